#!/usr/bin/env python
"""Static audit of the generated-loop GEMM kernels' ISA (gemm_nta / gemm_tna / gemm_f8a / gemm_tn8), run by clipa_amd.build on the device
assembly hipcc emitted (-save-temps) - part of the package so that a copy of clipa_amd without tools/ still builds.

The accumulators of gemm_nta_kernel live in a[0:255] between the tile's inline-asm statement and the v_accvgpr_read
statements of the epilogue - something hipcc does not know.  That is safe exactly as long as the compiler itself never
touches an accumulation register and never spills (cdna_hip_programming.md 5.7 item 4), so for every gemm_nta kernel:
  * no scratch: .private_segment_fixed_size 0, .vgpr_spill_count 0, no scratch_ instructions (SGPR spills into VGPR lanes
    are tolerated: they never touch memory or the accumulation registers);
  * outside ;;#ASMSTART ... ;;#ASMEND no instruction names an AGPR (v_accvgpr_*, a[..] operands);
  * the kernel gets its 512 registers (accum_offset 256 / agpr_count 256);
  * no 12- or 16-byte store is followed IMMEDIATELY by a VALU write (inline asm included) of the third or fourth register of its
    data: on gfx950 that write wins the race against the store's operand read (observed in gemm_f8a: lanes 12-15 of every 16
    stored the next chunk's values); the first two registers, or one instruction of distance, are safe (every bit-exact test
    of gemm_nta exercises them).
  * store counts: the tile statement waits with `s_waitcnt vmcnt(16 + S)`, S = the 16-byte stores the epilogue in front of it
    left in flight (32 per wave and output; NTA_TILE / F8A_TILE macro argument).  If hipcc ever emitted FEWER stores than S
    (merged / dropped), the wait would let the tile's first operands be read before they landed; MORE (split stores) is safe
    but slow.  Every copy of the epilogue (one per activation) is bracketed by `; CLIPA_EPI_BEGIN k` / `; CLIPA_EPI_END k`
    comment markers and must hold exactly S `buffer_store_dwordx4` in one straight-line block.
Usage: python -m clipa_amd.isa_audit <file.s>   -> exit code 0 / 1, findings on stdout.
"""
import os
import re
import sys


def store_data_races(path, txt, label=r"^(\w[\w.$]*):"):
    """[12- / 16-byte store] directly followed by a VALU write (inline asm included) of its third or fourth data register, in
    every function whose label matches `label` (default: every function of the file)."""
    def regs(tok):
        m = re.match(r"v\[(\d+):(\d+)\]", tok)
        if m:
            return list(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.match(r"v(\d+)$", tok)
        return [int(m.group(1))] if m else []
    problems, kernel, prev = [], None, None
    for ln, line in enumerate(txt, 1):
        m = re.match(label, line)
        if m:
            kernel, prev = m.group(1), None
            continue
        if line.startswith(".Lfunc_end"):
            kernel = None
        s = line.strip()
        if kernel is None or not s or s[0] in ";." or s.endswith(":"):
            continue
        code = s.split(";")[0].strip()
        if prev and code.startswith("v_") and not code.startswith("v_cmp"):
            dst = regs(code.split()[1].rstrip(","))
            if any(d in prev[1][2:] for d in dst):
                problems.append(f"{path}:{ln}: {kernel}: `{code}` overwrites the tail of the data of `{prev[0]}` one slot after it")
        prev = None
        if re.match(r"(buffer|global|flat)_store_dwordx[34]", code):
            prev = (code, regs(code.split()[1].rstrip(",")))
    return problems


def epilogue_store_counts(path, txt):
    """gemm_nta_kernel<EPI, PRE, .> / gemm_f8a_kernel<EPI, PRE, .>: every epilogue copy (between its `; CLIPA_EPI_BEGIN k` and
    `; CLIPA_EPI_END k` markers, one straight-line block) holds exactly the 16-byte stores the tile statement's wait assumes:
    32 per output (the e4m3 pre-activation copy of gemm_nta<., PRE = 2> leaves as 32 8-byte stores).  hipcc's structurizer makes a path walk over the basic blocks useless (correlated flow predicates), and
    totals per kernel say nothing (it tail-merges identical stores of the three activation copies) - hence the markers."""
    problems, kernel, open_k, count, count8, copies = [], None, None, 0, 0, 0

    def want(kname):
        """-> (16-byte stores, 8-byte stores) per epilogue copy.  gemm_nta_kernel<EPI, PRE (0 / 1 bf16 copy / 2 e4m3 copy / 3 bf16 activation output of the
        GELU-backward epilogue), AUX8, SCHED>, gemm_f8a_kernel<EPI, PRE (0 / 1 bf16 copy / 2 e4m3 copy), AUX8, OUTQ (e4m3 output: 8-byte stores), FMT>."""
        m = re.search(r"gemm_nta_kernelILi(\d+)ELi(\d+)E", kname)
        if m:
            pre = int(m.group(2))
            return 32 * (2 if pre in (1, 3) else 1), 32 if pre == 2 else 0
        m = re.search(r"gemm_f8a_kernelILi(\d+)ELi(\d+)ELb([01])ELi(\d+)E", kname)
        pre, outq = (int(m.group(2)), int(m.group(4))) if m else (0, 0)
        return 32 * ((pre == 1) + (outq == 0)), 32 * ((pre in (2, 3)) + (outq != 0))

    def close_kernel():
        if kernel is not None and copies == 0:
            problems.append(f"{path}: {kernel}: no CLIPA_EPI_BEGIN / END markers found (epilogue store count unchecked)")
        if kernel is not None and open_k is not None:
            problems.append(f"{path}: {kernel}: CLIPA_EPI_BEGIN {open_k} without its END")

    for ln, line in enumerate(txt, 1):
        m = re.match(r"^(_ZN\S*gemm_(?:nta|f8a)_kernel\S*):", line)
        if m:
            close_kernel()
            kernel, open_k, count, count8, copies = m.group(1), None, 0, 0, 0
            continue
        if line.startswith(".Lfunc_end"):
            close_kernel()
            kernel = None
        if kernel is None:
            continue
        s = line.strip()
        m = re.match(r";\s*CLIPA_EPI_(BEGIN|END)\s+(\d+)", s)
        if m and m.group(1) == "BEGIN":
            if open_k is not None:
                problems.append(f"{path}:{ln}: {kernel}: nested CLIPA_EPI_BEGIN")
            open_k, count, count8 = m.group(2), 0, 0
        elif m:
            if open_k != m.group(2):
                problems.append(f"{path}:{ln}: {kernel}: CLIPA_EPI_END {m.group(2)} does not close BEGIN {open_k}")
            elif (count, count8) != want(kernel):
                problems.append(f"{path}:{ln}: {kernel}: epilogue copy {open_k} issues {count} buffer_store_dwordx4 + {count8} "
                                f"buffer_store_dwordx2, the tile statement's vmcnt assumes {want(kernel)}")
            open_k, copies = None, copies + 1
        elif open_k is not None:
            code = s.split(";")[0].strip()
            if re.match(r"buffer_store_dwordx4\b", code):
                count += 1
            elif re.match(r"buffer_store_dwordx2\b", code):
                count8 += 1
            elif re.match(r"(buffer|global|flat)_store_", code):
                problems.append(f"{path}:{ln}: {kernel}: store of another width inside the epilogue: `{code}`")
            elif re.match(r"\.LBB\w+:", s) or code.startswith("s_cbranch") or code.startswith("s_branch") or code.startswith("s_setpc"):
                problems.append(f"{path}:{ln}: {kernel}: control flow inside an epilogue copy: `{s}`")
    close_kernel()
    return problems


def audit(path):
    txt = open(path).read().splitlines()
    problems = []
    kernel, in_asm = None, False
    agpr = re.compile(r"(?<![\w.])a\[?\d+")
    for ln, line in enumerate(txt, 1):
        m = re.match(r"^(_ZN\S*gemm_(?:[nt][nt]a|f8a|tn8)_kernel\S*):", line)
        if m:
            kernel = m.group(1)
            in_asm = False
            continue
        if line.startswith(".Lfunc_end") or line.strip().startswith(".end_amdhsa_kernel"):
            kernel = None
        if kernel is None:
            continue
        s = line.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if in_asm or not s or s.startswith(";") or s.startswith("."):
            continue
        code = s.split(";")[0]
        if "scratch_" in code or "s_scratch" in code:
            problems.append(f"{path}:{ln}: scratch access in {kernel}: {code}")
        if agpr.search(code) and not code.startswith("v_accvgpr_read_b32") :
            problems.append(f"{path}:{ln}: compiler-generated AGPR use in {kernel}: {code}")
        if code.startswith("v_accvgpr_write"):
            problems.append(f"{path}:{ln}: compiler-generated v_accvgpr_write in {kernel}: {code}")
    problems += store_data_races(path, txt, r"^(_ZN\S*gemm_(?:[nt][nt]a|f8a|tn8)_kernel\S*):")
    problems += epilogue_store_counts(path, txt)
    # metadata
    meta = "\n".join(txt)
    for m in re.finditer(r"\.name:\s+(\S*gemm_(?:[nt][nt]a|f8a|tn8)_kernel\S*)\n(.*?)\.wavefront_size", meta, re.S):
        name, body = m.group(1), m.group(2)
        for key in ("private_segment_fixed_size", "vgpr_spill_count"):
            v = re.search(rf"\.{key}:\s+(\d+)", body)
            if v and int(v.group(1)) != 0:
                problems.append(f"{name}: .{key} = {v.group(1)} (must be 0)")
        v = re.search(r"\.vgpr_count:\s+(\d+)", body)
        if v and int(v.group(1)) != 512:
            problems.append(f"{name}: .vgpr_count = {v.group(1)} (expected 512)")
    return problems


if __name__ == "__main__":
    if not os.path.isfile(sys.argv[1]):
        print(f"isa_audit: assembly file {sys.argv[1]} not found (hipcc -save-temps naming changed?)")
        sys.exit(2)
    bad = audit(sys.argv[1])
    for b in bad[:40]:
        print(b)
    print(f"isa_audit: {len(bad)} finding(s)")
    sys.exit(1 if bad else 0)
