#!/usr/bin/env python
"""Static audit of the gemm_nta kernels' ISA (run by clipa_amd.build after compiling gemm_nta.hip with -save-temps).

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
Usage: python tools/audit_nta.py <file.s>   -> exit code 0 / 1, findings on stdout.
"""
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


def audit(path):
    txt = open(path).read().splitlines()
    problems = []
    kernel, in_asm = None, False
    agpr = re.compile(r"(?<![\w.])a\[?\d+")
    for ln, line in enumerate(txt, 1):
        m = re.match(r"^(_ZN\S*gemm_(?:[nt][nt]a|f8a)_kernel\S*):", line)
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
    problems += store_data_races(path, txt, r"^(_ZN\S*gemm_(?:[nt][nt]a|f8a)_kernel\S*):")
    # metadata
    meta = "\n".join(txt)
    for m in re.finditer(r"\.name:\s+(\S*gemm_(?:[nt][nt]a|f8a)_kernel\S*)\n(.*?)\.wavefront_size", meta, re.S):
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
    bad = audit(sys.argv[1])
    for b in bad[:40]:
        print(b)
    print(f"audit_nta: {len(bad)} finding(s)")
    sys.exit(1 if bad else 0)
