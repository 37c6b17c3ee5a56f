#!/usr/bin/env python
"""Static audit of the gemm_nta kernels' ISA (run by clipa_amd.build after compiling gemm_nta.hip with -save-temps).

The accumulators of gemm_nta_kernel live in a[0:255] between the tile's inline-asm statement and the v_accvgpr_read
statements of the epilogue - something hipcc does not know.  That is safe exactly as long as the compiler itself never
touches an accumulation register and never spills (cdna_hip_programming.md 5.7 item 4), so for every gemm_nta kernel:
  * no scratch: .private_segment_fixed_size 0, .vgpr_spill_count 0, no scratch_ instructions (SGPR spills into VGPR lanes
    are tolerated: they never touch memory or the accumulation registers);
  * outside ;;#ASMSTART ... ;;#ASMEND no instruction names an AGPR (v_accvgpr_*, a[..] operands);
  * the kernel gets its 512 registers (accum_offset 256 / agpr_count 256).
Usage: python tools/audit_nta.py <file.s>   -> exit code 0 / 1, findings on stdout.
"""
import re
import sys


def audit(path):
    txt = open(path).read().splitlines()
    problems = []
    kernel, in_asm = None, False
    agpr = re.compile(r"(?<![\w.])a\[?\d+")
    for ln, line in enumerate(txt, 1):
        m = re.match(r"^(_ZN\S*gemm_[nt][nt]a_kernel\S*):", line)
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
    # metadata
    meta = "\n".join(txt)
    for m in re.finditer(r"\.name:\s+(\S*gemm_[nt][nt]a_kernel\S*)\n(.*?)\.wavefront_size", meta, re.S):
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
