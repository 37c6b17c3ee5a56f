#!/usr/bin/env python
"""Generator of clipa_amd/csrc/gemm_f8a_asm.inc: the hand-scheduled main loop of gemm_f8a_kernel (gemm_f8a.hip), the fp8
NT GEMM on four waves of 512 registers.

    python tools/gen_gemm_f8a.py            # rewrites the .inc (tests/test_gemm_nta_gen_cpu.py checks it is up to date)

Same LDS images, swizzle, ring, LDS-DMA and tile-edge handling as gen_gemm_nta.py (an operand image is 256 rows x 128
BYTES = 128 fp8 k-values), but one `v_mfma_f32_16x16x128_f8f6f4` consumes a whole image row of a block: a lane's operand
is chunks g and 4 + g of its row = the bf16 kernel's two k-halves side by side in 8 registers, so a K step is 64 MFMAs and
there is no "other k-half" to read under the first half of them.  The read-ahead follows the registers instead:

  registers  a[0:255]     accumulators, block (bj, ai) at a[4 (8 bj + ai) : +3]
             v[128:191]   A block ai (token rows) at v[128 + 8 ai : +7]  (registers 0-3 = chunk g, 4-7 = chunk 4 + g)
             v[192:255]   B block bj (weight rows) at v[192 + 8 bj : +7]
             v[96:119]    LDS-DMA offsets and fragment-read addresses as in gen_gemm_nta.py
  MFMA order bj-major (m = 8 bj + ai): weight block bj is dead after MFMA 8 bj + 7, token block ai after MFMA 56 + ai, and
             the next step needs them in the same order (B0 + A0 at its MFMA 0, A1 at 1, ... B1 at 8, ...).  So a step
               - reads ITS OWN last weight block B7 at the top (needed at MFMA 56), waits for token block ai of the read-ahead
                 in front of MFMA ai (counted lgkmcnt: LDS returns in order),
               - after MFMA 8: every fragment of this step is in registers -> barrier 1 frees the slot, the 16 LDS-DMA of
                 step + 2 go out one per two MFMAs (MFMAs 9..40),
               - after MFMA 42: `s_waitcnt vmcnt(16)` + barrier 2 publish step + 1 (issued a step ago), whose weight blocks
                 0..4 are read at once, 5 and 6 as they die (MFMAs 48, 56), token block ai right behind MFMA 56 + ai - eight
                 MFMAs (256 clocks) ahead of its first use.
  tile edge  as in gen_gemm_nta.py; the tile's vectors (row scales | bias | column scales, 1 KiB each) ride the last step as one
             LDS-DMA per wave 0..2 into a 3 KiB window behind the ring, older than the step's 16 pieces.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_gemm_nta import SLOT, IMG, PIECE, BLOCK, VOFF_A, VOFF_B, VADDR_A, VADDR_B, c_string, clobbers, setup_text, prologue_text  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "clipa_amd", "csrc", "gemm_f8a_asm.inc")

FA, FB = 128, 192
BAR1, DMA0, PUBLISH = 8, 9, 42


def acc(bj, ai):
    b = 4 * (8 * bj + ai)
    return f"a[{b}:{b + 3}]"


def blk(base, i):
    return f"v[{base + 8 * i}:{base + 8 * i + 7}]"


def reads(kind, b, slot):
    """Both 16-byte reads of block b (chunk g into registers 0-3, chunk 4 + g into 4-7) from ring slot `slot`."""
    base, addr = (FB, VADDR_B) if kind == "B" else (FA, VADDR_A)
    r = base + 8 * b
    return [f"ds_read_b128 v[{r}:{r + 3}], v{addr + 2 * slot} offset:{b * BLOCK}",
            f"ds_read_b128 v[{r + 4}:{r + 7}], v{addr + 2 * slot + 1} offset:{b * BLOCK}"]


def step_text(slot, srd, first, last, vmcnt):
    """One K step computing from registers; slot = ring slot of its data (re-filled for step + 2); see the module docstring."""
    pre = {m: [] for m in range(64)}        # in front of MFMA m
    fill = {m: [] for m in range(64)}       # behind MFMA m
    pre[0] += reads("B", 7, slot)
    for m in range(8):
        pre[m].append(f"s_waitcnt lgkmcnt({min(15, 16 - 2 * m)})")
    fill[BAR1] += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    if last:
        # the tile's vectors: one 1 KiB LDS-DMA per wave (a 0-byte descriptor for wave 3), OLDER than the step's 16 pieces
        fill[BAR1 - 2].append("s_mov_b32 m0, %[ldsv]")
        fill[BAR1 - 1].append("buffer_load_dwordx4 %[vvec], %[srdVec], 0 offen lds")
    for q in range(16):
        img, j = q // 8, q % 8
        voff = (VOFF_B if img else VOFF_A) + j
        s = "A" if img == 0 else "B"
        fill[DMA0 + 2 * q].append(f"s_add_u32 m0, %[ldsw], {slot * SLOT + img * IMG + j * PIECE}")
        fill[DMA0 + 2 * q + 1].append(f"buffer_load_dwordx4 v{voff}, %[{srd}{s}], %[sk] offen lds")
    assert DMA0 + 2 * 15 + 1 < PUBLISH - 1
    fill[PUBLISH - 1].append("s_add_u32 %[sk], %[sk], 128")
    if not last:
        fill[PUBLISH] += [f"s_waitcnt vmcnt({vmcnt})", "s_barrier"]
        for bj in range(5):
            fill[PUBLISH + 1 + bj] += reads("B", bj, slot ^ 1)
        fill[48] += reads("B", 5, slot ^ 1)
        fill[56] += reads("B", 6, slot ^ 1)
        for ai in range(8):
            fill[56 + ai] += reads("A", ai, slot ^ 1)
    lines = []
    for m in range(64):
        bj, ai = m // 8, m % 8
        lines += pre[m]
        c = "0" if first else acc(bj, ai)
        lines.append(f"v_mfma_f32_16x16x128_f8f6f4 {acc(bj, ai)}, {blk(FB, bj)}, {blk(FA, ai)}, {c}@FMT@")
        lines += fill[m]
    return lines


def tile_text():
    t = setup_text()
    t.append("s_mov_b32 %[sk], 256")
    t += ["s_waitcnt vmcnt(@VMS@)", "s_barrier"]
    for bj in range(7):
        t += reads("B", bj, 0)
    for ai in range(8):
        t += reads("A", ai, 0)
    t.append("s_waitcnt lgkmcnt(0)")
    t += step_text(0, "cur", True, False, "@VM0@")
    t += step_text(1, "cur", False, False, "16")
    t += ["s_cmp_eq_u32 %[nloop], 0", "s_cbranch_scc1 F8A_TAIL_%=", "s_mov_b32 %[cnt], %[nloop]", "F8A_LOOP_%=:"]
    t += step_text(0, "cur", False, False, "16")
    t += step_text(1, "cur", False, False, "16")
    t += ["s_sub_u32 %[cnt], %[cnt], 1", "s_cmp_lg_u32 %[cnt], 0", "s_cbranch_scc1 F8A_LOOP_%=", "F8A_TAIL_%=:",
          "s_mov_b32 %[sk], 0"]
    t += step_text(0, "nxt", False, False, "16")
    t += step_text(1, "nxt", False, True, None)
    # the vectors are older than the 16 LDS-DMA of the last step; the barrier makes every wave's vector visible to the epilogue;
    # MFMA results must be readable by v_accvgpr_read afterwards
    t += ["s_waitcnt vmcnt(16)", "s_barrier", "s_nop 7", "s_nop 7"]
    return t


def c_string3(lines, indent="  "):
    """c_string with a third stringified macro argument: the format suffix of the MFMAs."""
    out = []
    for l in lines:
        if "@FMT@" in l:
            a, b = l.split("@FMT@")
            out.append(f'{indent}"{a}" FMT "{b}\\n\\t"')
        else:
            out.append(c_string([l], indent))
    return "\n".join(out)


def render():
    p = ["// GENERATED by tools/gen_gemm_f8a.py - do not edit (tests/test_gemm_nta_gen_cpu.py compares it with the generator).",
         "// Main loop of gemm_f8a_kernel as inline-asm text; register map, pipeline and schedule: see the generator's docstring.",
         "#pragma once", "",
         "#define F8A_PROLOGUE_ASM \\"]
    p.append(" \\\n".join(c_string(prologue_text()).split("\n")))
    p.append("")
    p.append("// VMS: wait in front of the tile's first fragment reads (16 + stores the epilogue before it may leave in flight);")
    p.append("// VM0: publish wait of the tile's first step (likewise).  FMT: a string, the format suffix of the MFMAs (\" cbsz:1\" ...).")
    p.append("#define F8A_TILE_ASM(S, FMT) F8A_TILE_ASM_I(F8A_VM_##S, F8A_VM_##S, FMT)")
    p.append("#define F8A_TILE_ASM_I(VMS, VM0, FMT) F8A_TILE_ASM_II(VMS, VM0, FMT)")
    p.append("#define F8A_TILE_ASM_II(VMS, VM0, FMT) \\")
    p.append(" \\\n".join(c_string3(tile_text()).split("\n")))
    p.append("")
    for st in (0, 32, 64):
        p.append(f"#define F8A_VM_{st} {min(63, 16 + st)}")
    p.append("")
    p.append("#define F8A_CLOBBERS \\")
    p.append(" \\\n".join(clobbers().split("\n")))
    p.append("")
    return "\n".join(p)


if __name__ == "__main__":
    text = render()
    if "--check" in sys.argv:
        sys.exit(0 if open(OUT).read() == text else 1)
    open(OUT, "w").write(text)
    print("wrote", OUT, len(text.splitlines()), "lines")
