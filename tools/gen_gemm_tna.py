#!/usr/bin/env python
"""Generator of clipa_amd/csrc/gemm_tna_asm.inc: the hand-scheduled main loop of gemm_tna_kernel (gemm_tna.hip), the
weight-gradient GEMM  O[R,C] = sum_m P[m,R] Q[m,C]  on four waves of 512 registers.

    python tools/gen_gemm_tna.py            # rewrites the .inc (tests/test_gemm_nta_gen_cpu.py checks it is up to date)

Same pipeline as gen_gemm_nta.py (all fragments of a K step in registers, LDS slot re-filled two steps ahead, two barriers
per step); what differs is the operand geometry: the reduction index m is the SLOW axis of both operands, so the LDS images are
[64 m][256 columns] (512-byte rows, the gemm_tn3 image and swizzle) and MFMA fragments come from `ds_read_b64_tr_b16`
(hardware transpose: two reads per 16 x 32 fragment).

  registers  a[0:255]     accumulators, block (ri, ci) at a[4 (8 ri + ci) : +3]
             v[128:159]   P0 = fragments of the 8 P column blocks, k-half 0 (m rows 0..31 of the step)   v[160:191]  P1
             v[192:223]   Q0                                                                              v[224:255]  Q1
             v[96:103] / v[104:111]    per-lane byte offsets of the 8 P / 8 Q LDS-DMA pieces of a step (2 rows of 512 B each)
             v[112:119] / v[120:127]   fragment-read address of block b of P / Q in the CURRENT slot (flipped by XOR 0x10000
                                       once per step, between the reads of this step's k-half 1 and the next step's k-half 0):
                                       the image's chunk swizzle XORs the same address bits as the block index, so a block
                                       cannot be an instruction offset
             v[92:95]     a fragment of ones; COLSUM variant: sum_m P[m, r] = P^T . 1 is 16 more MFMAs per step for the
                          workgroups that own the bias gradient (accumulated in the output operands cs0..cs7), instead of VALU adds
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_gemm_nta import SLOT, IMG, PIECE, c_string, younger  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "clipa_amd", "csrc", "gemm_tna_asm.inc")

P0, P1, Q0, Q1 = 128, 160, 192, 224
VOFF_P, VOFF_Q = 96, 104
VADDR_P, VADDR_Q = 112, 120
ONES = 92
KHALF, HALF = 16384, 2048          # bytes: 32 image rows, 4 image rows

SCHEDULES = {
    # reads one per MFMA (32 per phase), slot freed at MFMA 36, DMA one per four MFMAs, publish leaves the DMA issued so far in flight
    0: dict(rd1_start=0, rd1_stride=1, bar1=36, dma_start=38, dma_stride=4, vmwait=92, rd0_start=94, rd0_stride=1, lgk_end=127, flip_start=40),
    # two reads per MFMA slot at the front, slot freed at MFMA 20, DMA one per five MFMAs
    1: dict(rd1_start=0, rd1_stride=0.5, bar1=20, dma_start=22, dma_stride=5, vmwait=100, rd0_start=102, rd0_stride=0.75, lgk_end=127, flip_start=24),
}


def acc(ri, ci):
    b = 4 * (8 * ri + ci)
    return f"a[{b}:{b + 3}]"


def vr(base, i):
    return f"v[{base + 4 * i}:{base + 4 * i + 3}]"


def frag_reads(base, addr, kk):
    """The 16 transposing reads of the 8 fragments (b = 0..7) of one operand's k-half kk: (text, destination base register)."""
    out = []
    for b in range(8):
        for half in range(2):
            r = base + 4 * b + 2 * half
            out.append(f"ds_read_b64_tr_b16 v[{r}:{r + 1}], v{addr + b} offset:{kk * KHALF + half * HALF}")
    return out


def step_text(S, slot, srd, first, last, vmcnt, colsum):
    fill = {m: [] for m in range(128)}
    # k-half-1 fragments of THIS step: P first (the outer operand of the MFMA order: its registers were last used earliest)
    reads = frag_reads(P1, VADDR_P, 1) + frag_reads(Q1, VADDR_Q, 1)
    for i, r in enumerate(reads):
        m = int(S["rd1_start"] + i * S["rd1_stride"])
        assert m < S["bar1"]
        fill[m].append(r)
    fill[S["bar1"]] += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    m = S["dma_start"]
    assert m > S["bar1"]
    for q in range(16):
        img, j = q // 8, q % 8
        voff = (VOFF_Q if img else VOFF_P) + j
        s = "P" if img == 0 else "Q"
        fill[m].append(f"s_add_u32 m0, %[ldsw], {slot * SLOT + img * IMG + j * PIECE}")
        fill[m + 1].append(f"buffer_load_dwordx4 v{voff}, %[{srd}{s}], %[sk{s}] offen lds")
        m += S["dma_stride"]
    last_dma = m - S["dma_stride"] + 1
    assert last_dma + 2 < 127
    fill[last_dma + 1].append("s_add_u32 %[skP], %[skP], %[sP64]")
    fill[last_dma + 2].append("s_add_u32 %[skQ], %[skQ], %[sQ64]")
    # flip the fragment-read addresses to the other slot (after this step's reads, before the read-ahead)
    for i in range(16):
        m = S["flip_start"] + 2 * i + 1
        assert S["bar1"] < m < S["vmwait"]
        reg = (VADDR_P + i) if i < 8 else (VADDR_Q + i - 8)
        fill[m].append(f"v_xor_b32 v{reg}, 0x10000, v{reg}")
    if not last:
        assert vmcnt in ("@VM0@", str(younger(S)))
        fill[S["vmwait"]] += [f"s_waitcnt vmcnt({vmcnt})", "s_barrier"]
        reads = frag_reads(P0, VADDR_P, 0) + frag_reads(Q0, VADDR_Q, 0)
        for i, r in enumerate(reads):
            m = int(S["rd0_start"] + i * S["rd0_stride"])
            assert S["vmwait"] < m < S["lgk_end"]
            fill[m].append(r)
        fill[S["lgk_end"]].append("s_waitcnt lgkmcnt(0)")
    lines = []
    for m in range(128):
        kk, r = m // 64, m % 64
        ri, ci = r // 8, r % 8
        fp, fq = (P0, Q0) if kk == 0 else (P1, Q1)
        c = "0" if (first and kk == 0) else acc(ri, ci)
        lines.append(f"v_mfma_f32_16x16x32_bf16 {acc(ri, ci)}, {vr(fp, ri)}, {vr(fq, ci)}, {c}")
        if colsum and ci == 7:       # column sums of P block ri: P^T . ones, accumulated in the asm OUTPUT operands cs0..cs7
            cs = f"%[cs{ri}]"
            lines.append(f"v_mfma_f32_16x16x32_bf16 {cs}, {vr(fp, ri)}, v[{ONES}:{ONES + 3}], {'0' if (first and kk == 0) else cs}")
        lines += fill[m]
    return lines


def setup_text():
    t = ["s_nop 4"]
    for b in range(8):
        t.append(f"v_xor_b32 v{VADDR_P + b}, {b << 5}, %[vP]")
        t.append(f"v_xor_b32 v{VADDR_Q + b}, {b << 5}, %[vQ]")
    t += [f"v_mov_b32 v{VOFF_P}, %[vPe]", f"v_mov_b32 v{VOFF_P + 1}, %[vPo]", f"v_mov_b32 v{VOFF_Q}, %[vQe]", f"v_mov_b32 v{VOFF_Q + 1}, %[vQo]"]
    for j in range(2, 8):
        t.append(f"v_add_u32 v{VOFF_P + j}, %[sP16], v{VOFF_P + j - 2}")
        t.append(f"v_add_u32 v{VOFF_Q + j}, %[sQ16], v{VOFF_Q + j - 2}")
    for i in range(4):
        t.append(f"v_mov_b32 v{ONES + i}, 0x3f803f80")
    return t


def tile_text(S, colsum):
    t = setup_text()
    # the workgroup's K steps 0 and 1
    t += ["s_mov_b32 %[skP], 0", "s_mov_b32 %[skQ], 0"]
    for step in range(2):
        for q in range(16):
            img, j = q // 8, q % 8
            voff = (VOFF_Q if img else VOFF_P) + j
            s = "P" if img == 0 else "Q"
            t.append(f"s_add_u32 m0, %[ldsw], {step * SLOT + img * IMG + j * PIECE}")
            t.append("s_nop 0")
            t.append(f"buffer_load_dwordx4 v{voff}, %[cur{s}], %[sk{s}] offen lds")
        t += ["s_add_u32 %[skP], %[skP], %[sP64]", "s_add_u32 %[skQ], %[skQ], %[sQ64]"]
    t += ["s_waitcnt vmcnt(16)", "s_barrier"]
    t += frag_reads(P0, VADDR_P, 0) + frag_reads(Q0, VADDR_Q, 0)
    t.append("s_waitcnt lgkmcnt(0)")
    k = str(younger(S))
    t += step_text(S, 0, "cur", True, False, k, colsum)
    t += step_text(S, 1, "cur", False, False, k, colsum)
    t += ["s_cmp_eq_u32 %[nloop], 0", "s_cbranch_scc1 TNA_TAIL_%=", "s_mov_b32 %[cnt], %[nloop]", "TNA_LOOP_%=:"]
    t += step_text(S, 0, "cur", False, False, k, colsum)
    t += step_text(S, 1, "cur", False, False, k, colsum)
    t += ["s_sub_u32 %[cnt], %[cnt], 1", "s_cmp_lg_u32 %[cnt], 0", "s_cbranch_scc1 TNA_LOOP_%=", "TNA_TAIL_%=:"]
    # the last two steps have nothing left to fetch: their LDS-DMA go through descriptors of 0 bytes (zeros land, counts stay uniform)
    t += step_text(S, 0, "nul", False, False, k, colsum)
    t += step_text(S, 1, "nul", False, True, None, colsum)
    t += ["s_waitcnt vmcnt(0)", "s_nop 7", "s_nop 7"]
    return t


def clobbers():
    regs = [f"v{i}" for i in range(ONES, 256)] + [f"a{i}" for i in range(256)]
    out, line = [], "  "
    for r in regs:
        tok = f'"{r}", '
        if len(line) + len(tok) > 124:
            out.append(line.rstrip())
            line = "  "
        line += tok
    out.append(line.rstrip().rstrip(","))
    return "\n".join(out)


def render():
    p = ["// GENERATED by tools/gen_gemm_tna.py - do not edit (tests/test_gemm_nta_gen_cpu.py compares it with the generator).",
         "// Main loop of gemm_tna_kernel as inline-asm text; register map, pipeline and schedule: see the generator's docstring.",
         "#pragma once", ""]
    for v, S in SCHEDULES.items():
        for cs in (0, 1):
            p.append(f"// schedule {v}{' + column sums' if cs else ''}: {S}")
            p.append(f"#define TNA_ASM_{v}_{cs} \\")
            p.append(" \\\n".join(c_string(tile_text(S, bool(cs))).split("\n")))
            p.append("")
    p.append("#define TNA_CLOBBERS \\")
    p.append(" \\\n".join(clobbers().split("\n")))
    p.append("")
    return "\n".join(p)


if __name__ == "__main__":
    text = render()
    if "--check" in sys.argv:
        sys.exit(0 if open(OUT).read() == text else 1)
    open(OUT, "w").write(text)
    print("wrote", OUT, len(text.splitlines()), "lines")
