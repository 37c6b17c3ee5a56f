#!/usr/bin/env python
"""Generator of clipa_amd/csrc/gemm_nta_asm.inc: the hand-scheduled main loop of gemm_nta_kernel (gemm_nta.hip).

    python tools/gen_gemm_nta.py            # rewrites the .inc (tests/test_gemm_nta_gen_cpu.py checks it is up to date)

Why a generator: the K loop of the 4-wave / 512-register bf16 GEMM is ONE inline-asm statement per output tile (every
issue slot placed by hand: 128 MFMAs, 32 fragment reads, 16 LDS-DMA, 2 barriers and 3 waits per K step); writing ~1500
lines of it by hand is how schedules rot.  The schedule is data (`SCHEDULES`), the text is derived.

Structure of one tile (256 x 256 outputs, K = 64 * nkt, nkt even >= 4; wave (wm, wn) owns 128 x 128 = 8 x 8 blocks of 16 x 16):

  registers  a[0:255]     accumulators, block (bj, ai) at a[4 (8 bj + ai) : +3]   (bj: weight-row block, ai: token-row block)
             v[128:159]   A0 = token-row fragments of k-half 0 (8 x 4 registers)   v[160:191]  A1 (k-half 1)
             v[192:223]   B0 = weight-row fragments of k-half 0                    v[224:255]  B1
             v[96:103] / v[104:111]   per-lane byte offsets of the 8 A / 8 B LDS-DMA pieces of a K step
             v[112:115] / v[116:119]  fragment-read addresses: A (slot 0 k0, slot 0 k1, slot 1 k0, slot 1 k1) / B likewise
  LDS        two ring slots of 64 KiB (A image 256 rows x 128 B at +0, B image at +32 KiB), the production image and swizzle
  pipeline   K step i computes from registers: its k-half-0 fragments were read at the end of step i-1, its k-half-1
             fragments are read under the first MFMAs; barrier 1 (every wave is done reading slot i&1) frees the slot for
             the LDS-DMA of step i+2 (two steps ahead: ~1.5 K steps to land instead of < 1 in gemm_nt2); `s_waitcnt vmcnt`
             + barrier 2 publish step i+1's operands, whose k-half-0 fragments are read under the last MFMAs.
  tile edge  the last two steps fetch the NEXT tile's first two K steps (other descriptors, k offset 0), so the ring
             never drains; the tile's bias vector is fetched to registers under the last step.
  (tried)    keeping 12 of a lane's 32 packed output chunks in registers and storing them from inside the next tile's first
             two K steps, one store per 20 MFMAs, with exact in-order vmcnt accounting (commit 'gemm_nta: 12 of a lane's 32
             output chunks ...'): bit-identical and 3..5 % SLOWER on every shape - a store issued inside the K loop stalls
             the matrix pipe as long as one issued after it (profiles/r03_gemm_nta_deferred_stores_ab.jsonl).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "clipa_amd", "csrc", "gemm_nta_asm.inc")

A0, A1, B0, B1 = 128, 160, 192, 224          # fragment register bases
VOFF_A, VOFF_B = 96, 104                     # LDS-DMA per-lane offsets
VADDR_A, VADDR_B = 112, 116                  # + 2 * slot + khalf
SLOT, IMG, PIECE, BLOCK = 65536, 32768, 4096, 2048

# schedule = issue slots (index of the MFMA after which the instruction is placed; 0..127) of everything that is not an MFMA
SCHEDULES_ALL = {
    # two barriers per step; fragment reads and LDS-DMA one per two MFMAs (the pacing of the vendor's assembly kernel)
    0: dict(rd1_start=0, rd1_stride=2, bar1=34, dma_start=36, dma_stride=2, vmwait=74, rd0_start=76, rd0_stride=2, lgk_end=126),
    # denser front: reads one per MFMA, slot freed at MFMA 20, DMA issued by MFMA 54, operands published later
    1: dict(rd1_start=0, rd1_stride=1, bar1=20, dma_start=22, dma_stride=2, vmwait=88, rd0_start=90, rd0_stride=2, lgk_end=126),
    # DMA spread one per three MFMAs, publish as late as the fragment reads allow
    2: dict(rd1_start=0, rd1_stride=2, bar1=34, dma_start=36, dma_stride=3, vmwait=92, rd0_start=94, rd0_stride=2, lgk_end=126),
    # DMA one per four MFMAs (the issue spans half a K step), read-ahead packed into the last 24 MFMAs
    3: dict(rd1_start=0, rd1_stride=2, bar1=34, dma_start=36, dma_stride=4, vmwait=100, rd0_start=102, rd0_stride=1, lgk_end=126),
    # early slot release (reads one per MFMA) + DMA one per four MFMAs
    4: dict(rd1_start=0, rd1_stride=1, bar1=20, dma_start=22, dma_stride=4, vmwait=96, rd0_start=98, rd0_stride=1, lgk_end=126),
    # schedule 2 with the publish point pushed to MFMA 104
    5: dict(rd1_start=0, rd1_stride=2, bar1=34, dma_start=36, dma_stride=3, vmwait=104, rd0_start=106, rd0_stride=1, lgk_end=126),
    # early release, DMA one per six MFMAs: the issue spans the step, the publish wait leaves the DMA issued before it in flight
    6: dict(rd1_start=0, rd1_stride=1, bar1=20, dma_start=22, dma_stride=6, vmwait=96, rd0_start=98, rd0_stride=1, lgk_end=126),
    # early release, DMA one per five MFMAs, publish after the last one
    7: dict(rd1_start=0, rd1_stride=1, bar1=20, dma_start=22, dma_stride=5, vmwait=100, rd0_start=102, rd0_stride=1, lgk_end=126),
}


# the schedules compiled into the library (the rest were A/B'd on hardware: profiles/r03_gemm_nta_schedules_*.jsonl)
SCHEDULES = {k: SCHEDULES_ALL[k] for k in (0, 3, 4)}


def younger(S):
    """LDS-DMA of a step issued before its publish wait (the wait sits after MFMA `vmwait`, DMA q after MFMA dma_start + 1 + q stride)."""
    return sum(1 for q in range(16) if S["dma_start"] + 1 + q * S["dma_stride"] <= S["vmwait"])


def acc(bj, ai):
    b = 4 * (8 * bj + ai)
    return f"a[{b}:{b + 3}]"


def vr(base, i):
    return f"v[{base + 4 * i}:{base + 4 * i + 3}]"


def step_text(S, slot, srd, first, last, vmcnt, bias):
    """One K step.  slot: ring slot it computes from (and re-fills for step + 2).  srd: 'cur' | 'nxt' descriptors of the
    LDS-DMA it issues.  first: accumulators start from 0.  last: last step of the tile (no publish / read-ahead; the
    bias fetch rides here).  vmcnt: text of the immediate of the publish wait."""
    fill = {m: [] for m in range(128)}
    # k-half-1 fragments of THIS step (weights first: their registers were last used earliest in the previous step)
    order = [("B", b) for b in range(8)] + [("A", a) for a in range(8)]
    m = S["rd1_start"]
    for kind, b in order:
        base, addr = (B1, VADDR_B) if kind == "B" else (A1, VADDR_A)
        fill[m].append(f"ds_read_b128 {vr(base, b)}, v{addr + 2 * slot + 1} offset:{b * BLOCK}")
        m += S["rd1_stride"]
    assert m - S["rd1_stride"] < S["bar1"]
    fill[S["bar1"]] += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    # LDS-DMA of step + 2 into the slot just freed
    m = S["dma_start"]
    assert m > S["bar1"]
    if bias:
        # last step of the tile: the bias vector is fetched to registers here, OLDER than the step's 16 LDS-DMA, so that the
        # statement's final vmcnt(16) covers it.  (Fetching the first residual / pre-activation chunks of the epilogue here as
        # well was measured and changed nothing - profiles/r03_gemm_nta_aux_prefetch_ab.jsonl: those epilogues are bound by
        # the store path, not by the operand latency.)
        for p in range(4):
            for h in range(2):
                fill[m - 1 - (p * 2 + h)].append(f"buffer_load_dwordx4 %[bias{p * 2 + h}], %[vbias], %[srdBias], 0 offen offset:{p * 128 + h * 16}")
    for q in range(16):
        img, j = q // 8, q % 8
        voff = (VOFF_B if img else VOFF_A) + j
        s = "A" if img == 0 else "B"
        fill[m].append(f"s_add_u32 m0, %[ldsw], {slot * SLOT + img * IMG + j * PIECE}")
        fill[m + 1].append(f"buffer_load_dwordx4 v{voff}, %[{srd}{s}], %[sk] offen lds")
        m += S["dma_stride"]
    last_dma = m - S["dma_stride"] + 1
    assert last_dma + 1 < 127
    fill[last_dma + 1].append("s_add_u32 %[sk], %[sk], 128")
    if not last:
        # step + 1's operands are older than everything this step has issued so far: `younger(S)` LDS-DMA may stay in flight
        assert vmcnt in ("@VM0@", str(younger(S)))
        fill[S["vmwait"]] += [f"s_waitcnt vmcnt({vmcnt})", "s_barrier"]
        m = S["rd0_start"]
        for kind, b in order:
            base, addr = (B0, VADDR_B) if kind == "B" else (A0, VADDR_A)
            fill[m].append(f"ds_read_b128 {vr(base, b)}, v{addr + 2 * (slot ^ 1)} offset:{b * BLOCK}")
            m += S["rd0_stride"]
        assert m - S["rd0_stride"] < S["lgk_end"]
        fill[S["lgk_end"]].append("s_waitcnt lgkmcnt(0)")
    lines = []
    for m in range(128):
        kk, r = m // 64, m % 64
        bj, ai = r // 8, r % 8
        fb, fa = (B0, A0) if kk == 0 else (B1, A1)
        c = "0" if (first and kk == 0) else acc(bj, ai)
        lines.append(f"v_mfma_f32_16x16x32_bf16 {acc(bj, ai)}, {vr(fb, bj)}, {vr(fa, ai)}, {c}")
        lines += fill[m]
    return lines


def setup_text():
    """Per-lane addresses from the four "v" inputs (recomputed per statement: nothing of ours lives in registers the
    compiler may touch between statements except the accumulators)."""
    t = ["s_nop 4",
         f"v_mov_b32 v{VADDR_A}, %[va]", f"v_xor_b32 v{VADDR_A + 1}, 64, v{VADDR_A}",
         f"v_add_u32 v{VADDR_A + 2}, 0x10000, v{VADDR_A}", f"v_add_u32 v{VADDR_A + 3}, 0x10000, v{VADDR_A + 1}",
         f"v_mov_b32 v{VADDR_B}, %[vb]", f"v_xor_b32 v{VADDR_B + 1}, 64, v{VADDR_B}",
         f"v_add_u32 v{VADDR_B + 2}, 0x10000, v{VADDR_B}", f"v_add_u32 v{VADDR_B + 3}, 0x10000, v{VADDR_B + 1}",
         f"v_mov_b32 v{VOFF_A}, %[voffA]", f"v_mov_b32 v{VOFF_B}, %[voffB]"]
    for j in range(1, 8):
        t.append(f"v_add_u32 v{VOFF_A + j}, %[sA32], v{VOFF_A + j - 1}")
        t.append(f"v_add_u32 v{VOFF_B + j}, %[sB32], v{VOFF_B + j - 1}")
    return t


def prologue_text():
    """Very first tile of a workgroup: fetch its K steps 0 and 1 (32 LDS-DMA per wave)."""
    t = setup_text()
    for step in range(2):
        t.append(f"s_mov_b32 %[sk], {step * 128}")
        for q in range(16):
            img, j = q // 8, q % 8
            voff = (VOFF_B if img else VOFF_A) + j
            s = "A" if img == 0 else "B"
            t.append(f"s_add_u32 m0, %[ldsw], {step * SLOT + img * IMG + j * PIECE}")
            t.append("s_nop 0")
            t.append(f"buffer_load_dwordx4 v{voff}, %[cur{s}], %[sk] offen lds")
    return t


def tile_text(S):
    order = [("B", b) for b in range(8)] + [("A", a) for a in range(8)]
    t = setup_text()
    t.append("s_mov_b32 %[sk], 256")
    # this tile's step 0 has landed (younger: step 1's 16 LDS-DMA + whatever the epilogue before us left in flight)
    t += ["s_waitcnt vmcnt(@VMS@)", "s_barrier"]
    for kind, b in order:
        base, addr = (B0, VADDR_B) if kind == "B" else (A0, VADDR_A)
        t.append(f"ds_read_b128 {vr(base, b)}, v{addr} offset:{b * BLOCK}")
    t.append("s_waitcnt lgkmcnt(0)")
    t += step_text(S, 0, "cur", True, False, "@VM0@", False)
    k = str(younger(S))
    t += step_text(S, 1, "cur", False, False, k, False)
    t += ["s_cmp_eq_u32 %[nloop], 0", "s_cbranch_scc1 NTA_TAIL_%=", "s_mov_b32 %[cnt], %[nloop]", "NTA_LOOP_%=:"]
    t += step_text(S, 0, "cur", False, False, k, False)
    t += step_text(S, 1, "cur", False, False, k, False)
    t += ["s_sub_u32 %[cnt], %[cnt], 1", "s_cmp_lg_u32 %[cnt], 0", "s_cbranch_scc1 NTA_LOOP_%=", "NTA_TAIL_%=:",
          "s_mov_b32 %[sk], 0"]
    t += step_text(S, 0, "nxt", False, False, k, False)
    t += step_text(S, 1, "nxt", False, True, None, True)
    # the bias is older than the 16 LDS-DMA of the last step; MFMA results must be readable by v_accvgpr_read afterwards
    t += ["s_waitcnt vmcnt(16)", "s_nop 7", "s_nop 7"]
    return t


def c_string(lines, indent="  "):
    out = []
    for l in lines:
        if "@VM0@" in l or "@VMS@" in l:       # the immediates that differ per instantiation: macro arguments, stringified
            name = "VM0" if "@VM0@" in l else "VMS"
            a, b = l.split(f"@{name}@")
            out.append(f'{indent}"{a}" #{name} "{b}\\n\\t"')
        else:
            out.append(f'{indent}"{l}\\n\\t"')
    return "\n".join(out)


def clobbers():
    regs = [f"v{i}" for i in range(96, 256)] + [f"a{i}" for i in range(256)]
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
    p = ["// GENERATED by tools/gen_gemm_nta.py - do not edit (tests/test_gemm_nta_gen_cpu.py compares it with the generator).",
         "// Main loop of gemm_nta_kernel as inline-asm text; register map, pipeline and schedule: see the generator's docstring.",
         "#pragma once", "",
         "#define NTA_PROLOGUE_ASM \\"]
    p.append(" \\\n".join(c_string(prologue_text()).split("\n")))
    p.append("")
    for v, S in SCHEDULES.items():
        k = younger(S)
        p.append(f"// schedule {v}: {S}")
        p.append(f"// VMS: wait in front of the tile's first fragment reads (16 + stores the epilogue before it may leave in flight);")
        p.append(f"// VM0: publish wait of the tile's first step ({k} + those stores).  Both capped at the counter's 63.")
        p.append(f"#define NTA_TILE_ASM_{v}(S) NTA_TILE_ASM_{v}_I(NTA_VMS_##S, NTA_VM0_{v}_##S)")
        for st in (0, 32, 64):
            p.append(f"#define NTA_VM0_{v}_{st} {min(63, k + st)}")
        p.append(f"#define NTA_TILE_ASM_{v}_I(VMS, VM0) NTA_TILE_ASM_{v}_II(VMS, VM0)")
        p.append(f"#define NTA_TILE_ASM_{v}_II(VMS, VM0) \\")
        p.append(" \\\n".join(c_string(tile_text(S)).split("\n")))
        p.append("")
    for st in (0, 32, 64):
        p.append(f"#define NTA_VMS_{st} {min(63, 16 + st)}")
    p.append("")
    p.append("#define NTA_CLOBBERS \\")
    p.append(" \\\n".join(clobbers().split("\n")))
    p.append("")
    return "\n".join(p)


if __name__ == "__main__":
    text = render()
    if "--check" in sys.argv:
        sys.exit(0 if open(OUT).read() == text else 1)
    open(OUT, "w").write(text)
    print("wrote", OUT, len(text.splitlines()), "lines")
