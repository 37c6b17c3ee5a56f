#!/usr/bin/env python
"""Generator of clipa_amd/csrc/gemm_tn8_asm.inc: the hand-scheduled main loop of gemm_tn8_kernel (gemm_tn8.hip), the fp8
weight-gradient GEMM  O[R,C] = sum_m P8[m,R] Q8[m,C]  on four waves of 512 registers.

    python tools/gen_gemm_tn8.py            # rewrites the .inc (tests/test_gemm_nta_gen_cpu.py checks it is up to date)

The reduction index m is the SLOW axis of both operands (as in gen_gemm_tna.py), the operands are fp8 bytes (as in
gen_gemm_f8a.py): a K step is 128 m-rows, an operand image [128 m][256 columns] of bytes (256-byte rows, 32 KiB - the ring, the
LDS-DMA piece size and the bytes per step are those of the other two kernels), and one `v_mfma_f32_16x16x128_f8f6f4` consumes a
whole step of a block pair.  Fragments come from `ds_read_b64_tr_b8` (hardware transpose of an [8 m][16 columns] byte block per
16 lanes; lanes 2j, 2j+1 of the 16 supply the two 8-byte halves of m-row j, lane c receives column c of the 8 rows - measured with
tools/probes/tr_b8/): four reads per block and step, read i of lane group g = rows 32 i + 8 g .. + 7 of the step, for both
operands alike (which m a register byte holds is immaterial as long as P and Q agree).

  LDS image   row m of the step at m * 256; its 16-byte chunk c (columns 16 c .. 16 c + 15 of the tile) at chunk position
              c ^ (m & 15): the 32 lanes of a half-wave read 16 different rows -> 16 different positions -> all 64 banks once.
              The swizzle XORs the bits of the block index, so a block is an address register (flipped to the other ring slot by
              XOR 0x10000 once per step), the read index i an instruction offset (i * 8192).
  registers   a[0:255]     accumulators, block (ri, ci) at a[4 (8 ri + ci) : +3]
              v[128:191]   P block ri at v[128 + 8 ri : +7] (read i in registers 2 i, 2 i + 1)     v[192:255]  Q block ci likewise
              v[96:103] / v[104:111]    per-lane byte offsets of the 8 P / 8 Q LDS-DMA pieces of a step (4 rows of 256 B per wave)
              v[112:119] / v[120:127]   fragment-read address of block b of P / Q in the ring slot the NEXT reads go to
  MFMA order  rows 0 .. 7 - T of P blocks row-major (m = 8 ri + ci), the last T rows column-major (ri fastest): P block ri < 8 - T
              is dead after its row, Q block ci after the T MFMAs of its column in the tail, the last T P blocks with the step.
              The next step's fragments follow the registers (gen_gemm_f8a.py): early P blocks right behind the publish
              barrier, Q block ci behind its last MFMA, the late P blocks at the top of the next step (they are needed by its
              MFMA 8 (8 - T)).  T = 1 is gen_gemm_f8a.py's order: 32 + 4 transposing reads in the last 8 MFMA slots, an LDS burst
              of ~512 clocks against 256 of matrix work; T = 2 halves the rate, T = 4 spreads the reads evenly at the price of
              a later slot release (the default: +1-5 % on 14 of 16 production shapes).  Waits are counted (`lgkmcnt`: LDS returns in order; the generator walks the
              issue order of two consecutive steps and derives every immediate).
  pipeline    as gen_gemm_f8a.py: barrier 1 (every wave has read the step's last fragments) frees the ring slot for the LDS-DMA
              of step + 2, `s_waitcnt vmcnt` + barrier 2 publish step + 1.  One tile per workgroup (split-M slices, no tile edge):
              the last two steps issue their LDS-DMA through 0-byte descriptors (zeros land, counts stay uniform).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_gemm_nta import SLOT, IMG, PIECE, c_string  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "clipa_amd", "csrc", "gemm_tn8_asm.inc")

FP, FQ = 128, 192
VOFF_P, VOFF_Q = 96, 104
VADDR_P, VADDR_Q = 112, 120
RD_I = 8192                         # bytes between the 32-row groups of a step (instruction offset of read i)

# schedule = tail rows T + issue slots (index of the MFMA behind which the instruction is placed)
SCHEDULES = {
    # gen_gemm_f8a.py's order and slots
    0: dict(T=1, top_per_slot=4, bar1=8, dma_start=9, dma_stride=2, publish=42),
    # two column-major tail rows: Q reads 4 per two MFMAs over the last 16 slots
    1: dict(T=2, top_per_slot=8, bar1=8, dma_start=9, dma_stride=2, publish=42),
    # four tail rows: one read per MFMA slot throughout; the slot is freed at MFMA 10, the publish wait leaves the LDS-DMA issued so far in flight
    2: dict(T=4, top_per_slot=2, bar1=10, dma_start=11, dma_stride=2, publish=33),
}


def acc(ri, ci):
    b = 4 * (8 * ri + ci)
    return f"a[{b}:{b + 3}]"


def blk(base, i):
    return f"v[{base + 8 * i}:{base + 8 * i + 7}]"


def mfma_order(T):
    """[(ri, ci)] of the 64 MFMAs of a step."""
    H = 8 - T
    head = [(ri, ci) for ri in range(H) for ci in range(8)]
    tail = [(H + t, ci) for ci in range(8) for t in range(T)]
    return head + tail


def reads(kind, b):
    """The four transposing reads of block b of one operand: (text, tag)."""
    base, addr = (FP, VADDR_P) if kind == "P" else (FQ, VADDR_Q)
    r = base + 8 * b
    return [(f"ds_read_b64_tr_b8 v[{r + 2 * i}:{r + 2 * i + 1}], v{addr + b} offset:{i * RD_I}", (kind, b)) for i in range(4)]


def younger(S):
    """LDS-DMA of a step issued before its publish wait."""
    return sum(1 for q in range(16) if S["dma_start"] + q * S["dma_stride"] + 1 <= S["publish"])


def step_items(S, slot, srd, first, last, vmcnt):
    """One K step as a list of items: ("mfma", ri, ci, first) | ("rd", text, tag) | ("need", tag) | ("txt", text).
    `need` markers become counted lgkmcnt waits in resolve()."""
    T = S["T"]
    H = 8 - T
    order = mfma_order(T)
    pre = {m: [] for m in range(64)}
    fill = {m: [] for m in range(64)}
    # the late P blocks of THIS step (needed by its MFMA 8 H), from the slot the address registers still point at
    late = [r for b in range(H, 8) for r in reads("P", b)]
    per = S["top_per_slot"]
    for i, r in enumerate(late):
        (pre if i < per else fill)[0 if i < per else (i // per - 1)].append(("rd",) + r)
    assert (len(late) - 1) // per - 1 < S["bar1"]
    fill[S["bar1"]] += [("txt", "s_waitcnt lgkmcnt(0)"), ("txt", "s_barrier")]
    m = S["dma_start"]
    assert m > S["bar1"]
    for q in range(16):
        img, j = q // 8, q % 8
        voff = (VOFF_Q if img else VOFF_P) + j
        s = "Q" if img else "P"
        fill[m].append(("txt", f"s_add_u32 m0, %[ldsw], {slot * SLOT + img * IMG + j * PIECE}"))
        fill[m + 1].append(("txt", f"buffer_load_dwordx4 v{voff}, %[{srd}{s}], %[sk{s}] offen lds"))
        m += S["dma_stride"]
    last_dma = m - S["dma_stride"] + 1
    assert last_dma + 2 < 63
    fill[last_dma + 1].append(("txt", "s_add_u32 %[skP], %[skP], %[sP128]"))
    fill[last_dma + 2].append(("txt", "s_add_u32 %[skQ], %[skQ], %[sQ128]"))
    # flip the fragment-read addresses to the other ring slot: behind the top reads, in front of the read-ahead
    f0 = max(S["bar1"] + 1, (len(late) - 1) // per)
    for i in range(16):
        mm = f0 + i
        assert mm < S["publish"]
        reg = (VADDR_P + i) if i < 8 else (VADDR_Q + i - 8)
        fill[mm].append(("txt", f"v_xor_b32 v{reg}, 0x10000, v{reg}"))
    if not last:
        assert S["publish"] < 8 * H + T - 1, "the publish barrier must precede the first Q read-ahead"
        fill[S["publish"]] += [("txt", f"s_waitcnt vmcnt({vmcnt})"), ("txt", "s_barrier")]
        for b in range(H):                      # early P blocks: as soon as published and dead (last MFMA of row b is 8 b + 7)
            mm = max(S["publish"] + 1 + b, 8 * b + 7)
            fill[mm] += [("rd",) + r for r in reads("P", b)]
        for ci in range(8):                     # Q block ci: behind the last MFMA of its tail column
            fill[8 * H + T * ci + T - 1] += [("rd",) + r for r in reads("Q", ci)]
    items = []
    for m, (ri, ci) in enumerate(order):
        items += pre[m]
        items += [("need", ("P", ri)), ("need", ("Q", ci)), ("mfma", ri, ci, first)]
        items += fill[m]
    return items


def resolve(prev, cur):
    """Text of step `cur`, its `need` markers resolved against the issue order of `prev` + `cur` (LDS reads return in order:
    block X is complete once at most n reads issued after X's last one are outstanding)."""
    seq = [it for it in prev if it[0] in ("rd", "txt")] + list(cur)
    base = len(seq) - len(cur)
    out = []
    have = None                      # strongest wait already in force since the last read was issued: (index in seq, n)
    for k in range(base, len(seq)):
        it = seq[k]
        if it[0] == "need":
            # reads issued before k, newest first, until X's last read or a full drain
            n, found = 0, False
            for j in range(k - 1, -1, -1):
                p = seq[j]
                if p[0] == "txt" and p[1] == "s_waitcnt lgkmcnt(0)":
                    break
                if p[0] == "rd":
                    if p[2] == it[1]:
                        found = True
                        break
                    n += 1
            if not found:
                continue                                  # complete since the last full drain
            n = min(n, 15)
            if have is not None and have <= n:
                continue
            out.append(f"s_waitcnt lgkmcnt({n})")
            have = n
        elif it[0] == "rd":
            out.append(it[1])
            have = None
        elif it[0] == "mfma":
            _, ri, ci, first = it
            c = "0" if first else acc(ri, ci)
            out.append(f"v_mfma_f32_16x16x128_f8f6f4 {acc(ri, ci)}, {blk(FP, ri)}, {blk(FQ, ci)}, {c}@FMT@")
        else:
            out.append(it[1])
            if it[1] == "s_waitcnt lgkmcnt(0)":
                have = 0
    return out


def setup_text():
    t = ["s_nop 4"]
    for b in range(8):
        t.append(f"v_xor_b32 v{VADDR_P + b}, {b << 4}, %[vP]")
        t.append(f"v_xor_b32 v{VADDR_Q + b}, {b << 4}, %[vQ]")
    t += [f"v_mov_b32 v{VOFF_P}, %[voffP]", f"v_mov_b32 v{VOFF_Q}, %[voffQ]"]
    for j in range(1, 8):
        t.append(f"v_add_u32 v{VOFF_P + j}, %[sP16], v{VOFF_P + j - 1}")
        t.append(f"v_add_u32 v{VOFF_Q + j}, %[sQ16], v{VOFF_Q + j - 1}")
    return t


def tile_text(S):
    T = S["T"]
    H = 8 - T
    t = setup_text()
    t += ["s_mov_b32 %[skP], 0", "s_mov_b32 %[skQ], 0"]
    for step in range(2):                       # the workgroup's K steps 0 and 1
        for q in range(16):
            img, j = q // 8, q % 8
            voff = (VOFF_Q if img else VOFF_P) + j
            s = "Q" if img else "P"
            t.append(f"s_add_u32 m0, %[ldsw], {step * SLOT + img * IMG + j * PIECE}")
            t.append("s_nop 0")
            t.append(f"buffer_load_dwordx4 v{voff}, %[cur{s}], %[sk{s}] offen lds")
        t += ["s_add_u32 %[skP], %[skP], %[sP128]", "s_add_u32 %[skQ], %[skQ], %[sQ128]"]
    t += ["s_waitcnt vmcnt(16)", "s_barrier"]
    pro = [r for b in range(H) for r in reads("P", b)] + [r for b in range(8) for r in reads("Q", b)]
    t += [r[0] for r in pro]
    t.append("s_waitcnt lgkmcnt(0)")
    k = str(younger(S))
    steady0 = step_items(S, 0, "cur", False, False, k)
    steady1 = step_items(S, 1, "cur", False, False, k)
    drained = [("txt", "s_waitcnt lgkmcnt(0)")]
    t += resolve(drained, step_items(S, 0, "cur", True, False, k))
    t += resolve(steady0, steady1)
    t += ["s_cmp_eq_u32 %[nloop], 0", "s_cbranch_scc1 TN8_TAIL_%=", "s_mov_b32 %[cnt], %[nloop]", "TN8_LOOP_%=:"]
    t += resolve(steady1, steady0)
    t += resolve(steady0, steady1)
    t += ["s_sub_u32 %[cnt], %[cnt], 1", "s_cmp_lg_u32 %[cnt], 0", "s_cbranch_scc1 TN8_LOOP_%=", "TN8_TAIL_%=:"]
    t += resolve(steady1, step_items(S, 0, "nul", False, False, k))
    t += resolve(steady0, step_items(S, 1, "nul", False, True, None))
    t += ["s_waitcnt vmcnt(0)", "s_nop 7", "s_nop 7"]
    return t


def c_string3(lines, indent="  "):
    out = []
    for l in lines:
        if "@FMT@" in l:
            a, b = l.split("@FMT@")
            out.append(f'{indent}"{a}" FMT "{b}\\n\\t"')
        else:
            out.append(c_string([l], indent))
    return "\n".join(out)


def clobbers():
    regs = [f"v{i}" for i in range(VOFF_P, 256)] + [f"a{i}" for i in range(256)]
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
    p = ["// GENERATED by tools/gen_gemm_tn8.py - do not edit (tests/test_gemm_nta_gen_cpu.py compares it with the generator).",
         "// Main loop of gemm_tn8_kernel as inline-asm text; register map, pipeline and schedule: see the generator's docstring.",
         "#pragma once", ""]
    for v, S in SCHEDULES.items():
        p.append(f"// schedule {v}: {S}.  FMT: a string, the format suffix of the MFMAs (\"\" = e4m3 x e4m3, \" cbsz:1\" = e5m2 P operand).")
        p.append(f"#define TN8_ASM_{v}(FMT) \\")
        p.append(" \\\n".join(c_string3(tile_text(S)).split("\n")))
        p.append("")
    p.append("#define TN8_CLOBBERS \\")
    p.append(" \\\n".join(clobbers().split("\n")))
    p.append("")
    return "\n".join(p)


if __name__ == "__main__":
    text = render()
    if "--check" in sys.argv:
        sys.exit(0 if open(OUT).read() == text else 1)
    open(OUT, "w").write(text)
    print("wrote", OUT, len(text.splitlines()), "lines")
