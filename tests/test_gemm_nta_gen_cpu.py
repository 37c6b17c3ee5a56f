"""gemm_nta (clipa_amd/csrc/gemm_nta.hip): the main loop is generated text and the accumulators live in registers hipcc does
not know about.  Without a GPU this checks that (1) the committed gemm_nta_asm.inc IS what tools/gen_gemm_nta.py generates,
(2) every schedule keeps the pipeline's ordering rules (slot freed before it is re-filled, publish wait after the step's last
LDS-DMA, fragment registers not re-loaded before their last use), and (3) the cross-compiled ISA passes clipa_amd/isa_audit.py:
no scratch, no compiler-generated access to an accumulation register, 512 registers per wave."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_inc_is_up_to_date():
    import gen_gemm_nta as G
    import gen_gemm_tna as T
    assert open(G.OUT).read() == G.render(), "run: python tools/gen_gemm_nta.py"
    assert open(T.OUT).read() == T.render(), "run: python tools/gen_gemm_tna.py"


@pytest.mark.parametrize("sched", [0, 1, 2, 3, 4, 5, 6, 7])          # every schedule that was A/B'd on hardware
def test_schedule_ordering_rules(sched):
    import gen_gemm_nta as G
    S = G.SCHEDULES_ALL[sched]
    lines = G.step_text(S, 0, "cur", False, False, str(G.younger(S)), False)
    pos = {k: [] for k in ("mfma", "rd", "dma", "m0", "bar", "vm", "lgk")}
    for i, l in enumerate(lines):
        key = ("mfma" if l.startswith("v_mfma") else "rd" if l.startswith("ds_read") else "dma" if l.startswith("buffer_load") else
               "m0" if l.startswith("s_add_u32 m0") else "bar" if l == "s_barrier" else "vm" if l.startswith("s_waitcnt vmcnt") else
               "lgk" if l.startswith("s_waitcnt lgkmcnt") else None)
        if key:
            pos[key].append(i)
    assert len(pos["mfma"]) == 128 and len(pos["rd"]) == 32 and len(pos["dma"]) == 16 and len(pos["bar"]) == 2
    rd1, rd0 = pos["rd"][:16], pos["rd"][16:]
    assert max(rd1) < pos["lgk"][0] < pos["bar"][0] < min(pos["dma"])          # slot freed (all waves) before the DMA re-fills it
    assert pos["vm"][0] < pos["bar"][1] < min(rd0)                             # publish: wait, then barrier, then reads
    # the publish wait leaves exactly the LDS-DMA this step has issued before it in flight
    issued = sum(1 for d in pos["dma"] if d < pos["vm"][0])
    assert lines[pos["vm"][0]] == f"s_waitcnt vmcnt({issued})" and issued == G.younger(S)
    assert max(rd0) < pos["lgk"][1]
    for d, m in zip(pos["dma"], pos["m0"]):                                    # an M0 write needs a wait state before its LDS-DMA
        assert m < d and any(m < x < d for x in pos["mfma"])
    # write-after-read on the fragment registers: k-half-0 registers are re-loaded only after the 64 MFMAs that read them
    assert min(rd0) > pos["mfma"][63]
    # k-half-1 registers of block b are re-loaded (at the top of the NEXT step) after their last reader of this step
    regs = lambda l: re.findall(r"v\[(\d+):", l)
    for i in rd1:
        dst = regs(lines[i])[0]
        readers = [j for j in pos["mfma"] if dst in regs(lines[j])[0:2] or f"v[{dst}:" in lines[j]]
        assert all(j > i for j in readers if j >= pos["mfma"][64])             # within a step the read precedes its k-half-1 users


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
def test_isa_audit(tmp_path):
    asm = tmp_path / "gemm_nta.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "clipa_amd", "csrc"), "-I",
           os.path.join(ROOT, "include"), "-Wno-unused-result", "-ffp-contract=fast", "-S", "--cuda-device-only", "-o", str(asm),
           os.path.join(ROOT, "clipa_amd", "csrc", "gemm_nta.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    a = subprocess.run([sys.executable, os.path.join(ROOT, "clipa_amd", "isa_audit.py"), str(asm)], capture_output=True, text=True)
    assert a.returncode == 0, a.stdout[-3000:]
    assert len(re.findall(r"\.name:\s+\S*gemm_nta_kernel", asm.read_text())) == 24      # 8 epilogue flavours (5 + the three e4m3 ones) x 3 schedules


@pytest.mark.parametrize("sched", [0, 1])
def test_tn_schedule_ordering_rules(sched):
    """gemm_tna: same rules, plus the slot flip of the fragment-read addresses sits between this step's reads and the read-ahead."""
    import gen_gemm_nta as G
    import gen_gemm_tna as T
    S = T.SCHEDULES[sched]
    lines = T.step_text(S, 0, "cur", False, False, str(G.younger(S)), True)
    idx = lambda pred: [i for i, l in enumerate(lines) if pred(l)]
    mf = idx(lambda l: l.startswith("v_mfma") and "%[cs" not in l)
    rd, dma, bar = idx(lambda l: l.startswith("ds_read_b64_tr_b16")), idx(lambda l: l.startswith("buffer_load")), idx(lambda l: l == "s_barrier")
    flip, vm = idx(lambda l: l.startswith("v_xor_b32")), idx(lambda l: l.startswith("s_waitcnt vmcnt"))
    assert len(mf) == 128 and len(rd) == 64 and len(dma) == 16 and len(bar) == 2 and len(flip) == 16
    assert len(idx(lambda l: "%[cs" in l)) == 16                                 # P^T . ones: one MFMA per P block and k-half
    rd1, rd0 = rd[:32], rd[32:]
    assert max(rd1) < bar[0] < min(dma) and bar[0] < min(flip) and max(flip) < vm[0] < bar[1] < min(rd0)
    assert min(rd0) > mf[63]
    issued = sum(1 for d in dma if d < vm[0])
    assert lines[vm[0]] == f"s_waitcnt vmcnt({issued})"


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
def test_tn_isa_audit(tmp_path):
    asm = tmp_path / "gemm_tna.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "clipa_amd", "csrc"), "-I",
           os.path.join(ROOT, "include"), "-Wno-unused-result", "-ffp-contract=fast", "-S", "--cuda-device-only", "-o", str(asm),
           os.path.join(ROOT, "clipa_amd", "csrc", "gemm_tna.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    a = subprocess.run([sys.executable, os.path.join(ROOT, "clipa_amd", "isa_audit.py"), str(asm)], capture_output=True, text=True)
    assert a.returncode == 0, a.stdout[-3000:]


def test_f8a_inc_is_up_to_date():
    import gen_gemm_f8a as F
    assert open(F.OUT).read() == F.render(), "run: python tools/gen_gemm_f8a.py"


def test_f8a_schedule_ordering_rules():
    """gemm_f8a: a step is 64 MFMAs that each consume a whole block row, so the read-ahead follows the registers: a block is
    re-loaded only after the last MFMA that reads it, and in front of the first one of the next step that does (counted waits)."""
    import gen_gemm_f8a as F
    lines = F.step_text(0, "cur", False, False, "16")
    idx = lambda pred: [i for i, l in enumerate(lines) if pred(l)]
    mf, rd, dma = idx(lambda l: l.startswith("v_mfma")), idx(lambda l: l.startswith("ds_read_b128")), idx(lambda l: l.startswith("buffer_load"))
    m0, bar, vm = idx(lambda l: l.startswith("s_add_u32 m0")), idx(lambda l: l == "s_barrier"), idx(lambda l: l.startswith("s_waitcnt vmcnt"))
    assert len(mf) == 64 and len(rd) == 32 and len(dma) == 16 and len(bar) == 2 and len(vm) == 1
    own = rd[:2]                                                        # this step's own B7, read at the top
    assert max(own) < mf[0] and bar[0] < min(dma) and max(dma) < vm[0] < bar[1] < min(rd[2:])
    assert lines[vm[0]] == "s_waitcnt vmcnt(16)"                        # all 16 pieces of step + 2 are younger than step + 1's
    for d, m in zip(dma, m0):
        assert m < d and any(m < x < d for x in mf)
    # the slot is freed only after every fragment of the step is in registers: lgkmcnt(0) in front of barrier 1
    assert lines[bar[0] - 1] == "s_waitcnt lgkmcnt(0)"
    # write-after-read: a read-ahead into block registers sits behind the last MFMA that names them
    for i in rd[2:]:
        lo = int(re.match(r"ds_read_b128 v\[(\d+):", lines[i]).group(1)) // 8 * 8
        users = [j for j in mf if f"v[{lo}:{lo + 7}]" in lines[j]]
        assert users and max(users) < i, lines[i]
    # counted waits in front of the first eight MFMAs: token block m has landed before MFMA m reads it
    for m in range(8):
        assert lines[mf[m] - 1] == f"s_waitcnt lgkmcnt({min(15, 16 - 2 * m)})"
    last = F.step_text(1, "nxt", False, True, None)                     # last step of a tile: no publish, no read-ahead, the vectors ride along
    assert sum(l.startswith("ds_read_b128") for l in last) == 2 and sum(l == "s_barrier" for l in last) == 1
    vec = [i for i, l in enumerate(last) if "%[srdVec]" in l]
    pieces = [i for i, l in enumerate(last) if l.startswith("buffer_load") and "%[srdVec]" not in l]
    assert len(vec) == 1 and len(pieces) == 16 and vec[0] < min(pieces)     # older than the 16 pieces: the tile's final vmcnt(16) covers it


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
def test_f8a_isa_audit(tmp_path):
    asm = tmp_path / "gemm_f8a.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "clipa_amd", "csrc"), "-I",
           os.path.join(ROOT, "include"), "-Wno-unused-result", "-ffp-contract=fast", "-S", "--cuda-device-only", "-o", str(asm),
           os.path.join(ROOT, "clipa_amd", "csrc", "gemm_f8a.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    a = subprocess.run([sys.executable, os.path.join(ROOT, "clipa_amd", "isa_audit.py"), str(asm)], capture_output=True, text=True)
    assert a.returncode == 0, a.stdout[-3000:]
    assert len(re.findall(r"\.name:\s+\S*gemm_f8a_kernel", asm.read_text())) == 24      # 12 epilogue flavours (5 + the two e4m3 pre-activation ones + 4 with a producer-quantised output + the operand-emitting GELU-backward) x e4m3 / e5m2 A operand


def test_audit_rejects_a_store_data_race(tmp_path):
    """The rule gemm_f8a's first hardware run paid for: a VALU write of the third / fourth data register in the slot right
    behind a 16-byte store."""
    from clipa_amd import isa_audit as audit_nta
    head = "_ZN10clipa_gemm12_GLOBAL__N_115gemm_f8a_kernelILi0ELb0ELi0EEEvNS_7F8AArgsE:\n"
    bad = head + "\tbuffer_store_dwordx4 v[6:9], v101, s[40:43], s96 offen\n\tv_pk_mul_f32 v[8:9], v[100:101], v[18:19]\n.Lfunc_end0:\n"
    ok = head + "\tbuffer_store_dwordx4 v[6:9], v101, s[40:43], s96 offen\n\ts_nop 0\n\tv_pk_mul_f32 v[8:9], v[100:101], v[18:19]\n.Lfunc_end0:\n"
    ok2 = head + "\tbuffer_store_dwordx4 v[6:9], v101, s[40:43], s96 offen\n\tv_accvgpr_read_b32 v6, a[4]\n.Lfunc_end0:\n"
    for name, text, n in (("bad", bad, 1), ("ok", ok, 0), ("ok2", ok2, 0)):
        p = tmp_path / (name + ".s")
        p.write_text(text)
        assert len(audit_nta.store_data_races(str(p), text.splitlines())) == n, name


def test_audit_counts_the_epilogue_stores(tmp_path):
    """The tile statement waits with vmcnt(16 + S), S = the 16-byte stores the epilogue in front of it issued: a copy of the
    epilogue with fewer (merged / dropped) stores would let operands be read before they landed.  Every copy is
    bracketed by markers and counted; a missing marker pair is a finding as well."""
    from clipa_amd import isa_audit
    head = "_ZN10clipa_gemm12_GLOBAL__N_115gemm_nta_kernelILi1ELi1ELb0ELi4EEEvNS_6NTArgsE:\n"       # <ACT, PRE = 1 (bf16 copy), ., .>
    st = "\tbuffer_store_dwordx4 v[6:9], v101, s[40:43], s96 offen\n\ts_nop 0\n"
    st8 = "\tbuffer_store_dwordx2 v[6:7], v101, s[40:43], s96 offen\n\ts_nop 0\n"
    def body(n, n8=0, h=head):
        return h + "\t;;#ASMSTART\n\t; CLIPA_EPI_BEGIN 0\n\t;;#ASMEND\n" + st * n + st8 * n8 + "\t;;#ASMSTART\n\t; CLIPA_EPI_END 0\n\t;;#ASMEND\n.Lfunc_end0:\n"
    head8 = head.replace("ILi1ELi1E", "ILi1ELi2E")                                                    # PRE = 2: e4m3 copy = 8-byte stores
    assert isa_audit.epilogue_store_counts("x", body(32, 32, head8).splitlines()) == []
    assert len(isa_audit.epilogue_store_counts("x", body(32, 31, head8).splitlines())) == 1
    assert len(isa_audit.epilogue_store_counts("x", body(64, 0, head8).splitlines())) == 1
    assert isa_audit.epilogue_store_counts("x", body(64).splitlines()) == []          # PRE: two outputs x 32
    assert len(isa_audit.epilogue_store_counts("x", body(63).splitlines())) == 1
    assert len(isa_audit.epilogue_store_counts("x", body(65).splitlines())) == 1
    assert len(isa_audit.epilogue_store_counts("x", (head + st * 64 + ".Lfunc_end0:\n").splitlines())) == 1   # no markers


def test_tn8_inc_is_up_to_date():
    import gen_gemm_tn8 as T8
    assert open(T8.OUT).read() == T8.render(), "run: python tools/gen_gemm_tn8.py"


@pytest.mark.parametrize("sched", [0, 1, 2])
def test_tn8_schedule_ordering_rules(sched):
    """gemm_tn8 (the fp8 weight-gradient GEMM): a step is 64 MFMAs over 128 reduction rows from 128 registers of fragments that
    are re-loaded as they die.  For every schedule of the generator, on a STEADY-STATE step (resolved against its predecessor):
    every MFMA covers its block pair once; a block's registers are re-loaded only behind the last MFMA of the step that reads
    them; in front of every MFMA the counted lgkmcnt wait guarantees both of its blocks have landed (checked by replaying the
    issue order with in-order LDS returns); the ring slot is freed (barrier 1) after the step's last own read and before the first
    LDS-DMA, the address registers are flipped between the step's own reads and the read-ahead, and the publish wait leaves exactly
    the LDS-DMA issued before it in flight."""
    import gen_gemm_tn8 as T8
    S = T8.SCHEDULES[sched]
    k = str(T8.younger(S))
    prev = T8.step_items(S, 0, "cur", False, False, k)
    cur = T8.step_items(S, 1, "cur", False, False, k)
    lines = T8.resolve(prev, cur)
    idx = lambda pred: [i for i, l in enumerate(lines) if pred(l)]
    mf, rd = idx(lambda l: l.startswith("v_mfma")), idx(lambda l: l.startswith("ds_read_b64_tr_b8"))
    dma, m0 = idx(lambda l: l.startswith("buffer_load")), idx(lambda l: l.startswith("s_add_u32 m0"))
    bar, vm, flip = idx(lambda l: l == "s_barrier"), idx(lambda l: l.startswith("s_waitcnt vmcnt")), idx(lambda l: l.startswith("v_xor_b32"))
    assert len(mf) == 64 and len(rd) == 64 and len(dma) == 16 and len(bar) == 2 and len(vm) == 1 and len(flip) == 16
    blocks = lambda l: tuple(int(x) for x in re.findall(r"v\[(\d+):\d+\]", l)[:2])
    pairs = {blocks(lines[i]) for i in mf}
    assert pairs == {(128 + 8 * r, 192 + 8 * c) for r in range(8) for c in range(8)}
    T = S["T"]
    own = [i for i in rd if i < bar[0]]                       # the late P blocks of this step, read from its own slot at the top
    assert len(own) == 4 * T and lines[bar[0] - 1] == "s_waitcnt lgkmcnt(0)"
    assert max(own) < bar[0] < min(dma) and max(own) < min(flip) and max(flip) < vm[0] < bar[1] < min(i for i in rd if i > bar[0])
    issued = sum(1 for d in dma if d < vm[0])
    assert lines[vm[0]] == f"s_waitcnt vmcnt({issued})" and issued == T8.younger(S)
    for d, m in zip(dma, m0):
        assert m < d and any(m < x < d for x in mf)
    # write-after-read inside the step: a read into a block's registers sits behind every MFMA of THIS step that reads the block,
    # except the top reads (whose users come later in the step)
    for i in rd:
        lo = int(re.match(r"ds_read_b64_tr_b8 v\[(\d+):", lines[i]).group(1)) // 8 * 8
        users = [j for j in mf if lo in blocks(lines[j])]
        assert users
        if i in own:
            assert min(users) > i
        else:
            assert max(users) < i, lines[i]
    # replay: LDS returns in order; `s_waitcnt lgkmcnt(n)` completes all but the n youngest reads.  Start from the predecessor's
    # read-ahead in flight (worst case: nothing of it has returned yet).
    seq = [l for l in T8.resolve(prev, prev)] + lines
    base = len(seq) - len(lines)
    inflight, landed = [], set()
    for pos, l in enumerate(seq):
        if l.startswith("ds_read_b64_tr_b8"):
            r = int(re.match(r"ds_read_b64_tr_b8 v\[(\d+):", l).group(1))
            landed.discard(r)
            inflight.append(r)
        elif l.startswith("s_waitcnt lgkmcnt"):
            n = int(re.search(r"lgkmcnt\((\d+)\)", l).group(1))
            while len(inflight) > n:
                landed.add(inflight.pop(0))
        elif l.startswith("v_mfma") and pos >= base:
            for lo in blocks(l):
                need = {lo + 2 * i for i in range(4)}
                assert need <= landed, (sched, l, sorted(need - landed))
    # the last step of a slice: no publish, no read-ahead
    last = T8.resolve(prev, T8.step_items(S, 1, "nul", False, True, None))
    assert sum(l.startswith("ds_read_b64_tr_b8") for l in last) == 4 * T and sum(l == "s_barrier" for l in last) == 1


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
def test_tn8_isa_audit(tmp_path):
    asm = tmp_path / "gemm_tn8.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "clipa_amd", "csrc"), "-I",
           os.path.join(ROOT, "include"), "-Wno-unused-result", "-ffp-contract=fast", "-S", "--cuda-device-only", "-o", str(asm),
           os.path.join(ROOT, "clipa_amd", "csrc", "gemm_tn8.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    a = subprocess.run([sys.executable, os.path.join(ROOT, "clipa_amd", "isa_audit.py"), str(asm)], capture_output=True, text=True)
    assert a.returncode == 0, a.stdout[-3000:]
    assert len(re.findall(r"\.name:\s+\S*gemm_tn8_kernel", asm.read_text())) == 6      # 3 schedules x e4m3 / e5m2 gradient operand
