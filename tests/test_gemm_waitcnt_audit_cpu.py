"""The window epilogue of gemm_nt2<bf16> / gemm_nt_f8 (clipa_amd/csrc/gemm_common.h) counts `s_waitcnt vmcnt(N)` by hand around
inline-asm loads.  This test cross-compiles the two kernels to gfx950 assembly (no GPU needed) and replays the counts with
tools/audit_hidden_loads.py: no instruction may touch the destination of a hidden load that can still be in flight, and no
instantiation may use scratch (a spill's loads and stores sit on the same counter: they cannot make a counted wait too short -
extra younger operations only make it stricter - but every reload is a memory round trip inside the epilogue, and the counts
are only exact, i.e. the stores only stay in flight, when nothing else is issued)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src,min_kernels", [("gemm_nt.hip", 2), ("gemm_f8.hip", 8)])
def test_hand_counted_waits_hold_on_the_assembly(tmp_path, src, min_kernels):
    asm = tmp_path / (src + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "clipa_amd", "csrc"), "-I",
           os.path.join(ROOT, "include"), "-Wno-unused-result", "-ffp-contract=fast", "-S", "--cuda-device-only", "-o", str(asm),
           os.path.join(ROOT, "clipa_amd", "csrc", src)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    text = asm.read_text()
    # no kernel of these files may spill: private segment 0 everywhere
    sizes = [int(x) for x in re.findall(r"\.private_segment_fixed_size:\s+(\d+)", text)]
    assert sizes and all(s == 0 for s in sizes), sizes
    a = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_hidden_loads.py"), str(asm)], capture_output=True, text=True)
    assert a.returncode == 0, a.stdout[-3000:]
    audited = [ln for ln in a.stdout.splitlines() if "hidden loads" in ln and not ln.split(":")[1].strip().startswith("0 hidden")]
    assert len(audited) >= min_kernels, a.stdout[-3000:]       # the aux epilogues (residual add, activation backward) were seen
    assert all(ln.rstrip().endswith("0 violations") for ln in audited), a.stdout[-3000:]


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
def test_no_store_data_race_in_any_kernel(tmp_path):
    """gfx950: a VALU write of the third / fourth data register of a 12- / 16-byte store in the slot right behind it overtakes the
    store's operand read (found on gemm_f8a's first hardware run; hipcc does not model it for stores with an SGPR offset).  Every
    kernel file of the library is cross-compiled and scanned (clipa_amd/isa_audit.py: store_data_races)."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, ROOT)
    from clipa_amd import isa_audit as audit_nta
    from clipa_amd.build import SOURCES

    def one(src):
        asm = tmp_path / (src + ".s")
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "clipa_amd", "csrc"), "-I",
               os.path.join(ROOT, "include"), "-Wno-unused-result", "-ffp-contract=fast", "-S", "--cuda-device-only", "-o", str(asm),
               os.path.join(ROOT, "clipa_amd", "csrc", src)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return audit_nta.store_data_races(str(asm), asm.read_text().splitlines())
    with ThreadPoolExecutor(max_workers=8) as ex:
        found = [p for ps in ex.map(one, SOURCES) for p in ps]
    assert not found, "\n".join(found[:20])
