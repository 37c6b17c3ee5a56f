"""Build libclipa_hip.so (gfx950) in-tree with hipcc.  `python -m clipa_amd.build [--force]`.

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libclipa_hip.so")
SOURCES = ["gemm_nt.hip", "gemm_nta.hip", "gemm_tn.hip", "gemm_tna.hip", "gemm_tn8.hip", "gemm_f8.hip", "gemm_f8a.hip", "quant.hip", "simce.hip", "layernorm.hip", "attention.hip", "attention_wide.hip", "misc.hip", "augment.hip", "runtime.hip"]
AUDITED = ("gemm_nta.hip", "gemm_tna.hip", "gemm_f8a.hip", "gemm_tn8.hip")          # sources whose device assembly clipa_amd/isa_audit.py checks
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", CSRC, "-I", os.path.join(ROOT, "include"),
         "-Wno-unused-result", "-ffp-contract=fast"]


def _digest():
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)) + ["../../include/clipa_hip.h"]:
        path = os.path.join(CSRC, name)
        if os.path.isfile(path):
            with open(path, "rb") as f:
                h.update(name.encode())
                h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "build.stamp")
    digest = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
        return LIB
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if src in AUDITED:
            cmd.append("-save-temps=obj")            # keeps the device assembly next to the object for the audit below
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        if src in AUDITED:
            # hand-counted waits / literally named registers are only valid for the code hipcc actually emitted: the static
            # audits are part of the build, not only of the test-suite
            asm = os.path.join(objdir, src.replace(".hip", "-hip-amdgcn-amd-amdhsa-gfx950.s"))
            if not os.path.isfile(asm):
                raise RuntimeError(f"ISA audit of {src}: device assembly {asm} not found - this hipcc names its -save-temps "
                                   f"files differently (present: {sorted(f for f in os.listdir(objdir) if f.endswith('.s'))})")
            a = subprocess.run([sys.executable, os.path.join(HERE, "isa_audit.py"), asm], capture_output=True, text=True)
            if a.returncode != 0:
                raise RuntimeError(f"ISA audit of {src} failed (clipa_amd/isa_audit.py):\n{a.stdout[-3000:]}\n{a.stderr[-2000:]}")
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(digest)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
