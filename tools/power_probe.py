"""Sample rocm-smi power / clocks while a GEMM loop runs (is the ~1000 TF/s plateau a power cap?).
   python tools/power_probe.py [nt|tn|idle]"""
import os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clipa_amd import ops
mode = sys.argv[1] if len(sys.argv) > 1 else "nt"
bf16 = torch.bfloat16
a = torch.randn(65536, 4096, device="cuda").to(bf16)
w = (torch.randn(4096, 4096, device="cuda") * 0.05).to(bf16)
stop = False
samples = []
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--csv"], capture_output=True, text=True, timeout=10).stdout
            samples.append(out.strip().splitlines())
        except Exception as e:
            samples.append([repr(e)])
        time.sleep(0.3)
th = threading.Thread(target=sampler); th.start()
t0 = time.time(); n = 0
torch.cuda.synchronize()
while time.time() - t0 < 6.0:
    if mode == "nt":
        for _ in range(20): ops.gemm_nt(a, w)
        n += 20
    elif mode == "tn":
        for _ in range(5): ops.gemm_tn(a, a[:, :1024], bf16)
        n += 5
    else:
        time.sleep(0.2)
    torch.cuda.synchronize()
el = time.time() - t0
stop = True; th.join()
if mode == "nt": print("TF/s", 2 * 65536 * 4096 * 4096 * n / el / 1e12)
if mode == "tn": print("TF/s", 2 * 65536 * 4096 * 1024 * n / el / 1e12)
print(samples[0][0] if samples and samples[0] else "")
for s in samples[1::3]:
    print(s[-1] if s else "")
