"""Is the CPU port (oracle/clip_oracle.py, the `cpu_baseline` of bench.py, kind "port") a fair stand-in for the reference's own
CPU path?  Times both on the SAME cores, batch and weights: the real clipa_torch CLIP + ClipLoss (imported from
/root/reference, build container only) and the oracle restatement - forward + loss + backward + AdamW, fp32.
    python tools/cpu_port_vs_reference.py [model] [pairs] > profiles/r02_cpu_port_vs_reference.txt"""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clipa_amd
from oracle import clip_oracle as O, ref_loader

model_name = sys.argv[1] if len(sys.argv) > 1 else "ViT-B-16"
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
threads = min(8, os.cpu_count() or 1)
torch.set_num_threads(threads)
cfg = clipa_amd.get_model_config(model_name)
S, ctx = cfg["vision_cfg"]["image_size"], cfg["text_cfg"]["context_length"]
img, txt = O.synthetic_batch(pairs, S, ctx, cfg["text_cfg"]["vocab_size"], seed=1)
shell = clipa_amd.CLIP(**cfg)
sd0 = {k: v.detach().clone().float() for k, v in shell.state_dict().items()}


def bench(step, n=3):
    step()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
    return min(ts), sorted(ts)[len(ts) // 2]


# --- the port
sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items() if v.is_floating_point()}
ocfg = O.oracle_cfg(cfg)
opt_o = torch.optim.AdamW(list(sd.values()), lr=1e-4, betas=(0.9, 0.95), eps=1e-6, weight_decay=0.2)
def step_port():
    opt_o.zero_grad()
    i, t, s = O.clip_forward(sd, ocfg, O.normalize_images(img), txt)
    loss, _ = O.clip_loss(i, t, s); loss.backward(); opt_o.step(); return float(loss)
lo = step_port()
p_min, p_med = bench(step_port)

# --- the reference itself
M, L, _ = ref_loader.load()
ref = M.CLIP(**cfg)
ref.load_state_dict(sd0, strict=True)
loss_fn = L.ClipLoss()
opt_r = torch.optim.AdamW(ref.parameters(), lr=1e-4, betas=(0.9, 0.95), eps=1e-6, weight_decay=0.2)
x = O.normalize_images(img)
def step_ref():
    opt_r.zero_grad()
    out = ref(x, txt)
    loss = loss_fn(*out) if isinstance(out, tuple) else loss_fn(**out)
    loss.backward(); opt_r.step(); return float(loss)
lr_ = step_ref()
r_min, r_med = bench(step_ref)
print(json.dumps({"model": model_name, "pairs_per_step": pairs, "threads": threads, "first_step_loss_port": lo, "first_step_loss_reference": lr_,
                  "port_s_per_step_min_med": [round(p_min, 3), round(p_med, 3)], "reference_s_per_step_min_med": [round(r_min, 3), round(r_med, 3)],
                  "port_pairs_per_s": round(pairs / p_med, 3), "reference_pairs_per_s": round(pairs / r_med, 3),
                  "port_over_reference_time": round(p_med / r_med, 3)}))
