#!/usr/bin/env python
"""bench.py - image-text pairs/sec of the full CLIP training step on N MI355X of one node.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N --steps K --warmup W          # N > 1 without a launcher: re-executes itself under torch.distributed.run
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = what clipa_torch/training/train.py:187-291 does per iteration: uint8 batch (already resident in
HBM) -> model forward -> global InfoNCE loss (local_loss + gather_with_grad, the reference GPU recipe) ->
backward -> DDP gradient all-reduce -> AdamW -> logit_scale clamp.  Default workload = BASELINE.json's headline
configuration at its per-GPU shape: ViT-L/16 @ 224, text-77, bf16 compute, local batch 4096 (weak scaling:
global batch 4096*N, i.e. 32768 at N=8).  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md: ~2.5 PF dense)
PEAK_HBM_TBPS = 8.0           # HBM3E peak of the same guide
PEAK_FP8_TFLOPS = 5000.0      # dense fp8 peak on v_mfma_f32_16x16x128_f8f6f4 (same guide: ~5 PF dense)
# the MFMA-bound kernels a step can be dominated by: profile key -> (peak TFLOP/s, description)
ROOFLINE_KERNELS = {
    "gemm_nt": (PEAK_BF16_TFLOPS, "gemm_nt (gemm_nta_kernel<EPI, PRE, SCHED>: 4 waves x 512 registers, hand-scheduled v_mfma_f32_16x16x32_bf16 main loop, 256x256x64 tile, all epilogue instantiations; gemm_nt2_kernel on ragged shapes)"),
    "gemm_nt_f8": (PEAK_FP8_TFLOPS, "gemm_nt_f8 (gemm_f8a_kernel: 4 waves x 512 registers, hand-scheduled v_mfma_f32_16x16x128_f8f6f4 main loop, 256x256x128 tile; gemm_nt_f8_kernel on ragged shapes)"),
    "gemm_tn_f8": (PEAK_FP8_TFLOPS, "gemm_tn_f8 (gemm_tn8_kernel: hand-scheduled fp8 weight-gradient GEMM, ds_read_b64_tr_b8 fragments, v_mfma_f32_16x16x128_f8f6f4, 256x256 tile, split-M)"),
    "gemm_tn": (PEAK_BF16_TFLOPS, "gemm_tn (gemm_tna_kernel: hand-scheduled bf16 weight-gradient GEMM, 256x256 tile, split-M; gemm_tn2 / gemm_tn3 on ragged shapes)"),
}


MEASURED_KERNEL_SOURCES = ("common.h", "gemm_common.h", "gemm_nt.hip", "gemm_nta.hip", "gemm_nta_asm.inc")


def kernel_source_sha16():
    """Identity of the sources of the kernels behind `gemm_nt` launches - what a PMC measurement in profiles/traffic.json
    belongs to (the fp8 and weight-gradient GEMMs live in other files and do not invalidate it)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "clipa_amd", "csrc")
    for f in MEASURED_KERNEL_SOURCES:
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def train_gflop_per_pair(cfg, S, ctx):
    """F_tower = layers*L*(24 D^2 + 4 L D), patch GEMM, projections; training = 3x forward;
    activation recompute NOT counted."""
    v, t, E = cfg["vision_cfg"], cfg["text_cfg"], cfg["embed_dim"]
    P, D = v["patch_size"], v["width"]
    g = S // P
    L = g * g + 1
    fwd = v["layers"] * L * (24 * D * D + 4 * L * D) + g * g * 2 * 3 * P * P * D + 2 * D * E
    Dt = t["width"]
    fwd += t["layers"] * ctx * (24 * Dt * Dt + 4 * ctx * Dt) + 2 * Dt * E
    return 3.0 * fwd / 1e9


def pruned_gflop_per_pair(cfg, S, ctx, image_pruned, text_pruned):
    """Model FLOPs of `train_gflop_per_pair` that the engine does NOT execute: the out-projection + MLP of a tower's LAST block
    on the rows its head never reads (engine.LastBlockFn: class-token / EOT pooling; forward, input gradient and weight gradient
    = 3 x 18 D^2 per dead token).  ViT-L/16 @ 224 + text-77: 11.1 + 2.4 GF of 409.2."""
    v, t = cfg["vision_cfg"], cfg["text_cfg"]
    g = S // v["patch_size"]
    dead = 0.0
    if image_pruned:
        dead += (g * g) * 18 * v["width"] ** 2
    if text_pruned:
        dead += (ctx - 1) * 18 * t["width"] ** 2
    return 3.0 * dead / 1e9


def kernel_entry(v, steps):
    """One line of the `kernels` block: MFMA-bound kernels (algorithmic FLOPs known) report TFLOP/s, HBM-bound ones (LayerNorm,
    row quantisers, patch gather, activation re-materialisation, crops) their algorithmic bytes per second and the fraction
    of the 8 TB/s HBM3E peak (MI355X_MICROARCH.md; ~6.3 TB/s is what a streaming kernel reaches on this part)."""
    e = {"launches": v["launches"], "ms_per_step": round(v["ms"] / steps, 2)}
    if v["work"] > 0:
        e["tflops"] = round(v["work"] / max(v["ms"], 1e-9) / 1e9, 1)
    elif v.get("bytes", 0) > 0:
        gbps = v["bytes"] / max(v["ms"], 1e-9) / 1e6
        e["hbm_gbps"] = round(gbps, 1)
        e["hbm_frac"] = round(gbps / (PEAK_HBM_TBPS * 1e3), 4)
    return e


def plan_keep(budget, layers_v, layers_t, mv_b, mt_b, lv_b, lt_b):
    """Spend `budget` bytes of HBM on kept activations where a byte saves the most step time: the "medium" tier of the image
    tower first (drops LN1, in-proj, attention, out-proj of the recompute: ~1.45 ms per GB at ViT-L/16), then its upgrades to
    the tier whose bytes per block are lv_b ("light8" for the bf16 engines: the c_fc recompute goes too, ~1.36 ms per GB; "light"
    for fp8), then the same two steps for the narrower text tower (~0.9 ms per GB).
    -> (upgraded_v, upgraded_t, medium_v, medium_t)."""
    budget = max(0, int(budget))
    mv = min(layers_v, budget // mv_b); budget -= mv * mv_b
    kv = min(mv, budget // (lv_b - mv_b)); budget -= kv * (lv_b - mv_b)
    mt = min(layers_t, budget // mt_b); budget -= mt * mt_b
    kt = min(mt, budget // (lt_b - mt_b))
    return int(kv), int(kt), int(mv - kv), int(mt - kt)


# What one GB of a kept tensor saves in the backward (ms per GB at ViT-L/16 + text-77, local batch 4096; profiles/NOTEBOOK.md 7d): the e4m3
# pre-activation replaces the c_fc recompute GEMM by an HBM-bound re-materialisation, the attention output / x1 / qkv each save the
# kernel that would recompute them (LN1 is needed for the weight gradient either way).  Only the ORDER matters to the planner.
KEEP_VALUE_MS_PER_GB = (("v", "h8", 1.5), ("t", "h8", 1.13), ("v", "a", 0.94), ("v", "x1", 0.91), ("t", "x1", 0.88),
                        ("v", "qkv", 0.79), ("t", "a", 0.70), ("t", "qkv", 0.66))


# The same order without the e4m3 pre-activation: every tensor of this list is kept bit for bit (bf16 as the forward wrote it), so
# the step's gradients are those of the all-recompute step bit for bit (`value_exact_tiers`); the bf16 pre-activation ("h", twice
# the bytes of "h8" for the same saved GEMM) ranks below qkv.
KEEP_VALUE_MS_PER_GB_EXACT = (("v", "a", 0.94), ("v", "x1", 0.91), ("t", "x1", 0.88), ("v", "qkv", 0.79), ("v", "h", 0.75),
                              ("t", "a", 0.70), ("t", "qkv", 0.66), ("t", "h", 0.57))


# The fp8 engine (round 6: per-tensor keep sets there too), priced on ViT-H/14 + text-77 at local batch 2048 (profiles/
# r06_bench_h14_B2048_fp8_wgrad8_first.json): the attention output saves the attention forward (1.7 ms per GB), the e4m3
# pre-activation LN2 + the c_fc GEMM (1.3), x1 the out-projection + a row quantiser (1.0), qkv LN1 + the in-projection (0.7).
KEEP_VALUE_MS_PER_GB_FP8 = (("v", "a", 1.68), ("v", "h8", 1.33), ("t", "h8", 1.05), ("v", "x1", 1.02), ("t", "x1", 0.94),
                            ("t", "a", 0.85), ("v", "qkv", 0.68), ("t", "qkv", 0.56))
KEEP_VALUE_MS_PER_GB_FP8_EXACT = (("v", "a", 1.68), ("v", "x1", 1.02), ("t", "x1", 0.94), ("t", "a", 0.85), ("v", "qkv", 0.68),
                                  ("v", "h", 0.66), ("t", "qkv", 0.56), ("t", "h", 0.52))


def plan_keep_tensors(budget, layers, nbytes, order=KEEP_VALUE_MS_PER_GB, pruned_last=()):
    """Greedy per-tensor plan for the bf16 engines: walk `order`, give each (tower, tensor) as many blocks as the budget still
    holds.  layers: {"v": n, "t": n}; nbytes: {(tower, tensor): bytes per block}; pruned_last: towers whose LAST block runs its
    out-projection / MLP on the pooled rows only (engine.LastBlockFn) - keep_counts gives a tensor to the LAST n blocks, so the
    first block of an "x1" / "h8" / "h" count is that one and holds a few MB instead of a token-level tensor.
    -> {tower: {tensor: blocks}}."""
    budget = max(0, int(budget))
    plan = {tw: {"h8": 0, "h": 0, "a": 0, "x1": 0, "qkv": 0} for tw in layers}
    for tw, name, _ in order:
        free = 1 if (tw in pruned_last and name in ("x1", "h8", "h")) else 0
        n = min(layers[tw], free + budget // nbytes[(tw, name)])
        plan[tw][name] = int(n)
        budget -= max(0, n - free) * nbytes[(tw, name)]
    return plan


def dist_reservation(world, local_batch, embed_dim, n_params, grad_bytes_per_param):
    """What a rank leaves free for the exchanges of a `world`-rank step, itemised (bytes) - memory the collective-free trial step of
    the planner never sees (explicit amounts instead of smaller fractions):
      rccl_buffers       RCCL's own hipMalloc'ed channel / proxy buffers (outside torch's allocator, so invisible to its peak
                         statistics): an estimate, 0.5 GiB + 0.35 GiB per peer (ring + tree connections of up to 32 channels x
                         4 MiB per protocol); the 1-rank records measure 0.3 GiB
      ddp_buckets        DistributedDataParallel's flat gradient buckets (one copy of every gradient)
      gathered_features  the fused [W B, 2 E] bf16 gather buffer, its fp32 gradient and the reduce-scatter output (loss.py)
      logits_workspace   the similarity / cross-entropy workspace at the gathered width (per-row statistics and bf16 dlogits
                         [B, W B] of simce)"""
    W = max(1, int(world))
    gathered = W * local_batch * 2 * embed_dim
    return {"rccl_buffers": int((0.5 + 0.35 * (W - 1)) * 2 ** 30) if W > 1 else 0,
            "ddp_buckets": int(n_params * grad_bytes_per_param) if W > 1 else 0,
            "gathered_features": int(gathered * (2 + 4) + local_batch * 2 * embed_dim * 4) if W > 1 else 0,
            "logits_workspace": int(2 * local_batch * W * local_batch * 2 + 16 * local_batch * 4) if W > 1 else 0}


def agree_budget(budget, device):
    """Every rank must keep the SAME blocks (identical collectives, identical step time; a rank that keeps more can run
    out of HBM alone): all ranks adopt the smallest activation budget any of them measured."""
    import torch.distributed as dist
    bt = torch.tensor([int(budget)], device=device, dtype=torch.int64)
    dist.all_reduce(bt, op=dist.ReduceOp.MIN)
    return int(bt.item())


def usable_cores(cap=64):
    """Host cores this process may really use: affinity mask and cgroup CPU quota, capped (an OpenMP team far
    larger than the quota spin-waits and is slower than a small one)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, cap))


def cpu_baseline(cfg, S, ctx, sample_pairs, threads):
    """The CPU port of the reference path (oracle/clip_oracle.py, fp32, torch CPU ops) on the host cores:
    forward + loss + backward + AdamW on a bounded sample of the same workload."""
    from oracle import clip_oracle as O
    import clipa_amd
    torch.set_num_threads(threads)
    m = clipa_amd.CLIP(**cfg)                      # parameter container only (shapes / init); maths is the oracle's
    sd = {k: v.detach().clone().float().requires_grad_(v.requires_grad) for k, v in m.named_parameters()}
    ocfg = O.oracle_cfg(cfg)
    opt = torch.optim.AdamW([p for p in sd.values() if p.requires_grad], lr=1e-4, betas=(0.9, 0.95), eps=1e-6,
                            weight_decay=0.2)
    img, txt = O.synthetic_batch(sample_pairs, S, ctx, cfg["text_cfg"]["vocab_size"], seed=1)

    def step():
        opt.zero_grad()
        i, t, s = O.clip_forward(sd, ocfg, O.normalize_images(img), txt)
        loss, _ = O.clip_loss(i, t, s)
        loss.backward()
        opt.step()
        return float(loss)

    step()                                          # warm-up
    n, t0 = 0, time.perf_counter()
    while n < 2 or (time.perf_counter() - t0 < 10.0 and n < 8):
        step()
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": sample_pairs / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
            # measured in the build container on the current oracle (tools/cpu_port_vs_reference.py, 8 threads - all the container
            # has; same weights / batch, first-step loss equal to 2e-7): ViT-B/16 0.93, ViT-L/16 1.07
            "port_over_reference": {"time_ratio": [0.93, 1.07], "source": "profiles/r06_cpu_port_vs_reference.txt"},
            "sample": f"{sample_pairs} pairs/step x {n} timed steps (+1 warm-up), fp32 fwd+loss+bwd+AdamW, "
                      f"oracle/clip_oracle.py on torch CPU ops"}


def fp8_child_line(args):
    """The same workload in fp8 (forward, input-gradient and weight-gradient products of the block linears on
    v_mfma_f32_16x16x128_f8f6f4), measured by a child process while this one has not allocated anything on the GPU yet.
    -> the summary that goes into the bench line as `fp8_same_workload` (errors are reported there, never raised)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--precision", "fp8", "--model", args.model,
           "--image-size", str(args.image_size), "--ctx", str(args.ctx), "--batch", str(args.batch), "--accum-freq", str(args.accum_freq),
           "--steps", str(args.fp8_line_steps), "--warmup", "1", "--no-cpu-baseline", "--h2d-steps", "0", "--plain-steps", "0",
           "--exact-steps", "0", "--unpad-steps", "0", "--fp8-line-steps", "0"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=args.fp8_line_timeout, env=env)
        rows = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not rows:
            return {"value": None, "error": (r.stderr or r.stdout)[-300:]}
        j = json.loads(rows[-1])
        ro = j.get("roofline") or {}
        return {"value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "steps": j["steps"], "dtype": j["dtype"],
                "loss": j.get("loss"), "workload": j["config"]["workload"], "model_flops_util_of_fp8_peak": j.get("model_flops_util"),
                "roofline": {k: ro.get(k) for k in ("kernel", "achieved", "peak", "unit", "frac")},
                "note": "child process `bench.py --precision fp8` on the same model / batch before this process allocated on the GPU; "
                        "parity of the fp8 step: tests/test_fp8_gpu.py"}
    except subprocess.TimeoutExpired:
        return {"value": None, "error": f"timed out after {args.fp8_line_timeout}s"}
    except Exception as e:      # the headline line must survive
        return {"value": None, "error": repr(e)[:300]}


def self_launch_argv(n_gpus, script_args, port=None):
    """`python bench.py --gpus N` (N > 1) outside a launcher: the argv this process re-executes itself with - one rank per GPU
    under torch.distributed.run on this node, rendezvous on 127.0.0.1 (the reference's `torchrun --nproc_per_node 8`,
    scripts/exp/gpu/vit_l16/i37_t8_pretrain.sh:1).  The torchrun form of the docstring keeps working: it sets WORLD_SIZE, and a
    process that finds WORLD_SIZE set never re-executes."""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_gpus)}",
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__)] + list(script_args)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="ViT-L-16")
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--ctx", type=int, default=77)
    ap.add_argument("--batch", type=int, default=4096, help="local (per-GPU) batch")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "amp_bf16", "fp8"],
                    help="bf16 = reference 'bf16' mode (bf16 weights); amp_bf16 = fp32 master weights; fp8 = BASELINE "
                         "configs[3]: bf16 weights / activations, e4m3 operands for the block GEMMs (forward + input gradient)")
    ap.add_argument("--fp8-predicted-scales", action="store_true",
                    help="--precision fp8: the engine knob fp8_predicted_scales (the MLP's 4 D-wide tensors leave their GEMMs as e4m3 "
                         "operands with predicted row scales; off by default: slightly lower gradient parity, DESIGN 4)")
    ap.add_argument("--keep-blocks", default="auto", help="'auto' or 'LV,LT[,MV,MT]': light-kept (and medium-kept) blocks per tower (image, text)")
    ap.add_argument("--keep-fraction", type=float, default=None,
                    help="share of the free HBM 'auto' may spend (default 0.97 single process, 0.94 with several ranks; a trial step + vote backs the plan off)")
    ap.add_argument("--tier-plan", action="store_true",
                    help="plan whole tiers (medium, then light8 upgrades) instead of tensor by tensor (the round-4 default for bf16)")
    ap.add_argument("--no-light8", action="store_true",
                    help="keep bit-exact tensors only: no e4m3 pre-activation (per-tensor plan: the bf16 pre-activation ranks below "
                         "qkv; --tier-plan: the bf16 'light' tier instead of 'light8').  The default run reports this plan's rate as "
                         "`value_exact_tiers` next to `value`")
    ap.add_argument("--exact-steps", type=int, default=3,
                    help="extra steps after the timed region under the keep plan restricted to bit-exact tensors (same budget, no e4m3 "
                         "pre-activation): reported as `value_exact_tiers`, never part of `value` (0 = skip)")
    ap.add_argument("--no-second-pass", action="store_true",
                    help="skip the second planning pass (spend only the budget derived from the all-recompute step's peak)")
    ap.add_argument("--second-pass-fraction", type=float, default=None,
                    help="share of the device memory still free after the planned trial step that the second planning pass may spend "
                         "(default 0.6 single process, 0.4 with several ranks; 0.8 measured 286 of 288 GiB reserved and one allocator retry)")
    ap.add_argument("--unpad-text", action="store_true",
                    help="run the TIMED region with the engine's unpad_text knob (causal text tower on the tokens up to each caption's "
                         "EOT only; identical features / loss / gradients).  Off by default: `value` is measured on the reference's "
                         "schedule (all 77 positions); the unpadded rate is reported next to it as `unpadded_text`")
    ap.add_argument("--unpad-steps", type=int, default=3,
                    help="extra steps after the timed region with unpad_text on (reported as `unpadded_text`, never part of `value`; 0 = skip)")
    ap.add_argument("--alloc-conf", default=None,
                    help="PyTorch caching-allocator settings applied before the first allocation, e.g. expandable_segments:True")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shapes", action="store_true", help="also report time and TF/s per GEMM / attention shape (stderr)")
    ap.add_argument("--accum-freq", type=int, default=1,
                    help="the reference's --accum-freq (train.py:216-256): the local batch is processed as this many micro-batches "
                         "with the feature-cache procedure (no-grad forward of all, then forward+backward of each against the "
                         "cached features of the others): one extra forward per pair buys a local batch that does not fit at once")
    ap.add_argument("--optimizer", default="adamw", choices=["adamw", "sharded"],
                    help="adamw = the reference's arrangement (DDP gradient all-reduce + a full fused AdamW per rank); sharded = "
                         "clipa_amd.zero.ShardedAdamW (gradient reduce-scatter, optimizer state / W, parameter all-gather; no DDP wrapper)")
    ap.add_argument("--exchange", default="reduce_scatter", choices=["reduce_scatter", "all_to_all"],
                    help="--optimizer sharded: RCCL reduce-scatter, or one-hop all-to-all over the xGMI mesh + local sum")
    ap.add_argument("--h2d-steps", type=int, default=5,
                    help="extra steps (after the timed region, not part of `value`) whose batches come from pinned host memory "
                         "through the device input pipeline: reported as `h2d_inclusive` (0 = skip)")
    ap.add_argument("--plain-steps", type=int, default=3,
                    help="extra steps after the timed region WITHOUT the per-launch HIP-event instrumentation of clipa_amd.ops: "
                         "reported as `uninstrumented_ms_per_step` next to `ms_per_step` (0 = skip)")
    ap.add_argument("--cpu-sample", type=int, default=8, help="pairs per CPU-baseline step")
    ap.add_argument("--cpu-timeout", type=int, default=240)
    ap.add_argument("--fp8-line-steps", type=int, default=3,
                    help="one GPU, bf16 run: BEFORE this process touches the GPU, the same workload is measured in fp8 (BASELINE configs[3]'s "
                         "precision; create_model(precision='fp8')) by a child `bench.py --precision fp8` with this many timed steps and "
                         "reported as `fp8_same_workload`, never part of `value` (0 = skip)")
    ap.add_argument("--fp8-line-timeout", type=int, default=300)
    args = ap.parse_args()

    if (args.gpus > 1 or os.environ.get("CLIPA_BENCH_FORCE_DIST") == "1") and "WORLD_SIZE" not in os.environ:
        # no launcher around us: become one (same arguments, one rank per GPU); never returns.  (CLIPA_BENCH_FORCE_DIST=1: the
        # one-GPU rehearsal of the multi-rank path goes through the same entry)
        argv = self_launch_argv(args.gpus, sys.argv[1:])
        print("bench.py: --gpus %d without a launcher, re-executing as: %s" % (args.gpus, " ".join(argv)), file=sys.stderr, flush=True)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("OMP_NUM_THREADS", "8")
        os.execv(argv[0], argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}; pass --nproc-per-node {args.gpus}", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print(f"bench.py (rank {rank} of {world}): no GPU visible - the MI355X engine has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    fp8_line = None
    if world == 1 and os.environ.get("CLIPA_BENCH_FORCE_DIST") != "1" and args.precision != "fp8" and args.fp8_line_steps > 0:
        fp8_line = fp8_child_line(args)
    if args.alloc_conf:
        torch.cuda.memory._set_allocator_settings(args.alloc_conf)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    # CLIPA_BENCH_FORCE_DIST=1 (under torchrun --nproc-per-node 1) walks the multi-rank code path - RCCL process
    # group, DDP wrapper, barriers, max-over-ranks reduction - on a single GPU: a pre-flight for the scaling runs
    dist_on = world > 1 or os.environ.get("CLIPA_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL prints a version banner ("RCCL version : ... Librccl path : ...") on the C stdout when its first communicator comes
        # up - in front of the ONE JSON line this script owes its caller.  File descriptor 1 points at stderr while the group and
        # its communicator are created (a first all-reduce forces the lazy part), then C's buffer is flushed and stdout restored.
        import ctypes
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=dev)       # RCCL over xGMI
            dist.all_reduce(torch.zeros(1, device=dev))
            torch.cuda.synchronize()
        finally:
            try:
                ctypes.CDLL(None).fflush(None)
            except Exception:
                pass
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    import clipa_amd
    from clipa_amd import ops
    from clipa_amd.optim import AdamW

    cfg = clipa_amd.get_model_config(args.model)
    cfg["vision_cfg"]["image_size"] = args.image_size
    cfg["text_cfg"]["context_length"] = args.ctx
    torch.manual_seed(0)                                       # same init on every rank (main.py:231)
    model = clipa_amd.create_model(args.model, precision=args.precision,
                                   device=dev, force_image_size=args.image_size, output_dict=True)
    if args.ctx != model.positional_embedding.shape[0]:
        model.positional_embedding = torch.nn.Parameter(model.positional_embedding[:args.ctx].clone())
    model.set_grad_checkpointing(True)                         # every reference GPU script passes --grad-checkpointing
    if args.fp8_predicted_scales:
        model.visual.transformer.fp8_predicted_scales = model.transformer.fp8_predicted_scales = True
    model.unpad_text = bool(args.unpad_text)
    named = list(model.named_parameters())
    exclude = lambda n, p: p.ndim < 2 or "bn" in n or "ln" in n or "bias" in n or "logit_scale" in n   # main.py:311-316
    groups = [{"params": [p for n, p in named if exclude(n, p) and p.requires_grad], "weight_decay": 0.},
              {"params": [p for n, p in named if not exclude(n, p) and p.requires_grad], "weight_decay": 0.2}]
    step_model = model
    if args.optimizer == "sharded":
        from clipa_amd.zero import ShardedAdamW
        opt = ShardedAdamW(groups, lr=5e-4, betas=(0.9, 0.95), eps=1e-6, clamp=(model.logit_scale, 0.0, math.log(100)),
                           exchange=args.exchange, force_collectives=dist_on)
    else:
        opt = AdamW(groups, lr=5e-4, betas=(0.9, 0.95), eps=1e-6,
                    clamp=(model.logit_scale, 0.0, math.log(100)))      # train.py:285-286, fused into the update kernel
        if dist_on:
            step_model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], static_graph=True)
    loss_fn = clipa_amd.ClipLoss(local_loss=True, gather_with_grad=True, cache_labels=True, rank=rank, world_size=world)
    loss_fn.bind(model)      # image-feature all-gather starts right after the image tower, on a side stream

    B = args.batch
    from clipa_amd.data import synthetic_batch
    # uint8 NHWC (channels_last) images + int64 token ids, resident in HBM before the timed region
    images, texts = synthetic_batch(B, args.image_size, args.ctx, cfg["text_cfg"]["vocab_size"], seed=1234 + rank, device=dev)

    A = max(1, args.accum_freq)
    if B % A != 0:
        print(f"bench.py: --batch {B} is not a multiple of --accum-freq {A}", file=sys.stderr)
        sys.exit(2)

    lens_cache = {}

    def host_lengths(tx):
        """unpad_text: the caption lengths as the loader's host side knows them (read back ONCE per resident batch here) - the
        engine then needs no device-to-host copy inside the step."""
        if not model.unpad_text:
            return None
        key = (tx.data_ptr(), tx.shape[0])
        if key not in lens_cache:
            lens_cache[key] = (tx.argmax(-1) + 1).cpu().tolist()      # a list: DDP moves tensor arguments of forward to the device
        return lens_cache[key]

    def step(images=images, texts=texts):
        opt.zero_grad(set_to_none=True)                        # (the sharded optimizer zeroes its flat buffers instead)
        if A == 1:
            out = step_model(images, texts, text_lengths=host_lengths(texts))
            loss = loss_fn(**out, output_dict=True)["contrastive_loss"]
            loss.backward()
        else:
            # training/train.py:216-256: cache every micro-batch's features without grad, then re-forward each micro-batch
            # with grad, splice its features into the cached lists, take the FULL-batch loss and back-propagate
            chunks = list(zip(images.chunk(A), texts.chunk(A)))
            feats = {"image_features": [], "text_features": []}
            with torch.no_grad():
                for im, tx in chunks:
                    o = step_model(im, tx, text_lengths=host_lengths(tx))
                    for k in feats:
                        feats[k].append(o[k])
            import contextlib
            for j, (im, tx) in enumerate(chunks):
                # gradients are exchanged once, by the last micro-batch's backward (DDP.no_sync / ShardedAdamW.no_sync); the
                # reference's loop all-reduces on every backward, which is correct with both but moves A x the bytes
                defer = contextlib.nullcontext() if (j == A - 1 or not dist_on) else \
                    (opt.no_sync() if args.optimizer == "sharded" else step_model.no_sync())
                with defer:
                    o = step_model(im, tx, text_lengths=host_lengths(tx))
                    scale = o.pop("logit_scale")
                    inputs = {k: torch.cat(v[:j] + [o[k]] + v[j + 1:]) for k, v in feats.items()}
                    loss = loss_fn(**inputs, logit_scale=scale, output_dict=True)["contrastive_loss"]
                    loss.backward()
        opt.step()                                             # AdamW + logit_scale clamp, multi-tensor kernels
        return loss

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    # Activation policy: every block recomputes in backward (the reference's --grad-checkpointing) except the
    # first `keep` blocks of each tower, which keep their GEMM / attention outputs in the HBM that is left over.
    # "auto" measures the peak of one all-recompute step and spends ~85 % of the remaining HBM.
    # the upgraded tier: "light8" (MLP pre-activation kept as e4m3 bytes, 4 D bytes per token on top of the medium set); since
    # round 6 the fp8 GEMM has the epilogues for it too
    use_l8 = not args.no_light8

    tensor_plan = None                 # bf16 engines under --keep-blocks auto: per-tensor counts (plan_keep_tensors)

    def set_keep(kv, kt, mv=0, mt=0):
        vtr, ttr = model.visual.transformer, model.transformer
        if tensor_plan is not None:
            vtr.keep_blocks = ttr.keep_blocks = vtr.light8_blocks = ttr.light8_blocks = vtr.medium_blocks = ttr.medium_blocks = 0
            vtr.keep_counts, ttr.keep_counts = dict(tensor_plan["v"]), dict(tensor_plan["t"])
            return
        vtr.keep_counts, ttr.keep_counts = ({"h8": 0, "h": 0, "a": 0, "x1": 0, "qkv": 0} for _ in range(2))
        if use_l8:
            vtr.light8_blocks, ttr.light8_blocks, vtr.keep_blocks, ttr.keep_blocks = kv, kt, 0, 0
        else:
            vtr.keep_blocks, ttr.keep_blocks, vtr.light8_blocks, ttr.light8_blocks = kv, kt, 0, 0
        vtr.medium_blocks, ttr.medium_blocks = mv, mt

    keep_v = keep_t = med_v = med_t = 0
    backoffs = 0
    second_pass = None                  # what the second planning pass (from the planned step's own free HBM) added, bytes
    exact_plan_fn = None                # budget -> plan restricted to bit-exact tensors (for `value_exact_tiers`)
    final_budget = None
    L_img = (args.image_size // cfg["vision_cfg"]["patch_size"]) ** 2 + 1
    vt, tt = model.visual.transformer, model.transformer
    layers = {"v": cfg["vision_cfg"]["layers"], "t": cfg["text_cfg"]["layers"]}
    # towers whose last block runs on the pooled rows only (engine.LastBlockFn): class-token image towers, every text tower
    image_pruned = model.visual._pool_mode() == ops.POOL_FIRST
    text_pruned = True
    pruned_last = tuple(tw for tw, on in (("v", image_pruned), ("t", text_pruned)) if on)
    warm = args.warmup
    total_mem = torch.cuda.get_device_properties(dev).total_memory
    # several ranks: what the exchanges need on top of what the planner's collective-free trial step measures, reserved explicitly;
    # the 1-rank pre-flight (CLIPA_BENCH_FORCE_DIST=1) reserves for the world size the run is a rehearsal of (default 8)
    assume_world = int(os.environ.get("CLIPA_BENCH_ASSUME_WORLD", "8")) if (dist_on and world == 1) else world
    n_params = sum(p.numel() for p in model.parameters() if p.requires_grad)
    reserve = dist_reservation(assume_world if dist_on else 1, B, cfg["embed_dim"], n_params, 2 if args.precision != "amp_bf16" else 4)
    reserve_total = sum(reserve.values())
    if args.keep_blocks == "auto":
        torch.cuda.reset_peak_memory_stats(dev)
        step()                                   # one extra untimed all-recompute step, only to measure its peak
        torch.cuda.synchronize()                 # (under DDP this peak already contains the reducer's buckets)
        peak = torch.cuda.max_memory_allocated(dev)
        # several ranks: the same fraction as alone less an explicit reservation for what the collective-free trial below never
        # allocates (dist_reservation: RCCL's buffers, DDP's buckets, the gathered features and their gradient); the vote after
        # the trial backs every rank off together if it was too much
        frac = args.keep_fraction if args.keep_fraction is not None else 0.97
        budget0 = int(frac * (total_mem - peak)) - (6 << 30) - reserve_total
        if dist_on:
            budget0 = agree_budget(budget0, dev)
        # Spend the budget where a byte saves the most step time: tensor by tensor for the bf16 engines (plan_keep_tensors), whole
        # tiers for fp8 / --tier-plan (plan_keep: medium first, then its upgrades).
        mv_b, mt_b = vt.medium_keep_bytes(B // A * L_img), tt.medium_keep_bytes(B // A * args.ctx)     # per micro-batch
        lv_b, lt_b = (vt.light8_keep_bytes(B // A * L_img), tt.light8_keep_bytes(B // A * args.ctx)) if use_l8 else \
            (vt.light_keep_bytes(B // A * L_img), tt.light_keep_bytes(B // A * args.ctx))

        per_tensor = not args.tier_plan
        tok_v, tok_t = B // A * L_img, B // A * args.ctx
        tb = {(tw, n): tr.tensor_keep_bytes(tok, n) for tw, tr, tok in (("v", vt, tok_v), ("t", tt, tok_t)) for n in ("h8", "h", "a", "x1", "qkv")}
        if args.precision == "fp8":
            order = KEEP_VALUE_MS_PER_GB_FP8 if use_l8 else KEEP_VALUE_MS_PER_GB_FP8_EXACT
        else:
            order = KEEP_VALUE_MS_PER_GB if use_l8 else KEEP_VALUE_MS_PER_GB_EXACT
        if per_tensor and args.precision != "fp8":
            exact_plan_fn = lambda budget: plan_keep_tensors(budget, layers, tb, KEEP_VALUE_MS_PER_GB_EXACT, pruned_last)

        def plan(budget):
            nonlocal tensor_plan
            if per_tensor:
                tensor_plan = plan_keep_tensors(budget, layers, tb, order, pruned_last)
                return 0, 0, 0, 0
            return plan_keep(budget, layers["v"], layers["t"], mv_b, mt_b, lv_b, lt_b)

        # Trial step under the plan; an allocator-fragmentation OOM shrinks the budget instead of failing the run.  With
        # several ranks a rank must never run out of memory INSIDE a collective (its peers would wait for it forever), so
        # the trial there is collective-free: forward + backward of the unwrapped model against a one-rank loss, the early
        # gather unbound, the sharded optimizer's exchange inside no_sync(), no optimizer step (the reducer of DDP ignores
        # gradient hooks of a forward it did not see) - the same activation memory as the real step.  Then every rank votes
        # (MIN all-reduce): all ranks keep the same blocks, and all of them back off if any of them could not fit the plan.
        def trial():
            if not dist_on:
                step()
                torch.cuda.synchronize()
                return
            import contextlib
            partner, model._gather_partner = model._gather_partner, None
            try:
                with (opt.no_sync() if args.optimizer == "sharded" else contextlib.nullcontext()):
                    local_loss = clipa_amd.ClipLoss()
                    for im, tx in zip(images.chunk(A), texts.chunk(A)):
                        out = model(im, tx)
                        local_loss(**out, output_dict=True)["contrastive_loss"].backward()
                torch.cuda.synchronize()
            finally:
                model._gather_partner = partner
                opt.zero_grad(set_to_none=True)

        def fits(budget):
            """Plan `budget`, run the trial step under it -> True if every rank got through."""
            nonlocal keep_v, keep_t, med_v, med_t
            keep_v, keep_t, med_v, med_t = plan(budget)
            set_keep(keep_v, keep_t, med_v, med_t)
            ok = 1
            try:
                trial()
            except torch.OutOfMemoryError:
                ok = 0
                opt.zero_grad(set_to_none=True)
                torch.cuda.empty_cache()
            if dist_on:
                vote = torch.tensor([ok], device=dev, dtype=torch.int32)
                dist.all_reduce(vote, op=dist.ReduceOp.MIN)
                ok = int(vote.item())
            return bool(ok)

        backoffs = 0
        for attempt in range(5):
            if fits(budget0):
                final_budget = budget0
                break
            backoffs += 1
            budget0 -= 12 << 30
        if final_budget is None:
            keep_v = keep_t = med_v = med_t = 0        # five plans did not fit: fall back to the all-recompute step that did
            tensor_plan = None
        elif not args.no_second_pass:
            # Second planning pass (the budget above comes from the ALL-RECOMPUTE step's peak, whose per-block transients
            # are about twice the planned step's): what the device still has free after the trial step of the plan - memory no
            # allocator segment covers - goes to the planner too, less a margin; one more trial decides, and a plan that does not
            # fit is taken back (the first plan is re-run, so the allocator is in the state the timed region will find).
            free_dev, _ = torch.cuda.mem_get_info(dev)
            sp_frac = args.second_pass_fraction if args.second_pass_fraction is not None else 0.6
            extra = int(sp_frac * free_dev) - (2 << 30) - reserve_total
            if dist_on:
                extra = agree_budget(extra, dev)
            snapshot = lambda t4: (tuple(int(v) for v in t4), json.dumps(tensor_plan, sort_keys=True))
            before = snapshot((keep_v, keep_t, med_v, med_t))
            changed = extra > 0 and snapshot(plan(final_budget + extra)) != before
            if changed and fits(final_budget + extra):
                second_pass = extra
                final_budget += extra
            elif changed:
                backoffs += 1
                if not fits(final_budget):             # (cannot happen short of fragmentation: it fitted a moment ago)
                    keep_v = keep_t = med_v = med_t = 0
                    tensor_plan, final_budget = None, None
            else:
                keep_v, keep_t, med_v, med_t = plan(final_budget)
    else:
        vals = [int(v) for v in args.keep_blocks.split(",")]
        keep_v, keep_t = vals[0], vals[1]
        med_v, med_t = (vals[2], vals[3]) if len(vals) >= 4 else (0, 0)
    set_keep(int(keep_v), int(keep_t), int(med_v), int(med_t))
    from clipa_amd import loss as loss_mod

    pressure_backoffs = 0

    def memory_pressure():
        """Several ranks: a rank that runs out of memory INSIDE a collective leaves its peers waiting for ever, so the decision to
        back off is taken between steps, by all ranks together: True if any rank's allocator had to give cached blocks back to the
        driver during the warm-up (num_alloc_retries moved: the step before an OutOfMemoryError) or sees < 1.5 GiB of free HBM."""
        st = torch.cuda.memory_stats(dev)
        free_dev, _ = torch.cuda.mem_get_info(dev)
        mine = torch.tensor([int(st.get("num_alloc_retries", 0)) - memory_pressure.seen, -int(free_dev)], device=dev, dtype=torch.int64)
        memory_pressure.seen = int(st.get("num_alloc_retries", 0))
        dist.all_reduce(mine, op=dist.ReduceOp.MAX)
        return int(mine[0]) > 0 or -int(mine[1]) < (3 << 29)
    memory_pressure.seen = int(torch.cuda.memory_stats(dev).get("num_alloc_retries", 0))

    def timed_region():
        nonlocal pressure_backoffs, final_budget, keep_v, keep_t, med_v, med_t
        for attempt in range(3):
            for _ in range(warm):
                step()
            fence()
            if not (dist_on and warm > 0 and final_budget is not None and attempt < 2 and memory_pressure()):
                break
            # every rank takes the same smaller plan (the budgets were agreed, so are the 12 GiB steps) and warms up again
            pressure_backoffs += 1
            final_budget -= 12 << 30
            keep_v, keep_t, med_v, med_t = plan(final_budget)
            set_keep(int(keep_v), int(keep_t), int(med_v), int(med_t))
            opt.zero_grad(set_to_none=True)
            torch.cuda.empty_cache()
        if dist_on:
            loss_mod.wait_timing_start()
        ops.profile_start(detail=args.shapes)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        fence()
        return time.perf_counter() - t0, ops.profile_stop(), loss

    # (single process: should the allocator run out INSIDE the warm-up / timed steps after the trial steps got through - it has
    # not happened - the plan backs off once more and the region starts over; `value` is always a complete region of K steps)
    for region_attempt in range(3):
        try:
            elapsed, prof, loss = timed_region()
            break
        except torch.OutOfMemoryError:
            if dist_on or final_budget is None or region_attempt == 2:
                raise
            ops.profile_stop()
            opt.zero_grad(set_to_none=True)
            torch.cuda.empty_cache()
            backoffs += 1
            final_budget -= 12 << 30
            keep_v, keep_t, med_v, med_t = plan(final_budget)
            set_keep(int(keep_v), int(keep_t), int(med_v), int(med_t))
    reserved_timed = torch.cuda.max_memory_reserved(dev)
    retries_timed = int(torch.cuda.memory_stats(dev).get("num_alloc_retries", 0))     # through the timed region (the extra regions
    gather_wait_ms = loss_mod.wait_timing_stop() / args.steps if dist_on else 0.0      # below re-shape the allocator's segments)
    last_loss = float(loss)
    if dist_on:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax)
    if not math.isfinite(last_loss):
        print(f"bench.py: non-finite loss {last_loss}", file=sys.stderr)
        sys.exit(3)
    # the same step without the ~1100 HIP events per step that feed `roofline` / `kernels` (clipa_amd.ops.profile_start):
    # quantifies what the instrumentation costs the timed region.  Never part of `value`.
    plain_ms = None
    if args.plain_steps > 0:
        fence()
        tp = time.perf_counter()
        for _ in range(args.plain_steps):
            step()
        fence()
        ep = time.perf_counter() - tp
        if dist_on:
            tm = torch.tensor([ep], device=dev, dtype=torch.float64)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            ep = float(tm)
        plain_ms = round(1e3 * ep / args.plain_steps, 2)

    # PCIe-inclusive rate (the reference's batch_time includes the H2D copy, train.py:187-189): the same step
    # fed from pinned host memory with staged uint8 NHWC images through DevicePrefetcher (copy stream, double buffered) +
    # the on-device RandomResizedCrop / ColorJitter / Grayscale of the reference GPU recipe (scripts/exp/gpu/vit_l16/i37_t8_pretrain.sh:13).
    # Never part of `value`.  Runs first among the extra regions: the allocator is still in the timed region's state (behind the
    # exact-tier region, which re-shapes the segments under another plan, it ran into allocator retries: 1298 instead of 2440 pairs/s).
    h2d = None
    if args.h2d_steps > 0:
        try:
            from clipa_amd.data import DeviceAugment, DevicePrefetcher
            stage = (args.image_size * 8 // 7 + 15) // 16 * 16                  # 224 -> 256, 84 -> 96
            gh = torch.Generator().manual_seed(99 + rank)
            base = torch.randint(0, 256, (64, stage, stage, 3), generator=gh, dtype=torch.uint8)
            pool = [base.roll(k, 0).repeat((B + 63) // 64, 1, 1, 1)[:B].contiguous().pin_memory() for k in range(2)]
            texts_host = texts.cpu().pin_memory()
            n_h2d = args.h2d_steps
            loader = ((pool[i % 2], texts_host) for i in range(n_h2d + 1))
            aug = DeviceAugment(args.image_size, scale=(0.4, 1.0), color_jitter=(0.32, 0.32, 0.32, 0.08), color_jitter_prob=0.8,
                                gray_scale_prob=0.2, seed=7 + rank)
            feed = iter(DevicePrefetcher(loader, dev, transform=aug, depth=2))
            step(*next(feed))                                                    # warm-up of the pipeline itself
            fence()
            th = time.perf_counter()
            for _ in range(n_h2d):
                loss_h = step(*next(feed))
            fence()
            eh = time.perf_counter() - th
            if dist_on:
                tm = torch.tensor([eh], device=dev, dtype=torch.float64)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                eh = float(tm)
            ops.check_token_ids(wait=True)
            h2d = {"value": round(B * world * n_h2d / eh, 2), "unit": "pairs/s", "ms_per_step": round(1e3 * eh / n_h2d, 2),
                   "steps": n_h2d, "loss": round(float(loss_h), 4),
                   "input": f"uint8 NHWC {stage}x{stage} staged images in pinned host memory -> DevicePrefetcher (depth 2, copy "
                            f"stream) -> on-device RandomResizedCrop(scale 0.4-1, bicubic) + ColorJitter(0.32, 0.32, 0.32, 0.08; p=0.8) + Grayscale(p=0.2) -> {args.image_size} px"}
            del pool, feed
        except (RuntimeError, torch.OutOfMemoryError) as e:                      # the headline line must survive
            h2d = {"value": None, "error": str(e)[:200]}

    # The same step under the keep plan restricted to BIT-EXACT tensors (no e4m3 pre-activation; VERDICT r4 next #1c): its
    # gradients are those of the all-recompute bf16 step bit for bit (tests/test_model_gpu.py::
    # test_headline_configuration_matches_oracle), where the default plan's carry the e4m3 rounding of the kept pre-activations
    # (cosine >= 0.995 against them, the engine's stated 0.99 against the fp32 oracle - same test).  Same activation budget less
    # 8 GiB: a block that re-runs c_fc holds its bf16 pre-activation and the GELU output together.  Never part of `value`.
    exact = None
    if args.exact_steps > 0 and exact_plan_fn is not None and use_l8 and tensor_plan is not None and final_budget is not None \
            and not dist_on:
        default_plan = tensor_plan
        try:
            tensor_plan = exact_plan_fn(final_budget - (8 << 30))
            set_keep(0, 0)
            step()                                                           # warm-up: the allocator re-shapes its segments
            fence()
            te = time.perf_counter()
            for _ in range(args.exact_steps):
                loss_e = step()
            fence()
            ee = time.perf_counter() - te
            exact = {"value": round(B * world * args.exact_steps / ee, 2), "unit": "pairs/s", "ms_per_step": round(1e3 * ee / args.exact_steps, 2),
                     "steps": args.exact_steps, "loss": round(float(loss_e), 4),
                     "plan": "kept tensors (image/text blocks) " + ", ".join(f"{n} {tensor_plan['v'][n]}/{tensor_plan['t'][n]}" for n in ("a", "x1", "qkv", "h")),
                     "note": "keep plan restricted to bit-exact tensors (bf16 as the forward wrote them; no e4m3 pre-activation): "
                             "gradients identical to the all-recompute bf16 step"}
        except (RuntimeError, torch.OutOfMemoryError) as e:
            opt.zero_grad(set_to_none=True)
            exact = {"value": None, "error": str(e)[:200]}
        tensor_plan = default_plan
        torch.cuda.empty_cache()
        set_keep(0, 0)

    # The engine's unpad_text knob (profiles/NOTEBOOK.md 7d): same inputs, same features / loss / gradients, the causal text tower on the
    # tokens up to each caption's EOT (SURVEY 8d's caption lengths ~ N(20, 8): about a quarter of the 77 positions).  Never
    # part of `value`, which stays on the reference's schedule unless --unpad-text asks otherwise.
    unpad = None
    # Several ranks: toggling the knob changes the autograd graph, which DistributedDataParallel(static_graph=True) records in its
    # first iteration - so this region (the LAST one that steps the model) drops the
    # wrapper and wraps the model again with the knob set, as a trainer would set it before its wrap.
    def unpad_region():
        nonlocal step_model
        rewrapped = False
        model.unpad_text = True
        try:
            if dist_on and args.optimizer == "adamw":
                import gc
                step_model = None
                gc.collect()                                                     # the old reducer unhooks itself when it dies
                step_model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], static_graph=True)
                rewrapped = True
            step()                                                           # warm-up (allocator, index structures)
            fence()
            tu = time.perf_counter()
            for _ in range(args.unpad_steps):
                loss_u = step()
            fence()
            eu = time.perf_counter() - tu
            if dist_on:
                tm = torch.tensor([eu], device=dev, dtype=torch.float64)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                eu = float(tm)
            lens = (texts.argmax(-1) + 1).float()
            res = {"value": round(B * world * args.unpad_steps / eu, 2), "unit": "pairs/s", "ms_per_step": round(1e3 * eu / args.unpad_steps, 2),
                   "steps": args.unpad_steps, "loss": round(float(loss_u), 4),
                   "text_positions_processed": round(float(lens.sum()) / (B * args.ctx), 4),
                   "note": "model.unpad_text = True: the causal text tower runs on the tokens up to each caption's EOT, packed; features, "
                           "loss and gradients are those of the padded step (tests/test_model_gpu.py::test_unpadded_text_tower_changes_nothing)"
                           + ("; DistributedDataParallel wrapper rebuilt with the knob set" if rewrapped else "")}
        except (RuntimeError, torch.OutOfMemoryError) as e:
            res = {"value": None, "error": str(e)[:200]}
        model.unpad_text = False
        return res

    if args.unpad_steps > 0 and not args.unpad_text:
        unpad = unpad_region()

    mine = {"rank": rank, "reserved_for_exchanges_gb": {k: round(v / 2**30, 2) for k, v in reserve.items()},
            "peak_hbm_gb": round(torch.cuda.max_memory_allocated(dev) / 2**30, 1),
            "peak_reserved_gb": round(torch.cuda.max_memory_reserved(dev) / 2**30, 1),
            "alloc_retries": int(torch.cuda.memory_stats(dev).get("num_alloc_retries", 0)),
            "kernel_ms_per_step": round(sum(v["ms"] for k, v in prof.items() if "|" not in k and "#" not in k) / args.steps, 2),
            "gather_wait_ms_per_step": round(gather_wait_ms, 3)}
    per_rank = [mine]
    if dist_on:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        pairs_s = B * world * args.steps / elapsed
        gf = train_gflop_per_pair(cfg, args.image_size, args.ctx)
        gf_pruned = pruned_gflop_per_pair(cfg, args.image_size, args.ctx, image_pruned, text_pruned)
        mfu_peak = PEAK_FP8_TFLOPS if args.precision == "fp8" else PEAK_BF16_TFLOPS
        # roofline of the DOMINANT kernel of this step (most time among the MFMA-bound GEMMs; gemm_nt for the bf16 headline)
        dom = max(ROOFLINE_KERNELS, key=lambda k: prof.get(k, {}).get("ms", 0.0))
        if args.precision != "fp8":
            dom = "gemm_nt"
        peak, dom_desc = ROOFLINE_KERNELS[dom]
        nt = prof.get(dom, {"launches": 0, "ms": 0.0, "work": 0.0, "bytes": 0.0})
        achieved = nt["work"] / (nt["ms"] * 1e-3) / 1e12 if nt["ms"] > 0 else 0.0
        # PMC traffic (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 passes) cannot be collected inside this process: it is read
        # from profiles/traffic.json, which names the kernel sources it was measured on - and is reported only while they are
        # unchanged (otherwise null + the reason)
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and dom == "gemm_nt":
            try:
                tj = json.load(open(tpath))
                wl = tj.get("workload")
                this_wl = {"model": args.model, "image_size": args.image_size, "ctx": args.ctx, "batch": B, "precision": args.precision}
                if wl is not None and wl != this_wl:
                    traffic_source = f"profiles/traffic.json was measured on {wl}, not on this workload"
                elif tj.get("kernel_source_sha16") == kernel_source_sha16():
                    traffic = tj.get("gemm_nt_hbm_bytes_per_launch")
                    traffic_source = f"profiles/traffic.json ({tj.get('measured', 'PMC passes')}; kernel sources {tj.get('kernel_source_sha16')})"
                else:
                    traffic_source = (f"stale: profiles/traffic.json was measured on kernel sources {tj.get('kernel_source_sha16')}, "
                                      f"this build is {kernel_source_sha16()}")
            except Exception:
                traffic = None
        line = {
            "metric": "image-text pairs/sec (whole job), full training step",
            "value": round(pairs_s, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 2), "uninstrumented_ms_per_step": plain_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp8" if args.precision == "fp8" else "bf16", "data": "synthetic",
            "config": {"workload": f"{args.model}@{args.image_size} + text-{args.ctx}, local batch {B}, "
                                   f"global batch {B * world}, " + (f"accum_freq {A} (feature cache: +1 forward per pair), " if A > 1 else "") +
                                   f"InfoNCE local_loss+gather_with_grad, AdamW, " + ("fp8 with predicted row scales of the MLP tensors, " if args.fp8_predicted_scales else "") + ("text tower on the tokens up to EOT (unpad_text), " if args.unpad_text else "") +
                                   ("block recompute except kept tensors (image/text blocks) " + ", ".join(f"{n} {tensor_plan['v'][n]}/{tensor_plan['t'][n]}" for n in (("h8",) if use_l8 else ()) + ("a", "x1", "qkv") + (() if use_l8 else ("h",))) if tensor_plan is not None else f"block recompute except {int(keep_v)}+{int(keep_t)} {'light8' if use_l8 else 'light'}-kept and {int(med_v)}+{int(med_t)} medium-kept (image+text) blocks"), "precision": args.precision, "parallelism": f"dp{world}" + ("" if args.optimizer == "adamw" else f" zero1/{args.exchange}"),
                       "global_batch": B * world, "train_gflop_per_pair": round(gf, 2)},
            # executed model FLOPs (the reference count less the dead rows of the pruned last blocks) against the peak of the
            # precision the block GEMMs run in; `mfu_reference_flops` = the same rate priced at the reference's FLOP count
            "model_flops_util": round(pairs_s / world * (gf - gf_pruned) / 1e3 / mfu_peak, 4),
            "mfu_reference_flops": round(pairs_s / world * gf / 1e3 / mfu_peak, 4),
            "mfu_peak_tflops": mfu_peak, "executed_gflop_per_pair": round(gf - gf_pruned, 2),
            "alloc_conf": args.alloc_conf, "loss": round(last_loss, 4), "peak_hbm_gb": round(torch.cuda.max_memory_allocated(dev) / 2**30, 1),
            "peak_reserved_gb": round(torch.cuda.max_memory_reserved(dev) / 2**30, 1), "peak_reserved_gb_through_timed_region": round(reserved_timed / 2**30, 1),
            "alloc_retries": retries_timed,
            "alloc_retries_incl_extra_regions": int(torch.cuda.memory_stats(dev).get("num_alloc_retries", 0)),
            "keep_plan_backoffs": backoffs, "memory_pressure_backoffs": pressure_backoffs,
            "keep_plan_second_pass_gb": None if second_pass is None else round(second_pass / 2**30, 1),
            "total_hbm_gb": round(total_mem / 2**30, 1),
            # what the step spends outside this library's kernels (rank 0): at one rank host gaps + optimizer glue, with several
            # ranks additionally the EXPOSED part of the exchanges (gradient all-reduce tail, feature gathers); the wait of
            # the compute stream for the feature gathers is event-timed separately (per_rank.gather_wait_ms_per_step)
            "non_kernel_ms_per_step": round(ms - mine["kernel_ms_per_step"], 2),
            "per_rank": per_rank,
            "roofline": {"bound": "mfma", "kernel": dom_desc, "achieved": round(achieved, 1),
                         "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                         "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes_per_launch": round(nt.get("bytes", 0.0) / max(nt["launches"], 1)),
                         "launches": nt["launches"],
                         "avg_launch_ms": round(nt["ms"] / max(nt["launches"], 1), 4),
                         # the family by epilogue (epi0 plain / bias, 1 GELU, 1+pre8 GELU + e4m3 copy, 2 residual add, 3 GELU-backward,
                         # 3,aux8 the same from e4m3 bytes, 3,aux8+act ... also writing the re-materialised GELU output - an HBM-bound
                         # pass folded into that launch): TFLOP/s and ms per step
                         # rocprofv3 lists every epilogue instantiation as a kernel of its own; the one with the most time (plain / bias):
                         "dominant_instantiation": (lambda v: None if v is None else {
                             "kernel": "gemm_nta_kernel<CLIPA_EPI_NONE> (plain / bias launches)", "achieved": round(v["work"] / max(v["ms"], 1e-9) / 1e9, 1),
                             "frac": round(v["work"] / max(v["ms"], 1e-9) / 1e9 / peak, 4), "launches": v["launches"]})(prof.get("gemm_nt#epi0")) if dom == "gemm_nt" else None,
                         "by_epilogue": {k[8:]: {"tflops": round(v["work"] / max(v["ms"], 1e-9) / 1e9, 1), "ms_per_step": round(v["ms"] / args.steps, 2),
                                                  "launches_per_step": v["launches"] // args.steps}
                                         for k, v in sorted(prof.items()) if k.startswith("gemm_nt#")} if dom == "gemm_nt" else None},
            "fp8_same_workload": None if fp8_line is None else {**fp8_line, "over_value": (round(fp8_line["value"] / pairs_s, 3) if fp8_line.get("value") else None)},
            "value_exact_tiers": exact,
            "h2d_inclusive": h2d,
            "unpadded_text": unpad,
            "kernels": {k: kernel_entry(v, args.steps) for k, v in prof.items() if "|" not in k and "#" not in k},
        }
        if args.shapes:
            for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
                if "|" in k:
                    print(f"SHAPE {k:48s} n={v['launches'] // args.steps:4d}/step {v['ms'] / args.steps:8.2f} ms/step "
                          f"{v['work'] / max(v['ms'], 1e-9) / 1e9:7.1f} TF/s", file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            # bounded: run in a child process with a wall-clock limit so the bench line is never held hostage
            import subprocess
            threads = usable_cores()
            code = ("import json,sys; sys.path.insert(0, %r); import bench, clipa_amd; "
                    "cfg = clipa_amd.get_model_config(%r); cfg['vision_cfg']['image_size'] = %d; "
                    "cfg['text_cfg']['context_length'] = %d; "
                    "print('CPUBASE ' + json.dumps(bench.cpu_baseline(cfg, %d, %d, %d, %d)))"
                    % (ROOT, args.model, args.image_size, args.ctx, args.image_size, args.ctx, args.cpu_sample, threads))
            try:
                r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=args.cpu_timeout,
                                   env={**os.environ, "HIP_VISIBLE_DEVICES": "", "OMP_NUM_THREADS": str(threads)})
                got = [l for l in r.stdout.splitlines() if l.startswith("CPUBASE ")]
                line["cpu_baseline"] = json.loads(got[-1][8:]) if got else {"value": None, "unit": "pairs/s", "cores": threads,
                                                                           "kind": "port", "sample": "failed: " + r.stderr[-200:]}
            except subprocess.TimeoutExpired:
                line["cpu_baseline"] = {"value": None, "unit": "pairs/s", "cores": threads, "kind": "port",
                                        "sample": f"timed out after {args.cpu_timeout}s ({args.cpu_sample} pairs/step)"}
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
