"""Evidence that the image-feature all-gather overlaps the text tower.  Runs the bench model under
a 1-rank RCCL group (torchrun --nproc-per-node 1) and records HIP-event timestamps of: end of the image tower, start /
end of the early all-gather on the side stream, start / end of the text tower on the compute stream.
    CLIPA_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
        --master-port 29611 tools/gather_overlap_probe.py [--batch 4096]"""
import argparse, json, os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096)
args = ap.parse_args()
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
import clipa_amd
import clipa_amd.loss as L
from clipa_amd.data import synthetic_batch

model = clipa_amd.create_model("ViT-L-16", precision="bf16", device=dev, force_image_size=224, output_dict=True)
model.set_grad_checkpointing(True)
loss_fn = clipa_amd.ClipLoss(local_loss=True, gather_with_grad=True, rank=0, world_size=1).bind(model)
loss_fn.world_size = 2              # pretend a second rank exists so that CLIP.forward issues the early gather ...
images, texts = synthetic_batch(args.batch, 224, 77, 49408, seed=1, device=dev)

marks = {}
orig_gather = L._all_gather_bf16


def timed_gather(local, world_size, group=None):
    side = L._side_stream(local.device)
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = torch.empty((local.shape[0], local.shape[1]), device=local.device, dtype=local.dtype)
    with torch.cuda.stream(side):
        e0.record(side)
        for _ in range(8):          # ... and move 8x the bytes a rank contributes (one 8-GPU all-gather's worth) on the 1-rank group
            dist.all_gather_into_tensor(out.view(torch.uint8), local.view(torch.uint8))
        e1.record(side)
    marks.setdefault("gather", []).append((e0, e1))
    return out, e1


L._all_gather_bf16 = timed_gather
orig_text = model.encode_text


def timed_text(text, normalize=False):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig_text(text, normalize=normalize)
    e1.record()
    marks.setdefault("text", []).append((e0, e1))
    return r


model.encode_text = timed_text
for it in range(3):
    marks.clear()
    t0 = torch.cuda.Event(enable_timing=True)
    t0.record()
    model.train()
    feats_i = model.encode_image(images, normalize=True)
    loss_fn.early_gather(feats_i)
    loss_fn._pending = None
    feats_t = model.encode_text(texts, normalize=True)
    torch.cuda.synchronize()
    g0, g1 = marks["gather"][0]
    x0, x1 = marks["text"][0]
    rec = {"iter": it, "batch": args.batch,
           "image_tower_end_ms": 0.0,
           "gather_start_ms": round(t0.elapsed_time(g0) - t0.elapsed_time(x0), 3),
           "gather_end_ms": round(t0.elapsed_time(g1) - t0.elapsed_time(x0), 3),
           "text_tower_start_ms": 0.0,
           "text_tower_end_ms": round(x0.elapsed_time(x1), 3),
           "gather_inside_text_tower": bool(t0.elapsed_time(g1) <= t0.elapsed_time(x1))}
    print(json.dumps(rec), flush=True)
dist.destroy_process_group()
