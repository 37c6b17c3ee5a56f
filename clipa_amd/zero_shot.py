"""Zero-shot classification through the MI355X engine.  Mirrors
clipa_torch/training/zero_shot.py:29-90 - `zero_shot_classifier` (per class: encode the prompt templates, L2-normalise,
average, re-normalise, stack into [E, C]) and `run` (encode images, normalise, logits = 100 * f @ classifier, top-1 /
top-5) - on pre-tokenised prompts: the tokenizer itself (open_clip/tokenizer.py) is host-side text processing outside
the hot path.  Every matrix product and normalisation below is a libclipa_hip.so launch (encode_text / encode_image
are the engine's towers, the class logits are one gemm_nt); only topk / eq bookkeeping is torch."""
import torch

from . import ops


def unwrap_model(model):                      # zero_shot.py:22-26
    return model.module if hasattr(model, "module") else model


@torch.no_grad()
def zero_shot_classifier(model, class_token_ids):
    """class_token_ids: int64 [C, T, ctx] (C classes x T prompt templates, tokenised) or a list of [T_c, ctx] tensors.
    -> bf16 classifier [C8, E] (class embeddings as rows, zero-padded to a multiple of 8 classes) and C."""
    m = unwrap_model(model)
    rows = []
    for toks in class_token_ids:
        emb = m.encode_text(toks, normalize=True)                       # [T, E] f32, L2-normalised (model.py:263)
        mean = emb.mean(dim=0, keepdim=True)
        rows.append(ops.l2norm_fwd(mean.contiguous())[0])               # class_embedding /= class_embedding.norm()
    w = torch.cat(rows, dim=0)                                          # [C, E]
    C = w.shape[0]
    C8 = (C + 7) // 8 * 8
    wb = torch.zeros((C8, w.shape[1]), device=w.device, dtype=torch.bfloat16)
    wb[:C] = ops.to_bf16(w)
    return wb, C


@torch.no_grad()
def zero_shot_logits(model, classifier, num_classes, images):
    """100 * normalize(encode_image(images)) @ classifier^T -> f32 [B, C].  images: uint8 / float [B,3,S,S] on the GPU."""
    f = unwrap_model(model).encode_image(images, normalize=True)
    return ops.gemm_nt(ops.to_bf16(f), classifier, alpha=100.0, out_f32=True)[:, :num_classes]


def accuracy(output, target, topk=(1,)):      # zero_shot.py:47-50
    pred = output.topk(max(topk), 1, True, True)[1].t()
    correct = pred.eq(target.view(1, -1).expand_as(pred))
    return [float(correct[:k].reshape(-1).float().sum()) for k in topk]


@torch.no_grad()
def run(model, classifier, num_classes, batches, topk=(1, 5)):
    """batches: iterable of (images, targets) already on the model's device -> accuracies for `topk`."""
    hits = [0.0] * len(topk)
    n = 0
    for images, target in batches:
        logits = zero_shot_logits(model, classifier, num_classes, images)
        for i, a in enumerate(accuracy(logits, target, topk=tuple(min(k, num_classes) for k in topk))):
            hits[i] += a
        n += images.shape[0]
    return [h / max(n, 1) for h in hits]
