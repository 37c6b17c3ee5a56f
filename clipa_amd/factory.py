"""Construction surface mirroring clipa_torch/open_clip/factory.py for the plain-CLIP (ViT) branch:
create_model (factory.py:121-259), create_loss (factory.py:262-290), create_model_and_transforms
(factory.py:293-352), load_checkpoint (factory.py:99-118).

Out of scope and rejected loudly: pretrained tags / HF hub download (network), timm / ResNet / CoCa /
HF-text towers, torchscript.  The two image transforms create_model_and_transforms returns are
clipa_amd.transform.image_transform (host-side, Pillow only).
"""
import logging
from typing import Optional, Tuple, Union

import torch

from . import configs
from .loss import ClipLoss
from .model import (CLIP, OPENAI_DATASET_MEAN, OPENAI_DATASET_STD, convert_weights_to_lp, resize_pos_embed,
                    resize_text_pos_embed)


def get_cast_dtype(precision: str):
    """open_clip/model.py:78-86."""
    if precision in ('bf16', 'fp8'):
        return torch.bfloat16
    if precision == 'fp16':
        raise NotImplementedError("clipa_amd: fp16 is not an MI355X engine mode (bf16 MFMA path only)")
    return None


def load_state_dict(checkpoint_path: str, map_location='cpu'):
    checkpoint = torch.load(checkpoint_path, map_location=map_location, weights_only=False)
    state_dict = checkpoint['state_dict'] if isinstance(checkpoint, dict) and 'state_dict' in checkpoint else checkpoint
    if next(iter(state_dict.items()))[0].startswith('module'):
        state_dict = {k[7:]: v for k, v in state_dict.items()}
    return state_dict


def load_checkpoint(model, checkpoint_path, strict=True, interpolation='bicubic', square_resize_only=False):
    """factory.py:110-118: state_dict load with positional-table resize (two-resolution CLIPA schedule)."""
    state_dict = load_state_dict(checkpoint_path)
    resize_pos_embed(state_dict, model, interpolation=interpolation)
    resize_text_pos_embed(state_dict, model)
    return model.load_state_dict(state_dict, strict=strict)


def create_model(
        model_name: str,
        pretrained: Optional[str] = None,
        precision: str = 'fp32',
        device: Union[str, torch.device] = 'cpu',
        jit: bool = False,
        force_quick_gelu: bool = False,
        force_custom_text: bool = False,
        force_patch_dropout: Optional[float] = None,
        force_image_size: Optional[Union[int, Tuple[int, int]]] = None,
        pretrained_image: bool = False,
        pretrained_hf: bool = True,
        cache_dir: Optional[str] = None,
        output_dict: Optional[bool] = None,
        require_pretrained: bool = False,
        pos_embed: str = None,
        interpolation: str = 'bicubic',
        square_resize_only: bool = False,
):
    model_name = model_name.replace('/', '-')
    if isinstance(device, str):
        device = torch.device(device)
    if jit or force_custom_text or pretrained_image:
        raise NotImplementedError("clipa_amd.create_model: jit / custom-text / timm-pretrained are outside the MI355X hot path")
    model_cfg = configs.get_model_config(model_name)
    if model_cfg is None:
        raise RuntimeError(f'Model config for {model_name} not found; available models {configs.list_models()}.')
    if force_quick_gelu:
        model_cfg["quick_gelu"] = True
    if force_patch_dropout is not None:
        model_cfg["vision_cfg"]["patch_dropout"] = force_patch_dropout    # factory.py:182-184
    if force_image_size is not None:
        model_cfg["vision_cfg"]["image_size"] = force_image_size          # factory.py:186-188
    if pos_embed is not None:
        model_cfg["vision_cfg"]["pos_embed"] = pos_embed                  # factory.py:190-192
    cast_dtype = get_cast_dtype(precision)
    model = CLIP(**model_cfg, cast_dtype=cast_dtype)
    if pretrained:
        import os
        if not os.path.exists(pretrained):
            raise RuntimeError(f'Pretrained weights ({pretrained}) must be a local checkpoint path (no network).')
        logging.info(f'Loading pretrained {model_name} weights ({pretrained}).')
        load_checkpoint(model, pretrained, interpolation=interpolation, square_resize_only=square_resize_only)
    elif require_pretrained:
        raise RuntimeError(f'Pretrained weights were required for (model: {model_name}) but not loaded.')
    model.to(device=device)
    if precision in ("fp16", "bf16", "fp8"):
        convert_weights_to_lp(model, dtype=torch.float16 if precision == 'fp16' else torch.bfloat16)
    if precision == "fp8":
        # MI355X engine mode without a reference counterpart (training/params.py:195-200 stops at bf16): bf16 master
        # weights and activations, fp8 (OCP e4m3) operands for the block GEMMs.  See engine._block_forward_fp8.
        model.visual.transformer.fp8 = True
        model.transformer.fp8 = True
    model.visual.image_mean = model_cfg.get("vision_cfg", {}).get('mean', None) or OPENAI_DATASET_MEAN
    model.visual.image_std = model_cfg.get("vision_cfg", {}).get('std', None) or OPENAI_DATASET_STD
    if output_dict and hasattr(model, "output_dict"):
        model.output_dict = True
    return model


def create_loss(args, model=None):
    """factory.py:262-290 (plain ClipLoss branch).  `model` (optional, engine extension): bind the loss to the model so
    that the image-feature all-gather starts under the text tower (ClipLoss.bind)."""
    if getattr(args, "distill", False) or "coca" in getattr(args, "model", "").lower():
        raise NotImplementedError("clipa_amd.create_loss: distillation / CoCa losses are out of scope")
    loss = ClipLoss(
        local_loss=args.local_loss,
        gather_with_grad=args.gather_with_grad,
        cache_labels=True,
        rank=args.rank,
        world_size=args.world_size,
        use_horovod=getattr(args, "horovod", False),
    )
    return loss.bind(model) if model is not None else loss


def create_model_and_transforms(model_name: str, pretrained: Optional[str] = None, precision: str = 'fp32',
                                device: Union[str, torch.device] = 'cpu', jit: bool = False,
                                force_quick_gelu: bool = False, force_custom_text: bool = False,
                                force_patch_dropout: Optional[float] = None,
                                force_image_size: Optional[Union[int, Tuple[int, int]]] = None,
                                pretrained_image: bool = False, pretrained_hf: bool = True,
                                image_mean: Optional[Tuple[float, ...]] = None,
                                image_std: Optional[Tuple[float, ...]] = None, aug_cfg=None,
                                cache_dir: Optional[str] = None, output_dict: Optional[bool] = None,
                                to_float_on_device: bool = False, pos_embed: str = None,
                                interpolation: str = 'bicubic', square_resize_only: bool = False):
    """Same signature and return value as factory.py:293-352: (model, preprocess_train, preprocess_val).  The transforms are
    clipa_amd.transform.image_transform - the reference's open_clip/transform.py:91-214 restated on Pillow alone (torchvision,
    which the reference builds them from, is not part of this image); they run on the host in the loader workers exactly like
    the reference's.  The engine's own device-side pipeline is clipa_amd.data.DeviceAugment."""
    model = create_model(model_name, pretrained, precision=precision, device=device, jit=jit,
                         force_quick_gelu=force_quick_gelu, force_custom_text=force_custom_text,
                         force_patch_dropout=force_patch_dropout, force_image_size=force_image_size,
                         pretrained_image=pretrained_image, pretrained_hf=pretrained_hf, cache_dir=cache_dir,
                         output_dict=output_dict, pos_embed=pos_embed, interpolation=interpolation,
                         square_resize_only=square_resize_only)
    if image_mean is not None:
        model.visual.image_mean = image_mean
    if image_std is not None:
        model.visual.image_std = image_std
    from .transform import image_transform
    image_mean = image_mean or getattr(model.visual, 'image_mean', None)
    image_std = image_std or getattr(model.visual, 'image_std', None)
    preprocess_train = image_transform(model.visual.image_size, is_train=True, mean=image_mean, std=image_std, aug_cfg=aug_cfg,
                                       to_float_on_device=to_float_on_device)
    preprocess_val = image_transform(model.visual.image_size, is_train=False, mean=image_mean, std=image_std,
                                     to_float_on_device=to_float_on_device, interpolation=interpolation,
                                     square_resize_only=square_resize_only)
    return model, preprocess_train, preprocess_val
