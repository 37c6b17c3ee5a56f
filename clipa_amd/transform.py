"""Host-side image transforms with the contract of clipa_torch/open_clip/transform.py:91-214 (`image_transform`), built on
Pillow alone.

Why this exists: the reference's `create_model_and_transforms` (factory.py:293-352) returns `(model, preprocess_train,
preprocess_val)` and `training/main.py:232,360` hands the two transforms to `get_data`; a trainer that does
`import clipa_amd as open_clip` needs them.  The reference builds them from torchvision, which is not part of this image; the
torchvision classes it uses (RandomResizedCrop, Resize, CenterCrop, ColorJitter, Grayscale, ToTensor, PILToTensor,
Normalize) are thin wrappers over the Pillow calls restated here one for one (torchvision/transforms/_functional_pil.py), so
a PIL image goes through the same Pillow kernels in the same order.  When torchvision IS importable the reference's own
`image_transform` can be used instead - nothing here is on the MI355X hot path: the engine's own input pipeline is
`clipa_amd.data.DeviceAugment` (the same operations as HIP kernels on staged uint8 batches).

Not covered (raise): `use_timm` augmentation and `resize_longest_max` (neither is used by the reference's CLIPA scripts).
"""
import math
import random
from dataclasses import asdict, dataclass
from typing import Any, Dict, Optional, Tuple, Union

import numpy as np
import torch

from .data import random_resized_crop_params
from .model import OPENAI_DATASET_MEAN, OPENAI_DATASET_STD


@dataclass
class AugmentationCfg:
    """open_clip/transform.py:16-29."""
    scale: Tuple[float, float] = (0.9, 1.0)
    ratio: Optional[Tuple[float, float]] = None
    color_jitter: Optional[Union[float, Tuple[float, float, float], Tuple[float, float, float, float]]] = None
    interpolation: Optional[str] = None
    re_prob: Optional[float] = None
    re_count: Optional[int] = None
    use_timm: bool = False
    color_jitter_prob: float = None
    gray_scale_prob: float = None


def _pil():
    from PIL import Image, ImageEnhance
    return Image, ImageEnhance


class Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, img):
        for t in self.transforms:
            img = t(img)
        return img

    def __repr__(self):
        return "Compose(" + ", ".join(getattr(t, "__name__", type(t).__name__) for t in self.transforms) + ")"


class RandomResizedCrop:
    """torchvision RandomResizedCrop on a PIL image: get_params box -> crop -> resize((size, size), BICUBIC)."""

    def __init__(self, size, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)):
        self.size = (size, size) if isinstance(size, int) else tuple(size)
        self.scale, self.ratio = tuple(scale), tuple(ratio)

    def __call__(self, img):
        Image, _ = _pil()
        w, h = img.size
        i, j, ch, cw = random_resized_crop_params(h, w, self.scale, self.ratio)
        return img.crop((j, i, j + cw, i + ch)).resize((self.size[1], self.size[0]), Image.BICUBIC)


class Resize:
    """torchvision Resize: an int is the SHORTER edge (aspect preserved, long edge = int(size * long / short)); a pair is (h, w)."""

    def __init__(self, size, interpolation="bicubic"):
        self.size, self.interpolation = size, interpolation

    def __call__(self, img):
        Image, _ = _pil()
        mode = Image.BICUBIC if self.interpolation == "bicubic" else Image.BILINEAR
        w, h = img.size
        if isinstance(self.size, int):
            short, long = (w, h) if w <= h else (h, w)
            if short == self.size:
                return img
            new_short, new_long = self.size, int(self.size * long / short)
            nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
        else:
            nh, nw = self.size
        return img.resize((nw, nh), mode)


class CenterCrop:
    def __init__(self, size):
        self.size = (size, size) if isinstance(size, int) else tuple(size)

    def __call__(self, img):
        Image, _ = _pil()
        th, tw = self.size
        w, h = img.size
        if tw > w or th > h:                       # torchvision pads with zeros first
            pl, pt = max((tw - w) // 2, 0), max((th - h) // 2, 0)
            canvas = Image.new(img.mode, (max(w, tw), max(h, th)))
            canvas.paste(img, (pl, pt))
            img = canvas
            w, h = img.size
        top, left = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))
        return img.crop((left, top, left + tw, top + th))


def _convert_to_rgb(image):
    return image.convert("RGB")


def _adjust_hue(img, hue_factor):
    """torchvision _functional_pil.adjust_hue: shift the H channel of the HSV image by hue_factor * 255 with uint8 wraparound."""
    Image, _ = _pil()
    mode = img.mode
    if mode in ("L", "1", "I", "F"):
        return img
    h, s, v = img.convert("HSV").split()
    np_h = np.array(h, dtype=np.uint8)
    with np.errstate(over="ignore"):
        np_h = np_h + np.uint8(int(hue_factor * 255) & 0xff)
    return Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert(mode)


class ColorJitter:
    """torchvision ColorJitter on a PIL image: the four adjustments in a random order, factors U(max(0, 1 - x), 1 + x) / U(-hue, hue)."""

    def __init__(self, brightness=0.0, contrast=0.0, saturation=0.0, hue=0.0):
        rng = lambda x: (max(0.0, 1.0 - x), 1.0 + x) if x else None
        self.b, self.c, self.s = rng(brightness), rng(contrast), rng(saturation)
        self.h = (-hue, hue) if hue else None

    def __call__(self, img):
        _, ImageEnhance = _pil()
        order = torch.randperm(4).tolist()
        u = lambda r: None if r is None else float(torch.empty(1).uniform_(r[0], r[1]))
        fb, fc, fs, fh = u(self.b), u(self.c), u(self.s), u(self.h)
        for fn in order:
            if fn == 0 and fb is not None:
                img = ImageEnhance.Brightness(img).enhance(fb)
            elif fn == 1 and fc is not None:
                img = ImageEnhance.Contrast(img).enhance(fc)
            elif fn == 2 and fs is not None:
                img = ImageEnhance.Color(img).enhance(fs)
            elif fn == 3 and fh is not None:
                img = _adjust_hue(img, fh)
        return img


class color_jitter:
    """open_clip/transform.py:61-75: ColorJitter with probability p."""

    def __init__(self, brightness=0., contrast=0., saturation=0., hue=0., p=0.8):
        assert 0. <= p <= 1.
        self.p = p
        self.transf = ColorJitter(brightness=brightness, contrast=contrast, saturation=saturation, hue=hue)

    def __call__(self, img):
        return self.transf(img) if random.random() < self.p else img


class gray_scale:
    """open_clip/transform.py:78-91: Grayscale(num_output_channels=3) with probability p."""

    def __init__(self, p=0.2):
        assert 0. <= p <= 1.
        self.p = p

    def __call__(self, img):
        Image, _ = _pil()
        if random.random() >= self.p:
            return img
        g = np.array(img.convert("L"), dtype=np.uint8)
        return Image.fromarray(np.dstack([g, g, g]), "RGB")


def pil_to_tensor(img):
    """torchvision PILToTensor: uint8 [C, H, W], no scaling (the `--to-float-on-device` wire format, train.py:191-197)."""
    a = np.array(img, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    return torch.from_numpy(a).permute(2, 0, 1).contiguous()


def to_tensor(img):
    """torchvision ToTensor: float32 [C, H, W] in [0, 1]."""
    return pil_to_tensor(img).to(torch.float32).div(255.0)


class Normalize:
    def __init__(self, mean, std):
        self.mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
        self.std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

    def __call__(self, t):
        return (t - self.mean) / self.std


def image_transform(
        image_size: int,
        is_train: bool,
        mean: Optional[Tuple[float, ...]] = None,
        std: Optional[Tuple[float, ...]] = None,
        resize_longest_max: bool = False,
        fill_color: int = 0,
        aug_cfg: Optional[Union[Dict[str, Any], AugmentationCfg]] = None,
        to_float_on_device: Optional[bool] = False,
        interpolation: str = 'bicubic',
        square_resize_only: bool = False,
):
    """The reference's transform factory (open_clip/transform.py:91-214): same signature, same argument defaults, same order of
    the transforms, on Pillow alone."""
    def _triple(v, default):
        """None -> the OpenAI statistics; a scalar -> the same value for the three channels."""
        v = default if not v else v
        return tuple(v) if isinstance(v, (list, tuple)) else (v, v, v)

    mean, std = _triple(mean, OPENAI_DATASET_MEAN), _triple(std, OPENAI_DATASET_STD)
    if isinstance(image_size, (list, tuple)) and len(image_size) == 2 and image_size[0] == image_size[1]:
        image_size = image_size[0]                      # a square size is passed on as an int
    aug_cfg = AugmentationCfg(**aug_cfg) if isinstance(aug_cfg, dict) else (aug_cfg or AugmentationCfg())
    normalize = Normalize(mean=mean, std=std)
    tail = [pil_to_tensor] if to_float_on_device else [to_tensor, normalize]
    if is_train:
        cfg = {k: v for k, v in asdict(aug_cfg).items() if v is not None}
        if cfg.pop('use_timm', False):
            raise NotImplementedError("clipa_amd.image_transform: use_timm augmentation needs timm (not part of the MI355X path)")
        t = [RandomResizedCrop(image_size, scale=cfg.pop('scale')), _convert_to_rgb]
        if aug_cfg.color_jitter_prob:
            assert aug_cfg.color_jitter is not None and len(aug_cfg.color_jitter) == 4
            t.append(color_jitter(*aug_cfg.color_jitter, p=aug_cfg.color_jitter_prob))
        if aug_cfg.gray_scale_prob:
            t.append(gray_scale(aug_cfg.gray_scale_prob))
        return Compose(t + tail)
    assert interpolation in ['bicubic', 'bilinear']
    assert not (resize_longest_max and square_resize_only)
    if resize_longest_max:
        raise NotImplementedError("clipa_amd.image_transform: resize_longest_max is not used by the CLIPA recipes")
    if square_resize_only:
        size = (image_size, image_size) if isinstance(image_size, int) else tuple(image_size)
        t = [Resize(size, interpolation)]
    else:
        t = [Resize(image_size, interpolation), CenterCrop(image_size)]
    return Compose(t + [_convert_to_rgb] + tail)
