"""clipa_amd - MI355X-native compute engine for the CLIPA / open_clip training step.

Public surface mirrors `open_clip` (clipa_torch/open_clip/__init__.py) for the ViT-CLIP hot path:
create_model, create_model_and_transforms, create_loss, CLIP, ClipLoss, convert_weights_to_lp,
get_cast_dtype, list_models, add_model_config.
"""
from .configs import add_model_config, get_model_config, list_models
from .factory import (create_loss, create_model, create_model_and_transforms, get_cast_dtype, load_checkpoint)
from .loss import ClipLoss
from .model import (CLIP, CLIPTextCfg, CLIPVisionCfg, OPENAI_DATASET_MEAN, OPENAI_DATASET_STD, convert_weights_to_lp,
                    get_2d_sincos_pos_embed, resize_pos_embed, resize_text_pos_embed)

from .data import DeviceAugment, DevicePrefetcher
from .transform import AugmentationCfg, image_transform
from .zero import ShardedAdamW

__version__ = "0.4.0"
