import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
MODEL_CASES = ["cls_erf", "gap_sincos_tanh", "bigvision_quick", "h14_dh80"]
# full model dimensions at BASELINE shapes (ViT-L/16, ViT-B/16, ViT-H/14 @ 224 + text-77; ViT-L/16 @ 84 GAP / sin-cos),
# and BASELINE config 1 at its stated dimensions: ViT-S/16 @ 112, text-32, batch 64), generated from the reference's own
# model_configs/*.json by oracle/make_golden.py
FULL_CASES = ["full_B16_224", "full_L16_224", "full_H14_224", "full_L16_84_gap", "full_S16_112_t32"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden:
    """One tests/golden/<case>.npz produced by oracle/make_golden.py from the real reference."""

    def __init__(self, name):
        from oracle import clip_oracle as O
        z = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
        self.name = name
        self.z = z
        self.cfg = json.loads(str(z["cfg"]))
        self.seed = int(z["seed"])
        self.shapes = {k: tuple(v) for k, v in json.loads(str(z["shapes"])).items()}
        self.frozen = [str(k) for k in z["frozen"] if str(k)]
        frozen_values = json.loads(str(z["frozen_values"]))
        self.sd = O.make_state_dict(self.shapes, self.seed, frozen=self.frozen)
        for k in self.frozen:
            self.sd[k] = torch.tensor(frozen_values[k], dtype=torch.float32)
        self.images_u8 = torch.from_numpy(z["images_u8"])
        self.texts = torch.from_numpy(z["texts"])
        self.ocfg = O.oracle_cfg(self.cfg)

    def t(self, key):
        return torch.from_numpy(np.asarray(self.z[key]))


@pytest.fixture(params=MODEL_CASES)
def golden(request):
    return Golden(request.param)


@pytest.fixture(params=FULL_CASES)
def golden_full(request):
    return Golden(request.param)


def load_golden(name):
    return Golden(name)
