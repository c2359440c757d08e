"""DEV-CONTAINER ONLY: import the real (Python) reference from /root/reference.

Used by ``oracle/make_golden.py`` and by the optional ``tests/test_oracle_vs_reference.py``
(skipped when /root/reference is absent, i.e. on the GPU box).  Nothing from the
reference is copied: it is imported in place, after stubbing the third-party
modules this image lacks (pytorch_lightning, torchvision, cv2, numba,
torch._six — SURVEY.md Appendix C).
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = os.environ.get("PARQ_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "model"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load():
    """Returns a namespace with PARQDecoder, AddRayPE, Camera, Pose, Obb3D."""
    if not available():
        raise RuntimeError("reference tree not found at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True          # never drop __pycache__ into the reference tree
    import torch

    if "pytorch_lightning" not in sys.modules:
        class LightningModule(torch.nn.Module):
            def save_hyperparameters(self, *a, **k):
                pass

            def log(self, *a, **k):
                pass

        class LightningDataModule:
            pass

        pl = _stub("pytorch_lightning", LightningModule=LightningModule)
        pl.utilities = _stub("pytorch_lightning.utilities", rank_zero_only=lambda f: f)
        pl.core = _stub("pytorch_lightning.core", LightningDataModule=LightningDataModule)
        pl.LightningDataModule = LightningDataModule
    if "torchvision" not in sys.modules:
        tv = _stub("torchvision")
        tv.transforms = _stub("torchvision.transforms")
        tv.models = _stub("torchvision.models")
        tv.models.detection = _stub("torchvision.models.detection")
        tv.models.detection.backbone_utils = _stub(
            "torchvision.models.detection.backbone_utils", resnet_fpn_backbone=None)
    if "cv2" not in sys.modules:
        _stub("cv2")
    if "numba" not in sys.modules:
        def jit(*a, **k):
            if len(a) == 1 and callable(a[0]) and not k:
                return a[0]
            return lambda f: f
        _stub("numba", jit=jit)
    if "torch._six" not in sys.modules:
        six = _stub("torch._six", string_classes=(str, bytes))
        torch._six = six

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from utils import Camera, Pose, Obb3D                      # noqa: E402
    from model.parq_decoder import PARQDecoder                 # noqa: E402
    from model.ray_positional_encoding import AddRayPE         # noqa: E402
    return types.SimpleNamespace(PARQDecoder=PARQDecoder, AddRayPE=AddRayPE, Camera=Camera,
                                 Pose=Pose, Obb3D=Obb3D,
                                 mean_size_path=os.path.join(REFERENCE_ROOT, "data", "average_scan2cad.txt"))


def build_reference_decoder(ref, cfg, weights: dict, double=False):
    """Instantiate the reference PARQDecoder and load synthetic weights into it
    through its own state_dict (so key compatibility is exercised too)."""
    import copy
    import numpy as np
    import torch

    cfg = copy.deepcopy(cfg)
    cfg.MEAN_SIZE_PATH = ref.mean_size_path
    dec = ref.PARQDecoder(cfg).eval()
    sd = dec.state_dict()
    new = {}
    for k in sd:
        src = k.replace("parq_module.decoder.mlp_heads.", "mlp_heads.")
        new[k] = torch.from_numpy(np.asarray(weights[src])).reshape(sd[k].shape).clone()
    missing = set(weights) - {k.replace("parq_module.decoder.mlp_heads.", "mlp_heads.") for k in sd}
    assert not missing, "synthetic weights not consumed by the reference: %s" % sorted(missing)
    dec.load_state_dict(new, strict=True)
    return dec.double() if double else dec
