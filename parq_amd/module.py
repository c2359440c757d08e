"""PARQ: the reference Lightning module's interface (model/parq_lightning.py:28-145) over the HIP path.

``forward(batch, batch_idx) -> (losses, outputs)`` reproduces model/parq_lightning.py:68-95 for the part
that is in scope this round: ray positional encoding + tokenisation + decoder.  The 2-D backbone is not
part of the decoder path (torchvision ResNet-FPN): pass any callable that adds
``all_features`` (B,T,C,h,w) and ``camera_feature`` to the batch, or feed batches that already carry them.
Lightning is used as the base class only when it is importable (it is not on the GPU image).
"""
from __future__ import annotations

import math

import torch
from torch import nn

from .decoder import PARQDecoder
from .ray_pe import AddRayPE

try:                                                    # pragma: no cover - absent on the target image
    from pytorch_lightning import LightningModule as _Base
except Exception:                                       # noqa: BLE001
    _Base = nn.Module


def _get(cfg, path):
    for k in path.split("."):
        cfg = cfg[k] if isinstance(cfg, dict) else getattr(cfg, k)
    return cfg


class PARQ(_Base):
    def __init__(self, cfg, backbone2d=None):
        super().__init__()
        self.cfg = cfg
        self.backbone2d = backbone2d
        tk = _get(cfg, "MODEL.TOKENIZER")
        self.add_ray_pe = AddRayPE(_get(tk, "OUT_CHANNELS"), _get(tk, "RAY_POINTS_SCALE"), _get(tk, "NUM_SAMPLES"),
                                   _get(tk, "MIN_DEPTH"), _get(tk, "MAX_DEPTH"))
        self.box3d_decoder = PARQDecoder(_get(cfg, "MODEL.DECODER"))
        self.for_vis = _get(cfg, "MODEL.DECODER.FOR_VIS")
        self.synced_metrics = {}          # validation metrics averaged over the ranks (what the reference logs with sync_dist=True)

    def set_data_parallel(self, on=True):
        """One process per GPU, scenes sharded (train.py:103-108 runs DDP): with ``on`` every trainable tensor of the module —
        the decoder's flat gradient arena and the ray-PE encoder's four tensors — is averaged over the default process group
        (RCCL over xGMI) inside the two backward nodes, i.e. two collectives per step instead of DDP's per-tensor buckets.
        Do not wrap the module in DistributedDataParallel as well."""
        self.box3d_decoder.dp_all_reduce = bool(on)
        self.add_ray_pe.dp_all_reduce = bool(on)
        return self

    def forward(self, batch, batch_idx=0):
        if self.backbone2d is not None:
            batch = self.backbone2d(batch)
        feats = batch["all_features"]
        # encoding + `images_feat = features + encoding` + both einops rearranges, fused (parq_lightning.py:72-85)
        input_tokens = self.add_ray_pe.tokens(feats, batch["camera_feature"], batch["T_camera_pseudoCam"],
                                              batch["T_world_pseudoCam"], batch["T_world_local"])
        outputs = self.box3d_decoder(input_tokens, batch["camera_feature"], batch["T_camera_pseudoCam"],
                                     batch["T_world_pseudoCam"], batch["T_world_local"],
                                     feat_hw=tuple(feats.shape[-2:]))
        if "obbs_padded" in batch:
            losses = self.box3d_decoder.loss(outputs, batch["obbs_padded"], batch["T_world_local"], batch["sym"])
        else:
            losses = {"total_loss": 0}
        return losses, outputs

    def validation_step(self, batch, batch_idx):
        """model/parq_lightning.py:102-112 without the image logging: with ground truth in the batch the scene-level F1
        trackers are advanced by this snippet batch."""
        losses, outputs = self.forward(batch, batch_idx)
        if "obbs_padded" in batch:
            self.box3d_decoder.update_metrics(outputs, batch["obbs_padded"], batch["T_world_local"], batch["scene_name"])
        return losses["total_loss"]

    def on_validation_epoch_start(self):
        self.box3d_decoder.reset_metrics()

    def validation_epoch_end(self, outs=None):
        """model/parq_lightning.py:118-142: {"0.25_f1", ...}; `eval.py` calls this directly and prints the result.  As in the
        reference the RETURNED dict is this rank's own; the scalars it logs with ``sync_dist=True`` (:133-140: mean over the
        ranks) are kept in ``self.synced_metrics`` and, under Lightning, logged the same way."""
        from .parallel import all_reduce_mean_scalars
        metrics = self.box3d_decoder.compute_metrics()
        self.synced_metrics = all_reduce_mean_scalars(metrics)
        if _Base is not nn.Module:                        # pragma: no cover - Lightning is absent on the target image
            for key, value in metrics.items():
                if isinstance(value, (int, float)):
                    self.log("val/metrics/{}".format(key), value, on_epoch=True, logger=True, sync_dist=True, rank_zero_only=False)
        return metrics

    def test_step(self, batch, batch_idx=0):
        return self.forward(batch, batch_idx)

    def training_step(self, batch, batch_idx):
        """model/parq_lightning.py:97-100.  In train mode the decoder forward is an autograd node whose backward is the HIP
        backward chain, and AddRayPE.tokens is an autograd node too: the decoder and the ray-PE
        encoder receive gradients, and d loss / d features is handed to whatever produced ``all_features``."""
        losses, _ = self.forward(batch, batch_idx)
        return losses["total_loss"]

    def configure_optimizers(self):
        """AdamW with the reference's batch-size learning-rate rule and its warm-up + cosine-restart schedule, stepped per
        epoch (model/parq_lightning.py:150-199).  The reference's `torch.optim._multi_tensor.AdamW` no longer exists in
        torch 2 (`foreach=True` is the same implementation) and its 1-tuple return value is not inherited: this returns the
        Lightning dictionary itself, or the bare optimizer when the config carries no schedule section."""
        from .schedule import CosineAnnealingWarmupRestarts
        base = _get(self.cfg, "OPTIMIZER.LEARNING_RATE")
        lr, eff = base, None
        try:
            eff = (_get(self.cfg, "DATAMODULE.BATCH_SIZE") * _get(self.cfg, "TRAINER.NUM_NODES") * _get(self.cfg, "TRAINER.GPUS") *
                   _get(self.cfg, "TRAINER.ACCUMULATE_GRAD_BATCHES"))
            if _get(self.cfg, "OPTIMIZER.AUTOSCALE_LR"):
                lr = base * eff / 256.0
        except (KeyError, AttributeError):
            pass
        optimizer = torch.optim.AdamW([p for p in self.parameters() if p.requires_grad], lr=lr, foreach=True)
        try:
            warmup = _get(self.cfg, "OPTIMIZER.WARMUP_EPOCHS")
            mult = _get(self.cfg, "OPTIMIZER.CYCLE_MULT")
            restarts = _get(self.cfg, "OPTIMIZER.NUM_RESTARTS")
            epochs = _get(self.cfg, "TRAINER.MAX_EPOCHS")
        except (KeyError, AttributeError):
            return optimizer
        lr_min = base / 256.0 if (eff is not None and eff <= 256) else base
        cycle0 = math.ceil(epochs / sum(pow(mult, i) for i in range(restarts)))
        scheduler = CosineAnnealingWarmupRestarts(optimizer, cycle0, mult, lr, lr_min, warmup)
        return {"optimizer": optimizer, "lr_scheduler": {"scheduler": scheduler, "interval": "epoch"}}
