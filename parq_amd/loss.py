"""Set-prediction loss of the PARQ decoder: host-side mirror of ``PARQDecoder.loss`` (model/parq_decoder.py:165-370),
its matcher (utils/matcher.py:31-115) and the 6-D rotation helpers (utils/ortho6d_transforms.py).

All arithmetic is plain torch on whatever device the decoder outputs live on (the tensors are (B, Q, <=10): this is not a
hot path); the linear-sum-assignment runs in scipy on the host exactly as in the reference.  The reference's quirks are kept,
because a drop-in must give the same numbers:
  * the matcher works on ``coord_pos`` (the INPUT reference points of the iteration), not on the predicted centres (:58);
  * besides the Hungarian pairs every prediction whose reference point lies within L1 < 0.2 of a ground-truth centre is
    matched to it, at most 10 per box, chosen with ``np.random.choice`` (:85-97) — seed numpy for reproducibility;
  * ``punish_mask`` is the mask built for the LAST ground-truth box of a scene, and the list only has entries for scenes
    that have boxes, so a batch with an empty scene before a non-empty one mis-indexes exactly as the reference does;
  * every term is averaged over ``valid_bs`` = number of (iteration, scene) pairs with at least one match (:292,307,364-367).
"""
from __future__ import annotations

import math

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

from .wrappers import Obb3D, Pose, raw


# ---------------------------------------------------------------- rotations (utils/ortho6d_transforms.py)
def rot_to_6d(R):
    """First two COLUMNS of the rotation matrix, concatenated (:16-18)."""
    return torch.cat((R[..., 0], R[..., 1]), dim=-1)


def _unit(v):
    n = torch.sqrt((v * v).sum(1)).clamp_min(1e-8)          # :22-33 (max with 1e-8)
    return v / n.unsqueeze(1)


def _cross(u, v):
    return torch.stack((u[:, 1] * v[:, 2] - u[:, 2] * v[:, 1],
                        u[:, 2] * v[:, 0] - u[:, 0] * v[:, 2],
                        u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0]), dim=1)


def rotation_from_ortho6d(o6):
    """(n, 6) -> (n, 3, 3) with columns x, y, z (Gram-Schmidt, :52-66)."""
    x = _unit(o6[:, 0:3])
    z = _unit(_cross(x, o6[:, 3:6]))
    y = _cross(z, x)
    return torch.stack((x, y, z), dim=2)


def roty(t, device="cpu"):
    """Rotation about the y axis (utils/parq_utils.py:214-218; float32 like torch.Tensor)."""
    c, s = math.cos(t), math.sin(t)
    return torch.tensor([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]], dtype=torch.float32, device=device)


# ---------------------------------------------------------------- matcher (utils/matcher.py)
class HungarianMatcherModified:
    def __init__(self, cost_class=2.0, cost_bbox=0.25, ratio=0.2, max_padding=10):
        self.cost_class, self.cost_bbox, self.ratio, self.max_padding = cost_class, cost_bbox, ratio, max_padding

    @torch.no_grad()
    def __call__(self, outputs, targets):
        prob = outputs["pred_logits"].softmax(-1)
        ref_pts = outputs["coord_pos"]
        indices, punish = [], []
        for b in range(prob.shape[0]):
            ids, centers = targets[b]["labels"], targets[b]["center"]
            if len(ids) == 0:
                # no object in this scene: no matches.  (The reference builds a boolean mask plus a FLOAT index here, :61-65,
                # and its loss then raises IndexError; a scene without boxes is simply skipped instead.)
                indices.append((np.zeros(0, np.int64), np.zeros(0, np.int64)))
                continue
            l1 = torch.cdist(ref_pts[b], centers, p=1)
            cost = self.cost_bbox * l1 - self.cost_class * prob[b][:, ids]
            rows, cols = linear_sum_assignment(cost.cpu())
            extra_p, extra_g = [], []
            near_all = (l1 < self.ratio).cpu().numpy()            # one host copy per scene instead of one per box
            mask_np = None
            for j in range(len(ids)):                           # reference points close to box j (:83-99)
                pidx = np.nonzero(near_all[:, j])[0]
                mask_np = np.ones(near_all.shape[0], dtype=bool)
                mask_np[pidx] = False
                if pidx.shape[0] > self.max_padding:
                    pidx = pidx[np.random.choice(pidx.shape[0], self.max_padding, replace=False)]
                mask_np[pidx] = True
                extra_p.append(pidx)
                extra_g.append(np.ones_like(pidx) * j)
            mask = torch.from_numpy(mask_np).to(l1.device)
            p = np.concatenate([rows, np.concatenate(extra_p)])
            g = np.concatenate([cols, np.concatenate(extra_g)])
            _, first = np.unique(p, return_index=True)           # one ground truth per prediction (:107-110)
            indices.append([p[first], g[first]])
            punish.append(mask)
        return indices, punish


# ---------------------------------------------------------------- targets (model/parq_decoder.py:165-203)
def parse_target(obbs_padded: Obb3D, T_world_local):
    T_local_world = Pose(raw(T_world_local)).inverse()
    out = []
    for i in range(raw(obbs_padded).shape[0]):
        boxes = obbs_padded[i].remove_padding()
        T_lo = Pose(raw(T_local_world[i])).compose(boxes.T_world_object)         # local <- object
        out.append({
            "labels": boxes.sem_id.squeeze(-1).long(),
            "center": T_lo.transform(boxes.bb3_center_object.unsqueeze(1)).squeeze(1),
            "size": boxes.bb3_size,
            "T_rig_object": T_lo.matrix.view(-1, 4, 4),
            "gt_corners": T_lo.transform(boxes.bb3corners_object),
            "gt_ortho6d": rot_to_6d(T_lo.R),
            "gt_corners_world": boxes.T_world_object.transform(boxes.bb3corners_object),
            "T_world_object": boxes.T_world_object,
        })
    return out


_ROTY_CACHE = {}


def _roty_table(m, device):
    """The m y-rotations of a symmetry class, (m, 3, 3) float32, built once per device."""
    key = (m, str(device))
    if key not in _ROTY_CACHE:
        _ROTY_CACHE[key] = torch.stack([roty((k * 2.0 / m) * math.pi) for k in range(m)]).to(device)
    return _ROTY_CACHE[key]


def rotation_loss_with_sym(rot_pred, rot_tgt, sym):
    """Mean over objects of the squared-error rotation loss, minimised over the y-rotations the object's symmetry class
    allows: 1 -> 2-fold, 2 -> 4-fold, 3 -> 36 samples of a full revolution (model/parq_decoder.py:205-262).  The reference
    loops over objects and candidates in Python (tens of thousands of tiny launches per step); here the objects of one
    class are evaluated together — same arithmetic per candidate."""
    folds = {1: 2, 2: 4, 3: 36}
    sym = sym.to(torch.int64) if sym.dtype.is_floating_point else sym
    per_obj = ((rot_pred - rot_tgt) ** 2).mean(dim=(1, 2))                     # classes without symmetry
    for cls, m in folds.items():
        sel = torch.nonzero(sym == cls).squeeze(1)
        if sel.numel() == 0:
            continue
        cand = rot_tgt[sel].unsqueeze(1) @ _roty_table(m, rot_pred.device).to(rot_pred.dtype)      # (n, m, 3, 3)
        err = ((rot_pred[sel].unsqueeze(1) - cand) ** 2).mean(dim=(2, 3))                           # (n, m)
        per_obj = per_obj.index_put((sel,), err.min(dim=1).values)
    return per_obj.mean()


def decoder_loss(out_dict_list, obbs_padded, T_world_local, sym=None, *, matcher, loss_weight, num_semcls, class_weight):
    """model/parq_decoder.py:264-370.  Returns the dict {center_loss, size_loss, rot_loss, cat_loss, total_loss}."""
    assert raw(obbs_padded).ndim == 3, tuple(raw(obbs_padded).shape)
    last = out_dict_list[-1]
    total = (last["ortho6d"].sum() * last["size_unnormalized"].sum() * last["center_unnormalized"].sum() * last["pred_logits"].sum() * 0)
    terms = {"center_loss": 0, "size_loss": 0, "rot_loss": 0, "cat_loss": 0}
    valid_bs = 0
    targets = parse_target(obbs_padded, T_world_local)
    for out in out_dict_list:
        indices, punish = matcher(out, targets)
        for i in range(raw(obbs_padded).shape[0]):
            pi, gi = indices[i][0], indices[i][1]
            if len(pi) == 0:
                continue
            valid_bs += 1
            c = (out["center_unnormalized"][i][pi] - targets[i]["center"][gi]).abs().mean() * loss_weight[0]
            s = (out["size_unnormalized"][i][pi] - targets[i]["size"][gi]).abs().mean() * loss_weight[1]
            rot_pred = rotation_from_ortho6d(out["ortho6d"][i][pi])
            rot_tgt = targets[i]["T_rig_object"][:, :3, :3][gi]
            if sym is not None:
                r = rotation_loss_with_sym(rot_pred, rot_tgt, sym[i][gi])
            else:
                r = ((rot_tgt - rot_pred) ** 2).mean()
            r = r * loss_weight[2]
            logits = out["pred_logits"][i]
            cls_t = torch.full(logits.shape[0:1], num_semcls, dtype=torch.int64, device=logits.device)     # background
            cls_t[pi] = targets[i]["labels"][gi]
            w = class_weight.to(logits.device)
            if punish is not None:
                per_q = torch.nn.functional.cross_entropy(logits, cls_t, weight=w, reduction="none")
                k = (per_q * punish[i]).sum() / punish[i].sum()
            else:
                k = torch.nn.functional.cross_entropy(logits, cls_t, weight=w)
            k = k * loss_weight[3]
            total = total + c + s + r + k
            terms["center_loss"] += c
            terms["size_loss"] += s
            terms["rot_loss"] += r
            terms["cat_loss"] += k
    if valid_bs != 0:
        total = total / valid_bs
        terms = {k: v / valid_bs for k, v in terms.items()}
    terms["total_loss"] = total
    return terms
