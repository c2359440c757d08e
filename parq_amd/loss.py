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

import contextlib
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
    """Per scene: labels, centres / sizes / rotations / corners of the ground-truth boxes in the local frame
    (model/parq_decoder.py:165-203).  Evaluated for the padded (B, max_box) array at once — one host read for the box counts —
    and sliced per scene (the reference loops over scenes with a dozen small launches each)."""
    X = raw(obbs_padded)
    keep = ~torch.all(X == -1, dim=-1)
    counts = keep.sum(-1).tolist() if X.ndim == 3 else None
    if counts is None:
        raise AssertionError("obbs_padded must be (B, max_box, 19)")
    boxes = Obb3D(X)
    T_local_world = Pose(raw(T_world_local)).inverse()                        # (B, 1, 12)
    T_lo = Pose(raw(T_local_world)).compose(boxes.T_world_object)             # local <- object, (B, max_box, 12)
    center = T_lo.transform(boxes.bb3_center_object.unsqueeze(-2)).squeeze(-2)
    corners_obj = boxes.bb3corners_object
    gt_corners = T_lo.transform(corners_obj)
    gt_corners_world = boxes.T_world_object.transform(corners_obj)
    T44 = T_lo.matrix
    o6 = rot_to_6d(T_lo.R)
    labels = boxes.sem_id.squeeze(-1).long()
    size = boxes.bb3_size
    out = []
    for i, n in enumerate(counts):
        out.append({
            "labels": labels[i, :n],
            "center": center[i, :n],
            "size": size[i, :n],
            "T_rig_object": T44[i, :n].reshape(-1, 4, 4),
            "gt_corners": gt_corners[i, :n],
            "gt_ortho6d": o6[i, :n],
            "gt_corners_world": gt_corners_world[i, :n],
            "T_world_object": Pose(raw(boxes.T_world_object)[i, :n]),
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
    matcher.last_valid_bs = valid_bs          # introspection for tests; the returned dict is the reference's
    if valid_bs != 0:
        total = total / valid_bs
        terms = {k: v / valid_bs for k, v in terms.items()}
    terms["total_loss"] = total
    return terms


_DIFF_KEYS = ("pred_logits", "center_unnormalized", "size_unnormalized", "ortho6d")
DEVICE_SET_LOSS = True        # device tensors: loss terms + output gradients by parq_set_loss (False: the torch expression, kept as the
                              # reference the kernel is tested against)


def _stacked(out_dict_list, key):
    """(I, B, Q, k) tensor of one output over the iterations: the tensor the per-iteration dicts are views of when they come from
    PARQDecoder's training forward (no copy, and autograd reaches the decoder's node directly), else a stack."""
    t0 = out_dict_list[0][key]
    base, I = t0._base, len(out_dict_list)
    if (base is not None and base.dim() == t0.dim() + 1 and base.shape[0] == I and base.is_contiguous()
            and all(o[key]._base is base and o[key].shape == t0.shape and o[key].storage_offset() == base.storage_offset() + i * base.stride(0)
                    for i, o in enumerate(out_dict_list))):
        return base
    return torch.stack([o[key] for o in out_dict_list]).contiguous()


class _SetLossFn(torch.autograd.Function):
    """parq_set_loss (parq_amd/csrc/setloss.hip): the four terms and d term / d output in three launches; each term depends on one
    output tensor, so the backward is four scalings."""

    @staticmethod
    def forward(ctx, logits, ctr, siz, r6, aux):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        I, B, Q, ncls = logits.shape
        dev = logits.device
        logits, ctr, siz, r6 = (t.detach().contiguous() for t in (logits, ctr, siz, r6))
        terms = torch.empty(4, dtype=torch.float32, device=dev)
        g = [torch.empty_like(t) for t in (logits, ctr, siz, r6)]
        scratch = torch.empty(I * B * Q, dtype=torch.int32, device=dev)
        P = aux["P"]
        host = aux["packed"]                               # int32 view: pairs [4][P] | coef [P] | row_weight [I*B*Q]
        base = host.data_ptr()
        w4 = (C.c_float * 4)(*[float(x) for x in aux["loss_weight"]])
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        _lib.check(lib.parq_set_loss(p(logits), p(ctr), p(siz), p(r6), I, B, Q, ncls, p(aux["t_center"]), p(aux["t_size"]), p(aux["t_rot"]),
                                     p(aux["t_lab"]), p(aux["t_sym"]), aux["nmax"], C.c_void_p(base), C.c_void_p(base + 16 * P), P,
                                     C.c_void_p(base + 20 * P), p(aux["class_weight"]), w4, p(terms), p(g[0]), p(g[1]), p(g[2]), p(g[3]),
                                     p(scratch), _lib.stream_ptr()), "parq_set_loss")
        ctx.grads = g
        c, s_, r, k = terms.unbind(0)
        return c, s_, r, k

    @staticmethod
    def backward(ctx, gc, gs, gr, gk):
        gl, gcen, gsiz, gr6 = ctx.grads
        return gl * gk, gcen * gc, gsiz * gs, gr6 * gr, None


def _device_set_loss(out_dict_list, I, B, Q, nmax, tc, t_size, t_rot, t_lab, t_sym, cw_dev, seg, pi_all, gi_all, bi_all, punish_np,
                     valid_np, valid_bs, loss_weight, num_semcls):
    """Tail of decoder_loss_batched for device tensors: host-side constants of the matching in ONE upload, then parq_set_loss."""
    dev = tc.device
    seg_np = np.concatenate(seg)
    P = int(seg_np.shape[0])
    cnt = np.bincount(seg_np, minlength=I * B).astype(np.float32)
    coef = (np.float32(1.0) / (cnt[seg_np] * np.float32(valid_bs))).astype(np.float32)
    pm = punish_np.astype(np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        rw = (pm / pm.sum(-1, keepdims=True) * valid_np[..., None].astype(np.float32) / np.float32(valid_bs)).astype(np.float32)
    pairs = np.stack([seg_np // B, np.concatenate(bi_all), np.concatenate(pi_all), np.concatenate(gi_all)]).astype(np.int32)
    packed = torch.from_numpy(np.concatenate([pairs.reshape(-1), coef.view(np.int32), rw.reshape(-1).view(np.int32)])).to(dev)
    assert num_semcls + 1 == out_dict_list[0]["pred_logits"].shape[-1]
    aux = {"P": P, "packed": packed, "loss_weight": loss_weight, "t_center": tc.contiguous(), "t_size": t_size, "t_rot": t_rot,
           "t_lab": t_lab, "t_sym": t_sym, "nmax": int(nmax), "class_weight": cw_dev}
    c, s_, r, k = _SetLossFn.apply(*[_stacked(out_dict_list, key) for key in _DIFF_KEYS], aux)
    return {"center_loss": c, "size_loss": s_, "rot_loss": r, "cat_loss": k, "total_loss": c + s_ + r + k}


_SIDE_STREAMS = {}


def _side_stream(dev):
    """One extra stream per device for the matcher inputs of iterations whose successors are still running on the main stream."""
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[key]


# ---------------------------------------------------------------- batched evaluation (same numbers, ~40 launches per step)
def decoder_loss_batched(out_dict_list, obbs_padded, T_world_local, sym=None, *, matcher, loss_weight, num_semcls, class_weight,
                         ready=None, targets_ready=None):
    """``decoder_loss`` with every (iteration, scene) pair evaluated together: one softmax / cdist / host copy for the matcher,
    the matched (prediction, box) pairs of all iterations and scenes gathered into flat index tensors, per-pair terms reduced
    with segment sums.  The matching itself (scipy LSAP, the np.random.choice cap, the punish-mask quirks) is the loop of
    ``HungarianMatcherModified`` on host copies, in the same (iteration, scene, box) order, so seeded runs give the same
    matches as ``decoder_loss``; the terms differ from it only by floating-point summation order."""
    X = raw(obbs_padded)
    assert X.ndim == 3, tuple(X.shape)
    I, B = len(out_dict_list), X.shape[0]
    dev = out_dict_list[-1]["pred_logits"].device
    # `ready` (PARQDecoder.loss on the outputs of its own training forward): ready(k) blocks until iteration k's outputs are written
    # and returns True if the forward had to be re-run (fp16 range fallback) -> the matching starts over.  The target preparation
    # and the per-iteration matcher inputs then run on a side stream while the device is still in the later iterations; the
    # matching order (iteration, scene, box) and with it the np.random draws are the same as without it.
    # The side stream must not read the caller's targets (obbs_padded, T_world_local, sym) before the work that produced them
    # has run: `targets_ready` is an event the module recorded on the main stream at the ENTRY of its training forward, so
    # targets enqueued before that forward — a pinned-memory .to(device, non_blocking=True), on-device augmentation behind the
    # previous step — are ordered in front of everything below, while the forward's iterations are not waited for.  Targets
    # produced AFTER the forward was enqueued are the caller's to order; without an event the side stream waits for the whole
    # main stream (correct, no overlap).
    main = torch.cuda.current_stream(dev) if (ready is not None and dev.type == "cuda") else None
    side = _side_stream(dev) if main is not None else None
    on_side = (lambda: torch.cuda.stream(side)) if side is not None else contextlib.nullcontext
    if side is not None:
        if targets_ready is not None:
            side.wait_event(targets_ready)
        else:
            side.wait_stream(main)
    with on_side():
        targets = parse_target(obbs_padded, T_world_local)
        nmax = max(1, max(len(t["labels"]) for t in targets))
        tc = torch.zeros(B, nmax, 3, device=dev, dtype=out_dict_list[-1]["coord_pos"].dtype)
        for b, t in enumerate(targets):
            tc[b, :len(t["labels"])] = t["center"]
        ids_np = [t["labels"].cpu().numpy() for t in targets]
        fused = DEVICE_SET_LOSS and dev.type == "cuda" and all(o[k].dtype == torch.float32 for o in out_dict_list for k in _DIFF_KEYS)
        if fused:                                          # padded targets of the device loss (parq_set_loss), built off the main stream
            t_size = torch.zeros(B, nmax, 3, device=dev); t_rot = torch.zeros(B, nmax, 3, 3, device=dev)
            t_lab = torch.zeros(B, nmax, dtype=torch.int32, device=dev)
            for b, t in enumerate(targets):
                n = len(t["labels"])
                t_size[b, :n] = t["size"]; t_rot[b, :n] = t["T_rig_object"][:, :3, :3]; t_lab[b, :n] = t["labels"].to(torch.int32)
            t_sym = raw(sym).to(dev)[:, :nmax].to(torch.int32).contiguous() if sym is not None else None
            cw_dev = class_weight.to(device=dev, dtype=torch.float32).contiguous()
            tc = tc.to(torch.float32)
    Q = out_dict_list[-1]["pred_logits"].shape[1]

    def matcher_inputs(k0, k1):
        """softmax probabilities and L1 centre distances of iterations k0 .. k1-1 as NumPy arrays (one host copy each)"""
        with torch.no_grad(), on_side():
            lg = torch.stack([out_dict_list[k]["pred_logits"] for k in range(k0, k1)])
            cp = torch.stack([out_dict_list[k]["coord_pos"] for k in range(k0, k1)])
            prob = lg.softmax(-1)
            l1 = torch.cdist(cp.flatten(0, 1), tc.repeat(k1 - k0, 1, 1), p=1).view(k1 - k0, B, Q, nmax)
            return prob.cpu().numpy(), l1.cpu().numpy()
    if ready is None:
        prob_all, l1_all = matcher_inputs(0, I)
    seg, pi_all, gi_all, bi_all = [], [], [], []          # flat matched pairs: segment (k*B+b), query, box, scene
    punish_np = np.ones((I, B, Q), dtype=bool)
    valid_np = np.zeros((I, B), dtype=bool)
    k = -1
    while k + 1 < I:
        k += 1
        if ready is not None:
            if ready(k):                                   # outputs rewritten by a re-run of the forward: start over
                seg, pi_all, gi_all, bi_all = [], [], [], []
                punish_np[:] = True
                valid_np[:] = False
                k = -1
                continue
            prob_k, l1_k = matcher_inputs(k, k + 1)
            prob_np, l1_np, kk_ = prob_k, l1_k, 0
        else:
            prob_np, l1_np, kk_ = prob_all, l1_all, k
        plist = []
        idx_k = []
        for b in range(B):
            ids = ids_np[b]
            n = len(ids)
            if n == 0:
                idx_k.append((np.zeros(0, np.int64), np.zeros(0, np.int64)))
                continue
            l1b = l1_np[kk_, b, :, :n]
            cost = matcher.cost_bbox * l1b - matcher.cost_class * prob_np[kk_, b][:, ids]
            rows, cols = linear_sum_assignment(cost)
            near = l1b < matcher.ratio
            if near.sum(0).max() <= matcher.max_padding:
                # no box has more neighbours than the cap: no random draw, every neighbour is an extra match (pairs ordered by box,
                # then query, as the per-box loop appends them) and the punish mask (taken from the LAST box only) keeps everything
                eg, ep = np.nonzero(near.T)
                mask_np = np.ones(Q, dtype=bool)
            else:
                extra_p, extra_g, mask_np = [], [], None
                for j in range(n):
                    pidx = np.nonzero(near[:, j])[0]
                    mask_np = np.ones(Q, dtype=bool)
                    mask_np[pidx] = False
                    if pidx.shape[0] > matcher.max_padding:
                        pidx = pidx[np.random.choice(pidx.shape[0], matcher.max_padding, replace=False)]
                    mask_np[pidx] = True
                    extra_p.append(pidx)
                    extra_g.append(np.ones_like(pidx) * j)
                ep, eg = np.concatenate(extra_p), np.concatenate(extra_g)
            p = np.concatenate([rows, ep])
            g = np.concatenate([cols, eg])
            _, first = np.unique(p, return_index=True)
            idx_k.append((p[first], g[first]))
            plist.append(mask_np)
        for b in range(B):
            pi, gi = idx_k[b]
            if len(pi) == 0:
                continue
            valid_np[k, b] = True
            seg.append(np.full(len(pi), k * B + b)); pi_all.append(pi); gi_all.append(gi); bi_all.append(np.full(len(pi), b))
            punish_np[k, b] = plist[b]                         # the reference indexes its list by scene number (quirk kept)
    if main is not None:
        main.wait_stream(side)                             # the target tensors were produced on the side stream
        for t_ in [tc] + [v for t in targets for v in t.values() if torch.is_tensor(v)] + \
                ([t_size, t_rot, t_lab, cw_dev] + ([t_sym] if t_sym is not None else []) if fused else []):
            t_.record_stream(main)
    valid_bs = int(valid_np.sum())
    if fused and valid_bs > 0:
        matcher.last_valid_bs = valid_bs
        return _device_set_loss(out_dict_list, I, B, Q, nmax, tc, t_size, t_rot, t_lab, t_sym, cw_dev, seg, pi_all, gi_all, bi_all,
                                punish_np, valid_np, valid_bs, loss_weight, num_semcls)
    logits = torch.stack([o["pred_logits"] for o in out_dict_list])                  # (I, B, Q, ncls)
    punish = torch.from_numpy(punish_np).to(torch.float32)
    valid = torch.from_numpy(valid_np)
    last = out_dict_list[-1]
    total0 = (last["ortho6d"].sum() * last["size_unnormalized"].sum() * last["center_unnormalized"].sum() * last["pred_logits"].sum() * 0)
    valid_bs = int(valid_np.sum())
    matcher.last_valid_bs = valid_bs
    if valid_bs == 0:
        return {"center_loss": 0, "size_loss": 0, "rot_loss": 0, "cat_loss": 0, "total_loss": total0}
    packed = torch.from_numpy(np.stack([np.concatenate(seg), np.concatenate(pi_all), np.concatenate(gi_all),
                                        np.concatenate(bi_all)]).astype(np.int64)).to(dev)          # one host -> device copy
    seg_t, pi_t, gi_t, bi_t = packed[0], packed[1], packed[2], packed[3]
    kk = seg_t // B
    nseg = I * B
    cnt = torch.zeros(nseg, device=dev).index_add_(0, seg_t, torch.ones_like(seg_t, dtype=torch.float32))

    def seg_mean(per_pair):                                   # mean over the pairs of every (iteration, scene)
        return torch.zeros(nseg, device=dev, dtype=per_pair.dtype).index_add_(0, seg_t, per_pair) / cnt.clamp_min(1)
    # ground truth, padded per scene
    t_size = torch.zeros(B, nmax, 3, device=dev); t_rot = torch.zeros(B, nmax, 3, 3, device=dev)
    t_lab = torch.zeros(B, nmax, dtype=torch.int64, device=dev)
    for b, t in enumerate(targets):
        n = len(t["labels"])
        t_size[b, :n] = t["size"]; t_rot[b, :n] = t["T_rig_object"][:, :3, :3]; t_lab[b, :n] = t["labels"]
    ctr = torch.stack([o["center_unnormalized"] for o in out_dict_list])
    siz = torch.stack([o["size_unnormalized"] for o in out_dict_list])
    r6 = torch.stack([o["ortho6d"] for o in out_dict_list])
    c_seg = seg_mean((ctr[kk, bi_t, pi_t] - tc[bi_t, gi_t]).abs().mean(-1)) * loss_weight[0]
    s_seg = seg_mean((siz[kk, bi_t, pi_t] - t_size[bi_t, gi_t]).abs().mean(-1)) * loss_weight[1]
    rot_pred = rotation_from_ortho6d(r6[kk, bi_t, pi_t])
    rot_tgt = t_rot[bi_t, gi_t]
    per_obj = ((rot_pred - rot_tgt) ** 2).mean(dim=(1, 2))
    if sym is not None:
        sy = raw(sym).to(dev)[bi_t, gi_t].to(torch.int64)
        for cls, m in {1: 2, 2: 4, 3: 36}.items():
            sel = torch.nonzero(sy == cls).squeeze(1)
            if sel.numel():
                cand = rot_tgt[sel].unsqueeze(1) @ _roty_table(m, dev).to(rot_pred.dtype)
                per_obj = per_obj.index_put((sel,), ((rot_pred[sel].unsqueeze(1) - cand) ** 2).mean(dim=(2, 3)).min(dim=1).values)
    r_seg = seg_mean(per_obj) * loss_weight[2]
    cls_t = torch.full((I, B, Q), num_semcls, dtype=torch.int64, device=dev)
    cls_t[kk, bi_t, pi_t] = t_lab[bi_t, gi_t]
    per_q = torch.nn.functional.cross_entropy(logits.flatten(0, 2), cls_t.flatten(), weight=class_weight.to(dev), reduction="none").view(I, B, Q)
    pm = punish.to(dev)
    k_seg = ((per_q * pm).sum(-1) / pm.sum(-1)).flatten() * loss_weight[3]
    vmask = valid.flatten().to(dev)
    terms = {"center_loss": (c_seg * vmask).sum() / valid_bs, "size_loss": (s_seg * vmask).sum() / valid_bs,
             "rot_loss": (r_seg * vmask).sum() / valid_bs, "cat_loss": (k_seg * vmask).sum() / valid_bs}
    terms["total_loss"] = total0 + terms["center_loss"] + terms["size_loss"] + terms["rot_loss"] + terms["cat_loss"]
    return terms
