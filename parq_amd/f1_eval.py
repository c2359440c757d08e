"""Scene-level F1 evaluation of the detections ("F1 tracker stays host-side").

Mirrors the behaviour of the reference's ``F1Calculator`` (utils/f1_eval.py:254-557) with the same public surface
(``step`` / ``compute_metrics`` / ``reset``): per scene, the boxes predicted on successive snippets are merged into tracks by
a Hungarian assignment on oriented-box IoU (utils/f1_eval.py:293-352), ground truth likewise (:416-471), and at the end every
track is greedily matched to a same-class ground-truth box at IoU thresholds 0.25 / 0.5 / 0.7 (:36-62, :473-502).

This is host code on NumPy + scipy's LSAP (as in the reference): the work is a few hundred box pairs per snippet.
The device side of the evaluation path (box building, validity window, NMS) is ``PARQDecoder.parse_pred``.

The oriented IoU follows the reference's convention (utils/f1_eval.py:46-48,77-107): corners are re-ordered with
[4,0,1,5,7,3,2,6] and rotated by +90 deg about x, the first four re-ordered corners (taken in reverse) are the footprint
polygon in the rotated (x, z) plane, the overlap height is min(y of corner 0) - max(y of corner 4), and the volume is the
product of three edge lengths.  The footprint intersection is a convex polygon clip + shoelace area (the reference clips the
same way and asks Qhull for the area of the result).
"""
from __future__ import annotations

import copy

import numpy as np
from scipy.optimize import linear_sum_assignment

CARE_CLASSES = {0: "chair", 1: "table", 2: "cabinet", 3: "trash bin", 4: "bookshelf", 5: "display", 6: "sofa", 7: "bathtub",
                8: "other"}
_NON_OBJECT = 9
_FACE_ORDER = np.array([4, 0, 1, 5, 7, 3, 2, 6])
_ROT_X90 = np.array([[1.0, 0.0, 0.0], [0.0, np.cos(np.pi / 2), -np.sin(np.pi / 2)], [0.0, np.sin(np.pi / 2), np.cos(np.pi / 2)]])


def _shoelace(poly: np.ndarray) -> float:
    x, y = poly[:, 0], poly[:, 1]
    return 0.5 * abs(float(np.dot(x, np.roll(y, 1)) - np.dot(y, np.roll(x, 1))))


def _clip_convex(subject, clip):
    """Sutherland-Hodgman: the part of polygon `subject` inside the convex counter-clockwise polygon `clip`
    (lists of (x, y)); None when nothing is left (utils/f1_eval.py:132-175: same inside test and intersection formula)."""
    out = list(subject)
    a = clip[-1]
    for b in clip:
        ex, ey = b[0] - a[0], b[1] - a[1]
        src, out = out, []
        prev = src[-1]
        prev_in = ex * (prev[1] - a[1]) > ey * (prev[0] - a[0])
        for cur in src:
            cur_in = ex * (cur[1] - a[1]) > ey * (cur[0] - a[0])
            if cur_in != prev_in:
                # crossing of the segment prev->cur with the line through a, b
                dcx, dcy = a[0] - b[0], a[1] - b[1]
                dpx, dpy = prev[0] - cur[0], prev[1] - cur[1]
                n1 = a[0] * b[1] - a[1] * b[0]
                n2 = prev[0] * cur[1] - prev[1] * cur[0]
                inv = 1.0 / (dcx * dpy - dcy * dpx)
                out.append(((n1 * dpx - n2 * dcx) * inv, (n1 * dpy - n2 * dcy) * inv))
            if cur_in:
                out.append(cur)
            prev, prev_in = cur, cur_in
        a = b
        if not out:
            return None
    return out


def canonical(corners_world: np.ndarray) -> np.ndarray:
    """World corners (8,3) in the IoU routine's frame (re-ordered, rotated about x)."""
    return (_ROT_X90 @ np.asarray(corners_world, dtype=np.float64)[_FACE_ORDER].T).T


def _edge(c, i, j):
    return float(np.sqrt(np.sum((c[i] - c[j]) ** 2)))


def iou3d(c1: np.ndarray, c2: np.ndarray):
    """(3-D IoU, footprint IoU) of two boxes given as canonical() corners; (0, 0) for NaN input or a degenerate overlap."""
    if np.isnan(c1).any() or np.isnan(c2).any():
        return 0.0, 0.0
    r1 = [(c1[i, 0], c1[i, 2]) for i in (3, 2, 1, 0)]
    r2 = [(c2[i, 0], c2[i, 2]) for i in (3, 2, 1, 0)]
    a1, a2 = _shoelace(np.array(r1)), _shoelace(np.array(r2))
    inter = _clip_convex(r1, r2)
    if inter is None:
        inter_area = 0.0
    else:
        if len(inter) < 3:
            return 0.0, 0.0                      # Qhull rejects such input in the reference -> its except branch
        inter_area = _shoelace(np.array(inter))
        if not inter_area > 1e-14 * max(a1, a2):
            return 0.0, 0.0                      # flat intersection: Qhull error in the reference -> (0, 0)
    union_2d = a1 + a2 - inter_area
    iou_2d = inter_area / union_2d if union_2d > 0 else float("nan")       # boxes standing on edge: 0/0 in the reference too
    top = min(c1[0, 1], c2[0, 1])
    bottom = max(c1[4, 1], c2[4, 1])
    inter_vol = inter_area * max(0.0, top - bottom)
    v1 = _edge(c1, 0, 1) * _edge(c1, 1, 2) * _edge(c1, 0, 4)
    v2 = _edge(c2, 0, 1) * _edge(c2, 1, 2) * _edge(c2, 0, 4)
    union = v1 + v2 - inter_vol
    return (inter_vol / union if union > 0 else 0.0), iou_2d


def _iou_matrix(dets, trks) -> np.ndarray:
    """float32 (len(dets), len(trks)) matrix of 3-D IoUs between entries [class, corners, score, ...]."""
    cd = [canonical(d[1]) for d in dets]
    ct = [canonical(t[1]) for t in trks]
    m = np.zeros((len(dets), len(trks)), dtype=np.float32)
    for i, a in enumerate(cd):
        for j, b in enumerate(ct):
            m[i, j] = iou3d(a, b)[0]
    return m


def _associate(dets, trks, iou_thresh):
    """Hungarian assignment on IoU; returns (matches [(det, trk)], unmatched detection indices in the reference's order:
    never-assigned ones first, then assigned pairs whose IoU is under the threshold)."""
    iou = _iou_matrix(dets, trks)
    rows, cols = linear_sum_assignment(-iou)
    assigned = set(int(r) for r in rows)
    unmatched = [d for d in range(len(dets)) if d not in assigned]
    matches = []
    for r, c in zip(rows, cols):
        if iou[r, c] < iou_thresh:
            unmatched.append(int(r))
        else:
            matches.append((int(r), int(c)))
    return matches, unmatched


def count_matches(total_gts, total_preds, total_tps, predictions, gts, threshold):
    """Greedy true-positive count of one scene (utils/f1_eval.py:36-62).  As in the reference, a prediction is not
    consumed by its first match: it scores once for EVERY still-unused same-class ground-truth box it overlaps."""
    used = set()
    cg = [canonical(g[1]) for g in gts]
    for g in gts:
        total_gts[g[0]] += 1
    for pred in predictions:
        cls = pred[0]
        total_preds[cls] += 1
        cp = canonical(pred[1])
        for i, g in enumerate(gts):
            if g[0] != cls:
                continue
            if iou3d(cg[i], cp)[0] > threshold and i not in used:
                used.add(i)
                total_tps[cls] += 1


def f1_from_counts(gts, preds, tps, verbose=False):
    """(accuracy, recall, F1) over the classes that have predictions (utils/f1_eval.py:178-215)."""
    n_gt = n_pred = n_tp = 0
    for c in CARE_CLASSES:
        if preds[c] == 0:
            continue
        if verbose:
            acc = tps[c] / preds[c] if gts[c] else 0
            rec = tps[c] / gts[c] if gts[c] else 0
            print("class %s: accuracy %s recall %s F1 %s" % (CARE_CLASSES[c], acc, rec, 2 * acc * rec / (acc + rec) if acc + rec else 0))
        n_gt += gts[c]
        n_pred += preds[c]
        n_tp += tps[c]
    acc = n_tp / n_pred if n_pred else 0
    rec = n_tp / n_gt if n_gt else 0
    f1 = 2 * acc * rec / (acc + rec) if acc + rec else 0
    return acc, rec, f1


class F1Calculator:
    """Accumulates per-scene prediction and ground-truth tracks over snippets; ``compute_metrics`` returns
    {"<thr>_accuracy", "<thr>_recall", "<thr>_f1"} for thr in f1_iou_thresh."""

    def __init__(self, conf_thresh, f1_iou_thresh=(0.25, 0.5, 0.7), verbose=False):
        self.f1_iou_thresh = list(f1_iou_thresh)
        self.conf_thresh = conf_thresh
        self.iou_thresh = 0.1
        self.verbose = verbose
        self.reset()

    def reset(self):
        self.preds = {}
        self.gts = {}

    # ---- one snippet batch ------------------------------------------------------------------------------------------
    def step(self, outputs, gt_list):
        """outputs: {"pred_corners_world" (B,Q,8,3), "sem_cls_prob" (B,Q,K), "pred_mask" (B,Q), "scene_name" [B]};
        gt_list: per scene {"labels" (n,), "gt_corners_world" (n,8,3)} (utils/f1_eval.py:277-291)."""
        dets = self.parse_predictions(outputs, self.conf_thresh)
        gts = self.make_gt_list(gt_list)
        names = outputs["scene_name"]
        self.matching_pred(dets, names)
        self.matching_gt(gts, names)

    @staticmethod
    def parse_predictions(outputs, conf_thresh):
        """Per scene the list of [class, corners (8,3), score, track id] of the boxes that are not `non-object`, score above
        the confidence threshold and kept by the mask (utils/f1_eval.py:372-414)."""
        prob = _np(outputs["sem_cls_prob"])
        corners = _np(outputs["pred_corners_world"])
        mask = _np(outputs["pred_mask"]).astype(bool)
        cls = prob.argmax(-1)
        score = prob.max(-1)
        keep = (cls != _NON_OBJECT) & (score > conf_thresh) & mask
        return [[[int(cls[b, j]), corners[b, j], score[b, j], -1] for j in np.nonzero(keep[b])[0]]
                for b in range(corners.shape[0])]

    @staticmethod
    def make_gt_list(gt_list):
        """[(class, corners + one N(0, 1e-6) scalar per box, 1)]; the jitter is the reference's (utils/f1_eval.py:354-370) and
        draws from NumPy's global generator in box order."""
        out = []
        for g in gt_list:
            labels, corners = _np(g["labels"]), _np(g["gt_corners_world"])
            out.append([(labels[j].item(), corners[j] + np.random.randn(1) * 0.001, 1) for j in range(corners.shape[0])])
        return out

    def matching_pred(self, detections, scene_names):
        for dets, name in zip(detections, scene_names):
            if name not in self.preds:
                for k, d in enumerate(dets):
                    d[-1] = k
                self.preds[name] = copy.deepcopy(dets)
                continue
            trks = self.preds[name]
            n_before = len(trks)
            matches, unmatched = _associate(dets, trks, self.iou_thresh)
            for d, t in matches:
                dets[d][-1] = trks[t][-1]
                if trks[t][2] < dets[d][2]:              # the more confident observation represents the track
                    trks[t] = dets[d]
            for k, d in enumerate(unmatched):
                dets[d][-1] = n_before + k
                trks.append(dets[d])
            self.preds[name] = copy.deepcopy(trks)
        return detections

    def matching_gt(self, gts, scene_names):
        snapshot = []
        for dets, name in zip(gts, scene_names):
            if name not in self.gts:
                self.gts[name] = dets
                snapshot.append(copy.deepcopy(dets))
                continue
            trks = self.gts[name]
            matches, unmatched = _associate(dets, trks, self.iou_thresh)
            for d, t in matches:
                if trks[t][2] < dets[d][2]:              # scores are the placeholder 1: never replaces
                    trks[t] = dets[d]
            for d in unmatched:
                trks.append(dets[d])
            snapshot.append(copy.deepcopy(trks))
        return snapshot

    # ---- end of epoch -------------------------------------------------------------------------------------------------
    def counts(self, threshold):
        total = [{k: 0 for k in CARE_CLASSES} for _ in range(3)]
        for scene, preds in self.preds.items():
            count_matches(total[0], total[1], total[2], preds, self.gts[scene], threshold)
        return total

    def compute_metrics(self):
        metrics = {}
        for thr in self.f1_iou_thresh:
            gts, preds, tps = self.counts(thr)
            acc, rec, f1 = f1_from_counts(gts, preds, tps, self.verbose)
            metrics["{}_accuracy".format(thr)] = acc
            metrics["{}_recall".format(thr)] = rec
            metrics["{}_f1".format(thr)] = f1
        return metrics


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)
