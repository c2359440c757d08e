"""Set-prediction loss (parq_amd/loss.py) against the golden captured from the reference's PARQDecoder.loss
(oracle/make_golden.py::make_loss_golden): Hungarian + proximity matching (np.random.choice cap, seeded), L1 centre / size,
symmetry-aware rotation loss, class-weighted cross-entropy with the punish mask, valid_bs averaging."""
import json
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
from make_golden import LOSS_CASE, loss_case_inputs  # noqa: E402  (inputs are regenerated from the seed: data only)
from parq_amd import Obb3D, Pose  # noqa: E402
from parq_amd.loss import (HungarianMatcherModified, decoder_loss, decoder_loss_batched, rot_to_6d,  # noqa: E402
                           rotation_from_ortho6d)

GOLD = os.path.join(os.path.dirname(__file__), "golden", "g10_loss.npz")


def _loss(sym_on, fn=decoder_loss):
    c = LOSS_CASE
    outs, obbs, T_wl, sym = loss_case_inputs(c)
    touts = [{k: torch.from_numpy(v) for k, v in o.items()} for o in outs]
    cw = torch.ones(10)
    cw[9] = 0.1
    np.random.seed(c["np_seed"])
    return fn(touts, Obb3D(torch.from_numpy(obbs)), Pose(torch.from_numpy(T_wl)), torch.from_numpy(sym) if sym_on else None,
              matcher=HungarianMatcherModified(cost_class=2, cost_bbox=0.25), loss_weight=[5.0, 5.0, 5.0, 1.0],
              num_semcls=9, class_weight=cw)


@pytest.mark.parametrize("fn", [decoder_loss, decoder_loss_batched])
@pytest.mark.parametrize("tag,sym_on", [("sym", True), ("nosym", False)])
def test_loss_matches_reference_golden(tag, sym_on, fn):
    z = np.load(GOLD)
    assert json.loads(bytes(z["meta"]).decode()) == json.loads(json.dumps(LOSS_CASE))
    got = _loss(sym_on, fn)
    for k in ("center_loss", "size_loss", "rot_loss", "cat_loss", "total_loss"):
        want = float(z["%s_%s" % (tag, k)])
        assert abs(float(got[k]) - want) < 2e-5 * max(1.0, abs(want)), (k, float(got[k]), want)


def test_rotation_6d_round_trip_and_empty_scene():
    R = rotation_from_ortho6d(torch.randn(5, 6, dtype=torch.float64))
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3, dtype=torch.float64).expand(5, 3, 3), atol=1e-12)
    assert torch.allclose(rotation_from_ortho6d(rot_to_6d(R)), R, atol=1e-12)
    # a scene without boxes is skipped (the reference raises there): loss of the other scenes is unchanged
    c = dict(LOSS_CASE)
    outs, obbs, T_wl, sym = loss_case_inputs(c)
    obbs2 = obbs.copy()
    obbs2[2] = -1.0
    touts = [{k: torch.from_numpy(v) for k, v in o.items()} for o in outs]
    cw = torch.ones(10)
    cw[9] = 0.1
    np.random.seed(1)
    l = decoder_loss(touts, Obb3D(torch.from_numpy(obbs2)), Pose(torch.from_numpy(T_wl)), None,
                     matcher=HungarianMatcherModified(2, 0.25), loss_weight=[5.0, 5.0, 5.0, 1.0], num_semcls=9, class_weight=cw)
    assert torch.isfinite(l["total_loss"])


def test_batched_loss_gradients_equal_the_loop_version():
    c = LOSS_CASE
    outs, obbs, T_wl, sym = loss_case_inputs(c)
    cw = torch.ones(10)
    cw[9] = 0.1
    grads = []
    for fn in (decoder_loss, decoder_loss_batched):
        touts = [{k: torch.from_numpy(v).clone().requires_grad_(k != "coord_pos") for k, v in o.items()} for o in outs]
        np.random.seed(c["np_seed"])
        l = fn(touts, Obb3D(torch.from_numpy(obbs)), Pose(torch.from_numpy(T_wl)), torch.from_numpy(sym),
               matcher=HungarianMatcherModified(2, 0.25), loss_weight=[5.0, 5.0, 5.0, 1.0], num_semcls=9, class_weight=cw)
        l["total_loss"].backward()
        grads.append([o[k].grad.clone() for o in touts for k in ("pred_logits", "center_unnormalized", "size_unnormalized", "ortho6d")])
    for a, b in zip(*grads):
        assert torch.allclose(a, b, atol=1e-6, rtol=1e-5)


def test_batched_loss_without_capped_clusters_equals_the_loop_version():
    """The matcher's fast path (no box with more than 10 neighbours: no random draw) against the per-box loop of the loop version:
    the clusters of the golden case are thinned to 6 + 3 reference points, one scene keeps none."""
    c = LOSS_CASE
    outs, obbs, T_wl, sym = loss_case_inputs(c)
    rng = np.random.RandomState(3)
    for o in outs:
        o["coord_pos"][:, 6:14] = rng.uniform(2.5, 3.0, (o["coord_pos"].shape[0], 8, 3)).astype(np.float32)   # far from every box
        o["coord_pos"][2, :18] = rng.uniform(2.5, 3.0, (18, 3)).astype(np.float32)
    cw = torch.ones(10)
    cw[9] = 0.1
    res = []
    for fn in (decoder_loss, decoder_loss_batched):
        touts = [{k: torch.from_numpy(v).clone() for k, v in o.items()} for o in outs]
        state = np.random.get_state()
        np.random.seed(77)
        l = fn(touts, Obb3D(torch.from_numpy(obbs)), Pose(torch.from_numpy(T_wl)), torch.from_numpy(sym),
               matcher=HungarianMatcherModified(2, 0.25), loss_weight=[5.0, 5.0, 5.0, 1.0], num_semcls=9, class_weight=cw)
        drew = np.random.random()                      # position of the global generator after the call
        np.random.set_state(state)
        res.append(({k: float(v) for k, v in l.items()}, drew))
    assert res[0][1] == res[1][1] == np.random.RandomState(77).random_sample()          # neither version drew a sample
    for k in res[0][0]:
        assert abs(res[0][0][k] - res[1][0][k]) < 1e-5 * max(1.0, abs(res[0][0][k])), k
