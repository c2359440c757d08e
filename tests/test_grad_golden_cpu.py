"""CPU tier: float64 autograd of the oracle (oracle/parq_oracle.py) + the host set loss (parq_amd/loss.py) against golden
g17_grads = the REFERENCE's own autograd (float64, eval mode) — oracle/make_golden.py::make_grad_golden.

What forward vectors cannot pin and this does: where the reference detaches (reference points between iterations,
model/transformer_parq.py:331-332), what it computes without a graph (class probabilities, :261-265), the arg-max mean-size
gather (utils/parq_utils.py:96-98), which parameters it leaves without gradient (`decoder.norm.*`), and the gradient of its set
loss (model/parq_decoder.py:264-370) including the matcher's decisions."""
import numpy as np
import pytest
import torch

from parq_amd import Obb3D, Pose, synth
from parq_amd.loss import HungarianMatcherModified, decoder_loss
from oracle import make_golden as MG
from oracle import parq_oracle as O
import golden_util as G


def oracle_run(c, kind):
    cfg, W, sc, cots, obbs, sym = MG.grad_case_inputs(c)
    od = O.OracleDecoder(cfg, W, synth.SCANNET_MEAN_SIZES, dtype=torch.float64)
    for k in od.W:
        od.W[k].requires_grad_(True)
    od.prepare(sc["tokens"], sc["camera"], sc["T_camera_pseudoCam"], sc["T_world_pseudoCam"], sc["T_world_local"])
    od.tokens.requires_grad_(True)
    ref = od.initial_ref()
    outs = []
    for k in range(cfg.TRANSFORMER.DEC_LAYERS):
        out, nxt, _ = od.iterate(ref, k)
        outs.append(out)
        ref = nxt.detach()
    if kind == "linear":
        loss = sum((o[k] * torch.from_numpy(cots[k][i]).double()).sum() for i, o in enumerate(outs) for k in MG.GRAD_KEYS)
        terms = {}
    else:
        cw = torch.ones(cfg.NUM_SEMCLS + 1)                 # the reference's float32 table (parq_decoder.py:46-48), then widened
        cw[cfg.NUM_SEMCLS] = 0.1
        cw = cw.double()
        np.random.seed(c["np_seed"])
        torch.set_default_dtype(torch.float64)       # as the fixture's generator: constants built with torch.tensor(...) are float64 too
        try:
            terms = decoder_loss(outs, Obb3D(torch.from_numpy(obbs).double()), Pose(torch.from_numpy(sc["T_world_local"]).double()),
                                 torch.from_numpy(sym).double(), matcher=HungarianMatcherModified(cost_class=2, cost_bbox=0.25),
                                 loss_weight=cfg.LOSS_WEIGHT, num_semcls=cfg.NUM_SEMCLS, class_weight=cw)
        finally:
            torch.set_default_dtype(torch.float32)
        loss = terms["total_loss"]
    loss.backward()
    return od, outs, loss, terms


@pytest.mark.parametrize("tag", sorted(MG.GRAD_CASES))
@pytest.mark.parametrize("kind", ["linear", "setloss"])
def test_oracle_autograd_equals_reference_autograd(tag, kind):
    meta, z = G.load_grads()
    c = meta["cases"][tag]
    assert c == MG.GRAD_CASES[tag]
    od, outs, loss, terms = oracle_run(c, kind)
    want = float(z["%s/%s/loss_value" % (tag, kind)])
    # set loss: our y-rotation tables are float32 constants as in the reference's normal (float32-default) run, the fixture's run
    # had float64 ones: 4e-10 relative on the rotation term
    tol_l, tol_g = (1e-12, 1e-9) if kind == "linear" else (2e-9, 1e-7)
    assert abs(float(loss.detach()) - want) < tol_l * max(1.0, abs(want)), (float(loss.detach()), want)
    for k, v in terms.items():
        assert abs(float(v.detach()) - float(z["%s/%s/loss/%s" % (tag, kind, k)])) < tol_l * 10, k
    for i, o in enumerate(outs):                                  # the free-running forward itself
        for k in G.KEYS:
            ref = z["%s/%s/out/it%d_%s" % (tag, kind, i, k)]
            assert np.abs(o[k].detach().numpy() - ref).max() < 1e-10, (i, k)
    names = G.grad_names(z, tag, kind)
    nograd = meta["%s/%s/nograd" % (tag, kind)]
    assert nograd == ["parq_module.decoder.norm.weight", "parq_module.decoder.norm.bias"]
    # exactly the reference's parameters carry a gradient (the oracle's weight dict has no decoder.norm: it is never applied)
    have = sorted(k for k, v in od.W.items() if v.grad is not None)
    assert have == names, set(have) ^ set(names)
    worst = 0.0
    for name in names:
        fro, mx = G.grad_errors(z, tag, kind, name, od.W[name].grad.numpy())
        worst = max(worst, fro, mx)
        assert fro < tol_g and mx < tol_g, (name, fro, mx)
    fro, mx = G.token_grad_errors(z, tag, kind, od.tokens.grad.numpy())
    assert fro < tol_g and mx < tol_g, ("d tokens", fro, mx)
    print("\n%s %s: oracle autograd vs reference autograd, worst relative error %.2e (tokens %.2e)" % (tag, kind, worst, max(fro, mx)))


def test_oracle_ray_pe_autograd_equals_reference_autograd(monkeypatch):
    """g20: the reference's AddRayPE under its own autograd (float64; tokens = features + encoding, tokenised as
    model/parq_lightning.py:72-85) against the oracle's ray_pe + tokenize: loss, a token sample, the four encoder gradients and
    d features."""
    import json
    import os
    z = np.load(os.path.join(G.GOLDEN_DIR, "g20_raype_grads.npz"))
    c = json.loads(bytes(z["meta"]).decode())
    assert c == MG.RAYPE_GRAD_CASE
    Wp, (cam, T_cp, T_wp, T_wl), feat, cot = MG.raype_grad_case_inputs(c)
    W64 = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in Wp.items()}
    f64 = torch.from_numpy(feat).double().requires_grad_(True)
    monkeypatch.setattr(O, "_as_torch", lambda Wd, dtype: Wd)            # leaf tensors passed through unchanged
    enc = O.ray_pe(cam, T_cp, T_wp, T_wl, W64, c["ray_points_scale"], dtype=torch.float64)
    tokens = O.tokenize(f64, enc)
    loss = (tokens * torch.from_numpy(cot).double()).sum()
    loss.backward()
    assert abs(float(loss.detach()) - float(z["loss_value"])) < 1e-9 * abs(float(z["loss_value"]))
    assert np.abs(tokens.detach().numpy()[:, ::11, ::7] - z["tokens_sample"]).max() < 1e-10
    for name, t in W64.items():
        g = t.grad.numpy().reshape(-1)
        if "grad/%s/full" % name in z.files:
            ref = z["grad/%s/full" % name]
            err = np.linalg.norm(g - ref) / np.linalg.norm(ref)
        else:
            ref = z["grad/%s/sample" % name]
            err = max(np.linalg.norm(g[::MG.GRAD_STRIDE] - ref) / np.linalg.norm(ref),
                      abs(np.linalg.norm(g) - z["grad/%s/norm" % name][0]) / z["grad/%s/norm" % name][0])
        assert err < 1e-9, (name, err)
    fg = f64.grad.numpy().reshape(-1)
    assert np.abs(fg[::MG.TOKEN_STRIDE] - z["dfeat/sample"]).max() < 1e-12
