"""GPU tier, the benchmark's own configuration (BASELINE cfg 3: 10 views 480x640 -> 120x160 feature maps, N = 192 000
tokens, 256 queries, 8 iterations, d = 256, 4 heads, FFN 768) held to the reference:

  * forward, both attention arithmetics, all 8 iterations teacher-forced, against golden g14_cfg3 captured from the real
    reference at exactly this configuration AND against the float64 oracle run on the box's host cores (1e-4 on
    |a-b| / max(1,|b|); where the reference's own fp32 run sits further than that from its float64 evaluation, its
    deviation bounds the comparison with the golden);
  * backward (training forward in the default split-precision mode + HIP backward chain) against float64 autograd of the
    oracle at the full key count;
  * the cfg-4 per-GPU shard (4 scenes in one call): gradients == sum of four single-scene runs.

Reference lines: model/transformer_parq.py:283-337 (loop), :365-386 (layer), model/parq_decoder.py:134-163."""
import os

import numpy as np
import pytest
import torch

from parq_amd import synth
from oracle import parq_oracle as O
import golden_util as G
from gpu_util import dev, make_decoder, scene_args, to_np, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4
V, FH, FW, Q, DIM, ITERS = 10, 120, 160, 256, 256, 8
GKEYS = ("pred_logits", "center_unnormalized", "size_unnormalized", "ortho6d")


@pytest.fixture(scope="module")
def g14():
    case, z = G.load("g14_cfg3")
    cfg, W, sc = G.inputs(case)
    assert (case["V"], case["h"], case["w"], cfg.NUM_QUERIES, cfg.TRANSFORMER.DEC_LAYERS) == (V, FH, FW, Q, ITERS)
    return cfg, W, sc, z, scene_args(sc)


@pytest.fixture(scope="module")
def truth(g14):
    """The float64 oracle (pinned to the reference's float64 run by golden g7, tests/test_oracle_golden.py) evaluated on the
    box's host cores for all 8 iterations at the golden's own per-iteration reference points."""
    cfg, W, sc, z, args = g14
    od = O.OracleDecoder(cfg, W, synth.SCANNET_MEAN_SIZES, dtype=torch.float64)
    od.prepare(sc["tokens"], sc["camera"], sc["T_camera_pseudoCam"], sc["T_world_pseudoCam"], sc["T_world_local"])
    refs = G.forced_refs(z, cfg.TRANSFORMER.SCALE)
    out = []
    with torch.no_grad():
        for k in range(ITERS):
            t, _, _ = od.iterate(torch.from_numpy(refs[k]).double(), k)
            out.append({key: v.numpy() for key, v in t.items()})
    return out


def _masked_err(a, b, z, k, key):
    vm, cm = G.safe_mask(z, k)
    err = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / np.maximum(1.0, np.abs(np.asarray(b, np.float64)))
    m = vm & cm if key == "size_unnormalized" else (np.ones_like(vm) if key == "coord_pos" else vm)
    e = err[m]
    return float(e.max()) if e.size else 0.0


@pytest.mark.parametrize("mode", ["split8", "split", "fp32"])
def test_cfg3_golden_teacher_forced(g14, truth, mode):
    """Every iteration of the headline configuration (64 key splits x 47 stages per head in the split modes — "split8", the default
    here: cross terms as MX-scaled fp8 products, and "split": three fp16 products; the exact-fp32 MFMA kernels are held to the
    same vectors), teacher-forced, against
      (a) the float64 evaluation of the reference's algorithm: 1e-4 on every output (measured: ~2e-6), and
      (b) golden g14 = the reference's own fp32 CPU run: 1e-4, or — on the outputs where the reference's fp32 run itself
          sits further than 9e-5 from its float64 evaluation at this size (N = 192 000 white-noise tokens: up to 1.4e-4 on
          size / logits, measured in this test) — that deviation + 1e-5.  No implementation that is not bit-compatible with
          the reference's rounding sequence can be closer to g14 than g14 is to the truth."""
    cfg, W, sc, z, args = g14
    dec = make_decoder(cfg, W)
    dec.attention_mode = mode
    dec.prepare(*args, feat_hw=(FH, FW))
    refs = G.forced_refs(z, cfg.TRANSFORMER.SCALE)
    worst_truth = worst_gold = worst_ref = 0.0
    # per-(iteration, output) table of the three errors and of the bound that applied: written next to the test output so that
    # which comparisons took the relaxed branch is on record (gpurun_out/parity_tables/; profiles/r03_cfg3_parity_table_<mode>.txt is a copy of one run)
    rows = ["# g14_cfg3 [%s] teacher-forced, error = max |a - b| / max(1, |b|) over the decision-safe elements" % mode,
            "# it output               HIP-vs-float64  HIP-vs-reference-fp32  reference-fp32-vs-float64  bound-on-HIP-vs-reference  branch"]
    for k in range(ITERS):
        out, _ = dec.iterate(k, dev(refs[k]))
        o = to_np(out)
        for key in G.KEYS:
            g = z["it%d_%s" % (k, key)]
            e_truth = _masked_err(o[key], truth[k][key], z, k, key)
            e_gold = _masked_err(o[key], g, z, k, key)
            e_ref = _masked_err(g, truth[k][key], z, k, key)              # the reference's own fp32-vs-float64 deviation
            worst_truth, worst_gold, worst_ref = max(worst_truth, e_truth), max(worst_gold, e_gold), max(worst_ref, e_ref)
            relaxed = not (e_ref < 9e-5)
            bound = e_ref + 1e-5 if relaxed else TOL
            rows.append("%4d %-20s %14.3e  %21.3e  %25.3e  %25.3e  %s" % (k, key, e_truth, e_gold, e_ref, bound, "relaxed" if relaxed else "1e-4"))
            assert e_truth < TOL, (mode, k, key, e_truth)
            assert e_gold < bound, (mode, k, key, e_gold, e_ref)
    rows.append("# relaxed comparisons: %d of %d; HIP-vs-reference above 1e-4 on %d of them" % (
        sum(r.endswith("relaxed") for r in rows), ITERS * len(G.KEYS),
        sum(r.endswith("relaxed") and float(r.split()[3]) >= TOL for r in rows[2:])))
    print("\n" + "\n".join(rows))
    try:
        d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "parity_tables")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "cfg3_parity_table_%s.txt" % mode), "w") as f:
            f.write("\n".join(rows) + "\n")
    except OSError:
        pass
    print("\ng14_cfg3 [%s]: HIP vs float64 oracle %.2e | HIP vs reference fp32 golden %.2e | reference fp32 vs float64 %.2e"
          % (mode, worst_truth, worst_gold, worst_ref))
    assert worst_truth <= worst_ref                                       # closer to the truth than the reference's fp32 run
    assert not dec.fp16_range_exceeded()


def test_cfg3_forward_api_first_iteration_matches_golden(g14):
    """The public forward() at the headline size: iteration 0 starts from sigmoid(refpoint.weight) on both sides, so it needs no
    teacher forcing (later free-running iterations on white-noise features are chaotic in the reference itself: SURVEY.md App. D)."""
    cfg, W, sc, z, args = g14
    with torch.no_grad():
        outs = make_decoder(cfg, W)(*args, feat_hw=(FH, FW))
    assert len(outs) == ITERS
    G.compare(to_np(outs[0]), z, 0, TOL, what="g14 forward()")


def _cotangents(seed, I, B):
    ncls = 10
    return {"pred_logits": synth.normal(seed, "cl", (I, B, Q, ncls)), "center_unnormalized": synth.normal(seed + 1, "cc", (I, B, Q, 3)),
            "size_unnormalized": synth.normal(seed + 2, "cs", (I, B, Q, 3)), "ortho6d": synth.normal(seed + 3, "cr", (I, B, Q, 6))}


def test_cfg3_size_backward_matches_float64_oracle_autograd():
    """Training forward (default split-precision attention) + HIP backward chain at the full key count (N = 192 000, 256
    queries, d = 256; 2 iterations so the one-launch batched cross-attention backward and the split-precision dW_kv run)
    against float64 autograd of the oracle on the host cores.  Smooth features + damped centre head keep the free-running
    second iteration comparable.  Tolerances as tests/test_gpu_backward.py: Frobenius-relative 2e-3,
    max-norm-relative 2e-2 per tensor."""
    I = 2
    cfg = synth.decoder_cfg(dim=DIM, queries=Q, heads=4, ffn=768, layers=I, dropout=0.0)
    W = synth.make_decoder_weights(cfg, 441, damped=True)
    sc = synth.make_scene(442, 1, V, FH, FW, DIM, smooth=True)
    cots = _cotangents(443, I, 1)
    # float64 autograd of the oracle
    od = O.OracleDecoder(cfg, W, synth.SCANNET_MEAN_SIZES, dtype=torch.float64)
    for k in od.W:
        od.W[k].requires_grad_(True)
    od.prepare(sc["tokens"], sc["camera"], sc["T_camera_pseudoCam"], sc["T_world_pseudoCam"], sc["T_world_local"])
    od.tokens.requires_grad_(True)
    ref = od.initial_ref()
    loss = 0.0
    oouts = []
    for k in range(I):
        out, nxt, _ = od.iterate(ref, k)
        oouts.append({key: out[key].detach().numpy() for key in GKEYS})
        for key in GKEYS:
            loss = loss + (out[key] * torch.from_numpy(cots[key][k]).double()).sum()
        ref = nxt.detach()
    loss.backward()
    want = {k: v.grad.numpy() for k, v in od.W.items() if v.grad is not None}
    want_tok = od.tokens.grad.numpy()
    del loss, out, nxt

    dec = make_decoder(cfg, W).train()
    assert dec._train_mode() == "split"
    outs = dec.forward_train(*scene_args(sc), feat_hw=(FH, FW))
    for k in range(I):
        for key in GKEYS:
            assert rel_err(outs[k][key].cpu().numpy(), oouts[k][key]) < TOL, (k, key)
    grads, d_tok = dec.backward({k: torch.from_numpy(v) for k, v in cots.items()})
    worst = {}
    for name, g in grads.items():
        if name not in want:
            assert float(g.abs().max()) == 0.0, name
            continue
        d = g.cpu().numpy().astype(np.float64) - want[name]
        worst[name] = (np.linalg.norm(d) / max(np.linalg.norm(want[name]), 1e-9), np.abs(d).max() / max(np.abs(want[name]).max(), 1e-9))
    print("\ncfg-3 size backward, worst (frobenius, max):", sorted(worst.items(), key=lambda kv: -kv[1][0])[:5])
    bad = {k: v for k, v in worst.items() if not (v[0] < 2e-3 and v[1] < 2e-2)}
    assert not bad, bad
    assert len(worst) >= 30
    dt = d_tok.cpu().numpy().astype(np.float64) - want_tok
    terr = (np.linalg.norm(dt) / np.linalg.norm(want_tok), np.abs(dt).max() / np.abs(want_tok).max())
    print("token gradient error (frobenius, max) %.3e %.3e" % terr)
    assert terr[0] < 2e-3 and terr[1] < 2e-2, terr


def test_cfg4_shard_four_scenes_gradients_equal_sum_of_single_scene_runs():
    """BASELINE cfg 4's per-GPU shard: 4 scenes of the cfg-3 geometry in ONE training forward/backward (8 iterations, the
    one-launch cross-attention backward over all (iteration, scene, head) tiles) against the four scenes run one at a time:
    the loss is a sum over scenes, so every weight gradient of the batched run must equal the sum of the four single-scene
    gradients and the token gradients must be the per-scene ones.  Different key-split counts and atomics order at B = 4 vs
    B = 1 perturb the free-running 8-iteration trajectories by rounding -> Frobenius-relative 2e-3 (the bar of the
    backward-vs-oracle tests; the measured figure is printed); dropout off (its masks are indexed by the row within the batch)."""
    Bn, I = 4, ITERS
    cfg = synth.decoder_cfg(dim=DIM, queries=Q, heads=4, ffn=768, layers=I, dropout=0.0)
    W = synth.make_decoder_weights(cfg, 451, damped=True)
    sc = synth.make_scene(452, Bn, V, FH, FW, DIM, smooth=True)
    cots = _cotangents(453, I, Bn)
    dec = make_decoder(cfg, W).train()
    args = scene_args(sc)
    outs = dec.forward_train(*args, feat_hw=(FH, FW))
    outs = [{k: v.clone() for k, v in o.items()} for o in outs]
    grads, d_tok = dec.backward({k: torch.from_numpy(v) for k, v in cots.items()})
    grads = {k: v.double() for k, v in grads.items()}
    d_tok = d_tok.clone()
    acc = {k: torch.zeros_like(v) for k, v in grads.items()}
    GTOL, worst_tok = 2e-3, 0.0          # Frobenius-relative, the bar of the backward-vs-oracle tests (free-running trajectories differ by rounding)
    for s in range(Bn):
        one = tuple(a[s:s + 1].contiguous() for a in args)
        o1 = dec.forward_train(*one, feat_hw=(FH, FW))
        for k in range(I):
            for key in GKEYS:
                # free-running over 8 iterations with different key-split counts at B = 4 and B = 1: rounding differences grow
                assert rel_err(o1[k][key][0].cpu().numpy(), outs[k][key][s].cpu().numpy()) < 1e-3, (s, k, key)
        g1, t1 = dec.backward({k: torch.from_numpy(np.ascontiguousarray(v[:, s:s + 1])) for k, v in cots.items()})
        for k, v in g1.items():
            acc[k] += v.double()
        rel = float((t1[0].double() - d_tok[s].double()).norm() / t1[0].double().norm())
        worst_tok = max(worst_tok, rel)
        assert rel < GTOL, ("tokens", s, rel)
    worst = ("", 0.0)
    for name, a in grads.items():
        nb = float(acc[name].norm())
        if nb == 0:
            assert float(a.norm()) == 0, name
            continue
        rel = float((a - acc[name]).norm()) / nb
        worst = max(worst, (name, rel), key=lambda t: t[1])
        assert rel < GTOL, (name, rel)
    print("\ncfg-4 shard (B=4) vs sum of four B=1 runs: worst relative difference %.2e (%s); tokens %.2e" % (worst[1], worst[0], worst_tok))
