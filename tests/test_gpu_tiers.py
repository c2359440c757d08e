"""GPU tier: what attention mode "split8" does when its peakedness guard trips (include/parq_hip.h parq_set_head_tiers,
parq_amd/decoder.py).  The reference has ONE arithmetic for the cross-attention (fp32, model/transformer_parq.py:377-380); mode "split8"
approximates it inside an error model that needs rows spread over many keys.  The contract tested here: a forward never returns
plausible numbers from outside that model — the flagged heads move to the fp16 x 3 arithmetic of mode "split" (per head, inside the
same forward: two launches over complementary head sets), and the forward that met the rows is either re-run ("sync", and the
first forward of a module) or NaN from the flagged iteration on ("lazy")."""
import warnings

import numpy as np
import pytest
import torch

from parq_amd import synth
from oracle import parq_oracle as O
import golden_util as G
from gpu_util import make_decoder, scene_args, to_np, rel_err

pytestmark = pytest.mark.gpu

WQ = "parq_module.decoder.layers.0.multihead_attn.in_proj_weight"


def _case(scale, heads=None, name="g15_cfg5_shape"):
    """A reference fixture's inputs with the cross-attention query projection scaled (all heads, or the 64 rows of each head in
    `heads`): x 1 spreads every row over thousands of keys, x 4 leaves rows that two or three keys carry."""
    case, z = G.load(name)
    cfg, W, sc = G.inputs(case)
    W = dict(W)
    w = W[WQ].copy()
    if heads is None:
        w[:w.shape[1]] *= scale
    else:
        for h in heads:
            w[64 * h: 64 * h + 64] *= scale
    W[WQ] = w
    return cfg, W, sc, G.forced_refs(z, cfg.TRANSFORMER.SCALE)


def _run(dec, sc):
    with torch.no_grad():
        out = [{k: v.clone() for k, v in o.items()} for o in dec(*scene_args(sc))]
    torch.cuda.synchronize()
    return out


def _split_reference(cfg, W, sc):
    ref = make_decoder(cfg, W)
    ref.attention_mode = "split"
    return _run(ref, sc)


def _worst(a, b):
    return max(rel_err(x[k].cpu().numpy(), y[k].cpu().numpy()) for x, y in zip(a, b) for k in x)


def test_sync_policy_reruns_with_the_flagged_heads_on_the_safe_tier():
    """All four heads of the x 4 fixture are peaked: "sync" moves them all (a module whose heads are all safe runs exactly mode
    "split": bit-identical outputs) and the caller never sees the first attempt."""
    cfg, W, sc, _ = _case(4.0)
    want = _split_reference(cfg, W, sc)
    dec = make_decoder(cfg, W)
    dec.range_check = "sync"
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        got = _run(dec, sc)
    assert dec.attention_mode == "split8" and dec.safe_heads == 0b1111
    for a, b in zip(got, want):
        for k in a:
            assert torch.equal(a[k], b[k]), k
    assert any("too few keys" in str(w.message) for w in caught)


def test_lazy_policy_never_returns_numbers_from_outside_the_error_model():
    """"lazy", past the first-forward check: the forward that meets peaked rows is NaN from the flagged iteration on (here: from
    iteration 0), the next call moves the heads and returns mode "split"'s numbers."""
    cfg, W, sc, _ = _case(4.0)
    want = _split_reference(cfg, W, sc)
    dec = make_decoder(cfg, W)
    dec.range_check = "lazy"               # opt-in (servers): the default is "sync"
    dec._peaky_checked = True              # as if an earlier (spread) forward had passed the first-call check: the lazy path proper
    dec._ensure_packed(torch.device("cuda", torch.cuda.current_device()))
    dec._peaky_checked = True              # (packing the weights re-arms the first-forward check)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        got = _run(dec, sc)
        assert dec.safe_heads == 0 and dec.attention_too_peaked()
        assert all(torch.isnan(v).all() for o in got for k, v in o.items() if k != "coord_pos"), "numbers from outside the error model"
        again = _run(dec, sc)              # polls the pinned word first
    assert dec.safe_heads == 0b1111
    for a, b in zip(again, want):
        for k in a:
            assert torch.equal(a[k], b[k]), k
    assert any("too few keys" in str(w.message) for w in caught)


def test_first_forward_is_checked_under_the_lazy_policy():
    cfg, W, sc, _ = _case(4.0)
    want = _split_reference(cfg, W, sc)
    dec = make_decoder(cfg, W)
    dec.range_check = "lazy"
    assert dec.attention_mode == "split8"
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        got = _run(dec, sc)
    assert dec.safe_heads == 0b1111
    for a, b in zip(got, want):
        for k in a:
            assert torch.equal(a[k], b[k]), k
    assert any("too few keys" in str(w.message) for w in caught)


def test_diffuse_scene_then_peaked_scene_same_module_lazy():
    """peakedness is a property of the scene as much as of the model.  Same module, "lazy": a diffuse scene passes (fast
    tier, numbers), then a scene whose tokens are 4 x larger (scores 4 x larger: peaked rows) — its outputs are NaN or mode "split"'s,
    never plain wrong; the call after that is right."""
    cfg, W, sc, _ = _case(1.0)
    dec = make_decoder(cfg, W)
    dec.range_check = "lazy"
    first = _run(dec, sc)
    assert dec.safe_heads == 0 and all(torch.isfinite(v).all() for o in first for v in o.values())
    sc2 = dict(sc)
    sc2["tokens"] = (sc["tokens"] * 4.0).astype(np.float32)
    want = _split_reference(cfg, W, sc2)
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        got = _run(dec, sc2)
        assert dec.attention_too_peaked(), "the x 4 scene is meant to trip the guard"
        for a, b in zip(got, want):
            for k in a:
                if k == "coord_pos":
                    continue
                assert torch.isnan(a[k]).all(), k           # (iteration 0 already meets peaked rows: everything is NaN)
        again = _run(dec, sc2)
    assert dec.safe_heads != 0
    assert _worst(again[:1], want[:1]) < 2e-5          # (free-running: iteration 0; all heads safe -> bit-identical anyway)


def test_diffuse_scene_then_peaked_scene_same_module_default_policy_returns_mode_split_numbers():
    """VERDICT r05, item 1 (reference behaviour: model/transformer_parq.py:377-380 never returns NaN; eval.py:45-48 consumes every
    snippet's outputs).  The DEFAULT policy on the same sequence — a diffuse scene, then a peaked one, same module, also with the
    forward replayed from its captured graph in between: the peaked scene's outputs are finite and equal mode "split"'s bit for bit
    (all four heads trip and move: a module whose heads are all safe IS mode "split"), in the call that met the rows, not the next."""
    cfg, W, sc, _ = _case(1.0)
    dec = make_decoder(cfg, W)
    assert dec.attention_mode == "split8" and dec.range_check == "sync", "the never-NaN policy is the default"
    for _ in range(3):                                   # (the third call replays the captured forward)
        first = _run(dec, sc)
        assert dec.safe_heads == 0 and all(torch.isfinite(v).all() for o in first for v in o.values())
    sc2 = dict(sc)
    sc2["tokens"] = (sc["tokens"] * 4.0).astype(np.float32)
    want = _split_reference(cfg, W, sc2)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        got = _run(dec, sc2)
    assert dec.safe_heads == 0b1111 and any("too few keys" in str(w.message) for w in caught)
    for a, b in zip(got, want):
        for k in a:
            assert torch.isfinite(a[k]).all(), k
            assert torch.equal(a[k], b[k]), k
    again = _run(dec, sc2)                               # and the calls after it (one of them from a graph) stay there
    again = _run(dec, sc2)
    for a, b in zip(again, want):
        for k in a:
            assert torch.equal(a[k], b[k]), k


def test_validation_step_over_three_snippets_with_one_peaked_snippet_default_policy():
    """The same at module level, as eval.py drives it (eval.py:45-48: `model.validation_step(batch, 0)` per snippet): three snippets
    through parq_amd.PARQ.validation_step with the default policy, the middle one with features x 6 (peaked cross-attention).  No snippet
    loses its detections: snippet 1 is within mode "split8"'s stated distance of a mode-"split" module (fast tier, iteration 0),
    snippets 2 and 3 equal the mode-"split" module bit for bit (the flagged heads moved inside snippet 2's call)."""
    from types import SimpleNamespace as NS
    from parq_amd import PARQ, Camera, Pose
    from gpu_util import dev
    B, V, h, w, Cd, Qn, I = 1, 4, 48, 64, 256, 64, 3
    dcfg = synth.decoder_cfg(dim=Cd, queries=Qn, heads=4, ffn=768, layers=I)
    pcfg = NS(MODEL=NS(TOKENIZER=NS(OUT_CHANNELS=Cd, RAY_POINTS_SCALE=dcfg.TRANSFORMER.SCALE, NUM_SAMPLES=64, MIN_DEPTH=0.25, MAX_DEPTH=5.25),
                       DECODER=dcfg))
    W = synth.make_decoder_weights(dcfg, 811)
    Wp = synth.make_ray_pe_weights(Cd, 812)

    def build(mode):
        model = PARQ(pcfg).eval()
        sd = model.state_dict()
        for k in sd:
            if k.startswith("box3d_decoder."):
                src = k[len("box3d_decoder."):].replace("parq_module.decoder.mlp_heads.", "mlp_heads.")
                sd[k] = torch.from_numpy(W[src]).reshape(sd[k].shape)
            else:
                sd[k] = torch.from_numpy(Wp[k[len("add_ray_pe."):]])
        model.load_state_dict(sd, strict=True)
        model = model.cuda()
        if mode:
            model.box3d_decoder.attention_mode = mode
        return model

    def snippet(seed, gain):
        cam, T_cp, T_wp, T_wl = synth.make_geometry(seed, B, V, h, w)
        feat = synth.normal(seed + 1, "feat", (B, V, Cd, h, w), std=1.0) * np.float32(gain)
        return {"all_features": dev(feat), "camera_feature": Camera(dev(cam)), "T_camera_pseudoCam": Pose(dev(T_cp)),
                "T_world_pseudoCam": Pose(dev(T_wp)), "T_world_local": Pose(dev(T_wl))}
    snippets = [snippet(820, 1.0), snippet(830, 6.0), snippet(840, 1.0)]

    def run(model):
        got = []
        real = model.box3d_decoder.forward
        with torch.no_grad(), warnings.catch_warnings(record=True):
            warnings.simplefilter("always")
            for b in snippets:
                seen = {}
                model.box3d_decoder.forward = lambda *a, _r=real, **k: seen.setdefault("o", _r(*a, **k))
                model.validation_step(dict(b), 0)
                got.append([{k: v.clone() for k, v in o.items()} for o in seen["o"]])
        model.box3d_decoder.forward = real
        torch.cuda.synchronize()
        return got
    model = build(None)
    assert model.box3d_decoder.range_check == "sync" and model.box3d_decoder.attention_mode == "split8"
    got = run(model)
    assert model.box3d_decoder.safe_heads == 0b1111, "the x 6 snippet is meant to trip the guard on every head"
    want = run(build("split"))
    for i in range(3):
        for o in got[i]:
            assert all(torch.isfinite(v).all() for v in o.values()), i
    assert _worst(got[0][:1], want[0][:1]) < 3e-5
    for i in (1, 2):
        for a, b in zip(got[i], want[i]):
            for k in a:
                assert torch.equal(a[k], b[k]), (i, k)


def test_off_policy_returns_numbers_and_reports_the_map():
    cfg, W, sc, _ = _case(4.0)
    dec = make_decoder(cfg, W)
    dec.range_check = "off"
    got = _run(dec, sc)
    assert dec.safe_heads == 0 and dec.attention_too_peaked()
    assert all(torch.isfinite(v).all() for o in got for v in o.values())
    pm = dec.attention_peaked_map()
    assert len(pm) == dec.num_layers and pm[0] != 0 and all(0 <= m < 16 for m in pm)
    assert dec.attention_min_row_sum() < 256


@pytest.mark.parametrize("heads", [[2], [0, 3]])
def test_only_the_peaked_heads_move(heads):
    """Per-head granularity: only the query rows of `heads` are sharpened.  "sync" ends with exactly those heads on the fp16 x 3 tier,
    the others stay on the mode-4 kernel, and the mixed forward is within 2e-5 of mode "split" and of float64 (teacher-forced)."""
    cfg, W, sc, refs = _case(4.0, heads=heads)
    want = _split_reference(cfg, W, sc)
    dec = make_decoder(cfg, W)
    dec.range_check = "sync"
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        got = _run(dec, sc)
    mask = sum(1 << h for h in heads)
    assert dec.safe_heads == mask, bin(dec.safe_heads)
    assert not dec.attention_too_peaked()                    # the last (mixed) run is inside the model
    d = _worst(got[:1], want[:1])            # free-running: later iterations amplify any rounding difference (white-noise features)
    # float64, teacher-forced on the reference's reference points, first two iterations
    od = O.OracleDecoder(cfg, W, synth.SCANNET_MEAN_SIZES, dtype=torch.float64)
    od.prepare(sc["tokens"], sc["camera"], sc["T_camera_pseudoCam"], sc["T_world_pseudoCam"], sc["T_world_local"])
    with torch.no_grad():
        dec.prepare(*scene_args(sc))
        e = 0.0
        for k in range(2):
            exact = od.iterate(torch.from_numpy(refs[k]).double(), k)[0]
            out, _ = dec.iterate(k, torch.from_numpy(refs[k]).cuda())
            e = max(e, max(rel_err(out[key].cpu().numpy(), exact[key].numpy()) for key in G.KEYS))
    print("\nheads %s peaked: mixed tiers vs split %.2e, vs float64 %.2e" % (heads, d, e))
    assert d < 2e-5 and e < 2e-5


@pytest.mark.parametrize("mask,B", [(0b0001, 1), (0b0110, 1), (0b1011, 2), (0b1000, 3)])
def test_mixed_tiers_equal_their_own_kernels_per_head(mask, B):
    """Forced tiers on spread attention (policy "off", nothing trips): in the cross-attention output of a mixed forward the columns
    of a safe head equal mode "split"'s and the columns of a fast head equal mode "split8"'s, up to the summation order of the key
    splits (each launch picks the split count that fills the chip with ITS heads)."""
    cfg = synth.decoder_cfg(dim=256, queries=96, heads=4, ffn=256, layers=2)
    W = synth.make_decoder_weights(cfg, seed=77)
    sc = synth.make_scene(78, B, 3, 24, 32, 256)
    attn, outs = {}, {}
    for mode, m in (("split", 0), ("split8", 0), ("mixed", mask)):
        dec = make_decoder(cfg, W)
        dec.range_check = "off"
        dec.attention_mode = "split" if mode == "split" else "split8"
        dec.safe_heads = m
        with torch.no_grad():
            dec.prepare(*scene_args(sc))
            out, nxt = dec.iterate(0)
            attn[mode] = dec.intermediate("attn").view(B * 96, 256).double().cpu().clone()
            outs[mode] = {k: v.double().cpu() for k, v in out.items()}
            dec.iterate(1)                                        # (a second iteration on the same tiers: sweep direction reversed)
    for h in range(4):
        src = "split" if (mask >> h) & 1 else "split8"
        a, b = attn["mixed"][:, 64 * h: 64 * h + 64], attn[src][:, 64 * h: 64 * h + 64]
        # a safe head: fp16 x 3, only the summation order over the key splits differs (2e-6); a fast head: a different split count moves
        # the reference maximum of every split, hence the fp16 rounding of its probabilities: mode 4's own noise at 2304 keys (2e-4)
        assert float((a - b).abs().max() / b.abs().max()) < (2e-6 if src == "split" else 2e-4), (h, src)
    # 2304 keys: the fast heads carry mode 4's short-row noise (tests/test_gpu_split8.py: 3e-5 .. 2e-4 at 64 .. 2304 keys)
    assert max(float(((outs["mixed"][k] - outs["split"][k]).abs() / outs["split"][k].abs().clamp(min=1)).max()) for k in outs["split"]) < 3e-4


def test_mixed_tiers_at_cfg3_size():
    """BASELINE cfg 3's geometry, one safe head of four (split counts 256 and 86, 393 MB regions per scene and head): the first
    iteration against mode "split"."""
    cfg = synth.decoder_cfg(dim=256, queries=256, heads=4, ffn=768, layers=1)
    W = synth.make_decoder_weights(cfg, seed=2024)
    cam, T_cp, T_wp, T_wl = (torch.from_numpy(a).cuda() for a in synth.make_geometry(3024, 1, 10, 120, 160))
    g = torch.Generator(device="cuda").manual_seed(3024)
    tokens = torch.randn(1, 10 * 120 * 160, 256, device="cuda", generator=g)
    outs = {}
    with torch.no_grad():
        for mode, m in (("split", 0), ("mixed", 0b0100)):
            dec = make_decoder(cfg, W)
            dec.attention_mode = "split" if mode == "split" else "split8"
            dec.range_check = "off"
            dec.safe_heads = m
            outs[mode] = {k: v.double().clone() for k, v in dec(tokens, cam, T_cp, T_wp, T_wl, feat_hw=(120, 160))[0].items()}
            assert not dec.attention_too_peaked()
            dec._ws.clear()
    diff = max(float(((outs["mixed"][k] - outs["split"][k]).abs() / outs["split"][k].abs().clamp(min=1)).max()) for k in outs["split"])
    print("\ncfg-3 size, one safe head: mixed vs split %.2e" % diff)
    assert diff < 2e-5


def test_reset_attention_tiers():
    cfg, W, sc, _ = _case(4.0)
    dec = make_decoder(cfg, W)
    dec.range_check = "sync"
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        _run(dec, sc)
        assert dec.safe_heads == 0b1111
        dec.reset_attention_tiers()
        assert dec.safe_heads == 0
        _run(dec, sc)
    assert dec.safe_heads == 0b1111


def test_a_head_returns_to_the_fast_tier_after_calm_forwards_under_the_sync_policy():
    """`tier_return_after` (range_check = "sync" only): a head that was moved to the fp16 x 3 tier by a peaked scene returns after N
    consecutive forwards in which all of its rows spread again — and moves back at once (re-run, never NaN) when the peaked scene comes back."""
    cfg, W, sc, _ = _case(1.0)
    dec = make_decoder(cfg, W)
    dec.range_check = "sync"
    dec.tier_return_after = 2
    dec.tier_return_margin = 1.0                        # (this fixture's spread rows sit at sums of 300 .. 1000: margin 4 would keep the heads)
    sc_peaked = dict(sc)
    sc_peaked["tokens"] = (sc["tokens"] * 4.0).astype(np.float32)
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        _run(dec, sc_peaked)
        moved = dec.safe_heads
        assert moved != 0
        mins = dec.attention_min_row_sum(per_head=True)
        assert len(mins) == 4 and all(m is not None for m in mins)
        _run(dec, sc)                                   # calm forward 1: still on the safe tier
        assert dec.safe_heads == moved
        assert min(m for m in dec.attention_min_row_sum(per_head=True)) >= 256
        _run(dec, sc)                                   # calm forward 2: the heads return
        assert dec.safe_heads == 0
        out = _run(dec, sc)                             # ... and the fast tier runs clean
        assert dec.safe_heads == 0 and not dec.attention_too_peaked()
        assert all(torch.isfinite(v).all() for o in out for v in o.values())
        got = _run(dec, sc_peaked)                      # the peaked scene again: re-run with the heads moved, numbers not NaN
        assert dec.safe_heads != 0
        assert all(torch.isfinite(v).all() for o in got for v in o.values())


def test_train_split8_step_on_peaked_attention_is_rerun_in_mode_split():
    """`train_split8` (opt-in): a training forward whose mode-4 heads meet peaked rows is poisoned like an inference forward; forward_train
    sees it before returning (one host synchronisation per step on this path), moves the heads and re-runs the step — with any safe head a
    training step runs as mode "split" — so outputs are finite and the gradients are those of a mode-"split" step (same dropout seed)."""
    B, V, h, w, Q, dim, I = 1, 2, 32, 36, 24, 256, 2
    cfg = synth.decoder_cfg(dim=dim, queries=Q, heads=4, ffn=128, layers=I, dropout=0.0)
    W = synth.make_decoder_weights(cfg, 71, damped=True)
    wq = W[WQ].copy()
    wq[:dim] *= 6.0                                                           # 2304 keys, sharpened: rows on a handful of keys
    W[WQ] = wq
    sc = synth.make_scene(72, B, V, h, w, dim, smooth=True)
    ncls = cfg.NUM_SEMCLS + 1
    cots = {"pred_logits": synth.normal(73, "cl", (I, B, Q, ncls)), "center_unnormalized": synth.normal(74, "cc", (I, B, Q, 3)),
            "size_unnormalized": synth.normal(75, "cs", (I, B, Q, 3)), "ortho6d": synth.normal(76, "cr", (I, B, Q, 6))}
    res = {}
    for opt in (True, False):
        dec = make_decoder(cfg, W)
        dec.train_split8 = opt
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            outs = dec.forward_train(*scene_args(sc))
            grads, d_tok = dec.backward({k: torch.from_numpy(v) for k, v in cots.items()})
        torch.cuda.synchronize()
        assert all(torch.isfinite(o[k]).all() for o in outs for k in o)
        if opt:
            assert dec.safe_heads != 0 and any("too few keys" in str(x.message) for x in caught)
        else:
            assert dec.safe_heads == 0
        res[opt] = {k: v.double().cpu() for k, v in grads.items()}
        res[opt]["__tokens__"] = d_tok.double().cpu()
    for name, a in res[True].items():
        b = res[False][name]
        assert torch.isfinite(a).all(), name
        if float(b.norm()) > 0:
            assert float((a - b).norm()) / float(b.norm()) < 1e-4, name      # both steps ran the fp16 x 3 kernels (atomics: ~1e-5 run to run)
