"""GPU tier, SURVEY.md §8f-2: the set loss (model/parq_decoder.py:264-370, utils/matcher.py:52-115) on CUDA tensors — the
tensors a training step really hands it — against golden g10 captured from the reference's PARQDecoder.loss, and the
BASELINE cfg-5 end-to-end leg (20 views 960x1280, 512 queries, 12 iterations, fp16 attention -> Hungarian matcher + box /
class / rotation losses) at full size."""
import json
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
from make_golden import LOSS_CASE, loss_case_inputs  # noqa: E402  (inputs regenerated from the seed: data only)
from parq_amd import Obb3D, Pose, synth  # noqa: E402
from parq_amd.loss import HungarianMatcherModified, decoder_loss, decoder_loss_batched  # noqa: E402
from gpu_util import dev, make_decoder  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "g10_loss.npz")
TERMS = ("center_loss", "size_loss", "rot_loss", "cat_loss", "total_loss")


@pytest.mark.parametrize("fn", [decoder_loss, decoder_loss_batched])
@pytest.mark.parametrize("tag,sym_on", [("sym", True), ("nosym", False)])
def test_loss_on_cuda_tensors_matches_reference_golden(tag, sym_on, fn):
    c = LOSS_CASE
    z = np.load(GOLD)
    assert json.loads(bytes(z["meta"]).decode()) == json.loads(json.dumps(c))
    outs, obbs, T_wl, sym = loss_case_inputs(c)
    touts = [{k: torch.from_numpy(v).cuda().requires_grad_(k != "coord_pos") for k, v in o.items()} for o in outs]
    cw = torch.ones(10)
    cw[9] = 0.1                                              # PARQDecoder keeps the class weights on the host (parq_decoder.py:46-48)
    np.random.seed(c["np_seed"])
    matcher = HungarianMatcherModified(cost_class=2, cost_bbox=0.25)
    got = fn(touts, Obb3D(torch.from_numpy(obbs).cuda()), Pose(torch.from_numpy(T_wl).cuda()),
             torch.from_numpy(sym).cuda() if sym_on else None, matcher=matcher, loss_weight=[5.0, 5.0, 5.0, 1.0],
             num_semcls=9, class_weight=cw)
    for k in TERMS:
        want = float(z["%s_%s" % (tag, k)])
        assert got[k].is_cuda
        assert abs(float(got[k]) - want) < 2e-5 * max(1.0, abs(want)), (k, float(got[k]), want)
    assert matcher.last_valid_bs == c["I"] * c["B"]
    got["total_loss"].backward()                             # the graph lives on the device end to end
    assert all(torch.isfinite(o[k].grad).all() for o in touts for k in o if k != "coord_pos")


def test_decoder_loss_method_on_cuda_matches_golden():
    """Through PARQDecoder.loss itself (the call model/parq_lightning.py:92 makes)."""
    c = LOSS_CASE
    z = np.load(GOLD)
    outs, obbs, T_wl, sym = loss_case_inputs(c)
    cfg = synth.decoder_cfg(dim=64, queries=c["Q"], heads=1, ffn=64, layers=c["I"])
    from parq_amd.decoder import PARQDecoder
    dec = PARQDecoder(cfg).cuda()
    touts = [{k: torch.from_numpy(v).cuda() for k, v in o.items()} for o in outs]
    for batched in (True, False):
        dec.loss_batched = batched
        np.random.seed(c["np_seed"])
        got = dec.loss(touts, Obb3D(torch.from_numpy(obbs).cuda()), Pose(torch.from_numpy(T_wl).cuda()), torch.from_numpy(sym).cuda())
        for k in TERMS:
            want = float(z["sym_%s" % k])
            assert abs(float(got[k]) - want) < 2e-5 * max(1.0, abs(want)), (batched, k)


def test_cfg5_full_size_fp16_forward_and_set_loss_end_to_end():
    """BASELINE cfg 5: 20 views 960x1280 -> 240x320 feature maps (N = 1 536 000 tokens), 512 queries, 12 iterations, fp16
    cross-attention, then the Hungarian matcher + box / class / rotation losses on the 12 output dicts.  The CPU oracle
    cannot run this size; checked here: every output and every loss term finite, no fp16 range overflow, valid_bs = 12
    (every iteration matched: the scene has boxes), and the batched loss equals the reference-ordered per-pair loop."""
    I, Qn, Vn, h, w = 12, 512, 20, 240, 320
    cfg = synth.decoder_cfg(dim=256, queries=Qn, heads=4, ffn=768, layers=I)
    W = synth.make_decoder_weights(cfg, 551, damped=True)
    dec = make_decoder(cfg, W)
    dec.attention_mode = "fp16"
    cam, T_cp, T_wp, T_wl = synth.make_geometry(552, 1, Vn, h, w)
    g = torch.Generator(device="cuda").manual_seed(553)
    tokens = torch.randn(1, Vn * h * w, 256, device="cuda", generator=g)
    outs = dec(tokens, dev(cam), dev(T_cp), dev(T_wp), dev(T_wl), feat_hw=(h, w))
    assert len(outs) == I
    for o in outs:
        for key, v in o.items():
            assert torch.isfinite(v).all(), key
    assert not dec.fp16_range_exceeded()
    obbs, sym = synth.make_boxes(554, 1, 14)
    res = {}
    for batched in (True, False):
        dec.loss_batched = batched
        np.random.seed(3)
        res[batched] = dec.loss(outs, Obb3D(dev(obbs)), Pose(dev(T_wl)), dev(sym))
        assert dec._matcher.last_valid_bs == I
        for k in TERMS:
            assert torch.isfinite(res[batched][k]).all() and float(res[batched][k]) >= 0, (batched, k)
    for k in TERMS:
        a, b = float(res[True][k]), float(res[False][k])
        assert abs(a - b) < 1e-5 * max(1.0, abs(b)), (k, a, b)
    print("\ncfg5 end-to-end loss:", {k: float(v) for k, v in res[True].items()})
    del tokens
    torch.cuda.empty_cache()


def test_cfg5_full_size_fp16_training_step_end_to_end():
    """BASELINE cfg 5 as a TRAINING step: 20 views 960x1280 (N = 1 536 000 tokens), 512 queries, 12 iterations, fp16
    cross-attention with dropout 0.1, Hungarian matcher + box / class / rotation losses, HIP backward through all 12 iterations
    (one batched cross-attention backward over 6000 key blocks), gradients for every tensor and for the tokens.  No CPU oracle
    runs this size: finiteness, non-zero gradients everywhere the small-size parity tests see them, loss terms finite."""
    I, Qn, Vn, h, w = 12, 512, 20, 240, 320
    cfg = synth.decoder_cfg(dim=256, queries=Qn, heads=4, ffn=768, layers=I, dropout=0.1)
    W = synth.make_decoder_weights(cfg, 651, damped=True)
    dec = make_decoder(cfg, W).train()
    dec.attention_mode = "fp16"
    cam, T_cp, T_wp, T_wl = synth.make_geometry(652, 1, Vn, h, w)
    g = torch.Generator(device="cuda").manual_seed(653)
    tokens = torch.randn(1, Vn * h * w, 256, device="cuda", generator=g).requires_grad_(True)
    obbs, sym = synth.make_boxes(654, 1, 14)
    torch.manual_seed(5)
    np.random.seed(7)
    outs = dec(tokens, dev(cam), dev(T_cp), dev(T_wp), dev(T_wl), feat_hw=(h, w))
    assert dec._mode_set == "fp16" and len(outs) == I
    losses = dec.loss(outs, Obb3D(dev(obbs)), Pose(dev(T_wl)), dev(sym))
    assert dec._matcher.last_valid_bs == I
    for k in TERMS:
        assert torch.isfinite(losses[k]).all(), k
    losses["total_loss"].backward()
    torch.cuda.synchronize()
    assert not dec.fp16_range_exceeded()
    n_grad = 0
    for name, p in dec.named_parameters():
        if "decoder.norm." in name:
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
        n_grad += int(float(p.grad.abs().max()) > 0)
    assert n_grad >= 40, n_grad
    assert torch.isfinite(tokens.grad).all() and float(tokens.grad.abs().max()) > 0
    print("\ncfg5 fp16 training step: loss %.4f, |d tokens| %.3e, %d tensors with gradient" % (float(losses["total_loss"]), float(tokens.grad.norm()), n_grad))
    del tokens, outs, losses
    dec._train_ws = None
    torch.cuda.empty_cache()


def _train_step_loss(dec, sc_t, obbs, T_wl, sym, seed):
    np.random.seed(seed)
    torch.manual_seed(seed)                                  # the dropout seed of the forward is drawn from torch's generator
    dec.zero_grad(set_to_none=True)
    outs = dec(*sc_t)
    terms = dec.loss(outs, obbs, T_wl, sym)
    terms["total_loss"].backward()
    return {k: float(v) for k, v in terms.items()}, {n: p.grad.clone() for n, p in dec.named_parameters() if p.grad is not None}


def test_loss_matching_overlapped_with_the_forward_gives_the_same_step():
    """PARQDecoder.loss on the outputs of the module's own training forward matches iteration k on the host while the device still
    runs iterations k+1.. (parq_wait_iteration, matcher inputs on a side stream).  Same seeds -> same matches, same loss terms,
    same gradients as with the synchronous order (overlap_loss_matching = False)."""
    cfg = synth.decoder_cfg(dim=256, queries=64, heads=4, ffn=256, layers=4, dropout=0.1)
    W = synth.make_decoder_weights(cfg, 811, damped=True)
    sc = synth.make_scene(812, 2, 3, 32, 36, 256)
    sc_t = [dev(sc[k]) for k in ("tokens", "camera", "T_camera_pseudoCam", "T_world_pseudoCam", "T_world_local")]
    obbs, sym = synth.make_boxes(813, 2, 9)
    obbs, sym, T_wl = Obb3D(dev(obbs)), dev(sym), Pose(sc_t[4])
    res = {}
    for overlap in (True, False):
        dec = make_decoder(cfg, W).train()
        dec.overlap_loss_matching = overlap
        res[overlap] = _train_step_loss(dec, sc_t, obbs, T_wl, sym, 5)
        assert (dec._train_pending is None)                  # resolved by loss() / never deferred
    (ta, ga), (tb, gb) = res[True], res[False]
    for k in TERMS:
        assert abs(ta[k] - tb[k]) <= 1e-6 * max(1.0, abs(tb[k])), (k, ta[k], tb[k])
    assert set(ga) == set(gb)
    for n in ga:
        den = float(gb[n].norm())
        assert float((ga[n] - gb[n]).norm()) <= 1e-4 * max(den, 1e-6), n      # float atomics in the backward: not bit-identical


def test_targets_enqueued_behind_a_long_kernel_are_waited_for_by_the_overlapped_loss():
    """The overlapped loss reads the caller's targets on a SIDE stream: targets that were produced asynchronously on
    the main stream before the forward — here a pinned-memory copy queued behind ~20 ms of matrix products, into device buffers
    that still hold other boxes — must be complete before parse_target reads them.  The side stream waits for the event recorded
    at the entry of the training forward; the step equals the one with synchronously prepared targets."""
    cfg = synth.decoder_cfg(dim=256, queries=64, heads=4, ffn=256, layers=4, dropout=0.0)
    W = synth.make_decoder_weights(cfg, 831, damped=True)
    sc = synth.make_scene(832, 2, 3, 32, 36, 256)
    sc_t = [dev(sc[k]) for k in ("tokens", "camera", "T_camera_pseudoCam", "T_world_pseudoCam", "T_world_local")]
    obbs_np, sym_np = synth.make_boxes(833, 2, 9)
    wrong_np, wrong_sym = synth.make_boxes(834, 2, 3)
    T_wl = Pose(sc_t[4])
    dec = make_decoder(cfg, W).train()
    assert dec.overlap_loss_matching
    want, _ = _train_step_loss(dec, sc_t, Obb3D(dev(obbs_np)), T_wl, dev(sym_np), 5)
    torch.cuda.synchronize()
    # device buffers hold OTHER boxes; the right ones arrive by an asynchronous copy behind a long main-stream kernel
    obbs_d, sym_d = dev(wrong_np), dev(wrong_sym)
    obbs_h, sym_h = torch.from_numpy(obbs_np).pin_memory(), torch.from_numpy(sym_np).pin_memory()
    big = torch.randn(8192, 8192, device="cuda")
    torch.cuda.synchronize()
    for _ in range(12):
        big = (big @ big).clamp_(-1.0, 1.0)                 # ~20 ms of queue in front of the copies
    obbs_d.copy_(obbs_h, non_blocking=True)
    sym_d.copy_(sym_h, non_blocking=True)
    got, _ = _train_step_loss(dec, sc_t, Obb3D(obbs_d), T_wl, sym_d, 5)
    for k in TERMS:
        assert abs(got[k] - want[k]) <= 1e-6 * max(1.0, abs(want[k])), (k, got[k], want[k])


def test_deferred_range_check_is_repaired_by_loss_and_raised_by_backward_otherwise():
    """With overlap_loss_matching the autograd forward does not wait for the device; a forward whose tokens leave the fp16 range
    (NaN outputs) is re-run with the exact fp32 kernels INSIDE loss() (finite loss, finite gradients); if the outputs were
    consumed by something else, backward() raises instead of handing NaN gradients to the optimizer."""
    import warnings
    cfg = synth.decoder_cfg(dim=256, queries=32, heads=4, ffn=128, layers=2, dropout=0.0)
    W = synth.make_decoder_weights(cfg, 821, damped=True)
    sc = synth.make_scene(822, 1, 2, 32, 36, 256)
    sc["tokens"] = (sc["tokens"] * np.float32(2e4)).astype(np.float32)
    sc_t = [dev(sc[k]) for k in ("tokens", "camera", "T_camera_pseudoCam", "T_world_pseudoCam", "T_world_local")]
    obbs, sym = synth.make_boxes(823, 1, 6)
    obbs, sym, T_wl = Obb3D(dev(obbs)), dev(sym), Pose(sc_t[4])
    dec = make_decoder(cfg, W).train()
    assert dec._train_mode() == "split" and dec.overlap_loss_matching
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        terms, grads = _train_step_loss(dec, sc_t, obbs, T_wl, sym, 7)
    assert any("fp16 range" in str(r.message) for r in rec)
    assert dec.attention_mode == "fp32" and all(np.isfinite(v) for v in terms.values())
    assert all(torch.isfinite(g).all() for g in grads.values())
    # outputs consumed outside loss(): backward refuses
    dec2 = make_decoder(cfg, W).train()
    outs = dec2(*sc_t)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(RuntimeError, match="fp16 operand range"):
            sum(o["pred_logits"].sum() for o in outs).backward()


@pytest.mark.parametrize("sym_on", [True, False])
def test_device_set_loss_kernel_matches_the_torch_expression_terms_and_gradients(sym_on):
    """parq_set_loss (three launches: terms + d term / d output) against the torch expression of decoder_loss_batched on the same
    device tensors and the same matching: every term within 2e-6 relative, every output gradient within 1e-5 of its norm —
    including the Gram-Schmidt backward of the rotation term and the symmetry-candidate minimum (classes 1, 2, 3 present)."""
    from parq_amd import loss as L
    I, B, Q, ncls = 3, 2, 48, 19
    rng = np.random.default_rng(901)
    obbs, sym = synth.make_boxes(903, B, 10)
    assert set(np.unique(sym[sym >= 0]).astype(int).tolist()) >= {1, 2, 3}
    T_wl = synth.make_geometry(904, B, 2, 8, 8)[3]
    base = {"pred_logits": rng.normal(size=(I, B, Q, ncls)), "center_unnormalized": rng.normal(size=(I, B, Q, 3)) * 2,
            "size_unnormalized": rng.random((I, B, Q, 3)) + 0.2, "ortho6d": rng.normal(size=(I, B, Q, 6)),
            "coord_pos": rng.normal(size=(I, B, Q, 3)) * 2}
    cw = torch.ones(ncls); cw[ncls - 1] = 0.1
    res = {}
    for fused in (True, False):
        L.DEVICE_SET_LOSS = fused
        try:
            leaves = {k: torch.from_numpy(v.astype(np.float32)).cuda().requires_grad_(k != "coord_pos") for k, v in base.items()}
            outs = [{k: t[i] for k, t in leaves.items()} for i in range(I)]
            np.random.seed(11)
            terms = decoder_loss_batched(outs, Obb3D(dev(obbs)), Pose(dev(T_wl)), dev(sym) if sym_on else None,
                                         matcher=HungarianMatcherModified(cost_class=2, cost_bbox=0.25), loss_weight=[2.0, 1.5, 0.7, 1.2],
                                         num_semcls=ncls - 1, class_weight=cw)
            (terms["center_loss"] * 1.0 + terms["size_loss"] * 0.5 + terms["rot_loss"] * 2.0 + terms["cat_loss"] * 1.5).backward()
            res[fused] = ({k: float(terms[k]) for k in TERMS}, {k: leaves[k].grad.clone() for k in L._DIFF_KEYS})
        finally:
            L.DEVICE_SET_LOSS = True
    (ta, ga), (tb, gb) = res[True], res[False]
    for k in TERMS:
        assert abs(ta[k] - tb[k]) <= 2e-6 * max(1.0, abs(tb[k])), (k, ta[k], tb[k])
    for k in ga:
        den = float(gb[k].norm())
        assert den > 0 and float((ga[k] - gb[k]).norm()) <= 1e-5 * den, (k, float((ga[k] - gb[k]).norm()), den)
