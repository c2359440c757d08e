"""GPU tier: the captured forward (include/parq_hip.h parq_forward_capture).  The reference's loop stalls
the host every iteration (model/transformer_parq.py:135,301; utils/parq_utils.py:96-98); this path enqueues ~90 launches per forward
from Python, and from its second forward with the same (shape, stream, weights, attention settings) on replays the iterations from ONE
HIP graph behind a directly launched prologue + K/V projection.  Contract tested here: the replay is the uncaptured forward bit for bit
for ANY tokens / cameras / output tensors (the graph holds no pointer of a particular call), a graph is never replayed across a change
of anything it was recorded with (weights, attention mode, head tiers, seam fusion), and the never-NaN policy still sees the forward that
runs from a graph."""
import ctypes as C
import warnings

import numpy as np
import pytest
import torch

from parq_amd import _lib, synth
from gpu_util import make_decoder, scene_args

pytestmark = pytest.mark.gpu

KEYS = ("pred_logits", "center_unnormalized", "size_unnormalized", "ortho6d", "sem_cls_prob", "coord_pos")


def _setup(seed=901, V=4, h=60, w=80, Q=64, I=3, mode=None, smooth=True):
    cfg = synth.decoder_cfg(dim=256, queries=Q, heads=4, ffn=256, layers=I)
    W = synth.make_decoder_weights(cfg, seed, damped=True)
    dec = make_decoder(cfg, W).eval()
    if mode:
        dec.attention_mode = mode
    sc = synth.make_scene(seed + 1, 1, V, h, w, 256, smooth=smooth)
    return cfg, W, dec, scene_args(sc), (h, w)


def _run(dec, args, hw):
    with torch.no_grad():
        out = [{k: v.clone() for k, v in o.items()} for o in dec(*args, feat_hw=hw)]
    torch.cuda.synchronize()
    return out


def _same(a, b):
    return all(torch.equal(x[k], y[k]) for x, y in zip(a, b) for k in KEYS)


def _replays(dec):
    return sum(e.replays for e in dec._ws.values())


@pytest.mark.parametrize("mode", ["split8", "split", "fp32", "bf16"])
def test_replayed_forward_is_the_uncaptured_forward_bit_for_bit(mode):
    cfg, W, dec, args, hw = _setup(mode=mode)
    dec.range_check = "off" if mode in ("split8",) else dec.range_check
    dec.use_graph = False
    want = _run(dec, args, hw)
    assert _replays(dec) == 0
    dec.use_graph = True
    first = _run(dec, args, hw)             # launch by launch (remembers the key)
    second = _run(dec, args, hw)            # captured here, and run from the graph
    third = _run(dec, args, hw)             # replayed
    assert _replays(dec) == 2, "the second forward with the same key captures and replays, the third replays"
    entry = next(reversed(dec._ws.values()))
    nodes = _lib.load().parq_graph_nodes(next(iter(entry.graphs.values())))
    assert nodes >= 8 * dec.num_layers, nodes
    for got in (first, second, third):
        assert _same(got, want)
    # fresh output tensors per call (the reference returns new tensors; a caller may keep the previous call's)
    with torch.no_grad():
        a = dec(*args, feat_hw=hw)
        keep = a[0]["pred_logits"].clone()
        b = dec(*args, feat_hw=hw)
    torch.cuda.synchronize()
    assert a[0]["pred_logits"].data_ptr() != b[0]["pred_logits"].data_ptr() and torch.equal(a[0]["pred_logits"], keep)


def test_one_graph_serves_every_call_of_its_shape_whatever_tensors_the_caller_passes():
    """The recorded iterations take this call's tokens, cameras and outputs from the workspace (left there by the directly launched
    prologue): other token tensors, other cameras / poses, fresh output tensors — all replay the same graph and match the uncaptured path."""
    cfg, W, dec, args, hw = _setup(seed=911)
    dec.range_check = "off"
    for _ in range(3):
        _run(dec, args, hw)
    assert _replays(dec) == 2
    sc2 = synth.make_scene(913, 1, 4, hw[0], hw[1], 256, smooth=True)
    args2 = scene_args(sc2)
    moved = (args[0],) + tuple(args2[1:])                  # same token tensor, another scene's cameras and poses
    dec.use_graph = False
    want_moved = _run(dec, moved, hw)
    want_2 = _run(dec, args2, hw)
    want_1 = _run(dec, args, hw)
    dec.use_graph = True
    n = _replays(dec)
    assert _same(_run(dec, moved, hw), want_moved)
    assert _same(_run(dec, args2, hw), want_2)
    assert _same(_run(dec, args, hw), want_1)
    _run(dec, args, hw)                                    # (use_graph was off in between: the key has to repeat once)
    assert _replays(dec) >= n + 3
    entry = next(reversed(dec._ws.values()))
    assert len(entry.graphs) == 1


def test_weight_update_mode_change_tier_move_and_seam_switch_never_replay_an_old_graph():
    cfg, W, dec, args, hw = _setup(seed=921)
    dec.range_check = "off"
    ref = make_decoder(cfg, W).eval()
    ref.range_check = "off"
    ref.use_graph = False

    def both(change):
        change(dec)
        change(ref)
        want = _run(ref, args, hw)
        for rep in range(3):                               # direct, capture, replay: all three are the new state's forward
            assert _same(_run(dec, args, hw), want), rep
    both(lambda d: None)
    with torch.no_grad():
        both(lambda d: d.mlp_heads["size_head"].layers["0"].bias.add_(0.25))         # an optimizer-style in-place update (version counter)
    both(lambda d: setattr(d, "attention_mode", "split"))
    both(lambda d: setattr(d, "attention_mode", "split8"))
    both(lambda d: setattr(d, "safe_heads", 0b0101))
    both(lambda d: setattr(d, "fuse_seams", False))
    both(lambda d: setattr(d, "fuse_seams", True))
    assert _replays(dec) >= 7
    assert len(dec._graveyard) <= 2, "graphs of dropped workspaces are destroyed once their last launch has completed"


def test_default_policy_sees_a_forward_that_runs_from_a_graph():
    """range_check = "sync" + a replayed forward: features that leave the fp16 range in the THIRD call (the replayed one) are re-run
    with the fp32 kernels inside that call — finite outputs, the module switched, a warning."""
    cfg, W, dec, args, hw = _setup(seed=931, mode="split")
    assert dec.range_check == "sync"
    tokens = args[0].clone()
    a = (tokens,) + tuple(args[1:])
    for _ in range(2):
        _run(dec, a, hw)
    assert _replays(dec) == 1
    entry = next(reversed(dec._ws.values()))               # (the fallback drops the workspace: keep a reference to count its replays)
    tokens.mul_(3e4)                                       # same buffer (same key: the graph replays), values beyond 60000
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        got = _run(dec, a, hw)
    assert entry.replays == 2 and dec.attention_mode == "fp32" and any("fp16 range" in str(r.message) for r in rec)
    assert all(torch.isfinite(o[k]).all() for o in got for k in KEYS)
    ref = make_decoder(cfg, W).eval()
    ref.attention_mode = "fp32"
    assert _same(got, _run(ref, a, hw))


def test_capture_entry_points_of_the_c_abi_directly():
    """parq_forward_capture + parq_forward_replay through ctypes, as INTEGRATION.md shows them: the replay writes what parq_forward
    writes, on any stream, into whatever outputs the call names; capture does not run anything; a graph is refused after the handle's
    settings changed."""
    cfg, W, dec, args, hw = _setup(seed=941, mode="split")
    dec.range_check = "off"
    dec.use_graph = False
    want = _run(dec, args, hw)
    lib, h = _lib.load(), dec._handle()
    sc, keep, dev_ = dec._scene(*args, feat_hw=hw)
    ws = dec._workspace(sc.B, sc.V, sc.h, sc.w, dev_)
    g = C.c_void_p()
    before = ws.clone()
    _lib.check(lib.parq_forward_capture(h, sc.B, sc.V, sc.h, sc.w, _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr(), C.byref(g)), "capture")
    torch.cuda.synchronize()
    assert torch.equal(ws.view(torch.int32), before.view(torch.int32)), "capturing must not execute anything"
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    for rep in range(2):
        outs = dec._alloc_outputs((dec.num_layers, sc.B, dec.num_queries), dev_)        # fresh outputs per replay
        po = _lib.ParqOutputs(*[_lib.ptr(t) for t in outs])
        _lib.check(lib.parq_forward_replay(h, g, C.byref(sc), _lib.ptr(ws), ws.numel() * 4, C.byref(po), C.c_void_p(side.cuda_stream)), "replay")
        side.synchronize()
        for i, key in enumerate(KEYS):
            for k in range(dec.num_layers):
                assert torch.equal(outs[i][k], want[k][key]), (rep, key, k)
    # settings changed since the capture: refused with a status code, nothing enqueued
    flipped = 0 if dec.fuse_seams else 1
    _lib.check(lib.parq_set_seam_fusion(h, flipped), "seam")
    assert lib.parq_forward_replay(h, g, C.byref(sc), _lib.ptr(ws), ws.numel() * 4, C.byref(po), _lib.stream_ptr()) == 3
    assert b"capture again" in lib.parq_last_error()
    _lib.check(lib.parq_set_seam_fusion(h, 1 - flipped), "seam")
    assert lib.parq_graph_destroy(g) == 0
    assert lib.parq_forward_replay(h, None, C.byref(sc), _lib.ptr(ws), ws.numel() * 4, C.byref(po), _lib.stream_ptr()) != 0
    dec.profile_enable(True)
    g2 = C.c_void_p()
    assert lib.parq_forward_capture(h, sc.B, sc.V, sc.h, sc.w, _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr(), C.byref(g2)) != 0
    assert b"profile" in lib.parq_last_error()
    dec.profile_enable(False)


def test_early_completion_signal_carries_the_epoch_and_the_flags():
    """parq_set_progress (include/parq_hip.h): the first launch behind the last iteration's cross-attention merge stores the call's epoch
    into the host-visible progress word — captured forward and launch-by-launch forward alike, a new epoch per call — and ORs what the
    forward raised into the mirror word BEFORE it; the default policy waits for that word instead of the stream."""
    cfg, W, dec, args, hw = _setup(seed=951, mode="split")
    assert dec.range_check == "sync"
    seen = []
    for rep in range(4):                                    # launch by launch, captured, replayed, replayed
        _run(dec, args, hw)
        entry = next(reversed(dec._ws.values()))
        seen.append((entry.epoch, int(dec._mirror_np[dec._MIRROR_SLOTS + entry.slot])))
    assert [e for e, _ in seen] == sorted({e for e, _ in seen}) and all(e == w and e > 0 for e, w in seen), seen
    assert _replays(dec) == 3
    # a flagged forward: the bits are in the mirror word when the epoch is (the re-run decision is taken from them)
    tokens = args[0].clone()
    tokens.mul_(3e4)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        got = _run(dec, (tokens,) + tuple(args[1:]), hw)
    assert dec.attention_mode == "fp32" and any("fp16 range" in str(r.message) for r in rec)
    assert all(torch.isfinite(o[k]).all() for o in got for k in KEYS)
