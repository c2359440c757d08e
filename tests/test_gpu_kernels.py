"""GPU tier, per kernel: each HIP kernel is called through the C ABI and compared with the
CPU oracle's statement of the same operation on seeded inputs.  Tolerance 1e-4 on
|a-b|/max(1,|b|) (fp32; north-star bar), tighter where the op is a plain fp32 chain."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from parq_amd import _lib, synth
from oracle import parq_oracle as O
from gpu_util import dev, lib, sptr, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K,relu,use_x2,use_r", [
    (256, 256, 256, 0, False, False), (256, 768, 256, 1, False, False), (256, 256, 768, 0, False, True),
    (64, 256, 384, 1, True, False), (37, 19, 64, 0, True, True), (2048, 10, 256, 0, False, False),
    (1, 3, 32, 0, False, False), (300, 525, 1024, 0, False, False)])
def test_linear(M, N, K, relu, use_x2, use_r):
    X = synth.normal(1, "X", (M, K)); X2 = synth.normal(2, "X2", (M, K))
    W = synth.normal(3, "W", (N, K), std=K ** -0.5); b = synth.normal(4, "b", (N,)); R = synth.normal(5, "R", (M, N))
    Y = torch.empty(M, N, device="cuda")
    dX, dX2, dW, db, dR = dev(X), dev(X2), dev(W), dev(b), dev(R)       # keep the device tensors alive
    rc = lib().parq_k_linear(_lib.ptr(dX), _lib.ptr(dX2) if use_x2 else None, _lib.ptr(dW), _lib.ptr(db),
                             _lib.ptr(dR) if use_r else None, _lib.ptr(Y), M, N, K, relu, sptr())
    _lib.check(rc, "parq_k_linear")
    A = torch.from_numpy(X).double() + (torch.from_numpy(X2).double() if use_x2 else 0)
    want = A @ torch.from_numpy(W).double().T + torch.from_numpy(b).double()
    if relu:
        want = want.clamp(min=0)
    if use_r:
        want = want + torch.from_numpy(R).double()
    assert rel_err(Y.cpu().numpy(), want.numpy()) < 2e-5


@pytest.mark.parametrize("M,N,K,use_x2,use_r,spread", [
    (256, 3072, 1024, True, False, 0), (256, 1024, 1024, False, True, 0), (64, 2064, 1024, False, False, 0),
    (256, 1024, 768, False, False, 0), (32, 1024, 1024, False, False, 40), (48, 256, 1024, True, False, -40),
    (512, 1024, 1024, False, True, 0),       # 32-row tiles (their grid has a workgroup per CU): LDS planes shared with the fold
    (1024, 2064, 1024, False, False, 0),     # 43 column tiles of 48: the grid is padded to 48 per row tile, 5 workgroups per row tile leave at once
    (1024, 3072, 1024, True, False, 25),     # the addend launch on 32-row tiles, rows x 2^+-25
    (4096, 2064, 1024, False, False, 0)])    # 129 sub-tiles at four rounds and more: full-width column tiles, the last one with one sub-tile
def test_linear_fp16x3_tile_is_fp32_class_at_any_magnitude(M, N, K, use_x2, use_r, spread):
    """parq_k_linear_half (chain.hip chain_linear_h3_kernel: the inference chain's tile at the shipped width): fp16 hi / lo operands,
    three fp16 MFMA products, fp32 accumulation — against a float64 product at the tolerance of the fp32 tile.  `spread`: rows of X and
    rows of W scaled by 2^(+-spread) alternately (values far outside the fp16 range: the per-row / per-column power-of-two scales make
    range a non-condition), plus an all-zero row and an all-zero weight row."""
    X = synth.normal(1, "X", (M, K)); X2 = synth.normal(2, "X2", (M, K))
    W = synth.normal(3, "W", (N, K), std=K ** -0.5); b = synth.normal(4, "b", (N,)); R = synth.normal(5, "R", (M, N))
    if spread:
        X = X * (2.0 ** (spread * (np.arange(M) % 3 - 1)))[:, None].astype(np.float32)
        W = W * (2.0 ** (-spread * (np.arange(N) % 3 - 1) * 0.5))[:, None].astype(np.float32)
        X[5] = 0.0
        W[7] = 0.0
        b = b * 0
    Y = torch.full((M, N), float("nan"), device="cuda")
    scratch = torch.empty(N * K + N, device="cuda")
    dX, dX2, dW, db, dR = dev(X), dev(X2), dev(W), dev(b), dev(R)
    rc = lib().parq_k_linear_half(_lib.ptr(dX), _lib.ptr(dX2) if use_x2 else None, _lib.ptr(dW), _lib.ptr(db),
                                  _lib.ptr(dR) if use_r else None, _lib.ptr(Y), M, N, K, 0, _lib.ptr(scratch), scratch.numel() * 4, sptr())
    _lib.check(rc, "parq_k_linear_half")
    A = torch.from_numpy(X).double() + (torch.from_numpy(X2).double() if use_x2 else 0)
    want = A @ torch.from_numpy(W).double().T + torch.from_numpy(b).double()
    if use_r:
        want = want + torch.from_numpy(R).double()
    got = Y.cpu().double()
    assert torch.isfinite(got).all()
    if spread:          # per-element scale: |row of X| |row of W| sqrt(K) is the size of an output
        scale = (A.norm(dim=1)[:, None] * torch.from_numpy(W).double().norm(dim=1)[None, :]).clamp(min=1e-300)
        assert ((got - want).abs() / scale).max().item() < 2e-6
        assert (got[:, 7] == 0).all() and (use_x2 or (got[5] == 0).all())
    else:
        assert rel_err(got.numpy(), want.numpy()) < 2e-5
        fp32 = torch.empty(M, N, device="cuda")
        _lib.check(lib().parq_k_linear(_lib.ptr(dX), _lib.ptr(dX2) if use_x2 else None, _lib.ptr(dW), _lib.ptr(db),
                                       _lib.ptr(dR) if use_r else None, _lib.ptr(fp32), M, N, K, 0, sptr()), "parq_k_linear")
        e16, e32 = (got - want).abs().max().item(), (fp32.cpu().double() - want).abs().max().item()
        assert e16 < 4 * e32 + 1e-6, (e16, e32)       # the same error class as the fp32 MFMA tile


def test_linear_fp16x3_rejects_what_it_has_no_tile_for():
    X = torch.zeros(16, 512, device="cuda")
    s = torch.empty(16 * 512 + 16, device="cuda")
    assert lib().parq_k_linear_half(_lib.ptr(X), None, _lib.ptr(X), None, None, _lib.ptr(X), 16, 16, 512, 0, _lib.ptr(s), s.numel() * 4, sptr()) == 1
    X = torch.zeros(16, 1024, device="cuda")
    assert lib().parq_k_linear_half(_lib.ptr(X), None, _lib.ptr(X), None, None, _lib.ptr(X), 16, 16, 1024, 0, _lib.ptr(s), 64, sptr()) == 1


def test_linear_rejects_bad_k():
    X = torch.zeros(4, 48, device="cuda")
    assert lib().parq_k_linear(_lib.ptr(X), None, _lib.ptr(X), None, None, _lib.ptr(X), 4, 4, 48, 0, sptr()) == 1
    assert b"multiple of 32" in lib().parq_last_error()


@pytest.mark.parametrize("M,Cn", [(256, 256), (7, 128), (33, 1024), (5, 64)])
def test_layernorm(M, Cn):
    X = synth.normal(1, "lnx", (M, Cn), std=2.0, mean=0.5)
    g = synth.uniform(2, "lng", (Cn,), 0.5, 1.5); b = synth.normal(3, "lnb", (Cn,))
    Y = torch.empty(M, Cn, device="cuda")
    dX, dg, db = dev(X), dev(g), dev(b)
    _lib.check(lib().parq_k_layernorm(_lib.ptr(dX), _lib.ptr(dg), _lib.ptr(db), _lib.ptr(Y), M, Cn, 1e-5, sptr()), "ln")
    want = F.layer_norm(torch.from_numpy(X).double(), (Cn,), torch.from_numpy(g).double(), torch.from_numpy(b).double(), 1e-5)
    assert rel_err(Y.cpu().numpy(), want.numpy()) < 1e-5


@pytest.mark.parametrize("B,H,Lq,Lk,dh", [
    (1, 4, 64, 9600, 64),       # cfg-1 cross-attention
    (2, 4, 256, 256, 64),       # self-attention shape
    (1, 2, 40, 429, 64),        # ragged Lq and Lk (tail masking)
    (1, 1, 32, 64, 64), (1, 1, 1, 1, 64), (2, 2, 100, 1000, 32), (1, 2, 48, 300, 128), (1, 4, 32, 600, 256)])
def test_attention(B, H, Lq, Lk, dh):
    Cn = H * dh
    q = synth.normal(1, "q", (B, Lq, Cn)); k = synth.normal(2, "k", (B, Lk, Cn)); v = synth.normal(3, "v", (B, Lk, Cn))
    nbytes = lib().parq_k_attention_scratch_bytes(B, H, Lq, Lk, dh)
    scratch = torch.empty(nbytes // 4 + 1, device="cuda")
    out = torch.empty(B, Lq, Cn, device="cuda")
    dq, dk, dv = dev(q), dev(k), dev(v)
    _lib.check(lib().parq_k_attention(_lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(out), B, H, Lq, Lk, dh,
                                     _lib.ptr(scratch), nbytes, sptr()), "attention")
    tq, tk, tv = (torch.from_numpy(x).double().view(B, -1, H, dh).transpose(1, 2) for x in (q, k, v))
    want = (torch.softmax(tq @ tk.transpose(-1, -2) / dh ** 0.5, -1) @ tv).transpose(1, 2).reshape(B, Lq, Cn)
    assert rel_err(out.cpu().numpy(), want.numpy()) < 2e-5


def test_attention_spike_forces_rescale():
    """A key that dominates late in the stream forces the online-softmax rescale branch."""
    B, H, Lq, Lk, dh = 1, 1, 32, 4096, 64
    q = synth.normal(1, "q", (B, Lq, dh)); k = synth.normal(2, "k", (B, Lk, dh)); v = synth.normal(3, "v", (B, Lk, dh))
    k[0, 3000] = 8.0 * q[0, 5]          # score ~ 8*|q|^2/8 >> everything before it
    k[0, 10] = 4.0 * q[0, 7]
    nbytes = lib().parq_k_attention_scratch_bytes(B, H, Lq, Lk, dh)
    scratch = torch.empty(nbytes // 4 + 1, device="cuda"); out = torch.empty(B, Lq, dh, device="cuda")
    dq, dk, dv = dev(q), dev(k), dev(v)
    _lib.check(lib().parq_k_attention(_lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(out), B, H, Lq, Lk, dh,
                                     _lib.ptr(scratch), nbytes, sptr()), "attention")
    tq, tk, tv = (torch.from_numpy(x).double() for x in (q, k, v))
    want = torch.softmax(tq @ tk.transpose(-1, -2) / dh ** 0.5, -1) @ tv
    assert rel_err(out.cpu().numpy(), want.numpy()) < 2e-5


@pytest.mark.parametrize("B,H,Lq,Lk", [(1, 4, 64, 9600), (2, 4, 256, 256), (1, 2, 40, 429), (1, 1, 32, 33), (1, 1, 1, 1),
                                       (1, 4, 256, 19200), (2, 1, 300, 1000)])
def test_attention_split_fp16x3(B, H, Lq, Lk):
    """Split-precision kernel (fp16 hi/lo, 3-term products): same tolerance as the fp32-MFMA kernel."""
    dh, Cn = 64, H * 64
    q = synth.normal(1, "q", (B, Lq, Cn)); k = synth.normal(2, "k", (B, Lk, Cn), std=2.0); v = synth.normal(3, "v", (B, Lk, Cn), std=3.0)
    k[0, 0, :dh] = 3.0 * q[0, 0, :dh]                 # a peaky row
    # exact fp16 rounding ties (and min-normal residuals): the hi/lo split must stay self-consistent
    ties = np.array([1 + 2.0 ** -11, -(2 + 2.0 ** -10), 0.5 + 2.0 ** -12, 3 * 2.0 ** -14 + 2.0 ** -25, 0.20623779296875], np.float32)
    k[0, Lk - 1, :5] = ties
    v[0, 0, :5] = ties
    q[0, Lq - 1, :5] = ties / np.float32(1.4426950408889634 / 8.0)
    nbytes = lib().parq_k_attention_split_scratch_bytes(B, H, Lq, Lk)
    scratch = torch.empty(nbytes // 4 + 1, device="cuda")
    out = torch.empty(B, Lq, Cn, device="cuda")
    dq, dk, dv = dev(q), dev(k), dev(v)
    _lib.check(lib().parq_k_attention_split(_lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(out), B, H, Lq, Lk,
                                           _lib.ptr(scratch), nbytes, sptr()), "attention_split")
    tq, tk, tv = (torch.from_numpy(x).double().view(B, -1, H, dh).transpose(1, 2) for x in (q, k, v))
    want = (torch.softmax(tq @ tk.transpose(-1, -2) / dh ** 0.5, -1) @ tv).transpose(1, 2).reshape(B, Lq, Cn)
    assert rel_err(out.cpu().numpy(), want.numpy()) < 2e-5
    assert int(scratch[:1].view(torch.int32).item()) == 0          # no fp16 range overflow flagged


@pytest.mark.parametrize("B,H,Lq,Lk", [(1, 4, 256, 9600), (2, 2, 100, 1000), (1, 1, 32, 33), (1, 2, 130, 4097), (2, 1, 256, 70000)])
def test_attention_split_head_dim_256(B, H, Lq, Lk):
    """Split-precision attention at head dim 256 (the reference's shipped DEC_DIM 1024 / 4 heads): wave pairs share a query tile and
    split the head dim.  fp32-class tolerance against float64; ragged Lq (inactive wave pairs, partial tiles) and Lk (masked last
    block), one to many key splits, rows with exact fp16 ties."""
    dh, Cn = 256, H * 256
    q = synth.normal(11, "q", (B, Lq, Cn)); k = synth.normal(12, "k", (B, Lk, Cn), std=1.5); v = synth.normal(13, "v", (B, Lk, Cn), std=3.0)
    k[0, 0, :dh] = 1.5 * q[0, 0, :dh]                 # a peaky row
    ties = np.array([1 + 2.0 ** -11, -(2 + 2.0 ** -10), 0.5 + 2.0 ** -12, 3 * 2.0 ** -14 + 2.0 ** -25, 0.20623779296875], np.float32)
    k[0, Lk - 1, 130:135] = ties
    v[0, 0, 200:205] = ties
    nbytes = lib().parq_k_attention_split256_scratch_bytes(B, H, Lq, Lk)
    scratch = torch.empty(nbytes // 4 + 1, device="cuda")
    out = torch.empty(B, Lq, Cn, device="cuda")
    dq, dk, dv = dev(q), dev(k), dev(v)
    _lib.check(lib().parq_k_attention_split256(_lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(out), B, H, Lq, Lk,
                                              _lib.ptr(scratch), nbytes, sptr()), "attention_split256")
    tq, tk, tv = (torch.from_numpy(x).double().view(B, -1, H, dh).transpose(1, 2) for x in (q, k, v))
    want = (torch.softmax(tq @ tk.transpose(-1, -2) / dh ** 0.5, -1) @ tv).transpose(1, 2).reshape(B, Lq, Cn)
    assert rel_err(out.cpu().numpy(), want.numpy()) < 2e-5
    assert int(scratch[:1].view(torch.int32).item()) == 0          # no fp16 range overflow flagged


@pytest.mark.parametrize("bf16,tol", [(0, 2e-3), (1, 1.5e-2)])
@pytest.mark.parametrize("B,H,Lq,Lk", [(1, 4, 64, 9600), (2, 2, 100, 1000), (1, 1, 32, 64), (1, 2, 256, 4097),
                                       (2, 2, 100, 1024), (1, 4, 256, 12800)])      # whole 64-key stages: the stage kernel (ragged Lq; write-through epilogue)
def test_attention_half(B, H, Lq, Lk, bf16, tol):
    """Single-product fp16 / bf16 attention (modes 2 / 3): tolerance = a few 16-bit ulps of the value scale
    (fp16 2^-11, bf16 2^-8 relative operand rounding), stated here; ragged Lq / Lk covered."""
    dh, Cn = 64, H * 64
    q = synth.normal(1, "q", (B, Lq, Cn)); k = synth.normal(2, "k", (B, Lk, Cn)); v = synth.normal(3, "v", (B, Lk, Cn), std=2.0)
    k[0, 0, :dh] = 2.0 * q[0, 0, :dh]                 # a peaky row
    nbytes = lib().parq_k_attention_half_scratch_bytes(B, H, Lq, Lk)
    scratch = torch.empty(nbytes // 4 + 1, device="cuda")
    out = torch.empty(B, Lq, Cn, device="cuda")
    dq, dk, dv = dev(q), dev(k), dev(v)
    _lib.check(lib().parq_k_attention_half(_lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(out), B, H, Lq, Lk, bf16,
                                          _lib.ptr(scratch), nbytes, sptr()), "attention_half")
    tq, tk, tv = (torch.from_numpy(x).double().view(B, -1, H, dh).transpose(1, 2) for x in (q, k, v))
    want = (torch.softmax(tq @ tk.transpose(-1, -2) / dh ** 0.5, -1) @ tv).transpose(1, 2).reshape(B, Lq, Cn)
    err = rel_err(out.cpu().numpy(), want.numpy())
    assert err < tol, err
    assert err > 1e-6            # it really is the reduced-precision path


def test_attention_split_spike_forces_deferred_rescale():
    """The split kernel defers the running-max update (threshold 2^10): keys that dominate late in
    the stream must take the rescale branch; a moderately larger key (< threshold) must not need it."""
    B, H, Lq, Lk, dh = 1, 1, 64, 8192, 64
    q = synth.normal(1, "q", (B, Lq, dh)); k = synth.normal(2, "k", (B, Lk, dh)); v = synth.normal(3, "v", (B, Lk, dh))
    k[0, 5000] = 8.0 * q[0, 5]          # score ~ |q|^2: far past the deferral threshold, late in the stream
    k[0, 7000] = 1.0 * q[0, 9]          # ~ 8 in the log2 domain: inside the threshold
    k[0, 100] = 4.0 * q[0, 40]
    nbytes = lib().parq_k_attention_split_scratch_bytes(B, H, Lq, Lk)
    scratch = torch.empty(nbytes // 4 + 1, device="cuda"); out = torch.empty(B, Lq, dh, device="cuda")
    dq, dk, dv = dev(q), dev(k), dev(v)
    _lib.check(lib().parq_k_attention_split(_lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(out), B, H, Lq, Lk,
                                           _lib.ptr(scratch), nbytes, sptr()), "attention_split")
    tq, tk, tv = (torch.from_numpy(x).double() for x in (q, k, v))
    want = torch.softmax(tq @ tk.transpose(-1, -2) / dh ** 0.5, -1) @ tv
    assert rel_err(out.cpu().numpy(), want.numpy()) < 2e-5


def test_attention_split_flags_fp16_overflow():
    B, H, Lq, Lk = 1, 1, 32, 64
    q = synth.normal(1, "q", (B, Lq, 64)); k = synth.normal(2, "k", (B, Lk, 64)); v = synth.normal(3, "v", (B, Lk, 64))
    v[0, 5, 7] = 7.0e4
    nbytes = lib().parq_k_attention_split_scratch_bytes(B, H, Lq, Lk)
    scratch = torch.empty(nbytes // 4 + 1, device="cuda"); out = torch.empty(B, Lq, 64, device="cuda")
    dq, dk, dv = dev(q), dev(k), dev(v)
    _lib.check(lib().parq_k_attention_split(_lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(out), B, H, Lq, Lk,
                                           _lib.ptr(scratch), nbytes, sptr()), "attention_split")
    assert int(scratch[:1].view(torch.int32).item()) == 1


def test_camera_local():
    cam, T_cp, T_wp, T_wl = synth.make_geometry(4, 3, 5, 12, 16)
    out = torch.empty(3, 5, 12, device="cuda")
    d1, d2, d3 = dev(T_cp), dev(T_wp), dev(T_wl)
    _lib.check(lib().parq_k_camera_local(_lib.ptr(d1), _lib.ptr(d2), _lib.ptr(d3), 3, 5, _lib.ptr(out), sptr()), "cl")
    want = O.camera_local_poses(torch.from_numpy(T_cp).double(), torch.from_numpy(T_wp).double(), torch.from_numpy(T_wl).double())
    assert rel_err(out.cpu().numpy(), want.numpy()) < 1e-6


@pytest.mark.parametrize("B,V,h,w,Cn,Q", [(1, 2, 60, 80, 256, 64), (2, 3, 11, 13, 128, 40), (1, 20, 12, 16, 64, 33),
                                          (1, 2, 15, 20, 1024, 32), (2, 10, 30, 40, 256, 256)])
def test_project_sample(B, V, h, w, Cn, Q):
    sc = synth.make_scene(21, B, V, h, w, Cn)
    scale = synth.DEFAULT_SCALE
    ref = synth.uniform(22, "ref", (B, Q, 3), 0.0, 1.0)
    T_cl = O.camera_local_poses(*(torch.from_numpy(sc[k]) for k in ("T_camera_pseudoCam", "T_world_pseudoCam", "T_world_local")))
    tgt = torch.empty(B, Q, Cn, device="cuda"); cp = torch.empty(B, Q, 3, device="cuda")
    dtok, dT, dcam, dref = dev(sc["tokens"]), T_cl.cuda().contiguous(), dev(sc["camera"]), dev(ref)
    _lib.check(lib().parq_k_project_sample(_lib.ptr(dtok), _lib.ptr(dT), _lib.ptr(dcam),
                                          _lib.ptr(dref), (C.c_float * 6)(*scale), B, V, h, w, Cn, Q, _lib.ptr(tgt), _lib.ptr(cp),
                                          sptr()), "project_sample")
    P = O.denormalize(torch.from_numpy(ref), scale)
    tok64, cam64 = torch.from_numpy(sc["tokens"]).double(), torch.from_numpy(sc["camera"]).double()
    T64 = O.camera_local_poses(*(torch.from_numpy(sc[k]).double() for k in ("T_camera_pseudoCam", "T_world_pseudoCam", "T_world_local")))
    P64 = O.denormalize(torch.from_numpy(ref).double(), scale)
    want64, p2d, valid = O.project_and_sample(tok64, P64, T64, cam64, h, w, False)
    # queries whose valid-view count could flip under rounding are excluded
    size = cam64[..., :2].unsqueeze(-2)
    margin = torch.minimum(p2d.abs(), (p2d - (size - 1)).abs()).min(-1).values.min(1).values
    ok = (margin > 1e-3).numpy()
    assert ok.mean() > 0.9
    # (the kernel is handed float32 poses here, so the float64 oracle gets the same rounded poses)
    want64b, _, _ = O.project_and_sample(tok64, P64, T_cl.double(), cam64, h, w, False)
    assert rel_err(tgt.cpu().numpy()[ok], want64b.numpy()[ok]) < 2e-6      # geometry in fp64, fp32 blend
    for ro in (False, True):                                               # the fp32 reference ops carry ~5e-5 of
        want, _, _ = O.project_and_sample(torch.from_numpy(sc["tokens"]), P, T_cl, torch.from_numpy(sc["camera"]), h, w, ro)
        assert rel_err(tgt.cpu().numpy()[ok], want.numpy()[ok]) < 1e-4     # their own pixel-coordinate rounding noise
    assert rel_err(cp.cpu().numpy(), P.numpy()) < 1e-6
    assert valid.any() and (~valid).any()      # the fixture exercises both branches


@pytest.mark.parametrize("for_vis,key", [(0, "mask_eval"), (1, "mask_vis")])
def test_parse_pred_matches_reference_golden(for_vis, key):
    """Device parse_pred + 3-D NMS against the golden computed with the reference's own building blocks (ortho6d -> R,
    Obb3D.separate_init, utils/nms.nms) on the CPU: boxes within 2e-6, keep masks identical."""
    import json
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
    from make_golden import PARSE_CASE, parse_case_inputs
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "g11_parse_pred.npz"))
    assert json.loads(bytes(z["meta"]).decode()) == json.loads(json.dumps(PARSE_CASE))
    x = parse_case_inputs(PARSE_CASE)
    B, Q = PARSE_CASE["B"], PARSE_CASE["Q"]
    d = {k: dev(v) for k, v in x.items()}
    obbs = torch.empty(B, Q, 19, device="cuda"); mask = torch.empty(B, Q, dtype=torch.uint8, device="cuda")
    ts = (C.c_float * 6)(*[float(t) for t in PARSE_CASE["track_scale"]])
    _lib.check(lib().parq_parse_pred(_lib.ptr(d["center"]), _lib.ptr(d["size"]), _lib.ptr(d["rot6"]), _lib.ptr(d["prob"]), B, Q, 10, ts,
                                    for_vis, 1, _lib.ptr(obbs), C.c_void_p(mask.data_ptr()), sptr()), "parse_pred")
    assert np.abs(obbs.cpu().numpy() - z["obbs"]).max() < 2e-6
    assert np.array_equal(mask.cpu().numpy().astype(bool), z[key])
    assert 0 < int(mask.sum()) < B * Q


def test_dropout_masks_of_row_pairs_overlap_like_independent_draws():
    """nn.Dropout draws i.i.d. masks; the library's counter-based masks must at least look pairwise AND jointly independent
    across rows: for every pair of rows the fraction of one row's dropped columns that the other row drops too has to be ~p
    (comparing the bare xor of a row hash and a column hash against the threshold gave 1/16 of all row pairs
    >60 % shared drops at p = 0.1).  512 rows x 4096 columns, all 130 816 pairs, 6-sigma band."""
    import ctypes as C
    from parq_amd import _lib
    from parq_amd.decoder import PARQDecoder
    p, rows, cols = 0.1, 512, 4096
    cfg = synth.decoder_cfg(dim=64, queries=16, heads=1, ffn=64, layers=2, dropout=p)
    dec = PARQDecoder(cfg).cuda()
    h = dec._handle(apply_mode=False)
    l = _lib.load()
    _lib.check(l.parq_set_dropout(h, p, 12345), "set_dropout")
    for site in (2, 4):
        m = torch.empty(rows, cols, device="cuda")
        _lib.check(l.parq_k_dropout_mask(h, 0, site, rows, cols, _lib.ptr(m), _lib.stream_ptr()), "mask")
        d = (m == 0).double()                                     # dropped indicator
        frac = float(d.mean())
        assert abs(frac - p) < 0.004, frac
        per_row = d.sum(1)
        assert float(per_row.min()) > 0.75 * p * cols and float(per_row.max()) < 1.25 * p * cols
        shared = d @ d.t()                                        # (rows, rows): columns dropped by both
        share = (shared / per_row[:, None]).fill_diagonal_(0.0)   # fraction of row i's drops that row j shares
        sigma = (p * (1 - p) / float(per_row.min())) ** 0.5
        worst = float(share.max())
        print("\ndropout site %d: drop rate %.4f, worst shared-drop fraction over all row pairs %.3f (independent: %.3f +- %.3f)"
              % (site, frac, worst, p, sigma))
        assert worst < p + 6 * sigma, worst
        # columns too: no column is dropped for (or spared by) whole groups of rows
        per_col = d.sum(0)
        assert float(per_col.max()) < p * rows + 7 * (p * (1 - p) * rows) ** 0.5
