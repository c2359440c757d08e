"""Deterministic, torch-independent synthetic inputs for the PARQ decoder path.

Everything is produced by a counter-based hash (splitmix64 over
``(seed, stream name, element index)``), so the same ``(seed, name, shape)``
gives bit-identical float32 arrays on every machine and numpy version.  The
golden fixtures under ``tests/golden`` store only *outputs*; their inputs are
regenerated from the seeds recorded next to them.

Conventions mirrored from the reference (file:line are relative to the
reference tree):

* Pose = 12-vector ``[R row-major (9) | t (3)]``            utils/wrappers.py:194-293
* Camera = 6-vector ``[w, h, fx, fy, cx, cy]``              utils/wrappers.py:441-476
* tokens = channels-last ``(B, V*h*w, C)``, token index
  ``(v*h + y)*w + x``                                       model/parq_lightning.py:78-85
* weight names = the decoder's ``state_dict`` keys           model/parq_decoder.py:35-82
"""
from __future__ import annotations

import zlib
from types import SimpleNamespace

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _stream_key(seed: int, name: str) -> np.uint64:
    h = zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF
    k = np.array([(int(seed) & 0xFFFFFFFF) << 32 | h], dtype=np.uint64)
    return _splitmix64(k)[0]


def _u01(seed: int, name: str, n: int, lane: int = 0) -> np.ndarray:
    """n float64 uniforms in (0, 1) from stream (seed, name, lane)."""
    with np.errstate(over="ignore"):
        key = _stream_key(seed, name + "#%d" % lane)
        idx = np.arange(n, dtype=np.uint64)
        bits = _splitmix64(idx ^ key)
        bits = _splitmix64(bits + key)
    # 53 random mantissa bits, shifted off zero
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def uniform(seed: int, name: str, shape, lo=0.0, hi=1.0) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = _u01(seed, name, n)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(seed: int, name: str, shape, std=1.0, mean=0.0) -> np.ndarray:
    """Box-Muller on two hashed uniform streams."""
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = _u01(seed, name, n, 1)
    u2 = _u01(seed, name, n, 2)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return (mean + std * z).astype(np.float32).reshape(shape)


# --------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------

DEFAULT_SCALE = [-3.0, 3.0, -2.0, 0.5, 0.25, 5.25]  # config/train.yaml:55


def decoder_cfg(dim=256, queries=256, heads=4, ffn=768, layers=8, num_semcls=9,
                scale=None, share_weights=True, dropout=0.1, mean_size_path=None):
    """A cfg namespace carrying exactly the fields PARQDecoder reads
    (model/parq_decoder.py:40-82; config/default.py:94-115)."""
    return SimpleNamespace(
        DIM_IN=dim, NUM_QUERIES=queries, NUM_SEMCLS=num_semcls,
        LOSS_WEIGHT=[5.0, 5.0, 5.0, 1.0], FOR_VIS=False,
        TRACK_SCALE=[-1.5, 1.5, -2, 1, 0, 2], SHARE_MLP_HEADS=True,
        MEAN_SIZE_PATH=mean_size_path, EVAL_TYPE="f1", CONF_THRESH=0.1,
        ENABLE_NMS=True,
        TRANSFORMER=SimpleNamespace(
            DEC_DIM=dim, QUERIES_DIM=dim, DEC_HEADS=heads, DEC_LAYERS=layers,
            DEC_FFN_DIM=ffn, DROPOUT_RATE=dropout,
            SCALE=list(scale if scale is not None else DEFAULT_SCALE),
            SHARE_WEIGHTS=share_weights))


# --------------------------------------------------------------------------
# weights (state-dict keys of the reference decoder)
# --------------------------------------------------------------------------

def decoder_param_shapes(cfg) -> dict:
    """name -> shape for every *unique* tensor in PARQDecoder.state_dict().

    The reference registers the four heads twice (``mlp_heads.*`` and
    ``parq_module.decoder.mlp_heads.*`` share storage, model/parq_decoder.py:66);
    only the ``mlp_heads.*`` spelling is listed here.
    """
    C = cfg.DIM_IN
    T = cfg.TRANSFORMER
    F = T.DEC_FFN_DIM
    ncls = cfg.NUM_SEMCLS + 1
    shapes = {"refpoint.weight": (cfg.NUM_QUERIES, 3)}
    # heads: Conv1d(k=1) weights are (out, in, 1)       model/generic_mlp.py:94-110
    shapes["mlp_heads.sem_cls_head.layers.0.weight"] = (ncls, C, 1)
    shapes["mlp_heads.sem_cls_head.layers.0.bias"] = (ncls,)
    shapes["mlp_heads.size_head.layers.0.weight"] = (3, C, 1)
    shapes["mlp_heads.size_head.layers.0.bias"] = (3,)
    for head, nout in (("center_head", 3), ("rotation_head", 6)):
        p = "mlp_heads.%s.layers." % head
        # [Conv(no bias), GN, ReLU, Dropout] x2 + Conv(bias): indices 0,1 | 4,5 | 8
        shapes[p + "0.weight"] = (C, C, 1)
        shapes[p + "1.weight"] = (C,)
        shapes[p + "1.bias"] = (C,)
        shapes[p + "4.weight"] = (C, C, 1)
        shapes[p + "5.weight"] = (C,)
        shapes[p + "5.bias"] = (C,)
        shapes[p + "8.weight"] = (nout, C, 1)
        shapes[p + "8.bias"] = (nout,)
    d = "parq_module.decoder."
    nl = 1 if T.SHARE_WEIGHTS else T.DEC_LAYERS
    for li in range(nl):
        lp = d + "layers.%d." % li
        for attn in ("self_attn", "multihead_attn"):
            shapes[lp + attn + ".in_proj_weight"] = (3 * C, C)
            shapes[lp + attn + ".in_proj_bias"] = (3 * C,)
            shapes[lp + attn + ".out_proj.weight"] = (C, C)
            shapes[lp + attn + ".out_proj.bias"] = (C,)
        shapes[lp + "linear1.weight"] = (F, C)
        shapes[lp + "linear1.bias"] = (F,)
        shapes[lp + "linear2.weight"] = (C, F)
        shapes[lp + "linear2.bias"] = (C,)
        for n in ("norm1", "norm2", "norm3"):
            shapes[lp + n + ".weight"] = (C,)
            shapes[lp + n + ".bias"] = (C,)
    shapes[d + "norm.weight"] = (C,)          # present in checkpoints, never applied
    shapes[d + "norm.bias"] = (C,)            # (model/transformer_parq.py:82-84,174)
    shapes[d + "position_encoder.0.weight"] = (C, 384)
    shapes[d + "position_encoder.0.bias"] = (C,)
    shapes[d + "position_encoder.2.weight"] = (C, C)
    shapes[d + "position_encoder.2.bias"] = (C,)
    return shapes


def make_decoder_weights(cfg, seed: int, damped: bool = False) -> dict:
    """Random-init weights of the reference architecture.

    Matrices are xavier-uniform (model/transformer_parq.py:90-93); biases and
    norm affines are *non-trivial* (small random) so that a kernel that drops a
    bias or an affine term fails parity.  ``damped`` scales the last layer of
    the centre head by 0.05, the fixture SURVEY Appendix D found stable enough
    for free-running 1e-4 parity.
    """
    out = {}
    for name, shape in decoder_param_shapes(cfg).items():
        if name == "refpoint.weight":
            w = normal(seed, name, shape, std=1.0)          # nn.Embedding default
        elif name.endswith("norm.weight") or (name.endswith(".weight") and len(shape) == 1):
            w = uniform(seed, name, shape, 0.8, 1.2)
        elif name.endswith("bias"):
            w = uniform(seed, name, shape, -0.1, 0.1)
        else:
            fan_out, fan_in = shape[0], shape[1]
            a = float(np.sqrt(6.0 / (fan_in + fan_out)))
            w = uniform(seed, name, shape, -a, a)
        if damped and name == "mlp_heads.center_head.layers.8.weight":
            w = (w * np.float32(0.05)).astype(np.float32)
        if damped and name == "mlp_heads.center_head.layers.8.bias":
            w = (w * np.float32(0.05)).astype(np.float32)
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def ray_pe_param_shapes(dim_out: int, num_samples: int = 64) -> dict:
    """AddRayPE.state_dict() (model/ray_positional_encoding.py:55-59)."""
    return {"encoder.0.weight": (dim_out, 3 * num_samples),
            "encoder.0.bias": (dim_out,),
            "encoder.2.weight": (dim_out, dim_out),
            "encoder.2.bias": (dim_out,)}


def make_ray_pe_weights(dim_out: int, seed: int, num_samples: int = 64) -> dict:
    out = {}
    for name, shape in ray_pe_param_shapes(dim_out, num_samples).items():
        if name.endswith("bias"):
            w = uniform(seed, "raype." + name, shape, -0.1, 0.1)
        else:
            a = float(np.sqrt(6.0 / (shape[0] + shape[1])))
            w = uniform(seed, "raype." + name, shape, -a, a)
        out[name] = w
    return out


# --------------------------------------------------------------------------
# scenes
# --------------------------------------------------------------------------

def _rotations(seed: int, name: str, n: int, amount: float) -> np.ndarray:
    """n proper rotations near identity: QR of (I + amount * N(0,1))."""
    a = np.eye(3, dtype=np.float64)[None] + amount * normal(seed, name, (n, 3, 3)).astype(np.float64)
    q, r = np.linalg.qr(a)
    q = q * np.sign(np.diagonal(r, axis1=1, axis2=2))[:, None, :]
    det = np.linalg.det(q)
    q[:, :, 2] *= det[:, None]
    return q


def _pose(Rm: np.ndarray, t: np.ndarray) -> np.ndarray:
    return np.concatenate([Rm.reshape(Rm.shape[0], 9), t], axis=1).astype(np.float32)


def make_geometry(seed: int, B: int, V: int, h: int, w: int):
    """Cameras at *feature* scale and the three pose stacks of one batch.

    camera (B,V,6): size (w,h), fx=fy=0.9*w, principal point at the centre —
    what ``Camera.scale(1/4)`` hands the decoder (model/resnet_fpn.py:85-90).
    T_world_local is the pseudo-camera pose of the middle view
    (datasets/transforms.py:201-208).
    """
    cam = np.zeros((B, V, 6), dtype=np.float32)
    cam[..., 0] = w
    cam[..., 1] = h
    cam[..., 2] = 0.9 * w
    cam[..., 3] = 0.9 * w
    cam[..., 4] = (w - 1) / 2.0
    cam[..., 5] = (h - 1) / 2.0
    n = B * V
    R_wp = _rotations(seed, "R_wp", n, 0.10)
    t_wp = 0.2 * normal(seed, "t_wp", (n, 3)).astype(np.float64)
    R_cp = _rotations(seed, "R_cp", n, 0.05)     # gravity-alignment rotation, no translation
    t_cp = np.zeros((n, 3))
    T_wp = _pose(R_wp, t_wp).reshape(B, V, 12)
    T_cp = _pose(R_cp, t_cp).reshape(B, V, 12)
    T_wl = T_wp[:, V // 2: V // 2 + 1, :].copy()
    return cam, T_cp, T_wp, T_wl


def make_tokens(seed: int, B: int, V: int, h: int, w: int, C: int, smooth: bool = False) -> np.ndarray:
    """(B, V*h*w, C) channels-last tokens: white noise, or — for the damped
    free-running fixture — three low-frequency sinusoids per channel."""
    if not smooth:
        return normal(seed, "tokens", (B, V * h * w, C))
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    amp = normal(seed, "tok.amp", (B, V, 3, C)).astype(np.float64)
    fx = uniform(seed, "tok.fx", (B, V, 3, C), -3.0, 3.0).astype(np.float64) / w
    fy = uniform(seed, "tok.fy", (B, V, 3, C), -3.0, 3.0).astype(np.float64) / h
    ph = uniform(seed, "tok.ph", (B, V, 3, C), 0.0, 2 * np.pi).astype(np.float64)
    out = np.zeros((B, V, h, w, C), dtype=np.float64)
    for k in range(3):
        arg = (2 * np.pi) * (xx[None, None, :, :, None] * fx[:, :, None, None, k, :]
                             + yy[None, None, :, :, None] * fy[:, :, None, None, k, :]) \
            + ph[:, :, None, None, k, :]
        out += amp[:, :, None, None, k, :] * np.sin(arg)
    return (out / np.sqrt(1.5)).astype(np.float32).reshape(B, V * h * w, C)


def make_scene(seed: int, B: int, V: int, h: int, w: int, C: int, smooth: bool = False):
    cam, T_cp, T_wp, T_wl = make_geometry(seed, B, V, h, w)
    tokens = make_tokens(seed, B, V, h, w, C, smooth=smooth)
    return dict(tokens=tokens, camera=cam, T_camera_pseudoCam=T_cp,
                T_world_pseudoCam=T_wp, T_world_local=T_wl)


def make_boxes(seed: int, B: int, nbox: int, max_box: int = 100):
    """Synthetic ground truth for the loss / training step: (B, max_box, 19) padded Obb3D rows
    [xmin,xmax,ymin,ymax,zmin,zmax | R (9) t (3) | class], all -1 = padding (utils/wrappers.py:297-409), and the
    symmetry classes (B, max_box) padded with -1 (datasets/scannet_dataset.py:151-165)."""
    obbs = -np.ones((B, max_box, 19), np.float32)
    sym = -np.ones((B, max_box), np.float32)
    half = uniform(seed, "box_half", (B, nbox, 3), 0.15, 0.7)
    ang = uniform(seed, "box_ang", (B, nbox), -np.pi, np.pi)
    t = uniform(seed, "box_t", (B, nbox, 3), -1.0, 1.0) * np.array([2.5, 1.0, 2.0], np.float32) + np.array([0.0, -0.75, 2.75], np.float32)
    cls = (uniform(seed, "box_cls", (B, nbox), 0.0, 8.999)).astype(np.int64)
    sy = (uniform(seed, "box_sym", (B, nbox), 0.0, 3.999)).astype(np.int64)
    for b in range(B):
        for j in range(nbox):
            c, s_ = np.cos(ang[b, j]), np.sin(ang[b, j])
            obbs[b, j, 0:6] = [-half[b, j, 0], half[b, j, 0], -half[b, j, 1], half[b, j, 1], -half[b, j, 2], half[b, j, 2]]
            obbs[b, j, 6:15] = [c, 0, s_, 0, 1, 0, -s_, 0, c]
            obbs[b, j, 15:18] = t[b, j]
            obbs[b, j, 18] = cls[b, j]
            sym[b, j] = sy[b, j]
    return obbs, sym


# per-class mean box sizes of the 8 named ScanNet classes in class-id order
# (chair, table, cabinet, trash bin, bookshelf, display, sofa, bathtub), then
# "other" and "non-object" = [1,1,1].  These are the rows BoxProcessor builds
# from its MEAN_SIZE_PATH file (utils/parq_utils.py:45-88).
SCANNET_MEAN_SIZES = np.array([
    [0.55067552, 0.84943989, 0.5786128],
    [1.24506049, 0.66165523, 0.72455878],
    [0.95658434, 0.99974904, 0.56246602],
    [0.36641966, 0.45580824, 0.27876528],
    [1.05132399, 1.3471979, 0.33744382],
    [0.60740744, 0.4752175, 0.16435075],
    [1.68820774, 0.76637348, 0.89351734],
    [0.85305378, 0.43925023, 0.51612006],
    [1.0, 1.0, 1.0],
    [1.0, 1.0, 1.0],
], dtype=np.float64)
