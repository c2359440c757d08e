"""Test infrastructure: CPU emulation of the cross-attention ARITHMETIC of the HIP kernels on top of the float64 oracle, to size the
error of an arithmetic before (and independently of) the kernel that implements it.

  "split"   q.k and p.v as three fp16 products, fp32 accumulate:  hi.hi + hi.lo + lo.hi           (flash_split.hip, mode 1)
  "fp16"    one fp16 product (operands rounded once)                                               (mode 2)
  "split8"  hi.hi as an fp16 product, the two cross terms as fp8 (e4m3) products:
            hi16.hi16 + e4m3(a) . e4m3(b_lo 2^10) 2^-10 + e4m3(a_lo 2^10) 2^-10 . e4m3(b)          (mode 4: MX-scaled fp8 MFMA, scale 2^-10)
            hi16 rounds toward zero as in the kernels; probabilities enter their fp8 forms scaled by 2^6 (e4m3 has no values under 2^-9)
            (the decoder's kernel uses this product for the SCORES only; its P V is one fp16 product of probabilities and values rounded
            to nearest, normalised by the sum of the same rounded probabilities: tests/calibrate_split8_guard.py models that form, "p16")

`patched(mode)` swaps oracle.mha for the emulation on every attention whose key axis is longer than 1024 (the cross-attention).
Everything else of the oracle stays float64, so the difference to the un-patched oracle IS the arithmetic's error."""
import contextlib
import math

import torch
import torch.nn.functional as F

from oracle import parq_oracle as O

E4 = torch.float8_e4m3fn


def _h(x):
    return x.float().half().double()


def _rtz16(x):
    """fp16 of x rounded toward zero — the hi part the kernels use (v_cvt_pkrtz_f16_f32)."""
    x = x.float()
    h = x.half().float()
    over = h.abs() > x.abs()
    h[over] = torch.nextafter(h[over].half(), torch.zeros_like(h[over]).half()).float()
    return h.double()


def _e4(x, scale=1.0):
    y = (x * scale).float().clamp(-448.0, 448.0).to(E4).double()
    return y / scale


def product(a, b, mode, a_scale8=1.0):
    """a (.., m, k) @ b (.., k, n) in the arithmetic `mode`; operands arrive as float64 holding fp32 values."""
    a, b = a.float().double(), b.float().double()
    if mode == "fp16":
        return _h(a) @ _h(b)
    ah, bh = _rtz16(a), _rtz16(b)
    al, bl = a - ah, b - bh
    if mode == "split":
        return ah @ bh + ah @ _h(bl) + _h(al) @ bh
    if mode == "split8":
        lo = 1024.0
        return ah @ bh + _e4(a, a_scale8) @ _e4(bl, lo) + _e4(al, lo * a_scale8) @ _e4(b)
    raise ValueError(mode)


PROJECTION_MODE = [None]          # None: K / V projection exact (float64); else the arithmetic of the projection GEMM


def mha_emulated(mode):
    def mha(query, key, value, in_w, in_b, out_w, out_b, H, reference_ops=False):
        B, L, C = query.shape
        S = key.shape[1]
        if S <= 1024:
            return _exact(query, key, value, in_w, in_b, out_w, out_b, H, reference_ops)
        dh = C // H
        q = F.linear(query, in_w[:C], in_b[:C]).view(B, L, H, dh).transpose(1, 2)
        if PROJECTION_MODE[0] is None:
            k = F.linear(key, in_w[C:2 * C], in_b[C:2 * C])
            v = F.linear(value, in_w[2 * C:], in_b[2 * C:])
        else:
            k = product(key, in_w[C:2 * C].t(), PROJECTION_MODE[0]) + in_b[C:2 * C]
            v = product(value, in_w[2 * C:].t(), PROJECTION_MODE[0]) + in_b[2 * C:]
        k, v = k.view(B, S, H, dh).transpose(1, 2), v.view(B, S, H, dh).transpose(1, 2)
        # the kernels scale q by log2(e) / sqrt(dh) before it is split and work in base 2
        s = product(q * (math.log2(math.e) / math.sqrt(dh)), k.transpose(-1, -2), mode)
        p = torch.exp2(s - s.max(-1, keepdim=True).values).float().double()
        o = product(p, v, mode, a_scale8=64.0) / p.sum(-1, keepdim=True)
        return F.linear(o.transpose(1, 2).reshape(B, L, C), out_w, out_b)
    return mha


_exact = O.mha


@contextlib.contextmanager
def patched(mode):
    O.mha = mha_emulated(mode)
    try:
        yield
    finally:
        O.mha = _exact
