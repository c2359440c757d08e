"""CPU oracle for the PARQ recurrent pixel-aligned decoder path.

TEST INFRASTRUCTURE ONLY.  This file is a plain-PyTorch (CPU) restatement of
the reference algorithm; it is imported by ``tests/``, by
``__graft_entry__.smoke()`` and by ``bench.py``'s ``cpu_baseline`` leg, and by
nothing else.  The product path (``parq_amd``) never routes through it.

Parity status: PINNED.  The reference ships no tests or golden vectors
(SURVEY.md §4), so the pin is made by this repo: ``oracle/make_golden.py``
imports the real reference from ``/root/reference`` in the dev container,
runs it on the seeded synthetic inputs of ``parq_amd/synth.py`` and commits
the *outputs* under ``tests/golden``; ``tests/test_oracle_golden.py`` checks
this restatement against every one of those vectors.

All ``file:line`` citations are relative to the reference tree.

Arithmetic note: the numerics of the path live in third-party torch ops
(``F.grid_sample``, ``nn.MultiheadAttention``, ``LayerNorm``, ``GroupNorm``;
reference pins torch 1.12.1, environment.yml:16).  ``reference_ops=True``
issues the same ATen op sequence the reference does (per-iteration K/V
in-projection, materialised (B*H,Q,N) softmax, head-averaged attention
weights, ``grid_sampler_2d``) and is what the CPU baseline times;
``reference_ops=False`` spells the same math out explicitly.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# SE(3) / pinhole helpers on raw 12- and 6-vectors
# --------------------------------------------------------------------------

def pose_R(p):                      # utils/wrappers.py:224-228
    return p[..., :9].reshape(p.shape[:-1] + (3, 3))


def pose_t(p):                      # utils/wrappers.py:230-233
    return p[..., 9:]


def pose_from_Rt(Rm, t):            # utils/wrappers.py:199-213
    return torch.cat([Rm.flatten(start_dim=-2), t], -1)


def pose_inverse(p):                # utils/wrappers.py:247-251
    Rm = pose_R(p).transpose(-1, -2)
    t = -(Rm @ pose_t(p).unsqueeze(-1)).squeeze(-1)
    return pose_from_Rt(Rm, t)


def pose_compose(a, b):             # utils/wrappers.py:253-257  (a ∘ b)
    Rm = pose_R(a) @ pose_R(b)
    t = pose_t(a) + (pose_R(a) @ pose_t(b).unsqueeze(-1)).squeeze(-1)
    return pose_from_Rt(Rm, t)


def pose_transform(p, x):           # utils/wrappers.py:259-267
    return x @ pose_R(p).transpose(-1, -2) + pose_t(p).unsqueeze(-2)


def camera_project(cam, p3d, eps=1e-3):     # utils/wrappers.py:502-522
    z = p3d[..., -1]
    in_front = z > eps
    z = z.clamp(min=eps)
    p2d = p3d[..., :-1] / z.unsqueeze(-1)
    p2d = p2d * cam[..., 2:4].unsqueeze(-2) + cam[..., 4:6].unsqueeze(-2)
    size = cam[..., :2].unsqueeze(-2)
    in_img = torch.all((p2d >= 0) & (p2d <= (size - 1)), -1)
    return p2d, in_front & in_img


# --------------------------------------------------------------------------
# small pieces
# --------------------------------------------------------------------------

def inverse_sigmoid(x, eps=1e-3):   # model/transformer_parq.py:38-42
    x = x.clamp(min=0, max=1)
    x1 = x.clamp(min=eps)
    x2 = (1 - x).clamp(min=eps)
    return torch.log(x1 / x2)


def pos2posemb3d(pos, num_pos_feats=128, temperature=10000):   # model/transformer_parq.py:45-64
    pos = pos * (2 * math.pi)
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = (temperature ** (2 * (dim_t // 2) / num_pos_feats)).to(pos.dtype)
    embs = []
    for axis in (1, 0, 2):                      # concat order (y, x, z)
        a = pos[..., axis, None] / dim_t
        e = torch.stack((a[..., 0::2].sin(), a[..., 1::2].cos()), dim=-1).flatten(-2)
        embs.append(e)
    return torch.cat(embs, dim=-1)


def denormalize(x, scale):          # model/transformer_parq.py:198-209
    lo = x.new_tensor(scale[0::2])
    hi = x.new_tensor(scale[1::2])
    return x * (hi - lo) + lo


def normalize(x, scale):            # model/transformer_parq.py:185-196
    lo = x.new_tensor(scale[0::2])
    hi = x.new_tensor(scale[1::2])
    return (x - lo) / (hi - lo)


def camera_local_poses(T_cp, T_wp, T_wl):   # model/transformer_parq.py:298-300
    """T_camera_local (B,V,12) = T_cp ∘ (inv(T_wp) ∘ T_wl)."""
    return pose_compose(T_cp, pose_compose(pose_inverse(T_wp), T_wl))


def bilinear_zeros_align(feat, u, v):
    """Explicit bilinear sample of channels-last feat (V,h,w,C) at pixel coords
    (u,v) (V,Q): zero padding, align_corners=True — the semantics of the
    reference's ``grid_sample`` call (model/transformer_parq.py:148-152)."""
    Vn, h, w, C = feat.shape
    x0 = torch.floor(u)
    y0 = torch.floor(v)
    out = feat.new_zeros(Vn, u.shape[1], C)
    vi = torch.arange(Vn)[:, None].expand(Vn, u.shape[1])
    for dy in (0, 1):
        for dx in (0, 1):
            xx = x0 + dx
            yy = y0 + dy
            wgt = (1 - (u - xx).abs()) * (1 - (v - yy).abs())
            ok = (xx >= 0) & (xx <= w - 1) & (yy >= 0) & (yy <= h - 1)
            xi = xx.clamp(0, w - 1).long()
            yi = yy.clamp(0, h - 1).long()
            out = out + feat[vi, yi, xi] * (wgt * ok.to(wgt.dtype)).unsqueeze(-1)
    return out


def project_and_sample(tokens, ref_denorm, T_cl, cam, h, w, reference_ops=False, raw=False):
    """model/transformer_parq.py:129-161.  tokens (B,N,C) channels-last,
    ref_denorm (B,Q,3) in the local frame, T_cl/cam (B,V,·).  Returns the
    view-averaged pixel-aligned features (B,Q,C), pixel coords, validity."""
    B, N, C = tokens.shape
    Vn = T_cl.shape[1]
    pc = pose_transform(T_cl, ref_denorm.unsqueeze(1))        # (B,V,Q,3)
    p2d, valid = camera_project(cam, pc)                      # (B,V,Q,2), (B,V,Q)
    if reference_ops:
        mem = tokens.view(B * Vn, h, w, C).permute(0, 3, 1, 2)
        grid = torch.stack([2 * p2d[..., 0] / (w - 1) - 1, 2 * p2d[..., 1] / (h - 1) - 1], dim=-1)
        grid = grid.view(B * Vn, 1, -1, 2)
        f = F.grid_sample(mem, grid, padding_mode="zeros", align_corners=True)
        f = f.view(B, Vn, C, -1).permute(0, 1, 3, 2)
    else:
        feat = tokens.view(B, Vn, h, w, C)
        # the reference round-trips pixel -> normalised grid -> pixel
        gx = 2 * p2d[..., 0] / (w - 1) - 1
        gy = 2 * p2d[..., 1] / (h - 1) - 1
        u = (gx + 1) * ((w - 1) / 2)
        v = (gy + 1) * ((h - 1) / 2)
        f = torch.stack([bilinear_zeros_align(feat[b], u[b], v[b]) for b in range(B)])
    f = f.sum(dim=1)                                           # ALL views summed
    cnt = valid.sum(dim=1)
    if raw:                                                    # view-sharded scenes: the undivided sum and the valid-view count
        return f, cnt, p2d, valid
    cnt = torch.where(cnt == 0, torch.ones_like(cnt), cnt)     # divide by max(#valid, 1)
    return f / cnt.unsqueeze(-1).to(f.dtype), p2d, valid


def layer_norm(x, w, b, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def mha(query, key, value, in_w, in_b, out_w, out_b, H, reference_ops=False):
    """nn.MultiheadAttention forward, batch-first here: (B,L,C) x (B,S,C).
    model/transformer_parq.py:345-346,373-380."""
    if reference_ops:
        o, _ = F.multi_head_attention_forward(
            query.transpose(0, 1), key.transpose(0, 1), value.transpose(0, 1),
            query.shape[-1], H, in_w, in_b, None, None, False, 0.0, out_w, out_b,
            training=False, need_weights=True)      # head-averaged weights computed, discarded
        return o.transpose(0, 1)
    B, L, C = query.shape
    S = key.shape[1]
    dh = C // H
    q = F.linear(query, in_w[:C], in_b[:C]).view(B, L, H, dh).transpose(1, 2)
    k = F.linear(key, in_w[C:2 * C], in_b[C:2 * C]).view(B, S, H, dh).transpose(1, 2)
    v = F.linear(value, in_w[2 * C:], in_b[2 * C:]).view(B, S, H, dh).transpose(1, 2)
    a = torch.softmax((q * (1.0 / math.sqrt(dh))) @ k.transpose(-1, -2), dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, L, C)
    return F.linear(o, out_w, out_b)


def cross_attention_shard(query, memory, in_w, in_b, H):
    """The cross-attention of ``mha`` restricted to the keys of ``memory`` (a shard of the scene's tokens), before the output
    projection: returns the shard-normalised output (B,H,L,dh) and the natural-log log-sum-exp of the shard's scores (B,H,L) —
    what the ranks of a view-sharded scene exchange (parq_amd/parallel.py merge_attention_shards)."""
    B, L, C = query.shape
    S = memory.shape[1]
    dh = C // H
    q = F.linear(query, in_w[:C], in_b[:C]).view(B, L, H, dh).transpose(1, 2)
    k = F.linear(memory, in_w[C:2 * C], in_b[C:2 * C]).view(B, S, H, dh).transpose(1, 2)
    v = F.linear(memory, in_w[2 * C:], in_b[2 * C:]).view(B, S, H, dh).transpose(1, 2)
    sc = (q * (1.0 / math.sqrt(dh))) @ k.transpose(-1, -2)
    lse = torch.logsumexp(sc, dim=-1)
    return torch.softmax(sc, dim=-1) @ v, lse


def head_mlp(x_bcq, Wd, prefix):
    """GenericMLP with hidden [C,C]: [Conv1d(no bias) -> GroupNorm(1,C) -> ReLU] x2
    -> Conv1d(bias).  model/generic_mlp.py:85-110.  x is (B,C,Q); GroupNorm(1,C)
    normalises over ALL C*Q elements of a scene."""
    y = F.conv1d(x_bcq, Wd[prefix + "0.weight"])
    y = F.relu(F.group_norm(y, 1, Wd[prefix + "1.weight"], Wd[prefix + "1.bias"], 1e-5))
    y = F.conv1d(y, Wd[prefix + "4.weight"])
    y = F.relu(F.group_norm(y, 1, Wd[prefix + "5.weight"], Wd[prefix + "5.bias"], 1e-5))
    return F.conv1d(y, Wd[prefix + "8.weight"], Wd[prefix + "8.bias"])


def box_heads(x, ref, Wd, scale, mean_sizes):
    """model/transformer_parq.py:211-281 + utils/parq_utils.py:90-105.
    x (B,Q,C) decoder output, ref (B,Q,3) normalised reference points."""
    t = x.permute(0, 2, 1).contiguous()
    logits = F.conv1d(t, Wd["mlp_heads.sem_cls_head.layers.0.weight"],
                      Wd["mlp_heads.sem_cls_head.layers.0.bias"]).transpose(1, 2)
    ctr = head_mlp(t, Wd, "mlp_heads.center_head.layers.").transpose(1, 2)
    coord_pos = denormalize(ref, scale)
    ctr = denormalize((ctr + inverse_sigmoid(ref)).sigmoid(), scale)
    size_scale = F.conv1d(t, Wd["mlp_heads.size_head.layers.0.weight"],
                          Wd["mlp_heads.size_head.layers.0.bias"]).transpose(1, 2)
    rot = head_mlp(t, Wd, "mlp_heads.rotation_head.layers.").transpose(1, 2)
    prob = torch.softmax(logits, dim=-1)
    cls = prob.argmax(-1)
    size = torch.exp(size_scale) * mean_sizes[cls].float().to(size_scale.dtype)   # table rounded to fp32 (:98)
    return {"pred_logits": logits, "center_unnormalized": ctr, "size_unnormalized": size,
            "ortho6d": rot, "sem_cls_prob": prob, "coord_pos": coord_pos}


# --------------------------------------------------------------------------
# the decoder
# --------------------------------------------------------------------------

def _as_torch(Wd, dtype):
    out = {}
    for k, v in Wd.items():
        t = torch.as_tensor(np.asarray(v)) if not torch.is_tensor(v) else v
        out[k] = t.detach().to("cpu", dtype)
    return out


class OracleDecoder:
    """Functional restatement of PARQDecoder.forward
    (model/parq_decoder.py:134-163 -> model/transformer_parq.py:95-126,283-337)."""

    def __init__(self, cfg, weights: dict, mean_sizes, dtype=torch.float32, reference_ops=False):
        self.cfg = cfg
        self.T = cfg.TRANSFORMER
        self.dtype = dtype
        self.W = _as_torch(weights, dtype)
        # float64 table cast to the compute type at use (utils/parq_utils.py:88,98)
        self.mean_sizes = torch.as_tensor(np.asarray(mean_sizes, dtype=np.float64))
        self.reference_ops = reference_ops

    # one decoder layer, post-norm                      model/transformer_parq.py:365-386
    def layer(self, tgt, memory, pos, li):
        W, H, ro = self.W, self.T.DEC_HEADS, self.reference_ops
        p = "parq_module.decoder.layers.%d." % li
        q = tgt + pos
        t2 = mha(q, q, tgt, W[p + "self_attn.in_proj_weight"], W[p + "self_attn.in_proj_bias"],
                 W[p + "self_attn.out_proj.weight"], W[p + "self_attn.out_proj.bias"], H, ro)
        x = layer_norm(tgt + t2, W[p + "norm1.weight"], W[p + "norm1.bias"])
        t2 = mha(x + pos, memory, memory,
                 W[p + "multihead_attn.in_proj_weight"], W[p + "multihead_attn.in_proj_bias"],
                 W[p + "multihead_attn.out_proj.weight"], W[p + "multihead_attn.out_proj.bias"], H, ro)
        self._x1, self._cross = x, t2
        x = layer_norm(x + t2, W[p + "norm2.weight"], W[p + "norm2.bias"])
        self._x2 = x
        t2 = F.linear(F.relu(F.linear(x, W[p + "linear1.weight"], W[p + "linear1.bias"])),
                      W[p + "linear2.weight"], W[p + "linear2.bias"])
        return layer_norm(x + t2, W[p + "norm3.weight"], W[p + "norm3.bias"])

    def prepare(self, tokens, camera, T_cp, T_wp, T_wl):
        dt = self.dtype
        self.tokens = torch.as_tensor(tokens).to(dt)
        self.cam = torch.as_tensor(camera).to(dt)
        self.T_cl = camera_local_poses(torch.as_tensor(T_cp).to(dt), torch.as_tensor(T_wp).to(dt),
                                       torch.as_tensor(T_wl).to(dt))
        w, h = self.cam[0, 0, :2].tolist()                   # model/transformer_parq.py:301-302
        self.h, self.w = int(h), int(w)

    def initial_ref(self):                                   # model/transformer_parq.py:122,309
        B = self.tokens.shape[0]
        return self.W["refpoint.weight"].sigmoid().unsqueeze(0).repeat(B, 1, 1)

    def iterate(self, ref, layer_num):
        """One recurrent iteration from normalised reference points (B,Q,3):
        returns (out_dict, next_ref, intermediates)."""
        W = self.W
        li = 0 if self.T.SHARE_WEIGHTS else layer_num
        d = "parq_module.decoder.position_encoder."
        pos = F.linear(F.relu(F.linear(pos2posemb3d(ref), W[d + "0.weight"], W[d + "0.bias"])),
                       W[d + "2.weight"], W[d + "2.bias"])
        tgt, p2d, valid = project_and_sample(self.tokens, denormalize(ref, self.T.SCALE), self.T_cl,
                                             self.cam, self.h, self.w, self.reference_ops)
        x = self.layer(tgt, self.tokens, pos, li)
        out = box_heads(x, ref, W, self.T.SCALE, self.mean_sizes)
        nxt = normalize(out["center_unnormalized"], self.T.SCALE)     # :331-332 (detached)
        return out, nxt, {"pos": pos, "tgt": tgt, "x": x, "p2d": p2d, "valid": valid,
                          "x1": self._x1, "x2": self._x2, "cross_out": self._cross}

    def iterate_sharded(self, ref, layer_num, merge_sample, merge_attention):
        """``iterate`` for a rank that holds only SOME views of the scene (``prepare`` was called with those views): the sampled
        feature sums / valid-view counts and the cross-attention over the local keys go through the two merge callbacks
        (parq_amd/parallel.py merge_sample_sums / merge_attention_shards); everything else is as in ``iterate`` / ``layer``.
        No reference analogue (the reference keeps a scene in one process): the single-process ``iterate`` is the truth."""
        W, H = self.W, self.T.DEC_HEADS
        li = 0 if self.T.SHARE_WEIGHTS else layer_num
        d = "parq_module.decoder.position_encoder."
        pos = F.linear(F.relu(F.linear(pos2posemb3d(ref), W[d + "0.weight"], W[d + "0.bias"])), W[d + "2.weight"], W[d + "2.bias"])
        sums, cnt, _, _ = project_and_sample(self.tokens, denormalize(ref, self.T.SCALE), self.T_cl, self.cam, self.h, self.w, raw=True)
        tgt = merge_sample(sums, cnt)
        p = "parq_module.decoder.layers.%d." % li
        q = tgt + pos
        t2 = mha(q, q, tgt, W[p + "self_attn.in_proj_weight"], W[p + "self_attn.in_proj_bias"],
                 W[p + "self_attn.out_proj.weight"], W[p + "self_attn.out_proj.bias"], H)
        x = layer_norm(tgt + t2, W[p + "norm1.weight"], W[p + "norm1.bias"])
        o, lse = cross_attention_shard(x + pos, self.tokens, W[p + "multihead_attn.in_proj_weight"], W[p + "multihead_attn.in_proj_bias"], H)
        o = merge_attention(o, lse)                                      # (B,H,L,dh) over ALL keys of the scene
        B, L = o.shape[0], o.shape[2]
        t2 = F.linear(o.transpose(1, 2).reshape(B, L, -1), W[p + "multihead_attn.out_proj.weight"], W[p + "multihead_attn.out_proj.bias"])
        x = layer_norm(x + t2, W[p + "norm2.weight"], W[p + "norm2.bias"])
        t2 = F.linear(F.relu(F.linear(x, W[p + "linear1.weight"], W[p + "linear1.bias"])), W[p + "linear2.weight"], W[p + "linear2.bias"])
        x = layer_norm(x + t2, W[p + "norm3.weight"], W[p + "norm3.bias"])
        out = box_heads(x, ref, W, self.T.SCALE, self.mean_sizes)
        return out, normalize(out["center_unnormalized"], self.T.SCALE)

    def forward(self, tokens, camera, T_cp, T_wp, T_wl, forced_refs=None):
        """Free-running (forced_refs=None) or teacher-forced: iteration k is fed
        forced_refs[k] instead of its own previous prediction."""
        self.prepare(tokens, camera, T_cp, T_wp, T_wl)
        ref = self.initial_ref()
        outs = []
        for k in range(self.T.DEC_LAYERS):
            if forced_refs is not None:
                ref = torch.as_tensor(forced_refs[k]).to(self.dtype)
            out, ref, _ = self.iterate(ref, k)
            outs.append(out)
        return outs


# --------------------------------------------------------------------------
# ray positional encoding (once per forward)
# --------------------------------------------------------------------------

def ray_pe(camera, T_cp, T_wp, T_wl, Wd, ray_points_scale, num_samples=64, min_depth=0.25,
           max_depth=5.25, dtype=torch.float32):
    """AddRayPE.forward (model/ray_positional_encoding.py:61-139) with
    ray_points / ray_points_snippet (utils/encoding_utils.py:23-100).
    Returns the encoding (B,V,C,h,w)."""
    cam = torch.as_tensor(camera).to(dtype)
    T_cp = torch.as_tensor(T_cp).to(dtype)
    T_wp = torch.as_tensor(T_wp).to(dtype)
    T_wl = torch.as_tensor(T_wl).to(dtype)
    W = _as_torch(Wd, dtype)
    B, Vn = cam.shape[:2]
    width = int(round(float(cam[0, 0, 0])))
    height = int(round(float(cam[0, 0, 1])))
    # integer pixel grid, no +0.5 (utils/encoding_utils.py:15-20)
    xs = torch.linspace(0.0, width, width + 1)[:-1].to(dtype)
    ys = torch.linspace(0.0, height, height + 1)[:-1].to(dtype)
    xx, yy = torch.meshgrid(xs, ys, indexing="xy")                   # (h,w)
    uv = torch.stack([xx, yy], -1).reshape(1, -1, 2)                 # (1,h*w,2)
    camf = cam.reshape(B * Vn, 6)
    rays = (uv - camf[:, None, 4:6]) / camf[:, None, 2:4]            # utils/wrappers.py:543-548
    rays = torch.cat([rays, rays.new_ones(B * Vn, rays.shape[1], 1)], -1)
    ramp = torch.linspace(0, 1, num_samples).view(1, 1, num_samples, 1)
    mind = torch.tensor([min_depth])[0]
    maxd = torch.tensor([max_depth])[0]
    depth = torch.exp(torch.log(mind) + torch.log(maxd / mind) * ramp).to(dtype)
    pts = (rays.unsqueeze(-2) * depth).view(B * Vn, -1, 3)           # camera frame
    pts = pose_transform(pose_inverse(T_cp.reshape(B * Vn, 12)), pts)    # -> pseudoCam
    T_lp = pose_compose(pose_inverse(T_wl), T_wp).reshape(B * Vn, 12)    # local <- pseudoCam
    pts = pose_transform(T_lp, pts)
    pts = pts.view(B * Vn, height, width, num_samples, 3)
    lo = pts.new_tensor(ray_points_scale[0::2])
    hi = pts.new_tensor(ray_points_scale[1::2])
    pts = inverse_sigmoid((pts - lo) / (hi - lo))
    pts = pts.reshape(B * Vn, height, width, num_samples * 3)
    enc = F.linear(F.relu(F.linear(pts, W["encoder.0.weight"], W["encoder.0.bias"])),
                   W["encoder.2.weight"], W["encoder.2.bias"])
    return enc.view(B, Vn, height, width, -1).permute(0, 1, 4, 2, 3)


def tokenize(features, encoding):
    """model/parq_lightning.py:75-85: (B,V,C,h,w) + PE -> (B, V*h*w, C)."""
    x = features + encoding
    B, Vn, C, h, w = x.shape
    return x.permute(0, 1, 3, 4, 2).reshape(B, Vn * h * w, C)
