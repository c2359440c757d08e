"""AddRayPE: host-side mirror of model/ray_positional_encoding.py:29-139 on the HIP path.

Same constructor arguments, same ``encoder.{0,2}`` state_dict keys, same ``forward`` signature
and result (the encoding, shaped (B, T, C, H, W)) as the reference module.  The arithmetic runs in
libparq_hip.so (``parq_ray_pe``); ``tokens()`` is the fused fast path that adds the feature maps
and returns the channels-last token tensor the decoder consumes, skipping the NCHW round trip of
model/parq_lightning.py:72-85.
"""
from __future__ import annotations

import ctypes as C
import weakref

import torch
from torch import nn

from . import _lib
from .wrappers import raw


class _RayPeFn(torch.autograd.Function):
    """tokens = features + encoding as one autograd node: backward = parq_ray_pe_backward (gradients of the encoder MLP and
    of the feature maps; the ray geometry has no learnable part)."""

    @staticmethod
    def forward(ctx, mod, features, camera, T_cp, T_wp, T_wl, w1, b1, w2, b2):
        out, dims, _ = mod._run(camera, T_cp, T_wp, T_wl, tuple(features.shape[-2:]), features, own_workspace=True)
        ctx.mod, ctx.dims = mod, dims
        ctx.geo = mod._last_geo
        # the node owns the workspace of its forward (the hidden layer the backward reads): a later call, while this node is
        # alive and has not run its backward, takes a workspace of its own (several outstanding forwards per module)
        ctx.hold = _WsHold(mod._ws, mod._gen)
        mod._ws_owner = weakref.ref(ctx.hold)
        ctx.want_feat = bool(features.requires_grad)
        return out

    @staticmethod
    def backward(ctx, g_tokens):
        mod = ctx.mod
        hold = ctx.hold
        if getattr(hold.ws, "_parq_gen", None) != hold.gen:
            raise RuntimeError("parq_amd.AddRayPE: second backward (retain_graph=True) through a tokens() call whose workspace was "
                               "released by the first backward and has since been reused by a later call of the same module")
        B, V, h, w = ctx.dims
        Cd, S = mod.dim_out, mod.num_samples
        cam, T_cp, T_wp, T_wl = ctx.geo
        dev = cam.device
        lib = _lib.load()
        g = g_tokens.to(dtype=torch.float32).contiguous()
        nbytes = lib.parq_ray_pe_backward_workspace_bytes(B, V, h, w, Cd, S)
        bws = torch.empty(nbytes // 4 + 1, dtype=torch.float32, device=dev)
        # the four parameter gradients live in ONE flat buffer, so data-parallel training averages them with a single
        # collective (train.py:103: DDP reduces every trainable parameter of the module, the encoder MLP included)
        sizes = (Cd * 3 * S, Cd, Cd * Cd, Cd)
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        dw1, db1, dw2, db2 = (t.view(shape) for t, shape in zip(flat.split(sizes), ((Cd, 3 * S), (Cd,), (Cd, Cd), (Cd,))))
        dfeat = torch.empty(B, V, Cd, h, w, device=dev) if ctx.want_feat else None
        w2 = mod.encoder[2].weight.detach().to(device=dev, dtype=torch.float32).contiguous()
        _lib.check(lib.parq_ray_pe_backward(_lib.ptr(cam), _lib.ptr(T_cp), _lib.ptr(T_wp), _lib.ptr(T_wl), _lib.ptr(w2),
                                            (C.c_float * 6)(*mod.ray_points_scale), mod.min_depth, mod.max_depth, S, B, V, h, w, Cd,
                                            _lib.ptr(g), _lib.ptr(hold.ws), _lib.ptr(bws), bws.numel() * 4, _lib.ptr(dw1), _lib.ptr(db1),
                                            _lib.ptr(dw2), _lib.ptr(db2), _lib.ptr(dfeat), _lib.stream_ptr()), "parq_ray_pe_backward")
        hold.consumed = True
        if mod.dp_all_reduce:
            from .parallel import all_reduce_mean_
            all_reduce_mean_(flat)
        return None, dfeat, None, None, None, None, dw1, db1, dw2, db2


class _WsHold:
    """The workspace one autograd node of AddRayPE reads in its backward, and the generation it was written in."""
    __slots__ = ("ws", "gen", "consumed", "__weakref__")

    def __init__(self, ws, gen):
        self.ws, self.gen, self.consumed = ws, gen, False


class AddRayPE(nn.Module):
    def __init__(self, dim_out: int, ray_points_scale=(-2, 2, -1.5, 0, 0.25, 4.25), num_samples: int = 64,
                 min_depth: float = 0.25, max_depth: float = 5.25):
        super().__init__()
        self.dim_out = dim_out
        self.ray_points_scale = [float(x) for x in ray_points_scale]
        self.num_samples = num_samples
        self.min_depth = float(min_depth)
        self.max_depth = float(max_depth)
        self.encoder = nn.Sequential(nn.Linear(3 * num_samples, dim_out), nn.ReLU(), nn.Linear(dim_out, dim_out))
        self._ws = None
        self._ws_key = None               # what the weight copies inside ``_ws`` were made from (see _run)
        self._ws_stream = None            # the launch stream ``_ws`` belongs to, and the parked (workspace, key) pairs of other streams:
        self._ws_parked = {}              # calls on different HIP streams (parq_amd.InFlight) each own their pose tables and weight copies
        self._gen = 0                     # forward counter, stamped on the workspace it wrote (``_parq_gen``)
        self._ws_owner = None             # weak reference to the _WsHold of the autograd node that owns ``_ws`` (if any)
        self.dp_all_reduce = False        # True: the backward all-reduces (mean) the encoder gradients over the default process group

    def _run(self, camera, T_cp, T_wp, T_wl, feat_hw, features, nchw=False, own_workspace=False):
        cam, T_cp, T_wp, T_wl = (raw(x) for x in (camera, T_cp, T_wp, T_wl))
        if not cam.is_cuda:
            raise RuntimeError("parq_amd.AddRayPE runs on the GPU only (there is no CPU fallback)")
        dev = cam.device
        prep = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
        cam, T_cp, T_wp, T_wl = prep(cam), prep(T_cp), prep(T_wp), prep(T_wl)
        self._last_geo = (cam, T_cp, T_wp, T_wl)
        B, V = cam.shape[:2]
        if feat_hw is None:
            wf, hf = cam[0, 0, :2].tolist()                 # the reference rounds the first camera's size too (:80-81)
            feat_hw = (int(round(hf)), int(round(wf)))
        h, w = feat_hw
        Cd = self.dim_out
        if features is not None:
            features = prep(features)
            assert features.shape == (B, V, Cd, h, w), tuple(features.shape)
        lib = _lib.load()
        self._gen += 1
        sid = int(torch.cuda.current_stream(dev).cuda_stream)
        if sid != self._ws_stream:        # another stream's call may still be reading this workspace: park it, take this stream's own
            if self._ws is not None:
                self._ws_parked[self._ws_stream] = (self._ws, self._ws_key)
            self._ws, self._ws_key = self._ws_parked.pop(sid, (None, None))
            self._ws_stream = sid
        # include/parq_hip.h: 1 = NCHW output, 2 = inference (the hidden layer is not kept for a backward: on the one-pass path it
        # never leaves the CU and the workspace shrinks by B*V*h*w*C floats)
        fused = Cd == 256 and self.num_samples == 64          # the library's one-pass path can write (B, V, C, h, w) directly
        nchw = bool(nchw and fused)
        flags = (1 if nchw else 0) | (0 if own_workspace else 2)
        nbytes = lib.parq_ray_pe_workspace_bytes_flags(B, V, h, w, Cd, self.num_samples, flags)
        owner = self._ws_owner() if self._ws_owner is not None else None
        if owner is not None and not owner.consumed and owner.ws is self._ws:
            self._ws = None                                  # an autograd node still needs the hidden layer saved there
        if self._ws is None or self._ws.numel() * 4 < nbytes or self._ws.device != dev:
            self._ws = torch.empty(nbytes // 4 + 1, dtype=torch.float32, device=dev)
            self._ws_key = None
        # the split / fragment-ordered weight copies inside the workspace are reused while the same workspace holds the same
        # weights (data pointer and version of the four parameters: optimizer steps and load_state_dict bump the version;
        # writes through .data need invalidate_weights()) in the same layout
        key = (self._ws.data_ptr(), flags & 2, B, V, h, w) + tuple((t.data_ptr(), t._version) for t in
                                                                     (self.encoder[0].weight, self.encoder[2].weight))
        if key == self._ws_key:
            flags |= 4
        self._ws_key = key
        out = torch.empty((B, V, Cd, h, w) if nchw else (B, V * h * w, Cd), dtype=torch.float32, device=dev)
        p = [prep(t.detach()) for t in (self.encoder[0].weight, self.encoder[0].bias, self.encoder[2].weight,
                                        self.encoder[2].bias)]
        _lib.check(lib.parq_ray_pe(_lib.ptr(cam), _lib.ptr(T_cp), _lib.ptr(T_wp), _lib.ptr(T_wl), _lib.ptr(p[0]),
                                   _lib.ptr(p[1]), _lib.ptr(p[2]), _lib.ptr(p[3]), (C.c_float * 6)(*self.ray_points_scale),
                                   self.min_depth, self.max_depth, self.num_samples, B, V, h, w, Cd, _lib.ptr(features),
                                   _lib.ptr(out), flags, _lib.ptr(self._ws), self._ws.numel() * 4, _lib.stream_ptr()),
                   "parq_ray_pe")
        self._ws._parq_gen = self._gen
        return out, (B, V, h, w), nchw

    def invalidate_weights(self):
        """After parameter writes that bypass autograd's version counter (``p.data.copy_()``, EMA swaps through ``.data``)."""
        self._ws_key = None

    def _needs_graph(self, features=None):
        return torch.is_grad_enabled() and (self.training or any(p.requires_grad for p in self.parameters())
                                            or bool(getattr(features, "requires_grad", False)))

    def forward(self, images_feat, camera=None, T_camera_pseudoCam=None, T_world_pseudoCam=None, T_world_local=None):
        """The encoding (B, T, C, H, W), as the reference returns it (images_feat only supplies the shape).  With gradients
        enabled and an encoder parameter that requires grad the result carries a graph, as the reference's does
        (model/ray_positional_encoding.py:128-136): the same autograd node as ``tokens`` on zero feature maps."""
        hw = tuple(images_feat.shape[-2:])
        if self._needs_graph():
            B, V = raw(camera).shape[:2]
            zeros = torch.zeros(B, V, self.dim_out, hw[0], hw[1], dtype=torch.float32, device=raw(camera).device)
            tok = self.tokens(zeros, camera, T_camera_pseudoCam, T_world_pseudoCam, T_world_local)
            return tok.view(B, V, hw[0], hw[1], self.dim_out).permute(0, 1, 4, 2, 3)
        with torch.no_grad():
            enc, (B, V, h, w), nchw = self._run(camera, T_camera_pseudoCam, T_world_pseudoCam, T_world_local, hw, None, nchw=True)
            return enc if nchw else enc.view(B, V, h, w, self.dim_out).permute(0, 1, 4, 2, 3)

    def tokens(self, images_feat, camera, T_camera_pseudoCam, T_world_pseudoCam, T_world_local):
        """features + encoding, tokenised channels-last (B, T*H*W, C) in one pass.  With gradients enabled and anything to
        differentiate (train mode, an encoder parameter or the feature maps requiring grad — eval mode included, like the
        reference) the call is an autograd node that owns the workspace holding the hidden layer for its backward."""
        if self._needs_graph(images_feat):
            e0, e2 = self.encoder[0], self.encoder[2]
            return _RayPeFn.apply(self, images_feat, camera, T_camera_pseudoCam, T_world_pseudoCam, T_world_local,
                                  e0.weight, e0.bias, e2.weight, e2.bias)
        with torch.no_grad():
            hw = tuple(images_feat.shape[-2:])
            out, _, _ = self._run(camera, T_camera_pseudoCam, T_world_pseudoCam, T_world_local, hw, images_feat)
            return out
