"""Pose / Camera tensor wrappers with the reference's calling convention.

Drivers written for the reference hand ``PARQDecoder.forward`` objects that expose
``._data`` (a (...,12) pose or (...,6) camera tensor), indexing, ``.cuda()`` and a few
geometric helpers (utils/wrappers.py:114-293, 441-553).  These classes provide the
same surface so such drivers run unchanged; the decoder itself only reads ``._data``
and does all geometry on the device inside the HIP kernels.

Layout: Pose = [R row-major (9) | t (3)];  Camera = [w, h, fx, fy, cx, cy].
"""
from __future__ import annotations

import numpy as np
import torch


def _as_tensor(x):
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x)
    return x


class TensorWrapper:
    """A thin, indexable view over one tensor whose last axis is the payload."""
    _width = None

    def __init__(self, data):
        data = _as_tensor(data)
        if self._width is not None and data.shape[-1] != self._width:
            raise AssertionError("%s expects last dim %d, got %s" % (type(self).__name__, self._width, tuple(data.shape)))
        self._data = data

    # ---- metadata
    shape = property(lambda self: self._data.shape[:-1])
    device = property(lambda self: self._data.device)
    dtype = property(lambda self: self._data.dtype)
    ndim = property(lambda self: self._data.ndim)

    # ---- tensor-like plumbing (every op re-wraps)
    def _wrap(self, t):
        return type(self)(t)

    def __getitem__(self, idx):
        return self._wrap(self._data[idx])

    def __setitem__(self, idx, item):
        self._data[idx] = item._data if isinstance(item, TensorWrapper) else item

    def __len__(self):
        return self._data.shape[0]

    def to(self, *a, **k):
        return self._wrap(self._data.to(*a, **k))

    def cpu(self):
        return self._wrap(self._data.cpu())

    def cuda(self, *a, **k):
        return self._wrap(self._data.cuda(*a, **k))

    def float(self):
        return self._wrap(self._data.float())

    def double(self):
        return self._wrap(self._data.double())

    def detach(self):
        return self._wrap(self._data.detach())

    def clone(self):
        return self._wrap(self._data.clone())

    def pin_memory(self):
        return self._wrap(self._data.pin_memory())

    def squeeze(self, dim=None):
        if dim is None:
            lead = [s for s in self._data.shape[:-1] if s != 1]
            return self._wrap(self._data.reshape(lead + [self._data.shape[-1]]))
        assert dim not in (-1, self._data.dim() - 1)
        return self._wrap(self._data.squeeze(dim))

    def unsqueeze(self, dim):
        assert dim not in (-1, self._data.dim())
        return self._wrap(self._data.unsqueeze(dim))

    def view(self, *shape):
        assert shape[-1] in (-1, self._data.shape[-1])
        return self._wrap(self._data.view(*shape))

    @classmethod
    def stack(cls, objs, dim=0):
        return cls(torch.stack([o._data for o in objs], dim=dim))

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        if func is torch.stack:
            return cls.stack(*args, **(kwargs or {}))
        return NotImplemented

    def __repr__(self):
        return "%s %s %s %s" % (type(self).__name__, tuple(self.shape), self.dtype, self.device)


class Pose(TensorWrapper):
    """SE(3) as a 12-vector (utils/wrappers.py:194-293)."""
    _width = 12

    @classmethod
    def from_Rt(cls, R, t):
        R, t = _as_tensor(R), _as_tensor(t)
        assert R.shape[-2:] == (3, 3) and t.shape[-1] == 3
        return cls(torch.cat([R.reshape(R.shape[:-2] + (9,)), t], dim=-1))

    @classmethod
    def from_4x4mat(cls, T):
        T = _as_tensor(T)
        return cls.from_Rt(T[..., :3, :3], T[..., :3, 3])

    @property
    def R(self):
        return self._data[..., :9].reshape(self._data.shape[:-1] + (3, 3))

    @property
    def t(self):
        return self._data[..., 9:]

    @property
    def matrix(self):
        top = torch.cat([self.R, self.t.unsqueeze(-1)], dim=-1)
        bot = top.new_zeros(top.shape[:-2] + (1, 4))
        bot[..., 0, 3] = 1
        return torch.cat([top, bot], dim=-2)

    def inverse(self):
        Rt = self.R.transpose(-1, -2)
        return Pose.from_Rt(Rt, -(Rt @ self.t.unsqueeze(-1)).squeeze(-1))

    def compose(self, other):
        """self ∘ other: apply ``other`` first."""
        return Pose.from_Rt(self.R @ other.R, self.t + (self.R @ other.t.unsqueeze(-1)).squeeze(-1))

    def transform(self, p3d):
        p3d = _as_tensor(p3d)
        return p3d @ self.R.transpose(-1, -2) + self.t.unsqueeze(-2)

    __matmul__ = compose
    __mul__ = transform


class Camera(TensorWrapper):
    """Pinhole camera as [w, h, fx, fy, cx, cy] (utils/wrappers.py:441-553)."""
    _width = 6
    eps = 1e-3

    size = property(lambda self: self._data[..., 0:2])
    f = property(lambda self: self._data[..., 2:4])
    c = property(lambda self: self._data[..., 4:6])

    def scale(self, scales):
        """Intrinsics after resizing the image by ``scales`` (pixel-centre convention:
        c' = (c + 0.5) s - 0.5)."""
        if isinstance(scales, (int, float)):
            scales = (scales, scales)
        s = self._data.new_tensor(scales)
        return Camera(torch.cat([self.size * s, self.f * s, (self.c + 0.5) * s - 0.5], dim=-1))

    def in_image(self, p2d):
        p2d = _as_tensor(p2d)
        hi = self.size.unsqueeze(-2) - 1
        return torch.all((p2d >= 0) & (p2d <= hi), dim=-1)

    def project(self, p3d):
        p3d = _as_tensor(p3d)
        z = p3d[..., 2]
        front = z > self.eps
        uv = p3d[..., :2] / z.clamp(min=self.eps).unsqueeze(-1)
        uv = uv * self.f.unsqueeze(-2) + self.c.unsqueeze(-2)
        return uv, front & self.in_image(uv)

    def unproject(self, uv):
        uv = _as_tensor(uv)
        B = uv.shape[0]
        xy = (uv - self.c.reshape(B, 1, 2)) / self.f.reshape(B, 1, 2)
        return torch.cat([xy, xy.new_ones(xy.shape[:-1] + (1,))], dim=-1)


class Obb3D(TensorWrapper):
    """Oriented 3-D boxes as 19-vectors [xmin,xmax,ymin,ymax,zmin,zmax | T_world_object (12) | sem_id]
    (utils/wrappers.py:297-436); a row of all -1 is padding."""
    _width = 19

    @classmethod
    def separate_init(cls, bb3_object, T_world_object, sem_id):
        bb3_object, T_world_object, sem_id = _as_tensor(bb3_object), raw(T_world_object), _as_tensor(sem_id)
        if sem_id.dim() != bb3_object.dim():
            sem_id = sem_id.unsqueeze(-1)
        return cls(torch.cat([bb3_object, T_world_object, sem_id.to(bb3_object.dtype)], dim=-1))

    bb3_object = property(lambda self: self._data[..., :6])
    bb3_min_object = property(lambda self: self._data[..., 0:6:2])
    bb3_max_object = property(lambda self: self._data[..., 1:6:2])
    bb3_center_object = property(lambda self: 0.5 * (self._data[..., 0:6:2] + self._data[..., 1:6:2]))
    bb3_size = property(lambda self: self._data[..., 1:6:2] - self._data[..., 0:6:2])
    T_world_object = property(lambda self: Pose(self._data[..., 6:18]))
    sem_id = property(lambda self: self._data[..., 18].unsqueeze(-1))

    @property
    def bb3corners_object(self):
        """The 8 corners (..., 8, 3): bottom face (z = zmin) counter-clockwise from (xmin, ymin), then the top face."""
        lo, hi = self.bb3_min_object, self.bb3_max_object
        pick = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
        pts = [torch.stack([(hi if sx else lo)[..., 0], (hi if sy else lo)[..., 1], (hi if sz else lo)[..., 2]], dim=-1)
               for sx, sy, sz in pick]
        return torch.stack(pts, dim=-2)

    def add_padding(self, max_box=100):
        assert self._data.ndim <= 2
        n = self._data.shape[0]
        if n >= max_box:
            return type(self)(self._data[:max_box])
        pad = -self._data.new_ones(max_box - n, self._data.shape[-1])
        return type(self)(torch.cat([self._data, pad], dim=0))

    def remove_padding(self):
        """Boxes before the first all -1 row count (the reference counts non-padding rows and takes that many leading rows)."""
        assert self._data.ndim <= 3
        if self._data.ndim == 1:
            return self
        keep = ~torch.all(self._data == -1, dim=-1)
        if self._data.ndim == 2:
            return type(self)(self._data[:int(keep.sum())])
        return [type(self)(self._data[b][:int(keep[b].sum())]) for b in range(self._data.shape[0])]


def raw(x):
    """The underlying tensor of a wrapper (ours or a duck-typed reference one) or a tensor."""
    return x._data if hasattr(x, "_data") else _as_tensor(x)
