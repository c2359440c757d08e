"""Helpers for the -m gpu parity tests: build a parq_amd.PARQDecoder on cuda:0 from the
seeded synthetic weights and run it through the C ABI."""
import ctypes as C

import numpy as np
import torch

from parq_amd import _lib, synth


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def make_decoder(cfg, W):
    from parq_amd.decoder import PARQDecoder
    dec = PARQDecoder(cfg).eval()
    sd = dec.state_dict()
    for k in sd:
        src = k.replace("parq_module.decoder.mlp_heads.", "mlp_heads.")
        sd[k] = torch.from_numpy(W[src]).reshape(sd[k].shape)
    dec.load_state_dict(sd, strict=True)
    return dec.cuda()


def scene_args(sc):
    return (dev(sc["tokens"]), dev(sc["camera"]), dev(sc["T_camera_pseudoCam"]), dev(sc["T_world_pseudoCam"]),
            dev(sc["T_world_local"]))


def infer(dec, *args, **kw):
    """An inference call the way the reference's drivers make it (eval.py:46: under torch.no_grad()).  Without no_grad an
    eval-mode module whose parameters require grad builds a graph — like the reference — i.e. runs the training-forward kernels."""
    with torch.no_grad():
        return dec(*args, **kw)


def to_np(out):
    return {k: v.detach().cpu().numpy() for k, v in out.items()}


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float((np.abs(a - b) / np.maximum(1.0, np.abs(b))).max())


def lib():
    assert torch.cuda.is_available()
    return _lib.load()


def sptr():
    return _lib.stream_ptr()
