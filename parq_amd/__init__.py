"""parq_amd — MI355X-native implementation of PARQ's recurrent pixel-aligned decoder path.

Public surface mirrors the reference (ymingxie/PARQ):
    PARQDecoder  (model/parq_decoder.py:30)      forward on the HIP kernel chain
    AddRayPE     (model/ray_positional_encoding.py:29)  ray-point PE (+ fused tokenisation)
    Pose, Camera (utils/wrappers.py:194,441)     tensor wrappers drivers pass in
    InFlight     (no counterpart: the reference calls its model once per batch, eval.py:46)  several forwards of one module in flight
The compute lives in ``parq_amd/_C/libparq_hip.so`` (C ABI: include/parq_hip.h).
"""
from .wrappers import Camera, Obb3D, Pose, TensorWrapper  # noqa: F401


def __getattr__(name):
    # lazy: importing the package must not require the HIP library (CPU-only tooling, synth)
    if name == "PARQDecoder":
        from .decoder import PARQDecoder
        return PARQDecoder
    if name == "AddRayPE":
        from .ray_pe import AddRayPE
        return AddRayPE
    if name == "PARQ":
        from .module import PARQ
        return PARQ
    if name == "InFlight":
        from .inflight import InFlight
        return InFlight
    raise AttributeError(name)
