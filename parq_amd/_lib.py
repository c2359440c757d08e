"""ctypes binding of libparq_hip.so (include/parq_hip.h).

The product path has NO fallback: if the HIP library is missing or a symbol
cannot be resolved, importing callers get a RuntimeError that says how to build
it.  ``import torch`` must come first so that the process-wide HIP runtime
(torch's bundled libamdhip64.so.7) is the one our kernels register with.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (loads the HIP runtime before our library)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_C", "libparq_hip.so")
# development build (-DPARQ_DEV_PROBES: environment A/B switches, probe kernels, in-kernel time stamps).  Never loaded by the
# package itself: tools/ and `bench.py --dev-lib` opt in with use_dev_library() before the first load().
DEV_LIB_PATH = os.path.join(_HERE, "_C", "libparq_hip_dev.so")

PROF_KV_PROJ, PROF_PROJECT_SAMPLE, PROF_CROSS_ATTN, PROF_SELF_ATTN, PROF_LINEAR, PROF_OTHER, PROF_MERGE = range(7)
PROF_NAMES = ["kv_proj", "project_sample", "cross_attn", "self_attn", "linear", "other", "cross_attn_merge"]


class ParqConfig(C.Structure):
    _fields_ = [("dim", C.c_int32), ("num_queries", C.c_int32), ("num_classes", C.c_int32),
                ("num_heads", C.c_int32), ("ffn_dim", C.c_int32), ("num_layers", C.c_int32),
                ("share_weights", C.c_int32), ("num_mean_sizes", C.c_int32), ("scale", C.c_float * 6)]


class ParqScene(C.Structure):
    _fields_ = [("B", C.c_int32), ("V", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
                ("tokens", C.c_void_p), ("camera", C.c_void_p), ("T_camera_pseudoCam", C.c_void_p),
                ("T_world_pseudoCam", C.c_void_p), ("T_world_local", C.c_void_p)]


class ParqOutputs(C.Structure):
    _fields_ = [("pred_logits", C.c_void_p), ("center_unnormalized", C.c_void_p),
                ("size_unnormalized", C.c_void_p), ("ortho6d", C.c_void_p),
                ("sem_cls_prob", C.c_void_p), ("coord_pos", C.c_void_p)]


class ParqOutputGrads(C.Structure):
    _fields_ = [("pred_logits", C.c_void_p), ("center_unnormalized", C.c_void_p),
                ("size_unnormalized", C.c_void_p), ("ortho6d", C.c_void_p)]


# every symbol include/parq_hip.h declares: (restype, argtypes)
_vp, _i32, _i64, _sz, _f = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t, C.c_float
SYMBOLS = {
    "parq_last_error": (C.c_char_p, []),
    "parq_version": (C.c_char_p, []),
    "parq_create": (C.c_int, [C.POINTER(ParqConfig), C.POINTER(_vp)]),
    "parq_destroy": (C.c_int, [_vp]),
    "parq_set_weight": (C.c_int, [_vp, C.c_char_p, _vp, _i64]),
    "parq_packed_weights_bytes": (_sz, [_vp]),
    "parq_pack_weights": (C.c_int, [_vp, _vp, _sz, _vp]),
    "parq_workspace_bytes": (_sz, [_vp, _i32, _i32, _i32, _i32]),
    "parq_forward": (C.c_int, [_vp, C.POINTER(ParqScene), _vp, _sz, C.POINTER(ParqOutputs), _vp]),
    "parq_forward_capture": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _sz, _vp, C.POINTER(_vp)]),
    "parq_forward_replay": (C.c_int, [_vp, _vp, C.POINTER(ParqScene), _vp, _sz, C.POINTER(ParqOutputs), _vp]),
    "parq_graph_nodes": (_i64, [_vp]),
    "parq_graph_destroy": (C.c_int, [_vp]),
    "parq_prepare": (C.c_int, [_vp, C.POINTER(ParqScene), _vp, _sz, _vp]),
    "parq_iterate": (C.c_int, [_vp, C.POINTER(ParqScene), _vp, _sz, _i32, _vp, C.POINTER(ParqOutputs), _vp, _vp]),
    "parq_workspace_lookup": (C.c_int, [_vp, _i32, _i32, _i32, _i32, C.c_char_p, C.POINTER(_sz), C.POINTER(_sz)]),
    "parq_set_attention_mode": (C.c_int, [_vp, _i32]),
    "parq_set_head_tiers": (C.c_int, [_vp, C.c_uint32, _i32]),
    "parq_set_seam_fusion": (C.c_int, [_vp, _i32]),
    "parq_set_range_mirror": (C.c_int, [_vp, _vp]),
    "parq_mirror_take": (_i32, [_vp]),
    "parq_set_progress": (C.c_int, [_vp, _vp, _i32]),
    "parq_shard_exchange_floats": (_sz, [_vp, _i32, _i32]),
    "parq_iterate_sharded": (C.c_int, [_vp, C.POINTER(ParqScene), _vp, _sz, _i32, _i32, _vp, C.POINTER(ParqOutputs), _vp, _vp, _vp, _i32, _vp]),
    "parq_profile_enable": (C.c_int, [_vp, _i32]),
    "parq_profile_read": (C.c_int, [_vp, _i32, C.POINTER(C.c_double), C.POINTER(_i64)]),
    "parq_set_dropout": (C.c_int, [_vp, _f, C.c_uint32]),
    "parq_set_backward_batched": (C.c_int, [_vp, _i32]),
    "parq_k_dropout_mask": (C.c_int, [_vp, _i32, _i32, _i64, _i64, _vp, _vp]),
    "parq_train_workspace_bytes": (_sz, [_vp, _i32, _i32, _i32, _i32]),
    "parq_grad_arena_bytes": (_sz, [_vp]),
    "parq_forward_train": (C.c_int, [_vp, C.POINTER(ParqScene), _vp, _sz, C.POINTER(ParqOutputs), _vp]),
    "parq_wait_iteration": (C.c_int, [_vp, C.c_int32]),
    "parq_backward": (C.c_int, [_vp, C.POINTER(ParqScene), _vp, _sz, C.POINTER(ParqOutputs), C.POINTER(ParqOutputGrads), _vp, _vp, _vp]),
    "parq_arena_lookup": (C.c_int, [_vp, C.c_char_p, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "parq_grad_bucket": (C.c_int, [_vp, _i32, C.POINTER(_i64), C.POINTER(_i64)]),
    "parq_backward_wait_bucket": (C.c_int, [_vp, _i32, _vp]),
    "parq_set_backward_streams": (C.c_int, [_vp, _i32]),
    "parq_ray_pe_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "parq_ray_pe_workspace_bytes_flags": (_sz, [_i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "parq_ray_pe": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_f), _f, _f, _i32, _i32, _i32, _i32, _i32, _i32,
                              _vp, _vp, _i32, _vp, _sz, _vp]),
    "parq_ray_pe_backward_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "parq_ray_pe_backward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.POINTER(_f), _f, _f, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp,
                                       _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    "parq_parse_pred": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, C.POINTER(_f), _i32, _i32, _vp, _vp, _vp]),
    "parq_set_loss": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp,
                                C.POINTER(_f), _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "parq_k_project_sample": (C.c_int, [_vp, _vp, _vp, _vp, C.POINTER(_f), _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "parq_k_camera_local": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "parq_k_linear": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "parq_k_linear_half": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "parq_k_attention_scratch_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32]),
    "parq_k_attention": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "parq_k_attention_split_scratch_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "parq_k_attention_split": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "parq_k_attention_split8": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "parq_k_attention_split256_scratch_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "parq_k_attention_split256": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "parq_k_attention_half_scratch_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "parq_k_attention_half": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "parq_k_layernorm": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _f, _vp]),
}

_lib = None


def use_dev_library():
    """Development only (tools/, `bench.py --dev-lib`): bind the -DPARQ_DEV_PROBES build instead of the product library."""
    global LIB_PATH
    if _lib is not None:
        raise RuntimeError("parq_amd: use_dev_library() must be called before the library is first loaded")
    LIB_PATH = DEV_LIB_PATH


def is_dev_library():
    return LIB_PATH == DEV_LIB_PATH


def load():
    """Load libparq_hip.so (once) and type every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "parq_amd: HIP extension %s is missing. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc, --offload-arch=gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError("parq_amd: %s does not export %s (stale build?)" % (LIB_PATH, name)) from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().parq_last_error().decode("utf-8", "replace")
        raise RuntimeError("parq_amd: %s failed (code %d): %s" % (what, rc, msg))


def ptr(t):
    """Device pointer of a CUDA float32 contiguous tensor (or None)."""
    if t is None:
        return None
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), \
        "expected a contiguous float32 CUDA tensor, got %s %s contiguous=%s" % (t.device, t.dtype, t.is_contiguous())
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
