"""PARQDecoder: host-side mirror of the reference module, running on the HIP kernel chain.

Same constructor argument (``cfg.MODEL.DECODER``), same ``forward`` signature and return
value, same ``state_dict`` keys as the reference class (model/parq_decoder.py:30-163,
SURVEY.md §5.4), so a reference checkpoint loads with ``strict=True`` and reference-style
drivers (eval.py:26-47) can swap the import.  All arithmetic of the forward pass runs in
libparq_hip.so (include/parq_hip.h); this file only owns parameters, buffers and the
packing of arguments.  There is no PyTorch/CPU fallback for the compute.

Scope: the inference forward (rows a-e) and the "next" rows: in ``train()`` mode under autograd the
forward is one autograd node whose backward is the HIP backward chain (``parq_backward``; the decoder layer's dropout,
DROPOUT_RATE, is applied with counter-based masks), ``loss`` mirrors the reference's set loss, ``parse_pred`` runs on the
device and ``update_metrics`` / ``compute_metrics`` / ``reset_metrics`` drive the scene-level F1 trackers (§8f-4).

Weights are packed into one device arena the first time they are used and again whenever a parameter changes.  Changes
are detected from ``(data_ptr, _version)`` of every parameter, which optimizers and ``load_state_dict`` bump; writes that
bypass autograd's version counter (``p.data.copy_()``, ``p.data.mul_()``, EMA / SWA swaps through ``.data``) are invisible
to it: call ``invalidate_weights()`` after such an update.
"""
from __future__ import annotations

import ctypes as C
import math
import weakref

import numpy as np
import torch
from torch import nn

from . import _lib
from .box_processor import mean_size_table
from .wrappers import raw

def _raw_stream(device):
    """hipStream_t of the current stream of `device` as an int (the fast path of torch.cuda.current_stream(device).cuda_stream)."""
    idx = device.index
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device() if idx is None else idx)


OUTPUT_KEYS = ("pred_logits", "center_unnormalized", "size_unnormalized", "ortho6d", "sem_cls_prob", "coord_pos")


# ---------------------------------------------------------------------------------------
# parameter containers that reproduce the reference's state_dict layout
# ---------------------------------------------------------------------------------------

# Every parameter of the decoder lives in one of the small container classes below.  They bump this counter whenever a Parameter or
# sub-module is (re-)assigned or deleted, which lets PARQDecoder._unique_params() reuse its walk of the module tree until then (the walk
# costs ~0.15 ms per call — under ``range_check = "sync"`` that is device idle time between two forwards).
_PARAM_EPOCH = [0]


class _Tracked:
    def __setattr__(self, name, value):
        if isinstance(value, (nn.Parameter, nn.Module)) or name in self.__dict__.get("_parameters", ()) or name in self.__dict__.get("_modules", ()):
            _PARAM_EPOCH[0] += 1
        super().__setattr__(name, value)

    def __delattr__(self, name):
        _PARAM_EPOCH[0] += 1
        super().__delattr__(name)

    def register_parameter(self, name, param):
        _PARAM_EPOCH[0] += 1
        super().register_parameter(name, param)

    def add_module(self, name, module):
        _PARAM_EPOCH[0] += 1
        super().add_module(name, module)


class _TDict(_Tracked, nn.ModuleDict):
    pass


class _TList(_Tracked, nn.ModuleList):
    def __setitem__(self, idx, module):
        _PARAM_EPOCH[0] += 1
        super().__setitem__(idx, module)

    def __delitem__(self, idx):
        _PARAM_EPOCH[0] += 1
        super().__delitem__(idx)


class _RefPoints(_Tracked, nn.Embedding):
    """``refpoint`` (model/parq_decoder.py:62: nn.Embedding(num_queries, 3)); only its weight is read."""


class _WB(_Tracked, nn.Module):
    """A leaf holding ``weight`` (and optionally ``bias``)."""

    def __init__(self, wshape, bias=True, bshape=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*wshape))
        if bias:
            self.bias = nn.Parameter(torch.empty(*(bshape or (wshape[0],))))
        else:
            self.register_parameter("bias", None)


def _conv_init(m: _WB):
    """torch Conv1d/Linear default init (kaiming_uniform(a=sqrt(5)))."""
    fan_in = m.weight.shape[1]
    bound = 1.0 / math.sqrt(fan_in)
    nn.init.uniform_(m.weight, -bound, bound)
    if m.bias is not None:
        nn.init.uniform_(m.bias, -bound, bound)


def _norm_init(m: _WB):
    nn.init.ones_(m.weight)
    nn.init.zeros_(m.bias)


class _Head(_Tracked, nn.Module):
    """GenericMLP parameter layout (model/generic_mlp.py:64-132): ``layers.<idx>``.

    hidden=[]     -> layers.0 = Conv1d(C, out)
    hidden=[C,C]  -> layers.0 Conv(no bias), .1 GroupNorm, (.2 ReLU, .3 Dropout),
                     .4 Conv(no bias), .5 GroupNorm, (.6, .7), .8 Conv(bias)
    """

    def __init__(self, C_in, out, hidden):
        super().__init__()
        layers = _TDict()
        if hidden:
            layers["0"] = _WB((C_in, C_in, 1), bias=False)
            layers["1"] = _WB((C_in,), bshape=(C_in,))
            layers["4"] = _WB((C_in, C_in, 1), bias=False)
            layers["5"] = _WB((C_in,), bshape=(C_in,))
            layers["8"] = _WB((out, C_in, 1))
            for k in ("0", "4", "8"):
                _conv_init(layers[k])
            for k in ("1", "5"):
                _norm_init(layers[k])
        else:
            layers["0"] = _WB((out, C_in, 1))
            _conv_init(layers["0"])
        self.layers = layers


class _MHA(_Tracked, nn.Module):
    """nn.MultiheadAttention parameter layout."""

    def __init__(self, C_):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * C_, C_))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * C_))
        self.out_proj = _WB((C_, C_))
        nn.init.zeros_(self.out_proj.bias)


class _Layer(_Tracked, nn.Module):
    def __init__(self, C_, F_):
        super().__init__()
        self.self_attn = _MHA(C_)
        self.multihead_attn = _MHA(C_)
        self.linear1 = _WB((F_, C_))
        self.linear2 = _WB((C_, F_))
        self.norm1, self.norm2, self.norm3 = (_WB((C_,), bshape=(C_,)) for _ in range(3))
        for m in (self.linear1, self.linear2):
            _conv_init(m)
        for m in (self.norm1, self.norm2, self.norm3):
            _norm_init(m)


class _DecoderParams(_Tracked, nn.Module):
    def __init__(self, C_, F_, n_layers):
        super().__init__()
        self.layers = _TList([_Layer(C_, F_) for _ in range(n_layers)])
        self.norm = _WB((C_,), bshape=(C_,))          # in checkpoints, never applied (transformer_parq.py:174)
        _norm_init(self.norm)
        pe = _TDict()
        pe["0"] = _WB((C_, 384))
        pe["2"] = _WB((C_, C_))
        _conv_init(pe["0"])
        _conv_init(pe["2"])
        self.position_encoder = pe
        self.mlp_heads = None                          # shared with PARQDecoder.mlp_heads (parq_decoder.py:66)


class _TransformerParams(_Tracked, nn.Module):
    def __init__(self, C_, F_, n_layers):
        super().__init__()
        self.decoder = _DecoderParams(C_, F_, n_layers)
        for p in self.parameters():                    # transformer_parq.py:90-93
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)


# ---------------------------------------------------------------------------------------

# cross-attention arithmetic (include/parq_hip.h, parq_set_attention_mode): "split" = fp16 hi/lo 3-term products
# (fp32-class accuracy, default at head dims 64 and 256), "fp32" = exact fp32 MFMA, "fp16" / "bf16" = single reduced-precision
# products (BASELINE configs 2 and 5; head dim 64 with dim 128 / 256, head dim 256 with dim a multiple of 128)
ATTENTION_MODES = {"fp32": 0, "split": 1, "fp16": 2, "bf16": 3, "split8": 4}


class _Stash:
    """What one training forward leaves for its backward: the training workspace (saved activations of every iteration, K / V,
    log-sum-exp rows), the call's state tuple (scene, inputs, output tensors, attention mode, dropout probability and seed) and
    the workspace generation it was written in.  Owned by the autograd node; the module only keeps a weak reference, so a later
    training forward knows whether the workspace is still needed (then it takes a fresh one) or free to be reused."""
    __slots__ = ("ws", "state", "gen", "consumed", "__weakref__")

    def __init__(self, ws, state, gen):
        self.ws, self.state, self.gen, self.consumed = ws, state, gen, False


class _WsEntry:
    """One inference workspace of a module: the K/V cache + activations tensor of a (shape, device, stream), the pinned mirror word the
    device raises for forwards that run in it, and the captured iterations of its forward (a HIP graph, include/parq_hip.h
    parq_forward_capture) with the settings they were recorded under."""
    __slots__ = ("ws", "slot", "stream", "graphs", "last_key", "replays", "epoch")

    def __init__(self, ws, slot, stream):
        self.ws, self.slot, self.stream = ws, slot, stream
        self.epoch = 0                # epoch of the last forward enqueued here (the device stores it into the slot's progress word)
        self.graphs = {}              # key -> parq_graph_t (one per (weights, attention settings): the current one; stale ones are retired)
        self.last_key = None          # key of the previous forward in this workspace (a graph is captured when a key repeats)
        self.replays = 0


class _WsCache(dict):
    """The module's inference workspaces, least recently used first.  Dropping an entry retires its graphs: they are destroyed once
    an event recorded behind their last launch has completed (``PARQDecoder._purge_graphs``)."""

    def __init__(self, owner):
        super().__init__()
        self._owner = weakref.ref(owner)

    def _retire(self, entry):
        dec = self._owner()
        if dec is not None and entry is not None and entry.graphs:
            dec._retire_graphs(entry)

    def pop(self, key, *default):
        had = key in self
        entry = super().pop(key, *default)
        if had:
            self._retire(entry)
        return entry

    def take(self, key):
        """Remove and return WITHOUT retiring (the entry is re-inserted as the most recently used)."""
        return super().pop(key, None)

    def clear(self):
        for entry in list(self.values()):
            self._retire(entry)
        super().clear()


class _TrainFn(torch.autograd.Function):
    """Autograd node of the whole decoder: forward = parq_forward_train, backward = parq_backward (HIP kernels)."""

    @staticmethod
    def forward(ctx, dec, tokens, camera, T_cp, T_wp, T_wl, feat_hw, *params):
        outs = dec.forward_train(tokens, camera, T_cp, T_wp, T_wl, feat_hw=feat_hw, defer_range_check=bool(dec.overlap_loss_matching) and dec.num_layers <= 16)
        stacked = dec._train_state[2]                       # six (I, B, Q, k) tensors
        ctx.dec = dec
        # this node OWNS the stash of its forward: a later forward of the module, while this node is alive and has not run its
        # backward, allocates a workspace of its own instead of overwriting this one (several outstanding forwards per module,
        # e.g. (loss(dec(a)) + loss(dec(b))).backward(), work like they do with the reference)
        ctx.stash = _Stash(dec._train_ws, dec._train_state, dec._train_gen)
        dec._stash_owner = weakref.ref(ctx.stash)
        ctx.want_tokens = bool(tokens.requires_grad)
        ctx.mark_non_differentiable(stacked[4], stacked[5])  # sem_cls_prob / coord_pos carry no gradient (transformer_parq.py:261-265)
        del outs
        return tuple(stacked)

    @staticmethod
    def backward(ctx, g_logits, g_center, g_size, g_rot, _g_prob, _g_coord):
        dec = ctx.dec
        if dec._phase_hook is not None:
            dec._phase_hook("hip_backward")                  # bench.py --phase-times: where the loss graph's autograd ends
        st = ctx.stash
        if getattr(st.ws, "_parq_gen", None) != st.gen:
            raise RuntimeError("parq_amd.PARQDecoder: second backward through a training forward (retain_graph=True) whose saved "
                               "activations were released by the first backward and have since been reused by a later forward of "
                               "the same module.  Run the repeated backward before the next forward.")
        grads, d_tokens = dec.backward({"pred_logits": g_logits, "center_unnormalized": g_center, "size_unnormalized": g_size,
                                        "ortho6d": g_rot}, want_token_grad=ctx.want_tokens, _stash=st)
        st.consumed = True                                   # the workspace may be reused by the next training forward
        per_param = []
        for name, p in dec._unique_params():
            per_param.append(grads.get(name) if p.requires_grad else None)
        return (None, d_tokens, None, None, None, None, None, *per_param)


class PARQDecoder(_Tracked, nn.Module):
    """Drop-in for ``model.parq_decoder.PARQDecoder`` (forward path)."""

    def __init__(self, cfg):
        super().__init__()
        T = cfg.TRANSFORMER
        assert T.QUERIES_DIM == T.DEC_DIM, "queries dim needs to equal DEC_DIM (transformer_parq.py:76-78)"
        assert cfg.DIM_IN == T.DEC_DIM
        if not cfg.SHARE_MLP_HEADS:
            raise NotImplementedError("SHARE_MLP_HEADS=False is broken in the reference (parq_decoder.py:119-123)")
        self.dim_in = cfg.DIM_IN
        self.num_queries = cfg.NUM_QUERIES
        self.num_semcls = cfg.NUM_SEMCLS
        self.loss_weight = cfg.LOSS_WEIGHT
        self.for_vis = cfg.FOR_VIS
        self.track_scale = cfg.TRACK_SCALE
        self.share_mlp_heads = cfg.SHARE_MLP_HEADS
        self.enable_nms = getattr(cfg, "ENABLE_NMS", True)
        # evaluation trackers, one per EVAL_TYPE entry (model/parq_decoder.py:73-80; only "f1" exists in the reference)
        from .f1_eval import F1Calculator
        eval_type = getattr(cfg, "EVAL_TYPE", "f1")
        self.metrics_calculator = [F1Calculator(getattr(cfg, "CONF_THRESH", 0.1))
                                   for et in (eval_type if isinstance(eval_type, (list, tuple)) else [eval_type]) if et == "f1"]
        self.num_heads = T.DEC_HEADS
        self.num_layers = T.DEC_LAYERS
        self.ffn_dim = T.DEC_FFN_DIM
        self.share_weights = bool(T.SHARE_WEIGHTS)
        self.dropout_rate = float(T.DROPOUT_RATE)
        self.scale = [float(x) for x in T.SCALE]
        self.mean_size_path = getattr(cfg, "MEAN_SIZE_PATH", None)

        Cd, ncls = self.dim_in, self.num_semcls + 1
        self._out_width = 2 * ncls + 3 + 3 + 6 + 3      # floats per (iteration, scene, query) over the six output tensors
        self.mlp_heads = _TDict([
            ("sem_cls_head", _Head(Cd, ncls, hidden=False)),
            ("center_head", _Head(Cd, 3, hidden=True)),
            ("size_head", _Head(Cd, 3, hidden=False)),
            ("rotation_head", _Head(Cd, 6, hidden=True)),
        ])
        self.parq_module = _TransformerParams(Cd, self.ffn_dim, 1 if self.share_weights else self.num_layers)
        self.parq_module.decoder.mlp_heads = self.mlp_heads
        self.refpoint = _RefPoints(self.num_queries, 3)

        # cross-attention arithmetic: "split" = fp16 hi/lo 3-term products on the fp16 matrix pipe
        # (fp32-class accuracy; head dim 64, and head dim 256 = the reference's shipped DEC_DIM 1024 / 4 heads),
        # "fp32" = fp32 MFMA.  include/parq_hip.h
        # "split8" = the same with the two cross terms of every product as MX-scaled fp8 products (inference at head dim 64, dim 256,
        # key counts that are multiples of 64 — every other case of this mode runs as "split"): 1e-6 from float64 at the outputs on
        # the reference's fixtures against 2e-6 for "split", same 1e-4 bound against the reference's vectors; the default where it applies
        dh_ = Cd // self.num_heads
        self.attention_mode = "split8" if (dh_ == 64 and Cd == 256) else "split" if dh_ in (64, 256) else "fp32"
        self._mean_sizes = mean_size_table(self.mean_size_path)     # (rows,3) float64
        self._h = None            # parq_handle
        self._arena = None
        self._arena_key = None
        self._ws = _WsCache(self)
        self._graveyard = []              # (graphs, event): captured forwards of dropped workspaces, destroyed once the event has completed
        self._arena_gen = 0               # bumped by every re-pack of the weight arena (part of a captured forward's key)
        self._arena_event = None          # recorded behind the first forward after a (re-)pack: other streams wait for it once
        self._arena_pack_stream = None
        self._arena_streams = {}          # stream id -> stream object of every stream that is ordered behind the current arena
        self._defer = None                # InFlight.submit(): list that receives the settle callable of a forward instead of a host wait
        self._epoch = 0                   # counter of inference forwards (parq_set_progress)
        self._profiling = False
        # Captured forward (include/parq_hip.h parq_forward_capture): the iterations of the second inference forward of a (shape,
        # stream, weights, attention settings) are recorded into a HIP graph and later forwards replay it behind their directly
        # launched prologue and K/V projection — 3 host calls instead of ~90 launches; the graph holds no pointer of a particular
        # call (any tokens / cameras / outputs replay it) and its results are bit-identical to the uncaptured path.  False = always
        # enqueue launch by launch.
        self.use_graph = True
        self._matcher = None
        self._train_ws = None
        self._train_state = None
        self._train_gen = 0               # bumped by every forward_train and stamped on its workspace (``_parq_gen``)
        self._stash_owner = None          # weak reference to the _Stash of the autograd node that owns ``_train_ws`` (if any)
        self._train_entry_event = None    # recorded on the caller's stream at the entry of forward_train (loss(): targets ready)
        self.loss_batched = True          # loss(): all (iteration, scene) pairs in ~40 launches (False: the reference's per-pair loop)
        self.dp_all_reduce = False        # True: backward() all-reduces the flat gradient arena over the default process group
        self.dp_bucketed = True           # ... in two buckets, the first overlapped with the cross-attention backward (False: one flat all-reduce)
        self._dp_stream = None
        self.backward_batched = True      # cross-attention backward of all iterations as one launch (False: per-iteration launches,
                                          # the cross-check form; include/parq_hip.h parq_set_backward_batched)
        self.overlap_loss_matching = True # loss() on the outputs of this module's own training forward matches iteration k on the
                                          # host while the device runs iterations k+1.. (parq_wait_iteration); the fp16-range check
                                          # of that forward is then resolved inside loss() (or raised by backward()) instead of
                                          # by a host synchronisation at the end of forward_train
        self._phase_hook = None           # optional callable(name), called at the entry of the HIP backward (bench.py --phase-times)
        self.loss_targets_late = False    # True: the targets handed to loss() may have been produced on the caller's stream AFTER the
                                          # training forward was enqueued (a late .to(device, non_blocking=True), on-device box
                                          # augmentation between dec(...) and dec.loss(...)): the loss's side stream then waits for the
                                          # whole main stream before it reads them (correct, no overlap with the forward's iterations)
        self._train_pending = None        # deferred range check of the last training forward: callable -> True if it re-ran
        self.max_workspaces = 2           # inference workspaces (each holds a K/V cache) kept alive, least recently used first out
        self._mean_dev = None
        # fp16-operand attention modes ("split", "split8", "fp16"): what to do when a token / K / V element leaves the fp16 range
        # (include/parq_hip.h: the device then writes NaN outputs instead of wrong numbers, and raises a flag), and when attention mode
        # "split8" meets rows that rest on too few keys for its error model (the merge kernel flags them per head; the iteration that
        # met the row and everything after it is written as NaN, the flagged heads are named in the pinned mirror word).
        #   "sync" (default): the reference never returns NaN on valid input (model/transformer_parq.py:283-337, eval.py:45-48), and
        #           neither does this module: every inference forward waits for its stream once, reads the pinned word of its
        #           workspace (a host load, no device copy) and, if it is raised, re-runs the forward — with the exact fp32 kernels
        #           after a range violation, with the flagged HEADS moved to the fp16 x 3 tier (``safe_heads``) after a too-peaked row
        #           — before it returns.  The caller always gets numbers.  Cost: the launch latency of the next forward is no longer
        #           hidden behind the previous one (bench.py ``guard_policy_cost``); with ``InFlight`` the wait moves to
        #           ``Ticket.result()`` and other forwards keep the device busy meanwhile.
        #   "lazy": no synchronisation on the forward path (servers that keep forwards in flight and check tickets themselves).  On a
        #           violation the NEXT call into the module that finds the pinned word set warns and moves the flagged heads / switches
        #           to "fp32" for good; forwards already in flight that met such inputs return NaN (never silently wrong numbers).  A
        #           module's FIRST inference forward per weight version is still checked synchronously and re-run.
        #   "off":  no poisoning, no tier change; ``fp16_range_exceeded()`` / ``attention_too_peaked()`` / ``attention_peaked_map()``
        #           on request.
        # Heads moved to the fp16 x 3 tier stay there (``reset_attention_tiers()``; "sync" only, opt-in: ``tier_return_after``).
        self.range_check = "sync"
        self._range_mirror = None         # pinned host int32 words the device raises on a range violation (see _set_mirror)
        self._mirror_np = None
        self._mirror_set = None
        # Training in mode "split8" (forward flash_split8_kernel with dropout, backward from its stage cache): 3 % faster per step at
        # BASELINE cfg 4, and the forward's ~1e-5 arithmetic noise (ten times mode "split"'s) reaches the gradients amplified by the
        # free-running chain (profiles/NOTES_r04.md).  Off by default: training steps then run in mode "split".
        self.train_split8 = False
        self._peaky_checked = False       # the first inference forward in mode "split8" (per weight version) has been checked for too-peaked rows
        # Per-head tiers of attention mode "split8" (include/parq_hip.h parq_set_head_tiers): bit h = head h runs the fp16 x 3
        # arithmetic of mode "split" inside a "split8" forward.  Heads are moved there by the peakedness guard (never back by
        # themselves: ``reset_attention_tiers()``); a module whose heads are all safe runs exactly mode "split".
        self.safe_heads = 0
        self._tiers_set = None            # (safe mask, poison) the native handle currently holds
        # In-launch hand-offs of the small-op chain (include/parq_hip.h parq_set_seam_fusion): the self out-projection and the query
        # projection behind norm1 as ONE launch whose query tiles wait (bounded) for row sums the other tiles publish.  OFF by default
        # since round 6: it bought +1.1 % when every launch was enqueued from the host; with the iterations replayed from a captured
        # graph two 4 x 300-step A/Bs on two boxes put it at -0.3 % and +0.6 % (profiles/r06_ab_seam_q_under_graph_replay.txt) — below what
        # a spin-wait between workgroups of one launch has to earn (HIP does not promise their dispatch order).  True switches it on;
        # after a timeout (never observed) the outputs of that forward are NaN (re-run under the default policy) and the module goes
        # back to one launch per dependent stage for good.
        self.fuse_seams = False
        self._seams_set = None
        # range_check = "sync" only (there a wrong guess costs a re-run, never a NaN forward): a head on the fp16 x 3 tier returns to the
        # fast tier after this many CONSECUTIVE forwards in which all of its rows kept a probability sum of at least tier_return_margin x
        # the guard threshold.  0 (default) = heads never return by themselves.
        self.tier_return_after = 0
        self.tier_return_margin = 4.0
        self._calm_streak = {}

    # ------------------------------------------------------------------ fp16 operand range (split / fp16 modes)
    def _flag_view(self, ws, B, V, h, w, words=1):
        off, n = C.c_size_t(), C.c_size_t()
        _lib.check(_lib.load().parq_workspace_lookup(self._handle(apply_mode=False), B, V, h, w, b"flags", C.byref(off), C.byref(n)),
                   "parq_workspace_lookup")
        return ws[off.value: off.value + words].view(torch.int32)

    def _range_fallback(self, where):
        import warnings
        warnings.warn("parq_amd.PARQDecoder: a token / K / V element left the fp16 range (|x| >= 60000) in attention mode %r (%s); "
                      "the affected outputs are NaN.  Switching attention_mode to 'fp32' (exact fp32 MFMA kernels) for this module."
                      % (self.attention_mode, where), RuntimeWarning, stacklevel=3)
        self.attention_mode = "fp32"

    def _seam_fallback(self, where):
        import warnings
        warnings.warn("parq_amd.PARQDecoder: an in-launch hand-off of the decoder chain timed out (%s); the outputs of that forward are NaN.  "
                      "Switching this module to one launch per dependent stage (fuse_seams = False)." % where, RuntimeWarning, stacklevel=3)
        self.fuse_seams = False

    def _peaky_fallback(self, heads, where):
        """Move the flagged heads (bit mask) of attention mode 'split8' to the fp16 x 3 tier."""
        import warnings
        heads = int(heads) & ((1 << self.num_heads) - 1)
        new = heads & ~self.safe_heads
        if not new:
            return False
        self.safe_heads |= heads
        names = ",".join(str(h) for h in range(self.num_heads) if (new >> h) & 1)
        left = self.num_heads - bin(self.safe_heads).count("1")
        warnings.warn("parq_amd.PARQDecoder: cross-attention head(s) %s have rows that rest on too few keys for attention mode 'split8' "
                      "(%s): its error model (fp8 cross terms, fp16 probabilities) assumes rows that spread over many keys.  Those heads "
                      "now run the fp16 x 3 arithmetic of mode 'split' (%d of %d heads stay on the fast tier); reset_attention_tiers() "
                      "undoes it." % (names, where, left, self.num_heads), RuntimeWarning, stacklevel=3)
        return True

    def reset_attention_tiers(self):
        """All heads back on the fast tier of attention mode 'split8'; the next inference forward is checked synchronously again."""
        self.safe_heads = 0
        self._peaky_checked = False

    # mirror words: one pinned int32 per inference workspace (slots 1 ..), slot 0 for the training / stepping / view-sharded entry
    # points.  The device ORs into the word of the workspace a forward runs in (the pointer is read at enqueue time, include/parq_hip.h);
    # the host takes a word with one atomic exchange, so bits raised by another forward in flight are never lost.  Behind the flag
    # words sit the slots' PROGRESS words (parq_set_progress): the device stores a forward's epoch there as soon as the forward can
    # raise no more flags — what policy "sync" waits for, one chain tail before the forward's end.
    _MIRROR_SLOTS = 16

    def _progress_ptr(self, slot):
        return self._range_mirror.data_ptr() + 4 * (self._MIRROR_SLOTS + int(slot))

    def _mirror_ptr(self, slot):
        return self._range_mirror.data_ptr() + 4 * int(slot)

    def _set_mirror(self, slot):
        if self._mirror_set != slot:
            _lib.check(_lib.load().parq_set_range_mirror(self._handle(apply_mode=False), C.c_void_p(self._mirror_ptr(slot))), "parq_set_range_mirror")
            self._mirror_set = slot

    def _mirror_take(self, slot):
        return int(_lib.load().parq_mirror_take(C.c_void_p(self._mirror_ptr(slot)))) if self._range_mirror is not None else 0

    def _free_slot(self):
        used = {e.slot for e in self._ws.values()}
        for sl in range(1, self._MIRROR_SLOTS):
            if sl not in used:
                return sl
        return self._MIRROR_SLOTS - 1                   # more workspaces than words: the last word is shared (bits are still never lost)

    def _range_poll(self):
        """Host loads of the pinned words earlier forwards raise from the device (no synchronisation): bit 0 = an operand left the
        fp16 range (outputs of that forward are NaN), bit 1 = attention mode 'split8' met a row carried by too few keys on the heads of
        bits 8.. (outputs of that forward are NaN from that iteration on unless ``range_check == "off"``), bit 2 = an in-launch hand-off
        timed out."""
        if self._mirror_np is None or not self._mirror_np[:self._MIRROR_SLOTS].any():
            return
        v = 0
        for sl in np.nonzero(self._mirror_np[:self._MIRROR_SLOTS])[0].tolist():
            v |= self._mirror_take(sl)
        if v == 0 or self.range_check == "off":
            return
        if v & 4:
            self._seam_fallback("detected after an earlier forward")
        if (v & 1) and self.attention_mode in ("split", "split8", "fp16"):
            self._range_fallback("detected after an earlier forward")
        elif (v & 2) and self.attention_mode == "split8":
            self._peaky_fallback(v >> 8, "detected after an earlier forward, whose outputs are NaN from that iteration on")

    def _range_after_forward(self, entry, sc, dev):
        """"sync" policy: wait for the forward just enqueued in `entry` and read what it raised; True = re-run it (with the fp32 kernels
        after a range violation, with the flagged heads on the fp16 x 3 tier after a too-peaked row in mode 'split8', with one launch
        per stage after a hand-off timeout).  The FIRST inference forward of a module in mode 'split8' (per weight version) is checked
        this way under every policy but "off" (one synchronisation, once): a model whose attention is too peaked for that mode is peaked
        from its first call on."""
        first = self.attention_mode == "split8" and not self._peaky_checked and self.range_check != "off"
        if not first and (self.range_check != "sync" or self.attention_mode not in ("split", "split8", "fp16")):
            return False
        want_calm = (self.range_check == "sync" and self.tier_return_after > 0 and self.safe_heads != 0 and self.attention_mode == "split8"
                     and self.num_heads <= 16)
        if want_calm:
            # also needs every head's smallest row sum of this forward: a device read of the flag words (waits for the stream)
            flags = self._flag_view(entry.ws, sc.B, sc.V, sc.h, sc.w, 48).tolist()
            v = self._mirror_take(entry.slot)
        else:
            self._wait_progress(entry, dev)                   # until this forward can raise no more flags (not: until it has finished)
            v = self._mirror_take(entry.slot)
            flags = None
        if first:
            self._peaky_checked = True
        if v == 0 and not want_calm:
            return False
        if self.range_check == "off":
            return False
        rerun = False
        if (v & 4) and self.fuse_seams:                           # a hand-off timed out: re-run with one launch per stage (every policy that looks)
            self._seam_fallback("re-running this forward")
            rerun = True
        if (v & 1) and self.attention_mode in ("split", "split8", "fp16"):
            if self.range_check == "sync" or first:
                self._range_fallback("re-running this forward")
                return True
        if (v & 2) and self.attention_mode == "split8":
            if self._peaky_fallback(v >> 8, "re-running this forward"):
                self._peaky_checked = False if first else self._peaky_checked     # the re-run is checked too: other heads may follow
                return True
        if rerun:
            return True
        if want_calm:
            # heads on the fp16 x 3 tier whose rows all spread again (flags[32 + h]: the head's smallest row sum of this forward)
            limit = 256.0 * float(self.tier_return_margin)
            for h in range(self.num_heads):
                if not (self.safe_heads >> h) & 1:
                    self._calm_streak.pop(h, None)
                    continue
                code = flags[32 + h]
                lmin = float(np.array([0x7fffffff - code], dtype=np.int32).view(np.float32)[0]) if code else 0.0
                self._calm_streak[h] = self._calm_streak.get(h, 0) + 1 if lmin >= limit else 0
                if self._calm_streak[h] >= int(self.tier_return_after):
                    self.safe_heads &= ~(1 << h)                  # this forward's numbers stand (fp16 x 3); the next one tries the fast tier
                    self._calm_streak.pop(h)
        return False

    def _wait_progress(self, entry, dev):
        """Spin on the slot's progress word until the device has stored this forward's epoch there (parq_set_progress: the first launch
        behind the last iteration's cross-attention merge does — nothing after it can raise a flag, so the decision "re-run or not" is
        final ~36 us before the outputs are; those stay stream-ordered as always).  A forward that never gets there (or a library
        without the signal) ends the wait through the stream itself."""
        word, want = self._mirror_np, entry.epoch
        idx = self._MIRROR_SLOTS + entry.slot
        if want:
            query = entry.stream.query
            for spin in range(1 << 30):
                if word[idx] == want:
                    return
                if (spin & 1023) == 1023 and query():       # the stream has drained: the word is as final as it gets
                    break
        torch.cuda.current_stream(dev).synchronize()

    # ------------------------------------------------------------------ native handle
    def _handle(self, apply_mode=True):
        if self._h is None:
            lib = _lib.load()
            cfg = _lib.ParqConfig(self.dim_in, self.num_queries, self.num_semcls + 1, self.num_heads, self.ffn_dim,
                                  self.num_layers, int(self.share_weights), int(self._mean_sizes.shape[0]),
                                  (C.c_float * 6)(*self.scale))
            h = C.c_void_p()
            _lib.check(lib.parq_create(C.byref(cfg), C.byref(h)), "parq_create")
            self._h = h
            self._mode_set = None
            self._tiers_set = None
            self._seams_set = None
            self._bwd_batched_set = None
            self._bwd_streams_set = None
            self._train_ws = None
            # pinned host memory is mapped into the device address space under the same pointer (hipHostMalloc)
            self._range_mirror = torch.zeros(2 * self._MIRROR_SLOTS, dtype=torch.int32).pin_memory()
            self._mirror_np = self._range_mirror.numpy()
            _lib.check(lib.parq_set_range_mirror(h, C.c_void_p(self._range_mirror.data_ptr())), "parq_set_range_mirror")
            self._mirror_set = 0
        det = 1 if torch.are_deterministic_algorithms_enabled() else 8
        if getattr(self, "_bwd_streams_set", None) != det:
            # torch.use_deterministic_algorithms(True): the iterations of the chain backward run in turn with plain accumulation
            # instead of on eight streams with float atomics (include/parq_hip.h parq_set_backward_streams)
            _lib.check(_lib.load().parq_set_backward_streams(self._h, det), "parq_set_backward_streams")
            self._bwd_streams_set = det
            self._train_ws = None
        if self._bwd_batched_set != bool(self.backward_batched):
            _lib.check(_lib.load().parq_set_backward_batched(self._h, int(bool(self.backward_batched))), "parq_set_backward_batched")
            self._bwd_batched_set = bool(self.backward_batched)
            self._train_ws = None                      # the training workspace is carved differently
        tiers = (int(self.safe_heads) if self.num_heads <= 16 else 0, 0 if self.range_check == "off" else 1)
        if self._tiers_set != tiers:
            _lib.check(_lib.load().parq_set_head_tiers(self._h, tiers[0], tiers[1]), "parq_set_head_tiers")
            self._tiers_set = tiers
        if self._seams_set != bool(self.fuse_seams):
            _lib.check(_lib.load().parq_set_seam_fusion(self._h, int(bool(self.fuse_seams))), "parq_set_seam_fusion")
            self._seams_set = bool(self.fuse_seams)
        if apply_mode and self._mode_set != self.attention_mode:
            if self.attention_mode not in ATTENTION_MODES:
                raise ValueError(f"attention_mode must be one of {sorted(ATTENTION_MODES)}")
            _lib.check(_lib.load().parq_set_attention_mode(self._h, ATTENTION_MODES[self.attention_mode]),
                       "parq_set_attention_mode")
            self._mode_set = self.attention_mode
            self._ws.clear()
        return self._h

    def _train_mode(self):
        """Attention arithmetic of the training entry points: ``attention_mode`` where the kernels exist — the split-precision
        forward at head dims 64 / 256 (the backward kernels then work on fp32 K / V rebuilt from the split cache), the fp16 /
        bf16 forward at head dim 64 (dim <= 256) and at head dim 256 (BASELINE cfg 5 trains in fp16: the backward differentiates
        straight through the rounded K / V) — else the exact-fp32 kernels."""
        dh = self.dim_in // self.num_heads
        if self.attention_mode in ("fp16", "bf16") and ((dh == 64 and self.dim_in <= 256) or (dh == 256 and self.dim_in % 128 == 0)):
            return self.attention_mode
        if self.attention_mode == "split8" and dh == 64 and self.train_split8:
            # opt-in (see __init__): the library runs mode 4 as mode "split" wherever its kernels do not apply (d != 256, ragged
            # key counts, the per-iteration backward); the batched backward reads the forward's stage cache (attn_bwd.hip CACHE == 8)
            return "split8"
        return "split" if dh in (64, 256) and self.attention_mode != "fp32" else "fp32"

    def _handle_in_mode(self, mode):
        """The handle switched to `mode` without touching the user-facing ``attention_mode`` (the training entry points need
        the exact-fp32 attention kernels: their backward reads the fp32 K/V cache); the next inference call switches back."""
        h = self._handle(apply_mode=False)
        if self._mode_set != mode:
            _lib.check(_lib.load().parq_set_attention_mode(h, ATTENTION_MODES[mode]), "parq_set_attention_mode")
            self._mode_set = mode
            self._ws.clear()
        return h

    def __del__(self):
        try:
            self._ws.clear()
            self._purge_graphs(wait=True)
            if self._h is not None:
                _lib.load().parq_destroy(self._h)
        except Exception:
            pass

    # ------------------------------------------------------------------ captured forwards (include/parq_hip.h parq_forward_capture)
    def _retire_graphs(self, entry):
        """The graphs of a workspace that is being dropped: destroyed once everything enqueued on its stream so far has completed."""
        graphs, entry.graphs = list(entry.graphs.values()), {}
        try:
            ev = torch.cuda.Event()
            ev.record(entry.stream)
        except Exception:                                 # noqa: BLE001 - interpreter shutdown
            ev = None
        self._graveyard.append((graphs, ev))
        self._purge_graphs()

    def _purge_graphs(self, wait=False):
        keep = []
        for graphs, ev in self._graveyard:
            if ev is not None and wait:
                ev.synchronize()
            if ev is None or ev.query():
                for g in graphs:
                    _lib.load().parq_graph_destroy(g)
            else:
                keep.append((graphs, ev))
        self._graveyard = keep

    def _unique_params(self):
        """(name, parameter) of every distinct parameter.  The module tree is walked once per change of its structure: every container
        class of this module counts assignments of parameters / sub-modules in ``_PARAM_EPOCH``."""
        cache = self.__dict__.get("_param_cache")
        if cache is not None and cache[0] == _PARAM_EPOCH[0]:
            return cache[1]
        seen, out = set(), []
        for name, p in self.named_parameters(remove_duplicate=True):
            if id(p) not in seen:
                seen.add(id(p))
                out.append((name, p))
        self.__dict__["_param_cache"] = (_PARAM_EPOCH[0], out)
        return out

    def invalidate_weights(self):
        """Force a re-pack of the weight arena at the next call.  Needed only after parameter writes that bypass the version
        counter (``p.data.copy_()`` / ``.data.mul_()`` / EMA or SWA swaps through ``.data``); optimizer steps, ``load_state_dict``
        and in-place ops on the parameters themselves are detected automatically."""
        self._arena_key = None

    def _ensure_packed(self, device):
        params = self._unique_params()
        key = (str(device),) + tuple((p.data_ptr(), p._version) for _, p in params)
        if key == self._arena_key:
            return
        lib, h = _lib.load(), self._handle()
        keep = []
        for name, p in params:
            if name.startswith("parq_module.decoder.norm."):
                continue                               # never applied by the forward pass
            t = p.detach()
            if t.device != device or t.dtype != torch.float32 or not t.is_contiguous():
                t = t.to(device=device, dtype=torch.float32).contiguous()
            keep.append(t)
            _lib.check(lib.parq_set_weight(h, name.encode(), _lib.ptr(t), t.numel()), "parq_set_weight(%s)" % name)
        # BoxProcessor table: float64 on the host, float32 at use (utils/parq_utils.py:88,98)
        self._mean_dev = torch.from_numpy(self._mean_sizes.astype(np.float32)).to(device).contiguous()
        _lib.check(lib.parq_set_weight(h, b"mean_sizes", _lib.ptr(self._mean_dev), self._mean_dev.numel()), "mean_sizes")
        nbytes = lib.parq_packed_weights_bytes(h)
        old = self._arena
        if old is not None:
            # forwards enqueued on OTHER streams may still read the old arena: the caching allocator must not hand its block out
            # again before they have passed
            for st in self._arena_streams.values():
                old.record_stream(st)
        cur = torch.cuda.current_stream(device)
        self._arena = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
        _lib.check(lib.parq_pack_weights(h, _lib.ptr(self._arena), nbytes, _lib.stream_ptr()), "parq_pack_weights")
        del keep
        self._arena_key = key
        self._arena_gen += 1
        self._arena_event, self._arena_pack_stream, self._arena_streams = None, cur, {}
        self._peaky_checked = False                    # new weights: the next inference forward in mode "split8" is checked synchronously

    def _order_behind_pack(self, device):
        """The weight arena is packed — and its lazily derived forms are built — on whichever stream got there first; a forward on any
        OTHER stream waits (on the device) for that work once per (re-)pack."""
        sid = _raw_stream(device)
        if sid in self._arena_streams:
            return
        st = torch.cuda.current_stream(device)
        if self._arena_event is not None:
            st.wait_event(self._arena_event)
        elif self._arena_pack_stream is not None and int(self._arena_pack_stream.cuda_stream) != sid:
            st.wait_stream(self._arena_pack_stream)
        self._arena_streams[sid] = st

    def _mark_first_forward(self, device):
        """Behind the first forward (inference or training) enqueued after a (re-)pack: the pack, the mode's 16-bit W_kv copy and the
        derived inference weights are all in front of this event."""
        if self._arena_event is None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(device))
            self._arena_event = ev

    def _workspace_entry(self, B, V, h, w, device, handle=None):
        """Workspace (K/V cache + activations) of a batch shape.  The ``max_workspaces`` most recently used shapes stay alive, so
        a driver that alternates two shapes (e.g. train / validation snippets) does not reallocate a K/V cache per call; each holds
        a K/V cache (393 MB per scene at BASELINE cfg 3), so ``max_workspaces = 1`` halves the module's footprint for single-shape
        drivers.  ``handle``: the native handle already switched to the mode of this call (else ``attention_mode`` is applied)."""
        if handle is None:
            handle = self._handle()                   # first: a pending mode change drops the cached workspaces (their carving differs)
        # keyed by the launch stream too: forwards enqueued on different streams (two scenes in flight: the small-op chain of one
        # leaves most of the chip to the K/V projection and cross-attention of the other, +18 % throughput at BASELINE cfg 3,
        # profiles/r05_two_in_flight.txt) each own a workspace; a workspace is allocated, used and freed in the order of ONE stream
        k = (B, V, h, w, device.index, _raw_stream(device))
        entry = self._ws.take(k)
        if entry is None:
            st = torch.cuda.current_stream(device)
            nbytes = _lib.load().parq_workspace_bytes(handle, B, V, h, w)
            if nbytes == 0:
                raise RuntimeError("parq_workspace_bytes returned 0 for B=%d V=%d h=%d w=%d" % (B, V, h, w))
            while len(self._ws) >= max(1, int(self.max_workspaces)):
                self._ws.pop(next(iter(self._ws)))    # dicts iterate in insertion order: the first key is the least recently used
            entry = _WsEntry(torch.empty(nbytes // 4, dtype=torch.float32, device=device), self._free_slot(), st)
        self._ws[k] = entry                           # (re-)insert as the most recently used
        return entry

    def _workspace(self, B, V, h, w, device, handle=None):
        return self._workspace_entry(B, V, h, w, device, handle).ws

    # ------------------------------------------------------------------ argument packing
    def _scene(self, tokens, camera, T_cp, T_wp, T_wl, feat_hw):
        tokens = raw(tokens)
        cam, T_cp, T_wp, T_wl = raw(camera), raw(T_cp), raw(T_wp), raw(T_wl)
        if not tokens.is_cuda:
            raise RuntimeError("parq_amd.PARQDecoder runs on the GPU only: move the module and its inputs to 'cuda' "
                               "(there is no CPU fallback)")
        dev = tokens.device

        def prep(t, last):
            if t.device != dev or t.dtype != torch.float32:
                t = t.to(device=dev, dtype=torch.float32)
            assert t.shape[-1] == last, (tuple(t.shape), last)
            return t if t.is_contiguous() else t.contiguous()
        tokens = prep(tokens, self.dim_in)
        cam, T_cp, T_wp, T_wl = prep(cam, 6), prep(T_cp, 12), prep(T_wp, 12), prep(T_wl, 12)
        B, N, _ = tokens.shape
        assert cam.dim() == 3 and cam.shape[0] == B, "camera must be (B,V,6)"
        V = cam.shape[1]
        if T_wl.dim() == 2:
            T_wl = T_wl.unsqueeze(1)
        assert T_cp.shape == (B, V, 12) and T_wp.shape == (B, V, 12) and T_wl.shape == (B, 1, 12)
        if feat_hw is None:
            # the reference reads the size from the first camera on the host too
            # (transformer_parq.py:301-302); pass feat_hw=(h,w) to avoid this sync
            wf, hf = cam[0, 0, :2].tolist()
            feat_hw = (int(hf), int(wf))
        h, w = int(feat_hw[0]), int(feat_hw[1])
        assert V * h * w == N, "tokens (N=%d) do not match V*h*w = %d*%d*%d" % (N, V, h, w)
        sc = _lib.ParqScene(B, V, h, w, _lib.ptr(tokens), _lib.ptr(cam), _lib.ptr(T_cp), _lib.ptr(T_wp), _lib.ptr(T_wl))
        return sc, (tokens, cam, T_cp, T_wp, T_wl), dev

    def _alloc_outputs(self, lead, device):
        ncls = self.num_semcls + 1
        widths = (ncls, 3, 3, 6, ncls, 3)
        return [torch.empty(*lead, wd, dtype=torch.float32, device=device) for wd in widths]

    def _alloc_outputs_flat(self, lead, device, flat=None):
        """The six output tensors of an inference call as views of ONE allocation (`flat`: carve that tensor instead of a fresh one);
        returns (views, flat)."""
        ncls = self.num_semcls + 1
        widths = (ncls, 3, 3, 6, ncls, 3)
        rows = 1
        for d in lead:
            rows *= int(d)
        if flat is None:
            flat = torch.empty(rows * sum(widths), dtype=torch.float32, device=device)
        outs, off = [], 0
        for wd in widths:
            outs.append(flat[off: off + rows * wd].view(*lead, wd))
            off += rows * wd
        return outs, flat

    def _check_mode(self):
        """Inference entry points ignore dropout exactly like nn.Dropout in eval mode; in train mode WITHOUT autograd
        (torch.no_grad around a training module) the reference would still drop: use forward_train for that."""
        if self.training and self.dropout_rate > 0:
            raise RuntimeError("parq_amd.PARQDecoder is in train mode with dropout %.2f but autograd is disabled: call "
                               ".eval() for inference or forward_train() for a dropout forward without a graph" % self.dropout_rate)

    # ------------------------------------------------------------------ forward (model/parq_decoder.py:134-163)
    def forward(self, intput_tokens, camera, T_camera_pseudoCam, T_world_pseudoCam, T_world_local, feat_hw=None):
        """Same dispatch as the reference module gets from autograd: with gradients enabled and anything to differentiate (a
        parameter or the tokens require grad) the result carries a graph — in ``train()`` mode with the decoder layer's dropout,
        in ``eval()`` mode without it (the reference differentiates in eval mode too, model/parq_decoder.py:134-163).  Under
        ``torch.no_grad()`` (eval.py:46, Lightning's validation loop), or with nothing that requires grad, the inference chain
        runs: no saved activations, the folded position MLP, one K/V workspace."""
        if torch.is_grad_enabled() and (self.training or self._needs_graph(intput_tokens)):
            if not self.training and not getattr(self, "_warned_eval_autograd", False):
                import warnings
                self._warned_eval_autograd = True
                warnings.warn("parq_amd.PARQDecoder.forward in eval() mode with gradients enabled builds an autograd graph, as the reference "
                              "module does: the training forward runs (attention mode 'split', saved activations of every iteration, one "
                              "training workspace per output kept alive).  Wrap inference in torch.no_grad() (eval.py:46) for the "
                              "inference chain.", RuntimeWarning, stacklevel=2)
            return self._forward_autograd(intput_tokens, camera, T_camera_pseudoCam, T_world_pseudoCam, T_world_local, feat_hw)
        with torch.no_grad():
            return self._forward_inference(intput_tokens, camera, T_camera_pseudoCam, T_world_pseudoCam, T_world_local, feat_hw)

    def _needs_graph(self, tokens):
        t = raw(tokens)
        return bool(getattr(t, "requires_grad", False)) or any(p.requires_grad for p in self.parameters())

    def _forward_autograd(self, tokens, camera, T_cp, T_wp, T_wl, feat_hw):
        """Forward under autograd (train mode, or eval mode with something to differentiate): one autograd node whose backward
        is the HIP backward chain.  Attention arithmetic = ``_train_mode()``: the split-precision kernels where the head dim has
        them (64 / 256) unless ``attention_mode == "fp32"``, else the exact fp32 MFMA kernels.  In train mode DROPOUT_RATE > 0
        applies the decoder layer's six dropout sites with counter-based masks (seeded from torch's generator per call); in eval
        mode dropout is off, as nn.Dropout.  Every node owns the saved activations of its forward (``_Stash``): several forwards
        may be outstanding, each holding one training workspace until its backward has run or its graph is dropped."""
        params = [p for _, p in self._unique_params()]
        stacked = _TrainFn.apply(self, raw(tokens), camera, T_cp, T_wp, T_wl, feat_hw, *params)
        return [{k: t[i] for k, t in zip(OUTPUT_KEYS, stacked)} for i in range(self.num_layers)]

    def _forward_inference(self, intput_tokens, camera, T_camera_pseudoCam, T_world_pseudoCam, T_world_local, feat_hw=None):
        self._check_mode()
        self._range_poll()
        sc, keep, dev = self._scene(intput_tokens, camera, T_camera_pseudoCam, T_world_pseudoCam, T_world_local, feat_hw)
        self._ensure_packed(dev)
        self._order_behind_pack(dev)
        lead = (self.num_layers, sc.B, self.num_queries)
        flat = torch.empty(self.num_layers * sc.B * self.num_queries * self._out_width, dtype=torch.float32, device=dev)
        entry = self._enqueue_forward(sc, keep, flat, dev)
        self.__dict__["_last_flat"] = flat                     # (InFlight: the one allocation behind the outputs)
        # (everything below runs while the device works on the forward)
        rows = self.num_layers * sc.B * self.num_queries
        ncls = self.num_semcls + 1
        widths = (ncls, 3, 3, 6, ncls, 3)
        per = [seg.view(*lead, wd).unbind(0) for seg, wd in zip(flat.split([rows * wd for wd in widths]), widths)]
        result = [dict(zip(OUTPUT_KEYS, [p[i] for p in per])) for i in range(self.num_layers)]

        def settle():
            """What policy "sync" owes the caller (and every policy owes a module's first forward): wait, look, and re-run — with the
            exact fp32 kernels (range), or with the flagged heads on the fp16 x 3 tier (each re-run can only add heads: at most
            num_heads of them) — into the SAME output tensors."""
            e = entry
            for _attempt in range(self.num_heads + 2):
                if not self._range_after_forward(e, sc, dev):
                    break
                e = self._enqueue_forward(sc, keep, flat, dev)
            return result

        if self._defer is not None and self.range_check == "sync" and self.attention_mode in ("split", "split8", "fp16"):
            self._defer.append(settle)         # InFlight.submit: the wait belongs to Ticket.result(), other forwards keep the device busy
            return result
        return settle()

    def _out_pointers(self, base, rows):
        """parq_outputs over one flat allocation at device address `base`: the six tensors back to back, `rows` rows each."""
        ncls = self.num_semcls + 1
        ptrs, off = [], 0
        for wd in (ncls, 3, 3, 6, ncls, 3):
            ptrs.append(C.c_void_p(base + 4 * off))
            off += rows * wd
        return _lib.ParqOutputs(*ptrs)

    def _enqueue_forward(self, sc, keep, flat, dev):
        """One inference forward into `flat` (the six output tensors back to back).  From the second forward of a (workspace, weights,
        attention settings) on, the iterations are replayed from a captured graph behind the directly launched prologue and K/V
        projection (parq_forward_replay); launch by launch otherwise (parq_forward).  Returns the workspace entry."""
        lib = _lib.load()
        h = self._handle()
        entry = self._workspace_entry(sc.B, sc.V, sc.h, sc.w, dev, handle=h)
        ws = entry.ws
        self._set_mirror(entry.slot)
        self._epoch = (self._epoch % 0x7ffffff0) + 1
        entry.epoch = self._epoch
        _lib.check(lib.parq_set_progress(h, C.c_void_p(self._progress_ptr(entry.slot)), self._epoch), "parq_set_progress")
        stream = C.c_void_p(entry.stream.cuda_stream)
        po = self._out_pointers(flat.data_ptr(), self.num_layers * sc.B * self.num_queries)
        graph = None
        if self.use_graph and not self._profiling:
            key = (self._arena_gen, self._mode_set, self._tiers_set, self._seams_set)
            graph = entry.graphs.get(key)
            if graph is None and entry.last_key == key:
                graph = self._capture(entry, key, sc, stream)
            entry.last_key = key
        if graph is not None:
            _lib.check(lib.parq_forward_replay(h, graph, C.byref(sc), _lib.ptr(ws), ws.numel() * 4, C.byref(po), stream), "parq_forward_replay")
            entry.replays += 1
        else:
            _lib.check(lib.parq_forward(h, C.byref(sc), _lib.ptr(ws), ws.numel() * 4, C.byref(po), stream), "parq_forward")
        self._mark_first_forward(dev)
        return entry

    def _capture(self, entry, key, sc, stream):
        """Record the iterations of this workspace's forward into a HIP graph (parq_forward_capture); graphs of earlier weights / settings
        of the same workspace are retired."""
        if entry.graphs:
            stale = _WsEntry(None, 0, entry.stream)
            stale.graphs, entry.graphs = entry.graphs, {}
            self._retire_graphs(stale)
        g = C.c_void_p()
        _lib.check(_lib.load().parq_forward_capture(self._handle(), sc.B, sc.V, sc.h, sc.w, _lib.ptr(entry.ws), entry.ws.numel() * 4, stream,
                                                    C.byref(g)), "parq_forward_capture")
        entry.graphs[key] = g
        return g

    # ------------------------------------------------------------------ training
    @torch.no_grad()
    def forward_train(self, intput_tokens, camera, T_camera_pseudoCam, T_world_pseudoCam, T_world_local, feat_hw=None,
                      defer_range_check=False):
        """Forward that keeps every iteration's activations for ``backward`` (attention arithmetic: ``_train_mode()``;
        dropout when the module is in train mode).  Returns the same list of dicts as ``forward``.  ``defer_range_check``
        (the autograd path with ``overlap_loss_matching``): do not wait for the device here; ``loss()`` resolves the fp16-range
        check of this forward while it matches, ``backward()`` raises if nobody did."""
        self._range_poll()                     # BEFORE the mode of this step is chosen: a fallback must not split forward / backward
        owner = self._stash_owner() if self._stash_owner is not None else None
        if owner is not None and not owner.consumed and owner.ws is self._train_ws:
            # an autograd node of an earlier forward still needs its activations: settle that forward's deferred range check
            # (it would re-run into the workspace we are about to leave) and give this forward a workspace of its own
            self._resolve_train_range()
            self._train_ws = None
        sc, keep, dev = self._scene(intput_tokens, camera, T_camera_pseudoCam, T_world_pseudoCam, T_world_local, feat_hw)
        self._ensure_packed(dev)
        self._order_behind_pack(dev)
        self._set_mirror(0)
        lib = _lib.load()
        # everything the caller enqueued so far (in particular the targets loss() will read on its side stream: an asynchronous
        # host-to-device copy, on-device augmentation) is ordered before this event
        self._train_entry_event = torch.cuda.Event()
        self._train_entry_event.record(torch.cuda.current_stream(dev))
        # decoder-layer dropout (train mode only, as nn.Dropout): a fresh mask seed per call, reused by backward()
        p_drop = float(self.dropout_rate) if self.training else 0.0
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p_drop > 0 else 0
        outs = self._alloc_outputs((self.num_layers, sc.B, self.num_queries), dev)
        po = _lib.ParqOutputs(*[_lib.ptr(t) for t in outs])

        def enqueue():
            mode = self._train_mode()
            h = self._handle_in_mode(mode)
            _lib.check(lib.parq_set_dropout(h, p_drop, seed), "parq_set_dropout")
            nbytes = lib.parq_train_workspace_bytes(h, sc.B, sc.V, sc.h, sc.w)
            old_ws = self._train_ws
            if self._train_ws is None or self._train_ws.numel() * 4 < nbytes or self._train_ws.device != dev:
                self._ws.clear()
                self._train_ws = torch.empty(nbytes // 4 + 1, dtype=torch.float32, device=dev)
            _lib.check(lib.parq_forward_train(h, C.byref(sc), _lib.ptr(self._train_ws), self._train_ws.numel() * 4, C.byref(po),
                                              _lib.stream_ptr()), "parq_forward_train")
            self._mark_first_forward(dev)
            # the stash is laid out for `mode`: backward() uses exactly this mode, whatever attention_mode says by then
            self._train_state = (sc, keep, outs, po, dev, mode, p_drop, seed)
            own = self._stash_owner() if self._stash_owner is not None else None
            if own is not None and old_ws is not None and own.ws is old_ws and not own.consumed and getattr(old_ws, "_parq_gen", None) == own.gen:
                # a re-run (range fallback) after the autograd node of THIS forward took the stash: new mode, and — where the new
                # mode's workspace is larger — a new workspace; the node must not keep the poisoned forward's activations
                own.state = self._train_state
                if self._train_ws is not old_ws:
                    own.ws = self._train_ws
                    self._train_ws._parq_gen = own.gen
            return mode
        mode = enqueue()
        self._train_pending = None
        if self.range_check != "off" and mode in ("split", "split8", "fp16"):
            # Training never lets a range violation reach the optimizer: a poisoned forward (NaN outputs, device flag, pinned host
            # word) is re-run with the exact fp32 kernels (same dropout seed, same output tensors) before its outputs are used.
            def rerun_if_poisoned(completed):
                """`completed`: the caller knows the forward has finished on the device (the pinned word is then current);
                otherwise the device flag is read, which waits for the stream."""
                if completed:
                    m = int(self._range_mirror[0])
                    rng, peaked = (m & 1) != 0, (m >> 8) if (m & 2) else 0
                else:
                    f = self._flag_view(self._train_ws, sc.B, sc.V, sc.h, sc.w, 2).tolist()
                    rng, peaked = f[0] != 0, f[1]
                if not rng and not (peaked and mode == "split8"):
                    return False
                self._range_mirror[0] = 0
                if rng:
                    self._range_fallback("re-running this training forward")
                else:                                        # train_split8: the flagged step is re-run in mode "split" (any safe head: mode 1)
                    self._peaky_fallback(peaked, "re-running this training forward in mode 'split'")
                enqueue()
                return True
            if defer_range_check:
                self._train_pending = rerun_if_poisoned       # resolved by loss() / backward(): no host synchronisation here
            else:
                rerun_if_poisoned(False)                     # one host synchronisation per step
        self._train_gen += 1
        self._train_ws._parq_gen = self._train_gen             # which forward's activations the workspace holds
        return [{k: t[i] for k, t in zip(OUTPUT_KEYS, outs)} for i in range(self.num_layers)]

    @torch.no_grad()
    def backward(self, grad_outputs, want_token_grad=True, _stash=None):
        """Backward of the last ``forward_train``.  ``grad_outputs``: dict with any of pred_logits / center_unnormalized /
        size_unnormalized / ortho6d -> (I, B, Q, k) cotangents (missing = zero).  Returns ({reference tensor name:
        gradient}, d_tokens or None); gradients of tensors registered under two names are returned once per name."""
        if _stash is None and self._train_state is None:
            raise RuntimeError("backward() needs a preceding forward_train()")
        state = _stash.state if _stash is not None else self._train_state
        train_ws = _stash.ws if _stash is not None else self._train_ws
        if state is self._train_state:                         # the most recent forward: its range check may still be pending
            self._resolve_train_range(in_backward=True)
        sc, keep, outs, po, dev, mode, p_drop, seed = state
        # the mode the stash was written in (forward_train), not whatever attention_mode says now: the workspace layout differs
        lib, h = _lib.load(), self._handle_in_mode(mode)
        _lib.check(lib.parq_set_dropout(h, p_drop, seed), "parq_set_dropout")      # the masks of THAT forward (another one may have run since)
        gs = []
        for key, wd in (("pred_logits", self.num_semcls + 1), ("center_unnormalized", 3), ("size_unnormalized", 3), ("ortho6d", 6)):
            g = grad_outputs.get(key)
            if g is not None:
                g = g.to(device=dev, dtype=torch.float32).contiguous()
                assert g.shape == (self.num_layers, sc.B, self.num_queries, wd), (key, tuple(g.shape))
            gs.append(g)
        pg = _lib.ParqOutputGrads(*[_lib.ptr(g) for g in gs])
        arena = torch.empty(lib.parq_grad_arena_bytes(h) // 4, dtype=torch.float32, device=dev)
        N = sc.V * sc.h * sc.w
        d_tokens = torch.empty(sc.B, N, self.dim_in, dtype=torch.float32, device=dev) if want_token_grad else None
        _lib.check(lib.parq_backward(h, C.byref(sc), _lib.ptr(train_ws), train_ws.numel() * 4, C.byref(po), C.byref(pg),
                                     _lib.ptr(arena), _lib.ptr(d_tokens), _lib.stream_ptr()), "parq_backward")
        if self.dp_all_reduce:
            # data-parallel training (train.py:103 DDP semantics: mean over ranks): the gradient arena is one flat buffer laid
            # out in the order parq_backward finishes it, so it is averaged in TWO buckets instead of one bucket per tensor —
            # the first (everything above the cross-attention) on a side stream as soon as phase 1 of the backward has written
            # it, i.e. beside the cross-attention backward that is still enqueued; the second at the end
            from .parallel import all_reduce_mean_, all_reduce_mean_buckets_
            if self.dp_bucketed and arena.is_cuda:
                off, cnt = C.c_int64(), C.c_int64()
                buckets = []
                for b in (0, 1):
                    _lib.check(lib.parq_grad_bucket(h, b, C.byref(off), C.byref(cnt)), "parq_grad_bucket")
                    buckets.append((off.value, cnt.value))
                which = [b for b in (0, 1) if buckets[b][1] > 0]

                def ready(i, stream, _which=which):
                    if stream is not None:
                        _lib.check(lib.parq_backward_wait_bucket(h, _which[i], C.c_void_p(stream.cuda_stream)), "parq_backward_wait_bucket")
                if self._dp_stream is None or self._dp_stream.device != arena.device:
                    self._dp_stream = torch.cuda.Stream(device=arena.device)
                all_reduce_mean_buckets_(arena, buckets, ready=ready, side_stream=self._dp_stream)
            else:
                all_reduce_mean_(arena)
        grads = {}
        off, rows, cols, ld = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        for name, p in self._unique_params():
            if name.startswith("parq_module.decoder.norm."):
                continue
            _lib.check(lib.parq_arena_lookup(h, name.encode(), C.byref(off), C.byref(rows), C.byref(cols), C.byref(ld)),
                       "parq_arena_lookup(%s)" % name)
            g = arena[off.value: off.value + rows.value * cols.value].view(rows.value, cols.value)
            grads[name] = g.reshape(p.shape)          # a view of the arena (fresh per call): no per-tensor copy kernels
        return grads, d_tokens

    # ------------------------------------------------------------------ stepping interface (tests, custom drivers)
    @torch.no_grad()
    def prepare(self, intput_tokens, camera, T_camera_pseudoCam, T_world_pseudoCam, T_world_local, feat_hw=None):
        self._check_mode()
        sc, keep, dev = self._scene(intput_tokens, camera, T_camera_pseudoCam, T_world_pseudoCam, T_world_local, feat_hw)
        self._ensure_packed(dev)
        self._order_behind_pack(dev)
        entry = self._workspace_entry(sc.B, sc.V, sc.h, sc.w, dev)
        ws = entry.ws
        self._set_mirror(entry.slot)
        _lib.check(_lib.load().parq_prepare(self._handle(), C.byref(sc), _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr()),
                   "parq_prepare")
        self._step = (sc, keep, ws, dev)

    @torch.no_grad()
    def iterate(self, layer_num, ref_in=None):
        """One recurrent iteration; ``ref_in`` (B,Q,3) normalised reference points or None to
        continue.  Returns (out_dict, next_ref)."""
        sc, keep, ws, dev = self._step
        outs = self._alloc_outputs((sc.B, self.num_queries), dev)
        po = _lib.ParqOutputs(*[_lib.ptr(t) for t in outs])
        nxt = torch.empty(sc.B, self.num_queries, 3, dtype=torch.float32, device=dev)
        if ref_in is not None:
            ref_in = ref_in.to(device=dev, dtype=torch.float32).contiguous()
            assert ref_in.shape == (sc.B, self.num_queries, 3)
        _lib.check(_lib.load().parq_iterate(self._handle(), C.byref(sc), _lib.ptr(ws), ws.numel() * 4, int(layer_num),
                                            _lib.ptr(ref_in), C.byref(po), _lib.ptr(nxt), _lib.stream_ptr()),
                   "parq_iterate")
        return dict(zip(OUTPUT_KEYS, outs)), nxt

    @torch.no_grad()
    def forward_view_sharded(self, intput_tokens, camera, T_camera_pseudoCam, T_world_pseudoCam, T_world_local, feat_hw=None,
                             group=None, forced_refs=None):
        """Intra-scene view sharding (SURVEY.md 8e "split-N"; include/parq_hip.h parq_iterate_sharded): THIS rank passes only
        ITS views of every scene — tokens (B, V_local*h*w, C) and the camera / pose tensors of those views — and every rank of
        ``group`` (default process group) calls this collectively.  The K/V projection, the K/V cache and the cross-attention
        stream are sharded over the ranks; per iteration two small collectives merge the ranks' contributions (SUM all-reduce
        of the sampled-feature sums and valid-view counts; all-gather of the per-shard attention outputs and log-sum-exp rows:
        ~0.27 MB each at Q = 256, d = 256), everything else is computed identically on every rank.  Returns the same list of
        per-iteration dicts as ``forward`` on every rank.  ``forced_refs``: optional list of (B,Q,3) reference points per
        iteration (teacher forcing, as ``iterate``).

        fp16 operand range: the ranks' range flags travel in the first exchange (one float, summed), so a violation in ANY rank's
        shard poisons the outputs on EVERY rank and every rank reads the same flag after the last iteration (one host
        synchronisation per forward on this path, which already hands two collectives per iteration to the host): under
        ``range_check = "sync"`` all ranks re-run the forward with the exact fp32 kernels, under "lazy" all ranks warn, return
        the NaN outputs and use the fp32 kernels from the next call on.  The ranks never end up in different attention modes."""
        import torch.distributed as dist
        self._check_mode()
        self._range_poll()
        sc, keep, dev = self._scene(intput_tokens, camera, T_camera_pseudoCam, T_world_pseudoCam, T_world_local, feat_hw)
        self._ensure_packed(dev)
        self._order_behind_pack(dev)
        self._set_mirror(0)
        lib = _lib.load()
        world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        for _attempt in range(2):
            # mode "split8" runs as "split" here: its peakedness guard looks at whole rows, a rank sees only its shard of the keys
            h = self._handle_in_mode("split") if self.attention_mode == "split8" else self._handle()
            ws = self._workspace(sc.B, sc.V, sc.h, sc.w, dev, handle=h)       # (not through _handle(): it would re-apply "split8")
            _lib.check(lib.parq_prepare(h, C.byref(sc), _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr()), "parq_prepare")
            na, nb = lib.parq_shard_exchange_floats(h, sc.B, 0), lib.parq_shard_exchange_floats(h, sc.B, 1)
            xa = torch.empty(na, dtype=torch.float32, device=dev)
            xb = torch.empty(nb, dtype=torch.float32, device=dev)
            gathered = torch.empty(world * nb, dtype=torch.float32, device=dev)
            results = []
            for k in range(self.num_layers):
                outs = self._alloc_outputs((sc.B, self.num_queries), dev)
                po = _lib.ParqOutputs(*[_lib.ptr(t) for t in outs])
                ref_in = None
                if forced_refs is not None:
                    ref_in = forced_refs[k].to(device=dev, dtype=torch.float32).contiguous()

                def phase(ph, xin, xout, _po=po, _ref=ref_in, _k=k):
                    _lib.check(lib.parq_iterate_sharded(h, C.byref(sc), _lib.ptr(ws), ws.numel() * 4, _k, ph, _lib.ptr(_ref), C.byref(_po),
                                                        None, _lib.ptr(xin), _lib.ptr(xout), world, _lib.stream_ptr()),
                               "parq_iterate_sharded(phase %d)" % ph)

                phase(0, None, xa)
                if world > 1:
                    dist.all_reduce(xa, group=group)
                phase(1, xa, xb)
                if world > 1:
                    dist.all_gather_into_tensor(gathered, xb, group=group)
                else:
                    gathered.copy_(xb)
                phase(2, gathered, None)
                results.append(dict(zip(OUTPUT_KEYS, outs)))
            if self.range_check == "off" or self.attention_mode not in ("split", "split8", "fp16"):
                break
            # identical on every rank (the flags were summed in the first exchange of every iteration)
            if int(self._flag_view(ws, sc.B, sc.V, sc.h, sc.w).item()) == 0:
                break
            self._range_mirror[0] = 0
            self._range_fallback("view-sharded forward: a rank's K/V shard left the range" +
                                 ("; re-running on every rank" if self.range_check == "sync" else ""))
            if self.range_check != "sync":
                break
        del keep
        return results

    def _last_ws(self):
        k, entry = list(self._ws.items())[-1]
        return k, entry.ws

    def fp16_range_exceeded(self):
        """True if the last prepare() / forward() / forward_train() saw a token, K or V element outside the fp16 range while
        building the 16-bit K/V cache (synchronises; meaningful in the "split" and "fp16" modes).  Outputs of such a call are
        NaN by construction (include/parq_hip.h); see ``range_check`` for the automatic handling."""
        if self._ws:
            (B, V, h, w, _, _), ws = self._last_ws()                   # the most recently used workspace
        elif self._train_ws is not None and self._train_state is not None:
            sc = self._train_state[0]
            (B, V, h, w), ws = (sc.B, sc.V, sc.h, sc.w), self._train_ws
        else:
            return False
        return bool(self._flag_view(ws, B, V, h, w).item() != 0)

    def attention_too_peaked(self):
        """True if the last inference forward in attention mode "split8" met a cross-attention row whose probabilities (relative to
        the row's reference maximum) sum to less than 256 on a head of the fast tier, i.e. a row that rests on too few keys for that
        mode's error model (synchronises).  ``range_check`` handles it: "sync" re-runs the forward with the flagged heads on the
        fp16 x 3 tier before returning, "lazy" moves them at the next call (``safe_heads``)."""
        if not self._ws:
            return False
        (B, V, h, w, _, _), ws = self._last_ws()
        return bool(self._flag_view(ws, B, V, h, w, 2)[1].item() != 0)

    def attention_peaked_map(self):
        """Per recurrent iteration of the last inference forward: bit mask of the heads on which attention mode "split8" met a row
        under the guard threshold (synchronises).  The guard acts per head; this map says in which iterations."""
        if not self._ws:
            return []
        (B, V, h, w, _, _), ws = self._last_ws()
        f = self._flag_view(ws, B, V, h, w, 64).tolist()
        return [f[8 + (k % 56)] for k in range(self.num_layers)]

    def attention_min_row_sum(self, per_head=False):
        """Smallest row probability sum (relative to the row's reference maximum) the mode-"split8" heads of the last inference forward
        saw, or None (synchronises).  The guard threshold is 256 (include/parq_hip.h, attention mode 4).  ``per_head``: a list with
        every head's smallest sum instead — heads on the fp16 x 3 tier included (what ``tier_return_after`` looks at)."""
        if not self._ws:
            return None
        (B, V, h, w, _, _), ws = self._last_ws()
        dec = lambda code: float(np.array([0x7fffffff - code], dtype=np.int32).view(np.float32)[0]) if code else None
        if per_head:
            f = self._flag_view(ws, B, V, h, w, 48).tolist()
            return [dec(f[32 + i]) for i in range(min(self.num_heads, 16))]
        return dec(int(self._flag_view(ws, B, V, h, w, 3)[2].item()))

    def intermediate(self, name):
        """View of a named workspace buffer after prepare()/iterate() (parity tests)."""
        sc, keep, ws, dev = self._step
        off, n = C.c_size_t(), C.c_size_t()
        _lib.check(_lib.load().parq_workspace_lookup(self._handle(), sc.B, sc.V, sc.h, sc.w, name.encode(),
                                                     C.byref(off), C.byref(n)), "parq_workspace_lookup")
        return ws[off.value: off.value + n.value]

    # ------------------------------------------------------------------ profiling hooks used by bench.py
    def profile_enable(self, on=True):
        self._profiling = bool(on)                      # the library's per-group events are host-side records: no graph replay meanwhile
        _lib.check(_lib.load().parq_profile_enable(self._handle(), int(on)), "parq_profile_enable")

    def profile_read(self):
        lib, h = _lib.load(), self._handle()
        res = {}
        for i, name in enumerate(_lib.PROF_NAMES):
            ms, n = C.c_double(), C.c_int64()
            _lib.check(lib.parq_profile_read(h, i, C.byref(ms), C.byref(n)), "parq_profile_read")
            res[name] = (ms.value, n.value)
        return res

    # ------------------------------------------------------------------ next-tier rows
    def loss(self, out_dict_list, obbs_padded, T_world_local, sym=None, *argv):
        """Hungarian-matched set loss, model/parq_decoder.py:264-370 (host-side torch + scipy as in the reference;
        parq_amd/loss.py).  Under autograd its gradient reaches the weights through the HIP backward chain."""
        from .loss import HungarianMatcherModified, decoder_loss, decoder_loss_batched
        if self._matcher is None:
            self._matcher = HungarianMatcherModified(cost_class=2, cost_bbox=0.25)          # parq_decoder.py:71
            self._class_weight = torch.ones(self.num_semcls + 1)
            self._class_weight[self.num_semcls] = 0.1                                        # background (:46-48)
        kw = dict(matcher=self._matcher, loss_weight=self.loss_weight, num_semcls=self.num_semcls, class_weight=self._class_weight)
        if not self.loss_batched:
            self._resolve_train_range()
            return decoder_loss(out_dict_list, obbs_padded, T_world_local, sym, **kw)
        ready = None
        st = self._train_state
        if (self.overlap_loss_matching and st is not None and len(out_dict_list) == self.num_layers and self.num_layers <= 16
                and out_dict_list[0]["pred_logits"].data_ptr() == st[2][0].data_ptr()):
            ready = self._train_ready                      # these ARE the outputs of the training forward in flight
        else:
            self._resolve_train_range()
        return decoder_loss_batched(out_dict_list, obbs_padded, T_world_local, sym, ready=ready,
                                    targets_ready=self._train_entry_event if (ready is not None and not self.loss_targets_late) else None, **kw)

    def wait_iteration(self, k):
        """Block the host until iteration k of the last training forward has written its outputs (parq_wait_iteration)."""
        _lib.check(_lib.load().parq_wait_iteration(self._handle(apply_mode=False), int(k)), "parq_wait_iteration")

    def _train_ready(self, k):
        """loss(): outputs of iteration k are final; True = the forward was re-run (range fallback) and every iteration changed."""
        self.wait_iteration(k)
        if self._train_pending is None:
            return False
        if (int(self._range_mirror[0]) & 3) == 0 and k + 1 < self.num_layers:
            return False                                   # nothing raised so far
        self.wait_iteration(self.num_layers - 1)           # the whole forward, then the pinned word is final
        pending, self._train_pending = self._train_pending, None
        return bool(pending(True))

    def _resolve_train_range(self, in_backward=False):
        """The deferred fp16-range check of the last training forward, for callers that did not go through the overlapped loss."""
        if self._train_pending is None:
            return
        pending, self._train_pending = self._train_pending, None
        if not in_backward:
            pending(False)                                 # outputs not consumed yet by this module: re-run in place if poisoned
            return
        self.wait_iteration(self.num_layers - 1)
        m = int(self._range_mirror[0])
        if (m & 3) != 0:
            self._range_mirror[0] = 0
            if m & 1:
                self._range_fallback("detected in backward()")
            else:
                self._peaky_fallback(m >> 8, "detected in backward()")
            raise RuntimeError("parq_amd.PARQDecoder: the training forward of this step left the fp16 operand range (or, with train_split8, "
                               "met attention rows too peaked for mode 'split8') and its outputs are NaN; they were consumed outside "
                               "PARQDecoder.loss, so the step cannot be repaired here.  Skip this step (the module now uses the safer "
                               "kernels), or set overlap_loss_matching = False to have the training forward check and re-run before it returns.")

    @torch.no_grad()
    def parse_pred(self, out_dict):
        """model/parq_decoder.py:372-424 on the device (parq_parse_pred: one workgroup per scene builds the boxes, the
        validity window and runs the 3-D NMS; the reference round-trips through NumPy).  Adds ``obbs_pred`` (Obb3D, (B,Q,19))
        and ``pred_mask`` (bool (B,Q)) to the last iteration's dict and returns it."""
        from .wrappers import Obb3D
        out = out_dict[-1] if isinstance(out_dict, (list, tuple)) else out_dict
        ctr, size, rot6, prob = (out[k].detach().to(torch.float32).contiguous() for k in
                                 ("center_unnormalized", "size_unnormalized", "ortho6d", "sem_cls_prob"))
        B, Q = ctr.shape[:2]
        obbs = torch.empty(B, Q, 19, dtype=torch.float32, device=ctr.device)
        mask = torch.empty(B, Q, dtype=torch.uint8, device=ctr.device)
        _lib.check(_lib.load().parq_parse_pred(_lib.ptr(ctr), _lib.ptr(size), _lib.ptr(rot6), _lib.ptr(prob), B, Q, self.num_semcls + 1,
                                               (C.c_float * 6)(*[float(x) for x in self.track_scale]), int(bool(self.for_vis)),
                                               int(bool(self.enable_nms)), _lib.ptr(obbs), C.c_void_p(mask.data_ptr()),
                                               _lib.stream_ptr()), "parq_parse_pred")
        out["obbs_pred"] = Obb3D(obbs)
        out["pred_mask"] = mask.bool()
        return out

    @torch.no_grad()
    def update_metrics(self, out_dict, obbs_padded, T_world_local, scene_name=None):
        """model/parq_decoder.py:426-459: boxes + keep-mask of the last iteration (``parse_pred``, on the device), box corners
        in world coordinates, then one step of every tracker (host side, parq_amd/f1_eval.py)."""
        from .loss import parse_target
        from .wrappers import Pose, raw
        assert raw(obbs_padded).ndim == 3, tuple(raw(obbs_padded).shape)
        out = self.parse_pred(out_dict)
        targets = parse_target(obbs_padded, T_world_local)
        obbs = out["obbs_pred"]
        out["scene_name"] = scene_name
        out["pred_corners_world"] = Pose(raw(T_world_local)).transform(obbs.T_world_object.transform(obbs.bb3corners_object))
        for calc in self.metrics_calculator:
            calc.step(out, targets)

    def log_images(self, out_dict, obbs_padded, Ts_world_pseudoCam, Ts_world_local, T_camera_pseudoCam, rgb_img=None, calib_rgb=None,
                   slaml_img=None, calib_slaml=None, slamr_img=None, calib_slamr=None):
        """model/parq_decoder.py:470-538 draws predicted / ground-truth boxes into the input images (cv2, utils/parq_utils.py:108-225)
        for TensorBoard; called from parq_lightning.py:280 (``get_log_images``).  Visualisation is outside this path's scope
        (SURVEY.md §2): same name, same argument list, an empty dict of images, so that a reference-style driver with image logging
        switched on keeps running instead of hitting AttributeError."""
        return {}

    def compute_metrics(self):
        metrics = {}
        for calc in self.metrics_calculator:
            metrics.update(calc.compute_metrics())
        return metrics

    def reset_metrics(self):
        for calc in self.metrics_calculator:
            calc.reset()
