"""Scene-sharded data parallelism for the decoder path.

Scenes are independent (the only cross-query coupling, GroupNorm, is within a scene), so N GPUs
run N processes that each own a contiguous shard of the scenes; inference needs NO data-path
collective.  The helpers here are the whole multi-GPU surface: rendezvous from the torchrun
environment ("nccl" = RCCL on ROCm, "gloo" on CPU-only hosts), shard arithmetic, a barrier
and the max-over-ranks reduction the benchmark contract asks for, plus an all-gather of
per-scene outputs for drivers that want the full batch on every rank.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1 process -> (0, 0, 1))."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: str | None = None, device: torch.device | None = None):
    """Join the process group described by the environment; returns (rank, local_rank, world)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_range(num_scenes: int, rank: int, world: int):
    """Contiguous, balanced shard [lo, hi) of `num_scenes` scenes owned by `rank`
    (the first num_scenes % world ranks get one extra scene)."""
    assert 0 <= rank < world
    base, extra = divmod(num_scenes, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a host scalar (the benchmark's wall time is the slowest rank's)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value: float, device=None):
    """[value of rank 0, ..., value of rank W-1] on every rank (the benchmark prints every rank's step time next to the
    maximum, so that a straggler — a rank on the far socket, a throttling GPU — is visible in the driver's log)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def all_gather_scenes(x: torch.Tensor, num_scenes: int) -> torch.Tensor:
    """Concatenate the per-rank scene shards of `x` (leading dim = local scenes) in rank order.
    Shards may be ragged (shard_range), so they are padded to the largest shard for the collective."""
    if not (dist.is_available() and dist.is_initialized()):
        return x
    world = dist.get_world_size()
    sizes = [shard_range(num_scenes, r, world) for r in range(world)]
    biggest = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((biggest,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    pad[: x.shape[0]] = x
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(out, sizes)], dim=0)


def all_reduce_mean_(flat: torch.Tensor) -> torch.Tensor:
    """Data-parallel gradient averaging of ONE flat buffer (the gradient arena of parq_backward has the layout of the packed
    weight arena, so a training step needs a single collective instead of one bucket per tensor): in place, mean over the
    ranks of the default process group (RCCL over xGMI on GPUs, gloo in the CPU tests); identity without a group."""
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        torch.distributed.all_reduce(flat)
        flat /= torch.distributed.get_world_size()
    return flat


def all_reduce_mean_buckets_(flat: torch.Tensor, buckets, ready=None, side_stream=None, _force=False) -> torch.Tensor:
    """``all_reduce_mean_`` of one flat gradient buffer in BUCKETS: ``buckets`` = [(offset, count), ...] in the order the
    producer finishes them (include/parq_hip.h parq_grad_bucket: bucket 0 is final after phase 1 of parq_backward, half a step
    before the rest).  ``ready(i, stream)`` (GPU: parq_backward_wait_bucket) makes ``stream`` wait, on the device, until bucket i
    is written; each bucket's all-reduce is then issued on ``side_stream`` so that it runs beside what the backward still has
    enqueued on the main stream, and the main stream joins the collectives at the end (what DDP's bucketed, overlapped all-reduce
    does for the reference, train.py:103-108).  Ranges outside the buckets are not touched; empty buckets are skipped.  On CPU
    tensors (the gloo tests) the buckets are reduced one after the other.  Same result as the flat all-reduce: a mean per element."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not _force):
        return flat                                  # (_force: the one-rank RCCL test runs the stream protocol anyway)
    world = dist.get_world_size()
    buckets = [(int(o), int(n)) for o, n in buckets if int(n) > 0]
    if not flat.is_cuda or side_stream is None:
        for i, (o, n) in enumerate(buckets):
            if ready is not None:
                ready(i, None)
            dist.all_reduce(flat[o:o + n])
            flat[o:o + n] /= world
        return flat
    main = torch.cuda.current_stream(flat.device)
    works = []
    for i, (o, n) in enumerate(buckets):
        if ready is not None:
            ready(i, side_stream)                    # device-side wait: the host runs ahead
        else:
            side_stream.wait_stream(main)
        with torch.cuda.stream(side_stream):
            view = flat[o:o + n]
            works.append(dist.all_reduce(view, async_op=True))
            # (RCCL: the collective is ordered behind `side_stream`; gloo on device tensors: the work object completes on wait())
    with torch.cuda.stream(side_stream):
        for w in works:
            w.wait()                                 # RCCL: the CURRENT stream — the side stream — waits for the collective
        for o, n in buckets:                         # (gloo on device tensors: wait() blocks the host until the result is back)
            flat[o:o + n] /= world
    main.wait_stream(side_stream)                    # the main stream joins once, behind the last bucket's mean
    flat.record_stream(side_stream)
    return flat


# ------------------------------------------------------------------------------------------------------------- host placement
def _parse_cpulist(text: str):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_local_cpus(index: int, sysfs: str = "/sys", visible: str | None = None):
    """Host CPUs on the NUMA node of the GPU that HIP enumerates as device ``index`` — WITHOUT touching the GPU runtime: the
    GPU agents appear in the KFD topology (``/sys/class/kfd/kfd/topology/nodes/*/properties``, nodes with simd_count > 0) in the
    order ROCr / HIP enumerate them; ``location_id`` / ``domain`` give the PCI address, whose ``local_cpulist`` is the answer.
    ``visible``: HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES style list of physical indices (default: from the environment).
    Returns [] when the topology cannot be read (containers without /sys/class/kfd)."""
    import glob
    if visible is None:
        visible = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or ""
    gpus = []
    nodes = glob.glob(os.path.join(sysfs, "class/kfd/kfd/topology/nodes/*"))
    for node in sorted(nodes, key=lambda p: int(os.path.basename(p)) if os.path.basename(p).isdigit() else 1 << 30):
        props = {}
        try:
            with open(os.path.join(node, "properties")) as f:
                for line in f:
                    k, _, v = line.strip().partition(" ")
                    props[k] = v
        except OSError:
            continue
        try:
            if int(props.get("simd_count", "0")) <= 0:
                continue
            loc, dom = int(props["location_id"]), int(props.get("domain", "0"))
        except (KeyError, ValueError):
            continue
        gpus.append("%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 0x7))
    try:
        phys = [int(x) for x in visible.split(",") if x.strip() != ""] if visible else list(range(len(gpus)))
        bdf = gpus[phys[index]]
    except (ValueError, IndexError):
        return []
    try:
        with open(os.path.join(sysfs, "bus/pci/devices", bdf, "local_cpulist")) as f:
            return _parse_cpulist(f.read())
    except (OSError, ValueError):
        return []


def gather_objects(obj):
    """Every rank's picklable object, in rank order ([obj] without a process group)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def cpulist_string(cpus) -> str:
    """[0, 1, 2, 5] -> "0-2,5" (the sysfs cpulist form)."""
    cpus = sorted(set(int(c) for c in cpus))
    runs, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        runs.append(str(cpus[i]) if i == j else "%d-%d" % (cpus[i], cpus[j]))
        i = j + 1
    return ",".join(runs)


def pin_to_local_cores(local_rank: int, local_world: int | None = None, sysfs: str = "/sys", gpu_index_of=None):
    """Pin THIS process (and the threads it starts later) to the host cores next to its GPU, in-process
    (``os.sched_setaffinity``; no re-exec, no numactl — call it BEFORE the first GPU call so that the runtime's helper threads
    inherit the mask).  Ranks whose GPUs share a NUMA node split that node's cores among themselves, so eight ranks on a
    two-socket host get disjoint eighths instead of all landing on socket 0 (the launcher's default), where four of them
    would drive their GPUs across the inter-socket link.  ``gpu_index_of``: local rank -> index of the GPU that rank uses (default:
    its own local rank; ranks that share a device — bench.py --share-device — pass ``lambda r: 0`` and split that GPU's node).
    Returns the CPU list it set, or [] if it left the mask alone (no topology information, or the intersection with the allowed
    set is empty)."""
    gpu = gpu_index_of if gpu_index_of is not None else (lambda r: r)
    if not hasattr(os, "sched_setaffinity"):
        return []
    if local_world is None:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    mine = gpu_local_cpus(gpu(local_rank), sysfs)
    if not mine:
        return []
    # ranks that share my NUMA node (same local_cpulist), in rank order: each takes a contiguous slice
    sharing = [r for r in range(max(1, local_world)) if gpu_local_cpus(gpu(r), sysfs) == mine] or [local_rank]
    allowed = sorted(set(mine) & set(os.sched_getaffinity(0)))
    if not allowed:
        return []
    k, n = (sharing.index(local_rank) if local_rank in sharing else 0), len(sharing)
    per = max(1, len(allowed) // n)
    part = allowed[k * per:(k + 1) * per] if k < n - 1 else allowed[k * per:]
    part = part or allowed
    try:
        os.sched_setaffinity(0, part)
    except OSError:
        return []
    return part


def all_reduce_mean_scalars(metrics: dict, device=None) -> dict:
    """Mean over ranks of a dict of host scalars — what Lightning's ``self.log(..., sync_dist=True)`` does with the
    validation metrics (model/parq_lightning.py:133-140).  Non-scalar entries (arrays, images) pass through untouched, as
    the reference skips them.  Ranks may hold DIFFERENT key sets (a rank that saw no valid scene returns fewer metrics): the key
    lists are gathered first, the reduction runs over their sorted union, and a key is averaged over the ranks that hold it —
    a rank without a key neither hangs the collective nor drags the mean towards zero.  Identity without a process group."""
    import numbers
    mine = sorted(k for k, v in metrics.items() if isinstance(v, numbers.Number) and not isinstance(v, bool))
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return dict(metrics)
    gathered = [None] * dist.get_world_size()
    dist.all_gather_object(gathered, mine)
    keys = sorted(set().union(*[set(g) for g in gathered]))
    if not keys:
        return dict(metrics)
    if device is None:
        device = (torch.device("cuda", torch.cuda.current_device())
                  if dist.get_backend() == "nccl" and torch.cuda.is_available() else "cpu")
    have = set(mine)
    t = torch.tensor([[float(metrics[k]) if k in have else 0.0 for k in keys], [1.0 if k in have else 0.0 for k in keys]],
                     dtype=torch.float64, device=device)
    dist.all_reduce(t)
    out = dict(metrics)
    for k, total, count in zip(keys, t[0].tolist(), t[1].tolist()):
        out[k] = total / count                    # count >= 1: the key came from somebody's list
    return out


# ---------------------------------------------------------------------------------------------------- intra-scene view sharding
# The two merges of a view-sharded scene (include/parq_hip.h parq_iterate_sharded; PARQDecoder.forward_view_sharded) as plain
# tensor arithmetic on any device.  The GPU path performs them inside the library (sample_finalize / attn_combine kernels) after the
# same collectives; these host-level forms define the protocol and are what the CPU test runs over gloo with the oracle as compute.

def merge_sample_sums(sums: torch.Tensor, counts: torch.Tensor, group=None) -> torch.Tensor:
    """View mean of the sampled features over ALL views of a scene from this rank's undivided sums (B,Q,C) and valid-view counts
    (B,Q): SUM all-reduce of both, then sum / max(count, 1) (model/transformer_parq.py:157-160)."""
    sums, counts = sums.clone(), counts.clone().to(sums.dtype)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(sums, group=group)
        dist.all_reduce(counts, group=group)
    return sums / torch.where(counts > 0, counts, torch.ones_like(counts)).unsqueeze(-1)


def merge_attention_shards(out: torch.Tensor, lse: torch.Tensor, group=None) -> torch.Tensor:
    """Attention output over all keys from per-rank outputs over disjoint key shards: out (B,H,L,dh) normalised within the
    shard, lse (B,H,L) natural-log log-sum-exp of the shard's scores.  All-gather, then the softmax-weighted combination
    sum_r exp(lse_r - max) out_r / sum_r exp(lse_r - max)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return out
    world = dist.get_world_size(group)
    outs = [torch.empty_like(out) for _ in range(world)]
    lses = [torch.empty_like(lse) for _ in range(world)]
    dist.all_gather(outs, out.contiguous(), group=group)
    dist.all_gather(lses, lse.contiguous(), group=group)
    L = torch.stack(lses)                                   # (R,B,H,L)
    wgt = torch.exp(L - L.max(dim=0, keepdim=True).values)
    return (torch.stack(outs) * wgt.unsqueeze(-1)).sum(0) / wgt.sum(0).unsqueeze(-1)


def view_shard(num_views: int, rank: int, world: int):
    """Views [lo, hi) of a scene owned by ``rank`` (contiguous, balanced; every rank needs at least one view)."""
    if num_views < world:
        raise ValueError("view sharding needs at least one view per rank (%d views, %d ranks)" % (num_views, world))
    return shard_range(num_views, rank, world)
