"""GPU tier: RCCL itself executes on the lease.  With one GPU the two-rank tests fall back to gloo (RCCL refuses two ranks on one
device), so the collectives the N > 1 paths use — SUM all-reduce of the flat gradient arena (train.py:103 DDP semantics), all-gather
of per-rank outputs, the barrier of bench.py — are run here through backend "nccl" (= RCCL on ROCm) in a one-rank group: library
load, communicator creation and the collective kernels on the device, with the values checked."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

WORKER = r"""
import os, sys, torch, torch.distributed as dist
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
flat = torch.arange(1475000, dtype=torch.float32, device=dev) * 1e-3          # 5.9 MB: the size of the decoder's gradient arena
want = flat.clone()
dist.all_reduce(flat)                                                        # SUM over one rank = identity, through the RCCL kernel
torch.cuda.synchronize()
assert torch.equal(flat, want)
parts = [torch.empty(4, 256, 19, device=dev)]
x = torch.randn(4, 256, 19, device=dev)
dist.all_gather(parts, x)
dist.barrier(device_ids=[0])
torch.cuda.synchronize()
assert torch.equal(parts[0], x)
sys.path.insert(0, sys.argv[1])
from parq_amd import parallel
assert parallel.max_over_ranks(1.25, device=dev) == 1.25
assert parallel.gather_over_ranks(2.5, device=dev) == [2.5]
# the bucketed gradient all-reduce of PARQDecoder.backward through RCCL: two buckets issued on a side stream behind a long
# main-stream kernel that PRODUCES the buffer (ready=None: the side stream waits for the main stream), mean over one rank =
# identity, the main stream joins at the end
big = torch.randn(4096, 4096, device=dev)
for _ in range(8):
    big = (big @ big).clamp_(-1.0, 1.0)
arena = (big.flatten()[:1475000] * 0.5 + 1.0).contiguous()                  # written behind the matrix products
want = arena.clone()
side = torch.cuda.Stream(device=dev)
parallel.all_reduce_mean_buckets_(arena, [(800000, 675000), (0, 800000)], side_stream=side, _force=True)
got = arena * 1.0                                                            # consumer on the main stream
torch.cuda.synchronize()
assert torch.equal(got, want)
dist.destroy_process_group()
print("RCCL_OK", torch.cuda.nccl.version())
"""


@pytest.mark.timeout(300)
def test_rccl_collectives_run_in_a_one_rank_group():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", WORKER, root], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    print("\n" + r.stdout.strip().splitlines()[-1])
