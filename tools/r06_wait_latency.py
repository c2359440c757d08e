"""How long the host takes to notice that a forward has finished (round 6, policy "sync"): the same captured forward launched from an
idle stream and waited for in different ways; wall time launch -> return."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from parq_amd import _lib  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
cfg, W, dec = bench.build_decoder(dev)
dec.range_check = "lazy"
inputs = bench.build_inputs(1, dev, 1000)
h, w = bench.WORKLOAD["feat_hw"]
for _ in range(30):
    dec(*inputs, feat_hw=(h, w))
torch.cuda.synchronize()
entry = next(reversed(dec._ws.values()))
g = next(iter(entry.graphs.values()))
lib = _lib.load()
st = torch.cuda.current_stream()
sp = C.c_void_p(st.cuda_stream)
hip = C.CDLL("libamdhip64.so")
pinned = torch.zeros(4, dtype=torch.int32).pin_memory()
pnp = pinned.numpy()
dflag = torch.ones(4, dtype=torch.int32, device=dev)


def run(wait, n=60):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        time.sleep(0.0005)
        t0 = time.perf_counter()
        lib.parq_graph_launch(g, sp)
        wait()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e6, ts[len(ts) // 10] * 1e6


def w_stream():
    st.synchronize()


def w_device():
    torch.cuda.synchronize()


def w_event():
    ev = torch.cuda.Event()
    ev.record(st)
    ev.synchronize()


def w_event_blocking():
    ev = torch.cuda.Event(blocking=True)
    ev.record(st)
    ev.synchronize()


def w_query():
    ev = torch.cuda.Event()
    ev.record(st)
    while not ev.query():
        pass


def w_stream_query():
    while not st.query():
        pass


def w_pinned_copy():
    pnp[0] = 0
    pinned.copy_(dflag, non_blocking=True)
    while pnp[0] == 0:
        pass


for name, fn in (("stream.synchronize", w_stream), ("torch.cuda.synchronize", w_device), ("event.synchronize", w_event),
                 ("event(blocking).synchronize", w_event_blocking), ("event.query spin", w_query), ("stream.query spin", w_stream_query),
                 ("async D2H of a flag into pinned memory + host spin", w_pinned_copy)):
    med, p10 = run(fn)
    print("%-55s median %.1f us   p10 %.1f us" % (name, med, p10))
