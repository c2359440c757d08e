"""What one fork + join between two HIP streams costs on this stack (round 2 measured ~100 us per forward for an event fork/join).
Three dependent kernels on one stream against the middle one on a second stream behind an event, joined by another event — with tiny
kernels (the host-side cost of the event calls shows) and with ~50 us kernels (the host runs ahead: what is left is the device-side
latency of the two cross-stream dependencies)."""
import time, torch
A, B = torch.cuda.Stream(), torch.cuda.Stream()
e1, e2 = torch.cuda.Event(), torch.cuda.Event()

def make(n):
    x = torch.zeros(n, device="cuda"); y = torch.zeros(n, device="cuda")
    def serial(k):
        with torch.cuda.stream(A):
            for _ in range(k):
                x.add_(1.0); y.add_(1.0); x.add_(1.0)
    def forked(k):
        for _ in range(k):
            with torch.cuda.stream(A):
                x.add_(1.0)
                e1.record(A)
            with torch.cuda.stream(B):
                B.wait_event(e1)
                y.add_(1.0)
                e2.record(B)
            with torch.cuda.stream(A):
                A.wait_event(e2)
                x.add_(1.0)
    return serial, forked

for label, n, reps in (("tiny kernels", 1024, 2000), ("~50 us kernels", 64 << 20, 300)):
    serial, forked = make(n)
    for fn in (serial, forked):
        fn(20); torch.cuda.synchronize()
    for rep in range(3):
        res = {}
        for name, fn in (("serial", serial), ("forked", forked)):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(reps); torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0) / reps * 1e6
        print("%s: serial %.2f us, forked %.2f us per group of three kernels: fork + join = %.2f us" % (label, res["serial"], res["forked"], res["forked"] - res["serial"]), flush=True)
