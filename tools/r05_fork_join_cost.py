"""What one fork + join between two HIP streams costs on this stack (round 2 measured ~100 us per forward for an event fork/join):
three tiny dependent kernels on one stream against the middle one on a second stream behind an event, joined by another event."""
import time, torch
x = torch.zeros(1024, device="cuda"); y = torch.zeros(1024, device="cuda")
A, B = torch.cuda.Stream(), torch.cuda.Stream()
e1, e2 = torch.cuda.Event(), torch.cuda.Event()

def serial(n):
    with torch.cuda.stream(A):
        for _ in range(n):
            x.add_(1.0); y.add_(1.0); x.add_(1.0)

def forked(n):
    for _ in range(n):
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

for fn in (serial, forked):
    fn(200); torch.cuda.synchronize()
for rep in range(3):
    for name, fn in (("serial", serial), ("forked", forked)):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(2000); torch.cuda.synchronize()
        print("%s: %.2f us per group of three kernels" % (name, (time.perf_counter() - t0) / 2000 * 1e6), flush=True)
