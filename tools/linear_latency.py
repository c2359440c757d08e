import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from parq_amd import _lib
lib = _lib.load()
def run(M, N, K, reps=200, flush=False):
    X = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * K ** -0.5; b = torch.randn(N, device="cuda")
    Y = torch.empty(M, N, device="cuda"); big = torch.empty(128 << 20, device="cuda")
    s = _lib.stream_ptr()
    for _ in range(10): lib.parq_k_linear(_lib.ptr(X), None, _lib.ptr(W), _lib.ptr(b), None, _lib.ptr(Y), M, N, K, 0, s)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    if not flush:
        e0.record()
        for _ in range(reps): lib.parq_k_linear(_lib.ptr(X), None, _lib.ptr(W), _lib.ptr(b), None, _lib.ptr(Y), M, N, K, 0, s)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    tot = 0.0
    for _ in range(20):
        big.fill_(1.0)            # sweep the caches (512 MB)
        e0.record(); lib.parq_k_linear(_lib.ptr(X), None, _lib.ptr(W), _lib.ptr(b), None, _lib.ptr(Y), M, N, K, 0, s); e1.record()
        torch.cuda.synchronize(); tot += e0.elapsed_time(e1)
    return tot / 20 * 1e3
for (M, N, K) in ((256, 256, 256), (256, 768, 256), (256, 256, 768), (256, 256, 384), (1024, 256, 256), (2048, 768, 256)):
    print((M, N, K), "warm back-to-back %.2f us   cold (after cache sweep, incl. event overhead) %.2f us" % (run(M, N, K), run(M, N, K, flush=True)))
