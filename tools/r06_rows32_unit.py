"""parq_k_linear at K = 1024 under the development switch PARQ_CHAIN_K1024_ROWS32 against a float64 product (round 6)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from parq_amd import _lib  # noqa: E402
_lib.use_dev_library()
lib = _lib.load()
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(5)
for M, N in ((256, 3072), (256, 1024), (64, 2048), (512, 1024)):
    K = 1024
    X = torch.randn(M, K, generator=g).to(dev)
    X2 = torch.randn(M, K, generator=g).to(dev)
    Wt = (torch.randn(N, K, generator=g) / 32).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    R = torch.randn(M, N, generator=g).to(dev)
    for name, x2, r in (("bias", None, None), ("x2+bias", X2, None), ("bias+res", None, R)):
        Y = torch.full((M, N), float("nan"), device=dev)
        rc = lib.parq_k_linear(_lib.ptr(X), _lib.ptr(x2) if x2 is not None else None, _lib.ptr(Wt), _lib.ptr(b), _lib.ptr(r) if r is not None else None,
                               _lib.ptr(Y), M, N, K, 0, _lib.stream_ptr())
        torch.cuda.synchronize()
        want = ((X.double() + (x2.double() if x2 is not None else 0)) @ Wt.double().t() + b.double() + (r.double() if r is not None else 0))
        err = (Y.double() - want).abs().max().item() / want.abs().max().item()
        bad = (~torch.isfinite(Y)).sum().item()
        rows = ((Y.double() - want).abs().amax(dim=1) > 1e-3 * want.abs().max()).nonzero().flatten()[:8].tolist()
        cols = ((Y.double() - want).abs().amax(dim=0) > 1e-3 * want.abs().max()).nonzero().flatten()[:8].tolist()
        print("ROWS32=%s M=%d N=%d %-9s rc=%d rel err %.2e nonfinite %d  first bad rows %s cols %s" % (os.environ.get("PARQ_CHAIN_K1024_ROWS32", "0"), M, N, name, rc, err, bad, rows, cols))
