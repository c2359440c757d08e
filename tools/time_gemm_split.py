"""GPU box: speed of the generic split-precision GEMM (raype.hip gemm_split_kernel) and of the fp32 TN GEMM on the shapes a
head-dim-256 attention backward built from them would use (development: sizing a plan)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parq_amd import _lib
lib = _lib.load()
gs = getattr(lib, "_ZN4parq17launch_gemm_splitEPKflPKvS3_S1_PfliiiiS1_iP12ihipStream_t")
gs.restype = C.c_int
gs.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
gt = getattr(lib, "_ZN4parq14launch_gemm_tnEPKflS1_lPfliiiiP12ihipStream_t")
gt.restype = C.c_int
gt.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda", 0)
s = _lib.stream_ptr()


def time_it(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


Lk = 192000
for QP in (256, 2048):
    X = torch.randn(Lk, 256, device=dev)
    Whi = torch.randn(QP, 256, device=dev).half(); Wlo = (torch.randn(QP, 256, device=dev) * 1e-3).half()
    bias = torch.zeros(QP, device=dev)
    Y = torch.empty(Lk, QP, device=dev)
    ms = time_it(lambda: gs(X.data_ptr(), 256, Whi.data_ptr(), Wlo.data_ptr(), bias.data_ptr(), Y.data_ptr(), QP, Lk, QP, 256, 0, None, 1, s))
    print("S^T-shape  M=%d N=%d K=256: %.3f ms = %.0f TFLOP/s algorithmic" % (Lk, QP, ms, 2.0 * Lk * QP * 256 / ms / 1e9))
    W2hi = torch.randn(256, QP, device=dev).half(); W2lo = (torch.randn(256, QP, device=dev) * 1e-3).half()
    b2 = torch.zeros(256, device=dev)
    Y2 = torch.empty(Lk, 256, device=dev)
    ms = time_it(lambda: gs(Y.data_ptr(), QP, W2hi.data_ptr(), W2lo.data_ptr(), b2.data_ptr(), Y2.data_ptr(), 256, Lk, 256, QP, 0, None, 1, s))
    print("dV-shape   M=%d N=256 K=%d: %.3f ms = %.0f TFLOP/s algorithmic" % (Lk, QP, ms, 2.0 * Lk * QP * 256 / ms / 1e9))
    G = torch.zeros(QP, 256, device=dev)
    ms = time_it(lambda: gt(Y.data_ptr(), QP, X.data_ptr(), 256, G.data_ptr(), 256, Lk, QP, 256, 0, s))
    print("dQ-shape (fp32 TN) rows=%d N=%d K=256: %.3f ms = %.0f TFLOP/s" % (Lk, QP, ms, 2.0 * Lk * QP * 256 / ms / 1e9))
    del X, Y, Y2
