"""Time the ray-point positional encoding + tokenisation (AddRayPE.tokens) at BASELINE cfg 3 size."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parq_amd import AddRayPE, synth, _lib
if any(k.startswith("PARQ_RAYPE") for k in os.environ):       # development switches live in the development library only
    _lib.use_dev_library()
# inference (the default): no graph, the hidden layer is never written; RAYPE_GRAD=1: the autograd forward, which keeps it
torch.set_grad_enabled(os.environ.get("RAYPE_GRAD", "0") == "1")
B, V, h, w, C = 1, 10, 120, 160, 256
pe = AddRayPE(C, synth.DEFAULT_SCALE, 64, 0.25, 5.25)
Wp = synth.make_ray_pe_weights(C, 7)
pe.load_state_dict({k: torch.from_numpy(v) for k, v in Wp.items()}, strict=True)
pe = pe.cuda().eval()
cam, T_cp, T_wp, T_wl = (torch.from_numpy(x).cuda() for x in synth.make_geometry(8, B, V, h, w))
feat = torch.randn(B, V, C, h, w, device="cuda")
for _ in range(3):
    tok = pe.tokens(feat, cam, T_cp, T_wp, T_wl)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(5):                                   # best of five runs of 20 calls
    e0.record()
    for _ in range(20):
        tok = pe.tokens(feat, cam, T_cp, T_wp, T_wl)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20)
print("AddRayPE.tokens cfg3: %.3f ms" % best)
e0.record()
for _ in range(10):
    enc = pe(feat, cam, T_cp, T_wp, T_wl)
e1.record(); torch.cuda.synchronize()
print("AddRayPE.forward (B,V,C,h,w encoding) cfg3: %.3f ms" % (e0.elapsed_time(e1) / 10))
