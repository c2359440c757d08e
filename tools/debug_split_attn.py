import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from parq_amd import synth, _lib
from gpu_util import dev, lib, sptr
B, H, Lq, Lk = [int(x) for x in sys.argv[1:5]]
peak = len(sys.argv) > 5
dh, Cn = 64, H * 64
q = synth.normal(1, "q", (B, Lq, Cn)); k = synth.normal(2, "k", (B, Lk, Cn), std=2.0); v = synth.normal(3, "v", (B, Lk, Cn), std=3.0)
if peak: k[0, 0, :dh] = 3.0 * q[0, 0, :dh]
nbytes = lib().parq_k_attention_split_scratch_bytes(B, H, Lq, Lk)
scratch = torch.empty(nbytes // 4 + 1, device="cuda"); out = torch.empty(B, Lq, Cn, device="cuda")
dq, dk, dv = dev(q), dev(k), dev(v)
_lib.check(lib().parq_k_attention_split(_lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(out), B, H, Lq, Lk, _lib.ptr(scratch), nbytes, sptr()), "x")
tq, tk, tv = (torch.from_numpy(x).double().view(B, -1, H, dh).transpose(1, 2) for x in (q, k, v))
S = tq @ tk.transpose(-1, -2) / dh ** 0.5
want = (torch.softmax(S, -1) @ tv).transpose(1, 2).reshape(B, Lq, Cn)
err = (out.cpu().double() - want).abs().view(B, Lq, H, dh).amax(-1)
print("max err", err.max().item(), "at (b,q,h)", np.unravel_index(err.argmax().item(), err.shape))
print("err per query (head0, b0) top:", torch.topk(err[0, :, 0], min(5, Lq)))
Sl = S * 1.4426950408889634
print("log2-score max per query head0:", Sl[0, 0].amax(-1)[:8], "first-stage(64 keys) max:", Sl[0, 0, :, :64].amax(-1)[:8])
bq = np.unravel_index(err.argmax().item(), err.shape)
b_, q_, h_ = [int(x) for x in bq]
sl = Sl[b_, h_, q_]
top = torch.topk(sl, 6)
print("query", q_, "head", h_, "top log2 scores", top.values.numpy().round(2), "at keys", top.indices.numpy())
for st in range(0, Lk, 64):
    pass
stage_max = sl[: (Lk // 64) * 64].view(-1, 64).amax(-1)
run = torch.cummax(stage_max, 0).values
jumps = (stage_max[1:] > run[:-1]).nonzero().flatten() + 1
print("stage maxima jumps at stages", jumps.numpy()[:20], "values", stage_max[jumps].numpy().round(2)[:20], "first", stage_max[0].item())
o = out.cpu().double().view(B, Lq, H, dh)[b_, q_, h_]; w = want.view(B, Lq, H, dh)[b_, q_, h_]
print("out", o[:6].numpy(), "\nwant", w[:6].numpy(), "\nratio", (o / w)[:6].numpy())
# ---- check the per-split partials of the worst (b,h,q)
nblk = (Lk + 31) // 32; nst = (nblk + 1) // 2
base = B * H * ((Lq + 255) // 256); ns = min(max(-(-256 // base), 1), nst, 256)
lp = (Lq + 31) // 32 * 32
cache_bytes = B * H * nblk * 16384
fl = scratch.view(torch.uint8)
off = 256 + cache_bytes
opart = fl[off: off + B * H * ns * 64 * lp * 4].view(torch.float32).view(B * H, ns, 64, lp).cpu().double()
off += B * H * ns * 64 * lp * 4
mpart = fl[off: off + B * H * ns * lp * 4].view(torch.float32).view(B * H, ns, lp).cpu().double()
off += B * H * ns * lp * 4
lpart = fl[off: off + B * H * ns * lp * 4].view(torch.float32).view(B * H, ns, lp).cpu().double()
bh = b_ * H + h_
worst = 0
for s_ in range(ns):
    t0, t1 = s_ * nst // ns, (s_ + 1) * nst // ns
    k0, k1 = t0 * 64, min(t1 * 64, Lk)
    sc = sl[k0:k1]
    m = mpart[bh, s_, q_]
    p = torch.exp2(sc - m)
    l_ref = p.sum(); o_ref = p @ tv[b_, h_, k0:k1]
    el = abs(lpart[bh, s_, q_] - l_ref) / l_ref
    eo = ((opart[bh, s_, :, q_] - o_ref).abs().max() / o_ref.abs().max())
    if max(el, eo) > 1e-5 or s_ < 2:
        print("split", s_, "stages", t0, t1, "m %.4f true max %.4f" % (m, sc.max()), "l rel err %.2e O rel err %.2e" % (el, eo))
bad = (err > 1e-5).nonzero()
print("bad (b,q,h):", bad.tolist()[:40], "count", len(bad))
qq = torch.from_numpy(q).view(B, Lq, H, dh)
for (bb, qi, hh) in bad.tolist()[:6]:
    x = qq[bb, qi, hh] * (1.4426950408889634 / 8.0)
    print((bb, qi, hh), "max|q*scale| %.4f min|.| %.3e" % (x.abs().max(), x.abs().min()), "err %.2e" % err[bb, qi, hh])
