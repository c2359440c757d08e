"""Time one training step (forward_train + backward) of the decoder at BASELINE cfg 3 / cfg 4 per-GPU shape."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parq_amd import PARQDecoder, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
V, h, w, Q, I = 10, 120, 160, 256, 8
C = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg = synth.decoder_cfg(dim=C, queries=Q, heads=4, ffn=768, layers=I, dropout=0.0)
W = synth.make_decoder_weights(cfg, 41, damped=True)
dec = PARQDecoder(cfg)
dec.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=False)
dec = dec.cuda().train()
cam, T_cp, T_wp, T_wl = (torch.from_numpy(x).cuda() for x in synth.make_geometry(42, B, V, h, w))
tokens = torch.randn(B, V * h * w, C, device="cuda")
g = {k: torch.randn(I, B, Q, wd, device="cuda") for k, wd in (("pred_logits", 10), ("center_unnormalized", 3), ("size_unnormalized", 3), ("ortho6d", 6))}
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for it in range(2):
    e[0].record()
    dec.forward_train(tokens, cam, T_cp, T_wp, T_wl, feat_hw=(h, w))
    e[1].record()
    dec.backward(g)
    e[2].record()
    torch.cuda.synchronize()
    print("B=%d forward_train %.2f ms, backward %.2f ms" % (B, e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])))
