"""parq_amd.PARQ.forward (ray-PE + tokenisation + decoder, model/parq_lightning.py:68-95) at BASELINE cfg 3, one scene per call: one at a
time against parq_amd.InFlight with two forwards in flight.  Feature maps resident in HBM (the backbone is outside the path)."""
import os, sys, time, torch
from types import SimpleNamespace as NS
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parq_amd import PARQ, Camera, Pose, InFlight, synth
torch.set_grad_enabled(False)
V, h, w, C, Q, I = 10, 120, 160, 256, 256, 8
dcfg = synth.decoder_cfg(dim=C, queries=Q, heads=4, ffn=768, layers=I)
cfg = NS(MODEL=NS(TOKENIZER=NS(OUT_CHANNELS=C, RAY_POINTS_SCALE=dcfg.TRANSFORMER.SCALE, NUM_SAMPLES=64, MIN_DEPTH=0.25, MAX_DEPTH=5.25), DECODER=dcfg))
model = PARQ(cfg).eval()
W, Wp = synth.make_decoder_weights(dcfg, 2024), synth.make_ray_pe_weights(C, 7)
sd = model.state_dict()
for k in sd:
    if k.startswith("box3d_decoder."):
        sd[k] = torch.from_numpy(W[k[len("box3d_decoder."):].replace("parq_module.decoder.mlp_heads.", "mlp_heads.")]).reshape(sd[k].shape)
    else:
        sd[k] = torch.from_numpy(Wp[k[len("add_ray_pe."):]])
model.load_state_dict(sd, strict=True)
model = model.cuda()

def batch(seed):
    cam, T_cp, T_wp, T_wl = (torch.from_numpy(x).cuda() for x in synth.make_geometry(seed, 1, V, h, w))
    return {"all_features": torch.randn(1, V, C, h, w, device="cuda") * 0.5, "camera_feature": Camera(cam), "T_camera_pseudoCam": Pose(T_cp),
            "T_world_pseudoCam": Pose(T_wp), "T_world_local": Pose(T_wl)}
batches = [batch(8), batch(9)]
runner = InFlight(model, depth=2)

def serial(n):
    for i in range(n):
        model(dict(batches[i & 1]), 0)

def in_flight(n):
    t = []
    for i in range(n):
        t.append(runner.submit(dict(batches[i & 1]), 0))
        if len(t) == 2:
            t.pop(0).result()
    for x in t:
        x.result()

for fn in (serial, in_flight):
    fn(30); torch.cuda.synchronize()
for rep in range(3):
    for name, fn in (("one at a time", serial), ("two in flight (InFlight)", in_flight)):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(200); torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 200 * 1e3
        print("PARQ.forward cfg3, %s: %.4f ms per scene = %.0f scenes/s = %.0f decoder-iterations/s" % (name, ms, 1e3 / ms, I * 1e3 / ms), flush=True)
