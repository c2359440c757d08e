"""GPU box: the K/V projection alone (parq_prepare in a loop; `python tools/kvproj_power.py`) or the whole forward
(`... forward`) with rocm-smi sampled beside it — time per call, socket power and shader clock.  Is the path power-limited?
PARQ_KVPROJ_PROBE / PARQ_FLASH_PROBE select ingredient-removed kernels."""
import os, subprocess, sys, threading, time
import torch
torch.set_grad_enabled(False)      # inference tool: the reference's drivers run these calls under no_grad (eval.py:46)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parq_amd import _lib
_lib.use_dev_library()          # the PARQ_*_PROBE switches exist in the development library only
import bench

device = torch.device("cuda", 0)
cfg, W, dec = bench.build_decoder(device)
inputs = bench.build_inputs(1, device, seed=1000)
h, w = bench.WORKLOAD["feat_hw"]
samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append(out)
        except Exception as e:          # noqa
            samples.append("ERR %r" % (e,))
        time.sleep(0.05)


with torch.no_grad():
    dec(*inputs, feat_hw=(h, w))
    for _ in range(20):
        dec.prepare(*inputs, feat_hw=(h, w))
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler)
    th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    what = sys.argv[1] if len(sys.argv) > 1 else "prepare"
    n = int(os.environ.get("N", 20000 if what == "prepare" else 2500))
    e0.record()
    for _ in range(n):
        if what == "prepare":
            dec.prepare(*inputs, feat_hw=(h, w))
        else:
            dec(*inputs, feat_hw=(h, w))
    e1.record()
    torch.cuda.synchronize()
    stop = True
    th.join()
print("%s, kvproj probe %s, flash probe %s: %.1f us per call (%d calls)" % (what, os.environ.get("PARQ_KVPROJ_PROBE", "0"), os.environ.get("PARQ_FLASH_PROBE", "0"), e0.elapsed_time(e1) / n * 1e3, n))
mid = samples[len(samples) // 4: max(len(samples) // 4 + 1, 3 * len(samples) // 4)]
print("samples %d; middle ones:" % len(samples))
for s in mid[:2] + mid[-2:]:
    print(s.strip().split("\n")[-1])
