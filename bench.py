#!/usr/bin/env python3
"""Benchmark of the PARQ recurrent decoder path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scenes-per-gpu B]

metric  : decoder-iterations/sec = scenes x recurrent iterations / wall time of
          PARQDecoder.forward (prologue incl. the hoisted K/V projection + I iterations)
workload: BASELINE cfg 3 — 10 views of 480x640 images -> 120x160 feature maps (stride 4),
          256 queries, 8 iterations, d=256, 4 heads, FFN 768, fp32, synthetic features,
          random-init weights.  One "step" = one forward over the rank's batch of scenes.
N > 1   : one process per GPU, scenes sharded data-parallel (independent, no data-path collective:
          SURVEY.md §8e); weak scaling, value = all ranks' iterations / max time.  `python bench.py --gpus N`
          works as typed: without a torchrun environment the parent starts N fresh ranks through
          `python -m torch.distributed.run` BEFORE it touches the GPU, forwards rank 0's JSON line and
          exits with the launcher's code; under torchrun (the driver's form) it is a rank.

Also reported in the same JSON line:
  roofline      dominant kernel (flash_split_pipe_kernel, the dense cross-attention): algorithmic FLOP per launch
                (4*Q*N*C per scene) / mean launch time from hipEvents recorded by the library
                on the launch stream during an instrumented repeat of the same K steps
                (split mode: fp16 hi/lo 3-term products on the fp16 matrix pipe, fp32 accumulation)
  roofline_project_sample   HBM-bound gather: algorithmic bytes (4*V*Q*C + Q*C)*4 per scene
  cpu_baseline  the oracle in reference_ops mode (same ATen op sequence as the reference)
                on the host cores, bounded sample (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOAD = dict(views=10, image_hw=(480, 640), feat_hw=(120, 160), queries=256, iters=8, dim=256, heads=4, ffn=768)
# --config: the other single-GPU configurations of BASELINE.json as driver-reproducible lines (same JSON, same roofline and
# bounded cpu_baseline objects).  cfg3 is the configuration the metric is quoted on (the default, the headline); cfg2 / cfg5 name
# a reduced-precision arithmetic (bf16 / fp16), which selects the single-product attention kernels — NOT the headline number.
CONFIGS = {
    "cfg3": dict(views=10, image_hw=(480, 640), feat_hw=(120, 160), queries=256, iters=8, mode=None,
                 text="BASELINE cfg3: 10 views 480x640 (feature maps 120x160, N=192000 tokens), 256 queries, 8 iterations"),
    "cfg2": dict(views=5, image_hw=(480, 640), feat_hw=(120, 160), queries=128, iters=4, mode="bf16",
                 text="BASELINE cfg2: 5 views 480x640 (feature maps 120x160, N=96000 tokens), 128 queries, 4 iterations, bf16 cross-attention"),
    "cfg5": dict(views=20, image_hw=(960, 1280), feat_hw=(240, 320), queries=512, iters=12, mode="fp16",
                 text="BASELINE cfg5: 20 views 960x1280 (feature maps 240x320, N=1536000 tokens), 512 queries, 12 iterations, fp16 cross-attention"),
    # the geometry the reference SHIPS (config/train.yaml:16-56: 3 frames per snippet of 240x320 images, DEC_DIM 1024 / 4 heads,
    # FFN 768, 256 queries, 8 iterations): N = 14 400 tokens, so the forward is the chain of small dependent launches — a LATENCY line
    "shipped": dict(views=3, image_hw=(240, 320), feat_hw=(60, 80), queries=256, iters=8, mode=None, dim=1024,
                    text="reference's shipped geometry (config/train.yaml): 3 views 240x320 (feature maps 60x80, N=14400 tokens), 256 queries, 8 iterations"),
}
PEAK_F32_MATRIX_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
PEAK_F16_MATRIX_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense fp16/bf16 MFMA peak
SPLIT_PASSES = 3                    # fp16 MFMAs issued per fp32-accurate product (hi*hi + hi*lo + lo*hi)
PEAK_HBM_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E spec peak
PREWARM_STEPS = 24                 # untimed forwards before the warm-up steps (see main): ~50 ms of device spin-up


def build_inputs(B, device, seed):
    from parq_amd import synth
    V, (h, w), C = WORKLOAD["views"], WORKLOAD["feat_hw"], WORKLOAD["dim"]
    cam, T_cp, T_wp, T_wl = synth.make_geometry(seed, B, V, h, w)
    g = torch.Generator(device=device).manual_seed(seed)
    tokens = torch.randn(B, V * h * w, C, device=device, generator=g)
    dev = lambda a: torch.from_numpy(a).to(device)
    return tokens, dev(cam), dev(T_cp), dev(T_wp), dev(T_wl)


def build_decoder(device, wq_scale=1.0):
    from parq_amd import synth
    from parq_amd.decoder import PARQDecoder
    cfg = synth.decoder_cfg(dim=WORKLOAD["dim"], queries=WORKLOAD["queries"], heads=WORKLOAD["heads"],
                            ffn=WORKLOAD["ffn"], layers=WORKLOAD["iters"])
    W = synth.make_decoder_weights(cfg, seed=2024)
    if wq_scale != 1.0:                  # sharpened cross-attention: the query rows of the in-projection (transformer_parq.py:377-380)
        key = "parq_module.decoder.layers.0.multihead_attn.in_proj_weight"
        wq = W[key].copy()
        wq[:WORKLOAD["dim"]] *= wq_scale
        W[key] = wq
    dec = PARQDecoder(cfg).eval()
    sd = dec.state_dict()
    for k in sd:
        sd[k] = torch.from_numpy(W[k.replace("parq_module.decoder.mlp_heads.", "mlp_heads.")]).reshape(sd[k].shape)
    dec.load_state_dict(sd, strict=True)
    return cfg, W, dec.to(device)


def cpu_baseline(cfg, W, inputs, min_seconds=12.0, max_iters=24, view_limit=None):
    """The reference's op sequence on the host cores (oracle, reference_ops=True): time
    recurrent iterations of scene 0 of the SAME workload until `min_seconds` of CPU work has
    been measured (the reference hoists nothing, so its cost is linear in the iteration count).
    `view_limit` (cfg 5: the materialised scores of all 20 views are 2 x 12.6 GB per iteration): only the first `view_limit`
    views are given to the CPU and the rate is scaled by view_limit / V — every per-iteration cost of the reference that
    matters (K/V projection, scores, softmax, head mean, PV) is linear in the number of keys; said in `sample`."""
    import copy
    from parq_amd import synth
    from oracle import parq_oracle as O
    cfg2 = copy.deepcopy(cfg)
    tokens, cam, T_cp, T_wp, T_wl = [t[:1].cpu() for t in inputs]
    V_all = cam.shape[1]
    scale = 1.0
    if view_limit is not None and view_limit < V_all:
        n_per_view = tokens.shape[1] // V_all
        tokens, cam, T_cp, T_wp = tokens[:, :view_limit * n_per_view], cam[:, :view_limit], T_cp[:, :view_limit], T_wp[:, :view_limit]
        scale = view_limit / V_all
    od = O.OracleDecoder(cfg2, W, synth.SCANNET_MEAN_SIZES, reference_ops=True)
    with torch.no_grad():
        od.prepare(tokens, cam, T_cp, T_wp, T_wl)
        ref = od.initial_ref()
        _, ref, _ = od.iterate(ref, 0)                     # warm-up (thread pools, allocator)
        t0 = time.perf_counter()
        budget_iters = 0
        while budget_iters < max_iters and (budget_iters < 2 or time.perf_counter() - t0 < min_seconds):
            _, ref, _ = od.iterate(ref, budget_iters % cfg.TRANSFORMER.DEC_LAYERS)
            budget_iters += 1
        dt = time.perf_counter() - t0
    return {"value": budget_iters / dt * scale, "unit": "decoder-iterations/sec", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": "%d recurrent iterations of 1 scene of the same workload (reference op sequence: per-iteration "
                      "K/V projection, materialised softmax, head-averaged weights), %.1f s, host has %d logical CPUs%s"
                      % (budget_iters, dt, os.cpu_count() or 0,
                         "" if scale == 1.0 else "; only %d of the %d views were given to the CPU (the materialised scores of all of them do "
                         "not fit a bounded sample) and the measured rate was scaled by %d/%d: the reference's per-iteration cost is "
                         "linear in the key count" % (view_limit, V_all, view_limit, V_all))}


def peaked_workload(device, inputs, h, w, steps, wq_scale=4.0):
    """The number a model with NON-diffuse attention gets: the same workload with the cross-attention query projection
    x 4 — rows that rest on a handful of keys, as trained detectors produce (xavier weights on white noise give
    near-uniform rows).  Default policy: the module's first forward is checked synchronously and re-run with the flagged heads on the
    fp16 x 3 tier (PARQDecoder.safe_heads); timed after the tiers have settled."""
    import warnings
    cfg, W, dec = build_decoder(device, wq_scale=wq_scale)
    B, I = inputs[0].shape[0], WORKLOAD["iters"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        settle = 0
        for settle in range(1, 9):                                   # the first call settles every head iteration 0 flags; later ones what is left
            before = dec.safe_heads
            out = dec(*inputs, feat_hw=(h, w))
            torch.cuda.synchronize()
            if dec.safe_heads == before and not dec.attention_too_peaked():
                break
        for _ in range(PREWARM_STEPS // 2):
            dec(*inputs, feat_hw=(h, w))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = dec(*inputs, feat_hw=(h, w))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        finite = all(bool(torch.isfinite(v).all()) for o in out for v in o.values())
        tripped_after = bool(dec.attention_too_peaked())
        nh = WORKLOAD["heads"]
        safe = dec.safe_heads
        # the error the guard avoided: the same forward with every head forced onto the fast tier, against mode "split" (iteration 0)
        ref = [{k: v.double().clone() for k, v in o.items()} for o in dec(*inputs, feat_hw=(h, w))][:1]
        dec.range_check = "off"
        dec.safe_heads = 0
        raw = dec(*inputs, feat_hw=(h, w))[:1]
        lmin = dec.attention_min_row_sum()
        dec.attention_mode = "split"
        strict = dec(*inputs, feat_hw=(h, w))[:1]
        torch.cuda.synchronize()
        err = lambda a, b: max(float(((x[k].double() - y[k].double()).abs() / y[k].double().abs().clamp(min=1.0)).max()) for x, y in zip(a, b) for k in x)
        e_tiers, e_raw = err(ref, strict), err(raw, strict)
    dec._ws.clear()
    del dec
    torch.cuda.empty_cache()
    return {"workload": "the timed workload with the cross-attention query projection x %g (peaked rows)" % wq_scale,
            "value": B * I * steps / dt, "unit": "decoder-iterations/sec", "ms_per_step": dt / steps * 1e3,
            "safe_heads": "0b" + format(safe, "0%db" % nh), "heads_on_the_fast_tier": nh - bin(safe).count("1"),
            "forwards_until_the_tiers_settled": settle, "outputs_finite": finite, "guard_tripped_in_the_timed_forwards": tripped_after,
            "smallest_row_probability_sum": lmin,
            "first_iteration_max_difference_to_mode_split": {"with_the_guard": e_tiers, "every_head_forced_onto_the_fast_tier": e_raw}}


def project_sample_b32(dec, device, h, w, scenes=32, steps=2):
    """The HBM-bound gather at a size that does not fit the 256 MB Infinity Cache (BASELINE.md §3/§4 ask for B = 1 AND B = 32):
    `scenes` synthetic scenes of the same workload (32 x 197 MB of tokens = 6.3 GB) through the same PARQDecoder.forward, the
    project+sample launches timed by the library's hipEvents on the launch stream.  Algorithmic bytes per launch =
    (4*V*Q*C + Q*C)*4 per scene."""
    V, Q, C = WORKLOAD["views"], WORKLOAD["queries"], WORKLOAD["dim"]
    inputs = build_inputs(scenes, device, seed=4242)
    dec(*inputs, feat_hw=(h, w))                                   # warm-up (allocates the 32-scene workspace)
    dec.profile_enable(True)
    for _ in range(steps):
        dec(*inputs, feat_hw=(h, w))
    torch.cuda.synchronize()
    prof = dec.profile_read()
    dec.profile_enable(False)
    ms, n = prof["project_sample"]
    fw_ms = sum(v[0] for v in prof.values()) / steps
    bytes_per_launch = (4.0 * V * Q * C + Q * C) * 4.0 * scenes
    gbs = bytes_per_launch / (ms / n * 1e-3) / 1e9
    del inputs
    dec._ws.clear()
    torch.cuda.empty_cache()
    return {"bound": "hbm", "kernel": "project_sample_kernel", "scenes": scenes, "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": gbs / PEAK_HBM_GBS, "traffic": pmc_traffic("project_sample_kernel", scenes),
            "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": ms / n, "launches": n,
            "forward_kernel_ms_at_%d_scenes" % scenes: fw_ms,
            "note": "tokens of %d scenes = %.1f GB (past the 256 MB Infinity Cache): a bandwidth figure; the B = 1 entry "
                    "(roofline_project_sample) is latency-bound.  `achieved` / `frac` are ALGORITHMIC bytes per second (BASELINE.md's "
                    "definition of the >= 40 %% target); `traffic` (counter bytes) is about half of them: a query's four corner rows "
                    "and neighbouring queries' texels are served again from L2 / the Infinity Cache, so the HBM side moves ~0.3 of peak"
                    % (scenes, scenes * V * h * w * C * 4 / 1e9)}


def _amdgpu_sysfs(index):
    """(hwmon power file, shader-clock file) of the amdgpu device behind cuda:`index`, or (None, None).  Plain sysfs reads: nothing
    is spawned (a child started from a GPU-initialised process under rocprofv3 --pmc inherits the profiler preload)."""
    import glob
    cards = []
    for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        try:
            with open(os.path.join(d, "vendor")) as f:
                if f.read().strip() != "0x1002":
                    continue
        except OSError:
            continue
        cards.append(d)
    if not cards:
        return None, None
    d = None
    try:                                        # the card whose PCI address is the HIP device's
        pr = torch.cuda.get_device_properties(index)
        bdf = "%04x:%02x:%02x." % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        for c in cards:
            if bdf in os.path.realpath(c):
                d = c
                break
    except Exception:                           # noqa: properties without PCI fields
        d = None
    if d is None:
        visible = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or ""
        try:
            phys = int(visible.split(",")[index]) if visible else index
        except (ValueError, IndexError):
            phys = index
        d = cards[phys] if phys < len(cards) else cards[0]
    power = None
    for name in ("power1_average", "power1_input"):
        hits = glob.glob(os.path.join(d, "hwmon", "hwmon*", name))
        if hits:
            power = hits[0]
            break
    freq = glob.glob(os.path.join(d, "hwmon", "hwmon*", "freq1_input"))          # current shader clock in Hz
    sclk = freq[0] if freq else os.path.join(d, "pp_dpm_sclk")
    return power, (sclk if os.path.exists(sclk) else None)


def device_state_under_load(step, seconds=1.0, index=0):
    """Socket power and shader clock sampled from sysfs (hwmon power1_average, pp_dpm_sclk) while the forward loops back to back —
    the two big kernels of this path run at the 1400 W cap with the clock throttled below 2.4 GHz (DESIGN.md section 4,
    profiles/r02_power_rocm_smi.txt), which is what `roofline.frac` against the nominal peak has to be read with.  Best effort:
    null when the files are not readable.  In-process reads only — no child process is ever started here."""
    import threading
    power_f, sclk_f = _amdgpu_sysfs(index)
    if power_f is None and sclk_f is None:
        return None
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            mhz = watts = None
            try:
                if sclk_f and sclk_f.endswith("freq1_input"):
                    with open(sclk_f) as f:
                        mhz = float(f.read().strip()) * 1e-6
                elif sclk_f:
                    with open(sclk_f) as f:
                        for line in f:
                            if "*" in line:
                                mhz = float(line.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
                if power_f:
                    with open(power_f) as f:
                        watts = float(f.read().strip()) * 1e-6
            except (OSError, ValueError, IndexError):
                pass
            if mhz is not None or watts is not None:
                samples.append((mhz, watts))
            time.sleep(0.02)

    th = threading.Thread(target=sampler)
    th.start()
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        for _ in range(10):
            step()
        torch.cuda.synchronize()
    stop[0] = True
    th.join()
    late = samples[len(samples) // 2:]
    if not late:
        return None
    mean = lambda xs: (sum(xs) / len(xs)) if xs else None
    return {"sclk_mhz": mean([x[0] for x in late if x[0] is not None]), "socket_power_w": mean([x[1] for x in late if x[1] is not None]),
            "samples": len(late), "source": "sysfs (%s, %s), read in-process" % (os.path.basename(sclk_f or "-"), os.path.basename(power_f or "-")), "sysfs_device": os.path.realpath(os.path.dirname(power_f or sclk_f)),
            "note": "sampled while the forward loops back to back; nominal shader clock 2400 MHz, socket cap 1400 W"}


def _pmc_record():
    """The newest committed rocprofv3 PMC record (profiles/rNN_pmc.json, written by tools/collect_profiles.sh +
    tools/make_pmc_json.py), or (None, None)."""
    import glob
    paths = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r[0-9][0-9]_pmc.json")))
    for path in reversed(paths):
        try:
            with open(path) as f:
                return json.load(f), os.path.relpath(path, os.path.dirname(os.path.abspath(__file__)))
        except Exception:      # noqa: unreadable record -> try the previous round's
            continue
    return None, None


_LIVE_PMC = {"by_kernel": None, "scenes": None, "note": None}


def live_pmc(args, B):
    """HBM-side bytes of THIS command measured in THIS run (the line could not see a traffic regression): after the timed
    region rank 0 starts two short child processes of this script under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate
    passes, --kernel-trace only, as MI355X_MICROARCH.md prescribes; the program itself behind `--`), each running a few forwards of the
    same configuration (`--pmc-child`), and reads the per-kernel means from their counter CSVs.  Any failure (no rocprofv3, a timeout)
    leaves the recorded figures of profiles/rNN_pmc.json in place; `roofline.traffic_source` says which it was."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    tool = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(tool):
        _LIVE_PMC["note"] = "rocprofv3 not found"
        return
    if under_profiler():
        # this process is itself running under rocprofv3 (tools/collect_profiles.sh): a nested profiler would inherit the outer one's
        # preload (the GPU is then initialised in the launcher that has to start the child), and hardware counters are exclusive
        _LIVE_PMC["note"] = "bench.py is itself running under a profiler: no nested rocprofv3 passes"
        return
    here = os.path.abspath(__file__)
    base = [sys.executable, here, "--pmc-child", "--steps", "2", "--warmup", "1", "--config", args.config, "--scenes-per-gpu", str(B)]
    if args.attention_mode:
        base += ["--attention-mode", args.attention_mode]
    res = {}
    t0 = time.perf_counter()
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            with tempfile.TemporaryDirectory(dir="/tmp") as d:
                env = scrubbed_env(TMPDIR="/tmp")
                r = subprocess.run([tool, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--"] + base,
                                   cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
                if r.returncode != 0:
                    _LIVE_PMC["note"] = "rocprofv3 --pmc %s exited with %d" % (ctr, r.returncode)
                    return
                acc = {}
                for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                    with open(path) as f:
                        for row in csv.DictReader(f):
                            if row.get("Counter_Name") == ctr:
                                acc.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
                for k, v in acc.items():
                    res.setdefault(k, {})[ctr] = sum(v) / len(v)
    except Exception as exc:      # noqa: BLE001 — a measurement extra must not end the bench
        _LIVE_PMC["note"] = "live PMC pass failed: %s" % type(exc).__name__
        return
    _LIVE_PMC.update(by_kernel=res, scenes=B, note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this command in this run (two child passes, %.0f s)" % (time.perf_counter() - t0))


_PROFILER_ENV = ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_", "ROCPROF_", "ROCP_", "HSA_TOOLS_LIB", "ROCTRACER_", "ROCTX_")


def under_profiler():
    """True when this process was started under rocprofv3 / rocprof (their tool library is preloaded or named in the environment)."""
    if any("rocprof" in v.lower() or "roctracer" in v.lower() for v in (os.environ.get("LD_PRELOAD", ""), os.environ.get("HSA_TOOLS_LIB", ""))):
        return True
    return any(k.startswith(_PROFILER_ENV) for k in os.environ)


def scrubbed_env(**extra):
    """A copy of the environment without anything a profiler put there (for child processes that get a profiler of their own)."""
    env = {k: v for k, v in os.environ.items() if not k.startswith(_PROFILER_ENV)}
    pre = [x for x in env.get("LD_PRELOAD", "").replace(":", " ").split() if "rocprof" not in x.lower() and "roctracer" not in x.lower()]
    if pre:
        env["LD_PRELOAD"] = ":".join(pre)
    else:
        env.pop("LD_PRELOAD", None)
    env.update(extra)
    return env


def pmc_traffic(kernel, scenes):
    """HBM bytes per launch of `kernel`: FETCH_SIZE x 2 + WRITE_SIZE (KB counters; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    gfx950 wide streaming reads) — measured in this run where `live_pmc` succeeded for this scene count, else the RECORDED figure of the
    newest committed passes (profiles/rNN_pmc.json); `traffic_source` in the line says which, or null."""
    live = _LIVE_PMC["by_kernel"]
    if live and _LIVE_PMC["scenes"] == scenes:
        f = [v["FETCH_SIZE"] for k, v in live.items() if kernel in k and "FETCH_SIZE" in v]
        w = [v["WRITE_SIZE"] for k, v in live.items() if kernel in k and "WRITE_SIZE" in v]
        if f and w:
            return (sum(f) / len(f) * 2.0 + sum(w) / len(w)) * 1024.0
    rec, _ = _pmc_record()
    try:
        e = rec["by_scenes"][str(scenes)][kernel]
        return (e["fetch_kb"] * e.get("fetch_correction", 2.0) + e["write_kb"]) * 1024.0
    except Exception:
        return None


def ray_pe_timing(B, device):
    """Not part of the metric (PARQDecoder.forward): the once-per-forward AddRayPE + tokenisation that precedes the
    decoder in PARQ.forward (model/parq_lightning.py:70-85), timed at the same workload for the end-to-end picture."""
    from parq_amd import AddRayPE, synth
    V, (h, w), C = WORKLOAD["views"], WORKLOAD["feat_hw"], WORKLOAD["dim"]
    pe = AddRayPE(C, synth.DEFAULT_SCALE, 64, 0.25, 5.25)
    Wp = synth.make_ray_pe_weights(C, 7)
    pe.load_state_dict({k: torch.from_numpy(v) for k, v in Wp.items()}, strict=True)
    pe = pe.to(device).eval()
    cam, T_cp, T_wp, T_wl = (torch.from_numpy(x).to(device) for x in synth.make_geometry(8, B, V, h, w))
    feat = torch.randn(B, V, C, h, w, device=device)
    for _ in range(2):
        pe.tokens(feat, cam, T_cp, T_wp, T_wl)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        pe.tokens(feat, cam, T_cp, T_wp, T_wl)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    N = V * h * w
    return {"tokens_ms": ms, "algorithmic_gflop": 2.0 * B * N * (192 * C + C * C) / 1e9,
            "min_traffic_gb": 2.0 * B * N * C * 4 / 1e9,
            "note": "AddRayPE.tokens (ray-point encoding + feature add + channels-last tokenisation), outside the metric"}


def train_bench(args):
    """--train: one data-parallel training step per "step" at the per-GPU shard of BASELINE config 4 (batch 32 scenes over 8 GPUs
    = 4 scenes per GPU, 10 views, 256 queries, 8 iterations): decoder forward with saved activations, the reference's set loss on
    synthetic boxes, HIP backward, ONE all-reduce of the flat gradient arena over RCCL, AdamW.  Reported beside the headline
    metric (which stays the inference number); dropout 0.1 as in config/train.yaml, exact-fp32 attention kernels."""
    from parq_amd import Obb3D, PARQDecoder, Pose, parallel, synth
    rank, local_rank, world = parallel.env_world()
    if world > 1:
        parallel.pin_to_local_cores(local_rank)       # before the first GPU call: NUMA-local host cores per rank
    device, backend = rank_device_and_backend(args, local_rank, world)
    parallel.init(backend=backend, device=device)
    red_dev = device if backend == "nccl" else None
    B = args.scenes_per_gpu if args.scenes_per_gpu > 1 else 4
    V, (h, w), Q, C, I = WORKLOAD["views"], WORKLOAD["feat_hw"], WORKLOAD["queries"], WORKLOAD["dim"], WORKLOAD["iters"]
    cfg = synth.decoder_cfg(dim=C, queries=Q, heads=WORKLOAD["heads"], ffn=WORKLOAD["ffn"], layers=I, dropout=args.dropout)   # config/train.yaml:53 (0.1)
    W = synth.make_decoder_weights(cfg, 41, damped=True)
    dec = PARQDecoder(cfg)
    dec.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=False)
    dec = dec.to(device).train()
    dec.dp_all_reduce = world > 1
    if args.attention_mode:
        dec.attention_mode = args.attention_mode
    dec.train_split8 = bool(args.train_split8)       # opt-in: mode 4 in the training forward, backward from its stage cache (decoder.py)
    inputs = build_inputs(B, device, seed=2000 + rank)
    if args.token_grad:                 # the gradient a trained backbone / ray-PE encoder in front of the decoder needs
        inputs = (inputs[0].requires_grad_(True),) + tuple(inputs[1:])
    obbs, sym = synth.make_boxes(3000 + rank, B, 12)
    obbs, sym = Obb3D(torch.from_numpy(obbs).to(device)), torch.from_numpy(sym).to(device)
    T_wl = Pose(inputs[4])
    opt = torch.optim.AdamW([p for p in dec.parameters() if p.requires_grad], lr=1e-4 * (B * world) / 256.0, foreach=True)
    np.random.seed(1 + rank)

    def step():
        opt.zero_grad(set_to_none=True)
        outs = dec(*inputs, feat_hw=(h, w))
        loss = dec.loss(outs, obbs, T_wl, sym)["total_loss"]
        loss.backward()
        torch.nn.utils.clip_grad_norm_(dec.parameters(), 1.0)
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    phase_ms = None
    if rank == 0 and args.phase_times:
        # untimed extra steps with an event between the phases (device time of each phase incl. the host stalls inside it)
        acc = []
        for _ in range(5):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
            dec._phase_hook = lambda name, e=ev[5]: e.record()
            opt.zero_grad(set_to_none=True)
            ev[0].record(); outs = dec(*inputs, feat_hw=(h, w))
            ev[1].record(); loss = dec.loss(outs, obbs, T_wl, sym)["total_loss"]
            ev[2].record(); loss.backward()
            ev[3].record(); torch.nn.utils.clip_grad_norm_(dec.parameters(), 1.0); opt.step()
            ev[4].record(); torch.cuda.synchronize()
            dec._phase_hook = None
            acc.append([ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[5]), ev[5].elapsed_time(ev[3]),
                        ev[3].elapsed_time(ev[4])])
        med = np.median(np.array(acc), axis=0)
        phase_ms = {"forward": float(med[0]), "loss": float(med[1]), "loss_autograd": float(med[2]), "hip_backward": float(med[3]),
                    "clip+adamw": float(med[4])}
    torch.cuda.synchronize(); parallel.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize(); parallel.barrier(); torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0
    dt = parallel.max_over_ranks(dt_own, device=red_dev)
    per_rank_ms = [x / args.steps * 1e3 for x in parallel.gather_over_ranks(dt_own, device=red_dev)]
    final_loss, train_mode = float(loss.detach()), dec._train_mode()
    opt_in = None
    if world == 1 and dec.attention_mode == "split8" and not dec.train_split8 and not (args.dev_lib or parq_env()):
        # after the timed region: the same steps with the training forward in mode split8 (PARQDecoder.train_split8, off by default)
        dec.train_split8 = True
        for _ in range(max(2, args.warmup)):
            step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        opt_in = {"train_attention_mode": dec._train_mode(), "ms_per_step": (time.perf_counter() - t1) / args.steps * 1e3}
        dec.train_split8 = False
    if rank == 0:
        print(json.dumps({
            "per_rank_ms_per_step": per_rank_ms,
            "metric": "training steps/sec (decoder forward + set loss + HIP backward + gradient all-reduce + AdamW)"
                      + (" [development library or PARQ_* set: not a headline]" if (args.dev_lib or parq_env()) else ""),
            "value": args.steps / dt, "unit": "steps/sec", "scenes_per_sec": args.steps * B * world / dt,
            "n_gpus": world, "collective_backend": backend, "rccl_ranks": world if backend == "nccl" else 0,
            "rccl_version": (".".join(str(x) for x in torch.cuda.nccl.version()) if world > 1 and hasattr(torch.cuda, "nccl") else None),
            "parq_env": parq_env(), "dev_lib": bool(args.dev_lib), "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "final_loss": final_loss, "phase_ms": phase_ms, "train_attention_mode": train_mode, "opt_in_train_split8": opt_in,
            "config": {"workload": "BASELINE cfg4 per-GPU shard: %d scenes, 10 views 480x640 (120x160 features), 256 queries, 8 iterations, "
                                   "d=256; dropout %g; 12 synthetic boxes per scene%s" % (B, args.dropout, "; token gradient" if args.token_grad else ""),
                       "scenes_per_gpu": B, "parallelism": "dp%d (gradient arena all-reduced in two buckets, the first overlapped with the cross-attention backward)" % world}}))
    if world > 1:
        parallel.barrier()
        torch.distributed.destroy_process_group()


def parq_env():
    """Every PARQ_* variable of this process.  The product library reads none of them (only the -DPARQ_DEV_PROBES build does), but
    the line records them and a run with any of them set — or with --dev-lib — is marked as not a headline number."""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("PARQ_")}


def rank_device_and_backend(args, local_rank, world):
    """One GPU per rank over RCCL ("nccl"); with --share-device (fewer GPUs than ranks) every rank sits on cuda:0 and the group
    runs over gloo.  Returns (device, backend or None for a single process)."""
    if world > 1 and args.share_device and torch.cuda.device_count() < world:
        torch.cuda.set_device(0)
        return torch.device("cuda", 0), "gloo"
    torch.cuda.set_device(local_rank)
    return torch.device("cuda", local_rank), ("nccl" if world > 1 else None)


def self_launch(n, share_device=False):
    """`python bench.py --gpus N` typed without a launcher: start N fresh ranks (one per GPU, RCCL rendezvous on
    127.0.0.1) as CHILD processes of this one, which has not initialised the GPU (no HIP call, no torch.cuda query
    besides device_count), and exit with the launcher's code.  Rank 0's JSON line is the child's stdout."""
    import socket
    import subprocess
    have = torch.cuda.device_count()                     # does not initialise the runtime on this image
    if have < n and not (share_device and have >= 1):
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (add --share-device to run the rank path on one GPU)" % (n, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scenes-per-gpu", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-b32", action="store_true", help="skip the 32-scene project+sample bandwidth measurement (6.3 GB of tokens)")
    ap.add_argument("--no-peaked", action="store_true", help="skip the peaked_workload record (the same workload with sharpened cross-attention)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 --pmc passes (roofline.traffic then comes from profiles/rNN_pmc.json)")
    ap.add_argument("--pmc-child", "--kernels-only", dest="pmc_child", action="store_true",
                    help="only the forwards of the configuration (warm-up + steps), nothing measured or printed: the command to put behind "
                         "`rocprofv3 ... --` so that per-kernel statistics hold the timed workload's kernels and nothing else (live_pmc uses it)")
    ap.add_argument("--train-split8", action="store_true", help="--train: run the training forward in mode split8 (PARQDecoder.train_split8)")
    ap.add_argument("--attention-mode", default=None, choices=["split", "split8", "fp32", "fp16", "bf16"],
                    help="cross-attention arithmetic; default = the library default (split8 at d = 256 / head dim 64: fp16 hi.hi + fp8 cross terms; "
                         "split: three fp16 terms). "
                         "fp16 / bf16 are the reduced-precision configurations (NOT the headline number)")
    ap.add_argument("--train", action="store_true", help="time the training step of BASELINE config 4's per-GPU shard instead")
    ap.add_argument("--dropout", type=float, default=0.1, help="--train: dropout rate (config/train.yaml:53 = 0.1, the default; other "
                    "values are for kernel A/B only)")
    ap.add_argument("--token-grad", action="store_true", help="--train: also compute d loss / d tokens (B x N x C; what a trained "
                    "backbone or the ray-PE encoder in front of the decoder consumes)")
    ap.add_argument("--phase-times", action="store_true", help="--train: also report the median device time of forward / loss / "
                    "backward / optimizer over 5 extra untimed steps (phase_ms)")
    ap.add_argument("--dev-lib", action="store_true", help="development only: bind parq_amd/_C/libparq_hip_dev.so (-DPARQ_DEV_PROBES: "
                    "environment A/B switches and probe kernels); the line is then marked as NOT a headline number")
    ap.add_argument("--share-device", action="store_true", help="N > 1 on a box with fewer than N GPUs: every rank uses cuda:0 and the "
                    "process group runs over gloo — exercises the rank path (sharding, barrier, max over ranks), not a scaling number")
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS), help="BASELINE.json configuration: cfg3 (default) is the one the "
                    "metric is quoted on = the headline; cfg2 (bf16) / cfg5 (fp16) are the reduced-precision configurations, emitted as "
                    "the same JSON line with their own roofline and bounded cpu_baseline (NOT the headline number)")
    ap.add_argument("--dim", type=int, default=256, help="decoder width; 256 = the BASELINE metric (default).  1024 = the reference's "
                    "shipped DEC_DIM (head dim 256): reported beside the headline, NOT the BASELINE metric")
    args = ap.parse_args()
    conf = CONFIGS[args.config]
    if "dim" in conf:
        args.dim = conf["dim"]
    WORKLOAD["dim"] = args.dim
    WORKLOAD.update({k: conf[k] for k in ("views", "image_hw", "feat_hw", "queries", "iters")})
    if conf["mode"] and not args.attention_mode:
        args.attention_mode = conf["mode"]
    if args.config != "cfg3":
        args.no_b32 = True
    if args.dim != 256 and args.config != "shipped":
        args.no_cpu_baseline = True                     # the bounded CPU sample is sized for the headline configuration
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, args.share_device))
    if args.dev_lib:
        from parq_amd import _lib
        _lib.use_dev_library()
    if args.train:
        return train_bench(args)

    from parq_amd import parallel
    rank, local_rank, world = parallel.env_world()
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d" % (args.gpus, world))
    # host placement BEFORE the first GPU call (the runtime's helper threads inherit the mask): each rank on the cores of its
    # GPU's NUMA node, ranks that share a node on disjoint slices.  In-process sched_setaffinity, no re-exec.
    shared_dev = world > 1 and args.share_device and torch.cuda.device_count() < world      # (device_count does not initialise the runtime)
    pinned = parallel.pin_to_local_cores(local_rank, gpu_index_of=(lambda r: 0) if shared_dev else None) if world > 1 else []
    torch.set_grad_enabled(False)     # the metric is the inference forward; the reference's drivers run it under no_grad (eval.py:46)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback in the product path)"
    device, backend = rank_device_and_backend(args, local_rank, world)
    parallel.init(backend=backend, device=device)                             # "nccl" is RCCL on ROCm
    red_dev = device if backend == "nccl" else None                           # gloo reduces host scalars on the host

    B = args.scenes_per_gpu
    I = WORKLOAD["iters"]
    h, w = WORKLOAD["feat_hw"]
    cfg, W, dec = build_decoder(device)
    if args.attention_mode:
        dec.attention_mode = args.attention_mode
    inputs = build_inputs(B, device, seed=1000 + rank)         # each rank owns its own scenes (sharded)

    def step():
        return dec(*inputs, feat_hw=(h, w))

    if args.pmc_child:                    # live_pmc(): the kernels of this configuration under rocprofv3 --pmc, nothing printed
        for _ in range(args.warmup + args.steps):
            step()
        torch.cuda.synchronize()
        return

    def barrier():
        torch.cuda.synchronize()
        parallel.barrier()
        torch.cuda.synchronize()

    # Device spin-up, untimed and before the W warm-up steps: the first ~12 forwards of a process run 2 - 12 % slower than the steady
    # state (lazy code-object loading on the first, then the clock / power controller settling: tools/step_times.py prints the
    # per-forward times).  A throughput number should not depend on how few steps the caller asks for.
    for _ in range(PREWARM_STEPS):
        step()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt_own = time.perf_counter() - t0
    dt = parallel.max_over_ranks(dt_own, device=red_dev)
    per_rank_ms = [x / args.steps * 1e3 for x in parallel.gather_over_ranks(dt_own, device=red_dev)]
    per_rank_cpus = parallel.gather_over_ranks(float(len(pinned)), device=red_dev)
    try:                                  # (object collectives pickle through the backend: never let a reporting extra end a multi-GPU run)
        per_rank_cpulists = [parallel.cpulist_string(c) for c in parallel.gather_objects(list(pinned))]
    except Exception as exc:              # noqa: BLE001
        per_rank_cpulists = ["unavailable: %s" % type(exc).__name__] * world

    # ---- per-step times from hipEvents on the launch stream (median of >= 20 runs); the wall-clock mean above stays
    # the contract's `value`, this is its cross-check and its spread
    n_ev = max(20, args.steps)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_ev + 1)]
    evs[0].record()
    for i in range(n_ev):
        step()
        evs[i + 1].record()
    torch.cuda.synchronize()
    step_ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(n_ev))
    pct = lambda q: step_ms[min(n_ev - 1, int(q * n_ev))]

    # ---- per-kernel-group times: hipEvents recorded by the library on the launch stream
    dec.profile_enable(True)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    prof = dec.profile_read()
    dec.profile_enable(False)

    # ---- the same forward with all three terms of every cross-attention product in fp16 (attention_mode "split"), when the timed
    # mode was "split8": untimed by the contract, reported beside `value` with the largest difference of the two modes' outputs
    strict = None
    guard_tripped = dec.attention_too_peaked() if hasattr(dec, "attention_too_peaked") else False
    guard_lmin = dec.attention_min_row_sum() if hasattr(dec, "attention_min_row_sum") else None
    if dec.attention_mode == "split8" and world == 1 and not (args.dev_lib or parq_env()):
        fast_out = [{k: v.clone() for k, v in o.items()} for o in step()]
        dec.attention_mode = "split"
        for _ in range(3):
            strict_out = step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            strict_out = step()
        torch.cuda.synchronize()
        dt_strict = time.perf_counter() - t1
        diff = max(float(((a[k].double() - b[k].double()).abs() / b[k].double().abs().clamp(min=1.0)).max())
                   for a, b in zip(fast_out[:1], strict_out[:1]) for k in a)
        strict = {"attention_mode": "split", "value": B * I * args.steps / dt_strict, "unit": "decoder-iterations/sec",
                  "ms_per_step": dt_strict / args.steps * 1e3,
                  "first_iteration_outputs_max_difference_to_the_timed_mode": diff,
                  "note": "same inputs, same process, after the timed region; the difference is taken on iteration 0 (later iterations are "
                          "free-running on white-noise features, where the recurrence amplifies any rounding difference — teacher-forced "
                          "parity of both modes: tests/test_gpu_headline.py, tests/test_gpu_reference_pins.py)"}
        dec.attention_mode = "split8"
        step()
        torch.cuda.synchronize()

    # ---- what the never-NaN default costs: the same K steps under each policy, alternating, after the timed region
    policy_cost = None
    if world == 1 and not (args.dev_lib or parq_env()) and dec.attention_mode in ("split", "split8", "fp16"):
        default_policy = dec.range_check

        def timed(policy, graph):
            dec.range_check, dec.use_graph = policy, graph
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            host = 0.0
            t1 = time.perf_counter()
            for _ in range(args.steps):
                th = time.perf_counter()
                step()
                host += time.perf_counter() - th
            torch.cuda.synchronize()
            dtp = time.perf_counter() - t1
            return {"value": B * I * args.steps / dtp, "ms_per_step": dtp / args.steps * 1e3, "host_ms_per_call": host / args.steps * 1e3}
        rounds = [(pol, gr) for _ in range(2) for pol in ("sync", "lazy") for gr in (True, False)]
        acc = {}
        for pol, gr in rounds:
            acc.setdefault((pol, gr), []).append(timed(pol, gr))
        dec.range_check, dec.use_graph = default_policy, True
        step()
        torch.cuda.synchronize()
        best = lambda k: max(acc[k], key=lambda r: r["value"])
        policy_cost = {"default_policy": default_policy, "unit": "decoder-iterations/sec",
                       "sync": best(("sync", True)), "lazy": best(("lazy", True)),
                       "sync_without_captured_forward": best(("sync", False)), "lazy_without_captured_forward": best(("lazy", False)),
                       "host_enqueue_ms": best(("lazy", True))["host_ms_per_call"],
                       "host_enqueue_ms_without_captured_forward": best(("lazy", False))["host_ms_per_call"],
                       "cost_of_the_default": 1.0 - best(("sync", True))["value"] / best(("lazy", True))["value"],
                       "note": "same process, after the timed region, %d steps per figure, two alternating repeats (best of each); `value` of the line is "
                               "measured under the default policy (sync: every forward waits for its stream and reads one pinned word before it "
                               "returns, and is re-run with safer arithmetic if the device flagged it); lazy = no wait on the forward path (a "
                               "flagged forward returns NaN); host_enqueue_ms = wall time of the Python call while the device is busy (lazy: "
                               "nothing waits)" % args.steps}

    # ---- two scenes in flight (untimed by the contract, reported beside `value`): the same module called alternately on two HIP
    # streams with two sets of inputs — the small-op chain of one forward leaves most of the chip to the K/V projection and
    # cross-attention of the other.  What a server that keeps a second scene queued gets; `value` stays one forward at a time.
    in_flight = in_flight3 = None
    if world == 1 and not (args.dev_lib or parq_env()):
        inputs2 = build_inputs(B, device, seed=5000 + rank)
        pair = [inputs, inputs2]
        side = [torch.cuda.Stream(device), torch.cuda.Stream(device)]
        for st in side:
            st.wait_stream(torch.cuda.current_stream(device))

        from parq_amd import InFlight
        serial_out = [[{k: v.clone() for k, v in o.items()} for o in dec(*p_in, feat_hw=(h, w))] for p_in in pair]     # one at a time
        torch.cuda.synchronize()
        runner = InFlight(dec, depth=2)
        last = [None, None]

        def go(n):
            tickets = []
            for i in range(n):
                tickets.append((i & 1, runner.submit(*pair[i & 1], feat_hw=(h, w))))
                if len(tickets) == 2:                              # the default policy's check of a forward happens in result()
                    j, t = tickets.pop(0)
                    last[j] = t.result()
            for j, t in tickets:
                last[j] = t.result()
        go(8)
        torch.cuda.synchronize()
        n2 = max(40, args.steps)
        t2 = time.perf_counter()
        go(n2)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t2
        same = all(torch.equal(a[k], b[k]) for j in (0, 1) for a, b in zip(last[j], serial_out[j]) for k in a)
        # the same with three forwards outstanding (a third stream / workspace / captured graph): launch-bound configurations gain
        # most from it (cfg 2: 13.6 k -> 16.3 k it/s; cfg 3: 7.2 k -> 7.5 k)
        inputs3 = build_inputs(B, device, seed=9000 + rank)
        trio = [inputs, inputs2, inputs3]
        runner3 = InFlight(dec, depth=3)

        def go3(n):
            tickets = []
            for i in range(n):
                tickets.append(runner3.submit(*trio[i % 3], feat_hw=(h, w)))
                if len(tickets) == 3:
                    tickets.pop(0).result()
            for t in tickets:
                t.result()
        go3(9)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        go3(n2)
        torch.cuda.synchronize()
        dt3 = time.perf_counter() - t3
        in_flight3 = {"streams": 3, "value": B * I * n2 / dt3, "unit": "decoder-iterations/sec", "ms_per_step": dt3 / n2 * 1e3, "steps": n2,
                      "policy": dec.range_check, "note": "parq_amd.InFlight(depth=3), otherwise as two_scenes_in_flight; not the contract's `value`"}
        del inputs3, runner3
        del serial_out, last
        in_flight = {"streams": 2, "value": B * I * n2 / dt2, "unit": "decoder-iterations/sec", "ms_per_step": dt2 / n2 * 1e3, "steps": n2,
                     "outputs_bit_identical_to_one_at_a_time": bool(same), "policy": dec.range_check,
                     "note": "one module, parq_amd.InFlight(depth=2): forwards submitted alternately on two HIP streams (a workspace and a "
                             "captured graph per stream: tests/test_gpu_streams.py, results bit-identical to serial calls); under the default "
                             "policy every forward is checked in Ticket.result() before its outputs are handed out; not the contract's `value`"}
        del inputs2

    if rank == 0 and world == 1 and not args.no_pmc and not (args.dev_lib or parq_env()):
        live_pmc(args, B)                 # two short child runs under rocprofv3 --pmc (after every timed region of this process)
    if rank == 0:
        V, Q, C = WORKLOAD["views"], WORKLOAD["queries"], WORKLOAD["dim"]
        N = V * h * w
        total_iters = world * B * I * args.steps
        live = _LIVE_PMC["by_kernel"] is not None and _LIVE_PMC["scenes"] == B      # HBM-side bytes measured in this run (live_pmc)
        ca_ms, ca_n = prof["cross_attn"]
        ps_ms, ps_n = prof["project_sample"]
        flop_per_launch = 4.0 * Q * N * C * B                       # QK^T + PV, all heads, B scenes
        ach_tflops = flop_per_launch / (ca_ms / ca_n * 1e-3) / 1e12 if ca_n else None
        bytes_per_launch = (4.0 * V * Q * C + Q * C) * 4.0 * B
        ps_gbs = bytes_per_launch / (ps_ms / ps_n * 1e-3) / 1e9 if ps_n else None
        mode = dec.attention_mode
        split = mode in ("split", "split8")
        split8 = mode == "split8" and C == 256 and C // WORKLOAD["heads"] == 64 and N % 64 == 0
        half = mode in ("fp16", "bf16")
        # dominant kernel: cross-attention QK^T + PV.  In "split" mode every fp32-accurate product costs
        # SPLIT_PASSES fp16 MFMAs, so the matrix roof for ALGORITHMIC flops is the dense fp16 peak / 3.
        mfma_peak = (PEAK_F16_MATRIX_TFLOPS / SPLIT_PASSES if split else
                     PEAK_F16_MATRIX_TFLOPS if half else PEAK_F32_MATRIX_TFLOPS)
        kv_bytes = 2.0 * N * C * (2.0 if half else 4.0) * B
        roofline = {"bound": "mfma",
                    "kernel": (("flash_split_pipe_kernel" if C // WORKLOAD["heads"] == 64 else "flash_split256_kernel") +
                               " (cross-attention QK^T+PV, fp16 hi/lo 3-term products, fp32 accumulate)" if split
                               else "flash_split_pipe_kernel<4,0,1> (cross-attention QK^T+PV, single %s products, fp32 accumulate)" % mode if half
                               else "flash_f32_kernel (cross-attention QK^T+PV, fp32 MFMA)"),
                    "achieved": ach_tflops, "peak": mfma_peak, "unit": "TFLOP/s",
                    "frac": (ach_tflops / mfma_peak) if ach_tflops else None,
                    "traffic": pmc_traffic("flash_split_pipe_kernel", B) if (split and C == 256 and (args.config == "cfg3" or live)) else None,
                    "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE: see traffic_source); algorithmic stream = 2*N*C*4*B bytes",
                    "avg_launch_ms": (ca_ms / ca_n) if ca_n else None, "launches": ca_n,
                    "algorithmic_gflop_per_launch": flop_per_launch / 1e9,
                    "peak_note": ("dense fp16 MFMA peak 2500 TFLOP/s / 3 passes per product; the fp32-MFMA peak is %.1f"
                                  % PEAK_F32_MATRIX_TFLOPS) if split else "dense fp16/bf16 MFMA peak" if half else "fp32 MFMA peak",
                    "hbm_stream_gbs": (kv_bytes / (ca_ms / ca_n * 1e-3) / 1e9) if ca_n else None,
                    "note": "launch time from hipEvents around this kernel alone (its merge kernel is group cross_attn_merge)"}
        if not split8 and ca_n and kv_bytes / (PEAK_HBM_GBS * 1e9) > flop_per_launch / (mfma_peak * 1e12):
            # the roof that bounds the launch is the larger of its two floors: with few queries per key (cfg 2: 128; the 16-bit modes at
            # 256) streaming the K/V cache once at 8 TB/s takes longer than the launch's products at the dense matrix peak
            stream_gbs = kv_bytes / (ca_ms / ca_n * 1e-3) / 1e9
            roofline.update({"bound": "hbm", "achieved": stream_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": stream_gbs / PEAK_HBM_GBS,
                             "algorithmic_bytes_per_launch": kv_bytes,
                             "peak_note": "HBM3E 8 TB/s (floor %.1f us) against %.1f us of products at the matrix peak; a plain streaming kernel "
                                          "reaches 5.6 TB/s on this part" % (kv_bytes / (PEAK_HBM_GBS * 1e9) * 1e6, flop_per_launch / (mfma_peak * 1e12) * 1e6),
                             "mfma": {"achieved": ach_tflops, "peak": mfma_peak, "unit": "TFLOP/s", "frac": ach_tflops / mfma_peak}})
        if split8:
            # mode 4: one fp16 product + two MX-fp8 products per algorithmic product.  Matrix roof for algorithmic flops:
            # 2500 / 1.5 = 1667 TFLOP/s = 30 us per launch at cfg 3; the K/V stream of the launch (N C x (4 + 2) bytes:
            # K hi16 + hi8 + lo8, V fp16) is 37 us at 8 TB/s, so HBM is the roof that bounds this kernel (stream-only build of the
            # kernel: 54 us = 5.5 TB/s, the streaming ceiling of this part; profiles/r04_split8_ingredient_probes.txt)
            kv_bytes = N * C * 6.0 * B                            # K: hi16 + hi8 + lo8, V: fp16 (24 KB per 64-key stage and head)
            stream_gbs = (kv_bytes / (ca_ms / ca_n * 1e-3) / 1e9) if ca_n else None
            mx_peak = PEAK_F16_MATRIX_TFLOPS / 1.5
            roofline.update({"bound": "hbm", "kernel": "flash_split8_kernel (cross-attention: Q K^T = fp16 hi.hi + MX-scaled fp8 e4m3 cross terms, "
                                                       "P V = fp16 product with a self-consistent normaliser, fp32 accumulate)",
                             "achieved": stream_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": (stream_gbs / PEAK_HBM_GBS) if stream_gbs else None,
                             "traffic": pmc_traffic("flash_split8_kernel", B) if (args.config == "cfg3" or live) else None,
                             "algorithmic_bytes_per_launch": kv_bytes, "hbm_stream_gbs": stream_gbs,
                             "peak_note": "HBM3E 8 TB/s; a plain streaming kernel reaches 5.6 TB/s on this part (profiles/r02_hbm_stream_ceiling.txt)",
                             "mfma": {"achieved": ach_tflops, "peak": mx_peak, "unit": "TFLOP/s", "frac": (ach_tflops / mx_peak) if ach_tflops else None,
                                      "note": "algorithmic flops against 2500 / 1.5: Q K^T = one fp16 pass + two fp8 passes at twice the rate, P V = one fp16 pass"}})
        not_headline = bool(args.dev_lib or parq_env())
        kv_ms, kv_n = prof["kv_proj"]
        kvp_bytes = ((N * C * 4.0 + N * C * 6.0) * B if split8 else
                     3.0 * N * C * 4.0 * B if not half else (N * C * 4.0 + 2.0 * N * C * 2.0) * B)     # tokens in, K and V cache images out
        roofline["hbm_frac"] = (roofline["hbm_stream_gbs"] / PEAK_HBM_GBS) if roofline["hbm_stream_gbs"] else None
        traffic_source = (_LIVE_PMC["note"] if live else
                          "recorded: %s (rocprofv3 --pmc passes of the same command), not measured in this run%s"
                          % (_pmc_record()[1], "" if not _LIVE_PMC["note"] else " (%s)" % _LIVE_PMC["note"]))
        roofline["traffic_source"] = traffic_source
        roofline_kv = {"bound": "hbm", "kernel": "kvproj_dma_kernel (hoisted K/V in-projection, once per forward)",
                       "achieved": (kvp_bytes / (kv_ms / kv_n * 1e-3) / 1e9) if kv_n else None, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                       "frac": (kvp_bytes / (kv_ms / kv_n * 1e-3) / 1e9 / PEAK_HBM_GBS) if kv_n else None,
                       "traffic": pmc_traffic("kvproj_dma_kernel", B) if (args.config == "cfg3" or live) else None, "traffic_source": traffic_source,
                       "algorithmic_bytes_per_launch": kvp_bytes, "avg_launch_ms": (kv_ms / kv_n) if kv_n else None, "launches": kv_n,
                       "streaming_ceiling_ms": kvp_bytes / 5.6e12 * 1e3,
                       "note": "reads N*C fp32 tokens, writes the K and V cache images; a plain streaming kernel with this 1:2 read/write "
                               "shape reaches 5.6 TB/s on MI355X (profiles/r02_hbm_stream_ceiling.txt; 24 of 32 KB per stage written in mode "
                               "split8); also 4*N*C^2 = %.1f GFLOP x 3 "
                               "fp16 passes on the matrix pipe" % (4.0 * N * C * C * B / 1e9)}
        out = {
            "metric": "decoder-iterations/sec (%d views, %d queries, d=%d)%s%s%s" % (
                V, Q, C, "" if (C == 256 or args.config == "shipped") else " [not the BASELINE metric: non-default --dim]",
                "" if args.config == "cfg3" else " [%s: not the configuration the metric is quoted on]" % args.config,
                " [development library or PARQ_* set: not a headline]" if not_headline else ""),
            "value": total_iters / dt, "unit": "decoder-iterations/sec",
            "n_gpus": world, "collective_backend": backend, "rccl_ranks": world if backend == "nccl" else (1 if world == 1 else 0),
            "parq_env": parq_env(), "dev_lib": bool(args.dev_lib),
            "steps": args.steps, "warmup": args.warmup, "prewarm_steps": PREWARM_STEPS,
            "ms_per_step": dt / args.steps * 1e3,
            "per_rank_ms_per_step": per_rank_ms,       # every rank's own time over the same K steps (value uses the maximum): stragglers show
            "per_rank_pinned_cpus": [int(x) for x in per_rank_cpus],   # host cores each rank pinned itself to (0 = mask left alone)
            "per_rank_pinned_cpulist": per_rank_cpulists,
            "rccl_version": (".".join(str(x) for x in torch.cuda.nccl.version()) if world > 1 and hasattr(torch.cuda, "nccl") else None),
            "step_ms_hipevents": {"median": pct(0.5), "p10": pct(0.1), "p90": pct(0.9), "min": step_ms[0], "n": n_ev,
                                  "iterations_per_sec_at_median": B * I / (pct(0.5) * 1e-3),
                                  "note": "rank 0, one hipEvent pair per forward on the launch stream; `value` is the contract's wall-clock figure"},
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": ("f32 (cross-attention scores: fp16 hi.hi + MX-fp8 e4m3 cross terms; P V: fp16 probabilities and values with a "
                                          "self-consistent normaliser; fp32 accumulation — 4e-6..1.5e-5 from float64 at the outputs on the reference's fixtures, the "
                                          "reference's own fp32 run: 6e-5..1.4e-4; guarded: rows on too few keys fall back to fp16 x 3; K/V projection "
                                          "as fp16 hi/lo split products; `strict_fp16x3` is the same run with all three terms in fp16)" if split8 else
                                          "f32 (cross-attention and K/V projection as fp16 hi/lo split products with fp32 accumulation)" if split
                                          else "%s cross-attention and K/V projection operands, fp32 accumulation, fp32 elsewhere (reduced precision: not the headline configuration)" % mode if half
                                          else "f32"),
            "data": "synthetic",
            "config": {"workload": "%s, d=%d, 4 heads, FFN 768, ResNet-FPN-shaped synthetic features" % (conf["text"], C),
                       "scenes_per_gpu": B, "parallelism": "dp%d (scene-sharded, no data-path collective)" % world},
            "parity_metric": "outputs are held to the reference within 1e-4 on |a-b| / max(1,|b|) (absolute below 1, relative above; "
                             "tests/golden_util.py); reduced-precision modes: bf16 2e-3, fp16 3e-4",
            "roofline": roofline,
            "roofline_kv_proj": roofline_kv,
            "roofline_project_sample": {"bound": "hbm", "kernel": "project_sample_kernel", "scenes": B,
                                        "achieved": ps_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                        "frac": (ps_gbs / PEAK_HBM_GBS) if ps_gbs else None,
                                        "traffic": pmc_traffic("project_sample_kernel", B) if (args.config == "cfg3" or live) else None,
                                        "algorithmic_bytes_per_launch": bytes_per_launch,
                                        "avg_launch_ms": (ps_ms / ps_n) if ps_n else None, "launches": ps_n,
                                        "note": ("latency-bound at one scene: a %.1f us launch over a 197 MB token tensor that sits in the 256 MB "
                                                 "Infinity Cache; the bandwidth figure is roofline_project_sample_b32" % (ps_ms / ps_n * 1e3))
                                                if (B == 1 and ps_n) else None},
            "kernel_groups_ms_per_step": {k: v[0] / args.steps for k, v in prof.items()},
        }
        if strict is not None:
            out["strict_fp16x3"] = strict
        if in_flight is not None:
            out["two_scenes_in_flight"] = in_flight
            out["three_scenes_in_flight"] = in_flight3
        if policy_cost is not None:
            out["guard_policy_cost"] = policy_cost
            out["host_enqueue_ms"] = policy_cost["host_enqueue_ms"]
        if hasattr(dec, "attention_too_peaked"):
            # mode "split8" is kept only while every cross-attention row spreads over enough keys (its error model); a tripped guard
            # would have switched the module to "split" and `dtype` / `roofline` above would say so
            out["attention_mode"] = mode
            out["attention_guard"] = {"row_probability_sum_threshold": 256, "tripped": bool(guard_tripped),
                                      "safe_heads": "0b" + format(int(getattr(dec, "safe_heads", 0)), "0%db" % WORKLOAD["heads"]),
                                      "smallest_row_probability_sum": guard_lmin, "policy": dec.range_check}
            if split8 and world == 1 and B == 1 and not args.no_peaked and not (args.dev_lib or parq_env()):
                out["peaked_workload"] = peaked_workload(device, inputs, h, w, args.steps)
        if C == 256:
            out["ray_pe"] = ray_pe_timing(B, device)
        if world == 1:
            out["device_state_under_load"] = device_state_under_load(step, index=local_rank)
        if world == 1 and C == 256 and B == 1 and not args.no_b32:
            out["roofline_project_sample_b32"] = project_sample_b32(dec, device, h, w)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, W, inputs, view_limit=4 if args.config == "cfg5" else None)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        parallel.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
