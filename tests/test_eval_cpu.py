"""CPU tier: host-side evaluation and schedule logic against vectors captured from the reference
(oracle/make_golden.py: g12 = utils/f1_eval.py F1Calculator, g13 = utils/train_utils.py CosineAnnealingWarmupRestarts)."""
import json
import os
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

import golden_util as G
from parq_amd import synth
from parq_amd.f1_eval import F1Calculator, canonical, iou3d
from parq_amd.schedule import CosineAnnealingWarmupRestarts, lr_at


def _g(name):
    return np.load(os.path.join(G.GOLDEN_DIR, name))


def test_oriented_iou_matches_reference_values():
    g = _g("g12_f1.npz")
    assert len(g["iou_pairs"]) == 24 and (g["iou_values"][:, 0] > 0.05).sum() >= 10
    for pair, want in zip(g["iou_pairs"], g["iou_values"]):
        got = iou3d(canonical(pair[0]), canonical(pair[1]))
        assert abs(got[0] - want[0]) < 1e-12 and abs(got[1] - want[1]) < 1e-12       # same clip + area arithmetic
    far = canonical(g["iou_pairs"][0][0] + 50.0)
    assert iou3d(canonical(g["iou_pairs"][0][0]), far) == (0.0, 0.0)
    nan = canonical(g["iou_pairs"][0][0]) * np.nan
    assert iou3d(nan, far) == (0.0, 0.0)
    edge_on = g["iou_pairs"][0][0][:, [0, 2, 1]]                        # a box whose footprint in this convention is a segment
    assert iou3d(canonical(edge_on), canonical(edge_on))[0] == 0.0


def test_f1_tracker_matches_reference_run():
    """Four snippet batches over three scenes (ragged batch, changing scene order): same tracks (count, class, id, score) and the
    same nine metrics as the reference's calculator, with NumPy's global generator seeded the same way."""
    from oracle.make_golden import F1_CASE, f1_case_inputs
    g = _g("g12_f1.npz")
    assert json.loads(bytes(g["meta"]).decode()) == json.loads(json.dumps(F1_CASE))
    calc = F1Calculator(F1_CASE["conf"])
    np.random.seed(F1_CASE["np_seed"])
    for st in f1_case_inputs(F1_CASE):
        out = {"pred_corners_world": torch.from_numpy(st["corners"]), "sem_cls_prob": torch.from_numpy(st["prob"]),
               "pred_mask": torch.from_numpy(st["mask"]), "scene_name": st["scenes"]}
        calc.step(out, [{"labels": torch.from_numpy(x["labels"]), "gt_corners_world": torch.from_numpy(x["corners"])} for x in st["gts"]])
    for name in F1_CASE["scenes"]:
        assert len(calc.preds[name]) == int(g["ntrack_" + name]) and len(calc.gts[name]) == int(g["ngt_" + name])
        assert np.array_equal([t[0] for t in calc.preds[name]], g["trackcls_" + name])
        assert np.array_equal([t[-1] for t in calc.preds[name]], g["trackid_" + name])
        assert np.array_equal(np.array([t[2] for t in calc.preds[name]], np.float64), g["trackscore_" + name])
    metrics = calc.compute_metrics()
    assert set(metrics) == {"%s_%s" % (t, k) for t in (0.25, 0.5, 0.7) for k in ("accuracy", "recall", "f1")}
    for k, v in metrics.items():
        assert v == float(g["metric_" + k]), k                                          # counts -> identical ratios
    assert metrics["0.25_f1"] > metrics["0.5_f1"] > metrics["0.7_f1"] > 0                # the case separates the thresholds
    calc.reset()
    assert calc.preds == {} and calc.gts == {}


def test_f1_tracker_empty_snippets():
    calc = F1Calculator(0.1)
    empty = {"pred_corners_world": np.zeros((1, 4, 8, 3), np.float32), "sem_cls_prob": np.full((1, 4, 10), 0.1, np.float32),
             "pred_mask": np.zeros((1, 4), bool), "scene_name": ["s"]}
    gt0 = [{"labels": np.zeros((0,), np.int64), "gt_corners_world": np.zeros((0, 8, 3), np.float32)}]
    calc.step(empty, gt0)
    calc.step(empty, gt0)                      # second visit: assignment on 0 x 0 matrices
    m = calc.compute_metrics()
    assert all(v == 0 for v in m.values())


def test_lr_schedule_matches_reference_sequences():
    g = _g("g13_lr_schedule.npz")
    cases = json.loads(bytes(g["meta"]).decode())
    assert len(cases) == 3
    for i, c in enumerate(cases):
        opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=c["max_lr"])
        sch = CosineAnnealingWarmupRestarts(opt, c["first"], c["mult"], c["max_lr"], c["min_lr"], c["warmup"])
        seq = [opt.param_groups[0]["lr"]]
        for _ in range(c["steps"]):
            opt.step()
            sch.step()
            seq.append(opt.param_groups[0]["lr"])
        assert np.array_equal(np.array(seq, np.float64), g["lr_%d" % i]), i               # same arithmetic -> bit-equal
        # resuming from a checkpointed epoch gives the same rate as stepping there
        opt2 = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=c["max_lr"])
        sch2 = CosineAnnealingWarmupRestarts(opt2, c["first"], c["mult"], c["max_lr"], c["min_lr"], c["warmup"])
        sch2.load_state_dict({**sch.state_dict(), "last_epoch": 17})
        assert opt2.param_groups[0]["lr"] == lr_at(17, c["first"], c["mult"], c["max_lr"], c["min_lr"], c["warmup"]) == seq[17]


def test_configure_optimizers_follows_reference_recipe():
    """model/parq_lightning.py:150-199: lr = base * effective batch / 256 when AUTOSCALE_LR, floor = base / 256 for an
    effective batch <= 256, first cycle = ceil(epochs / sum(mult^i))."""
    from parq_amd import PARQ
    dcfg = synth.decoder_cfg(dim=64, queries=8, heads=1, ffn=64, layers=1)
    tok = NS(OUT_CHANNELS=64, RAY_POINTS_SCALE=dcfg.TRANSFORMER.SCALE, NUM_SAMPLES=64, MIN_DEPTH=0.25, MAX_DEPTH=5.25)
    cfg = NS(MODEL=NS(TOKENIZER=tok, DECODER=dcfg),
             OPTIMIZER=NS(LEARNING_RATE=1e-4, AUTOSCALE_LR=True, WARMUP_EPOCHS=2, CYCLE_MULT=2.0, NUM_RESTARTS=2),
             DATAMODULE=NS(BATCH_SIZE=4), TRAINER=NS(NUM_NODES=1, GPUS=8, ACCUMULATE_GRAD_BATCHES=1, MAX_EPOCHS=30))
    conf = PARQ(cfg).configure_optimizers()
    opt, sch = conf["optimizer"], conf["lr_scheduler"]["scheduler"]
    assert conf["lr_scheduler"]["interval"] == "epoch" and isinstance(opt, torch.optim.AdamW)
    assert sch.max_lr == pytest.approx(1e-4 * 32 / 256) and sch.min_lr == pytest.approx(1e-4 / 256)
    assert sch.first_cycle_steps == 10 and sch.warmup_steps == 2 and sch.cycle_mult == 2.0
    assert opt.param_groups[0]["lr"] == sch.min_lr                                   # warm-up starts from the floor
    cfg.OPTIMIZER = NS(LEARNING_RATE=1e-4, AUTOSCALE_LR=False)                       # no schedule section -> bare optimizer
    assert isinstance(PARQ(cfg).configure_optimizers(), torch.optim.AdamW)
