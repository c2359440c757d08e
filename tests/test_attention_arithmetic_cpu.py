"""CPU tier: the arithmetic MODEL of the attention modes (tests/emulate_attention_arithmetic.py) on a reference fixture — what each
mode's rounding does to the decoder outputs, independent of any kernel.  The GPU tier measures the kernels at the same numbers
(tests/test_gpu_split8.py prints 3.5e-6 for "split8" on g19, this model 3.5e-6): the error of mode 4 is its arithmetic, not its code."""
import torch

from parq_amd import synth
from oracle import parq_oracle as O
import golden_util as G
import emulate_attention_arithmetic as E


def test_modelled_error_of_the_attention_modes_on_a_reference_fixture():
    case, z = G.load("g15_cfg5_shape")                       # 20 views of 12 x 16 features = 3840 keys (a multiple of 64), 512 queries
    cfg, W, sc = G.inputs(case)
    refs = G.forced_refs(z, cfg.TRANSFORMER.SCALE)
    od = O.OracleDecoder(cfg, W, synth.SCANNET_MEAN_SIZES, dtype=torch.float64)
    od.prepare(sc["tokens"], sc["camera"], sc["T_camera_pseudoCam"], sc["T_world_pseudoCam"], sc["T_world_local"])
    worst = {}
    with torch.no_grad():
        for k in range(2):
            ref = torch.from_numpy(refs[k]).double()
            exact = od.iterate(ref, k)[0]
            for mode in ("split", "split8", "fp16"):
                with E.patched(mode):
                    out = od.iterate(ref, k)[0]
                e = max(float(((out[key] - exact[key]).abs() / exact[key].abs().clamp(min=1)).max()) for key in G.KEYS)
                worst[mode] = max(worst.get(mode, 0.0), e)
    print("\nmodelled distance from float64 (g15, two iterations):", {m: "%.1e" % e for m, e in worst.items()})
    assert worst["split"] < 2e-7                                      # three fp16 terms: fp32-rounding class
    assert worst["split"] < worst["split8"] < 2e-5                    # fp8 cross terms: ~16 significant bits per product
    assert worst["split8"] < worst["fp16"] / 8 and worst["fp16"] < 3e-4
