"""GPU tier, stage by stage: the named intermediates of one decoder iteration (parq_workspace_lookup) against the float64 oracle,
so that every §8(a) function is pinned on its own and not only through the end of the iteration:

  posemb        pos2posemb3d (model/transformer_parq.py:45-64) of the NEXT reference points, written by the decode kernel
  pos_hidden    relu(position_encoder[0]) of this iteration's reference points (:317) — through `posemb_kernel`
  tgt           project + grid_sample + mean over valid views (:129-161, 321)
  self_qkv      self-attention in-projection of (tgt + pos | tgt + pos | tgt) (:372-374) — the position MLP's last layer is
                folded into this GEMM and into cross_q, so these two pin position_encoder[2]
  xa_prenorm1   tgt + self-attention block, checked after norm1 (:375-376)
  cross_q       cross-attention query projection of norm1(...) + pos (:377)
  xb_prenorm2   + cross-attention block over all N keys, checked after norm2 (:377-381)
  xc_prenorm3   + feed-forward block, checked after norm3 (:382-384)
Teacher-forced reference points from the fixtures; bound 5e-6 for the stages in front of the long cross-attention, 2e-5 behind it
(|a-b| / max(1,|b|))."""
import pytest
import torch
import torch.nn.functional as F

from parq_amd import synth
from oracle import parq_oracle as O
import golden_util as G
from gpu_util import dev, make_decoder, scene_args

pytestmark = pytest.mark.gpu


def _err(a, r):
    a, r = a.cpu().double().reshape(-1), r.double().reshape(-1)
    assert a.numel() == r.numel()
    return float(((a - r).abs() / r.abs().clamp(min=1)).max())


@pytest.mark.parametrize("name", ["g1_cfg1", "g2_forced", "g8_unshared", "g4_edges"])
def test_named_intermediates_against_the_float64_oracle(name):
    case, z = G.load(name)
    cfg, W, sc = G.inputs(case)
    dec = make_decoder(cfg, W)
    refs = G.forced_refs(z, cfg.TRANSFORMER.SCALE)
    od = O.OracleDecoder(cfg, W, synth.SCANNET_MEAN_SIZES, dtype=torch.float64)
    od.prepare(sc["tokens"], sc["camera"], sc["T_camera_pseudoCam"], sc["T_world_pseudoCam"], sc["T_world_local"])
    Wd = od.W
    worst = {}
    with torch.no_grad():
        dec.prepare(*scene_args(sc))
        for k in range(G.num_iters(z)):
            _, ref_next = dec.iterate(k, dev(refs[k]))
            ref = torch.from_numpy(refs[k]).double()
            _, _, it = od.iterate(ref, k)
            p = "parq_module.decoder.layers.%d." % (0 if cfg.TRANSFORMER.SHARE_WEIGHTS else k)
            d = "parq_module.decoder.position_encoder."
            C = it["tgt"].shape[-1]
            Wi, bi = Wd[p + "self_attn.in_proj_weight"], Wd[p + "self_attn.in_proj_bias"]
            Wc, bc = Wd[p + "multihead_attn.in_proj_weight"], Wd[p + "multihead_attn.in_proj_bias"]

            def ln(x, n):
                return O.layer_norm(x.cpu().double().reshape(it["x"].shape), Wd[p + n + ".weight"], Wd[p + n + ".bias"])
            want = {
                "posemb": (dec.intermediate("posemb"), O.pos2posemb3d(ref_next.cpu().double()), 5e-6),
                "pos_hidden": (dec.intermediate("pos_hidden"), F.relu(F.linear(O.pos2posemb3d(ref), Wd[d + "0.weight"], Wd[d + "0.bias"])), 5e-6),
                "tgt": (dec.intermediate("tgt"), it["tgt"], 5e-6),
                "self_qkv": (dec.intermediate("self_qkv"),
                             torch.cat([F.linear(it["tgt"] + it["pos"], Wi[:2 * C], bi[:2 * C]), F.linear(it["tgt"], Wi[2 * C:], bi[2 * C:])], -1), 5e-6),
                "norm1(xa_prenorm1)": (ln(dec.intermediate("xa_prenorm1"), "norm1"), it["x1"], 5e-6),
                "cross_q": (dec.intermediate("cross_q"), F.linear(it["x1"] + it["pos"], Wc[:C], bc[:C]), 5e-6),
                "norm2(xb_prenorm2)": (ln(dec.intermediate("xb_prenorm2"), "norm2"), it["x2"], 2e-5),
                "norm3(xc_prenorm3)": (ln(dec.intermediate("xc_prenorm3"), "norm3"), it["x"], 2e-5),
            }
            for nm, (a, r, tol) in want.items():
                e = _err(a, r)
                worst[nm] = max(worst.get(nm, 0.0), e)
                assert e < tol, (name, k, nm, e)
    print("\n%s stage errors:" % name, {k: "%.1e" % v for k, v in worst.items()})
