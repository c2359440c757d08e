"""GPU tier: forwards of ONE module in flight on several HIP streams.  The reference's decoder is called once per scene on the
default stream (model/parq_decoder.py:44-62 through eval.py:46); here a call only enqueues, and a server may keep two scenes in flight
so that the small-op chain of one runs beside the K/V projection / cross-attention of the other.  Contract: every stream's forward
owns its workspace (PARQDecoder._workspace is keyed by the launch stream) and the results are bit for bit those of serial calls."""
import pytest
import torch

from parq_amd import synth
from gpu_util import make_decoder, scene_args

pytestmark = pytest.mark.gpu


def _scene(seed, V, h, w, dim):
    return synth.make_scene(seed, 1, V, h, w, dim, smooth=True)


@pytest.mark.parametrize("mode", ["split8", "split"])
def test_two_scenes_in_flight_on_two_streams_equal_serial_forwards(mode):
    V, h, w, Q, dim, I = 4, 60, 80, 64, 256, 3
    cfg = synth.decoder_cfg(dim=dim, queries=Q, heads=4, ffn=256, layers=I)
    W = synth.make_decoder_weights(cfg, 311, damped=True)
    dec = make_decoder(cfg, W).eval()
    dec.attention_mode = mode
    dec.range_check = "off"
    scenes = [_scene(312 + i, V, h, w, dim) for i in range(2)]
    args = [scene_args(sc) for sc in scenes]
    with torch.no_grad():
        serial = []
        for a in args:
            outs = dec(*a, feat_hw=(h, w))
            serial.append([{k: v.clone() for k, v in o.items()} for o in outs])
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        for s in streams:
            s.wait_stream(torch.cuda.current_stream())
        for rep in range(6):                                   # alternating enqueue: both forwards are in flight together
            got = [None, None]
            for i in (0, 1):
                with torch.cuda.stream(streams[i]):
                    outs = dec(*args[i], feat_hw=(h, w))
                    got[i] = [{k: v.clone() for k, v in o.items()} for o in outs]
            torch.cuda.synchronize()
            for i in (0, 1):
                for k in range(I):
                    for key in serial[i][k]:
                        assert torch.equal(got[i][k][key], serial[i][k][key]), (rep, i, k, key)
    keys = list(dec._ws)
    assert len(keys) >= 2 and len({kk[-1] for kk in keys}) >= 2          # one workspace per stream


def test_workspace_of_a_stream_is_reused_by_that_stream_only():
    V, h, w, Q, dim, I = 2, 12, 16, 16, 256, 2
    cfg = synth.decoder_cfg(dim=dim, queries=Q, heads=4, ffn=128, layers=I)
    W = synth.make_decoder_weights(cfg, 321, damped=True)
    dec = make_decoder(cfg, W).eval()
    dec.range_check = "off"                                    # (384 keys: every row rests on few keys)
    a = scene_args(_scene(322, V, h, w, dim))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.no_grad():
        with torch.cuda.stream(s1):
            dec(*a, feat_hw=(h, w))
            ws1 = next(reversed(dec._ws.values()))
            dec(*a, feat_hw=(h, w))
            assert next(reversed(dec._ws.values())) is ws1
        with torch.cuda.stream(s2):
            dec(*a, feat_hw=(h, w))
            assert next(reversed(dec._ws.values())) is not ws1
    torch.cuda.synchronize()
    assert len(dec._ws) == 2
