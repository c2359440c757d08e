"""Several forwards of one module in flight, one HIP stream each.

The reference calls its model once per batch on the default stream (eval.py:46, model/parq_lightning.py:68-95).  A call into this
package only ENQUEUES; at one scene per call the decoder's recurrent chain of small launches leaves most of the chip idle, and a
second scene's K/V projection / cross-attention fills it: two scenes in flight run +15-18 % decoder-iterations/s at BASELINE cfg 3
(profiles/r05_two_in_flight.txt; a third adds nothing).  ``InFlight`` is the small amount of stream plumbing that takes:

    runner = InFlight(model, depth=2)            # model: PARQDecoder, AddRayPE, PARQ - anything whose call enqueues
    tickets = [runner.submit(*batch) for batch in batches[:2]]
    for nxt in batches[2:]:
        out = tickets.pop(0).result()             # orders the CALLER's current stream behind that forward
        ...                                       # consume `out` on the current stream
        tickets.append(runner.submit(*nxt))

Inference only (call it under ``torch.no_grad()``); results are bit for bit those of one-at-a-time calls (tests/test_gpu_streams.py).
Each stream owns a workspace of the module (PARQDecoder keys its workspace cache by stream): memory grows with ``depth``.
"""
import torch

__all__ = ["InFlight", "Ticket"]


def _tensors(obj):
    """Tensors reachable from the arguments / results of a call: tensors, the package's wrappers (``_data``), containers."""
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(getattr(obj, "_data", None), torch.Tensor):
        yield obj._data
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors(v)


class Ticket:
    """One submitted forward.  ``result()`` makes the caller's current stream wait for it and hands out its outputs."""

    def __init__(self, outputs, event, stream):
        self._out, self._ev, self._stream = outputs, event, stream

    def done(self):
        return self._ev.query()

    def result(self):
        cur = torch.cuda.current_stream(self._stream.device)
        cur.wait_event(self._ev)
        for t in _tensors(self._out):           # allocated on the side stream, consumed on the caller's: tell the caching allocator
            if t.is_cuda:
                t.record_stream(cur)
        return self._out


class InFlight:
    def __init__(self, module, depth=2, device=None):
        if depth < 1:
            raise ValueError("InFlight: depth >= 1")
        if device is None:
            p = next(module.parameters(), None) if hasattr(module, "parameters") else None
            device = p.device if p is not None and p.is_cuda else torch.device("cuda", torch.cuda.current_device())
        self.module, self.device = module, torch.device(device)
        self._streams = [torch.cuda.Stream(self.device) for _ in range(depth)]
        self._next = 0
        if hasattr(module, "max_workspaces"):    # one workspace per stream stays cached (PARQDecoder)
            module.max_workspaces = max(int(module.max_workspaces), depth)
        dec = getattr(getattr(module, "box3d_decoder", None), "max_workspaces", None)
        if dec is not None:
            module.box3d_decoder.max_workspaces = max(int(dec), depth)

    @property
    def depth(self):
        return len(self._streams)

    def submit(self, *args, **kwargs):
        """Enqueue ``module(*args, **kwargs)`` on the next stream, behind whatever the caller's current stream has enqueued so far
        (the producers of the arguments).  Returns a Ticket."""
        side = self._streams[self._next]
        self._next = (self._next + 1) % len(self._streams)
        cur = torch.cuda.current_stream(self.device)
        side.wait_stream(cur)
        for t in _tensors((args, kwargs)):       # the arguments were allocated on the caller's stream and are read on `side`
            if t.is_cuda:
                t.record_stream(side)
        with torch.cuda.stream(side):
            out = self.module(*args, **kwargs)
            ev = torch.cuda.Event()
            ev.record(side)
        return Ticket(out, ev, side)

    def drain(self):
        for s in self._streams:
            s.synchronize()
