"""Several forwards of one module in flight, one HIP stream each.

The reference calls its model once per batch on the default stream (eval.py:46, model/parq_lightning.py:68-95).  A call into this
package only ENQUEUES; at one scene per call the decoder's recurrent chain of small launches leaves most of the chip idle, and a
second scene's K/V projection / cross-attention fills it: two scenes in flight run +15-18 % decoder-iterations/s at BASELINE cfg 3
(profiles/r05_two_in_flight.txt; a third adds nothing).  ``InFlight`` is the small amount of stream plumbing that takes:

    runner = InFlight(model, depth=2)            # model: PARQDecoder, AddRayPE, PARQ - anything whose call enqueues
    tickets = [runner.submit(*batch) for batch in batches[:2]]
    for nxt in batches[2:]:
        out = tickets.pop(0).result()             # waits for THAT forward (the other one runs meanwhile), checks it, orders the caller's stream behind it
        ...                                       # consume `out` on the current stream
        tickets.append(runner.submit(*nxt))

Inference only (call it under ``torch.no_grad()``); results are bit for bit those of one-at-a-time calls (tests/test_gpu_streams.py).
Each stream owns a workspace of the module (PARQDecoder keys its workspace cache by stream): memory grows with ``depth``.
"""
import torch

__all__ = ["InFlight", "Ticket"]


def _storages(obj):
    """One CUDA tensor per distinct storage reachable from `obj` (the 48 output tensors of a decoder forward are views of ONE
    allocation: telling the caching allocator once is enough, and 47 calls cheaper)."""
    seen, out = set(), []
    for t in _tensors(obj):
        if t.is_cuda:
            key = t.untyped_storage().data_ptr()
            if key not in seen:
                seen.add(key)
                out.append(t)
    return out


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
    """One submitted forward.  ``result()`` hands out its outputs, ordered in front of whatever the caller enqueues next on its
    current stream.  Under the decoder's default policy (``range_check = "sync"``: a forward never returns NaN, it is re-run with safer
    arithmetic instead) the check that a one-at-a-time call makes before it returns is made HERE: ``result()`` waits on the host for
    this forward — the other forwards in flight keep the device busy meanwhile — and, if the device flagged it, re-runs it on its
    stream before handing the outputs out.  Under ``range_check = "lazy"`` / ``"off"`` nothing waits on the host; ``valid()`` tells
    whether the outputs are numbers."""

    def __init__(self, outputs, event, stream, settle=(), storages=None):
        self._out, self._ev, self._stream, self._settle = outputs, event, stream, list(settle)
        self._storages = storages                # one tensor per allocation behind `outputs`, if the submitter knows them

    def done(self):
        return self._ev.query()

    def result(self):
        if self._settle:
            with torch.cuda.stream(self._stream):
                for fn in self._settle:         # host wait + the pinned word of this forward's workspace; a flagged forward is re-run
                    fn()
                self._settle = []
                self._ev = torch.cuda.Event()
                self._ev.record(self._stream)
        cur = torch.cuda.current_stream(self._stream.device)
        cur.wait_event(self._ev)
        for t in (self._storages if self._storages is not None else _storages(self._out)):
            t.record_stream(cur)                # allocated on the side stream, consumed on the caller's: tell the caching allocator
        return self._out

    def valid(self):
        """True if every floating-point output of this forward is finite (waits for the forward; one small reduction per tensor).
        For callers that run ``range_check = "lazy"``: a forward that met inputs outside its arithmetic's guarantees is all NaN."""
        self._ev.synchronize()
        with torch.cuda.stream(self._stream):
            return all(bool(torch.isfinite(t).all()) for t in _tensors(self._out) if t.is_cuda and t.is_floating_point())


def _decoders(module):
    """The PARQDecoder(s) inside `module` (itself, or the ``box3d_decoder`` of a PARQ module)."""
    out = []
    for m in (module, getattr(module, "box3d_decoder", None)):
        if m is not None and hasattr(m, "_defer") and hasattr(m, "range_check"):
            out.append(m)
    return out


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
        self._decs = _decoders(module)
        for dec in self._decs:                   # one workspace per stream stays cached (PARQDecoder)
            dec.max_workspaces = max(int(dec.max_workspaces), depth)

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
        for t in _storages((args, kwargs)):      # the arguments were allocated on the caller's stream and are read on `side`
            t.record_stream(side)
        decs = self._decs
        settle = []
        for d in decs:
            d.__dict__["_defer"] = settle        # the decoder hands its post-forward check over instead of waiting inside the call
            d.__dict__["_last_flat"] = None      # (plain dict writes: nn.Module.__setattr__ costs 2.5 us a piece)
        try:
            with torch.cuda.stream(side):
                out = self.module(*args, **kwargs)
                ev = torch.cuda.Event()
                ev.record(side)
        finally:
            for d in decs:
                d.__dict__["_defer"] = None
        # the decoder's 48 output tensors are views of one allocation: hand the ticket that allocation instead of letting it walk them
        flats = [d.__dict__.get("_last_flat") for d in decs]
        own = [f for f in flats if f is not None] if (decs and (self.module is decs[0] or getattr(self.module, "box3d_decoder", None) is decs[0])) else None
        return Ticket(out, ev, side, settle, own or None)

    def drain(self):
        for s in self._streams:
            s.synchronize()
