"""Shared helpers for the golden-vector tests (inputs are regenerated from the
seeds stored in each fixture; see oracle/make_golden.py)."""
import json
import os

import numpy as np

from parq_amd import synth
from oracle import make_golden as MG

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KEYS = MG.KEYS


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    case = json.loads(bytes(z["meta"]).decode())
    return case, z


def inputs(case):
    return MG.case_inputs(case)


def num_iters(z):
    return len([k for k in z.files if k.endswith("_coord_pos")])


def forced_refs(z, scale, dtype=np.float32):
    """norm(coord_pos_k): the reference's own per-iteration input reference points."""
    lo = np.asarray(scale[0::2], dtype)
    hi = np.asarray(scale[1::2], dtype)
    return [((z["it%d_coord_pos" % k].astype(dtype) - lo) / (hi - lo)).astype(dtype)
            for k in range(num_iters(z))]


def safe_mask(z, k, valid_margin=2e-3, cls_margin=1e-3):
    """(B,Q) mask of queries whose discrete decisions (valid-view count, arg-max
    class) are not within rounding of flipping."""
    return (z["it%d_valid_margin" % k] > valid_margin), (z["it%d_cls_margin" % k] > cls_margin)


def compare(out, z, k, tol, what=""):
    """max-abs error of the six outputs of iteration k against the golden, with
    the discontinuous elements masked out.  Error metric: |a-b| / max(1,|b|)."""
    vm, cm = safe_mask(z, k)
    worst = {}
    for key in KEYS:
        a = np.asarray(out[key], dtype=np.float64)
        b = z["it%d_%s" % (k, key)].astype(np.float64)
        assert a.shape == b.shape, (key, a.shape, b.shape)
        err = np.abs(a - b) / np.maximum(1.0, np.abs(b))
        m = vm.copy()
        if key == "size_unnormalized":
            m &= cm
        if key == "coord_pos":
            m = np.ones_like(vm)
        e = err[m]
        worst[key] = float(e.max()) if e.size else 0.0
    bad = {k2: v for k2, v in worst.items() if not v <= tol}
    assert not bad, "%s iteration %d exceeds %g: %s (all: %s)" % (what, k, tol, bad, worst)
    return worst


# ---------------------------------------------------------------------------------------------------------------------------
# g17: gradients of the reference's own autograd (oracle/make_golden.py::make_grad_golden)

def load_grads():
    z = np.load(os.path.join(GOLDEN_DIR, "g17_grads.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    return meta, z


def grad_names(z, tag, kind):
    pre = "%s/%s/grad/" % (tag, kind)
    return sorted({k[len(pre):].rsplit("/", 1)[0] for k in z.files if k.startswith(pre)})


def grad_errors(z, tag, kind, name, g):
    """(Frobenius-relative, max-relative) error of gradient tensor `g` against what the fixture keeps of the reference's:
    the whole tensor, or its norm and every GRAD_STRIDE-th element."""
    g = np.asarray(g, np.float64).reshape(-1)
    pre = "%s/%s/grad/%s/" % (tag, kind, name)
    if pre + "full" in z.files:
        ref = z[pre + "full"]
        d = g - ref
        return np.linalg.norm(d) / max(np.linalg.norm(ref), 1e-300), np.abs(d).max() / max(np.abs(ref).max(), 1e-300)
    ref = z[pre + "sample"]
    d = g[::MG.GRAD_STRIDE] - ref
    nrm = z[pre + "norm"]
    fro = max(np.linalg.norm(d) / max(np.linalg.norm(ref), 1e-300), abs(np.linalg.norm(g) - nrm[0]) / max(nrm[0], 1e-300))
    return fro, np.abs(d).max() / max(np.abs(ref).max(), 1e-300)


def token_grad_errors(z, tag, kind, g):
    g = np.asarray(g, np.float64).reshape(-1)
    ref = z["%s/%s/dtokens/sample" % (tag, kind)]
    d = g[::MG.TOKEN_STRIDE] - ref
    nrm = z["%s/%s/dtokens/norm" % (tag, kind)]
    fro = max(np.linalg.norm(d) / max(np.linalg.norm(ref), 1e-300), abs(np.linalg.norm(g) - nrm[0]) / max(nrm[0], 1e-300))
    return fro, np.abs(d).max() / max(np.abs(ref).max(), 1e-300)
