"""Host-side mirror of the reference BoxProcessor's mean-size table
(utils/parq_utils.py:45-88).  The arithmetic (softmax, arg-max gather, exp) runs
on the device inside the HIP box-decode kernel; this module only builds the
(10,3) table the kernel gathers from."""
from __future__ import annotations

import numpy as np

from .synth import SCANNET_MEAN_SIZES

# class id -> name, utils/parq_utils.py:47-57 ("other" has no file entry)
_CLASS_NAMES = ["chair", "table", "cabinet", "trash bin", "bookshelf", "display", "sofa", "bathtub", "other"]


def parse_mean_size_file(path: str) -> np.ndarray:
    """Parse a ``name[,alias...]: [sx sy sz]`` file into the reference's table:
    one row per named class found (in class-id order), then two [1,1,1] rows
    ("other", "non-object")."""
    by_name = {}
    with open(path, "r") as f:
        for line in f:
            if ": " not in line:
                continue
            names, size = line.split(": ", 1)
            vals = [float(tok) for tok in size.strip().strip("[]").split()]
            by_name[names] = vals[:3]
    rows = []
    for cname in _CLASS_NAMES:
        for names, vals in by_name.items():
            if cname in names.split(","):
                rows.append(vals)
                break
    rows.append([1.0, 1.0, 1.0])
    rows.append([1.0, 1.0, 1.0])
    return np.asarray(rows, dtype=np.float64)


def mean_size_table(path=None) -> np.ndarray:
    return parse_mean_size_file(path) if path else SCANNET_MEAN_SIZES.copy()
