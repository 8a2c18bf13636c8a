"""Seeded weight/inputs generators shared by make_golden.py (runs with the reference) and
the tests (run without it).  numpy's legacy RandomState is bit-stable across versions."""
import numpy as np


def np_weights(shapes, seed, scale=None):
    """uniform(-a, a) per tensor with the Xavier bound (or `scale`); 1-D tensors get small
    non-zero values so bias paths are exercised.  Keys are visited in sorted order."""
    rs = np.random.RandomState(seed)
    out = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        if len(shp) >= 2:
            rf = int(np.prod(shp[2:])) if len(shp) > 2 else 1
            bound = scale if scale is not None else float(np.sqrt(6.0 / ((shp[0] + shp[1]) * rf)))
        else:
            bound = 0.1
        out[k] = rs.uniform(-bound, bound, size=shp).astype(np.float32)
    return out
