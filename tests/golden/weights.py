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


N_SKETCH = 32


def grad_sketch(names, grads):
    """Whole-tensor fingerprints of full-width gradients that are too large to commit: for every tensor (visited in
    sorted name order, tensor i sketched with RandomState(7000 + i)) N_SKETCH projections onto unit-variance normal
    vectors, scaled by 1/sqrt(N_SKETCH).  For any two tensors a, b:  |S a - S b| / |S a| estimates the relative L2
    distance |a - b| / |a| (Johnson-Lindenstrauss; +-25 % at 32 projections) over the WHOLE tensor - every element
    counts, which an 8 x 8 corner does not give.  Returns {name: float32[N_SKETCH]}."""
    out = {}
    for i, k in enumerate(sorted(names)):
        g = np.asarray(grads[k], dtype=np.float32).reshape(-1)
        rs = np.random.RandomState(7000 + i)
        sk = np.empty(N_SKETCH, np.float64)
        for j in range(N_SKETCH):                              # one row at a time: the largest tensor has 188 k elements
            sk[j] = float(np.dot(rs.standard_normal(g.size).astype(np.float32).astype(np.float64), g.astype(np.float64)))
        out[k] = (sk / np.sqrt(N_SKETCH)).astype(np.float32)
    return out
