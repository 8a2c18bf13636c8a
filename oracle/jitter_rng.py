"""ORACLE (test infrastructure only): numpy restatement of the device jitter generator
(aew_jitter_t / k_jitter, include/aewavenet.h) and of the reference's Jitter class.

reference_like(): what jitter.py:13-33 computes, with the random draw left as a parameter.  At HEAD the
conditional table is indexed `cond2d[p1][p1]` (jitter.py:30), so the special row `cond2d[2][1]` is never
read and every draw is iid [p, 1-2p, p]; `intended=True` indexes `[p2][p1]` as the docstring describes.
device_indices(): the same chain driven by the device's counter RNG u(b,t) = mix64(seed, step, b, t) / 2^53
- the arithmetic of k_jitter, integer for integer, so outputs are bit-identical.
"""
import numpy as np

_M = (1 << 64) - 1


def mix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xbf58476d1ce4e5b9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94d049bb133111eb)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform(seed, step, B, n):
    """u[b][t] in [0,1) as double, exactly as aew_jitter_u()."""
    with np.errstate(over="ignore"):
        h = mix64(np.uint64(seed & _M) + np.uint64(0x9e3779b97f4a7c15))
        h = mix64(h ^ (np.uint64(step & _M) + np.uint64(0x9e3779b97f4a7c15)))
        b = np.arange(B, dtype=np.uint64)[:, None]
        t = np.arange(n, dtype=np.uint64)[None, :]
        h = mix64(h ^ ((b << np.uint64(32)) | t))
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def chain(u, p, intended=False):
    """X[b][t] in {0,1,2} from uniforms u (jitter.py:25-31 with cumulative-table sampling)."""
    p = float(np.float32(p))
    s = 1.0 - 2.0 * p
    B, n = u.shape
    x = np.ones((B, n), dtype=np.int64)
    for b in range(B):
        x2 = x1 = 1
        for t in range(n):
            v = 1
            if t >= 2:
                c0, c1 = p, p + s
                if intended and x2 == 2 and x1 == 1:
                    c0, c1 = 0.0, s / (p + s)
                v = int(u[b, t] >= c0) + int(u[b, t] >= c1)
            x[b, t] = v
            x2, x1 = x1, v
    return x


def device_indices(seed, step, B, n, p, mode=0):
    x = chain(uniform(seed, step, B, n), p, intended=(mode == 1))
    return x + np.arange(-1, n - 1, dtype=np.int64)[None, :]
