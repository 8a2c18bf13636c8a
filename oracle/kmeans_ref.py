"""ORACLE (test infrastructure only): Lloyd's k-means as the device runs it (ae_wavenet_amd/kmeans.py), on the exact
C chain of oracle/exact.py, and its relation to what the reference calls.

The reference initialises the codebook with scipy.cluster.vq.kmeans(samples, n_codes) (autoencoder_model.py:194;
scipy is a third-party dependency, 1.x, not vendored).  Its published algorithm (`_kmeans`): assign every sample to
the nearest code (Euclidean), move each code to the mean of its samples, stop when the mean distortion changes by
less than `thresh`; codes without samples are dropped; `kmeans` draws the initial codes from the samples and keeps
the best of `iter` restarts, or runs once from a caller-supplied initial codebook.  lloyd() below is the same
iteration from a given initial codebook (the form tests pin against scipy), except that a code without samples
keeps its position so that the codebook keeps its size.
"""
import numpy as np

from . import exact


def lloyd(samples, init, n_iter):
    """n_iter iterations (assignment + centroid step).  Returns codes, and per iteration the assignment and the mean
    squared distance before the centroid step."""
    x = np.ascontiguousarray(samples, np.float32)
    emb = np.array(init, np.float32, copy=True)
    K = emb.shape[0]
    hist = []
    for _ in range(n_iter):
        ind, dist, _ = exact.vq_nearest(x, emb, "sq_l2")
        z_sum, n_sum = exact.vq_stats(x, ind, K)
        has = n_sum > 0
        emb[has] = (z_sum[has] / n_sum[has, None]).astype(np.float32)
        hist.append((ind, float(dist.astype(np.float64).mean())))
    return emb, hist
