"""k-means codebook initialisation on the device (autoencoder_model.py:171-199).

The reference collects encoder outputs on the host and calls scipy.cluster.vq.kmeans(samples, n_codes): random
initial codes drawn from the samples, Lloyd iterations until the mean distortion stops improving.  Here the samples
stay in HBM and one Lloyd iteration is a three-op launch plan built from the ops the training step already has:

    VQ_NEAREST (squared L2, exact fp32 chain)  ->  VQ_STATS (z_sum, n_sum, deterministic order)
    ->  VQ_EMA with gamma = 0, gamma_comp = 1, update_codebook = 2   (emb[k] = z_sum[k] / n_sum[k] where n_sum[k] > 0)

plus a REDUCE of the distances for the convergence test.  Differences from scipy, on purpose: a code that loses all
its samples keeps its position (scipy drops it and returns fewer codes, which the reference's `bn.emb[...] = km`
cannot store); one run from one initial draw (scipy keeps the best of 20 restarts: pass n_init > 1 for that).
oracle/kmeans_ref.py restates the iteration on the exact C chain; the device result is bit-identical to it.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib as L
from .plan import Plan, Workspace


class DeviceKMeans:
    def __init__(self, n: int, d: int, K: int, device, n_split: int = 4):
        self.n, self.d, self.K = n, d, K
        self.dev = torch.device(device)
        ws = self.ws = Workspace(device)
        f32, i64 = torch.float32, torch.int64
        self.samples = ws.alloc("km.samples", n * d, f32)
        self.emb = ws.alloc("km.emb", K * d, f32)
        self.ind = ws.alloc("km.ind", n, i64)
        self.dist = ws.alloc("km.dist", n, f32)
        self.zq = ws.alloc("km.zq", n * d, f32)
        self.z_sum = ws.alloc("km.z_sum", K * d, f32)
        self.n_sum = ws.alloc("km.n_sum", K, f32)
        self.numer = ws.alloc("km.numer", K * d, f32)
        self.denom = ws.alloc("km.denom", K, f32)
        self.red = ws.alloc("km.red", 8, f32)
        self.scratch = ws.alloc("km.scratch", 2 * n * n_split, i64)
        p = self.plan = Plan("kmeans")
        vn = L.VqNearest()
        vn.ze, vn.emb, vn.Q, vn.K, vn.d, vn.d_pitch, vn.metric = (self.samples.data_ptr(), self.emb.data_ptr(), n, K,
                                                                  d, d, 1)
        vn.ind, vn.dist, vn.zq = self.ind.data_ptr(), self.dist.data_ptr(), self.zq.data_ptr()
        vn.scratch, vn.n_split = self.scratch.data_ptr(), n_split
        p.add(L.OP_VQ_NEAREST, vn, "km.assign")
        rd = L.Reduce()
        rd.n_terms = 1
        rd.x[0], rd.n[0], rd.scale[0] = self.dist.data_ptr(), n, 1.0 / n
        rd.out = self.red.data_ptr()
        p.add(L.OP_REDUCE, rd, "km.distortion")
        vs = L.VqStats()
        vs.ze, vs.ind, vs.Q, vs.K, vs.d, vs.d_pitch = self.samples.data_ptr(), self.ind.data_ptr(), n, K, d, d
        vs.z_sum, vs.n_sum = self.z_sum.data_ptr(), self.n_sum.data_ptr()
        p.add(L.OP_VQ_STATS, vs, "km.stats")
        em = L.VqEma()
        em.numer, em.denom, em.z_sum, em.n_sum = (self.numer.data_ptr(), self.denom.data_ptr(), self.z_sum.data_ptr(),
                                                  self.n_sum.data_ptr())
        em.emb, em.K, em.d, em.update_codebook, em.gamma, em.gamma_comp = self.emb.data_ptr(), K, d, 2, 0.0, 1.0
        p.add(L.OP_VQ_EMA, em, "km.centroids")

    def fit(self, samples: torch.Tensor, init: torch.Tensor, max_iter: int = 100, thresh: float = 1e-5,
            check_every: int = 4, stream: int = 0) -> Tuple[torch.Tensor, float, int]:
        """samples [n][d] fp32, init [K][d] initial codes.  Runs Lloyd iterations until the mean squared distance
        to the nearest code improves by less than `thresh` (relative) or max_iter is reached; the convergence value
        is read back only every `check_every` iterations.  Returns (codes [K][d], mean squared distance of the
        assignment that produced them, iterations run)."""
        n, d, K = self.n, self.d, self.K
        if tuple(samples.shape) != (n, d) or tuple(init.shape) != (K, d):
            raise L.AewError("kmeans: shape")
        self.samples[:n * d].copy_(samples.reshape(-1).to(self.dev, torch.float32))
        self.emb[:K * d].copy_(init.reshape(-1).to(self.dev, torch.float32))
        self.numer.zero_()
        self.denom.zero_()
        prev, it, cur = None, 0, float("inf")
        while it < max_iter:
            self.plan.run(stream)
            it += 1
            if it % check_every == 0 or it == max_iter:
                cur = float(self.red[1].item())
                if prev is not None and prev - cur <= thresh * max(prev, 1e-30):
                    break
                prev = cur
        if cur == float("inf"):
            cur = float(self.red[1].item())
        return self.emb[:K * d].view(K, d).clone(), cur, it


def initial_codes(samples: torch.Tensor, K: int, seed: int) -> torch.Tensor:
    """K distinct samples (scipy's `_kpoints`, with a seeded torch generator instead of numpy's global RNG)."""
    g = torch.Generator().manual_seed(seed)
    idx = torch.randperm(samples.shape[0], generator=g)[:K]
    return samples[idx.to(samples.device)].clone()


def kmeans(samples: torch.Tensor, K: int, seed: int = 0, n_init: int = 1, max_iter: int = 100, thresh: float = 1e-5,
           km: Optional[DeviceKMeans] = None):
    """Best of n_init runs.  samples on the device."""
    n, d = samples.shape
    km = km or DeviceKMeans(n, d, K, samples.device)
    best = None
    for r in range(n_init):
        codes, dist, it = km.fit(samples, initial_codes(samples, K, seed + r), max_iter, thresh)
        if best is None or dist < best[1]:
            best = (codes, dist, it)
    return best
