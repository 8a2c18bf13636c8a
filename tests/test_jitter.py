"""Device time-jitter generator against its oracle and against the reference's Jitter class.

The device uses a counter RNG, so its stream cannot equal numpy's MT19937 draw for draw; what is pinned:
  * oracle/jitter_rng.py restates the kernel's arithmetic -> bit-identical indices (GPU test);
  * tests/golden/jitter_stats.json holds statistics of the reference class itself (jitter.py:13-33, numpy
    seed 0, written by make_golden.py): structure, offset frequencies, pair frequencies; the generator is
    held to them within sampling error.  They also document that at HEAD the reference's
    "no three in a row" rule never fires (it indexes cond2d[p1][p1]): mode 0 reproduces that, mode 1 the
    documented rule."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import jitter_rng

HERE = os.path.dirname(os.path.abspath(__file__))
STATS = json.load(open(os.path.join(HERE, "golden", "jitter_stats.json")))


def _three_equal(idx):
    return int(((idx[:, :-2] == idx[:, 1:-1]) & (idx[:, 1:-1] == idx[:, 2:])).sum())


@pytest.mark.parametrize("p", [0.12, 0.3])
def test_distribution_matches_the_reference_class(p):
    ref = STATS[str(p)]
    n, reps = ref["n"], 4000
    idx = jitter_rng.device_indices(seed=2507, step=3, B=reps, n=n, p=p, mode=0)
    off = idx - np.arange(n)[None, :]
    assert ref["first_two_identity"] and (off[:, :2] == 0).all()
    assert off.min() == ref["min_off"] == -1 and off.max() == ref["max_off"] == 1
    body = off[:, 2:]
    tot = body.size
    ref_tot = sum(ref["counts"])
    for v, c in zip((-1, 0, 1), ref["counts"]):
        f_ref, f = c / ref_tot, (body == v).sum() / tot
        sigma = np.sqrt(f_ref * (1 - f_ref) * (1 / ref_tot + 1 / tot))
        assert abs(f - f_ref) < 4.5 * sigma, (v, f, f_ref)
    # consecutive draws: the reference's table gives the same row for every history (iid) - pair frequencies
    pc = ref["pair_counts"]
    pairs_tot = sum(pc.values())
    a, b = body[:, :-1].reshape(-1), body[:, 1:].reshape(-1)
    for key, c in pc.items():
        va, vb = (int(x) for x in key.split(","))
        f_ref, f = c / pairs_tot, ((a == va) & (b == vb)).mean()
        sigma = np.sqrt(f_ref * (1 - f_ref) * (1 / pairs_tot + 1 / a.size))
        assert abs(f - f_ref) < 5 * sigma, (key, f, f_ref)
    # ... so three equal indices in a row DO occur at HEAD, in the reference and here
    assert ref["same_nonzero_triples"] > 0 and _three_equal(idx) > 0


def test_intended_rule_forbids_three_equal_indices():
    idx = jitter_rng.device_indices(seed=1, step=0, B=2000, n=72, p=0.3, mode=1)
    assert _three_equal(idx) == 0
    off = idx - np.arange(72)[None, :]
    assert (off[:, :2] == 0).all() and off.min() == -1 and off.max() == 1
    # the renormalised row only removes mass from offset -1 after (+1, 0): offset -1 becomes rarer than +1
    body = off[:, 2:]
    assert (body == -1).sum() < (body == 1).sum()


def test_stream_depends_on_seed_step_row_only():
    a = jitter_rng.device_indices(7, 5, 4, 40, 0.2)
    assert np.array_equal(a, jitter_rng.device_indices(7, 5, 4, 40, 0.2))
    assert not np.array_equal(a, jitter_rng.device_indices(8, 5, 4, 40, 0.2))
    assert not np.array_equal(a, jitter_rng.device_indices(7, 6, 4, 40, 0.2))
    assert np.array_equal(a[:2], jitter_rng.device_indices(7, 5, 2, 40, 0.2))       # rows independent of B
    assert np.array_equal(a[:, :25], jitter_rng.device_indices(7, 5, 4, 25, 0.2))   # prefix-stable in n (mode 0)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
def test_device_equals_oracle_bit_for_bit(mode):
    from ae_wavenet_amd.jitter import DeviceJitter
    for seed, B, n, p in ((2507, 8, 72, 0.12), (2 ** 63 + 11, 3, 450, 0.3), (0, 64, 29, 0.5), (5, 1, 2, 0.1)):
        jit = DeviceJitter(p, seed=seed, intended=bool(mode))
        for call in range(3):
            got = jit(B, n, "cuda:0").cpu().numpy()
            want = jitter_rng.device_indices(seed, call, B, n, p, mode)
            assert got.dtype == np.int64 and np.array_equal(got, want), (seed, B, n, p, call)


@pytest.mark.gpu
def test_prefetcher_delivers_batches_in_order_with_device_jitter():
    from ae_wavenet_amd.jitter import DeviceJitter
    from ae_wavenet_amd.loader import DevicePrefetcher
    g = torch.Generator().manual_seed(0)
    host = [(torch.randint(0, 256, (4, 300), generator=g).float(), torch.randn(4, 39, 21 + k, generator=g),
             torch.randint(0, 40, (4,), generator=g), torch.zeros(4, 21 + k, dtype=torch.int64), ("path", k))
            for k in range(7)]
    jit = DeviceJitter(0.12, seed=9)
    got = list(DevicePrefetcher(iter(host), "cuda:0", depth=2, jitter=jit))
    assert len(got) == len(host)
    for k, (h, d) in enumerate(zip(host, got)):
        assert d[0].is_cuda and torch.equal(d[0].cpu(), h[0]) and torch.equal(d[1].cpu(), h[1])
        assert torch.equal(d[2].cpu(), h[2]) and d[4] == ("path", k)
        want = jitter_rng.device_indices(9, k, 4, 21 + k, 0.12, 0)       # call k of the generator
        assert np.array_equal(d[3].cpu().numpy(), want)
