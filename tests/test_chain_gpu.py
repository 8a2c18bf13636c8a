"""Chained NT launches (AEW_OP_NT_CHAIN, csrc/aew_chain.hip) on the MI355X: the gated stack (wavenet.py:100-109, 354-357)
and its backward as launches of several dependent GEMMs with tile-granular hand-off must reproduce the one-launch-per-op
plan BIT FOR BIT - same kernel bodies, same summation order; a missing or too-narrow dependency, a store that was not
visible when its reader started, or a wait that gave up shows as a mismatch (or as the launch's timeout flag)."""
import ctypes as C

import numpy as np
import pytest
import torch

from ae_wavenet_amd import _lib as L, config, model as M, plan as PLN
from tests.test_gpu_parity import DEV, np_weights, seeded_full_engine

pytestmark = pytest.mark.gpu


def _mask(eng):
    """gradient elements that come out of deterministic paths (bias-type sums are fp32 atomics: round-off)"""
    atomic = [n for n in eng.ps.names() if n.endswith(".bias") or "speaker_embedding" in n]
    mask = torch.ones(eng.ps.numel, dtype=torch.bool, device=DEV)
    for n in atomic:
        o = (eng.ps.view(n, True).data_ptr() - eng.ps.grads.data_ptr()) // 4
        mask[o:o + eng.ps.numel_of(n)] = False
    return mask


def _step(eng):
    eng.init_ema_from_emb()
    loss = float(eng.forward())
    eng.backward()
    torch.cuda.synchronize()
    return loss, eng.logits().clone(), eng.ps.grads[:eng.ps.numel].clone()


def _chains(eng):
    return [lab for pl in (eng.fwd_b, eng.bwd) for lab in getattr(pl, "nt_chains", {})]


def _no_timeouts(eng):
    for pl in (eng.fwd_b, eng.bwd):
        assert PLN.chain_timeouts(pl) == [], f"a hand-off wait of plan {pl.name} gave up: {PLN.chain_stats(pl)}"


@pytest.mark.parametrize("n_chain", [64, 2])
def test_chained_stack_is_bit_identical_to_one_launch_per_op(monkeypatch, n_chain):
    """BASELINE configs[1] size (B = 8, w = 5000): whole stack per direction as one launch (64) and the pair of a layer
    per launch (2), as captured graphs (several replays) and eager."""
    monkeypatch.setattr(M.TrainEngine, "nt_chain", 0)
    monkeypatch.setattr(M.TrainEngine, "nt_chain_bwd", 0)
    monkeypatch.delenv("AEW_NT_CHAIN", raising=False)
    hps, eng0, wts, emb, inp = seeded_full_engine(B=8, w=5000, seed=11)
    wav, mel, voice, jitter = [t.to(DEV) for t in inp]
    eng0.set_inputs(wav, mel, voice, jitter)
    assert _chains(eng0) == []
    l_ref, lg_ref, g_ref = _step(eng0)
    mask = _mask(eng0)
    del eng0
    torch.cuda.empty_cache()
    monkeypatch.setattr(M.TrainEngine, "nt_chain", n_chain)
    monkeypatch.setattr(M.TrainEngine, "nt_chain_bwd", n_chain)
    _, eng, _, _, _ = seeded_full_engine(B=8, w=5000, seed=11)
    eng.set_inputs(wav, mel, voice, jitter)
    labs = _chains(eng)
    NL = len(eng.geom.layers)
    # forward: the stack (G1.0 .. G1.19: 2 NL - 1 ops) and the post network's pair; backward: d.post2, d.post1, dz / dx x NL
    assert len(labs) == (3 if n_chain >= 2 * NL + 2 else (NL - 1) + 1 + (NL + 1)), labs
    for rep in range(4):
        l, lg, g = _step(eng)
        _no_timeouts(eng)
        assert l == l_ref, (rep, l, l_ref)
        assert torch.equal(lg, lg_ref), rep
        assert torch.equal(g[mask], g_ref[mask]), rep
        # (ABI 20: the bias-type sums have one order as well - nothing is masked any more)
        assert torch.equal(g, g_ref), (rep, (g - g_ref).abs().max().item())
    eng.use_graphs = False
    l, lg, g = _step(eng)
    _no_timeouts(eng)
    assert l == l_ref and torch.equal(lg, lg_ref) and torch.equal(g[mask], g_ref[mask])
    # the launch-time switch: the same plan with chaining off runs its stage ops one by one
    t = L.current_tuning(nt_chain=0)
    eng.tuning = t
    l, lg, g = _step(eng)
    assert l == l_ref and torch.equal(lg, lg_ref) and torch.equal(g[mask], g_ref[mask])
    # the engine watches the launches' STICKY timeout word (copied to pinned host memory behind each plan, looked at before
    # the next forward): a wait that gave up raises instead of training on (tests/test_trust_gpu.py: with the host ahead)
    eng._chain_watch("check")                                   # nothing pending / nothing set: silent
    eng.chain_guard[:1].fill_(7)
    eng._chain_watch("fwd")
    torch.cuda.synchronize()
    with pytest.raises(L.AewError, match="gave up"):
        eng._chain_watch("check")
    eng.clear_chain_guard()


@pytest.mark.parametrize("B,w", [(3, 700), (2, 100)])
def test_chained_stack_with_waiting_tiles(monkeypatch, B, w):
    """Small windows: a stage is a few dozen tiles, so most of the 40 stages are resident at once and consumer tiles really
    spin on their producers' counters (at BASELINE size a producer has normally finished before its consumer starts).
    Odd batch: padding blocks in every stage.  The stand-alone launcher runs these sizes on its 64-row shapes; the chain
    is forced onto them (nt_chain_force) with the one-window form off, so both execute the plain k-ascending order."""
    lib = L.load()
    lib.aew_set_nt_window(0)
    try:
        monkeypatch.delenv("AEW_NT_CHAIN", raising=False)
        res = {}
        for n_chain in (0, 64):
            monkeypatch.setattr(M.TrainEngine, "nt_chain", n_chain)
            monkeypatch.setattr(M.TrainEngine, "nt_chain_bwd", n_chain)
            monkeypatch.setattr(M.TrainEngine, "nt_chain_force", True)
            hps, eng, wts, emb, inp = seeded_full_engine(B=B, w=w, seed=5)
            eng.set_inputs(*[t.to(DEV) for t in inp])
            assert (len(_chains(eng)) == 3) == (n_chain == 64)
            outs = []
            for rep in range(3):
                outs.append(_step(eng))
                _no_timeouts(eng)
            eng.use_graphs = False
            outs.append(_step(eng))
            _no_timeouts(eng)
            res[n_chain] = (outs, _mask(eng))
            del eng
            torch.cuda.empty_cache()
        (ref, mask), (got, _) = res[0], res[64]
        for (l0, lg0, g0), (l1, lg1, g1) in zip(ref, got):
            assert l0 == l1
            assert torch.equal(lg0, lg1)
            assert torch.equal(g0[mask], g1[mask])
    finally:
        lib.aew_set_nt_window(64)


def test_chained_stack_under_uneven_load(monkeypatch):
    """The guide's rule for every in-launch hand-off: test it under UNEVEN load.  While the chained forward and backward run,
    a second stream keeps the chip busy with work of its own (large device copies and fp32 matrix products in bursts of
    different length), so that tiles of the chain are delayed unevenly, producers and consumers drift apart, and more waits
    are real.  Results must stay bit-identical to the quiet run, no wait may give up."""
    monkeypatch.setattr(M.TrainEngine, "nt_chain", 64)
    monkeypatch.setattr(M.TrainEngine, "nt_chain_bwd", 64)
    monkeypatch.delenv("AEW_NT_CHAIN", raising=False)
    hps, eng, wts, emb, inp = seeded_full_engine(B=8, w=5000, seed=11)
    eng.set_inputs(*[t.to(DEV) for t in inp])
    assert len(_chains(eng)) == 3
    l_ref, lg_ref, g_ref = _step(eng)
    _no_timeouts(eng)
    mask = _mask(eng)
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=DEV)
    big = torch.empty(64 << 20, dtype=torch.float32, device=DEV)
    waited = []
    for rep in range(4):
        with torch.cuda.stream(side):
            for k in range(6 + 5 * rep):                      # bursts of different length per replay
                if k % 3 == 0:
                    big[:32 << 20].copy_(big[32 << 20:])
                else:
                    a = (a @ a).clamp_(-1, 1)
        l, lg, g = _step(eng)                                 # (synchronises the device at its end)
        _no_timeouts(eng)
        waited.append([v[1] for pl in (eng.fwd_b, eng.bwd) for v in PLN.chain_stats(pl).values()])
        assert l == l_ref and torch.equal(lg, lg_ref), rep
        assert torch.equal(g[mask], g_ref[mask]), rep
    print("tiles that waited per chained launch, by replay under load:", waited)


def test_chain_timeout_is_reported_not_hung():
    """A stage table whose first stage never publishes: every consumer's wait gives up after spin_max polls, the launch
    ends, and the timeout flag names a stage."""
    hps, eng, wts, emb, inp = seeded_full_engine(B=2, w=100, seed=5)      # (too small for the engine to chain anything itself)
    assert _chains(eng) == []
    eng.set_inputs(*[t.to(DEV) for t in inp])
    eng.forward()
    torch.cuda.synchronize()
    fb = eng.fwd_b
    idx = [i for i, lab in enumerate(fb.labels) if lab.startswith(("G1.", "G2."))]
    sub = PLN.Plan("stuck")
    sub.ops, sub.labels = [fb.ops[i] for i in idx[:4]], [fb.labels[i] for i in idx[:4]]
    for op in sub.ops:
        op.join = 0
    made = PLN.insert_nt_chains(sub, eng.ws, "chain.stuck", lambda lab: True, force=True, spin_max=2000)
    assert made == [(1, 4)] and sub.labels[0].startswith("zero:")      # (one zero op for the counters, then the chain)
    stages, cd = sub.nt_chains[sub.labels[1]]
    # break the table on the device: stage 0 does not publish
    raw = eng.ws.get("chain.stuck.0.stages")
    host = (L.NtStage * 4).from_buffer_copy(bytes(raw[:(C.sizeof(L.NtStage) * 4 + 7) // 8].cpu().numpy().tobytes())[:C.sizeof(L.NtStage) * 4])
    host[0].publish = 0
    b = bytes(host)
    raw[:(len(b) + 7) // 8].copy_(torch.frombuffer(bytearray(b + b"\0" * (-len(b) % 8)), dtype=torch.int64))
    sub.run(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert PLN.chain_timeouts(sub) == [sub.labels[1]]
