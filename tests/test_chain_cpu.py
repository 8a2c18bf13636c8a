"""Chained NT launches (AEW_OP_NT_CHAIN): the HOST side - aew_nt_chain_build's stage table - checked without a GPU.

The table is what makes the one-launch form of the gated stack (wavenet.py:100-109, 354-357) and of its backward safe:
a tile of a stage may start as soon as the producer row tiles its dependency records name have published.  The test
re-derives, from the plain segment / view records of the stage descriptors and nothing else, which rows every tile reads
and writes, then replays the launch under adversarial random schedules that honour ONLY the table's waits and the
in-order dispatch of workgroups: every row a tile reads must have been written by a finished tile."""
import ctypes as C
import random

import numpy as np
import pytest
import torch

from ae_wavenet_amd import _lib as L, config, model as M

BM = 256


def _engine(monkeypatch, n_chain, **kw):
    monkeypatch.setattr(M.TrainEngine, "nt_chain", n_chain)
    monkeypatch.setattr(M.TrainEngine, "nt_chain_bwd", n_chain)
    monkeypatch.delenv("AEW_NT_CHAIN", raising=False)
    hps = config.make_hps(kw.pop("arch", "vqvae-ema"), n_win_batch=kw.pop("w", 5000), **kw)
    return M.TrainEngine(hps, 8, "cpu", n_mel=39)


def _refs(g):
    """(inputs, out0, side outputs) of a descriptor as (ptr, batch_stride, pitch, step, off, lo, hi) records"""
    rec = lambda v: (v.ptr or 0, v.batch_stride, v.row_pitch, v.row_step, v.row_off, v.row_lo, v.row_hi)
    ins = [rec(g.seg[i]) for i in range(g.n_segs)]
    if g.epi == L.EPI_DFG or (g.epi == L.EPI_STORE and g.flags & L.EF_ADD_AUX0):
        ins.append(rec(g.aux0))
    if g.epi == L.EPI_DFG or (g.epi == L.EPI_STORE and g.flags & (L.EF_MUL_POS1 | L.EF_OUT1_POS1)):
        ins.append(rec(g.aux1))
    return [r for r in ins if r[0]], rec(g.out0)


def _rows(ref, m0, m1):
    """buffer rows [a, b] that GEMM rows [m0, m1] touch through `ref` (None: none)"""
    ptr, bs, pitch, step, off, lo, hi = ref
    a, b = m0 * step + off, m1 * step + off
    a, b = max(min(a, b), lo), min(max(a, b), hi - 1)
    return (a, b) if a <= b else None


def _simulate(stages, seed, slots=64):
    """Random schedule: workgroups are DISPATCHED in index order into `slots` resident slots; a resident tile whose waits
    (the table's records) are satisfied may RUN at any later time, in any order.  Returns the number of tiles checked."""
    rng = random.Random(seed)
    n = len(stages)
    meta = []
    for s in stages:
        ins, out0 = _refs(s.g)
        meta.append((ins, out0))
    done = [np.zeros(s.n_mt, dtype=np.int32) for s in stages]          # per row tile: finished N tiles (batch element 0)
    written = {}                                                       # out0 ptr -> bool array over buffer rows
    for s, (ins, out0) in zip(stages, meta):
        written[out0[0]] = np.zeros(out0[6] + 1, dtype=bool)
    order = [(si, mt, nt) for si, s in enumerate(stages) for mt in range(s.n_mt) for nt in range(s.n_nt)]
    nxt, resident, checked = 0, [], 0
    lib = L.load()
    while nxt < len(order) or resident:
        while nxt < len(order) and len(resident) < slots:
            resident.append(order[nxt])
            nxt += 1
        ready = []
        for t in resident:
            si, mt, nt = t
            s = stages[si]
            ok = True
            for d in range(s.n_deps):
                lo, hi = C.c_int(0), C.c_int(0)
                L.check(lib.aew_nt_chain_dep_tiles(C.byref(s), d, mt * BM, C.byref(lo), C.byref(hi)), "dep_tiles")
                dep = s.dep[d]
                prod = next(p for p in range(n) if stages[p].cnt_base == dep.cnt_base)
                for u in range(lo.value, hi.value + 1):
                    ok = ok and done[prod][u] >= dep.need
            if ok:
                ready.append(t)
        assert ready, "deadlock: no resident tile can run (a wait names a tile that is dispatched later)"
        t = rng.choice(ready) if rng.random() < 0.7 else ready[-1]        # often the YOUNGEST runnable tile first
        resident.remove(t)
        si, mt, nt = t
        s = stages[si]
        ins, out0 = meta[si]
        m0, m1 = mt * BM, min(mt * BM + BM, s.g.M) - 1
        for ref in ins:
            if ref[0] in written:
                rr = _rows(ref, m0, m1)
                if rr is not None:
                    w = written[ref[0]]
                    assert w[rr[0]:rr[1] + 1].all(), \
                        f"stage {si} tile {mt} reads rows {rr} of a buffer the chain writes before they are complete"
        checked += 1
        done[si][mt] += 1
        if done[si][mt] == s.n_nt:
            rr = _rows(out0, m0, m1)
            if rr is not None:
                written[out0[0]][rr[0]:rr[1] + 1] = True
    return checked


@pytest.mark.parametrize("n_chain", [64, 2])
def test_stage_table_orders_every_read_after_its_write(monkeypatch, n_chain):
    eng = _engine(monkeypatch, n_chain)
    chains = [(pl.name, lab, st) for pl in (eng.fwd_b, eng.bwd) for lab, (st, _) in getattr(pl, "nt_chains", {}).items()]
    NL = len(eng.geom.layers)
    assert len(chains) == (3 if n_chain == 64 else (NL - 1) + 1 + (NL + 1))
    for name, lab, stages in chains[:3] + chains[-2:]:
        # every stage but the last is waited for; stages are laid out at multiples of 8 blocks in stage order
        assert all(s.publish for s in stages[:-1]) and not stages[-1].publish
        assert all(s.first_block % 8 == 0 for s in stages)
        assert all(b.first_block == a.first_block + a.n_blocks for a, b in zip(stages, stages[1:]))
        assert all(s.n_deps >= 1 for s in stages[1:]) and stages[0].n_deps == 0
        for seed in range(2):
            assert _simulate(stages, seed) == sum(s.n_mt * s.n_nt for s in stages)


def test_dilated_taps_widen_the_wait(monkeypatch):
    """wavenet.py:100: layer l reads x[t] and x[t + d] - its tile waits for the producer rows up to m_last + d; the
    backward's dx reads dfg[t - d]."""
    eng = _engine(monkeypatch, 64)
    stages, _ = eng.fwd_b.nt_chains["chain[G1.0..G1.19]"]
    dil = [lg.dil for lg in eng.geom.layers]
    for l in range(1, len(dil)):
        s = stages[2 * l]                                  # G1 of layer l
        assert s.g.epi == L.EPI_GATED and s.n_deps == 1
        assert (s.dep[0].d_lo, s.dep[0].d_hi) == (0, dil[l])
        g2 = stages[2 * l - 1]
        assert g2.g.epi == L.EPI_STORE and (g2.dep[0].d_lo, g2.dep[0].d_hi) == (0, 0)
    (bst, _), = [v for v in eng.bwd.nt_chains.values()]
    assert [s.g.epi for s in bst[:2]] == [L.EPI_STORE, L.EPI_STORE]            # d.post2, d.post1 head the run
    for i, l in enumerate(range(len(dil) - 1, -1, -1)):
        dz, dx = bst[2 + 2 * i], bst[2 + 2 * i + 1]
        assert dz.g.epi == L.EPI_DFG and dx.g.epi == L.EPI_STORE
        assert (dx.dep[0].d_lo, dx.dep[0].d_hi) == (-dil[l], 0)


def test_builder_refuses_what_it_cannot_order():
    lib = L.load()
    from ae_wavenet_amd.plan import Workspace, Mat, make_nt
    ws = Workspace("cpu")
    X = Mat.new(ws, "x", 2, 4096, 128, L.BF16)
    Y = Mat.new(ws, "y", 2, 4096, 128, L.BF16)
    W = Mat.new(ws, "w", 1, 128, 128, L.BF16)
    a = make_nt(L.BF16, 4096, 128, 128, 2, [X.seg(128)], W.ptr, out0=Y.view())
    b = make_nt(L.BF16, 4096, 128, 128, 2, [Y.seg(128)], W.ptr, out0=X.view())      # writes what stage 0 reads
    c = make_nt(L.BF16, 2048, 128, 128, 2, [Y.seg(128, row_step=2)], W.ptr, out0=Mat.new(ws, "z", 2, 2048, 128, L.BF16).view())

    def build(descs, force=1):
        n = len(descs)
        arr = (L.GemmNT * n)(*descs)
        st = (L.NtStage * n)()
        bm = (C.c_uint16 * 4096)()
        nb, nc, se = C.c_int(0), C.c_int(0), C.c_int(0)
        return lib.aew_nt_chain_build(C.byref(arr), n, C.byref(st), C.byref(bm), 8 * 4096, C.byref(nb), C.byref(nc), C.byref(se), force), st

    assert build([a, b])[0] == L.E_UNSUP                   # write-after-read inside the run
    assert build([a, c])[0] == L.E_UNSUP                   # strided read of a produced buffer
    rc, st = build([a, make_nt(L.BF16, 4000, 128, 128, 2, [Y.seg(128, row_off=96)], W.ptr,
                               out0=Mat.new(ws, "q", 2, 4096, 128, L.BF16).view())])
    assert rc == 0 and (st[1].dep[0].d_lo, st[1].dep[0].d_hi, st[1].dep[0].c_hi) == (96, 96, 4095)
    assert build([a], force=1)[0] == 0 and build([a], force=0)[0] == L.E_UNSUP        # 32 tiles: the launcher's 64-row shapes
