"""Diagnostic: which gradient tensors differ between the serial and the two-lane schedule."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_gpu_parity import seeded_full_engine, DEV
from ae_wavenet_amd import _lib as L

hps, eng, wts, emb, inp = seeded_full_engine(B=8, w=5000, seed=11)
eng.set_inputs(*[t.to(DEV) for t in inp])
lib = L.load()


def step():
    eng.init_ema_from_emb()
    loss = float(eng.forward())
    eng.backward()
    torch.cuda.synchronize()
    return loss, eng.ps.grads[:eng.ps.numel].clone()


def recapture():
    for pl in (eng.fwd_a, eng.fwd_b, eng.bwd):
        pl.invalidate_graph()


def report(tag, g, ref):
    bad = []
    for n in eng.ps.names():
        a, b = eng.ps.view(n, True), None
    off = 0
    for n in eng.ps.names():
        k = eng.ps.numel_of(n)
        o = eng.ps.view(n, True).data_ptr() - eng.ps.grads.data_ptr()
        o //= 4
        d = (g[o:o + k] - ref[o:o + k]).abs().max().item()
        if d != 0:
            bad.append((n, d, ref[o:o + k].abs().max().item()))
    print(tag, "differing tensors:", len(bad))
    for n, d, m in bad[:40]:
        print(f"   {n:50s} maxdiff {d:.3e} (scale {m:.3e})")


lib.aew_set_lanes(0); recapture()
l0, g0 = step()
l1, g1 = step()
report("serial vs serial", g1, g0)
lib.aew_set_lanes(1); recapture()
for i in range(3):
    l2, g2 = step()
    report(f"lanes run {i} vs serial", g2, g0)
