#!/usr/bin/env python3
"""Chained NT launches: run the BASELINE-size engine with and without chaining and print, replay by replay, whether loss /
logits / gradients are bit-identical and what the launches' wait statistics say (timeout flag, tiles that waited, longest
wait in polls).  python tools/chain_debug.py [n_chain] [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ae_wavenet_amd import model as M, plan as PLN
from tests.test_gpu_parity import DEV, seeded_full_engine
from tests.test_chain_gpu import _mask, _step

n_chain = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
B, w = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (8, 5000)
M.TrainEngine.nt_chain = 0
M.TrainEngine.nt_chain_force = True
hps, eng0, wts, emb, inp = seeded_full_engine(B=B, w=w, seed=11)
inp = [t.to(DEV) for t in inp]
eng0.set_inputs(*inp)
l_ref, lg_ref, g_ref = _step(eng0)
mask = _mask(eng0)
del eng0
torch.cuda.empty_cache()
M.TrainEngine.nt_chain = n_chain
_, eng, _, _, _ = seeded_full_engine(B=B, w=w, seed=11)
eng.set_inputs(*inp)
for mode in ("graph", "eager"):
    eng.use_graphs = mode == "graph"
    for rep in range(reps):
        l, lg, g = _step(eng)
        st = {pl.name: list(PLN.chain_stats(pl).values()) for pl in (eng.fwd_b, eng.bwd)}
        bad_lg = int((lg != lg_ref).sum())
        bad_g = int((g[mask] != g_ref[mask]).sum())
        print(f"{mode} rep {rep}: loss {'==' if l == l_ref else '!='} ({l} vs {l_ref}) logits differing {bad_lg} grads differing {bad_g}  "
              f"stats (flag, waited, max polls) {st}", flush=True)
