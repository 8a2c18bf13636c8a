#!/usr/bin/env python3
"""Soak test of the chained launches: two engines of the BASELINE configuration - one with every GEMM its own launch, one
with the forward (and optionally the backward) chained - are driven in lockstep over many steps with a fresh batch each,
and after every step the loss, the logits' checksum and the gradient buffer must be bit-identical (bias-type gradients,
fp32-atomic sums, at round-off); no hand-off wait may give up.  A rare race in the in-launch hand-off would show here.
    python tools/chain_soak.py [steps] [bwd: 0|1] [noise: 0|1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ae_wavenet_amd import model as M, plan as PLN
from tests.test_gpu_parity import DEV, seeded_full_engine
from tests.test_chain_gpu import _mask

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
bwd = int(sys.argv[2]) if len(sys.argv) > 2 else 1
noise = int(sys.argv[3]) if len(sys.argv) > 3 else 1
M.TrainEngine.nt_chain, M.TrainEngine.nt_chain_bwd = 0, 0
hps, ref, wts, emb, inp = seeded_full_engine(B=8, w=5000, seed=3)
M.TrainEngine.nt_chain, M.TrainEngine.nt_chain_bwd = 64, (64 if bwd else 0)
_, eng, _, _, _ = seeded_full_engine(B=8, w=5000, seed=3)
mask = _mask(eng)
g = eng.geom
gen = torch.Generator(device="cpu").manual_seed(17)
side = torch.cuda.Stream()
a = torch.randn(2048, 2048, device=DEV)
bad = 0
waited_max = 0
for it in range(steps):
    wav = torch.randint(0, 256, (8, g.enc_in_len), generator=gen).float().to(DEV)
    mel = torch.randn(8, 39, g.mel_len, generator=gen).to(DEV)
    voice = torch.randint(0, 40, (8,), generator=gen).to(DEV)
    jitter = torch.arange(g.embed_len).repeat(8, 1).to(DEV)
    out = []
    for e in (ref, eng):
        e.set_inputs(wav, mel, voice, jitter)
        e.init_ema_from_emb()
        if noise and e is eng and it % 3 == 0:              # uneven load from a second stream, every third step
            with torch.cuda.stream(side):
                for _ in range(1 + it % 7):
                    a = (a @ a).clamp_(-1, 1)
        loss = e.forward()
        e.backward()
        out.append((loss.clone(), e.logits().double().sum(), e.ps.grads[:e.ps.numel].clone()))
    torch.cuda.synchronize()
    (l0, s0, g0), (l1, s1, g1) = out
    st = [v for pl in (eng.fwd_b, eng.bwd) for v in PLN.chain_stats(pl).values()]
    waited_max = max(waited_max, max(v[1] for v in st))
    ok = bool(l0 == l1) and bool(s0 == s1) and torch.equal(g0[mask], g1[mask]) and all(v[0] == 0 for v in st)
    if not ok:
        bad += 1
        print(f"step {it}: MISMATCH loss {float(l0)} {float(l1)} logits {float(s0)} {float(s1)} grads differing "
              f"{int((g0[mask] != g1[mask]).sum())} stats {st}", flush=True)
    if it % 100 == 0:
        print(f"step {it}: ok so far ({bad} mismatches), chain stats {st}", flush=True)
print(f"soak: {steps} steps, {bad} mismatches, most tiles waiting in one launch {waited_max}")
sys.exit(1 if bad else 0)
