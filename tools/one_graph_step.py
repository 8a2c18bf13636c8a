#!/usr/bin/env python3
"""Does it matter that a step is four hipGraph launches (fwd_a, fwd_b, bwd, codebook) + Adam?  The same ops as ONE
captured graph (plan boundaries become joins) against the engine's normal path, on the bench workload.
    python tools/one_graph_step.py       # on the GPU box
"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ae_wavenet_amd import _lib as L, autoencoder_model as ae, config
from ae_wavenet_amd.plan import Plan

dev = "cuda:0"
hps = config.make_hps("vqvae-ema", n_win_batch=5000, n_batch=8)
torch.manual_seed(0)
model = ae.AutoEncoder(hps, n_mel=39).to(dev)              # Xavier weights, codebook gain 10 (the engine alone starts from zeros)
eng = model._ensure_engine(8)
g = eng.geom
gen = torch.Generator().manual_seed(1)
eng.set_inputs(torch.randint(0, 256, (8, g.enc_in_len), generator=gen).float().to(dev), torch.randn(8, 39, g.mel_len, generator=gen).to(dev),
               torch.randint(0, 40, (8,), generator=gen).to(dev), torch.arange(g.embed_len).repeat(8, 1).to(dev))


def timed(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def normal():
    eng.forward(); eng.backward(); eng.adam_step(1e-4, 1.0)


one = Plan("step")
scratch = torch.zeros(64, device=dev)
for k, sub in enumerate((eng.fwd_a, eng.fwd_b, eng.bwd, eng.cb)):
    if k:                                                # the previous plan's implicit end-of-plan join
        z = L.Zero()
        z.ptr, z.bytes = scratch.data_ptr(), 16
        one.add(L.OP_ZERO, z, "join", join=True)
    for op, lab in zip(sub.ops, sub.labels):
        one.ops.append(L.Op.from_buffer_copy(op))
        one.labels.append(lab)
one._arr = None


def fused():
    one.run_graph(eng._stream())
    eng.adam_step(1e-4, 1.0)


a = timed(normal)
l0 = float(eng.loss_buf[0])
b = timed(fused)
l1 = float(eng.loss_buf[0])
a2 = timed(normal)
b2 = timed(fused)
print(f"four graphs + adam: {a:.3f} / {a2:.3f} ms    one graph + adam: {b:.3f} / {b2:.3f} ms   (loss {l0:.1f} / {l1:.1f})")
