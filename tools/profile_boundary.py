#!/usr/bin/env python3
"""Host-side profile of the step through the module boundary (bench.py's timed region): where the Python time goes."""
import cProfile
import os
import pstats
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

sys.argv = ["bench.py"] + sys.argv[1:]
args = bench.parse()
device = torch.device("cuda", 0)
from ae_wavenet_amd.jitter import DeviceJitter
from ae_wavenet_amd.loader import DevicePrefetcher
hps, model, opt = bench.make_model(args, device)
eng = model._ensure_engine(args.batch)
loader = DevicePrefetcher(bench.host_batches(model, args.batch, 0), device, depth=2, jitter=DeviceJitter(args.jitter_prob, seed=1))


def step():
    wav, mel, voice, jitter = next(loader)
    opt.zero_grad()
    pred, target, loss = model.run(wav, mel, voice, jitter)
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host time per step {1e3 * (t1 - t0) / 20:.2f} ms, with final sync {1e3 * (t2 - t0) / 20:.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
