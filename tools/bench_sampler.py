"""Throughput / latency of the autoregressive sampler at arch.vqvae-ema width (synthetic weights and conditioning).

  python tools/bench_sampler.py [--steps 4000] [--batches 1,2,4,8,16] [--flag-stride 64]

Prints, per number of 16-stream batches in flight: us per time step (all batches), samples/s per stream and in total,
and the implied hand-off time (step time / 44 hand-offs on the critical path with one batch).  Free-running
(drawn) generation after one primed position, so the feedback loop through SAMPLE is in the timed path."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ae_wavenet_amd import config, sampler as S                                   # noqa: E402
from ae_wavenet_amd.engine import ParamStore, decoder_param_specs                 # noqa: E402
from ae_wavenet_amd.plan import Workspace                                         # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--batches", default="1,2,4,8,16,32")
    ap.add_argument("--flag-stride", type=int, default=64)
    ap.add_argument("--nap", type=int, default=-1, help="nap_eighths override (0..7)")
    ap.add_argument("--profile", action="store_true", help="per-role phase clock for the 1-batch run")
    ap.add_argument("--deep", action="store_true", help="30 layers x 512 residual channels")
    args = ap.parse_args()
    dev = "cuda:0"
    hps = config.make_hps("vqvae-ema", **(dict(n_blocks=3, n_res=512) if args.deep else {}))
    ws = Workspace(dev)
    ps = ParamStore(ws, decoder_param_specs(hps, hps.bn_n_out, "decoder."))
    gen = torch.Generator().manual_seed(0)
    for k in ps.names():
        t = torch.empty(ps.shape[k])
        if len(ps.shape[k]) >= 2:
            torch.nn.init.xavier_uniform_(t, generator=gen)
        else:
            t.uniform_(-0.1, 0.1, generator=gen)
        ps.view(k).copy_(t)
    smp = S.Sampler(hps, ps, "decoder.", dev, flag_stride=args.flag_stride)
    if args.nap >= 0:
        smp.nap_eighths = args.nap
    g = smp.g
    T = args.steps
    hops = 2 * g.NL + 4
    print(f"{g.NL} layers, R={g.R}, {g.n_actors()} actors, {T} steps, flag stride {args.flag_stride}")
    for nb in [int(v) for v in args.batches.split(",")]:
        n = 16 * nb
        cond = (torch.randn(n, T, 128, generator=gen) * 0.5).to(torch.bfloat16).to(dev)
        bias = (torch.randn(n, g.NL, g.n_pairs * 32, generator=gen) * 0.1).to(dev)
        forced = torch.full((n, T), -1, dtype=torch.int32)
        forced[:, 0] = 128
        forced = forced.to(dev)
        smp.generate(cond, bias, forced[:, :64].contiguous(), seed=1)          # warm-up (code load)
        best = 1e30
        for _ in range(3):
            wav, _ = smp.generate(cond, bias, forced, seed=2, timing=True)
            best = min(best, smp.last["kernel_ms"])
        if args.profile and nb == 1:
            smp.generate(cond, bias, forced, seed=2, profile=True)
            for k, v in smp.last["profile"].items():
                print(f"    {k:9s} us per item: wait {v[0]:7.2f}  load+compute {v[1]:6.2f}  publish {v[2]:6.2f}")
        us = best * 1e3 / T
        print(f"batches {nb:3d} ({n:4d} streams): {us:8.2f} us/step  {1e6 / us:9.0f} samples/s/stream  "
              f"{n * 1e6 / us / 1e6:8.3f} M samples/s total   ({us / nb:6.2f} us per batch-step, "
              f"{us / hops:5.2f} us per hand-off if serial)   distinct values {len(np.unique(wav.cpu().numpy()))}")


if __name__ == "__main__":
    main()
