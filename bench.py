#!/usr/bin/env python3
"""bench.py — 16 kHz audio samples / second / training step of the VQ-VAE-EMA WaveNet
autoencoder (par/arch.vqvae-ema.json shape) on N MI355X, data-parallel.

A step = forward + backward + gradient all-reduce (N>1) + fused Adam + EMA codebook update
over one batch of 8 windows x 5000 output samples per GPU (BASELINE.json configs[1]; weak
scaling), synthetic inputs resident in HBM before the timed region, random-init weights.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints one JSON line.  `roofline` is for the dominant kernel (the bf16 NT GEMM that
carries forward + dgrad of the gated stack): algorithmic FLOPs / HIP-event time measured on
the plan's stream.  `cpu_baseline` times the oracle (torch fp32 CPU port of the reference) on
one window of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def build_engine(args, device):
    from ae_wavenet_amd import config, model as M
    hps = config.make_hps("vqvae-ema", n_win_batch=args.n_win, n_batch=args.batch)
    eng = M.TrainEngine(hps, B=args.batch, device=device, n_mel=39)
    gen = torch.Generator().manual_seed(2507)                     # hparams.py:92 random_seed
    for k in eng.ps.names():
        shp = eng.ps.shape[k]
        t = torch.empty(shp)
        if len(shp) >= 2:
            torch.nn.init.xavier_uniform_(t, generator=gen)      # netmisc.py:10-14
        else:
            t.zero_()
        eng.ps.view(k).copy_(t)
    emb = torch.empty(hps.bn_vq_n_embed, hps.bn_n_out)
    torch.nn.init.xavier_uniform_(emb, gain=10, generator=gen)    # vqema_bn.py:97
    eng.emb.copy_(emb)
    eng.init_ema_from_emb()
    return hps, eng


def synth_batch(eng, rank, device):
    g = eng.geom
    gen = torch.Generator().manual_seed(1000 + rank)
    B = eng.B
    wav = torch.randint(0, 256, (B, g.enc_in_len), generator=gen).float()
    mel = torch.randn(B, 39, g.mel_len, generator=gen)
    voice = torch.randint(0, 40, (B,), generator=gen)
    jitter = torch.arange(g.embed_len).repeat(B, 1)
    return [t.to(device) for t in (wav, mel, voice, jitter)]


def cpu_baseline(hps, eng, seconds_budget=40.0):
    """Oracle (torch fp32 CPU restatement of the reference) forward+backward on ONE window of
    the same workload (B=1, same n_win), all host cores."""
    from oracle import ref_model as R
    from ae_wavenet_amd import geometry
    g = eng.geom
    sd = {k: eng.ps.view(k).detach().cpu().clone().requires_grad_(True) for k in eng.ps.names()}
    emb = eng.emb.detach().cpu().clone()
    gen = torch.Generator().manual_seed(5)
    wav = torch.randint(0, 256, (1, g.enc_in_len), generator=gen).float()
    mel = torch.randn(1, 39, g.mel_len, generator=gen)
    voice = torch.randint(0, 40, (1,), generator=gen)
    jitter = torch.arange(g.embed_len).repeat(1, 1)
    cores = torch.get_num_threads()
    times = []
    t_start = time.time()
    for it in range(3):
        t0 = time.time()
        out = R.ae_run(sd, {"emb": emb}, hps, g, wav, mel, voice, jitter, loss_mode="intended", take_compat=False)
        out["loss"].backward()
        times.append(time.time() - t0)
        for v in sd.values():
            v.grad = None
        if time.time() - t_start > seconds_budget:
            break
    best = min(times)
    return {"value": g.n_win / best, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"oracle fwd+bwd, 1 window of {g.n_win} samples (B=1), best of {len(times)} "
                      f"({best:.2f} s/step)"}


def pmc_traffic():
    """HBM bytes per launch (a number, as the bench contract asks) of the dominant kernel from the committed PMC passes
    (profiles/r01_pmc_hbm_traffic.csv: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this
    same command, FETCH doubled per MI355X_MICROARCH.md).  bench.py cannot collect PMCs itself."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.csv")
    try:
        import csv
        n = tot = 0.0
        for r in csv.DictReader(open(path)):
            if "k_gemm_nt_bf16" in r["kernel"]:
                k = float(r["launches"])
                n += k
                tot += k * (float(r["fetch_MB_corrected_x2"]) + float(r["write_MB"])) * 1e6
        return round(tot / n) if n else None
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--n-win", dest="n_win", type=int, default=5000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--lanes", type=int, default=1, help="0: serial plan order (no side-lane overlap)")
    ap.add_argument("--nt-wave-rows", dest="nt_wave_rows", type=int, default=64, help="bf16 NT shape (64|128|256)")
    ap.add_argument("--nt-pipe", dest="nt_pipe", type=int, default=1, help="0 plain loop, 1 pipelined, 2 pipelined K=64 tiles")
    ap.add_argument("--tn-blocks", dest="tn_blocks", type=int, default=0, help="split-K block target of the TN ops")
    ap.add_argument("--tn-small", dest="tn_small", default="", help="max_tiles,target_blocks for small-output TN ops")
    ap.add_argument("--chains", type=int, default=1, help="1: gated stack as one full-batch chain; 2: two half-batch chains")
    ap.add_argument("--nt-rows192", dest="nt_rows192", type=int, default=-1, help="192-row NT tiles: 0 never, 1 cost model, 2 always")
    ap.add_argument("--nt-small", dest="nt_small", type=int, default=-1, help="tile-count threshold for 64-row NT tiles")
    ap.add_argument("--side-lanes", dest="side_lanes", type=int, default=0, help="side lanes the wgrads rotate over (1..4)")
    ap.add_argument("--per-op", default=None, help="write per-op HIP-event times (ms) to this file")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # AEW_BENCH_SHARE_GPU=1 + AEW_BENCH_BACKEND=gloo: all ranks on cuda:0 over gloo - a functional check of the
    # multi-process path on a one-GPU box (tools/measure_round.sh), not a measurement
    share = os.environ.get("AEW_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("AEW_BENCH_BACKEND", "nccl")
    if share:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dp = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        from ae_wavenet_amd.dp import DataParallel
        dp = DataParallel()

    from ae_wavenet_amd import _lib as L
    lib = L.load()
    lib.aew_set_lanes(args.lanes)
    lib.aew_set_nt_wave_rows(args.nt_wave_rows)
    lib.aew_set_nt_pipe(args.nt_pipe)
    if args.nt_rows192 >= 0:
        lib.aew_set_nt_rows192(args.nt_rows192)
    if args.nt_small >= 0:
        lib.aew_set_nt_small_tiles(args.nt_small)
    if args.tn_blocks:
        lib.aew_set_tn_target_blocks(args.tn_blocks)
    if args.tn_small:
        a, b = (int(v) for v in args.tn_small.split(","))
        lib.aew_set_tn_small(a, b)
    from ae_wavenet_amd import engine as _E
    _E.DecoderPlan.split_chains = args.chains == 2
    if args.side_lanes:
        _E.DecoderPlan.n_side_lanes = args.side_lanes
    if os.environ.get("AEW_SPLIT_MULTISEG"):
        _E.DecoderPlan.split_multiseg = os.environ["AEW_SPLIT_MULTISEG"] == "1"
    hps, eng = build_engine(args, device)
    if dp is not None:
        dp.broadcast_params(eng)
    batch = synth_batch(eng, rank, device)
    eng.set_inputs(*batch)
    ema_ar = dp.allreduce_ema if dp is not None else None
    # sum-type loss (VQEMA 'intended'): summed gradients across ranks == single-process global batch
    gscale = 1.0

    def step():
        if dp is not None:
            dp.train_step(eng, args.lr, gscale)     # collectives overlapped (EMA stats, decoder / encoder grads)
        else:
            eng.forward()
            eng.backward()
            eng.adam_step(args.lr, gscale)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(eng.loss_buf[0])

    # ---- per-kernel timing for the roofline (outside the timed region) ----------------------
    roof = None
    kern = {}
    if rank == 0:
        import ctypes as C
        lib.aew_timing_enable(1)
        n_t = 3
        for _ in range(n_t):
            eng.forward(None, timing=True)
            eng.backward(timing=True)
            eng.adam_step(args.lr, gscale)
        cap = 1 << 16
        ms = (C.c_float * cap)()
        tags = (C.c_int32 * cap)()
        cnt = C.c_int(0)
        L.check(lib.aew_timing_read(ms, tags, cap, C.byref(cnt)), "timing_read")
        lib.aew_timing_enable(0)
        cls_ms = {}
        cls_n = {}
        for i in range(min(cnt.value, cap)):
            c = tags[i] // 100
            cls_ms[c] = cls_ms.get(c, 0.0) + ms[i] / n_t
            cls_n[c] = cls_n.get(c, 0) + 1
            kern[tags[i]] = kern.get(tags[i], 0.0) + ms[i] / n_t
        if args.per_op:
            labels = eng.fwd_a.labels + eng.fwd_b.labels + eng.bwd.labels + \
                (eng.cb.labels if hasattr(eng, "cb") else []) + eng.opt.labels
            per = {}
            for i in range(min(cnt.value, cap)):
                lab = labels[i % len(labels)]
                per[lab] = per.get(lab, 0.0) + ms[i] / n_t
            with open(args.per_op, "w") as fh:
                for lab, v in sorted(per.items(), key=lambda kv: -kv[1]):
                    fh.write(f"{v:9.4f}  {lab}\n")
        fl = eng.flops_per_step()
        # forward + dgrad of the gated stack / post network run on the bf16 NT kernel,
        # wgrad on the bf16 TN kernel  (90.4 MFLOP per output sample per step)
        nt_flops, nt_ms = fl["step"] * 2.0 / 3.0, cls_ms.get(1, 0.0)
        n_launch = max(cls_n.get(1, 0) // n_t, 1)
        achieved = nt_flops / (nt_ms * 1e-3) / 1e12 if nt_ms > 0 else 0.0
        roof = {"bound": "mfma", "kernel": "k_gemm_nt_bf16", "achieved": round(achieved, 2), "peak": 2500.0,
                "unit": "TFLOP/s", "frac": round(achieved / 2500.0, 4), "traffic": pmc_traffic(),
                "launches_per_step": n_launch, "avg_launch_ms": round(nt_ms / n_launch, 5),
                "alg_flops_per_launch": nt_flops / n_launch,
                "tn_bf16": {"achieved": round((fl["step"] / 3.0) / (cls_ms.get(2, 1e9) * 1e-3) / 1e12, 2),
                            "ms_per_step": round(cls_ms.get(2, 0.0), 4)},
                "ms_per_step_by_class": {str(k): round(v, 4) for k, v in sorted(cls_ms.items())}}
        # the same launches against the HBM roofline: measured bytes per launch (PMC) / measured time per launch.  At
        # K = 256..1024 with 3-4 activation tensors in and out the stack's intensity (~240 FLOP/B) is below the ridge
        # (2500 / 8 = 312), see profiles/r01_op_roofline.txt
        roof["traffic_source"] = "profiles/r01_pmc_hbm_traffic.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH x 2)"
        if roof["traffic"]:
            tbps = roof["traffic"] / (nt_ms / n_launch * 1e-3) / 1e12
            roof["hbm_view"] = {"achieved": round(tbps, 3), "peak": 8.0, "unit": "TB/s", "frac": round(tbps / 8.0, 4)}

    cpu = None
    if world > 1:
        cpu = {"value": None, "unit": "samples/s", "cores": 0, "kind": "port",
               "sample": "measured at --gpus 1 only (rank 0 would hold the other ranks for ~20 s)"}
    elif rank == 0 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(hps, eng)
        except Exception as e:                                   # never lose the GPU line
            cpu = {"value": None, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
                   "sample": f"failed: {type(e).__name__}: {e}"}

    if rank == 0:
        samples = world * args.batch * args.n_win * args.steps
        out = {
            "metric": "16kHz audio samples/sec/step (VQ-VAE-EMA train)",
            "value": samples / dt, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "VQ-VAE-EMA (arch.vqvae-ema shape: 2x10 gated layers, 368 res / 256 dil "
                                   "/ 256 skip, K=4096 d=32) fwd+bwd+Adam",
                       "global_batch": world * args.batch, "n_win_batch": args.n_win,
                       "parallelism": f"dp{world}", "loss": float(loss_val),
                       "decoder": "bf16 MFMA, fp32 accumulate", "encoder_vq": "fp32 MFMA exact chain"},
            "roofline": roof, "cpu_baseline": cpu,
            "kernel_ms_by_tag": {str(k): round(v, 4) for k, v in sorted(kern.items())},
        }
        print(json.dumps(out))
    if world > 1:
        sys.stdout.flush()
        dist.barrier()                      # rank 0 arrives last (per-op timing pass): everybody leaves together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
