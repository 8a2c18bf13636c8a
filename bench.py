#!/usr/bin/env python3
"""bench.py — 16 kHz audio samples / second / training step of the VQ-VAE-EMA WaveNet
autoencoder (par/arch.vqvae-ema.json shape) on N MI355X, data-parallel.

A step (chassis.py:151-171) = `model.run(wav, mel, voice, jitter)` -> `loss.backward()` ->
`FusedAdam.step()` (gradient / EMA-statistic collectives inside for N > 1) on the drop-in module surface
(autoencoder_model.AutoEncoder), over one batch of 8 windows x 5000 output samples per GPU (BASELINE.json configs[1];
weak scaling), synthetic data, random-init weights.  The headline `value` is that step with the batches (a pool of 8
distinct ones, jitter indices included) already resident in HBM, as the contract asks.  `host_fed` is the same step fed
from pinned host memory through the device prefetcher (H2D on a copy stream, jitter generated on the device): the
PCIe-inclusive rate, reported beside `value`.  `engine_only` repeats the measurement below the module surface
(TrainEngine forward / backward / adam on one resident batch: round 1's number).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N ...          # spawns N ranks itself (torch.distributed.run) when WORLD_SIZE is unset
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints one JSON line.  `roofline` is for the dominant kernel (the bf16 NT GEMM that carries forward + dgrad of
the gated stack): algorithmic FLOPs / HIP-event time measured on the plan's stream.  `cpu_baseline` times the oracle
(torch fp32 CPU port of the reference) on a bounded sample of the same workload, Adam included, with and without the
reference's diagnostic second backward.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--arch", default="vqvae-ema", help="config.make_hps architecture; anything but the default (vae, deep: BASELINE configs[3], [4] per GPU at --batch 4 --n-win 65536) is a side measurement, not the bench line")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--n-win", dest="n_win", type=int, default=5000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-box", action="store_true",
                    help="skip the box fingerprint (profiling passes: its ~450 probe launches would sit in the kernel tables)")
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--jitter-prob", dest="jitter_prob", type=float, default=0.12, help="par/train.basic.json")
    ap.add_argument("--lanes", type=int, default=0, help="1: every side-lane op on its own stream / graph branch (default 0: plan order, except the tail branch below)")
    ap.add_argument("--tail-lane", dest="tail_lane", type=int, default=0, help="4: the last grouped wgrad launch as a graph branch beside the rest of the backward (A/B aid); 0: serial")
    ap.add_argument("--nt-wave-rows", dest="nt_wave_rows", type=int, default=64, help="bf16 NT shape (64|128|256)")
    ap.add_argument("--nt-pipe", dest="nt_pipe", type=int, default=1, help="0 plain loop, 1 pipelined, 2 pipelined K=64 tiles")
    ap.add_argument("--tn-blocks", dest="tn_blocks", type=int, default=0, help="split-K block target of the TN ops")
    ap.add_argument("--tn-small", dest="tn_small", default="", help="max_tiles,target_blocks for small-output TN ops")
    ap.add_argument("--chains", type=int, default=1, help="1: gated stack as one full-batch chain; 2: two half-batch chains (graph branches)")
    ap.add_argument("--wgrad-cursor", dest="wgrad_cursor", default="", help="epoch,slack: pace the grouped weight-gradient launch with the row cursor (default: only launches of > 1024 tiles, 4,2); 'off' = never")
    ap.add_argument("--chains-bwd", dest="chains_bwd", type=int, default=1, help="the same for the backward's dz / dx chain")
    ap.add_argument("--nt-rows192", dest="nt_rows192", type=int, default=-1, help="192-row NT tiles: 0 never, 1 cost model, 2 always")
    ap.add_argument("--nt-small", dest="nt_small", type=int, default=-1, help="tile-count threshold for 64-row NT tiles")
    ap.add_argument("--side-lanes", dest="side_lanes", type=int, default=0, help="side lanes the wgrads rotate over (1..4)")
    ap.add_argument("--nt-deep", dest="nt_deep", type=int, default=-1, help="64-row NT launches of <= this many blocks use the 5-stage ring")
    ap.add_argument("--nt-small-waves", dest="nt_small_waves", type=int, default=-1, help="waves per block of the deep 64-row NT shape (2|8)")
    ap.add_argument("--nf-loaders", dest="nf_loaders", type=int, default=-1, help="fp32 NT: dedicated loader waves (0|1)")
    ap.add_argument("--nf-deep", dest="nf_deep", type=int, default=-1, help="fp32 NT: launches of <= this many blocks use the deep ring (0 never)")
    ap.add_argument("--nt-window", dest="nt_window", type=int, default=-1, help="largest dilation on the one-window NT kernel (0 off)")
    ap.add_argument("--check-replicas", dest="check_replicas", action="store_true",
                    help="N > 1: report the largest parameter difference between the ranks after the run")
    ap.add_argument("--nt-chain", dest="nt_chain", default="", help="f or f,b: stages per chained NT launch of the decoder forward / backward (default: the engine's 64,0; 0 = one launch per GEMM)")
    ap.add_argument("--merge-packs", dest="merge_packs", type=int, default=-1, help="1: all weight-layout packs of a step as one launch")
    ap.add_argument("--per-op", default=None, help="write per-op HIP-event times (ms) to this file")
    ap.add_argument("--engine-only", action="store_true", help="headline = the engine-level step (no module surface / loader)")
    return ap.parse_args()


def spawn(args):
    """`python bench.py --gpus N` without a launcher: one process per GPU through torch.distributed.run on
    127.0.0.1 (train.py:58-60 semantics: one process per device).  Fails loudly if the node has fewer devices."""
    import torch
    share = os.environ.get("AEW_BENCH_SHARE_GPU") == "1"
    have = torch.cuda.device_count()
    if not share and have < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: this node exposes {have} GPU(s); refusing to report a {args.gpus}-GPU number "
                         "(AEW_BENCH_SHARE_GPU=1 AEW_BENCH_BACKEND=gloo runs all ranks on cuda:0 as a functional check)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    raise SystemExit(subprocess.call(cmd, env=env))


def make_model(args, device):
    import torch
    from ae_wavenet_amd import autoencoder_model as ae, config, optim
    hps = config.make_hps(args.arch, n_win_batch=args.n_win, n_batch=args.batch, jitter_prob=args.jitter_prob)
    torch.manual_seed(2507)                                       # hparams.py:92 random_seed
    model = ae.AutoEncoder(hps, n_mel=39).to(device)              # Xavier weights, zero biases, codebook gain 10
    opt = optim.FusedAdam(model, lr=args.lr)
    return hps, model, opt


def host_batches(model, B, rank, n_pool=8):
    """A pool of distinct host batches (pinned by the prefetcher): what the reference's DataLoader workers hand to the
    training loop (data.py:218-240: wav, mel, voice, jitter; the jitter indices are generated on the device here)."""
    import torch
    g = model.geom
    gen = torch.Generator().manual_seed(1000 + rank)
    pool = []
    for _ in range(n_pool):
        wav = torch.randint(0, 256, (B, g.enc_in_len), generator=gen).float()
        mel = torch.randn(B, 39, g.mel_len, generator=gen)
        voice = torch.randint(0, 40, (B,), generator=gen)
        pool.append((wav, mel, voice, None))

    def it():
        i = 0
        while True:
            yield pool[i % n_pool]
            i += 1
    return it()


CPU_BASELINE_METHOD = ("r06: thread sweep on one window (with the reference's diagnostic backward), the same step without it at the "
                       "fastest count, then ONE full-batch step at that count on the weights and batch the GPU has just run "
                       "(the parity record); value = fastest one-window rate, value_full_batch beside it")


def cpu_baseline(hps, eng, seconds_budget=75.0, full_batch=True):
    """The oracle (torch fp32 CPU restatement of the reference) as the best the CPU path does on this host, on a bounded
    sample of the same workload - and, in the same pass, the PARITY RECORD of the as-benchmarked configuration.
    One training step = forward, the reference's diagnostic autograd.grad over (mel, encoding)
    (autoencoder_model.py:252-257: what its run() does), backward, Adam over all parameters (chassis.py:151-171).
    (1) one window of the batch, one timed step at each of {4, 8, 16, 32, 64, all} threads (after an untimed warm-up
        step; the sweep stops once two counts in a row are slower than the best);
    (2) the same one-window step WITHOUT the diagnostic backward at the fastest count (the pair);
    (3) one step on the FULL batch at that count, on a fresh copy of the engine's current weights and codebook and on a
        batch the GPU engine runs first (forward + backward, chained launch and all, B = 8, w = 5000): the oracle's loss,
        code indices and gradients against the GPU's = `parity` (chassis.py:151-157).
    `value` is the fastest one-window samples/s, `value_full_batch` the full-batch rate (the GPU number's own batch);
    both are soft numbers - torch's CPU convolutions at these sizes move by 10x with the thread count alone (128 threads:
    0.6 k samples/s, 16 threads: 8 k) - `sweep` lists everything measured."""
    import torch
    from oracle import ref_model as R
    g = eng.geom
    n_all = torch.get_num_threads()

    def fresh():
        sd_ = {k: eng.ps.view(k).detach().cpu().clone().requires_grad_(True) for k in eng.ps.names()}
        return sd_, torch.optim.Adam(list(sd_.values()), lr=1e-4)
    sd, opt = fresh()
    emb = eng.emb.detach().cpu().clone() if hasattr(eng, "emb") else None

    def batch(nb):
        gen = torch.Generator().manual_seed(5)
        return (torch.randint(0, 256, (nb, g.enc_in_len), generator=gen).float(), torch.randn(nb, 39, g.mel_len, generator=gen),
                torch.randint(0, 40, (nb,), generator=gen), torch.arange(g.embed_len).repeat(nb, 1))

    def step(data, diag, sd_=None, opt_=None, keep=None):
        sd_, opt_ = (sd, opt) if sd_ is None else (sd_, opt_)
        wav, mel, voice, jitter = data
        t0 = time.time()
        opt_.zero_grad()
        m = mel.clone().requires_grad_(True)
        out = R.ae_run(sd_, {"emb": emb}, hps, g, wav, m, voice, jitter, loss_mode="intended", take_compat=False)
        if diag:
            torch.autograd.grad(out["loss"], (m, out["encoding_bn"]), retain_graph=True, allow_unused=True)
        out["loss"].backward()
        if keep is not None:                                   # (before Adam moves the weights: the gradients of THIS step)
            keep["loss"] = float(out["loss"].detach())
            keep["ind"] = out["min_ind"].reshape(-1).clone() if "min_ind" in out else None
            keep["grads"] = {k: v.grad.detach().clone() for k, v in sd_.items() if v.grad is not None}
        opt_.step()
        return time.time() - t0
    t_start = time.time()
    one = batch(1)
    step(one, False)                                       # warm-up (allocator, thread pool, first touch of 95 MB of moments)
    sweep = []
    parity = None
    without_diag = None
    cands = sorted({t for t in (4, 8, 16, 32, 64, n_all) if 1 <= t <= n_all})
    try:
        for t in cands:
            torch.set_num_threads(t)
            dt = step(one, True)
            sweep.append({"threads": t, "windows": 1, "seconds": round(dt, 3), "samples_per_s": round(g.n_win / dt, 1)})
            if time.time() - t_start > seconds_budget * 0.5 and len(sweep) >= 2:
                break
            # past the optimum more threads only cost time (128 threads: 8 s per step against 0.6 s at 16): stop once two
            # counts in a row are slower than the best, so that the whole baseline stays a ~25 s sample
            if len(sweep) >= 3 and all(r["samples_per_s"] < max(x["samples_per_s"] for x in sweep) for r in sweep[-2:]):
                break
        best = max(sweep, key=lambda r: r["samples_per_s"])
        torch.set_num_threads(best["threads"])
        dt = step(one, False)
        without_diag = {"threads": best["threads"], "windows": 1, "seconds": round(dt, 3), "samples_per_s": round(g.n_win / dt, 1)}
        if full_batch and time.time() - t_start < seconds_budget * 0.6:
            nb = eng.B
            data = batch(nb)
            # ---- the GPU's step on this batch, from the engine's current state (no optimizer step: the oracle below starts
            # from the same weights).  vq.ema of this extra forward advances the EMA accumulators once more; the bench is over.
            dev = eng.device
            sd2, opt2 = fresh()                                # the weights and the codebook this forward reads (the backward
            if hasattr(eng, "emb"):                            # below refreshes the codebook from the EMA statistics)
                emb = eng.emb.detach().cpu().clone()
            eng.set_inputs(*[t_.to(dev) for t_ in data])
            gl = float(eng.forward())
            eng.backward()
            torch.cuda.synchronize()
            g_ind = eng.ind[:eng.Q].cpu() if hasattr(eng, "ind") else None
            g_grads = {k: eng.ps.view(k, True).detach().cpu().clone() for k in eng.ps.names()}
            keep = {}
            dt = step(data, True, sd2, opt2, keep)
            sweep.append({"threads": best["threads"], "windows": nb, "seconds": round(dt, 3),
                          "samples_per_s": round(nb * g.n_win / dt, 1)})
            rl2, cos = [], []
            for k, gr in keep["grads"].items():
                a, b = g_grads[k].double().reshape(-1), gr.double().reshape(-1)
                nb_ = float(b.norm())
                if nb_ > 0:
                    rl2.append(float((a - b).norm()) / nb_)
                    cos.append(float(torch.dot(a, b)) / (float(a.norm()) * nb_ + 1e-300))
            rl2.sort()
            parity = {"config": f"B = {nb}, w = {g.n_win}, the engine as benchmarked (chained forward: "
                                f"{bool(getattr(eng.fwd_b, 'nt_chains', None))}) vs oracle/ref_model.ae_run on the same weights, codebook and batch",
                      "loss_gpu": gl, "loss_oracle": keep["loss"], "loss_rel": abs(gl / keep["loss"] - 1.0),
                      "n_queries": int(g_ind.numel()) if g_ind is not None else 0,
                      "indices_equal": bool(torch.equal(g_ind, keep["ind"])) if g_ind is not None and keep["ind"] is not None else None,
                      "n_indices_differ": int((g_ind != keep["ind"]).sum()) if g_ind is not None and keep["ind"] is not None else None,
                      "grad_tensors": len(rl2), "grad_rel_l2_median": round(rl2[len(rl2) // 2], 5) if rl2 else None,
                      "grad_rel_l2_worst": round(rl2[-1], 5) if rl2 else None, "grad_cosine_min": round(min(cos), 5) if cos else None}
    finally:
        torch.set_num_threads(n_all)
    best = max((r for r in sweep if r["windows"] == 1), key=lambda r: r["samples_per_s"])
    fullb = next((r for r in sweep if r["windows"] > 1), None)
    return {"value": best["samples_per_s"], "unit": "samples/s", "cores": best["threads"], "kind": "port",
            "value_full_batch": fullb["samples_per_s"] if fullb else None,
            "with_diagnostic_backward": best["samples_per_s"],
            "without_diagnostic_backward": without_diag["samples_per_s"] if without_diag else None,
            "soft_number": "torch CPU convolutions at these sizes: 10x by thread count alone (see sweep); a reported baseline, not a target",
            "host_threads_available": n_all, "method": CPU_BASELINE_METHOD, "sweep": sweep,
            "sample": f"oracle forward + the reference's diagnostic backward + backward + Adam: one step each at "
                      f"{[r['threads'] for r in sweep if r['windows'] == 1]} threads on 1 of the batch's {eng.B} windows "
                      f"({g.n_win} samples), the same without the diagnostic backward, then one step on "
                      + (f"all {eng.B} windows at {fullb['threads']} threads" if fullb else "no full batch (time budget)")
                      + f"; fastest one-window step: {best['threads']} threads, {best['seconds']} s; "
                        f"{time.time() - t_start:.0f} s of host time in total"}, parity


def box_record(lib, eng, device):
    """What THIS box sustains, measured in this process before the timed region (a 7.29 vs 6.92 ms pair of
    driver runs could not be attributed): the pure-MFMA rate, a 90 MB device copy, and the stand-alone launch time of one
    gated GEMM and one residual GEMM of the stack (layer 2: d = 4, the one-window kernel; K = 256)."""
    import ctypes as C
    import torch
    from ae_wavenet_amd import _lib as L
    rec = {}
    nbytes = 90 << 20
    scratch = torch.empty(2 * nbytes, dtype=torch.uint8, device=device)
    out = (C.c_float * 2)()
    st = torch.cuda.current_stream(device).cuda_stream
    L.check(lib.aew_probe_box(scratch.data_ptr(), scratch.numel(), nbytes, C.c_void_p(st), out), "aew_probe_box")
    rec["mfma_bf16_tflops"] = round(float(out[0]), 1)
    rec["copy_90MB_TBps"] = round(float(out[1]), 3)
    # the same copy 600 times back to back (~15 ms of full memory load), the last 400 timed: the burst figure above is equal
    # to 1 % on boxes whose step differs by 4 % (round 5: 6.63 .. 6.89 ms)
    a, b = scratch[:nbytes], scratch[nbytes:]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(200):
        b.copy_(a)
    e0.record()
    for _ in range(400):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    rec["sustained_copy_TBps"] = round(2 * nbytes * 400 / (1e-3 * e0.elapsed_time(e1)) / 1e12, 3)
    del a, b, scratch
    for name in ("G1.2", "G2.2"):
        fb = eng.fwd_b
        if name not in fb.labels:
            continue
        sp = eng._sub_plan("probe", fb, lambda i, lab: lab == name)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            sp.run(st)
        ev0.record()
        n = 20
        for _ in range(n):
            sp.run(st)
        ev1.record()
        torch.cuda.synchronize()
        rec[f"launch_us_{name}"] = round(1e3 * ev0.elapsed_time(ev1) / n, 2)
        if name == "G1.2":
            # ... and the same launch 400 times back to back (~20 ms of full load), the last 200 timed: two boxes with the
            # same burst figures have differed by 3 % on the step (round 5: 6.63 vs 6.81 ms) - clocks under sustained load
            for _ in range(200):
                sp.run(st)
            ev0.record()
            for _ in range(200):
                sp.run(st)
            ev1.record()
            torch.cuda.synchronize()
            rec[f"sustained_us_{name}"] = round(1e3 * ev0.elapsed_time(ev1) / 200, 2)
    try:
        rec["device"] = torch.cuda.get_device_properties(device).name
    except Exception:
        pass
    return rec


def kernel_source_sha():
    """sha256 over the kernel sources and the C header: ties a committed PMC table to the kernels it was measured on."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "ae-wavenet_amd", "csrc")
    for path in sorted(os.path.join(csrc, f) for f in os.listdir(csrc)) + [os.path.join(ROOT, "include", "aewavenet.h")]:
        if path.endswith((".hip", ".h")):
            h.update(os.path.basename(path).encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()


def pmc_traffic(kernel_prefix, tag="r06"):
    """HBM bytes per launch (a number, as the bench contract asks) of the dominant kernel from the committed PMC passes
    (profiles/<tag>_pmc_hbm_traffic.csv: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command,
    FETCH doubled per MI355X_MICROARCH.md; written by tools/measure_round.sh together with <tag>_pmc_hbm_traffic.sha =
    kernel_source_sha() of the sources it ran).  bench.py cannot collect PMCs itself, so the table is only used when
    its sha matches the sources of THIS build; otherwise traffic is null.  Returns (bytes | None, source note)."""
    path = os.path.join(ROOT, "profiles", f"{tag}_pmc_hbm_traffic.csv")
    sha_path = path[:-4] + ".sha"
    try:
        want = open(sha_path).read().strip()
    except OSError:
        return None, f"no profiles/{tag}_pmc_hbm_traffic.sha"
    if want != kernel_source_sha():
        return None, f"profiles/{tag}_pmc_hbm_traffic.csv is stale (kernel sources changed since it was measured)"
    try:
        import csv
        n = tot = 0.0
        for r in csv.DictReader(open(path)):
            if kernel_prefix + "<" in r["kernel"]:
                k = float(r["launches"])
                n += k
                tot += k * (float(r["fetch_MB_corrected_x2"]) + float(r["write_MB"])) * 1e6
        if n:
            return round(tot / n), f"profiles/{tag}_pmc_hbm_traffic.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH x 2)"
    except Exception as e:
        return None, f"{type(e).__name__}: {e}"
    return None, f"kernel {kernel_prefix} not in profiles/{tag}_pmc_hbm_traffic.csv"


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn(args)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # AEW_BENCH_SHARE_GPU=1 + AEW_BENCH_BACKEND=gloo: all ranks on cuda:0 over gloo - a functional check of the
    # multi-process path on a one-GPU box (tools/measure_round.sh), not a measurement
    share = os.environ.get("AEW_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("AEW_BENCH_BACKEND", "nccl")
    if share:
        local = 0
    if not share and torch.cuda.device_count() <= local:
        raise SystemExit(f"rank {rank}: no cuda:{local} on this node ({torch.cuda.device_count()} devices)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dp = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")
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
    if args.nt_window >= 0:
        lib.aew_set_nt_window(args.nt_window)
    if args.nf_loaders >= 0:
        lib.aew_set_nf_loaders(args.nf_loaders)
    if args.nf_deep >= 0:
        lib.aew_set_nf_deep(args.nf_deep)
    if args.nt_small_waves >= 0:
        lib.aew_set_nt_small_waves(args.nt_small_waves)
    if args.nt_deep >= 0:
        lib.aew_set_nt_small_deep(args.nt_deep)
    if args.tn_blocks:
        lib.aew_set_tn_target_blocks(args.tn_blocks)
    if os.environ.get("AEW_TNB") is not None:                    # A/B aid: big-tile wgrad kernel off (0) / split-K block target
        v = int(os.environ["AEW_TNB"])
        lib.aew_set_tn_big(1 if v > 0 else 0, v)
    if args.tn_small:
        a, b = (int(v) for v in args.tn_small.split(","))
        lib.aew_set_tn_small(a, b)
    from ae_wavenet_amd import engine as _E
    _E.DecoderPlan.split_chains = args.chains == 2
    _E.DecoderPlan.split_chains_bwd = args.chains_bwd == 2
    if args.wgrad_cursor == "off":
        _E.DecoderPlan.wgrad_cursor = False
    elif args.wgrad_cursor:
        ce, cd = (int(v) for v in args.wgrad_cursor.split(","))
        _E.DecoderPlan.wgrad_cursor = True
        L.check(lib.aew_set_tn_cursor(ce, cd), "aew_set_tn_cursor")
    _E.DecoderPlan.tail_lane = args.tail_lane
    from ae_wavenet_amd import model as _M0
    if args.nt_chain:
        v = [int(x) for x in args.nt_chain.split(",")]
        _M0.TrainEngine.nt_chain, _M0.TrainEngine.nt_chain_bwd = v[0], (v[1] if len(v) > 1 else 0)
    if args.merge_packs >= 0:
        _M0.TrainEngine.merge_packs = bool(args.merge_packs)
    if args.side_lanes:
        _E.DecoderPlan.n_side_lanes = args.side_lanes
    if os.environ.get("AEW_DIAG_EARLY") is not None:             # A/B aid: 0 = per-step diagnostics at the tail of the forward plan
        from ae_wavenet_amd import model as _M
        _M.TrainEngine.diag_early = os.environ["AEW_DIAG_EARLY"] == "1"
    if os.environ.get("AEW_SPLIT_MULTISEG"):
        _E.DecoderPlan.split_multiseg = os.environ["AEW_SPLIT_MULTISEG"] == "1"

    from ae_wavenet_amd.jitter import DeviceJitter
    from ae_wavenet_amd.loader import DevicePrefetcher
    hps, model, opt = make_model(args, device)
    eng = model._ensure_engine(args.batch)
    sharded = os.environ.get("AEW_DP_SHARDED", "1") == "1"       # reduce-scatter + sharded Adam + all-gather (dp.py)
    bf16_grads = os.environ.get("AEW_DP_BF16_GRADS", "0") == "1"
    if dp is not None:
        dp.attach(model, sharded=sharded, bf16_grads=bf16_grads)  # collectives inside run() / backward() / step()
        dp.broadcast_params(eng)
    loader = DevicePrefetcher(host_batches(model, args.batch, rank), device, depth=2,
                              jitter=DeviceJitter(args.jitter_prob, seed=2507 + rank))

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the step through the boundary ----------------------------------------------------------------------
    # `value`: inputs already resident in HBM when the timed region starts (the bench contract; the reference's run()
    # takes device tensors too, chassis.py:151-171) - a pool of distinct batches with their jitter indices.
    # `host_fed`: the same step fed from pinned host batches through the prefetcher (PCIe-inclusive; never `value`).
    jit = DeviceJitter(args.jitter_prob, seed=99 + rank)
    src = host_batches(model, args.batch, rank)
    resident = []
    for _ in range(8):
        wav, mel, voice, _none = next(src)
        resident.append((wav.to(device), mel.to(device), voice.to(device), jit(args.batch, mel.shape[2], device)))
    cursor = [0]

    def step():
        wav, mel, voice, jitter = resident[cursor[0] % len(resident)]
        cursor[0] += 1
        opt.zero_grad()
        pred, target, loss = model.run(wav, mel, voice, jitter)
        loss.backward()
        opt.step()

    def host_step():
        wav, mel, voice, jitter = next(loader)
        opt.zero_grad()
        pred, target, loss = model.run(wav, mel, voice, jitter)
        loss.backward()
        opt.step()

    from ae_wavenet_amd import model as _Mm
    gscale = dp.grad_scale(_Mm.MEAN_LOSS[eng.bn_type]) if dp is not None else 1.0

    def engine_step():
        if dp is not None and sharded:                            # the same exchange schedule as the surface step
            dp.train_step_sharded(eng, args.lr, gscale, bf16_grads=bf16_grads)
        elif dp is not None:
            dp.train_step(eng, args.lr, gscale)
        else:
            eng.forward()
            eng.backward()
            eng.adam_step(args.lr, 1.0)

    def timed(fn):
        for _ in range(args.warmup):
            fn()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    box = None
    if rank == 0 and not args.no_box:
        try:
            box = box_record(lib, eng, device)
        except Exception as e:                                   # never lose the bench line
            box = {"error": f"{type(e).__name__}: {e}"}
    head_fn, other_fn = (engine_step, step) if args.engine_only else (step, engine_step)
    dt = timed(head_fn)
    loss_val = float(eng.loss_buf[0])
    dt_other = timed(other_fn)
    dt_host = timed(host_step)
    dp_info = None
    if dp is not None:
        # how much of the exchange the compute stream actually waits for: events around every collective wait, over a
        # few extra steps outside the timed region (every rank; rank 0's numbers are printed)
        dp.finish()
        fence()
        dp.timing = True
        n_x = 5
        for _ in range(n_x):
            step()
        dp.finish()
        ex = dp.exposed_ms()
        dp.timing = False
        dp_info = {"exposed_collective_ms_per_step": round(sum(ex.values()) / n_x, 4),
                   "by_wait_ms_per_step": {k: round(v / n_x, 4) for k, v in sorted(ex.items())},
                   "backend": backend, "ranks_share_one_gpu": share}
        if args.check_replicas:
            fence()
            n = eng.ps.numel
            hi, lo_ = eng.ps.params[:n].clone(), eng.ps.params[:n].clone()
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
            dp_info["replica_param_max_diff"] = float((hi - lo_).abs().max())
            if eng.bn_type == "vqvae-ema":
                hi, lo_ = eng.emb.clone(), eng.emb.clone()
                dist.all_reduce(hi, op=dist.ReduceOp.MAX)
                dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
                dp_info["replica_codebook_max_diff"] = float((hi - lo_).abs().max())

    # ---- per-kernel timing for the roofline (outside the timed region) ----------------------
    # Two passes of HIP-event timing over the engine's plans (eager launches on the plans' stream, serial plan order):
    #   as run  (aew_timing_enable(2)): a chained launch (AEW_OP_NT_CHAIN: the forward's gated stack as ONE kernel,
    #           csrc/aew_chain.hip) is timed as the one launch it is in the timed region - what a rocprofv3 kernel
    #           trace of this command shows (k_nt_chain<0>, and k_gemm_nt_bf16 / _win for the stand-alone launches)
    #   serial  (aew_timing_enable(1)): every GEMM as its own launch - the per-kernel view of rounds 1-4, kept so that
    #           the numbers stay comparable and every body's own time stays visible (`roofline.serial`)
    def roofline_block():
        kern = {}
        import ctypes as C
        plans = [eng.fwd_a, eng.fwd_b, eng.bwd] + ([eng.cb] if hasattr(eng, "cb") else []) + [eng.opt]
        ops_all = [op for pl in plans for op in pl.ops]
        labels_all = [lab for pl in plans for lab in pl.labels]
        n_t = 3

        def timing_pass(mode):
            lib.aew_timing_enable(mode)
            for _ in range(n_t):
                eng.forward(None, timing=True)
                eng.backward(timing=True)
                eng.adam_step(args.lr, 1.0)
            cap = 1 << 16
            ms = (C.c_float * cap)()
            tags = (C.c_int32 * cap)()
            cnt = C.c_int(0)
            L.check(lib.aew_timing_read(ms, tags, cap, C.byref(cnt)), "timing_read")
            lib.aew_timing_enable(0)
            n = min(cnt.value, cap)
            assert n == n_t * len(ops_all), (n, len(ops_all))
            return [ms[i] for i in range(n)], [tags[i] for i in range(n)]

        KNAME = {0: "k_gemm_nt_bf16", 1: "k_gemm_nt_bf16_p64", 2: "k_fn", 6: "k_gemm_nt_bf16_win", 7: "k_nt_chain"}

        def ex_flops(g):
            return 2.0 * g.M * g.batch * g.N_pad * g.K_total + (2.0 * g.M * g.batch * g.N2_pad * (g.N_pad // 2) if g.W2 else 0.0)

        def by_kernel(ms, chained):
            """executed FLOPs, launches and time per kernel name; chained: an AEW_OP_NT_CHAIN op carries its stages' FLOPs
            (and their time: the stage ops are empty intervals in that pass)"""
            byk = {k: {"ms": 0.0, "launches": 0, "exec_flops": 0.0, "stages": 0} for k in KNAME}
            in_chain = 0
            for i in range(len(ms)):
                op = ops_all[i % len(ops_all)]
                if chained and op.kind == L.OP_NT_CHAIN and op.u.chain.stages:
                    n_ops = op.u.chain.n_ops
                    fl_ = sum(ex_flops(ops_all[(i % len(ops_all)) + 1 + q].u.nt) for q in range(n_ops))
                    byk[7]["ms"] += ms[i] / n_t
                    byk[7]["launches"] += 1
                    byk[7]["stages"] += n_ops
                    byk[7]["exec_flops"] += fl_ / n_t
                    in_chain = n_ops
                    continue
                if in_chain > 0:
                    in_chain -= 1
                    continue
                if op.kind != L.OP_GEMM_NT or op.u.nt.dtype != L.BF16:
                    continue
                kid = lib.aew_nt_kernel(C.byref(op.u.nt))
                if kid not in byk:
                    continue
                byk[kid]["ms"] += ms[i] / n_t
                byk[kid]["launches"] += 1
                byk[kid]["stages"] += 1
                byk[kid]["exec_flops"] += ex_flops(op.u.nt) / n_t
            for v in byk.values():
                v["launches"] //= n_t
                v["stages"] //= n_t
            return byk

        ms2, tags2 = timing_pass(2)
        from ae_wavenet_amd import plan as _PLN
        torch.cuda.synchronize()                                 # (read before the serial pass: its zero op clears them)
        chain_waits = {pl.name: {lab: dict(zip(("timeout_flag", "tiles_that_waited", "longest_wait_polls"), v))
                                 for lab, v in _PLN.chain_stats(pl).items()} for pl in (eng.fwd_b, eng.bwd)
                       if getattr(pl, "nt_chains", None)}
        ms1, tags1 = timing_pass(1)
        cls_ms, cls_n = {}, {}
        for i in range(len(ms2)):
            c = tags2[i] // 100
            # (a chained launch is tagged 0: its time belongs to class 1, the bf16 NT kernels)
            if ops_all[i % len(ops_all)].kind == L.OP_NT_CHAIN:
                c = 1
            cls_ms[c] = cls_ms.get(c, 0.0) + ms2[i] / n_t
            cls_n[c] = cls_n.get(c, 0) + (1 if ms2[i] > 0 else 0)
            kern[tags2[i]] = kern.get(tags2[i], 0.0) + ms2[i] / n_t
        if args.per_op:
            per = {}
            for i in range(len(ms1)):
                lab = labels_all[i % len(labels_all)]
                per[lab] = per.get(lab, 0.0) + ms1[i] / n_t
            for i in range(len(ms2)):                                # the chained launches, as run
                lab = labels_all[i % len(labels_all)]
                if lab.startswith("chain["):
                    per[lab] = per.get(lab, 0.0) + ms2[i] / n_t
            with open(args.per_op, "w") as fh:
                fh.write("# serial pass (every GEMM its own launch); chain[...] lines: the chained launch as run, NOT additional time\n")
                for lab, v in sorted(per.items(), key=lambda kv: -kv[1]):
                    fh.write(f"{v:9.4f}  {lab}\n")
        fl = eng.flops_per_step()
        # Forward + dgrad of the gated stack / post network run on the bf16 NT kernel bodies, wgrad on the bf16 TN kernel
        # (90.4 MFLOP per output sample per step, 2/3 of it NT).  The NT work is spread over kernel names - the
        # tiled kernel k_gemm_nt_bf16 and its one-window form k_gemm_nt_bf16_win (stand-alone launches: the backward),
        # k_nt_chain (the SAME two bodies, the forward's stack as one launch), k_gemm_nt_bf16_p64 (64-row tiles for launches
        # of a few tiles) and k_fn (the two 20-segment GEMMs) - so times are grouped by the kernel each op dispatches to,
        # the way a rocprofv3 kernel trace groups them by name, and the algorithmic NT FLOPs are apportioned by the FLOPs
        # the descriptors execute (padding is near-uniform).
        nt_flops = fl["step"] * 2.0 / 3.0

        def finish(byk):
            ex_all = sum(v["exec_flops"] for v in byk.values()) or 1.0
            for v in byk.values():
                v["alg_flops"] = nt_flops * v["exec_flops"] / ex_all
                v["tflops"] = v["alg_flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0
            return byk

        def family(byk, ids):
            d = {k: sum(byk[i][k] for i in ids) for k in ("ms", "launches", "stages", "exec_flops", "alg_flops")}
            d["tflops"] = d["alg_flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
            return d
        byk2, byk1 = finish(by_kernel(ms2, True)), finish(by_kernel(ms1, False))
        # The dominant kernel: the tiled bf16 NT bodies (nt_tile / win_tile of csrc/aew_gemm.hip, aew_win.hip) under
        # whichever kernel name they run - stand-alone (k_gemm_nt_bf16, k_gemm_nt_bf16_win) or as stages of a chained launch
        # (k_nt_chain).  `achieved` = their algorithmic FLOPs / their HIP-event time AS RUN in the timed region.
        dom, dom1 = family(byk2, (0, 6, 7)), family(byk1, (0, 6))
        n_launch = max(dom["launches"], 1)
        achieved = dom["tflops"]
        traffic, src = None, ""
        tr = [(k, pmc_traffic(KNAME[k])) for k in (0, 6, 7) if byk2[k]["launches"]]
        if tr and all(t[1][0] is not None for t in tr):
            traffic = round(sum(t[1][0] * byk2[t[0]]["launches"] for t in tr) / n_launch)
            src = tr[0][1][1]
        elif tr:
            src = next(t[1][1] for t in tr if t[1][0] is None)

        def view(byk):
            return {KNAME[k]: {"ms_per_step": round(v["ms"], 4), "launches": v["launches"], "gemm_stages": v["stages"],
                               "tflops": round(v["tflops"], 1), "frac": round(v["tflops"] / 2500.0, 4)}
                    for k, v in byk.items() if v["launches"]}
        nt_ms_all = cls_ms.get(1, 0.0)
        roof = {"bound": "mfma",
                "kernel": "the tiled bf16 NT bodies: k_nt_chain (forward stack, one launch) + k_gemm_nt_bf16 + k_gemm_nt_bf16_win (stand-alone)",
                "achieved": round(achieved, 2), "peak": 2500.0,
                "unit": "TFLOP/s", "frac": round(achieved / 2500.0, 4), "traffic": traffic,
                "launches_per_step": n_launch, "gemm_stages_per_step": dom["stages"],
                "avg_launch_ms": round(dom["ms"] / n_launch, 5), "ms_per_step": round(dom["ms"], 4),
                "alg_flops_per_launch": dom["alg_flops"] / n_launch,
                "by_kernel": view(byk2),
                "serial": {"note": "the same GEMMs, every one its own launch (aew_timing_enable(1): the per-kernel view of rounds 1-4)",
                           "achieved": round(dom1["tflops"], 2), "frac": round(dom1["tflops"] / 2500.0, 4),
                           "launches_per_step": dom1["launches"], "ms_per_step": round(dom1["ms"], 4),
                           "avg_launch_ms": round(dom1["ms"] / max(dom1["launches"], 1), 5), "by_kernel": view(byk1)},
                "all_bf16_nt": {"achieved": round(nt_flops / (nt_ms_all * 1e-3) / 1e12, 2) if nt_ms_all > 0 else 0.0,
                                "ms_per_step": round(nt_ms_all, 4)},
                "tn_bf16": {"achieved": round((fl["step"] / 3.0) / (cls_ms.get(2, 1e9) * 1e-3) / 1e12, 2),
                            "frac": round((fl["step"] / 3.0) / (cls_ms.get(2, 1e9) * 1e-3) / 1e12 / 2500.0, 4),
                            "ms_per_step": round(cls_ms.get(2, 0.0), 4),
                            "note": "weight gradients: k_gemm_tn_bf16_grp (grouped, one result per matrix) + k_gemm_tn_bf16"},
                "ms_per_step_by_class": {str(k): round(v, 4) for k, v in sorted(cls_ms.items())}}
        nt_ms = dom["ms"]
        # the same launches against the HBM roofline: measured bytes per launch (PMC) / measured time per launch.  At
        # K = 256..1024 with 3-4 activation tensors in and out the stack's intensity (~240 FLOP/B) is below the ridge
        # (2500 / 8 = 312), see profiles/r01_op_roofline.txt
        roof["traffic_source"] = src
        if roof["traffic"]:
            tbps = roof["traffic"] / (nt_ms / n_launch * 1e-3) / 1e12
            roof["hbm_view"] = {"achieved": round(tbps, 3), "peak": 8.0, "unit": "TB/s", "frac": round(tbps / 8.0, 4)}
        roof["chain_waits"] = chain_waits
        return roof, kern

    roof, kern = None, {}
    if rank == 0:
        try:
            roof, kern = roofline_block()
        except Exception as e:                                   # never lose the bench line
            import traceback
            roof = {"error": f"{type(e).__name__}: {e}", "trace": traceback.format_exc()[-600:]}

    cpu, parity = None, None
    if world > 1:
        cpu = {"value": None, "unit": "samples/s", "cores": 0, "kind": "port",
               "sample": "measured at --gpus 1 only (rank 0 would hold the other ranks for ~30 s)"}
    elif rank == 0 and not args.no_cpu_baseline:
        try:
            cpu, parity = cpu_baseline(hps, eng)
        except Exception as e:                                   # never lose the GPU line
            cpu = {"value": None, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
                   "sample": f"failed: {type(e).__name__}: {e}"}

    if rank == 0:
        n_ranks = dist.get_world_size() if world > 1 else 1
        samples = n_ranks * args.batch * args.n_win * args.steps
        names = ("engine", "boundary") if args.engine_only else ("boundary", "engine")
        out = {
            "metric": "16kHz audio samples/sec/step (VQ-VAE-EMA train)",
            "value": samples / dt, "unit": "samples/s", "n_gpus": n_ranks, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("VQ-VAE-EMA (arch.vqvae-ema shape: 2x10 gated layers, 368 res / 256 dil "
                                    "/ 256 skip, K=4096 d=32) fwd+bwd+Adam") if args.arch == "vqvae-ema" else
                                   f"SIDE MEASUREMENT, not the bench line: --arch {args.arch} fwd+bwd+Adam",
                       "headline_config": args.arch == "vqvae-ema" and args.batch == 8 and args.n_win == 5000,
                       "timed_region": {"boundary": "AutoEncoder.run(device tensors) -> loss.backward() -> FusedAdam.step(), batches resident in HBM",
                                        "engine": "TrainEngine.forward / backward / adam_step on one resident batch"}[names[0]],
                       "global_batch": n_ranks * args.batch, "n_win_batch": args.n_win, "jitter_prob": args.jitter_prob,
                       "parallelism": f"dp{n_ranks}" + ("" if n_ranks == 1 else (" reduce-scatter + sharded Adam + all-gather" if sharded else " all-reduce")
                                                        + (" (bf16 gradient transport)" if bf16_grads else "")),
                       "loss": float(loss_val),
                       "decoder": "bf16 MFMA, fp32 accumulate", "encoder_vq": "fp32 MFMA exact chain (forward), bf16 MFMA (backward)"},
            names[1] + "_only" if names[1] == "engine" else "through_boundary":
                {"ms_per_step": 1e3 * dt_other / args.steps, "value": samples / dt_other, "unit": "samples/s"},
            "host_fed": {"ms_per_step": 1e3 * dt_host / args.steps, "value": samples / dt_host, "unit": "samples/s",
                         "path": "pinned host batch -> DevicePrefetcher (H2D on a copy stream, jitter generated on the device) -> the "
                                 "same step: PCIe-inclusive, reported beside `value`, never as it"},
            "roofline": roof, "cpu_baseline": cpu, "parity": parity, "box": box, "data_parallel": dp_info,
            "kernel_ms_by_tag": {str(k): round(v, 4) for k, v in sorted(kern.items())},
        }
        print(json.dumps(out))
    if world > 1:
        sys.stdout.flush()
        dist.barrier()                      # rank 0 arrives last (per-op timing pass): everybody leaves together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
