#!/usr/bin/env python3
"""How much independent work fits UNDER the backward's dgrad chain?  (priced before building it.)

The review's proposal: run the backward's dz / dx stages (wavenet.py:354-357 reversed; 2.04 ms, stages of 448 / 672 tiles
on 512 resident slots) as one dynamically scheduled launch and let the grouped weight gradient's tiles (1.6 ms of
independent work) fill the slots the dependent stages leave idle.  What that can return is bounded by what a CU that
already holds one dgrad tile can still give to a second, unrelated workgroup - this probe measures it with the hardware's
own scheduler instead of a hand-written one:

  A  the dgrad chain of the real backward plan (d.post2, d.post1, dz.19, dx.19 ... dx.0: 42 dependent launches), on a
     HIGH-priority stream;
  B  filler: weight-gradient work in SHORT workgroups - 8-wave 128 x 256 tiles (k_gemm_tn_bf16_grp8: the NT bodies'
     footprint, 72 KiB and 8 waves, so that one fits beside one dgrad workgroup on a CU) over 1024-row contractions
     (~40 us each, like a dgrad tile), operands streamed from HBM with the real launch's reuse (~3x) - on a LOW-priority
     stream: the dispatcher gives it the slots A does not ask for.

Reported: A alone, B alone, both started together (each stream's own elapsed time and the wall time of the pair).
  harvest = T(A alone) + T(B alone) - T(pair): what co-scheduling returned;
  the slowdown of A under B = what the dependent chain loses when its tiles share their CUs.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1024, help="contraction length of a filler tile (32 rows per K step)")
    ap.add_argument("--launches", type=int, default=5)
    ap.add_argument("--descs", type=int, default=512, help="filler matrices (512 x 512: 8 tiles each) per launch")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--tile", type=int, default=384, help="384: 8-wave tiles (72 KiB); 128: the 4-wave 128 x 128 tiles (48 KiB)")
    ap.add_argument("--chain-bwd", dest="chain_bwd", type=int, default=0, help="64: the dgrad chain as ONE chained launch")
    args = ap.parse_args()
    import torch
    from ae_wavenet_amd import _lib as L, autoencoder_model as ae, config, model as M
    from ae_wavenet_amd.plan import Mat, Plan, TnGroupBuilder, Workspace, make_tn
    dev = torch.device("cuda", 0)
    M.TrainEngine.nt_chain_bwd = args.chain_bwd
    hps = config.make_hps("vqvae-ema", n_win_batch=5000, n_batch=8, jitter_prob=0.12)
    torch.manual_seed(2507)
    model = ae.AutoEncoder(hps, n_mel=39).to(dev)
    eng = model._ensure_engine(8)
    g = eng.geom
    gen = torch.Generator().manual_seed(0)
    eng.set_inputs(torch.randint(0, 256, (8, g.enc_in_len), generator=gen).float().to(dev), torch.randn(8, 39, g.mel_len, generator=gen).to(dev),
                   torch.randint(0, 40, (8,), generator=gen).to(dev), torch.arange(g.embed_len).repeat(8, 1).to(dev))
    for _ in range(3):
        eng.forward(); eng.backward(); eng.adam_step(1e-4)
    torch.cuda.synchronize()
    sel = ("d.post", "dz.", "dx.", "chain[", "zero:chain.bwd")
    dgrad = eng._sub_plan("dgrad", eng.bwd, lambda i, lab: lab.startswith(sel))
    n_nt = sum(1 for lab in dgrad.labels if lab.startswith(("d.post", "dz.", "dx.")))
    # ---- filler
    ws = Workspace(dev)
    nd, rows = args.descs, args.rows
    G = Mat.new(ws, "fill.G", nd, rows, 512, L.BF16)
    A = Mat.new(ws, "fill.A", nd, rows, 512, L.BF16)
    ws.get("fill.G").normal_(); ws.get("fill.A").normal_()
    out = ws.alloc("fill.out", 512 * 512, torch.float32)
    fill = Plan("filler")
    gb = TnGroupBuilder(ws, "fill.tng", args.tile)
    for d in range(nd):
        t = make_tn(L.BF16, rows, 1, 512, 512, G.seg(512, b0=d), [A.seg(512, b0=d)])
        t.out, t.out_batch_stride = out.data_ptr(), 512 * 512
        gb.add(t, f"f{d}")
    gb.emit(fill, "filler")
    n_wg = sum(1 for r in gb.tile_map() if r >= 0)
    hi, lo = torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev, priority=0)

    def run_a():
        dgrad.run_graph(hi.cuda_stream)

    def run_b():
        for _ in range(args.launches):
            fill.run_graph(lo.cuda_stream)

    def timed(fa, fb):
        ea0, ea1, eb0, eb1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if fb:
            eb0.record(lo); fb(); eb1.record(lo)
        if fa:
            ea0.record(hi); fa(); ea1.record(hi)
        torch.cuda.synchronize()
        wall = 1e3 * (time.perf_counter() - t0)
        return (ea0.elapsed_time(ea1) if fa else 0.0, eb0.elapsed_time(eb1) if fb else 0.0, wall)
    for _ in range(3):
        timed(run_a, None); timed(None, run_b); timed(run_a, run_b)
    res = {"A": [], "B": [], "AB": []}
    for _ in range(args.reps):
        res["A"].append(timed(run_a, None))
        res["B"].append(timed(None, run_b))
        res["AB"].append(timed(run_a, run_b))
    med = lambda xs: sorted(xs)[len(xs) // 2]
    a = med([r[0] for r in res["A"]]); b = med([r[1] for r in res["B"]])
    ab_a = med([r[0] for r in res["AB"]]); ab_b = med([r[1] for r in res["AB"]]); ab_w = med([max(r[0], r[1]) for r in res["AB"]])
    print(f"dgrad chain ({n_nt} GEMMs{', ONE chained launch' if args.chain_bwd else ''}) alone: {a:.3f} ms")
    print(f"filler ({args.launches} launches x {n_wg} workgroups of {rows // 32} K steps, tile {args.tile}) alone: {b:.3f} ms")
    print(f"together: dgrad stream {ab_a:.3f} ms, filler stream {ab_b:.3f} ms, both done after ~{ab_w:.3f} ms")
    print(f"harvest = {a + b - ab_w:.3f} ms of {a + b:.3f}; the dgrad chain slows by {ab_a - a:+.3f} ms under the filler")


if __name__ == "__main__":
    main()
