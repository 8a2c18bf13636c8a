#!/usr/bin/env python3
"""Tile-wave quantisation (round 4): a launch of T tiles on 256 CUs x 2 block slots ends with a ragged round (G2: 2.53 tiles
per CU -> the CUs with three set the time).  Would finishing with HALF tiles help?  Emulated without a new kernel: the same
op as two launches on two streams - rows [0, R) of every window on 256-row tiles, rows [R, M) on 128-row tiles
(aew_set_nt_mem128(1) at launch time) - so the half tiles fill slots as the full ones drain.  Against the single launch.
    python tools/tail_split_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ae_wavenet_amd import _lib as L
from ae_wavenet_amd.plan import Mat, Plan, Workspace, make_nt

lib = L.load()
lib.aew_set_nt_rows192(0)
dev = "cuda:0"
B = 8
Rp, Dp, Sp = 384, 256, 256
N_REP = 6


def plan_of(g):
    p = Plan("ts")
    for _ in range(N_REP):
        p.add(L.OP_GEMM_NT, g, "g", 1)
    return p


def shifted(g, r0, rows):
    """the same op restricted to GEMM rows [r0, r0 + rows) of every batch element"""
    import copy
    h = copy.deepcopy(g) if False else type(g).from_buffer_copy(bytes(g))
    h.M = rows
    for i in range(h.n_segs):
        h.seg[i].row_off += r0 * h.seg[i].row_step
    for nm in ("out0", "out1", "out2", "aux0", "aux1"):
        v = getattr(h, nm)
        if v.ptr:
            v.row_off += r0 * v.row_step
    return h


def main():
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()

    def timed(fn):
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(4):
            e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record(s0)
            s1.wait_event(e0)
            fn()
            ea.record(s0)
            eb.record(s1)
            torch.cuda.synchronize()
            best = min(best, max(e0.elapsed_time(ea), e0.elapsed_time(eb)) * 1e3 / N_REP)
        return best
    for M in (6900, 6000):
        ws = Workspace(dev)
        z = Mat.new(ws, "z", B, M, Dp, L.BF16)
        x = Mat.new(ws, "x", B, M + 64, Rp, L.BF16)
        xo = Mat.new(ws, "xo", B, M + 64, Rp, L.BF16)
        dskp = Mat.new(ws, "dskp", B, M, Sp, L.BF16)
        pf = Mat.new(ws, "pf", B, M, Dp, L.BF16)
        pg = Mat.new(ws, "pg", B, M, Dp, L.BF16)
        dfg = Mat.new(ws, "dfg", B, M, 2 * Dp, L.BF16)
        Wrs = Mat.new(ws, "Wrs", 1, Rp, Dp, L.BF16)
        WrsT = Mat.new(ws, "WrsT", 1, Dp, Rp + Sp, L.BF16)
        for n in ("z", "x", "dskp", "pf", "pg", "xo"):
            ws.get(n).copy_((torch.rand(ws.get(n).shape, device=dev) * 2 - 1).bfloat16())
        for n in ("Wrs", "WrsT"):
            ws.get(n).copy_(((torch.rand(ws.get(n).shape, device=dev) * 2 - 1) * 0.05).bfloat16())
        ops = {
            "G2": (make_nt(L.BF16, M, 368, Rp, B, [z.seg(Dp)], Wrs.ptr, flags=L.EF_ADD_AUX0, out0=xo.view(), aux0=x.view(row_off=16)), 3),
            "dz": (make_nt(L.BF16, M, Dp, Dp, B, [xo.seg(Rp, hi=M), dskp.seg(Sp)], WrsT.ptr, epi=L.EPI_DFG, aux0=pf.view(),
                           aux1=pg.view(), out0=dfg.view()), 2),
        }
        for name, (g, n_nt) in ops.items():
            full = plan_of(g)
            lib.aew_set_nt_mem128(0)
            full.run(s0.cuda_stream)
            t_full = timed(lambda: full.run(s0.cuda_stream))
            tiles = -(-M // 256) * B * n_nt
            line = f"{name} rows {M}: {tiles} tiles = {tiles / 256:.2f} per CU, one launch {t_full:6.1f} us |"
            for k_full in sorted({(256 * j) // (B * n_nt) for j in (1, 2, 3)} | {(512) // (B * n_nt)}):
                r_full = 256 * k_full
                if r_full <= 0 or r_full >= M - 128:
                    continue
                pa, pb = plan_of(shifted(g, 0, r_full)), plan_of(shifted(g, r_full, M - r_full))

                def both():
                    lib.aew_set_nt_mem128(0)
                    pa.run(s0.cuda_stream)
                    lib.aew_set_nt_mem128(1)
                    pb.run(s1.cuda_stream)
                    lib.aew_set_nt_mem128(0)
                both()
                t = timed(both)
                n_a, n_b = k_full * B * n_nt, -(-(M - r_full) // 128) * B * n_nt
                line += f" {n_a} full + {n_b} half: {t:6.1f} |"
            print(line)
        del ws


if __name__ == "__main__":
    main()
