#!/usr/bin/env python3
"""What bounds the memory-bound half of the gated stack (round 4): the residual 1x1 + add (G2: K = 256, N = 384,
wavenet.py:108-109) and dz (K = 640, N = 256, DFG epilogue, backward of wavenet.py:100-102).  Every variant runs the
SAME kernel with the same instruction stream; only the descriptor's views change so that one stream of HBM traffic
at a time is served from cache instead (batch_stride = 0: all 8 batch elements share one window, 8x fewer distinct
bytes) or dropped (flag off).  The differences price each stream.
    python tools/membound_probe.py          # on the GPU box
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ae_wavenet_amd import _lib as L
from ae_wavenet_amd.plan import Mat, Plan, Workspace, make_nt

lib = L.load()
dev = "cuda:0"
B, M = 8, int(os.environ.get("ROWS", "6900"))
Rp, Dp, Sp = 384, 256, 256
REP = 10


def timeit(g, n=REP):
    p = Plan("probe")
    for _ in range(n):
        p.add(L.OP_GEMM_NT, g, "g", 1)
    st = torch.cuda.current_stream().cuda_stream
    p.run(st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(4):
        e0.record()
        p.run(st)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


def shared(v):
    """the same view with every batch element mapped onto batch element 0 (8x fewer distinct bytes)"""
    v.batch_stride = 0
    return v


def main():
    ws = Workspace(dev)
    z = Mat.new(ws, "z", B, M, Dp, L.BF16)
    x = Mat.new(ws, "x", B, M + 64, Rp, L.BF16)
    xo = Mat.new(ws, "xo", B, M, Rp, L.BF16)
    Wrs = Mat.new(ws, "Wrs", 1, Rp, Dp, L.BF16)
    dxn = Mat.new(ws, "dxn", B, M, Rp, L.BF16)
    dskp = Mat.new(ws, "dskp", B, M, Sp, L.BF16)
    pf = Mat.new(ws, "pf", B, M, Dp, L.BF16)
    pg = Mat.new(ws, "pg", B, M, Dp, L.BF16)
    dfg = Mat.new(ws, "dfg", B, M, 2 * Dp, L.BF16)
    WrsT = Mat.new(ws, "WrsT", 1, Dp, Rp + Sp, L.BF16)
    for n in ("z", "x", "dxn", "dskp", "pf", "pg"):
        ws.get(n).copy_((torch.rand(ws.get(n).shape, device=dev) * 2 - 1).bfloat16())
    for n in ("Wrs", "WrsT"):
        ws.get(n).copy_(((torch.rand(ws.get(n).shape, device=dev) * 2 - 1) * 0.05).bfloat16())
    # pure streaming reference: copy of as many bytes as G2 / dz move
    for mb in (98, 180):
        a = torch.empty(mb * 500000 // 2, dtype=torch.bfloat16, device=dev)
        b = torch.empty_like(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record()
            b.copy_(a)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3)
        print(f"copy {mb / 2:.0f} MB -> {mb / 2:.0f} MB: {best:7.1f} us = {mb * 1e6 / (best * 1e-6) / 1e12:.2f} TB/s")
        del a, b

    def g2(aux=True, sh_a=False, sh_aux=False, sh_out=False, N=Rp):
        a = z.seg(Dp)
        if sh_a:
            a.batch_stride = 0
        ax = x.view(row_off=16)
        ov = xo.view()
        if sh_aux:
            shared(ax)
        if sh_out:
            shared(ov)
        return make_nt(L.BF16, M, min(N, 368), N, B, [a], Wrs.ptr, flags=L.EF_ADD_AUX0 if aux else 0, out0=ov,
                       aux0=ax if aux else None)

    mb_g2 = B * M * (Dp + 2 * Rp) * 2 / 1e6
    print(f"== G2: {B} x {M} rows, K = {Dp}, N = {Rp}: {2.0 * B * M * Dp * Rp / 1e9:.1f} GFLOP, least bytes {mb_g2:.0f} MB")
    rows = [("full", g2()), ("no aux (flag off)", g2(aux=False)), ("aux shared (cache hits)", g2(sh_aux=True)),
            ("out shared (writes absorbed)", g2(sh_out=True)), ("A shared (K loop from cache)", g2(sh_a=True)),
            ("A + aux shared", g2(sh_a=True, sh_aux=True)), ("everything shared", g2(sh_a=True, sh_aux=True, sh_out=True)),
            ("no aux, A + out shared (no HBM at all)", g2(aux=False, sh_a=True, sh_out=True)),
            ("N = 128 (one N tile), full", g2(N=128)), ("N = 256, full", g2(N=256))]
    for name, g in rows:
        us = timeit(g)
        print(f"   {name:42s} {us:7.1f} us")

    def dz(sh_a=False, sh_aux=False, sh_out=False, skip_only=False):
        segs = ([] if skip_only else [dxn.seg(Rp)]) + [dskp.seg(Sp)]
        Wp = WrsT.ptr
        if sh_a:
            for s in segs:
                s.batch_stride = 0
        a0, a1, ov = pf.view(), pg.view(), dfg.view()
        if sh_aux:
            shared(a0), shared(a1)
        if sh_out:
            shared(ov)
        return make_nt(L.BF16, M, Dp, Dp, B, segs, Wp, epi=L.EPI_DFG, aux0=a0, aux1=a1, out0=ov)

    mb_dz = B * M * (Rp + Sp + 2 * Dp + 2 * Dp) * 2 / 1e6
    print(f"== dz: {B} x {M} rows, K = {Rp + Sp}, N = {Dp}: {2.0 * B * M * (Rp + Sp) * Dp / 1e9:.1f} GFLOP, least bytes {mb_dz:.0f} MB")
    rows = [("full", dz()), ("aux shared (pf, pg from cache)", dz(sh_aux=True)), ("out shared (writes absorbed)", dz(sh_out=True)),
            ("A shared (K loop from cache)", dz(sh_a=True)), ("aux + out shared", dz(sh_aux=True, sh_out=True)),
            ("everything shared", dz(sh_a=True, sh_aux=True, sh_out=True)), ("K = 256 only (skip term), full", dz(skip_only=True)),
            ("K = 256 only, aux + out shared", dz(skip_only=True, sh_aux=True, sh_out=True))]
    for name, g in rows:
        us = timeit(g)
        print(f"   {name:42s} {us:7.1f} us")


if __name__ == "__main__":
    main()
