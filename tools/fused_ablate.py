#!/usr/bin/env python3
"""Where the time of the fused gated layer (impl 2, csrc/aew_fn.hip) goes: the op on the bench workload's shapes with
parts switched off through aew_gemm_nt_t.reserved (bit 0 no LDS reads / MFMA in the first GEMM, bit 1 no operand DMA,
bit 2 no epilogues (gate math, stores), bit 3 no MFMA in the second GEMM), next to the unfused pair on the tiled kernels.
    python tools/fused_ablate.py        # on the GPU box
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ae_wavenet_amd import _lib as L
from ae_wavenet_amd.plan import Mat, Plan, Workspace, make_nt

lib = L.load()
dev = "cuda:0"
B, Rp, Cp, Dp = 8, 384, 128, 256


def run(T, d, label):
    ws = Workspace(dev)
    x = Mat.new(ws, "x", B, T, Rp, L.BF16); cond = Mat.new(ws, "cond", B, T, Cp, L.BF16)
    W = Mat.new(ws, "W", 1, 2 * Dp, 2 * Rp + Cp, L.BF16)
    W2 = Mat.new(ws, "W2", 1, Rp, Dp, L.BF16)
    z = Mat.new(ws, "z", B, T, Dp, L.BF16); pf = Mat.new(ws, "pf", B, T, Dp, L.BF16); pg = Mat.new(ws, "pg", B, T, Dp, L.BF16)
    xn = Mat.new(ws, "xn", B, T, Rp, L.BF16)
    bias = ws.alloc("bias", B * 2 * Dp, torch.float32)
    for n in ("x", "cond"):
        ws.get(n).copy_((torch.rand(ws.get(n).shape, device=dev) * 2 - 1).bfloat16())
    for n in ("W", "W2"):
        ws.get(n).copy_(((torch.rand(ws.get(n).shape, device=dev) * 2 - 1) * 0.05).bfloat16())
    M = T - d
    flops = 2.0 * B * M * ((2 * Rp + Cp) * 2 * Dp + Dp * Rp)
    segs = [x.seg(Rp), x.seg(Rp, row_off=d), cond.seg(Cp, row_off=d)]
    gk = dict(epi=L.EPI_GATED, out0=z.view(), out1=pf.view(), out2=pg.view(), bias_ptr=bias.data_ptr(), bias_bs=2 * Dp)

    def timed(ops, n=10):
        p = Plan("t")
        for _ in range(n):
            for o in ops:
                p.add(L.OP_GEMM_NT, o, "o", 1)
        st = torch.cuda.current_stream().cuda_stream
        p.run(st)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); p.run(st); e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / n)
        return best

    print(f"== {label}: B={B} rows={M} dil={d}  {flops / 1e9:.1f} GFLOP")
    g1 = make_nt(L.BF16, M, Dp, 2 * Dp, B, segs, W.ptr, impl=0, **gk)
    g2 = make_nt(L.BF16, M, Rp, Rp, B, [z.seg(Dp)], W2.ptr, flags=L.EF_ADD_AUX0, out0=xn.view(), aux0=x.view(row_off=d))
    t1, t2 = timed([g1]), timed([g2])
    print(f"   tiled kernels: gated {t1:7.1f} us + residual {t2:6.1f} us = {t1 + t2:7.1f} us   ({flops / (t1 + t2) / 1e6:7.1f} TFLOP/s)")
    g1f = make_nt(L.BF16, M, Dp, 2 * Dp, B, segs, W.ptr, impl=2, **gk)
    t = timed([g1f])
    print(f"   full-N gated alone (impl 2)            {t:7.1f} us")
    names = {0: "fused layer (impl 2)", 1: "  no LDS reads / MFMA in GEMM 1", 2: "  no operand DMA", 4: "  no epilogues", 8: "  no MFMA in GEMM 2",
             5: "  no GEMM 1 compute, no epilogues", 13: "  no compute at all, no epilogues", 15: "  barriers + launch only", 6: "  no DMA, no epilogues",
             9: "  no compute in either GEMM", 16: "  gate math but no global stores (incl. x_next)"}
    for v in (0, 16, 4, 2, 6, 13, 15):
        gf = make_nt(L.BF16, M, Dp, 2 * Dp, B, segs, W.ptr, impl=2, W2_ptr=W2.ptr, N2=Rp, N2_pad=Rp, out3=xn.view(),
                     aux0=x.view(row_off=d), **gk)
        gf.reserved = v
        t = timed([gf])
        print(f"   {names[v]:38s} {t:7.1f} us   ({flops / t / 1e6:7.1f} TFLOP/s-equivalent)")


run(7046, 16, "early layer")
run(6023, 512, "late layer")
run(6023, 16, "late layer rows, small dilation")
