#!/usr/bin/env python3
"""s_memtime phase clock and cut-down variants of the thin bf16 NT kernel on the decoder's four shapes (round 4).
Needs the tools library (kernels built with -DAEW_FN_ABLATE=1):
    AEW_LIB_PATH=ae-wavenet_amd/lib/libaewavenet_hip_abl.so python tools/phase_clock.py        # on the GPU box
Per shape: time of the full op and of variants that leave parts out (aew_gemm_nt_t.reserved bits: 1 MFMA, 2 fragment
reads, 4 LDS-DMA, 8 epilogue, 16 exit after setup, 32 exit at entry), then the phase clock of wave 0 of every block:
cycles in {prologue, vmcnt wait, barrier, fragment wait, MFMA + DMA issue, drain, epilogue issue, stores retire}."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ae_wavenet_amd import _lib as L
from ae_wavenet_amd.plan import Mat, Plan, Workspace, make_nt

lib = L.load()
lib.aew_set_nt_window(0)
lib.aew_set_nt_rows192(0)
dev = "cuda:0"
B, M = 8, int(os.environ.get("ROWS", "6900"))
Rp, Dp, Sp, Cp = 384, 256, 256, 128
NAMES = ["prologue", "vmcnt wait", "barrier", "fragment wait", "MFMA + DMA issue", "drain idle DMA", "epilogue issue",
         "stores retire"]


def run(g, n=10):
    p = Plan("pc")
    for _ in range(n):
        p.add(L.OP_GEMM_NT, g, "g", 1)
    st = torch.cuda.current_stream().cuda_stream
    p.run(st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        p.run(st)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


def main():
    ws = Workspace(dev)
    z = Mat.new(ws, "z", B, M, Dp, L.BF16)
    x = Mat.new(ws, "x", B, M + 64, Rp, L.BF16)
    cond = Mat.new(ws, "cond", B, M + 64, Cp, L.BF16)
    xo = Mat.new(ws, "xo", B, M + 64, Rp, L.BF16)
    Wrs = Mat.new(ws, "Wrs", 1, Rp, Dp, L.BF16)
    Wfg = Mat.new(ws, "Wfg", 1, 2 * Dp, 2 * Rp + Cp, L.BF16)
    WfgT = Mat.new(ws, "WfgT", 1, Rp, 4 * Dp, L.BF16)
    dskp = Mat.new(ws, "dskp", B, M, Sp, L.BF16)
    pf = Mat.new(ws, "pf", B, M, Dp, L.BF16)
    pg = Mat.new(ws, "pg", B, M, Dp, L.BF16)
    dfg = Mat.new(ws, "dfg", B, M, 2 * Dp, L.BF16)
    WrsT = Mat.new(ws, "WrsT", 1, Dp, Rp + Sp, L.BF16)
    bias = ws.alloc("bias", B * 2 * Dp, torch.float32)
    for n in ("z", "x", "cond", "dskp", "pf", "pg", "dfg"):
        ws.get(n).copy_((torch.rand(ws.get(n).shape, device=dev) * 2 - 1).bfloat16())
    for n in ("Wrs", "WrsT", "Wfg", "WfgT"):
        ws.get(n).copy_(((torch.rand(ws.get(n).shape, device=dev) * 2 - 1) * 0.05).bfloat16())
    cnt = torch.zeros(16, dtype=torch.int64, device=dev)
    d = 128
    shapes = {
        "G1 (K 896, N 512, gated)": lambda: make_nt(
            L.BF16, M, Dp, 2 * Dp, B, [x.seg(Rp), x.seg(Rp, row_off=d), cond.seg(Cp, row_off=d)], Wfg.ptr, epi=L.EPI_GATED,
            out0=z.view(), out1=pf.view(), out2=pg.view(), bias_ptr=bias.data_ptr(), bias_bs=2 * Dp, counter_ptr=cnt.data_ptr()),
        "G2 (K 256, N 384, + residual)": lambda: make_nt(
            L.BF16, M, 368, Rp, B, [z.seg(Dp)], Wrs.ptr, flags=L.EF_ADD_AUX0, out0=xo.view(), aux0=x.view(row_off=16),
            counter_ptr=cnt.data_ptr()),
        "dz (K 640, N 256, DFG)": lambda: make_nt(
            L.BF16, M, Dp, Dp, B, [xo.seg(Rp), dskp.seg(Sp)], WrsT.ptr, epi=L.EPI_DFG, aux0=pf.view(), aux1=pg.view(),
            out0=dfg.view(), counter_ptr=cnt.data_ptr()),
        "dx (K 1024, N 384, + dx_next)": lambda: make_nt(
            L.BF16, M, 368, Rp, B, [dfg.seg(2 * Dp), dfg.seg(2 * Dp, row_off=-d)], WfgT.ptr, flags=L.EF_ADD_AUX0,
            out0=xo.view(), aux0=x.view(row_off=16), counter_ptr=cnt.data_ptr()),
    }
    variants = [(0, "full"), (8, "no epilogue"), (1, "no MFMA"), (2, "no fragment reads"), (4, "no LDS-DMA"),
                (1 | 2, "DMA + barriers + epilogue"), (4 | 8, "no DMA, no epilogue"), (1 | 2 | 8, "DMA + barriers only"),
                (1 | 2 | 4, "barriers + epilogue"), (16, "exit after setup"), (32, "exit at entry")]
    for name, mk in shapes.items():
        g = mk()
        tiles = ((M + 255) // 256) * B * (g.N_pad // 128)
        print(f"== {name}: {2.0 * B * M * g.K_total * g.N / 1e9:.1f} GFLOP, {tiles} tiles of 256 x 128, {g.K_total // 32} K steps")
        for bits, vn in variants:
            g = mk()
            g.reserved = bits if bits else 2048          # 2048: the ablation build's kernel with every part on
            print(f"   {vn:28s} {run(g):7.1f} us")
        g = mk()
        g.reserved = 1024
        cnt.zero_()
        p = Plan("clk")
        p.add(L.OP_GEMM_NT, g, "g", 1)
        p.run(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        c = cnt.cpu().tolist()
        nb = max(c[8], 1)
        tot = sum(c[:8])
        print(f"   phase clock, wave 0 of {nb} blocks (s_memtime ticks per block; total {tot / nb:.0f}):")
        for n_, v in zip(NAMES, c[:8]):
            print(f"      {n_:20s} {v / nb:9.0f}  {100.0 * v / max(tot, 1):5.1f} %")


if __name__ == "__main__":
    main()
