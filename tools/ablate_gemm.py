#!/usr/bin/env python3
"""Ablation timing of the bf16 NT GEMM on a G1-shaped problem (gated layer of the decoder:
B=8, 7000 rows, K = 2*384 + 128, N = 512).  Variants switch off parts of the kernel via
aew_gemm_nt_t.reserved (bit0 MFMA, bit1 LDS fragment reads, bit2 LDS-DMA, bit3 epilogue)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ae_wavenet_amd import _lib as L
from ae_wavenet_amd.plan import Mat, Plan, Workspace, make_nt

lib = L.load()
lib.aew_set_nt_wave_rows(int(os.environ.get('WAVE_ROWS', '64')))
lib.aew_set_nt_pipe(int(os.environ.get('PIPE', '1')))
dev = "cuda:0"
B, T, Rp, Cp, Dp = 8, 7046, int(os.environ.get('RP', '384')), 128, 256
ws = Workspace(dev)
x = Mat.new(ws, "x", B, T, Rp, L.BF16); cond = Mat.new(ws, "cond", B, T, Cp, L.BF16)
W = Mat.new(ws, "W", 1, 2 * Dp, 2 * Rp + Cp, L.BF16)
z = Mat.new(ws, "z", B, T, Dp, L.BF16); pf = Mat.new(ws, "pf", B, T, Dp, L.BF16); pg = Mat.new(ws, "pg", B, T, Dp, L.BF16)
bias = ws.alloc("bias", B * 2 * Dp, torch.float32)
for n in ("x", "cond"):
    ws.get(n).copy_((torch.rand(ws.get(n).shape, device=dev) * 2 - 1).bfloat16())
ws.get("W").copy_(((torch.rand(ws.get("W").shape, device=dev) * 2 - 1) * 0.05).bfloat16())
d = 16
M = T - d
flops = 2.0 * B * M * (2 * Rp + Cp) * 2 * Dp
names = {0: "full", 1: "no MFMA", 2: "no ds_read", 4: "no DMA", 8: "no epilogue", 6: "MFMA only (+epi)",
         14: "MFMA only", 13: "ds_read only", 11: "DMA only", 15: "barriers only", 12: "no DMA, no epi",
         32: "exit at entry", 16: "exit after setup"}
for v in [int(x) for x in os.environ.get('ABL', '0,8,1,2,4,12,14,13,11,15').split(',')]:
    g = make_nt(L.BF16, M, Dp, 2 * Dp, B, [x.seg(Rp), x.seg(Rp, row_off=d), cond.seg(Cp, row_off=d)], W.ptr,
                epi=L.EPI_GATED, out0=z.view(), out1=pf.view(), out2=pg.view(), bias_ptr=bias.data_ptr(),
                bias_bs=2 * Dp)
    g.reserved = v
    p = Plan("abl")
    for _ in range(20):
        p.add(L.OP_GEMM_NT, g, "g1", 1)
    p.run(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    p.run(torch.cuda.current_stream().cuda_stream)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"variant {v:2d} {names.get(v, ''):18s}: {us:8.1f} us  ({flops / us / 1e6:7.1f} TFLOP/s-equivalent)")

if os.environ.get('CLOCK'):
    cnt = torch.zeros(16, dtype=torch.int64, device=dev)
    g = make_nt(L.BF16, M, Dp, 2 * Dp, B, [x.seg(Rp), x.seg(Rp, row_off=d), cond.seg(Cp, row_off=d)], W.ptr,
                epi=L.EPI_GATED, out0=z.view(), out1=pf.view(), out2=pg.view(), bias_ptr=bias.data_ptr(),
                bias_bs=2 * Dp, counter_ptr=cnt.data_ptr())
    g.reserved = 1024 | int(os.environ.get('CLOCK'))
    p = Plan("clk")
    p.add(L.OP_GEMM_NT, g, "g1", 1)
    p.run(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    c = cnt.cpu().tolist()
    nb, steps = c[8], (2 * Rp + Cp) // 32
    names = ["prologue", "vmcnt wait", "barrier", "LDS fragment wait", "DMA issue + MFMA issue", "drain idle DMA", "epilogue issue", "stores retire"]
    print(f"phase clock (s_memtime ticks = 100 MHz? see total), {nb} blocks, wave 0 of each, {steps} K steps:")
    tot = sum(c[:8])
    for n, v in zip(names, c[:8]):
        per = v / nb
        print(f"   {n:26s} {per:10.0f} ticks/block  {100.0 * v / tot:5.1f} %   per K step {per / steps:8.1f}")
