#!/usr/bin/env python3
"""In-loop rate of the bf16 NT kernel variants on plain GEMM shapes (STORE epilogue, one segment), to separate
the K loop from prologue / epilogue / tile-wave quantisation:  a square 8192 x 4096 x 4096 problem (long K, many
tiles) next to the decoder's shapes (M ~ 48000, N = 256..512, K = 256..1024).
    python tools/gemm_sweep.py            # on the GPU box
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ae_wavenet_amd import _lib as L
from ae_wavenet_amd.plan import Mat, Plan, Workspace, make_nt

lib = L.load()
dev = "cuda:0"
VARIANTS = [("thin 256x128 (default)", 64, 1), ("fat pipe 256x128", 128, 1), ("wide pipe 256x256 K32", 256, 1),
            ("p64 256x256 K64", 256, 2), ("p64 128x128 K64", 0, 1), ("p64 256x128 K64", 1, 1)]
VARIANTS.append(("full-N loader/consumer (impl 2)", -1, 0))
VARIANTS.append(("  impl 2, consumers idle (DMA only)", -2, 1))
VARIANTS.append(("  impl 2, loaders idle (LDS+MFMA only)", -2, 2))
VARIANTS.append(("  impl 2, barriers only", -2, 3))
SHAPES = [("square 8192x4096x4096", 1, 8192, 4096, 4096),
          ("long-K 8x8192 N512 K4096", 8, 8192, 512, 4096), ("long-K 8x8192 N384 K4096", 8, 8192, 384, 4096),
          ("long-K 8x8192 N256 K4096", 8, 8192, 256, 4096), ("long-K 8x8192 N128 K4096", 8, 8192, 128, 4096),
          ("G1-like 8x6000 N512 K896", 8, 6000, 512, 896), ("dx-like 8x6200 N384 K1024", 8, 6200, 384, 1024),
          ("dz-like 8x6000 N256 K640", 8, 6000, 256, 640), ("G2-like 8x6000 N384 K256", 8, 6000, 384, 256)]


def main():
    ws = Workspace(dev)
    for si, (name, B, M, N, K) in enumerate(SHAPES):
        Np = (N + 127) // 128 * 128
        x = Mat.new(ws, f"x{si}", B, M, K, L.BF16)
        W = Mat.new(ws, f"W{si}", 1, Np, K, L.BF16)
        y = Mat.new(ws, f"y{si}", B, M, Np, L.BF16)
        ws.get(f"x{si}").copy_((torch.rand(ws.get(f"x{si}").shape, device=dev) * 2 - 1).bfloat16())
        ws.get(f"W{si}").copy_(((torch.rand(ws.get(f"W{si}").shape, device=dev) * 2 - 1) * 0.05).bfloat16())
        flops = 2.0 * B * M * N * K
        print(f"== {name}: {flops / 1e9:.1f} GFLOP")
        ref = None
        for vname, rows, pipe in VARIANTS:
            if rows >= 0:
                lib.aew_set_nt_wave_rows(rows)
                lib.aew_set_nt_pipe(pipe)
            elif Np > 512:
                continue
            if rows == 256 and Np % 256:
                continue
            g = make_nt(L.BF16, M, N, Np, B, [x.seg(K)], W.ptr, out0=y.view(), impl=2 if rows < 0 else 0)
            if rows == -2:
                g.reserved = pipe
            p = Plan("sweep")
            for _ in range(10):
                p.add(L.OP_GEMM_NT, g, "g", 1)
            st = torch.cuda.current_stream().cuda_stream
            p.run(st)
            torch.cuda.synchronize()
            out = y.tensor().float().clone()
            if ref is None:
                ref = out
            if rows == -2:
                out = ref
            err = float((out - ref).abs().max())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(3):
                e0.record()
                p.run(st)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / 10)
            print(f"   {vname:28s} {best:9.1f} us  {flops / best / 1e6:8.1f} TFLOP/s   max|diff vs first| {err:.3g}")
        lib.aew_set_nt_wave_rows(64)
        lib.aew_set_nt_pipe(1)


if __name__ == "__main__":
    main()
