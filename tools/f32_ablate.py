#!/usr/bin/env python3
"""Where the time of the fp32 exact-chain NT kernel (k_gemm_nt_f32: encoder forward) goes, on the encoder's shapes, with
parts switched off through aew_gemm_nt_t.reserved (tools build only: hipcc ... -DAEW_FN_ABLATE=1 -o
lib/libaewavenet_hip_abl.so):  1 no MFMA, 2 no operand DMA, 4 no fragment reads, 8 no epilogue, 16 no barrier.
    AEW_LIB_PATH=ae-wavenet_amd/lib/libaewavenet_hip_abl.so python tools/f32_ablate.py     # on the GPU box
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ae_wavenet_amd import _lib as L
from ae_wavenet_amd.plan import Mat, Plan, Workspace, make_nt

lib = L.load()
dev = "cuda:0"
SHAPES = [("enc.1  8 x 68 rows, 3 taps of 768", 8, 68, 768, 3), ("enc.5  8 x 29 rows, 1 tap of 768", 8, 29, 768, 1),
          ("enc.2  8 x 33 rows, 4 taps stride 2", 8, 33, 768, 4)]
CASES = [(0, "full"), (8, "no epilogue"), (1, "no MFMA"), (4, "no fragment reads"), (5, "no MFMA, no fragment reads"),
         (2, "no operand DMA"), (7, "DMA waits + barriers only (no DMA, reads, MFMA)"), (13, "DMA + barriers only"),
         (29, "DMA only, no barrier"), (7 + 32, "skeleton without the transposes"), (7 + 64, "skeleton without issue bookkeeping"),
         (7 + 32 + 64, "skeleton without transposes and issue bookkeeping"), (7 + 16 + 32 + 64 + 128, "empty loop"),
         (32, "full without transposes (wrong result)")]


def main():
    for deep in (0,):
        lib.aew_set_nf_deep(deep)
        print(f"==== aew_set_nf_deep({deep})")
        for si, (name, B, M, E, taps) in enumerate(SHAPES):
            ws = Workspace(dev)
            x = Mat.new(ws, "x", B, M + taps, E, L.F32)
            W = Mat.new(ws, "W", 1, E, E * taps, L.F32)
            y = Mat.new(ws, "y", B, M, E, L.F32)
            ws.get("x").copy_(torch.rand(ws.get("x").shape, device=dev) * 2 - 1)
            ws.get("W").copy_((torch.rand(ws.get("W").shape, device=dev) * 2 - 1) * 0.05)
            segs = [x.seg(E, row_off=j) for j in range(taps)]
            flops = 2.0 * B * M * E * E * taps
            print(f"== {name}: {flops / 1e9:.2f} GFLOP, chain {E * taps // 4} MFMAs = {E * taps // 4 * 32 / 2.4e3:.1f} us")
            for bits, label in CASES:
                g = make_nt(L.F32, M, E, E, B, segs, W.ptr, out0=y.view())
                g.reserved = bits
                p = Plan("t")
                for _ in range(10):
                    p.add(L.OP_GEMM_NT, g, "g", 1)
                st = torch.cuda.current_stream().cuda_stream
                p.run(st)
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); p.run(st); e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) * 1e3 / 10)
                print(f"   {label:50s} {best:7.1f} us")


if __name__ == "__main__":
    main()
