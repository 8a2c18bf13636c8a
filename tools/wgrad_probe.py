#!/usr/bin/env python3
"""Does the grouped weight-gradient launch wait for its operands' L2 misses (round 4)?  The stack's
group - 20 x (fg 512 x 896 | res 384 x 256) tiles over B x rows - with the same kernel, tile map and instruction stream,
(a) on per-layer buffers as in the model (3.3 GB of distinct operands, PMC: 7.8 GB fetched through the fabric), (b) with
all 20 layers reading ONE layer's buffers (165 MB distinct: the re-fetches of drifting tiles hit in L2 / MALL, nothing
comes from HBM after the first pass).  (a) - (b) is the most any tile order / k-cursor scheme could return.
    python tools/wgrad_probe.py          # on the GPU box
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ae_wavenet_amd import _lib as L
from ae_wavenet_amd.plan import Mat, Plan, Workspace, make_tn, TnGroupBuilder

lib = L.load()
dev = "cuda:0"
B, M, NL = 8, int(os.environ.get("ROWS", "6500")), 20
Rp, Dp, Cp = 384, 256, 128


def layer_rows():
    """rows each layer's matrices contract over per batch element: VARY=1 the model's (7045 ... 5000: the output length
    shrinks by the dilation per layer), else ROWS for every layer"""
    if os.environ.get("VARY", "0") != "1":
        return [M] * NL
    rows, n = [], 7046
    for l in range(NL):
        n -= 1 << (l % 10)
        rows.append(n)
    if os.environ.get("VARY_MEAN", "0") == "1":                 # the same total work, spread evenly
        return [sum(rows) // NL] * NL
    return rows


def build(shared, tile=128, order=None):
    global M
    rows = layer_rows()
    M = max(rows)
    ws = Workspace(dev)
    nbuf = 1 if shared else NL
    x = [Mat.new(ws, f"x{l}", B, M + 600, Rp, L.BF16) for l in range(nbuf)]
    dfg = [Mat.new(ws, f"dfg{l}", B, M, 2 * Dp, L.BF16) for l in range(nbuf)]
    z = [Mat.new(ws, f"z{l}", B, M, Dp, L.BF16) for l in range(nbuf)]
    dx = [Mat.new(ws, f"dx{l}", B, M, Rp, L.BF16) for l in range(nbuf)]
    cond = Mat.new(ws, "cond", B, M + 600, Cp, L.BF16)
    for n, t in ws.bufs.items():
        t.copy_((torch.rand(t.shape, device=dev) * 2 - 1).to(t.dtype))
    gb = TnGroupBuilder(ws, "tng", tile)
    outs = []
    for l in range(NL):
        k = 0 if shared else l
        d = 1 << (l % 10)
        Ml = rows[l]
        t = make_tn(L.BF16, Ml, B, 2 * Dp, 2 * Dp, dfg[k].seg(2 * Dp), [x[k].seg(Rp), x[k].seg(Rp, row_off=d), cond.seg(Cp, row_off=d)])
        o = ws.alloc(f"ofg{l}", 2 * Dp * (2 * Rp + Cp), torch.float32)
        t.out, t.out_batch_stride = o.data_ptr(), 2 * Dp * (2 * Rp + Cp)
        gb.add(t, f"fg{l}")
        t = make_tn(L.BF16, Ml, B, 368, Rp, dx[k].seg(Rp), [z[k].seg(Dp)])
        o = ws.alloc(f"ors{l}", Rp * Dp, torch.float32)
        t.out, t.out_batch_stride = o.data_ptr(), Rp * Dp
        gb.add(t, f"res{l}")
    p = Plan("g")
    gb.emit(p, "group")
    return ws, p, gb


def timeit(p):
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
        best = min(best, e0.elapsed_time(e1))
    return best


def main():
    flops = sum(layer_rows()) * 2.0 * B * (2 * Dp * (2 * Rp + Cp) + Rp * Dp)
    print("rows per layer:", layer_rows())
    for shared in (False, True):
        ws, p, gb = build(shared)
        ms = timeit(p)
        print(f"{'ONE layer buffers shared by all 20 layers' if shared else 'per-layer buffers (as in the model)':44s} "
              f"{ms:7.3f} ms  {flops / ms / 1e9:6.0f} TFLOP/s   distinct operand bytes {ws.nbytes() / 1e9:.2f} GB, "
              f"{sum(1 for r in gb.tile_map() if r >= 0)} tiles")
        del ws, p, gb
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
