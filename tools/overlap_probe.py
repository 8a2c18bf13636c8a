#!/usr/bin/env python3
"""Can a K loop and an epilogue share a CU (round 4)?  The thin bf16 NT kernel cut in two with the ablation switches of
the tools library - "no epilogue" (K loop only) and "barriers + epilogue" (no LDS-DMA, no fragment reads, no MFMA) - run
alone, back to back, and CONCURRENTLY on two streams (each kernel one block per CU slot: 72 KiB LDS, 128 VGPRs, so a pair
is co-resident on every CU).  If the pair takes max(K, E) the two phases use different resources and an in-kernel pipeline
(epilogue of tile i under the K loop of tile i + 1) can return up to sum - max per launch; if it takes K + E they queue
on the same path and no schedule will help.
    AEW_LIB_PATH=ae-wavenet_amd/lib/libaewavenet_hip_abl.so python tools/overlap_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ae_wavenet_amd import _lib as L
from ae_wavenet_amd.plan import Mat, Plan, Workspace, make_nt

lib = L.load()
lib.aew_set_nt_window(int(os.environ.get("WINDOW", "0")))      # 64: G1 / dx on the one-window kernel (product library)
lib.aew_set_nt_rows192(int(os.environ.get("ROWS192", "0")))
dev = "cuda:0"
B, M = 8, int(os.environ.get("ROWS", "6900"))
Rp, Dp, Sp, Cp = 384, 256, 256, 128
N_REP = 6


def plan_of(g, n=N_REP):
    p = Plan("ov")
    for _ in range(n):
        p.add(L.OP_GEMM_NT, g, "g", 1)
    return p


def main():
    ws = Workspace(dev)
    mats = {}
    for nm, rows, pitch in (("z", M, Dp), ("x", M + 64, Rp), ("cond", M + 64, Cp), ("xo", M + 64, Rp), ("dskp", M, Sp),
                            ("pf", M, Dp), ("pg", M, Dp), ("dfg", M, 2 * Dp),
                            ("z2", M, Dp), ("xo2", M + 64, Rp), ("pf2", M, Dp), ("pg2", M, Dp), ("dfg2", M, 2 * Dp)):
        mats[nm] = Mat.new(ws, nm, B, rows, pitch, L.BF16)
        ws.get(nm).copy_((torch.rand(ws.get(nm).shape, device=dev) * 2 - 1).bfloat16())
    Ws = {}
    for nm, r, c in (("Wrs", Rp, Dp), ("Wfg", 2 * Dp, 2 * Rp + Cp), ("WfgT", Rp, 4 * Dp), ("WrsT", Dp, Rp + Sp)):
        Ws[nm] = Mat.new(ws, nm, 1, r, c, L.BF16)
        ws.get(nm).copy_(((torch.rand(ws.get(nm).shape, device=dev) * 2 - 1) * 0.05).bfloat16())
    bias = ws.alloc("bias", B * 2 * Dp, torch.float32)
    m = mats
    d = int(os.environ.get("DIL", "128"))

    def G1(o):   # o: output set (the concurrent pair writes different buffers)
        return make_nt(L.BF16, M, Dp, 2 * Dp, B, [m["x"].seg(Rp), m["x"].seg(Rp, row_off=d), m["cond"].seg(Cp, row_off=d)],
                       Ws["Wfg"].ptr, epi=L.EPI_GATED, out0=m["z" + o].view(), out1=m["pf" + o].view(), out2=m["pg" + o].view(),
                       bias_ptr=bias.data_ptr(), bias_bs=2 * Dp)

    def G2(o):
        return make_nt(L.BF16, M, 368, Rp, B, [m["z"].seg(Dp)], Ws["Wrs"].ptr, flags=L.EF_ADD_AUX0, out0=m["xo" + o].view(),
                       aux0=m["x"].view(row_off=16))

    def DZ(o):
        return make_nt(L.BF16, M, Dp, Dp, B, [m["xo"].seg(Rp), m["dskp"].seg(Sp)], Ws["WrsT"].ptr, epi=L.EPI_DFG,
                       aux0=m["pf"].view(), aux1=m["pg"].view(), out0=m["dfg" + o].view())

    def DX(o):
        return make_nt(L.BF16, M, 368, Rp, B, [m["dfg"].seg(2 * Dp), m["dfg"].seg(2 * Dp, row_off=-d)], Ws["WfgT"].ptr,
                       flags=L.EF_ADD_AUX0, out0=m["xo" + o].view(), aux0=m["x"].view(row_off=16))
    s0 = torch.cuda.Stream()
    s1 = torch.cuda.Stream()

    def timed(fn):
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(4):
            e0, e1a, e1b = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record(s0)
            s1.wait_event(e0)
            fn()
            e1a.record(s0)
            e1b.record(s1)
            torch.cuda.synchronize()
            best = min(best, max(e0.elapsed_time(e1a), e0.elapsed_time(e1b)) * 1e3 / N_REP)
        return best
    for name, mk in (("G1", G1), ("G2", G2), ("dz", DZ), ("dx", DX)):
        gk, ge, gf, gf2 = mk(""), mk("2"), mk(""), mk("2")
        gk.reserved, ge.reserved, gf.reserved, gf2.reserved = 8, 1 | 2 | 4, 2048, 2048
        pk, pe, pf_, pf2 = plan_of(gk), plan_of(ge), plan_of(gf), plan_of(gf2)
        for p in (pk, pe, pf_, pf2):
            p.run(s0.cuda_stream)
        tk = timed(lambda: pk.run(s0.cuda_stream))
        te = timed(lambda: pe.run(s0.cuda_stream))
        tf = timed(lambda: pf_.run(s0.cuda_stream))
        tser = timed(lambda: (pk.run(s0.cuda_stream), pe.run(s0.cuda_stream)))
        tcon = timed(lambda: (pk.run(s0.cuda_stream), pe.run(s1.cuda_stream)))
        tff = timed(lambda: (pf_.run(s0.cuda_stream), pf2.run(s1.cuda_stream)))
        # the same launch as two half-batch launches (batch elements 0-3 | 4-7), back to back and concurrently
        ha, hb = mk(""), mk("")
        ha.reserved = hb.reserved = 2048
        ha.batch = hb.batch = B // 2
        for fld in ("out0", "out1", "out2", "aux0", "aux1"):
            v = getattr(hb, fld)
            if v.ptr:
                v.ptr += (B // 2) * v.batch_stride * 2
        for i in range(hb.n_segs):
            hb.seg[i].ptr += (B // 2) * hb.seg[i].batch_stride * 2
        if hb.bias:
            hb.bias = hb.bias + 4 * (B // 2) * hb.bias_bs
        pa, pb = plan_of(ha), plan_of(hb)
        for p in (pa, pb):
            p.run(s0.cuda_stream)
        th_ser = timed(lambda: (pa.run(s0.cuda_stream), pb.run(s0.cuda_stream)))
        th_con = timed(lambda: (pa.run(s0.cuda_stream), pb.run(s1.cuda_stream)))
        print(f"    halves (B = {B // 2} each): back to back {th_ser:6.1f}, concurrently {th_con:6.1f}")
        print(f"{name}: full {tf:6.1f} us | K only {tk:6.1f}  E only {te:6.1f}  sum {tk + te:6.1f} | back to back {tser:6.1f} | "
              f"concurrent K || E {tcon:6.1f} | two full launches concurrently {tff:6.1f} (2 x full = {2 * tf:6.1f})")


if __name__ == "__main__":
    main()
