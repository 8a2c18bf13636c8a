#!/usr/bin/env python3
"""Round 4: is a launch's fixed cost amortised by a bigger launch, or only by a second queue?  G2 / dz as one B = 8 launch,
two of them back to back, two on two streams, and ONE B = 16 launch.  (ORD=1 with the tile-order probe build
-DAEW_TILE_ORDER_PROBE=1: other orders of the row tiles.)    python tools/b16_probe.py"""
import os, sys
sys.path.insert(0, "/root/repo")
import torch
from ae_wavenet_amd import _lib as L
from ae_wavenet_amd.plan import Mat, Plan, Workspace, make_nt
lib = L.load(); lib.aew_set_nt_rows192(0)
dev="cuda:0"; M=6900; Rp,Dp,Sp=384,256,256; N_REP=6
def plan_of(gs):
    p=Plan("p")
    for _ in range(N_REP):
        for g in gs: p.add(L.OP_GEMM_NT,g,"g",1)
    return p
s0,s1=torch.cuda.Stream(),torch.cuda.Stream()
def timed(fn):
    torch.cuda.synchronize(); best=1e9
    for _ in range(4):
        e0,ea,eb=(torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(s0); s1.wait_event(e0); fn(); ea.record(s0); eb.record(s1); torch.cuda.synchronize()
        best=min(best,max(e0.elapsed_time(ea),e0.elapsed_time(eb))*1e3/N_REP)
    return best
ws=Workspace(dev)
B=16
z=Mat.new(ws,"z",B,M,Dp,L.BF16); x=Mat.new(ws,"x",B,M+64,Rp,L.BF16); xo=Mat.new(ws,"xo",B,M+64,Rp,L.BF16)
dskp=Mat.new(ws,"dskp",B,M,Sp,L.BF16); pf=Mat.new(ws,"pf",B,M,Dp,L.BF16); pg=Mat.new(ws,"pg",B,M,Dp,L.BF16); dfg=Mat.new(ws,"dfg",B,M,2*Dp,L.BF16)
Wrs=Mat.new(ws,"Wrs",1,Rp,Dp,L.BF16); WrsT=Mat.new(ws,"WrsT",1,Dp,Rp+Sp,L.BF16)
for n in ("z","x","dskp","pf","pg","xo"): ws.get(n).copy_((torch.rand(ws.get(n).shape,device=dev)*2-1).bfloat16())
for n in ("Wrs","WrsT"): ws.get(n).copy_(((torch.rand(ws.get(n).shape,device=dev)*2-1)*0.05).bfloat16())
def G2(b0,nb): return make_nt(L.BF16,M,368,Rp,nb,[z.seg(Dp,b0=b0)],Wrs.ptr,flags=L.EF_ADD_AUX0,out0=xo.view(b0=b0),aux0=x.view(row_off=16,b0=b0))
def DZ(b0,nb): return make_nt(L.BF16,M,Dp,Dp,nb,[xo.seg(Rp,hi=M,b0=b0),dskp.seg(Sp,b0=b0)],WrsT.ptr,epi=L.EPI_DFG,aux0=pf.view(b0=b0),aux1=pg.view(b0=b0),out0=dfg.view(b0=b0))
def with_ord(g, o):
    g.reserved = o << 16
    return g
for name,mk in (("G2",G2),("dz",DZ)):
    if os.environ.get("ORD"):
        res=[]
        for o in (0,1,2,3):
            p8,p16=plan_of([with_ord(mk(0,8),o)]),plan_of([with_ord(mk(0,16),o)])
            for p in (p8,p16): p.run(s0.cuda_stream)
            res.append((o,timed(lambda:p8.run(s0.cuda_stream)),timed(lambda:p16.run(s0.cuda_stream))))
        print(name, "tile order -> (B=8, B=16) us:", "  ".join(f"{o}: {a:5.1f} {b:5.1f}" for o,a,b in res))
        continue
    a,b,c=plan_of([mk(0,8)]),plan_of([mk(8,8)]),plan_of([mk(0,16)])
    ab=plan_of([mk(0,8),mk(8,8)])
    for p in (a,b,c,ab): p.run(s0.cuda_stream)
    t1=timed(lambda:a.run(s0.cuda_stream))
    t2=timed(lambda:ab.run(s0.cuda_stream))
    t3=timed(lambda:(a.run(s0.cuda_stream),b.run(s1.cuda_stream)))
    t4=timed(lambda:c.run(s0.cuda_stream))
    print(f"{name}: one B=8 launch {t1:6.1f} us | two B=8 launches back to back {t2:6.1f} | two B=8 launches on two streams {t3:6.1f} | ONE B=16 launch {t4:6.1f}")
