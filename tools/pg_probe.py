import os, sys
sys.path.insert(0, "/root/repo")
import torch
from ae_wavenet_amd import _lib as L
from ae_wavenet_amd.plan import Mat, Plan, Workspace, make_nt
lib = L.load()
dev="cuda:0"; B=8; Rp,Dp,Cp=384,256,128
def timeit(g,n=10):
    p=Plan("p")
    for _ in range(n): p.add(L.OP_GEMM_NT,g,"g",1)
    st=torch.cuda.current_stream().cuda_stream; p.run(st); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); best=1e9
    for _ in range(4):
        e0.record(); p.run(st); e1.record(); torch.cuda.synchronize(); best=min(best,e0.elapsed_time(e1)*1e3/n)
    return best
for M,d in ((6900,16),(6900,256),(6000,16)):
    ws=Workspace(dev)
    x=Mat.new(ws,"x",B,M+600,Rp,L.BF16); cond=Mat.new(ws,"cond",B,M+600,Cp,L.BF16)
    z=Mat.new(ws,"z",B,M,Dp,L.BF16); pf=Mat.new(ws,"pf",B,M,Dp,L.BF16); pg=Mat.new(ws,"pg",B,M,Dp,L.BF16)
    W=Mat.new(ws,"W",1,2*Dp,2*Rp+Cp,L.BF16); bias=ws.alloc("bias",B*2*Dp,torch.float32)
    for n in ("x","cond"): ws.get(n).copy_((torch.rand(ws.get(n).shape,device=dev)*2-1).bfloat16())
    ws.get("W").copy_(((torch.rand(ws.get("W").shape,device=dev)*2-1)*0.05).bfloat16())
    def G1(o1=True,o2=True):
        return make_nt(L.BF16,M,Dp,2*Dp,B,[x.seg(Rp),x.seg(Rp,row_off=d),cond.seg(Cp,row_off=d)],W.ptr,epi=L.EPI_GATED,
                       out0=z.view(),out1=pf.view() if o1 else None,out2=pg.view() if o2 else None,bias_ptr=bias.data_ptr(),bias_bs=2*Dp)
    print(f"rows {M} d {d}: G1 z+pf+pg {timeit(G1()):6.1f} us | z+pf {timeit(G1(True,False)):6.1f} | z only {timeit(G1(False,False)):6.1f}")
    del ws
