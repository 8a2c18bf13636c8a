"""Launch cost per graph node: a plan of N empty-ish kernels (GEMM that exits at entry, ABL bit 32) replayed as
a hipGraph vs eagerly; plus the op count of the real step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from ae_wavenet_amd import _lib as L
from ae_wavenet_amd.plan import Mat, Plan, Workspace, make_nt

lib = L.load()
dev = "cuda:0"
ws = Workspace(dev)
x = Mat.new(ws, "x", 8, 7046, 384, L.BF16); W = Mat.new(ws, "W", 1, 512, 384, L.BF16)
z = Mat.new(ws, "z", 8, 7046, 256, L.BF16)
bias = ws.alloc("bias", 8 * 512, torch.float32)
st = torch.cuda.current_stream().cuda_stream
for N in (100, 400):
    g = make_nt(L.BF16, 7000, 256, 512, 8, [x.seg(384)], W.ptr, epi=L.EPI_GATED, out0=z.view(), out1=z.view(),
                out2=z.view(), bias_ptr=bias.data_ptr(), bias_bs=512)
    g.reserved = 32                                   # exit at entry
    p = Plan("gap")
    for _ in range(N):
        p.add(L.OP_GEMM_NT, g, "k", 1)
    for mode, fn in (("graph", lambda: p.run_graph(st)), ("eager", lambda: p.run(st))):
        fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        print(f"{N} empty kernels, {mode}: {(time.perf_counter() - t) / 10 / N * 1e6:.2f} us per node")
from test_gpu_parity import seeded_full_engine
hps, eng, *_ = seeded_full_engine(B=8, w=5000, seed=1)
n = {pl.name: len(pl.ops) for pl in (eng.fwd_a, eng.fwd_b, eng.bwd, eng.cb, eng.opt)}
print("ops per step:", n, "total", sum(n.values()))
