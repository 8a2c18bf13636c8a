"""Do lane-1 ops really overlap lane-0 ops?  Two single-block reductions (k_reduce, ~ms each) on
different lanes: replayed as a hipGraph and eagerly, against the serial time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ae_wavenet_amd import _lib as L
from ae_wavenet_amd.plan import Plan, Workspace

lib = L.load()
dev = "cuda:0"
ws = Workspace(dev)
n = 8_000_000
x = ws.alloc("x", n, torch.float32)
o = ws.alloc("o", 64, torch.float32)


def red(i):
    r = L.Reduce()
    r.n_terms = 1
    r.x[0], r.n[0], r.scale[0], r.post_scale[0] = x.data_ptr(), n, 1.0, 1.0
    r.out = o.data_ptr() + 32 * i
    return r


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


st = torch.cuda.current_stream().cuda_stream
for lanes in (0, 1):
    lib.aew_set_lanes(lanes)
    p = Plan("t")
    p.add(L.OP_REDUCE, red(0), "a")
    with p.side():
        p.add(L.OP_REDUCE, red(1), "b")
    p.add(L.OP_REDUCE, red(2), "c")
    with p.side():
        p.add(L.OP_REDUCE, red(3), "d")
    print(f"lanes={lanes}: graph {timed(lambda: p.run_graph(st)):.3f} ms   eager {timed(lambda: p.run(st)):.3f} ms  (4 single-block kernels)")

# pattern 2: the side op comes FIRST (like the decoder weight pack at the head of fwd_a)
for lanes in (0, 1):
    lib.aew_set_lanes(lanes)
    p = Plan("t2")
    with p.side():
        p.add(L.OP_REDUCE, red(0), "a")
    p.add(L.OP_REDUCE, red(1), "b")
    p.add(L.OP_REDUCE, red(2), "c")
    print(f"side-first lanes={lanes}: graph {timed(lambda: p.run_graph(st)):.3f} ms   eager {timed(lambda: p.run(st)):.3f} ms  (3 kernels; overlap -> 2x single)")
