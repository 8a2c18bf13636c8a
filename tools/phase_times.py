"""Wall time of each plan (graph replay) with lanes on / off, B=8 w=5000."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_gpu_parity import seeded_full_engine, DEV
from ae_wavenet_amd import _lib as L

lib = L.load()
hps, eng, wts, emb, inp = seeded_full_engine(B=8, w=5000, seed=11)
eng.set_inputs(*[t.to(DEV) for t in inp])
st = torch.cuda.current_stream().cuda_stream
for lanes in (1, 0):
    lib.aew_set_lanes(lanes)
    for pl in (eng.fwd_a, eng.fwd_b, eng.bwd):
        pl.invalidate_graph()
    eng.forward(); eng.backward(); torch.cuda.synchronize()
    out = []
    for pl in (eng.fwd_a, eng.fwd_b, eng.bwd, eng.opt):
        for _ in range(3):
            pl.run_graph(st)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            pl.run_graph(st)
        torch.cuda.synchronize()
        out.append(f"{pl.name} {(time.perf_counter() - t) / 20 * 1e3:.3f}")
    print(f"lanes={lanes}: " + "  ".join(out))
