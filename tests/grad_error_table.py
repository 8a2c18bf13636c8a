"""Per-tensor gradient error of the full-width step against the fp32 oracle (same setup as
tests/test_gpu_parity.py::test_full_width_step_vs_oracle): max-normalised error, cosine, and where the worst
element sits.  A diagnostic script (python tests/grad_error_table.py on the GPU box), kept under tests/ because it
uses the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_gpu_parity import seeded_full_engine, DEV
from oracle import ref_model as R

B, w = int(os.environ.get("B", "2")), int(os.environ.get("W", "100"))
hps, eng, wts, emb, inp = seeded_full_engine(B=B, w=w)
eng.set_inputs(*[t.to(DEV) for t in inp])
loss = eng.forward(); eng.backward(); torch.cuda.synchronize()
sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in wts.items()}
out = R.ae_run(sd, {"emb": torch.from_numpy(emb)}, hps, eng.geom, *inp, loss_mode="intended", take_compat=False)
out["loss"].backward()
rows = []
for k in eng.ps.names():
    ref = sd[k].grad
    if ref is None or ref.abs().max() == 0:
        continue
    got = eng.ps.view(k, grad=True).cpu()
    d = (got - ref).abs()
    e = d.max().item() / ref.abs().max().item()
    med = d.median().item() / ref.abs().max().item()
    cos = torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0).item()
    idx = np.unravel_index(int(d.argmax()), tuple(ref.shape))
    rows.append((e, cos, med, k, idx, tuple(ref.shape)))
rows.sort(reverse=True)
for e, cos, med, k, idx, shp in rows[:14]:
    print(f"{e:7.4f}  cos {cos:.5f}  median {med:.2e}  {k:48s} worst at {idx} of {shp}")
es = np.array([r[0] for r in rows])
print(f"{len(rows)} tensors: median of max-normalised error {np.median(es):.4f}, 90th pct {np.percentile(es, 90):.4f}")
