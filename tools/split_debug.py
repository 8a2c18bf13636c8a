#!/usr/bin/env python3
"""Which launch of the split-K hint (TrainEngine.small_split) moves the gradients, and by how much: one engine per hinted
launch with the hint left on that launch only, against the unsplit engine on the same inputs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from ae_wavenet_amd import model as M  # noqa: E402
from tests.test_gpu_parity import DEV, seeded_full_engine  # noqa: E402


def run(e, inp):
    e.init_ema_from_emb()
    e.set_inputs(*[t.to(DEV) for t in inp])
    loss = float(e.forward())
    e.backward()
    torch.cuda.synchronize()
    return loss, e.enc.dy[0].tensor().clone(), {k: e.ps.view(k, grad=True).clone() for k in e.ps.names()}


def main():
    B, w = 8, 5000
    _, e0, _, _, inp = seeded_full_engine(B=B, w=w)
    l0, m0, g0 = run(e0, inp)
    l0b, m0b, g0b = run(e0, inp)
    print("unsplit replay: mel equal", torch.equal(m0, m0b))
    M.TrainEngine.small_split = int(os.environ.get("TARGET", "256"))
    _, e1, _, _, _ = seeded_full_engine(B=B, w=w)
    labels = [lab for lab, _ in e1.small_split_made]
    del e1
    for only in labels + ["ALL"]:
        _, e, _, _, _ = seeded_full_engine(B=B, w=w)
        for pl in (e.fwd_a, e.fwd_b, e.bwd):
            for op, lab in zip(pl.ops, pl.labels):
                if op.kind == M.L.OP_GEMM_NT and op.u.nt.dtype == M.L.BF16 and op.u.nt.k_split > 1 and only not in ("ALL", lab):
                    op.u.nt.k_split = 0
        l1, m1, g1 = run(e, inp)
        rel = lambda a, b: (a - b).norm().item() / max(b.norm().item(), 1e-30)
        rows = sorted(((rel(g1[k], g0[k]), k) for k in g0 if g0[k].abs().max() > 0), reverse=True)[:3]
        print(f"{only:12s} S={dict(e.small_split_made).get(only)}  loss dev {abs(l1 / l0 - 1):.1e}  mel {rel(m1, m0):.2e}  "
              + "  ".join(f"{k}:{r:.1e}" for r, k in rows))
        del e


if __name__ == "__main__":
    main()
