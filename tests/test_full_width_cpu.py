"""Full-width launch plans against the fp32 oracle, on the CPU (no GPU needed).

The GPU tests hold the full-width training step to the oracle only within bf16 effects (15 % / 20 % max-normalised per
gradient: ReLU-mask flips), and the tight buffer-by-buffer check compares the GPU with the interpreter of the SAME
descriptors - an error that exists only in the full-width descriptors (channel pads 368 -> 384, K packing 874 -> 896,
lead offsets, the grouped-wgrad tile map) would be common to both.  This test closes that chain: the plan interpreter
(tests/plan_emulator.py) executes the exact full-width `aew_op_t` records with fp32 storage ("wide": no bf16 rounding,
hence no mask flips) and has to reproduce `oracle.ref_model.ae_run` - itself pinned to the reference's goldens
(tests/test_oracle_vs_golden.py) - to 2e-4 in the logits, the loss and EVERY parameter gradient.

  V     par/arch.vqvae-ema.json shape (2 x 10 layers, 368 / 256 / 256), B = 1, w = 100 (the small parity size of SURVEY 8)
  DEEP  BASELINE configs[4] architecture (3 x 10 layers, 512 residual channels), short window
"""
import os
import sys

import numpy as np
import pytest
import torch

from ae_wavenet_amd import _lib as L, config, model as M, plan as PL
from tests.plan_emulator import Emu

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from weights import np_weights  # noqa: E402


def _step_vs_oracle(hps, seed, n_embed, tol=2e-4):
    from oracle import ref_model as R
    B = 1
    eng = M.TrainEngine(hps, B=B, device="cpu", n_mel=39, update_codebook_every_step=False)
    shapes = {k: eng.ps.shape[k] for k in eng.ps.names()}
    wts = np_weights(shapes, seed)
    for k, v in wts.items():
        eng.ps.view(k).copy_(torch.from_numpy(v))
    rs = np.random.RandomState(seed + 1)
    emb = (rs.standard_normal((n_embed, hps.bn_n_out)) * 0.7).astype(np.float32)
    eng.emb.copy_(torch.from_numpy(emb))
    eng.init_ema_from_emb()
    g = eng.geom
    inp = (torch.from_numpy(rs.randint(0, 256, (B, g.enc_in_len)).astype(np.float32)),
           torch.from_numpy(rs.standard_normal((B, 39, g.mel_len)).astype(np.float32)),
           torch.from_numpy(rs.randint(0, 40, (B,)).astype(np.int64)), torch.arange(g.embed_len).repeat(B, 1))
    eng.set_inputs(*inp)
    emu = Emu(eng.ws)
    for plan in (eng.fwd_a, eng.fwd_b, eng.bwd):
        emu.run(plan)
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in wts.items()}
    out = R.ae_run(sd, {"emb": torch.from_numpy(emb)}, hps, g, *inp, loss_mode="intended", take_compat=False)
    out["loss"].backward()
    assert np.array_equal(eng.ind[:eng.Q].numpy(), out["min_ind"].reshape(-1).numpy()), "code indices"
    ref_lg = out["quant"].detach()
    lg = eng.logits().permute(0, 2, 1)
    assert (lg - ref_lg).abs().max().item() <= tol * ref_lg.abs().max().item()
    assert abs(float(eng.loss_buf[0]) / float(out["loss"].detach()) - 1) < tol
    worst = (0.0, "")
    n_checked = 0
    for k in eng.ps.names():
        ref = sd[k].grad
        if ref is None or ref.abs().max().item() == 0:
            continue
        got = eng.ps.view(k, grad=True)
        e = (got - ref).abs().max().item() / ref.abs().max().item()
        worst = max(worst, (e, k))
        n_checked += 1
        assert e < tol, (k, e)
    assert n_checked > 150                                   # every weight / bias of encoder, bottleneck and decoder
    return worst


@pytest.fixture
def wide(monkeypatch):
    monkeypatch.setitem(PL.TORCH_DT, L.BF16, torch.float32)
    monkeypatch.setitem(PL.ESIZE, L.BF16, 4)


def test_full_width_plan_vs_oracle_cpu(wide):
    hps = config.make_hps("vqvae-ema", n_win_batch=100, bn_vq_n_embed=4096)
    assert (hps.n_res, hps.n_dil, hps.n_skp) == (368, 256, 256)
    worst = _step_vs_oracle(hps, seed=3, n_embed=4096)
    print("V full width, wide interpreter vs oracle: worst gradient", worst)


def test_deep_plan_vs_oracle_cpu(wide):
    hps = config.make_hps("deep", n_win_batch=64, bn_vq_n_embed=512)
    assert hps.n_res == 512
    worst = _step_vs_oracle(hps, seed=21, n_embed=512)
    print("DEEP (30 x 512), wide interpreter vs oracle: worst gradient", worst)
