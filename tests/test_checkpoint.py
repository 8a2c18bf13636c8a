"""Checkpoint interchange with the reference.

tests/golden/reference_format.ckpt was written by tests/golden/make_golden.py with the REFERENCE's own
classes: the dict of checkpoint.py:87-98 with a reference MfccInverter's state_dict, the state of a
torch.optim.Adam after two steps and hps as a pickled hparams.Hyperparams.  reference_ckpt_next_step.npz
holds the gradients of a third step and the parameters the reference's Adam produced from them."""
import os

import numpy as np
import pytest
import torch

from ae_wavenet_amd import checkpoint, config, mfcc_inverter as mi, optim

HERE = os.path.dirname(os.path.abspath(__file__))
CKPT = os.path.join(HERE, "golden", "reference_format.ckpt")
DEV = "cuda:0"


def _model(ckpt):
    hps = config.from_checkpoint_hps(ckpt["hps"])
    return hps, mi.MfccInverter(hps)


def test_reads_a_file_written_by_the_reference_classes():
    with pytest.raises(ModuleNotFoundError):
        torch.load(CKPT, weights_only=False)              # plain torch.load needs the reference's `hparams`
    ck = checkpoint.load(CKPT)
    assert isinstance(ck["hps"], config.Hyperparams) and ck["hps"].n_res == ck["hps"]["n_res"]
    assert (ck["epoch"], ck["step"], ck["optim_step"]) == (3, 1234, 2)
    hps, m = _model(ck)
    opt = optim.FusedAdam(m)
    pos = checkpoint.restore(m, opt, ck)
    assert pos["optim_step"] == 2 and pos["hps"].n_win_batch == hps.n_win_batch
    # every reference parameter landed, bit for bit (the `_lead` / `left_wing_size` buffers are filtered)
    sd = m.state_dict()
    ref = checkpoint.filtered_state(ck)
    own = dict(m.named_parameters())
    assert set(own) <= set(ref)
    for k in own:
        assert torch.equal(sd[k], ref[k]), k
    # Adam moments and step came through in torch.optim.Adam's own layout
    o = opt.state_dict()
    names = [n for n, _ in m.named_parameters()]
    assert o["param_groups"][0]["params"] == list(range(len(names)))
    assert o["param_groups"][0]["lr"] == ck["optim"]["param_groups"][0]["lr"]
    for i, n in enumerate(names):
        assert float(o["state"][i]["step"]) == 2.0
        assert torch.equal(o["state"][i]["exp_avg"], ck["optim"]["state"][i]["exp_avg"]), n
        assert torch.equal(o["state"][i]["exp_avg_sq"], ck["optim"]["state"][i]["exp_avg_sq"]), n


def test_writes_a_file_the_reference_loader_accepts(tmp_path):
    ck = checkpoint.load(CKPT)
    hps, m = _model(ck)
    opt = optim.FusedAdam(m)
    checkpoint.restore(m, opt, ck)
    out = tmp_path / "roundtrip.ckpt"
    checkpoint.save(str(out), m, opt, hps, epoch=4, step=99, optim_step=7)
    # what Checkpoint.__init__ does (checkpoint.py:25-28, 52-63), with stock torch only
    raw = torch.load(str(out), weights_only=False)
    assert type(raw["hps"]) is dict and raw["hps"]["n_res"] == hps.n_res          # Hyperparams(**ckpt['hps']) works
    assert (raw["epoch"], raw["step"], raw["optim_step"]) == (4, 99, 7)
    params = [torch.nn.Parameter(torch.empty_like(p)) for p in m.parameters()]
    stock = torch.optim.Adam(params)
    stock.load_state_dict(raw["optim"])                       # the reference's restore path
    for i, p in enumerate(params):
        st = stock.state[p]
        assert float(st["step"]) == 2.0
        assert torch.equal(st["exp_avg"], ck["optim"]["state"][i]["exp_avg"])
    for k, v in checkpoint.filtered_state(ck).items():
        if k in raw["model_state_dict"]:
            assert torch.equal(raw["model_state_dict"][k], v), k


@pytest.mark.gpu
def test_restored_adam_continues_like_the_reference():
    """Third Adam step from the restored state == the step the reference's torch.optim.Adam took."""
    ck = checkpoint.load(CKPT)
    nxt = np.load(os.path.join(HERE, "golden", "reference_ckpt_next_step.npz"))
    hps, m = _model(ck)
    m = m.to(DEV)
    opt = optim.FusedAdam(m, lr=1.0)                          # lr must come from the checkpoint
    checkpoint.restore(m, opt, ck)
    g = m.geom
    B = 2
    gen = torch.Generator().manual_seed(3)
    wav = torch.randint(0, hps.n_quant, (B, g.enc_in_len), generator=gen).float().to(DEV)
    mel = torch.randn(B, hps.n_lc_in, g.mel_len, generator=gen).to(DEV)
    voice = torch.randint(0, hps.n_speakers, (B,), generator=gen).to(DEV)
    jitter = torch.arange(g.embed_len).repeat(B, 1).to(DEV)
    pred, target, loss = m.run(wav, mel, voice, jitter)       # builds the engine (weights re-homed)
    loss.backward()
    eng = m._engine
    for n, _ in m.named_parameters():                         # the gradients the reference used
        eng.ps.view(n, grad=True).copy_(torch.from_numpy(nxt["grad." + n]).to(DEV))
    opt.step()
    torch.cuda.synchronize()
    assert eng.step_count == 3
    for n, _ in m.named_parameters():
        got = eng.ps.view(n).cpu().numpy()
        want = nxt["after." + n]
        assert np.allclose(got, want, rtol=2e-6, atol=2e-7), (n, np.abs(got - want).max())
