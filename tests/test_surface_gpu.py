"""Module-surface behaviour on the GPU (through the C ABI): things the reference harness relies on implicitly.

* an unclipped jitter row (the reference's Jitter emits n at position n-1, jitter.py:29-33; a row made for the mel
  frames is longer than the encoder output) is clamped, never read / scattered out of bounds;
* autograd's upstream gradient reaches every .grad ((loss * k).backward());
* Adam moments / step survive model.to(), a change of batch size and sample() (checkpoint.py:82-102 saves after .to);
* init_codebook's sample collection and the sampler's conditioning pass leave the code histogram alone
  (autoencoder_model.py:171-199 runs encoder + bottleneck.linear only);
* the sampler re-packs its weights after FusedAdam changed them.
"""
import numpy as np
import pytest
import torch

from ae_wavenet_amd import config

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _tiny(bn="vqvae-ema", every_step=True, **kw):
    from ae_wavenet_amd import autoencoder_model as ae
    args = dict(n_res=64, n_dil=32, n_skp=32, n_post=32, n_lc_out=16, enc_n_out=64, bn_n_out=72 if bn != "vqvae-ema" else 8,
                bn_vq_n_embed=64, n_win_batch=96, n_blocks=2, n_block_layers=3, n_global_embed=4, n_speakers=5)
    args.update(kw)
    hps = config.make_hps(bn, **args)
    torch.manual_seed(11)
    m = ae.AutoEncoder(hps, n_mel=39, update_codebook_every_step=every_step).to(DEV)
    return hps, m


def _batch(m, B, seed=1, jitter=None):
    g = m.geom
    gen = torch.Generator().manual_seed(seed)
    wav = torch.randint(0, 256, (B, g.enc_in_len), generator=gen).float().to(DEV)
    mel = torch.randn(B, 39, g.mel_len, generator=gen).to(DEV)
    voice = torch.randint(0, 5, (B,), generator=gen).to(DEV)
    if jitter is None:
        jitter = torch.arange(g.embed_len).repeat(B, 1)
    return wav, mel, voice, jitter.to(DEV)


def test_unclipped_jitter_is_clamped():
    # d = 72 > 64 channels: an unclamped index N would leave the 64-element allocation slack
    hps, m = _tiny("ae")
    g = m.geom
    B, Ne, Nm = 3, g.embed_len, g.mel_len
    rs = np.random.RandomState(0)
    raw = np.arange(Nm)[None, :] - 1 + rs.randint(0, 3, size=(B, Nm))     # t - 1 + x, x in {0, 1, 2}: up to Nm at the end
    raw[:, 0], raw[:, 1] = 0, 1
    raw[:, Ne - 1] = Ne                                                    # the case the reference can emit
    raw[0, 2] = -1
    wav, mel, voice, jit = _batch(m, B, jitter=torch.from_numpy(raw))
    pred, _, loss = m.run(wav, mel, voice, jit)
    loss.backward()
    torch.cuda.synchronize()
    got = pred.clone(), m._engine.ps.grads[:m._engine.ps.numel].clone(), m._engine.dec.dlc_src.tensor().clone()
    assert torch.isfinite(got[0]).all() and torch.isfinite(got[1]).all()
    clipped = torch.from_numpy(np.clip(raw[:, :Ne], 0, Ne - 1)).to(DEV)
    m.zero_grad()
    pred2, _, loss2 = m.run(wav, mel, voice, clipped)
    loss2.backward()
    torch.cuda.synchronize()
    assert torch.equal(got[0], pred2)
    assert torch.allclose(got[2], m._engine.dec.dlc_src.tensor(), rtol=1e-5, atol=1e-7)      # fp32 atomics: order only
    assert torch.allclose(got[1], m._engine.ps.grads[:m._engine.ps.numel], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("bn", ["vqvae-ema", "vae", "ae", "vqvae"])
def test_upstream_gradient_reaches_every_grad(bn):
    d = 16 if bn == "vae" else 8                               # != embed_len (8): eps layout is unambiguous
    hps, m = _tiny(bn, every_step=False, bn_n_out=d)          # frozen codebook: the three runs see the same forward
    if bn == "vae":
        m.objective.update_anneal_weight(0.3)
    batch = _batch(m, 2)
    eps = torch.randn(2, m.geom.embed_len, d, generator=torch.Generator().manual_seed(5)).to(DEV) if bn == "vae" else None
    _, _, loss = m.run(*batch, eps=eps)
    loss.backward()
    torch.cuda.synchronize()
    n = m._engine.ps.numel
    g1 = m._engine.ps.grads[:n].clone()
    assert float(g1.abs().max()) > 0
    m.zero_grad()
    _, _, loss = m.run(*batch, eps=eps)
    (loss * 4.0).backward()                                    # a power of two: bf16 / fp32 roundings scale exactly
    torch.cuda.synchronize()
    g4 = m._engine.ps.grads[:n].clone()
    assert torch.allclose(g4, 4.0 * g1, rtol=2e-4, atol=1e-7 * float(g1.abs().max()))
    m.zero_grad()
    _, _, loss = m.run(*batch, eps=eps)
    loss.backward()                                            # and back: the scalar is per call, not sticky
    torch.cuda.synchronize()
    assert torch.allclose(m._engine.ps.grads[:n], g1, rtol=2e-4, atol=1e-7 * float(g1.abs().max()))
    # no zero_grad(): a further backward ADDS to .grad (nn.Module semantics); a parameter whose .grad was set to None
    # alone starts from zero
    first = next(iter(m.parameters()))
    first.grad = None
    _, _, loss = m.run(*batch, eps=eps)
    (loss * 2.0).backward()
    torch.cuda.synchronize()
    g3 = m._engine.ps.grads[:n]
    k = first.numel()
    o = (m._engine.ps.view(m._pnames[0][0], grad=True).data_ptr() - m._engine.ps.grads.data_ptr()) // 4
    want = 3.0 * g1
    want[o:o + k] = 2.0 * g1[o:o + k]
    assert first.grad is not None and torch.allclose(g3, want, rtol=3e-4, atol=2e-7 * float(g1.abs().max()))
    # FusedAdam.zero_grad(set_to_none=False): .grad reads as zeros at once (torch's contract), and the next backward starts
    # from them; the default (set_to_none=True) detaches the views instead
    from ae_wavenet_amd import optim
    opt = optim.FusedAdam(m, lr=1e-4)
    opt.zero_grad(set_to_none=False)
    torch.cuda.synchronize()
    assert all(p.grad is not None and float(p.grad.abs().max()) == 0.0 for p in m.parameters())
    _, _, loss = m.run(*batch, eps=eps)
    loss.backward()
    torch.cuda.synchronize()
    assert torch.allclose(m._engine.ps.grads[:n], g1, rtol=2e-4, atol=1e-7 * float(g1.abs().max()))
    opt.zero_grad()
    assert all(p.grad is None for p in m.parameters())


@pytest.mark.parametrize("kind", ["vqvae-ema", "vae", "mfcc"])
def test_step_metrics_come_from_the_plan(kind):
    """rec / tprb_m / com / mel_grad_sd / bn_grad_sd / mel_grad_mean / enc_az are reduced by ops of the step's own
    plans (side lanes); here they are held to the same reductions done with torch on the buffers they summarise."""
    if kind == "mfcc":
        from ae_wavenet_amd import mfcc_inverter as mi
        hps = config.make_hps("mi", n_res=64, n_dil=32, n_skp=32, n_post=32, n_lc_out=16, n_win_batch=96, n_blocks=2,
                              n_block_layers=3, n_global_embed=4, n_speakers=5)
        torch.manual_seed(3)
        m = mi.MfccInverter(hps).to(DEV)
        g = m.geom
        gen = torch.Generator().manual_seed(2)
        n_mel = hps.n_lc_in
        batch = (torch.randint(0, 256, (2, g.enc_in_len), generator=gen).float().to(DEV),
                 torch.randn(2, n_mel, g.mel_len, generator=gen).to(DEV),
                 torch.randint(0, 5, (2,), generator=gen).to(DEV), torch.arange(g.embed_len).repeat(2, 1).to(DEV))
    else:
        hps, m = _tiny(kind, bn_n_out=16 if kind == "vae" else 8)
        batch = _batch(m, 2)
    _, _, loss = m.run(*batch)
    loss.backward()
    torch.cuda.synchronize()
    eng, mt = m._engine, m.objective.metrics
    B, w = eng.B, eng.n_win
    n_pos = B * (w - 1)

    def close(a, b, tol=2e-5):
        a, b = float(a), float(b)
        assert abs(a - b) <= tol * max(1.0, abs(b)), (a, b)
    close(mt["rec"], eng.dec.nll[:B * w].double().sum() / n_pos)
    close(m.tprb_m, eng.dec.ptgt[:B * w].double().sum() / n_pos)
    if kind == "vqvae-ema":
        close(mt["com"], (eng.min_dist[:eng.Q].double() * hps.bn_vq_gamma).mean())
    if kind == "mfcc":
        mg = eng.dec.dlc_src.tensor()[:, :, :eng.n_mel].double()
        close(mt["mel_grad_sd"], mg.std())
        close(mt["mel_grad_mean"], mg.mean(), 1e-6)
    else:
        close(mt["mel_grad_sd"], eng.enc.dy[0].tensor()[:, :, :eng.n_mel].double().std())
        close(mt["bn_grad_sd"], eng.dec.dlc_src.tensor()[:, :, :hps.bn_n_out].double().std())
        for i in range(9):
            numel = B * eng.geom.enc_lens[i + 1] * hps.enc_n_out
            close(m.encoder.metrics[f"enc_az_{i}"], float(eng.enc.zero_cnt[i]) / numel, 1e-9)
    assert float(mt["mel_grad_sd"]) > 0


def test_adam_state_survives_engine_rebuilds():
    from ae_wavenet_amd import optim
    hps, m = _tiny()
    opt = optim.FusedAdam(m, lr=1e-3)
    batch = _batch(m, 2)
    for _ in range(2):
        opt.zero_grad()
        _, _, loss = m.run(*batch)
        loss.backward()
        opt.step()
    ref = opt.state_dict()["state"]
    assert len(ref) == len(list(m.parameters())) and float(ref[0]["step"]) == 2.0
    w_ref = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}

    def same(state):
        assert len(state) == len(ref)
        for i in ref:
            assert float(state[i]["step"]) == 2.0
            assert torch.equal(state[i]["exp_avg"], ref[i]["exp_avg"]) and torch.equal(state[i]["exp_avg_sq"], ref[i]["exp_avg_sq"]), i

    # the reference's save path: model.to(cpu), then optim.state_dict()  (checkpoint.py:82-98)
    m.to("cpu")
    same(opt.state_dict()["state"])
    m.to(DEV)
    same(opt.state_dict()["state"])
    # eval-mode sampling (B = 1 engine) between training steps
    m.eval()
    m.set_n_replicas(2)
    out = m(*[t[:1] for t in batch])
    assert out.shape[0] == 3
    m.train()
    same(opt.state_dict()["state"])
    # a different training batch size
    b3 = _batch(m, 3, seed=2)
    m._ensure_engine(3)
    same(opt.state_dict()["state"])
    for k, v in m.state_dict().items():
        assert torch.equal(v.detach().cpu(), w_ref[k]), k       # weights and codebook buffers travelled too
    _, _, loss = m.run(*b3)
    loss.backward()
    opt.step()
    st = opt.state_dict()["state"]
    assert float(st[0]["step"]) == 3.0                            # bias correction continues at step 3, not 1
    # and a restored optimizer state lands in whatever engine is live / comes next
    opt2 = optim.FusedAdam(m, lr=1e-3)
    opt2.load_state_dict({"state": ref, "param_groups": opt.state_dict()["param_groups"]})
    same(opt2.state_dict()["state"])


def test_encode_and_conditioning_leave_the_histogram_alone():
    hps, m = _tiny()
    batch = _batch(m, 2)
    m.run(*batch)
    eng = m._engine
    torch.cuda.synchronize()
    hist = eng.ind_hist.clone()
    assert float(hist.sum()) == eng.Q
    eng.encode()
    eng.conditioning()
    torch.cuda.synchronize()
    assert torch.equal(eng.ind_hist, hist)

    def source():
        while True:
            yield batch
    m.init_codebook(source(), 5 * eng.Q)                        # 80 samples >= K = 64 codes
    torch.cuda.synchronize()
    assert torch.equal(m._engine.ind_hist, hist)
    assert m.init_codebook_iters >= 1


def test_sampler_repacks_after_fused_adam():
    from ae_wavenet_amd import optim
    hps, m = _tiny()
    opt = optim.FusedAdam(m, lr=1e-2)
    b1 = _batch(m, 1)
    m.eval()
    m(*b1)
    stamp0, smp0 = m._sampler
    m(*b1)
    assert m._sampler[1] is smp0                                # unchanged weights: the packed copy is reused
    m.train()
    _, _, loss = m.run(*b1)
    loss.backward()
    opt.step()                                                   # raw-pointer update: no tensor _version bump
    m.eval()
    m(*b1)
    assert m._sampler[0] != stamp0 and m._sampler[1] is not smp0
