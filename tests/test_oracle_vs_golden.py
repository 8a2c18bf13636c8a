"""Pin the oracle (oracle/) against outputs captured from the unmodified reference modules
(tests/golden/*.npz).  CPU only.  Tolerances: the oracle is fp32 torch like the reference,
so agreement is at fp32 round-off; VQ indices are compared exactly."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from ae_wavenet_amd import config, geometry
from oracle import exact, ref_model as R

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from weights import grad_sketch, np_weights  # noqa: E402

RTOL, ATOL = 2e-5, 2e-6


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name), allow_pickle=False))


def sd_from(z, prefix="w."):
    return {k[len(prefix):]: torch.from_numpy(v).requires_grad_(True)
            for k, v in z.items() if k.startswith(prefix)}


def tiny_hps(z, **over):
    h = json.loads(str(z["hps_json"]))
    n_mel = h.pop("n_mel_ch", None)
    hps = config.make_hps(**{k: v for k, v in h.items() if k not in over}, **over)
    return hps, n_mel


def close(a, b, rtol=RTOL, atol=ATOL):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["identity", "jitter"])
def test_mfcc_inverter_tiny(golden_dir, tag):
    z = load(golden_dir, f"mi_tiny_{tag}.npz")
    hps, _ = tiny_hps(z, global_model="mfcc_inverter")
    geom = geometry.model_geometry(hps, False, hps.n_win_batch)
    sd = sd_from(z)
    wav, mel = torch.from_numpy(z["wav"]), torch.from_numpy(z["mel"]).requires_grad_(True)
    voice, jitter = torch.from_numpy(z["voice"]), torch.from_numpy(z["jitter"])
    pred, target, loss = R.mi_run(sd, hps, geom, wav, mel, voice, jitter, take_compat=True)
    close(pred, z["pred"])
    close(target, z["target"], 0, 0)
    close(loss, z["loss"])
    loss.backward()
    close(mel.grad, z["mel_grad"], 1e-4, 1e-7)
    assert abs(float(mel.grad.std()) - float(z["mel_grad_sd"])) < 1e-6
    for k, p in sd.items():
        close(p.grad if p.grad is not None else torch.zeros_like(p), z["grad." + k], 2e-4, 2e-6)


def _run_ae(z, loss_mode="intended"):
    hps, n_mel = tiny_hps(z, global_model="autoencoder")
    geom = geometry.model_geometry(hps, True, hps.n_win_batch)
    gj = json.loads(str(z["geo_json"]))
    assert geom.enc_in_len == gj["enc_in_len"] and list(geom.trim_dec_in) == gj["trim_dec_in"]
    assert list(geom.trim_ups_out) == gj["trim_ups_out"] and geom.embed_len == gj["embed_len"]
    sd = sd_from(z)
    bufs = {"emb": torch.from_numpy(z["emb0"])} if "emb0" in z and hps.bn_type == "vqvae-ema" else {}
    wav = torch.from_numpy(z["wav"])
    mel = torch.from_numpy(z["mel"]).requires_grad_(True)
    voice, jitter = torch.from_numpy(z["voice"]), torch.from_numpy(z["jitter"])
    eps = torch.from_numpy(z["eps"]) if "eps" in z else None
    anneal = float(z["anneal"]) if "anneal" in z else 0.0
    if hps.bn_type == "vae":
        hps.bn_free_nats = float(z["free_nats"])
    out = R.ae_run(sd, bufs, hps, geom, wav, mel, voice, jitter, loss_mode=loss_mode,
                   take_compat=True, eps=eps, anneal=anneal)
    return hps, sd, mel, out


def _check_grads(z, tag, sd, mel, out, loss):
    names = list(sd)
    gs = torch.autograd.grad(loss, [sd[k] for k in names] + [mel, out["encoding_bn"]],
                             retain_graph=True, allow_unused=True)
    for k, g in zip(names + ["@mel", "@encoding_bn"], gs):
        ref = z[f"{tag}.{k}"]
        if ref.size == 0:
            assert g is None or float(g.abs().max()) == 0.0, k
            continue
        assert g is not None, k
        scale = max(float(np.abs(ref).max()), 1e-12)
        np.testing.assert_allclose(g.numpy(), ref, rtol=5e-4, atol=5e-5 * scale, err_msg=k)


@pytest.mark.parametrize("jk", ["random", "identity"])
def test_autoencoder_vqema_tiny(golden_dir, jk):
    z = load(golden_dir, f"ae_tiny_vqvae-ema_{jk}.npz")
    hps, sd, mel, out = _run_ae(z, "intended")
    close(out["encoding"], z["encoding"])
    close(out["ze"], z["ze"])
    assert np.array_equal(out["min_ind"].numpy(), z["min_ind"])          # bit-exact indices
    close(out["min_dist"], z["min_dist"])
    close(out["encoding_bn"], z["encoding_bn"])
    close(out["pred"], z["pred"], 5e-5, 5e-6)
    close(out["loss"], z["loss_intended"], 1e-5)
    _check_grads(z, "gint", sd, mel, out, out["loss"])
    head = R.vqema_loss(out["pred"], out["target"], out["min_dist"], hps.bn_vq_gamma, "head")
    close(head, z["loss_head"], 1e-5)
    _check_grads(z, "ghead", sd, mel, out, head)
    # EMA statistics and codebook refresh
    K = z["emb0"].shape[0]
    z_sum, n_sum = R.vqema_stats(out["ze"], out["min_ind"], K)
    close(z_sum, z["z_sum"])
    close(n_sum, z["n_sum"], 0, 0)
    emb0 = torch.from_numpy(z["emb0"])
    numer, denom = R.vqema_ema(emb0 * (1 - 0.99), torch.full((K,), 1 - 0.99), z_sum, n_sum, 0.99)
    close(numer, z["ema_numer"], 1e-5, 1e-7)
    close(denom, z["ema_denom"], 1e-5, 1e-7)
    close(R.vqema_codebook(numer, denom), z["emb1"], 1e-4, 1e-6)
    assert abs(np.mean(out["enc_frac_zero"]) - np.mean(z["enc_frac_zero"])) < 1e-9


def test_autoencoder_vae_tiny(golden_dir):
    z = load(golden_dir, "ae_tiny_vae_random.npz")
    hps, sd, mel, out = _run_ae(z)
    close(out["mu"], z["mu"])
    close(out["sigma_sq"], z["sigma_sq"])
    close(out["encoding_bn"], z["encoding_bn"])
    close(out["pred"], z["pred"], 5e-5, 5e-6)
    close(out["loss"], z["loss"], 1e-5)
    close(out["kl"], z["metric.kl_div_loss"], 1e-5)
    _check_grads(z, "g", sd, mel, out, out["loss"])


def test_autoencoder_ae_tiny(golden_dir):
    z = load(golden_dir, "ae_tiny_ae_identity.npz")
    hps, sd, mel, out = _run_ae(z)
    close(out["ze"], z["ze"])
    close(out["pred"], z["pred"], 5e-5, 5e-6)
    close(out["loss"], z["loss"], 1e-5)
    close(out["norm"], z["metric.norm"], 1e-5)
    _check_grads(z, "g", sd, mel, out, out["loss"])


def test_autoencoder_vq_tiny(golden_dir):
    z = load(golden_dir, "ae_tiny_vqvae_identity.npz")
    hps, sd, mel, out = _run_ae(z)
    assert np.array_equal(out["min_ind"].numpy(), z["min_ind"])
    close(out["min_dist"], z["min_dist"])
    close(out["loss"], z["loss_intended"], 1e-5)
    _check_grads(z, "gint", sd, mel, out, out["loss"])


# ------------------------------------------------------------------------------------------
# full-width single modules (weights regenerated from the recorded seeds)
# ------------------------------------------------------------------------------------------
def _gated_shapes(final):
    s = {"conv_signal.weight": (256, 368, 2), "conv_signal.bias": (256,),
         "conv_gate.weight": (256, 368, 2), "conv_gate.bias": (256,),
         "proj_signal.weight": (256, 138, 1), "proj_gate.weight": (256, 138, 1),
         "dil_skp.weight": (256, 256, 1)}
    if not final:
        s["dil_res.weight"] = (368, 256, 1)
    return s


@pytest.mark.parametrize("tag", ["mid", "final"])
def test_gated_layer_full_width(golden_dir, tag):
    z = load(golden_dir, f"gated_full_{tag}.npz")
    final = tag == "final"
    dil = int(z["dil"])
    w = np_weights(_gated_shapes(final), int(z["seed"]))
    sd = {"L." + k: torch.from_numpy(v) for k, v in w.items()}
    rs = np.random.RandomState(int(z["in_seed"]))
    x = torch.from_numpy(rs.uniform(-1, 1, (2, 368, 40)).astype(np.float32))
    cond = torch.from_numpy(rs.uniform(-1, 1, (2, 138, 40 - dil + 5)).astype(np.float32))
    sig, skp = R.gated_layer(sd, "L.", x, cond, dil, 5, 9, final)
    close(sig, z["sig"], 2e-5, 5e-6)
    close(skp, z["skp"], 2e-5, 5e-6)


def vqema_full_inputs(z):
    w = np_weights({"linear.weight": (32, 768, 1)}, int(z["w_seed"]))
    rs = np.random.RandomState(int(z["in_seed"]))
    emb = (rs.standard_normal((4096, 32)) * 0.7).astype(np.float32)
    x = (rs.standard_normal((2, 768, 29)) * 2.0).astype(np.float32)
    return w["linear.weight"], emb, x


def test_vqema_full_width(golden_dir):
    z = load(golden_dir, "vqema_full.npz")
    W, emb, x = vqema_full_inputs(z)
    ze = torch.nn.functional.conv1d(torch.from_numpy(x), torch.from_numpy(W))
    close(ze, z["ze"], 2e-5, 2e-6)
    # torch oracle on the reference's own ze -> identical indices
    zr = torch.from_numpy(z["ze"])
    md, mi, zq = R.vq_forward(zr, torch.from_numpy(emb), "scaled_l2")
    assert np.array_equal(mi.numpy(), z["min_ind"])
    close(md, z["min_dist"], 1e-5, 1e-7)
    close(zq, z["zq"], 0, 0)
    md2, mi2, _ = R.vq_forward(zr, torch.from_numpy(emb), "sq_l2")
    assert np.array_equal(mi2.numpy(), z["l2_min_ind"])
    # the two metrics genuinely disagree on this data (SURVEY C-2)
    assert (z["l2_min_ind"] != z["min_ind"]).sum() > 5
    # exact-order C oracle on the reference's ze: same indices; distances to 1 ulp-ish
    q = z["ze"].transpose(0, 2, 1).reshape(-1, 32)
    ind, dist, sec = exact.vq_nearest(q, emb, "scaled_l2")
    assert np.array_equal(ind.reshape(2, 29), z["min_ind"])
    np.testing.assert_allclose(dist.reshape(2, 29), z["min_dist"], rtol=2e-6)
    # recorded top-2 margins show no near-tie in this fixture
    assert z["margin"].min() > 1e-5
    np.testing.assert_allclose((sec - dist).reshape(2, 29), z["margin"], rtol=1e-3, atol=1e-6)
    ind2, dist2, _ = exact.vq_nearest(q, emb, "sq_l2")
    assert np.array_equal(ind2.reshape(2, 29), z["l2_min_ind"])
    # stats / EMA / codebook in the exact-order oracle vs the reference's values
    z_sum, n_sum = exact.vq_stats(q, ind, 4096)
    np.testing.assert_array_equal(n_sum, z["n_sum"])
    np.testing.assert_allclose(z_sum[ind], z["z_sum_rows"], rtol=1e-6, atol=1e-6)
    numer, denom = exact.ema(emb * np.float32(1 - 0.99), np.full(4096, 1 - 0.99, np.float32),
                             z_sum, n_sum, 0.99)
    np.testing.assert_allclose(denom, z["ema_denom"], rtol=1e-6)
    np.testing.assert_allclose(numer[ind], z["ema_numer_rows"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(exact.codebook(numer, denom)[ind], z["emb1_rows"], rtol=1e-5, atol=1e-6)


ENC_SHAPES = {}
for _i, (_f, _cin) in enumerate(zip((3, 3, 4, 3, 3, 1, 1, 1, 1), (39,) + (768,) * 8)):
    ENC_SHAPES[f"net.{_i}.conv.weight"] = (768, _cin, _f)
    ENC_SHAPES[f"net.{_i}.conv.bias"] = (768,)


def test_encoder_full_width(golden_dir):
    z = load(golden_dir, "encoder_full.npz")
    w = np_weights(ENC_SHAPES, int(z["w_seed"]))
    rs = np.random.RandomState(int(z["in_seed"]))
    mel = rs.standard_normal((2, 39, 30)).astype(np.float32)
    out, frac = R.encoder_forward({k: torch.from_numpy(v) for k, v in w.items()}, "",
                                  torch.from_numpy(mel))
    close(out, z["out"], 2e-5, 2e-5)
    np.testing.assert_allclose(frac, z["frac_zero"], atol=2e-4)
    # exact-order C oracle: same function, fixed summation order
    out_c = exact.encoder_cl(w, "", mel.transpose(0, 2, 1))
    np.testing.assert_allclose(out_c.transpose(0, 2, 1), z["out"], rtol=2e-5, atol=2e-5)


def test_lc_upsample_full_width(golden_dir):
    z = load(golden_dir, "lc_upsample_full.npz")
    hps = config.make_hps("vqvae-ema")
    shapes = {"lc_conv.weight": (128, 32, 3), "lc_conv.bias": (128,)}
    for i, f in enumerate(hps.lc_upsample_filt_sizes):
        shapes[f"lc_upsample.{i}.tconv.weight"] = (128, 128, f)
        shapes[f"lc_upsample.{i}.tconv.bias"] = (128,)
    # the generator seeded *all* WaveNet parameters in sorted-name order; rebuild that order
    full = wavenet_param_shapes(hps, 32)
    w = np_weights(full, int(z["w_seed"]))
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    rs = np.random.RandomState(int(z["in_seed"]))
    lc = torch.from_numpy(rs.standard_normal((2, 32, 7)).astype(np.float32))
    y = torch.nn.functional.conv1d(lc, sd["lc_conv.weight"], sd["lc_conv.bias"])
    y = R.upsample_stack(sd, "", y, hps.lc_upsample_filt_sizes, hps.lc_upsample_strides)
    assert y.shape[2] == int(z["out_len"])
    close(y[:, ::5, :], z["out_sub"], 2e-5, 2e-6)


def wavenet_param_shapes(hps, n_lc_in, prefix=""):
    """Parameter names/shapes of the reference WaveNet module."""
    R_, D, S, P, Q = hps.n_res, hps.n_dil, hps.n_skp, hps.n_post, hps.n_quant
    C = hps.n_lc_out + hps.n_global_embed
    s = {"lc_conv.weight": (hps.n_lc_out, n_lc_in, 3), "lc_conv.bias": (hps.n_lc_out,),
         "cond.speaker_embedding.weight": (hps.n_global_embed, hps.n_speakers),
         "cond.speaker_embedding.bias": (hps.n_global_embed,),
         "base_layer.weight": (R_, Q, 1), "base_layer.bias": (R_,),
         "post1.weight": (P, S, 1), "post1.bias": (P,),
         "post2.weight": (Q, P, 1), "post2.bias": (Q,)}
    for i, f in enumerate(hps.lc_upsample_filt_sizes):
        s[f"lc_upsample.{i}.tconv.weight"] = (hps.n_lc_out, hps.n_lc_out, f)
        s[f"lc_upsample.{i}.tconv.bias"] = (hps.n_lc_out,)
    n_layers = hps.n_blocks * hps.n_block_layers
    for i in range(n_layers):
        p = f"conv_layers.{i}."
        s[p + "conv_signal.weight"] = (D, R_, 2)
        s[p + "conv_signal.bias"] = (D,)
        s[p + "conv_gate.weight"] = (D, R_, 2)
        s[p + "conv_gate.bias"] = (D,)
        s[p + "proj_signal.weight"] = (D, C, 1)
        s[p + "proj_gate.weight"] = (D, C, 1)
        s[p + "dil_skp.weight"] = (S, D, 1)
        if i != n_layers - 1:
            s[p + "dil_res.weight"] = (R_, D, 1)
    return {prefix + k: v for k, v in s.items()}


@pytest.mark.parametrize("fixture", ["mi_full.npz", "mi_full_real.npz"])
def test_mfcc_inverter_full_width(golden_dir, fixture):
    """Full-width (13.5 M parameter) MfccInverter, B=2, w=100: loss, logits, gradients.  mi_full_real.npz: windows of
    real mu-law audio from the reference's dat/librispeech.some.dat (BASELINE configs[0])."""
    z = load(golden_dir, fixture)
    z["wav"] = z["wav"].astype(np.float32)
    hps = config.make_hps("mi", n_win_batch=100)
    shapes = json.loads(str(z["param_names"]))
    assert shapes == {k: list(v) for k, v in wavenet_param_shapes(hps, 39, "wavenet.").items()}
    w = np_weights(shapes, int(z["seed"]))
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in w.items()}
    geom = geometry.model_geometry(hps, False, 100)
    mel = torch.from_numpy(z["mel"]).requires_grad_(True)
    pred, target, loss = R.mi_run(sd, hps, geom, torch.from_numpy(z["wav"]), mel,
                                  torch.from_numpy(z["voice"]), torch.from_numpy(z["jitter"]))
    close(loss, z["loss"], 1e-5)
    close(target, z["target"], 0, 0)
    close(pred[:, :, ::9], z["pred_sub"], 1e-4, 2e-5)
    loss.backward()
    close(mel.grad, z["mel_grad"], 2e-3, 1e-8)
    for k in z:
        if k.startswith("grad."):
            close(sd[k[5:]].grad, z[k], 2e-3, 1e-7)
        elif k.startswith("gradslice."):
            close(sd[k[10:]].grad[:8, :8], z[k], 2e-3, 1e-7)
    # every gradient, whole tensor: norm and seeded random projections recorded from the reference's gradients
    sk = grad_sketch(list(shapes), {k: sd[k].grad.numpy() for k in shapes})
    for k in shapes:
        ref = z["gsketch." + k]
        assert np.linalg.norm(sk[k] - ref) <= 2e-3 * max(np.linalg.norm(ref), 1e-12), k
        close(torch.tensor(float(np.linalg.norm(sd[k].grad.numpy().astype(np.float64)))), z["gnorm." + k], 1e-3, 1e-12)


def test_recloss(golden_dir):
    z = load(golden_dir, "recloss.npz")
    pred = torch.from_numpy(z["pred"]).requires_grad_(True)
    loss = R.rec_loss(pred, torch.from_numpy(z["target"]))
    close(loss, z["loss"], 1e-6)
    loss.backward()
    close(pred.grad, z["grad"], 1e-5, 1e-9)


def test_adam_matches_torch():
    """a17: third-party torch.optim.Adam (defaults) is the reference's optimizer."""
    torch.manual_seed(3)
    p0 = torch.randn(1000)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([p], lr=1e-3)
    m = torch.zeros(1000)
    v = torch.zeros(1000)
    q = p0.clone()
    for step in range(1, 6):
        g = torch.randn(1000) * (10.0 ** (step - 3))
        p.grad = g.clone()
        opt.step()
        q, m, v = R.adam_step(q, g, m, v, step, 1e-3)
        np.testing.assert_allclose(q.numpy(), p.detach().numpy(), rtol=2e-6, atol=1e-7)


# ------------------------------------------------------------------------------------------
# the reference's autoregressive sampler (WaveNet.forward_test, wavenet.py:367-531)
# ------------------------------------------------------------------------------------------
def sampler_case(golden_dir):
    """Fixture mi_tiny_sampler.npz: the sequences the unmodified reference sampler generated for two replicas and
    the probabilities it passed to torch.multinomial at its first 96 steps.  Returns what both the oracle test
    below and the GPU sampler test need."""
    z = load(golden_dir, "mi_tiny_sampler.npz")
    hps, _ = tiny_hps(z, global_model="mfcc_inverter", n_win_batch=int(z["n_win_batch"]))
    geom = geometry.model_geometry(hps, False, hps.n_win_batch)
    rf = int(z["rf"]) - 1                                      # reference counts the current sample in its "rf"
    assert [geom.trim_dec_in[0], geom.trim_dec_in[0] + geom.dec_in_len] == z["wav_cond_offset"].tolist()
    assert geom.dec_in_len == geom.n_win + rf
    out = z["out"].astype(np.int64)                            # [1 + R][n_ts]: row 0 = the given wav
    assert np.array_equal(out[0], z["wav"][0, geom.trim_dec_in[0]:].astype(np.int64))
    assert np.array_equal(out[1:, :rf + 1], np.repeat(out[:1, :rf + 1], out.shape[0] - 1, 0))   # primed with rf + 1 samples
    return z, hps, geom, rf, out


def test_reference_sampler_is_the_training_graph_run_incrementally(golden_dir):
    """What the parity of the MI355X sampler rests on: the probabilities the reference sampler draws from at
    position c are softmax of the TRAINING graph's output (the oracle's decoder_forward, itself pinned above) for
    the sequence generated so far.  Step s of the reference draws position rf + 1 + s."""
    z, hps, geom, rf, out = sampler_case(golden_dir)
    sd = {k: v.detach() for k, v in sd_from(z).items()}
    mel, voice, jitter = torch.from_numpy(z["mel"]), torch.from_numpy(z["voice"]), torch.from_numpy(z["jitter"])
    n = geom.n_win
    for r in range(out.shape[0] - 1):
        wav = torch.from_numpy(z["wav"]).clone()
        o = geom.trim_dec_in[0]
        wav[0, o:o + geom.dec_in_len] = torch.from_numpy(out[1 + r, :geom.dec_in_len]).float()
        with torch.no_grad():
            quant = R.decoder_forward(sd, "wavenet.", hps, wav, mel, voice, jitter, o, geom.dec_in_len,
                                      geom.trim_ups_out, take_compat=True)
        p = torch.softmax(quant[0], 0).t().numpy()             # [n_win][Q]; row i: position i + rf + 1
        steps = min(n, z["probs"].shape[0])
        np.testing.assert_allclose(p[:steps], z["probs"][:steps, r], rtol=2e-4, atol=1e-7)
