"""Plan construction verified on CPU: a TrainEngine is built on device 'cpu' (plans only — it
cannot execute them), the CPU plan interpreter (tests/plan_emulator.py) runs the exact op
records, and results are compared with the golden vectors captured from the reference.

Two passes: 'wide' stores the decoder's bf16 tensors as fp32 (tight tolerances: checks every
index map / pack record / epilogue flag), 'bf16' keeps the real storage types (loose
tolerance: sizes the rounding error the GPU path will show)."""
import json
import os

import numpy as np
import pytest
import torch

from ae_wavenet_amd import _lib as L, config, model as M, plan as PL
from tests.plan_emulator import Emu


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name), allow_pickle=False))


@pytest.fixture(params=["wide", "bf16"])
def mode(request, monkeypatch):
    if request.param == "wide":
        monkeypatch.setitem(PL.TORCH_DT, L.BF16, torch.float32)
        monkeypatch.setitem(PL.ESIZE, L.BF16, 4)
    return request.param


def tiny_hps(z, **over):
    h = json.loads(str(z["hps_json"]))
    n_mel = h.pop("n_mel_ch", None)
    return config.make_hps(**{k: v for k, v in h.items() if k not in over}, **over), n_mel


def make_engine(z, kind, n_mel, loss_mode="intended"):
    hps, nm = tiny_hps(z, global_model=kind)
    eng = M.TrainEngine(hps, B=z["wav"].shape[0], device="cpu", n_mel=n_mel or nm, loss_mode=loss_mode,
                        take_compat=True, update_codebook_every_step=False)
    for k in eng.ps.names():
        eng.ps.view(k).copy_(torch.from_numpy(z["w." + k]))
    return hps, eng


def run(eng, z, eps=None):
    eng.set_inputs(torch.from_numpy(z["wav"]), torch.from_numpy(z["mel"]), torch.from_numpy(z["voice"]),
                   torch.from_numpy(z["jitter"]), eps=eps)
    emu = Emu(eng.ws)
    emu.run(eng.fwd_a)
    emu.run(eng.fwd_b)
    emu.run(eng.bwd)
    return emu


def tol(mode, tight, loose):
    return tight if mode == "wide" else loose


def close_dir(mode, got, ref, tight=2e-4):
    """wide: max-normalised error; bf16 toy nets: direction only (see check_grads)."""
    if mode == "wide":
        assert np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12) < tight
    else:
        cos = float((got * ref).sum() / max(np.linalg.norm(got) * np.linalg.norm(ref), 1e-30))
        assert cos > 0.9, cos


def check_grads(eng, z, tag, mode, skip=()):
    worst = 0.0
    for k in eng.ps.names():
        ref = z[f"{tag}.{k}"]
        got = eng.ps.view(k, grad=True).numpy()
        if ref.size == 0:
            assert np.abs(got).max() == 0, k
            continue
        scale = max(np.abs(ref).max(), 1e-12)
        err = np.abs(got - ref).max() / scale
        worst = max(worst, err)
        if mode == "wide":
            assert err < 2e-4, (k, err)
        else:
            # bf16 storage on a 12-channel toy net: single ReLU-mask flips move small tensors by
            # O(1/12); require direction agreement only (full-width accuracy is measured on the GPU)
            cos = float((got * ref).sum() / max(np.linalg.norm(got) * np.linalg.norm(ref), 1e-30))
            assert cos > 0.9, (k, cos, err)
    return worst


@pytest.mark.parametrize("tag", ["identity", "jitter"])
def test_mfcc_inverter_plan(golden_dir, mode, tag):
    z = load(golden_dir, f"mi_tiny_{tag}.npz")
    hps, eng = make_engine(z, "mfcc_inverter", 7)
    run(eng, z)
    pred = eng.logits()[:, :-1, :].permute(0, 2, 1).numpy()
    np.testing.assert_allclose(pred, z["pred"], rtol=tol(mode, 1e-4, 5e-2), atol=tol(mode, 2e-5, 3e-2))
    assert abs(float(eng.loss_buf[0]) - float(z["loss"])) < tol(mode, 1e-5, 2e-2)
    check_grads(eng, z, "grad", mode)
    # d(loss)/d(mel) comes out of the jitter scatter (channels-last)
    mg = eng.dec.dlc_src.tensor()[:, :, :7].permute(0, 2, 1).numpy()
    close_dir(mode, mg, z["mel_grad"])
    # the per-step statistics run() reports are plan ops too (AEW_OP_MOMENTS / the "metrics" reduction)
    if "mel_grad_sd" in z:
        assert abs(float(eng.gstat[1]) - float(z["mel_grad_sd"])) < tol(mode, 1e-6, 2e-2) * max(1.0, float(z["mel_grad_sd"]))
    assert abs(float(eng.gstat[0]) - float(mg.mean())) < 1e-6 and abs(float(eng.gstat[1]) - float(mg.std(ddof=1))) < 1e-6
    n_pos = eng.B * (eng.n_win - 1)
    assert abs(float(eng.met_buf[1]) - float(eng.dec.nll.sum()) / n_pos) < 1e-5


@pytest.mark.parametrize("jk,loss_mode,gtag,ltag", [("random", "intended", "gint", "loss_intended"),
                                                    ("identity", "intended", "gint", "loss_intended"),
                                                    ("random", "head", "ghead", "loss_head")])
def test_autoencoder_vqema_plan(golden_dir, mode, jk, loss_mode, gtag, ltag):
    z = load(golden_dir, f"ae_tiny_vqvae-ema_{jk}.npz")
    hps, eng = make_engine(z, "autoencoder", None, loss_mode)
    eng.emb.copy_(torch.from_numpy(z["emb0"]))
    eng.init_ema_from_emb()
    run(eng, z)
    ze = eng.lin.tensor()[:, :, :eng.d].permute(0, 2, 1).numpy()
    np.testing.assert_allclose(ze, z["ze"], rtol=2e-5, atol=2e-6)
    assert np.array_equal(eng.ind[:eng.Q].view(eng.B, -1).numpy(), z["min_ind"])
    np.testing.assert_allclose(eng.min_dist[:eng.Q].view(eng.B, -1).numpy(), z["min_dist"], rtol=1e-5)
    np.testing.assert_allclose(eng.z_sum.numpy(), z["z_sum"], rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(eng.n_sum.numpy(), z["n_sum"])
    np.testing.assert_allclose(eng.ema_numer.numpy(), z["ema_numer"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(eng.ema_denom.numpy(), z["ema_denom"], rtol=1e-5)
    pred = eng.logits()[:, :-1, :].permute(0, 2, 1).numpy()
    np.testing.assert_allclose(pred, z["pred"], rtol=tol(mode, 1e-4, 5e-2), atol=tol(mode, 2e-5, 3e-2))
    assert abs(float(eng.loss_buf[0]) / float(z[ltag]) - 1) < tol(mode, 1e-5, 5e-3)
    check_grads(eng, z, gtag, mode)
    mg = eng.enc.dy[0].tensor()[:, :, :9].permute(0, 2, 1).float().numpy()
    close_dir(mode, mg, z[gtag + ".@mel"])
    bg = eng.dec.dlc_src.tensor()[:, :, :eng.d].permute(0, 2, 1).numpy()
    ref = z[gtag + ".@encoding_bn"]
    if ref.size:
        close_dir(mode, bg, ref)
    # codebook refresh
    Emu(eng.ws).run(eng.cb)
    np.testing.assert_allclose(eng.emb.numpy(), z["emb1"], rtol=1e-4, atol=1e-6)
    # enc_az metric (wave_encoder.py:46)
    cnt = eng.enc.zero_cnt[:9].numpy().astype(np.float64)
    numel = np.array([eng.B * eng.geom.enc_lens[i + 1] * hps.enc_n_out for i in range(9)], np.float64)
    np.testing.assert_allclose(cnt / numel, z["enc_frac_zero"], atol=1e-9)
    check_diagnostics(np.concatenate([eng.diag.numpy()[:6], eng.diag_pk.numpy()[6:9]]), z, exact=(mode == "wide"))


def check_diagnostics(dg, z, exact):
    """AEW_OP_VQ_DIAG against the metrics the reference's VQEMALoss reported (vqema_bn.py:251-264) and, for the
    peak statistics (not stored in the fixtures), the same formulas on the reference's own `pred`."""
    for i, k in enumerate(("min_ze", "max_ze", "min_emb", "max_emb", "hst_ent")):
        np.testing.assert_allclose(dg[i], z["metric." + k], rtol=1e-4, err_msg=k)
    assert dg[5] == float(z["metric.nunq"])
    lp = torch.log_softmax(torch.from_numpy(z["pred"]).double(), 1)          # (B, Q, w-1)
    pk, am = lp.max(dim=1)
    tol_ = 1e-4 if exact else 3e-2
    np.testing.assert_allclose(dg[6], float(pk.mean()), rtol=tol_, atol=tol_)
    np.testing.assert_allclose(dg[7], float(pk.std()), rtol=tol_, atol=tol_)
    assert abs(dg[8] - am.unique().numel()) <= (0 if exact else 3)


def test_unfolded_wgrad_and_ones_channel_colsum(golden_dir, monkeypatch):
    """Force the per-batch (non-folded) wgrad layout on the toy nets: exercises multi-slab gradient
    unpacking and the bias gradients taken from the constant-one pad channel of x."""
    monkeypatch.setitem(PL.TORCH_DT, L.BF16, torch.float32)
    monkeypatch.setitem(PL.ESIZE, L.BF16, 4)
    from ae_wavenet_amd import engine as E
    monkeypatch.setattr(E.DecoderPlan, "wgrad_group", 0)       # one TN op per matrix (the grouped form has no slabs)
    lib = L.load()
    lib.aew_set_tn_fold_rows(0)
    try:
        z = load(golden_dir, "mi_tiny_jitter.npz")
        hps, eng = make_engine(z, "mfcc_inverter", 7)
        assert any(lab.startswith("colsum.dfg (from wgrad") for lab in eng.bwd.labels)
        run(eng, z)
        check_grads(eng, z, "grad", "wide")
        z = load(golden_dir, "ae_tiny_vqvae-ema_random.npz")
        hps, eng = make_engine(z, "autoencoder", None)
        eng.emb.copy_(torch.from_numpy(z["emb0"]))
        eng.init_ema_from_emb()
        run(eng, z)
        check_grads(eng, z, "gint", "wide")
    finally:
        lib.aew_set_tn_fold_rows(4096)


@pytest.mark.parametrize("group,split,tile", [(64, 0, 256), (64, 0, 128), (8, 0, 256), (3, 0, 128), (1, 0, 256), (64, 3, 256)])
def test_grouped_wgrad_plan(golden_dir, monkeypatch, group, split, tile):
    """The default backward: the gated stack's weight gradients as grouped launches (AEW_OP_GEMM_TN_GROUP, one
    result per matrix, every tile named once in the tile map: the interpreter checks that) and the per-batch
    bias / speaker sums from the running ones-channel snapshots, for several group sizes."""
    monkeypatch.setitem(PL.TORCH_DT, L.BF16, torch.float32)
    monkeypatch.setitem(PL.ESIZE, L.BF16, 4)
    from ae_wavenet_amd import engine as E
    monkeypatch.setattr(E.DecoderPlan, "wgrad_group", group)
    monkeypatch.setattr(E.DecoderPlan, "wgrad_split_layers", split)   # top layers as one split-K op per matrix
    monkeypatch.setattr(E.DecoderPlan, "wgrad_tile", tile)
    z = load(golden_dir, "mi_tiny_jitter.npz")
    hps, eng = make_engine(z, "mfcc_inverter", 7)
    NL = len(eng.geom.layers)
    kinds = [op.kind for op in eng.bwd.ops]
    # the stack's groups + the upsampler / lc-conv group (the MFCC inverter has no encoder group)
    assert kinds.count(L.OP_GEMM_TN_GROUP) == max(1, -(-(NL - split) // group)) + 1
    assert not any(lab.startswith(("db.post", "db.base", "db.lc", "wgrad.base", "wgrad.up", "wgrad.lc", "wgrad.p"))
                   for lab in eng.bwd.labels)
    assert sum(lab.startswith("wgrad.fg") for lab in eng.bwd.labels) == split
    spk = [op.u.spkb for op in eng.bwd.ops if op.kind == L.OP_SPK_BWD]
    if group < NL and split == 0:
        # several groups: the first one (layers NL - group .., skip, post network) is followed by the speaker / gated-bias
        # gradients of ITS layers and the early unpack - everything from that layer up is final there (the data-parallel
        # schedule exchanges that region first: TrainEngine.bwd_a1 / dec_hi_offset); the rest of the layers at the end
        hi = NL - group
        assert len(spk) == 2 and [(s_.layer_range & 0xffff, s_.layer_range >> 16) for s_ in spk] == [(hi, NL - hi), (0, hi)]
        i_grp = eng.bwd.labels.index("wgrad.group0 (layers %d.., skip, post)" % hi)
        assert eng.bwd.labels[i_grp + 1: i_grp + 3] == ["spk_bwd (upper layers)", "unpack grads (decoder, upper layers)"]
        assert eng.dec_hi_offset == min(eng.ps.off[n] for n in eng.ps.names() if f"conv_layers.{hi}." in n)
        assert len(eng.bwd_a1.ops) == i_grp + 3 and len(eng.bwd_a1.ops) + len(eng.bwd_a2.ops) == len(eng.bwd_a.ops)
    else:
        assert len(spk) == 1 and spk[0].colsum_running == NL - split and spk[0].layer_range == 0
        assert eng.dec_hi_offset is None and eng.bwd_a1 is None
    run(eng, z)
    check_grads(eng, z, "grad", "wide")


def test_schedule_options_leave_the_plan_results_alone(golden_dir, monkeypatch):
    """DecoderPlan.split_chains / split_chains_bwd (half-batch chains on lanes 4 / 5), tail_lane (the last grouped launch as a
    branch; ignored when the chain is split) and wgrad_cursor (progress words + the op that zeroes them in front of the
    grouped launch) re-schedule or pace; serially interpreted, the plans give the golden gradients."""
    monkeypatch.setitem(PL.TORCH_DT, L.BF16, torch.float32)
    monkeypatch.setitem(PL.ESIZE, L.BF16, 4)
    from ae_wavenet_amd import engine as E
    z = load(golden_dir, "mi_tiny_jitter.npz")
    for cfg in (dict(split_chains=True, split_chains_bwd=True, tail_lane=4), dict(tail_lane=4, wgrad_cursor=True)):
        for k in ("split_chains", "split_chains_bwd", "tail_lane", "wgrad_cursor"):
            monkeypatch.setattr(E.DecoderPlan, k, cfg.get(k, 0))
        hps, eng = make_engine(z, "mfcc_inverter", 7)
        labs, ops = eng.bwd.labels, eng.bwd.ops
        i_grp = next(i for i, lab in enumerate(labs) if lab.startswith("wgrad.group0"))
        if cfg.get("split_chains_bwd"):
            assert eng.B % 2 == 0 and "dz.0.c0" in labs and "dx.0.c1" in labs and "dz.0" not in labs
            assert {ops[labs.index(f"dz.1.c{c}")].lane for c in (0, 1)} == {4, 5}
            assert ops[i_grp].lane not in (4, 5) and ops[i_grp].join == 1         # meets both chains; no tail branch behind a chain
            assert ops[labs.index("dcond")].join == 1 and eng.dec.tail_lane_used == 0
            assert any(lab.endswith(".c1") for lab in eng.fwd_b.labels)
        else:
            assert ops[i_grp].lane == 4 and labs[i_grp + 1] == "spk_bwd" and ops[i_grp + 1].lane == 4
            assert labs[i_grp - 1].startswith("zero:") and labs[i_grp - 1].endswith(".cursors") and ops[i_grp - 1].lane == 4
            assert E.DecoderPlan.wgrad_cursor is True
            assert ops[i_grp].u.tng.cursor_stride == 64 and ops[i_grp].u.tng.cursors
            assert ops[labs.index("unpack grads (decoder)")].lane == 4
        run(eng, z)
        check_grads(eng, z, "grad", "wide")


def test_autoencoder_vae_plan(golden_dir, mode):
    z = load(golden_dir, "ae_tiny_vae_random.npz")
    hps, eng = make_engine(z, "autoencoder", None)
    eng.hps.bn_free_nats = float(z["free_nats"])
    # rebuild so the free-nats clamp is baked with the fixture's value
    hps2 = config.make_hps(**{**dict(hps), "bn_free_nats": float(z["free_nats"])})
    eng = M.TrainEngine(hps2, B=2, device="cpu", n_mel=9, take_compat=True)
    for k in eng.ps.names():
        eng.ps.view(k).copy_(torch.from_numpy(z["w." + k]))
    eng.set_anneal_weight(float(z["anneal"]))
    run(eng, z, eps=torch.from_numpy(z["eps"]))
    pred = eng.logits()[:, :-1, :].permute(0, 2, 1).numpy()
    np.testing.assert_allclose(pred, z["pred"], rtol=tol(mode, 1e-4, 5e-2), atol=tol(mode, 2e-5, 3e-2))
    assert abs(float(eng.loss_buf[0]) / float(z["loss"]) - 1) < tol(mode, 1e-5, 5e-3)
    assert abs(float(eng.loss_buf[2]) / float(z["metric.kl_div_loss"]) - 1) < 1e-5
    check_grads(eng, z, "g", mode)


def test_autoencoder_ae_plan(golden_dir, mode):
    z = load(golden_dir, "ae_tiny_ae_identity.npz")
    hps, eng = make_engine(z, "autoencoder", None)
    run(eng, z)
    assert abs(float(eng.loss_buf[0]) / float(z["loss"]) - 1) < tol(mode, 1e-5, 5e-3)
    check_grads(eng, z, "g", mode)


def test_autoencoder_vq_plan(golden_dir, mode):
    z = load(golden_dir, "ae_tiny_vqvae_identity.npz")
    hps, eng = make_engine(z, "autoencoder", None)
    run(eng, z)
    assert np.array_equal(eng.ind[:eng.Q].view(eng.B, -1).numpy(), z["min_ind"])
    assert abs(float(eng.loss_buf[0]) / float(z["loss_intended"]) - 1) < tol(mode, 1e-5, 5e-3)
    check_grads(eng, z, "gint", mode)


def test_adam_plan(mode):
    hps = config.make_hps("mi", n_res=8, n_dil=8, n_skp=8, n_post=8, n_lc_out=8, n_global_embed=2,
                          n_speakers=3, n_blocks=1, n_block_layers=2, n_win_batch=5, n_lc_in=4)
    eng = M.TrainEngine(hps, B=1, device="cpu", n_mel=4)
    torch.manual_seed(0)
    n = eng.ps.numel
    p0 = torch.randn(n)
    eng.ps.params[:n].copy_(p0)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=3e-4)
    for step in range(3):
        g = torch.randn(n)
        eng.ps.grads[:n].copy_(g)
        ref.grad = g.clone()
        opt.step()
        eng.step_count += 1
        a = eng.opt.array()[0].u.adam
        a.lr, a.bc1, a.bc2 = 3e-4, 1 - 0.9 ** eng.step_count, 1 - 0.999 ** eng.step_count
        Emu(eng.ws).run(eng.opt)
        np.testing.assert_allclose(eng.ps.params[:n].numpy(), ref.detach().numpy(), rtol=2e-6, atol=1e-7)


def test_copy_table_tiled_form_addresses_the_same_elements():
    """CopyTableBuilder._tiled_form reorders / merges the dims of a transposing record; the (source, destination) offset
    pairs it describes must be exactly those of the record as given, the tile must fit the kernel's LDS buffer, and the
    records the element-wise forms already handle sequentially must be left alone."""
    import itertools
    from ae_wavenet_amd import _lib as L
    from ae_wavenet_amd.plan import CopyTableBuilder

    def pairs(dims, ss, ds):
        out = set()
        for idx in itertools.product(*[range(d) for d in dims]):
            out.add((sum(i * s for i, s in zip(idx, ss)), sum(i * s for i, s in zip(idx, ds))))
        return out
    cases = [((1, 48, 40, 3), (0, 120, 3, 1), (0, 1, 144, 48)),          # [o][c][k] -> [(c, k)][o]
             ((1, 1, 37, 50), (0, 0, 50, 1), (0, 0, 1, 37)),              # ragged transpose
             ((1, 1, 48, 32), (0, 0, 64, 2), (0, 0, 1, 48))]              # source stride 2 (every other tap)
    for dims, ss, ds in cases:
        got = CopyTableBuilder._tiled_form(list(dims), list(ss), list(ds), L.F32, L.BF16, 1, False)
        assert got is not None, (dims, ss, ds)
        d2, s2, t2, ta, tb = got
        assert pairs(d2, s2, t2) == pairs(dims, ss, ds)
        assert t2[2] == 1 and 0 < s2[3] <= 2 and ta * tb <= 1024 and tb * (ta | 1) <= 1600
    for dims, ss, ds in [((1, 8, 100, 3), (0, 300, 3, 1), (0, 300, 1, 100)),   # tap <-> channel inside a row
                         ((1, 8, 3, 100), (0, 300, 100, 1), (0, 300, 1, 3)),   # ... and back
                         ((4, 23, 2, 16), (736, 2, 1, 46), (32, 128, 64, 1)),   # gate-permuted groups: 16-wide runs stay element-wise
                         ((1, 1, 16, 64), (0, 0, 64, 1), (0, 0, 64, 1))]:    # plain copy
        assert CopyTableBuilder._tiled_form(list(dims), list(ss), list(ds), L.F32, L.F32, 1, False) is None
    assert CopyTableBuilder._tiled_form([1, 1, 64, 64], [0, 0, 64, 1], [0, 0, 1, 64], L.F32, L.F32, 4, False) is None   # slab sums


def test_copy_table_interleave_form_addresses_the_same_elements():
    """CopyTableBuilder._interleave_form: same (source, destination) offset pairs as the record given, dims in the order
    the kernel's register-permutation forms expect, nothing selected when shape or alignment is not covered."""
    import itertools
    from ae_wavenet_amd import _lib as L
    from ae_wavenet_amd.plan import CopyTableBuilder

    def pairs(dims, ss, ds):
        return {(sum(i * s for i, s in zip(idx, ss)), sum(i * s for i, s in zip(idx, ds)))
                for idx in itertools.product(*[range(d) for d in dims])}
    got = CopyTableBuilder._interleave_form([1, 6, 3, 40], [0, 120, 40, 1], [0, 120, 1, 3], 0, 0, L.F32, L.F32, 1, False)
    d, s, t, k, mode = got
    assert (k, mode) == (-3, 1) and d[2] == 3 and s[3] == 1 and t[2] == 1 and t[3] == 3
    assert pairs(d, s, t) == pairs([1, 6, 3, 40], [0, 120, 40, 1], [0, 120, 1, 3])
    got = CopyTableBuilder._interleave_form([2, 6, 48, 2], [1152, 96, 2, 1], [2304, 192, 1, 64], 64, 32, L.F32, L.BF16, 1, False)
    d, s, t, k, mode = got
    assert (k, mode) == (-2, 2) and d[3] == 2 and s[2] == 2 and t[2] == 1 and t[3] == 64
    assert pairs(d, s, t) == pairs([2, 6, 48, 2], [1152, 96, 2, 1], [2304, 192, 1, 64])
    none = [([1, 6, 3, 42], [0, 126, 42, 1], [0, 126, 1, 3], 0, 0, L.F32, L.F32),        # 42 channels: not a multiple of 4
            ([1, 6, 3, 40], [0, 120, 40, 1], [0, 120, 1, 3], 4, 0, L.F32, L.F32),        # source not 16-byte aligned
            ([1, 6, 5, 40], [0, 200, 40, 1], [0, 200, 1, 5], 0, 0, L.F32, L.F32),        # five taps
            ([1, 6, 44, 2], [0, 88, 2, 1], [0, 176, 1, 44], 0, 0, L.F32, L.BF16)]        # bf16 planes need multiples of 8
    for dims, ss, ds, sp, dp, a, b in none:
        assert CopyTableBuilder._interleave_form(dims, ss, ds, sp, dp, a, b, 1, False) is None


def test_split_grouped_descriptor_keeps_its_bias_gradient(monkeypatch):
    """Round-3 ADVICE: with lc rows x B > 1024 the lc-conv weight gradient becomes a SPLIT grouped descriptor, which has
    no block that sees every row and therefore no column-sum by-product; the bias gradient must then come from a
    column-sum op (it raised NotImplementedError).  Grouped plan == one-op-per-matrix plan on every gradient."""
    monkeypatch.setitem(PL.TORCH_DT, L.BF16, torch.float32)
    monkeypatch.setitem(PL.ESIZE, L.BF16, 4)
    from ae_wavenet_amd import engine as E
    B, n_win = 3, 2500
    monkeypatch.setattr(E.DecoderPlan, "ups_split_rows", 16)            # (1024 in production: reached at B = 16, w = 20000)
    hps = config.make_hps("mi", n_res=8, n_dil=8, n_skp=8, n_post=8, n_lc_out=8, n_global_embed=2, n_speakers=3,
                          n_blocks=1, n_block_layers=2, n_win_batch=n_win, n_lc_in=4)
    gen = torch.Generator().manual_seed(5)
    grads = {}
    for grp in (64, 0):
        eng = M.TrainEngine(hps, B=B, device="cpu", n_mel=4, wgrad_group=grp)
        assert eng.dec.ups_in[0].rows * B > 16
        assert "db.lc" in eng.bwd.labels
        gen.manual_seed(5)
        for k in eng.ps.names():
            eng.ps.view(k).copy_(torch.randn(eng.ps.shape[k], generator=gen) * 0.3)
        g = eng.geom
        wav = torch.randint(0, 256, (B, g.enc_in_len), generator=gen).float()
        mel = torch.randn(B, 4, g.mel_len, generator=gen)
        voice = torch.randint(0, 3, (B,), generator=gen)
        jitter = torch.arange(g.embed_len).repeat(B, 1)
        eng.set_inputs(wav, mel, voice, jitter)
        emu = Emu(eng.ws)
        for p in (eng.fwd_a, eng.fwd_b, eng.bwd):
            emu.run(p)
        grads[grp] = {k: eng.ps.view(k, grad=True).clone() for k in eng.ps.names()}
    for k, ref in grads[0].items():
        got = grads[64][k]
        scale = max(float(ref.abs().max()), 1e-20)
        assert float((got - ref).abs().max()) / scale < 2e-4, k
    assert float(grads[64]["wavenet.lc_conv.bias"].abs().max()) > 0


def test_first_forward_plan_reads_no_decoder_parameter(golden_dir, monkeypatch):
    """TrainEngine.pack_dec_late: the decoder's weight layouts are packed at the head of fwd_b, and nothing in fwd_a - op
    descriptors or the records of its pack tables - points into the decoder's region of the flat parameter buffer.  That is
    what lets a data-parallel step keep the decoder's parameter all-gather in flight under the encoder forward
    (dp.DataParallel.forward).  (A data-parallel rank's form of the plans: merge_packs off - a lone process packs every
    layout in one launch at the head of fwd_a instead.)"""
    import ctypes as C
    monkeypatch.setattr(M.TrainEngine, "merge_packs", False)
    z = load(golden_dir, "ae_tiny_vqvae-ema_random.npz")
    hps, eng = make_engine(z, "autoencoder", None)
    base, lo, n = eng.ps.params.data_ptr(), eng.dec_grad_offset, eng.ps.numel
    assert 0 < lo < n
    d_lo, d_hi = base + 4 * lo, base + 4 * n

    def ints(obj):
        if isinstance(obj, (C.Structure, C.Union)):
            return [v for f in obj._fields_ for v in ints(getattr(obj, f[0]))]
        if isinstance(obj, C.Array):
            return [v for e in obj for v in ints(e)]
        return [obj] if isinstance(obj, int) else []

    assert "pack weights (decoder)" in eng.fwd_b.labels and "pack weights (decoder)" not in eng.fwd_a.labels
    for op, lab in zip(eng.fwd_a.ops, eng.fwd_a.labels):
        hits = [v for v in ints(getattr(op.u, L.OP_FIELD[op.kind])) if d_lo <= v < d_hi]
        assert not hits, (lab, [hex(v) for v in hits])
    for tb in (eng.pack_first, eng.pack_tbl):
        if tb is not None:
            assert not [r for r in tb.recs if d_lo <= (r.src or 0) < d_hi]
    assert eng.pack_dec.recs and all(d_lo <= r.src < d_hi for r in eng.pack_dec.recs)
