"""MFCC front-end on the device (ae_wavenet_amd/mfcc.py, AEW_OP_MFCC) against oracle/mfcc_ref.py, the numpy / scipy
restatement of mfcc.py:39-76.  librosa is absent from this image, so the oracle's header says which of its parts are
pinned (the scipy / numpy calls librosa makes) and which are restated; the tests below add the cross-checks that are
possible without it."""
import numpy as np
import pytest
import scipy.signal
import torch

from oracle import mfcc_ref as R

DEV = "cuda:0"


def test_host_tables_against_independent_forms():
    from ae_wavenet_amd import mfcc as M
    # the product's mel filterbank (scalar loops) and the oracle's (vectorised ramps) are written independently
    assert np.abs(M._mel_weights(16000, 400, 80) - R.mel_filterbank(16000, 400, 80)).max() < 1e-12
    fb = R.mel_filterbank(16000, 400, 80)
    assert fb.shape == (80, 201) and (fb >= 0).all() and ((fb > 0).sum(1) >= 1).all()
    # Slaney scale landmarks: 1000 Hz = 15 mel, and the map round-trips
    assert abs(float(R.hz_to_mel(1000.0)) - 15.0) < 1e-12
    f = np.array([0.0, 300.0, 999.0, 1000.0, 4000.0, 8000.0])
    assert np.abs(R.mel_to_hz(R.hz_to_mel(f)) - f).max() < 1e-9
    # Savitzky-Golay rows against the scipy call librosa.feature.delta makes
    for order in (1, 2):
        rows = M._savgol_rows(order)
        x = np.random.RandomState(order).standard_normal(40)
        ref = scipy.signal.savgol_filter(x, 9, deriv=order, polyorder=order, mode="interp")
        got = np.array([rows[9 + f * 9:18 + f * 9] @ x[:9] if f < 4 else
                        rows[45 + (f - 36) * 9:54 + (f - 36) * 9] @ x[31:] if f >= 36 else rows[:9] @ x[f - 4:f + 5]
                        for f in range(40)])
        assert np.abs(got - ref).max() < 1e-12


def test_restated_stages_against_a_third_party_restatement_of_librosa():
    """librosa itself is absent, so the three stages the oracle restates WITHOUT a check of its own - the Slaney mel filterbank,
    the dB conversion with its top_db floor, the centred reflect-padded framing - are held here to an independent
    implementation of the same librosa calls that this image does carry: `transformers.audio_utils` (mel_filter_bank with
    norm = mel_scale = "slaney", spectrogram(center=True, pad_mode="reflect", power=2), power_to_db(db_range=80)), which its
    own project validates against librosa.  Not a pin to librosa - the header of oracle/mfcc_ref.py keeps saying "partly
    unpinned" - but two independently written restatements of the published algorithm that agree to round-off."""
    AU = pytest.importorskip("transformers.audio_utils")
    sr, win, hop, n_mels = 16000, 400, 160, 80
    fb = AU.mel_filter_bank(num_frequency_bins=1 + win // 2, num_mel_filters=n_mels, min_frequency=0.0, max_frequency=sr / 2,
                            sampling_rate=sr, norm="slaney", mel_scale="slaney")                 # [bins][mels]
    ours = R.mel_filterbank(sr, win, n_mels)
    assert fb.shape == ours.T.shape and np.abs(fb - ours.T).max() < 1e-9 * np.abs(ours).max() + 1e-12
    rs = np.random.RandomState(3)
    y = 128 + 60 * np.sin(2 * np.pi * np.arange(11800) * 0.01) + rs.normal(0, 12, 11800)
    window = scipy.signal.get_window("hann", win, fftbins=True)
    # power mel spectrogram through THEIR framing (centre, reflect pad) and filterbank, dB through THEIR conversion ...
    theirs_db = AU.spectrogram(y, window, frame_length=win, hop_length=hop, fft_length=win, power=2.0, center=True,
                               pad_mode="reflect", mel_filters=fb, mel_floor=1e-10, log_mel="dB", reference=1.0,
                               min_value=1e-10, db_range=80.0, dtype=np.float64)                 # [mels][frames]
    # ... against the oracle's own stages on the same samples (mfcc_and_deltas's body without the reference's pad / trim)
    ypad = np.pad(y, win // 2, mode="reflect")
    n_frames = 1 + (len(ypad) - win) // hop
    frames = np.stack([ypad[f * hop:f * hop + win] * window for f in range(n_frames)])
    mel = (np.abs(np.fft.rfft(frames, n=win, axis=1)) ** 2) @ ours.T
    db = 10.0 * np.log10(np.maximum(1e-10, mel))
    db = np.maximum(db, db.max() - 80.0)
    assert theirs_db.shape == db.T.shape
    assert np.abs(theirs_db - db.T).max() < 1e-8 * np.abs(db).max()


def test_oracle_frame_arithmetic():
    """mfcc.py:47-72 for the 400 / 160 analysis window: 40 zeros of left pad, one frame trimmed on each side, and the
    frame counts SURVEY's configuration table lists (11 760 samples -> 72 frames, 6 960 -> 42)."""
    assert R.wings(400) == (199, 200)
    for n, frames in ((11760, 72), (6960, 42), (3600, 21)):
        out = R.mfcc_and_deltas(np.zeros(n) + 128.0)
        assert out.shape == (39, frames)
    # a constant signal: energy only in the lowest bands, every delta zero away from the edges of the call
    out = R.mfcc_and_deltas(np.full(11760, 100.0))
    assert np.abs(out[13:, 8:-8]).max() < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("B,n", [(8, 11760), (2, 3600), (1, 40000)])
def test_device_mfcc_matches_oracle(B, n):
    from ae_wavenet_amd import mfcc as M
    rs = np.random.RandomState(n)
    # mu-law-like windows: a slowly varying tone + noise, quantised to 0..255 (what data.py:228-230 feeds)
    t = np.arange(n)[None, :]
    wav = 128 + 60 * np.sin(2 * np.pi * t * rs.uniform(0.001, 0.05, (B, 1))) + rs.normal(0, 12, (B, n))
    wav = np.clip(np.round(wav), 0, 255).astype(np.float32)
    dm = M.DeviceMfcc(DEV)
    got = dm(torch.from_numpy(wav).to(DEV)).cpu().numpy()
    assert got.shape == (B, 39, dm.n_frames(n))
    for b in range(B):
        ref = R.mfcc_and_deltas(wav[b])
        scale = np.abs(ref[:13]).max()
        err = np.abs(got[b] - ref).max()
        assert err < 2e-4 * scale, (b, err, scale)                  # fp32 DFT / log against float64
    # the engine's frame count for this window length is the one the front-end produces
    again = dm(torch.from_numpy(wav).to(DEV)).cpu().numpy()
    assert np.array_equal(again, got)


@pytest.mark.gpu
def test_prefetcher_computes_mel_on_the_device():
    from ae_wavenet_amd import loader, mfcc as M
    rs = np.random.RandomState(0)
    batches = [(torch.from_numpy(rs.randint(0, 256, (4, 3600)).astype(np.float32)), None,
                torch.from_numpy(rs.randint(0, 5, (4,)))) for _ in range(3)]
    dm = M.DeviceMfcc(DEV)
    seen = 0
    for (wav, mel, voice), src in zip(loader.DevicePrefetcher(iter(batches), DEV, mfcc=dm), batches):
        assert mel.shape == (4, 39, dm.n_frames(3600)) and mel.is_cuda
        ref = R.mfcc_and_deltas(src[0][1].numpy())
        assert np.abs(mel[1].cpu().numpy() - ref).max() < 2e-4 * np.abs(ref[:13]).max()
        assert torch.equal(wav.cpu(), src[0]) and torch.equal(voice.cpu(), src[2])
        seen += 1
    assert seen == 3


@pytest.mark.gpu
def test_model_mfcc_is_the_reference_callable():
    """model.mfcc(window) -> (39, F) numpy, the call data.Collate makes (data.py:230); librosa is not importable here,
    so the models carry the device front-end behind the reference's ProcessWav surface."""
    from ae_wavenet_amd import config, mfcc_inverter as mi
    m = mi.MfccInverter(config.make_hps("mi"))
    assert m.mfcc.n_out == 39 and m.mfcc.window_sz == 400 and m.mfcc.hop_sz == 160
    wav = np.random.RandomState(4).randint(0, 256, 6960).astype(np.float32)
    got = m.mfcc(wav)
    ref = R.mfcc_and_deltas(wav)
    assert got.shape == ref.shape == (39, 42)
    assert np.abs(got - ref).max() < 2e-4 * np.abs(ref[:13]).max()
