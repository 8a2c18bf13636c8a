"""ORACLE (test infrastructure only): the reference's MFCC front-end, mfcc.py:39-76, in numpy / scipy float64.

PARITY PARTLY UNPINNED.  mfcc.py calls librosa.feature.mfcc + librosa.feature.delta; librosa (0.7-era: the reference
also calls librosa.output.write_wav, removed in 0.8) is a third-party dependency that is absent from this image and
not vendored by the reference, and the reference holds no test vectors for this path.  Restated here from librosa's
published algorithm:
    stft(y, n_fft=win, hop_length=hop, win_length=win, window='hann', center=True, pad_mode='reflect') -> |.|^2
    -> filters.mel(sr, n_fft, n_mels, fmin=0, fmax=sr/2, htk=False, norm='slaney')
    -> power_to_db(ref=1.0, amin=1e-10, top_db=80.0) -> scipy.fftpack.dct(type=2, norm='ortho')[:n_mfcc]
    delta(x, width=9, order=k) = scipy.signal.savgol_filter(x, 9, deriv=k, polyorder=k, axis=-1, mode='interp')
What IS pinned: the window (scipy.signal.get_window), the DFT (numpy.fft), the DCT (scipy.fft.dct) and the delta
filter (scipy.signal.savgol_filter) are the very library calls librosa makes.  What is restated without librosa to
check against: the Slaney mel filterbank, the dB conversion, the reflect padding / framing, and the reference's own
left-pad / trim arithmetic (mfcc.py:47-72, with vconv.VirtualConv's wing sizes for a 400/160 filter).  Round 5: the
first three are held to a SECOND, independently written restatement of the same librosa calls that this image carries
(transformers.audio_utils: mel_filter_bank / spectrogram / power_to_db, validated against librosa by its own project) -
they agree to round-off (tests/test_mfcc.py) - which narrows, but does not close, the gap: still no librosa output.
"""
import numpy as np
import scipy.fft
import scipy.signal


def hz_to_mel(f):
    """Slaney: linear below 1 kHz, logarithmic above (librosa.hz_to_mel, htk=False)."""
    f = np.asarray(f, np.float64)
    f_sp = 200.0 / 3
    mel = f / f_sp
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mel)


def mel_to_hz(m):
    m = np.asarray(m, np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin=0, fmax=sr/2, htk=False, norm='slaney') -> [n_mels][1 + n_fft/2]."""
    fft_f = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(0.0), hz_to_mel(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    w = np.zeros((n_mels, len(fft_f)))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return w * enorm[:, None]


def wings(win_sz):
    """vconv.VirtualConv(filter_info=win_sz): l_wing = (win-1)//2, r_wing = win-1-l_wing (vconv.py:91-94)."""
    l = (win_sz - 1) // 2
    return l, win_sz - 1 - l


def mfcc_and_deltas(wav, sample_rate=16000, win_sz=400, hop_sz=160, n_mels=80, n_mfcc=13):
    """ProcessWav.__call__ (mfcc.py:39-76).  wav (T,) -> (3 * n_mfcc, F)."""
    wav = np.asarray(wav, np.float64)
    l_wing, r_wing = wings(win_sz)
    adj_l = l_wing + (1 if win_sz % 2 == 0 else 0)
    left_pad, trim_left, trim_right = adj_l % hop_sz, adj_l // hop_sz, r_wing // hop_sz
    y = np.concatenate((np.zeros(left_pad), wav))
    ypad = np.pad(y, win_sz // 2, mode="reflect")                    # center=True
    n_frames = 1 + (len(ypad) - win_sz) // hop_sz
    assert n_frames == 1 + len(y) // hop_sz                          # mfcc_pred_output_size (mfcc.py:60-68)
    win = scipy.signal.get_window("hann", win_sz, fftbins=True)
    frames = np.stack([ypad[f * hop_sz:f * hop_sz + win_sz] * win for f in range(n_frames)])
    power = np.abs(np.fft.rfft(frames, n=win_sz, axis=1)) ** 2       # [F][1 + win/2]
    mel = power @ mel_filterbank(sample_rate, win_sz, n_mels).T      # [F][n_mels]
    db = 10.0 * np.log10(np.maximum(1e-10, mel))
    db = np.maximum(db, db.max() - 80.0)                             # top_db over the whole call
    mfcc = scipy.fft.dct(db, type=2, norm="ortho", axis=1)[:, :n_mfcc].T     # [n_mfcc][F]
    mfcc = mfcc[:, trim_left:(-trim_right or None)]
    d1 = scipy.signal.savgol_filter(mfcc, 9, deriv=1, polyorder=1, axis=-1, mode="interp")
    d2 = scipy.signal.savgol_filter(mfcc, 9, deriv=2, polyorder=2, axis=-1, mode="interp")
    return np.concatenate((mfcc, d1, d2), axis=0)
