"""MFCC + delta + delta-delta front-end on the device (the reference: mfcc.py:39-76 `ProcessWav`, run per window on
the host by the DataLoader's collate function, data.py:230, through librosa).

`DeviceMfcc(...)(wav)` takes the batch of mu-law windows that is already on the device for the decoder and returns the
conditioning input `mel` (B, 3 * n_mfcc, frames) - the same frames the reference's left-pad / trim arithmetic yields
(mfcc.py:47-72).  This module builds the constant tables (Hann window, DFT twiddles, Slaney mel filterbank, DCT-II
rows, Savitzky-Golay taps incl. the 'interp' edge rows) and the descriptor; the transform itself is AEW_OP_MFCC.
Parity: oracle/mfcc_ref.py (librosa is absent here: the oracle restates its algorithm and says which parts are
pinned to the scipy / numpy calls librosa itself makes).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib as L
from .plan import Plan, Workspace


def _mel_weights(sr: int, n_fft: int, n_mels: int) -> np.ndarray:
    """Triangular filters on the Slaney mel scale, area-normalised (librosa.filters.mel defaults of the 0.7 era)."""
    f_sp, brk = 200.0 / 3.0, 1000.0
    step = math.log(6.4) / 27.0

    def to_mel(f):
        return f / f_sp if f < brk else brk / f_sp + math.log(f / brk) / step

    def to_hz(m):
        return f_sp * m if m < brk / f_sp else brk * math.exp(step * (m - brk / f_sp))

    lo, hi = to_mel(0.0), to_mel(sr / 2.0)
    edges = [to_hz(lo + (hi - lo) * i / (n_mels + 1)) for i in range(n_mels + 2)]
    n_bins = n_fft // 2 + 1
    w = np.zeros((n_mels, n_bins))
    for m in range(n_mels):
        a, c, b = edges[m], edges[m + 1], edges[m + 2]
        for k in range(n_bins):
            f = k * (sr / 2.0) / (n_bins - 1)
            up, down = (f - a) / (c - a), (b - f) / (b - c)
            w[m, k] = max(0.0, min(up, down)) * 2.0 / (b - a)
    return w


def _savgol_rows(order: int, width: int = 9) -> np.ndarray:
    """[9 interior taps | 4 x 9 left-edge rows | 4 x 9 right-edge rows] of the order-th derivative of the least-squares
    polynomial of degree `order` over `width` points (scipy.signal.savgol_filter(deriv=polyorder=order, mode='interp'):
    interior = fit centred on the point; the first / last width//2 points evaluate the fit of the first / last window)."""
    half = width // 2

    def taps(x):
        # deriv == polyorder: the derivative of the fitted polynomial is order! times its top coefficient, the same
        # at every position of the window
        P = np.linalg.pinv(np.vander(x, order + 1, increasing=True))       # polynomial coefficients = P @ y
        return math.factorial(order) * P[order]

    centred = taps(np.arange(-half, half + 1, dtype=np.float64))
    edge = taps(np.arange(width, dtype=np.float64))
    return np.concatenate([centred, np.tile(edge, half), np.tile(edge, half)])


class DeviceMfcc:
    def __init__(self, device, sample_rate=16000, win_sz=400, hop_sz=160, n_mels=80, n_mfcc=13):
        self.dev = torch.device(device)
        self.sr, self.win, self.hop, self.n_mels, self.n_mfcc = sample_rate, win_sz, hop_sz, n_mels, n_mfcc
        self.n_out = 3 * n_mfcc                                     # mfcc.py:35
        l_wing = (win_sz - 1) // 2                                  # vconv.VirtualConv wings of the analysis window
        r_wing = win_sz - 1 - l_wing
        adj = l_wing + (1 if win_sz % 2 == 0 else 0)                # mfcc.py:47-52
        self.left_pad, self.trim_left, self.trim_right = adj % hop_sz, adj // hop_sz, r_wing // hop_sz
        n = np.arange(win_sz)
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(self.dev)
        self.window = f32(0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_sz))             # periodic Hann
        self.twiddle = f32(np.stack([np.cos(2.0 * np.pi * n / win_sz), np.sin(2.0 * np.pi * n / win_sz)], 1))
        self.melw = f32(_mel_weights(sample_rate, win_sz, n_mels))
        k, j = np.arange(n_mfcc)[:, None], np.arange(n_mels)[None, :]
        d = np.cos(np.pi * (2 * j + 1) * k / (2.0 * n_mels)) * np.sqrt(2.0 / n_mels)
        d[0] *= np.sqrt(0.5)                                                         # DCT-II, norm='ortho'
        self.dct = f32(d)
        self.sg = f32(np.stack([_savgol_rows(1), _savgol_rows(2)]))
        self._plans = {}
        L.load()

    def n_frames(self, n: int) -> int:
        """Frames returned for windows of n samples."""
        return 1 + (self.left_pad + n) // self.hop - self.trim_left - self.trim_right

    def _plan(self, B: int, n: int):
        key = (B, n)
        if key not in self._plans:
            ws = Workspace(self.dev)
            nf = 1 + (self.left_pad + n) // self.hop
            Ft = nf - self.trim_left - self.trim_right
            wav = ws.alloc("mfcc.wav", B * n, torch.float32)
            out = ws.alloc("mfcc.out", B * self.n_out * Ft, torch.float32)
            scratch = ws.alloc("mfcc.scratch", B * nf * (self.n_mels + 1 + self.n_mfcc), torch.float32)
            m = L.Mfcc()
            m.wav, m.wav_bs, m.n, m.B = wav.data_ptr(), n, n, B
            m.win, m.hop, m.n_bins, m.n_mels, m.n_mfcc = self.win, self.hop, self.win // 2 + 1, self.n_mels, self.n_mfcc
            m.left_pad, m.trim_left, m.trim_right, m.n_frames = self.left_pad, self.trim_left, self.trim_right, nf
            m.window, m.twiddle, m.melw = self.window.data_ptr(), self.twiddle.data_ptr(), self.melw.data_ptr()
            m.dct, m.sg, m.scratch = self.dct.data_ptr(), self.sg.data_ptr(), scratch.data_ptr()
            m.out, m.out_bs, m.out_pitch = out.data_ptr(), self.n_out * Ft, Ft
            p = Plan("mfcc")
            p.add(L.OP_MFCC, m, "mfcc")
            self._plans[key] = (p, wav, out, Ft, ws)
        return self._plans[key]

    def __call__(self, wav: torch.Tensor) -> torch.Tensor:
        """wav (B, n) float tensor on the device (the mu-law values the reference feeds, data.py:228-230) ->
        (B, 3 * n_mfcc, frames) float32."""
        if wav.dim() != 2 or wav.device != self.dev:
            raise L.AewError("DeviceMfcc: (B, n) tensor on the device expected")
        B, n = wav.shape
        p, buf, out, Ft, _ = self._plan(B, n)
        buf[:B * n].view(B, n).copy_(wav)
        p.run(torch.cuda.current_stream(self.dev).cuda_stream)
        return out[:B * self.n_out * Ft].view(B, self.n_out, Ft).clone()


class ProcessWav:
    """Drop-in for the reference's `mfcc.ProcessWav` (mfcc.py:27-76): same constructor arguments and attributes, same
    call - a 1-D numpy window in, a (3 * n_mfcc, frames) numpy array out - computed by AEW_OP_MFCC instead of librosa.
    It is what `model.mfcc` is when the reference's librosa front-end cannot be imported, so `DataProcessor(hps, dat,
    model.mfcc, ...)` (checkpoint.py:42) keeps working in the main process (n_loader_workers = 0); with worker
    processes use `DevicePrefetcher(..., mfcc=DeviceMfcc(...))`, which computes whole batches on the copy stream."""

    def __init__(self, sample_rate=16000, win_sz=400, hop_sz=160, n_mels=80, n_mfcc=13, name=None, device="cuda:0"):
        self.sample_rate, self.window_sz, self.hop_sz, self.n_mels, self.n_mfcc = sample_rate, win_sz, hop_sz, n_mels, n_mfcc
        self.n_out = n_mfcc * 3
        self.name, self._device, self._dm = name, device, None

    def __call__(self, wav):
        if self._dm is None:
            self._dm = DeviceMfcc(self._device, self.sample_rate, self.window_sz, self.hop_sz, self.n_mels, self.n_mfcc)
        x = torch.as_tensor(np.ascontiguousarray(wav), dtype=torch.float32).reshape(1, -1).to(self._dm.dev)
        return self._dm(x)[0].cpu().numpy()
