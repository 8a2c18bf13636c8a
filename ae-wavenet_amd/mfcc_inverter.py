"""Drop-in `MfccInverter(hps)`: the reference's decoder-only model surface
(mfcc_inverter.py:9-109) on the MI355X-native training engine."""
from __future__ import annotations

from .surface import HipModelBase


class MfccInverter(HipModelBase):
    def __init__(self, hps, take_compat: bool = False):
        super().__init__(hps, "mfcc_inverter", take_compat=take_compat, n_mel=hps.n_lc_in)
        try:
            import mfcc as _mfcc                # the reference's librosa front-end (mfcc.py:39-76)
            self.mfcc = _mfcc.ProcessWav(sample_rate=hps.sample_rate, win_sz=hps.mfcc_win_sz,
                                         hop_sz=hps.mfcc_hop_sz, n_mels=hps.n_mels, n_mfcc=hps.n_mfcc)
        except Exception:                       # librosa absent: the same callable on the device (AEW_OP_MFCC)
            from .mfcc import ProcessWav
            self.mfcc = ProcessWav(sample_rate=hps.sample_rate, win_sz=hps.mfcc_win_sz, hop_sz=hps.mfcc_hop_sz,
                                   n_mels=hps.n_mels, n_mfcc=hps.n_mfcc)

    @property
    def wavenet(self):
        return self
