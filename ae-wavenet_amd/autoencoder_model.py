"""Drop-in `AutoEncoder(hps)`: the reference's autoencoder module surface
(autoencoder_model.py:23-260) on the MI355X-native training engine.

The reference class is mid-refactor at HEAD and cannot be constructed (SURVEY C-5); this class
follows the live calling convention of `MfccInverter` (single `hps` argument, `run(wav, mel,
voice, jitter)`), with the semantics of autoencoder_model.py:206-259.

    import autoencoder_model as ae        # resolves to this file when ae-wavenet_amd/ is first on sys.path
    model = ae.AutoEncoder(hps)
"""
from __future__ import annotations

import torch

from .surface import HipModelBase, _BottleneckFacade, _EncoderFacade


class AutoEncoder(HipModelBase):
    def __init__(self, hps, loss_mode: str = "intended", take_compat: bool = False,
                 update_codebook_every_step: bool = True, n_mel=None):
        super().__init__(hps, "autoencoder", loss_mode, take_compat, update_codebook_every_step, n_mel)
        self.encoder = _EncoderFacade()
        self.bottleneck = _BottleneckFacade(self)
        try:                                    # the reference's MFCC front-end, when importable
            import mfcc as _mfcc                # noqa: F401  (librosa-based, CPU; mfcc.py:39-76)
            self.mfcc = _mfcc.ProcessWav(sample_rate=hps.sample_rate, win_sz=hps.mfcc_win_sz,
                                         hop_sz=hps.mfcc_hop_sz, n_mels=hps.n_mels, n_mfcc=hps.n_mfcc)
        except Exception:                       # librosa absent: the same callable on the device (AEW_OP_MFCC)
            from .mfcc import ProcessWav
            self.mfcc = ProcessWav(sample_rate=hps.sample_rate, win_sz=hps.mfcc_win_sz, hop_sz=hps.mfcc_hop_sz,
                                   n_mels=hps.n_mels, n_mfcc=hps.n_mfcc)

    @property
    def decoder(self):                          # model.decoder.set_n_replicas(n) / .n_quant (wavenet.py:296)
        return self

    def init_codebook(self, data_source, n_samples):
        """k-means initialisation of the codebook from encoder outputs
        (autoencoder_model.py:171-199).  `data_source` yields (wav, mel, voice, jitter, ...)."""
        if self.bn_type not in ("vqvae", "vqvae-ema"):
            raise RuntimeError("init_codebook only applies to the vqvae model types")
        from . import kmeans as KM
        d, K = self.hps.bn_n_out, self.hps.bn_vq_n_embed
        samples, e = None, 0
        with torch.no_grad():
            while e != n_samples:
                batch = next(data_source)
                wav, mel, voice, jitter = batch[:4]
                eng = self._ensure_engine(wav.shape[0])
                if samples is None:
                    samples = torch.empty(n_samples, d, dtype=torch.float32, device=eng.device)
                eng.set_inputs(wav, mel, voice, jitter)
                eng.encode()                                     # encoder + bottleneck.linear only (no VQ side effects)
                ze = eng.lin.tensor()[:, :, :d].reshape(-1, d)
                c = min(n_samples - e, ze.shape[0])
                samples[e:e + c] = ze[:c]                        # stays in HBM
                e += c
        # Lloyd iterations on the device (kmeans.py); the reference calls scipy.cluster.vq.kmeans on the host here
        codes, self.init_codebook_distortion, self.init_codebook_iters = KM.kmeans(samples, K, seed=getattr(self, "kmeans_seed", 0))
        eng = self._engine
        emb = codes
        if self.bn_type == "vqvae-ema":
            eng.emb[:emb.shape[0]].copy_(emb)
            eng.init_ema_from_emb()
        else:
            eng.ps.view("bottleneck.emb")[:emb.shape[0]].copy_(emb)
