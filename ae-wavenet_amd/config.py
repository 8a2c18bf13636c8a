"""Hyper-parameter object and the ``par/*.json`` key map.

The reference has two config systems: the live ``hparams.py`` registry
(hparams.py:23-110, attribute-dict ``Hyperparams``) and the legacy prefixed JSON files
``par/arch.*.json`` / ``par/train.*.json`` (parse_tools.py:83-166).  The build's
modules take the *live* form (a dict with attribute access) and this module converts
the JSON schema into it, so either can drive the same model.
"""
from __future__ import annotations

import json
from typing import Any, Dict, Iterable


class Hyperparams(dict):
    """dict with attribute access (same contract as reference hparams.py:6-20)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(f"attribute {k} undefined")

    def __setattr__(self, k, v):
        self[k] = v

    def __getstate__(self):
        return self

    def __setstate__(self, state):
        self.update(state)


# Defaults of the live registry (hparams.py:39-96) plus the autoencoder keys that only
# exist in the JSON schema (parse_tools.py:93-161).
_DEFAULTS: Dict[str, Any] = dict(
    # mfcc (hparams.py:39-46)
    sample_rate=16000, mfcc_win_sz=400, mfcc_hop_sz=160, n_mels=80, n_mfcc=13, n_lc_in=39,
    # wavenet (hparams.py:51-68)
    filter_sz=2, n_lc_out=128, lc_upsample_strides=[5, 4, 4, 2],
    lc_upsample_filt_sizes=[25, 16, 16, 16], n_res=368, n_dil=256, n_skp=256, n_post=256,
    n_quant=256, n_blocks=2, n_block_layers=10, n_global_embed=10, n_speakers=40,
    jitter_prob=0.0, free_nats=9, bias=True,
    # model selection (hparams.py:74-76, parse_tools.py:109-111)
    global_model="mfcc_inverter",
    # autoencoder-only keys (par/arch.*.json; defaults parse_tools.py:117-123)
    enc_n_out=768, bn_type="vqvae-ema", bn_n_out=32, bn_vq_gamma=0.25,
    bn_vq_ema_gamma=0.99, bn_vq_n_embed=4096, bn_free_nats=9,
    bn_anneal_weight_steps=[0], bn_anneal_weight_vals=[0.0],
    # train (hparams.py:82-96)
    hw="GPU", n_batch=16, n_win_batch=5000, n_epochs=10, save_interval=1000,
    progress_interval=1, skip_loop_body=False, n_loader_workers=4, log_dir="/tmp",
    random_seed=2507, learning_rate_steps=[0, 4e6, 6e6, 8e6],
    learning_rate_rates=[1e-4, 5e-5, 5e-5, 5e-5], ckpt_template="%.ckpt", ckpt_file=None,
)

# par-key -> hps-key.  Both spellings of the MFCC window keys occur
# (par/arch.basic.json:3-4 vs par/arch.mi.json:4-5).
_PAR_TO_HPS = {
    "pre_sample_rate": "sample_rate",
    "pre_mfcc_win_sz": "mfcc_win_sz", "pre_win_sz": "mfcc_win_sz",
    "pre_mfcc_hop_sz": "mfcc_hop_sz", "pre_hop_sz": "mfcc_hop_sz",
    "pre_n_mels": "n_mels", "pre_n_mfcc": "n_mfcc",
    "dec_filter_sz": "filter_sz", "dec_n_lc_out": "n_lc_out",
    "dec_lc_upsample_strides": "lc_upsample_strides",
    "dec_lc_upsample_filt_sizes": "lc_upsample_filt_sizes",
    "dec_n_res": "n_res", "dec_n_dil": "n_dil", "dec_n_skp": "n_skp", "dec_n_post": "n_post",
    "dec_n_quant": "n_quant", "dec_n_blocks": "n_blocks",
    "dec_n_block_layers": "n_block_layers", "dec_n_global_embed": "n_global_embed",
    "mi_n_lc_in": "n_lc_in",
}

# Named architecture / training sets equivalent to the reference's par/*.json files
# (values only; the files themselves stay in the reference checkout).
ARCH = {
    "vqvae-ema": dict(global_model="autoencoder", enc_n_out=768, bn_type="vqvae-ema",
                      bn_n_out=32, bn_vq_gamma=0.25, bn_vq_ema_gamma=0.99,
                      lc_upsample_strides=[5, 4, 4, 4]),
    "vqvae": dict(global_model="autoencoder", enc_n_out=768, bn_type="vqvae", bn_n_out=64,
                  bn_vq_gamma=0.25, lc_upsample_strides=[5, 4, 4, 4]),
    "vae": dict(global_model="autoencoder", enc_n_out=768, bn_type="vae", bn_n_out=64,
                lc_upsample_strides=[5, 4, 4, 4]),
    "ae": dict(global_model="autoencoder", enc_n_out=768, bn_type="ae", bn_n_out=64,
               lc_upsample_strides=[5, 4, 4, 4]),
    "mi": dict(global_model="mfcc_inverter", n_lc_in=39, lc_upsample_strides=[5, 4, 4, 2]),
    # BASELINE.json config 5: 30 dilation layers x 512 residual channels
    "deep": dict(global_model="autoencoder", enc_n_out=768, bn_type="vqvae-ema", bn_n_out=32,
                 bn_vq_gamma=0.25, bn_vq_ema_gamma=0.99, lc_upsample_strides=[5, 4, 4, 4],
                 n_blocks=3, n_res=512),
}


def make_hps(*named: str, **overrides) -> Hyperparams:
    """Defaults <- named ARCH sets (in order) <- keyword overrides.  Unknown keys are
    rejected like hparams.py:31-33."""
    h = Hyperparams(_DEFAULTS)
    for n in named:
        h.update(ARCH[n])
    for k, v in overrides.items():
        if k not in _DEFAULTS:
            raise ValueError(f"{k} not in default args")
        h[k] = v
    _finalize(h)
    return h


def from_checkpoint_hps(saved: Dict[str, Any], **overrides) -> Hyperparams:
    """hps as stored in a reference checkpoint (checkpoint.py:27-29: `Hyperparams(**ckpt['hps'])`, then
    the command-line overrides).  Keys this build does not use (output_dir, hw, ...) are carried along."""
    h = Hyperparams(_DEFAULTS)
    h.update(saved)
    h.update(overrides)
    _finalize(h)
    return h


def from_par(arch: Dict[str, Any], train: Dict[str, Any] | None = None, **overrides) -> Hyperparams:
    """Build hps from the contents of a ``par/arch.*.json`` (+ optional
    ``par/train.*.json``) dictionary."""
    h = Hyperparams(_DEFAULTS)
    if "global_model" not in arch:
        h["global_model"] = "autoencoder"          # parse_tools.py:109-111 default
    for src in (arch, train or {}):
        for k, v in src.items():
            k2 = _PAR_TO_HPS.get(k, k)
            if k2 not in _DEFAULTS:
                raise ValueError(f"unknown config key {k}")
            h[k2] = v
    for k, v in overrides.items():
        if k not in _DEFAULTS:
            raise ValueError(f"{k} not in default args")
        h[k] = v
    _finalize(h)
    return h


def load_par(arch_path: str, train_path: str | None = None, **overrides) -> Hyperparams:
    with open(arch_path) as fh:
        arch = json.load(fh)
    train = None
    if train_path:
        with open(train_path) as fh:
            train = json.load(fh)
    return from_par(arch, train, **overrides)


def _finalize(h: Hyperparams) -> None:
    if h.global_model == "autoencoder":
        # decoder LC input = bottleneck output (autoencoder_model.py:86)
        h["n_lc_in"] = h.bn_n_out
    if len(h.lc_upsample_strides) != len(h.lc_upsample_filt_sizes):
        raise ValueError("upsample strides / filter sizes length mismatch")
    for f, s in zip(h.lc_upsample_filt_sizes, h.lc_upsample_strides):
        if f % s:
            raise ValueError("each upsample filter size must be a multiple of its stride "
                             "(doc/upsampling_notes.txt:117-180)")
    if h.filter_sz != 2:
        raise ValueError("the gated-layer kernels implement filter_sz == 2 only")
