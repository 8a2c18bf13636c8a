"""ORACLE — test infrastructure only.  Never imported by the product path.

Plain-PyTorch fp32 CPU restatement of the reference's training hot path, written as
pure functions over a flat ``{state_dict key: tensor}`` mapping (the reference's own key
names).  Backward comes from autograd, so every hand-written HIP
backward kernel has an independent checker.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this package.

Pinned by tests/test_oracle_vs_golden.py against outputs captured from the unmodified
reference modules (tests/golden/*.npz, generator tests/golden/make_golden.py).

Each function cites the reference lines it restates (paths relative to the reference
checkout).  Tensors are NCL like the reference's.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

ENC_FILTERS = (3, 3, 4, 3, 3, 1, 1, 1, 1)          # wave_encoder.py:58-61
ENC_STRIDES = (1, 1, 2, 1, 1, 1, 1, 1, 1)
ENC_RESIDUAL = (False, True, False, True, True, True, True, True, True)


# --------------------------------------------------------------------------------------
# encoder  (wave_encoder.py:34-50, 53-103)
# --------------------------------------------------------------------------------------
def encoder_forward(sd: SD, pre: str, mel: Tensor) -> Tuple[Tensor, List[float]]:
    """mel (B, n_in, F) -> (B, n_out, N_e); also the per-layer zero-activation fraction
    (wave_encoder.py:46, 83-89)."""
    x = mel
    frac = []
    for i, (f, s, res) in enumerate(zip(ENC_FILTERS, ENC_STRIDES, ENC_RESIDUAL)):
        y = F.relu(F.conv1d(x, sd[f"{pre}net.{i}.conv.weight"], sd[f"{pre}net.{i}.conv.bias"],
                            stride=s))
        if res:
            lw = (f - 1) // 2                      # vconv.py:91-94 wings of a centred filter
            y = y + x[:, :, lw:lw + y.shape[2]]
        frac.append(float((y == 0).double().mean()))
        x = y
    return x, frac


# --------------------------------------------------------------------------------------
# bottlenecks
# --------------------------------------------------------------------------------------
def scaled_l2(z: Tensor, emb: Tensor) -> Tensor:
    """dist[b,k,n] = ||z[b,:,n]-emb[k]|| / (||z[b,:,n]|| + ||emb[k]||)  (vqema_bn.py:67-76).
    z (B,d,N), emb (K,d) -> (B,K,N)."""
    diff = z.unsqueeze(1) - emb.unsqueeze(0).unsqueeze(3)          # B,K,d,N
    num = (diff ** 2).sum(dim=2).sqrt()
    den = (z ** 2).sum(dim=1, keepdim=True).sqrt() + (emb ** 2).sum(dim=1).sqrt()[None, :, None]
    return num / den


def squared_l2(z: Tensor, emb: Tensor) -> Tensor:
    """Plain squared L2 (vq_bn.py:39)."""
    diff = z.unsqueeze(1) - emb.unsqueeze(0).unsqueeze(3)
    return (diff ** 2).sum(dim=2)


class _StraightThrough(torch.autograd.Function):
    """Value of `zq`, gradient routed to `ze` (ReplaceGradFn, vqema_bn.py:33-45)."""

    @staticmethod
    def forward(ctx, zq, ze):
        return zq.clone()

    @staticmethod
    def backward(ctx, g):
        return torch.zeros_like(g), g


def vq_forward(ze: Tensor, emb: Tensor, metric: str):
    """Nearest code per (b, n); first minimum wins on ties like torch.min (vqema_bn.py:141).
    Returns min_dist (B,N) [differentiable w.r.t. ze], min_ind (B,N) int64, zq (B,d,N) with
    straight-through gradient."""
    d = scaled_l2(ze, emb.detach()) if metric == "scaled_l2" else squared_l2(ze, emb.detach())
    min_dist, min_ind = d.min(dim=1)
    zq = emb.detach()[min_ind].permute(0, 2, 1)                    # gather_md(...).permute(1,0,2)
    return min_dist, min_ind, _StraightThrough.apply(zq, ze)


def vqema_stats(ze: Tensor, min_ind: Tensor, n_embed: int):
    """z_sum[k] = sum of ze over queries assigned to k, n_sum[k] = count
    (vqema_bn.py:172-188)."""
    d = ze.shape[1]
    flat_ind = min_ind.reshape(-1)
    flat_ze = ze.detach().permute(0, 2, 1).reshape(-1, d)
    z_sum = torch.zeros(n_embed, d, dtype=ze.dtype).index_add_(0, flat_ind, flat_ze)
    n_sum = torch.zeros(n_embed, dtype=ze.dtype).index_add_(0, flat_ind,
                                                            torch.ones_like(flat_ind, dtype=ze.dtype))
    return z_sum, n_sum


def vqema_ema(numer, denom, z_sum, n_sum, gamma):
    """vqema_bn.py:190-195."""
    return gamma * numer + (1.0 - gamma) * z_sum, gamma * denom + (1.0 - gamma) * n_sum


def vqema_codebook(numer, denom):
    """vqema_bn.py:216-222."""
    return numer / denom.unsqueeze(1)


def vae_forward(sd: SD, pre: str, enc_out: Tensor, eps: Tensor):
    """1x1 -> split mu / log sigma^2 -> mu + sigma*eps  (vae_bn.py:26-62).  `eps` is the
    injected N(0,1) draw (the reference calls randn_like, vae_bn.py:51)."""
    lin = F.conv1d(enc_out, sd[f"{pre}linear.weight"])
    d = lin.shape[1] // 2
    mu, log_sigma_sq = lin[:, :d], lin[:, d:]
    sigma = torch.exp(0.5 * log_sigma_sq)
    return mu + sigma * eps, mu, sigma ** 2.0


# --------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------
def nll_terms(pred: Tensor, target: Tensor) -> Tensor:
    """-log softmax(pred)[target]; pred (B,Q,N), target (B,N) float-encoded ints
    (wavenet.py:543-547)."""
    return -torch.gather(F.log_softmax(pred, 1), 1, target.long().unsqueeze(1)).squeeze(1)


def rec_loss(pred, target):
    """RecLoss (wavenet.py:541-552)."""
    return nll_terms(pred, target).mean()


def vqema_loss(pred, target, min_dist, gamma, mode):
    """VQEMALoss (vqema_bn.py:231-266).  mode 'head' = what HEAD computes (commitment
    only, :246); 'intended' = the author's full line (:244) rec.sum() + com.sum()."""
    com = (min_dist * gamma).sum()
    if mode == "head":
        return com
    return nll_terms(pred, target).sum() + com


def vq_loss(pred, target, ze, emb, min_ind, min_dist, gamma):
    """VQLoss (vq_bn.py:72-115): rec.sum() + l2.sum() + com.sum(); the undefined L2Error
    (vq_bn.py:70) taken as the squared error between sg(ze) and the selected code."""
    sel = emb[min_ind]                                             # B,N,d
    l2 = ((ze.detach().permute(0, 2, 1) - sel) ** 2).sum()
    return nll_terms(pred, target).sum() + l2 + (min_dist * gamma).sum()


def sgvb_loss(pred, target, mu, sigma_sq, anneal, free_nats):
    """SGVBLoss (vae_bn.py:76-125): mean NLL + anneal * max(KL_sum, free_nats)."""
    neg_kl = 0.5 * torch.sum(1.0 + torch.log(sigma_sq) - mu * mu - sigma_sq)
    kl = -neg_kl
    return nll_terms(pred, target).mean() + anneal * torch.clamp(kl, min=free_nats), kl


def ae_loss(pred, target, ze, norm_gamma=0.001):
    """AELoss (ae_bn.py:29-46)."""
    ze_norm = (ze ** 2.0).sum(dim=1).sqrt()
    norm = norm_gamma * torch.abs(ze_norm - 1.0).mean()
    return nll_terms(pred, target).mean() + norm, norm


# --------------------------------------------------------------------------------------
# decoder  (wavenet.py:323-364)
# --------------------------------------------------------------------------------------
def lc_gather(lc: Tensor, jitter: Tensor, take_compat: bool) -> Tensor:
    """Jitter gather.  take_compat reproduces the reference's torch.take on the flattened
    tensor (wavenet.py:330-336, SURVEY C-1): every channel c of batch b receives
    lc[0, b, jitter[b, t]].  Otherwise the intended lc[b, c, jitter[b, t]]."""
    B, C, N = lc.shape
    if take_compat:
        flat_idx = jitter + torch.arange(0, jitter.numel(), jitter.shape[1]).unsqueeze(1)
        return torch.take(lc, flat_idx.unsqueeze(1).expand(-1, C, -1))
    return torch.gather(lc, 2, jitter.unsqueeze(1).expand(-1, C, -1))


def upsample_stack(sd: SD, pre: str, x: Tensor, filt_sizes: Sequence[int], strides: Sequence[int]):
    """4x ConvTranspose1d with padding f - s (wavenet.py:142-165)."""
    for i, (f, s) in enumerate(zip(filt_sizes, strides)):
        x = F.conv_transpose1d(x, sd[f"{pre}lc_upsample.{i}.tconv.weight"],
                               sd[f"{pre}lc_upsample.{i}.tconv.bias"], stride=s, padding=f - s)
    return x


def layer_dilations(hps) -> List[int]:
    return [2 ** l for _ in range(hps.n_blocks) for l in range(hps.n_block_layers)]


def gated_layer(sd: SD, pre: str, x: Tensor, cond: Tensor, dil: int, cond_lead: int,
                skip_lead: int, final: bool):
    """GatedResidualCondConv.forward (wavenet.py:91-111)."""
    c = cond[:, :, cond_lead:]
    filt = F.conv1d(x, sd[f"{pre}conv_signal.weight"], sd.get(f"{pre}conv_signal.bias"),
                    dilation=dil) + F.conv1d(c, sd[f"{pre}proj_signal.weight"])
    gate = F.conv1d(x, sd[f"{pre}conv_gate.weight"], sd.get(f"{pre}conv_gate.bias"),
                    dilation=dil) + F.conv1d(c, sd[f"{pre}proj_gate.weight"])
    z = torch.tanh(filt) * torch.sigmoid(gate)
    skp = F.conv1d(z[:, :, skip_lead:], sd[f"{pre}dil_skp.weight"])
    if final:
        sig = x[:, :, dil:]
    else:
        sig = F.conv1d(z, sd[f"{pre}dil_res.weight"]) + x[:, :, dil:]
    return sig, skp


def decoder_forward(sd: SD, pre: str, hps, wav: Tensor, lc: Tensor, voice: Tensor,
                    jitter: Tensor, wav_off: int, dec_in_len: int, trim_ups: Tuple[int, int],
                    take_compat: bool, capture: Optional[dict] = None) -> Tensor:
    """WaveNet.forward_train (wavenet.py:323-364).  wav (B, n_wav) float-encoded ints,
    lc (B, n_lc_in, N_e).  Returns quant (B, Q, n_win)."""
    n_win = dec_in_len - sum(layer_dilations(hps))
    lcj = lc_gather(lc, jitter, take_compat)
    lcc = F.conv1d(lcj, sd[f"{pre}lc_conv.weight"], sd.get(f"{pre}lc_conv.bias"))
    dense = upsample_stack(sd, pre, lcc, hps.lc_upsample_filt_sizes, hps.lc_upsample_strides)
    dense = dense[:, :, trim_ups[0]:trim_ups[1]]
    # speaker embedding: one-hot -> Linear -> broadcast -> concat (wavenet.py:127-140)
    gc = sd[f"{pre}cond.speaker_embedding.weight"][:, voice.long()].t()
    if f"{pre}cond.speaker_embedding.bias" in sd:
        gc = gc + sd[f"{pre}cond.speaker_embedding.bias"]
    cond = torch.cat((dense, gc.unsqueeze(2).expand(-1, -1, dense.shape[2])), dim=1)
    # base layer on the one-hot = column gather (wavenet.py:348-351)
    idx = wav[:, wav_off:wav_off + dec_in_len].long()
    sig = sd[f"{pre}base_layer.weight"][:, :, 0][:, idx].permute(1, 0, 2)
    if f"{pre}base_layer.bias" in sd:
        sig = sig + sd[f"{pre}base_layer.bias"][None, :, None]
    skp_sum = torch.zeros(wav.shape[0], hps.n_skp, n_win)
    dils = layer_dilations(hps)
    rf = sum(dils)
    c = 0
    if capture is not None:
        capture.update(cond=cond, x0=sig)
    for i, d in enumerate(dils):
        c += d
        sig, skp = gated_layer(sd, f"{pre}conv_layers.{i}.", sig, cond, d, c, rf - c,
                               final=(i == len(dils) - 1))
        skp_sum = skp_sum + skp
    post1 = F.conv1d(F.relu(skp_sum), sd[f"{pre}post1.weight"], sd.get(f"{pre}post1.bias"))
    quant = F.conv1d(F.relu(post1), sd[f"{pre}post2.weight"], sd.get(f"{pre}post2.bias"))
    if capture is not None:
        capture.update(skp_sum=skp_sum)
    return quant


# --------------------------------------------------------------------------------------
# whole models
# --------------------------------------------------------------------------------------
def mi_run(sd: SD, hps, geom, wav, mel, voice, jitter, take_compat=True):
    """MfccInverter.run (mfcc_inverter.py:89-108): returns pred, target, loss."""
    quant = decoder_forward(sd, "wavenet.", hps, wav, mel, voice, jitter, geom.trim_dec_in[0],
                            geom.dec_in_len, geom.trim_ups_out, take_compat)
    o = geom.wav_out_off
    target = wav[:, o:o + geom.n_win][:, 1:]
    pred = quant[..., :-1]
    return pred, target, rec_loss(pred, target)


def ae_run(sd: SD, buffers: SD, hps, geom, wav, mel, voice, jitter, loss_mode="intended",
           take_compat=True, eps: Optional[Tensor] = None, anneal: float = 0.0):
    """AutoEncoder.forward/run wiring (autoencoder_model.py:206-259) for all bottleneck
    types.  `buffers` holds the codebook 'emb' for the VQ types.  Returns a dict with
    pred, target, loss and the bottleneck internals."""
    enc_out, frac = encoder_forward(sd, "encoder.", mel)
    out = dict(encoding=enc_out, enc_frac_zero=frac)
    bn = hps.bn_type
    if bn in ("vqvae-ema", "vqvae"):
        ze = F.conv1d(enc_out, sd["bottleneck.linear.weight"])
        emb = buffers["emb"] if bn == "vqvae-ema" else sd["bottleneck.emb"]
        metric = "scaled_l2" if bn == "vqvae-ema" else "sq_l2"
        min_dist, min_ind, code = vq_forward(ze, emb, metric)
        out.update(ze=ze, min_ind=min_ind, min_dist=min_dist)
    elif bn == "vae":
        code, mu, sigma_sq = vae_forward(sd, "bottleneck.", enc_out, eps)
        out.update(mu=mu, sigma_sq=sigma_sq)
    elif bn == "ae":
        code = F.conv1d(enc_out, sd["bottleneck.linear.weight"], sd.get("bottleneck.linear.bias"))
        out.update(ze=code)
    else:
        raise ValueError(bn)
    out["encoding_bn"] = code
    quant = decoder_forward(sd, "decoder.", hps, wav, code, voice, jitter, geom.trim_dec_in[0],
                            geom.dec_in_len, geom.trim_ups_out, take_compat)
    o = geom.wav_out_off
    target = wav[:, o:o + geom.n_win][:, 1:]
    pred = quant[..., :-1]
    if bn == "vqvae-ema":
        loss = vqema_loss(pred, target, min_dist, hps.bn_vq_gamma, loss_mode)
    elif bn == "vqvae":
        loss = vq_loss(pred, target, ze, emb, min_ind, min_dist, hps.bn_vq_gamma)
    elif bn == "vae":
        loss, kl = sgvb_loss(pred, target, mu, sigma_sq, anneal, hps.bn_free_nats)
        out["kl"] = kl
    else:
        loss, norm = ae_loss(pred, target, code)
        out["norm"] = norm
    out.update(pred=pred, target=target, loss=loss, quant=quant)
    return out


# --------------------------------------------------------------------------------------
# Adam (third-party torch.optim.Adam, defaults betas=(0.9,0.999) eps=1e-8; checkpoint.py:49-50)
# --------------------------------------------------------------------------------------
def adam_step(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    """One Adam update restated from the published algorithm (Kingma & Ba 2015, Alg. 1) in
    the form torch.optim.Adam uses: denom = sqrt(v)/sqrt(1-b2^t) + eps."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * m / denom, m, v
