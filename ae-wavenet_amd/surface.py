"""Shared machinery of the drop-in nn.Module surfaces (see autoencoder_model.py and
mfcc_inverter.py in this package).

The reference harness drives a model through (chassis.py:151-171, checkpoint.py:35-67):
    model = Model(hps); model.get_input_size(w); model.override(w); model.mfcc
    optim = Adam(model.parameters()); model.to(device); model.train()
    quant, target, loss = model.run(wav, mel, voice, jitter); loss.backward(); optim.step()
    model.objective.metrics, model.encoder.metrics, model.bn_type,
    model.bottleneck.update_codebook(), model.init_codebook(...)

Parameters are nn.Parameters whose storage is a view into the engine's flat fp32 buffer and
whose .grad is a view into the flat gradient buffer, so `loss.backward()` (one
autograd.Function around the whole HIP training step) fills every .grad without per-parameter
autograd traffic, and any torch optimizer — or the fused `optim.FusedAdam` — can step them.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import nn

from . import _lib as L
from . import geometry as G
from .model import TrainEngine


class _StepFn(torch.autograd.Function):
    """loss = f(parameters, batch) with a hand-written backward (the bwd plan)."""

    @staticmethod
    def forward(ctx, anchor, owner):
        ctx.owner = owner
        dp = owner._dp
        if dp is not None and dp.sharded and not dp._solo():
            loss = dp.forward(owner._engine)                 # the decoder's parameter all-gather stays in flight under the encoder
        else:
            loss = owner._engine.forward(owner._ema_allreduce)
        return loss.clone()

    @staticmethod
    def backward(ctx, g):
        # g = d(L)/d(loss) from autograd ((loss * k).backward(), loss scaling, ...).  It is copied device-side into
        # the engine's `gmul` scalar, which the ops where the loss gradient enters the backward plan multiply in, so
        # every .grad carries it.  The backward plan WRITES the flat gradient buffer; nn.Module semantics (a second
        # backward() without zero_grad() adds to .grad) are kept by carrying the previous values over explicitly - a
        # path the reference harness never takes (it zeroes every step, chassis.py:157-160; FusedAdam.zero_grad() /
        # zero_grad(set_to_none=True) cost nothing here).
        owner = ctx.owner
        owner._engine.set_upstream_grad(g)
        dp = owner._dp
        prev = owner._grads_carried()
        if dp is not None and dp.sharded and not dp._solo():
            dp.backward_exchange(owner._engine, dp.bf16_grads)   # reduce-scatter issued under the encoder backward
        else:
            if dp is not None:
                dp.allreduce_kl(owner._engine)               # VAE: the clamp's gate sees the global KL
            owner._engine.backward()
        owner._after_backward(g)
        if prev is not None:
            eng = owner._engine
            eng.ps.grads[:eng.ps.numel].add_(prev)
        # (the anchor's "gradient": a cached zero, not a fill kernel per step)
        z = getattr(owner, "_anchor_zero", None)
        if z is None or z.device != owner._anchor.device:
            z = owner._anchor_zero = torch.zeros_like(owner._anchor)
        return z, None


class Objective:
    """Stand-in for the reference loss modules' attribute surface (metrics dict,
    update_anneal_weight, free_nats, anneal_weight; chassis.py:124,149,215-220)."""

    def __init__(self, owner):
        self._owner = owner
        self.metrics: Dict[str, torch.Tensor] = {}
        self.free_nats = torch.tensor(float(getattr(owner.hps, "bn_free_nats", 0.0)))
        self.anneal_weight = torch.tensor(0.0)

    def update_anneal_weight(self, w):
        self.anneal_weight = torch.tensor(float(w))
        eng = self._owner._engine
        if eng is not None and eng.bn_type == "vae":
            eng.set_anneal_weight(float(w))


class _EncoderFacade(nn.Module):
    def __init__(self):
        super().__init__()
        self.metrics: Dict[str, torch.Tensor] = {}


class _BottleneckFacade(nn.Module):
    def __init__(self, owner):
        super().__init__()
        self.__dict__["_owner"] = owner

    def update_codebook(self):
        """vqema_bn.py:216-222 (called by chassis.py:175-176)."""
        eng = self._owner._engine
        if eng is not None and eng.bn_type == "vqvae-ema":
            eng.update_codebook()


class HipModelBase(nn.Module):
    """Common implementation; subclasses set `kind` and the parameter prefix layout."""

    def __init__(self, hps, kind: str, loss_mode: str = "intended", take_compat: bool = False,
                 update_codebook_every_step: bool = True, n_mel: Optional[int] = None):
        super().__init__()
        if hps.global_model != kind:
            hps = type(hps)(hps)
            hps["global_model"] = kind
        self.hps = hps
        self.kind = kind
        self.bn_type = hps.bn_type if kind == "autoencoder" else "none"
        self._opts = dict(loss_mode=loss_mode, take_compat=take_compat,
                          update_codebook_every_step=update_codebook_every_step, n_mel=n_mel)
        self.window_batch_size = hps.n_win_batch
        self._engine: Optional[TrainEngine] = None
        self._engines: Dict[int, TrainEngine] = {}      # engines by batch size (train B, sampling B = 1, ...)
        self._opt_carry = None                           # (step, exp_avg flat, exp_avg_sq flat) while no engine holds them
        self._grads_cleared = True                       # no backward yet / FusedAdam.zero_grad() since the last one
        self._opt_carry_partial = False                  # carry saved from a sharded DP engine without a moment gather
        self._weights_epoch = 0                          # bumped whenever parameter values change behind torch's back
        self._device = torch.device("cpu")
        self._pending_state: Optional[Dict[str, torch.Tensor]] = None
        self._ema_allreduce = None
        self._dp = None
        self.objective = Objective(self)
        # inference surface (chassis.py:296, 313: model.wavenet.set_n_replicas / .n_quant; `model.wavenet` of the
        # MFCC inverter is the model itself here)
        self.n_quant, self.n_replicas = hps.n_quant, 1
        self.sample_seed = 0
        self._sampler = None
        self._anchor = torch.zeros((), requires_grad=True)
        # geometry attributes of the reference classes (autoencoder_model.py:119-146,
        # mfcc_inverter.py:38-65)
        self._set_geometry(self.window_batch_size)
        # Parameters exist from construction (Checkpoint builds Adam before .to(device),
        # checkpoint.py:48-50); they are re-homed into the engine's flat buffer on first use.
        self._make_cpu_params()

    def set_n_replicas(self, n_replicas):                       # wavenet.py:296-297
        self.n_replicas = int(n_replicas)

    # ---- geometry / harness queries ------------------------------------------------------
    def _set_geometry(self, w):
        g = G.model_geometry(self.hps, self.kind == "autoencoder", w)
        self.geom = g
        self.enc_in_len, self.enc_in_mel_len, self.embed_len = g.enc_in_len, g.mel_len, g.embed_len
        self.dec_in_len = g.dec_in_len
        self.trim_dec_in = torch.tensor(g.trim_dec_in)
        self.trim_dec_out = torch.tensor(g.trim_dec_out)
        self.trim_ups_out = torch.tensor(g.trim_ups_out)

    def get_input_size(self, output_size):
        """wav samples per window (wavenet.py:287-294 via checkpoint.py:40)."""
        return G.input_size(self.hps, self.kind == "autoencoder", output_size)

    def override(self, n_win_batch=None):
        """mfcc_inverter.py:30-35 (checkpoint.py:46)."""
        if n_win_batch is not None and n_win_batch != self.window_batch_size:
            self.window_batch_size = n_win_batch
            self._set_geometry(n_win_batch)
            self._drop_engine()

    # ---- parameters --------------------------------------------------------------------------
    def _specs(self):
        from .engine import bottleneck_param_specs, decoder_param_specs, encoder_param_specs
        h = self.hps
        if self.kind == "autoencoder":
            n_mel = self._opts["n_mel"] or 3 * h.n_mfcc
            return (encoder_param_specs(n_mel, h.enc_n_out) + bottleneck_param_specs(h)
                    + decoder_param_specs(h, h.n_lc_in, "decoder."))
        return decoder_param_specs(h, h.n_lc_in, "wavenet.")

    def _make_cpu_params(self):
        """Xavier-uniform weights, zero biases (netmisc.py:10-14); VQ codebooks with the
        reference's gains (vq_bn.py:21)."""
        self._pnames = []
        for name, shape in self._specs():
            t = torch.empty(shape)
            if len(shape) >= 2:
                nn.init.xavier_uniform_(t)
            else:
                t.zero_()
            pname = name.replace(".", "__")
            self.register_parameter(pname, nn.Parameter(t))
            self._pnames.append((name, pname))
        if self.bn_type == "vqvae-ema":
            K, d = self.hps.bn_vq_n_embed, self.hps.bn_n_out
            emb = torch.empty(K, d)
            nn.init.xavier_uniform_(emb, gain=10)                        # vqema_bn.py:97
            comp = 1.0 - self.hps.bn_vq_ema_gamma
            self.register_buffer("bn_emb", emb)
            self.register_buffer("bn_ema_numer", emb * comp)             # vqema_bn.py:117-118
            self.register_buffer("bn_ema_denom", torch.full((K,), comp))
            self.register_buffer("bn_ind_hist", torch.zeros(K))

    # nn.Module hooks so that state_dict / named_parameters use the reference's dotted names
    def named_parameters(self, prefix="", recurse=True, remove_duplicate=True):
        for name, pname in self._pnames:
            yield (prefix + ("." if prefix else "") + name), self._parameters[pname]

    def parameters(self, recurse=True):
        for _, p in self.named_parameters():
            yield p

    _BUF_NAMES = {"bn_emb": "bottleneck.emb", "bn_ema_numer": "bottleneck.ema_numer",
                  "bn_ema_denom": "bottleneck.ema_denom", "bn_ind_hist": "bottleneck.ind_hist"}

    def _dp_finish(self):
        """Data parallel, sharded schedule: the parameter all-gathers of the last optimizer step may still be in
        flight on the collective stream - everything that reads the parameters outside run() waits for them first."""
        if self._dp is not None:
            self._dp.finish()

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        self._dp_finish()
        self._sync_buffers_from_engine()
        sd = destination if destination is not None else {}
        for name, p in self.named_parameters():
            sd[prefix + name] = p if keep_vars else p.detach()
        for b, nm in self._BUF_NAMES.items():
            if b in self._buffers:
                sd[prefix + nm] = self._buffers[b]
        return sd

    def load_state_dict(self, state_dict, strict=True, assign=False):
        own = dict(self.named_parameters())
        missing = [k for k in own if k not in state_dict]
        unexpected = []
        inv = {v: k for k, v in self._BUF_NAMES.items()}
        with torch.no_grad():
            for k, v in state_dict.items():
                if k in own:
                    own[k].copy_(v.reshape(own[k].shape))
                elif k in inv and inv[k] in self._buffers:
                    self._buffers[inv[k]].copy_(v)
                else:
                    unexpected.append(k)
        self._push_buffers_to_engine()
        self._weights_epoch += 1
        if strict and (missing or [u for u in unexpected if "_lead" not in u and "eye" not in u
                                   and "residual_offsets" not in u]):
            raise RuntimeError(f"state_dict mismatch: missing {missing}, unexpected {unexpected}")
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    # ---- device placement / engine ---------------------------------------------------------
    def _apply(self, fn, recurse=True):
        # .to(device) / .float() ...: let nn.Module move the parameter tensors (views into the
        # engine's flat buffer are copied out by this), then drop the engines; one is rebuilt
        # lazily on the next run() and the values (and the Adam moments) are copied back in.
        if self._engine is not None:
            self._dp_finish()
            self._sync_buffers_from_engine()
            self._save_opt_carry()
        out = super()._apply(fn, recurse)
        self._engine = None
        self._engines = {}
        self._device = next(iter(self._parameters.values())).device
        if self._opt_carry is not None:
            st, m, v = self._opt_carry
            self._opt_carry = (st, m.to(self._device), v.to(self._device))
        self._anchor = torch.zeros((), requires_grad=True, device=self._device)
        return out

    def _drop_engine(self):
        if self._engine is not None:
            self._dp_finish()
            self._pull_params_to_cpu()
            self._save_opt_carry()
        self._engine = None
        self._engines = {}

    def _save_opt_carry(self):
        """Adam moments / step count live in the engine's flat buffers; keep them when the engine goes away
        (model.to(), override(), a different batch size, sample()), so optimizer state survives exactly like
        torch.optim.Adam's per-parameter state does in the reference's save / restore flow (checkpoint.py:82-102)."""
        eng = self._engine
        if eng is not None and (eng.step_count > 0 or self._opt_carry is None):
            n = eng.ps.numel
            self._opt_carry = (eng.step_count, eng.adam_m[:n].detach().clone(), eng.adam_v[:n].detach().clone())
            # sharded data parallel: unless the moments were just gathered this rank's copy is valid for its own shards
            # only - enough to continue training on a rebuilt engine (same shard layout), not to write a checkpoint
            self._opt_carry_partial = self._dp is not None and not self._dp.moments_complete(eng)

    def _opt_state_flat(self):
        """(step, exp_avg flat, exp_avg_sq flat) from the live engine, else from the carry, else None."""
        eng = self._engine
        if eng is not None:
            if self._dp is not None and not self._dp.moments_complete(eng):
                raise L.AewError("the Adam moments are sharded across the data-parallel ranks (each rank holds its own "
                                 "1/world): call dp.sync_optimizer_state(model) - or checkpoint.save(...) - on ALL ranks "
                                 "before reading optimizer state on any of them")
            n = eng.ps.numel
            return eng.step_count, eng.adam_m[:n], eng.adam_v[:n]
        if self._opt_carry is not None and getattr(self, "_opt_carry_partial", False):
            raise L.AewError("the carried Adam moments were saved from a sharded data-parallel engine without "
                             "dp.sync_optimizer_state(model): they are complete for this rank's shards only")
        return self._opt_carry

    def _pull_params_to_cpu(self):
        eng = self._engine
        self._dp_finish()
        self._sync_buffers_from_engine()
        with torch.no_grad():
            for name, pname in self._pnames:
                self._parameters[pname].data = eng.ps.view(name).detach().clone()

    def _sync_buffers_from_engine(self):
        eng = self._engine
        if eng is not None and eng.bn_type == "vqvae-ema":
            self._buffers["bn_emb"] = eng.emb.detach().clone()
            self._buffers["bn_ema_numer"] = eng.ema_numer.detach().clone()
            self._buffers["bn_ema_denom"] = eng.ema_denom.detach().clone()
            self._buffers["bn_ind_hist"] = eng.ind_hist.detach().clone()

    def _push_buffers_to_engine(self):
        eng = self._engine
        if eng is not None and eng.bn_type == "vqvae-ema":
            eng.emb.copy_(self._buffers["bn_emb"])
            eng.ema_numer.copy_(self._buffers["bn_ema_numer"])
            eng.ema_denom.copy_(self._buffers["bn_ema_denom"])
            eng.ind_hist.copy_(self._buffers["bn_ind_hist"])

    def _ensure_engine(self, B: int):
        eng = self._engine
        if eng is not None and eng.B == B:
            return eng
        if self._device.type != "cuda":
            raise L.AewError("this model executes only through the HIP library on an MI355X; "
                             "move it to a cuda device first (no CPU execution path exists)")
        L.load()
        if eng is not None:
            self._dp_finish()
            # another batch size (e.g. sample() uses B = 1 between training steps): parameters, EMA buffers and
            # Adam moments move to the other engine; engines are kept per batch size so that switching back does
            # not rebuild plans and graphs
            self._pull_params_to_cpu()
            self._save_opt_carry()
        new = self._engines.get(B)
        if new is None:
            new = TrainEngine(self.hps, B, self._device, n_win=self.window_batch_size, **self._opts)
            self._engines[B] = new
        eng = new
        # re-home the parameters into the flat buffer (values preserved)
        with torch.no_grad():
            for name, pname in self._pnames:
                p = self._parameters[pname]
                view = eng.ps.view(name)
                view.copy_(p.data.to(self._device))
                p.data = view
                p.grad = eng.ps.view(name, grad=True)
        self._grads_cleared = True                               # gradients do not move between engines
        self._engine = eng
        self._weights_epoch += 1
        self._push_buffers_to_engine()
        if self._opt_carry is not None:
            st, m, v = self._opt_carry
            n = eng.ps.numel
            eng.adam_m[:n].copy_(m)
            eng.adam_v[:n].copy_(v)
            eng.step_count = int(st)
        if self._dp is not None:
            self._dp.prepare_vae(eng)
        if eng.bn_type == "vae":
            eng.set_anneal_weight(float(self.objective.anneal_weight))
        return eng

    # ---- the hot path ----------------------------------------------------------------------
    def run(self, wav, mel, voice, jitter, eps=None):
        """(wav, mel, voice, jitter) -> (pred, target, loss)   [chassis.py:152]

        wav (B, enc_in_len) float32 holding mu-law ints; mel (B, n_mel, frames) float32;
        voice (B,) int64; jitter (B, >= embed_len) int64.  pred (B, Q, w-1) logits, target
        (B, w-1), loss scalar with a grad_fn whose backward fills every parameter's .grad."""
        B = wav.shape[0]
        eng = self._ensure_engine(B)
        if self._dp is not None and not (self._dp.sharded and not self._dp._solo()):
            self._dp.finish()                                # (sharded: _StepFn.forward waits region by region)
        eng.set_inputs(wav, mel, voice, jitter, eps=eps)
        loss = _StepFn.apply(self._anchor, self)
        w, g = eng.n_win, eng.geom
        pred = eng.logits()[:, :w - 1, :].permute(0, 2, 1)
        target = wav[:, g.wav_out_off + 1: g.wav_out_off + w]
        self._fill_forward_metrics(eng)
        return pred, target, loss

    def forward(self, wav, mel, voice, jitter):
        """train(): logits (B, Q, w) for the batch (teacher-forced).  eval(): the reference's inference call
        (mfcc_inverter.py:80-85 -> WaveNet.forward_test, wavenet.py:367-531): generates `n_replicas` continuations
        of the single input window and returns (1 + n_replicas, dec_in_len) mu-law values, row 0 = the input."""
        if not self.training:
            return self.sample(wav, mel, voice, jitter)
        eng = self._ensure_engine(wav.shape[0])
        self._dp_finish()
        eng.set_inputs(wav, mel, voice, jitter)
        eng.forward(self._ema_allreduce)
        return eng.logits().permute(0, 2, 1)

    def sample(self, wav, mel, voice, jitter, n_prime: Optional[int] = None, seed: Optional[int] = None):
        """Autoregressive generation on the persistent-kernel sampler (sampler.py).  The first n_prime positions of
        the decoder window are fed from `wav` (default: receptive field + 1, like the reference, which starts
        drawing at base_global_rf, wavenet.py:423-431); the rest are drawn.  One input window (B = 1); replicas are
        parallel streams (wavenet.py:378-381)."""
        from . import sampler as S
        if wav.shape[0] != 1:
            raise L.AewError("sampling takes one window; replicas come from set_n_replicas()")
        R = max(1, int(self.n_replicas))
        self._dp_finish()
        with torch.no_grad():
            eng = self._ensure_engine(1)
            eng.set_inputs(wav, mel, voice, jitter)
            cond, bias = eng.conditioning()
            # the sampler keeps its own packed copy of the weights: re-pack whenever they may have changed.  FusedAdam
            # updates parameters by raw pointer (no tensor _version bump), hence the explicit counters
            stamp = (eng.serial, eng.ps.params._version, eng.weights_version, self._weights_epoch)
            if self._sampler is None or self._sampler[0] != stamp:
                self._sampler = (stamp, S.from_engine(eng))
            smp = self._sampler[1]
            g = eng.geom
            T, rf = g.dec_in_len, smp.g.rf()
            n16 = (R + 15) // 16 * 16
            given = wav[:, g.trim_dec_in[0]:g.trim_dec_in[0] + T].to(torch.int32)
            forced = given.expand(n16, -1).clone()
            forced[:, min(T, rf + 1 if n_prime is None else max(1, n_prime)):] = -1
            if seed is None:
                seed, self.sample_seed = self.sample_seed, self.sample_seed + 1
            out, _ = smp.generate(cond.expand(n16, -1, -1).contiguous(), bias.expand(n16, -1, -1).contiguous(),
                                  forced.contiguous(), seed=seed)
            return torch.cat([given, out[:R]], 0).float()

    def _grads_carried(self):
        """What the coming backward has to ADD to the gradients it writes: None when the gradients were cleared
        (FusedAdam.zero_grad(), or every .grad is None as zero_grad(set_to_none=True) leaves them), else a copy of the
        current values (zeros where a single .grad is None) - torch's accumulate-into-.grad rule."""
        eng = self._engine
        cleared, self._grads_cleared = self._grads_cleared, False
        if cleared:
            return None
        grads = [(name, self._parameters[pname].grad) for name, pname in self._pnames]
        if all(gr is None for _, gr in grads):
            return None
        if self._dp is not None and self._dp.sharded and not self._dp._solo():
            raise L.AewError("backward() onto existing gradients (no zero_grad() since the last backward) is not supported "
                             "under the sharded data-parallel schedule: each rank holds the reduced gradient of its own "
                             "shards only")
        prev = eng.ps.grads[:eng.ps.numel].clone()
        for name, gr in grads:
            view = eng.ps.view(name, grad=True)
            o = (view.data_ptr() - eng.ps.grads.data_ptr()) // 4
            if gr is None:
                prev[o:o + view.numel()].zero_()
            elif gr.data_ptr() != view.data_ptr():               # somebody assigned their own tensor
                prev[o:o + view.numel()].copy_(gr.detach().reshape(-1).to(prev))
        return prev

    def _after_backward(self, g):
        eng = self._engine
        # re-attach .grad views (optim.zero_grad(set_to_none=True) detaches them)
        for name, pname in self._pnames:
            self._parameters[pname].grad = eng.ps.view(name, grad=True)
        if self._dp is not None and not self._dp.sharded:
            self._dp.allreduce_grads(eng)
        # gradient statistics: AEW_OP_MOMENTS ops on a side lane of the backward plan (views, no kernels here)
        m = self.objective.metrics
        gs = eng.gstat
        if self.kind == "autoencoder":
            m["mel_grad_sd"], m["bn_grad_sd"] = gs[1], gs[5]
        else:
            m["mel_grad_sd"], m["mel_grad_mean"] = gs[1], gs[0]

    def _az_numel(self, eng):
        key = (eng.serial,)
        if getattr(self, "_az_cache", (None,))[0] != key:
            n = [eng.B * eng.geom.enc_lens[i + 1] * self.hps.enc_n_out for i in range(9)]
            self._az_cache = (key, torch.tensor(n, dtype=torch.float64, device=eng.enc.zero_cnt.device))
        return self._az_cache[1]

    def _fill_forward_metrics(self, eng):
        w, B = eng.n_win, eng.B
        n_pos = B * (w - 1)
        m = self.objective.metrics
        mb = eng.met_buf                                           # the "metrics" reduction op of the forward plan
        m["rec"] = mb[1]
        self.tprb_m = mb[2]                                        # chassis.py:266-270
        dg, pk = eng.diag, eng.diag_pk                             # AEW_OP_VQ_DIAG ops of the forward plan (side lanes)
        m["pk_m"], m["pk_sd"], m["pk_nuq"] = pk[6], pk[7], pk[8]    # vqema_bn.py:261-263
        if eng.bn_type in ("vqvae-ema", "vqvae"):
            m["com"] = mb[3]
            m["min_ze"], m["max_ze"], m["min_emb"], m["max_emb"] = dg[0], dg[1], dg[2], dg[3]   # vqema_bn.py:254-257
            if eng.bn_type == "vqvae-ema":
                m["hst_ent"], m["nunq"] = dg[4], dg[5]              # vqema_bn.py:258-260
            else:
                m["nunq"] = eng.ind[:eng.Q].unique().numel()
        elif eng.bn_type == "vae":
            m["kl_div_loss"], m["log_pred_loss"] = eng.loss_buf[2], m["rec"]
        elif eng.bn_type == "ae":
            m["norm"] = eng.loss_buf[2]
        if eng.enc is not None and hasattr(self, "encoder"):
            az = eng.enc.zero_cnt[:9].double() / self._az_numel(eng)   # one division; the entries are views
            for i in range(9):
                self.encoder.metrics[f"enc_az_{i}"] = az[i]
