"""Training-step engine behind the drop-in module surfaces.

`TrainEngine` owns the workspace, the flat parameter / gradient buffers and the forward /
backward / optimizer plans for one (model, batch size, window).  `autoencoder_model.AutoEncoder`
and `mfcc_inverter.MfccInverter` wrap it in the reference's nn.Module surface
(autoencoder_model.py:206-259, mfcc_inverter.py:89-108); `bench.py` drives it directly.
"""
from __future__ import annotations

import os

import math
from typing import Dict, List, Optional

import torch

from . import _lib as L
from . import geometry as G
from .engine import (BF, F3, DecoderPlan, EncoderPlan, Packer, ParamStore, TAG_ADAM, TAG_ENC, det_colsum, exact_split_args,
                     TAG_LOSS, TAG_MISC, TAG_PACK, TAG_VQ, bottleneck_param_specs,
                     decoder_param_specs, encoder_param_specs)
from .plan import CopyTableBuilder, Mat, Plan, Workspace, insert_nt_chains, make_nt, make_tn, null_view, ru, split_small_nt

MEAN_LOSS = {"none": True, "ae": True, "vae": True, "vqvae": False, "vqvae-ema": False}


_SERIAL = [0]


class TrainEngine:
    """kind: 'autoencoder' | 'mfcc_inverter'.

    loss_mode (vqvae-ema only): 'intended' = rec.sum() + com.sum() (vqema_bn.py:244);
    'head' = commitment only, the debug state HEAD ships (vqema_bn.py:246, SURVEY C-3).
    take_compat: reproduce the reference's torch.take jitter gather (SURVEY C-1).
    update_codebook_every_step: refresh emb from the EMA statistics each step (standard
    VQ-VAE-EMA); the reference only refreshes at global_step == 10000 (chassis.py:175-176).
    """
    DIAG_LANE = 3             # codebook diagnostics (behind vq.ema on the same lane; read at the end of the forward)
    LANE_PACK_DEC = 1         # side lane of the decoder's forward-layout weight pack (joined by the end of fwd_a) ...
    pack_dec_late = True      # the decoder's weight pack at the head of fwd_b instead of fwd_a (see build: data-parallel overlap)
    small_split = 0           # > 0: bf16 NT launches of few 64 x 64 blocks contract K as up to 8 ranges on separate workgroups, as
                              # long as ranges x blocks stays at or under this (aew_gemm_nt_t.k_split hint, plan.split_small_nt)
    merge_packs = None        # True: all weight-layout packs of a step as ONE copy-table launch at the head of fwd_a (three launches
                              # fewer: 6.837 -> 6.802 ms per step interleaved, r05).  None = on unless the process is a rank of a
                              # data-parallel group: there the decoder's pack stays at the head of fwd_b (pack_dec_late), so that
                              # the decoder's parameter all-gather can run under the encoder forward (dp.DataParallel.forward)
    graph_lanes = 2           # lane mode of graph captures when the process-wide mode is 0 (aew_set_lanes): 2 = lanes 4 / 5 are
                              # branches, everything else in plan order.  Eager runs stay serial (one cross-stream edge costs
                              # more there than the branch returns: 7.27 vs 6.90 ms per step).  0: never
    LANE_PACK_LATE = 2        # ... and of the backward-layout pack at the head of fwd_b (joined by its end).  4 / 5: the lanes
                              # aew_set_lanes(2) honours alone (A/B: each a single fork / join, profiles/r04_notes.md §16)
    PACK_LANE = 2             # side lane of the forward-layout weight pack (layers 1.. of the encoder, biases, bottleneck)
    nt_chain = 64             # chained NT launches (AEW_OP_NT_CHAIN, csrc/aew_chain.hip): runs of dependent NT ops of the decoder
                              # FORWARD - the gated stack's G1 / G2 ops (wavenet.py:354-357), the post network's pair - as
                              # launches of up to this many stages with tile-granular hand-off (64: the whole stack is ONE
                              # launch of 27 856 tiles instead of 39; 2: the pair of a layer; 0: one launch per op).  Bit-identical
                              # results; per-op timing (aew_timing_enable(1)) always runs the ops one by one.  Measured,
                              # interleaved on one box: fwd_b plan 2.089 -> 1.911 ms, step 6.865 -> 6.748 ms (profiles/r05_notes.md)
    nt_chain_bwd = 0          # the same for the BACKWARD's d.post / dz / dx run.  Measured neutral to slightly negative (bwd plan
                              # 4.085 -> 4.134 ms): a dz stage is 320-448 tiles - fewer than the 512 resident slots - so its
                              # consumers (dx) are resident before it has finished and 11 000 of the launch's 20 432 tiles spin
                              # on a counter, where the forward's stages (640-896 tiles) leave 900 of 27 856 waiting.  Option
    nt_chain_bwd_phase = 0    # with nt_chain_bwd = 2: 0 pairs (dz.l, dx.l); 1 pairs (dx.l, dz.l-1) - the dependency whose producer stage is
                              # larger than the 512 resident tiles, so that its consumers find it finished
    nt_chain_max_stage_tiles = 2048   # runs whose stages average more tiles than this (four waves of the 512 resident slots) stay
                              # stand-alone launches: the deep decoder (30 x 512, 64k windows: 8 224 tiles per gated stage) loses
                              # 1 % to chaining (62.9 vs 62.2 ms per step) - nothing to gain at 16 tile waves per launch, and the
                              # hand-off (write-through stores, one acquire per tile) is not free.  0 = no limit
    nt_chain_force = False    # tests: chain also the sizes the stand-alone launcher runs on its small-launch shapes
    nt_chain_flags = 0        # aew_nt_chain_t.flags (measurement aids)
    nt_chain_spin_max = 0     # polls before a hand-off wait gives up (0: the library's 1 << 18, ~0.3 s)
    diag_early = True         # per-step diagnostics placed where their inputs become final (False: at the tail of the
                              # forward plan; A/B: 8.02 -> 7.99 ms per step)

    def __init__(self, hps, B: int, device, n_mel: Optional[int] = None, loss_mode: str = "intended",
                 take_compat: bool = False, update_codebook_every_step: bool = True, impl: int = 0,
                 n_win: Optional[int] = None, use_graphs: Optional[bool] = None, wgrad_group: Optional[int] = None,
                 tuning=None):
        self.hps, self.B, self.impl = hps, B, impl
        _SERIAL[0] += 1
        self.serial = _SERIAL[0]          # unique per engine (id() can be recycled after garbage collection)
        self.weights_version = 0          # bumped by adam_step(): FusedAdam writes parameters by raw pointer
        # plans run as captured hipGraphs unless told otherwise.  An explicit constructor argument wins; the environment
        # (AEW_USE_GRAPHS=0: eager stream launches, AEW_WGRAD_GROUP=n: layers per grouped wgrad launch, 0 = round 2's one
        # op per matrix) only supplies the default of an argument left at None (A/B and bisecting aid).
        if use_graphs is None:
            use_graphs = os.environ.get("AEW_USE_GRAPHS", "1") == "1"
        self.use_graphs = bool(use_graphs)
        if wgrad_group is None:
            wgrad_group = int(os.environ["AEW_WGRAD_GROUP"]) if os.environ.get("AEW_WGRAD_GROUP") is not None \
                else DecoderPlan.wgrad_group
        self.wgrad_group = int(wgrad_group)
        self.tuning = tuning              # _lib.Tuning (aew_tuning_t) for this engine's launches, or None
        self.merge_packs_auto = self.merge_packs is None          # (dp.DataParallel warns when an automatic choice went the
        if self.merge_packs is None:                              #  wrong way: engine built before init_process_group)
            import torch.distributed as _dist
            self.merge_packs = not (_dist.is_available() and _dist.is_initialized() and _dist.get_world_size() > 1)
        self.kind = hps.global_model
        self.bn_type = hps.bn_type if self.kind == "autoencoder" else "none"
        self.loss_mode, self.take_compat = loss_mode, take_compat
        self.update_codebook_every_step = update_codebook_every_step
        self.device = torch.device(device)
        self.n_win = n_win or hps.n_win_batch
        with_enc = self.kind == "autoencoder"
        self.geom = G.model_geometry(hps, with_enc, self.n_win)
        self.n_mel = n_mel if n_mel is not None else hps.n_lc_in if not with_enc else 3 * hps.n_mfcc
        self.ws = ws = Workspace(self.device)
        g = self.geom
        # ---- parameters
        dec_pre = "decoder." if with_enc else "wavenet."
        specs = []
        if with_enc:
            specs += encoder_param_specs(self.n_mel, hps.enc_n_out)
            specs += bottleneck_param_specs(hps)
        specs += decoder_param_specs(hps, hps.n_lc_in, dec_pre)
        self.ps = ps = ParamStore(ws, specs)
        self.dec_pre = dec_pre
        # ---- static inputs
        self.in_wav = ws.alloc("in.wav", B * g.enc_in_len, torch.float32)[:B * g.enc_in_len].view(B, g.enc_in_len)
        self.in_mel = ws.alloc("in.mel", B * self.n_mel * g.mel_len, torch.float32)[:B * self.n_mel * g.mel_len] \
            .view(B, self.n_mel, g.mel_len)
        self.in_voice = ws.alloc("in.voice", B, torch.int64)[:B]
        self.in_jitter = ws.alloc("in.jitter", B * g.embed_len, torch.int64)[:B * g.embed_len].view(B, g.embed_len)
        self.loss_buf = ws.alloc("loss", 8, torch.float32)
        # upstream gradient d(L)/d(loss) of the current backward call (autograd's `g`): the ops where the loss
        # gradient enters the backward read it from device memory, so (loss * k).backward() scales every gradient
        # without a host sync and without touching the captured graphs
        self.gmul = ws.alloc("loss.gmul", 4, torch.float32)
        self.gmul[:1].fill_(1.0)
        # STICKY word of the chained launches (aew_nt_chain_t.sticky): no plan clears it; a hand-off wait that gave up leaves
        # (stage + 1) here, and the optimizer / EMA ops read it as their guard - the step does not reach the parameters
        self.chain_guard = ws.alloc("chain.guard", 4, torch.int32)
        # ---- tables
        # encoder / bottleneck tables stay on the main lane (their packed weights are needed at once and
        # their gradients come last); the decoder's are side-lane work: its weight pack overlaps the
        # (tiny-grid) encoder forward, its gradient unpack overlaps the encoder backward
        self.pack_tbl = CopyTableBuilder(ws, "tbl.pack")
        self.unpack_tbl = CopyTableBuilder(ws, "tbl.unpack")
        self.pack_dec = CopyTableBuilder(ws, "tbl.pack_dec")
        self.unpack_dec = CopyTableBuilder(ws, "tbl.unpack_dec")
        self.in_tbl = CopyTableBuilder(ws, "tbl.in")
        self.pack_late = CopyTableBuilder(ws, "tbl.pack_late")      # dgrad layouts of the encoder weights: backward only
        # the encoder's first layer waits for its own (small) weight pack only; the rest of the forward-layout packs
        # (57 MB of fp32 encoder weights, biases, bottleneck) run on a side lane under layer 0
        self.pack_first = CopyTableBuilder(ws, "tbl.pack_first") if with_enc else None
        self.pk = Packer(ps, self.pack_tbl, self.unpack_tbl, late_tbl=self.pack_late, first_tbl=self.pack_first)
        self.pk_dec = Packer(ps, self.pack_dec, self.unpack_dec)
        Mp = ru(self.n_mel, 64)
        self.mel_cl = Mat.new(ws, "mel_cl", B, g.mel_len, Mp, F3)
        self.in_tbl.add(self.in_mel.data_ptr(), self.mel_cl.ptr, [B, g.mel_len, self.n_mel],
                        [self.n_mel * g.mel_len, 1, g.mel_len], [self.mel_cl.bs, Mp, 1], F3, F3)
        # ---- sub-plans
        self.enc: Optional[EncoderPlan] = None
        if with_enc:
            self.enc = EncoderPlan(ws, ps, hps, g, B, self.n_mel, self.mel_cl, self.pk, impl, in_tbl=self.in_tbl,
                                   in_mel=self.in_mel, wgrad_group=self.wgrad_group)
            self._alloc_bottleneck()
            lc_src = self.code
        else:
            lc_src = self.mel_cl
        self.dec = DecoderPlan(ws, ps, hps, g, B, dec_pre, hps.n_lc_in, lc_src, self.in_wav, self.in_voice,
                               self.in_jitter, take_compat, self.pk_dec, impl, wgrad_group=self.wgrad_group)
        self.dec.unpack_early_tbl = CopyTableBuilder(ws, "tbl.unpack_dec_early")
        self._build()
        self.adam_state = None
        self.step_count = 0

    # --------------------------------------------------------------------------------------
    def _alloc_bottleneck(self):
        ws, hps, B, g = self.ws, self.hps, self.B, self.geom
        bn, E, d = self.bn_type, hps.enc_n_out, hps.bn_n_out
        self.d, self.dp = d, ru(d, 64)
        Ne = g.embed_len
        self.Q = B * Ne
        Ep = ru(E, 64)
        nlin = 2 * d if bn == "vae" else d
        self.nlin, self.nlin_p = nlin, ru(nlin, 64)
        self.lin = Mat.new(ws, "bn.lin", B, Ne, self.nlin_p, F3)          # ze / (mu|logvar)
        self.Wl = Mat.new(ws, "bn.wp.lin", 1, self.nlin_p, Ep, F3)
        self.WlT = Mat.new(ws, "bn.wp.linT", 1, Ep, self.nlin_p, F3)
        self.pk.rec("bottleneck.linear.weight", 0, [E, 1], [nlin, E], self.Wl, 0, [Ep, 1])
        self.pk.rec("bottleneck.linear.weight", 0, [E, 1], [nlin, E], self.WlT, 0, [1, self.nlin_p])
        self.dlin = Mat.new(ws, "bn.dlin", B, Ne, self.nlin_p, F3)
        if bn in ("vqvae-ema", "vqvae"):
            K = hps.bn_vq_n_embed
            self.K = K
            self.code = Mat.new(ws, "bn.zq", B, Ne, self.dp, F3)
            self.ind = ws.alloc("bn.ind", self.Q, torch.int64)
            self.min_dist = ws.alloc("bn.min_dist", self.Q, torch.float32)
            if bn == "vqvae-ema":
                self.emb = ws.alloc("bn.emb", K * d, torch.float32)[:K * d].view(K, d)
                self.ema_numer = ws.alloc("bn.ema_numer", K * d, torch.float32)[:K * d].view(K, d)
                self.ema_denom = ws.alloc("bn.ema_denom", K, torch.float32)[:K]
                # one buffer so that the data-parallel exchange of the EMA statistics is ONE all-reduce
                self.zn_sum = ws.alloc("bn.zn_sum", K * d + K, torch.float32)[:K * d + K]
                self.z_sum = self.zn_sum[:K * d].view(K, d)
                self.n_sum = self.zn_sum[K * d:]
                self.ind_hist = ws.alloc("bn.ind_hist", K, torch.float32)[:K]
            else:
                self.emb = self.ps.view("bottleneck.emb")
        elif bn == "vae":
            self.code = Mat.new(ws, "bn.sample", B, Ne, self.dp, F3)
            self.eps = ws.alloc("bn.eps", self.Q * d, torch.float32)[:self.Q * d].view(B, Ne, d)
            self.kl_terms = ws.alloc("bn.kl_terms", self.Q, torch.float32)
            self.anneal_weight = 0.0
            self.dp_world = 1                                                # set by dp.DataParallel (see set_anneal_weight)
            self.anneal_buf = ws.alloc("bn.anneal", 4, torch.float32)[:1]   # read by the loss op
            self.anneal_bwd = ws.alloc("bn.anneal_bwd", 4, torch.float32)[:1]   # read by the KL-gradient op
        elif bn == "ae":
            self.code = self.lin
            self.norm_terms = ws.alloc("bn.norm_terms", self.Q, torch.float32)
            bt = ws.alloc("bn.wp.bias", self.nlin_p, torch.float32)
            self.pack_tbl.add(self.ps.ptr("bottleneck.linear.bias"), bt.data_ptr(), [d], [1], [1], F3, F3)
            self.lin_bias = bt
        else:
            raise ValueError(f"unknown bn_type {bn}")

    # --------------------------------------------------------------------------------------
    def _build(self):
        ws, hps, B, g, ps, impl = self.ws, self.hps, self.B, self.geom, self.ps, self.impl
        bn = self.bn_type
        w = self.n_win
        n_pos = B * (w - 1)
        # ===== forward, part A: pack, inputs, encoder, bottleneck up to the EMA statistics
        self.fwd_a = fa = Plan("fwd_a")
        self.fwd_b = fb = Plan("fwd_b")
        pack_slot = len(fa.ops)
        self.in_tbl.emit(fa, "mel->channels-last")
        if self.enc is not None:
            self.enc.build_forward(fa, join_before_layer1=("lane", self.PACK_LANE) if self.pack_first is not None else False)
            E, Ep = hps.enc_n_out, ru(hps.enc_n_out, 64)
            y9 = self.enc.y[9]
            flags = L.EF_BIAS if bn == "ae" else 0
            skw = exact_split_args(ws, "bn.ks", EncoderPlan.k_split if impl == 0 else 0, E, Ep, g.embed_len * B, self.nlin_p)
            fa.add(L.OP_GEMM_NT, make_nt(F3, g.embed_len, ru(self.nlin, 4), self.nlin_p, B, [y9.seg(Ep)],
                                         self.Wl.ptr, flags=flags, out0=self.lin.view(),
                                         bias_ptr=self.lin_bias.data_ptr() if bn == "ae" else 0, impl=impl, **skw),
                   "bn.linear", TAG_VQ)
            if bn in ("vqvae-ema", "vqvae"):
                vq = L.VqNearest()
                vq.ze, vq.emb = self.lin.ptr, self.emb.data_ptr()
                vq.Q, vq.K, vq.d, vq.d_pitch = self.Q, self.K, self.d, self.nlin_p
                vq.metric = 0 if bn == "vqvae-ema" else 1
                vq.ind, vq.dist, vq.zq = self.ind.data_ptr(), self.min_dist.data_ptr(), self.code.ptr
                vq.n_split = max(1, min(16, self.K // 256))       # each block scans >= 256 codes (one per thread)
                if vq.n_split > 1:
                    vq.scratch = ws.alloc("bn.vq_part", 2 * self.Q * vq.n_split, torch.float32).data_ptr()
                assert self.dp == self.nlin_p
                fa.add(L.OP_VQ_NEAREST, vq, "vq.nearest", TAG_VQ)
                if bn == "vqvae-ema":
                    st = L.VqStats()
                    st.ze, st.ind = self.lin.ptr, self.ind.data_ptr()
                    st.Q, st.K, st.d, st.d_pitch = self.Q, self.K, self.d, self.nlin_p
                    st.z_sum, st.n_sum, st.hist = self.z_sum.data_ptr(), self.n_sum.data_ptr(), self.ind_hist.data_ptr()
                    fa.add(L.OP_VQ_STATS, st, "vq.stats", TAG_VQ)
                    # the diagnostics op (side lane of fwd_b) reads this step's LOCAL code counts; a data-parallel
                    # caller all-reduces zn_sum in place while fwd_b runs, so it gets its own copy
                    self.n_sum_diag = ws.alloc("bn.n_sum_diag", self.K, torch.float32)
                    ntbl = CopyTableBuilder(ws, "tbl.nsum")
                    ntbl.add(self.n_sum.data_ptr(), self.n_sum_diag.data_ptr(), [self.K], [1], [1], F3, F3)
                    ntbl.emit(fa, "n_sum -> diagnostics copy")
                    em = L.VqEma()
                    em.numer, em.denom = self.ema_numer.data_ptr(), self.ema_denom.data_ptr()
                    em.z_sum, em.n_sum, em.emb = self.z_sum.data_ptr(), self.n_sum.data_ptr(), self.emb.data_ptr()
                    em.K, em.d, em.update_codebook = self.K, self.d, 0
                    em.gamma = float(hps.bn_vq_ema_gamma)
                    em.gamma_comp = float(1.0 - hps.bn_vq_ema_gamma)
                    em.guard = self.chain_guard.data_ptr()
                    # EMA runs at the START of part B (after the optional cross-rank sum); the
                    # codebook refresh is deferred to after backward so that forward and backward
                    # of one step see the same emb
                    with fb.side(3):                           # nothing in the forward / backward reads the accumulators
                        fb.add(L.OP_VQ_EMA, em, "vq.ema", TAG_VQ)
            elif bn == "vae":
                va = self._vae_op(False)
                fa.add(L.OP_VAE, va, "vae.sample", TAG_VQ)
            elif bn == "ae":
                an = self._ae_norm_op(False)
                fa.add(L.OP_AE_NORM, an, "ae.norm", TAG_VQ)
        # ===== forward, part B: decoder + loss
        if not self.merge_packs:
            with fb.side(self.LANE_PACK_LATE):                 # only the backward reads these: hidden under the decoder
                self.pack_late.emit(fb, "pack weights (backward layouts)")
            for op in fb.ops[-1:] if self.pack_late.recs else []:
                op.tag = TAG_PACK
        # Per-step diagnostics (what the reference's loss modules report, vqema_bn.py:251-264, chassis.py:266-270) are
        # side-lane ops placed where their inputs become final, not after the plan's last op (a side op starts after
        # every main-lane op that precedes it in the plan: appended at the end they were ~0.09 ms of exposed tail):
        #   codebook / encoder-output statistics  right here, under the whole decoder forward
        #   peak log-probability statistics       after the logits GEMM, under the softmax
        #   rec / tprb_m / com                    after the softmax, under the loss reduction
        n_pos_m = B * (w - 1)
        self.diag = ws.alloc("diag.out", 16, torch.float32)              # [0..5] (vqema_bn.py:254-260)
        self.diag_pk = ws.alloc("diag.peak", 16, torch.float32)          # [6..8] (vqema_bn.py:261-263)
        self.diag_scratch = ws.alloc("scratch.diag", 512, torch.float32)
        self.diag_scratch_pk = ws.alloc("scratch.diag_pk", 512, torch.float32)   # (the two diagnostics ops may overlap on lanes)
        self.met_buf = ws.alloc("diag.metrics", 8, torch.float32)        # [1] rec, [2] tprb_m, [3] com
        if bn in ("vqvae-ema", "vqvae"):
            dv = L.VqDiag()
            dv.ze, dv.Q, dv.d, dv.d_pitch = self.lin.ptr, self.Q, hps.bn_n_out, self.lin.pitch
            dv.K = hps.bn_vq_n_embed
            dv.emb = self.emb.data_ptr() if bn == "vqvae-ema" else ps.ptr("bottleneck.emb")
            if bn == "vqvae-ema":
                dv.hist, dv.n_sum = self.ind_hist.data_ptr(), self.n_sum_diag.data_ptr()
            dv.B, dv.w, dv.n_quant = B, w, hps.n_quant
            dv.scratch, dv.out = self.diag_scratch.data_ptr(), self.diag.data_ptr()
            with fb.side(self.DIAG_LANE):                       # not lane 1: decoder layer 0 waits for that one
                fb.add(L.OP_VQ_DIAG, dv, "diagnostics (codebook)", TAG_LOSS)

        # peak statistics of the predicted distribution (vqema_bn.py:261-263).  With the reference's 256 classes the
        # softmax kernel has each row in registers and writes the per-position peak log-probability / arg-max next to the
        # nll; the diagnostics op then reduces 2 x 160 KB instead of reading the 41 MB of logits a second time
        fused_peak = hps.n_quant == 256 and self.dec.Qp == 256
        if fused_peak:
            self.peak_buf = ws.alloc("diag.peak_pos", B * w, torch.float32)
            self.amax_buf = ws.alloc("diag.amax_pos", B * w, torch.int32)
            self.dec.peak_ptrs = (self.peak_buf.data_ptr(), self.amax_buf.data_ptr())

        def peak_diag(plan):
            dg = L.VqDiag()
            lgm = self.dec.logits
            dg.logits, dg.bs, dg.pitch, dg.B, dg.w, dg.n_quant = lgm.ptr, lgm.bs, lgm.pitch, B, w, hps.n_quant
            dg.scratch, dg.out = self.diag_scratch_pk.data_ptr(), self.diag_pk.data_ptr()
            if fused_peak:
                dg.peak, dg.amax = self.dec.peak_ptrs
            with plan.side(1):
                plan.add(L.OP_VQ_DIAG, dg, "diagnostics (peak)", TAG_LOSS)

        def after_logits(plan):
            if not fused_peak:
                peak_diag(plan)

        def after_nll(plan):
            if fused_peak:
                peak_diag(plan)
            met = L.Reduce()
            mterms = [(self.dec.nll.data_ptr(), B * w, 1.0 / n_pos_m), (self.dec.ptgt.data_ptr(), B * w, 1.0 / n_pos_m)]
            if bn in ("vqvae-ema", "vqvae"):
                mterms.append((self.min_dist.data_ptr(), self.Q, float(hps.bn_vq_gamma) / self.Q))
            met.n_terms = len(mterms)
            for i, (p_, n_, s_) in enumerate(mterms):
                met.x[i], met.n[i], met.scale[i], met.post_scale[i] = p_, n_, s_, 1.0
            met.out = self.met_buf.data_ptr()
            with plan.side(1):
                plan.add(L.OP_REDUCE, met, "metrics", TAG_LOSS)

        if self.diag_early:
            self.dec.build_forward(fb, after_logits=after_logits, after_nll=after_nll)
        else:                                                  # A/B: all of them after the plan's last main-lane op
            self.dec.build_forward(fb)
            self._diag_tail = (after_logits, after_nll)
        # AEW_NT_CHAIN = "f" or "f,b": stages per chained launch of the forward / backward (A/B and bisecting aid)
        env = os.environ.get("AEW_NT_CHAIN")
        n_f, n_b = int(self.nt_chain), int(self.nt_chain_bwd)
        if env is not None and env.strip():
            try:
                v = [int(x) for x in env.split(",") if x.strip()]
            except ValueError:
                raise ValueError(f"AEW_NT_CHAIN={env!r}: expected 'f' or 'f,b' (stages per chained launch, forward / backward)")
            if v:
                n_f, n_b = v[0], (v[1] if len(v) > 1 else 0)
        self.nt_chain_used = n_f if impl == 0 else 0
        self.nt_chain_bwd_used = n_b if impl == 0 else 0
        ckw = dict(force=bool(self.nt_chain_force), flags=int(self.nt_chain_flags), max_stage_tiles=int(self.nt_chain_max_stage_tiles),
                   sticky_ptr=self.chain_guard.data_ptr(), tuning=self.tuning, spin_max=int(self.nt_chain_spin_max))
        if self.nt_chain_used >= 2:
            insert_nt_chains(fb, ws, "chain.fwd", lambda lab: lab.startswith(("G1.", "G2.", "post1", "post2")), max_len=self.nt_chain_used, **ckw)
        red = L.Reduce()
        nll_ptr = self.dec.nll.data_ptr()
        terms = []
        if bn in ("none",):
            terms = [(nll_ptr, B * w, 1.0 / n_pos)]
        elif bn == "vqvae-ema":
            com = (self.min_dist.data_ptr(), self.Q, float(hps.bn_vq_gamma))
            terms = [com] if self.loss_mode == "head" else [(nll_ptr, B * w, 1.0), com]
        elif bn == "vqvae":
            terms = [(nll_ptr, B * w, 1.0), (self.min_dist.data_ptr(), self.Q, 1.0 + float(hps.bn_vq_gamma))]
        elif bn == "vae":
            terms = [(nll_ptr, B * w, 1.0 / n_pos), (self.kl_terms.data_ptr(), self.Q, -0.5)]
        elif bn == "ae":
            terms = [(nll_ptr, B * w, 1.0 / n_pos), (self.norm_terms.data_ptr(), self.Q, 0.001 / self.Q)]
        red.n_terms = len(terms)
        for i, (p_, n_, s_) in enumerate(terms):
            red.x[i], red.n[i], red.scale[i] = p_, n_, s_
            red.post_scale[i] = 1.0
        if bn == "vae":
            red.clamp[1], red.clamp_min[1], red.post_scale[1] = 1, float(hps.bn_free_nats), 0.0
            red.post_scale_dev[1] = self.anneal_buf.data_ptr()
            self._red_index = len(fb.ops)
        red.out = self.loss_buf.data_ptr()
        fb.add(L.OP_REDUCE, red, "loss", TAG_LOSS)
        for hook in getattr(self, "_diag_tail", ()):
            hook(fb)
        # ===== backward
        self.bwd = bw = Plan("bwd")
        bw.zero(ws, "grads")
        mean = MEAN_LOSS[bn]
        nll_scale = (1.0 / n_pos) if mean else 1.0
        if bn == "vqvae-ema" and self.loss_mode == "head":
            nll_scale = 0.0
        self.dec.build_backward(bw, nll_scale)
        if self.nt_chain_bwd_used >= 2:
            heads = ("d.post2", "d.post1")[int(self.nt_chain_bwd_phase):]    # (phase 1: pairs are (dx.l, dz.l-1), not (dz.l, dx.l))
            insert_nt_chains(bw, ws, "chain.bwd", lambda lab: lab.startswith(heads + ("dz.", "dx.")),
                             max_len=self.nt_chain_bwd_used, **ckw)
        # gradient statistics of run() (autoencoder_model.py:252-257 mel_grad_sd / bn_grad_sd; mfcc_inverter.py:100-106
        # mel_grad_sd / mel_grad_mean): by-products of this backward, reduced on a side lane as soon as their input is
        # final (d(loss)/d(code) here, under the encoder backward)
        self.gstat = ws.alloc("diag.gstat", 8, torch.float32)            # [0:4] mel (mean, std, sum, sumsq), [4:8] bn

        def moments(mat, cols, slot, nm):
            mo = L.Moments()
            mo.x, mo.rows, mo.cols, mo.batch = mat.view(), mat.rows, cols, B
            mo.out = self.gstat.data_ptr() + 4 * slot
            with bw.side(1):
                bw.add(L.OP_MOMENTS, mo, f"grad stats ({nm})", TAG_LOSS)
        with bw.side(getattr(self.dec, "tail_lane_used", 0) or 1):                 # after every decoder wgrad on any side lane
            self.unpack_dec.emit(bw, "unpack grads (decoder)", join=True)
        if self.enc is not None:
            moments(self.dec.dlc_src, hps.bn_n_out, 4, "bn")
        if self.enc is not None:
            dcode = self.dec.dlc_src                      # d(loss)/d(code) [B][Ne][dp]
            Ep = ru(hps.enc_n_out, 64)
            y9 = self.enc.y[9]
            if bn in ("vqvae-ema", "vqvae"):
                vb = L.VqBwd()
                vb.ze, vb.emb, vb.ind, vb.dzq = self.lin.ptr, self.emb.data_ptr(), self.ind.data_ptr(), dcode.ptr
                vb.Q, vb.d, vb.d_pitch = self.Q, self.d, self.nlin_p
                vb.metric = 0 if bn == "vqvae-ema" else 1
                vb.coef = float(hps.bn_vq_gamma)
                vb.demb_coef = 1.0
                vb.dze = self.dlin.ptr
                vb.demb = ps.ptr("bottleneck.emb", True) if bn == "vqvae" else None
                vb.gmul = self.gmul.data_ptr()
                bw.add(L.OP_VQ_BWD, vb, "vq.bwd", TAG_VQ)
            elif bn == "vae":
                self._vae_bwd_index = len(bw.ops)
                bw.add(L.OP_VAE, self._vae_op(True, dcode), "vae.bwd", TAG_VQ)
            elif bn == "ae":
                bw.add(L.OP_AE_NORM, self._ae_norm_op(True, dcode), "ae.norm.bwd", TAG_VQ)
                cs = L.Colsum()
                cs.x = self.dlin.seg(64)
                cs.dtype, cs.M, cs.N, cs.batch = F3, g.embed_len, self.d, B
                cs.out, cs.out_bs, cs.accumulate = ps.ptr("bottleneck.linear.bias", True), 0, 1
                det_colsum(ws, cs, "det.db.bn")
                with bw.side():
                    bw.add(L.OP_COLSUM, cs, "db.bn", TAG_VQ)
            # linear wgrad / dgrad
            t = make_tn(F3, g.embed_len, B, self.nlin, self.nlin_p, self.dlin.seg(64), [y9.seg(Ep)], impl=impl)
            slabs = L.tn_slabs(t)
            gt = ws.alloc("bn.wg.lin", slabs * self.nlin_p * Ep, torch.float32)
            t.out, t.out_batch_stride = gt.data_ptr(), self.nlin_p * Ep
            with bw.side():
                bw.add(L.OP_GEMM_TN, t, "wgrad.bn.linear", TAG_VQ)
            self.pk.rec("bottleneck.linear.weight", 0, [hps.enc_n_out, 1], [self.nlin, hps.enc_n_out], None, 0,
                        [Ep, 1], g_ptr=gt.data_ptr(), slabs=slabs, slab_stride=self.nlin_p * Ep)
            bw.add(L.OP_GEMM_NT, make_nt(F3, g.embed_len, hps.enc_n_out, Ep, B, [self.dlin.seg(self.nlin_p)],
                                         self.WlT.ptr, flags=L.EF_OUT1_POS1, out0=self.enc.dy[9].view(),
                                         out1=self.enc.dpre[9].view(), aux1=self.enc.r[9].view(), impl=impl),
                   "d.bn.linear", TAG_VQ)
            self.enc.build_backward(bw, need_input_grad=True)
        # (mel gradient statistics: its input is the last thing the backward writes)
        moments(self.enc.dy[0] if self.enc is not None else self.dec.dlc_src, self.n_mel, 0, "mel")
        self.unpack_tbl.emit(bw, "unpack grads", join=True)        # reads the side-lane encoder wgrad slabs
        # launches of a few dozen 64 x 64 blocks (the upsampler / encoder data gradients): split-K hint
        self.small_split_made = []
        if int(self.small_split) > 0 and impl == 0:
            for pl in (fa, fb, bw):
                self.small_split_made += split_small_nt(pl, ws, "ks." + pl.name, int(self.small_split))
        # Deferred EMA (data parallel): the EMA accumulators are not read again before the codebook
        # refresh, so the cross-rank sum of z_sum | n_sum can run asynchronously under the whole
        # decoder forward + backward; fwd_b_noema / ema_plan are fwd_b without / only its leading vq.ema op
        # The decoder's weight layouts are packed at the head of fwd_b (behind vq.ema), not of fwd_a: nothing in fwd_a
        # reads a decoder parameter, so a data-parallel caller lets the all-gather of the updated decoder shards run
        # under the encoder forward and waits for it only between the two plans (forward(before_decoder=...)).  Serial
        # plans: the same launches in the same total time.
        if self.pack_dec_late and not self.merge_packs and self.pack_dec.recs:
            pkd = Plan("pack_dec")
            with pkd.side(self.LANE_PACK_DEC):
                self.pack_dec.emit(pkd, "pack weights (decoder)")
            at = 1 if (fb.labels and fb.labels[0] == "vq.ema") else 0
            for op in pkd.ops:
                op.tag = TAG_PACK
            fb.ops[at:at] = pkd.ops
            fb.labels[at:at] = pkd.labels
            fb.keep += pkd.keep
            self._pack_dec_emitted = True
        self.fwd_b_noema, self.ema_plan = None, None
        if fb.labels and fb.labels[0] == "vq.ema":
            self.fwd_b_noema, self.ema_plan = Plan("fwd_b_noema"), Plan("ema")
            self.fwd_b_noema.ops, self.fwd_b_noema.labels = fb.ops[1:], fb.labels[1:]
            self.ema_plan.ops, self.ema_plan.labels = fb.ops[:1], fb.labels[:1]
        self._ema_work = None
        # The same ops as two plans, cut after the decoder's gradients are final: lets a data-parallel
        # caller start reducing the decoder gradients (the contiguous tail of the flat buffer) while the
        # bottleneck / encoder backward still runs (dp.DataParallel.backward_allreduce)
        cut = bw.labels.index("unpack grads (decoder)") + 1 if "unpack grads (decoder)" in bw.labels else len(bw.ops)
        self.bwd_a, self.bwd_b = Plan("bwd_a"), Plan("bwd_b")
        self.bwd_a.ops, self.bwd_a.labels = bw.ops[:cut], bw.labels[:cut]
        self.bwd_b.ops, self.bwd_b.labels = bw.ops[cut:], bw.labels[cut:]
        # ... and, with the stack's weight gradients as two grouped launches (wgrad_group < layers, e.g. AEW_WGRAD_GROUP=10),
        # bwd_a itself in two: after bwd_a1 every gradient from layer `hi_first_layer` up (+ the post network: the tail of
        # the flat buffer from dec_hi_offset) is final, so that exchange starts under the second half of the chain
        self.bwd_a1 = self.bwd_a2 = None
        self.dec_hi_offset = None
        hi = getattr(self.dec, "hi_first_layer", None)
        lab_hi = "unpack grads (decoder, upper layers)"
        if hi is not None and lab_hi in bw.labels[:cut]:
            c1 = bw.labels.index(lab_hi) + 1
            self.bwd_a1, self.bwd_a2 = Plan("bwd_a1"), Plan("bwd_a2")
            self.bwd_a1.ops, self.bwd_a1.labels = bw.ops[:c1], bw.labels[:c1]
            self.bwd_a2.ops, self.bwd_a2.labels = bw.ops[c1:cut], bw.labels[c1:cut]
            self.dec_hi_offset = min(ps.off[n] for n in ps.names() if n.startswith(self.dec.pre + f"conv_layers.{hi}."))
            later = [n for n in ps.names() if ps.off[n] >= self.dec_hi_offset]
            assert all(n.startswith(self.dec.pre + "post") or
                       (n.startswith(self.dec.pre + "conv_layers.") and int(n[len(self.dec.pre) + 12:].split(".")[0]) >= hi)
                       for n in later), "layers >= hi and the post network form the tail of the flat buffer"
        dec_names = [n for n in ps.names() if n.startswith(self.dec.pre)]
        self.dec_grad_offset = min(ps.off[n] for n in dec_names)
        assert all(ps.off[n] >= self.dec_grad_offset for n in dec_names) and \
            all(ps.off[n] < self.dec_grad_offset for n in ps.names() if n not in dec_names), "decoder params form the tail"
        if bn == "vqvae-ema":
            # deferred codebook refresh (vqema_bn.py:216-222)
            self.cb = Plan("codebook")
            em = L.VqEma()
            em.numer, em.denom = self.ema_numer.data_ptr(), self.ema_denom.data_ptr()
            z0 = ws.alloc("bn.zero_k", self.K * self.d, torch.float32)
            em.z_sum, em.n_sum, em.emb = z0.data_ptr(), z0.data_ptr(), self.emb.data_ptr()
            em.K, em.d, em.update_codebook = self.K, self.d, 1
            em.gamma, em.gamma_comp = 1.0, 0.0     # numer/denom unchanged: emb = numer/denom
            em.guard = self.chain_guard.data_ptr()
            self.cb.add(L.OP_VQ_EMA, em, "vq.codebook", TAG_VQ)
        # pack goes first in fwd_a (its table is complete only now)
        pk_plan = Plan("pack")
        if self.merge_packs:
            # every weight layout of the step in ONE table launch at the head of fwd_a (serial plans: three launches fewer)
            allp = CopyTableBuilder(ws, "tbl.pack_all")
            for tb in (self.pack_first, self.pack_tbl, self.pack_dec, self.pack_late):
                if tb is not None:
                    allp.recs.extend(tb.recs)
            allp.emit(pk_plan, "pack weights (all layouts)")
        elif self.pack_first is not None and self.pack_first.recs:
            if not getattr(self, "_pack_dec_emitted", False):
                with pk_plan.side(self.LANE_PACK_DEC):         # joined by the end of fwd_a
                    self.pack_dec.emit(pk_plan, "pack weights (decoder)")
            self.pack_first.emit(pk_plan, "pack weights (encoder layer 0)")
            with pk_plan.side(self.PACK_LANE):
                self.pack_tbl.emit(pk_plan, "pack weights")
        else:
            if not getattr(self, "_pack_dec_emitted", False):
                with pk_plan.side(self.LANE_PACK_DEC):         # joined by the end of fwd_a
                    self.pack_dec.emit(pk_plan, "pack weights (decoder)")
            self.pack_tbl.emit(pk_plan, "pack weights")
        fa.ops[pack_slot:pack_slot] = pk_plan.ops
        fa.labels[pack_slot:pack_slot] = pk_plan.labels
        for op in pk_plan.ops:
            op.tag = TAG_PACK
        fa.keep += pk_plan.keep
        # ===== optimizer
        self.opt = Plan("adam")
        self.adam_m = ws.alloc("adam.m", ps.numel, torch.float32)
        self.adam_v = ws.alloc("adam.v", ps.numel, torch.float32)
        ad = L.Adam()
        ad.p, ad.g, ad.m, ad.v = ps.params.data_ptr(), ps.grads.data_ptr(), self.adam_m.data_ptr(), self.adam_v.data_ptr()
        ad.n = ps.numel
        ad.lr, ad.beta1, ad.beta2, ad.eps, ad.bc1, ad.bc2, ad.grad_scale = 1e-4, 0.9, 0.999, 1e-8, 1.0, 1.0, 1.0
        ad.guard = self.chain_guard.data_ptr()
        self.opt.add(L.OP_ADAM, ad, "adam", TAG_ADAM)

    def _vae_op(self, backward: bool, dcode: Optional[Mat] = None) -> L.Vae:
        va = L.Vae()
        va.lin, va.lin_pitch, va.eps = self.lin.ptr, self.nlin_p, self.eps.data_ptr()
        va.Q, va.d, va.d_pitch = self.Q, self.d, self.dp
        va.sample, va.kl_terms = self.code.ptr, self.kl_terms.data_ptr()
        va.backward = int(backward)
        if backward:
            va.dsample = dcode.ptr
            va.kl_coef = float(self.anneal_weight)
            va.kl_coef_dev = self.anneal_bwd.data_ptr()
            va.kl_value = self.loss_buf.data_ptr() + 4 * 2       # out[1 + term 1]
            va.free_nats = float(self.hps.bn_free_nats)
            va.dlin = self.dlin.ptr
            va.gmul = self.gmul.data_ptr()
        return va

    def _ae_norm_op(self, backward: bool, dcode: Optional[Mat] = None) -> L.AeNorm:
        an = L.AeNorm()
        an.ze, an.Q, an.d, an.d_pitch = self.lin.ptr, self.Q, self.d, self.nlin_p
        an.term = self.norm_terms.data_ptr()
        an.backward = int(backward)
        if backward:
            an.dze_in, an.coef, an.dze = dcode.ptr, 0.001 / self.Q, self.dlin.ptr
            an.gmul = self.gmul.data_ptr()
        return an

    # --------------------------------------------------------------------------------------
    # execution
    # --------------------------------------------------------------------------------------
    def _stream(self) -> int:
        if self.device.type == "cuda":
            return torch.cuda.current_stream(self.device).cuda_stream
        raise L.AewError("the HIP plans can only run on a GPU device; this engine was built on "
                         f"'{self.device}' (plan construction only)")

    def set_inputs(self, wav, mel, voice, jitter, eps=None):
        g = self.geom
        self.in_wav.copy_(wav)
        self.in_mel.copy_(mel)
        self.in_voice.copy_(voice)
        self.in_jitter.copy_(jitter[:, :g.embed_len])
        if self.bn_type == "vae":
            if eps is None:
                self.eps.normal_()
            else:
                self.eps.copy_(eps.permute(0, 2, 1) if eps.shape[1] == self.d and eps.dim() == 3 else eps)

    def set_anneal_weight(self, a: float):
        """SGVBLoss.update_anneal_weight (vae_bn.py:72-73).  The weight changes every step
        (chassis.py:148-149), so the ops read it from device memory (`anneal_buf`): no descriptor is
        patched and no captured graph is invalidated."""
        self.anneal_weight = float(a)
        self.anneal_buf.fill_(float(a))
        # Data parallel (dp.DataParallel): the optimizer scales the SUMMED gradient by 1/world (mean-type NLL), but
        # the KL term is a sum over all windows of the global batch, so its gradient must not be divided: pre-multiply
        self.anneal_bwd.fill_(float(a) * float(getattr(self, "dp_world", 1)))

    def _run(self, plan, timing=False):
        # self.tuning: an aew_tuning_t of this engine's own (None: the process-wide switches).  Note that plan
        # CONSTRUCTION (split-K slab counts) always follows the process-wide record.
        if self.use_graphs and not timing:
            tun = self.tuning
            if self.graph_lanes and plan._graph is None and any(op.lane >= 4 for op in plan.ops):
                # capture: lanes 4 / 5 (DecoderPlan.tail_lane: the last grouped weight-gradient launch as a branch beside
                # the rest of the backward) become graph branches unless the caller chose a lane mode itself
                tun = L.current_tuning() if tun is None else L.Tuning.from_buffer_copy(tun)
                if tun.lanes == 0:
                    tun.lanes = self.graph_lanes
            plan.run_graph(self._stream(), tuning=tun)
        else:
            plan.run(self._stream(), tuning=self.tuning)

    def _sub_plan(self, name: str, src: Plan, keep) -> Plan:
        """A plan made of the ops of `src` whose (index, label) passes `keep` (shares the op records)."""
        sp = Plan(name)
        idx = [i for i, lab in enumerate(src.labels) if keep(i, lab)]
        sp.ops, sp.labels = [src.ops[i] for i in idx], [src.labels[i] for i in idx]
        return sp

    def encode(self):
        """Encoder + bottleneck.linear only (what autoencoder_model.py:171-199 runs to collect k-means samples):
        no nearest-code search, no EMA statistics, no index histogram update.  Result in `self.lin`."""
        if getattr(self, "_encode_plan", None) is None:
            fa = self.fwd_a
            stop = fa.labels.index("bn.linear") + 1
            self._encode_plan = self._sub_plan("encode", fa, lambda i, lab: i < stop)
        self._run(self._encode_plan, False)

    def conditioning(self):
        """Encoder / bottleneck and the conditioning half of the decoder forward only (jitter gather, lc_conv,
        upsampler, speaker bias): what the autoregressive sampler needs (wavenet.py:379-391).  No EMA statistics
        or update, no index histogram update (vq.stats is skipped), no loss.
        Returns (cond bf16 [B][T][Cp], gated bias fp32 [B][NL][2*Dp])."""
        if getattr(self, "_cond_plan", None) is None:
            fb = self.fwd_b
            stop = fb.labels.index("G1.0") if "G1.0" in fb.labels else next(
                i for i, l in enumerate(fb.labels) if l.startswith("G1.0"))
            self._cond_plan = self._sub_plan("conditioning", fb, lambda i, lab: i < stop and lab != "vq.ema" and
                                             not lab.startswith("chain["))
            self._cond_plan_a = self._sub_plan("conditioning_a", self.fwd_a, lambda i, lab: lab != "vq.stats")
        self._run(self._cond_plan_a, False)
        self._run(self._cond_plan, False)
        d = self.dec
        return d.cond.tensor(), d.bias_bl[:self.B * d.NL * 2 * d.Dp].view(self.B, d.NL, 2 * d.Dp)

    # ---- chained launches: a hand-off wait that gave up must not pass silently --------------------------------------
    def _chain_flags(self):
        """[(label, 1-element int32 view of the launch's timeout flag)] over the engine's chained launches."""
        if getattr(self, "_chain_flag_views", None) is None:
            v = []
            for pl in (self.fwd_b, self.bwd):
                for lab, (stages, cd) in getattr(pl, "nt_chains", {}).items():
                    n = sum(s.n_mt * s.g.batch for s in stages)
                    v.append((lab, cd[n:n + 1]))
            self._chain_flag_views = v
        return self._chain_flag_views

    CHAIN_WATCH_SLOTS = 4

    def _chain_watch(self, after: str):
        """The bounded spin of a chained launch (csrc/aew_chain.hip) ends a wait whose producer never arrives by setting a
        flag and letting the tile run on - with operands that may be incomplete.  Two things keep such a step from
        training on:
        * on the DEVICE the wait also raises the engine's sticky word (`chain_guard`, aew_nt_chain_t.sticky), which no plan
          clears and which the Adam and EMA / codebook ops read as their guard: from the poisoned step on they are no-ops,
          however far the host has run ahead;
        * on the HOST the sticky word is copied to pinned memory behind every plan that holds chained launches (`after` =
          "fwd" | "bwd"; asynchronous) into a ring of CHAIN_WATCH_SLOTS slots; "check" (the head of every forward) looks at
          every copy that has landed and raises.  A slot is reused only after its copy has been looked at - when the host is
          a full ring ahead it waits for the oldest copy (by then several plans old) - so no timeout is ever overwritten
          unseen, and one is reported at most CHAIN_WATCH_SLOTS plans late.  `chain_guard_check()` is the synchronous form
          for any point where the caller synchronises anyway."""
        if not self._chain_flags() or self.device.type != "cuda":
            return
        st = getattr(self, "_chain_host", None)
        if st is None:
            n = self.CHAIN_WATCH_SLOTS
            st = self._chain_host = {"buf": torch.zeros(n, dtype=torch.int32).pin_memory(),
                                     "ev": [torch.cuda.Event() for _ in range(n)], "pending": []}
        if after == "check":
            self._chain_poll(False)
            return
        if len(st["pending"]) >= self.CHAIN_WATCH_SLOTS:
            self._chain_poll(True)
        used = set(st["pending"])
        slot = next(i for i in range(self.CHAIN_WATCH_SLOTS) if i not in used)
        st["buf"][slot:slot + 1].copy_(self.chain_guard[:1], non_blocking=True)
        st["ev"][slot].record()
        st["pending"].append(slot)

    def _chain_poll(self, wait_oldest: bool):
        st = self._chain_host
        while st["pending"]:
            slot = st["pending"][0]
            if not st["ev"][slot].query():
                if not wait_oldest:
                    return
                st["ev"][slot].synchronize()
            wait_oldest = False
            st["pending"].pop(0)
            if int(st["buf"][slot]) != 0:
                self._chain_raise(int(st["buf"][slot]))

    def _chain_raise(self, stage: int):
        torch.cuda.synchronize(self.device)
        bad = [lab for lab, f in self._chain_flags() if int(f.item()) != 0]      # (the per-launch flags of the LAST run)
        raise L.AewError(f"chained launch: a tile's wait for its producers gave up (aew_nt_chain_t spin limit; sticky word = "
                         f"stage {stage - 1} + 1; flagged in the last run: {bad}).  The optimizer and EMA updates have been "
                         "no-ops since that step (device-side guard); parameters are those of the step before.  "
                         "engine.clear_chain_guard() re-arms; AEW_NT_CHAIN=0 runs one launch per GEMM")

    def chain_guard_check(self):
        """Synchronous check of the sticky timeout word (for callers that synchronise anyway): raises like forward() would."""
        if self.device.type == "cuda":
            v = int(self.chain_guard[0].item())
            if v:
                self._chain_raise(v)

    def clear_chain_guard(self):
        """Re-arm after a reported timeout: clears the sticky word and forgets the copies in flight."""
        self.chain_guard.zero_()
        if getattr(self, "_chain_host", None) is not None:
            torch.cuda.synchronize(self.device)
            self._chain_host["pending"].clear()
            self._chain_host["buf"].zero_()

    def forward(self, ema_allreduce=None, timing=False, before_decoder=None):
        """timing=True forces eager launches (the per-op event timing needs them).
        before_decoder(): called between the two forward plans - the first point at which a decoder parameter is read
        (pack_dec_late); a data-parallel caller waits there for the all-gather of the decoder's parameter shards.
        ema_allreduce(z_sum, n_sum): cross-rank sum of the EMA statistics.  If it returns a work handle
        (async collective) the EMA accumulation is deferred to finish_ema(), called by backward() - or by the next
        forward() if no backward came in between (forward-only use: the accumulation must not be lost)."""
        self._chain_watch("check")
        self.finish_ema(timing)
        self._run(self.fwd_a, timing)
        if before_decoder is not None:
            before_decoder()
        if ema_allreduce is not None and self.bn_type == "vqvae-ema":
            work = ema_allreduce(self.z_sum, self.n_sum)
            if work is not None and self.ema_plan is not None:
                self._ema_work = work
                self._run(self.fwd_b_noema, timing)
                if not timing:
                    self._chain_watch("fwd")
                return self.loss_buf[0]
        self._run(self.fwd_b, timing)
        if not timing:
            self._chain_watch("fwd")
        return self.loss_buf[0]

    def finish_ema(self, timing=False):
        """Deferred vq.ema (see forward): wait for the statistics' all-reduce, then accumulate."""
        if self._ema_work is not None:
            self._ema_work.wait()
            self._ema_work = None
            self._run(self.ema_plan, timing)

    def set_upstream_grad(self, g):
        """d(L)/d(loss) for the next backward(): a python float or a 0-d tensor (copied device-side, no sync)."""
        if torch.is_tensor(g):
            self.gmul[:1].copy_(g.detach().reshape(1).to(self.gmul.dtype), non_blocking=True)
        else:
            self.gmul[:1].fill_(float(g))

    def backward(self, timing=False, after_decoder=None, after_decoder_hi=None):
        """after_decoder: optional callback invoked between the decoder part of the backward (all
        decoder gradients final in ps.grads[dec_grad_offset:]) and the bottleneck / encoder part.
        after_decoder_hi: optional callback invoked as soon as ps.grads[dec_hi_offset:] is final (engines built with two
        grouped weight-gradient launches only: dec_hi_offset is not None)."""
        if after_decoder is None or not self.bwd_b.ops:
            self._run(self.bwd, timing)
            if after_decoder_hi is not None and self.dec_hi_offset is not None:
                after_decoder_hi()
            if after_decoder is not None:
                after_decoder()
        else:
            if after_decoder_hi is not None and self.bwd_a1 is not None:
                self._run(self.bwd_a1, timing)
                after_decoder_hi()
                self._run(self.bwd_a2, timing)
            else:
                self._run(self.bwd_a, timing)
            after_decoder()
            self._run(self.bwd_b, timing)
        if not timing and self.nt_chain_bwd_used >= 2:
            self._chain_watch("bwd")
        self.finish_ema(timing)
        if self.bn_type == "vqvae-ema" and self.update_codebook_every_step:
            self._run(self.cb, timing)

    def update_codebook(self):
        self.finish_ema()
        self.cb.run(self._stream())

    def adam_step(self, lr: float, grad_scale: float = 1.0, betas=(0.9, 0.999), eps: float = 1e-8,
                  lo: int = 0, hi: Optional[int] = None, count: bool = True):
        """One Adam step over the flat buffer, or over its element range [lo, hi) (multiples of 4): a
        data-parallel caller updates the decoder tail while the encoder gradients are still being
        reduced (`count=False` on all but the first range of a step)."""
        if count:
            self.step_count += 1
        self.weights_version += 1
        hi = self.ps.numel if hi is None else hi
        assert lo % 4 == 0 and (hi % 4 == 0 or hi == self.ps.numel) and 0 <= lo < hi <= self.ps.numel
        a = self.opt.array()[0].u.adam
        a.p, a.g = self.ps.params.data_ptr() + 4 * lo, self.ps.grads.data_ptr() + 4 * lo
        a.m, a.v = self.adam_m.data_ptr() + 4 * lo, self.adam_v.data_ptr() + 4 * lo
        a.n = hi - lo
        a.lr, a.beta1, a.beta2, a.eps = lr, betas[0], betas[1], eps
        a.bc1 = 1.0 - betas[0] ** self.step_count
        a.bc2 = 1.0 - betas[1] ** self.step_count
        a.grad_scale = grad_scale
        self.opt.run(self._stream())

    # --------------------------------------------------------------------------------------
    # views for the module surface / tests
    # --------------------------------------------------------------------------------------
    def logits(self) -> torch.Tensor:
        """(B, w, Q) fp32 channels-last view."""
        return self.dec.logits.tensor()[:, :, :self.hps.n_quant]

    def init_ema_from_emb(self):
        """vqema_bn.py:117-118."""
        comp = 1.0 - self.hps.bn_vq_ema_gamma
        self.ema_numer.copy_(self.emb * comp)
        self.ema_denom.fill_(comp)

    def flops_per_step(self) -> Dict[str, float]:
        """Algorithmic FLOPs of one training step by the formula of SURVEY Appendix B /
        BASELINE.md §4 (decoder stack + post network; x3 for fwd + dgrad + wgrad)."""
        h, g = self.hps, self.geom
        R, D, S, P, Q = h.n_res, h.n_dil, h.n_skp, h.n_post, h.n_quant
        C = h.n_lc_out + h.n_global_embed
        nl = len(g.layers)
        mac = 0
        for i, lg in enumerate(g.layers):
            mac += lg.out_len * (2 * 2 * R * D + 2 * C * D + (D * R if i < nl - 1 else 0))
        mac += nl * g.n_win * D * S + g.n_win * (S * P + P * Q)
        fwd = 2.0 * mac * self.B
        return {"fwd": fwd, "step": 3.0 * fwd}
