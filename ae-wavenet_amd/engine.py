"""Launch plans for the WaveNet-autoencoder training step.

Builds, once per (model, batch, window), the forward and backward plans that
`aew_run_plan` executes: weight packing, encoder (fp32 exact chain), bottleneck, the
conditioning path, the gated dilated stack, the post network, the loss, and the complete
hand-written backward including weight-gradient GEMMs and gradient unpacking.

Reference arithmetic restated by these plans (citations into the reference checkout):
  encoder            wave_encoder.py:34-50, 53-103
  bottlenecks        vqema_bn.py:125-222, vq_bn.py:28-61, vae_bn.py:26-62, ae_bn.py:11-16
  decoder            wavenet.py:91-111 (gated layer), :127-140 (conditioning), :142-165
                     (upsampling), :323-364 (forward_train)
  losses             wavenet.py:541-552, vqema_bn.py:231-266, vq_bn.py:72-115,
                     vae_bn.py:76-125, ae_bn.py:29-46
  wiring             autoencoder_model.py:206-259, mfcc_inverter.py:89-108
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L
from . import geometry as G
from .plan import (CopyTableBuilder, ESIZE, Mat, Plan, TnGroupBuilder, Workspace, make_nt, make_tn, null_view, ru)

BF, F3 = L.BF16, L.F32

# timing tags (reported by aew_timing_read; bench.py groups kernel time by these)
TAG_G1, TAG_G2, TAG_DZ, TAG_DX, TAG_WG_FG, TAG_WG_RS, TAG_DCOND, TAG_POST, TAG_UPS, TAG_ENC, \
    TAG_VQ, TAG_LOSS, TAG_PACK, TAG_MISC, TAG_ADAM = range(1, 16)


# ------------------------------------------------------------------------------------------
# parameters
# ------------------------------------------------------------------------------------------
def decoder_param_specs(hps, n_lc_in: int, pre: str) -> List[Tuple[str, Tuple[int, ...]]]:
    """Names / shapes / registration order of the reference WaveNet module
    (wavenet.py:184-259)."""
    R, D, S, P, Q = hps.n_res, hps.n_dil, hps.n_skp, hps.n_post, hps.n_quant
    Cc = hps.n_lc_out + hps.n_global_embed
    sp: List[Tuple[str, Tuple[int, ...]]] = [(pre + "lc_conv.weight", (hps.n_lc_out, n_lc_in, 3))]
    if hps.bias:
        sp.append((pre + "lc_conv.bias", (hps.n_lc_out,)))
    for i, f in enumerate(hps.lc_upsample_filt_sizes):
        sp.append((pre + f"lc_upsample.{i}.tconv.weight", (hps.n_lc_out, hps.n_lc_out, f)))
        sp.append((pre + f"lc_upsample.{i}.tconv.bias", (hps.n_lc_out,)))
    sp.append((pre + "cond.speaker_embedding.weight", (hps.n_global_embed, hps.n_speakers)))
    sp.append((pre + "cond.speaker_embedding.bias", (hps.n_global_embed,)))
    sp.append((pre + "base_layer.weight", (R, Q, 1)))
    if hps.bias:
        sp.append((pre + "base_layer.bias", (R,)))
    nl = hps.n_blocks * hps.n_block_layers
    for i in range(nl):
        p = pre + f"conv_layers.{i}."
        for nm in ("conv_signal", "conv_gate"):
            sp.append((p + nm + ".weight", (D, R, 2)))
            if hps.bias:
                sp.append((p + nm + ".bias", (D,)))
        sp.append((p + "proj_signal.weight", (D, Cc, 1)))
        sp.append((p + "proj_gate.weight", (D, Cc, 1)))
        sp.append((p + "dil_skp.weight", (S, D, 1)))
        if i != nl - 1:
            sp.append((p + "dil_res.weight", (R, D, 1)))
    for nm, shp in (("post1", (P, S, 1)), ("post2", (Q, P, 1))):
        sp.append((pre + nm + ".weight", shp))
        if hps.bias:
            sp.append((pre + nm + ".bias", (shp[0],)))
    return sp


def encoder_param_specs(n_in: int, n_out: int, pre: str = "encoder."):
    sp = []
    cin = n_in
    for i, f in enumerate(G.ENCODER_FILTERS):
        sp.append((pre + f"net.{i}.conv.weight", (n_out, cin, f)))
        sp.append((pre + f"net.{i}.conv.bias", (n_out,)))
        cin = n_out
    return sp


def bottleneck_param_specs(hps, pre: str = "bottleneck."):
    bn, E, d = hps.bn_type, hps.enc_n_out, hps.bn_n_out
    if bn == "vae":
        return [(pre + "linear.weight", (2 * d, E, 1))]
    if bn == "ae":
        return [(pre + "linear.weight", (d, E, 1)), (pre + "linear.bias", (d,))]
    if bn == "vqvae":
        # nn.Module registers direct Parameters before sub-module parameters (vq_bn.py:13,20)
        return [(pre + "emb", (hps.bn_vq_n_embed, d)), (pre + "linear.weight", (d, E, 1))]
    return [(pre + "linear.weight", (d, E, 1))]


class ParamStore:
    """All trainable parameters in one flat fp32 buffer (+ a same-shaped gradient buffer):
    one all-reduce, one fused Adam launch, and the pack/unpack tables address it by offset."""

    def __init__(self, ws: Workspace, specs: Sequence[Tuple[str, Tuple[int, ...]]]):
        self.ws = ws
        self.shape: Dict[str, Tuple[int, ...]] = {}
        self.off: Dict[str, int] = {}
        o = 0
        for n, shp in specs:
            self.shape[n], self.off[n] = tuple(shp), o
            numel = 1
            for s in shp:
                numel *= s
            o += ru(numel, 4)
        self.numel = o
        self.params = ws.alloc("params", o, torch.float32)
        self.grads = ws.alloc("grads", o, torch.float32)

    def names(self):
        return list(self.shape)

    def numel_of(self, n):
        k = 1
        for s in self.shape[n]:
            k *= s
        return k

    def view(self, n, grad=False) -> torch.Tensor:
        t = self.grads if grad else self.params
        return t[self.off[n]:self.off[n] + self.numel_of(n)].view(self.shape[n])

    def ptr(self, n, grad=False) -> int:
        t = self.grads if grad else self.params
        return t.data_ptr() + 4 * self.off[n]

    def has(self, n):
        return n in self.shape


# ------------------------------------------------------------------------------------------
# pack / unpack helpers
# ------------------------------------------------------------------------------------------
def _gate_groups(D: int):
    """Output channel co -> packed row (co//16)*32 + co%16 (+16 for gate).  Yields
    (co_start, n_groups, group_len) pieces covering [0, D)."""
    full = D // 16
    if full:
        yield 0, full, 16
    if D % 16:
        yield full * 16, 1, D % 16


class Packer:
    """Emits strided-copy records in both directions for one packed matrix:
    pack   params(fp32)        -> packed weights (bf16 / f32)
    unpack wgrad slabs (fp32)  -> flat grads (fp32), summing `slabs` partials."""

    def __init__(self, ps: ParamStore, pack_tbl: CopyTableBuilder, unpack_tbl: CopyTableBuilder,
                 late_tbl: Optional[CopyTableBuilder] = None, first_tbl: Optional[CopyTableBuilder] = None):
        self.ps, self.pack_tbl, self.unpack_tbl = ps, pack_tbl, unpack_tbl
        self.late_tbl = late_tbl            # packs that only the backward reads (dgrad layouts): off the forward's start
        self.first_tbl = first_tbl          # packs the very first GEMM of the step needs (first=True): the only ones it waits for

    def rec(self, pname: str, p_off: int, p_strides: Sequence[int], dims: Sequence[int],
            w_mat: Optional[Mat], w_off: int, w_strides: Sequence[int],
            g_ptr: int = 0, g_strides: Optional[Sequence[int]] = None, slabs: int = 0, slab_stride: int = 0,
            g_off: Optional[int] = None, late: bool = False, first: bool = False):
        """One rectangular piece.  p_* address the parameter tensor (elements, relative to the
        parameter), w_* the packed forward matrix, g_* the wgrad slab (defaults to the same
        layout as the packed matrix)."""
        ps = self.ps
        if w_mat is not None:
            tbl = self.late_tbl if (late and self.late_tbl is not None) else self.pack_tbl
            if first and not late and self.first_tbl is not None:
                tbl = self.first_tbl
            tbl.add(ps.ptr(pname) + 4 * p_off, w_mat.ptr + w_off * ESIZE[w_mat.dtype],
                    dims, p_strides, w_strides, F3, w_mat.dtype)
        if g_ptr:
            gs = list(g_strides if g_strides is not None else w_strides)
            go = w_off if g_off is None else g_off
            self.unpack_tbl.add(g_ptr + 4 * go, ps.ptr(pname, grad=True) + 4 * p_off, dims, gs,
                                p_strides, F3, F3, red_n=slabs, red_stride=slab_stride)


# ------------------------------------------------------------------------------------------
# decoder
# ------------------------------------------------------------------------------------------
class DecoderPlan:
    """Conditioning path + gated stack + post network + NLL, forward and backward."""

    EARLY_LANE = 5            # side lane of the early halves below (the wgrads rotate over lanes 1..n_side_lanes)
    split_multiseg = False    # True: skip-sum / cond-gradient GEMMs as two halves, the early half on a side lane.
                              # Measured 8.31-8.34 vs 8.34-8.39 ms/step, but the bf16 partial sum costs accuracy (one
                              # gradient at 15.1 % of its max against the oracle, limit 15 %): off
    n_side_lanes = 4          # weight gradients / column sums rotate over this many side lanes (1..4): they are
                              # mutually independent, so they need no ordering among themselves.  Measured
                              # ms/step: 1 lane 8.44, 2: 8.54, 3: 8.59, 4: 8.19 (one lane per op kind: 8.93)
    # Ops that run on the full-N loader / consumer kernels (csrc/aew_fn.hip, impl 2; bit-identical results):
    #   "layer"  the gated pair of a layer as ONE fused op (gated GEMM -> z tile in LDS -> residual 1x1 + add)
    #   "G1"     gated GEMM alone on the full-N kernel (when "layer" is off or for the last layer)
    #   "dx" "dz" "skip" "dcond" "post"   the other large NT GEMMs of the stack
    # Default: the two long-K multi-segment GEMMs (K = 20 x 256: skip sum, cond gradient), where the loader / consumer
    # pipeline wins (0.170 vs 0.187 ms, 0.224 vs 0.248 ms).  The fused layer and the other ops are measured SLOWER than
    # the tiled kernels on this workload (fused layer 100-121 us vs 77-104 us for the pair; profiles/r02_notes.md:
    # with one block per CU nothing runs under the epilogue's 220 KB of stores per tile) and stay opt-in (AEW_FN_OPS).
    fn_ops = frozenset(("skip", "dcond"))
    wgrad_group = 64          # weight gradients of the gated stack (fg + res per layer), the skip weights and the post
                              # network as grouped launches (AEW_OP_GEMM_TN_GROUP) of this many layers: every output
                              # tile contracts over the whole time axis and batch in one block - one fp32 result per
                              # matrix instead of 8-90 split-K slabs that the unpack has to read back, and the ones-
                              # channel column gives the per-batch bias sums as running snapshots.  >= n_layers: ONE
                              # launch after the chain (768 tiles for the 20-layer stack = 3 blocks per CU); smaller
                              # groups run under the chain but were measured slower (a group of 8 layers is ~270 blocks
                              # of 0.9 ms each: one block per CU is bound by its own fill latency, and the chain's
                              # kernels lose the LDS those blocks hold: 8.63 / 9.68 ms per step with groups of 8 / 4
                              # against 7.68 with one launch and 7.75 with one split-K op per matrix).  0: one TN op per
                              # matrix (round 2's form)
    side_inputs_first = True  # forward: spk_bias / base_gather at the head of the plan's side lane (False: after the upsamplers)
    wgrad_tile = None         # output tile of the grouped wgrad launch: 128 (three 4-wave blocks per CU) | 256 (one 8-wave
                              # block per CU, half the operand bytes staged per FLOP) | None = by the size of the launch:
                              # 256 when the stack's matrices make more than 1024 tiles of 128 (more than the 768 block slots
                              # hold at once), else 128.  Measured: arch.vqvae-ema (768 tiles) 1.66 ms with 128, 2.21 ms with
                              # 256 - a lone block fills its LDS at ~26 GB/s whatever its ring depth, three independent blocks
                              # per CU reach 43 GB/s together; the deep decoder (30 x 512: 1450 tiles, bound by what it
                              # re-fetches) 67.1 ms per step with 128, 64.8 with 128 + row cursor, 62.9 with 256
                              # (profiles/r04_notes.md 24, 25)
    wgrad_cursor = None       # the grouped weight-gradient launches of the stack paced by the row cursor
                              # (aew_gemm_tn_group_t.cursors): True / False / None = iff the launch has more tiles than
                              # TnGroupBuilder.CURSOR_AUTO_TILES.  Measured: arch.vqvae-ema (768 tiles, one resident wave) 7.03 ->
                              # 7.26 ms per step although the launch fetches 28 % less - it is not bound by what it fetches;
                              # the deep decoder (30 x 512, 1450 tiles of 268 k rows) 66.5 -> 64.4 ms (weight-gradient class
                              # 24.9 -> 23.0 ms) - there it is (profiles/r04_notes.md 21, 24)
    split_chains_bwd = False  # the backward's dz / dx chain as two half-batch chains on lanes 4 / 5 (see split_chains)
    tail_lane = 0             # 4: the LAST grouped weight-gradient launch, the speaker / gated-bias gradients that read its column
                              # sums and the decoder's gradient unpack form one side branch on this lane (lane mode 2 honours
                              # lanes 4 / 5 alone; TrainEngine.graph_lanes), and everything else after the dgrad chain - cond
                              # gradient, upsampler / LC backward, bottleneck and encoder backward: ~0.6 ms of small launches
                              # that depend on no weight gradient - runs beside it on the main lane: ONE fork after dx.0, one
                              # join before the last unpack.  Measured on three boxes, interleaved: 6.83 vs 6.91, 6.99 vs 6.97,
                              # 6.99 vs 6.90 ms per step - the grouped launch holds three blocks on most CUs for its whole 1.6
                              # ms, so the small launches only get the leftover slots, and the fork / join costs what that
                              # returns (profiles/r04_notes.md 18).  0 (default): serial plan order
    wgrad_split_layers = 0    # with grouped wgrads: the TOP this-many layers keep one split-K TN op per matrix (they run
                              # under the dgrad chain and fill its tile-wave tails), the rest go to the grouped launch
    ups_split_rows = 1024     # upsampler / lc-conv weight gradients (few output tiles): contractions longer than this many
                              # rows x batch are cut into 512-row chunks (one block and one slab each) in the grouped launch
    split_one_lane = False    # with split_chains / split_chains_bwd: both half-batch chains on the MAIN lane, interleaved stage by
                              # stage - no graph branches; meant for the chained launch (TrainEngine.nt_chain), where the stages of
                              # the two halves then alternate inside ONE launch and each half's tiles run while the other half's
                              # wait for their producers
    split_chains = False      # True: gated stack as two half-batch chains on lanes 4 / 5 (build_forward); split_chains_bwd: the
                              # same for the backward's dz / dx chain.  Captured as graph branches (TrainEngine.graph_lanes) both
                              # together return 0.07-0.125 ms per step (five boxes, interleaved: 6.95 -> 6.85, 6.97 -> 6.905,
                              # 6.98 -> 6.885, 6.85 -> 6.755, 6.945 -> 6.82; either one alone: -0.09 ... +0.03), bit-identical
                              # results (tests/test_gpu_parity.py).  Default off: two concurrent half-batch launches each run
                              # longer than half a full-batch launch, so every per-kernel duration - HIP events in serial
                              # timing mode, rocprofv3 under overlap - and with it the roofline line of bench.py would stop
                              # describing the kernel; 1.5 % of the step is less than the box-to-box spread (profiles/r04_notes.md 19)

    def __init__(self, ws: Workspace, ps: ParamStore, hps, geom: G.ModelGeom, B: int, pre: str,
                 n_lc_in: int, lc_src: Mat, wav: torch.Tensor, voice: torch.Tensor,
                 jitter: torch.Tensor, take_compat: bool, packer: Packer, impl: int = 0,
                 wgrad_group: Optional[int] = None):
        self.ws, self.ps, self.hps, self.g, self.B, self.pre = ws, ps, hps, geom, B, pre
        self.impl = impl
        import os as _os
        if _os.environ.get("AEW_FN_OPS") is not None:          # A/B aid: comma list, "" = none
            self.fn_ops = frozenset(v for v in _os.environ["AEW_FN_OPS"].split(",") if v)
        if wgrad_group is not None:                            # resolved once by TrainEngine (argument, else AEW_WGRAD_GROUP,
            self.wgrad_group = int(wgrad_group)                # else the class default) and shared with the EncoderPlan
        if hps.n_global_embed > 16:
            # k_spk_bwd holds a layer's speaker-projection row in registers (AEW_SPK_MAXG)
            raise ValueError(f"n_global_embed = {hps.n_global_embed}: the speaker-gradient kernel supports at most 16")
        self.gmul_ptr = ws.bufs["loss.gmul"].data_ptr() if "loss.gmul" in ws.bufs else 0
        self.n_lc_in = n_lc_in
        self.lc_src, self.wav, self.voice, self.jitter = lc_src, wav, voice, jitter
        self.take_compat = take_compat
        self.pk = packer
        h = hps
        self.R, self.D, self.S, self.P, self.Q = h.n_res, h.n_dil, h.n_skp, h.n_post, h.n_quant
        self.Clc, self.Gc = h.n_lc_out, h.n_global_embed
        self.Rp, self.Dp, self.Sp, self.Pp, self.Qp = (ru(v, 128) for v in (self.R, self.D, self.S, self.P, self.Q))
        self.Cp, self.Lp = ru(self.Clc, 128), ru(n_lc_in, 128)
        self.NL = len(geom.layers)
        self.T, self.w = geom.dec_in_len, geom.n_win
        self.Ne = geom.embed_len
        self._alloc()
        self._pack_records()

    # -- buffers --------------------------------------------------------------------------
    def _alloc(self):
        ws, B, g = self.ws, self.B, self.g
        p = self.pre
        M = lambda n, rows, pitch, dt, cols=None, guard=0: Mat.new(ws, p + n, B, rows, pitch, dt, cols, guard)
        self.lcj = M("lcj", self.Ne, self.Lp, BF)
        self.lc_lens = g.lc_lens                      # [Ne, Ne-2, after each upsampler]
        self.ups_in: List[Mat] = [M("lc1", self.lc_lens[1], self.Cp, BF)]
        n_ups = len(self.hps.lc_upsample_strides)
        # upsampler outputs carry guard rows: the merged-phase GEMM writes `stride` matrix rows per GEMM row
        for i in range(n_ups - 1):
            self.ups_in.append(M(f"ups{i}", self.lc_lens[2 + i], self.Cp, BF, guard=8))
        self.cond = M("cond", self.T, self.Cp, BF, guard=8)
        self.bias_bl = ws.alloc(p + "bias_bl", B * self.NL * 2 * self.Dp, torch.float32)
        self.gc = ws.alloc(p + "gc", B * self.Gc, torch.float32)
        self.x: List[Mat] = [M(f"x{l}", lg.in_len, self.Rp, BF) for l, lg in enumerate(g.layers)]
        self.z = [M(f"z{l}", lg.out_len, self.Dp, BF) for l, lg in enumerate(g.layers)]
        # local derivatives dz/dfilt, dz/dgate saved by the gated epilogue for backward
        self.pf = [M(f"pf{l}", lg.out_len, self.Dp, BF) for l, lg in enumerate(g.layers)]
        self.pg = [M(f"pg{l}", lg.out_len, self.Dp, BF) for l, lg in enumerate(g.layers)]
        self.h0 = M("h0", self.w, self.Sp, BF)
        self.skp_part = M("skp_part", self.w, self.Sp, BF)      # skip sum of the lower half of the stack (pre-relu)
        self.h1 = M("h1", self.w, self.Pp, BF)
        self.logits = M("logits", self.w, self.Qp, F3)
        self.nll = ws.alloc(p + "nll", B * self.w, torch.float32)
        self.ptgt = ws.alloc(p + "ptgt", B * self.w, torch.float32)
        # backward
        self.onehot = M("onehot", self.T, self.Qp, BF)
        self.dlogits = M("dlogits", self.w, self.Qp, BF)
        self.dh1 = M("dh1", self.w, self.Pp, BF)
        self.dskp = M("dskp", self.w, self.Sp, BF)
        # one dx buffer per layer (not a ping-pong pair): the res / base wgrads that read dx_{l+1} run
        # on the side lane and may lag the dx chain by several layers
        self.dx = [M(f"dx{l}", g.layers[l].in_len, self.Rp, BF) for l in range(self.NL)]
        self.dfg = [M(f"dfg{l}", lg.out_len, 2 * self.Dp, BF) for l, lg in enumerate(g.layers)]
        self.colsum_fg = ws.alloc(p + "colsum_fg", B * self.NL * 2 * self.Dp, torch.float32)
        self.dcond = M("dcond", self.T, self.Cp, BF)
        self.dcond_part = M("dcond_part", self.T, self.Cp, BF)  # cond gradient of the upper half of the stack
        self.dups = [M(f"d{m.name[len(p):]}", m.rows, self.Cp, BF) for m in self.ups_in]
        self.dlcj = M("dlcj", self.Ne, self.Lp, F3)
        # gradient w.r.t. the LC source (fp32, same shape as lc_src)
        self.dlc_src = Mat.new(ws, p + "dlc_src", B, self.lc_src.rows, self.lc_src.pitch, F3)
        # offset tables for the speaker-bias ops
        offs = {k: [] for k in ("bs", "bg", "ps", "pg")}
        for l in range(self.NL):
            q = p + f"conv_layers.{l}."
            offs["bs"].append(self.ps.off.get(q + "conv_signal.bias", -1))
            offs["bg"].append(self.ps.off.get(q + "conv_gate.bias", -1))
            offs["ps"].append(self.ps.off[q + "proj_signal.weight"])
            offs["pg"].append(self.ps.off[q + "proj_gate.weight"])
        self.off_tbl = {}
        for k, v in offs.items():
            t = ws.alloc(p + "off_" + k, self.NL, torch.int64)
            t[:self.NL].copy_(torch.tensor(v, dtype=torch.int64))
            self.off_tbl[k] = t

    def _impl(self, kind: str) -> int:
        """impl of an NT op of the given kind: the full-N kernels (2) where selected, unless the engine was built on
        the scalar check kernels (impl 1)."""
        return 2 if (self.impl == 0 and kind in self.fn_ops) else self.impl

    def _wmat(self, name, rows, cols, dt=BF) -> Mat:
        return Mat.new(self.ws, self.pre + "wp." + name, 1, rows, cols, dt)

    def _gslab(self, name, rows, cols, slabs) -> Tuple[int, int]:
        t = self.ws.alloc(self.pre + "wg." + name, slabs * rows * cols, torch.float32)
        return t.data_ptr(), rows * cols

    # -- packed weight matrices and their pack/unpack records ------------------------------
    def _pack_records(self):
        pk, p, ps = self.pk, self.pre, self.ps
        R, D, S, P, Q, Clc = self.R, self.D, self.S, self.P, self.Q, self.Clc
        Rp, Dp, Sp, Pp, Qp, Cp, Lp = self.Rp, self.Dp, self.Sp, self.Pp, self.Qp, self.Cp, self.Lp
        Cc = Clc + self.Gc
        NL = self.NL
        self.Wfg, self.Wrs, self.WrsT, self.WfgT = [], [], [], []
        self.tn: Dict[str, L.GemmTN] = {}         # wgrad ops by name (built in build_backward)
        self.gbuf: Dict[str, Tuple[int, int, int]] = {}  # name -> (ptr, slab stride, slabs)
        Kfg = 2 * Rp + Cp
        # the 20-segment GEMMs (skip sum, cond gradient) run as two halves: the half whose inputs exist
        # mid-chain goes to a side lane (fills tile-wave tails), the other half adds it as aux
        self.n_lo = NL // 2 if (self.split_multiseg and NL >= 4) else NL      # layers [0, n_lo) | [n_lo, NL)
        n_lo, n_hi = self.n_lo, NL - self.n_lo
        self.VfgT_lo = self._wmat("VfgT_lo", Cp, n_lo * 2 * Dp)
        self.VfgT_hi = self._wmat("VfgT_hi", Cp, max(n_hi, 1) * 2 * Dp)
        self.Wskp_lo = self._wmat("skp_lo", Sp, n_lo * Dp)
        self.Wskp_hi = self._wmat("skp_hi", Sp, max(n_hi, 1) * Dp)
        # base layer as a row gather: transposed fp32 copy [Q][Rp] (wavenet.py:348-351)
        self.Wbase_t = self._wmat("base_t", Q, Rp, F3)
        pk.rec(p + "base_layer.weight", 0, [Q, 1], [R, Q], self.Wbase_t, 0, [1, Rp])
        for l in range(NL):
            last = l == NL - 1
            q = p + f"conv_layers.{l}."
            Wfg = self._wmat(f"fg{l}", 2 * Dp, Kfg)
            WfgT = self._wmat(f"fgT{l}", Rp, 4 * Dp)
            self.Wfg.append(Wfg); self.WfgT.append(WfgT)
            for gate, nm in ((0, "signal"), (1, "gate")):
                for co0, ng, gl in _gate_groups(D):
                    row0 = (co0 // 16) * 32 + 16 * gate
                    # conv weight [D][R][2] -> Wfg rows, tap-major K; and WfgT (dgrad) layout
                    pk.rec(q + f"conv_{nm}.weight", co0 * R * 2, [16 * R * 2, R * 2, 2, 1], [ng, gl, R, 2],
                           Wfg, row0 * Kfg, [32 * Kfg, Kfg, 1, Rp])
                    pk.rec(q + f"conv_{nm}.weight", co0 * R * 2, [16 * R * 2, R * 2, 2, 1], [ng, gl, R, 2],
                           WfgT, row0, [32, 1, 4 * Dp, 2 * Dp])
                    # conditioning projection [D][Cc][1], first Clc columns only
                    pk.rec(q + f"proj_{nm}.weight", co0 * Cc, [16 * Cc, Cc, 1], [ng, gl, Clc],
                           Wfg, row0 * Kfg + 2 * Rp, [32 * Kfg, Kfg, 1])
                    Vt, lv, nv = (self.VfgT_lo, l, self.n_lo) if l < self.n_lo else (self.VfgT_hi, l - self.n_lo, NL - self.n_lo)
                    pk.rec(q + f"proj_{nm}.weight", co0 * Cc, [16 * Cc, Cc, 1], [ng, gl, Clc],
                           Vt, lv * 2 * Dp + row0, [32, 1, nv * 2 * Dp])
            # residual 1x1 (forward) and [res | skip]^T (backward dz); the skip 1x1s of ALL layers
            # form one matrix Wskp [Sp][NL*Dp] for the deferred skip GEMM (wavenet.py:103,357)
            Nrs = Sp if last else Rp + Sp
            Wrs = None if last else self._wmat(f"rs{l}", Rp, Dp)
            WrsT = self._wmat(f"rsT{l}", Dp, Nrs)
            self.Wrs.append(Wrs); self.WrsT.append(WrsT)
            so = 0 if last else Rp
            if not last:
                pk.rec(q + "dil_res.weight", 0, [D, 1], [R, D], Wrs, 0, [Dp, 1])
                pk.rec(q + "dil_res.weight", 0, [D, 1], [R, D], WrsT, 0, [1, Nrs])
            Ws, ls, ns = (self.Wskp_lo, l, self.n_lo) if l < self.n_lo else (self.Wskp_hi, l - self.n_lo, NL - self.n_lo)
            pk.rec(q + "dil_skp.weight", 0, [D, 1], [S, D], Ws, ls * Dp, [ns * Dp, 1])
            pk.rec(q + "dil_skp.weight", 0, [D, 1], [S, D], WrsT, so, [1, Nrs])
        # post network
        self.Wp1, self.Wp1T = self._wmat("p1", Pp, Sp), self._wmat("p1T", Sp, Pp)
        self.Wp2, self.Wp2T = self._wmat("p2", Qp, Pp), self._wmat("p2T", Pp, Qp)
        pk.rec(p + "post1.weight", 0, [S, 1], [P, S], self.Wp1, 0, [Sp, 1])
        pk.rec(p + "post1.weight", 0, [S, 1], [P, S], self.Wp1T, 0, [1, Pp])
        pk.rec(p + "post2.weight", 0, [P, 1], [Q, P], self.Wp2, 0, [Pp, 1])
        pk.rec(p + "post2.weight", 0, [P, 1], [Q, P], self.Wp2T, 0, [1, Qp])
        # fp32 bias vectors padded to N_pad
        self.bias_vec: Dict[str, int] = {}
        for nm, n, npad in (("post1", P, Pp), ("post2", Q, Qp), ("lc_conv", Clc, Cp)):
            t = self.ws.alloc(p + "wp.bias." + nm, npad, torch.float32)
            self.bias_vec[nm] = t.data_ptr()
            if ps.has(p + nm + ".bias"):
                pk.pack_tbl.add(ps.ptr(p + nm + ".bias"), t.data_ptr(), [n], [1], [1], F3, F3)
        # LC conv [Clc][n_lc_in][3]
        nin = self.n_lc_in
        self.Wlc, self.WlcT = self._wmat("lc", Cp, 3 * Lp), self._wmat("lcT", Lp, 3 * Cp)
        pk.rec(p + "lc_conv.weight", 0, [nin * 3, 3, 1], [Clc, nin, 3], self.Wlc, 0, [3 * Lp, 1, Lp])
        pk.rec(p + "lc_conv.weight", 0, [nin * 3, 3, 1], [Clc, nin, 3], self.WlcT, 0, [1, 3 * Cp, Cp])
        # upsamplers: ConvTranspose1d weight [ci][co][k]
        self.Wup: List[List[Mat]] = []
        self.WupT: List[Mat] = []
        for i, (f, s) in enumerate(zip(self.hps.lc_upsample_filt_sizes, self.hps.lc_upsample_strides)):
            nm = p + f"lc_upsample.{i}.tconv.weight"
            phases = []
            Kph = (f // s) * Cp
            Wall = self._wmat(f"up{i}", s * Cp, Kph)             # phase ph = rows ph*Cp .. (ph+1)*Cp
            for ph in range(s):
                Wm = Mat(self.ws, Wall.name, 1, Cp, Kph, BF, base_off=ph * Cp * Kph)
                # [co][j*Cp + ci] <- W[ci][co][ph + s*j]
                pk.rec(nm, ph, [f, Clc * f, s], [Clc, Clc, f // s], Wm, 0, [(f // s) * Cp, 1, Cp])
                phases.append(Wm)
            self.Wup.append(phases)
            self.Wup_all = getattr(self, "Wup_all", []) + [Wall]
            WT = self._wmat(f"upT{i}", Cp, f * Cp)
            # [ci][k*Cp + co] <- W[ci][co][k]
            pk.rec(nm, 0, [Clc * f, f, 1], [Clc, Clc, f], WT, 0, [f * Cp, 1, Cp])
            self.WupT.append(WT)
            t = self.ws.alloc(p + f"wp.bias.up{i}", s * Cp, torch.float32)     # the bias once per phase
            self.bias_vec[f"up{i}"] = t.data_ptr()
            for ph in range(s):
                pk.pack_tbl.add(ps.ptr(p + f"lc_upsample.{i}.tconv.bias"), t.data_ptr() + 4 * ph * Cp, [Clc], [1], [1],
                                F3, F3)

    # -- forward ---------------------------------------------------------------------------
    def build_forward(self, plan: Plan, need_onehot: bool = True, after_logits=None, after_nll=None):
        """after_logits / after_nll: callables(plan) that append side-lane ops reading the logits / the per-position
        nll (the caller's per-step diagnostics): placed here they run under the softmax and the loss reduction
        instead of after the plan's last main-lane op."""
        B, g, hps, p = self.B, self.g, self.hps, self.pre
        Rp, Dp, Sp, Pp, Qp, Cp, Lp = self.Rp, self.Dp, self.Sp, self.Pp, self.Qp, self.Cp, self.Lp
        impl = self.impl
        # 4./5. (emitted FIRST when side_inputs_first: they depend on nothing in this plan, so the side lane runs them
        # under the conditioning path instead of starting after its last op)
        def side_inputs():
            # 4. speaker-conditioned gated bias (wavenet.py:127-140 folded)
            sb = L.SpkBias()
            self._fill_spk(sb)
            sb.bias, sb.gc = self.bias_bl.data_ptr(), self.gc.data_ptr()
            with plan.side():                                      # independent of the conditioning path
                plan.add(L.OP_SPK_BIAS, sb, "spk_bias", TAG_MISC)
            # 5. base layer = column gather (wavenet.py:348-351)
            bg = L.BaseGather()
            bg.wav, bg.wav_pitch, bg.wav_off = self.wav.data_ptr(), self.wav.shape[1], g.trim_dec_in[0]
            bg.W = self.ps.ptr(p + "base_layer.weight")
            bg.Wt = self.Wbase_t.ptr
            bg.bias = self.ps.ptr(p + "base_layer.bias") if self.ps.has(p + "base_layer.bias") else None
            bg.B, bg.T, bg.R, bg.R_pad, bg.Q = B, self.T, self.R, Rp, self.Q
            bg.x, bg.x_bs, bg.x_pitch = self.x[0].ptr, self.x[0].bs, self.x[0].pitch
            if need_onehot:
                bg.onehot, bg.oh_bs, bg.oh_pitch, bg.Q_pad = self.onehot.ptr, self.onehot.bs, self.onehot.pitch, Qp
            bg.ones_channel = int(self.R < Rp)
            with plan.side():
                plan.add(L.OP_BASE_GATHER, bg, "base_gather", TAG_MISC)
        if self.side_inputs_first:
            side_inputs()
        # 1. jitter gather (wavenet.py:330-336)
        lg_ = L.LcGather()
        lg_.src, lg_.src_bs, lg_.src_pitch = self.lc_src.ptr, self.lc_src.bs, self.lc_src.pitch
        lg_.jitter, lg_.jit_pitch = self.jitter.data_ptr(), self.jitter.shape[1]
        lg_.dst, lg_.dst_bs, lg_.dst_pitch = self.lcj.ptr, self.lcj.bs, self.lcj.pitch
        lg_.B, lg_.N, lg_.C, lg_.C_pad = B, self.Ne, self.n_lc_in, Lp
        lg_.take_compat = int(self.take_compat)
        plan.add(L.OP_LC_GATHER, lg_, "lc_gather", TAG_UPS)
        # 2. LC conv k=3 (wavenet.py:337)
        lc1 = self.ups_in[0]
        plan.add(L.OP_GEMM_NT, make_nt(
            BF, lc1.rows, Cp, Cp, B, [self.lcj.seg(Lp, row_off=t) for t in range(3)], self.Wlc.ptr,
            flags=L.EF_BIAS, out0=lc1.view(), bias_ptr=self.bias_vec["lc_conv"], impl=impl), "lc_conv", TAG_UPS)
        # 3. transposed-conv upsamplers as polyphase GEMMs (wavenet.py:154,338)
        n_ups = len(hps.lc_upsample_strides)
        for i, (f, s) in enumerate(zip(hps.lc_upsample_filt_sizes, hps.lc_upsample_strides)):
            X = self.ups_in[i]
            Lin, Lout, pad = X.rows, self.lc_lens[2 + i], f - s
            last = i == n_ups - 1
            trim0 = g.trim_ups_out[0] if last else 0
            Y = self.cond if last else self.ups_in[i + 1]
            if pad % s == 0 and f % s == 0 and s <= 8:
                # all phases in ONE GEMM: N = s*Cp, GEMM row q holds matrix rows s*q + ph - trim0 (ph = 0..s-1),
                # i.e. the output is addressed as [rows/s][s*Cp]; partial rows at the ends land in the guard rows
                q0 = pad // s
                q_lo = trim0 // s
                q_hi = min(-(-(Y.rows + trim0) // s), (Lout - 1) // s + 1)
                Mq = q_hi - q_lo
                assert s * q_lo - trim0 >= -8 and s * (q_hi - 1) + s - 1 - trim0 < Y.rows + 8
                segs = [X.seg(Cp, row_off=q0 + q_lo - j) for j in range(f // s)]
                ov = L.View()
                ov.ptr = Y.ptr + 2 * (s * q_lo - trim0) * Cp
                ov.batch_stride, ov.row_pitch, ov.row_step, ov.row_off = Y.bs, s * Cp, 1, 0
                ov.row_lo, ov.row_hi, ov.dtype = 0, Mq, BF
                plan.add(L.OP_GEMM_NT, make_nt(
                    BF, Mq, s * Cp, s * Cp, B, segs, self.Wup_all[i].ptr, flags=L.EF_BIAS, out0=ov,
                    bias_ptr=self.bias_vec[f"up{i}"], impl=impl), f"ups{i}", TAG_UPS)
                continue
            for ph in range(s):
                q0 = -((ph - pad) // s)                 # ceil((pad - ph)/s): first q with o >= 0
                o0 = s * q0 + ph - pad
                if o0 >= Lout:
                    continue
                Mq = (Lout - 1 - o0) // s + 1
                segs = [X.seg(Cp, row_off=q0 - j) for j in range(f // s)]
                plan.add(L.OP_GEMM_NT, make_nt(
                    BF, Mq, Cp, Cp, B, segs, self.Wup[i][ph].ptr, flags=L.EF_BIAS,
                    out0=Y.view(row_off=o0 - trim0, row_step=s), bias_ptr=self.bias_vec[f"up{i}"],
                    impl=impl), f"ups{i}.ph{ph}", TAG_UPS)
        if not self.side_inputs_first:
            side_inputs()
        # 6. gated dilated stack (wavenet.py:91-111, 355-357)
        NL = self.NL
        # layer 0 waits for lane 1 only (x[0], gated biases), not for the other lanes' work that only later steps read
        g1_join = ("lane", 1) if self.side_inputs_first else True
        # Optionally (split_chains) the gated stack runs as two independent half-batch chains, chain 0 on the
        # main lane and chain 1 on the side lane, so that each fills the other's tile-wave tails (a full-batch
        # layer GEMM is 1.1-1.5 waves of tiles and nothing else is runnable in the forward).
        n_chains = 2 if (self.split_chains and B % 2 == 0 and B >= 2) else 1
        nb = B // n_chains
        for l, lg in enumerate(g.layers):
            last = l == NL - 1
            x = self.x[l]
            P_l = lg.out_len
            for c in range(n_chains):
                b0 = c * nb
                # two chains: BOTH on side lanes (4 and 5: the ones aew_set_lanes(2) honours alone).  A side op is ordered after every main-lane op emitted before
                # it, so a chain left on the main lane would hold the other one back at every layer; with no main-lane op
                # between the first layer and the skip sum the two lanes run free
                plan.lane = (4 + c) if (n_chains > 1 and not self.split_one_lane) else 0
                segs = [x.seg(Rp, b0=b0), x.seg(Rp, row_off=lg.dil, b0=b0),
                        self.cond.seg(Cp, row_off=lg.cond_lead, b0=b0)]
                sfx = f".c{c}" if n_chains > 1 else ""
                gkw = dict(epi=L.EPI_GATED, out0=self.z[l].view(b0=b0), out1=self.pf[l].view(b0=b0),
                           out2=self.pg[l].view(b0=b0),
                           bias_ptr=self.bias_bl.data_ptr() + 4 * (l * 2 * Dp + b0 * NL * 2 * Dp), bias_bs=NL * 2 * Dp)
                if not last and self._impl("layer") == 2:
                    # the gated pair as ONE op (wavenet.py:100-109): z never leaves the CU between the two GEMMs
                    plan.add(L.OP_GEMM_NT, make_nt(
                        BF, P_l, Dp, 2 * Dp, nb, segs, self.Wfg[l].ptr, impl=2, W2_ptr=self.Wrs[l].ptr, N2=Rp, N2_pad=Rp,
                        out3=self.x[l + 1].view(b0=b0), aux0=x.view(row_off=lg.dil, b0=b0), **gkw),
                        f"G1.{l}" + sfx, TAG_G1, join=g1_join if (l == 0 and c == 0) else False)
                    continue
                plan.add(L.OP_GEMM_NT, make_nt(
                    BF, P_l, Dp, 2 * Dp, nb, segs, self.Wfg[l].ptr, impl=self._impl("G1"), **gkw),
                    f"G1.{l}" + sfx, TAG_G1,
                    join=((g1_join if c == 0 else False) if self.split_one_lane else (True if n_chains > 1 else g1_join))
                    if l == 0 else False)                              # x[0] and the gated biases come from the side lane
                if not last:
                    # residual 1x1 + add (wavenet.py:108-109); the final layer has no residual output
                    plan.add(L.OP_GEMM_NT, make_nt(
                        BF, P_l, Rp, Rp, nb, [self.z[l].seg(Dp, b0=b0)], self.Wrs[l].ptr, flags=L.EF_ADD_AUX0,
                        out0=self.x[l + 1].view(b0=b0), aux0=x.view(row_off=lg.dil, b0=b0), impl=impl),
                        f"G2.{l}" + sfx, TAG_G2)
            if self.n_lo < NL and l == self.n_lo - 1:
                # early half of the skip sum (layers 0 .. n_lo-1): all its inputs exist now
                plan.lane = 0
                segs = [self.z[k].seg(Dp, row_off=g.layers[k].skip_lead) for k in range(self.n_lo)]
                with plan.side(self.EARLY_LANE):
                    plan.add(L.OP_GEMM_NT, make_nt(BF, self.w, Sp, Sp, B, segs, self.Wskp_lo.ptr,
                                                   out0=self.skp_part.view(), impl=impl), "skip_lo", TAG_G2,
                             join=n_chains > 1)
        plan.lane = 0
        # skip path of all layers as ONE GEMM: relu(sum_l Wk_l . z_l[u + skip_lead_l]) -> h0
        # (wavenet.py:103,355-359).  K = NL*256; the fp32 skip sum never touches HBM.
        # With split_multiseg the layers [0, n_lo) were already summed on a side lane right after
        # G1 of layer n_lo-1 (skp_part, below in the layer loop); this GEMM adds them and applies the relu.
        lo = self.n_lo
        if lo < NL:
            segs = [self.z[l].seg(Dp, row_off=g.layers[l].skip_lead) for l in range(lo, NL)]
            plan.add(L.OP_GEMM_NT, make_nt(BF, self.w, Sp, Sp, B, segs, self.Wskp_hi.ptr,
                                           flags=L.EF_ADD_AUX0 | L.EF_RELU_POST, out0=self.h0.view(),
                                           aux0=self.skp_part.view(), impl=self._impl("skip")), "skip_all", TAG_G2, join=True)
        else:
            segs = [self.z[l].seg(Dp, row_off=lg.skip_lead) for l, lg in enumerate(g.layers)]
            plan.add(L.OP_GEMM_NT, make_nt(BF, self.w, Sp, Sp, B, segs, self.Wskp_lo.ptr, flags=L.EF_RELU,
                                           out0=self.h0.view(), impl=self._impl("skip")), "skip_all", TAG_G2, join=True)
        # 7. post network (wavenet.py:359-360)
        plan.add(L.OP_GEMM_NT, make_nt(BF, self.w, Pp, Pp, B, [self.h0.seg(Sp)], self.Wp1.ptr,
                                       flags=L.EF_BIAS | L.EF_RELU, out0=self.h1.view(),
                                       bias_ptr=self.bias_vec["post1"], impl=self._impl("post")), "post1", TAG_POST)
        plan.add(L.OP_GEMM_NT, make_nt(BF, self.w, Qp, Qp, B, [self.h1.seg(Pp)], self.Wp2.ptr,
                                       flags=L.EF_BIAS, out0=self.logits.view(),
                                       bias_ptr=self.bias_vec["post2"], impl=impl), "post2", TAG_POST)
        if after_logits is not None:
            after_logits(plan)
        # 8. fused log-softmax + NLL (wavenet.py:543-547)
        plan.add(L.OP_SOFTMAX_NLL, self._softmax(False, 0.0), "softmax_nll", TAG_LOSS)
        if after_nll is not None:
            after_nll(plan)

    def _fill_spk(self, sb):
        p, ps = self.pre, self.ps
        sb.params, sb.voice = ps.params.data_ptr(), self.voice.data_ptr()
        sb.off_bias_sig, sb.off_bias_gate = self.off_tbl["bs"].data_ptr(), self.off_tbl["bg"].data_ptr()
        sb.off_proj_sig, sb.off_proj_gate = self.off_tbl["ps"].data_ptr(), self.off_tbl["pg"].data_ptr()
        sb.off_spk_w = ps.off[p + "cond.speaker_embedding.weight"]
        sb.off_spk_b = ps.off.get(p + "cond.speaker_embedding.bias", -1)
        sb.B, sb.L, sb.D, sb.D_pad, sb.C_lc, sb.G = self.B, self.NL, self.D, self.Dp, self.Clc, self.Gc
        sb.n_speakers = self.hps.n_speakers

    def _spk_det(self, sb: "L.SpkBwd", tag: str):
        """Scratch + ticket of the deterministic speaker-embedding sums (aew_spk_bwd_t.det_scratch: one [16][16] block of terms
        per (layer, filt | gate, chunk of 16 batch elements), added in a fixed order by the last block)."""
        chunks = (self.B + 15) // 16
        sb.det_scratch = self.ws.alloc(self.pre + f"det.spk_{tag}.scratch", self.NL * 2 * chunks * 256, torch.float32).data_ptr()
        sb.det_tickets = self.ws.alloc(self.pre + f"det.spk_{tag}.tickets", 4, torch.int32, zero=True).data_ptr()

    def _softmax(self, backward: bool, scale: float) -> L.SoftmaxNll:
        sm = L.SoftmaxNll()
        sm.logits, sm.bs, sm.pitch = self.logits.ptr, self.logits.bs, self.logits.pitch
        sm.wav, sm.wav_pitch, sm.tgt_off = self.wav.data_ptr(), self.wav.shape[1], self.g.wav_out_off
        sm.B, sm.w, sm.Q, sm.Q_pad = self.B, self.w, self.Q, self.Qp
        sm.nll, sm.ptgt = self.nll.data_ptr(), self.ptgt.data_ptr()
        sm.dlogits, sm.dl_bs, sm.dl_pitch = self.dlogits.ptr, self.dlogits.bs, self.dlogits.pitch
        sm.scale, sm.backward = scale, int(backward)
        if not backward and getattr(self, "peak_ptrs", None):
            sm.peak, sm.amax = self.peak_ptrs
        if backward and self.gmul_ptr:
            sm.gmul = self.gmul_ptr
        return sm

    # -- backward --------------------------------------------------------------------------
    def _colsum(self, plan: Plan, X: Mat, M: int, N: int, out_ptr: int, out_bs: int = 0,
                row_off: int = 0, label: str = "colsum"):
        cs = L.Colsum()
        cs.x = X.seg(128, row_off=row_off)
        cs.dtype, cs.M, cs.N, cs.batch = X.dtype, M, N, self.B
        cs.out, cs.out_bs, cs.accumulate = out_ptr, out_bs, 1      # target pre-zeroed by the plan
        det_colsum(self.ws, cs, self.pre + "det." + label)
        with plan.side(self._next_lane()):
            plan.add(L.OP_COLSUM, cs, label, TAG_MISC)

    def _wgrad_tile(self) -> int:
        if self.wgrad_tile:
            return self.wgrad_tile
        n128 = self.NL * ((2 * self.Dp // 128) * ((2 * self.Rp + self.Cp) // 128) + (self.Rp // 128) * (self.Dp // 128))
        return 256 if n128 > TnGroupBuilder.CURSOR_AUTO_TILES else 128

    def _next_lane(self, kind: str = "") -> int:
        n = min(self.n_side_lanes, Plan.N_SIDE, 3)             # lanes 4 / 5 are for explicit branches (tail_lane, split_chains)
        self._lane_rr = getattr(self, "_lane_rr", 0) % max(1, n) + 1
        return self._lane_rr

    def _wgrad(self, plan: Plan, name: str, dtype: int, Mc: int, N: int, N_pad: int, gseg: L.Seg,
               segs: Sequence[L.Seg], tag: int) -> Tuple[int, int, int]:
        t = make_tn(dtype, Mc, self.B, N, N_pad, gseg, segs, impl=self.impl)
        slabs = L.tn_slabs(t)
        ptr, stride = self._gslab(name, N_pad, t.K_total, slabs)
        t.out, t.out_batch_stride = ptr, stride
        with plan.side(self._next_lane(name)):                     # off the dgrad chain
            plan.add(L.OP_GEMM_TN, t, "wgrad." + name, tag)
        self.gbuf[name] = (ptr, stride, slabs)
        return ptr, stride, slabs

    def build_backward(self, plan: Plan, nll_scale: float):
        """nll_scale = d(loss)/d(per-position nll): 1/(B*(w-1)) for mean-type losses, 1 for
        sum-type (times the upstream gradient)."""
        B, g, hps, p, ps = self.B, self.g, self.hps, self.pre, self.ps
        R, D, S, P, Q, Clc = self.R, self.D, self.S, self.P, self.Q, self.Clc
        Rp, Dp, Sp, Pp, Qp, Cp, Lp = self.Rp, self.Dp, self.Sp, self.Pp, self.Qp, self.Cp, self.Lp
        impl, NL, w, T = self.impl, self.NL, self.w, self.T
        pk = self.pk
        plan.add(L.OP_SOFTMAX_NLL, self._softmax(True, nll_scale), "softmax_grad", TAG_LOSS)
        plan.zero(self.ws, p + "colsum_fg")
        # gradients of the post network and of the upper half of the stack are unpacked mid-chain (side
        # lane), so that less unpack work is left when the chain ends (shorter drain before the encoder
        # backward / before a data-parallel caller may start reducing the decoder gradients)
        early_tbl = getattr(self, "unpack_early_tbl", None)
        late_tbl = pk.unpack_tbl
        if early_tbl is not None:
            pk.unpack_tbl = early_tbl
        # grouped weight gradients (wgrad_group layers per launch, impl 0 only: the check kernels keep one op per
        # matrix).  A group is emitted right after the dz GEMM of its lowest layer, on a side lane; the last one after
        # the chain, together with the skip and post-network weight gradients.  With the ones channel in x the fg
        # descriptors also deliver the running per-batch column sums of dfg.
        grouped = self.wgrad_group > 0 and self.impl == 0
        # running per-batch column sums of dfg out of the grouped launch: column R of the weight gradient where x carries
        # the ones channel (R < Rp), else the launch's own all-ones operand (aew_gemm_tn_t.snap_k = -1)
        snap_ok = grouped
        grp: Optional[TnGroupBuilder] = None
        n_groups = 0
        tail_descs = []                                        # (name, descriptor): join the last group
        # Several groups (wgrad_group < NL; data parallel: AEW_WGRAD_GROUP=NL/2): the post-network and skip weight
        # gradients - operands complete before the chain starts - join the FIRST group instead of the last, and their
        # results are unpacked with it.  Every gradient from layer NL - wgrad_group up (the tail of the flat buffer,
        # `hi_first_layer`) is then final right after "unpack grads (decoder, upper layers)": a data-parallel caller
        # starts its reduce-scatter there, under the second half of the chain (TrainEngine.bwd_a1 / bwd_a2).
        multi = grouped and self.wgrad_group < NL
        if multi and self.wgrad_split_layers > 0:
            # (the upper-layers spk_bwd differences RUNNING column sums; layers kept as split-K ops deliver per-batch sums)
            raise ValueError("wgrad_split_layers > 0 cannot be combined with several grouped weight-gradient launches (wgrad_group < layers)")
        self.hi_first_layer = NL - self.wgrad_group if (multi and early_tbl is not None and snap_ok) else None
        self._spk_hi_from = None
        layers_in_grp = 0

        def group_add(name, t, tag):
            nonlocal grp
            if grp is None:
                grp = TnGroupBuilder(self.ws, p + f"tng{n_groups}", self._wgrad_tile())
                grp.cursor = self.wgrad_cursor
            ptr, stride = self._gslab(name, t.N_pad, t.K_total, 1)
            t.out, t.out_batch_stride = ptr, stride
            grp.add(t, "wgrad." + name)
            self.gbuf[name] = (ptr, stride, 1)
            return ptr, stride, 1

        def wgrad_late(name, dtype, Mc, N, N_pad, gseg, segs, tag, bias_grad=0):
            """A weight gradient whose operands exist early but whose result is only needed at the end: grouped mode
            defers it into the last group (one result, unpacked by the late table).  bias_grad: address of the layer's
            bias gradient = column sums of the G operand, a by-product of the grouped launch (0: none)."""
            if not grouped:
                return self._wgrad(plan, name, dtype, Mc, N, N_pad, gseg, segs, tag) + (pk.unpack_tbl,)
            t = make_tn(dtype, Mc, B, N, N_pad, gseg, segs)
            t.colsum_out = bias_grad or None
            if multi and name != "base":                       # (the base layer's needs dx of layer 0: last group)
                return group_add(name, t, tag) + (pk.unpack_tbl,)
            ptr, stride = self._gslab(name, t.N_pad, t.K_total, 1)
            t.out, t.out_batch_stride = ptr, stride
            tail_descs.append((name, t))
            self.gbuf[name] = (ptr, stride, 1)
            return ptr, stride, 1, late_tbl

        # ---- post network
        if ps.has(p + "post2.bias") and not grouped:
            self._colsum(plan, self.dlogits, w, Q, ps.ptr(p + "post2.bias", True), label="db.post2")
        gp, gs, gn, tbl = wgrad_late("p2", BF, w, Q, Qp, self.dlogits.seg(Qp), [self.h1.seg(Pp)], TAG_POST,
                                     bias_grad=ps.ptr(p + "post2.bias", True) if ps.has(p + "post2.bias") else 0)
        keep_tbl, pk.unpack_tbl = pk.unpack_tbl, tbl
        pk.rec(p + "post2.weight", 0, [P, 1], [Q, P], None, 0, [Pp, 1], g_ptr=gp, slabs=gn, slab_stride=gs)
        pk.unpack_tbl = keep_tbl
        plan.add(L.OP_GEMM_NT, make_nt(BF, w, Pp, Pp, B, [self.dlogits.seg(Qp)], self.Wp2T.ptr,
                                       flags=L.EF_MUL_POS1, out0=self.dh1.view(), aux1=self.h1.view(),
                                       impl=impl), "d.post2", TAG_POST)
        if ps.has(p + "post1.bias") and not grouped:
            self._colsum(plan, self.dh1, w, P, ps.ptr(p + "post1.bias", True), label="db.post1")
        gp, gs, gn, tbl = wgrad_late("p1", BF, w, P, Pp, self.dh1.seg(Pp), [self.h0.seg(Sp)], TAG_POST,
                                     bias_grad=ps.ptr(p + "post1.bias", True) if ps.has(p + "post1.bias") else 0)
        keep_tbl, pk.unpack_tbl = pk.unpack_tbl, tbl
        pk.rec(p + "post1.weight", 0, [S, 1], [P, S], None, 0, [Sp, 1], g_ptr=gp, slabs=gn, slab_stride=gs)
        pk.unpack_tbl = keep_tbl
        plan.add(L.OP_GEMM_NT, make_nt(BF, w, Sp, Sp, B, [self.dh1.seg(Pp)], self.Wp1T.ptr,
                                       flags=L.EF_MUL_POS1, out0=self.dskp.view(), aux1=self.h0.view(),
                                       impl=impl), "d.post1", TAG_POST)
        skp = None
        if multi:
            # skip weights of all layers (see below) into the first group: dskp and every z exist already
            skp = wgrad_late("skp_all", BF, w, S, Sp, self.dskp.seg(Sp),
                             [self.z[l].seg(Dp, row_off=g.layers[l].skip_lead) for l in range(NL)], TAG_WG_RS)
            for l in range(NL):
                pk.rec(p + f"conv_layers.{l}.dil_skp.weight", 0, [D, 1], [S, D], None, 0, [NL * Dp, 1],
                       g_ptr=skp[0], slabs=skp[2], slab_stride=skp[1], g_off=l * Dp)
        # ---- gated stack, last layer first
        Kfg = 2 * Rp + Cp
        Cc = Clc + self.Gc
        dx_next: Optional[Mat] = None
        colsum_tbl = CopyTableBuilder(self.ws, p + "tbl.colsum")
        # split_chains_bwd: the dz / dx chain as two independent half-batch chains on lanes 4 / 5 (the mirror of the
        # forward's split_chains; grouped weight gradients only, so that nothing else sits inside the chain)
        n_chains = 2 if (self.split_chains_bwd and grouped and not multi and snap_ok and self.wgrad_split_layers == 0
                         and B % 2 == 0 and self.n_lo >= NL) else 1
        nb = B // n_chains
        # (the tail branch is for the one-chain plan only: queued behind a chain on lane 4 it needs lanes 4 and 5 to wait
        # for each other in turn, and capturing that shape crashes inside the runtime - ROCm 7.2)
        tail = self.tail_lane_used = self.tail_lane if n_chains == 1 else 0
        for l in range(NL - 1, -1, -1):
            lg = g.layers[l]
            last = l == NL - 1
            P_l, d = lg.out_len, lg.dil
            q = p + f"conv_layers.{l}."
            segs = []
            if not last:
                segs.append(dx_next.seg(Rp, hi=P_l))
            segs.append(self.dskp.seg(Sp, row_off=-lg.skip_lead))
            for c in range(n_chains):
                b0 = c * nb
                plan.lane = (4 + c) if (n_chains > 1 and not self.split_one_lane) else 0
                csegs = segs if n_chains == 1 else \
                    ([] if last else [dx_next.seg(Rp, hi=P_l, b0=b0)]) + [self.dskp.seg(Sp, row_off=-lg.skip_lead, b0=b0)]
                plan.add(L.OP_GEMM_NT, make_nt(BF, P_l, Dp, Dp, nb, csegs, self.WrsT[l].ptr, epi=L.EPI_DFG,
                                               aux0=self.pf[l].view(b0=b0), aux1=self.pg[l].view(b0=b0),
                                               out0=self.dfg[l].view(b0=b0), impl=self._impl("dz")),
                         f"dz.{l}" + (f".c{c}" if n_chains > 1 else ""), TAG_DZ)
            plan.lane = 0
            if self.n_lo < NL and l == self.n_lo:
                hsegs = [self.dfg[k].seg(2 * Dp, row_off=-g.layers[k].cond_lead) for k in range(self.n_lo, NL)]
                with plan.side(self.EARLY_LANE):               # upper half of the cond gradient: inputs complete
                    plan.add(L.OP_GEMM_NT, make_nt(BF, T, Cp, Cp, B, hsegs, self.VfgT_hi.ptr,
                                                   out0=self.dcond_part.view(), impl=impl), "dcond_hi", TAG_DCOND)
            x = self.x[l]
            fg_segs = [x.seg(Rp), x.seg(Rp, row_off=d), self.cond.seg(Cp, row_off=lg.cond_lead)]
            if grouped and l < NL - self.wgrad_split_layers:
                if not last:
                    gp, gs, gn = group_add(f"res{l}", make_tn(BF, P_l, B, R, Rp, dx_next.seg(Rp, hi=P_l),
                                                              [self.z[l].seg(Dp)]), TAG_WG_RS)
                    pk.rec(q + "dil_res.weight", 0, [D, 1], [R, D], None, 0, [Dp, 1], g_ptr=gp, slabs=gn, slab_stride=gs)
                t = make_tn(BF, P_l, B, 2 * Dp, 2 * Dp, self.dfg[l].seg(2 * Dp), fg_segs)
                if snap_ok:
                    t.snap_out, t.snap_bs, t.snap_k = self.colsum_fg.data_ptr() + 4 * l * 2 * Dp, NL * 2 * Dp, (R if self.R < Rp else -1)
                gp, gs, gn = group_add(f"fg{l}", t, TAG_WG_FG)
            else:
                if not last:
                    gp, gs, gn = self._wgrad(plan, f"res{l}", BF, P_l, R, Rp, dx_next.seg(Rp, hi=P_l),
                                             [self.z[l].seg(Dp)], TAG_WG_RS)
                    pk.rec(q + "dil_res.weight", 0, [D, 1], [R, D], None, 0, [Dp, 1], g_ptr=gp, slabs=gn, slab_stride=gs)
                gp, gs, gn = self._wgrad(plan, f"fg{l}", BF, P_l, 2 * Dp, 2 * Dp, self.dfg[l].seg(2 * Dp), fg_segs,
                                         TAG_WG_FG)
            spb = gn // B if (gn % B == 0 and gn >= B) else 0     # slabs per batch (0: batch folded)
            if snap_ok and l < NL - self.wgrad_split_layers:
                pass                                               # running column sums come from the grouped wgrad
            elif self.R < Rp and spb > 0:
                # x carries a constant 1.0 in pad channel R (base_gather ones_channel), so column R
                # of this wgrad is sum_t dfg[t][n]: gather it per batch for the bias / speaker grads
                colsum_tbl.add(gp + 4 * R, self.colsum_fg.data_ptr() + 4 * l * 2 * Dp, [B, 2 * Dp],
                               [spb * gs, Kfg], [NL * 2 * Dp, 1], F3, F3, red_n=spb, red_stride=gs)
            else:
                self._colsum(plan, self.dfg[l], P_l, 2 * Dp, self.colsum_fg.data_ptr() + 4 * l * 2 * Dp,
                             out_bs=NL * 2 * Dp, label=f"colsum.dfg{l}")
            for gate, nm in ((0, "signal"), (1, "gate")):
                for co0, ng, gl in _gate_groups(D):
                    row0 = (co0 // 16) * 32 + 16 * gate
                    pk.rec(q + f"conv_{nm}.weight", co0 * R * 2, [16 * R * 2, R * 2, 2, 1], [ng, gl, R, 2],
                           None, row0 * Kfg, [32 * Kfg, Kfg, 1, Rp], g_ptr=gp, slabs=gn, slab_stride=gs)
                    pk.rec(q + f"proj_{nm}.weight", co0 * Cc, [16 * Cc, Cc, 1], [ng, gl, Clc],
                           None, row0 * Kfg + 2 * Rp, [32 * Kfg, Kfg, 1], g_ptr=gp, slabs=gn, slab_stride=gs)
            if grouped and l < NL - self.wgrad_split_layers:
                layers_in_grp += 1
            if grouped and grp is not None and layers_in_grp >= self.wgrad_group and l > 0:
                layers_in_grp = 0
                with plan.side(self._next_lane("tng")):
                    grp.emit(plan, f"wgrad.group{n_groups} (layers {l}.., skip, post)" if n_groups == 0 else
                             f"wgrad.group{n_groups} (layers {l}..)", TAG_WG_FG)
                grp = None
                n_groups += 1
                if early_tbl is not None:
                    if multi and snap_ok:
                        # gated-bias / speaker-projection gradients of the layers in this group (from the running column
                        # sums it just wrote): everything of layers >= l is final after the unpack below
                        sb_hi = L.SpkBwd()
                        self._fill_spk(sb_hi)
                        sb_hi.colsum, sb_hi.gc, sb_hi.grads = self.colsum_fg.data_ptr(), self.gc.data_ptr(), ps.grads.data_ptr()
                        sb_hi.colsum_running = NL
                        sb_hi.layer_range = l | ((NL - l) << 16)
                        self._spk_det(sb_hi, "hi")
                        self._spk_hi_from = l
                        with plan.side(1):
                            plan.add(L.OP_SPK_BWD, sb_hi, "spk_bwd (upper layers)", TAG_MISC, join=True)
                    with plan.side(1):                             # the first group's gradients: unpacked mid-chain
                        early_tbl.emit(plan, "unpack grads (decoder, upper layers)", join=True)
                    pk.unpack_tbl = late_tbl
                    early_tbl = None
            dx = self.dx[l]
            for c in range(n_chains):
                b0 = c * nb
                plan.lane = (4 + c) if (n_chains > 1 and not self.split_one_lane) else 0
                segs = [self.dfg[l].seg(2 * Dp, b0=b0), self.dfg[l].seg(2 * Dp, row_off=-d, b0=b0)]
                plan.add(L.OP_GEMM_NT, make_nt(
                    BF, lg.in_len, Rp, Rp, nb, segs, self.WfgT[l].ptr,
                    flags=0 if last else L.EF_ADD_AUX0,
                    out0=dx.view(hi=lg.in_len, b0=b0),
                    aux0=null_view() if last else dx_next.view(row_off=-d, hi=P_l, b0=b0), impl=self._impl("dx")),
                    f"dx.{l}" + (f".c{c}" if n_chains > 1 else ""), TAG_DX)
            plan.lane = 0
            dx_next = dx
            if not grouped and early_tbl is not None and l == NL // 2:
                with plan.side(1):                                 # after the wgrads issued so far, on any lane
                    early_tbl.emit(plan, "unpack grads (decoder, upper layers)", join=True)
                pk.unpack_tbl = late_tbl
                early_tbl = None
        pk.unpack_tbl = late_tbl
        dx0 = dx_next
        # ---- skip weights of all layers: ONE wgrad with NL segments (mirror of the deferred skip GEMM):
        # dW_skp[s][l*Dp + k] = sum_t dskp[t][s] * z_l[t + skip_lead_l][k].  640 tiles x batch fill the
        # chip without row splits, so it writes B slabs instead of ~128 per layer.
        # ---- base layer (wavenet.py:351)
        # (as a TN GEMM over a materialised one-hot matrix: 0.044 ms + 29 MB written by base_gather.  The scatter-add form
        # - rows of dx0 added into LDS images per 64 channels, AEW-internal experiment of round 2 - took 0.58 ms: the
        # rows have to be fetched one dependent (wav[t] -> dx0[t]) load pair at a time and the partial images merged
        # with ~10 M global atomics; the GEMM streams the same bytes at full rate)
        has_bb = ps.has(p + "base_layer.bias")
        if has_bb and not grouped:
            self._colsum(plan, dx0, T, R, ps.ptr(p + "base_layer.bias", True), label="db.base")
        gp_b, gs_b, gn_b, _tbl = wgrad_late("base", BF, T, R, Rp, dx0.seg(Rp), [self.onehot.seg(Qp)], TAG_MISC,
                                            bias_grad=ps.ptr(p + "base_layer.bias", True) if has_bb else 0)
        pk.rec(p + "base_layer.weight", 0, [Q, 1], [R, Q], None, 0, [Qp, 1], g_ptr=gp_b, slabs=gn_b, slab_stride=gs_b)
        if skp is None:
            gp, gs, gn, _tbl = wgrad_late("skp_all", BF, w, S, Sp, self.dskp.seg(Sp),
                                          [self.z[l].seg(Dp, row_off=g.layers[l].skip_lead) for l in range(NL)], TAG_WG_RS)
        if grouped:
            for name, t in tail_descs:
                if grp is None:
                    grp = TnGroupBuilder(self.ws, p + f"tng{n_groups}", self._wgrad_tile())
                    grp.cursor = self.wgrad_cursor
                grp.add(t, "wgrad." + name)
            with plan.side(tail or self._next_lane("tng")):
                grp.emit(plan, f"wgrad.group{n_groups} (last layers, skip, post)" if not multi else
                         f"wgrad.group{n_groups} (layers 0.., base)", TAG_WG_FG, join=n_chains > 1)
            grp = None
            if early_tbl is not None:                              # a single group: nothing was unpacked mid-chain
                pk.unpack_tbl = late_tbl
                late_tbl.recs.extend(early_tbl.recs)
                early_tbl = None
        for l in range(NL if skp is None else 0):
            pk.rec(p + f"conv_layers.{l}.dil_skp.weight", 0, [D, 1], [S, D], None, 0, [NL * Dp, 1],
                   g_ptr=gp, slabs=gn, slab_stride=gs, g_off=l * Dp)
        # ---- speaker / gated-bias gradients (tail_lane: right behind the last grouped launch, before any main-lane op,
        # so that the branch needs no second edge from the main lane)
        def emit_spk():
            sbw = L.SpkBwd()
            self._fill_spk(sbw)
            sbw.colsum, sbw.gc, sbw.grads = self.colsum_fg.data_ptr(), self.gc.data_ptr(), ps.grads.data_ptr()
            sbw.colsum_running = max(0, NL - self.wgrad_split_layers) if snap_ok else 0
            if getattr(self, "_spk_hi_from", None):                    # the upper layers were done after the first group
                sbw.layer_range = 0 | (self._spk_hi_from << 16)
            self._spk_det(sbw, "lo")
            with plan.side(tail or 1):                                 # reads the side lanes' wgrad slabs: side join
                colsum_tbl.emit(plan, "colsum.dfg (from wgrad column R)", join=True)
                plan.add(L.OP_SPK_BWD, sbw, "spk_bwd", TAG_MISC, join=not colsum_tbl.recs)
        if tail:
            emit_spk()
        # ---- conditioning gradient over all layers' dfg (wavenet.py:100-101 cond terms).  With split_multiseg the
        # layers [n_lo, NL) were summed on a side lane mid-chain (dcond_part); this GEMM adds them.
        lo = self.n_lo
        if lo < NL:
            segs = [self.dfg[l].seg(2 * Dp, row_off=-g.layers[l].cond_lead) for l in range(lo)]
            plan.add(L.OP_GEMM_NT, make_nt(BF, T, Cp, Cp, B, segs, self.VfgT_lo.ptr, flags=L.EF_ADD_AUX0,
                                           out0=self.dcond.view(), aux0=self.dcond_part.view(), impl=impl),
                     "dcond", TAG_DCOND, join=("lane", self.EARLY_LANE))
        else:
            segs = [self.dfg[l].seg(2 * Dp, row_off=-lg.cond_lead) for l, lg in enumerate(g.layers)]
            plan.add(L.OP_GEMM_NT, make_nt(BF, T, Cp, Cp, B, segs, self.VfgT_lo.ptr, out0=self.dcond.view(),
                                           impl=self._impl("dcond")), "dcond", TAG_DCOND,
                     join=n_chains > 1)          # two chains: the main lane meets them here at the latest (every lane mode)
        if not tail:
            emit_spk()
        # ---- upsamplers, last stage first (wavenet.py:154)
        n_ups = len(hps.lc_upsample_strides)
        ugrp = TnGroupBuilder(self.ws, p + "tng_ups", 128) if grouped else None     # upsampler + LC-conv wgrads: one launch

        def wgrad_ups(name, Mc, N, N_pad, gseg, segs, bias_grad=0, gmat=None):
            """One or a few output tiles with up to 8 x 1780 rows to contract: cut into chunks of ~512 rows (one block
            and one slab each) unless the whole contraction is that short (then the column-sum by-product is available)."""
            if ugrp is None:
                return self._wgrad(plan, name, BF, Mc, N, N_pad, gseg, segs, TAG_UPS)
            t = make_tn(BF, Mc, B, N, N_pad, gseg, segs)
            split = Mc * B > self.ups_split_rows
            if not split:
                t.colsum_out = bias_grad or None
            elif bias_grad:
                # a split descriptor has no block that sees every row: the bias gradient (column sums of the G
                # operand) comes from a column-sum op, as in the ungrouped plan
                self._colsum(plan, gmat, Mc, N, bias_grad, label="db." + name)
            slabs = ugrp.set_split(t, 512) if split else 1
            ptr, stride = self._gslab(name, t.N_pad, t.K_total, slabs)
            t.out, t.out_batch_stride = ptr, stride
            ugrp.add(t, "wgrad." + name)
            self.gbuf[name] = (ptr, stride, slabs)
            return ptr, stride, slabs

        for i in range(n_ups - 1, -1, -1):
            f, s = hps.lc_upsample_filt_sizes[i], hps.lc_upsample_strides[i]
            pad = f - s
            X, dX = self.ups_in[i], self.dups[i]
            last = i == n_ups - 1
            dY = self.dcond if last else self.dups[i + 1]
            trim0 = g.trim_ups_out[0] if last else 0
            self._colsum(plan, dY, dY.rows, Clc, ps.ptr(p + f"lc_upsample.{i}.tconv.bias", True),
                         label=f"db.up{i}")
            segs = [dY.seg(Cp, row_step=s, row_off=k - pad - trim0) for k in range(f)]
            gp, gs, gn = wgrad_ups(f"up{i}", X.rows, Clc, Cp, X.seg(Cp), segs)
            pk.rec(p + f"lc_upsample.{i}.tconv.weight", 0, [Clc * f, f, 1], [Clc, Clc, f], None, 0,
                   [f * Cp, 1, Cp], g_ptr=gp, slabs=gn, slab_stride=gs)
            plan.add(L.OP_GEMM_NT, make_nt(BF, X.rows, Cp, Cp, B, segs, self.WupT[i].ptr, out0=dX.view(),
                                           impl=impl), f"d.ups{i}", TAG_UPS)
        # ---- LC conv (wavenet.py:337)
        dlc1, lc1 = self.dups[0], self.ups_in[0]
        has_lb = ps.has(p + "lc_conv.bias")
        if has_lb and ugrp is None:
            self._colsum(plan, dlc1, lc1.rows, Clc, ps.ptr(p + "lc_conv.bias", True), label="db.lc")
        nin = self.n_lc_in
        gp, gs, gn = wgrad_ups("lc", lc1.rows, Clc, Cp, dlc1.seg(Cp), [self.lcj.seg(Lp, row_off=t) for t in range(3)],
                               bias_grad=ps.ptr(p + "lc_conv.bias", True) if has_lb else 0, gmat=dlc1)
        if ugrp is not None:
            with plan.side(self._next_lane("tng")):
                ugrp.emit(plan, "wgrad.group (upsamplers, lc conv)", TAG_UPS)
        pk.rec(p + "lc_conv.weight", 0, [nin * 3, 3, 1], [Clc, nin, 3], None, 0, [3 * Lp, 1, Lp],
               g_ptr=gp, slabs=gn, slab_stride=gs)
        plan.add(L.OP_GEMM_NT, make_nt(BF, self.Ne, ru(nin, 8), Lp, B,
                                       [dlc1.seg(Cp, row_off=-t) for t in range(3)], self.WlcT.ptr,
                                       out0=self.dlcj.view(), impl=impl), "d.lc_conv", TAG_UPS)
        # ---- jitter scatter back to the LC source
        if self.Ne > 4096:                                         # (up to 4096 conditioning vectors per window the scatter runs in
            plan.zero(self.ws, self.dlc_src.name)                  #  its gather form, which writes every element: no atomics, no zeroing)
        sc = L.LcScatter()
        sc.d, sc.d_bs, sc.d_pitch = self.dlcj.ptr, self.dlcj.bs, self.dlcj.pitch
        sc.jitter, sc.jit_pitch = self.jitter.data_ptr(), self.jitter.shape[1]
        sc.dsrc, sc.dsrc_bs, sc.dsrc_pitch = self.dlc_src.ptr, self.dlc_src.bs, self.dlc_src.pitch
        sc.B, sc.N, sc.C, sc.take_compat = B, self.Ne, nin, int(self.take_compat)
        plan.add(L.OP_LC_SCATTER, sc, "lc_scatter", TAG_UPS)


# ------------------------------------------------------------------------------------------
# encoder + bottleneck (fp32 exact chain)
# ------------------------------------------------------------------------------------------
def det_colsum(ws: Workspace, cs: "L.Colsum", name: str):
    """Scratch + tickets of the deterministic form of a column-sum op (aew_colsum_t.det_scratch / det_tickets: partial sums per
    row chunk, added in a fixed order by the last arriver - no fp32 atomics).  The tickets start at zero and every launch
    leaves them at zero."""
    import ctypes as _C
    nf, nt = _C.c_int64(0), _C.c_int32(0)
    L.check(L.load().aew_colsum_det_size(_C.byref(cs), _C.byref(nf), _C.byref(nt)), "aew_colsum_det_size")
    cs.det_scratch = ws.alloc(name + ".scratch", max(4, nf.value), torch.float32).data_ptr()
    cs.det_tickets = ws.alloc(name + ".tickets", max(4, nt.value), torch.int32, zero=True).data_ptr()


def exact_split_args(ws: Workspace, name: str, S: int, cin: int, k_total: int, rows: int, n_pad: int) -> dict:
    """make_nt keywords for the split-K form of an exact fp32 GEMM (aew_gemm_nt_t.k_split): S contiguous k-ranges on
    separate workgroups, combined in a fixed order - the canonical summation order of the op, oracle/exact.py ksplit_for
    states the same rule: the input channels need no padding (cin a multiple of 64) and the K axis is a multiple of 32 * S.
    Allocates the partial-sum slabs and the (self-resetting) tickets.  {} = not split."""
    if S not in (2, 4) or cin % 64 or k_total % (32 * S):
        return {}
    rows_pad = ru(rows, 32)
    wsb = ws.alloc(name + ".ws", S * rows_pad * n_pad, torch.float32)
    tk = ws.alloc(name + ".tickets", (rows_pad // 16) * (n_pad // 64) * 4, torch.int32)
    return dict(k_split=S, ksplit_ws_ptr=wsb.data_ptr(), ksplit_tickets_ptr=tk.data_ptr())


class EncoderPlan:
    """Encoder (wave_encoder.py:34-103).  Forward: fp32, exact k-ascending chains (bit-exact code indices against
    oracle/exact_chain.c).  Backward: bf16 operands, fp32 accumulation on the bf16 MFMA kernels - the forward's
    exactness contract does not extend to gradients, and the fp32 kernels spent 0.48 ms per step here (9.5 % of the
    step for 0.3 % of its FLOPs: one serial chain per output, every 16-row tile re-streaming the weight matrix).  The
    forward's epilogue writes what the backward reads as bf16: the activations (wgrad operand), the pre-activation
    (relu mask)."""

    k_split = 0               # 2 | 4: the exact fp32 GEMMs (encoder layers, bottleneck linear) as S contiguous k-ranges on separate
                              # workgroups with a fixed-order combine (aew_gemm_nt_t.k_split) - a DIFFERENT canonical summation
                              # order, which oracle/exact.py's KSPLIT must then state too (the GPU test sets both).  Built for
                              # the round-4 review's item 5 and measured: enc.# 0.266 ms unsplit, 0.291 ms with S = 4 (720
                              # workgroups instead of 180: the serial K loop per workgroup is a quarter, the launch is not
                              # faster - these layers are bound by what a workgroup streams per K tile, profiles/r04_notes.md
                              # 9), fwd_a 0.379 / 0.370 / 0.387 ms for S = 0 / 2 / 4.  Off: the order stays one chain per output

    def __init__(self, ws: Workspace, ps: ParamStore, hps, geom: G.ModelGeom, B: int, n_mel: int,
                 mel_cl: Mat, packer: Packer, impl: int = 0, in_tbl: Optional[CopyTableBuilder] = None,
                 in_mel: Optional[torch.Tensor] = None, wgrad_group: Optional[int] = None):
        self.ws, self.ps, self.hps, self.g, self.B, self.impl = ws, ps, hps, geom, B, impl
        self.wgrad_group = DecoderPlan.wgrad_group if wgrad_group is None else int(wgrad_group)
        self.pk = packer
        self.n_mel, self.Mp, self.Mb = n_mel, ru(n_mel, 64), ru(n_mel, 128)
        self.E, self.Ep, self.Eb = hps.enc_n_out, ru(hps.enc_n_out, 64), ru(hps.enc_n_out, 128)
        self.lens = geom.enc_lens                      # 10 entries
        E, Ep, Eb = self.E, self.Ep, self.Eb
        self.y: List[Mat] = [mel_cl]                   # fp32 activations (forward chain)
        self.yb: List[Mat] = [Mat.new(ws, "enc.yb0", B, self.lens[0], self.Mb, BF)]    # bf16 copies (wgrad operand)
        if in_tbl is not None and in_mel is not None:
            in_tbl.add(in_mel.data_ptr(), self.yb[0].ptr, [B, geom.mel_len, n_mel], [n_mel * geom.mel_len, 1, geom.mel_len],
                       [self.yb[0].bs, self.Mb, 1], F3, BF)
        self.r: List[Optional[Mat]] = [None]           # bf16 pre-activations (relu mask of the backward)
        for i in range(9):
            self.y.append(Mat.new(ws, f"enc.y{i + 1}", B, self.lens[i + 1], Ep, F3))
            self.yb.append(Mat.new(ws, f"enc.yb{i + 1}", B, self.lens[i + 1], Eb, BF))
            self.r.append(Mat.new(ws, f"enc.r{i + 1}", B, self.lens[i + 1], Eb, BF))
        self.dy = [Mat.new(ws, f"enc.dy{i}", B, self.lens[i], self.Mb if i == 0 else Eb, BF) for i in range(10)]
        self.dpre = [None] + [Mat.new(ws, f"enc.dpre{i}", B, self.lens[i], Eb, BF) for i in range(1, 10)]
        self.zero_cnt = ws.alloc("enc.zero_cnt", 9, torch.int64)
        self.W, self.WT, self.bias = [], [], []
        cin, cinp, cinb = n_mel, self.Mp, self.Mb
        for i, (f, s) in enumerate(zip(G.ENCODER_FILTERS, G.ENCODER_STRIDES)):
            nm = f"encoder.net.{i}.conv."
            Wm = Mat.new(ws, f"enc.wp.{i}", 1, Ep, f * cinp, F3)
            packer.rec(nm + "weight", 0, [cin * f, f, 1], [E, cin, f], Wm, 0, [f * cinp, 1, cinp], first=(i == 0))
            self.W.append(Wm)
            # dgrad layouts, bf16: [ci][k*Eb + co] <- W[co][ci][k]  (per output phase for strided layers)
            if s == 1:
                WT = Mat.new(ws, f"enc.wpT.{i}", 1, cinb, f * Eb, BF)
                packer.rec(nm + "weight", 0, [cin * f, f, 1], [E, cin, f], WT, 0, [1, f * Eb, Eb], late=True)
                self.WT.append([WT])
            else:
                phs = []
                for ph in range(s):
                    WT = Mat.new(ws, f"enc.wpT.{i}.{ph}", 1, cinb, (f // s) * Eb, BF)
                    # [ci][j*Eb + co] <- W[co][ci][ph + s*j]
                    packer.rec(nm + "weight", ph, [cin * f, f, s], [E, cin, f // s], WT, 0, [1, (f // s) * Eb, Eb],
                               late=True)
                    phs.append(WT)
                self.WT.append(phs)
            bt = ws.alloc(f"enc.wp.bias{i}", Ep, torch.float32)
            (packer.first_tbl if (i == 0 and packer.first_tbl is not None) else packer.pack_tbl).add(
                ps.ptr(nm + "bias"), bt.data_ptr(), [E], [1], [1], F3, F3)
            self.bias.append(bt)
            cin, cinp, cinb = E, Ep, Eb
        self.gbuf = {}

    def build_forward(self, plan: Plan, join_before_layer1=False):
        """join_before_layer1: join spec (see Plan.add) for the layer-1 GEMM - the side lane that packs the weights of
        layers 1.. while layer 0 (whose own weights were packed on the main lane) runs."""
        B, impl, Ep = self.B, self.impl, self.Ep
        plan.zero(self.ws, "enc.zero_cnt")
        cinp = self.Mp
        for i, (f, s, res) in enumerate(zip(G.ENCODER_FILTERS, G.ENCODER_STRIDES, G.ENCODER_RESIDUAL)):
            X, Y, Rm = self.y[i], self.y[i + 1], self.r[i + 1]
            segs = [X.seg(cinp, row_step=s, row_off=k) for k in range(f)]
            flags = L.EF_BIAS | L.EF_RELU | L.EF_OUT1_PRE | L.EF_COUNT_ZERO | L.EF_OUT2_COPY | (L.EF_ADD_AUX0 if res else 0)
            cin = self.n_mel if i == 0 else self.E
            skw = exact_split_args(self.ws, f"enc.ks{i}", self.k_split if impl == 0 else 0, cin, f * cinp, Y.rows * B, Ep)
            plan.add(L.OP_GEMM_NT, make_nt(
                F3, Y.rows, self.E, Ep, B, segs, self.W[i].ptr, flags=flags, out0=Y.view(), out1=Rm.view(),
                out2=self.yb[i + 1].view(),
                aux0=X.view(row_off=(f - 1) // 2) if res else null_view(), bias_ptr=self.bias[i].data_ptr(),
                counter_ptr=self.zero_cnt.data_ptr() + 8 * i, impl=impl, **skw), f"enc.{i}", TAG_ENC,
                join=join_before_layer1 if i == 1 else False)
            cinp = Ep

    def build_backward(self, plan: Plan, need_input_grad: bool = True):
        """Expects dy[9] and dpre[9] already written (by the bottleneck backward)."""
        B, impl, E, Eb, ps, pk = self.B, self.impl, self.E, self.Eb, self.ps, self.pk
        # grouped mode (DecoderPlan.wgrad_group): the nine weight gradients as ONE launch after the dgrad chain (each
        # contracts over 8 x 29..70 rows: nine launches of a few microseconds of work each otherwise), the bias
        # gradients as its column-sum by-product
        grp = TnGroupBuilder(self.ws, "enc.tng", 128) if (self.wgrad_group > 0 and impl == 0) else None
        for i in range(8, -1, -1):
            f, s, res = G.ENCODER_FILTERS[i], G.ENCODER_STRIDES[i], G.ENCODER_RESIDUAL[i]
            X = self.yb[i]
            cin, cinb = (self.n_mel, self.Mb) if i == 0 else (E, Eb)
            dpre, dyo = self.dpre[i + 1], self.dy[i + 1]
            Lo = dpre.rows
            t = make_tn(BF, Lo, B, E, Eb, dpre.seg(Eb), [X.seg(cinb, row_step=s, row_off=k) for k in range(f)],
                        impl=impl)
            if grp is None:
                cs = L.Colsum()
                cs.x = dpre.seg(128)
                cs.dtype, cs.M, cs.N, cs.batch = BF, Lo, E, B
                cs.out, cs.out_bs, cs.accumulate = ps.ptr(f"encoder.net.{i}.conv.bias", True), 0, 1
                det_colsum(self.ws, cs, f"det.db.enc{i}")
                with plan.side(1 + (2 * i) % DecoderPlan.n_side_lanes):
                    plan.add(L.OP_COLSUM, cs, f"db.enc{i}", TAG_ENC)
            slabs = L.tn_slabs(t) if grp is None else 1
            gt = self.ws.alloc(f"enc.wg.{i}", slabs * Eb * t.K_total, torch.float32)
            t.out, t.out_batch_stride = gt.data_ptr(), Eb * t.K_total
            if grp is None:
                with plan.side(1 + (2 * i + 1) % DecoderPlan.n_side_lanes):
                    plan.add(L.OP_GEMM_TN, t, f"wgrad.enc{i}", TAG_ENC)
            else:
                t.colsum_out = ps.ptr(f"encoder.net.{i}.conv.bias", True)
                grp.add(t, f"wgrad.enc{i}")
            pk.rec(f"encoder.net.{i}.conv.weight", 0, [cin * f, f, 1], [E, cin, f], None, 0,
                   [f * cinb, 1, cinb], g_ptr=gt.data_ptr(), slabs=slabs, slab_stride=Eb * t.K_total)
            if i == 0 and not need_input_grad:
                continue
            dX = self.dy[i]
            Li = dX.rows
            lw = (f - 1) // 2
            for ph in range(s):
                Mq = (Li - ph + s - 1) // s
                if Mq <= 0:
                    continue
                segs = [dpre.seg(Eb, row_off=-j) for j in range(f // s)] if s > 1 else \
                       [dpre.seg(Eb, row_off=-k) for k in range(f)]
                flags = (L.EF_ADD_AUX0 if res else 0) | (L.EF_OUT1_POS1 if i > 0 else 0)
                plan.add(L.OP_GEMM_NT, make_nt(
                    BF, Mq, ru(cin, 8), cinb, B, segs, self.WT[i][ph].ptr, flags=flags,
                    out0=dX.view(row_step=s, row_off=ph),
                    out1=self.dpre[i].view(row_step=s, row_off=ph) if i > 0 else null_view(),
                    aux0=dyo.view(row_step=s, row_off=ph - lw) if res else null_view(),
                    aux1=self.r[i].view(row_step=s, row_off=ph) if i > 0 else null_view(), impl=impl),
                    f"d.enc{i}.ph{ph}", TAG_ENC)
        if grp is not None:
            with plan.side(1):
                grp.emit(plan, "wgrad.group (encoder)", TAG_ENC)
