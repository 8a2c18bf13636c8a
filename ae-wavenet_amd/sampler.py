"""Autoregressive sampler: host side of aew_sampler_run (include/aewavenet.h, csrc/aew_sampler.hip).

Replaces WaveNet.forward_test (wavenet.py:367-531) as driven by InferenceChassis (chassis.py:283-349): given the
conditioning rows of every time step (the same lc_conv -> upsampler path the training forward computes,
wavenet.py:379-391) it generates the mu-law sequence sample by sample.  Differences from the reference, on purpose:
  * no per-sample Python / kernel launches: ONE persistent kernel whose wavefronts keep all weights in registers;
  * streams are generated 16 at a time (one MFMA tile); several 16-stream batches pipeline through the layers
    (the reference's n_replicas maps to streams);
  * the draw is an inverse-CDF lookup driven by a counter RNG (seed, stream, position) instead of
    torch.multinomial, so a generation is reproducible and checkable (oracle/jitter_rng.py has the numpy form);
  * priming: positions with forced[stream][t] >= 0 are fed as given (teacher forcing), the rest are drawn; the
    reference primes with the first `base_global_rf` samples of the input wav in the same way (wavenet.py:423-431).

This module only builds descriptors and packs weights (layout + bf16 cast); all arithmetic of a generation runs in
the HIP kernel.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib as L


def _ru(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def _frag_blob(W: torch.Tensor, nk: int) -> torch.Tensor:
    """W [<=32 rows][<=nk*32] fp32 -> MFMA A-fragment order [nk][2][64][8] bf16:
    blob[k][n][lane][j] = W[n*16 + (lane & 15)][k*32 + (lane >> 4)*8 + j]."""
    Wp = torch.zeros(32, nk * 32, dtype=torch.float32, device=W.device)
    Wp[:W.shape[0], :W.shape[1]] = W
    return Wp.view(2, 16, nk, 4, 8).permute(2, 0, 3, 1, 4).contiguous().view(nk, 2, 64, 8).to(torch.bfloat16)


class SamplerGeometry:
    """Channel padding and actor counts for one decoder configuration."""

    def __init__(self, hps):
        self.R, self.D, self.S, self.P, self.Q = hps.n_res, hps.n_dil, hps.n_skp, hps.n_post, hps.n_quant
        self.Clc, self.Gc = hps.n_lc_out, hps.n_global_embed
        self.dils = [2 ** l for _ in range(hps.n_blocks) for l in range(hps.n_block_layers)]
        self.NL = len(self.dils)
        self.Rk, self.Dk, self.Sk, self.Pk, self.Ck = (_ru(v, 32) for v in (self.R, self.D, self.S, self.P, self.Clc))
        self.n_pairs = _ru(self.D, 16) // 16                       # EARLY / LATE actors per layer
        self.n_res = (_ru(self.R, 16) // 16 + 1) // 2              # RES actors per layer (2 tiles each)
        self.n_skp = (_ru(self.S, 16) // 16 + 1) // 2
        self.n_p1 = (_ru(self.P, 16) // 16 + 1) // 2
        self.n_p2 = (self.Q // 16 + 1) // 2
        self.n_smp = 4
        self.kr_max = 12 if self.Rk <= 384 else 16
        if self.Rk > 512 or self.Dk > 256 or self.Sk > 256 or self.Pk > 256 or self.Ck > 128:
            raise L.AewError("sampler: channel counts beyond the compiled bounds (R<=512, D/S/P<=256, lc<=128)")
        if self.Q % 16 or self.Q > 256:
            raise L.AewError("sampler: n_quant must be a multiple of 16, at most 256")
        if max(self.n_pairs, self.n_res, self.n_skp, self.n_p1, self.n_p2) > 32:
            raise L.AewError("sampler: more than 32 producers in one wait set")

    def n_actors(self) -> int:
        return (self.NL * (2 * self.n_pairs + self.n_skp) + (self.NL - 1) * self.n_res + self.n_p1 + self.n_p2
                + self.n_smp)

    def rf(self) -> int:
        return sum(self.dils)


class Sampler:
    """Packs the decoder weights once; generate() runs one generation.

    ps: ParamStore (fp32 parameters by reference name), pre: 'decoder.' / 'wavenet.' prefix."""

    def __init__(self, hps, ps, pre: str, device, flag_stride: int = 64):
        self.g = g = SamplerGeometry(hps)
        self.hps, self.pre, self.dev = hps, pre, torch.device(device)
        self.flag_stride = flag_stride
        self.nap_eighths = 0                                       # napping between items: measured neutral (r01_sampler.txt)
        self.lib = L.load()
        self._pack(ps)

    # ---- weights ---------------------------------------------------------------------------------------------
    def _pack(self, ps):
        g, p = self.g, self.pre
        blobs: List[torch.Tensor] = []
        self.off: Dict[Tuple, int] = {}                             # (kind, layer, index) -> element offset

        def put(key, W, nk):
            self.off[key] = sum(b.numel() for b in blobs)
            blobs.append(_frag_blob(W.to(self.dev), nk).reshape(-1))

        kr, kc, kd = g.Rk // 32, g.Ck // 32, g.Dk // 32
        for l in range(g.NL):
            q = p + f"conv_layers.{l}."
            ws, wg = ps.view(q + "conv_signal.weight").float(), ps.view(q + "conv_gate.weight").float()   # [D][R][2]
            cs, cg = ps.view(q + "proj_signal.weight").float(), ps.view(q + "proj_gate.weight").float()   # [D][Clc+Gc][1]
            for pi in range(g.n_pairs):
                r0, r1 = 16 * pi, min(16 * pi + 16, g.D)
                for tap, kind in ((0, "early"), (1, "late")):     # tap 0 = h(t-d), tap 1 = h(t)   (wavenet.py:100-101)
                    W = torch.zeros(32, g.Rk, device=self.dev)
                    W[:r1 - r0, :g.R] = ws[r0:r1, :, tap]
                    W[16:16 + r1 - r0, :g.R] = wg[r0:r1, :, tap]
                    put((kind, l, pi), W, kr)
                W = torch.zeros(32, g.Ck, device=self.dev)
                W[:r1 - r0, :g.Clc] = cs[r0:r1, :g.Clc, 0]
                W[16:16 + r1 - r0, :g.Clc] = cg[r0:r1, :g.Clc, 0]
                put(("cond", l, pi), W, kc)
            if l < g.NL - 1:
                wr = ps.view(q + "dil_res.weight").float()[:, :, 0]                                       # [R][D]
                for qi in range(g.n_res):
                    put(("res", l, qi), wr[32 * qi:32 * qi + 32], kd)
            wk = ps.view(q + "dil_skp.weight").float()[:, :, 0]                                           # [S][D]
            for qi in range(g.n_skp):
                put(("skip", l, qi), wk[32 * qi:32 * qi + 32], kd)
        w1 = ps.view(p + "post1.weight").float()[:, :, 0]                                                 # [P][S]
        for qi in range(g.n_p1):
            put(("post1", 0, qi), w1[32 * qi:32 * qi + 32], g.Sk // 32)
        w2 = ps.view(p + "post2.weight").float()[:, :, 0]                                                 # [Q][P]
        for qi in range(g.n_p2):
            put(("post2", 0, qi), w2[32 * qi:32 * qi + 32], g.Pk // 32)
        self.blob = torch.cat(blobs)
        # shared biases (fp32, padded to whole actors) and the base-layer table [Q][Rk] bf16 (wavenet.py:348-351, 452)
        self.b1 = torch.zeros(g.n_p1 * 32, device=self.dev)
        self.b2 = torch.zeros(g.n_p2 * 32, device=self.dev)
        if ps.has(p + "post1.bias"):
            self.b1[:g.P] = ps.view(p + "post1.bias").float()
        if ps.has(p + "post2.bias"):
            self.b2[:g.Q] = ps.view(p + "post2.bias").float()
        base = ps.view(p + "base_layer.weight").float()[:, :, 0].t().contiguous()                         # [Q][R]
        if ps.has(p + "base_layer.bias"):
            base = base + ps.view(p + "base_layer.bias").float()[None, :]
        self.base_t = torch.zeros(g.Q, g.Rk, dtype=torch.bfloat16, device=self.dev)
        self.base_t[:, :g.R] = base.to(torch.bfloat16)

    # ---- one generation ----------------------------------------------------------------------------------------
    def generate(self, cond: torch.Tensor, bias: torch.Tensor, forced: torch.Tensor, seed: int = 0,
                 want_logits: bool = False, spin_max: int = 0, stream: int = 0, timing: bool = False,
                 profile: bool = False, dry_run: bool = False):
        """cond   bf16 [n_streams][>= T][>= Ck] (zero-padded channels; any row / stream strides that are multiples
                  of 8 elements): the upsampled local conditioning at every position (engine: dec.cond)
        bias   fp32 [n_streams][NL][>= n_pairs*32]: per-stream gated bias, (16 filt | 16 gate) per channel tile,
                  speaker term folded in (engine: dec.bias_bl as written by the spk_bias op)
        forced int32 [n_streams][T]: >= 0 feeds that value at the position, < 0 draws; column 0 must be >= 0
        Returns (wav int32 [n_streams][T], logits fp32 [n_streams][T][Q] or None).  logits[:, t] is the
        distribution of position t+1 given positions <= t.  timing: HIP-event time of the kernel in self.last
        (stream must be torch's current stream, i.e. 0 = the default stream).  dry_run: build the actor table only
        (works with CPU tensors) and return a description of it - tests/test_sampler.py replays the protocol from
        that under random schedules."""
        g = self.g
        n_streams, T = forced.shape
        if n_streams % 16:
            raise L.AewError("sampler: streams come in batches of 16")
        nb = n_streams // 16
        if cond.dtype != torch.bfloat16 or bias.dtype != torch.float32 or forced.dtype != torch.int32:
            raise L.AewError("sampler: cond bf16, bias fp32, forced int32 expected")
        if cond.shape[0] != n_streams or cond.shape[1] < T or cond.shape[2] < g.Ck or cond.stride(2) != 1:
            raise L.AewError("sampler: cond shape")
        if cond.stride(0) % 8 or cond.stride(1) % 8 or cond.data_ptr() % 16:
            raise L.AewError("sampler: cond alignment")
        if bias.shape[0] != n_streams or bias.shape[1] != g.NL or bias.shape[2] < g.n_pairs * 32 or bias.stride(2) != 1:
            raise L.AewError("sampler: bias shape")
        if bool((forced[:, 0] < 0).any()):
            raise L.AewError("sampler: position 0 must be given")
        dev, fs = self.dev, self.flag_stride
        z = lambda *shape, dt=torch.bfloat16: torch.zeros(*shape, dtype=dt, device=dev)
        rings = [d + 1 for d in g.dils]
        hbuf = [z(nb, rings[l], 16, g.Rk) for l in range(g.NL)]
        epart = [z(nb, 2, g.n_pairs, 64, 8, dt=torch.float32) for _ in range(g.NL)]
        zbuf = [z(nb, 16, g.Dk) for _ in range(g.NL)]
        skp = z(nb, 16, g.Sk, dt=torch.float32)
        p1 = z(nb, 16, g.Pk)
        logits = z(nb, 16, g.Q, dt=torch.float32)
        wav_out = torch.full((n_streams, T), -1, dtype=torch.int32, device=dev)
        logits_out = z(n_streams, T, g.Q, dt=torch.float32) if want_logits else None
        forced = forced.to(dev).contiguous()

        def sbuf(t: Optional[torch.Tensor], byte_off=0, *, bstride=0, entry=0, pitch=0, ring=1, layout=0) -> L.Sbuf:
            s = L.Sbuf()
            s.ptr = (t.data_ptr() + byte_off) if t is not None else None
            s.bstride, s.entry, s.pitch, s.ring, s.layout = bstride, entry, pitch, ring, layout
            return s

        # rows written by several actors are producer-contiguous (aew_sbuf_t layouts 1 / 2): h_l and relu(post1) as
        # [32-channel group][stream][32 ch], z_l as [16-channel pair][stream][16 ch]
        def h_at(l, group=0):                                        # h_l entry, or the block of one 32-channel group
            return sbuf(hbuf[l], 1024 * group, bstride=rings[l] * 16 * g.Rk * 2, entry=16 * g.Rk * 2, pitch=64,
                        ring=rings[l], layout=1)

        def z_at(l, pair=0):
            return sbuf(zbuf[l], 512 * pair, bstride=16 * g.Dk * 2, pitch=32, layout=2)

        # ---- actors, grouped so that each producer group owns consecutive flags --------------------------------
        acts: List[L.Actor] = []
        xcd_of: List[int] = []
        groups: Dict[Tuple, Tuple[int, int]] = {}                    # (kind, layer) -> (first flag id, count)

        def new(role, layer, index, xcd) -> L.Actor:
            a = L.Actor()
            a.role, a.layer, a.index, a.nt = role, layer, index, 2
            acts.append(a)
            xcd_of.append(xcd)
            return a

        def group(kind, layer, n):
            groups[(kind, layer)] = (len(acts), n)

        def wptr(key):
            return self.blob.data_ptr() + 2 * self.off[key]

        kr, kc, kd = g.Rk // 32, g.Ck // 32, g.Dk // 32
        # consecutive layers share an XCD (its L2 serves their hand-offs); only every third hand-off crosses XCDs
        xcd_layer = [(l * 8) // (g.NL + 1) for l in range(g.NL + 1)]
        post_xcd = xcd_layer[g.NL]
        for l, d in enumerate(g.dils):
            x = xcd_layer[l]
            group("early", l, g.n_pairs)
            for pi in range(g.n_pairs):
                a = new(L.ACT_EARLY, l, pi, x)
                a.nk, a.nk2, a.dil = kr, kc, d
                a.w, a.w2 = wptr(("early", l, pi)), wptr(("cond", l, pi))
                a.bias = bias.data_ptr() + 4 * (l * bias.stride(1) + pi * 32)
                a.bias_pitch = bias.stride(0)
                a.in0 = h_at(l)
                a.in1 = sbuf(cond, bstride=16 * cond.stride(0) * 2, entry=cond.stride(1) * 2, pitch=cond.stride(0) * 2,
                             ring=max(T, 1))
                a.out = sbuf(epart[l], pi * 2048, bstride=2 * g.n_pairs * 2048, entry=g.n_pairs * 2048, ring=2)
            group("late", l, g.n_pairs)
            for pi in range(g.n_pairs):
                a = new(L.ACT_LATE, l, pi, x)
                a.nk = kr
                a.w = wptr(("late", l, pi))
                a.in0 = h_at(l)
                a.in1 = sbuf(epart[l], pi * 2048, bstride=2 * g.n_pairs * 2048, entry=g.n_pairs * 2048, ring=2)
                a.out = z_at(l, pi)
            if l < g.NL - 1:
                group("res", l, g.n_res)
                for qi in range(g.n_res):
                    a = new(L.ACT_RES, l, qi, x)
                    a.nk, a.nt = kd, min(2, _ru(g.R, 16) // 16 - 2 * qi)
                    a.w = wptr(("res", l, qi))
                    a.in0, a.in1, a.out = z_at(l), h_at(l, qi), h_at(l + 1, qi)
            group("skip", l, g.n_skp)
            for qi in range(g.n_skp):
                a = new(L.ACT_SKIP, l, qi, x)
                a.nk, a.nt = kd, min(2, _ru(g.S, 16) // 16 - 2 * qi)
                a.w = wptr(("skip", l, qi))
                a.in0 = z_at(l)
                sk = sbuf(skp, 4 * 32 * qi, bstride=16 * g.Sk * 4, pitch=g.Sk * 4)
                a.in1 = sk if l > 0 else sbuf(None)
                a.out = sk
        group("post1", 0, g.n_p1)
        for qi in range(g.n_p1):
            a = new(L.ACT_POST1, g.NL, qi, post_xcd)
            a.nk, a.nt = g.Sk // 32, min(2, _ru(g.P, 16) // 16 - 2 * qi)
            a.w, a.bias = wptr(("post1", 0, qi)), self.b1.data_ptr() + 4 * 32 * qi
            a.in0 = sbuf(skp, bstride=16 * g.Sk * 4, pitch=g.Sk * 4)
            a.out = sbuf(p1, 1024 * qi, bstride=16 * g.Pk * 2, pitch=64, layout=1)
        group("post2", 0, g.n_p2)
        for qi in range(g.n_p2):
            a = new(L.ACT_POST2, g.NL, qi, post_xcd)
            a.nk, a.nt = g.Pk // 32, min(2, g.Q // 16 - 2 * qi)
            a.w, a.bias = wptr(("post2", 0, qi)), self.b2.data_ptr() + 4 * 32 * qi
            a.in0 = sbuf(p1, bstride=16 * g.Pk * 2, pitch=64, layout=1)
            a.out = sbuf(logits, 4 * 32 * qi, bstride=16 * g.Q * 4, pitch=g.Q * 4)
            if logits_out is not None:
                a.out2 = sbuf(logits_out, 4 * 32 * qi, bstride=16 * T * g.Q * 4, entry=g.Q * 4, pitch=T * g.Q * 4, ring=T)
        group("sample", 0, g.n_smp)
        for qi in range(g.n_smp):
            a = new(L.ACT_SAMPLE, g.NL, qi, post_xcd)
            a.w = self.base_t.data_ptr()
            a.n_quant, a.row_bytes = g.Q, g.Rk * 2
            a.in0 = sbuf(logits, bstride=16 * g.Q * 4, pitch=g.Q * 4)
            a.out = h_at(0)
        n_act = len(acts)
        assert n_act == g.n_actors()
        # slots: workgroup L runs on XCD L % 8; a layer's actors share an XCD (and its L2)
        per_xcd: List[List[int]] = [[] for _ in range(8)]
        for k, x in enumerate(xcd_of):
            per_xcd[x].append(k)
        depth = max(len(v) for v in per_xcd)
        n_slots = 8 * depth
        flags = torch.zeros(n_slots * fs, dtype=torch.int32, device=dev)     # flag ids < n_actors <= n_slots
        status = torch.zeros(4, dtype=torch.int32, device=dev)

        def wait(kind, layer, lag, only: Optional[int] = None) -> L.Wait:
            first, n = groups[(kind, layer)]
            w = L.Wait()
            if only is not None:
                first, n = first + only, 1
            w.flags, w.n, w.lag = flags.data_ptr() + 4 * fs * first, n, lag
            return w

        for k, a in enumerate(acts):
            a.flag = flags.data_ptr() + 4 * fs * k
            l = a.layer
            h_src = ("sample", 0) if l == 0 else ("res", l - 1)      # who publishes h_l
            if a.role == L.ACT_EARLY:
                a.wait[0] = wait(*h_src, a.dil)                      # h_l(t - d)
                a.wait[1] = wait("late", l, 2, only=a.index)         # its partial slot (t & 1) is free again
            elif a.role == L.ACT_LATE:
                a.wait[0] = wait(*h_src, 0)
                a.wait[1] = wait("early", l, 0, only=a.index)
            elif a.role == L.ACT_RES:
                a.wait[0] = wait("late", l, 0)
            elif a.role == L.ACT_SKIP:
                a.wait[0] = wait("late", l, 0)
                if l > 0:
                    a.wait[1] = wait("skip", l - 1, 0, only=a.index)
            elif a.role == L.ACT_POST1:
                a.wait[0] = wait("skip", g.NL - 1, 0)
            elif a.role == L.ACT_POST2:
                a.wait[0] = wait("post1", 0, 0)
            elif a.role == L.ACT_SAMPLE:
                a.wait[0] = wait("post2", 0, 0)
        table = (L.Actor * n_slots)()
        for s in range(n_slots):
            table[s].role = L.ACT_NONE
        for x in range(8):
            for pos, k in enumerate(per_xcd[x]):
                C.memmove(C.byref(table, (x + 8 * pos) * C.sizeof(L.Actor)), C.byref(acts[k]), C.sizeof(L.Actor))
        if dry_run:
            bufs = {f"h{l}": hbuf[l] for l in range(g.NL)}
            bufs.update({f"z{l}": zbuf[l] for l in range(g.NL)})
            bufs.update({f"e{l}": epart[l] for l in range(g.NL)})
            bufs.update(skp=skp, p1=p1, logits=logits)
            return dict(table=table, n_slots=n_slots, nb=nb, T=T, flag_stride=fs, flags=flags.data_ptr(),
                        buffers={k: (v.data_ptr(), v.numel() * v.element_size()) for k, v in bufs.items()}, keep=bufs)
        raw = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(dev)
        sp = L.Sampler()
        sp.actors, sp.n_slots, sp.n_batches, sp.n_steps = raw.data_ptr(), n_slots, nb, T
        sp.flag_stride, sp.kr_max, sp.spin_max = fs, g.kr_max, spin_max
        sp.flags, sp.status = flags.data_ptr(), status.data_ptr()
        sp.forced, sp.wav_out, sp.seed = forced.data_ptr(), wav_out.data_ptr(), seed & ((1 << 64) - 1)
        sp.nap_eighths = self.nap_eighths
        prof = torch.zeros(n_slots, 4, dtype=torch.int64, device=dev) if profile else None
        sp.prof = prof.data_ptr() if profile else None
        self.last = dict(n_slots=n_slots, n_actors=n_act, depth=depth)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)] if timing else None
        if ev:
            ev[0].record()
        L.check(self.lib.aew_sampler_run(C.byref(sp), stream), "aew_sampler_run")
        if ev:
            ev[1].record()
            ev[1].synchronize()
            self.last["kernel_ms"] = ev[0].elapsed_time(ev[1])
        st = status.cpu().tolist()                                   # synchronises; keeps every buffer above alive
        if profile:                                                  # per-role phase clock (s_memtime runs at 100 MHz)
            pc = prof.cpu()
            roles: Dict[int, List[int]] = {}
            for sl in range(n_slots):
                if table[sl].role >= 0 and int(pc[sl, 3]) > 0:
                    roles.setdefault((table[sl].role, min(table[sl].layer, 1)), []).append(sl)
            names = {0: "EARLY", 1: "LATE", 2: "RES", 3: "SKIP", 4: "POST1", 5: "POST2", 6: "SAMPLE"}
            self.last["profile"] = {
                f"{names[r]}{'' if r > 3 else ('.0' if l0 == 0 else '.l')}": [
                    float(pc[sl_list, c].double().mean() / pc[sl_list, 3].double().mean()) * 0.01 for c in range(3)]
                for (r, l0), sl_list in sorted(roles.items())}           # us per item: wait, work, publish
        if st[0]:
            raise L.AewError(f"sampler: actor in slot {st[1]} gave up waiting at t={st[2]}, batch {st[3]}")
        return wav_out, logits_out


def from_engine(eng, flag_stride: int = 64) -> Sampler:
    """Sampler over the decoder of a TrainEngine (same parameters, same prefix)."""
    return Sampler(eng.hps, eng.ps, eng.dec.pre, eng.device, flag_stride)


def engine_conditioning(eng) -> Tuple[torch.Tensor, torch.Tensor]:
    """(cond, bias) of the engine's last forward(): the conditioning rows dec.cond [B][T][Cp] bf16 and the gated bias
    dec.bias_bl [B][NL][2*Dp] fp32 (spk_bias op) - exactly what the training GEMMs read."""
    d = eng.dec
    cond = d.cond.tensor()
    bias = d.bias_bl[:eng.B * d.NL * 2 * d.Dp].view(eng.B, d.NL, 2 * d.Dp)
    return cond, bias
