"""TEST INFRASTRUCTURE — CPU interpreter for launch plans (include/aewavenet.h semantics).

Executes the very ctypes `aew_op_t` records a TrainEngine built on device 'cpu' would send
to `aew_run_plan`, using torch CPU ops.  It exists so that plan construction (segment tables,
row maps, pack/unpack records, epilogue flags) can be verified against the oracle in the
GPU-less CI container.  It is not part of the product: ae-wavenet_amd/ never imports it and
has no CPU execution path.
"""
import ctypes as C

import torch

from ae_wavenet_amd import _lib as L

BF, F3 = L.BF16, L.F32


class Emu:
    def __init__(self, ws):
        self.ws = ws

    # ---- raw memory helpers ---------------------------------------------------------------
    def flat(self, ptr):
        name, off = self.ws.resolve(ptr)
        return self.ws.get(name), off

    def rd(self, ptr, idx, dtype=None):
        t, off = self.flat(ptr)
        return t[off + idx].float() if t.dtype in (torch.bfloat16, torch.float32) else t[off + idx]

    def wr(self, ptr, idx, val):
        t, off = self.flat(ptr)
        t[off + idx] = val.to(t.dtype)

    # ---- views ------------------------------------------------------------------------------
    @staticmethod
    def vrows(v, m):
        row = m * v.row_step + v.row_off
        ok = (row >= v.row_lo) & (row < v.row_hi)
        return row, ok

    def vload(self, v, b, m, n0, ncols):
        """[len(m)][ncols] float32, zero where the view row is invalid."""
        row, ok = self.vrows(v, m)
        out = torch.zeros(m.numel(), ncols)
        if ok.any():
            idx = (b * v.batch_stride + row[ok] * v.row_pitch)[:, None] + n0 + torch.arange(ncols)[None, :]
            out[ok] = self.rd(v.ptr, idx)
        return out

    def vstore(self, v, b, m, n0, val, col_ok=None):
        row, ok = self.vrows(v, m)
        if not ok.any():
            return
        ncols = val.shape[1]
        idx = (b * v.batch_stride + row[ok] * v.row_pitch)[:, None] + n0 + torch.arange(ncols)[None, :]
        self.wr(v.ptr, idx, val[ok])

    def seg_matrix(self, s, b, M, esize_dtype):
        m = torch.arange(M)
        row = m * s.row_step + s.row_off
        ok = (row >= s.row_lo) & (row < s.row_hi)
        A = torch.zeros(M, s.k_len)
        if ok.any():
            idx = (b * s.batch_stride + row[ok] * s.row_pitch)[:, None] + torch.arange(s.k_len)[None, :]
            A[ok] = self.rd(s.ptr, idx)
        return A

    # ---- ops --------------------------------------------------------------------------------
    def run(self, plan):
        for op, lab in zip(plan.array(), plan.labels):     # the array is what aew_run_plan receives
            try:
                getattr(self, "op_%d" % op.kind)(getattr(op.u, L.OP_FIELD[op.kind]))
            except Exception as e:
                raise RuntimeError(f"emulator failed at op '{lab}': {e}") from e

    def op_1(self, g):   # GEMM_NT
        Wt, woff = self.flat(g.W)
        W = Wt[woff:woff + g.N_pad * g.K_total].float().view(g.N_pad, g.K_total)
        m = torch.arange(g.M)
        for b in range(g.batch):
            A = torch.cat([self.seg_matrix(g.seg[s], b, g.M, g.dtype) for s in range(g.n_segs)], dim=1)
            Cm = A @ W.t()                                     # [M][N_pad]
            if g.epi == L.EPI_STORE:
                self._epi_store(g, b, m, Cm[:, :g.N])
            elif g.epi == L.EPI_GATED:
                Dp = g.N_pad // 2
                Cp = Cm.view(g.M, Dp // 16, 2, 16)
                bias, boff = self.flat(g.bias)
                bb = bias[boff + b * g.bias_bs: boff + b * g.bias_bs + g.N_pad].view(Dp // 16, 2, 16)
                f = (Cp[:, :, 0, :] + bb[None, :, 0, :]).reshape(g.M, Dp)[:, :g.N]
                q = (Cp[:, :, 1, :] + bb[None, :, 1, :]).reshape(g.M, Dp)[:, :g.N]
                a, s = torch.tanh(f), torch.sigmoid(q)
                self.vstore(g.out0, b, m, 0, a * s)
                self.vstore(g.out1, b, m, 0, s * (1 - a * a))
                self.vstore(g.out2, b, m, 0, a * s * (1 - s))
                if g.W2:
                    # fused gated layer (aewavenet.h): residual 1x1 over the z just STORED (its storage rounding
                    # included: the kernel's LDS z tile holds the same bf16 values) + the aux0 addend -> out3
                    K2 = g.N_pad // 2
                    W2t, w2off = self.flat(g.W2)
                    W2 = W2t[w2off:w2off + g.N2_pad * K2].float().view(g.N2_pad, K2)
                    zs = torch.zeros(g.M, K2)
                    zs[:, :g.N] = self.vload(g.out0, b, m, 0, g.N)
                    res = (zs @ W2.t())[:, :g.N2]             # (the product shape of the two-op form: same BLAS blocking)
                    if g.aux0.ptr:
                        res = res + self.vload(g.aux0, b, m, 0, g.N2)
                    self.vstore(g.out3, b, m, 0, res)
            elif g.epi == L.EPI_RES_SKIP:
                ns = g.n_split
                if ns > 0:
                    res = Cm[:, :ns] + self.vload(g.aux0, b, m, 0, ns)
                    self.vstore(g.out0, b, m, 0, res)
                sk = Cm[:, ns:g.N]
                if g.flags & L.EF_ACCUM:
                    sk = sk + self.vload(g.out1, b, m, 0, sk.shape[1])
                self.vstore(g.out1, b, m, 0, sk)
                if g.flags & L.EF_OUT2_RELU:
                    self.vstore(g.out2, b, m, 0, torch.relu(sk))
            elif g.epi == L.EPI_DFG:
                dz = Cm[:, :g.N]
                df = dz * self.vload(g.aux0, b, m, 0, g.N)
                dg = dz * self.vload(g.aux1, b, m, 0, g.N)
                out = torch.stack((df.view(g.M, g.N // 16, 16), dg.view(g.M, g.N // 16, 16)), dim=2)
                self.vstore(g.out0, b, m, 0, out.reshape(g.M, 2 * g.N))

    def _epi_store(self, g, b, m, v):
        fl = g.flags
        N = v.shape[1]
        if fl & L.EF_BIAS:
            bias, boff = self.flat(g.bias)
            v = v + bias[boff + b * g.bias_bs: boff + b * g.bias_bs + N][None, :]
        if fl & L.EF_RELU:
            v = torch.relu(v)
        if fl & L.EF_OUT1_PRE:
            self.vstore(g.out1, b, m, 0, v)
        if fl & L.EF_ADD_AUX0:
            v = v + self.vload(g.aux0, b, m, 0, N)
        if fl & L.EF_RELU_POST:
            v = torch.relu(v)
        if fl & (L.EF_MUL_POS1 | L.EF_OUT1_POS1):
            a = self.vload(g.aux1, b, m, 0, N)
            wv = torch.where(a > 0, v, torch.zeros_like(v))
            if fl & L.EF_OUT1_POS1:
                self.vstore(g.out1, b, m, 0, wv)
            if fl & L.EF_MUL_POS1:
                v = wv
        if fl & L.EF_COUNT_ZERO:
            row, ok = self.vrows(g.out0, m)
            t, off = self.flat(g.counter)
            t[off] += int((v[ok] == 0).sum())
        self.vstore(g.out0, b, m, 0, v)
        if fl & L.EF_OUT2_COPY:
            self.vstore(g.out2, b, m, 0, v)

    def op_2(self, t):   # GEMM_TN
        slabs = L.tn_slabs(t)
        out, ooff = self.flat(t.out)
        # the split/fold heuristic only changes how partial sums are distributed over slabs;
        # the emulator writes the per-batch sum into slab b (or everything into slab 0 when the
        # library folds the batch) and zero elsewhere
        fold = bool(L.load().aew_tn_fold(C.byref(t)))
        region = out[ooff: ooff + slabs * t.out_batch_stride]
        region.zero_()
        for b in range(t.batch):
            Gm = self.seg_matrix(self._with_k(t.g, t.N_pad), b, t.Mc, t.dtype)
            A = torch.cat([self.seg_matrix(t.seg[s], b, t.Mc, t.dtype) for s in range(t.n_segs)], dim=1)
            dW = Gm.t() @ A                                     # [N_pad][K_total]
            sl = 0 if fold else b * (slabs // t.batch)
            o = ooff + sl * t.out_batch_stride
            out[o:o + t.N_pad * t.K_total] += dW.reshape(-1)

    def op_25(self, p):  # NT_CHAIN: the stage ops that follow it in the plan are the chain (aewavenet.h) - they run one by one
        pass

    def op_24(self, p):  # GEMM_TN_GROUP: every descriptor contracted over all rows of all batch elements, one result
        rt, roff = self.flat(p.descs)
        nbytes = p.n_descs * C.sizeof(L.GemmTN)
        raw = bytes(rt[roff:roff + (nbytes + 7) // 8].numpy().tobytes())
        descs = (L.GemmTN * p.n_descs).from_buffer_copy(raw[:nbytes])
        tm = self.rd(p.tile_map, torch.arange(p.n_blocks)).tolist()
        covered = {}
        for rec in tm:
            if rec >= 0:
                covered.setdefault(int(rec) >> 22, []).append(((int(rec) >> 12) & 0x3ff, int(rec) & 0xfff))
        for d, t in enumerate(descs):
            # the tile map must name every (chunk, output tile) of every descriptor exactly once
            k128, n128 = t.K_total // 128, t.N_pad // 128
            if p.tile == 384:                                  # 8-wave tiles: 128 x 256 or 256 x 128 by the descriptor's shape
                ori = 1 if (t.K_total % 256 == 0 and t.N_pad % 256 != 0) else 0
                n_tiles = ((k128 + 1) // 2) * n128 if ori else k128 * ((n128 + 1) // 2)
            else:
                n_tiles = k128 * n128 if p.tile == 128 else ((k128 + 1) // 2) * ((n128 + 1) // 2)
            n_chunks = t.grp_splits * t.batch if t.grp_splits > 0 else 1
            assert sorted(covered.get(d, [])) == [(c, tl) for c in range(n_chunks) for tl in range(n_tiles)], f"tile map of descriptor {d}"
            out, ooff = self.flat(t.out)
            if t.grp_splits > 0:                               # partial sums per (batch element, row chunk)
                for b in range(t.batch):
                    Gm = self.seg_matrix(self._with_k(t.g, t.N_pad), b, t.Mc, t.dtype)
                    A = torch.cat([self.seg_matrix(t.seg[s], b, t.Mc, t.dtype) for s in range(t.n_segs)], dim=1)
                    for sp in range(t.grp_splits):
                        r0, r1 = sp * t.grp_rows, min(t.Mc, (sp + 1) * t.grp_rows)
                        o = ooff + (b * t.grp_splits + sp) * t.out_batch_stride
                        out[o:o + t.N_pad * t.K_total] = (Gm[r0:r1].t() @ A[r0:r1]).reshape(-1)
                continue
            dW = torch.zeros(t.N_pad, t.K_total)
            for b in range(t.batch):
                Gm = self.seg_matrix(self._with_k(t.g, t.N_pad), b, t.Mc, t.dtype)
                A = torch.cat([self.seg_matrix(t.seg[s], b, t.Mc, t.dtype) for s in range(t.n_segs)], dim=1)
                dW += Gm.t() @ A
                if t.snap_out and t.snap_k >= 0:
                    self.wr(t.snap_out, b * t.snap_bs + torch.arange(t.N_pad), dW[:, t.snap_k])
                elif t.snap_out:                               # snap_k = -1: running column sums of G
                    run_cs = Gm.sum(0) + (run_cs if b else 0.0)
                    self.wr(t.snap_out, b * t.snap_bs + torch.arange(t.N_pad), run_cs)
            out[ooff:ooff + t.N_pad * t.K_total] = dW.reshape(-1)
            if t.colsum_out:
                cs = torch.zeros(t.N_pad)
                for b in range(t.batch):
                    cs += self.seg_matrix(self._with_k(t.g, t.N_pad), b, t.Mc, t.dtype).sum(0)
                self.wr(t.colsum_out, torch.arange(t.N), cs[:t.N])

    @staticmethod
    def _with_k(seg, k):
        s = L.Seg()
        C.memmove(C.byref(s), C.byref(seg), C.sizeof(L.Seg))
        s.k_len = k
        return s

    def op_3(self, tb):  # COPY_TABLE
        rt, roff = self.flat(tb.recs)
        raw = bytes(rt[roff:roff + (tb.n_recs * C.sizeof(L.CopyRec) + 7) // 8].numpy().tobytes())
        recs = (L.CopyRec * tb.n_recs).from_buffer_copy(raw[:tb.n_recs * C.sizeof(L.CopyRec)])
        for r in recs:
            d = list(r.dims)
            grids = torch.meshgrid(*[torch.arange(x) for x in d], indexing="ij")
            so = sum(gr * s for gr, s in zip(grids, r.ss)).reshape(-1)
            do = sum(gr * s for gr, s in zip(grids, r.ds)).reshape(-1)
            acc = torch.zeros(so.numel())
            for q in range(r.red_n):
                acc = acc + self.rd(r.src, so + q * r.red_stride)
            acc = acc * r.scale
            if r.accumulate:
                acc = acc + self.rd(r.dst, do)
            self.wr(r.dst, do, acc)

    def op_4(self, p):   # VQ_NEAREST
        idx = torch.arange(p.Q)[:, None] * p.d_pitch + torch.arange(p.d)[None, :]
        z = self.rd(p.ze, idx)
        emb = self.rd(p.emb, torch.arange(p.K * p.d)).view(p.K, p.d)
        diff = z[:, None, :] - emb[None, :, :]
        dd = (diff ** 2).sum(-1)
        if p.metric == 0:
            dist = dd.sqrt() / ((z ** 2).sum(-1).sqrt()[:, None] + (emb ** 2).sum(-1).sqrt()[None, :])
        else:
            dist = dd
        md, mi = dist.min(dim=1)
        self.wr(p.ind, torch.arange(p.Q), mi)
        self.wr(p.dist, torch.arange(p.Q), md)
        zq = torch.zeros(p.Q, p.d_pitch)
        zq[:, :p.d] = emb[mi]
        self.wr(p.zq, torch.arange(p.Q * p.d_pitch), zq.reshape(-1))

    def op_5(self, p):   # VQ_STATS
        idx = torch.arange(p.Q)[:, None] * p.d_pitch + torch.arange(p.d)[None, :]
        z = self.rd(p.ze, idx)
        ind = self.rd(p.ind, torch.arange(p.Q))
        zs = torch.zeros(p.K, p.d).index_add_(0, ind, z)
        ns = torch.zeros(p.K).index_add_(0, ind, torch.ones(p.Q))
        self.wr(p.z_sum, torch.arange(p.K * p.d), zs.reshape(-1))
        self.wr(p.n_sum, torch.arange(p.K), ns)
        if p.hist:
            self.wr(p.hist, torch.arange(p.K), self.rd(p.hist, torch.arange(p.K)) + ns)

    def _guarded(self, p) -> bool:
        """aew_adam_t.guard / aew_vq_ema_t.guard: a non-zero device word turns the op into a no-op"""
        return bool(p.guard) and int(self.rd(p.guard, torch.arange(1))[0]) != 0

    def op_6(self, p):   # VQ_EMA
        if self._guarded(p):
            return
        kd = torch.arange(p.K * p.d)
        k = torch.arange(p.K)
        nu = p.gamma * self.rd(p.numer, kd) + p.gamma_comp * self.rd(p.z_sum, kd)
        de = p.gamma * self.rd(p.denom, k) + p.gamma_comp * self.rd(p.n_sum, k)
        self.wr(p.numer, kd, nu)
        self.wr(p.denom, k, de)
        if p.update_codebook == 1:
            self.wr(p.emb, kd, (nu.view(p.K, p.d) / de[:, None]).reshape(-1))
        elif p.update_codebook == 2:                             # k-means centroid step: empty codes stay
            old = self.rd(p.emb, kd).view(p.K, p.d)
            new = torch.where(de[:, None] > 0, nu.view(p.K, p.d) / de[:, None].clamp_min(1e-30), old)
            self.wr(p.emb, kd, new.reshape(-1))

    def _gmul(self, p) -> float:
        """optional device scalar: upstream d(L)/d(loss) of the backward call (aewavenet.h, `gmul`)"""
        return float(self.rd(p.gmul, torch.arange(1))[0]) if p.gmul else 1.0

    def op_7(self, p):   # VQ_BWD
        idx = torch.arange(p.Q)[:, None] * p.d_pitch + torch.arange(p.d)[None, :]
        z = self.rd(p.ze, idx)
        ind = self.rd(p.ind, torch.arange(p.Q))
        emb_rows = self.rd(p.emb, ind[:, None] * p.d + torch.arange(p.d)[None, :])
        dzq = self.rd(p.dzq, idx)
        t = z - emb_rows
        if p.metric == 0:
            u = (t ** 2).sum(-1, keepdim=True).sqrt()
            zn = (z ** 2).sum(-1, keepdim=True).sqrt()
            v = zn + (emb_rows ** 2).sum(-1, keepdim=True).sqrt()
            dj = t / (u * v) - u * z / (v * v * zn)
        else:
            dj = 2 * t
        gm = self._gmul(p)
        out = torch.zeros(p.Q, p.d_pitch)
        out[:, :p.d] = dzq + p.coef * gm * dj
        self.wr(p.dze, torch.arange(p.Q * p.d_pitch), out.reshape(-1))
        if p.demb:
            ft, off = self.flat(p.demb)
            K = int(ind.max()) + 1
            acc = torch.zeros(K, p.d).index_add_(0, ind, -2.0 * t * p.demb_coef * gm)
            ft[off: off + K * p.d] += acc.reshape(-1)

    def _gather_index(self, p, B, N, C_):
        j = self.rd(p.jitter, torch.arange(B)[:, None] * p.jit_pitch + torch.arange(N)[None, :])   # B,N
        j = j.clamp(0, N - 1)                    # the kernels clamp (the reference's Jitter can emit N at position N-1)
        b = torch.arange(B)[:, None, None].expand(B, N, C_)
        c = torch.arange(C_)[None, None, :].expand(B, N, C_)
        jj = j[:, :, None].expand(B, N, C_)
        if p.take_compat:
            flat = b * N + jj
            return flat // (C_ * N), (flat // N) % C_, flat % N
        return b, c, jj

    def op_8(self, p):   # LC_GATHER
        bb, cc, nn = self._gather_index(p, p.B, p.N, p.C)
        v = self.rd(p.src, bb * p.src_bs + nn * p.src_pitch + cc)
        out = torch.zeros(p.B, p.N, p.C_pad)
        out[:, :, :p.C] = v
        idx = (torch.arange(p.B)[:, None, None] * p.dst_bs + torch.arange(p.N)[None, :, None] * p.dst_pitch
               + torch.arange(p.C_pad)[None, None, :])
        self.wr(p.dst, idx, out)

    def op_9(self, p):   # LC_SCATTER
        bb, cc, nn = self._gather_index(p, p.B, p.N, p.C)
        idx = (torch.arange(p.B)[:, None, None] * p.d_bs + torch.arange(p.N)[None, :, None] * p.d_pitch
               + torch.arange(p.C)[None, None, :])
        gsrc = self.rd(p.d, idx)
        t, off = self.flat(p.dsrc)
        if p.N <= 4096:
            # gather form (k_lc_scatter_det): every element of dsrc[b][j][0:C] is WRITTEN, the plan does not zero the target
            tgt = (torch.arange(p.B)[:, None, None] * p.dsrc_bs + torch.arange(p.N)[None, :, None] * p.dsrc_pitch
                   + torch.arange(p.C)[None, None, :])
            t[(off + tgt).reshape(-1)] = 0
        t.index_add_(0, (off + bb * p.dsrc_bs + nn * p.dsrc_pitch + cc).reshape(-1), gsrc.reshape(-1).to(t.dtype))

    def _spk_common(self, p):
        par, poff = self.flat(p.params)
        voice = self.rd(p.voice, torch.arange(p.B))
        tb = lambda ptr: self.rd(ptr, torch.arange(p.L))
        return par, poff, voice, tb(p.off_bias_sig), tb(p.off_bias_gate), tb(p.off_proj_sig), tb(p.off_proj_gate)

    @staticmethod
    def _pack_idx(D):
        co = torch.arange(D)
        return (co // 16) * 32 + co % 16

    def op_10(self, p):  # SPK_BIAS
        par, poff, voice, obs, obg, ops_, opg = self._spk_common(p)
        Cc = p.C_lc + p.G
        Wsp = par[poff + p.off_spk_w: poff + p.off_spk_w + p.G * p.n_speakers].view(p.G, p.n_speakers)
        gc = Wsp[:, voice].t().clone()                                          # B,G
        if p.off_spk_b >= 0:
            gc = gc + par[poff + p.off_spk_b: poff + p.off_spk_b + p.G][None, :]
        self.wr(p.gc, torch.arange(p.B * p.G), gc.reshape(-1))
        out = torch.zeros(p.B, p.L, 2 * p.D_pad)
        pi = self._pack_idx(p.D)
        for l in range(p.L):
            for gate, ob, ov in ((0, obs[l], ops_[l]), (1, obg[l], opg[l])):
                V = par[poff + ov: poff + ov + p.D * Cc].view(p.D, Cc)[:, p.C_lc:]     # D,G
                v = gc @ V.t()                                                # B,D
                if ob >= 0:
                    v = v + par[poff + ob: poff + ob + p.D][None, :]
                out[:, l, pi + 16 * gate] = v
        self.wr(p.bias, torch.arange(out.numel()), out.reshape(-1))

    def op_11(self, p):  # SPK_BWD
        par, poff, voice, obs, obg, ops_, opg = self._spk_common(p)
        gr, goff = self.flat(p.grads)
        Cc = p.C_lc + p.G
        cs = self.rd(p.colsum, torch.arange(p.B * p.L * 2 * p.D_pad)).view(p.B, p.L, 2 * p.D_pad)
        if p.colsum_running:                                    # lowest layers: sums over batch elements 0..b -> per element
            r = min(p.colsum_running, p.L)
            cs = cs.clone()
            cs[1:, :r] = cs[1:, :r] - self.rd(p.colsum, torch.arange(p.B * p.L * 2 * p.D_pad)).view(p.B, p.L, 2 * p.D_pad)[:-1, :r]
        gc = self.rd(p.gc, torch.arange(p.B * p.G)).view(p.B, p.G)
        pi = self._pack_idx(p.D)
        dgc = torch.zeros(p.B, p.G)
        l0 = p.layer_range & 0xffff
        ln = (p.layer_range >> 16) if p.layer_range else p.L
        for l in range(l0, l0 + ln):
            for gate, ob, ov in ((0, obs[l], ops_[l]), (1, obg[l], opg[l])):
                c = cs[:, l, pi + 16 * gate]                                   # B,D
                if ob >= 0:
                    gr[goff + ob: goff + ob + p.D] = c.sum(0)
                V = par[poff + ov: poff + ov + p.D * Cc].view(p.D, Cc)[:, p.C_lc:]
                gv = gr[goff + ov: goff + ov + p.D * Cc].view(p.D, Cc)
                gv[:, p.C_lc:] = c.t() @ gc
                dgc += c @ V
        gw = gr[goff + p.off_spk_w: goff + p.off_spk_w + p.G * p.n_speakers].view(p.G, p.n_speakers)
        for b in range(p.B):
            gw[:, voice[b]] += dgc[b]
        if p.off_spk_b >= 0:
            gr[goff + p.off_spk_b: goff + p.off_spk_b + p.G] += dgc.sum(0)

    def op_12(self, p):  # BASE_GATHER
        q = self.rd(p.wav, torch.arange(p.B)[:, None] * p.wav_pitch + p.wav_off + torch.arange(p.T)[None, :]).long()
        W = self.rd(p.W, torch.arange(p.R * p.Q)).view(p.R, p.Q)
        x = torch.zeros(p.B, p.T, p.R_pad)
        x[:, :, :p.R] = W.t()[q]
        if p.bias:
            x[:, :, :p.R] += self.rd(p.bias, torch.arange(p.R))[None, None, :]
        if p.ones_channel:
            x[:, :, p.R] = 1.0
        idx = (torch.arange(p.B)[:, None, None] * p.x_bs + torch.arange(p.T)[None, :, None] * p.x_pitch
               + torch.arange(p.R_pad)[None, None, :])
        self.wr(p.x, idx, x)
        if p.onehot:
            oh = torch.nn.functional.one_hot(q, p.Q_pad).float()
            idx = (torch.arange(p.B)[:, None, None] * p.oh_bs + torch.arange(p.T)[None, :, None] * p.oh_pitch
                   + torch.arange(p.Q_pad)[None, None, :])
            self.wr(p.onehot, idx, oh)

    def op_13(self, p):  # SOFTMAX_NLL
        idx = (torch.arange(p.B)[:, None, None] * p.bs + torch.arange(p.w)[None, :, None] * p.pitch
               + torch.arange(p.Q)[None, None, :])
        lg = self.rd(p.logits, idx)
        tgt = torch.zeros(p.B, p.w, dtype=torch.long)
        tgt[:, :p.w - 1] = self.rd(p.wav, torch.arange(p.B)[:, None] * p.wav_pitch + p.tgt_off + 1
                                   + torch.arange(p.w - 1)[None, :]).long()
        lsm = torch.log_softmax(lg, -1)
        live = torch.ones(p.B, p.w)
        live[:, p.w - 1] = 0
        if not p.backward:
            lp = torch.gather(lsm, 2, tgt[:, :, None]).squeeze(2)
            self.wr(p.nll, torch.arange(p.B * p.w), (-lp * live).reshape(-1))
            if p.ptgt:
                self.wr(p.ptgt, torch.arange(p.B * p.w), (lp.exp() * live).reshape(-1))
            if p.peak:
                pk, am = lsm.max(-1)
                self.wr(p.peak, torch.arange(p.B * p.w), pk.reshape(-1))
                self.wr(p.amax, torch.arange(p.B * p.w), am.reshape(-1).int())
        else:
            g = (lsm.exp() - torch.nn.functional.one_hot(tgt, p.Q).float()) * (p.scale * self._gmul(p)) * live[:, :, None]
            out = torch.zeros(p.B, p.w, p.Q_pad)
            out[:, :, :p.Q] = g
            idx = (torch.arange(p.B)[:, None, None] * p.dl_bs + torch.arange(p.w)[None, :, None] * p.dl_pitch
                   + torch.arange(p.Q_pad)[None, None, :])
            self.wr(p.dlogits, idx, out)

    def op_14(self, p):  # COLSUM
        t, off = self.flat(p.out)
        for b in range(p.batch):
            if not p.accumulate:
                t[off + b * p.out_bs: off + b * p.out_bs + p.N] = 0
        for b in range(p.batch):
            A = self.seg_matrix(self._with_k(p.x, p.N), b, p.M, p.dtype)
            t[off + b * p.out_bs: off + b * p.out_bs + p.N] += A.sum(0)

    def op_15(self, p):  # REDUCE
        tot = 0.0
        t, off = self.flat(p.out)
        for i in range(p.n_terms):
            v = self.rd(p.x[i], torch.arange(p.n[i])).sum() * p.scale[i]
            t[off + 1 + i] = v
            ps = float(self.rd(p.post_scale_dev[i], torch.arange(1))[0]) if p.post_scale_dev[i] else p.post_scale[i]
            tot = tot + (ps * torch.clamp(v, min=p.clamp_min[i]) if p.clamp[i] else v)
        t[off] = tot

    def op_16(self, a):  # ADAM
        if self._guarded(a):
            return
        n = torch.arange(a.n)
        p, g, m, v = (self.rd(x, n) for x in (a.p, a.g, a.m, a.v))
        g = g * a.grad_scale
        m = a.beta1 * m + (1 - a.beta1) * g
        v = a.beta2 * v + (1 - a.beta2) * g * g
        p = p - (a.lr / a.bc1) * m / (v.sqrt() / (a.bc2 ** 0.5) + a.eps)
        self.wr(a.p, n, p); self.wr(a.m, n, m); self.wr(a.v, n, v)

    def op_17(self, z):  # ZERO
        name, off = self.ws.resolve(z.ptr)
        t = self.ws.get(name)
        t[off: off + z.bytes // t.element_size()] = 0

    def op_18(self, p):  # VAE
        q = torch.arange(p.Q)[:, None]
        j = torch.arange(p.d)[None, :]
        mu = self.rd(p.lin, q * p.lin_pitch + j)
        ls = self.rd(p.lin, q * p.lin_pitch + p.d + j)
        e = self.rd(p.eps, q * p.d + j)
        sigma = torch.exp(0.5 * ls)
        if not p.backward:
            s = torch.zeros(p.Q, p.d_pitch)
            s[:, :p.d] = mu + sigma * e
            self.wr(p.sample, torch.arange(p.Q * p.d_pitch), s.reshape(-1))
            s2 = sigma * sigma
            self.wr(p.kl_terms, torch.arange(p.Q), (1 + torch.log(s2) - mu * mu - s2).sum(1))
        else:
            klc = float(self.rd(p.kl_coef_dev, torch.arange(1))[0]) if p.kl_coef_dev else p.kl_coef
            klc *= self._gmul(p)
            if p.kl_value:
                klc = klc if float(self.rd(p.kl_value, torch.arange(1))[0]) >= p.free_nats else 0.0
            ds = self.rd(p.dsample, q * p.d_pitch + j)
            self.wr(p.dlin, q * p.lin_pitch + j, ds + klc * mu)
            self.wr(p.dlin, q * p.lin_pitch + p.d + j, ds * e * 0.5 * sigma + klc * (-0.5) * (1 - sigma * sigma))

    def op_19(self, p):  # AE_NORM
        q = torch.arange(p.Q)[:, None]
        j = torch.arange(p.d)[None, :]
        z = self.rd(p.ze, q * p.d_pitch + j)
        nrm = (z ** 2).sum(1).sqrt()
        if not p.backward:
            self.wr(p.term, torch.arange(p.Q), (nrm - 1).abs())
        else:
            g = self.rd(p.dze_in, q * p.d_pitch + j) + p.coef * self._gmul(p) * torch.sign(nrm - 1)[:, None] * z / nrm[:, None]
            out = torch.zeros(p.Q, p.d_pitch)
            out[:, :p.d] = g
            self.wr(p.dze, torch.arange(p.Q * p.d_pitch), out.reshape(-1))

    def op_20(self, p):  # JITTER (counter RNG restated in oracle/jitter_rng.py)
        from oracle import jitter_rng
        idx = jitter_rng.device_indices(p.seed, p.step, p.B, p.n, p.p, p.mode)
        for b in range(p.B):
            self.wr(p.out, torch.arange(p.n) + b * p.out_pitch, torch.from_numpy(idx[b]))

    def op_21(self, p):  # VQ_DIAG (vqema_bn.py:155-160, 251-264; util.py:98-105)
        o = torch.zeros(9)
        if p.ze:
            z = self.rd(p.ze, torch.arange(p.Q)[:, None] * p.d_pitch + torch.arange(p.d)[None, :])
            nr = (z ** 2).sum(1).sqrt()
            o[0], o[1] = nr.min(), nr.max()
        if p.emb:
            e = self.rd(p.emb, torch.arange(p.K * p.d)).view(p.K, p.d)
            nr = (e ** 2).sum(1).sqrt()
            o[2], o[3] = nr.min(), nr.max()
        if p.hist:
            h = self.rd(p.hist, torch.arange(p.K)).double()
            n = h / h.sum()
            o[4] = -(n * torch.where(n == 0, torch.zeros_like(n), torch.log2(n))).sum()
        if p.n_sum:
            o[5] = (self.rd(p.n_sum, torch.arange(p.K)) > 0).sum()
        if p.peak:
            pk = self.rd(p.peak, torch.arange(p.B * p.w)).view(p.B, p.w)[:, :p.w - 1].double()
            am = self.rd(p.amax, torch.arange(p.B * p.w)).view(p.B, p.w)[:, :p.w - 1]
            o[6], o[7], o[8] = pk.mean(), pk.std(), am.unique().numel()
        elif p.logits:
            idx = (torch.arange(p.B)[:, None, None] * p.bs + torch.arange(p.w - 1)[None, :, None] * p.pitch
                   + torch.arange(p.n_quant)[None, None, :])
            pk, am = torch.log_softmax(self.rd(p.logits, idx).double(), -1).max(-1)
            o[6], o[7], o[8] = pk.mean(), pk.std(), am.unique().numel()
        self.wr(p.out, torch.arange(9), o)

    def op_23(self, p):  # MOMENTS (gradient statistics of run(): autoencoder_model.py:252-257, mfcc_inverter.py:100-106)
        m = torch.arange(p.rows)
        x = torch.stack([self.vload(p.x, b, m, 0, p.cols) for b in range(p.batch)]).double()
        n = x.numel()
        o = torch.tensor([x.mean(), x.std() if n > 1 else 0.0, x.sum(), (x * x).sum()]) if n else torch.zeros(4)
        self.wr(p.out, torch.arange(4), o.float())


def emulate(eng):
    """Make a TrainEngine built on 'cpu' executable: every plan it would send to aew_run_plan goes through the
    interpreter instead (test infrastructure for the GPU-less container: data-parallel schedules, optimizer flows)."""
    emu = Emu(eng.ws)
    eng._stream = lambda: 0
    eng._run = lambda plan, timing=False: emu.run(plan)
    for name in ("opt", "cb"):
        pl = getattr(eng, name, None)
        if pl is not None:
            pl.run = (lambda p: (lambda stream=0: emu.run(p)))(pl)
    return eng
