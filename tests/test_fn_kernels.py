"""Full-N NT kernels with loader / consumer waves and the fused gated layer (csrc/aew_fn.hip, impl = 2).

GPU: every case runs as impl 2 and as impl 0 (the tiled kernel, itself held to the scalar check kernel and the CPU
interpreter by test_gpu_parity.py::test_gemm_nt); outputs must agree BIT FOR BIT (same MFMA, K ascending, same
epilogue arithmetic).  A fused descriptor (W2 != NULL: gated GEMM -> z tile in LDS -> residual 1x1 + add,
wavenet.py:100-109) executes under impl 0 as the two-op sequence, which is the definition of its result.
CPU: the interpreter's fused op equals its two-op sequence.
"""
import pytest
import torch

from ae_wavenet_amd import _lib as L, plan as PL
from ae_wavenet_amd.plan import Mat, Plan, Workspace, make_nt
from tests.plan_emulator import Emu

DEV = "cuda:0"
BF, F3 = L.BF16, L.F32


def _fill(ws, name, gen, scale=1.0):
    t = ws.get(name)
    t.copy_(((torch.rand(t.shape, generator=gen) * 2 - 1) * scale).to(t.dtype))


def _mirror(ws_cpu, dev):
    ws = Workspace(dev)
    for n, t in ws_cpu.bufs.items():
        ws.bufs[n] = t.to(dev)
    return ws


class Case:
    """name, batch, M, (N, N_pad), K segments, epilogue kind"""

    def __init__(self, name, B, M, N, N_pad, ks, kind, N2=0, N2_pad=0, d=5):
        self.name, self.B, self.M, self.N, self.N_pad, self.ks, self.kind = name, B, M, N, N_pad, ks, kind
        self.N2, self.N2_pad, self.d = N2, N2_pad, d
        self.rows = M + 64

    def alloc(self, ws):
        B, R = self.B, self.rows
        kmax = max(self.ks)
        ws.alloc("A", B * R * kmax, torch.bfloat16)
        ws.alloc("A2", B * R * kmax, torch.bfloat16)
        ws.alloc("W", self.N_pad * sum(self.ks), torch.bfloat16)
        for n in ("O0", "O1", "O2"):
            ws.alloc(n, B * R * 1024, torch.bfloat16)
        ws.alloc("X0", B * R * 512, torch.bfloat16)
        ws.alloc("X1", B * R * 512, torch.bfloat16)
        ws.alloc("bias", B * 512, torch.float32)
        if self.N2_pad:
            ws.alloc("W2", self.N2_pad * (self.N_pad // 2), torch.bfloat16)
            ws.alloc("O3", B * R * 512, torch.bfloat16)

    def fill(self, ws, gen):
        for n in ("A", "A2", "X0", "X1", "bias"):
            _fill(ws, n, gen)
        _fill(ws, "W", gen, 0.06)
        # padded output channels of a packed weight matrix are zero (the engine's pack tables guarantee it)
        if self.N2_pad:
            _fill(ws, "W2", gen, 0.08)
            K2 = self.N_pad // 2
            w2 = ws.get("W2")[:self.N2_pad * K2].view(self.N2_pad, K2)
            w2[:, self.N:] = 0
            w2[self.N2:, :] = 0

    def op(self, ws, impl):
        B, M, R = self.B, self.M, self.rows
        kmax = max(self.ks)
        A = Mat(ws, "A", B, R, kmax, BF)
        A2 = Mat(ws, "A2", B, R, kmax, BF)
        segs = []
        for i, k in enumerate(self.ks):
            src = A if i % 2 == 0 else A2
            off = [0, self.d, -3, 7][i % 4]
            segs.append(src.seg(k, row_off=off, hi=R - 9 if i == 1 else None))
        Wm = Mat(ws, "W", 1, self.N_pad, sum(self.ks), BF)
        O0 = Mat(ws, "O0", B, R, 1024, BF)
        O1 = Mat(ws, "O1", B, R, 1024, BF)
        O2 = Mat(ws, "O2", B, R, 1024, BF)
        X0 = Mat(ws, "X0", B, R, 512, BF)
        X1 = Mat(ws, "X1", B, R, 512, BF)
        bias = ws.get("bias")
        kw = dict(impl=impl)
        kind = self.kind
        if kind == "bias_relu":
            return make_nt(BF, M, self.N, self.N_pad, B, segs, Wm.ptr, flags=L.EF_BIAS | L.EF_RELU, out0=O0.view(row_off=2),
                           bias_ptr=bias.data_ptr(), bias_bs=512, **kw)
        if kind == "add_relu_post":
            return make_nt(BF, M, self.N, self.N_pad, B, segs, Wm.ptr, flags=L.EF_ADD_AUX0 | L.EF_RELU_POST, out0=O0.view(),
                           aux0=X0.view(row_off=-2), **kw)
        if kind == "add":
            return make_nt(BF, M, self.N, self.N_pad, B, segs, Wm.ptr, flags=L.EF_ADD_AUX0, out0=O0.view(hi=M - 3),
                           aux0=X0.view(row_off=-self.d, hi=M - 20), **kw)
        if kind == "mask":
            return make_nt(BF, M, self.N, self.N_pad, B, segs, Wm.ptr, flags=L.EF_MUL_POS1, out0=O0.view(), aux1=X1.view(), **kw)
        if kind == "plain":
            return make_nt(BF, M, self.N, self.N_pad, B, segs, Wm.ptr, out0=O0.view(), **kw)
        if kind == "dfg":
            return make_nt(BF, M, self.N, self.N_pad, B, segs, Wm.ptr, epi=L.EPI_DFG, out0=O0.view(), aux0=X0.view(),
                           aux1=X1.view(), **kw)
        if kind in ("gated", "fused"):
            extra = {}
            if kind == "fused":
                W2 = Mat(ws, "W2", 1, self.N2_pad, self.N_pad // 2, BF)
                O3 = Mat(ws, "O3", B, R, 512, BF)
                extra = dict(W2_ptr=W2.ptr, N2=self.N2, N2_pad=self.N2_pad, out3=O3.view(), aux0=X0.view(row_off=self.d))
            return make_nt(BF, M, self.N, self.N_pad, B, segs, Wm.ptr, epi=L.EPI_GATED, out0=O0.view(), out1=O1.view(),
                           out2=O2.view(), bias_ptr=bias.data_ptr(), bias_bs=512, **kw, **extra)
        raise ValueError(kind)

    def outputs(self):
        return ("O0", "O1", "O2") + (("O3",) if self.N2_pad else ())


CASES = [
    Case("store128_bias_relu", 2, 300, 120, 128, [128, 64], "bias_relu"),
    Case("store256_add_relu_post_4seg", 3, 1000, 256, 256, [128, 128, 64, 192], "add_relu_post"),
    Case("store384_add_taps", 2, 700, 368, 384, [512, 512], "add", d=16),
    Case("store512_mask", 2, 333, 512, 512, [128], "mask"),
    Case("store384_plain_tiny", 1, 40, 384, 384, [64], "plain"),
    Case("dfg256", 2, 450, 256, 256, [384, 256], "dfg"),
    Case("dfg128", 2, 450, 128, 128, [128, 128], "dfg"),
    Case("gated512", 2, 500, 256, 512, [384, 384, 128], "gated", d=64),
    Case("gated256", 3, 260, 128, 256, [128, 128, 128], "gated", d=2),
    Case("fused_4_3", 2, 500, 256, 512, [384, 384, 128], "fused", N2=368, N2_pad=384, d=8),
    Case("fused_2_1", 3, 260, 32, 256, [128, 128, 128], "fused", N2=64, N2_pad=128, d=4),
    Case("fused_2_2", 2, 700, 128, 256, [128, 128, 128], "fused", N2=200, N2_pad=256, d=1),
    # full-size shapes of the bench workload: chunks of 14 units = tiles of 5 + 5 + 4 row groups, batch 8
    Case("fused_4_3_fullsize", 8, 7030, 256, 512, [384, 384, 128], "fused", N2=368, N2_pad=384, d=16),
    Case("store384_fullsize", 8, 7046, 368, 384, [512, 512], "add", d=512),
    # >= 16 K tiles, N <= 256: the three-stage operand ring (loaders two K tiles ahead, counted vmcnt).  One tile per
    # block, two tiles per block (the K-tile stream crosses a tile boundary), five segments with row offsets
    Case("store256_ring3_fullsize", 8, 7046, 256, 256, [256] * 5, "add", d=3),
    Case("store128_ring3_multiseg", 2, 3000, 128, 128, [512, 512, 256], "bias_relu"),
    Case("store128_ring3_two_tiles", 1, 70000, 104, 128, [512, 512], "plain"),
    Case("store256_ring3_mask", 3, 1000, 200, 256, [1024], "mask"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_fn_matches_tiled_kernel_bit_for_bit(case):
    gen = torch.Generator().manual_seed(3)
    ws_c = Workspace("cpu")
    case.alloc(ws_c)
    case.fill(ws_c, gen)
    res = {}
    for impl in (0, 2):
        ws_g = _mirror(ws_c, DEV)
        p = Plan("fn")
        p.add(L.OP_GEMM_NT, case.op(ws_g, impl), "nt")
        p.run(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        res[impl] = {n: ws_g.get(n).float().cpu() for n in case.outputs()}
    for n in case.outputs():
        a, b = res[0][n], res[2][n]
        assert float(a.abs().max()) > 0 or n in ("O1", "O2"), n
        bad = (a != b).nonzero()
        assert bad.numel() == 0, (n, "first mismatch at flat index", int(bad[0]), float(a.reshape(-1)[bad[0]]),
                                  float(b.reshape(-1)[bad[0]]), "mismatches", int(bad.shape[0]))


@pytest.mark.gpu
def test_fn_switch_off_falls_back():
    """aew_set_fn(0): impl 2 descriptors run on the tiled kernels (same results)."""
    lib = L.load()
    case = CASES[9]
    gen = torch.Generator().manual_seed(4)
    ws_c = Workspace("cpu")
    case.alloc(ws_c)
    case.fill(ws_c, gen)
    res = []
    try:
        for on in (1, 0):
            lib.aew_set_fn(on)
            ws_g = _mirror(ws_c, DEV)
            p = Plan("fn")
            p.add(L.OP_GEMM_NT, case.op(ws_g, 2), "nt")
            p.run(torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            res.append({n: ws_g.get(n).float().cpu() for n in case.outputs()})
    finally:
        lib.aew_set_fn(1)
    for n in case.outputs():
        assert torch.equal(res[0][n], res[1][n]), n


@pytest.mark.parametrize("case", [c for c in CASES if c.kind == "fused" and c.M <= 700], ids=lambda c: c.name)
def test_interpreter_fused_equals_two_ops(case):
    """CPU: the plan interpreter's fused op == gated op followed by the residual GEMM over the stored z."""
    gen = torch.Generator().manual_seed(5)
    ws_a = Workspace("cpu")
    case.alloc(ws_a)
    case.fill(ws_a, gen)
    ws_b = Workspace("cpu")
    for n, t in ws_a.bufs.items():
        ws_b.bufs[n] = t.clone()
    p = Plan("fused")
    p.add(L.OP_GEMM_NT, case.op(ws_a, 0), "fused")
    Emu(ws_a).run(p)
    g = case.op(ws_b, 0)
    W2_ptr, N2, N2_pad = g.W2, g.N2, g.N2_pad
    g.W2 = None
    O0 = Mat(ws_b, "O0", case.B, case.rows, 1024, BF)
    X0 = Mat(ws_b, "X0", case.B, case.rows, 512, BF)
    O3 = Mat(ws_b, "O3", case.B, case.rows, 512, BF)
    g2 = make_nt(BF, case.M, N2, N2_pad, case.B, [O0.seg(case.N_pad // 2)], W2_ptr, flags=L.EF_ADD_AUX0, out0=O3.view(),
                 aux0=X0.view(row_off=case.d))
    q = Plan("two")
    q.add(L.OP_GEMM_NT, g, "gated")
    q.add(L.OP_GEMM_NT, g2, "res")
    Emu(ws_b).run(q)
    for n in case.outputs():
        assert torch.equal(ws_a.get(n).float(), ws_b.get(n).float()), n


@pytest.mark.gpu
@pytest.mark.parametrize("bn", ["vqvae-ema"])
def test_engine_with_all_fn_ops_matches_tiled_engine(bn, monkeypatch):
    """A whole training step with every covered op on the full-N kernels (fused gated layers included) against the
    same step on the tiled kernels: forward bit-identical, gradients equal up to the atomically accumulated bias sums."""
    from ae_wavenet_amd import config, engine as E, model as M
    hps = config.make_hps(bn, n_res=192, n_dil=128, n_skp=128, n_post=128, n_lc_out=128, enc_n_out=64, bn_n_out=16,
                          bn_vq_n_embed=64, n_win_batch=700, n_blocks=2, n_block_layers=4, n_global_embed=4, n_speakers=5)
    res = {}
    for tag, ops in (("tiled", frozenset()), ("fn", frozenset(("layer", "G1", "dx", "dz", "skip", "dcond", "post")))):
        monkeypatch.setattr(E.DecoderPlan, "fn_ops", ops)
        eng = M.TrainEngine(hps, B=3, device=DEV, n_mel=39, update_codebook_every_step=False)
        labels = eng.fwd_b.labels
        assert ("G2.0" in labels) == (tag == "tiled")             # fused: no separate residual GEMMs
        gen = torch.Generator().manual_seed(9)
        for k in eng.ps.names():
            shp = eng.ps.shape[k]
            t = torch.empty(shp)
            torch.nn.init.xavier_uniform_(t, generator=gen) if len(shp) >= 2 else t.uniform_(-0.1, 0.1, generator=gen)
            eng.ps.view(k).copy_(t)
        eng.emb.copy_(torch.randn(64, 16, generator=gen))
        eng.init_ema_from_emb()
        g = eng.geom
        wav = torch.randint(0, 256, (3, g.enc_in_len), generator=gen).float()
        mel = torch.randn(3, 39, g.mel_len, generator=gen)
        voice = torch.randint(0, 5, (3,), generator=gen)
        jitter = torch.arange(g.embed_len).repeat(3, 1)
        eng.set_inputs(wav.to(DEV), mel.to(DEV), voice.to(DEV), jitter.to(DEV))
        loss = float(eng.forward())
        eng.backward()
        torch.cuda.synchronize()
        res[tag] = (loss, eng.logits().clone(), eng.ps.grads[:eng.ps.numel].clone())
    assert res["tiled"][0] == res["fn"][0]
    assert torch.equal(res["tiled"][1], res["fn"][1])
    ga, gb = res["tiled"][2], res["fn"][2]
    assert torch.allclose(ga, gb, rtol=1e-4, atol=1e-6 * float(ga.abs().max()))
