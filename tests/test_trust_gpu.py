"""A step that can be trusted (SURVEY §7 "deterministic mode needed for parity runs"):

* a chained launch whose hand-off wait gave up must not reach the parameters, however far the host runs ahead of the
  device, and must be reported (csrc/aew_chain.hip sticky word -> aew_adam_t.guard / aew_vq_ema_t.guard;
  TrainEngine._chain_watch);
* the whole training step has ONE summation order (aew_tuning_t.deterministic): column sums (k_colsum), the
  speaker-embedding sums (k_spk_bwd), the jitter scatter (k_lc_scatter_det) and the codebook gradient of the plain VQ
  bottleneck (k_vq_bwd) carry no fp32 atomics, so two runs of the same steps agree BIT FOR BIT in every gradient,
  parameter and moment (the reference's own step is deterministic on CPU: chassis.py:151-171).
"""
import ctypes as C

import numpy as np
import pytest
import torch

from ae_wavenet_amd import _lib as L, config, model as M, plan as PLN
from tests.test_gpu_parity import DEV, np_weights, seeded_full_engine

pytestmark = pytest.mark.gpu


def _break_first_chain(eng):
    """Stage 0 of the forward's first chained launch stops publishing (edited in the device copy of its stage table):
    every consumer of it spins to the limit and gives up."""
    lab = next(iter(eng.fwd_b.nt_chains))
    stages, _ = eng.fwd_b.nt_chains[lab]
    n = len(stages)
    raw = eng.ws.get("chain.fwd.0.stages")
    nbytes = C.sizeof(L.NtStage) * n
    host = (L.NtStage * n).from_buffer_copy(bytes(raw[:(nbytes + 7) // 8].cpu().numpy().tobytes())[:nbytes])
    assert host[0].publish == 1
    host[0].publish = 0
    b = bytes(host)
    raw[:(len(b) + 7) // 8].copy_(torch.frombuffer(bytearray(b + b"\0" * (-len(b) % 8)), dtype=torch.int64))
    return lab


def test_timeout_with_the_host_running_ahead_raises_and_leaves_the_parameters(monkeypatch):
    """The forward's chain gives up in EVERY step from step 1 on while the host queues steps without ever synchronising
    (the bench loop, any loop without a per-step .item()).  Round 5's check looked at the pinned flags only if the copy
    event had already completed and let the next forward overwrite them.  Now: the device-side guard keeps Adam, the EMA
    accumulation and the codebook refresh from applying anything from the poisoned step on, and the host raises within
    CHAIN_WATCH_SLOTS plans."""
    monkeypatch.setattr(M.TrainEngine, "nt_chain", 64)
    monkeypatch.setattr(M.TrainEngine, "nt_chain_force", True)
    # (forced chains at this size: most stages are resident at once and consumers really wait, ~1e3 polls; a broken stage
    # costs its first waiter 20 000 polls = ~20 ms, the others give up with it)
    monkeypatch.setattr(M.TrainEngine, "nt_chain_spin_max", 20000)
    monkeypatch.delenv("AEW_NT_CHAIN", raising=False)
    lib = L.load()
    lib.aew_set_nt_window(0)                                  # (forced small chains run the plain bodies)
    try:
        hps = config.make_hps("vqvae-ema", n_win_batch=100)
        eng = M.TrainEngine(hps, B=2, device=DEV, n_mel=39, update_codebook_every_step=True)
        assert eng.fwd_b.nt_chains, "the engine was expected to chain its forward stack"
        for k, v in np_weights({k: eng.ps.shape[k] for k in eng.ps.names()}, 3).items():
            eng.ps.view(k).copy_(torch.from_numpy(v))
        rs = np.random.RandomState(4)
        eng.emb.copy_(torch.from_numpy((rs.standard_normal((hps.bn_vq_n_embed, hps.bn_n_out)) * 0.7).astype(np.float32)))
        eng.init_ema_from_emb()
        g = eng.geom
        eng.set_inputs(torch.from_numpy(rs.randint(0, 256, (2, g.enc_in_len)).astype(np.float32)).to(DEV),
                       torch.from_numpy(rs.standard_normal((2, 39, g.mel_len)).astype(np.float32)).to(DEV),
                       torch.from_numpy(rs.randint(0, 40, (2,)).astype(np.int64)).to(DEV),
                       torch.arange(g.embed_len).repeat(2, 1).to(DEV))
        # step 0: healthy
        eng.forward(); eng.backward(); eng.adam_step(1e-3)
        torch.cuda.synchronize()
        assert int(eng.chain_guard[0]) == 0
        good = {"p": eng.ps.params[:eng.ps.numel].clone(), "m": eng.adam_m[:eng.ps.numel].clone(), "emb": eng.emb.clone(),
                "numer": eng.ema_numer.clone(), "denom": eng.ema_denom.clone()}
        _break_first_chain(eng)
        # a long kernel queue in front, so that the host is certainly several steps ahead of the device
        a = torch.randn(6144, 6144, device=DEV)
        for _ in range(40):
            a = (a @ a).clamp_(-1, 1)
        raised_at = None
        for step in range(1, 12):
            try:
                eng.forward(); eng.backward(); eng.adam_step(1e-3)
            except L.AewError as e:
                assert "gave up" in str(e)
                raised_at = step
                break
        assert raised_at is not None, "eleven poisoned steps queued and nothing raised"
        # the ring holds CHAIN_WATCH_SLOTS copies, one per forward: the report comes at most that many steps late (+ 1: the
        # forward whose check finds the ring full)
        assert raised_at <= eng.CHAIN_WATCH_SLOTS + 2, raised_at
        torch.cuda.synchronize()
        assert int(eng.chain_guard[0]) != 0
        assert torch.equal(eng.ps.params[:eng.ps.numel], good["p"]), "a poisoned step reached the parameters"
        assert torch.equal(eng.adam_m[:eng.ps.numel], good["m"])
        # the codebook too.  (The EMA ACCUMULATION of the first poisoned step runs at the head of the second forward plan -
        # from the encoder's outputs, before the chain - and is applied; every later one and every codebook refresh is not.)
        assert torch.equal(eng.emb, good["emb"])
        with pytest.raises(L.AewError, match="gave up"):
            eng.chain_guard_check()                            # the synchronous form
        with pytest.raises(L.AewError, match="gave up"):
            eng.forward()                                      # still set: every further forward reports it
        eng.clear_chain_guard()
        eng._chain_watch("check")                              # re-armed: silent
    finally:
        lib.aew_set_nt_window(64)


def _run_steps(arch, B, w, steps, seed, jitter_on, det=1):
    t = L.current_tuning(deterministic=det)
    hps = config.make_hps(arch, n_win_batch=w)
    eng = M.TrainEngine(hps, B=B, device=DEV, n_mel=39, tuning=t)
    for k, v in np_weights({k: eng.ps.shape[k] for k in eng.ps.names()}, seed).items():
        eng.ps.view(k).copy_(torch.from_numpy(v))
    rs = np.random.RandomState(seed + 1)
    if arch in ("vqvae-ema", "vqvae"):
        eng.emb.copy_(torch.from_numpy((rs.standard_normal(tuple(eng.emb.shape)) * 0.7).astype(np.float32)))
        if arch == "vqvae-ema":
            eng.init_ema_from_emb()
    g = eng.geom
    out = []
    for i in range(steps):
        jit = torch.arange(g.embed_len).repeat(B, 1)
        if jitter_on:
            jit = (jit + torch.from_numpy(rs.randint(-1, 2, (B, g.embed_len)))).clamp_(0, g.embed_len - 1)
        eps = torch.from_numpy(rs.standard_normal((B, g.embed_len, hps.bn_n_out)).astype(np.float32)).to(DEV) if arch == "vae" else None
        eng.set_inputs(torch.from_numpy(rs.randint(0, 256, (B, g.enc_in_len)).astype(np.float32)).to(DEV),
                       torch.from_numpy(rs.standard_normal((B, 39, g.mel_len)).astype(np.float32)).to(DEV),
                       torch.from_numpy(rs.randint(0, 40, (B,)).astype(np.int64)).to(DEV), jit.to(DEV), eps=eps)
        if arch == "vae":
            eng.set_anneal_weight(0.3 + 0.1 * i)
        loss = eng.forward()
        eng.backward()
        torch.cuda.synchronize()
        out.append((float(loss), eng.ps.grads[:eng.ps.numel].clone()))
        eng.adam_step(1e-3)
    torch.cuda.synchronize()
    state = (eng.ps.params[:eng.ps.numel].clone(), eng.adam_m[:eng.ps.numel].clone(), eng.adam_v[:eng.ps.numel].clone())
    names = [(n, (eng.ps.view(n).data_ptr() - eng.ps.params.data_ptr()) // 4, eng.ps.numel_of(n)) for n in eng.ps.names()]
    del eng
    torch.cuda.empty_cache()
    return out, state, names


@pytest.mark.parametrize("arch,B,w,jitter_on", [("vqvae-ema", 8, 5000, True), ("vae", 8, 1000, True), ("vqvae", 3, 300, True),
                                                 ("ae", 20, 200, False)])
def test_training_steps_are_bit_reproducible(arch, B, w, jitter_on):
    """Three optimizer steps, twice, from the same state and the same batches: every gradient element of every step,
    every parameter and both Adam moments agree bit for bit - the bias-type gradients included (until round 5 they were
    fp32 atomics and had to be masked: tests/test_chain_gpu._mask).  Covers BASELINE configs[1] (B = 8, w = 5000,
    chained forward), the VAE with jitter on (configs[3]), the plain VQ bottleneck's codebook gradient and a batch
    beyond 16 (the speaker-gradient kernel's chunked form)."""
    a, sa, names = _run_steps(arch, B, w, 3, 21, jitter_on)
    b, sb, _ = _run_steps(arch, B, w, 3, 21, jitter_on)
    for i, ((la, ga), (lb, gb)) in enumerate(zip(a, b)):
        assert la == lb, (i, la, lb)
        if not torch.equal(ga, gb):
            d = (ga - gb).abs()
            bad = [(n, float(d[o:o + k].max())) for n, o, k in names if float(d[o:o + k].max()) > 0]
            raise AssertionError(f"step {i}: gradients differ between two runs: {bad[:8]}")
    for x, y, what in zip(sa, sb, ("parameters", "m", "v")):
        assert torch.equal(x, y), what


def test_deterministic_sums_equal_the_atomic_ones_to_roundoff():
    """aew_tuning_t.deterministic = 0 keeps round 5's fp32 atomics (A/B of the cost): the same sums in another order."""
    a, _, names = _run_steps("vqvae-ema", 4, 600, 1, 8, True, det=1)
    b, _, _ = _run_steps("vqvae-ema", 4, 600, 1, 8, True, det=0)
    ga, gb = a[0][1], b[0][1]
    assert a[0][0] == b[0][0]
    for n, o, k in names:
        x, y = ga[o:o + k], gb[o:o + k]
        if n.endswith(".bias") or "speaker_embedding" in n:
            assert float((x - y).abs().max()) <= 1e-5 * float(y.abs().max()) + 1e-12, n
        else:
            assert torch.equal(x, y), n


@pytest.mark.parametrize("dtype,N,M,B,per_batch", [("bf16", 128, 7046, 8, False), ("bf16", 600, 900, 3, True), ("f32", 64, 29, 8, False),
                                                     ("f32", 300, 2500, 2, True)])
def test_colsum_deterministic_form(dtype, N, M, B, per_batch):
    """AEW_OP_COLSUM in its ticketed form (aew_colsum_t.det_scratch / det_tickets) over the shapes the plans use and some they do
    not - narrow and multi-column-block matrices, shared and per-batch outputs, bf16 and fp32, a row range that masks rows:
    equal to the fp64 column sums to fp32 round-off, bit-equal between repeats (the tickets re-arm themselves), bit-equal
    between two launches on different streams of work (order independence), and accumulating into what `out` held."""
    from ae_wavenet_amd.engine import det_colsum
    from ae_wavenet_amd.plan import Mat, Plan, Workspace
    ws = Workspace(DEV)
    dt = L.BF16 if dtype == "bf16" else L.F32
    pitch = (N + 127) // 128 * 128
    X = Mat.new(ws, "x", B, M, pitch, dt)
    gen = torch.Generator().manual_seed(N + M)
    X.tensor().copy_(torch.randn(B, M, pitch, generator=gen).to(ws.get("x").dtype))      # (the buffer is padded past B M pitch)
    out = ws.alloc("out", B * N if per_batch else N, torch.float32)
    cs = L.Colsum()
    cs.x = X.seg(128, row_off=3, hi=M - 5)                       # rows 3 .. M - 6 of every batch element
    Mv = M - 8
    cs.dtype, cs.M, cs.N, cs.batch = dt, Mv, N, B
    cs.out, cs.out_bs, cs.accumulate = out.data_ptr(), (N if per_batch else 0), 1
    det_colsum(ws, cs, "det.t")
    p = Plan("cs")
    p.add(L.OP_COLSUM, cs, "colsum")
    x = X.tensor()[:, 3:3 + Mv, :N].double()
    ref = x.sum(1) if per_batch else x.sum((0, 1))
    res = []
    for rep in range(3):
        out.fill_(0.5)                                           # accumulate = 1: the sums are ADDED to what is there
        if rep == 2:                                             # other work in flight beside it
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                a = torch.randn(2048, 2048, device=DEV)
                for _ in range(10):
                    a = (a @ a).clamp_(-1, 1)
        p.run(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        res.append(out[:ref.numel()].clone())
    got = res[0].double().reshape(ref.shape) - 0.5
    scale = float(x.abs().sum(1).max()) if per_batch else float(x.abs().sum((0, 1)).max())
    assert float((got - ref).abs().max()) <= 2e-6 * scale, float((got - ref).abs().max()) / scale
    assert torch.equal(res[0], res[1]) and torch.equal(res[0], res[2])
    assert int(ws.get("det.t.tickets").abs().max()) == 0        # every ticket word back at zero
    # ... and the atomic form gives the same sums to round-off
    out.fill_(0.5)
    p.run(torch.cuda.current_stream().cuda_stream, tuning=L.current_tuning(deterministic=0))
    torch.cuda.synchronize()
    assert float((out[:ref.numel()].double().reshape(ref.shape) - 0.5 - ref).abs().max()) <= 2e-6 * scale
