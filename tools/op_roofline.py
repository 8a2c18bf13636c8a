"""Per-op roofline table: which launches of the step are MFMA-bound and which are HBM-bound.

  python tools/op_roofline.py [profiles/r01_per_op_ms.txt]

Builds the BASELINE configs[1] plan on the CPU (descriptors only), and for every GEMM op computes its FLOPs and the
bytes it has to move at least once (distinct A-segment buffers x rows x k_len, the weight matrix, every output and
aux view; fp32 slabs for the TN ops), then divides by the measured per-op time of the given file.  Segments that
re-read the same buffer at another row offset (the dilated taps) are counted once: they hit in L2 / MALL."""
import collections
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ae_wavenet_amd import _lib as L, config, model as M            # noqa: E402

ES = {L.BF16: 2, L.F32: 4}


def nt_cost(g):
    rows = g.M * g.batch
    seen, a_bytes = set(), 0
    for i in range(g.n_segs):
        s = g.seg[i]
        key = (s.ptr - s.row_off * 0, s.k_len)                       # same buffer at another row offset = one read
        base = s.ptr
        if base in seen:
            continue
        seen.add(base)
        a_bytes += rows * s.k_len * ES[g.dtype]
    w_bytes = g.N_pad * g.K_total * ES[g.dtype]
    io = 0
    for v in (g.out0, g.out1, g.out2, g.aux0, g.aux1):
        if v.ptr:
            width = g.N_pad // 2 if (g.epi == L.EPI_GATED) else g.N_pad
            if g.epi == L.EPI_RES_SKIP:
                width = g.n_split if v is g.out0 or v is g.aux0 else g.N_pad - g.n_split
            if g.epi == L.EPI_DFG and (v is g.aux0 or v is g.aux1):
                width = g.N_pad // 2
            io += rows * width * ES[v.dtype]
    flops = 2.0 * rows * g.N_pad * g.K_total
    return flops, a_bytes + w_bytes + io


def tn_cost(t):
    rows = t.Mc * t.batch
    a = sum(rows * t.seg[i].k_len * ES[t.dtype] for i in range(t.n_segs))
    gb = rows * t.N_pad * ES[t.dtype]
    slabs = L.load().aew_tn_slabs(L.C.byref(t)) if os.path.exists(L.LIB_PATH) else 1
    out = slabs * t.N_pad * t.K_total * 4
    return 2.0 * rows * t.N_pad * t.K_total, a + gb + out


def tn_group_cost(descs):
    """A grouped launch: FLOPs of every descriptor; least bytes two ways - every distinct operand buffer of the group
    read ONCE (the x of a layer is the A operand of its fg matrix at two row offsets and the dx of the next layer the G
    operand of the residual matrix: one read each if the group's tiles marched in step), and every segment of every
    descriptor read once (what a launch per matrix would have to move at least)."""
    fl = once = every = 0.0
    seen = set()
    for t in descs:
        rows = t.Mc * t.batch
        fl += 2.0 * rows * t.N_pad * t.K_total
        out = t.N_pad * t.K_total * 4 * max(1, t.grp_splits * t.batch if t.grp_splits > 0 else 1)
        once += out
        every += out
        for ptr, width in [(t.g.ptr, t.N_pad)] + [(t.seg[i].ptr, t.seg[i].k_len) for i in range(t.n_segs)]:
            b = rows * width * ES[t.dtype]
            every += b
            if ptr not in seen:
                seen.add(ptr)
                once += b
    return fl, once, every


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "profiles", "r01_per_op_ms.txt")
    ms = collections.OrderedDict()
    for line in open(path):
        m = re.match(r"\s*([\d.]+)\s+(.+)$", line)
        if m:
            ms[m.group(2).strip()] = ms.get(m.group(2).strip(), 0.0) + float(m.group(1))
    hps = config.make_hps("vqvae-ema", n_win_batch=5000, n_batch=8)
    eng = M.TrainEngine(hps, B=8, device="cpu", n_mel=39)
    groups = collections.OrderedDict()
    extra = {}
    for plan in (eng.fwd_a, eng.fwd_b, eng.bwd):
        for op, lab in zip(plan.ops, plan.labels):
            if op.kind == L.OP_GEMM_NT:
                fl, by = nt_cost(op.u.nt)
            elif op.kind == L.OP_GEMM_TN:
                fl, by = tn_cost(op.u.tn)
            elif op.kind == L.OP_GEMM_TN_GROUP and lab in getattr(plan, "tn_groups", {}):
                fl, by, every = tn_group_cost(plan.tn_groups[lab])
                extra[re.sub(r"\d+", "#", lab)] = every
            else:
                continue
            if lab not in ms:
                continue
            key = re.sub(r"\d+", "#", lab)
            g = groups.setdefault(key, [0.0, 0.0, 0.0, 0])
            g[0] += fl; g[1] += by; g[2] += ms[lab]; g[3] += 1
    print(f"{'op group':44s} {'n':>3s} {'ms':>7s} {'TFLOP/s':>8s} {'of 2500':>8s} {'TB/s':>6s} {'of 8':>6s}  bound")
    for k, (fl, by, t, n) in sorted(groups.items(), key=lambda kv: -kv[1][2]):
        if t <= 0:
            continue
        tf, tb = fl / t / 1e9, by / t / 1e9
        bound = "HBM" if tb / 8.0 > tf / 2500.0 else "MFMA"
        note = ""
        if k in extra:
            note = (f"   (grouped launch: {by / 1e9:.2f} GB with every operand buffer read once, {extra[k] / 1e9:.2f} GB with every "
                    f"segment of every matrix read once = {extra[k] / t / 1e9:.2f} TB/s)")
        print(f"{k[:44]:44s} {n:3d} {t:7.3f} {tf:8.0f} {tf / 2500:8.1%} {tb:6.2f} {tb / 8:6.1%}  {bound}{note}")


if __name__ == "__main__":
    main()
