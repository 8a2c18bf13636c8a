#!/usr/bin/env python3
"""Interleaved A/B of the engine-level training step under different kernel settings, ONE process, one box
(box-to-box spread is +-0.15 ms, so only interleaved runs on one box are comparable: profiles/r02_notes.md).

    python tools/ab_step.py base "nt_window=0" "nt_window=16" [--rounds 4 --steps 20 --per-op nt_window=0]

A configuration is a comma list of `setter=value` (aew_set_<setter>(value)) and `E.attr=value` (a DecoderPlan class
attribute that shapes the plan, e.g. E.wgrad_group=0: one engine is built per distinct set of them); `base` = the
defaults.  Setters that
change kernel choice at launch time only (nt_window, nt_rows192, nt_small_tiles, ...) are safe to flip on a built
engine: the captured graphs are dropped and re-captured.  Prints ms/step per configuration and round, and with
--per-op the per-op HIP-event table (serial plan order) of the named configurations.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DEFAULTS = {"nt_window": 64, "nt_small_deep": 256, "nt_rows192": 1, "nt_small_tiles": 128, "lanes": 0, "fn": 1, "fn_ring3": 16, "graphs": 1, "nt_small_n64": 256, "nt_mem128": 0, "nt_deep": 0, "tn_cursor": 0, "deterministic": 1, "tn_mfma32": 0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="+")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--n-win", dest="n_win", type=int, default=5000)
    ap.add_argument("--arch", default="vqvae-ema", help="config.make_hps architecture (deep: --batch 4 --n-win 65536)")
    ap.add_argument("--per-op", dest="per_op", action="append", default=[])
    ap.add_argument("--out", default=None, help="directory for the per-op tables")
    ap.add_argument("--plan", default=None, help="time ONE plan of the engine (fwd_a | fwd_b | bwd | opt) instead of the step")
    args = ap.parse_args()
    import torch
    from ae_wavenet_amd import _lib as L, autoencoder_model as ae, config, engine as E
    lib = L.load()
    dev = torch.device("cuda", 0)
    hps = config.make_hps(args.arch, n_win_batch=args.n_win, n_batch=args.batch, jitter_prob=0.12)
    e_defaults = {}

    def parse(cfg):
        vals, evals = dict(DEFAULTS), {}
        if cfg != "base":
            for kv in cfg.split(","):
                k, v = kv.split("=")
                if k.startswith("E."):
                    evals[k[2:]] = int(v)
                else:
                    vals[k] = int(v)
        return vals, tuple(sorted(evals.items()))

    engines = {}

    def owner(k):                                  # E.tiled: the copy-table builder's switch; everything else DecoderPlan's
        from ae_wavenet_amd import plan as PLN
        if k.startswith("LANE_") or k in ("merge_packs", "graph_lanes", "pack_dec_late", "small_split", "nt_chain", "nt_chain_bwd", "nt_chain_bwd_phase", "nt_chain_force", "nt_chain_flags"):
            from ae_wavenet_amd import model as MDL
            return MDL.TrainEngine
        if k == "k_split":                         # (timing only: the oracle's canonical order is EncoderPlan.k_split's default)
            return E.EncoderPlan
        return PLN.CopyTableBuilder if k in ("tiled", "interleave", "tile_cap") else E.DecoderPlan

    def engine_for(ekey):
        if ekey not in engines:
            for k, v in ekey:
                e_defaults.setdefault(k, getattr(owner(k), k))
                setattr(owner(k), k, v)
            torch.manual_seed(2507)
            model = ae.AutoEncoder(hps, n_mel=39).to(dev)
            eng = model._ensure_engine(args.batch)
            for k, v in e_defaults.items():
                setattr(owner(k), k, v)
            g = eng.geom
            gen = torch.Generator().manual_seed(0)
            wav = torch.randint(0, 256, (args.batch, g.enc_in_len), generator=gen).float().to(dev)
            mel = torch.randn(args.batch, 39, g.mel_len, generator=gen).to(dev)
            voice = torch.randint(0, 40, (args.batch,), generator=gen).to(dev)
            jitter = torch.arange(g.embed_len).repeat(args.batch, 1).to(dev)
            eng.set_inputs(wav, mel, voice, jitter)
            engines[ekey] = (model, eng)
        return engines[ekey][1]

    cur = [None]

    def apply(cfg):
        vals, ekey = parse(cfg)
        eng = engine_for(ekey)
        cur[0] = eng
        for k, v in vals.items():
            if k == "graphs":                      # 0: plans run eagerly (real streams + events) instead of as hipGraphs
                eng.use_graphs = bool(v)
                continue
            if k in ("deterministic", "tn_mfma32"):   # aew_tuning_t fields without a setter: through the record
                import ctypes as _C
                t = L.current_tuning(**{k: v})
                L.check(lib.aew_tuning_set(_C.byref(t)), "aew_tuning_set")
                continue
            if k == "tn_cursor":                   # epoch * 10 + slack (0 = the defaults 4 / 2 where the plan paces a launch)
                lib.aew_set_tn_cursor(v // 10, v % 10)
                continue
            getattr(lib, "aew_set_" + k)(v)
        for p in (eng.fwd_a, eng.fwd_b, eng.bwd, getattr(eng, "bwd_a", None), getattr(eng, "bwd_b", None),
                  getattr(eng, "cb", None), getattr(eng, "fwd_b_noema", None), getattr(eng, "ema_plan", None)):
            if p is not None:
                p.invalidate_graph()

    def step():
        eng = cur[0]
        if args.plan:
            eng._run(getattr(eng, args.plan), False)
            return
        eng.forward()
        eng.backward()
        eng.adam_step(1e-4, 1.0)

    def timed(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    res = {c: [] for c in args.configs}
    for r in range(args.rounds):
        for c in args.configs:
            apply(c)
            for _ in range(3):
                step()
            res[c].append(timed(args.steps))
    for c in args.configs:
        v = res[c]
        print(f"{c:40s} " + " ".join(f"{x:7.3f}" for x in v) + f"   min {min(v):.3f} ms/step", flush=True)
    import ctypes as C
    for c in args.per_op:
        apply(c)
        eng = cur[0]
        lib.aew_set_lanes(0)
        for _ in range(2):
            step()
        lib.aew_timing_enable(1)
        n_t = 3
        for _ in range(n_t):
            eng.forward(None, timing=True)
            eng.backward(timing=True)
            eng.adam_step(1e-4, 1.0)
        cap = 1 << 16
        ms = (C.c_float * cap)()
        tags = (C.c_int32 * cap)()
        cnt = C.c_int(0)
        L.check(lib.aew_timing_read(ms, tags, cap, C.byref(cnt)), "timing_read")
        lib.aew_timing_enable(0)
        labels = eng.fwd_a.labels + eng.fwd_b.labels + eng.bwd.labels + (eng.cb.labels if hasattr(eng, "cb") else []) + eng.opt.labels
        per, cls = {}, {}
        for i in range(min(cnt.value, cap)):
            lab = labels[i % len(labels)]
            per[lab] = per.get(lab, 0.0) + ms[i] / n_t
            k = tags[i] // 100
            cls[k] = cls.get(k, 0.0) + ms[i] / n_t
        grp = {}
        import re
        for lab, v in per.items():
            key = re.sub(r"\d+", "#", lab)
            grp[key] = grp.get(key, 0.0) + v
        print(f"--- per-op (serial order), config {c}: classes " + " ".join(f"{k}:{v:.3f}" for k, v in sorted(cls.items()))
              + f" sum {sum(cls.values()):.3f} ms")
        for key, v in sorted(grp.items(), key=lambda kv: -kv[1])[:40]:
            print(f"   {v:8.4f}  {key}")
        if args.out:
            os.makedirs(args.out, exist_ok=True)
            with open(os.path.join(args.out, "per_op_" + c.replace("=", "").replace(",", "_") + ".txt"), "w") as fh:
                for lab, v in sorted(per.items(), key=lambda kv: -kv[1]):
                    fh.write(f"{v:9.4f}  {lab}\n")
        apply("base")


if __name__ == "__main__":
    main()
