#!/usr/bin/env python3
"""RCCL under the product's collectives on ONE GPU (train.py:58-60, chassis.py:168-169,188-190).

A process group of one rank over backend "nccl" (= RCCL on ROCm) and a DataParallel(force_collectives=True) that does not
take its world == 1 short cuts: every collective of the data-parallel step is then really issued to RCCL with the product's
own views, dtypes and stream ordering - reduce_scatter_tensor (fp32 and through the bf16 transport copy), sharded Adam,
the deferred head / decoder all_gather_into_tensor pair, the async all_reduce of the EMA statistics under the decoder
forward + backward, all_reduce of the scalar KL, broadcast of parameters and codebook, the all-gather of the Adam moments
- against plans replayed as captured hipGraphs on torch's current stream.  With one rank every collective is an
identity, so the steps must equal the plain single-process steps BIT FOR BIT (fp32 transport): every parameter, both Adam
moments, codebook and EMA numerator after three steps.  (Until round 5 the bias-type gradients - column sums and the
speaker-embedding sums - were fp32 atomics and had to be masked out; since ABI 20 they have one summation order,
aew_tuning_t.deterministic, and "plain_again" - the plain run against itself - is the yardstick that shows it.)
Prints one JSON line."""
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0, device_id=dev)
    from ae_wavenet_amd import autoencoder_model as ae, config, optim
    from ae_wavenet_amd.dp import DataParallel
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "cases": {}}
    n_win = int(os.environ.get("AEW_ONE_RANK_NWIN", "1000"))
    steps = 3

    def run(arch, dp_cfg):
        hps = config.make_hps(arch, n_win_batch=n_win, n_batch=8, jitter_prob=0.12)
        from ae_wavenet_amd import model as M
        # the plans of a real data-parallel rank (world > 1): decoder pack at the head of fwd_b, so that the deferred
        # head / decoder all-gather order and the region-wise wait are what is exercised
        M.TrainEngine.merge_packs = False if dp_cfg is not None else None
        torch.manual_seed(2507)
        model = ae.AutoEncoder(hps, n_mel=39).to(dev)
        opt = optim.FusedAdam(model, lr=1e-4)
        eng = model._ensure_engine(8)
        assert eng.use_graphs
        dp = None
        if dp_cfg is not None:
            dp = DataParallel(force_collectives=True)
            dp.attach(model, sharded=dp_cfg["sharded"], bf16_grads=dp_cfg.get("bf16", False))
            dp.broadcast_params(eng)
            dp.timing = True
        g = model.geom
        gen = torch.Generator().manual_seed(77)
        losses = []
        for i in range(steps):
            wav = torch.randint(0, 256, (8, g.enc_in_len), generator=gen).float().to(dev)
            mel = torch.randn(8, 39, g.mel_len, generator=gen).to(dev)
            voice = torch.randint(0, 40, (8,), generator=gen).to(dev)
            # jitter on (BASELINE configs[3]): offsets in {-1, 0, +1}, so conditioning vectors are read twice / not at all and
            # the backward's scatter really sums (its gather form: one order)
            jitter = (torch.arange(g.embed_len).repeat(8, 1) + torch.randint(-1, 2, (8, g.embed_len), generator=gen)) \
                .clamp_(0, g.embed_len - 1).to(dev)
            if arch == "vae":
                model.objective.update_anneal_weight(0.3 + 0.1 * i)
                torch.manual_seed(100 + i)                      # eps of the reparameterisation
            opt.zero_grad()
            pred, target, loss = model.run(wav, mel, voice, jitter)
            loss.backward()
            opt.step()
            losses.append(float(loss))
            if i == 0:                                          # the state after ONE step: what is compared bit for bit
                if dp is not None:
                    dp.finish()
                torch.cuda.synchronize()
                first = {"params": eng.ps.params[:eng.ps.numel].clone(), "m": eng.adam_m[:eng.ps.numel].clone(),
                         "v": eng.adam_v[:eng.ps.numel].clone()}
        exposed = None
        if dp is not None:
            dp.sync_optimizer_state(model)                      # finish() + all-gather of the sharded moments
            exposed = {k: round(v / steps, 4) for k, v in dp.exposed_ms().items()}
        torch.cuda.synchronize()
        # every element is compared: the step has ONE summation order (no mask; AEW_ONE_RANK_MASK_BIAS=1 restores round 5's
        # mask of the bias-type gradients for an A/B against aew_tuning_t.deterministic = 0)
        det = torch.ones(eng.ps.numel, dtype=torch.bool, device=dev)
        for nm in eng.ps.names() if os.environ.get("AEW_ONE_RANK_MASK_BIAS") == "1" else []:
            if nm.endswith(".bias") or "speaker_embedding" in nm:
                o = (eng.ps.view(nm).data_ptr() - eng.ps.params.data_ptr()) // 4
                det[o:o + eng.ps.numel_of(nm)] = False
        state = {"params": eng.ps.params[:eng.ps.numel].clone(), "m": eng.adam_m[:eng.ps.numel].clone(),
                 "v": eng.adam_v[:eng.ps.numel].clone(), "_det": det, "_first": first,
                 "_names": [(nm, (eng.ps.view(nm).data_ptr() - eng.ps.params.data_ptr()) // 4, eng.ps.numel_of(nm)) for nm in eng.ps.names()]}
        if eng.bn_type == "vqvae-ema":
            state["emb"], state["numer"] = eng.emb.clone(), eng.ema_numer.clone()
        del model, opt, eng
        torch.cuda.empty_cache()
        return losses, state, exposed

    for arch in ("vqvae-ema", "vae"):
        ref_l, ref_s, _ = run(arch, None)
        # "plain_again": the plain step a second time - what one process reproduces of itself (the yardstick for the rest)
        for name, cfg in (("plain_again", None), ("sharded", {"sharded": True}), ("all_reduce", {"sharded": False}),
                          ("sharded_bf16_grads", {"sharded": True, "bf16": True})):
            l, s, ex = run(arch, cfg)
            det = ref_s["_det"]
            keys = [k for k in ref_s if not k.startswith("_")]

            def same(k):                                        # flat-buffer tensors: the deterministic elements only
                return torch.equal(s[k][det], ref_s[k][det]) if s[k].numel() == det.numel() else torch.equal(s[k], ref_s[k])
            f1, f0 = s["_first"], ref_s["_first"]
            rec = {"losses": l, "ref_losses": ref_l, "exposed_collective_ms_per_step": ex,
                   "bit_equal_after_first_step": all(torch.equal(f1[k][det], f0[k][det]) for k in f0),
                   "first_step_max_abs_diff_deterministic": {k: float((f1[k] - f0[k])[det].abs().max()) for k in f0},
                   "max_abs_diff": {k: float((s[k] - ref_s[k]).abs().max()) for k in keys},
                   "max_rel_diff_atomic_sums": {k: (float(((s[k] - ref_s[k])[~det].abs().max() / ref_s[k][~det].abs().max().clamp_min(1e-30)))
                                                    if bool((~det).any()) else 0.0) for k in keys if s[k].numel() == det.numel()},
                   "max_abs_diff_deterministic": {k: float((s[k] - ref_s[k])[det].abs().max()) for k in keys if s[k].numel() == det.numel()},
                   "bit_equal": all(same(k) for k in keys)}
            if not rec["bit_equal"]:                            # which parameters: the five largest differences by name
                eng_names = ref_s["_names"]
                d = (s["params"] - ref_s["params"]).abs() * det
                top = []
                for nm, o, n_ in eng_names:
                    v = float(d[o:o + n_].max())
                    if v > 0:
                        top.append((v, nm))
                rec["params_differing"] = sorted(top, reverse=True)[:5]
            out["cases"][f"{arch}.{name}"] = rec
    print(json.dumps(out))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
