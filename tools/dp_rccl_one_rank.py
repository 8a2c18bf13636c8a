#!/usr/bin/env python3
"""RCCL under the product's collectives on ONE GPU (train.py:58-60, chassis.py:168-169,188-190).

A process group of one rank over backend "nccl" (= RCCL on ROCm) and a DataParallel(force_collectives=True) that does not
take its world == 1 short cuts: every collective of the data-parallel step is then really issued to RCCL with the product's
own views, dtypes and stream ordering - reduce_scatter_tensor (fp32 and through the bf16 transport copy), sharded Adam,
the deferred head / decoder all_gather_into_tensor pair, the async all_reduce of the EMA statistics under the decoder
forward + backward, all_reduce of the scalar KL, broadcast of parameters and codebook, the all-gather of the Adam moments
- against plans replayed as captured hipGraphs on torch's current stream.  With one rank every collective is an
identity, so the step must equal the plain single-process step BIT FOR BIT (fp32 transport).  Prints one JSON line."""
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
            jitter = torch.arange(g.embed_len).repeat(8, 1).to(dev)
            if arch == "vae":
                model.objective.update_anneal_weight(0.3 + 0.1 * i)
                torch.manual_seed(100 + i)                      # eps of the reparameterisation
            opt.zero_grad()
            pred, target, loss = model.run(wav, mel, voice, jitter)
            loss.backward()
            opt.step()
            losses.append(float(loss))
        exposed = None
        if dp is not None:
            dp.sync_optimizer_state(model)                      # finish() + all-gather of the sharded moments
            exposed = {k: round(v / steps, 4) for k, v in dp.exposed_ms().items()}
        torch.cuda.synchronize()
        state = {"params": eng.ps.params[:eng.ps.numel].clone(), "m": eng.adam_m[:eng.ps.numel].clone(),
                 "v": eng.adam_v[:eng.ps.numel].clone()}
        if eng.bn_type == "vqvae-ema":
            state["emb"], state["numer"] = eng.emb.clone(), eng.ema_numer.clone()
        del model, opt, eng
        torch.cuda.empty_cache()
        return losses, state, exposed

    for arch in ("vqvae-ema", "vae"):
        ref_l, ref_s, _ = run(arch, None)
        for name, cfg in (("sharded", {"sharded": True}), ("all_reduce", {"sharded": False}),
                          ("sharded_bf16_grads", {"sharded": True, "bf16": True})):
            l, s, ex = run(arch, cfg)
            rec = {"losses": l, "ref_losses": ref_l, "exposed_collective_ms_per_step": ex,
                   "max_abs_diff": {k: float((s[k] - ref_s[k]).abs().max()) for k in ref_s},
                   "bit_equal": all(torch.equal(s[k], ref_s[k]) for k in ref_s)}
            out["cases"][f"{arch}.{name}"] = rec
    print(json.dumps(out))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
