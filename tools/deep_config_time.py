"""Step time of BASELINE configs[4]'s per-GPU shape (DEEP: 30 layers x 512 residual channels, 64k-sample
windows, B = 4 per GPU) and of configs[3] (VAE bottleneck, B = 8, w = 5000): forward + backward + Adam."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ae_wavenet_amd import config as C_, model as M

DEV = "cuda:0"
for arch, B, w in (("deep", 4, 65536), ("vae", 8, 5000)):
    hps = C_.make_hps(arch, n_win_batch=w)
    eng = M.TrainEngine(hps, B=B, device=DEV, n_mel=39)
    gen = torch.Generator().manual_seed(1)
    for k in eng.ps.names():
        t = torch.empty(eng.ps.shape[k])
        torch.nn.init.xavier_uniform_(t, generator=gen) if t.dim() >= 2 else t.zero_()
        eng.ps.view(k).copy_(t)
    g = eng.geom
    if eng.bn_type == "vqvae-ema":
        eng.emb.normal_(); eng.init_ema_from_emb()
    wav = torch.randint(0, 256, (B, g.enc_in_len), generator=gen).float().to(DEV)
    mel = torch.randn(B, 39, g.mel_len, generator=gen).to(DEV)
    voice = torch.randint(0, 40, (B,), generator=gen).to(DEV)
    jitter = torch.arange(g.embed_len).repeat(B, 1).to(DEV)
    eps = torch.randn(B, g.embed_len, hps.bn_n_out, generator=gen).to(DEV) if arch == "vae" else None
    eng.set_inputs(wav, mel, voice, jitter, eps=eps)
    for _ in range(3):
        eng.forward(); eng.backward(); eng.adam_step(1e-4)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5 if arch == "deep" else 20
    for _ in range(n):
        eng.forward(); eng.backward(); eng.adam_step(1e-4)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    fl = eng.flops_per_step() if hasattr(eng, "flops_per_step") else {"step": 0}
    print(f"{arch:5s} B={B} w={w}: {dt * 1e3:8.2f} ms/step  {B * w / dt / 1e6:6.2f} M samples/s  "
          f"{fl['step'] / dt / 1e12:6.1f} TFLOP/s  workspace {eng.ws.nbytes() / 2**30:.1f} GiB")
    del eng; torch.cuda.empty_cache()
