"""Single-rank RCCL smoke of the overlapped exchange: the collective path (async all-reduce issued between
the two backward graphs, waited before Adam) must leave the gradients exactly as a plain backward does."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import time
import torch, torch.distributed as dist
from test_gpu_parity import seeded_full_engine, DEV
from ae_wavenet_amd.dp import DataParallel

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
hps, eng, wts, emb, inp = seeded_full_engine(B=8, w=5000, seed=5)
eng.set_inputs(*[t.to(DEV) for t in inp])
d = DataParallel()
eng.init_ema_from_emb(); eng.forward(); eng.backward(); torch.cuda.synchronize()
ref = eng.ps.grads[:eng.ps.numel].clone()
d.world = 2                                   # take the collective path although the group has one rank
for it in range(3):
    eng.init_ema_from_emb(); eng.forward(d.allreduce_ema); d.backward_allreduce(eng); torch.cuda.synchronize()
    g = eng.ps.grads[:eng.ps.numel]
    rel = (g - ref).abs().max().item() / ref.abs().max().item()
    assert rel < 1e-5, rel      # bias sums are accumulated with fp32 atomics: round-off level only
# full overlapped step against the plain one: same parameters after two optimizer steps
import copy
state = {k: eng.ws.get(k).clone() for k in ("params", "adam.m", "adam.v", "bn.emb", "bn.ema_numer", "bn.ema_denom")}
def restore():
    for k, v in state.items():
        eng.ws.get(k).copy_(v)
    eng.step_count = 0
restore()
for it in range(2):
    eng.forward(); eng.backward(); eng.adam_step(1e-3)
torch.cuda.synchronize()
p_ref = eng.ps.params[:eng.ps.numel].clone(); e_ref = eng.emb.clone()
restore()
for it in range(2):
    d.train_step(eng, 1e-3)
torch.cuda.synchronize()
dp_ = (eng.ps.params[:eng.ps.numel] - p_ref).abs().max().item()
de_ = (eng.emb - e_ref).abs().max().item()
assert dp_ < 1e-5 and de_ == 0.0, (dp_, de_)
torch.cuda.synchronize(); t = time.perf_counter()
for it in range(10):
    d.train_step(eng, 1e-4)
torch.cuda.synchronize()
print(f"overlapped-exchange path ok (params {dp_:.1e}, codebook exact); {(time.perf_counter() - t) / 10 * 1e3:.3f} ms/step with 1-rank collectives")
dist.destroy_process_group()
