"""Data parallel on the GPU: `bench.py --gpus 2` spawns two ranks through torch.distributed.run and runs the real
training step (module surface + FusedAdam, collectives inside) on both (train.py:58-60: one process per device;
chassis.py:168-169: the optimizer step carries the gradient exchange).

With two devices visible the ranks run one per device over RCCL ("nccl").  The test box of this pool has ONE MI355X and
RCCL refuses two ranks on one device ("duplicate GPU"), so there both ranks share cuda:0 (AEW_BENCH_SHARE_GPU=1) over
gloo - device tensors staged through the host by torch, the same torch.distributed calls: a functional check, not a
timing; the test prints which one ran.  What is held:
  * the process group really has 2 ranks (the JSON line's n_gpus comes from dist.get_world_size());
  * after the steps every parameter and the codebook are IDENTICAL on both ranks (all-reduce MAX - MIN == 0);
  * the reduce-scatter + sharded-Adam + all-gather schedule and the all-reduce schedule reach the same loss.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(backend, share, sharded, timeout=420):
    env = dict(os.environ, AEW_BENCH_BACKEND=backend, AEW_DP_SHARDED="1" if sharded else "0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env["AEW_BENCH_SHARE_GPU"] = "1" if share else "0"
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--check-replicas", "--n-win", "1000"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return None, (r.stdout[-2000:] + "\n" + r.stderr[-4000:])
    return json.loads(lines[-1]), ""


def test_two_ranks_train_identically_through_bench():
    import torch
    if torch.cuda.device_count() >= 2:
        backend, share = "nccl", False                  # two devices: RCCL, one rank per device
    else:
        # one device: RCCL refuses two ranks on it ("duplicate GPU", measured on this pool), so both ranks share cuda:0
        # over gloo - the same torch.distributed calls, staged through the host by torch
        backend, share = "gloo", True
    out, err = _bench(backend, share, True)
    assert out is not None, err
    print(f"two ranks over {backend} ({'one shared device: not a measurement' if share else 'two devices'}): "
          f"{out['ms_per_step']:.2f} ms/step, data_parallel = {out['data_parallel']}")
    dp = out["data_parallel"]
    assert out["n_gpus"] == 2 and dp["backend"] == backend and dp["ranks_share_one_gpu"] == share
    assert "reduce-scatter" in out["config"]["parallelism"]
    assert dp["replica_param_max_diff"] == 0.0 and dp["replica_codebook_max_diff"] == 0.0
    assert dp["exposed_collective_ms_per_step"] >= 0.0 and "params.all_gather" in dp["by_wait_ms_per_step"]
    loss_sharded = out["config"]["loss"]
    ref, err = _bench(backend, share, False)
    assert ref is not None, err
    assert "all-reduce" in ref["config"]["parallelism"] and ref["data_parallel"]["replica_param_max_diff"] == 0.0
    # same data, same seeds, same number of optimizer steps: the two schedules differ by fp32 summation order only
    assert abs(loss_sharded / ref["config"]["loss"] - 1) < 2e-3, (loss_sharded, ref["config"]["loss"])
