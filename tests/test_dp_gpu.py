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


def _bench(backend, share, sharded, timeout=420, gpus=2, n_win=1000, extra_env=None):
    env = dict(os.environ, AEW_BENCH_BACKEND=backend, AEW_DP_SHARDED="1" if sharded else "0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env["AEW_BENCH_SHARE_GPU"] = "1" if share else "0"
    env.pop("WORLD_SIZE", None)
    env.update(extra_env or {})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--check-replicas", "--n-win", str(n_win)]
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


def test_eight_ranks_spawn_and_report_through_bench():
    """The driver's first 8-GPU run (BASELINE configs[2]) must not be the first time `bench.py --gpus 8` exists: the spawn
    path, eight ranks in one process group, the three-wait-point sharded step and the JSON line (`n_gpus: 8`,
    `data_parallel.by_wait_ms_per_step`) - here with all eight ranks on the test box's device over gloo (a functional check,
    small windows), or one per device over RCCL where the box has eight.  train.py:58-60, chassis.py:168-169."""
    import torch
    if torch.cuda.device_count() >= 8:
        backend, share = "nccl", False
    else:
        backend, share = "gloo", True
    out, err = _bench(backend, share, True, timeout=900, gpus=8, n_win=100)
    assert out is not None, err
    dp = out["data_parallel"]
    print(f"eight ranks over {backend}: {out['ms_per_step']:.2f} ms/step ({'one shared device: not a measurement' if share else '8 devices'}), "
          f"waits {dp['by_wait_ms_per_step']}")
    assert out["n_gpus"] == 8 and out["config"]["global_batch"] == 64 and out["scaling"] == "weak"
    assert dp["backend"] == backend and dp["ranks_share_one_gpu"] == share
    assert dp["replica_param_max_diff"] == 0.0 and dp["replica_codebook_max_diff"] == 0.0
    assert {"grads.decoder", "grads.encoder", "params.all_gather"} <= set(dp["by_wait_ms_per_step"]), dp
    assert out["cpu_baseline"]["value"] is None and out["value"] > 0


def test_rccl_carries_every_product_collective_on_one_rank():
    """RCCL ("nccl") with ONE rank and DataParallel(force_collectives=True): the data-parallel step's collectives -
    reduce-scatter (fp32 / bf16 transport), all-gather pairs, async EMA all-reduce, scalar KL all-reduce, broadcast, the
    moments' all-gather - are really issued, through the module surface, against captured graphs.  One rank makes each of
    them an identity: the fp32 schedules must reproduce the plain single-process training steps bit for bit
    (chassis.py:168-169: the optimizer step carries the exchange).  tools/dp_rccl_one_rank.py runs it in a fresh process."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dp_rccl_one_rank.py")], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and lines, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    out = json.loads(lines[-1])
    assert out["backend"] == "nccl" and out["world"] == 1
    print("exposed collective ms per step (one rank, RCCL):",
          {k: v["exposed_collective_ms_per_step"] for k, v in out["cases"].items()})
    lr, steps = 1e-4, 3
    for name, rec in out["cases"].items():
        if name.endswith("plain_again"):
            # the plain run against itself over three steps: the whole state bit for bit - no masked elements since the
            # bias-type sums have one order (aew_tuning_t.deterministic, ABI 20)
            assert rec["bit_equal"] and rec["losses"] == rec["ref_losses"], (name, rec["max_abs_diff"], rec.get("params_differing"))
            continue
        assert all(abs(a / b - 1) < 1e-6 for a, b in zip(rec["losses"][:1], rec["ref_losses"][:1])), (name, rec["losses"])
        if name.endswith("bf16_grads"):
            # the gradient passes through a bf16 copy once: Adam's normalised update may flip for the smallest ones
            assert all(abs(a / b - 1) < 2e-2 for a, b in zip(rec["losses"], rec["ref_losses"])), (name, rec["losses"])
            assert rec["max_abs_diff"]["params"] < 2.05 * steps * lr, (name, rec["max_abs_diff"])
        else:
            # after the first step: bit for bit - parameters and both Adam moments - on everything one process reproduces at
            # all (bias-type gradients are fp32-atomic column sums: round-off)
            assert rec["bit_equal_after_first_step"], (name, rec["first_step_max_abs_diff_deterministic"])
            # ... and after all three: every parameter, both moments, codebook, EMA numerator, every element
            assert rec["bit_equal"] and rec["losses"] == rec["ref_losses"], (name, rec["max_abs_diff"], rec.get("params_differing"))
        if "sharded" in name:                                       # async collectives: their waits were really taken
            ex = rec["exposed_collective_ms_per_step"]
            assert {"grads.decoder", "grads.encoder", "params.all_gather"} <= set(ex), (name, ex)
