"""bench.py --gpus N without a launcher spawns N ranks itself (one process per GPU over torch.distributed.run on
127.0.0.1), and refuses to report an N-GPU number on a node with fewer devices."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _args(n):
    import bench
    old = sys.argv
    sys.argv = ["bench.py", "--gpus", str(n), "--steps", "2", "--warmup", "1"]
    try:
        return bench, bench.parse()
    finally:
        sys.argv = old


def test_spawn_builds_a_torchrun_command(monkeypatch):
    bench, args = _args(2)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setenv("AEW_BENCH_SHARE_GPU", "1")            # this container has no GPU
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.spawn(args)
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    i = cmd.index("--master-addr")
    assert cmd[i + 1] == "127.0.0.1" and "--master-port" in cmd
    assert cmd[-6:] == ["--gpus", "2", "--steps", "2", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_refuses_more_ranks_than_devices(monkeypatch):
    import torch
    bench, args = _args(8)
    monkeypatch.delenv("AEW_BENCH_SHARE_GPU", raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit) as e:
        bench.spawn(args)
    assert "exposes 1 GPU" in str(e.value.code)


def test_main_spawns_only_without_a_launcher(monkeypatch):
    import bench
    called = []
    monkeypatch.setattr(bench, "spawn", lambda a: (_ for _ in ()).throw(SystemExit(called.append(a.gpus) or 0)))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit):
        bench.main()
    assert called == [4]
