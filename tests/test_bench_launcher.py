"""bench.py without a launcher: `python bench.py --gpus N` (WORLD_SIZE unset) must start its own N ranks as child
processes through torch.distributed.run BEFORE it imports torch or the library (CPU test: nothing is launched)."""
import importlib
import sys
import types

import pytest


@pytest.fixture
def bench(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    mod = importlib.import_module("bench")
    assert mod.torch is None and mod.native is None, "bench.py must not import torch / the library at import time"
    return mod


def test_self_launch_command(bench, monkeypatch):
    seen = {}

    def fake_run(cmd, **kw):
        seen["cmd"] = cmd
        return types.SimpleNamespace(returncode=7)

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    rc = bench.self_launch(types.SimpleNamespace(gpus=4), ["--gpus", "4", "--steps", "3", "--warmup", "1"])
    cmd = seen["cmd"]
    assert rc == 7
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    k = [i for i, c in enumerate(cmd) if c.endswith("bench.py")][0]
    assert cmd[k + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]


def test_main_launches_before_touching_the_gpu(bench, monkeypatch):
    calls = []
    monkeypatch.setattr(bench, "self_launch", lambda a, argv: calls.append((a.gpus, list(argv))) or 0)
    monkeypatch.setattr(bench, "late_imports", lambda: pytest.fail("the parent of a self-launched run imported torch"))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and calls == [(2, ["--gpus", "2", "--steps", "1", "--warmup", "0"])]


def test_world_size_mismatch_is_an_error_message(bench, monkeypatch):
    """under a launcher with another world size: SystemExit with an explanation, not an assertion"""
    monkeypatch.setenv("WORLD_SIZE", "2")
    fake_torch = types.SimpleNamespace(cuda=types.SimpleNamespace(set_device=lambda d: None))
    monkeypatch.setattr(bench, "torch", fake_torch)
    a = types.SimpleNamespace(gpus=4, same_device=False, backend="gloo", force_collectives=False)
    with pytest.raises(SystemExit) as e:
        bench.Job(a)
    assert "--gpus 4 but WORLD_SIZE=2" in str(e.value)
