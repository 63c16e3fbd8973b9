"""bench.py --gpus N must start N ranks itself (the driver's single-command invocation) without the launching process
ever touching the GPU; started by torchrun (WORLD_SIZE set) it must run as a rank instead."""
import importlib
import sys
import types

import pytest


@pytest.fixture()
def bench(monkeypatch):
    sys.modules.pop("bench", None)
    return importlib.import_module("bench")


def test_gpus_flag_spawns_ranks_without_touching_the_gpu(bench, monkeypatch):
    calls = {}

    def fake_run(cmd, env=None, **kw):
        if "-c" in cmd:  # the device-count probe: a child process, so that the launcher itself stays free of torch and HIP
            calls["probe"] = list(cmd)
            return types.SimpleNamespace(returncode=0, stdout="8\n", stderr="")
        calls["cmd"], calls["env"] = list(cmd), dict(env or {})
        return types.SimpleNamespace(returncode=0)

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench, "run", lambda args: (_ for _ in ()).throw(AssertionError("the launcher must not run a rank")))
    torch_cuda_before = "torch.cuda" in sys.modules and getattr(sys.modules["torch.cuda"], "_initialized", False)
    rc = bench.main(["--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert rc == 0
    assert "device_count" in calls["probe"][-1]
    cmd = calls["cmd"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    script = cmd.index(str((bench.ROOT / "bench.py").resolve()))
    assert cmd[script + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]  # the ranks see the same flags
    assert calls["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    if "torch.cuda" in sys.modules:  # nothing above may have initialised the GPU runtime
        assert getattr(sys.modules["torch.cuda"], "_initialized", False) == torch_cuda_before


def test_launcher_relays_a_failing_child(bench, monkeypatch):
    monkeypatch.setattr(bench.subprocess, "run",
                        lambda cmd, env=None, **kw: types.SimpleNamespace(returncode=3, stdout="8\n", stderr=""))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.main(["--gpus", "4"]) == 3


def test_launcher_refuses_a_node_with_fewer_devices(bench, monkeypatch, capsys):
    """--gpus 8 on a node that shows 4 devices: no rank is started, nothing is printed on stdout, exit code 3."""
    started = []

    def fake_run(cmd, env=None, **kw):
        if "-c" in cmd:
            return types.SimpleNamespace(returncode=0, stdout="4\n", stderr="")
        started.append(cmd)
        return types.SimpleNamespace(returncode=0)

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.main(["--gpus", "8"]) == 3 and not started
    out = capsys.readouterr()
    assert out.out == "" and "shows 4 device(s)" in out.err


def test_inside_torchrun_it_is_a_rank(bench, monkeypatch):
    seen = {}
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(bench.subprocess, "run", lambda *a, **k: (_ for _ in ()).throw(AssertionError("no second launch")))
    monkeypatch.setattr(bench, "run", lambda args: seen.setdefault("gpus", args.gpus) and 0)
    assert bench.main(["--gpus", "2"]) == 0 and seen["gpus"] == 2


def test_single_gpu_default_runs_in_process(bench, monkeypatch):
    seen = {}
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench.subprocess, "run", lambda *a, **k: (_ for _ in ()).throw(AssertionError("no launch at N = 1")))
    monkeypatch.setattr(bench, "run", lambda args: seen.update(steps=args.steps) or 0)
    assert bench.main([]) == 0 and seen["steps"] == 40
