"""`python bench.py --gpus N` must start its N ranks itself (round prompt: the driver may run the plain form).

CPU part: the launcher builds the torch.distributed.run command, passes the user's arguments through and returns the child's
exit code, before anything GPU-related is imported.  GPU part (`-m gpu`): the plain form really runs two ranks on the box's
one GPU (gloo transport: RCCL wants one device per rank) on a tiny shape and rank 0 prints a line with n_gpus == 2."""
import json
import os
import subprocess
import sys
import types

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_plain_form_starts_the_ranks_as_a_child_process(monkeypatch):
    bench = _load_bench()
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=7)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                    # the child's return code is relayed
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(REPO, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_a_launched_rank_does_not_launch_again(monkeypatch):
    """With RANK / WORLD_SIZE in the environment (torch.distributed.run's children) main() must go on to the benchmark, not
    start another job: the launcher function is not reached."""
    bench = _load_bench()
    called = []
    monkeypatch.setattr(bench, "launch_ranks", lambda n: called.append(n) or 0)
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])

    class Stop(Exception):
        pass

    import builtins
    real_import = builtins.__import__

    def guard(name, *a, **kw):
        if name == "nerve_cl":
            raise Stop()                                         # reached the benchmark body
        return real_import(name, *a, **kw)

    monkeypatch.setattr(builtins, "__import__", guard)
    with pytest.raises(Stop):
        bench.main()
    assert called == []


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_plain_form_runs_two_ranks_on_the_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2",
                        "--warmup", "1", "--batch", "1", "--height", "32", "--width", "48", "--features", "16", "--blocks",
                        "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=840)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 2 and line["config"]["parallelism"] == "dp2"
    assert line["value"] > 0 and line["cpu_baseline"] is None
