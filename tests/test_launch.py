"""CPU: the one-process-per-GPU launcher (unet_zoo_amd/launch.py) and bench.py's rank wiring — what replaces the
reference's single-process nn.DataParallel set-up (unet_zoo/utils/multi_gpu.py:20-31).  Two real child processes,
gloo, a stub step; and `bench.py --gpus 2` on a box with fewer than two GPUs must fail loudly, never print a
one-rank number."""
import io
import json
import os
import subprocess
import sys
import textwrap

import pytest
import torch

from unet_zoo_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = textwrap.dedent('''
    import json, os, sys
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["MASTER_ADDR"] == "127.0.0.1"
    assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    dist.init_process_group("gloo")
    # the stub "step": every rank contributes its shard's gradient, all ranks end with the average
    g = torch.full((4,), float(rank + 1))
    dist.all_reduce(g)
    g /= world
    t = torch.tensor([1.0 + rank])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)      # max-over-ranks timing, as bench.py reports it
    print(f"noise from rank {rank}")
    if len(sys.argv) > 1 and sys.argv[1] == "fail" and rank == 1:
        sys.exit(7)
    if len(sys.argv) > 1 and sys.argv[1] == "fail":
        import time; time.sleep(60)               # must be terminated by the launcher
    if rank == 0:
        print(json.dumps({"n_gpus": world, "avg": g.tolist(), "max_t": t.item()}), flush=True)
    dist.destroy_process_group()
''')


def _stub(tmp_path):
    p = os.path.join(tmp_path, "stub_rank.py")
    with open(p, "w") as f:
        f.write(STUB)
    return p


def test_spawn_two_ranks_relays_rank0_only(tmp_path):
    out, err = io.StringIO(), io.StringIO()
    rc = launch.spawn_ranks(2, [sys.executable, _stub(tmp_path)], need_gpus=False, stdout=out, stderr=err)
    assert rc == 0, err.getvalue()
    lines = [l for l in out.getvalue().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d == {"n_gpus": 2, "avg": [1.5] * 4, "max_t": 2.0}
    assert "noise from rank 0" in out.getvalue() and "noise from rank 1" not in out.getvalue()
    assert "[rank 1] noise from rank 1" in err.getvalue()


def test_failing_rank_stops_the_others_and_propagates(tmp_path):
    out, err = io.StringIO(), io.StringIO()
    rc = launch.spawn_ranks(2, [sys.executable, _stub(tmp_path), "fail"], need_gpus=False, stdout=out, stderr=err)
    assert rc == 7
    assert "rank 1 exited with code 7" in err.getvalue()
    assert not [l for l in out.getvalue().splitlines() if l.startswith("{")]


def test_refuses_fewer_gpus_than_ranks_and_nesting(tmp_path):
    err = io.StringIO()
    have = torch.cuda.device_count()
    rc = launch.spawn_ranks(have + 1, [sys.executable, "-c", "print(1)"], stdout=io.StringIO(), stderr=err)
    assert rc == 2 and "GPU(s) visible" in err.getvalue()
    err = io.StringIO()
    rc = launch.spawn_ranks(1, [sys.executable, "-c", "print(1)"], need_gpus=False, stdout=io.StringIO(), stderr=err,
                            env={**os.environ, "RANK": "0", "WORLD_SIZE": "2"})
    assert rc == 2 and "refusing to nest" in err.getvalue()
    assert launch.rank_info({"RANK": "3", "LOCAL_RANK": "1", "WORLD_SIZE": "4"}) == (3, 1, 4)
    assert launch.rank_info({}) == (0, 0, 1)


def _bench(args, env_extra=None, drop=()):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE") + tuple(drop)}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=300)


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="needs a box with fewer than two GPUs")
def test_bench_gpus_2_fails_loudly_without_two_gpus():
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "1"])
    assert r.returncode != 0
    assert "GPU(s) visible" in r.stderr
    assert "n_gpus" not in r.stdout          # no JSON line of any kind


def test_bench_rejects_rank_mismatch_and_ablation_env():
    r = _bench(["--gpus", "2"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert r.returncode == 2 and "--gpus 2 but WORLD_SIZE=1" in r.stderr and not r.stdout.strip()
    r = _bench(["--gpus", "1"], {"UZ_TUNE": "256"})
    assert r.returncode == 2 and "ablation" in r.stderr and not r.stdout.strip()


def test_launch_module_cli(tmp_path):
    r = subprocess.run([sys.executable, "-m", "unet_zoo_amd.launch", "--gpus", "2", "--cpu", _stub(tmp_path)],
                       cwd=ROOT, capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")})
    assert r.returncode == 0, r.stderr
    assert json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 2


SLEEPER = textwrap.dedent('''
    import os, sys, time
    open(os.path.join(sys.argv[1], "pid_%s" % os.environ["RANK"]), "w").write(str(os.getpid()))
    time.sleep(120)
''')


def _alive(pid):
    try:
        os.kill(pid, 0)
    except OSError:
        return False
    # a zombie still answers kill(0): look at its state
    try:
        with open(f"/proc/{pid}/stat") as f:
            return f.read().split(") ")[1][0] != "Z"
    except OSError:
        return False


@pytest.mark.parametrize("sig", ["TERM", "KILL"])
def test_a_killed_launcher_leaves_no_rank_process_behind(tmp_path, sig):
    """SIGTERM to `python -m unet_zoo_amd.launch` (a harness timeout, an operator) must take the rank processes down
    with it -- they would otherwise keep their GPUs and sit in a rendezvous or a collective; SIGKILL too (the children
    ask the kernel for SIGTERM on the launcher's death)"""
    import signal as _signal
    import time
    script = os.path.join(tmp_path, "sleeper.py")
    with open(script, "w") as f:
        f.write(SLEEPER)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    p = subprocess.Popen([sys.executable, "-m", "unet_zoo_amd.launch", "--gpus", "2", "--cpu", script, str(tmp_path)],
                         env=env, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    try:
        deadline = time.time() + 60
        while time.time() < deadline and not all(os.path.exists(os.path.join(tmp_path, f"pid_{r}")) for r in (0, 1)):
            time.sleep(0.1)
        pids = [int(open(os.path.join(tmp_path, f"pid_{r}")).read()) for r in (0, 1)]
        assert all(_alive(q) for q in pids)
        p.send_signal(_signal.SIGTERM if sig == "TERM" else _signal.SIGKILL)
        rc = p.wait(timeout=30)
        assert rc == (128 + _signal.SIGTERM if sig == "TERM" else -_signal.SIGKILL)
        deadline = time.time() + 15
        while time.time() < deadline and any(_alive(q) for q in pids):
            time.sleep(0.1)
        assert not any(_alive(q) for q in pids), "rank processes outlived the launcher"
    finally:
        if p.poll() is None:
            p.kill()
