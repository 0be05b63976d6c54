"""One process per GPU: the launcher behind ``bench.py --gpus N`` and any ``ddp_rccl`` training script.

The reference parallelises with a single-process ``nn.DataParallel`` wrapper
(unet_zoo/utils/multi_gpu.py:20-31); here every GPU gets its own process and gradients meet in RCCL
all-reduces.  ``spawn_ranks`` is what turns "run on N GPUs" into N processes:

* it must be called BEFORE the calling process makes any GPU call (it only counts devices, which does not
  initialise the runtime), because a process that has touched the GPU must not be duplicated or re-executed;
* every child is a FRESH interpreter (``subprocess``, no fork of torch state) with ``RANK``, ``LOCAL_RANK``,
  ``WORLD_SIZE``, ``MASTER_ADDR=127.0.0.1``, ``MASTER_PORT`` and ``HSA_ENABLE_IPC_MODE_LEGACY=0`` set — the same
  contract ``python -m torch.distributed.run`` gives, so a script works under either;
* rank 0's stdout is relayed verbatim (the one JSON line of bench.py), the other ranks' stdout goes to stderr with a
  ``[rank k]`` prefix;
* the exit code is non-zero when any rank fails (the survivors are terminated by their exact PIDs) and when fewer
  than N GPUs are visible — never a silent single-rank run.
"""
from __future__ import annotations

import os
import socket
import signal
import subprocess
import sys
import threading
import time
from typing import Dict, List, Optional, Sequence

RANK_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE")


def visible_gpus() -> int:
    """Number of visible GPUs WITHOUT initialising the HIP runtime in this process."""
    import torch
    return int(torch.cuda.device_count())


def under_launcher(env: Optional[Dict[str, str]] = None) -> bool:
    """True inside a rank process (this launcher's or torch.distributed.run's)."""
    env = os.environ if env is None else env
    return "WORLD_SIZE" in env and "RANK" in env


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


_LAUNCH_STAMP = None


def rank_env(rank: int, world: int, port: int, base: Optional[Dict[str, str]] = None) -> Dict[str, str]:
    """Environment of one rank (single node: LOCAL_RANK == RANK).  UZ_RUN_TIMESTAMP: one value for every rank of this
    launcher process (config.Config derives the run directory from it)."""
    global _LAUNCH_STAMP
    if _LAUNCH_STAMP is None:
        _LAUNCH_STAMP = time.strftime("%Y%m%d-%H%M%S")
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    env.setdefault("UZ_RUN_TIMESTAMP", _LAUNCH_STAMP)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    return env


def _pump(stream, sink, prefix: str) -> None:
    for line in iter(stream.readline, ""):
        sink.write(prefix + line)
        sink.flush()
    stream.close()


def _die_with_parent() -> None:
    """child side, before exec: have the kernel send SIGTERM when the launcher dies (even by SIGKILL) -- Linux prctl"""
    try:
        import ctypes
        ctypes.CDLL(None, use_errno=True).prctl(1, signal.SIGTERM)   # PR_SET_PDEATHSIG
    except Exception:
        pass


def spawn_ranks(world: int, argv: Sequence[str], *, need_gpus: bool = True, port: Optional[int] = None,
                env: Optional[Dict[str, str]] = None, poll_s: float = 0.1, stdout=None, stderr=None) -> int:
    """Run ``argv`` as `world` rank processes; returns the exit code the caller should exit with.

    argv        the child command line (e.g. [sys.executable, "bench.py", "--gpus", "4", ...])
    need_gpus   refuse (exit code 2, message on stderr) when fewer than `world` GPUs are visible
    """
    stdout = sys.stdout if stdout is None else stdout
    stderr = sys.stderr if stderr is None else stderr
    if world < 1:
        stderr.write(f"launch: --gpus must be >= 1, got {world}\n")
        return 2
    if under_launcher(env):
        stderr.write("launch: already inside a rank process (RANK / WORLD_SIZE are set); refusing to nest\n")
        return 2
    if need_gpus:
        have = visible_gpus()
        if have < world:
            stderr.write(f"launch: {world} ranks requested but only {have} GPU(s) visible; refusing to run "
                         f"fewer ranks than asked (no number is reported)\n")
            return 2
    port = free_port() if port is None else port
    procs: List[subprocess.Popen] = []
    pumps: List[threading.Thread] = []
    rc = 0
    stop = {"sig": 0}

    def _on_signal(signum, _frame):      # SIGTERM / SIGINT to the launcher: stop the ranks, then leave with 128 + signum
        stop["sig"] = signum

    old_handlers = {}
    if threading.current_thread() is threading.main_thread():
        for sg in (signal.SIGTERM, signal.SIGINT):
            old_handlers[sg] = signal.signal(sg, _on_signal)

    def _stop_children(grace_s: float = 10.0) -> None:
        """terminate, then kill, every child still alive -- by the exact PIDs this function started, never a pattern"""
        live = [p for p in procs if p.poll() is None]
        for p in live:
            p.terminate()
        deadline = time.time() + grace_s
        while any(p.poll() is None for p in live) and time.time() < deadline:
            time.sleep(poll_s)
        for p in live:
            if p.poll() is None:
                p.kill()
        for p in live:
            try:
                p.wait(timeout=5.0)
            except subprocess.TimeoutExpired:
                pass

    try:
        for r in range(world):
            p = subprocess.Popen(list(argv), env=rank_env(r, world, port, env), stdout=subprocess.PIPE,
                                 stderr=subprocess.PIPE, text=True, bufsize=1, preexec_fn=_die_with_parent)
            procs.append(p)
            out_sink, out_prefix = (stdout, "") if r == 0 else (stderr, f"[rank {r}] ")
            for stream, sink, prefix in ((p.stdout, out_sink, out_prefix),
                                         (p.stderr, stderr, f"[rank {r}] " if world > 1 else "")):
                t = threading.Thread(target=_pump, args=(stream, sink, prefix), daemon=True)
                t.start()
                pumps.append(t)
        alive = set(range(world))
        while alive:
            if stop["sig"]:
                stderr.write(f"launch: signal {stop['sig']}; stopping the ranks\n")
                rc = 128 + stop["sig"]
                break
            for r in sorted(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    stderr.write(f"launch: rank {r} exited with code {code}; stopping the other ranks\n")
            if rc != 0:
                break
            if alive:
                time.sleep(poll_s)
    finally:
        # whatever ended the loop -- a failed rank, a signal, Popen raising for rank k > 0, an exception in here --
        # no rank process outlives the launcher
        _stop_children()
        for sg, h in old_handlers.items():
            signal.signal(sg, h)
    for t in pumps:
        t.join(timeout=5.0)
    return rc


def rank_info(env: Optional[Dict[str, str]] = None):
    """(rank, local_rank, world) of this process; (0, 0, 1) outside a launcher."""
    env = os.environ if env is None else env
    return int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0")), int(env.get("WORLD_SIZE", "1"))


def main(argv: Optional[Sequence[str]] = None) -> int:
    """python -m unet_zoo_amd.launch --gpus N script.py [script args]"""
    import argparse
    ap = argparse.ArgumentParser(prog="python -m unet_zoo_amd.launch",
                                 description="start one rank process per GPU of this node")
    ap.add_argument("--gpus", type=int, required=True)
    ap.add_argument("--cpu", action="store_true", help="do not require GPUs (gloo rehearsals)")
    ap.add_argument("script")
    ap.add_argument("args", nargs=argparse.REMAINDER)
    a = ap.parse_args(argv)
    return spawn_ranks(a.gpus, [sys.executable, a.script] + a.args, need_gpus=not a.cpu)


if __name__ == "__main__":
    sys.exit(main())
