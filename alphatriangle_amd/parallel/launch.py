"""Single-command multi-GPU start-up: one parent process that starts one fresh child per GPU.

The reference starts its whole worker pool from one command
(alphatriangle/training/worker_manager.py:39-75 creates N Ray actors,
alphatriangle/training/setup.py:133-151 sizes the pool).  The equivalent here is a
parent that has made no GPU call at all and starts ``n`` children of the same
program with torchrun-style environment variables (``RANK``, ``LOCAL_RANK``,
``WORLD_SIZE``, ``MASTER_ADDR``, ``MASTER_PORT``); every child then joins one RCCL
job through :func:`alphatriangle_amd.parallel.dist.init_from_env`.

The parent never re-executes itself and never touches the GPU: it only forwards
rank 0's stdout, supervises the children and turns "one rank died" into "every
rank stops, exit code non-zero" within a bounded time.
"""

from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys
import time
from typing import Sequence

LAUNCH_ENV_KEYS = ("RANK", "LOCAL_RANK", "WORLD_SIZE")


def under_launcher() -> bool:
    """True when this process is already one rank of a multi-process job (torchrun or :func:`launch_local_ranks`)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def world_size_from_env() -> int:
    return int(os.environ.get("WORLD_SIZE", "1"))


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def launch_local_ranks(
    n: int,
    argv: Sequence[str],
    *,
    extra_env: dict[str, str] | None = None,
    grace_s: float = 20.0,
    poll_s: float = 0.2,
) -> int:
    """Start ``n`` child processes running ``argv`` (one per local GPU) and wait for all of them.

    ``argv`` is the full child command line (``[sys.executable, script, ...]``).  Rank 0 inherits
    this process's stdout; the other ranks' stdout goes to stderr so that a program which prints one
    result line from rank 0 still prints exactly one line.  Returns 0 when every child exited 0.
    When a child exits non-zero (or is killed), the remaining children get SIGTERM, then SIGKILL
    after ``grace_s`` seconds, and the first failing code is returned (signals map to 128 + n).
    """
    if n < 1:
        raise ValueError("need at least one rank")
    port = free_port()
    base = dict(os.environ)
    for k in LAUNCH_ENV_KEYS:
        base.pop(k, None)
    base.update(
        WORLD_SIZE=str(n),
        MASTER_ADDR="127.0.0.1",
        MASTER_PORT=str(port),
        ATRI_LAUNCHED="1",
    )
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if extra_env:
        base.update(extra_env)
    procs: list[subprocess.Popen] = []
    try:
        for r in range(n):
            env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen(list(argv), env=env, stdout=None if r == 0 else sys.stderr))
        first_bad = 0
        deadline: float | None = None
        while True:
            codes = [p.poll() for p in procs]
            if all(c is not None for c in codes):
                break
            bad = [c for c in codes if c not in (None, 0)]
            if bad and deadline is None:
                first_bad = bad[0]
                deadline = time.monotonic() + grace_s
                for p in procs:
                    if p.poll() is None:
                        p.send_signal(signal.SIGTERM)
            if deadline is not None and time.monotonic() > deadline:
                for p in procs:
                    if p.poll() is None:
                        p.kill()
                deadline = float("inf")
            time.sleep(poll_s)
        codes = [p.returncode for p in procs]
        if first_bad == 0:
            first_bad = next((c for c in codes if c != 0), 0)
        return (128 - first_bad) if first_bad < 0 else first_bad
    except KeyboardInterrupt:
        for p in procs:
            if p.poll() is None:
                p.send_signal(signal.SIGINT)
        for p in procs:
            try:
                p.wait(timeout=grace_s)
            except subprocess.TimeoutExpired:
                p.kill()
        return 130
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
