"""One process per GPU, started from a plain `python3 program.py --gpus N` command line.

The parent never touches the GPU (no torch.cuda, no HIP context: it imports nothing that does): it starts N
fresh children of the same command line with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set,
relays what rank 0 prints, enforces a time limit and reports the first failing rank's status as its own.
"""
from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys
import threading
import time


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def needs_spawn(gpus: int, environ=os.environ) -> bool:
    """True for the parent of an N>1 run: more than one GPU asked for and no rank environment present."""
    return gpus > 1 and "RANK" not in environ and int(environ.get("WORLD_SIZE", "1")) == 1


def spawn_ranks(argv: list[str], world: int, *, time_limit_s: float = 1500.0, environ=os.environ, out=sys.stdout, err=sys.stderr,
                is_result=None) -> int:
    """Runs `sys.executable argv...` once per rank and returns the exit status for the parent: 0 when every rank
    exited 0, else the first non-zero status seen (124 on the time limit).  Rank 0's stdout is relayed to `out`
    (only the lines for which is_result(line) holds when that is given - libraries print to stdout too - the rest
    goes to `err`); the other ranks' stdout is dropped (they print nothing by contract); every rank's stderr goes to `err`.
    A rank that fails, or the time limit, ends the others: each child leads its own process group and the whole
    group is signalled, so nothing a rank started is left behind."""
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env, stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL,
                                      stderr=None if err is sys.stderr else err, text=True, start_new_session=True))

    def end_all(sig):
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, sig)
                except (ProcessLookupError, PermissionError):
                    pass

    relayed = []

    def relay():  # rank 0's stdout is drained as it comes, so that a long line never fills the pipe
        for line in procs[0].stdout:
            relayed.append(line)

    reader = threading.Thread(target=relay, daemon=True)
    reader.start()
    status = 0
    deadline = time.monotonic() + time_limit_s
    try:
        pending = set(range(world))
        while pending:
            for r in sorted(pending):
                rc = procs[r].poll()
                if rc is None:
                    continue
                pending.discard(r)
                if rc != 0 and status == 0:
                    status = rc if rc > 0 else 128 - rc
                    print(f"rank {r} exited with status {rc}: ending the other ranks", file=err, flush=True)
                    end_all(signal.SIGTERM)
            if pending and time.monotonic() > deadline:
                print(f"time limit of {time_limit_s:.0f} s reached: ending ranks {sorted(pending)}", file=err, flush=True)
                status = status or 124
                end_all(signal.SIGTERM)
                time.sleep(5.0)
                end_all(signal.SIGKILL)
                break
            if pending:
                time.sleep(0.05)
    finally:
        end_all(signal.SIGKILL)
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass
        reader.join(timeout=10)
        for line in relayed:
            (out if is_result is None or is_result(line) else err).write(line)
        out.flush()
        err.flush()
    return status
