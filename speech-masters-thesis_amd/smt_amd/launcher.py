"""One process per GPU on one node: start N rank processes of a script (reference train.py:389-458,568 does this with
``mp.spawn`` inside a process that may already hold a GPU context; here the parent never touches the GPU and never
exec()s -- the children are fresh interpreters with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment,
exactly what ``python -m torch.distributed.run`` would hand them).

Stdlib only; the parent may import torch (``torch.cuda.device_count()`` does not initialise the GPU on this image)
but makes no HIP call.
"""
import os
import socket
import subprocess
import sys
import time


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this host driver
    return env


def under_launcher(environ=None):
    e = os.environ if environ is None else environ
    return "RANK" in e and "WORLD_SIZE" in e


def spawn_ranks(world, cmd, base_env=None, poll_s=0.2, grace_s=10.0, port=None):
    """Run ``cmd`` (argv list) once per rank; rank 0 inherits stdout (it prints the result line), every rank inherits
    stderr.  Returns 0 when all ranks exit 0; when one rank fails the others are terminated (by PID) and its exit
    code is returned."""
    port = free_port() if port is None else port
    procs = [subprocess.Popen(cmd, env=rank_env(r, world, port, base_env),
                              stdout=None if r == 0 else subprocess.DEVNULL) for r in range(world)]
    failed = 0
    try:
        live = set(range(world))
        while live:
            for r in sorted(live):
                rc = procs[r].poll()
                if rc is None:
                    continue
                live.discard(r)
                if rc != 0 and not failed:
                    failed = rc
                    print(f"[launcher] rank {r} exited with code {rc}; stopping the other ranks", file=sys.stderr)
            if failed:
                break
            time.sleep(poll_s)
    finally:
        deadline = time.time() + grace_s
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return failed
