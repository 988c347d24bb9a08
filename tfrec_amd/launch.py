"""Starting one process per GPU on a node without an external launcher.

The reference is a single process (run_model.py:12-23).  ``launch_ranks`` starts N child processes of a script with
the environment ``torch.distributed.run`` would give them (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT);
the parent never touches the GPU and replaces no process image, it only waits.  All children are watched: when one
exits with an error the others are ended, so that a rank that dies before the rendezvous does not leave the rest
waiting for it until the process-group timeout.
"""
import os
import socket
import subprocess
import sys
import time


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, script, argv, extra_env=None, poll_s=0.2):
    """Runs ``script argv`` as ``n`` ranks; rank 0's stdout is passed through.  Returns the exit code (the first
    non-zero one of any rank, else 0)."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, WORLD_SIZE=str(n), RANK=str(r), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    failed = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and failed == 0:
                failed = code
                sys.stderr.write("rank %d exited with code %d; ending the other ranks\n" % (r, code))
                for o in alive:
                    procs[o].terminate()
        if alive:
            time.sleep(poll_s)
    for p in procs:       # (terminated ranks: give them a moment, then insist)
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
    return failed


def init_process_group(backend="nccl", device=None, timeout_s=300):
    """Joins the process group described by the environment (one rank per GPU).  A short timeout: a rank that never
    arrives should end the run, not hold it for half an hour."""
    import datetime

    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    kw = {"timeout": datetime.timedelta(seconds=timeout_s)}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, **kw)
    return dist.get_rank(), dist.get_world_size()
