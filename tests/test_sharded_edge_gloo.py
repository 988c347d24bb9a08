"""CPU tests (gloo) of the row-sharded BPR step on degenerate shapes: fewer users or items than ranks (ranks that own
nothing), ranks whose part of a batch is empty, a batch of one triple.  Several ranks must give what one rank gives.
Local compute is the test-only CPU backend (tests/sharded_cpu_backend.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _batches(U, I, B, steps, only_user):
    rng = np.random.default_rng(77)
    out = []
    for _ in range(steps):
        u = rng.integers(0, U, B).astype(np.int32)
        if only_user is not None:
            u[:] = only_user                  # every triple belongs to one rank: the others step with empty batches
        out.append((u, rng.integers(0, I, B).astype(np.int32), rng.integers(0, I, B).astype(np.int32)))
    return out


def _worker(rank, world, port, out_dir, mode, U, I, D, B, steps, only_user):
    for p in (ROOT, HERE, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from sharded_cpu_backend import CpuBackend
    from tfrec_amd.sharded import ShardedBprTrainer
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    tr = ShardedBprTrainer(CpuBackend(), U, I, D, 0.05, 0.02, seed=5, device=torch.device("cpu"), item_mode=mode)
    lo, hi = tr.user_range
    for (u, i, j) in _batches(U, I, B, steps, only_user):
        m = (u >= lo) & (u < hi)
        tr.step(torch.from_numpy(u[m]), torch.from_numpy(i[m]), torch.from_numpy(j[m]))
    P, Q, b = tr.gather_full()
    if rank == 0:
        np.savez(os.path.join(out_dir, "w%d.npz" % world), P=P.numpy(), Q=Q.numpy(), b=b.numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("mode", ["sharded", "dense"])
@pytest.mark.parametrize("U,I,B,steps,world,only_user", [
    (2, 3, 7, 2, 3, None),        # fewer users than ranks: one rank owns no user
    (5, 2, 9, 2, 3, None),        # fewer items than ranks
    (1, 1, 5, 2, 2, None),        # one user, one item
    (9, 7, 1, 3, 3, None),        # batches of one triple: two ranks step with nothing
    (9, 7, 6, 2, 3, 8),           # every triple on the last rank
])
def test_degenerate_shards_give_the_one_rank_result(tmp_path, mode, U, I, B, steps, world, only_user):
    out = str(tmp_path)
    _worker(0, 1, 0, out, mode, U, I, 4, B, steps, only_user)
    mp.spawn(_worker, args=(world, _free_port(), out, mode, U, I, 4, B, steps, only_user), nprocs=world, join=True)
    one, many = np.load(os.path.join(out, "w1.npz")), np.load(os.path.join(out, "w%d.npz" % world))
    for k in "PQb":
        np.testing.assert_allclose(many[k], one[k], rtol=1e-5, atol=1e-7)
