"""CPU tests (gloo, world size 1, 2 and 3) of the multi-GPU APR step in tfrec_amd.sharded: user-sharded
P with a replicated item table, two all-reduces per step.  The result has to be that of the unsharded
step (model/APR.py:116-136, 156-164) on the union of the ranks' batches.  Local compute is a
test-only CPU backend (tests/sharded_cpu_backend.py)."""
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

U, I, D, B, STEPS = 45, 37, 8, 160, 3
LR, REG, REG_ADV, EPS = 0.05, 0.01, 1.0, 0.05


def _problem():
    rng = np.random.default_rng(77)
    P = rng.uniform(-0.3, 0.3, (U, D)).astype(np.float32)
    Q = rng.uniform(-0.3, 0.3, (I, D)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, I).astype(np.float32)
    # users 40.. never appear: the last rank of a 3-way split may get an empty batch
    batches = [(rng.integers(0, 40, B).astype(np.int32), (rng.integers(0, 7, B) * 5 % I).astype(np.int32),
                rng.integers(0, I, B).astype(np.int32)) for _ in range(STEPS)]
    return (P, Q, b), batches


def _worker(rank, world, port, out_dir):
    for p in (ROOT, HERE, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from sharded_cpu_backend import CpuBackend
    from tfrec_amd.sharded import ShardedAprTrainer
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    params, batches = _problem()
    tr = ShardedAprTrainer(CpuBackend(), U, I, D, LR, REG, REG_ADV, EPS, params=params, device=torch.device("cpu"))
    lo, hi = tr.user_range
    losses = []
    for n, (u, i, j) in enumerate(batches):
        m = (u >= lo) & (u < hi)
        loss = torch.zeros(3, dtype=torch.float64)
        # the global batch size is given on even steps and counted by the trainer on odd ones
        tr.step(torch.from_numpy(u[m]), torch.from_numpy(i[m]), torch.from_numpy(j[m]), global_batch=B if n % 2 == 0 else None, loss=loss)
        losses.append(loss.numpy().copy())
    P, Q, b = tr.gather_full()
    if rank == 0:
        np.savez(os.path.join(out_dir, "w%d.npz" % world), P=P.numpy(), Q=Q.numpy(), b=b.numpy(), loss=np.array(losses),
                 accQ=tr._accQb.numpy(), dQ=tr.dQ.numpy(), db=tr.db.numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.fixture(scope="module")
def runs(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("sharded_apr"))
    _worker(0, 1, 0, out)
    for world in (2, 3):
        mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    return {w: np.load(os.path.join(out, "w%d.npz" % w)) for w in (1, 2, 3)}


def test_sharded_apr_equals_unsharded(runs, oracle):
    (P, Q, b), batches = _problem()
    accP, accQ, accb = (np.full_like(x, 0.1) for x in (P, Q, b))
    dP, dQ, db = (np.zeros_like(x) for x in (P, Q, b))
    losses = []
    for (u, i, j) in batches:
        losses.append(oracle.apr_step(P, Q, b, accP, accQ, accb, u, i, j, LR, REG, REG_ADV, EPS, deltas=(dP, dQ, db)))
    touched = np.zeros(I, bool)
    touched[batches[-1][1]] = True
    touched[batches[-1][2]] = True
    for w in (1, 2, 3):
        r = runs[w]
        np.testing.assert_allclose(r["P"], P, rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(r["Q"], Q, rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(r["b"], b, rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(r["accQ"][: I * D].reshape(I, D), accQ, rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(r["accQ"][I * D:], accb, rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(r["loss"], np.array(losses), rtol=1e-5)
        # the deltas of the last step: equal on the touched rows, zero elsewhere
        np.testing.assert_allclose(r["dQ"][touched], dQ[touched], rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(r["db"][touched], db[touched], rtol=1e-4, atol=1e-7)
        assert not r["dQ"][~touched].any() and not r["db"][~touched].any()
