"""CPU tests (gloo, world size 2 and 3) of the multi-GPU exchange logic in tfrec_amd.sharded: the
row-sharded BPR step (item table sharded with all-to-all lookups, or replicated with one all-reduce
of gradient sums) and the item-sharded evaluation give the results of the unsharded path.
Local compute is a test-only CPU backend (tests/sharded_cpu_backend.py)."""
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

U, I, D, B, STEPS, K = 57, 41, 8, 200, 3, 10
MODES = ("sharded", "dense")


def _problem():
    rng = np.random.default_rng(123)
    batches = [(rng.integers(0, U, B).astype(np.int32), rng.integers(0, 6, B).astype(np.int32) * 3 % I,
                rng.integers(0, I, B).astype(np.int32)) for _ in range(STEPS)]
    tr_ptr = np.arange(0, 4 * (U + 1), 4, dtype=np.int64)
    tr_idx = np.concatenate([np.sort(rng.choice(I, 4, replace=False)) for _ in range(U)]).astype(np.int32)
    te_ptr = np.zeros(U + 1, np.int64)
    te_rows = []
    for u in range(U):
        row = np.sort(rng.choice(I, int(rng.integers(0, 3)), replace=False)).astype(np.int32)
        te_rows.append(row); te_ptr[u + 1] = te_ptr[u] + len(row)
    te_idx = np.concatenate(te_rows).astype(np.int32)
    test_users = np.array([u for u in range(U) if te_ptr[u + 1] > te_ptr[u]], np.int32)
    return batches, (tr_ptr, tr_idx), (te_ptr, te_idx), test_users


def _worker(rank, world, port, out_dir, item_mode):
    for p in (ROOT, HERE, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from sharded_cpu_backend import CpuBackend
    from tfrec_amd.sharded import ShardedBprTrainer, ShardedEvaluator
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    batches, train, test, test_users = _problem()
    cpu = torch.device("cpu")
    tr = ShardedBprTrainer(CpuBackend(), U, I, D, 0.05, 0.02, seed=5, device=cpu, item_mode=item_mode)
    assert tr.item_mode == item_mode
    lo, hi = tr.user_range
    for (u, i, j) in batches:
        m = (u >= lo) & (u < hi)              # every triple goes to the owner of its user
        tr.step(torch.from_numpy(u[m]), torch.from_numpy(i[m].astype(np.int32)), torch.from_numpy(j[m]))
    P, Q, b = tr.gather_full()
    ev = ShardedEvaluator(CpuBackend(), tr, tuple(torch.from_numpy(x) for x in train), tuple(torch.from_numpy(x) for x in test),
                          torch.from_numpy(test_users), top_k=K, mode=0, chunk_users=16)
    metrics = ev.evaluate()
    if rank == 0:
        np.savez(os.path.join(out_dir, "w%d_%s.npz" % (world, item_mode)), P=P.numpy(), Q=Q.numpy(), b=b.numpy(), m=metrics.numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.fixture(scope="module")
def runs(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("sharded"))
    res = {}
    for mode in MODES:
        _worker(0, 1, 0, out, mode)
        for world in (2, 3):
            mp.spawn(_worker, args=(world, _free_port(), out, mode), nprocs=world, join=True)
        for w in (1, 2, 3):
            res[(w, mode)] = np.load(os.path.join(out, "w%d_%s.npz" % (w, mode)))
    return res


def test_sharded_training_equals_unsharded(runs, oracle):
    batches, _, _, _ = _problem()
    P = np.random.default_rng(11).uniform(-0.05, 0.05, (U, D)).astype(np.float32)
    Q = np.random.default_rng(12).uniform(-0.05, 0.05, (I, D)).astype(np.float32)
    b = np.zeros(I, np.float32)
    for (u, i, j) in batches:
        oracle.bpr_step(P, Q, b, u, i.astype(np.int32), j, 0.05, 0.02)
    for key in [(w, m) for w in (1, 2, 3) for m in MODES]:
        np.testing.assert_allclose(runs[key]["P"], P, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(runs[key]["Q"], Q, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(runs[key]["b"], b, rtol=1e-5, atol=1e-7)


def test_sharded_evaluation_equals_unsharded(runs, oracle):
    _, train, test, test_users = _problem()
    for m in MODES:
        r = runs[(1, m)]
        mean, _, _ = oracle.eval_fullrank(r["P"], r["Q"], r["b"], test_users, train, test, K, score_mode=1)
        for w in (1, 2, 3):
            np.testing.assert_allclose(runs[(w, m)]["m"], mean, rtol=1e-6, atol=1e-7)
        # item sharding is invisible in the result: same numbers for every world size
        assert np.allclose(runs[(2, m)]["m"], runs[(3, m)]["m"], rtol=1e-6)
