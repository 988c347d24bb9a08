"""CPU test (gloo, world 1/2/3) of tfrec_amd.sharded.ShardedEpochSampler: device-style epochs drawn per
rank for the users it owns -- BPR (one negative) and DNS (pools + best-scored candidate at epoch start)
-- give the triples the whole matrix would give, and collective epochs of steps give the parameters of
one process stepping over the union of the ranks' batches."""
import os
import socket
import sys

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
U, I, D, BATCH, NEG = 61, 47, 8, 40, 5


def _matrix():
    rng = np.random.default_rng(21)
    rows, cols = [], []
    for u in range(U - 1):
        c = rng.permutation(I)[: int(rng.integers(0, 9))]
        rows += [u] * len(c); cols += c.tolist()
    rows.append(U - 1); cols.append(0)        # the last user's only item is item 0: the reference does not train it (utils/tools.py:52-62)
    m = sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(U, I)); m.sort_indices()
    return m


def _worker(rank, world, port, out_dir, item_mode):
    for p in (ROOT, HERE, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from sharded_cpu_backend import CpuBackend
    from tfrec_amd.sharded import ShardedBprTrainer, ShardedEpochSampler
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    tr = ShardedBprTrainer(CpuBackend(), U, I, D, 0.05, 0.02, seed=5, device=torch.device("cpu"), item_mode=item_mode)
    m = _matrix()
    out = {}
    for n_neg, tag in ((1, "bpr"), (NEG, "dns")):
        s = ShardedEpochSampler(CpuBackend(), tr, m, seed=13, n_neg=n_neg)
        u, p, n = s.draw(0)
        out[tag + "_draw"] = np.stack([u.numpy() + tr.user_range[0], p.numpy(), n.numpy()]) if s.nnz else np.zeros((3, 0), np.int32)
        eu, ep, en = s.epoch(1)
        out[tag + "_epoch1"] = np.stack([eu.numpy(), ep.numpy(), en.numpy()]) if s.nnz else np.zeros((3, 0), np.int32)
    # two collective epochs with the DNS sampler
    s = ShardedEpochSampler(CpuBackend(), tr, m, seed=13, n_neg=NEG)
    lists = []
    for e in range(2):
        eu, ep, en = s.epoch(e)          # what train_epoch will draw (pure function of the epoch and the model at its start)
        lists.append(np.stack([eu.numpy(), ep.numpy(), en.numpy()]) if s.nnz else np.zeros((3, 0), np.int32))
        s.train_epoch(e, BATCH)
    P, Q, b = tr.gather_full()
    np.savez(os.path.join(out_dir, "w%d_%s_r%d.npz" % (world, item_mode, rank)), P=P.numpy(), Q=Q.numpy(), b=b.numpy(),
             l0=lists[0], l1=lists[1], **out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.fixture(scope="module")
def runs(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("epochs"))
    res = {}
    for mode in ("dense", "sharded"):
        _worker(0, 1, 0, out, mode)
        for world in (2, 3):
            mp.spawn(_worker, args=(world, _free_port(), out, mode), nprocs=world, join=True)
        for w in (1, 2, 3):
            res[(w, mode)] = [np.load(os.path.join(out, "w%d_%s_r%d.npz" % (w, mode, r))) for r in range(w)]
    return res


def test_shard_draws_equal_the_whole_matrix_draws(runs, oracle):
    sys.path.insert(0, ROOT)
    from tfrec_amd.utils.tools import csr_trained_rows
    m = csr_trained_rows(_matrix())
    assert m.nnz == _matrix().nnz - 1
    P = np.random.default_rng(11).uniform(-0.05, 0.05, (U, D)).astype(np.float32)
    Q = np.random.default_rng(12).uniform(-0.05, 0.05, (I, D)).astype(np.float32)
    b = np.zeros(I, np.float32)
    u, p, n1, _ = oracle.sample_negatives(m.indptr.astype(np.int64), m.indices.astype(np.int32), U, I, 1, 13, 0)
    _, _, pools, _ = oracle.sample_negatives(m.indptr.astype(np.int64), m.indices.astype(np.int32), U, I, NEG, 13, 0)
    picked, _ = oracle.dns_select(P, Q, b, u, pools)
    for key, rs in runs.items():
        got = np.concatenate([r["bpr_draw"] for r in rs], axis=1)
        assert np.array_equal(got, np.stack([u, p, n1.ravel()])), key
        got = np.concatenate([r["dns_draw"] for r in rs], axis=1)
        assert np.array_equal(got, np.stack([u, p, picked])), key
        # a rank's shuffled epoch is a permutation of its own draw for that epoch
        for r in rs:
            e = r["bpr_epoch1"]
            assert e.shape[1] == r["bpr_draw"].shape[1]
            assert np.array_equal(np.sort(e[1] * U + e[0]), np.sort(r["bpr_draw"][1] * U + r["bpr_draw"][0]))


def test_collective_epochs_equal_one_process_over_the_union_of_batches(runs, oracle):
    for (w, mode), rs in runs.items():
        P = np.random.default_rng(11).uniform(-0.05, 0.05, (U, D)).astype(np.float32)
        Q = np.random.default_rng(12).uniform(-0.05, 0.05, (I, D)).astype(np.float32)
        b = np.zeros(I, np.float32)
        for e in range(2):
            lists = [r["l%d" % e] for r in rs]
            steps = max((l.shape[1] + BATCH - 1) // BATCH for l in lists)
            for s in range(steps):
                part = [l[:, s * BATCH:(s + 1) * BATCH] for l in lists]
                uu, ii, jj = (np.concatenate([x[c] for x in part]).astype(np.int32) for c in range(3))
                if len(uu):
                    oracle.bpr_step(P, Q, b, uu, ii, jj, 0.05, 0.02)
        np.testing.assert_allclose(rs[0]["P"], P, rtol=1e-5, atol=1e-7, err_msg=str((w, mode)))
        np.testing.assert_allclose(rs[0]["Q"], Q, rtol=1e-5, atol=1e-7, err_msg=str((w, mode)))
        np.testing.assert_allclose(rs[0]["b"], b, rtol=1e-5, atol=1e-7, err_msg=str((w, mode)))
