"""CPU test (gloo, world 1/2/3) of the model plug-ins started as one process per GPU: run_model.py-shaped code
(ModelFactory-style config -> Dataset-like object -> FoldOutEvaluator.from_config -> Model(sess, config, dataset,
evaluator).train_model()) for BPR, DNSBPR and APR.  The local compute is the test-only CPU stand-in
(tests/sharded_cpu_backend.py) passed where a Device goes.  Checked: the global batches do not depend on the number
of ranks (they are the batches one process forms), so parameters and logged metrics agree across world sizes; rank 0
alone logs, in the reference's line format."""
import os
import socket
import sys

import numpy as np
import pytest
import scipy.sparse as sp
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
U, I, D = 61, 47, 8


def _matrices():
    rng = np.random.default_rng(21)
    rows, cols, trows, tcols = [], [], [], []
    for u in range(U):
        c = rng.permutation(I)[: int(rng.integers(2, 12))]
        k = max(1, len(c) // 4)
        rows += [u] * (len(c) - k); cols += c[k:].tolist()
        if u % 3:
            trows += [u] * k; tcols += c[:k].tolist()
    tr = sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(U, I)); tr.sort_indices()
    te = sp.csr_matrix((np.ones(len(trows)), (trows, tcols)), shape=(U, I)); te.sort_indices()
    return tr, te


class _Data:
    def __init__(self):
        self.train_matrix, self.test_matrix = _matrices()
        self.num_users, self.num_items = U, I


class _Log:
    def __init__(self):
        self.lines = []

    def info(self, msg):
        self.lines.append(str(msg))


CONFIGS = {
    "BPR": {"factors_num": D, "lr": 0.05, "reg": 0.02, "epochs": 2, "batch_size": 64, "top_k": 7, "metric_sum": "float64"},
    "DNSBPR": {"factors_num": D, "lr": 0.05, "reg": 0.02, "epochs": 2, "batch_size": 64, "neg_num": 4, "top_k": 7, "data_name": "t",
               "item_mode": "sharded"},
    "APR": {"factors_num": D, "lr": 0.05, "reg": 0.01, "reg_adv": 1.0, "eps": 0.05, "epochs": 2, "batch_size": 50, "top_k": 7},
}


def _worker(rank, world, port, out_dir):
    for p in (ROOT, HERE, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.chdir(out_dir)                        # DNSBPR saves its parameters into the working directory
    from sharded_cpu_backend import CpuBackend
    from tfrec_amd.evaluator import FoldOutEvaluator
    from tfrec_amd.model import APR, BPR, DNSBPR
    from tfrec_amd.utils import save_params
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    data = _Data()
    rng = np.random.default_rng(3)
    pre = os.path.join(out_dir, "pre_w%d_r%d.npz" % (world, rank))
    save_params(pre, rng.uniform(-0.3, 0.3, (U, D)).astype(np.float32), rng.uniform(-0.3, 0.3, (I, D)).astype(np.float32),
                rng.uniform(-0.1, 0.1, I).astype(np.float32))
    out = {}
    for name, Model in (("BPR", BPR), ("DNSBPR", DNSBPR), ("APR", APR)):
        cfg = dict(CONFIGS[name], force_sharded=True)
        if name == "APR":
            cfg["pretrain_file"] = pre
        log = _Log()
        Model.logger = log
        ev = FoldOutEvaluator.from_config(cfg, data.train_matrix, data.test_matrix)
        model = Model(CpuBackend(), cfg, data, ev)
        assert model.sharded and model.world == world
        model.train_model()
        P, Q, b = model.parameters()
        out[name + "_P"], out[name + "_Q"], out[name + "_b"] = P, Q, b
        if rank == 0:
            # the reference's log: the metric header, then one line per epoch with 5 * top_k tab-separated numbers
            assert log.lines[0] == ev.metrics_info() and "Precision@1:1:7" in log.lines[0]
            epoch_lines = [ln for ln in log.lines[1:] if not ln.startswith("pretrain")]
            assert len(epoch_lines) == 2 and epoch_lines[0].startswith("epoch 0:\t" if name != "APR" else "0:\t")
            out[name + "_metrics"] = np.array([float(x) for x in epoch_lines[-1].split("\t")[1:]])
            assert out[name + "_metrics"].shape == (35,)
        else:
            assert model.logger is not log and log.lines == []        # the other ranks stay quiet
    if rank == 0:
        np.savez(os.path.join(out_dir, "models_w%d.npz" % world), **out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.fixture(scope="module")
def runs(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("models_sharded"))
    cwd = os.getcwd()
    try:
        _worker(0, 1, 0, out)
    finally:
        os.chdir(cwd)
    for world in (2, 3):
        mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    return {w: np.load(os.path.join(out, "models_w%d.npz" % w)) for w in (1, 2, 3)}


@pytest.mark.parametrize("name", ["BPR", "DNSBPR", "APR"])
def test_models_give_the_same_run_on_any_number_of_ranks(runs, name):
    one = runs[1]
    for w in (2, 3):
        for k in ("P", "Q", "b"):
            np.testing.assert_allclose(runs[w]["%s_%s" % (name, k)], one["%s_%s" % (name, k)], rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(runs[w][name + "_metrics"], one[name + "_metrics"], rtol=0, atol=0.02)
    m = one[name + "_metrics"]
    assert np.all(np.isfinite(m)) and m.max() <= 1.0 and m.min() >= 0.0 and m[:7].max() > 0.0


def test_global_batches_are_the_single_process_batches(oracle):
    """epoch_global on one rank of a world of one: batch k is positions [k*B, (k+1)*B) of the permutation a single
    process draws, applied to the triples in storage order -- what DeviceEpochSampler.epoch + slicing gives."""
    import torch
    sys.path.insert(0, HERE)
    from sharded_cpu_backend import CpuBackend
    from tfrec_amd.sharded import ShardedBprTrainer, ShardedEpochSampler
    from tfrec_amd.utils.tools import csr_trained_rows
    tr, _ = _matrices()
    t = ShardedBprTrainer(CpuBackend(), U, I, D, 0.05, 0.02, seed=5, device=torch.device("cpu"), item_mode="dense")
    s = ShardedEpochSampler(CpuBackend(), t, tr, seed=13, n_neg=1, shuffle="global")
    B = 32
    u, p, n, bounds = s.epoch_global(3, B)
    csr = csr_trained_rows(tr)
    su, sp_, sn, _ = oracle.sample_negatives(csr.indptr.astype(np.int64), csr.indices.astype(np.int32), U, I, 1, 13, 6, 0)
    perm = oracle.shuffle_perm(csr.nnz, 13, 7)
    assert bounds == list(range(0, csr.nnz, B)) + [csr.nnz]
    assert np.array_equal(u.numpy(), su[perm]) and np.array_equal(p.numpy(), sp_[perm]) and np.array_equal(n.numpy(), sn.reshape(-1)[perm])
