"""GPU tests of the device epoch pipeline (tfrec_sample_negatives, tfrec_shuffle_perm,
tfrec_gather_rows_i32, tfrec_epoch_triples) against the CPU oracle: sampled indices and permutations are bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from tfrec_amd.device import Device
    d = Device(0)
    yield d
    d.close()


def _rows(rng, U, I, lo, hi):
    rows = [np.sort(rng.choice(I, int(rng.integers(lo, hi + 1)), replace=False)).astype(np.int32) for _ in range(U)]
    indptr = np.zeros(U + 1, np.int64)
    indptr[1:] = np.cumsum([len(r) for r in rows])
    return indptr, np.concatenate(rows)


@pytest.mark.parametrize("U,I,lo,hi,n_neg", [(1000, 500, 0, 40, 1), (300, 64, 30, 63, 4), (5000, 20000, 1, 300, 1), (3, 7, 6, 6, 32)])
def test_sampled_negatives_equal_the_oracle(dev, oracle, U, I, lo, hi, n_neg):
    rng = np.random.default_rng(U + I)
    indptr, indices = _rows(rng, U, I, lo, hi)
    want = oracle.sample_negatives(indptr, indices, U, I, n_neg, 2018, 6)
    u, p, n = dev.sample_negatives(dev.array(indptr, np.int64), dev.array(indices, np.int32), U, I, n_neg, 2018, 6)
    dev.synchronize()
    assert np.array_equal(u.numpy(), want[0]) and np.array_equal(p.numpy(), want[1]) and np.array_equal(n.numpy(), want[2])


@pytest.mark.parametrize("U,I,lo,hi,first", [(1000, 500, 0, 40, 0), (5000, 20000, 1, 300, 12345), (40, 9, 0, 8, 7), (1, 5, 1, 1, 0)])
def test_epoch_triples_equal_the_oracle_and_the_three_calls(dev, oracle, U, I, lo, hi, first):
    """tfrec_epoch_triples (packed draws, one shuffle read per interaction) gives what the oracle's draws under the
    oracle's permutation give, and what tfrec_sample_negatives + tfrec_shuffle_perm + tfrec_gather_rows_i32 give."""
    rng = np.random.default_rng(3 * U + I)
    indptr, indices = _rows(rng, U, I, lo, hi)
    nnz = len(indices)
    d_ptr, d_idx = dev.array(indptr, np.int64), dev.array(indices, np.int32)
    u, p, n = dev.epoch_triples(d_ptr, d_idx, U, I, 31, 8, 9, first_interaction=first)
    dev.synchronize()
    wu, wp, wn, _ = oracle.sample_negatives(indptr, indices, U, I, 1, 31, 8, first=first)
    perm = oracle.shuffle_perm(nnz, 31, 9)
    assert np.array_equal(u.numpy(), wu[perm]) and np.array_equal(p.numpy(), wp[perm]) and np.array_equal(n.numpy(), wn[perm, 0])
    su, sp, sn = dev.sample_negatives(d_ptr, d_idx, U, I, 1, 31, 8, first_interaction=first)
    dperm = dev.shuffle_perm(nnz, 31, 9)
    for got, src in ((u, su), (p, sp), (n, sn)):
        assert np.array_equal(got.numpy().reshape(-1), dev.gather_rows_i32(src, dperm).numpy().reshape(-1))
    dev.synchronize()


def test_epoch_triples_of_nothing_and_of_a_full_row(dev):
    empty = dev.epoch_triples(dev.array(np.zeros(4, np.int64), np.int64), dev.empty(0, np.int32), 3, 5, 1, 0, 1)
    dev.synchronize()
    assert all(x.numpy().size == 0 for x in empty)
    # a user who holds every item: nothing can be drawn, reported at synchronize
    dev.epoch_triples(dev.array(np.array([0, 2, 6], np.int64), np.int64), dev.array(np.array([1, 3, 0, 1, 2, 3], np.int32), np.int32), 2, 4, 1, 0, 1)
    with pytest.raises(IndexError):
        dev.synchronize()
    dev.synchronize()


def test_ml100k_epoch_equals_the_oracle_and_full_rows_are_reported(dev, oracle, golden_ml100k):
    from test_gpu_models import ml100k
    tr = ml100k(golden_ml100k).train_matrix
    from tfrec_amd.data import DeviceEpochSampler
    s = DeviceEpochSampler(dev, tr, seed=2018)
    for e in (0, 3):
        u, p, n = s.epoch(e)
        dev.synchronize()
        wu, wp, wn, _ = oracle.sample_negatives(tr.indptr.astype(np.int64), tr.indices.astype(np.int32), 943, 1682, 1, 2018, 2 * e)
        perm = oracle.shuffle_perm(tr.nnz, 2018, 2 * e + 1)
        assert np.array_equal(u.numpy(), wu[perm]) and np.array_equal(p.numpy(), wp[perm]) and np.array_equal(n.numpy(), wn[perm])
    # a user who holds every item: nothing can be drawn, the call is reported at synchronize
    indptr = np.array([0, 2, 6], np.int64)
    indices = np.array([1, 3, 0, 1, 2, 3], np.int32)
    dev.sample_negatives(dev.array(indptr, np.int64), dev.array(indices, np.int32), 2, 4, 1, 1, 0)
    with pytest.raises(IndexError):
        dev.synchronize()
    dev.synchronize()


@pytest.mark.parametrize("n", [1, 2, 63, 4097, 300000, 3000000])
def test_shuffle_and_gather_equal_the_oracle(dev, oracle, n):
    perm = dev.shuffle_perm(n, 77, 5)
    dev.synchronize()
    want = oracle.shuffle_perm(n, 77, 5)
    assert np.array_equal(perm.numpy(), want)
    if n >= 3000000:
        # the library sorts by the high key word and orders equal high words afterwards: this size has hundreds of them
        hi = np.sort(oracle.shuffle_keys(n, 77, 5) >> np.uint64(32))
        assert int((hi[1:] == hi[:-1]).sum()) > 100
    src = np.arange(3 * n, dtype=np.int32).reshape(n, 3) * 7
    out = dev.gather_rows_i32(dev.array(src, np.int32), perm)
    dev.synchronize()
    assert np.array_equal(out.numpy(), src[want])


def test_bpr_with_the_device_sampler_follows_the_oracle_schedule(dev, oracle, golden_ml100k):
    """model BPR with ``sampler = device``: one epoch on ML-100k equals the oracle run over the
    oracle-drawn, oracle-shuffled epoch."""
    from test_gpu_models import ml100k, _Log
    from tfrec_amd.evaluator import FoldOutEvaluator
    from tfrec_amd.model import BPR
    ds = ml100k(golden_ml100k)
    ev = FoldOutEvaluator(ds.train_matrix, ds.test_matrix, top_k=50)
    cfg = {"factors_num": 64, "lr": 0.01, "reg": 0.05, "epochs": 1, "batch_size": 512, "sampler": "device", "seed": 5}
    BPR.logger = _Log()
    model = BPR(dev, cfg, ds, ev)
    P, Q, b = model.mf.parameters()
    model.train_model()
    tr = ds.train_matrix
    u, p, n, _ = oracle.sample_negatives(tr.indptr.astype(np.int64), tr.indices.astype(np.int32), 943, 1682, 1, 5, 0)
    perm = oracle.shuffle_perm(tr.nnz, 5, 1)
    u, p, n = u[perm], p[perm], n[perm].ravel()
    ws = oracle.Workspace(943, 1682, 64)
    for lo in range(0, len(u), 512):
        oracle.bpr_step(P, Q, b, u[lo:lo + 512], p[lo:lo + 512], n[lo:lo + 512], 0.01, 0.05, ws=ws)
    P1, Q1, b1 = model.mf.parameters()
    for a, c in ((P1, P), (Q1, Q), (b1, b)):
        np.testing.assert_allclose(a, c, rtol=2e-4, atol=2e-6)


def test_dnsbpr_with_the_device_sampler_follows_the_oracle_schedule(dev, oracle, tmp_path, monkeypatch):
    """DNS with ``sampler = device``: pools drawn on the GPU, arg-max under the current parameters,
    shuffle, steps -- two epochs equal the oracle run (selected negatives exact, parameters 2e-4)."""
    from test_gpu_models import synthetic, _Log
    from tfrec_amd.evaluator import FoldOutEvaluator
    from tfrec_amd.model import DNSBPR
    monkeypatch.chdir(tmp_path)
    ds = synthetic(4, U=150, I=120)
    U, I = ds.train_matrix.shape
    ev = FoldOutEvaluator(ds.train_matrix, ds.test_matrix, top_k=10)
    cfg = {"factors_num": 32, "lr": 0.05, "reg": 0.01, "epochs": 2, "batch_size": 64, "neg_num": 8, "sampler": "device", "seed": 9,
           "data_name": "synthetic"}
    DNSBPR.logger = _Log()
    model = DNSBPR(dev, cfg, ds, ev)
    P, Q, b = model.mf.parameters()
    model.train_model()
    tr = ds.train_matrix
    ws = oracle.Workspace(U, I, 32)
    for e in range(2):
        u, p, pools, _ = oracle.sample_negatives(tr.indptr.astype(np.int64), tr.indices.astype(np.int32), U, I, 8, 9, 2 * e)
        n, _ = oracle.dns_select(P, Q, b, u, pools)
        perm = oracle.shuffle_perm(tr.nnz, 9, 2 * e + 1)
        u, p, n = u[perm], p[perm], n[perm]
        for lo in range(0, len(u), 64):
            oracle.bpr_step(P, Q, b, u[lo:lo + 64], p[lo:lo + 64], n[lo:lo + 64], 0.05, 0.01, ws=ws)
    P1, Q1, b1 = model.mf.parameters()
    for a, c in ((P1, P), (Q1, Q), (b1, b)):
        np.testing.assert_allclose(a, c, rtol=2e-4, atol=2e-6)


def test_apr_with_the_device_sampler_follows_the_oracle_schedule(dev, oracle, tmp_path):
    """APR with ``sampler = device`` (model/APR.py:156-184 with the epoch drawn and shuffled on the GPU): two epochs
    equal the oracle's APR steps over the oracle-drawn, oracle-shuffled epochs."""
    from test_gpu_models import synthetic, _Log
    from tfrec_amd.evaluator import FoldOutEvaluator
    from tfrec_amd.model import APR
    from tfrec_amd.utils import save_params
    ds = synthetic(6, U=150, I=120)
    U, I = ds.train_matrix.shape
    rng = np.random.default_rng(4)
    P = rng.uniform(-0.3, 0.3, (U, 20)).astype(np.float32); Q = rng.uniform(-0.3, 0.3, (I, 20)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, I).astype(np.float32)
    pre = str(tmp_path / "pre.npz")
    save_params(pre, P, Q, b)
    ev = FoldOutEvaluator(ds.train_matrix, ds.test_matrix, top_k=10)
    cfg = {"factors_num": 20, "lr": 0.05, "reg": 0.01, "reg_adv": 1, "eps": 0.05, "epochs": 2, "batch_size": 128, "pretrain_file": pre,
           "sampler": "device", "seed": 17}
    APR.logger = _Log()
    model = APR(dev, cfg, ds, ev)
    model.train_model()
    tr = ds.train_matrix
    aP, aQ, ab = [np.full_like(x, 0.1) for x in (P, Q, b)]
    ws = oracle.Workspace(U, I, 20)
    for e in range(2):
        u, p, n, _ = oracle.sample_negatives(tr.indptr.astype(np.int64), tr.indices.astype(np.int32), U, I, 1, 17, 2 * e)
        perm = oracle.shuffle_perm(tr.nnz, 17, 2 * e + 1)
        u, p, n = u[perm], p[perm], n[perm].ravel()
        for lo in range(0, len(u), 128):
            oracle.apr_step(P, Q, b, aP, aQ, ab, u[lo:lo + 128], p[lo:lo + 128], n[lo:lo + 128], 0.05, 0.01, 1.0, 0.05, ws=ws)
    for got, want in zip(model.mf.parameters(), (P, Q, b)):
        np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-6)
