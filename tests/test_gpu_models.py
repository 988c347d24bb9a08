"""GPU tests of the model / evaluator plug-in surface (the reference's API for this path):
BPR, DNSBPR and APR recommenders trained through the HIP library against the same schedule
driven through the CPU oracle, the dense evaluator drop-in, and run_model.py end to end."""
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import ROOT, csr_from_pairs

pytestmark = pytest.mark.gpu


class _Log:
    def __init__(self):
        self.lines = []

    def info(self, msg):
        self.lines.append(msg)


class _DS:
    def __init__(self, train, test):
        self.train_matrix, self.test_matrix = train, test


@pytest.fixture(scope="module")
def dev():
    from tfrec_amd.device import Device
    d = Device(0)
    yield d
    d.close()


def ml100k(golden_ml100k):
    g = golden_ml100k
    U, I = g["P"].shape[0], g["Q"].shape[0]
    return _DS(csr_from_pairs(g["train_u"], g["train_i"], U, I), csr_from_pairs(g["test_u"], g["test_i"], U, I))


def synthetic(seed, U=120, I=90):
    rng = np.random.default_rng(seed)
    rows, cols, trows, tcols = [], [], [], []
    for u in range(U):
        c = rng.permutation(I)[: int(rng.integers(3, 14))]
        rows += [u] * (len(c) - 1); cols += c[:-1].tolist(); trows.append(u); tcols.append(int(c[-1]))
    tr = sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(U, I)); tr.sort_indices()
    te = sp.csr_matrix((np.ones(len(trows)), (trows, tcols)), shape=(U, I)); te.sort_indices()
    return _DS(tr, te)


def test_bpr_ml100k_two_epochs_match_the_oracle_schedule(dev, oracle, golden_ml100k):
    """conf/BPR.ini on ML-100k with d=64: same seed, same sampler, same batch order; parameters
    after two epochs (314 steps) and the metrics agree with the oracle-driven run."""
    from tfrec_amd.evaluator import FoldOutEvaluator
    from tfrec_amd.model import BPR
    ds = ml100k(golden_ml100k)
    ev = FoldOutEvaluator(ds.train_matrix, ds.test_matrix, top_k=50)
    cfg = {"factors_num": 64, "lr": 0.01, "reg": 0.05, "epochs": 2, "batch_size": 512}
    BPR.logger = _Log()
    np.random.seed(2018)
    model = BPR(dev, cfg, ds, ev)
    P, Q, b = model.mf.parameters()
    assert P.shape == (943, 64) and abs(P).max() <= 0.05 and not b.any()
    model.train_model()
    lines = BPR.logger.lines
    assert lines[0].startswith("Precision@1:1:50") and len(lines) == 3
    got = np.array([float(x) for x in lines[-1].split("\t")[1:]], dtype=np.float32)
    assert got.shape == (250,)

    rs = np.random.RandomState(2018)
    user_pos = oracle.csr_to_user_dict(ds.train_matrix)
    ws = oracle.Workspace(943, 1682, 64)
    for ep in range(2):
        u, i, j = oracle.bpr_epoch_triples(rs, user_pos, 1682)
        for idx in oracle.batch_order(rs, len(u), 512):
            oracle.bpr_step(P, Q, b, u[idx], i[idx], j[idx], 0.01, 0.05, ws=ws)
    P1, Q1, b1 = model.mf.parameters()
    for a, c in ((P1, P), (Q1, Q), (b1, b)):
        np.testing.assert_allclose(a, c, rtol=2e-4, atol=2e-6)
    users = np.array(sorted(ev.user_pos_test.keys()), np.int32)
    mean, _, _ = oracle.eval_fullrank(P1, Q1, b1, users, (ds.train_matrix.indptr, ds.train_matrix.indices),
                                      (ds.test_matrix.indptr, ds.test_matrix.indices), 50, score_mode=1)
    np.testing.assert_allclose(got.reshape(5, 50), mean, rtol=1e-6, atol=1e-7)
    # the dense route (_predict + eval_rating_matrix) gives the same numbers up to score rounding
    from tfrec_amd.evaluator.backend import eval_rating_matrix
    R = model._predict().astype(np.float32)
    keep = R.copy()
    dense = eval_rating_matrix(R, ev.user_pos_train, ev.user_pos_test, top_k=50)
    assert np.array_equal(R, keep)                     # the caller's matrix is left alone
    np.testing.assert_allclose(dense, mean, rtol=2e-3, atol=2e-3)


def test_dense_evaluator_drop_in_matches_reference_vectors(golden_eval):
    from tfrec_amd.evaluator.backend import eval_rating_matrix
    g = {k.split("/", 1)[1]: v for k, v in golden_eval.items() if k.startswith("subset/")}
    R = (g["P"] @ g["Q"].T + g["b"]).astype(np.float32)
    tr = {u: g["train_indices"][g["train_indptr"][u]:g["train_indptr"][u + 1]] for u in range(len(g["P"]))
          if g["train_indptr"][u + 1] > g["train_indptr"][u]}
    te = {int(u): g["test_indices"][g["test_indptr"][u]:g["test_indptr"][u + 1]] for u in g["test_users"]}
    for k in g["k_list"]:
        out = eval_rating_matrix(R, tr, te, top_k=int(k))
        assert out.dtype == np.float32 and out.shape == (5, int(k))
        assert np.array_equal(out, g["metrics_k%d" % k])
    with pytest.raises(TypeError):
        eval_rating_matrix(R.tolist(), tr, te)


def test_dnsbpr_epoch(dev, oracle):
    from tfrec_amd.evaluator import FoldOutEvaluator
    from tfrec_amd.model import DNSBPR
    ds = synthetic(1)
    ev = FoldOutEvaluator(ds.train_matrix, ds.test_matrix, top_k=10)
    cfg = {"factors_num": 20, "lr": 0.01, "reg": 0.05, "epochs": 1, "batch_size": 64, "neg_num": 15, "data_name": "unit"}
    DNSBPR.logger = _Log()
    np.random.seed(7)
    model = DNSBPR(dev, cfg, ds, ev)
    P, Q, b = model.mf.parameters()
    np.random.seed(7)
    it = model.get_training_data()
    users, pos, neg = it.columns
    # the oracle draws the same pools and picks the same negatives wherever the scores are not tied
    rs = np.random.RandomState(7)
    U, I = ds.train_matrix.shape
    pools = []
    for u, p in oracle.csr_to_user_dict(ds.train_matrix).items():
        pools.append(oracle.choice_excluding(rs, I, 15 * len(p), p).reshape(len(p), 15))
    pools = np.concatenate(pools).astype(np.int32)
    ref, margin = oracle.dns_select(P, Q, b, users, pools)
    ok = margin > 1e-6
    assert ok.mean() > 0.95 and np.array_equal(neg[ok], ref[ok])
    assert all(neg[t] in pools[t] for t in range(len(neg)))
    for u, p, n in zip(users, pos, neg):
        row = ds.train_matrix.indices[ds.train_matrix.indptr[u]:ds.train_matrix.indptr[u + 1]]
        assert n not in row


def test_apr_epoch_matches_the_oracle_schedule(dev, oracle, tmp_path):
    from tfrec_amd.evaluator import FoldOutEvaluator
    from tfrec_amd.model import APR
    from tfrec_amd.utils import save_params
    ds = synthetic(2)
    U, I = ds.train_matrix.shape
    rng = np.random.default_rng(3)
    P = rng.uniform(-0.3, 0.3, (U, 20)).astype(np.float32); Q = rng.uniform(-0.3, 0.3, (I, 20)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, I).astype(np.float32)
    pre = str(tmp_path / "pre.npz")
    save_params(pre, P, Q, b)
    ev = FoldOutEvaluator(ds.train_matrix, ds.test_matrix, top_k=10)
    cfg = {"factors_num": 20, "lr": 0.05, "reg": 0.0, "reg_adv": 1, "eps": 0.05, "epochs": 1, "batch_size": 128, "pretrain_file": pre}
    APR.logger = _Log()
    np.random.seed(11)
    model = APR(dev, cfg, ds, ev)
    model.train_model()
    assert APR.logger.lines[1].startswith("pretrain:") and len(APR.logger.lines) == 3
    rs = np.random.RandomState(11)
    aP, aQ, ab = [np.full_like(x, 0.1) for x in (P, Q, b)]
    ws = oracle.Workspace(U, I, 20)
    u, i, j = oracle.bpr_epoch_triples(rs, oracle.csr_to_user_dict(ds.train_matrix), I)
    for idx in oracle.batch_order(rs, len(u), 128):
        oracle.apr_step(P, Q, b, aP, aQ, ab, u[idx], i[idx], j[idx], 0.05, 0.0, 1.0, 0.05, ws=ws)
    for got, want in zip(model.mf.parameters(), (P, Q, b)):
        np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("extra", ["", "sampler = device\n"])
def test_run_model_end_to_end(tmp_path, extra):
    """run_model.py with the shipped ini layout: one epoch of BPR on ML-100k, one log line of 250 numbers
    (also with the epoch drawn on the GPU: ``sampler = device`` in the model's ini)."""
    os.makedirs(tmp_path / "conf")
    (tmp_path / "conf" / "BPR.ini").write_text("[hyperparameters]\nlr = 0.01\nreg = 0.05\nfactors_num = 20\nbatch_size = 512\nepochs = 1\n" + extra)
    ini = open(os.path.join(ROOT, "tfrec.ini")).read().replace("./dataset/", os.path.join(ROOT, "dataset") + "/")
    (tmp_path / "tfrec.ini").write_text(ini)
    if not os.path.exists(os.path.join(ROOT, "dataset", "movielens-100k-train.txt")):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "export_ml100k.py")])
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "run_model.py")], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if "epoch 0:" in l][-1]
    vals = [float(x) for x in line.split("epoch 0:\t")[1].split("\t")]
    assert len(vals) == 250 and 0.0 < vals[0] <= 1.0
    logs = os.listdir(tmp_path / "log" / "ml100k" / "BPR")
    assert len(logs) == 1 and logs[0].startswith("BPR_log_lr=0.01_reg=0.05_factors_num=20_batch_size=512_epochs=1_")
    assert ("sampler=device" in logs[0]) == bool(extra)


def test_pop_goes_through_the_dense_evaluator(dev, oracle, golden_ml100k):
    """Pop has no factors: FoldOutEvaluator takes its _predict() matrix through eval_rating_matrix on the
    GPU (model/Pop.py:21-24, evaluator/evaluator.py:29-32).  Item counts tie often; ties rank by ascending
    item id here and in the oracle (the reference's order among equal scores is unspecified)."""
    from tfrec_amd.evaluator import FoldOutEvaluator
    from tfrec_amd.model import Pop
    ds = ml100k(golden_ml100k)
    ev = FoldOutEvaluator(ds.train_matrix, ds.test_matrix, top_k=50)
    Pop.logger = _Log()
    model = Pop(None, {}, ds, ev)
    model.train_model()
    counts = np.asarray((ds.train_matrix != 0).sum(axis=0)).ravel().astype(np.float32)
    assert np.array_equal(model.ranking_score, counts)
    R = model._predict()
    assert R.shape == ds.train_matrix.shape and R.flags["C_CONTIGUOUS"] and R.dtype == np.float32
    got = np.array([float(x) for x in Pop.logger.lines[-1].split("result:\t")[1].split("\t")], np.float32)
    want, _ = oracle.eval_dense(R, oracle.csr_to_user_dict(ds.train_matrix), oracle.csr_to_user_dict(ds.test_matrix), 50)
    np.testing.assert_allclose(got, np.asarray(want).reshape(-1), rtol=1e-5, atol=1e-7)
    assert got[0] > 0.05          # popularity is a real baseline on ML-100k


@pytest.mark.parametrize("sampler", ["host", "device"])
def test_resume_from_a_checkpoint_continues_the_same_run(dev, tmp_path, sampler):
    """Two epochs in one go against one epoch, a fresh process-like restart from the checkpoint, and the
    second epoch: the same parameters bit for bit (host sampler: numpy's generator state travels in the
    checkpoint; device sampler: draws are a function of seed, epoch and position)."""
    from tfrec_amd.evaluator import FoldOutEvaluator
    from tfrec_amd.model import APR, BPR
    ds = synthetic(21)
    ev = FoldOutEvaluator(ds.train_matrix, ds.test_matrix, top_k=10)
    base = {"factors_num": 16, "lr": 0.05, "reg": 0.01, "epochs": 2, "batch_size": 64, "sampler": sampler}
    BPR.logger = _Log()
    np.random.seed(5)
    whole = BPR(dev, dict(base), ds, ev)
    whole.train_model()
    ck = str(tmp_path / ("bpr_%s.npz" % sampler))
    np.random.seed(5)
    first = BPR(dev, dict(base, epochs=1, checkpoint_file=ck), ds, ev)
    first.train_model()
    np.random.seed(12345)                                  # a restart does not inherit the generator
    second = BPR(dev, dict(base, checkpoint_file=ck, resume=True), ds, ev)
    second.train_model()
    assert any("resumed from" in l for l in BPR.logger.lines)
    for a, c in zip(second.mf.parameters(), whole.mf.parameters()):
        assert np.array_equal(a, c)
    if sampler == "host":
        # APR: the Adagrad accumulators travel too
        P, Q, b = whole.mf.parameters()
        pre = str(tmp_path / "pre.npz")
        np.savez(pre, P=P, Q=Q, b=b)
        acfg = {"factors_num": 16, "lr": 0.05, "reg": 0.0, "reg_adv": 1.0, "eps": 0.05, "epochs": 2, "batch_size": 64, "pretrain_file": pre}
        APR.logger = _Log()
        np.random.seed(6)
        awhole = APR(dev, dict(acfg), ds, ev)
        awhole.train_model()
        ack = str(tmp_path / "apr.npz")
        np.random.seed(6)
        APR(dev, dict(acfg, epochs=1, checkpoint_file=ack), ds, ev).train_model()
        np.random.seed(777)
        asecond = APR(dev, dict(acfg, checkpoint_file=ack, resume=True), ds, ev)
        asecond.train_model()
        for a, c in zip(asecond.mf.parameters(), awhole.mf.parameters()):
            assert np.array_equal(a, c)
        for a, c in zip(asecond.accumulators, awhole.accumulators):
            assert np.array_equal(a.numpy(), c.numpy())


def test_bpr_large_batches_run_prepared_steps_on_the_versioned_table(dev, oracle, golden_ml100k):
    """Batches of 65,536 triples and more take the path the benchmark times (versioned user table, index of the
    next batch sorted ahead, tfrec_bpr_sgd_step_prepared_v; a short last batch included): two epochs on ML-100k equal
    the oracle-driven schedule, and the user table is whole again (flattened) when the epoch is over."""
    from tfrec_amd.evaluator import FoldOutEvaluator
    from tfrec_amd.model import BPR
    ds = ml100k(golden_ml100k)
    ev = FoldOutEvaluator(ds.train_matrix, ds.test_matrix, top_k=50)
    cfg = {"factors_num": 32, "lr": 0.0005, "reg": 0.05, "epochs": 2, "batch_size": 65536}
    assert cfg["batch_size"] >= BPR.VERSIONED_FROM
    BPR.logger = _Log()
    np.random.seed(2018)
    model = BPR(dev, cfg, ds, ev)
    P, Q, b = model.mf.parameters()
    model.train_model()
    assert model._ut is not None                       # one table for the model's life, not one per epoch
    rs = np.random.RandomState(2018)
    user_pos = oracle.csr_to_user_dict(ds.train_matrix)
    ws = oracle.Workspace(943, 1682, 32)
    for ep in range(2):
        u, i, j = oracle.bpr_epoch_triples(rs, user_pos, 1682)
        for idx in oracle.batch_order(rs, len(u), 65536):
            oracle.bpr_step(P, Q, b, u[idx], i[idx], j[idx], 0.0005, 0.05, ws=ws)
    for a, c in zip(model.mf.parameters(), (P, Q, b)):
        np.testing.assert_allclose(a, c, rtol=2e-4, atol=2e-6)


def test_evaluator_options_top_k_score_dtype_metric_sum(dev, oracle, golden_ml100k):
    """The optional evaluator settings of the configuration: ``top_k`` (the first k columns of a longer evaluation),
    ``metric_sum = float64`` (device sums: the reference's float32 means to 5e-6), ``score_dtype = bf16`` (close to
    the float32 ranking on a trained model), and the header line that names k."""
    from tfrec_amd.evaluator import FoldOutEvaluator
    g = golden_ml100k
    ds = ml100k(g)

    class _M:
        def device_factors(self):
            return dev, dP, dQ, db
    dP, dQ, db = dev.array(g["P"]), dev.array(g["Q"]), dev.array(g["b"])
    ref = FoldOutEvaluator(ds.train_matrix, ds.test_matrix).evaluate(_M()).reshape(5, 50)
    assert FoldOutEvaluator(ds.train_matrix, ds.test_matrix).metrics_info() == "Precision@1:1:50, Recall@1:1:50, MAP@1:1:50, NDCG@1:1:50, MRR@1:1:50"
    ev20 = FoldOutEvaluator.from_config({"top_k": 20}, ds.train_matrix, ds.test_matrix)
    assert ev20.metrics_info().startswith("Precision@1:1:20")
    np.testing.assert_allclose(ev20.evaluate(_M()).reshape(5, 20), ref[:, :20], rtol=1e-6, atol=1e-7)
    f64 = FoldOutEvaluator.from_config({"metric_sum": "float64"}, ds.train_matrix, ds.test_matrix).evaluate(_M()).reshape(5, 50)
    np.testing.assert_allclose(f64, ref, rtol=5e-6, atol=1e-7)
    bf = FoldOutEvaluator.from_config({"score_dtype": "bf16", "metric_sum": "float64"}, ds.train_matrix, ds.test_matrix).evaluate(_M()).reshape(5, 50)
    assert np.abs(bf - ref).max() < 0.02 and not np.array_equal(bf, ref)
    with pytest.raises(ValueError):
        FoldOutEvaluator.from_config({"score_dtype": "fp8"}, ds.train_matrix, ds.test_matrix)
    with pytest.raises(ValueError):
        FoldOutEvaluator.from_config({"metric_sum": "kahan"}, ds.train_matrix, ds.test_matrix)


def _run_model(tmp, model, model_ini, tfrec_extra=""):
    os.makedirs(tmp / "conf", exist_ok=True)
    (tmp / "conf" / (model + ".ini")).write_text("[hyperparameters]\n" + model_ini)
    ini = open(os.path.join(ROOT, "tfrec.ini")).read().replace("./dataset/", os.path.join(ROOT, "dataset") + "/")
    ini = ini.replace("model=BPR", "model=" + model) + tfrec_extra
    (tmp / "tfrec.ini").write_text(ini)
    if not os.path.exists(os.path.join(ROOT, "dataset", "movielens-100k-train.txt")):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "export_ml100k.py")])
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "run_model.py")], cwd=tmp, env=env, capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    return out.stdout


@pytest.mark.parametrize("model,ini,marker", [
    ("BPR", "lr = 0.01\nreg = 0.05\nfactors_num = 20\nbatch_size = 512\nepochs = 2\nsampler = device\ntop_k = 20\nmetric_sum = float64\n", "epoch 1:\t"),
    ("DNSBPR", "lr = 0.01\nreg = 0.05\nfactors_num = 20\nneg_num = 8\nbatch_size = 512\nepochs = 1\nsampler = device\ntop_k = 20\n"
               "metric_sum = float64\nitem_mode = sharded\n", "epoch 0:\t"),
])
def test_run_model_on_two_ranks_gives_the_single_gpu_log(tmp_path, model, ini, marker):
    """run_model.py with ``devices = 2`` starts two ranks of itself (both on the one test GPU, gloo between them), shards
    the tables, and rank 0 logs: the metrics of the last epoch agree with the single-process run of the same
    configuration (same draws, same global batches; sums associate differently across the ranks)."""
    one = _run_model(tmp_path / "one", model, ini)
    two = _run_model(tmp_path / "two", model, ini + "devices = 2\ndist_backend = 'gloo'\nsingle_device = True\n")
    vals = []
    for out in (one, two):
        line = [ln for ln in out.splitlines() if marker in ln][-1]
        vals.append(np.array([float(x) for x in line.split(marker)[1].split("\t")]))
        assert vals[-1].shape == (100,)
    assert vals[0][0] > 0.0
    np.testing.assert_allclose(vals[1], vals[0], rtol=0, atol=0.01)
    logs = os.listdir(tmp_path / "two" / "log" / "ml100k" / model)
    assert len(logs) == 1                                  # rank 0 alone opens a log file
    assert sum(1 for ln in two.splitlines() if marker in ln) == 1
