"""CPU tests: the oracle (oracle/mf_oracle.c, oracle/oracle.py) against the golden vectors
the reference's own code produced (tests/golden/make_golden.py)."""
import hashlib

import numpy as np
import pytest

from conftest import csr_from_pairs


def test_ml100k_known_answer(oracle, golden_ml100k):
    g = golden_ml100k
    U, I = g["P"].shape[0], g["Q"].shape[0]
    tr = csr_from_pairs(g["train_u"], g["train_i"], U, I)
    te = csr_from_pairs(g["test_u"], g["test_i"], U, I)
    users = g["test_users"]
    assert len(users) == 459
    mean, rank, _ = oracle.eval_fullrank(g["P"], g["Q"], g["b"], users,
                                         (tr.indptr, tr.indices), (te.indptr, te.indices), 50, score_mode=0)
    # published-by-measurement known answer (P@1, P@10, NDCG@50)
    assert abs(mean[0, 0] - 0.61220044) < 1e-6
    assert abs(mean[0, 9] - 0.45076251) < 1e-6
    assert abs(mean[3, 49] - 0.33114389) < 1e-6
    np.testing.assert_allclose(mean, g["metrics"], rtol=1e-6, atol=1e-7)
    if bool(g["tie_free"]):
        assert np.array_equal(rank, g["ranks"].astype(np.int32))
    else:   # rank lists may differ only where the reference's scores tie within rounding
        assert (rank == g["ranks"].astype(np.int32)).mean() > 0.999


@pytest.mark.parametrize("case", ["small", "subset", "short_truth", "neg_inf_tail", "c1_shape"])
@pytest.mark.parametrize("mode", [0, 1])
def test_eval_cases(oracle, golden_eval, case, mode):
    g = {k.split("/", 1)[1]: v for k, v in golden_eval.items() if k.startswith(case + "/")}
    for k in g["k_list"]:
        mean, rank, _ = oracle.eval_fullrank(g["P"], g["Q"], g["b"], g["test_users"],
                                             (g["train_indptr"], g["train_indices"]),
                                             (g["test_indptr"], g["test_indices"]), int(k), score_mode=mode)
        ref_rank = g["ranks_k%d" % k]
        if case == "neg_inf_tail":
            # the order inside the -inf tail is unspecified in the reference; compare the finite part
            R = g["P"] @ g["Q"].T + g["b"]
            for r, u in enumerate(g["test_users"]):
                n_fin = g["Q"].shape[0] - (g["train_indptr"][u + 1] - g["train_indptr"][u])
                n_fin = min(int(n_fin), int(k))
                assert np.array_equal(rank[r, :n_fin], ref_rank[r, :n_fin])
                tail = set(rank[r, n_fin:].tolist())
                assert tail <= set(g["train_indices"][g["train_indptr"][u]:g["train_indptr"][u + 1]].tolist())
        else:
            assert np.array_equal(rank, ref_rank)
        np.testing.assert_allclose(mean, g["metrics_k%d" % k], rtol=1e-6, atol=1e-7)


def test_metrics_bit_exact_vs_reference_cpp(oracle):
    """Per-user curves and the float mean reproduce the reference C++ bit for bit."""
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(5)
    n, k, I = 200, 50, 400
    ranks = np.stack([rng.permutation(I)[:k] for _ in range(n)]).astype(np.int32)
    truths = [rng.permutation(I)[: int(rng.integers(1, 60))].astype(np.int32) for _ in range(n)]
    ref = oracle.ref_evaluate(ranks, truths)
    per = np.stack([oracle.metrics_user(ranks[r], truths[r]) for r in range(n)])
    mine = oracle.metrics_mean(per)
    assert np.array_equal(mine, ref)


def test_sampler_restatement(oracle, golden_sampler, golden_ml100k):
    g, m = golden_sampler, golden_ml100k
    U, I = m["P"].shape[0], m["Q"].shape[0]
    tr = csr_from_pairs(m["train_u"], m["train_i"], U, I)
    user_pos = oracle.csr_to_user_dict(tr)
    rs = np.random.RandomState(2018)
    for ep in range(2):
        users, pos, neg = oracle.bpr_epoch_triples(rs, user_pos, I)
        assert np.array_equal(neg, g["epoch%d_neg" % ep].astype(np.int32))
        batches = oracle.batch_order(rs, len(users), 512)
        assert len(batches) == int(g["epoch%d_n_batches" % ep])
        assert len(batches[-1]) == int(g["epoch%d_last_len" % ep])
        flat = np.concatenate(batches)
        trip = np.stack([users[flat], pos[flat], neg[flat]]).astype(np.int32)
        assert hashlib.sha256(trip.tobytes()).hexdigest() == str(g["epoch%d_batched_sha" % ep])
        assert np.array_equal(trip[:, :512], g["epoch%d_first_batch" % ep])


def test_csr_to_user_dict_quirk(oracle, golden_sampler):
    import scipy.sparse as sp
    quirk = sp.csr_matrix((np.ones(4), ([1, 2, 2, 3], [0, 0, 3, 2])), shape=(4, 5))
    kept = sorted(oracle.csr_to_user_dict(quirk).keys())
    assert kept == golden_sampler["quirk_kept_rows"].tolist() == [2, 3]


def test_dns_pick(oracle, golden_sampler):
    g = golden_sampler
    rs = np.random.RandomState(99)
    P, Q, b = g["dns_P"], g["dns_Q"], g["dns_b"]
    I = Q.shape[0]
    indptr, indices = g["dns_indptr"], g["dns_indices"]
    pools, picks = [], []
    for u in range(P.shape[0]):
        items = indices[indptr[u]:indptr[u + 1]]
        if not (items.size and items.any()):
            continue
        logits = (P[u:u + 1] @ Q.T + b).reshape(-1)
        pool2, pick = oracle.dns_pool_and_pick(rs, logits, items, I, 15)
        pools.append(pool2); picks.append(pick)
    pool = np.concatenate(pools).astype(np.int32)
    assert np.array_equal(pool, g["dns_pool"])
    assert np.array_equal(np.concatenate(picks), g["dns_pick"])
    chosen, margin = oracle.dns_select(P, Q, b, g["dns_users"], pool, score_mode=0)
    ok = margin > 1e-6
    assert ok.mean() > 0.9 and np.array_equal(chosen[ok], g["dns_pick"][ok])
    chosen1, _ = oracle.dns_select(P, Q, b, g["dns_users"], pool, score_mode=1)
    assert np.array_equal(chosen1[ok], g["dns_pick"][ok])


BPR_CASES = ["b1", "b7", "b128", "b512", "b512dup", "d10"]


@pytest.mark.parametrize("name", BPR_CASES)
def test_bpr_step_vs_autograd(oracle, golden_steps, name):
    g = golden_steps
    P, Q, b = [g["bpr_%s_%s" % (name, k)].copy() for k in "PQb"]
    rank, regl = oracle.bpr_step(P, Q, b, g["bpr_%s_u" % name], g["bpr_%s_i" % name], g["bpr_%s_j" % name], 0.01, 0.05)
    np.testing.assert_allclose([rank, regl], g["bpr_%s_loss" % name], rtol=1e-9)
    for arr, k in ((P, "P1"), (Q, "Q1"), (b, "b1")):
        np.testing.assert_allclose(arr, g["bpr_%s_%s" % (name, k)], rtol=2e-7, atol=1e-9)


@pytest.mark.parametrize("name", ["b1", "b128", "b128reg", "b512"])
def test_apr_step_vs_autograd(oracle, golden_steps, name):
    g = golden_steps
    P, Q, b = [g["apr_%s_%s" % (name, k)].copy() for k in "PQb"]
    aP, aQ, ab = [np.full_like(x, 0.1) for x in (P, Q, b)]
    dP, dQ, db = [np.zeros_like(x) for x in (P, Q, b)]
    losses = oracle.apr_step(P, Q, b, aP, aQ, ab, g["apr_%s_u" % name], g["apr_%s_i" % name], g["apr_%s_j" % name],
                             0.05, float(g["apr_%s_reg" % name]), 1.0, 0.05, deltas=(dP, dQ, db))
    np.testing.assert_allclose(losses, g["apr_%s_loss" % name], rtol=1e-6)
    for arr, k in ((dP, "dP"), (dQ, "dQ"), (db, "db")):
        np.testing.assert_allclose(arr, g["apr_%s_%s" % (name, k)], rtol=1e-6, atol=1e-9)
    for arr, k in ((P, "P1"), (Q, "Q1"), (b, "b1"), (aP, "aP1"), (aQ, "aQ1"), (ab, "ab1")):
        np.testing.assert_allclose(arr, g["apr_%s_%s" % (name, k)], rtol=1e-6, atol=1e-8)


def test_multithreaded_oracle_step_equals_the_scalar_one(oracle):
    """oracle_bpr_step_mt (the benchmark's multi-core CPU baseline) leaves the same bits in every parameter as
    oracle_bpr_step, with hot items and duplicate users, and leaves the shared accumulators clean."""
    rng = np.random.default_rng(11)
    U, I, d, B = 700, 300, 20, 6000
    P = rng.uniform(-0.3, 0.3, (U, d)).astype(np.float32)
    Q = rng.uniform(-0.3, 0.3, (I, d)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, I).astype(np.float32)
    u = rng.integers(0, U, B).astype(np.int32)
    i = (rng.zipf(1.3, B) % I).astype(np.int32)
    j = rng.integers(0, I, B).astype(np.int32)
    ws = oracle.Workspace(U, I, d)
    a = [x.copy() for x in (P, Q, b)]
    la = oracle.bpr_step(*a, u, i, j, 0.01, 0.05, ws=ws)
    for threads in (2, 3, 8):
        c = [x.copy() for x in (P, Q, b)]
        lc = oracle.bpr_step(*c, u, i, j, 0.01, 0.05, ws=ws, threads=threads)
        assert all(np.array_equal(x, y) for x, y in zip(a, c))
        np.testing.assert_allclose(lc, la, rtol=1e-12)
    e = [x.copy() for x in (P, Q, b)]
    oracle.bpr_step(*e, u, i, j, 0.01, 0.05, ws=ws)
    assert all(np.array_equal(x, y) for x, y in zip(a, e))
