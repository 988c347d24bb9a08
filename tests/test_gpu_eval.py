"""GPU parity tests of full-rank evaluation: fused MFMA scoring + top-k, dense top-k, merge and
metrics (through the C ABI) against the golden vectors of the reference's own C++ evaluator and
against the CPU oracle.  Rank lists and hit patterns must match exactly; metric means exactly in
the reference-order float32 mode and within 5e-6 in the float64 mode."""
import numpy as np
import pytest

from conftest import csr_from_pairs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from tfrec_amd.device import Device
    d = Device(0)
    yield d
    d.close()


def dcsr(dev, indptr, indices):
    return dev.array(indptr, np.int64), dev.array(indices, np.int32)


def test_ml100k_known_answer(dev, golden_ml100k):
    g = golden_ml100k
    U, I = g["P"].shape[0], g["Q"].shape[0]
    tr = csr_from_pairs(g["train_u"], g["train_i"], U, I)
    te = csr_from_pairs(g["test_u"], g["test_i"], U, I)
    P, Q, b = dev.array(g["P"]), dev.array(g["Q"]), dev.array(g["b"])
    users = dev.array(g["test_users"], np.int32)
    train, test = dcsr(dev, tr.indptr, tr.indices), dcsr(dev, te.indptr, te.indices)
    ids = dev.empty((len(g["test_users"]), 50), np.int32)
    m_ref_order = dev.eval_fullrank(P, Q, b, users, train, test, 50, mean_mode=1, out_ids=ids)
    assert np.array_equal(ids.numpy(), g["ranks"].astype(np.int32))
    assert np.array_equal(m_ref_order, g["metrics"])          # bit for bit the reference's float32 means
    # float64 tree sum: differs from the reference only by the reference's own float32 summation error
    m = dev.eval_fullrank(P, Q, b, users, train, test, 50, mean_mode=0)
    np.testing.assert_allclose(m, g["metrics"], rtol=5e-6, atol=1e-7)
    assert abs(m[0, 0] - 0.61220044) < 1e-6 and abs(m[3, 49] - 0.33114389) < 1e-6


@pytest.mark.parametrize("case", ["small", "subset", "short_truth", "neg_inf_tail", "c1_shape"])
def test_eval_cases_factor_form(dev, golden_eval, case):
    g = {k.split("/", 1)[1]: v for k, v in golden_eval.items() if k.startswith(case + "/")}
    P, Q, b = dev.array(g["P"]), dev.array(g["Q"]), dev.array(g["b"])
    users = dev.array(g["test_users"], np.int32)
    train, test = dcsr(dev, g["train_indptr"], g["train_indices"]), dcsr(dev, g["test_indptr"], g["test_indices"])
    for k in g["k_list"]:
        k = int(k)
        ids = dev.empty((len(g["test_users"]), k), np.int32)
        m = dev.eval_fullrank(P, Q, b, users, train, test, k, mean_mode=1, out_ids=ids)
        got, ref = ids.numpy(), g["ranks_k%d" % k]
        if case == "neg_inf_tail":
            for r, u in enumerate(g["test_users"]):
                row = g["train_indices"][g["train_indptr"][u]:g["train_indptr"][u + 1]]
                n_fin = min(g["Q"].shape[0] - len(row), k)
                assert np.array_equal(got[r, :n_fin], ref[r, :n_fin])
                assert np.array_equal(got[r, n_fin:], row[:k - n_fin])     # masked tail: train items, ascending
        else:
            assert np.array_equal(got, ref)
        assert np.array_equal(m, g["metrics_k%d" % k])


@pytest.mark.parametrize("case", ["small", "subset", "neg_inf_tail"])
def test_dense_topk_and_metrics(dev, golden_eval, oracle, case):
    """The dense drop-in path: c_top_k_array_index + evaluate replacements."""
    g = {k.split("/", 1)[1]: v for k, v in golden_eval.items() if k.startswith(case + "/")}
    R = (g["P"] @ g["Q"].T + g["b"]).astype(np.float32)
    users = g["test_users"]
    dR = dev.array(R[users])
    mask = dcsr(dev, g["train_indptr"], g["train_indices"])
    test = dcsr(dev, g["test_indptr"], g["test_indices"])
    du = dev.array(users, np.int32)
    for k in g["k_list"]:
        k = int(k)
        ids, scores = dev.topk_dense(dR, k, mask=mask, mask_rows=du)
        got, ref = ids.numpy(), g["ranks_k%d" % k]
        if case == "neg_inf_tail":
            dev.topk_fill_masked(ids, du, k, mask)
            fin = scores.numpy() > -np.inf
            assert np.array_equal(got[fin], ref[fin])
        else:
            assert np.array_equal(got, ref)
            sc = scores.numpy()
            assert np.array_equal(sc, np.take_along_axis(R[users], got, axis=1))
        m = dev.rank_metrics(ids, du, k, test, mean_mode=1)
        assert np.array_equal(m, g["metrics_k%d" % k])


def test_f32_scores_are_the_fma_chain(dev, oracle):
    """TFREC_SCORE_F32 is bit-for-bit acc=b; acc=fmaf(p[f],q[f],acc) for f ascending."""
    rng = np.random.default_rng(9)
    for (U, I, d, k) in [(70, 3000, 64, 100), (33, 1000, 20, 10), (40, 777, 128, 50), (10, 300, 7, 5)]:
        P = rng.normal(size=(U, d)).astype(np.float32); Q = rng.normal(size=(I, d)).astype(np.float32)
        b = rng.normal(size=I).astype(np.float32)
        users = np.arange(U, dtype=np.int32)
        ids, scores = dev.score_topk(dev.array(P), dev.array(Q), dev.array(b), dev.array(users), k)
        ids, scores = ids.numpy(), scores.numpy()
        for u in range(U):
            row = oracle.score_row(P[u], Q, b, mode=1)
            ref = oracle.topk_row(row, k)
            assert np.array_equal(ids[u], ref), (U, I, d, k, u)
            assert np.array_equal(scores[u], row[ref])


def test_ties_break_by_ascending_id(dev):
    U, I, d, k = 5, 600, 8, 20
    P = np.ones((U, d), np.float32); Q = np.zeros((I, d), np.float32); b = np.zeros(I, np.float32)
    b[100:140] = 1.0                      # forty equal best scores
    ids, _ = dev.score_topk(dev.array(P), dev.array(Q), dev.array(b), dev.array(np.arange(U, dtype=np.int32)), k)
    assert np.array_equal(ids.numpy(), np.tile(np.arange(100, 120, dtype=np.int32), (U, 1)))
    R = np.tile(b, (U, 1))
    ids2, _ = dev.topk_dense(dev.array(R), k)
    assert np.array_equal(ids2.numpy(), ids.numpy())


def test_nan_and_infinite_scores_have_a_defined_ranking(dev):
    """A diverged model: +inf scores rank first (ascending id among them), an item whose score is NaN never
    enters a list, and a user whose every score is NaN gets an empty list (ids -1, scores -inf) -- in both
    scoring modes and in the dense top-k (the scoring kernels are built without NaN-honouring comparisons, so this
    behaviour is pinned here rather than left to the compiler)."""
    from tfrec_amd.device import SCORE_BF16, SCORE_F32
    rng = np.random.default_rng(3)
    U, I, d, k = 70, 900, 32, 20
    P = rng.normal(size=(U, d)).astype(np.float32)
    Q = rng.normal(size=(I, d)).astype(np.float32)
    b = np.zeros(I, np.float32)
    b[[5, 17]] = np.inf
    b[33] = -np.inf
    b[[40, 41]] = np.nan
    Q[100, 3] = np.nan
    P[7, :] = np.nan
    nan_items = {40, 41, 100}
    eu = dev.array(np.arange(U, dtype=np.int32), np.int32)
    with np.errstate(invalid="ignore"):
        R = (P.astype(np.float64) @ Q.T.astype(np.float64) + b).astype(np.float32)
    finite_best = np.argsort(-np.where(np.isfinite(R), R, -np.inf), axis=1, kind="stable")
    results = [dev.score_topk(dev.array(P), dev.array(Q), dev.array(b), eu, k, mode=m) for m in (SCORE_F32, SCORE_BF16)]
    results.append(dev.topk_dense(dev.array(R), k))
    dev.synchronize()
    for which, (ids, sc) in enumerate(results):
        ids, sc = ids.numpy(), sc.numpy()
        for u in range(U):
            if u == 7:
                assert (ids[u] == -1).all() and np.isneginf(sc[u]).all(), which
                continue
            assert ids[u, 0] == 5 and ids[u, 1] == 17 and np.isposinf(sc[u, :2]).all(), (which, u)
            assert not (set(ids[u].tolist()) & nan_items) and 33 not in ids[u], (which, u)
            assert len(set(ids[u].tolist())) == k and ids[u].min() >= 0 and ids[u].max() < I, (which, u)
            assert np.isfinite(sc[u, 2:]).all() and (np.diff(sc[u, 2:]) <= 0).all(), (which, u)
            if which != 1:          # exact arithmetic: the finite part of the list is the float ranking itself
                assert np.array_equal(ids[u, 2:6], finite_best[u, :4]), (which, u)


def test_item_sharded_merge_equals_single_pass(dev):
    rng = np.random.default_rng(4)
    U, I, d, k, G = 300, 9000, 64, 100, 3
    P = rng.normal(size=(U, d)).astype(np.float32); Q = rng.normal(size=(I, d)).astype(np.float32)
    b = rng.normal(size=I).astype(np.float32)
    indptr = np.arange(0, 20 * (U + 1), 20, dtype=np.int64)
    indices = np.concatenate([np.sort(rng.choice(I, 20, replace=False)) for _ in range(U)]).astype(np.int32)
    dP, dQ, db = dev.array(P), dev.array(Q), dev.array(b)
    users = dev.array(np.arange(U, dtype=np.int32))
    train = dcsr(dev, indptr, indices)
    full_ids, full_sc = dev.score_topk(dP, dQ, db, users, k, train=train)
    bounds = [0, 2500, 2500, 9000]       # includes an empty shard
    ids_all = dev.empty((G, U, k), np.int32); sc_all = dev.empty((G, U, k), np.float32)
    hi, hs = [], []
    for gi in range(G):
        a, s = dev.score_topk(dP, dQ, db, users, k, train=train, item_range=(bounds[gi], bounds[gi + 1]))
        hi.append(a.numpy()); hs.append(s.numpy())
    ids_all.copy_from(np.stack(hi)); sc_all.copy_from(np.stack(hs))
    mi, ms = dev.topk_merge(ids_all, sc_all, G, U, k)
    assert np.array_equal(mi.numpy(), full_ids.numpy()) and np.array_equal(ms.numpy(), full_sc.numpy())
    got = full_ids.numpy()
    for u in range(U):
        assert not set(got[u]) & set(indices[indptr[u]:indptr[u + 1]])


def test_few_users_many_items_uses_item_splits(dev, oracle):
    rng = np.random.default_rng(6)
    U, I, d, k = 7, 40000, 32, 50
    P = rng.normal(size=(U, d)).astype(np.float32); Q = rng.normal(size=(I, d)).astype(np.float32)
    b = rng.normal(size=I).astype(np.float32)
    ids, sc = dev.score_topk(dev.array(P), dev.array(Q), dev.array(b), dev.array(np.arange(U, dtype=np.int32)), k)
    for u in range(U):
        row = oracle.score_row(P[u], Q, b, mode=1)
        assert np.array_equal(ids.numpy()[u], oracle.topk_row(row, k))


@pytest.mark.parametrize("k", [100, 240])
def test_a_handful_of_users_merges_item_splits_in_two_levels(dev, oracle, k):
    """Five users against 300,000 items: the item range goes to many more workgroups than one merge launch takes
    lists (2048 candidates per user), so the partial lists are merged in two levels; same lists as one pass."""
    rng = np.random.default_rng(k)
    U, I, d = 5, 300000, 32
    P = rng.normal(size=(U, d)).astype(np.float32); Q = rng.normal(size=(I, d)).astype(np.float32)
    b = rng.normal(size=I).astype(np.float32)
    deg = 40
    indptr = np.arange(0, deg * (U + 1), deg, dtype=np.int64)
    rows = [oracle.score_row(P[u], Q, b, mode=1) for u in range(U)]
    # mask each user's best items, so that masking across splits matters
    indices = np.concatenate([np.sort(np.argsort(-rows[u], kind="stable")[:deg]) for u in range(U)]).astype(np.int32)
    ids, sc = dev.score_topk(dev.array(P), dev.array(Q), dev.array(b), dev.array(np.arange(U, dtype=np.int32)), k,
                             train=dcsr(dev, indptr, indices))
    dev.synchronize()
    for u in range(U):
        row = rows[u].copy()
        row[indices[indptr[u]:indptr[u + 1]]] = -np.inf
        assert np.array_equal(ids.numpy()[u], oracle.topk_row(row, k))
        assert np.array_equal(sc.numpy()[u], row[ids.numpy()[u]])


def test_bf16_scoring(dev, oracle):
    rng = np.random.default_rng(8)
    U, I, d, k = 200, 5000, 128, 100
    P = rng.normal(size=(U, d)).astype(np.float32); Q = rng.normal(size=(I, d)).astype(np.float32)
    b = rng.normal(size=I).astype(np.float32)
    from tfrec_amd.device import SCORE_BF16
    ids, sc = dev.score_topk(dev.array(P), dev.array(Q), dev.array(b), dev.array(np.arange(U, dtype=np.int32)), k, mode=SCORE_BF16)
    ids, sc = ids.numpy(), sc.numpy()
    agree = 0
    for u in range(U):
        row = oracle.score_row(P[u], Q, b, mode=2)           # bf16-rounded operands, exact accumulation
        # scores: float32 accumulation of exact bf16 products -> 1e-5 relative to the row's magnitude
        np.testing.assert_allclose(sc[u], row[ids[u]], rtol=0, atol=2e-5 * np.abs(row).max())
        ref = oracle.topk_row(row, k)
        agree += len(set(ref.tolist()) & set(ids[u].tolist()))
        # wherever the reference ranking has a clear margin the id must agree position by position
        gaps = -np.diff(row[ref])
        clear = np.concatenate([[True], gaps > 1e-4]) & np.concatenate([gaps > 1e-4, [True]])
        assert np.array_equal(ids[u][clear], ref[clear])
    assert agree >= 0.999 * U * k


def test_large_shape_properties(dev):
    """Size-independent checks at a size the oracle would not finish quickly."""
    rng = np.random.default_rng(10)
    U, I, d, k = 20000, 60000, 64, 100
    P = dev.uniform((U, d), -0.5, 0.5, 1); Q = dev.uniform((I, d), -0.5, 0.5, 2); b = dev.uniform((I,), -0.1, 0.1, 3)
    users = np.sort(rng.choice(U, 5000, replace=False)).astype(np.int32)
    deg = 30
    indptr = np.arange(0, deg * (U + 1), deg, dtype=np.int64)
    indices = np.sort(rng.integers(0, I, (U, deg)), axis=1).astype(np.int32).ravel()
    train = dcsr(dev, indptr, indices)
    ids, sc = dev.score_topk(P, Q, b, dev.array(users), k, train=train)
    ids, sc = ids.numpy(), sc.numpy()
    assert (np.diff(sc, axis=1) <= 0).all()                    # sorted
    assert (ids >= 0).all() and (ids < I).all()
    srt = np.sort(ids, axis=1)
    assert (np.diff(srt, axis=1) > 0).all()                    # no duplicates
    Ph, Qh, bh = P.numpy(), Q.numpy(), b.numpy()
    for r in rng.choice(len(users), 40, replace=False):
        u = users[r]
        row = (Ph[u].astype(np.float64) @ Qh.T.astype(np.float64)) + bh
        row[indices[indptr[u]:indptr[u + 1]]] = -np.inf
        best = np.argsort(-row, kind="stable")[:k]
        gaps = -np.diff(row[best])
        if gaps.min() > 1e-5:
            assert np.array_equal(ids[r], best)
        assert abs(sc[r, -1] - row[best[-1]]) < 1e-5
        assert not set(ids[r]) & set(indices[indptr[u]:indptr[u + 1]])


@pytest.mark.parametrize("mode_name", ["f32", "bf16"])
def test_full_size_item_table_top100(mode_name):
    """The item side at its full size (2M items, d=128, k=100, ten masked train items per user): the fused
    scoring + top-k against a float64 matrix product and torch.topk for 512 users.  f32 scoring: the same
    ids wherever neighbouring scores are further apart than the float32 rounding of a 128-term dot product;
    bf16 scoring: the lists overlap the exact ones almost entirely and every reported score is the bf16
    score of its item.  Both: scores descend, no id twice, no masked id."""
    import torch
    from tfrec_amd.device import Device, SCORE_BF16, SCORE_F32
    cuda = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=cuda)
    with torch.cuda.stream(stream):
        d2 = Device(0, stream=stream.cuda_stream)
        U, I, d, k, n, deg = 100_000, 2_000_000, 128, 100, 512, 10
        g = torch.Generator(device=cuda); g.manual_seed(17)
        P = (torch.rand(U, d, device=cuda, generator=g) - 0.5) * 0.4
        Q = (torch.rand(I, d, device=cuda, generator=g) - 0.5) * 0.4
        b = torch.randn(I, device=cuda, generator=g) * 0.05
        users = torch.sort(torch.randperm(U, device=cuda, generator=g)[:n]).values.to(torch.int32)
        indptr = torch.arange(0, deg * (U + 1), deg, device=cuda, dtype=torch.int64)
        indices = torch.sort(torch.randint(0, I, (U, deg), device=cuda, generator=g, dtype=torch.int32), dim=1).values.reshape(-1).contiguous()
        ids = torch.empty(n, k, device=cuda, dtype=torch.int32)
        sc = torch.empty(n, k, device=cuda)
        mode = SCORE_F32 if mode_name == "f32" else SCORE_BF16
        d2.score_topk(P, Q, b, users, k, train=(indptr, indices), mode=mode, out_ids=ids, out_scores=sc)
        d2.synchronize()
        Pu = P[users.long()]
        if mode_name == "bf16":
            R = Pu.to(torch.bfloat16).double() @ Q.to(torch.bfloat16).double().T + b.double()
        else:
            R = Pu.double() @ Q.double().T + b.double()
        masked = indices.view(U, deg)[users.long()].long()
        R.scatter_(1, masked, float("-inf"))
        top = torch.topk(R, k + 1, dim=1)
        idl = ids.long()
        assert bool((sc[:, 1:] <= sc[:, :-1]).all())
        assert bool((torch.sort(idl, dim=1).values.diff(dim=1) > 0).all())
        assert bool(torch.isfinite(R.gather(1, idl)).all())                     # no masked item reported
        # every reported score is the (float64-accumulated) score of its item in the mode's operand precision
        assert torch.allclose(sc.double(), R.gather(1, idl), rtol=0, atol=2e-5)
        gaps = top.values[:, :-1] - top.values[:, 1:]                           # gap below each of the k best
        above = torch.cat([torch.full((n, 1), float("inf"), device=cuda, dtype=gaps.dtype), gaps[:, :-1]], dim=1)
        clear = (gaps > 2e-5) & (above > 2e-5)                                  # positions no rounding can move
        assert float(clear.float().mean()) > 0.3
        assert torch.equal(idl[clear], top.indices[:, :k][clear])
        # rows with near-ties: same sets up to the items whose scores are within rounding of the cut
        same = (idl.unsqueeze(2) == top.indices[:, :k].unsqueeze(1)).any(2).float().mean()
        assert float(same) > 0.999
        d2.close()


@pytest.mark.parametrize("k,deg", [(50, 20), (100, 24), (100, 40), (10, 100)])
def test_masked_items_at_the_top_of_every_ranking(dev, k, deg):
    """Train items that would fill the top of the list (their biases tower over everything else): whether the
    scan checks them at every compaction (long rows) or only in the final pass (short rows, k + row length
    survivors kept meanwhile), the result is the exact top-k of the unmasked items."""
    rng = np.random.default_rng(k * 1000 + deg)
    U, I, d = 700, 30000, 64
    P = rng.uniform(-0.3, 0.3, (U, d)).astype(np.float32)
    Q = rng.uniform(-0.3, 0.3, (I, d)).astype(np.float32)
    b = rng.normal(0, 0.05, I).astype(np.float32)
    hot = rng.choice(I, 3 * deg, replace=False)
    b[hot] += 10.0
    rows = np.stack([np.sort(rng.choice(hot, deg, replace=False)) for _ in range(U)]).astype(np.int32)
    indptr = np.arange(0, deg * (U + 1), deg, dtype=np.int64)
    users = np.arange(U, dtype=np.int32)
    ids, sc = dev.score_topk(dev.array(P), dev.array(Q), dev.array(b), dev.array(users), k, train=dcsr(dev, indptr, rows.ravel()))
    ids, sc = ids.numpy(), sc.numpy()
    R = P.astype(np.float64) @ Q.T.astype(np.float64) + b
    np.put_along_axis(R, rows.astype(np.int64), -np.inf, axis=1)
    best = np.argsort(-R, axis=1, kind="stable")[:, :k]
    want = np.take_along_axis(R, best, axis=1)
    assert np.abs(sc - want).max() < 1e-4
    top1 = np.take_along_axis(R, np.argsort(-R, axis=1, kind="stable")[:, :k + 1], axis=1)
    below = -np.diff(top1, axis=1)                                          # gap under each of the k best
    above = np.concatenate([np.full((U, 1), np.inf), below[:, :-1]], axis=1)
    clear = (below > 1e-5) & (above > 1e-5)                                  # positions no float32 rounding can move
    assert clear.mean() > 0.5
    assert np.array_equal(ids[clear], best[clear])
    for u in range(0, U, 50):
        assert not set(ids[u]) & set(rows[u])


@pytest.mark.parametrize("mode_name", ["f32", "bf16"])
@pytest.mark.parametrize("d", [8, 20, 32, 48, 64, 100, 128, 200, 256])
@pytest.mark.parametrize("k", [1, 50, 100, 113, 240])
def test_every_kernel_instantiation(dev, mode_name, d, k):
    """Every (padded factor size, list capacity, score mode) instantiation of the scoring kernel -- they differ in
    waves per workgroup, tile length and registers -- on a user count that leaves waves and workgroups partly
    filled, with masked rows of mixed length, against a float64 product in the mode's operand precision."""
    import torch
    from tfrec_amd.device import SCORE_BF16, SCORE_F32
    if mode_name == "f32" and d > 128:
        with pytest.raises(Exception):
            dev.score_topk(dev.zeros((4, d), np.float32), dev.zeros((9, d), np.float32), dev.zeros(9, np.float32),
                           dev.array(np.arange(4, dtype=np.int32)), 1, mode=SCORE_F32)
        return
    rng = np.random.default_rng(d * 1000 + k)
    U, I, n = 900, 7001, 837
    P = rng.uniform(-0.5, 0.5, (U, d)).astype(np.float32)
    Q = rng.uniform(-0.5, 0.5, (I, d)).astype(np.float32)
    b = rng.normal(0, 0.1, I).astype(np.float32)
    users = np.sort(rng.choice(U, n, replace=False)).astype(np.int32)
    lens = rng.choice([0, 3, 20, 150, 700], U)
    indptr = np.zeros(U + 1, np.int64); indptr[1:] = np.cumsum(lens)
    indices = np.concatenate([np.sort(rng.choice(I, m, replace=False)) for m in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
    mode = SCORE_F32 if mode_name == "f32" else SCORE_BF16
    ids, sc = dev.score_topk(dev.array(P), dev.array(Q), dev.array(b), dev.array(users), k, train=dcsr(dev, indptr, indices), mode=mode)
    ids, sc = ids.numpy(), sc.numpy()
    if mode_name == "bf16":
        rnd = lambda x: torch.from_numpy(x).to(torch.bfloat16).double().numpy()
    else:
        rnd = lambda x: x.astype(np.float64)
    R = rnd(P[users]) @ rnd(Q).T + b
    for r, u in enumerate(users):
        R[r, indices[indptr[u]:indptr[u + 1]]] = -np.inf
    order = np.argsort(-R, axis=1, kind="stable")[:, :k + 1]
    top = np.take_along_axis(R, order, axis=1)
    assert np.abs(sc - top[:, :k]).max() < 2e-5
    below = top[:, :k] - top[:, 1:]
    above = np.concatenate([np.full((n, 1), np.inf), below[:, :-1]], axis=1)
    clear = (below > 2e-5) & (above > 2e-5)
    assert clear.mean() > 0.5
    assert np.array_equal(ids[clear], order[:, :k][clear])
    assert (np.diff(sc, axis=1) <= 0).all()


@pytest.mark.parametrize("case", ["small", "subset", "short_truth", "c1_shape"])
def test_sums_only_metrics_agree_with_the_per_user_curves(dev, golden_eval, case):
    """The two ways tfrec_rank_metrics_sums forms its sums -- straight from the rank lists (no per-user table; precision,
    NDCG and MRR then come from integer hit counts in float64) and from the per-user float32 curves (each curve
    rounded to float per user, as evaluator/backend/cpp/metrics.h:59-100 does) -- agree to 2e-7 relative, and both
    means agree with the reference's float32 means of the golden cases to 5e-6."""
    g = {k.split("/", 1)[1]: v for k, v in golden_eval.items() if k.startswith(case + "/")}
    users = dev.array(g["test_users"], np.int32)
    test = dcsr(dev, g["test_indptr"], g["test_indices"])
    n = len(g["test_users"])
    for k in g["k_list"]:
        k = int(k)
        ids = dev.array(g["ranks_k%d" % k].astype(np.int32))
        direct = dev.rank_metrics_sums(ids, users, k, test).numpy()
        rows = dev.empty((n, 5 * k), np.float32)
        via_rows = dev.rank_metrics_sums(ids, users, k, test, per_user=rows).numpy()
        dev.synchronize()
        np.testing.assert_allclose(direct, via_rows, rtol=2e-7, atol=1e-9)
        np.testing.assert_allclose(via_rows, rows.numpy().astype(np.float64).sum(0), rtol=1e-12)
        for sums in (direct, via_rows):
            np.testing.assert_allclose((sums / n).reshape(5, k), g["metrics_k%d" % k], rtol=5e-6, atol=1e-7)


@pytest.mark.parametrize("shape", ["partly_filled", "few_users_many_splits", "k150_long_lists", "shard_rows", "one_short_tile"])
def test_both_bf16_kernels_give_identical_lists(dev, shape):
    """bf16 scoring at a padded factor size of 128 runs score_topk_pp_kernel (64 users per wave, tiles straight into a
    swizzled LDS image); TFREC_AMD_SCORE_KERNEL=1 keeps score_topk_kernel for that shape.  Both form every score from the
    same products in the same order and rank under the same total order, so ids AND scores must be bit-identical --
    whatever the two kernels' thresholds did on the way (the second one's compactions cut by histogram)."""
    import os
    from tfrec_amd.device import SCORE_BF16
    U, I, d, n, k, compact = {"partly_filled": (3000, 9001, 128, 1337, 100, False),
                              "few_users_many_splits": (400, 70000, 100, 70, 100, False),
                              "k150_long_lists": (1500, 6000, 128, 1100, 150, False),
                              "shard_rows": (2000, 5003, 96, 900, 50, True),
                              "one_short_tile": (700, 131, 128, 600, 20, False)}[shape]
    rng = np.random.default_rng(len(shape) * 7 + k)
    P = rng.uniform(-0.5, 0.5, (U, d)).astype(np.float32)
    Q = rng.uniform(-0.5, 0.5, (I, d)).astype(np.float32)
    b = rng.normal(0, 0.3, I).astype(np.float32)
    users = np.sort(rng.choice(U, n, replace=False)).astype(np.int32)
    lens = rng.choice([0, 2, 30, 200, 600], U) if I > 1000 else rng.choice([0, 2, 30], U)
    indptr = np.zeros(U + 1, np.int64); indptr[1:] = np.cumsum(lens)
    indices = np.concatenate([np.sort(rng.choice(I, m, replace=False)) for m in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
    train = dcsr(dev, indptr, indices)
    dQ, db, du = dev.array(Q), dev.array(b), dev.array(users)

    def run():
        if compact:     # the item-sharded evaluator's call: row r of P belongs to users[r], ids are offset
            return dev.score_topk_shard(dev.array(P[users]), dQ, db, 1000, du, k, train, SCORE_BF16)
        return dev.score_topk(dev.array(P), dQ, db, du, k, train=train, mode=SCORE_BF16)

    old = os.environ.get("TFREC_AMD_SCORE_KERNEL")
    try:
        os.environ["TFREC_AMD_SCORE_KERNEL"] = "1"
        ids1, sc1 = run()
        ids1, sc1 = ids1.numpy().copy(), sc1.numpy().copy()
        os.environ["TFREC_AMD_SCORE_KERNEL"] = "0"
        ids2, sc2 = run()
        ids2, sc2 = ids2.numpy().copy(), sc2.numpy().copy()
    finally:
        if old is None:
            os.environ.pop("TFREC_AMD_SCORE_KERNEL", None)
        else:
            os.environ["TFREC_AMD_SCORE_KERNEL"] = old
    assert np.array_equal(ids1, ids2)
    assert np.array_equal(sc1.view(np.uint32), sc2.view(np.uint32))
    assert (ids1 >= 0).sum() > 0.9 * ids1.size or I < k + 700      # real lists, not empty ones
