"""GPU parity tests of the training steps: HIP path (through the C ABI) against the CPU oracle
and the committed golden vectors.  Tolerances: post-step parameters and losses within 1e-5
relative (float32 path vs the float64 oracle); selected indices exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-5, 2e-7


@pytest.fixture(scope="module")
def dev():
    from tfrec_amd.device import Device
    d = Device(0)
    yield d
    d.close()


def run_bpr(dev, P, Q, b, u, i, j, lr, reg, want_loss=True):
    dP, dQ, db = dev.array(P), dev.array(Q), dev.array(b)
    du, di, dj = dev.array(u, np.int32), dev.array(i, np.int32), dev.array(j, np.int32)
    loss = dev.zeros(2, np.float64) if want_loss else None
    dev.bpr_sgd_step(dP, dQ, db, du, di, dj, lr, reg, loss)
    dev.synchronize()
    return dP.numpy(), dQ.numpy(), db.numpy(), (loss.numpy() if want_loss else None)


BPR_CASES = ["b1", "b7", "b128", "b512", "b512dup", "d10"]


@pytest.mark.parametrize("name", BPR_CASES)
def test_bpr_step_golden(dev, golden_steps, name):
    g = golden_steps
    P, Q, b = [g["bpr_%s_%s" % (name, k)] for k in "PQb"]
    P1, Q1, b1, loss = run_bpr(dev, P, Q, b, g["bpr_%s_u" % name], g["bpr_%s_i" % name], g["bpr_%s_j" % name], 0.01, 0.05)
    np.testing.assert_allclose(P1, g["bpr_%s_P1" % name], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(Q1, g["bpr_%s_Q1" % name], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(b1, g["bpr_%s_b1" % name], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(loss, g["bpr_%s_loss" % name], rtol=RTOL)


def zipf_ids(rng, n, size, a=1.0):
    w = 1.0 / np.arange(1, n + 1) ** a
    return rng.permutation(n)[rng.choice(n, size=size, p=w / w.sum())]


@pytest.mark.parametrize("U,I,d,B,kind", [
    (500, 400, 64, 4096, "uniform"),
    (500, 400, 128, 3000, "uniform"),
    (64, 48, 20, 5000, "uniform"),          # every row repeated ~100 times
    (3000, 2000, 64, 30000, "zipf"),        # hot items: chains of partials longer than one workgroup walks alone
    (5, 7, 32, 20000, "uniform"),           # a handful of rows, thousands of occurrences each
    (300, 200, 10, 777, "uniform"),         # d not a multiple of 4: scalar row path
    (300, 200, 256, 600, "uniform"),
    (1, 2, 8, 300, "uniform"),
    (3, 2, 64, 200000, "uniform"),          # one row spans thousands of tiles: pre-summed workgroup partials
])
def test_bpr_step_vs_oracle(dev, oracle, U, I, d, B, kind):
    rng = np.random.default_rng(U * 7 + B)
    P = rng.uniform(-0.3, 0.3, (U, d)).astype(np.float32)
    Q = rng.uniform(-0.3, 0.3, (I, d)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, I).astype(np.float32)
    u = rng.integers(0, U, B).astype(np.int32)
    if kind == "zipf":
        i = zipf_ids(rng, I, B).astype(np.int32)
        j = rng.integers(0, I, B).astype(np.int32)
    else:
        i = rng.integers(0, I, B).astype(np.int32)
        j = rng.integers(0, I, B).astype(np.int32)
    Po, Qo, bo = P.copy(), Q.copy(), b.copy()
    rank, regl = oracle.bpr_step(Po, Qo, bo, u, i, j, 0.01, 0.05)
    P1, Q1, b1, loss = run_bpr(dev, P, Q, b, u, i, j, 0.01, 0.05)
    # tolerances are relative to the size of the UPDATE for heavily repeated rows: compare increments
    for new, ref, old in ((P1, Po, P), (Q1, Qo, Q), (b1, bo, b)):
        scale = max(np.abs(ref - old).max(), 1e-3)
        assert np.abs(new - ref).max() <= 1e-5 * scale + 2e-7 * np.abs(ref).max()
    np.testing.assert_allclose(loss, [rank, regl], rtol=1e-5)
    # without the loss output the same parameters come out, bit for bit
    P2, Q2, b2, _ = run_bpr(dev, P, Q, b, u, i, j, 0.01, 0.05, want_loss=False)
    assert np.array_equal(P1, P2) and np.array_equal(Q1, Q2) and np.array_equal(b1, b2)


@pytest.mark.parametrize("path", ["versioned", "versioned_prepared", "dense_v", "dense_prepared_v", "dense_chunked"])
def test_long_tile_steps_on_a_versioned_table_vs_oracle(dev, oracle, path):
    """Batches of 300,000 triples (above the 131,072 at which the row passes switch to tiles of 64 occurrences and
    the index to the radix passes), Zipf positives, two consecutive steps on a versioned user table, straight against
    the CPU oracle: the plain versioned step, the step on batches sorted ahead, and the replicated-item-table
    step (gradient table + axpy) unprepared, prepared and with its item half in three ranges of item ids."""
    rng = np.random.default_rng(300000)
    U, I, d, B = 20000, 5000, 64, 300000
    lr, reg = 0.002, 0.05
    P = rng.uniform(-0.3, 0.3, (U, d)).astype(np.float32)
    Q = rng.uniform(-0.3, 0.3, (I, d)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, I).astype(np.float32)
    batches = [(rng.integers(0, U, B).astype(np.int32), zipf_ids(rng, I, B).astype(np.int32), rng.integers(0, I, B).astype(np.int32))
               for _ in range(2)]
    Po, Qo, bo = P.copy(), Q.copy(), b.copy()
    refs = []
    for u, i, j in batches:
        before = (Po.copy(), Qo.copy(), bo.copy())
        oracle.bpr_step(Po, Qo, bo, u, i, j, lr, reg)
        refs.append((before, (Po.copy(), Qo.copy(), bo.copy())))
    dP, dQ, db = dev.array(P), dev.array(Q), dev.array(b)
    ut = dev.user_table(dP)
    dbt = [[dev.array(x) for x in bt] for bt in batches]
    G = dev.zeros(I * d + I, np.float32)
    events = None
    if path == "dense_chunked":
        import torch
        events = [torch.cuda.Event() for _ in range(3)]
        for e in events:
            e.record()
    if path in ("versioned_prepared", "dense_prepared_v"):
        dev.prepare_batch(0, U, I, *dbt[0])
    for s in range(2):
        if path == "versioned":
            dev.bpr_sgd_step_v(ut, dQ, db, *dbt[s], lr, reg)
        elif path == "versioned_prepared":
            if s == 0:
                dev.prepare_batch(1, U, I, *dbt[1])
            dev.bpr_sgd_step_prepared_v(s, ut, dQ, db, lr, reg)
        else:
            if path == "dense_v":
                dev.bpr_dense_grad_step_v(ut, dQ, db, G, *dbt[s], lr, reg)
            elif path == "dense_prepared_v":
                if s == 0:
                    dev.prepare_batch(1, U, I, *dbt[1])
                dev.bpr_dense_grad_step_v(ut, dQ, db, G, None, None, None, lr, reg, slot=s)
            else:
                dev.bpr_dense_grad_step_chunked(ut, dQ, db, G, *dbt[s], lr, reg, [e.cuda_event for e in events])
            dev.axpy(dQ, G.view(0, I * d), -lr, n=I * d)
            dev.axpy(db, G.view(I * d, I * d + I), -lr, n=I)
        # after every step: the flattened table and the item tables against the oracle's (increments compared: hot
        # items collect tens of thousands of summed terms)
        ut.flatten()
        dev.synchronize()
        (old, ref) = refs[s]
        for new, r, o in zip((dP.numpy(), dQ.numpy(), db.numpy()), ref, old):
            scale = max(np.abs(r - o).max(), 1e-3)
            assert np.abs(new - r).max() <= 1e-5 * (s + 1) * scale + 2e-7 * np.abs(r).max(), (path, s)     # (step 2 starts from step 1's rounding)
    ut.close()


@pytest.mark.parametrize("B", [1, 2, 85, 86, 170, 171, 341, 342, 682, 683, 1365, 1366, 2730, 2731])
def test_bpr_step_at_the_size_classes_of_the_single_workgroup_sort(dev, oracle, B):
    """3B pairs on either side of every size class of the one-workgroup index sort (256, 512, 1024, 2048, 4096, 8192
    pairs) and of its hand-over to the radix passes: the step equals the oracle's; and two such batches sorted
    together (batch number above the row id in the keys) give, consumed one after the other, what two plain steps give."""
    rng = np.random.default_rng(B)
    U, I, d = 97, 61, 16
    P = rng.uniform(-0.3, 0.3, (U, d)).astype(np.float32)
    Q = rng.uniform(-0.3, 0.3, (I, d)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, I).astype(np.float32)
    u, i, j = [rng.integers(0, n, 2 * B).astype(np.int32) for n in (U, I, I)]
    Po, Qo, bo = P.copy(), Q.copy(), b.copy()
    oracle.bpr_step(Po, Qo, bo, u[:B], i[:B], j[:B], 0.01, 0.05)
    P1, Q1, b1, _ = run_bpr(dev, P, Q, b, u[:B], i[:B], j[:B], 0.01, 0.05)
    for new, ref, old in ((P1, Po, P), (Q1, Qo, Q), (b1, bo, b)):
        scale = max(np.abs(ref - old).max(), 1e-3)
        assert np.abs(new - ref).max() <= 1e-5 * scale + 2e-7 * np.abs(ref).max()
    a = [dev.array(x) for x in (P, Q, b)]
    du, di, dj = dev.array(u, np.int32), dev.array(i, np.int32), dev.array(j, np.int32)
    for s in range(2):
        dev.bpr_sgd_step(*a, du.view(s * B, (s + 1) * B), di.view(s * B, (s + 1) * B), dj.view(s * B, (s + 1) * B), 0.01, 0.05)
    c = [dev.array(x) for x in (P, Q, b)]
    dev.prepare_batches(0, U, I, du, di, dj, 2)
    for s in range(2):
        dev.bpr_sgd_step_prepared(0, *c, 0.01, 0.05)
    dev.synchronize()
    for x, y in zip(a, c):
        assert np.array_equal(x.numpy(), y.numpy())


def test_bpr_step_is_reproducible(dev):
    rng = np.random.default_rng(3)
    U, I, d, B = 200, 100, 64, 50000
    P = rng.uniform(-0.3, 0.3, (U, d)).astype(np.float32)
    Q = rng.uniform(-0.3, 0.3, (I, d)).astype(np.float32)
    b = np.zeros(I, np.float32)
    u, i, j = [rng.integers(0, n, B).astype(np.int32) for n in (U, I, I)]
    a = run_bpr(dev, P, Q, b, u, i, j, 0.01, 0.05)
    c = run_bpr(dev, P, Q, b, u, i, j, 0.01, 0.05)
    for x, y in zip(a, c):
        assert np.array_equal(x, y)


def test_prepared_steps_equal_plain_steps(dev):
    """Index preparation running ahead on the auxiliary stream changes nothing in the results."""
    rng = np.random.default_rng(12)
    U, I, d, B, steps = 3000, 2000, 64, 20000, 6
    P = rng.uniform(-0.3, 0.3, (U, d)).astype(np.float32)
    Q = rng.uniform(-0.3, 0.3, (I, d)).astype(np.float32)
    b = np.zeros(I, np.float32)
    batches = [[dev.array(rng.integers(0, n, B).astype(np.int32)) for n in (U, I, I)] for _ in range(steps)]
    a = [dev.array(x) for x in (P, Q, b)]
    for u, i, j in batches:
        dev.bpr_sgd_step(*a, u, i, j, 0.01, 0.05)
    c = [dev.array(x) for x in (P, Q, b)]
    loss = dev.zeros(2, np.float64)
    dev.prepare_batch(0, U, I, *batches[0])
    for s in range(steps):
        if s + 1 < steps:
            dev.prepare_batch((s + 1) & 1, U, I, *batches[s + 1])
        dev.bpr_sgd_step_prepared(s & 1, *c, 0.01, 0.05, loss if s == steps - 1 else None)
    dev.synchronize()
    for x, y in zip(a, c):
        assert np.array_equal(x.numpy(), y.numpy())
    assert loss.numpy()[0] > 0
    with pytest.raises(ValueError):
        dev.bpr_sgd_step_prepared(0, *c, 0.01, 0.05)        # the slot was consumed
    # several batches sorted in one pass, consumed one per step
    e = [dev.array(x) for x in (P, Q, b)]
    host = [[x.numpy() for x in bt] for bt in batches]
    for g0 in range(0, steps, 3):
        grp = [dev.array(np.concatenate([host[s][c] for s in range(g0, g0 + 3)])) for c in range(3)]
        dev.prepare_batches((g0 // 3) & 1, U, I, *grp, 3)
        for s in range(3):
            dev.bpr_sgd_step_prepared((g0 // 3) & 1, *e, 0.01, 0.05)
    dev.synchronize()
    for x, y in zip(a, e):
        assert np.array_equal(x.numpy(), y.numpy())


def test_bpr_empty_batch_and_bad_index(dev):
    P = np.ones((4, 8), np.float32); Q = np.ones((5, 8), np.float32); b = np.zeros(5, np.float32)
    e = np.zeros(0, np.int32)
    P1, Q1, b1, loss = run_bpr(dev, P, Q, b, e, e, e, 0.1, 0.1)
    assert np.array_equal(P1, P) and np.array_equal(Q1, Q) and loss.tolist() == [0.0, 0.0]
    dP, dQ, db = dev.array(P), dev.array(Q), dev.array(b)
    dev.bpr_sgd_step(dP, dQ, db, dev.array([0, 9], np.int32), dev.array([1, 1], np.int32), dev.array([2, 2], np.int32), 0.1, 0.1)
    with pytest.raises(IndexError):
        dev.synchronize()
    dev.synchronize()       # the flag is cleared once reported
    with pytest.raises(NotImplementedError):
        big = dev.zeros((2, 2048), np.float32)
        dev.bpr_sgd_step(big, big, db, dev.array([0], np.int32), dev.array([0], np.int32), dev.array([1], np.int32), 0.1, 0.1)


def run_apr(dev, P, Q, b, u, i, j, lr, reg, reg_adv, eps):
    arrs = [dev.array(x) for x in (P, Q, b)]
    accs = [dev.full(x.shape, 0.1) for x in (P, Q, b)]
    dels = [dev.zeros(x.shape) for x in (P, Q, b)]
    loss = dev.zeros(3, np.float64)
    dev.apr_adagrad_step(*arrs, *accs, *dels, dev.array(u, np.int32), dev.array(i, np.int32), dev.array(j, np.int32),
                         lr, reg, reg_adv, eps, loss)
    dev.synchronize()
    return [x.numpy() for x in arrs], [x.numpy() for x in accs], [x.numpy() for x in dels], loss.numpy()


@pytest.mark.parametrize("name", ["b1", "b128", "b128reg", "b512"])
def test_apr_step_golden(dev, golden_steps, name):
    g = golden_steps
    P, Q, b = [g["apr_%s_%s" % (name, k)] for k in "PQb"]
    u, i, j = [g["apr_%s_%s" % (name, k)] for k in "uij"]
    new, acc, dl, loss = run_apr(dev, P, Q, b, u, i, j, 0.05, float(g["apr_%s_reg" % name]), 1.0, 0.05)
    tu = np.unique(u); ti = np.unique(np.concatenate([i, j]))
    np.testing.assert_allclose(dl[0][tu], g["apr_%s_dP" % name][tu], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(dl[1][ti], g["apr_%s_dQ" % name][ti], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(dl[2][ti], g["apr_%s_db" % name][ti], rtol=1e-4, atol=1e-7)
    for arr, k in zip(new + acc, ("P1", "Q1", "b1", "aP1", "aQ1", "ab1")):
        np.testing.assert_allclose(arr, g["apr_%s_%s" % (name, k)], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(loss, g["apr_%s_loss" % name], rtol=1e-5)


@pytest.mark.parametrize("U,I,d,B,reg", [(400, 300, 64, 5000, 0.0), (50, 40, 20, 3000, 0.02), (800, 500, 128, 2000, 0.0)])
def test_apr_step_vs_oracle(dev, oracle, U, I, d, B, reg):
    rng = np.random.default_rng(B)
    P = rng.uniform(-0.3, 0.3, (U, d)).astype(np.float32)
    Q = rng.uniform(-0.3, 0.3, (I, d)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, I).astype(np.float32)
    u, i, j = [rng.integers(0, n, B).astype(np.int32) for n in (U, I, I)]
    Po, Qo, bo = P.copy(), Q.copy(), b.copy()
    aP, aQ, ab = [np.full_like(x, 0.1) for x in (P, Q, b)]
    losses = oracle.apr_step(Po, Qo, bo, aP, aQ, ab, u, i, j, 0.05, reg, 1.0, 0.05)
    new, acc, dl, loss = run_apr(dev, P, Q, b, u, i, j, 0.05, reg, 1.0, 0.05)
    for got, ref, old in zip(new, (Po, Qo, bo), (P, Q, b)):
        scale = max(np.abs(ref - old).max(), 1e-3)
        assert np.abs(got - ref).max() <= 5e-5 * scale + 2e-7 * np.abs(ref).max()
    for got, ref in zip(acc, (aP, aQ, ab)):
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(loss, losses, rtol=1e-5)


def test_dns_select_golden_and_random(dev, oracle, golden_sampler):
    g = golden_sampler
    P, Q, b = g["dns_P"], g["dns_Q"], g["dns_b"]
    users, pool = g["dns_users"], g["dns_pool"]
    got = dev.dns_select(dev.array(P), dev.array(Q), dev.array(b), dev.array(users, np.int32), dev.array(pool, np.int32)).numpy()
    ref, margin = oracle.dns_select(P, Q, b, users, pool)
    ok = margin > 1e-5
    assert ok.mean() > 0.9
    assert np.array_equal(got[ok], g["dns_pick"][ok]) and np.array_equal(got[ok], ref[ok])
    # every pick is one of that positive's candidates
    assert all(got[t] in pool[t] for t in range(len(got)))
    rng = np.random.default_rng(1)
    for (U, I, d, n, Cn) in [(1000, 5000, 128, 4000, 32), (300, 700, 64, 1000, 15), (100, 50, 20, 500, 7)]:
        P = rng.uniform(-0.5, 0.5, (U, d)).astype(np.float32)
        Q = rng.uniform(-0.5, 0.5, (I, d)).astype(np.float32)
        b = rng.uniform(-0.1, 0.1, I).astype(np.float32)
        users = rng.integers(0, U, n).astype(np.int32)
        cand = rng.integers(0, I, (n, Cn)).astype(np.int32)
        cand[::7, 3] = cand[::7, 1]       # duplicated candidates: equal scores, the first one must win
        got = dev.dns_select(dev.array(P), dev.array(Q), dev.array(b), dev.array(users), dev.array(cand)).numpy()
        ref, margin = oracle.dns_select(P, Q, b, users, cand)
        ok = margin > 1e-5
        assert ok.mean() > 0.95 and np.array_equal(got[ok], ref[ok])


def test_predict(dev):
    rng = np.random.default_rng(2)
    U, I, d, n = 300, 200, 20, 1000
    P = rng.normal(size=(U, d)).astype(np.float32); Q = rng.normal(size=(I, d)).astype(np.float32)
    b = rng.normal(size=I).astype(np.float32)
    u = rng.integers(0, U, n).astype(np.int32); i = rng.integers(0, I, n).astype(np.int32)
    y = dev.predict(dev.array(P), dev.array(Q), dev.array(b), dev.array(u), dev.array(i)).numpy()
    ref = (P[u].astype(np.float64) * Q[i]).sum(1) + b[i]
    np.testing.assert_allclose(y, ref, rtol=1e-5, atol=1e-5)


def test_fill_uniform_is_counter_based(dev):
    a = dev.uniform((1000, 64), -0.05, 0.05, 2018).numpy()
    c = dev.uniform((64000,), -0.05, 0.05, 2018).numpy()
    assert np.array_equal(a.ravel(), c)
    assert a.min() >= -0.05 and a.max() < 0.05 and abs(a.mean()) < 1e-3 and abs(a.std() - 0.1 / np.sqrt(12)) < 1e-3


@pytest.mark.parametrize("n,bits", [(1, 1), (63, 5), (4096, 8), (4097, 13), (100003, 21), (3 * (1 << 20), 21), (300000, 32), (70000, 17)])
def test_sort_pairs_is_a_stable_sort(dev, n, bits):
    """The index sort under every training step, against numpy's stable sort (exact)."""
    rng = np.random.default_rng(n + bits)
    hi = (1 << bits) - 1
    keys = rng.integers(0, hi + 1, n, dtype=np.uint64).astype(np.uint32)
    if n > 1000:
        keys[: n // 3] = keys[0]                      # a long run of one key
        keys[n // 3: n // 2] &= np.uint32(0xff)       # and a crowded corner of the key space
    vals = np.arange(n, dtype=np.uint32)
    ko, vo = dev.sort_pairs(dev.array(keys, np.uint32), dev.array(vals, np.uint32), bits)
    dev.synchronize()
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(ko.numpy(), keys[order])
    assert np.array_equal(vo.numpy(), vals[order])


def test_axpy_and_dense_grad_step_contract(dev):
    """tfrec_axpy is an exact fused multiply-add; the dense gradient step rejects bad arguments and,
    followed by axpy, equals the plain step bit for bit (same arithmetic in the same order)."""
    rng = np.random.default_rng(5)
    for n in (1, 7, 1024, 4099):
        y, x = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
        dy = dev.array(y)
        dev.axpy(dy, dev.array(x), -0.37)
        dev.synchronize()
        want = np.array([np.float32(np.float64(np.float32(-0.37)) * np.float64(a) + np.float64(b)) for a, b in zip(x, y)], np.float32)
        assert np.array_equal(dy.numpy(), want)
    U, I, d, B = 300, 200, 64, 5000
    P = rng.uniform(-0.3, 0.3, (U, d)).astype(np.float32)
    Q = rng.uniform(-0.3, 0.3, (I, d)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, I).astype(np.float32)
    u, i, j = [rng.integers(0, n, B).astype(np.int32) for n in (U, I, I)]
    P1, Q1, b1, _ = run_bpr(dev, P, Q, b, u, i, j, 0.01, 0.05, want_loss=False)
    dP = dev.array(P)
    G = dev.array(np.full(I * d + I, 7.0, np.float32))       # stale contents must not leak into the sums
    dQ, db = dev.array(Q), dev.array(b)
    dev.bpr_dense_grad_step(dP, dQ, db, G, dev.array(u), dev.array(i), dev.array(j), 0.01, 0.05)
    g = G.numpy()
    untouched = np.setdiff1d(np.arange(I), np.concatenate([i, j]))
    assert not g[:I * d].reshape(I, d)[untouched].any() and not g[I * d:][untouched].any()
    assert np.array_equal(dQ.numpy(), Q) and np.array_equal(db.numpy(), b)      # the item tables are read-only here
    dev.axpy(dQ, G.view(0, I * d), -0.01, n=I * d)
    dev.axpy(db, G.view(I * d, I * d + I), -0.01, n=I)
    dev.synchronize()
    assert np.array_equal(dP.numpy(), P1) and np.array_equal(dQ.numpy(), Q1) and np.array_equal(db.numpy(), b1)
    with pytest.raises(ValueError):
        dev.bpr_dense_grad_step(dP, dQ, db, G, dev.array(u), dev.array(i), dev.array(j), 0.01, 0.05, slot=5)
    with pytest.raises(ValueError):
        dev.bpr_dense_grad_step(dP, dQ, db, G, None, None, None, 0.01, 0.05, slot=0)     # nothing prepared in slot 0


def _sum_rows(n_rows, idx_vals):
    """Float64 sums of value rows by index without atomics (hot items make index_add_ crawl): sort, running
    sum, differences at the segment ends."""
    import torch
    idx = torch.cat([iv[0] for iv in idx_vals])
    val = torch.cat([iv[1] for iv in idx_vals])
    order = torch.argsort(idx)
    idx_s = idx[order]
    run = val[order].cumsum(0)
    last = torch.cat([torch.nonzero(idx_s[1:] != idx_s[:-1]).squeeze(1), torch.tensor([idx.shape[0] - 1], device=idx.device)])
    seg = run[last]
    seg[1:] = seg[1:] - run[last[:-1]]
    out = torch.zeros((n_rows,) + tuple(val.shape[1:]), device=val.device, dtype=val.dtype)
    out[idx_s[last]] = seg
    return out


def test_full_size_step_against_float64_sums(dev):
    _full_size_step_check(1_000_000, 200_000, 64, 1 << 22)


def test_full_size_step_10m_users_2m_items_d128(dev):
    """The large configuration's table shape on one GPU (10M users x 2M items, d=128: 6.1 GB of tables that no
    cache holds), 2M triples: same checks as at the bench size."""
    _full_size_step_check(10_000_000, 2_000_000, 128, 1 << 21)


def test_full_size_prepared_steps_on_a_versioned_table_against_float64_sums(dev):
    """The path bench.py times for its headline number, at that size (1M users x 200k items, d=64, batches of 4M
    triples: long tiles, three radix passes): a versioned user table, batches sorted ahead on the auxiliary stream,
    tfrec_bpr_sgd_step_prepared_v, two consecutive batches (after the first, a quarter of the rows live in the
    second buffer), flatten.  Every row against the float64 statement of the two steps."""
    _full_size_step_check(1_000_000, 200_000, 64, 1 << 22, path="versioned_prepared")


def test_full_size_versioned_steps_10m_users_2m_items_d128(dev):
    """What bench.py times at the large configuration's table shape: tfrec_bpr_sgd_step_v, 4M triples per step,
    10M users x 2M items, d=128, two consecutive steps, against float64 sums."""
    _full_size_step_check(10_000_000, 2_000_000, 128, 1 << 22, path="versioned")


def _full_size_step_check(U, I, d, B, path="plain"):
    if path != "plain":
        return _full_size_versioned_check(U, I, d, B, path)
    """A BPR step at full size (the bench workload: 1M users x 200k items, d=64, 4M triples, Zipf positives): every
    row of P, Q and b after one step against an independent float64 statement of model/BPR.py:36-48 built
    from torch gathers and sorted segment sums on the same device, plus two size-independent properties: a step with
    lr = 0 leaves every parameter bit for bit, and without regularisation the biases only move mass between
    items (each triple takes s from its negative's bias gradient and gives it to its positive's)."""
    import torch
    from tfrec_amd.device import Device
    cuda = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=cuda)
    with torch.cuda.stream(stream):
        d2 = Device(0, stream=stream.cuda_stream)
        lr, reg = 0.01, 0.05
        g = torch.Generator(device=cuda); g.manual_seed(99)
        P = (torch.rand(U, d, device=cuda, generator=g) - 0.5) * 0.6
        Q = (torch.rand(I, d, device=cuda, generator=g) - 0.5) * 0.6
        b = torch.randn(I, device=cuda, generator=g) * 0.1
        u = torch.randint(0, U, (B,), device=cuda, generator=g, dtype=torch.int32)
        w = 1.0 / torch.arange(1, I + 1, device=cuda, dtype=torch.float64)
        cdf = torch.cumsum(w, 0) / w.sum()
        i = torch.searchsorted(cdf, torch.rand(B, device=cuda, generator=g, dtype=torch.float64)).clamp_(max=I - 1).to(torch.int32)
        j = torch.randint(0, I, (B,), device=cuda, generator=g, dtype=torch.int32)

        def reference(reg_):
            ul, il, jl = u.long(), i.long(), j.long()
            p, qi, qj = P[ul].double(), Q[il].double(), Q[jl].double()
            z = (p * (qi - qj)).sum(1) + (b[il].double() - b[jl].double()) + 1e-7
            s = torch.sigmoid(-z)
            gP = _sum_rows(U, [(ul, -s[:, None] * (qi - qj) + reg_ * p)])
            gQ = _sum_rows(I, [(il, -s[:, None] * p + reg_ * qi), (jl, s[:, None] * p + reg_ * qj)])
            gb = _sum_rows(I, [(il, -s + reg_ * b[il].double()), (jl, s + reg_ * b[jl].double())])
            return P.double() - lr * gP, Q.double() - lr * gQ, b.double() - lr * gb

        P1, Q1, b1 = P.clone(), Q.clone(), b.clone()
        d2.bpr_sgd_step(P1, Q1, b1, u, i, j, lr, reg)
        d2.synchronize()
        for got, want in zip((P1, Q1, b1), reference(reg)):
            err = (got.double() - want).abs()
            tol = ATOL + RTOL * want.abs()
            # hot items sum tens of thousands of float32 terms: their error is bounded relative to the row, not the entry
            row_scale = want.abs().amax(dim=-1, keepdim=True) if want.dim() == 2 else want.abs()
            assert bool((err <= tol + RTOL * row_scale).all()), float((err - tol).max())
        # lr = 0: nothing moves
        P2, Q2, b2 = P.clone(), Q.clone(), b.clone()
        d2.bpr_sgd_step(P2, Q2, b2, u, i, j, 0.0, reg)
        d2.synchronize()
        assert torch.equal(P2, P) and torch.equal(Q2, Q) and torch.equal(b2, b)
        # reg = 0: the bias gradients of a step sum to zero
        P3, Q3, b3 = P.clone(), Q.clone(), b.clone()
        d2.bpr_sgd_step(P3, Q3, b3, u, i, j, lr, 0.0)
        d2.synchronize()
        moved = (b3.double() - b.double())
        assert abs(float(moved.sum())) <= 1e-5 * float(moved.abs().sum())
        d2.close()


def _float64_bpr_step(P, Q, b, u, i, j, lr, reg):
    """model/BPR.py:36-48 in float64 on the device: (P, Q, b) float64 tensors after one step on float64 inputs."""
    U, I = P.shape[0], Q.shape[0]
    ul, il, jl = u.long(), i.long(), j.long()
    p, qi, qj = P[ul], Q[il], Q[jl]
    import torch
    z = (p * (qi - qj)).sum(1) + (b[il] - b[jl]) + 1e-7
    s = torch.sigmoid(-z)
    gP = _sum_rows(U, [(ul, -s[:, None] * (qi - qj) + reg * p)])
    gQ = _sum_rows(I, [(il, -s[:, None] * p + reg * qi), (jl, s[:, None] * p + reg * qj)])
    gb = _sum_rows(I, [(il, -s + reg * b[il]), (jl, s + reg * b[jl])])
    # what an item's bias update is summed from: -s per occurrence as a positive, +s per occurrence as a negative.  The
    # two sides cancel, so the float32 sum is accurate relative to the sum of the magnitudes, not to the result.
    b_terms = lr * _sum_rows(I, [(il, s), (jl, s)])
    return P - lr * gP, Q - lr * gQ, b - lr * gb, b_terms


def _full_size_versioned_check(U, I, d, B, path):
    """Two consecutive steps on a versioned user table (``path``: "versioned" = tfrec_bpr_sgd_step_v, "versioned_prepared"
    = batches sorted ahead with tfrec_batch_prepare_many + tfrec_bpr_sgd_step_prepared_v), then flatten.  Every row
    of P, Q, b after the first step against a float64 step from the initial tables, and after both steps against a
    float64 step from the device's own result of the first (the steps are reproducible bit for bit, so the run of two
    steps passed through exactly that state, with the rows of the first batch's users in the second buffer); and
    lr = 0 leaves every parameter bit for bit, in whichever buffer its row lives."""
    import torch
    from tfrec_amd.device import Device
    cuda = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=cuda)
    with torch.cuda.stream(stream):
        d2 = Device(0, stream=stream.cuda_stream)
        lr, reg = 0.01, 0.05
        g = torch.Generator(device=cuda); g.manual_seed(199)
        P = (torch.rand(U, d, device=cuda, generator=g) - 0.5) * 0.6
        Q = (torch.rand(I, d, device=cuda, generator=g) - 0.5) * 0.6
        b = torch.randn(I, device=cuda, generator=g) * 0.1
        w = 1.0 / torch.arange(1, I + 1, device=cuda, dtype=torch.float64)
        cdf = torch.cumsum(w, 0) / w.sum()
        batches = []
        for _ in range(2):
            u = torch.randint(0, U, (B,), device=cuda, generator=g, dtype=torch.int32)
            i = torch.searchsorted(cdf, torch.rand(B, device=cuda, generator=g, dtype=torch.float64)).clamp_(max=I - 1).to(torch.int32)
            j = torch.randint(0, I, (B,), device=cuda, generator=g, dtype=torch.int32)
            batches.append((u, i, j))

        def run(lr_, n_steps):
            P1, Q1, b1 = P.clone(), Q.clone(), b.clone()
            ut = d2.user_table(P1)
            if path == "versioned_prepared":
                d2.prepare_batches(0, U, I, *batches[0], 1)
                for s in range(n_steps):
                    if s + 1 < n_steps:
                        d2.prepare_batches((s + 1) & 1, U, I, *batches[s + 1], 1)
                    d2.bpr_sgd_step_prepared_v(s & 1, ut, Q1, b1, lr_, reg)
            else:
                for bt in batches[:n_steps]:
                    d2.bpr_sgd_step_v(ut, Q1, b1, *bt, lr_, reg)
            ut.close()          # flattens
            d2.synchronize()
            return P1, Q1, b1

        start = (P, Q, b)
        for n_steps in (1, 2):
            got = run(lr, n_steps)
            *want, b_terms = _float64_bpr_step(*[x.double() for x in start], *batches[n_steps - 1], lr, reg)
            for gt, wt in zip(got, want):
                err = (gt.double() - wt).abs()
                tol = ATOL + RTOL * wt.abs()
                # hot items sum tens of thousands of float32 terms: their error is bounded relative to the row (and, for
                # the bias, to the magnitude of the terms that cancel in it), not to the entry
                row_scale = wt.abs().amax(dim=-1, keepdim=True) if wt.dim() == 2 else wt.abs() + b_terms
                assert bool((err <= tol + RTOL * row_scale).all()), (n_steps, float((err - tol - RTOL * row_scale).max()))
            del want, b_terms
            start = got
        del start, got
        same = run(0.0, 2)
        assert torch.equal(same[0], P) and torch.equal(same[1], Q) and torch.equal(same[2], b)
        d2.close()


def test_full_size_apr_step_against_float64_sums(dev):
    """The APR step (model/APR.py:86-136) at the bench size (1M users x 200k items, d=64, 4M triples) against
    a float64 statement built from torch gathers and sorted segment sums: deltas, Adagrad accumulators, parameters
    and the three loss terms.  Property: the bias delta is one vector of norm eps."""
    import torch
    from tfrec_amd.device import Device
    cuda = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=cuda)
    with torch.cuda.stream(stream):
        d2 = Device(0, stream=stream.cuda_stream)
        U, I, d, B = 1_000_000, 200_000, 64, 1 << 22
        lr, reg, reg_adv, eps = 0.05, 0.01, 1.0, 0.05
        g = torch.Generator(device=cuda); g.manual_seed(5)
        P = (torch.rand(U, d, device=cuda, generator=g) - 0.5) * 0.6
        Q = (torch.rand(I, d, device=cuda, generator=g) - 0.5) * 0.6
        b = torch.randn(I, device=cuda, generator=g) * 0.1
        u = torch.randint(0, U, (B,), device=cuda, generator=g, dtype=torch.int32)
        w = 1.0 / torch.arange(1, I + 1, device=cuda, dtype=torch.float64)
        cdf = torch.cumsum(w, 0) / w.sum()
        i = torch.searchsorted(cdf, torch.rand(B, device=cuda, generator=g, dtype=torch.float64)).clamp_(max=I - 1).to(torch.int32)
        j = torch.randint(0, I, (B,), device=cuda, generator=g, dtype=torch.int32)
        f64 = torch.float64
        ul, il, jl = u.long(), i.long(), j.long()

        def clipsig(r):
            return torch.where((r >= -80.0) & (r <= 1e8), torch.sigmoid(-r), torch.zeros_like(r))

        def softplus_neg(r):
            return torch.nn.functional.softplus(-r.clamp(-80.0, 1e8))

        def rows(shape, idx_vals):
            return _sum_rows(shape[0], idx_vals)

        p, qi, qj = P[ul].double(), Q[il].double(), Q[jl].double()
        bi, bj = b[il].double(), b[jl].double()
        diff = qi - qj
        r = (p * diff).sum(1) + (bi - bj)
        s = clipsig(r)
        gP = rows((U, d), [(ul, -s[:, None] * diff)])
        gQ = rows((I, d), [(il, -s[:, None] * p), (jl, s[:, None] * p)])
        gb = rows((I,), [(il, -s), (jl, s)])
        dP = eps * gP / gP.pow(2).sum(1, keepdim=True).clamp_min(1e-12).sqrt()
        dQ = eps * gQ / gQ.pow(2).sum(1, keepdim=True).clamp_min(1e-12).sqrt()
        db = eps * gb / gb.pow(2).sum().clamp_min(1e-12).sqrt()
        del gP, gQ
        pa, diffa = p + dP[ul], (qi + dQ[il]) - (qj + dQ[jl])
        ra = (pa * diffa).sum(1) + ((bi + db[il]) - (bj + db[jl]))
        sa = reg_adv * clipsig(ra)
        creg = 4.0 * reg / (B * d)
        c = s[:, None] * p + sa[:, None] * pa
        GP = rows((U, d), [(ul, -s[:, None] * diff - sa[:, None] * diffa + creg * p)])
        GQ = rows((I, d), [(il, -c + creg * qi), (jl, c + creg * qj)])
        Gb = rows((I,), [(il, -(s + sa) + creg * d * bi), (jl, (s + sa) + creg * d * bj)])
        want_loss = torch.stack([softplus_neg(r).sum(), softplus_neg(ra).sum(),
                                 ((p * p + qi * qi + qj * qj).sum(1) + d * (bi * bi + bj * bj)).sum() / (B * d)])
        del p, qi, qj, pa, diffa, diff, c
        want = []
        for wt, G in ((P, GP), (Q, GQ), (b, Gb)):
            acc = 0.1 + G * G
            want.append((wt.double() - lr * G / acc.sqrt(), acc))

        P1, Q1, b1 = P.clone(), Q.clone(), b.clone()
        acc = [torch.full_like(x, 0.1) for x in (P, Q, b)]
        dl = [torch.zeros_like(x) for x in (P, Q, b)]
        loss = torch.zeros(3, device=cuda, dtype=f64)
        d2.apr_adagrad_step(P1, Q1, b1, *acc, *dl, u, i, j, lr, reg, reg_adv, eps, loss=loss)
        d2.synchronize()

        def close(got, ref, rtol):
            err = (got.double() - ref).abs()
            scale = ref.abs().amax(dim=-1, keepdim=True) if ref.dim() == 2 else ref.abs().max()
            return bool((err <= 1e-7 + rtol * ref.abs() + rtol * scale).all())

        assert close(dl[0], dP, 1e-4) and close(dl[1], dQ, 1e-4) and close(dl[2], db, 1e-4)
        for got, got_acc, (ref, ref_acc) in zip((P1, Q1, b1), acc, want):
            assert close(got, ref, 2e-5) and close(got_acc, ref_acc, 1e-4)
        assert torch.allclose(loss, want_loss, rtol=1e-5, atol=0.0)
        assert abs(float(dl[2].double().pow(2).sum().sqrt()) - eps) <= 1e-5 * eps
        d2.close()


def test_full_size_dns_selection(dev):
    """Dynamic negative sampling at the large configuration's shape (2M items, d=128, 32 candidates per
    positive, model/DNSBPR.py:74-79): the chosen candidate is the first best-scored one under a float64
    scoring of all candidates; where the two best candidates are closer than float32 rounding either may win,
    but the chosen one always scores within rounding of the best."""
    import torch
    from tfrec_amd.device import Device
    cuda = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=cuda)
    with torch.cuda.stream(stream):
        d2 = Device(0, stream=stream.cuda_stream)
        U, I, d, n, C = 500_000, 2_000_000, 128, 1 << 18, 32
        g = torch.Generator(device=cuda); g.manual_seed(23)
        P = (torch.rand(U, d, device=cuda, generator=g) - 0.5) * 0.4
        Q = (torch.rand(I, d, device=cuda, generator=g) - 0.5) * 0.4
        b = torch.randn(I, device=cuda, generator=g) * 0.05
        users = torch.randint(0, U, (n,), device=cuda, generator=g, dtype=torch.int32)
        cand = torch.randint(0, I, (n, C), device=cuda, generator=g, dtype=torch.int32)
        cand[:, 5] = cand[:, 2]                                 # a repeated candidate: the first occurrence wins
        out = torch.empty(n, device=cuda, dtype=torch.int32)
        d2.dns_select(P, Q, b, users, cand, out=out)
        d2.synchronize()
        cl = cand.long()
        sc = torch.einsum("nd,ncd->nc", P[users.long()].double(), Q[cl].double()) + b[cl].double()
        best = sc.max(dim=1)
        chosen_pos = (cl == out.long().unsqueeze(1)).float().argmax(dim=1)       # first slot holding the chosen id
        assert bool((cl.gather(1, chosen_pos.unsqueeze(1)).squeeze(1) == out.long()).all())     # it is a candidate
        chosen_score = sc.gather(1, chosen_pos.unsqueeze(1)).squeeze(1)
        assert bool((best.values - chosen_score <= 2e-5).all())
        second = sc.clone().scatter_(1, best.indices.unsqueeze(1), float("-inf")).max(dim=1).values
        clear = best.values - second > 2e-5
        assert float(clear.float().mean()) > 0.9
        assert torch.equal(out.long()[clear], cl.gather(1, best.indices.unsqueeze(1)).squeeze(1)[clear])
        d2.close()
