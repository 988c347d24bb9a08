"""GPU test of the sharded trainer / evaluator with the HIP backend on one rank (the exchange
steps degenerate to copies, the library entry points of the sharded path are the ones that run):
results must agree with the unsharded entry points."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("item_mode", ["sharded", "dense"])
def test_single_rank_sharded_path_matches_unsharded(oracle, item_mode):
    from tfrec_amd.sharded import ShardedBprTrainer, ShardedEvaluator, make_device
    dev = make_device(0)
    cuda = torch.device("cuda", 0)
    U, I, d, B, k = 5000, 3000, 64, 40000, 50
    tr = ShardedBprTrainer(dev, U, I, d, 0.01, 0.05, seed=2018, item_mode=item_mode)
    P0, Q0, b0 = [x.clone() for x in (tr.P, tr.Q, tr.b)]
    rng = np.random.default_rng(0)
    batches = [[torch.from_numpy(rng.integers(0, n, B).astype(np.int32)).to(cuda) for n in (U, I, I)] for _ in range(3)]
    for u, i, j in batches:
        tr.step(u, i, j)
    if item_mode == "dense":
        # and once more with the index work prepared ahead on the auxiliary stream: two different batches
        # back to back behind a busy main stream (the preparations read their index arrays only once that work
        # has drained, long after prepare() has returned), given as int64 (prepare casts)
        x = torch.randn(4096, 4096, device=cuda)
        for _ in range(30):
            x = (x @ x).clamp_(-1, 1)
        tr.prepare(0, *[t.long() for t in batches[0]])
        tr.prepare(1, *[t.long() for t in batches[1]])
        tr.step_prepared(0)
        tr.step_prepared(1)
        batches = batches + batches[:2]
    for u, i, j in batches:
        dev.bpr_sgd_step(P0, Q0, b0, u, i, j, 0.01, 0.05)
    dev.synchronize()
    for a, c in ((tr.P, P0), (tr.Q, Q0), (tr.b, b0)):
        if item_mode == "dense":     # same arithmetic in the same order on one rank: the same bits
            assert np.array_equal(a.cpu().numpy(), c.cpu().numpy())
        else:
            np.testing.assert_allclose(a.cpu().numpy(), c.cpu().numpy(), rtol=1e-5, atol=1e-7)
    # the shard initialiser reproduces the whole-table stream
    whole = dev.uniform((U, d), -0.05, 0.05, 2018 * 2 + 1).numpy()
    part = ShardedBprTrainer(dev, U, I, d, 0.01, 0.05, seed=2018, item_mode=item_mode)
    assert np.array_equal(part.P.cpu().numpy(), whole)

    deg = 12
    tptr = torch.arange(0, deg * (U + 1), deg, dtype=torch.int64, device=cuda)
    tidx = torch.from_numpy(np.sort(rng.integers(0, I, (U, deg)), axis=1).astype(np.int32).ravel()).to(cuda)
    eptr = torch.arange(0, 2 * (U + 1), 2, dtype=torch.int64, device=cuda)
    eidx = torch.from_numpy(np.sort(rng.integers(0, I, (U, 2)), axis=1).astype(np.int32).ravel()).to(cuda)
    users = torch.arange(0, U, 3, dtype=torch.int32, device=cuda)
    ev = ShardedEvaluator(dev, tr, (tptr, tidx), (eptr, eidx), users, top_k=k, chunk_users=700)
    m = ev.evaluate().cpu().numpy()
    ref = dev.eval_fullrank(tr.P, tr.Q, tr.b, users, (tptr, tidx), (eptr, eidx), k, mean_mode=0)
    np.testing.assert_allclose(m, ref, rtol=1e-6, atol=1e-7)
    dev.close()


@pytest.mark.parametrize("item_mode", ["sharded", "dense"])
@pytest.mark.parametrize("U,I,d,sizes", [(1, 1, 4, [0, 1, 0, 3]), (3, 2, 8, [0, 0, 5]), (50, 40, 64, [0, 7, 0]), (2, 300, 16, [1, 0, 2])])
def test_empty_and_tiny_batches_on_tiny_tables(item_mode, U, I, d, sizes):
    """A rank's part of a batch may be empty (its users drew nothing) and tables may have a single row: the sharded step
    then does nothing, or what the plain step does."""
    from tfrec_amd.sharded import ShardedBprTrainer, make_device
    dev = make_device(0)
    cuda = torch.device("cuda", 0)
    rng = np.random.default_rng(U + I)
    tr = ShardedBprTrainer(dev, U, I, d, 0.01, 0.05, seed=2018, item_mode=item_mode)
    P0, Q0, b0 = [x.clone() for x in (tr.P, tr.Q, tr.b)]
    for B in sizes:
        u, i, j = [torch.from_numpy(rng.integers(0, n, B).astype(np.int32)).to(cuda) for n in (U, I, I)]
        tr.step(u, i, j)
        if B:
            dev.bpr_sgd_step(P0, Q0, b0, u, i, j, 0.01, 0.05)
    dev.synchronize()
    for a, c in ((tr.P, P0), (tr.Q, Q0), (tr.b, b0)):
        np.testing.assert_allclose(a.cpu().numpy(), c.cpu().numpy(), rtol=1e-5, atol=1e-7)
    dev.close()


@pytest.mark.parametrize("K", [2, 5, 32])
def test_item_half_in_ranges_of_item_ids_equals_the_whole(K):
    """The replicated-table step with its item half cut into K ranges of item ids (what the overlapped all-reduce
    runs on): same parameters as the uncut step, up to the association of the sums; batches large and small
    (long and short tiles), prepared and not, hot items, ranges that receive nothing."""
    from tfrec_amd.sharded import ShardedBprTrainer, make_device
    dev = make_device(0)
    cuda = torch.device("cuda", 0)
    U, I, d = 3000, 1003, 64
    # (a small learning rate: the hottest item collects tens of thousands of summed gradients per step)
    one = ShardedBprTrainer(dev, U, I, d, 0.0005, 0.05, seed=11, item_mode="dense", overlap_chunks=1)
    cut = ShardedBprTrainer(dev, U, I, d, 0.0005, 0.05, seed=11, item_mode="dense", overlap_chunks=K)
    assert cut.overlap_chunks == K and one.overlap_chunks == 1
    rng = np.random.default_rng(K)
    batches = []
    for B, item_hi in ((200000, I), (300, I), (50000, 40), (0, I), (1, I), (150000, I)):       # (one batch is empty)
        u = rng.integers(0, U, B).astype(np.int32)
        i = np.minimum((rng.pareto(1.0, B) * 3).astype(np.int64), item_hi - 1).astype(np.int32)      # a few very hot items
        j = rng.integers(0, item_hi, B).astype(np.int32)
        batches.append([torch.from_numpy(x).to(cuda) for x in (u, i, j)])
    for tr in (one, cut):
        for bt in batches[:5]:
            tr.step(*bt)
        tr.prepare(0, *batches[5])
        tr.step_prepared(0, prepare_next=(1,) + tuple(batches[0]))
        tr.step_prepared(1)
    torch.cuda.synchronize()
    for a, c in ((cut.P, one.P), (cut.Q_full, one.Q_full), (cut.b_full, one.b_full)):
        np.testing.assert_allclose(a.cpu().numpy(), c.cpu().numpy(), rtol=2e-5, atol=1e-6)
    dev.close()


def test_chunked_step_rejects_bad_event_lists():
    """The chunked entry point refuses an empty or oversized event list and a null event, and says why."""
    from tfrec_amd.sharded import make_device
    dev = make_device(0)
    cuda = torch.device("cuda", 0)
    U, I, d, B = 50, 40, 8, 30
    P = torch.zeros(U, d, device=cuda); Q = torch.zeros(I, d, device=cuda); b = torch.zeros(I, device=cuda)
    G = torch.zeros(I * (d + 1), device=cuda)
    u = torch.zeros(B, dtype=torch.int32, device=cuda)
    ut = dev.user_table(P)
    ev = torch.cuda.Event(); ev.record()
    for events in ([], [ev.cuda_event] * 33, [0]):
        with pytest.raises(ValueError):
            dev.bpr_dense_grad_step_chunked(ut, Q, b, G, u, u, u, 0.01, 0.05, events)
    dev.bpr_dense_grad_step_chunked(ut, Q, b, G, u, u, u, 0.01, 0.05, [ev.cuda_event])      # one range = the unchunked step
    dev.synchronize()
    ut.close()
    dev.close()


def _two_rank_worker(rank, world, port, out_dir):
    """Two ranks share the one GPU of the test box; gloo moves the tensors (RCCL needs a GPU per rank)."""
    import os
    import sys
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from tfrec_amd.sharded import ShardedBprTrainer, ShardedEvaluator, make_device
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    dev = make_device(0)
    cuda = torch.device("cuda", 0)
    U, I, d, B, k = 4000, 1500, 64, 30000, 20
    # (the item half in three ranges of item ids, each range's rows all-reduced on a second stream as it completes)
    tr = ShardedBprTrainer(dev, U, I, d, 0.01, 0.05, seed=7, item_mode="dense", overlap_chunks=3)
    assert tr.overlap_chunks == 3
    rng = np.random.default_rng(1)
    lo, hi = tr.user_range
    batches = []
    for _ in range(3):
        u, i, j = rng.integers(0, U, B).astype(np.int32), rng.integers(0, I, B).astype(np.int32), rng.integers(0, I, B).astype(np.int32)
        m = (u >= lo) & (u < hi)
        batches.append([torch.from_numpy(x[m]).to(cuda) for x in (u, i, j)])
    tr.prepare(0, *batches[0])
    for s in range(3):
        nxt = ((s + 1) & 1,) + tuple(batches[s + 1]) if s + 1 < 3 else None
        tr.step_prepared(s & 1, prepare_next=nxt)
    P, Q, b = tr.gather_full()
    deg = 6
    tptr = torch.arange(0, deg * (U + 1), deg, dtype=torch.int64, device=cuda)
    tidx = torch.from_numpy(np.sort(rng.integers(0, I, (U, deg)), axis=1).astype(np.int32).ravel()).to(cuda)
    eptr = torch.arange(0, 2 * (U + 1), 2, dtype=torch.int64, device=cuda)
    eidx = torch.from_numpy(np.sort(rng.integers(0, I, (U, 2)), axis=1).astype(np.int32).ravel()).to(cuda)
    users = torch.arange(0, U, 5, dtype=torch.int32, device=cuda)
    m = ShardedEvaluator(dev, tr, (tptr, tidx), (eptr, eidx), users, top_k=k, chunk_users=300).evaluate()
    m_items = ShardedEvaluator(dev, tr, (tptr, tidx), (eptr, eidx), users, top_k=k, chunk_users=300, layout="items").evaluate()
    dev.synchronize()
    if rank == 0:
        np.savez(os.path.join(out_dir, "two.npz"), P=P.cpu().numpy(), Q=Q.cpu().numpy(), b=b.cpu().numpy(), m=m.cpu().numpy(),
                 m_items=m_items.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()
    dev.close()


def test_two_ranks_dense_mode_match_one_rank(tmp_path):
    """World size 2 with the HIP backend on both ranks: training and evaluation equal the unsharded run."""
    import socket
    import torch.multiprocessing as mp
    from tfrec_amd.device import Device
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = np.load(str(tmp_path / "two.npz"))
    dev = Device(0)
    U, I, d, B, k = 4000, 1500, 64, 30000, 20
    P = dev.uniform((U, d), -0.05, 0.05, 7 * 2 + 1)
    Q = dev.uniform((I, d), -0.05, 0.05, 7 * 2 + 2)
    b = dev.zeros(I, np.float32)
    rng = np.random.default_rng(1)
    for _ in range(3):
        u, i, j = rng.integers(0, U, B).astype(np.int32), rng.integers(0, I, B).astype(np.int32), rng.integers(0, I, B).astype(np.int32)
        dev.bpr_sgd_step(P, Q, b, dev.array(u, np.int32), dev.array(i, np.int32), dev.array(j, np.int32), 0.01, 0.05)
    dev.synchronize()
    np.testing.assert_allclose(r["P"], P.numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(r["Q"], Q.numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(r["b"], b.numpy(), rtol=1e-5, atol=1e-7)
    deg = 6
    tptr = np.arange(0, deg * (U + 1), deg, dtype=np.int64)
    tidx = np.sort(rng.integers(0, I, (U, deg)), axis=1).astype(np.int32).ravel()
    eptr = np.arange(0, 2 * (U + 1), 2, dtype=np.int64)
    eidx = np.sort(rng.integers(0, I, (U, 2)), axis=1).astype(np.int32).ravel()
    users = np.arange(0, U, 5, dtype=np.int32)
    # evaluate the two-rank model unsharded: the item-sharded merge must give the same means
    P2, Q2, b2 = dev.array(r["P"]), dev.array(r["Q"]), dev.array(r["b"])
    ref = dev.eval_fullrank(P2, Q2, b2, dev.array(users, np.int32), (dev.array(tptr, np.int64), dev.array(tidx, np.int32)),
                            (dev.array(eptr, np.int64), dev.array(eidx, np.int32)), k, mean_mode=0)
    np.testing.assert_allclose(r["m"], ref, rtol=1e-6, atol=1e-7)            # user-sharded (replicated item table)
    np.testing.assert_allclose(r["m_items"], ref, rtol=1e-6, atol=1e-7)      # item-sharded: local top-k, gather, merge
    dev.close()


def _one_rank_rccl_worker(rank, world, port, out_dir):
    """A process group of ONE rank over RCCL (backend "nccl") on the test GPU: every collective of the trainers and
    the evaluator runs through RCCL (trivially), on the streams and tensors it meets in a many-GPU run."""
    import os
    import sys
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from tfrec_amd.sharded import ShardedBprTrainer, ShardedEvaluator, make_device
    dev = make_device(0)
    cuda = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=cuda)
    U, I, d, B, k = 3000, 1100, 64, 25000, 20
    rng = np.random.default_rng(4)
    batches = [[torch.from_numpy(rng.integers(0, n, B).astype(np.int32)).to(cuda) for n in (U, I, I)] for _ in range(3)]
    out = {}
    for mode, chunks in (("dense", 3), ("sharded", None)):
        tr = ShardedBprTrainer(dev, U, I, d, 0.01, 0.05, seed=9, item_mode=mode, overlap_chunks=chunks)
        assert tr._coll and tr.world == 1
        if mode == "dense":
            assert tr.overlap_chunks == 3
            tr.prepare(0, *batches[0])
            for s in range(3):
                tr.step_prepared(s & 1, prepare_next=(((s + 1) & 1,) + tuple(batches[s + 1])) if s + 1 < 3 else None)
        else:
            for plan in tr.plan_steps(batches):
                tr.step_planned(plan)
        P, Q, b = tr.gather_full()
        deg = 5
        tptr = torch.arange(0, deg * (U + 1), deg, dtype=torch.int64, device=cuda)
        tidx = torch.from_numpy(np.sort(np.random.default_rng(5).integers(0, I, (U, deg)), axis=1).astype(np.int32).ravel()).to(cuda)
        eptr = torch.arange(0, 2 * (U + 1), 2, dtype=torch.int64, device=cuda)
        eidx = torch.from_numpy(np.sort(np.random.default_rng(6).integers(0, I, (U, 2)), axis=1).astype(np.int32).ravel()).to(cuda)
        users = torch.arange(0, U, 4, dtype=torch.int32, device=cuda)
        m = ShardedEvaluator(dev, tr, (tptr, tidx), (eptr, eidx), users, top_k=k, chunk_users=200).evaluate()
        torch.cuda.synchronize()
        out.update({mode + "_P": P.cpu().numpy(), mode + "_Q": Q.cpu().numpy(), mode + "_b": b.cpu().numpy(), mode + "_m": m.cpu().numpy()})
    # the APR step (two all-reduces of the item gradient table per step, the loss summed over the ranks)
    from test_gpu_sharded import _apr_problem, APR_HP
    from tfrec_amd.sharded import ShardedAprTrainer
    params, abatches = _apr_problem(U, I, d, 8000, 2, 5)
    apr = ShardedAprTrainer(dev, U, I, d, APR_HP["lr"], APR_HP["reg"], APR_HP["reg_adv"], APR_HP["eps"], params=params)
    loss = torch.zeros(3, dtype=torch.float64, device=cuda)
    for u, i, j in abatches:
        apr.step(*(torch.from_numpy(x).to(cuda) for x in (u, i, j)), loss=loss)
    aP, aQ, ab = apr.gather_full()
    torch.cuda.synchronize()
    out.update({"apr_P": aP.cpu().numpy(), "apr_Q": aQ.cpu().numpy(), "apr_b": ab.cpu().numpy(), "apr_loss": loss.cpu().numpy()})
    np.savez(os.path.join(out_dir, "rccl1.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()
    dev.close()


def test_one_rank_process_group_over_rccl(tmp_path):
    """Both layouts, the overlapped all-reduce and the evaluator with RCCL as the backend (one rank): the results of
    the plain single-GPU entry points."""
    import socket
    import torch.multiprocessing as mp
    from tfrec_amd.device import Device
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_one_rank_rccl_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    r = np.load(str(tmp_path / "rccl1.npz"))
    dev = Device(0)
    U, I, d, B, k = 3000, 1100, 64, 25000, 20
    P = dev.uniform((U, d), -0.05, 0.05, 9 * 2 + 1)
    Q = dev.uniform((I, d), -0.05, 0.05, 9 * 2 + 2)
    b = dev.zeros(I, np.float32)
    rng = np.random.default_rng(4)
    for _ in range(3):
        u, i, j = [dev.array(rng.integers(0, n, B).astype(np.int32), np.int32) for n in (U, I, I)]
        dev.bpr_sgd_step(P, Q, b, u, i, j, 0.01, 0.05)
    dev.synchronize()
    for mode in ("dense", "sharded"):
        np.testing.assert_allclose(r[mode + "_P"], P.numpy(), rtol=1e-5, atol=1e-7, err_msg=mode)
        np.testing.assert_allclose(r[mode + "_Q"], Q.numpy(), rtol=1e-5, atol=1e-7, err_msg=mode)
        np.testing.assert_allclose(r[mode + "_b"], b.numpy(), rtol=1e-5, atol=1e-7, err_msg=mode)
    np.testing.assert_allclose(r["dense_m"], r["sharded_m"], rtol=1e-5, atol=1e-7)
    assert r["dense_m"].shape == (5, k) and np.isfinite(r["dense_m"]).all()
    # APR: the single-GPU entry point on the same batches
    params, abatches = _apr_problem(U, I, d, 8000, 2, 5)
    aP, aQ, ab = (dev.array(x.copy()) for x in params)
    acc = [dev.full(x.shape, 0.1) for x in params]
    dl = [dev.zeros(x.shape, np.float32) for x in params]
    ref_loss = dev.zeros(3, np.float64)
    for u, i, j in abatches:
        dev.apr_adagrad_step(aP, aQ, ab, *acc, *dl, dev.array(u, np.int32), dev.array(i, np.int32), dev.array(j, np.int32),
                             APR_HP["lr"], APR_HP["reg"], APR_HP["reg_adv"], APR_HP["eps"], loss=ref_loss)
    dev.synchronize()
    for got, want in ((r["apr_P"], aP), (r["apr_Q"], aQ), (r["apr_b"], ab)):
        np.testing.assert_allclose(got, want.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r["apr_loss"], ref_loss.numpy(), rtol=1e-6)
    dev.close()


@pytest.mark.parametrize("item_mode", ["dense", "sharded"])
def test_sharded_epoch_sampler_on_the_gpu(oracle, item_mode):
    """ShardedEpochSampler with the HIP backend (one rank): DNS pools, picks and one collective epoch
    equal the oracle's statement of the same epoch."""
    import scipy.sparse as sp
    from tfrec_amd.sharded import ShardedBprTrainer, ShardedEpochSampler, make_device
    dev = make_device(0)
    U, I, d, C, B = 400, 300, 32, 6, 256
    rng = np.random.default_rng(2)
    rows, cols = [], []
    for u in range(U):
        c = rng.permutation(I)[: int(rng.integers(0, 12))]
        rows += [u] * len(c); cols += c.tolist()
    from tfrec_amd.utils.tools import csr_trained_rows
    m = csr_trained_rows(sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(U, I)))
    tr = ShardedBprTrainer(dev, U, I, d, 0.05, 0.01, seed=3, item_mode=item_mode)
    P, Q, b = [x.cpu().numpy().copy() for x in tr.gather_full()]
    s = ShardedEpochSampler(dev, tr, m, seed=17, n_neg=C)
    u, p, n = [x.cpu().numpy() for x in s.draw(0)]
    wu, wp, pools, _ = oracle.sample_negatives(m.indptr.astype(np.int64), m.indices.astype(np.int32), U, I, C, 17, 0)
    want, margin = oracle.dns_select(P, Q, b, wu, pools)
    assert np.array_equal(u, wu) and np.array_equal(p, wp)
    ok = margin > 1e-6
    assert ok.mean() > 0.9 and np.array_equal(n[ok], want[ok])
    eu, ep, en = [x.cpu().numpy() for x in s.epoch(0)]
    perm = oracle.shuffle_perm(m.nnz, 17, 1)
    assert np.array_equal(eu, u[perm]) and np.array_equal(ep, p[perm]) and np.array_equal(en, n[perm])
    steps = s.train_epoch(0, B)
    assert steps == (m.nnz + B - 1) // B
    ws = oracle.Workspace(U, I, d)
    for lo in range(0, m.nnz, B):
        oracle.bpr_step(P, Q, b, eu[lo:lo + B], ep[lo:lo + B], en[lo:lo + B], 0.05, 0.01, ws=ws)
    P1, Q1, b1 = [x.cpu().numpy() for x in tr.gather_full()]
    for a, c in ((P1, P), (Q1, Q), (b1, b)):
        np.testing.assert_allclose(a, c, rtol=1e-4, atol=1e-6)
    dev.close()


def _apr_problem(U, I, d, B, steps, seed):
    rng = np.random.default_rng(seed)
    P = rng.uniform(-0.3, 0.3, (U, d)).astype(np.float32)
    Q = rng.uniform(-0.3, 0.3, (I, d)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, I).astype(np.float32)
    items = (rng.zipf(1.3, (steps, B)) % I).astype(np.int32)         # hot items: long duplicate runs
    batches = [(rng.integers(0, U, B).astype(np.int32), items[s], rng.integers(0, I, B).astype(np.int32)) for s in range(steps)]
    return (P, Q, b), batches


APR_HP = dict(lr=0.05, reg=0.01, reg_adv=1.0, eps=0.05)


def test_single_rank_apr_matches_unsharded_step(oracle):
    """The four-call replicated-item APR step on one rank against tfrec_apr_adagrad_step and the oracle."""
    from tfrec_amd.sharded import ShardedAprTrainer, make_device
    dev = make_device(0)
    cuda = torch.device("cuda", 0)
    U, I, d, B, steps = 3000, 2000, 64, 20000, 3
    params, batches = _apr_problem(U, I, d, B, steps, 5)
    tr = ShardedAprTrainer(dev, U, I, d, APR_HP["lr"], APR_HP["reg"], APR_HP["reg_adv"], APR_HP["eps"], params=params)
    P, Q, b = (dev.array(x.copy()) for x in params)
    accP, accQ, accb = (dev.full(x.shape, 0.1) for x in params)
    dP, dQ, db = (dev.zeros(x.shape, np.float32) for x in params)
    oP, oQ, ob = (x.copy() for x in params)
    oaP, oaQ, oab = (np.full_like(x, 0.1) for x in params)
    ows = oracle.Workspace(U, I, d)
    loss = torch.zeros(3, dtype=torch.float64, device=cuda)
    ref_loss = dev.zeros(3, np.float64)
    for u, i, j in batches:
        tu, ti, tj = (torch.from_numpy(x).to(cuda) for x in (u, i, j))
        tr.step(tu, ti, tj, global_batch=B, loss=loss)
        dev.apr_adagrad_step(P, Q, b, accP, accQ, accb, dP, dQ, db, dev.array(u, np.int32), dev.array(i, np.int32), dev.array(j, np.int32),
                             APR_HP["lr"], APR_HP["reg"], APR_HP["reg_adv"], APR_HP["eps"], loss=ref_loss)
        want = oracle.apr_step(oP, oQ, ob, oaP, oaQ, oab, u, i, j, APR_HP["lr"], APR_HP["reg"], APR_HP["reg_adv"], APR_HP["eps"], ws=ows)
        dev.synchronize()
        np.testing.assert_allclose(loss.cpu().numpy(), ref_loss.numpy(), rtol=1e-6)
        np.testing.assert_allclose(loss.cpu().numpy(), np.array(want), rtol=1e-5)
    for got, unsharded, orc in ((tr.P, P, oP), (tr.Q_full, Q, oQ), (tr.b_full, b, ob), (tr.accP, accP, oaP),
                                (tr._accQb[: I * d].view(I, d), accQ, oaQ), (tr._accQb[I * d:], accb, oab)):
        g = got.cpu().numpy()
        np.testing.assert_allclose(g, unsharded.numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(g, orc, rtol=1e-4, atol=1e-6)
    # phase two refuses to run without the phase one of the same batch before it
    with pytest.raises(Exception):
        dev.apr_dense_adagrad_grads(tr.P, tr.Q_full, tr.b_full, tr.accP, tr.dP, tr.dQ, tr.db, tr._G, B, B, 0.05, 0.0, 1.0)
    dev.close()


def _two_rank_apr_worker(rank, world, port, out_dir):
    import os
    import sys
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from test_gpu_sharded import _apr_problem, APR_HP
    from tfrec_amd.sharded import ShardedAprTrainer, make_device
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    dev = make_device(0)
    cuda = torch.device("cuda", 0)
    U, I, d, B, steps = 3000, 2000, 64, 20000, 3
    params, batches = _apr_problem(U, I, d, B, steps, 5)
    tr = ShardedAprTrainer(dev, U, I, d, APR_HP["lr"], APR_HP["reg"], APR_HP["reg_adv"], APR_HP["eps"], params=params)
    lo, hi = tr.user_range
    losses = []
    for u, i, j in batches:
        m = (u >= lo) & (u < hi)
        loss = torch.zeros(3, dtype=torch.float64, device=cuda)
        tr.step(*(torch.from_numpy(x[m]).to(cuda) for x in (u, i, j)), loss=loss)
        losses.append(loss.cpu().numpy())
    P, Q, b = tr.gather_full()
    dev.synchronize()
    if rank == 0:
        np.savez(os.path.join(out_dir, "apr2.npz"), P=P.cpu().numpy(), Q=Q.cpu().numpy(), b=b.cpu().numpy(), loss=np.array(losses))
    dist.barrier()
    dist.destroy_process_group()
    dev.close()


def test_two_ranks_apr_match_one_rank(tmp_path):
    """World size 2 (both ranks on the one GPU, gloo between them): the APR steps equal the unsharded ones."""
    import socket
    import torch.multiprocessing as mp
    from tfrec_amd.device import Device
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_two_rank_apr_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = np.load(str(tmp_path / "apr2.npz"))
    dev = Device(0)
    U, I, d, B, steps = 3000, 2000, 64, 20000, 3
    params, batches = _apr_problem(U, I, d, B, steps, 5)
    P, Q, b = (dev.array(x.copy()) for x in params)
    accP, accQ, accb = (dev.full(x.shape, 0.1) for x in params)
    dP, dQ, db = (dev.zeros(x.shape, np.float32) for x in params)
    loss = dev.zeros(3, np.float64)
    for n, (u, i, j) in enumerate(batches):
        dev.apr_adagrad_step(P, Q, b, accP, accQ, accb, dP, dQ, db, dev.array(u, np.int32), dev.array(i, np.int32), dev.array(j, np.int32),
                             APR_HP["lr"], APR_HP["reg"], APR_HP["reg_adv"], APR_HP["eps"], loss=loss)
        np.testing.assert_allclose(r["loss"][n], loss.numpy(), rtol=1e-5)
    np.testing.assert_allclose(r["P"], P.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(r["Q"], Q.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(r["b"], b.numpy(), rtol=1e-4, atol=1e-6)
    dev.close()


# ---- the large configuration's layout: every table row-sharded, DNS pools of 32, item-sharded evaluation ------
C4 = dict(U=6000, I=4000, d=128, C=32, B=8192, k=100, seed=29)


def _c4_matrix():
    import scipy.sparse as sp
    rng = np.random.default_rng(C4["seed"])
    rows, cols = [], []
    for u in range(C4["U"]):
        c = rng.permutation(C4["I"])[: int(rng.integers(1, 14))]
        rows += [u] * len(c); cols += c.tolist()
    from tfrec_amd.utils.tools import csr_trained_rows
    return csr_trained_rows(sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(C4["U"], C4["I"])))


def _c4_eval_inputs(cuda):
    rng = np.random.default_rng(C4["seed"] + 1)
    U, I = C4["U"], C4["I"]
    m = _c4_matrix()
    tptr = torch.from_numpy(m.indptr.astype(np.int64)).to(cuda)
    tidx = torch.from_numpy(m.indices.astype(np.int32)).to(cuda)
    eptr = torch.arange(0, 3 * (U + 1), 3, dtype=torch.int64, device=cuda)
    eidx = torch.from_numpy(np.sort(rng.integers(0, I, (U, 3)), axis=1).astype(np.int32).ravel()).to(cuda)
    users = torch.arange(0, U, 2, dtype=torch.int32, device=cuda)
    return (tptr, tidx), (eptr, eidx), users


def _two_rank_c4_worker(rank, world, port, out_dir):
    import os
    import sys
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=True)      # a stuck collective names itself instead of hanging the run
    from test_gpu_sharded import C4, _c4_matrix, _c4_eval_inputs
    from tfrec_amd.device import SCORE_BF16
    from tfrec_amd.sharded import ShardedBprTrainer, ShardedEpochSampler, ShardedEvaluator, make_device
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    dev = make_device(0)
    cuda = torch.device("cuda", 0)
    tr = ShardedBprTrainer(dev, C4["U"], C4["I"], C4["d"], 0.05, 0.01, seed=C4["seed"], item_mode="sharded")
    s = ShardedEpochSampler(dev, tr, _c4_matrix(), seed=C4["seed"], n_neg=C4["C"])
    start = [x.cpu().numpy() for x in tr.gather_full()]
    lists = []
    for e in range(2):
        eu, ep, en = s.epoch(e)          # what train_epoch draws (a pure function of the epoch and the model at its start)
        lists.append(np.stack([x.cpu().numpy() for x in (eu, ep, en)]) if s.nnz else np.zeros((3, 0), np.int32))
        s.train_epoch(e, C4["B"])
    P, Q, b = tr.gather_full()
    train, test, users = _c4_eval_inputs(cuda)
    m32 = ShardedEvaluator(dev, tr, train, test, users, top_k=C4["k"], chunk_users=1000).evaluate()
    m16 = ShardedEvaluator(dev, tr, train, test, users, top_k=C4["k"], chunk_users=1000, mode=SCORE_BF16).evaluate()
    dev.synchronize()
    np.savez(os.path.join(out_dir, "c4_r%d.npz" % rank), P=P.cpu().numpy(), Q=Q.cpu().numpy(), b=b.cpu().numpy(),
             P0=start[0], Q0=start[1], b0=start[2], l0=lists[0], l1=lists[1], m32=m32.cpu().numpy(), m16=m16.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()
    dev.close()


def test_two_ranks_all_tables_sharded_dns_epochs_and_item_sharded_eval(tmp_path, oracle):
    """World size 2 with the HIP backend on both ranks (one GPU, gloo between them), every table row-sharded
    (all-to-all lookups), DNS pools of 32 candidates, d=128, item-sharded evaluation at k=100: the epoch lists are
    the whole matrix's draws with the best candidate under the epoch-start model, the parameters after two
    collective epochs are those of one GPU stepping over the union of the ranks' batches, and the sharded
    metrics are those of the unsharded evaluation of the same model."""
    import socket
    import torch.multiprocessing as mp
    from tfrec_amd.device import Device, SCORE_BF16
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_two_rank_c4_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    rs = [np.load(str(tmp_path / ("c4_r%d.npz" % r))) for r in range(2)]
    U, I, d, C, B, k = (C4[x] for x in ("U", "I", "d", "C", "B", "k"))
    dev = Device(0)
    # epoch 0: pools drawn for the whole matrix, best candidate under the initial model (same kernel arithmetic)
    m = _c4_matrix()
    wu, wp, pools, _ = oracle.sample_negatives(m.indptr.astype(np.int64), m.indices.astype(np.int32), U, I, C, C4["seed"], 0)
    P0, Q0, b0 = (dev.array(rs[0][x]) for x in ("P0", "Q0", "b0"))
    want = dev.dns_select(P0, Q0, b0, dev.array(wu, np.int32), dev.array(pools, np.int32)).numpy()
    got = np.concatenate([r["l0"] for r in rs], axis=1)
    key = lambda a: np.lexsort((a[2], a[1], a[0]))
    ref = np.stack([wu, wp, want])
    assert got.shape == ref.shape
    assert np.array_equal(got[:, key(got)], ref[:, key(ref)])
    # the two epochs replayed on one GPU over the union of the ranks' batches
    P, Q, b = P0, Q0, b0
    for e in range(2):
        lists = [r["l%d" % e] for r in rs]
        steps = max((l.shape[1] + B - 1) // B for l in lists)
        for st in range(steps):
            part = [l[:, st * B:(st + 1) * B] for l in lists]
            uu, ii, jj = (np.ascontiguousarray(np.concatenate([x[c] for x in part]), dtype=np.int32) for c in range(3))
            dev.bpr_sgd_step(P, Q, b, dev.array(uu), dev.array(ii), dev.array(jj), 0.05, 0.01)
    dev.synchronize()
    for name, t in (("P", P), ("Q", Q), ("b", b)):
        np.testing.assert_allclose(rs[0][name], t.numpy(), rtol=1e-5, atol=1e-7, err_msg=name)
        assert np.array_equal(rs[0][name], rs[1][name])
    # the sharded metrics against the unsharded evaluation of the two-rank model
    cuda = torch.device("cuda", 0)
    train, test, users = _c4_eval_inputs(cuda)
    P2, Q2, b2 = (torch.from_numpy(rs[0][x]).to(cuda) for x in ("P", "Q", "b"))
    for mode, name in ((0, "m32"), (SCORE_BF16, "m16")):
        ref = dev.eval_fullrank(P2, Q2, b2, users, train, test, k, mode=mode, mean_mode=0)
        np.testing.assert_allclose(rs[0][name], ref, rtol=1e-6, atol=1e-7, err_msg=name)
        assert np.array_equal(rs[0][name], rs[1][name])
    dev.close()


@pytest.mark.parametrize("I,B,W", [(1000, 300, 3), (50000, 40000, 8), (4097, 9000, 2), (7, 5, 1)])
def test_items_dedup_and_rows_gather(I, B, W):
    """The requester's de-duplication (distinct items ascending, slots, per-owner counts) and the owner's row
    gather against numpy."""
    from tfrec_amd.device import Device
    dev = Device(0)
    rng = np.random.default_rng(I + B)
    pos = (rng.zipf(1.2, B) % I).astype(np.int32)
    neg = rng.integers(0, I, B).astype(np.int32)
    per = (I + W - 1) // W
    uniq, ps, ns = dev.empty(min(2 * B, I), np.int32), dev.empty(B, np.int32), dev.empty(B, np.int32)
    counts, total = dev.empty(W, np.int64), dev.empty(1, np.int64)
    dev.items_dedup(dev.array(pos), dev.array(neg), I, per, W, uniq, ps, ns, counts, total)
    dev.synchronize()
    want, inv = np.unique(np.concatenate([pos, neg]), return_inverse=True)
    n = int(total.numpy()[0])
    assert n == len(want) and np.array_equal(uniq.numpy()[:n], want)
    assert np.array_equal(ps.numpy(), inv[:B]) and np.array_equal(ns.numpy(), inv[B:])
    assert np.array_equal(counts.numpy(), np.bincount(want // per, minlength=W))
    d = 20
    Q = rng.standard_normal((I, d)).astype(np.float32)
    b = rng.standard_normal(I).astype(np.float32)
    rows, bias = dev.empty((n, d), np.float32), dev.empty(n, np.float32)
    dev.rows_gather(dev.array(Q), dev.array(b), dev.array(want.astype(np.int32)), rows, bias)
    dev.synchronize()
    assert np.array_equal(rows.numpy(), Q[want]) and np.array_equal(bias.numpy(), b[want])
    dev.close()
