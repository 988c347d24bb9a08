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
        # and once more with the index work prepared ahead on the auxiliary stream
        tr.prepare(0, *batches[0])
        tr.prepare(1, *batches[1])
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
