"""GPU tests of the versioned user table (tfrec_utable) and the *_v steps: through the C ABI, bit for bit
against the plain steps on the same inputs (same arithmetic in the same order; only where the pre-step
user rows are read from differs)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from tfrec_amd.device import Device
    d = Device(0)
    yield d
    d.close()


def _problem(rng, U, I, d, B, steps):
    P = rng.uniform(-0.3, 0.3, (U, d)).astype(np.float32)
    Q = rng.uniform(-0.3, 0.3, (I, d)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, I).astype(np.float32)
    batches = [(rng.integers(0, U, B).astype(np.int32), (rng.zipf(1.3, B) % I).astype(np.int32), rng.integers(0, I, B).astype(np.int32))
               for _ in range(steps)]
    return P, Q, b, batches


@pytest.mark.parametrize("U,I,d,B", [(500, 400, 64, 6000), (3000, 700, 128, 5000), (200, 150, 20, 3000), (40000, 900, 32, 9000)])
def test_versioned_steps_equal_plain_steps(dev, U, I, d, B):
    """Consecutive steps over overlapping users (rows move back and forth between the two buffers), with and
    without the loss output; an empty batch; flatten restores the plain table."""
    rng = np.random.default_rng(U + d)
    P, Q, b, batches = _problem(rng, U, I, d, B, 4)
    a = [dev.array(x) for x in (P, Q, b)]
    c = [dev.array(x) for x in (P, Q, b)]
    ut = dev.user_table(c[0])
    la, lc = dev.zeros(2, np.float64), dev.zeros(2, np.float64)
    for s, (u, i, j) in enumerate(batches):
        du, di, dj = dev.array(u), dev.array(i), dev.array(j)
        want_loss = s % 2 == 0
        dev.bpr_sgd_step(*a, du, di, dj, 0.01, 0.05, la if want_loss else None)
        dev.bpr_sgd_step_v(ut, c[1], c[2], du, di, dj, 0.01, 0.05, lc if want_loss else None)
        dev.synchronize()
        assert np.array_equal(a[1].numpy(), c[1].numpy()) and np.array_equal(a[2].numpy(), c[2].numpy()), s
        assert np.array_equal(la.numpy(), lc.numpy())
    dev.bpr_sgd_step_v(ut, c[1], c[2], dev.array(batches[0][0][:0]), dev.array(batches[0][1][:0]), dev.array(batches[0][2][:0]), 0.01, 0.05)
    if U > B:       # (with every user in every batch, an even number of steps leaves all rows back in the first buffer)
        assert not np.array_equal(a[0].numpy(), c[0].numpy())       # some current rows live in the second buffer
    ut.flatten()
    dev.synchronize()
    assert np.array_equal(a[0].numpy(), c[0].numpy())
    # and the table keeps working after a flatten
    u, i, j = batches[1]
    dev.bpr_sgd_step(*a, dev.array(u), dev.array(i), dev.array(j), 0.01, 0.05)
    dev.bpr_sgd_step_v(ut, c[1], c[2], dev.array(u), dev.array(i), dev.array(j), 0.01, 0.05)
    ut.close()                                                   # flattens
    dev.synchronize()
    for x, y in zip(a, c):
        assert np.array_equal(x.numpy(), y.numpy())


def test_versioned_prepared_and_dense_gradient_steps(dev):
    """Prepared batches (one and several per preparation) and the replicated-item-table step on a versioned table."""
    rng = np.random.default_rng(8)
    U, I, d, B = 2000, 600, 64, 4096
    P, Q, b, batches = _problem(rng, U, I, d, B, 6)
    a = [dev.array(x) for x in (P, Q, b)]
    c = [dev.array(x) for x in (P, Q, b)]
    ut = dev.user_table(c[0])
    dev_b = [[dev.array(x) for x in bt] for bt in batches]
    dev.prepare_batch(0, U, I, *dev_b[0])
    for s in range(3):
        if s + 1 < 3:
            dev.prepare_batch((s + 1) & 1, U, I, *dev_b[s + 1])
        dev.bpr_sgd_step_prepared_v(s & 1, ut, c[1], c[2], 0.01, 0.05)
        dev.bpr_sgd_step(*a, *dev_b[s], 0.01, 0.05)
    grp = [dev.array(np.concatenate([batches[s][k] for s in (3, 4, 5)])) for k in range(3)]
    dev.prepare_batches(0, U, I, *grp, 3)
    for s in (3, 4, 5):
        dev.bpr_sgd_step_prepared_v(0, ut, c[1], c[2], 0.01, 0.05)
        dev.bpr_sgd_step(*a, *dev_b[s], 0.01, 0.05)
    # replicated item table: gradient sums into G, then the update
    G = dev.zeros(I * d + I, np.float32)
    dev.bpr_dense_grad_step_v(ut, c[1], c[2], G, *dev_b[0], 0.01, 0.05)
    dev.axpy(c[1], G.view(0, I * d), -0.01, n=I * d)
    dev.axpy(c[2], G.view(I * d, I * d + I), -0.01, n=I)
    dev.bpr_sgd_step(*a, *dev_b[0], 0.01, 0.05)
    ut.flatten()
    dev.synchronize()
    for x, y in zip(a, c):
        assert np.array_equal(x.numpy(), y.numpy())
    # bad arguments
    with pytest.raises(ValueError):
        dev.bpr_sgd_step_prepared_v(1, ut, c[1], c[2], 0.01, 0.05)          # nothing prepared
    u_bad = batches[0][0].copy(); u_bad[3] = U + 5
    dev.bpr_sgd_step_v(ut, c[1], c[2], dev.array(u_bad), dev_b[0][1], dev_b[0][2], 0.01, 0.05)
    with pytest.raises(IndexError):
        dev.synchronize()
    ut.close()
