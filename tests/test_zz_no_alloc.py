"""GPU test of tfrec_bpr_sgd_step_noalloc (the step whose device scratch is a fixed array inside the library):
through the C ABI, against tfrec_bpr_sgd_step on the same inputs, bit for bit, on two calls in a row, the second
with a smaller batch; oversized batches are refused before anything runs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problem(rng, U, I, d, B):
    P = rng.uniform(-0.3, 0.3, (U, d)).astype(np.float32)
    Q = rng.uniform(-0.3, 0.3, (I, d)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, I).astype(np.float32)
    u = rng.integers(0, U, B).astype(np.int32)
    i = (rng.zipf(1.3, B) % I).astype(np.int32)          # hot items: long duplicate runs, partial sums, fix-ups
    j = rng.integers(0, I, B).astype(np.int32)
    return P, Q, b, u, i, j


@pytest.mark.parametrize("d", [64, 128, 20])
def test_no_allocation_step_equals_the_plain_step(d):
    from tfrec_amd.device import Device
    dev = Device(0)
    rng = np.random.default_rng(3 + d)
    U, I = 3000, 1200
    ws_before = None
    for call, B in enumerate((20000, 512, 7000)):      # a large batch first, then smaller ones over its leftovers
        P, Q, b, u, i, j = _problem(rng, U, I, d, B)
        du, di, dj = dev.array(u), dev.array(i), dev.array(j)
        a = [dev.array(x) for x in (P, Q, b)]
        c = [dev.array(x) for x in (P, Q, b)]
        la, lc = dev.zeros(2, np.float64), dev.zeros(2, np.float64)
        dev.bpr_sgd_step(*a, du, di, dj, 0.01, 0.05, la)
        dev.synchronize()
        if ws_before is None:
            ws_before = dev.workspace_bytes()
        dev.bpr_sgd_step_noalloc(*c, du, di, dj, 0.01, 0.05, lc)
        dev.synchronize()
        for x, y in zip(a, c):
            assert np.array_equal(x.numpy(), y.numpy()), (d, call)
        assert np.array_equal(la.numpy(), lc.numpy())
    assert dev.workspace_bytes() == ws_before            # the context's own workspace was neither used nor grown
    # without the loss output, and an empty batch
    P, Q, b, u, i, j = _problem(rng, U, I, d, 300)
    a = [dev.array(x) for x in (P, Q, b)]
    c = [dev.array(x) for x in (P, Q, b)]
    dev.bpr_sgd_step(*a, dev.array(u), dev.array(i), dev.array(j), 0.01, 0.05)
    dev.bpr_sgd_step_noalloc(*c, dev.array(u), dev.array(i), dev.array(j), 0.01, 0.05)
    dev.bpr_sgd_step_noalloc(*c, dev.array(u[:0]), dev.array(i[:0]), dev.array(j[:0]), 0.01, 0.05)
    dev.synchronize()
    for x, y in zip(a, c):
        assert np.array_equal(x.numpy(), y.numpy())
    dev.close()


def test_no_allocation_step_refuses_what_does_not_fit():
    from tfrec_amd.device import Device
    dev = Device(0)
    rng = np.random.default_rng(1)
    U, I, d, B = 1000, 1000, 64, 200_000
    P, Q, b, u, i, j = _problem(rng, U, I, d, B)
    c = [dev.array(x) for x in (P, Q, b)]
    with pytest.raises(NotImplementedError, match="scratch"):
        dev.bpr_sgd_step_noalloc(*c, dev.array(u), dev.array(i), dev.array(j), 0.01, 0.05)
    dev.synchronize()
    for x, y in zip((P, Q, b), c):
        assert np.array_equal(x, y.numpy())              # nothing ran
    # an out-of-range index is reported as by the plain step
    u2 = u[:1000].copy(); u2[7] = U
    dev.bpr_sgd_step_noalloc(*c, dev.array(u2), dev.array(i[:1000]), dev.array(j[:1000]), 0.01, 0.05)
    with pytest.raises(IndexError):
        dev.synchronize()
    dev.close()
