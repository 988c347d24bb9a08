"""Not collected by pytest (run by hand on a GPU box): how far the float32 APR-Adagrad step is from the float64 oracle as the number of
occurrences per row grows (a handful of rows, up to 320,000 triples): the error of the accumulators grows smoothly with the length of
the (ill-conditioned: terms of both signs) float32 sums, from 2e-6 at a thousand occurrences per row to 1e-4 at a hundred thousand."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle
from tfrec_amd.device import Device
dev = Device(0)
for (U, I, d) in [(4, 8, 64), (7, 8, 16)]:
    for B in [1000, 4000, 16000, 40000, 80000, 160000, 320000]:
        rng = np.random.default_rng(B + U)
        P = rng.uniform(-0.3, 0.3, (U, d)).astype(np.float32); Q = rng.uniform(-0.3, 0.3, (I, d)).astype(np.float32); b = rng.uniform(-0.1, 0.1, I).astype(np.float32)
        u = rng.integers(0, U, B).astype(np.int32); i = rng.integers(0, I, B).astype(np.int32); j = rng.integers(0, I, B).astype(np.int32)
        Po, Qo, bo = P.copy(), Q.copy(), b.copy()
        aP, aQ, ab = [np.full_like(x, 0.1) for x in (P, Q, b)]
        oracle.apr_step(Po, Qo, bo, aP, aQ, ab, u, i, j, 0.05, 0.0, 1.0, 0.05)
        arrs = [dev.array(x) for x in (P, Q, b)]; accs = [dev.full(x.shape, 0.1) for x in (P, Q, b)]; dels = [dev.zeros(x.shape) for x in (P, Q, b)]
        loss = dev.zeros(3, np.float64)
        dev.apr_adagrad_step(*arrs, *accs, *dels, dev.array(u, np.int32), dev.array(i, np.int32), dev.array(j, np.int32), 0.05, 0.0, 1.0, 0.05, loss)
        dev.synchronize()
        rel = [float(np.max(np.abs(g.numpy() - r) / np.maximum(np.abs(r), 1e-30))) for g, r in zip(accs, (aP, aQ, ab))]
        upd = [float(np.abs(g.numpy() - r).max() / max(np.abs(r - o).max(), 1e-9)) for g, r, o in zip(arrs, (Po, Qo, bo), (P, Q, b))]
        print("U=%d I=%d d=%d B=%6d: acc rel err P %.1e Q %.1e b %.1e | param err / update P %.1e Q %.1e b %.1e | acc max %.3g" % (U, I, d, B, *rel, *upd, aQ.max()), flush=True)
