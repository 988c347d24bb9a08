"""Not collected by pytest (run by hand on a GPU box): random shapes of tfrec_score_topk against a float64 product (and, with integer-valued factors, against
the exact ranking including ties); every fifth case the whole evaluation (lists and metric curves) against the oracle.  python tests/fuzz_eval.py [cases] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfrec_amd.device import Device, SCORE_BF16, SCORE_F32  # noqa: E402


def one(dev, rng, case):
    mode_name = "bf16" if rng.random() < 0.6 else "f32"
    d = int(rng.choice([1, 7, 8, 16, 20, 32, 33, 48, 64, 65, 100, 127, 128] + ([] if mode_name == "f32" else [129, 200, 256])))
    edge = int(rng.choice([32, 64, 96, 128, 192, 256, 384, 512, 768, 1536, 3072]))
    I = int(rng.choice([rng.integers(1, 300), edge + rng.integers(-1, 2), rng.integers(300, 40000)]))
    I = max(I, 1)
    U = int(rng.choice([rng.integers(1, 70), rng.integers(70, 1500), rng.integers(1500, 6000)]))
    n = int(rng.integers(1, U + 1))
    if rng.random() < 0.1:
        # a handful of users against many items: the item range is split over workgroups and the lists merged
        n = int(rng.integers(1, min(U, 12) + 1))
        I = int(rng.integers(50000, 300000))
    elif U > 1500:
        I = min(I, 6000)
    k = int(min(rng.choice([1, 2, 10, 50, 100, 113, 128, 200, 240]), 240))
    model = str(rng.choice(["uniform", "biased", "integers", "constant"]))
    if model == "integers":
        # small integers: every product and sum is exact in bf16 operands and float32 sums, so the ranking is exact, ties included
        P = rng.integers(-2, 3, (U, d)).astype(np.float32)
        Q = rng.integers(-2, 3, (I, d)).astype(np.float32)
        b = rng.integers(-3, 4, I).astype(np.float32)
    elif model == "constant":
        P = np.full((U, d), 0.5, np.float32); Q = np.full((I, d), 0.25, np.float32); b = np.zeros(I, np.float32)
    else:
        P = rng.uniform(-0.5, 0.5, (U, d)).astype(np.float32)
        Q = rng.uniform(-0.5, 0.5, (I, d)).astype(np.float32)
        b = rng.normal(0, 0.1 if model == "uniform" else 3.0, I).astype(np.float32)
    if rng.random() < 0.15:
        b[rng.random(I) < 0.3] = -np.inf
    users = np.sort(rng.choice(U, n, replace=False)).astype(np.int32)
    train = None
    indptr = indices = None
    if rng.random() < 0.7:
        lens = np.minimum(rng.choice([0, 1, 3, 20, 150, 700, 5000], U), I)
        indptr = np.zeros(U + 1, np.int64); indptr[1:] = np.cumsum(lens)
        indices = np.concatenate([np.sort(rng.choice(I, m, replace=False)) for m in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
        train = (dev.array(indptr, np.int64), dev.array(indices, np.int32))
    mode = SCORE_F32 if mode_name == "f32" else SCORE_BF16
    desc = "case %d: %s d=%d U=%d n=%d I=%d k=%d model=%s mask=%s" % (case, mode_name, d, U, n, I, k, model, train is not None)
    ids, sc = dev.score_topk(dev.array(P), dev.array(Q), dev.array(b), dev.array(users), k, train=train, mode=mode)
    dev.synchronize()
    ids, sc = ids.numpy(), sc.numpy()
    rnd = (lambda x: torch.from_numpy(x).to(torch.bfloat16).double().numpy()) if mode_name == "bf16" else (lambda x: x.astype(np.float64))
    R = rnd(P[users]) @ rnd(Q).T + b.astype(np.float64)
    if indptr is not None:
        for r, u in enumerate(users):
            R[r, indices[indptr[u]:indptr[u + 1]]] = -np.inf
    kk = min(k, I)
    order = np.argsort(-R, axis=1, kind="stable")[:, :kk]
    top = np.take_along_axis(R, order, axis=1)
    valid = np.isfinite(top)
    # entries beyond the valid items of a user: id -1, score -inf
    got_valid = ids[:, :kk] >= 0
    if not np.array_equal(got_valid, valid):
        return desc + ": valid-entry pattern differs"
    if k > kk and not ((ids[:, kk:] == -1).all()):
        return desc + ": entries past the item count are not -1"
    if model in ("integers", "constant"):
        if not np.array_equal(np.where(valid, ids[:, :kk], -1), np.where(valid, order, -1)):
            bad = np.argwhere(np.where(valid, ids[:, :kk], -1) != np.where(valid, order, -1))[0]
            return desc + ": exact ranking differs at user row %d rank %d: got %d want %d" % (bad[0], bad[1], ids[bad[0], bad[1]], order[bad[0], bad[1]])
        if not np.array_equal(sc[:, :kk][valid], top[valid].astype(np.float32)):
            return desc + ": exact scores differ"
        return None
    tol = 2e-5 * max(1.0, float(np.abs(top[valid]).max()) if valid.any() else 1.0)
    if valid.any() and np.abs(sc[:, :kk][valid] - top[valid]).max() > tol:
        return desc + ": scores differ by %g" % np.abs(sc[:, :kk][valid] - top[valid]).max()
    nxt = np.argsort(-R, axis=1, kind="stable")[:, :kk + 1]
    topn = np.take_along_axis(R, nxt, axis=1)
    if topn.shape[1] == kk + 1:
        below = topn[:, :kk] - topn[:, 1:]
    else:
        below = np.concatenate([topn[:, :-1] - topn[:, 1:], np.full((n, 1), np.inf)], axis=1)
    above = np.concatenate([np.full((n, 1), np.inf), below[:, :-1]], axis=1)
    clear = valid & (below > tol) & (above > tol)
    if not np.array_equal(ids[:, :kk][clear], order[clear]):
        return desc + ": ids differ where the ranking is clear"
    s = np.where(np.isfinite(sc), sc, -3e38)
    if not (np.diff(s, axis=1) <= 0).all():
        return desc + ": scores not descending"
    return None


def one_metrics(dev, rng, case):
    """tfrec_eval_fullrank in f32 with the reference's float32 means (evaluator/backend/cpp/include/evaluate.h) on a random
    shape against the oracle: rank lists and the 5 x k metric curves bit for bit."""
    from oracle import oracle
    d = int(rng.choice([1, 8, 16, 20, 32, 64, 100, 128]))
    U, I = int(rng.integers(1, 1500)), int(rng.integers(260, 8000))
    k = int(rng.choice([1, 5, 10, 50, 100, 200]))
    n = int(rng.integers(1, U + 1))
    if rng.random() < 0.5:
        P = rng.integers(-2, 3, (U, d)).astype(np.float32); Q = rng.integers(-2, 3, (I, d)).astype(np.float32)
        b = rng.integers(-3, 4, I).astype(np.float32)
    else:
        P = rng.uniform(-0.5, 0.5, (U, d)).astype(np.float32); Q = rng.uniform(-0.5, 0.5, (I, d)).astype(np.float32)
        b = rng.normal(0, 0.1, I).astype(np.float32)
    def rows(max_len, min_len=0):
        lens = rng.integers(min_len, max_len + 1, U)
        ptr = np.zeros(U + 1, np.int64); ptr[1:] = np.cumsum(lens)
        idx = np.concatenate([np.sort(rng.choice(I, m, replace=False)) for m in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
        return ptr, idx
    tr, te = rows(int(rng.choice([0, 5, 50]))), rows(int(rng.choice([1, 3, 30])))
    # the evaluated users are users with test items (evaluator/abstract_evaluator.py keeps no others)
    have = np.flatnonzero(np.diff(te[0]) > 0)
    if len(have) == 0:
        return None
    users = np.sort(rng.choice(have, min(n, len(have)), replace=False)).astype(np.int32)
    n = len(users)
    ids = dev.empty((n, k), np.int32)
    m = dev.eval_fullrank(dev.array(P), dev.array(Q), dev.array(b), dev.array(users), (dev.array(tr[0], np.int64), dev.array(tr[1], np.int32)),
                          (dev.array(te[0], np.int64), dev.array(te[1], np.int32)), k, mode=SCORE_F32, mean_mode=1, out_ids=ids)
    mean, rank, _ = oracle.eval_fullrank(P, Q, b, users, tr, te, k, score_mode=1)      # the float32 chain the f32 kernel computes
    desc = "case %d: metrics d=%d U=%d n=%d I=%d k=%d" % (case, d, U, n, I, k)
    if not np.array_equal(ids.numpy(), rank):
        r, c = np.argwhere(ids.numpy() != rank)[0]
        g, w = int(ids.numpy()[r, c]), int(rank[r, c])
        sg, sw = [float(np.float32(P[users[r]].astype(np.float64) @ Q[x].astype(np.float64) + b[x])) for x in (g, w)]
        return desc + ": rank lists differ from the oracle's at row %d rank %d: %d (score %.9g) vs %d (score %.9g), %d entries in all" % (
            r, c, g, sg, w, sw, int((ids.numpy() != rank).sum()))
    if not np.array_equal(np.asarray(m), mean, equal_nan=True):
        return desc + ": metric curves differ by %g" % np.abs(np.asarray(m) - mean).max()
    return None


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dev = Device(0)
    rng = np.random.default_rng(seed)
    bad = 0
    t0 = time.time()
    for c in range(cases):
        msg = one_metrics(dev, rng, c) if c % 5 == 4 else one(dev, rng, c)
        if msg:
            bad += 1
            print("FAIL", msg, flush=True)
        if c % 25 == 24:
            print("%d cases, %d failures, %.0f s" % (c + 1, bad, time.time() - t0), flush=True)
    print("done: %d cases, %d failures" % (cases, bad), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
