"""TEST INFRASTRUCTURE ONLY: Python face of the CPU oracle.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module; the product package never does.

* ctypes bindings to ``oracle/libmf_oracle.so`` (``oracle/mf_oracle.c``), the C
  restatement of the reference's step math and evaluator;
* numpy restatements of the reference's host-side sampling and batching
  (``utils/tools.py:39-62``, ``data/dataiterator.py:52-54,87-95``,
  ``model/BPR.py:50-62``, ``model/DNSBPR.py:65-80``);
* optional bindings to the reference's own evaluator built in ``oracle/_ref``.

Parity status: evaluator and sampler parts are pinned by golden vectors produced
with the reference's own code (tests/golden/make_golden.py); the TensorFlow step
math is a restatement only -- parity unpinned against TensorFlow itself.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "libmf_oracle.so")
        if not os.path.exists(path):
            sys.path.insert(0, HERE)
            import build as _b  # noqa
            _b.build_oracle()
        L = C.CDLL(path)
        L.oracle_ws_create.restype = C.c_void_p
        L.oracle_ws_create.argtypes = [C.c_int64, C.c_int64, C.c_int]
        L.oracle_ws_destroy.argtypes = [C.c_void_p]
        L.oracle_bpr_step.restype = C.c_int
        L.oracle_bpr_step.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, _i32p, _i32p, _i32p, C.c_int64,
                                      C.c_double, C.c_double, _f64p]
        L.oracle_bpr_step_mt.restype = C.c_int
        L.oracle_bpr_step_mt.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, _i32p, _i32p, _i32p, C.c_int64,
                                         C.c_double, C.c_double, _f64p, _f64p, C.c_int]
        L.oracle_apr_step.restype = C.c_int
        L.oracle_apr_step.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p,
                                      C.c_void_p, C.c_void_p, C.c_void_p,
                                      _i32p, _i32p, _i32p, C.c_int64,
                                      C.c_double, C.c_double, C.c_double, C.c_double, _f64p]
        L.oracle_dns_select.restype = C.c_int
        L.oracle_dns_select.argtypes = [_f32p, _f32p, _f32p, C.c_int, _i32p, _i32p, C.c_int64, C.c_int,
                                        C.c_int, _i32p, _f64p]
        L.oracle_score_row.argtypes = [_f32p, _f32p, _f32p, C.c_int64, C.c_int, C.c_int, _f32p]
        L.oracle_topk_row.argtypes = [_f32p, C.c_int64, C.c_int, _i32p]
        L.oracle_metrics_user.argtypes = [_i32p, C.c_int, _i32p, C.c_int, _f32p]
        L.oracle_metrics_mean.argtypes = [_f32p, C.c_int64, C.c_int, _f32p]
        L.oracle_eval_fullrank.restype = C.c_int
        L.oracle_eval_fullrank.argtypes = [_f32p, _f32p, _f32p, C.c_int64, C.c_int, _i32p, C.c_int64,
                                           C.c_void_p, C.c_void_p, _i64p, _i32p,
                                           C.c_int, C.c_int, C.c_void_p, C.c_void_p, _f32p]
        _lib = L
    return _lib


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


class Workspace:
    def __init__(self, U, I, d):
        self.U, self.I, self.d = int(U), int(I), int(d)
        self.h = lib().oracle_ws_create(self.U, self.I, self.d)
        if not self.h:
            raise MemoryError("oracle workspace")

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_ws_destroy(self.h)
            self.h = None


def bpr_step(P, Q, b, users, pos, neg, lr, reg, ws=None, threads=1):
    """In-place BPR-SGD minibatch step on float32 arrays; returns (rank_loss, reg_loss).
    ``threads`` > 1 runs the multi-threaded statement of the same step (parameters bit-identical)."""
    ws = ws or Workspace(P.shape[0], Q.shape[0], P.shape[1])
    out = np.zeros(2)
    if threads > 1:
        rc = lib().oracle_bpr_step_mt(ws.h, P, Q, b, _c(users, np.int32), _c(pos, np.int32), _c(neg, np.int32),
                                      len(users), float(lr), float(reg), out, np.empty(len(users)), int(threads))
    else:
        rc = lib().oracle_bpr_step(ws.h, P, Q, b, _c(users, np.int32), _c(pos, np.int32), _c(neg, np.int32),
                                   len(users), float(lr), float(reg), out)
    if rc:
        raise IndexError("index out of range in oracle_bpr_step")
    return float(out[0]), float(out[1])


def apr_step(P, Q, b, accP, accQ, accb, users, pos, neg, lr, reg, reg_adv, eps, ws=None, deltas=None):
    """In-place APR step (delta computation + sparse Adagrad); returns (L, L_adv, mean_sq)."""
    ws = ws or Workspace(P.shape[0], Q.shape[0], P.shape[1])
    out = np.zeros(3)
    dp = dq = db = None
    if deltas is not None:
        dP, dQ, dB = deltas
        dp, dq, db = dP.ctypes.data, dQ.ctypes.data, dB.ctypes.data
    rc = lib().oracle_apr_step(ws.h, P, Q, b, accP, accQ, accb, dp, dq, db,
                               _c(users, np.int32), _c(pos, np.int32), _c(neg, np.int32), len(users),
                               float(lr), float(reg), float(reg_adv), float(eps), out)
    if rc:
        raise IndexError("index out of range in oracle_apr_step")
    return tuple(float(x) for x in out)


def dns_select(P, Q, b, users, cand, score_mode=0):
    """cand [n_pos, C] -> (chosen item per positive, margin to the best different candidate)."""
    cand = _c(cand, np.int32)
    n, Cn = cand.shape
    out = np.zeros(n, np.int32)
    margin = np.zeros(n)
    lib().oracle_dns_select(P, Q, b, P.shape[1], _c(users, np.int32), cand, n, Cn, score_mode, out, margin)
    return out, margin


def score_row(p, Q, b, mode=0):
    out = np.empty(Q.shape[0], np.float32)
    lib().oracle_score_row(_c(p, np.float32), Q, b, Q.shape[0], Q.shape[1], mode, out)
    return out


def topk_row(score, k):
    out = np.empty(k, np.int32)
    lib().oracle_topk_row(_c(score, np.float32), len(score), k, out)
    return out


def metrics_user(rank, truth):
    rank = _c(rank, np.int32)
    out = np.empty((5, len(rank)), np.float32)
    lib().oracle_metrics_user(rank, len(rank), _c(truth, np.int32), len(truth), out)
    return out


def metrics_mean(per_user):
    per_user = _c(per_user, np.float32)
    n, five, k = per_user.shape
    out = np.empty((5, k), np.float32)
    lib().oracle_metrics_mean(per_user, n, k, out)
    return out


def eval_fullrank(P, Q, b, test_users, train_csr, test_csr, k, score_mode=0, want_rank=True, want_per_user=False):
    """train_csr/test_csr: (indptr int64 [U+1], indices int32) tuples; train_csr may be None."""
    test_users = _c(test_users, np.int32)
    n = len(test_users)
    rank = np.empty((n, k), np.int32) if want_rank else None
    per = np.empty((n, 5, k), np.float32) if want_per_user else None
    mean = np.empty((5, k), np.float32)
    tp = ti = None
    keep = []
    if train_csr is not None:
        a, bb = _c(train_csr[0], np.int64), _c(train_csr[1], np.int32)
        keep += [a, bb]
        tp, ti = a.ctypes.data, bb.ctypes.data
    rc = lib().oracle_eval_fullrank(P, Q, b, Q.shape[0], Q.shape[1], test_users, n, tp, ti,
                                    _c(test_csr[0], np.int64), _c(test_csr[1], np.int32), k, score_mode,
                                    rank.ctypes.data if want_rank else None,
                                    per.ctypes.data if want_per_user else None, mean)
    if rc:
        raise MemoryError
    return mean, rank, per


def eval_dense(R, train_dict, test_dict, k):
    """Restatement of eval_rating_matrix (evaluator/backend/cpp/evaluate.py:10-44) on a dense
    matrix, without mutating the caller's array.  Returns ([5,k] mean, [n,k] ranks)."""
    R = np.array(R, dtype=np.float32, copy=True)
    if train_dict is not None:
        for u, items in train_dict.items():
            R[u][np.asarray(items)] = -np.inf      # float32(-sys.float_info.max) is -inf (evaluate.py:27)
    users = sorted(test_dict.keys())                # evaluate.py:30
    if len(R) != len(users):
        R = R[users]                                # evaluate.py:37-38
    ranks = np.stack([topk_row(R[r], k) for r in range(len(users))])
    per = np.stack([metrics_user(ranks[r], test_dict[u]) for r, u in enumerate(users)])
    return metrics_mean(per), ranks


# --------------------------------------------------------------------------- #
# host-side sampling / batching restatements                                  #
# --------------------------------------------------------------------------- #
def sample_negatives(indptr, indices, n_users, n_items, n_neg, seed, stream, first=0):
    """CPU statement of the device sampler (tfrec_sample_negatives): (users, pos, neg[nnz, n_neg], n_full_rows)."""
    L = lib()
    L.oracle_sample_negatives.restype = C.c_int64
    L.oracle_sample_negatives.argtypes = [_i64p, _i32p, C.c_int64, C.c_int64, C.c_int, C.c_uint64, C.c_uint64, C.c_int64, _i32p, _i32p, _i32p]
    indptr, indices = _c(indptr, np.int64), _c(indices, np.int32)
    nnz = len(indices)
    users, pos, neg = np.empty(nnz, np.int32), np.empty(nnz, np.int32), np.empty((nnz, n_neg), np.int32)
    full = L.oracle_sample_negatives(indptr, indices, n_users, n_items, n_neg, seed, stream, first, users, pos, neg)
    return users, pos, neg, int(full)


def shuffle_keys(n, seed, stream):
    """The 64-bit hash keys tfrec_shuffle_perm orders the elements by."""
    L = lib()
    L.oracle_shuffle_keys.restype = None
    L.oracle_shuffle_keys.argtypes = [C.c_int64, C.c_uint64, C.c_uint64, np.ctypeslib.ndpointer(np.uint64, flags="C")]
    keys = np.empty(n, np.uint64)
    L.oracle_shuffle_keys(n, seed, stream, keys)
    return keys


def shuffle_perm(n, seed, stream):
    """CPU statement of tfrec_shuffle_perm: stable argsort of the 64-bit hash keys."""
    return np.argsort(shuffle_keys(n, seed, stream), kind="stable").astype(np.int32)


def csr_to_user_dict(csr):
    """utils/tools.py:52-62 -- including its quirk: a row is kept only if ``any(indices)``
    is true, so empty rows AND rows whose only item id is 0 are dropped."""
    out = {}
    indptr, indices = csr.indptr, csr.indices
    for u in range(csr.shape[0]):
        row = indices[indptr[u]:indptr[u + 1]]
        if row.size and row.any():
            out[u] = row
    return out


def choice_excluding(rs, n_items, size, exclusion):
    """utils/tools.py:39-49 with numpy's legacy ``RandomState.choice(p=...)`` spelled out:
    p = 1[not excluded] / count  ->  cdf = cumsum(p); cdf /= cdf[-1];
    draws = cdf.searchsorted(random_sample(size), side='right').
    ``rs`` is a numpy RandomState (the reference uses the global one)."""
    p = np.ones(n_items, dtype=np.int64)
    p[exclusion] = 0
    p = p / np.sum(p)
    cdf = p.cumsum()
    cdf /= cdf[-1]
    u = rs.random_sample(size)
    return cdf.searchsorted(u, side="right")


def bpr_epoch_triples(rs, user_pos, n_items):
    """model/BPR.py:50-62: users in dict order, one uniform non-positive per positive."""
    users, pos, neg = [], [], []
    for u, items in user_pos.items():
        ng = choice_excluding(rs, n_items, len(items), items)
        users.append(np.full(len(items), u, np.int32))
        pos.append(np.asarray(items, np.int32))
        neg.append(ng.astype(np.int32))
    return np.concatenate(users), np.concatenate(pos), np.concatenate(neg)


def batch_order(rs, n, batch_size):
    """data/dataiterator.py:52-54, 87-95: one permutation per epoch, consecutive chunks,
    last short batch kept."""
    perm = rs.permutation(n)
    return [perm[s:s + batch_size] for s in range(0, n, batch_size)]


def dns_pool_and_pick(rs, logits, items, n_items, neg_num):
    """model/DNSBPR.py:72-79 for one user, given that user's all-item logits."""
    pool = choice_excluding(rs, n_items, neg_num * len(items), items)
    pool2 = pool.reshape(len(items), neg_num)
    pick = np.argmax(logits[pool].reshape(len(items), neg_num), axis=1)
    return pool2, pool2[np.arange(len(items)), pick]


# --------------------------------------------------------------------------- #
# the reference's own evaluator (only where oracle/_ref was built)             #
# --------------------------------------------------------------------------- #
_ref = None


def ref_lib():
    """ctypes handle on oracle/_ref/libref_eval.so or None."""
    global _ref
    if _ref is None:
        path = os.path.join(HERE, "_ref", "libref_eval.so")
        if not os.path.exists(path):
            return None
        L = C.CDLL(path)
        L.ref_topk_rows.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, _i32p]
        L.ref_evaluate.argtypes = [_i32p, C.c_int, C.c_int, _i32p, _i64p, C.c_int, _f32p]
        L.ref_metrics_one.argtypes = [_i32p, C.c_int, _i32p, C.c_int, _f32p]
        _ref = L
    return _ref


def ref_topk_rows(R, k, threads=8):
    R = _c(R, np.float32)
    out = np.zeros((R.shape[0], k), np.int32)
    ref_lib().ref_topk_rows(R, R.shape[0], R.shape[1], k, threads, out)
    return out


def ref_evaluate(ranks, truths, threads=8):
    ranks = _c(ranks, np.int32)
    ptr = np.zeros(len(truths) + 1, np.int64)
    ptr[1:] = np.cumsum([len(t) for t in truths])
    flat = _c(np.concatenate([np.asarray(t, np.int32) for t in truths]), np.int32)
    out = np.zeros((5, ranks.shape[1]), np.float32)
    ref_lib().ref_evaluate(ranks, ranks.shape[0], ranks.shape[1], flat, ptr, threads, out)
    return out
