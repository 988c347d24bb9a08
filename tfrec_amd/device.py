"""Device context: the Python face of the C ABI (include/tfrec_amd.h).

It stands where the reference passes a ``tf.Session`` (run_model.py:14-16): models receive a
``Device`` as their ``sess`` argument.  Buffers are either ``DevArray`` objects (memory owned by
this module through tfrec_dev_malloc) or torch tensors on the same GPU (borrowed through
``data_ptr()``); both are just a pointer, a shape and a dtype to the library.
"""
import ctypes as C

import numpy as np

from . import _lib

SCORE_F32 = _lib.SCORE_F32
SCORE_BF16 = _lib.SCORE_BF16


class DevArray:
    """A typed device allocation (row-major)."""

    def __init__(self, dev, shape, dtype):
        self.dev = dev
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        _lib.check(dev.lib.tfrec_dev_malloc(dev.h, max(self.nbytes, 1), C.byref(p)))
        self.ptr = p.value

    def __len__(self):
        return self.shape[0]

    def numpy(self):
        out = np.empty(self.shape, self.dtype)
        _lib.check(self.dev.lib.tfrec_memcpy_d2h(self.dev.h, out.ctypes.data, self.ptr, self.nbytes))
        return out

    def copy_from(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        if host.nbytes != self.nbytes:
            raise ValueError("size mismatch: %d vs %d bytes" % (host.nbytes, self.nbytes))
        _lib.check(self.dev.lib.tfrec_memcpy_h2d(self.dev.h, self.ptr, host.ctypes.data, self.nbytes))
        return self

    def view(self, lo, hi):
        """Rows [lo, hi) as a non-owning array (the parent must outlive it)."""
        return DevView(self, lo, hi)

    def free(self):
        if self.ptr and self.dev.h:
            self.dev.lib.tfrec_dev_free(self.dev.h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DevView:
    """A row range of a DevArray; just a pointer and a shape."""

    def __init__(self, parent, lo, hi):
        if not (0 <= lo <= hi <= parent.shape[0]):
            raise IndexError("view [%d,%d) of %d rows" % (lo, hi, parent.shape[0]))
        row_bytes = parent.nbytes // max(parent.shape[0], 1)
        self.parent = parent
        self.dev = parent.dev
        self.dtype = parent.dtype
        self.shape = (hi - lo,) + parent.shape[1:]
        self.nbytes = row_bytes * (hi - lo)
        self.ptr = parent.ptr + row_bytes * lo

    def numpy(self):
        out = np.empty(self.shape, self.dtype)
        _lib.check(self.dev.lib.tfrec_memcpy_d2h(self.dev.h, out.ctypes.data, self.ptr, self.nbytes))
        return out


class UserTable:
    """A versioned user table (``tfrec_utable``): the caller's ``P`` paired with a second row buffer and one version
    byte per user inside the library, so that a training step need not stage the pre-step user rows for its item
    half.  Between ``flatten()`` calls ``P`` is NOT complete: call ``flatten()`` before anything else reads it."""

    def __init__(self, dev, P):
        self.dev, self.P = dev, P
        U, d = _shape(P)
        h = C.c_void_p()
        _lib.check(dev.lib.tfrec_utable_create(dev.h, _ptr(P), U, d, C.byref(h)))
        self.h = h.value
        self.shape = (U, d)

    def flatten(self):
        """Makes ``P`` complete and current again (stream-ordered); returns ``P``."""
        _lib.check(self.dev.lib.tfrec_utable_flatten(self.dev.h, self.h))
        return self.P

    def close(self):
        """Flattens, then releases the second row buffer and the version bytes.  After the Device was closed the
        buffers are still released (without a flatten: nothing can read ``P`` through a closed context anyway)."""
        if self.h and self.dev.h:
            h, self.h = self.h, None
            _lib.check(self.dev.lib.tfrec_utable_destroy(self.dev.h, h))
        elif self.h:
            h, self.h = self.h, None
            self.dev.lib.tfrec_utable_destroy(None, h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _ptr(x):
    """Device pointer of a DevArray, a torch tensor, an int, or None."""
    if x is None:
        return None
    if isinstance(x, (DevArray, DevView)):
        return x.ptr
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):
        if not x.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return x.data_ptr()
    raise TypeError("not a device buffer: %r" % type(x))


def _shape(x):
    return tuple(x.shape)


class Device:
    """One GPU, one stream, one workspace.  Not re-entrant."""

    def __init__(self, device=0, stream=None):
        self.lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self.lib.tfrec_ctx_create(int(device), stream, C.byref(h)))
        self.h = h.value
        self.device = int(device)

    def close(self):
        if self.h:
            self.lib.tfrec_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- memory -----------------------------------------------------------------------------
    def empty(self, shape, dtype=np.float32):
        return DevArray(self, shape, dtype)

    def array(self, host, dtype=None):
        host = np.ascontiguousarray(host, dtype=dtype)
        return DevArray(self, host.shape, host.dtype).copy_from(host)

    def zeros(self, shape, dtype=np.float32):
        a = DevArray(self, shape, dtype)
        _lib.check(self.lib.tfrec_memset_zero(self.h, a.ptr, a.nbytes))
        return a

    def full(self, shape, value):
        a = DevArray(self, shape, np.float32)
        _lib.check(self.lib.tfrec_fill_const(self.h, a.ptr, a.nbytes // 4, float(value)))
        return a

    def uniform(self, shape, lo, hi, seed):
        a = DevArray(self, shape, np.float32)
        _lib.check(self.lib.tfrec_fill_uniform(self.h, a.ptr, a.nbytes // 4, float(lo), float(hi), int(seed)))
        return a

    def synchronize(self):
        """Waits for the stream; raises IndexError if a batch held an out-of-range index."""
        _lib.check(self.lib.tfrec_ctx_synchronize(self.h))

    def reserve(self, max_batch=0, d=0, max_eval_users=0, k=0):
        _lib.check(self.lib.tfrec_ctx_reserve(self.h, int(max_batch), int(d), int(max_eval_users), int(k)))

    def workspace_bytes(self):
        return int(self.lib.tfrec_ctx_workspace_bytes(self.h))

    def set_profiling(self, on=True):
        _lib.check(self.lib.tfrec_ctx_set_profiling(self.h, 1 if on else 0))

    def phase_times(self):
        """[(phase name, milliseconds)] of the last profiled call (waits for it)."""
        ms = (C.c_float * 16)()
        n = C.c_int(0)
        _lib.check(self.lib.tfrec_ctx_phase_times(self.h, ms, 16, C.byref(n)))
        return [(self.lib.tfrec_ctx_phase_name(self.h, i).decode(), float(ms[i])) for i in range(n.value)]

    @property
    def stream(self):
        return self.lib.tfrec_ctx_stream(self.h)

    # -- model math -------------------------------------------------------------------------
    def predict(self, P, Q, b, users, items, out=None):
        U, d = _shape(P)
        I = _shape(Q)[0]
        n = _shape(users)[0]
        out = out if out is not None else self.empty(n, np.float32)
        _lib.check(self.lib.tfrec_mf_predict(self.h, _ptr(P), _ptr(Q), _ptr(b), U, I, d, _ptr(users), _ptr(items), n, _ptr(out)))
        return out

    # -- epoch pipeline ---------------------------------------------------------------------
    def sample_negatives(self, indptr, indices, n_users, n_items, n_neg, seed, stream, users_out=None, pos_out=None, neg_out=None,
                         first_interaction=0):
        """Per stored interaction (CSR order): user, item and ``n_neg`` uniform draws from the items not
        in the user's row (model/BPR.py:50-62, utils/tools.py:39-49).  Returns (users, pos, neg[nnz, n_neg]).
        ``first_interaction``: position of this matrix's first interaction in the whole training set (row shards)."""
        nnz = _shape(indices)[0]
        users_out = users_out if users_out is not None else self.empty(nnz, np.int32)
        pos_out = pos_out if pos_out is not None else self.empty(nnz, np.int32)
        neg_out = neg_out if neg_out is not None else self.empty((nnz, n_neg), np.int32)
        _lib.check(self.lib.tfrec_sample_negatives(self.h, _ptr(indptr), _ptr(indices), int(n_users), int(n_items), nnz, int(n_neg),
                                                   int(seed) & (2 ** 64 - 1), int(stream) & (2 ** 64 - 1), int(first_interaction),
                                                   _ptr(users_out), _ptr(pos_out), _ptr(neg_out)))
        return users_out, pos_out, neg_out

    def epoch_triples(self, indptr, indices, n_users, n_items, seed, draw_stream, shuffle_stream, users_out=None, pos_out=None, neg_out=None,
                      first_interaction=0):
        """One epoch of BPR triples, drawn (one negative per stored interaction) and shuffled: what ``sample_negatives``,
        ``shuffle_perm`` and three ``gather_rows_i32`` give, in one call (model/BPR.py:50-62, data/dataiterator.py:52-54).
        Returns (users, pos, neg), int32 [nnz] each."""
        nnz = _shape(indices)[0]
        users_out = users_out if users_out is not None else self.empty(nnz, np.int32)
        pos_out = pos_out if pos_out is not None else self.empty(nnz, np.int32)
        neg_out = neg_out if neg_out is not None else self.empty(nnz, np.int32)
        m = 2 ** 64 - 1
        _lib.check(self.lib.tfrec_epoch_triples(self.h, _ptr(indptr), _ptr(indices), int(n_users), int(n_items), nnz, int(seed) & m,
                                                int(draw_stream) & m, int(shuffle_stream) & m, int(first_interaction),
                                                _ptr(users_out), _ptr(pos_out), _ptr(neg_out)))
        return users_out, pos_out, neg_out

    def shuffle_perm(self, n, seed, stream, out=None):
        """A permutation of range(n), a pure function of (seed, stream, n) (data/dataiterator.py:52-54)."""
        out = out if out is not None else self.empty(n, np.int32)
        _lib.check(self.lib.tfrec_shuffle_perm(self.h, int(n), int(seed) & (2 ** 64 - 1), int(stream) & (2 ** 64 - 1), _ptr(out)))
        return out

    def gather_rows_i32(self, src, perm, out=None):
        """out[r] = src[perm[r]] for an int32 array of rows."""
        shp = _shape(src)
        n = _shape(perm)[0]
        width = int(np.prod(shp[1:])) if len(shp) > 1 else 1
        out = out if out is not None else self.empty((n,) + tuple(shp[1:]), np.int32)
        _lib.check(self.lib.tfrec_gather_rows_i32(self.h, _ptr(src), shp[0], _ptr(perm), n, width, _ptr(out)))
        return out

    def sort_pairs(self, keys, vals, key_bits, out_keys=None, out_vals=None):
        """Stable sort of uint32 (key, value) pairs by the low ``key_bits`` bits of the key."""
        n = _shape(keys)[0]
        out_keys = out_keys if out_keys is not None else self.empty(n, np.uint32)
        out_vals = out_vals if out_vals is not None else self.empty(n, np.uint32)
        _lib.check(self.lib.tfrec_sort_pairs_u32(self.h, _ptr(keys), _ptr(vals), _ptr(out_keys), _ptr(out_vals), n, int(key_bits)))
        return out_keys, out_vals

    def bpr_sgd_step(self, P, Q, b, users, pos, neg, lr, reg, loss=None):
        """One minibatch step (model/BPR.py:36-48 + :72).  ``loss``: optional device float64[2]."""
        U, d = _shape(P)
        I = _shape(Q)[0]
        B = _shape(users)[0]
        _lib.check(self.lib.tfrec_bpr_sgd_step(self.h, _ptr(P), _ptr(Q), _ptr(b), U, I, d, _ptr(users), _ptr(pos), _ptr(neg), B,
                                               float(lr), float(reg), _ptr(loss)))

    def user_table(self, P):
        """Wraps ``P`` in a versioned user table for the ``*_v`` steps (see ``UserTable``)."""
        return UserTable(self, P)

    def bpr_sgd_step_v(self, ut, Q, b, users, pos, neg, lr, reg, loss=None):
        """``bpr_sgd_step`` on a versioned user table."""
        I = _shape(Q)[0]
        B = _shape(users)[0]
        _lib.check(self.lib.tfrec_bpr_sgd_step_v(self.h, ut.h, _ptr(Q), _ptr(b), I, _ptr(users), _ptr(pos), _ptr(neg), B,
                                                 float(lr), float(reg), _ptr(loss)))

    def bpr_sgd_step_prepared_v(self, slot, ut, Q, b, lr, reg, loss=None):
        I = _shape(Q)[0]
        _lib.check(self.lib.tfrec_bpr_sgd_step_prepared_v(self.h, int(slot), ut.h, _ptr(Q), _ptr(b), I, float(lr), float(reg), _ptr(loss)))

    def bpr_dense_grad_step_v(self, ut, Q, b, G, users, pos, neg, lr, reg, slot=-1, loss=None):
        I = _shape(Q)[0]
        B = _shape(users)[0] if users is not None else 0
        _lib.check(self.lib.tfrec_bpr_dense_grad_step_v(self.h, int(slot), ut.h, _ptr(Q), _ptr(b), _ptr(G), I, _ptr(users), _ptr(pos),
                                                        _ptr(neg), B, float(lr), float(reg), _ptr(loss)))

    def bpr_dense_grad_step_chunked(self, ut, Q, b, G, users, pos, neg, lr, reg, chunk_events, slot=-1, loss=None):
        """``bpr_dense_grad_step_v`` with the item half in ``len(chunk_events)`` ranges of item ids; ``chunk_events``
        are raw HIP event handles (ints), recorded on the context's stream as the ranges' rows of ``G`` complete."""
        I = _shape(Q)[0]
        B = _shape(users)[0] if users is not None else 0
        n = len(chunk_events)
        evs = (C.c_void_p * n)(*[C.c_void_p(int(e)) for e in chunk_events])
        _lib.check(self.lib.tfrec_bpr_dense_grad_step_chunked(self.h, int(slot), ut.h, _ptr(Q), _ptr(b), _ptr(G), I, _ptr(users), _ptr(pos),
                                                              _ptr(neg), B, float(lr), float(reg), _ptr(loss), n, C.cast(evs, C.c_void_p)))

    def bpr_sgd_step_noalloc(self, P, Q, b, users, pos, neg, lr, reg, loss=None):
        """``bpr_sgd_step`` for callers that cannot allocate: all device scratch comes from a fixed array inside
        the library (16 MiB per GPU); larger batches raise NotImplementedError.  Calls that overlap on one GPU race."""
        U, d = _shape(P)
        I = _shape(Q)[0]
        B = _shape(users)[0]
        _lib.check(self.lib.tfrec_bpr_sgd_step_noalloc(self.h, _ptr(P), _ptr(Q), _ptr(b), U, I, d, _ptr(users), _ptr(pos), _ptr(neg),
                                                       B, float(lr), float(reg), _ptr(loss)))

    def prepare_batch(self, slot, U, I, users, pos, neg):
        """Sorts a batch on the auxiliary stream (overlaps the row passes of the running step)."""
        B = _shape(users)[0]
        _lib.check(self.lib.tfrec_batch_prepare(self.h, int(slot), int(U), int(I), _ptr(users), _ptr(pos), _ptr(neg), B))

    def prepare_batches(self, slot, U, I, users, pos, neg, K):
        """Sorts K consecutive batches (arrays of K*B triples) at once; K prepared steps consume them."""
        n = _shape(users)[0]
        if n % K:
            raise ValueError("%d triples are not %d equal batches" % (n, K))
        _lib.check(self.lib.tfrec_batch_prepare_many(self.h, int(slot), int(U), int(I), _ptr(users), _ptr(pos), _ptr(neg), n // K, int(K)))

    def bpr_sgd_step_prepared(self, slot, P, Q, b, lr, reg, loss=None):
        U, d = _shape(P)
        I = _shape(Q)[0]
        _lib.check(self.lib.tfrec_bpr_sgd_step_prepared(self.h, int(slot), _ptr(P), _ptr(Q), _ptr(b), U, I, d, float(lr), float(reg),
                                                        _ptr(loss)))

    def bpr_grad_step(self, P, Qc, bc, users, pos_slot, neg_slot, lr, reg, loss=None):
        """Sharded step, compute side: P updated in place, Qc/bc (fetched item rows) replaced by gradient sums."""
        U, d = _shape(P)
        Ic = _shape(Qc)[0]
        B = _shape(users)[0]
        _lib.check(self.lib.tfrec_bpr_grad_step(self.h, _ptr(P), _ptr(Qc), _ptr(bc), U, Ic, d, _ptr(users), _ptr(pos_slot),
                                                _ptr(neg_slot), B, float(lr), float(reg), _ptr(loss)))

    def bpr_grad_step_prepared(self, slot, P, Qc, bc, lr, reg, loss=None):
        """``bpr_grad_step`` on a batch sorted ahead with ``prepare_batch(slot, U, Ic, users, pos_slot, neg_slot)``."""
        U, d = _shape(P)
        Ic = _shape(Qc)[0]
        _lib.check(self.lib.tfrec_bpr_grad_step_prepared(self.h, int(slot), _ptr(P), _ptr(Qc), _ptr(bc), U, Ic, d, float(lr), float(reg),
                                                         _ptr(loss)))

    def bpr_grad_step_prepared_v(self, slot, ut, Qc, bc, lr, reg, loss=None):
        """``bpr_grad_step_prepared`` on a versioned user table."""
        Ic = _shape(Qc)[0]
        _lib.check(self.lib.tfrec_bpr_grad_step_prepared_v(self.h, int(slot), ut.h, _ptr(Qc), _ptr(bc), Ic, float(lr), float(reg), _ptr(loss)))

    def items_dedup(self, pos, neg, I, items_per_owner, n_owners, uniq, pos_slot, neg_slot, counts, n_uniq=None):
        """Distinct items of a batch (ascending), every triple's slots in that list, distinct items per owner."""
        B = _shape(pos)[0]
        _lib.check(self.lib.tfrec_items_dedup(self.h, _ptr(pos), _ptr(neg), B, int(I), int(items_per_owner), int(n_owners), _ptr(uniq),
                                              _ptr(pos_slot), _ptr(neg_slot), _ptr(counts), _ptr(n_uniq)))

    def rows_gather(self, Q, b, ids, out_rows, out_b):
        """out_rows[t] = Q[ids[t]], out_b[t] = b[ids[t]]."""
        I, d = _shape(Q)
        n = _shape(ids)[0]
        _lib.check(self.lib.tfrec_rows_gather(self.h, _ptr(Q), _ptr(b), I, d, _ptr(ids), n, _ptr(out_rows), _ptr(out_b)))

    def bpr_dense_grad_step(self, P, Q, b, G, users, pos, neg, lr, reg, slot=-1, loss=None):
        """Step of a GPU holding a P shard and a full Q, b copy: P updated in place, the flat table
        ``G`` ([I*d | I] floats) zeroed and filled with the gradient sums of the touched items.
        ``slot`` 0/1 consumes a batch prepared with ``prepare_batch`` (index arrays may be None)."""
        U, d = _shape(P)
        I = _shape(Q)[0]
        B = _shape(users)[0] if users is not None else 0
        _lib.check(self.lib.tfrec_bpr_dense_grad_step(self.h, int(slot), _ptr(P), _ptr(Q), _ptr(b), _ptr(G), U, I, d, _ptr(users),
                                                      _ptr(pos), _ptr(neg), B, float(lr), float(reg), _ptr(loss)))

    def axpy(self, y, x, alpha, n=None):
        """y[i] = fma(alpha, x[i], y[i])."""
        n = int(n) if n is not None else int(np.prod(_shape(y)))
        _lib.check(self.lib.tfrec_axpy(self.h, _ptr(y), _ptr(x), float(alpha), n))

    def rows_sgd_apply(self, Q, b, ids, grad_rows, grad_b, lr):
        """Sharded step, owner side: Q[ids] -= lr * grad_rows (duplicates summed in order), b likewise."""
        I, d = _shape(Q)
        n = _shape(ids)[0]
        _lib.check(self.lib.tfrec_rows_sgd_apply(self.h, _ptr(Q), _ptr(b), I, d, _ptr(ids), _ptr(grad_rows), _ptr(grad_b), n, float(lr)))

    def apr_adagrad_step(self, P, Q, b, accP, accQ, accb, dP, dQ, db, users, pos, neg, lr, reg, reg_adv, eps, loss=None):
        """Both session calls of model/APR.py:163-164.  ``loss``: optional device float64[3]."""
        U, d = _shape(P)
        I = _shape(Q)[0]
        B = _shape(users)[0]
        _lib.check(self.lib.tfrec_apr_adagrad_step(self.h, _ptr(P), _ptr(Q), _ptr(b), _ptr(accP), _ptr(accQ), _ptr(accb),
                                                   _ptr(dP), _ptr(dQ), _ptr(db), U, I, d, _ptr(users), _ptr(pos), _ptr(neg), B,
                                                   float(lr), float(reg), float(reg_adv), float(eps), _ptr(loss)))

    # -- APR with a replicated item table (see include/tfrec_amd.h for the four-call protocol) --
    def apr_dense_delta_grads(self, P, Q, b, dP, G, users, pos, neg, eps):
        U, d = _shape(P)
        I = _shape(Q)[0]
        B = _shape(users)[0]
        _lib.check(self.lib.tfrec_apr_dense_delta_grads(self.h, _ptr(P), _ptr(Q), _ptr(b), _ptr(dP), _ptr(G), U, I, d,
                                                        _ptr(users), _ptr(pos), _ptr(neg), B, float(eps)))

    def apr_delta_from_grads(self, G, dQ, db, eps):
        I, d = _shape(dQ)
        _lib.check(self.lib.tfrec_apr_delta_from_grads(self.h, _ptr(G), _ptr(dQ), _ptr(db), I, d, float(eps)))

    def apr_dense_adagrad_grads(self, P, Q, b, accP, dP, dQ, db, G, n_local, n_total, lr, reg, reg_adv, loss=None):
        U, d = _shape(P)
        I = _shape(Q)[0]
        _lib.check(self.lib.tfrec_apr_dense_adagrad_grads(self.h, _ptr(P), _ptr(Q), _ptr(b), _ptr(accP), _ptr(dP), _ptr(dQ), _ptr(db),
                                                          _ptr(G), U, I, d, int(n_local), int(n_total), float(lr), float(reg),
                                                          float(reg_adv), _ptr(loss)))

    def adagrad_apply(self, w, acc, g, lr, n=None):
        """acc += g*g; w -= lr*g/sqrt(acc) wherever g != 0."""
        n = int(n) if n is not None else int(np.prod(_shape(w)))
        _lib.check(self.lib.tfrec_adagrad_apply(self.h, _ptr(w), _ptr(acc), _ptr(g), float(lr), n))

    def dns_select(self, P, Q, b, users, cand, out=None):
        """cand: int32 [n_pos, C] -> best-scored candidate per positive (model/DNSBPR.py:74-79)."""
        U, d = _shape(P)
        I = _shape(Q)[0]
        n, Cn = _shape(cand)
        out = out if out is not None else self.empty(n, np.int32)
        _lib.check(self.lib.tfrec_dns_select(self.h, _ptr(P), _ptr(Q), _ptr(b), U, I, d, _ptr(users), _ptr(cand), n, Cn, _ptr(out)))
        return out

    # -- evaluation -------------------------------------------------------------------------
    def score_topk(self, P, Q, b, eval_users, k, train=None, mode=SCORE_F32, item_range=None, out_ids=None, out_scores=None):
        """train: (indptr int64 dev, indices int32 dev) or None.  Returns (ids [n,k], scores [n,k])."""
        d = _shape(P)[1]
        I = _shape(Q)[0]
        n = _shape(eval_users)[0]
        lo, hi = item_range if item_range is not None else (0, I)
        out_ids = out_ids if out_ids is not None else self.empty((n, k), np.int32)
        out_scores = out_scores if out_scores is not None else self.empty((n, k), np.float32)
        tp, ti = (train if train is not None else (None, None))
        _lib.check(self.lib.tfrec_score_topk(self.h, _ptr(P), _ptr(Q), _ptr(b), I, d, lo, hi, _ptr(eval_users), n, _ptr(tp), _ptr(ti),
                                             int(k), int(mode), _ptr(out_ids), _ptr(out_scores)))
        return out_ids, out_scores

    def score_topk_shard(self, P_rows, Q_shard, b_shard, item_id_lo, eval_users, k, train=None, mode=SCORE_F32,
                         out_ids=None, out_scores=None):
        """One item shard's top-k: P_rows[r] is the factor row of eval_users[r]; ids are global."""
        n, d = _shape(P_rows)
        ni = _shape(Q_shard)[0]
        out_ids = out_ids if out_ids is not None else self.empty((n, k), np.int32)
        out_scores = out_scores if out_scores is not None else self.empty((n, k), np.float32)
        tp, ti = (train if train is not None else (None, None))
        _lib.check(self.lib.tfrec_score_topk_shard(self.h, _ptr(P_rows), _ptr(Q_shard), _ptr(b_shard), ni, int(item_id_lo), d,
                                                   _ptr(eval_users), n, _ptr(tp), _ptr(ti), int(k), int(mode), _ptr(out_ids),
                                                   _ptr(out_scores)))
        return out_ids, out_scores

    def topk_merge(self, ids, scores, G, n, k, out_ids=None, out_scores=None):
        out_ids = out_ids if out_ids is not None else self.empty((n, k), np.int32)
        out_scores = out_scores if out_scores is not None else self.empty((n, k), np.float32)
        _lib.check(self.lib.tfrec_topk_merge(self.h, _ptr(ids), _ptr(scores), int(G), int(n), int(k), _ptr(out_ids), _ptr(out_scores)))
        return out_ids, out_scores

    def topk_fill_masked(self, ids, eval_users, k, train):
        n = _shape(eval_users)[0]
        _lib.check(self.lib.tfrec_topk_fill_masked(self.h, _ptr(ids), _ptr(eval_users), n, int(k), _ptr(train[0]), _ptr(train[1])))

    def topk_dense(self, R, k, mask=None, mask_rows=None, out_ids=None, out_scores=None):
        rows, cols = _shape(R)
        out_ids = out_ids if out_ids is not None else self.empty((rows, k), np.int32)
        out_scores = out_scores if out_scores is not None else self.empty((rows, k), np.float32)
        mp, mi = (mask if mask is not None else (None, None))
        _lib.check(self.lib.tfrec_topk_dense(self.h, _ptr(R), rows, cols, int(k), _ptr(mask_rows), _ptr(mp), _ptr(mi),
                                             _ptr(out_ids), _ptr(out_scores)))
        return out_ids, out_scores

    def rank_metrics(self, ids, eval_users, k, test, mean_mode=0, per_user=None):
        """test: (indptr, indices) device CSR.  Returns float32 [5,k]: Precision, Recall, MAP, NDCG, MRR."""
        n = _shape(eval_users)[0]
        out = np.empty((5, k), np.float32)
        _lib.check(self.lib.tfrec_rank_metrics(self.h, _ptr(ids), _ptr(eval_users), n, int(k), _ptr(test[0]), _ptr(test[1]),
                                               int(mean_mode), _ptr(per_user), out.ctypes.data))
        return out

    def rank_metrics_sums(self, ids, eval_users, k, test, sums=None, per_user=None):
        n = _shape(eval_users)[0]
        sums = sums if sums is not None else self.empty(5 * k, np.float64)
        _lib.check(self.lib.tfrec_rank_metrics_sums(self.h, _ptr(ids), _ptr(eval_users), n, int(k), _ptr(test[0]), _ptr(test[1]),
                                                    _ptr(per_user), _ptr(sums)))
        return sums

    def eval_fullrank(self, P, Q, b, eval_users, train, test, k, mode=SCORE_F32, mean_mode=0, out_ids=None):
        d = _shape(P)[1]
        I = _shape(Q)[0]
        n = _shape(eval_users)[0]
        out = np.empty((5, k), np.float32)
        tp, ti = (train if train is not None else (None, None))
        _lib.check(self.lib.tfrec_eval_fullrank(self.h, _ptr(P), _ptr(Q), _ptr(b), I, d, _ptr(eval_users), n, _ptr(tp), _ptr(ti),
                                                _ptr(test[0]), _ptr(test[1]), int(k), int(mode), int(mean_mode), _ptr(out_ids),
                                                out.ctypes.data))
        return out
