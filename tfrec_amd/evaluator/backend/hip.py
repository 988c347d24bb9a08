"""Evaluator backend on the GPU, with the reference backend's entry point
(evaluator/backend/cpp/evaluate.py:10-44) plus a factor-form entry that never builds the
dense matrix."""
import numpy as np
from scipy.sparse import csr_matrix

from ...device import Device, SCORE_F32
from ...utils.tools import typeassert

_default_device = None


def default_device():
    global _default_device
    if _default_device is None:
        _default_device = Device(0)
    return _default_device


def _dict_to_csr(d, n_rows):
    """{row: items} -> (indptr int64 [n_rows+1], indices int32, rows sorted ascending inside)."""
    indptr = np.zeros(n_rows + 1, np.int64)
    for u, items in d.items():
        indptr[u + 1] = len(items)
    np.cumsum(indptr, out=indptr)
    indices = np.empty(int(indptr[-1]), np.int32)
    for u, items in d.items():
        indices[indptr[u]:indptr[u + 1]] = np.sort(np.asarray(items, dtype=np.int32))
    return indptr, indices


@typeassert(rating_matrix=np.ndarray, user_pos_train=dict, user_pos_test=dict, top_k=int, thread_num=int)
def eval_rating_matrix(rating_matrix, user_pos_train, user_pos_test, top_k=50, thread_num=None, device=None, mean_mode=1):
    """Drop-in for the reference's ``eval_rating_matrix``: float32 [5, top_k] with rows
    Precision, Recall, MAP, NDCG, MRR.  ``thread_num`` is accepted and ignored.

    Deviations from the reference, on purpose: the caller's matrix is NOT modified (the
    reference writes -inf into the train positions, evaluate.py:27); equal scores rank by
    ascending item id (the reference leaves their order to std::partial_sort_copy)."""
    dev = device or default_device()
    R = np.ascontiguousarray(rating_matrix, dtype=np.float32)
    n_rows = R.shape[0]
    users = np.array(sorted(user_pos_test.keys()), dtype=np.int32)          # evaluate.py:30
    if len(users) == 0:
        raise ValueError("no test users")
    rows = R if len(R) == len(users) else R[users]                           # evaluate.py:37-38
    hi = int(max(n_rows, users.max() + 1, max(user_pos_train.keys(), default=0) + 1 if user_pos_train else 0))
    test = _dict_to_csr(user_pos_test, hi)
    dR = dev.array(rows)
    dusers = dev.array(users)
    dtest = (dev.array(test[0]), dev.array(test[1]))
    mask = None
    if user_pos_train is not None:
        tr = _dict_to_csr(user_pos_train, hi)
        mask = (dev.array(tr[0]), dev.array(tr[1]))
    ids, _ = dev.topk_dense(dR, top_k, mask=mask, mask_rows=dusers if mask is not None else None)
    if mask is not None:
        dev.topk_fill_masked(ids, dusers, top_k, mask)
    return dev.rank_metrics(ids, dusers, top_k, dtest, mean_mode=mean_mode)


def eval_factors(P, Q, b, train_csr, test_csr, test_users, top_k=50, device=None, mode=SCORE_F32, mean_mode=1):
    """Factor-form evaluation: P, Q, b are device buffers; train_csr/test_csr are pairs of device
    arrays (indptr int64, indices int32) over all users; test_users a device int32 array."""
    dev = device or default_device()
    return dev.eval_fullrank(P, Q, b, test_users, train_csr, test_csr, top_k, mode=mode, mean_mode=mean_mode)
