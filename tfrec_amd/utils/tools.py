"""Host-side helpers with the reference's names and behaviour (utils/tools.py)."""
import functools
import heapq
import inspect
import time

import numpy as np
from scipy.sparse import csr_matrix


def typeassert(*type_args, **type_kwargs):
    """Decorator checking argument types by name, raising TypeError('Argument x must be T')
    like the reference's (utils/tools.py:10-24).  Arguments left at None are accepted when the
    reference's own callers rely on that (thread_num=None in eval_rating_matrix raises there;
    here None always means "not given")."""
    def decorate(func):
        sig = inspect.signature(func)
        wanted = sig.bind_partial(*type_args, **type_kwargs).arguments

        @functools.wraps(func)
        def wrapper(*args, **kwargs):
            for name, value in sig.bind(*args, **kwargs).arguments.items():
                if name in wanted and value is not None and not isinstance(value, wanted[name]):
                    raise TypeError("Argument {} must be {}".format(name, wanted[name]))
            return func(*args, **kwargs)
        return wrapper
    return decorate


def timer(func):
    """Prints the wall time of a call (utils/tools.py:27-36)."""
    @functools.wraps(func)
    def inner(*args, **kwargs):
        t0 = time.time()
        out = func(*args, **kwargs)
        print("%s function cost: %fs" % (func.__name__, time.time() - t0))
        return out
    return inner


def random_choice(a, size=None, replace=True, p=None, exclusion=None):
    """np.random.choice with some entries excluded (utils/tools.py:39-49): the probability
    vector gets zeros at ``exclusion`` and is renormalised; draws come from numpy's global
    legacy generator, so a run seeded like the reference (np.random.seed(2018)) draws the
    same negatives."""
    if exclusion is not None:
        p = np.ones_like(a) if p is None else np.array(p, copy=True)
        p = np.ndarray.flatten(p)
        p[exclusion] = 0
        p = p / np.sum(p)
    return np.random.choice(a, size=size, replace=replace, p=p)


@typeassert(sparse_matrix_data=csr_matrix)
def csr_to_user_dict(sparse_matrix_data):
    """{row: indices} for the rows of a CSR matrix (utils/tools.py:52-62).  The reference keeps
    a row only if ``any(indices)`` is true, which drops empty rows and also rows whose only
    item is item 0; that behaviour is kept because it decides who is trained and evaluated."""
    indptr, indices = sparse_matrix_data.indptr, sparse_matrix_data.indices
    out = {}
    for row in range(sparse_matrix_data.shape[0]):
        items = indices[indptr[row]:indptr[row + 1]]
        if items.size and items.any():
            out[row] = items
    return out


def csr_trained_rows(sparse_matrix_data):
    """The matrix with the rows ``csr_to_user_dict`` drops emptied (a row whose only item is item 0), sorted
    indices: what a sampler that works on the CSR arrays themselves must see to train the same users as the
    reference's dictionary-based one (utils/tools.py:52-62, model/BPR.py:50-62)."""
    csr = sparse_matrix_data.tocsr().copy()
    csr.sort_indices()
    counts = np.diff(csr.indptr)
    first = csr.indices[np.minimum(csr.indptr[:-1], max(csr.nnz - 1, 0))] if csr.nnz else np.zeros(0, csr.indices.dtype)
    drop = (counts == 1) & (first == 0) if csr.nnz else np.zeros(csr.shape[0], bool)
    if drop.any():
        keep = np.ones(csr.nnz, bool)
        keep[csr.indptr[:-1][drop]] = False
        new_counts = np.where(drop, 0, counts)
        csr = csr_matrix((csr.data[keep], csr.indices[keep], np.concatenate([[0], np.cumsum(new_counts)])), shape=csr.shape)
    return csr


@typeassert(sparse_matrix_data=csr_matrix)
def csr_to_user_item_pair(sparse_matrix_data):
    """Parallel lists of (user, item) over the non-zeros (utils/tools.py:65-72)."""
    counts = np.diff(sparse_matrix_data.indptr)
    users = np.repeat(np.arange(sparse_matrix_data.shape[0]), counts).tolist()
    return users, sparse_matrix_data.indices.tolist()


def argmax_top_k(a, top_k=50):
    """Indices of the top_k largest entries; ties go to the LARGER index, as the reference's
    heapq over (value, index) pairs does (utils/tools.py:75-77)."""
    best = heapq.nlargest(top_k, zip(a, range(len(a))))
    return np.array([i for _, i in best], dtype=np.intc)
