"""Epochs of BPR triples built on the GPU (no counterpart in the reference, whose epochs are built on
the host by model/BPR.py:50-62 + utils/tools.py:39-49 + data/dataiterator.py:52-54).

Same distribution and exclusion rule as the reference -- per stored interaction one negative (or
``n_neg``) drawn uniformly, with replacement, from the items the user has not interacted with; then
a shuffle -- but the random numbers are the library's counter-based draws, not numpy's global
stream, which cannot be drawn in parallel.  Epoch ``e`` of a sampler is a pure function of
``(seed, e)`` and of the training matrix.

Who is trained is the reference's rule: the users ``csr_to_user_dict`` keeps (utils/tools.py:52-62 drops a user
whose only item is item 0).  A user who holds every item has no negative to draw: the library reports it through
its sticky error flag, so the next ``synchronize`` raises instead of training on a positive."""
import numpy as np

from ..utils.tools import csr_trained_rows


class DeviceEpochSampler:
    def __init__(self, dev, train_matrix, seed=2018, n_neg=1, first_interaction=0):
        csr = csr_trained_rows(train_matrix)
        self.dev = dev
        self.n_users, self.n_items = csr.shape
        self.nnz = int(csr.nnz)
        self.n_neg = int(n_neg)
        self.seed = int(seed)
        self.first_interaction = int(first_interaction)    # position of row 0's first interaction in the whole set (row shards)
        self.indptr = dev.array(csr.indptr.astype(np.int64), np.int64)
        self.indices = dev.array(csr.indices.astype(np.int32), np.int32)
        # outputs in storage order (made when first needed: the one-negative epoch never holds them), reused every epoch;
        # shuffled copies are what the steps read
        self._users = self._pos = self._neg = self._perm = None
        self.users = dev.empty(self.nnz, np.int32)
        self.pos = dev.empty(self.nnz, np.int32)
        self.neg = dev.empty((self.nnz, self.n_neg), np.int32)

    def epoch(self, e, shuffle=True):
        """Device arrays (users, pos, neg) of epoch ``e``: int32 [nnz], [nnz], [nnz, n_neg].
        Streams 2e and 2e+1 of the seed feed the draws and the shuffle."""
        dev = self.dev
        if shuffle and self.n_neg == 1:
            # one call: the draws stay packed (user, item, negative) until the shuffle has moved them
            dev.epoch_triples(self.indptr, self.indices, self.n_users, self.n_items, self.seed, 2 * e, 2 * e + 1,
                              users_out=self.users, pos_out=self.pos, neg_out=self.neg, first_interaction=self.first_interaction)
            return self.users, self.pos, self.neg
        if self._users is None:
            self._users = dev.empty(self.nnz, np.int32)
            self._pos = dev.empty(self.nnz, np.int32)
            self._neg = dev.empty((self.nnz, self.n_neg), np.int32)
            self._perm = dev.empty(self.nnz, np.int32)
        dev.sample_negatives(self.indptr, self.indices, self.n_users, self.n_items, self.n_neg, self.seed, 2 * e,
                             users_out=self._users, pos_out=self._pos, neg_out=self._neg, first_interaction=self.first_interaction)
        if not shuffle:
            return self._users, self._pos, self._neg
        dev.shuffle_perm(self.nnz, self.seed, 2 * e + 1, out=self._perm)
        dev.gather_rows_i32(self._users, self._perm, out=self.users)
        dev.gather_rows_i32(self._pos, self._perm, out=self.pos)
        dev.gather_rows_i32(self._neg, self._perm, out=self.neg)
        return self.users, self.pos, self.neg
