"""Minibatch iteration over parallel columns (data/dataiterator.py:158-192).

Same observable behaviour as the reference's DataIterator: with ``shuffle=True`` every
``__iter__`` draws ONE ``np.random.permutation(n)`` from numpy's global generator
(data/dataiterator.py:52-54), batches are consecutive chunks of that order
(data/dataiterator.py:87-95) and the last short batch is kept unless ``drop_last``.
Iterating yields one list per column, like the reference; ``epoch_order()`` and
``columns_in_epoch_order()`` expose the same epoch as arrays, which is what the device path
uploads once per epoch instead of feeding 512 triples at a time.
"""
import numpy as np


class DataIterator(object):
    def __init__(self, *data, batch_size=1, shuffle=False, drop_last=False):
        if not data:
            raise ValueError("at least one column is required")
        n = len(data[0])
        for col in data:
            if len(col) != n:
                raise AssertionError("all columns must have the same length")
        if not isinstance(batch_size, int) or isinstance(batch_size, bool) or batch_size <= 0:
            raise ValueError("batch_size should be a positive integeral value, but got batch_size={}".format(batch_size))
        if not isinstance(drop_last, bool):
            raise ValueError("drop_last should be a boolean value, but got drop_last={}".format(drop_last))
        self.columns = [np.asarray(col) for col in data]
        self.n = n
        self.batch_size = batch_size
        self.shuffle = bool(shuffle)
        self.drop_last = drop_last

    def __len__(self):
        if self.drop_last:
            return self.n // self.batch_size
        return (self.n + self.batch_size - 1) // self.batch_size

    def epoch_order(self):
        """Sample order of one epoch (consumes one permutation from np.random when shuffling)."""
        order = np.random.permutation(self.n) if self.shuffle else np.arange(self.n)
        if self.drop_last:
            order = order[: (self.n // self.batch_size) * self.batch_size]
        return order

    def columns_in_epoch_order(self):
        """(columns reordered for one epoch, batch boundaries): batch k is rows bounds[k]:bounds[k+1]."""
        order = self.epoch_order()
        bounds = list(range(0, len(order), self.batch_size)) + [len(order)]
        return [col[order] for col in self.columns], bounds

    def __iter__(self):
        cols, bounds = self.columns_in_epoch_order()
        for k in range(len(bounds) - 1):
            batch = [col[bounds[k]:bounds[k + 1]].tolist() for col in cols]
            yield batch[0] if len(batch) == 1 else batch
