"""Train/test interaction matrices from UIR / UIRT text files (data/dataset.py:5-46)."""
import numpy as np
from scipy.sparse import csr_matrix


class Dataset(object):
    """``config`` keys as in tfrec.ini: data_name, train_file, test_file, format (UIR|UIRT),
    separator.  Ids are taken as they are in the files (0-based, dense); the matrix shape is
    (#distinct users, #distinct items) over train and test together, like the reference."""

    def __init__(self, config):
        self.name = config.get("data_name")
        self.train_file = config.get("train_file")
        self.test_file = config.get("test_file")
        fmt = config.get("format")
        if fmt not in ("UIR", "UIRT"):
            raise ValueError("There is not data format '%s'" % fmt)
        ncol = 3 if fmt == "UIR" else 4
        sep = config.get("separator")
        train = self._read(self.train_file, sep, ncol)
        test = self._read(self.test_file, sep, ncol)
        both = np.concatenate([train, test])
        self.num_users = len(np.unique(both[:, 0]))
        self.num_items = len(np.unique(both[:, 1]))
        self.num_ratings = len(both)
        shape = (self.num_users, self.num_items)
        self.train_matrix = self._csr(train, shape)
        self.test_matrix = self._csr(test, shape)

    @staticmethod
    def _read(path, sep, ncol):
        data = np.loadtxt(path, delimiter=sep, ndmin=2, usecols=range(ncol))
        return data

    @staticmethod
    def _csr(data, shape):
        m = csr_matrix((data[:, 2], (data[:, 0].astype(np.int64), data[:, 1].astype(np.int64))), shape=shape)
        m.sort_indices()
        return m
