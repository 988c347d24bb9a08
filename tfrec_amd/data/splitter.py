"""Raw interaction file -> filtered, re-numbered train / test files (data/splitter.py:8-50,
data/utils.py:13-98; driven by split_data.py with split_data.ini).

Host-side preparation, outside the GPU path.  The reference keeps one Python object per line and
per user; here the file is a set of integer columns and every user a slice of one lexicographic
sort.  The random splits consume numpy's legacy generator exactly as the reference does (one
``np.random.shuffle`` of the user's lines per user, users in ascending order), so a seeded run
writes the same files.
"""
import numpy as np

COLUMNS = {"uirt": ("user", "item", "rating", "time"), "uir": ("user", "item", "rating")}


def load_columns(filename, sep, names):
    """Integer columns of a delimited text file, one 1-D array per name."""
    raw = np.loadtxt(filename, delimiter=sep, ndmin=2, usecols=range(len(names)))
    return {n: raw[:, c].astype(np.int64) for c, n in enumerate(names)}


def _keep(cols, mask):
    return {k: v[mask] for k, v in cols.items()}


def filter_columns(cols, user_min=None, item_min=None):
    """Drops items with fewer than ``item_min`` lines, THEN users with fewer than ``user_min`` of the
    remaining lines (data/utils.py:13-29: one pass each, in that order)."""
    if item_min is not None and item_min > 0:
        _, inv, cnt = np.unique(cols["item"], return_inverse=True, return_counts=True)
        cols = _keep(cols, cnt[inv] >= item_min)
    if user_min is not None and user_min > 0:
        _, inv, cnt = np.unique(cols["user"], return_inverse=True, return_counts=True)
        cols = _keep(cols, cnt[inv] >= user_min)
    return cols


def remap_columns(cols):
    """Users and items re-numbered 0.. in ascending order of their raw ids (data/utils.py:32-44).
    Returns (columns, raw user ids by new id, raw item ids by new id)."""
    users, uinv = np.unique(cols["user"], return_inverse=True)
    items, iinv = np.unique(cols["item"], return_inverse=True)
    out = dict(cols)
    out["user"], out["item"] = uinv.astype(np.int64), iinv.astype(np.int64)
    return out, users, items


def _sorted_by_user(cols, by_time):
    # numpy's sort(order=["user", x]) of a record array breaks ties by the remaining fields in column order
    lead = ["user", "time" if by_time else "item"]
    keys = lead + [n for n in cols if n not in lead]
    order = np.lexsort(tuple(cols[n] for n in reversed(keys)))
    cols = _keep(cols, order)
    bounds = np.flatnonzero(np.diff(cols["user"], prepend=-1, append=-1))   # start of every user's run, and the end
    return cols, bounds


def _split(cols, by_time, n_train_of):
    """Per user (ascending): optionally shuffle the user's lines, the first n_train_of(len) go to the
    first section, the rest to the second; ``n_train_of`` returning None keeps everything in the first
    without shuffling."""
    cols, bounds = _sorted_by_user(cols, by_time)
    first, second = [], []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        n = int(hi - lo)
        cut = n_train_of(n)
        idx = np.arange(lo, hi)
        if cut is None:
            first.append(idx)
            continue
        if not by_time:
            np.random.shuffle(idx)                      # same draws and swaps as shuffling the lines themselves
        first.append(idx[:cut])
        second.append(idx[cut:])
    take = lambda parts: _keep(cols, np.concatenate(parts) if parts else np.zeros(0, np.int64))
    return take(first), take(second)


def split_by_ratio(cols, ratio=0.8, by_time=True):
    """The first ceil(ratio * n) lines of every user (by time, or in shuffled order) train, the rest test
    (data/utils.py:47-72)."""
    return _split(cols, by_time, lambda n: int(np.ceil(ratio * n)))


def split_by_loo(cols, by_time=True):
    """Leave one out: the last line of every user with at least three lines tests (data/utils.py:75-98)."""
    return _split(cols, by_time, lambda n: None if n < 3 else n - 1)


def as_table(cols, names):
    return np.stack([cols[n] for n in names], axis=1) if len(cols[names[0]]) else np.zeros((0, len(names)), np.int64)


class Splitter(object):
    """``config``: the [split] section of split_data.ini as a dict of strings (data/splitter.py:9-17):
    data_file, ratio, file_format, separator, user_min, item_min, by_time, splitter."""

    def __init__(self, config):
        if not isinstance(config, dict):
            raise TypeError("Argument config must be <class 'dict'>")
        from ..model.ModelFactory import parse_value      # literal parsing of ini values (the reference eval()s them)
        self.filename = config["data_file"]
        self.ratio = parse_value(config["ratio"])
        self.file_format = config["file_format"]
        self.sep = parse_value(config["separator"])
        self.user_min = parse_value(config["user_min"])
        self.item_min = parse_value(config["item_min"])
        self.by_time = parse_value(config["by_time"])
        self.spliter = config["splitter"]

    def split(self):
        fmt = self.file_format.lower()
        if fmt not in COLUMNS:
            raise ValueError("There is not data format '%s'" % self.file_format)
        names = COLUMNS[fmt]
        by_time = fmt == "uirt" and self.by_time is not False
        cols = load_columns(self.filename, self.sep, names)
        cols = filter_columns(cols, user_min=self.user_min, item_min=self.item_min)
        cols, users, items = remap_columns(cols)
        if self.spliter == "ratio":
            train, test = split_by_ratio(cols, ratio=float(self.ratio), by_time=by_time)
        elif self.spliter == "loo":
            train, test = split_by_loo(cols, by_time=by_time)
        else:
            raise ValueError("There is not splitter '%s'" % self.spliter)
        np.savetxt(self.filename + ".train", as_table(train, names), fmt="%d", delimiter="\t")
        np.savetxt(self.filename + ".test", as_table(test, names), fmt="%d", delimiter="\t")
        np.savetxt(self.filename + ".user2id", np.stack([users, np.arange(len(users))], axis=1), fmt="%d", delimiter="\t")
        np.savetxt(self.filename + ".item2id", np.stack([items, np.arange(len(items))], axis=1), fmt="%d", delimiter="\t")
        return train, test
