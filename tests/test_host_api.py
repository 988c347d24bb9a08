"""CPU tests of the host-side mirror of the reference interface (no GPU, no library calls)."""
import hashlib
import os
import pickle

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import csr_from_pairs
from tfrec_amd.data import DataIterator
from tfrec_amd.model.ModelFactory import parse_value
from tfrec_amd.utils import argmax_top_k, csr_to_user_dict, random_choice, typeassert
from tfrec_amd.utils.params_io import load_params, read_param_pickle_safely, save_params


def test_sampler_and_iterator_reproduce_the_reference_epoch(golden_sampler, golden_ml100k):
    """utils.random_choice + DataIterator under np.random.seed(2018) give the reference's triples,
    batch order and batch count bit for bit (vectors made by the reference's own code)."""
    g, m = golden_sampler, golden_ml100k
    U, I = m["P"].shape[0], m["Q"].shape[0]
    user_pos = csr_to_user_dict(csr_from_pairs(m["train_u"], m["train_i"], U, I))
    all_items = np.arange(I)
    np.random.seed(2018)
    for ep in range(2):
        users, pos, neg = [], [], []
        for u, p in user_pos.items():
            ng = random_choice(all_items, size=len(p), exclusion=p)
            users += [u] * len(p); pos += p.tolist(); neg += ng.tolist()
        assert np.array_equal(np.array(neg), g["epoch%d_neg" % ep].astype(np.int64))
        it = DataIterator(users, pos, neg, batch_size=512, shuffle=True)
        batches = list(it)
        assert len(batches) == len(it) == int(g["epoch%d_n_batches" % ep])
        assert len(batches[-1][0]) == int(g["epoch%d_last_len" % ep])
        assert isinstance(batches[0][0], list)
        flat = np.concatenate([np.array(bt, dtype=np.int32) for bt in batches], axis=1)
        assert hashlib.sha256(flat.tobytes()).hexdigest() == str(g["epoch%d_batched_sha" % ep])
        assert np.array_equal(np.array(batches[0], np.int32), g["epoch%d_first_batch" % ep])


def test_iterator_epoch_arrays_match_iteration():
    np.random.seed(5)
    it = DataIterator(list(range(10)), list(range(10, 20)), batch_size=3, shuffle=True)
    np.random.seed(9)
    a = list(it)
    np.random.seed(9)
    cols, bounds = it.columns_in_epoch_order()
    assert bounds == [0, 3, 6, 9, 10]
    for k, bt in enumerate(a):
        assert bt[0] == cols[0][bounds[k]:bounds[k + 1]].tolist() and bt[1] == cols[1][bounds[k]:bounds[k + 1]].tolist()
    it2 = DataIterator(list(range(10)), batch_size=3, shuffle=False, drop_last=True)
    assert list(it2) == [[0, 1, 2], [3, 4, 5], [6, 7, 8]] and len(it2) == 3
    with pytest.raises(ValueError):
        DataIterator([1, 2], batch_size=0)
    with pytest.raises(AssertionError):
        DataIterator([1, 2], [1], batch_size=1)


def test_csr_to_user_dict_quirk(golden_sampler):
    quirk = sp.csr_matrix((np.ones(4), ([1, 2, 2, 3], [0, 0, 3, 2])), shape=(4, 5))
    assert sorted(csr_to_user_dict(quirk).keys()) == golden_sampler["quirk_kept_rows"].tolist()
    with pytest.raises(TypeError):
        csr_to_user_dict(np.zeros((2, 2)))


def test_csr_trained_rows_keeps_exactly_the_users_of_csr_to_user_dict(golden_sampler):
    """The CSR the device samplers draw from holds the rows csr_to_user_dict keeps, and only those."""
    from tfrec_amd.utils.tools import csr_trained_rows
    quirk = sp.csr_matrix((np.ones(4), ([1, 2, 2, 3], [0, 0, 3, 2])), shape=(4, 5))
    t = csr_trained_rows(quirk)
    assert np.flatnonzero(np.diff(t.indptr)).tolist() == golden_sampler["quirk_kept_rows"].tolist()
    rng = np.random.default_rng(0)
    for _ in range(50):
        U, I = 30, 6
        rows, cols = [], []
        for u in range(U):
            c = rng.permutation(I)[: int(rng.integers(0, 3))]
            rows += [u] * len(c); cols += c.tolist()
        m = sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(U, I)); m.sort_indices()
        t, kept = csr_trained_rows(m), csr_to_user_dict(m)
        for u in range(U):
            got = t.indices[t.indptr[u]:t.indptr[u + 1]]
            assert np.array_equal(got, kept[u]) if u in kept else got.size == 0
    assert csr_trained_rows(sp.csr_matrix((3, 4))).nnz == 0


def test_typeassert_and_argmax_top_k():
    @typeassert(a=int, b=str)
    def f(a, b="x", c=None):
        return a
    assert f(1) == 1 and f(1, b="y") == 1
    with pytest.raises(TypeError):
        f("1")
    with pytest.raises(TypeError):
        f(1, b=2)
    # equal values: the LARGER index comes first, as heapq over (value, index) does in the reference
    assert argmax_top_k(np.array([1.0, 3.0, 3.0, 2.0]), 3).tolist() == [2, 1, 3]


def test_parse_value():
    assert parse_value("0.1/16") == 0.1 / 16
    assert parse_value("'\\t'") == "\t"
    assert parse_value("dataset/ml100k_dnsbpr.pkl") == "dataset/ml100k_dnsbpr.pkl"
    assert parse_value("512") == 512 and parse_value("[1, 2]") == [1, 2] and parse_value("-0.5") == -0.5
    assert parse_value("__import__('os').system('true')") == "__import__('os').system('true')"


def test_param_files(tmp_path):
    rng = np.random.default_rng(0)
    P, Q, b = rng.normal(size=(7, 4)).astype(np.float32), rng.normal(size=(5, 4)).astype(np.float32), rng.normal(size=5).astype(np.float32)
    for name in ("m.npz", "m.pkl"):
        path = str(tmp_path / name)
        save_params(path, P, Q, b)
        for trusted in (False, True):
            got = load_params(path, trusted=trusted)
            assert all(np.array_equal(x, y) for x, y in zip(got, (P, Q, b)))
    # every pickle protocol numpy arrays are written with (2: array bytes as latin-1 text through _codecs.encode;
    # 3, 4: bytes in the state tuple; 5: a bytearray handed to _frombuffer), read without unpickling; shapes that
    # share an element count (U == I, I == U * d) must not be confused, float64 files are narrowed
    path = str(tmp_path / "old.pkl")
    for U, I, d in ((7, 5, 4), (5, 5, 4), (3, 12, 4), (1, 1, 1), (300, 200, 16)):
        for dt in (np.float32, np.float64):
            P2, Q2, b2 = rng.normal(size=(U, d)).astype(dt), rng.normal(size=(I, d)).astype(dt), rng.normal(size=I).astype(dt)
            for proto in (2, 3, 4, 5):
                with open(path, "wb") as f:
                    pickle.dump([P2, Q2, b2], f, protocol=proto)
                got = read_param_pickle_safely(path)
                for x, y in zip(got, (P2, Q2, b2)):
                    assert x.dtype == np.float32 and x.shape == y.shape and np.array_equal(x, y.astype(np.float32)), (U, I, d, dt, proto)
            # read-only arrays: protocol 5 then writes their buffers as bytes, not as a bytearray
            for a in (P2, Q2, b2):
                a.setflags(write=False)
            with open(path, "wb") as f:
                pickle.dump([P2, Q2, b2], f, protocol=5)
            for x, y in zip(read_param_pickle_safely(path), (P2, Q2, b2)):
                assert x.shape == y.shape and np.array_equal(x, y.astype(np.float32)), (U, I, d, dt, "read-only")
    # globals other than numpy's are refused under every protocol (4 and 5 name them by two texts and STACK_GLOBAL)
    bad = str(tmp_path / "bad.pkl")
    for proto in (2, 3, 4, 5):
        with open(bad, "wb") as f:
            pickle.dump([os.path.join, 1], f, protocol=proto)
        with pytest.raises(ValueError, match="not a plain numpy"):
            read_param_pickle_safely(bad)


def test_checkpoint_round_trip(tmp_path):
    """Arrays, epoch number and numpy's legacy generator state survive save_checkpoint/load_checkpoint;
    the file holds plain arrays only (it loads with allow_pickle=False)."""
    from tfrec_amd.utils import load_checkpoint, save_checkpoint
    rng = np.random.default_rng(0)
    arrays = {"P": rng.normal(size=(5, 3)).astype(np.float32), "b": rng.normal(size=7).astype(np.float32)}
    np.random.seed(99)
    np.random.random_sample(11)
    np.random.standard_normal(3)          # leaves a cached gaussian in the state
    state = np.random.get_state()
    want = np.random.random_sample(5)
    path = str(tmp_path / "ck.npz")
    save_checkpoint(path, arrays, 4, rng_state=state)
    got, epoch, st = load_checkpoint(path)
    assert epoch == 4 and set(got) == {"P", "b"}
    assert all(np.array_equal(got[k], arrays[k]) for k in arrays)
    np.random.set_state(st)
    assert np.array_equal(np.random.random_sample(5), want)
    save_checkpoint(path, arrays, 5)       # overwrite; no generator state (device sampler)
    _, epoch, st = load_checkpoint(path)
    assert epoch == 5 and st is None
    assert not [f for f in os.listdir(tmp_path) if "tmp" in f]


SPLIT_COLS = ("user", "item", "rating", "time")


@pytest.fixture(scope="module")
def golden_splitter():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "splitter.npz"))


@pytest.mark.parametrize("name,umin,imin", [("nofilter", 0, 0), ("filtered", 3, 2)])
def test_splitter_reproduces_the_reference_splits(golden_splitter, tmp_path, name, umin, imin):
    """Filtering, re-numbering and the four splits (ratio / leave-one-out, by time / shuffled with numpy's
    seeded legacy generator) equal what the reference's data/utils.py produced for the same file."""
    from tfrec_amd.data import splitter as sp_
    g = golden_splitter
    path = str(tmp_path / "raw.txt")
    np.savetxt(path, g["raw"], fmt="%d", delimiter=",")
    cols = sp_.filter_columns(sp_.load_columns(path, ",", SPLIT_COLS), user_min=umin, item_min=imin)
    cols, users, items = sp_.remap_columns(cols)
    assert np.array_equal(users, g[name + "_users"]) and np.array_equal(items, g[name + "_items"])
    assert np.array_equal(sp_.as_table(cols, SPLIT_COLS), g[name + "_remapped"])
    for by_time in (True, False):
        np.random.seed(7)
        tr, te = sp_.split_by_ratio(cols, ratio=0.8, by_time=by_time)
        assert np.array_equal(sp_.as_table(tr, SPLIT_COLS), g["%s_ratio_%d_train" % (name, by_time)])
        assert np.array_equal(sp_.as_table(te, SPLIT_COLS), g["%s_ratio_%d_test" % (name, by_time)])
        np.random.seed(7)
        tr, te = sp_.split_by_loo(cols, by_time=by_time)
        assert np.array_equal(sp_.as_table(tr, SPLIT_COLS), g["%s_loo_%d_train" % (name, by_time)])
        assert np.array_equal(sp_.as_table(te, SPLIT_COLS), g["%s_loo_%d_test" % (name, by_time)])


def test_splitter_files_feed_the_dataset_reader(golden_splitter, tmp_path):
    """Splitter.split() with the ini-style string config writes the four files; Dataset reads the pair back."""
    from tfrec_amd.data import Dataset, Splitter
    path = str(tmp_path / "raw.dat")
    np.savetxt(path, golden_splitter["raw"], fmt="%d", delimiter=",")
    cfg = {"data_file": path, "ratio": "0.8", "file_format": "UIRT", "separator": '","', "user_min": "3", "item_min": "2",
           "by_time": "True", "splitter": "loo"}
    tr, te = Splitter(cfg).split()
    g = golden_splitter
    assert np.array_equal(np.loadtxt(path + ".train", dtype=np.int64, ndmin=2), g["filtered_loo_1_train"])
    assert np.array_equal(np.loadtxt(path + ".test", dtype=np.int64, ndmin=2), g["filtered_loo_1_test"])
    u2i = np.loadtxt(path + ".user2id", dtype=np.int64, ndmin=2)
    assert np.array_equal(u2i[:, 0], g["filtered_users"]) and np.array_equal(u2i[:, 1], np.arange(len(u2i)))
    ds = Dataset({"data_name": "x", "train_file": path + ".train", "test_file": path + ".test", "format": "UIRT", "separator": "\t"})
    assert ds.train_matrix.nnz == len(tr["user"]) and ds.test_matrix.nnz == len(te["user"])
    assert ds.train_matrix.shape == (len(g["filtered_users"]), len(g["filtered_items"]))
    with pytest.raises(ValueError):
        Splitter(dict(cfg, splitter="kfold")).split()
    with pytest.raises(ValueError):
        Splitter(dict(cfg, file_format="XYZ")).split()
    with pytest.raises(TypeError):
        Splitter([("data_file", path)])
