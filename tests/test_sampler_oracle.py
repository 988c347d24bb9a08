"""CPU tests of the oracle's statement of the device epoch sampler: the rules the reference's host
sampler obeys (model/BPR.py:50-62, utils/tools.py:39-49: per stored interaction a uniform draw, with
replacement, from the items outside the user's row) hold, and the draws are a pure function of
(seed, stream).  The numpy stream itself is not reproduced (parity with it is unpinned by design:
it cannot be drawn in parallel); the HIP kernels are checked against this oracle bit for bit in
tests/test_gpu_sampler.py."""
import numpy as np


def _rows(rng, U, I, lo, hi):
    rows = [np.sort(rng.choice(I, int(rng.integers(lo, hi + 1)), replace=False)).astype(np.int32) for _ in range(U)]
    indptr = np.zeros(U + 1, np.int64)
    indptr[1:] = np.cumsum([len(r) for r in rows])
    return indptr, (np.concatenate(rows) if indptr[-1] else np.zeros(0, np.int32)), rows


def test_negatives_are_outside_the_row_and_reproducible(oracle):
    rng = np.random.default_rng(0)
    U, I = 300, 97
    indptr, indices, rows = _rows(rng, U, I, 0, 60)
    u, p, n, full = oracle.sample_negatives(indptr, indices, U, I, 3, 11, 4)
    assert full == 0 and n.shape == (len(indices), 3)
    assert np.array_equal(p, indices)
    assert np.array_equal(u, np.repeat(np.arange(U), np.diff(indptr)).astype(np.int32))
    for uu in range(U):
        got = n[indptr[uu]:indptr[uu + 1]].ravel()
        assert got.size == 0 or (got.min() >= 0 and got.max() < I)
        assert not np.isin(got, rows[uu]).any()
    u2, p2, n2, _ = oracle.sample_negatives(indptr, indices, U, I, 3, 11, 4)
    assert np.array_equal(n, n2)
    assert not np.array_equal(n, oracle.sample_negatives(indptr, indices, U, I, 3, 11, 5)[2])      # another epoch
    assert not np.array_equal(n, oracle.sample_negatives(indptr, indices, U, I, 3, 12, 4)[2])      # another seed


def test_negatives_are_uniform_over_the_complement(oracle):
    """One user with 10 of 50 items, 40,000 draws: chi-square over the 40 allowed items."""
    I = 50
    row = np.arange(0, 50, 5, dtype=np.int32)
    indptr = np.array([0, len(row)], np.int64)
    n = oracle.sample_negatives(indptr, row, 1, I, 4000, 3, 0)[2].ravel()
    counts = np.bincount(n, minlength=I)
    assert counts[row].sum() == 0
    allowed = np.setdiff1d(np.arange(I), row)
    expect = n.size / len(allowed)
    chi2 = ((counts[allowed] - expect) ** 2 / expect).sum()
    assert chi2 < 80.0        # 39 degrees of freedom: P(chi2 > 80) < 1e-4


def test_nearly_full_and_full_rows(oracle):
    I = 9
    rows = [np.array([0, 1, 2, 3, 4, 5, 6, 8], np.int32), np.arange(9, dtype=np.int32), np.array([4], np.int32)]
    indptr = np.array([0, 8, 17, 18], np.int64)
    u, p, n, full = oracle.sample_negatives(indptr, np.concatenate(rows), 3, I, 2, 1, 0)
    assert full == 1
    assert (n[0:8] == 7).all()                 # the only free item, reached by rejection or by the upward walk
    assert (n[8:17] == 0).all()                # nothing to draw: reported, zeros written
    assert (n[17] != 4).all()


def test_shuffle_is_a_reproducible_permutation(oracle):
    for n in (1, 2, 1000, 70001):
        p = oracle.shuffle_perm(n, 9, 3)
        assert np.array_equal(np.sort(p), np.arange(n))
        assert np.array_equal(p, oracle.shuffle_perm(n, 9, 3))
    a, b = oracle.shuffle_perm(1000, 9, 3), oracle.shuffle_perm(1000, 9, 4)
    assert not np.array_equal(a, b)
    # no position is favoured: mean displacement of a uniform permutation is n/3
    p = oracle.shuffle_perm(70001, 9, 3).astype(np.int64)
    assert abs(np.abs(p - np.arange(70001)).mean() / 70001 - 1 / 3) < 0.01


def test_row_shards_draw_what_the_whole_matrix_draws(oracle):
    rng = np.random.default_rng(3)
    U, I = 200, 150
    indptr, indices, _ = _rows(rng, U, I, 0, 30)
    _, _, whole, _ = oracle.sample_negatives(indptr, indices, U, I, 4, 8, 2)
    for lo, hi in ((0, 70), (70, 71), (71, 200)):
        sub_ptr = (indptr[lo:hi + 1] - indptr[lo]).astype(np.int64)
        sub_idx = indices[indptr[lo]:indptr[hi]]
        u, p, n, _ = oracle.sample_negatives(sub_ptr, sub_idx, hi - lo, I, 4, 8, 2, first=int(indptr[lo]))
        assert np.array_equal(n, whole[indptr[lo]:indptr[hi]])
        assert np.array_equal(u, np.repeat(np.arange(hi - lo), np.diff(sub_ptr)).astype(np.int32))
