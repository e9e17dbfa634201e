import numpy as np

import dsgd_amd


def test_generator_shape_and_determinism():
    a = dsgd_amd.synth.generate(5000, seed=0)
    b = dsgd_amd.synth.generate(5000, seed=0)
    assert a.n_rows == 5000 and a.dim == 47236
    for x, y in ((a.row_ptr, b.row_ptr), (a.col, b.col), (a.val, b.val), (a.label, b.label)):
        np.testing.assert_array_equal(x, y)
    nnz = np.diff(a.row_ptr)
    assert nnz.min() >= 1 and nnz.max() <= 1200 and 60 < nnz.mean() < 90
    assert a.col.min() >= 1 and a.col.max() <= 47236
    # ascending, duplicate-free columns per row; unit L2 norm per row
    for i in range(0, 5000, 97):
        c = a.col[a.row_ptr[i]:a.row_ptr[i + 1]]
        assert np.all(np.diff(c) > 0)
        v = a.val[a.row_ptr[i]:a.row_ptr[i + 1]].astype(np.float64)
        assert abs((v * v).sum() - 1.0) < 1e-5
    assert 0.40 < (a.label == 1).mean() < 0.54
    assert set(np.unique(a.label)) == {-1, 1}


def test_generator_shards_are_slices_of_one_stream():
    whole = dsgd_amd.synth.generate(1000, seed=3)
    part = dsgd_amd.synth.generate(300, seed=3, row0=500)
    sub = whole.rows(500, 800)
    np.testing.assert_array_equal(part.row_ptr, sub.row_ptr)
    np.testing.assert_array_equal(part.col, sub.col)
    np.testing.assert_array_equal(part.val, sub.val)
    np.testing.assert_array_equal(part.label, sub.label)


def test_algorithmic_bytes():
    a = dsgd_amd.synth.generate(2000, seed=1)
    assert abs(a.algorithmic_bytes_per_row() - (8.0 * a.nnz / 2000 + 12)) < 1e-9
