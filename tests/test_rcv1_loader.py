"""The RCV1 text loader against the reference's loader semantics (utils/Dataset.scala:13-60) -- CPU only."""

import os

import numpy as np
import pytest

import dsgd_amd
from dsgd_amd import rcv1


def write(folder, files, qrels):
    os.makedirs(folder, exist_ok=True)
    for name, text in files.items():
        with open(os.path.join(folder, name), "w") as f:
            f.write(text)
    with open(os.path.join(folder, rcv1.QRELS), "w") as f:
        f.write(qrels)


def test_round_trip_of_the_synthetic_workload(tmp_path):
    data = dsgd_amd.synth.generate(3000, seed=3)
    ids = rcv1.export(str(tmp_path), data, n_train_file=1000)
    back, back_ids = rcv1.load(str(tmp_path), full=True, with_ids=True)
    assert back.n_rows == 3000  # train file, then test_pt0..3, rows in file order (Dataset.scala:47-58)
    np.testing.assert_array_equal(back_ids, ids)
    np.testing.assert_array_equal(back.row_ptr, data.row_ptr)
    np.testing.assert_array_equal(back.col, data.col)
    np.testing.assert_array_equal(back.label, data.label)  # incl. "CCAT then another topic" => -1, "ECAT then CCAT" => +1
    np.testing.assert_array_equal(back.val, data.val)      # %.9g is an fp32 round trip
    only_train = rcv1.load(str(tmp_path), full=False)
    assert only_train.n_rows == 1000                       # full = false reads the train file only (:47)


def test_reference_quirks(tmp_path):
    files = {rcv1.FILES[0]: ("10  3:0.5 7:0.25\n"          # official form: two spaces, token 1 empty
                             "11 9:1.0 4:0.5 6:0.125\n"    # ONE space: `.drop(2)` loses the first feature (:27)
                             "12  5:1 5:2 2:3\n"           # `.toMap`: the last value of a repeated key wins (:32)
                             "13  8:1e-25 1:0.5:junk\n"    # values stay as written; parts after a 2nd ':' are ignored (:29-30)
                             "14  \n")}                    # no features at all: an empty vector
    qrels = ("CCAT 10 1\n"
             "CCAT 11 1\nGCAT 11 1\n"                       # last line per document wins (:53): -1
             "MCAT 12 1\nCCAT 12 1\n"                       # ... and here +1
             "ECAT 13 1\n"
             "CCAT 14 1\n"
             "CCAT 99 1\n")                                 # labels of documents that never show up are harmless
    write(str(tmp_path), files, qrels)
    d, ids = rcv1.load(str(tmp_path), full=False, with_ids=True)
    assert ids.tolist() == [10, 11, 12, 13, 14]
    assert d.label.tolist() == [1, -1, 1, -1, 1]
    rows = [dict(zip(d.col[d.row_ptr[i]:d.row_ptr[i + 1]].tolist(), d.val[d.row_ptr[i]:d.row_ptr[i + 1]].tolist()))
            for i in range(5)]
    assert rows[0] == {3: 0.5, 7: 0.25}
    assert rows[1] == {4: 0.5, 6: 0.125}
    assert rows[2] == {5: 2.0, 2: 3.0}
    assert rows[3] == {8: np.float32(1e-25), 1: 0.5}
    assert rows[4] == {}


@pytest.mark.parametrize("files,qrels,what", [
    ({rcv1.FILES[0]: "10  3:0.5\n"}, "ECAT 11 1\n", "no label"),                 # labels(id) throws (:58)
    ({rcv1.FILES[0]: "x  3:0.5\n"}, "CCAT 10 1\n", "malformed"),                 # parts(0).toInt
    ({rcv1.FILES[0]: "10  3=0.5\n"}, "CCAT 10 1\n", "malformed"),                # elems(1) missing
    ({rcv1.FILES[0]: "10  3:abc\n"}, "CCAT 10 1\n", "malformed"),                # toDouble
    ({rcv1.FILES[0]: " 10  3:0.5\n"}, "CCAT 10 1\n", "malformed"),               # leading space: parts(0) == ""
    ({rcv1.FILES[0]: "10  3:0.5\n"}, "CCAT\n", "malformed"),                     # parts(1) missing in the qrels
    ({}, "CCAT 10 1\n", "cannot open"),
])
def test_what_makes_the_reference_throw(tmp_path, files, qrels, what):
    write(str(tmp_path), files, qrels)
    with pytest.raises(ValueError) as ei:
        rcv1.load(str(tmp_path), full=False)
    assert what in str(ei.value)


def test_full_needs_all_five_files(tmp_path):
    write(str(tmp_path), {rcv1.FILES[0]: "10  3:0.5\n"}, "CCAT 10 1\n")
    assert rcv1.load(str(tmp_path), full=False).n_rows == 1
    with pytest.raises(ValueError):
        rcv1.load(str(tmp_path), full=True)


# ---- the product parser against the INDEPENDENT restatement (oracle/ref_loader.py) on LYRL2004-formatted text --------
SAMPLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lyrl2004_sample")


def canon(row_ptr, col, val):
    """rows as sorted (column, fp32 value) lists: a row is a Map in the reference, its iteration order is unspecified"""
    return [sorted(zip(col[row_ptr[i]:row_ptr[i + 1]].tolist(), np.asarray(val[row_ptr[i]:row_ptr[i + 1]], np.float32).tolist()))
            for i in range(len(row_ptr) - 1)]


@pytest.mark.parametrize("full", [True, False])
def test_native_parser_equals_the_independent_restatement_on_the_lyrl2004_sample(full):
    from oracle import ref_loader

    rp, col, val, lab, ids = ref_loader.rcv1(SAMPLE, full=full)
    d, d_ids = rcv1.load(SAMPLE, full=full, with_ids=True)
    assert d.n_rows == (24 if full else 8) == len(lab)
    np.testing.assert_array_equal(d_ids, ids)
    np.testing.assert_array_equal(d.label, lab)
    np.testing.assert_array_equal(d.row_ptr, rp)
    assert canon(d.row_ptr, d.col, d.val) == canon(rp, col, val)
    if full:
        # pinned by hand from the committed files: ids ascend from 2286 across the five files in order; a document is
        # +1 only when CCAT is the LAST of its qrels lines (codes are listed in code order: C15 < CCAT < E12 ...)
        assert ids[0] == 2286 and (np.diff(ids) > 0).all() and int(rp[-1]) == 857
        qrels = {}
        for line in open(os.path.join(SAMPLE, rcv1.QRELS)):
            code, did, _ = line.split()
            qrels.setdefault(int(did), []).append(code)
        assert lab.tolist() == [1 if qrels[int(i)][-1] == "CCAT" else -1 for i in ids]
        assert any("CCAT" in qrels[int(i)] and qrels[int(i)][-1] != "CCAT" for i in ids)   # the quirk is exercised
        assert 0 < (lab > 0).sum() < len(lab)
        # cosine-normalised rows, 16 significant digits in the text: fp32 after parsing
        norms = [float(np.sqrt((np.asarray(d.val[d.row_ptr[i]:d.row_ptr[i + 1]], np.float64) ** 2).sum())) for i in range(24)]
        assert max(abs(x - 1.0) for x in norms) < 1e-6


def test_native_parser_equals_the_restatement_on_the_quirk_lines(tmp_path):
    from oracle import ref_loader

    files = {rcv1.FILES[0]: ("10  3:0.5 7:0.25\n11 9:1.0 4:0.5 6:0.125\n12  5:1 5:2 2:3\n13  8:1e-25 1:0.5:junk\n14  \n"
                             "15  2:1.5 \n16  4:2  6:3\n")}   # trailing space; a doubled space inside the features
    qrels = "CCAT 10 1\nCCAT 11 1\nGCAT 11 1\nMCAT 12 1\nCCAT 12 1\nECAT 13 1\nCCAT 14 1\nCCAT 15 1\nCCAT 16 1\n"
    write(str(tmp_path), files, qrels)
    try:
        ref = ref_loader.rcv1(str(tmp_path), full=False)
    except (ValueError, IndexError, KeyError) as e:
        ref = e
    try:
        d, ids = rcv1.load(str(tmp_path), full=False, with_ids=True)
    except ValueError as e:
        d = e
    # `16  4:2  6:3`: the doubled space yields an empty token, "".split(':')(0).toInt throws in the reference
    assert isinstance(ref, Exception) == isinstance(d, Exception)
    if isinstance(ref, Exception):
        files[rcv1.FILES[0]] = files[rcv1.FILES[0]].replace("16  4:2  6:3\n", "16  4:2 6:3\n")
        write(str(tmp_path), files, qrels)
        ref = ref_loader.rcv1(str(tmp_path), full=False)
        d, ids = rcv1.load(str(tmp_path), full=False, with_ids=True)
    rp, col, val, lab, rids = ref
    np.testing.assert_array_equal(ids, rids)
    np.testing.assert_array_equal(d.label, lab)
    assert canon(d.row_ptr, d.col, d.val) == canon(rp, col, val)
