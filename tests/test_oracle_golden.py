"""CPU: the oracle restatement (oracle/sa_oracle.c) against the reference-generated golden vectors
and the known-answer table of SURVEY.md §8(c).  This is what pins the oracle."""
import json

import numpy as np
import pytest

from tests.golden_util import GOLDEN_DIR, golden_cases, load_case, tri_to_full


@pytest.mark.parametrize("name", golden_cases())
def test_oracle_matches_reference_golden(name, oracle, sa):
    store, scoring, expected, full = load_case(name)
    got = oracle.align(store, scoring, triangular=True)
    assert np.array_equal(got, expected)
    got_full = oracle.align(store, scoring, triangular=False)
    assert np.array_equal(got_full, tri_to_full(expected, store.num))
    assert np.all(np.diag(got_full) == 0)  # diagonal never written (io/output.c:76-81)
    if full is not None:
        assert np.array_equal(got_full, full)


# upper triangle, rows 0..6 vs later columns, blosum62
EDGE = [b"A", b"A", b"W", b"ARNDW", b"WWWW*", b"BZX", b"HEAGAWGHEE", b"PAWHEAE"]
KNOWN = {
    ("nw", (("gap_pen", 4),)): [[4, -3, -12, -19, -8, -32, -20], [-3, -12, -19, -8, -32, -20], [-5, -5, -10, -25, -13],
                                [-7, -6, -13, -7], [-17, -19, -9], [-24, -12], [12]],
    ("nw", (("gap_pen", 0),)): [[4, 0, 4, 0, 0, 4, 4], [0, 4, 0, 0, 4, 4], [11, 11, 0, 11, 11], [11, 4, 15, 15],
                                [0, 11, 11], [5, 5], [33]],
    ("ga", (("gap_open", 10), ("gap_extend", 1))): [[4, -3, -9, -16, -11, -19, -16], [-3, -9, -16, -11, -19, -16],
                                                    [-2, -2, -13, -16, -13], [-15, -9, -16, -15], [-20, -14, -12],
                                                    [-12, -14], [3]],
    ("sw", (("gap_open", 10), ("gap_extend", 1))): [[4, 0, 4, 0, 0, 4, 4], [0, 4, 0, 0, 4, 4], [11, 11, 0, 11, 11],
                                                    [11, 4, 11, 11], [0, 11, 11], [5, 4], [18]],
}


@pytest.mark.parametrize("key", list(KNOWN))
def test_oracle_known_answers(key, oracle, sa):
    method, gaps = key
    scoring = sa.Scoring.from_names(method, "blosum62", **dict(gaps))
    store = sa.SequenceStore.from_sequences(EDGE)
    full = oracle.align(store, scoring)
    for r, row in enumerate(KNOWN[key]):
        assert list(full[r, r + 1:]) == row
    # per-pair entry point, both argument orders (matrices are symmetric)
    for r in range(len(EDGE)):
        for c in range(r + 1, len(EDGE)):
            assert oracle.pair(scoring, EDGE[c], EDGE[r]) == full[r, c]
            assert oracle.pair(scoring, EDGE[r], EDGE[c]) == full[r, c]


def test_sw_zero_gaps_equals_nw_zero_gap(oracle, sa):
    store = sa.SequenceStore.from_sequences(EDGE)
    a = oracle.align(store, sa.Scoring.from_names("sw", "blosum62", gap_open=0, gap_extend=0))
    b = oracle.align(store, sa.Scoring.from_names("nw", "blosum62", gap_pen=0))
    assert np.array_equal(a, b)


def test_oracle_range_and_pairs_agree_with_full(oracle, sa):
    from tests.synth import make_protein_set
    store = sa.SequenceStore.from_sequences(make_protein_set(40, 5, 70, 11))
    scoring = sa.Scoring.from_names("ga", "blosum62", gap_open=10, gap_extend=1)
    tri = oracle.align(store, scoring, triangular=True)
    assert np.array_equal(oracle.align_range(store, scoring, 100, 333), tri[100:433])
    idx = np.array([0, 779, 5, 400, 400, 17], np.int64)
    assert np.array_equal(oracle.align_pairs(store, scoring, idx), tri[idx])
    for p in (0, 1, 2, 3, 779, 12345678901):
        i, j = oracle.unpack(p)
        assert 0 <= i < j and j * (j - 1) // 2 + i == p


def test_oracle_filter_matches_reference_golden(oracle, sa):
    z = np.load(GOLDEN_DIR / "filter_f0.9.npz")
    meta = np.ascontiguousarray(z["meta"], np.int32)
    store = sa.SequenceStore(blob=np.ascontiguousarray(z["blob"]), meta=meta, num=meta.shape[0], max=int(meta[:, 1].max()))
    thr = json.loads(str(z["params"]))["threshold"]
    keep = oracle.filter(store, thr)
    assert list(np.nonzero(keep)[0]) == list(z["kept"])
    assert oracle.filter(store, 0.0).all()  # threshold <= 0 keeps everything (bio/filter.c:16-17)
