"""CPU, only where /root/reference was present at build time: the oracle restatement and the option
tables against the reference's own code (oracle/_ref/libseqalign_ref.so), live."""
import numpy as np
import pytest

from tests.oracle_binding import RefLib, ref_available
from tests.synth import make_dna_set, make_protein_set

pytestmark = pytest.mark.skipif(not ref_available(), reason="oracle/_ref not built (no reference checkout)")

CASES = [
    ("nw", "blosum62", dict(gap_pen=4)), ("nw", "pam120", dict(gap_pen=7)), ("ga", "blosum62", dict(gap_open=10, gap_extend=1)),
    ("ga", "blosum80", dict(gap_open=5, gap_extend=9)), ("sw", "blosum62", dict(gap_open=10, gap_extend=1)),
    ("sw", "blosum50", dict(gap_open=1, gap_extend=1)), ("ga", "blosum62", dict(gap_open=6, gap_extend=6)),
]


@pytest.mark.parametrize("method,matrix,gaps", CASES)
def test_oracle_equals_reference_random(method, matrix, gaps, oracle, sa):
    store = sa.SequenceStore.from_sequences(make_protein_set(120, 1, 160, 31))
    scoring = sa.Scoring.from_names(method, matrix, **gaps)
    ref = RefLib(method, matrix, **gaps)
    try:
        p = ref.params()
        assert np.array_equal(p["lut"], scoring.lut) and np.array_equal(p["sub"], scoring.sub)
        assert p["method"] == scoring.method_name
        assert np.array_equal(ref.align(store, triangular=True), oracle.align(store, scoring, triangular=True))
        assert np.array_equal(ref.align(store, triangular=False), oracle.align(store, scoring, triangular=False))
    finally:
        ref.close()


def test_every_matrix_matches_reference(sa):
    for k, name in enumerate(sa.matrix_names()):
        kw = dict(gap_pen=1)
        ref = RefLib("nw", name, **kw)
        try:
            p = ref.params()
            s = sa.Scoring.from_names("nw", name, **kw)
            assert np.array_equal(p["lut"], s.lut), name
            assert np.array_equal(p["sub"], s.sub), name
        finally:
            ref.close()


def test_dna_and_filter(oracle, sa):
    store = sa.SequenceStore.from_sequences(make_dna_set(80, 30, 200, 33, iupac=True))
    scoring = sa.Scoring.from_names("sw", "dnafull", gap_open=16, gap_extend=4)
    ref = RefLib("sw", "dnafull", gap_open=16, gap_extend=4)
    try:
        assert np.array_equal(ref.align(store, triangular=True), oracle.align(store, scoring, triangular=True))
    finally:
        ref.close()
