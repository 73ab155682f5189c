"""GPU (-m gpu): the multi-device body of sa_hip_align that follows BASELINE.json's north_star literally -- dense shares
per device, RCCL all-gather (ncclCommInitAll + grouped ncclAllGather, bound at run time), placement on every device,
delivery (csrc/sa_gather.hip).  A one-GPU box runs it with a ONE-device communicator (SA_HIP_GATHER=1):
the same calls, the same collective, a clique of one.  Results against the oracle and the reference's goldens."""
import os

import numpy as np
import pytest

from tests.golden_util import golden_cases, load_case, tri_to_full
from tests.synth import make_dna_set, make_protein_set

pytestmark = pytest.mark.gpu


@pytest.fixture()
def gather_on():
    os.environ["SA_HIP_GATHER"] = "1"
    yield
    del os.environ["SA_HIP_GATHER"]


@pytest.mark.parametrize("method,gaps", [("nw", dict(gap_pen=4)), ("ga", dict(gap_open=10, gap_extend=1)), ("sw", dict(gap_open=10, gap_extend=1))])
def test_gathered_align_matches_oracle(method, gaps, sa, oracle, gather_on):
    store = sa.SequenceStore.from_sequences(make_protein_set(1500, 20, 330, 41))  # 8- and 16-lane packed classes
    scoring = sa.Scoring.from_names(method, "blosum62", **gaps)
    want = oracle.align(store, scoring, triangular=True)
    got = sa.hip_align(store, scoring, triangular=True)
    assert sa.last_align_path() == "gather"
    assert np.array_equal(got, want)
    assert sa.last_align_seconds() > 0 and sa.last_align_breakdown()["phase_ms"] > 0
    full = sa.hip_align(store, scoring, triangular=False)          # N x N: the shells of this device's column range
    assert sa.last_align_path() == "gather"
    assert np.array_equal(full, tri_to_full(want, store.num)) and not np.diag(full).any()
    assert sa.hip_align(store, scoring, write=False) is None        # -W: compute, gather, place, copy nothing


def test_gathered_align_every_kernel_family_and_s32_exchange(sa, oracle, gather_on):
    """long sequences: scores do not fit int16 (s32 shares), s32 systolic and strip-mined columns; a scoring only the
    pair-per-wave kernels reproduce"""
    seqs = make_protein_set(60, 1, 150, 42) + make_protein_set(12, 900, 1400, 43) + make_dna_set(8, 1100, 2300, 44)
    store = sa.SequenceStore.from_sequences([s.replace(b"U", b"A") for s in seqs])
    for method, matrix, gaps in (("nw", "blosum62", dict(gap_pen=4)), ("ga", "blosum62", dict(gap_open=2, gap_extend=7)), ("sw", "pam250", dict(gap_open=11, gap_extend=2))):
        scoring = sa.Scoring.from_names(method, matrix, **gaps)
        got = sa.hip_align(store, scoring, triangular=True)
        assert sa.last_align_path() == "gather"
        assert np.array_equal(got, oracle.align(store, scoring, triangular=True))


@pytest.mark.parametrize("name", golden_cases())
def test_gathered_align_matches_reference_goldens(name, sa, gather_on):
    store, scoring, expected, full = load_case(name)
    got = sa.hip_align(store, scoring, triangular=True)
    assert sa.last_align_path() == "gather"
    assert np.array_equal(got, expected)
    if full is not None:
        assert np.array_equal(sa.hip_align(store, scoring, triangular=False), full)


def test_default_path_on_one_device_is_the_direct_one(sa, oracle):
    store = sa.SequenceStore.from_sequences(make_protein_set(300, 30, 120, 45))
    scoring = sa.Scoring.from_names("nw", "blosum62", gap_pen=4)
    got = sa.hip_align(store, scoring, triangular=True)
    assert sa.last_align_path() == ("gather" if sa.device_count() > 1 else "slices")
    assert np.array_equal(got, oracle.align(store, scoring, triangular=True))
