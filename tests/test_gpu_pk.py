"""GPU (-m gpu): the packed-u16 kernels (sa_systolic_pk.inc: two column sequences per register, sliding frame) against the
oracle on the cases their design makes delicate: column-class boundaries and padding, odd column counts (a column
paired with itself), streams of very short rows (several frame shifts in flight), packed ranges that cut a column pair,
scorings that only partly fit u16 (mixed packed / s32 launches), and agreement with the s32 kernel family."""
import numpy as np
import pytest

from tests.synth import AMINO20, make_dna_set, make_protein_set, splitmix64

pytestmark = pytest.mark.gpu

METHODS = [("nw", dict(gap_pen=4)), ("ga", dict(gap_open=10, gap_extend=1)), ("sw", dict(gap_open=10, gap_extend=1))]


def seq_of(length, seed):
    r = splitmix64(np.arange(length), seed) % np.uint64(20)
    return np.frombuffer(AMINO20, np.uint8)[r.astype(np.int64)].tobytes()


@pytest.mark.parametrize("method,gaps", METHODS)
def test_class_boundaries_and_padding(method, gaps, sa, oracle):
    """column lengths 8K-1, 8K, 8K+1 for every class K (padding 1, 0, 7), three columns per class (an odd count: the last
    one is paired with itself), rows of every length in between"""
    lens = []
    for k in range(1, 25):
        lens += [8 * k - 1, 8 * k, 8 * k + 1] if k < 24 else [8 * k - 1, 8 * k]
    lens += [1, 2, 3, 5, 193, 200]  # below the first class; the first 16-lane class
    for k in range(13, 41):           # 16-lane groups: W = 16 K = 208 .. 640 here, 641 .. 1024 below (classes whose range does not fit fall to
        lens += [16 * k - 1, 16 * k, 16 * k + 1]  # the s32 kernels: with a 1-residue row in the store most of these)
    lens += [700, 1025]
    seqs = [seq_of(n, 1000 + i) for i, n in enumerate(lens)]
    store = sa.SequenceStore.from_sequences(seqs)
    scoring = sa.Scoring.from_names(method, "blosum62", **gaps)
    assert np.array_equal(sa.hip_align(store, scoring, triangular=True), oracle.align(store, scoring, triangular=True))


@pytest.mark.parametrize("method,gaps", METHODS)
def test_wide_classes_up_to_1024_columns(method, gaps, sa, oracle):
    """16-lane classes K = 41..64 (columns 641..1024: the row token is the profile offset in 256-byte units there), at
    their boundaries 16K-1, 16K, 16K+1, in a store whose shortest sequence keeps one frame shift in flight -- so that
    these columns really run on the packed kernels -- plus a column beyond 1024 (strip-mined s32 class) beside them"""
    lens = [16, 17, 40, 100, 333]
    for k in range(41, 65):
        lens += [16 * k - 1, 16 * k] + ([16 * k + 1] if k < 64 else [])
    lens += [1100]
    seqs = [seq_of(n, 4000 + i) for i, n in enumerate(lens)]
    store = sa.SequenceStore.from_sequences(seqs)
    scoring = sa.Scoring.from_names(method, "blosum62", **gaps)
    assert np.array_equal(sa.hip_align(store, scoring, triangular=True), oracle.align(store, scoring, triangular=True, threads=16))


@pytest.mark.parametrize("method,gaps", METHODS)
def test_streams_of_very_short_rows(method, gaps, sa, oracle):
    """hundreds of rows of length 1..3 in front of long columns: a terminator every other stream position, i.e. the
    maximum number of frame shifts between a value's birth and its capture"""
    seqs = [seq_of(1 + (i % 3), 50 + i) for i in range(700)] + [seq_of(n, 9000 + n) for n in (64, 100, 101, 127, 128, 150, 191, 192,
                                                                                              200, 256, 300, 352, 384)]
    store = sa.SequenceStore.from_sequences(seqs)
    scoring = sa.Scoring.from_names(method, "blosum62", **gaps)
    assert np.array_equal(sa.hip_align(store, scoring, triangular=True), oracle.align(store, scoring, triangular=True, threads=8))


def test_ranges_that_cut_column_pairs(sa, oracle):
    """sa_ctx_align_range on ranges that start / end inside a column, so that the two columns of a packed tile have
    different row ranges (or only one of them belongs to the range)"""
    import torch
    seqs = make_protein_set(200, 90, 104, 17) + make_protein_set(60, 250, 270, 18)  # few classes, many pairs per class
    store = sa.SequenceStore.from_sequences(seqs)
    rng = np.random.default_rng(5)
    for method, gaps in METHODS:
        scoring = sa.Scoring.from_names(method, "blosum62", **gaps)
        want = oracle.align(store, scoring, triangular=True)
        with sa.Context(store, scoring, 0) as ctx:
            for _ in range(12):
                lo = int(rng.integers(0, store.pairs - 1))
                cnt = int(rng.integers(1, min(store.pairs - lo, 6000) + 1))
                out = torch.full((cnt,), -12345, dtype=torch.int32, device="cuda")
                ctx.align_range(lo, cnt, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                assert np.array_equal(out.cpu().numpy(), want[lo:lo + cnt]), (method, lo, cnt)


@pytest.mark.parametrize("method,matrix,gaps", [
    ("nw", "blosum62", dict(gap_pen=40)),                    # DELTA so large that only the short classes fit u16
    ("nw", "blosum62", dict(gap_pen=1)),                     # S - 2g < 0 for some pairs: no packed class at all
    ("ga", "pam250", dict(gap_open=30, gap_extend=2)),
    ("ga", "blosum62", dict(gap_open=3, gap_extend=3 + 0)),  # becomes NW (equal gaps)
    ("sw", "blosum62", dict(gap_open=12, gap_extend=4)),     # larger per-step drift of the row domain
    ("sw", "blosum62", dict(gap_open=2, gap_extend=5)),      # |open| < |extend|: not the packed formulation
])
def test_scorings_that_partly_fit_u16(method, matrix, gaps, sa, oracle):
    seqs = make_protein_set(150, 1, 390, 23)
    store = sa.SequenceStore.from_sequences(seqs)
    scoring = sa.Scoring.from_names(method, matrix, **gaps)
    assert np.array_equal(sa.hip_align(store, scoring, triangular=True), oracle.align(store, scoring, triangular=True))


def test_packed_and_s32_kernel_families_agree(sa, monkeypatch):
    """SA_HIP_NO_PK routes the same job to the s32 kernels: two independent implementations, one result"""
    sets = [make_protein_set(500, 60, 380, 31), make_dna_set(400, 100, 390, 32, iupac=True)]
    for seqs, matrix in zip(sets, ("blosum62", "nuc44")):
        store = sa.SequenceStore.from_sequences(seqs)
        for method, gaps in METHODS:
            scoring = sa.Scoring.from_names(method, matrix, **gaps)
            monkeypatch.delenv("SA_HIP_NO_PK", raising=False)
            a = sa.hip_align(store, scoring, triangular=True)
            monkeypatch.setenv("SA_HIP_NO_PK", "1")
            b = sa.hip_align(store, scoring, triangular=True)
            monkeypatch.delenv("SA_HIP_NO_PK")
            assert np.array_equal(a, b), (matrix, method)


@pytest.mark.parametrize("method,gaps", METHODS)
@pytest.mark.parametrize("shortest", [200, 16, 7, 1])
def test_frame_budget_follows_the_shortest_sequence(method, gaps, shortest, sa, oracle):
    """The number of frame shifts in flight -- and with it the value range, the choice between the f16 three-way-max
    and the u16 two-way-max form of the 16-lane kernels, and the largest packed class -- follows the store's shortest
    sequence (sa_plan.h: sa_pk_live): columns of 193..650 residues against rows whose shortest has 200 / 16 / 7 / 1
    residues (1, 1, 2 and 8 shifts in flight with 16-lane groups), the short rows repeated so that terminators really
    come min_len + 1 positions apart."""
    seqs = [seq_of(n, 7000 + n) for n in range(200, 385, 3)] + [seq_of(n, 7000 + n) for n in range(390, 660, 13)]
    if shortest < 200:
        seqs = [seq_of(shortest + (i % 2), 300 + i) for i in range(120)] + seqs
    store = sa.SequenceStore.from_sequences(seqs)
    scoring = sa.Scoring.from_names(method, "blosum62", **gaps)
    assert np.array_equal(sa.hip_align(store, scoring, triangular=True), oracle.align(store, scoring, triangular=True, threads=8))


def test_several_launches_one_after_the_other_or_side_by_side(sa, oracle, monkeypatch):
    """a store whose columns need three packed bundles, s32 classes and the strip-mined launch: the launches of one range run
    one after the other on the caller's stream (default) or side by side on side streams (SA_HIP_CONCURRENT_CLASSES,
    round 3's schedule) -- the same scores either way, whole and in ranges that cut columns"""
    import torch
    lens = list(range(3, 200, 7)) + [220, 330, 520, 700, 900, 1024, 1100, 1500] + [40] * 300 + [150] * 200
    seqs = [seq_of(n, 5000 + i) for i, n in enumerate(lens)]
    store = sa.SequenceStore.from_sequences(seqs)
    scoring = sa.Scoring.from_names("ga", "blosum62", gap_open=10, gap_extend=1)
    want = oracle.align(store, scoring, triangular=True)
    for concurrent in (False, True):
        if concurrent:
            monkeypatch.setenv("SA_HIP_CONCURRENT_CLASSES", "1")
        with sa.Context(store, scoring, 0) as ctx:
            out = torch.empty(ctx.pairs, dtype=torch.int32, device="cuda")
            st = torch.cuda.current_stream().cuda_stream
            cuts = [0, 777, ctx.pairs // 3, ctx.pairs // 3 + 1, ctx.pairs - 5, ctx.pairs]
            for a, b in zip(cuts, cuts[1:]):
                ctx.align_range(a, b - a, out.data_ptr() + 4 * a, st)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), want), "side by side" if concurrent else "one after the other"
