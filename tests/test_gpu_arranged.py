"""GPU (-m gpu): arranged row streams of the packed kernels (sa_plan.cpp: sa_arrange_rows; sa_systolic_pk.inc) against
the oracle, at sizes where they engage and on the cases their bookkeeping makes delicate: device-memory output (blocks
of several sizes, scores scattered inside a block through rowmap), host output (one tile per block, scores leaving in
row order through posmap), ranges that start inside a column (the first columns of the range cannot use blocks),
int16 output, few distinct lengths (all rounds pure) and all-distinct lengths (all rounds mixed), and agreement with
SA_HIP_NO_SORT-style row order by construction (the oracle knows nothing of either)."""
import numpy as np
import pytest

from tests.synth import make_dna_set, make_protein_set

pytestmark = pytest.mark.gpu

METHODS = [("nw", dict(gap_pen=4)), ("ga", dict(gap_open=10, gap_extend=1)), ("sw", dict(gap_open=10, gap_extend=1))]


def device_range(sa, ctx, lo, n, use16=False):
    import torch
    buf = torch.full((n + 8,), -77, dtype=torch.int16 if use16 else torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    (ctx.align_range16 if use16 else ctx.align_range)(lo, n, buf.data_ptr(), s)
    torch.cuda.synchronize()
    out = buf.cpu().numpy()
    assert (out[n:] == -77).all(), "wrote past the range"
    return out[:n].astype(np.int32)


@pytest.mark.parametrize("method,gaps", METHODS)
def test_device_output_blocks_of_several_sizes(method, gaps, sa, oracle):
    """4300 sequences: two blocks of 2048 rows, then 1024 / 512 / 256-row blocks in the tails of the columns; whole range,
    a range cut inside columns at both ends, and int16 output"""
    seqs = make_protein_set(4300, 30, 70, 31)
    store = sa.SequenceStore.from_sequences(seqs)
    scoring = sa.Scoring.from_names(method, "blosum62", **gaps)
    want = oracle.align(store, scoring, triangular=True, threads=16)
    with sa.Context(store, scoring, 0) as ctx:
        got = device_range(sa, ctx, 0, store.pairs)
        assert np.array_equal(got, want)
        lo, hi = 2_345_678, store.pairs - 1_234_567  # both inside a column
        assert np.array_equal(device_range(sa, ctx, lo, hi - lo), want[lo:hi])
        assert ctx.scores_fit16
        assert np.array_equal(device_range(sa, ctx, lo, hi - lo, use16=True), want[lo:hi])


@pytest.mark.parametrize("method,gaps", METHODS)
def test_host_output_in_row_order(method, gaps, sa, oracle):
    """the same store through the host-delivery loop: a block is one tile and its scores leave in row order"""
    seqs = make_dna_set(3300, 40, 90, 32, iupac=True)
    store = sa.SequenceStore.from_sequences(seqs)
    scoring = sa.Scoring.from_names(method, "nuc44", **gaps)
    want = oracle.align(store, scoring, triangular=True, threads=16)
    dest = sa.PinnedMatrix(store.pairs)
    try:
        with sa.Context(store, scoring, 0) as ctx:
            dest.array[:] = -77
            ctx.align_host(dest.array, triangular=True)
            assert np.array_equal(dest.array, want)
            lo, n = 1_000_003, 3_000_001
            dest.array[:] = -77
            ctx.align_host(dest.array, triangular=True, start=lo, count=n)
            assert np.array_equal(dest.array[lo:lo + n], want[lo:lo + n])
            assert (dest.array[:lo] == -77).all() and (dest.array[lo + n:] == -77).all()
    finally:
        dest.close()


@pytest.mark.parametrize("lengths", ["two", "distinct"])
def test_pure_and_mixed_rounds(lengths, sa, oracle):
    """only two lengths (every round pure, terminators of a wave's streams in step) / every length different inside a
    block (every round mixed); NW, device output"""
    base = make_protein_set(2600, 150, 150, 33)
    if lengths == "two":
        seqs = [s[:60] if k % 3 else s[:97] for k, s in enumerate(base)]
    else:
        seqs = [s[:20 + (k * 37) % 131] for k, s in enumerate(base)]
    store = sa.SequenceStore.from_sequences(seqs)
    scoring = sa.Scoring.from_names("nw", "blosum62", gap_pen=4)
    want = oracle.align(store, scoring, triangular=True, threads=16)
    with sa.Context(store, scoring, 0) as ctx:
        assert np.array_equal(device_range(sa, ctx, 0, store.pairs), want)
