"""GPU (-m gpu): randomized parity sweep of the fast path's corner cases against the oracle: every column-length
class boundary (n = W-1, W, W+1), streams of length-1 sequences, all methods, random matrices and gap values up to
the s8-profile limit and beyond (generic fallback), packed sub-ranges that cut columns, and every stream length the
planner can choose.  Seeds are fixed: the case list is identical on every run."""
import numpy as np
import pytest

from tests.synth import AMINO20, splitmix64

pytestmark = pytest.mark.gpu

# every kernel class boundary: W = 8..128 step 8 (8-lane groups), 144..256 step 16, 288..512 step 32, 576..1024 step 64
W_CLASSES = list(range(8, 129, 8)) + list(range(144, 257, 16)) + list(range(288, 513, 32)) + list(range(576, 1025, 64))


def rand_seqs(rng, n, lens, alphabet):
    a = np.frombuffer(alphabet, np.uint8)
    return [a[rng.integers(0, len(a), int(l))].tobytes() for l in lens]


def check(sa, oracle, seqs, scoring, sub=None):
    store = sa.SequenceStore.from_sequences(seqs)
    want = oracle.align(store, scoring, triangular=True)
    got = sa.hip_align(store, scoring, triangular=True)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, f"{bad.size} mismatches, first at packed index {bad[:5]}: got {got[bad[:5]]} want {want[bad[:5]]}"


@pytest.mark.parametrize("method", ["nw", "ga", "sw"])
def test_class_boundaries(method, sa, oracle):
    """column sequences of length W-1, W, W+1 for every class, mixed with short and long row sequences"""
    rng = np.random.default_rng(100)
    gaps = dict(gap_pen=4) if method == "nw" else dict(gap_open=10, gap_extend=1)
    scoring = sa.Scoring.from_names(method, "blosum62", **gaps)
    for w in W_CLASSES:
        lens = [1, 2, 3, w - 1, w, w + 1, 17, w // 2, w, 5, w - 1, 1, w + 1, 33]
        if w > 256:  # fewer row sequences for the wide classes keeps the oracle side short
            lens = [1, w - 1, w, w + 1, 17, w // 2, 5, w + 1, 33]
        check(sa, oracle, rand_seqs(rng, len(lens), lens, AMINO20), scoring)


def test_random_parameter_sweep(sa, oracle):
    rng = np.random.default_rng(101)
    matrices = ["blosum62", "blosum45", "blosum100", "pam30", "pam250", "pam500", "blosum30"]
    for case in range(60):
        method = ["nw", "ga", "sw"][case % 3]
        matrix = matrices[int(rng.integers(0, len(matrices)))]
        if method == "nw":
            gaps = dict(gap_pen=int(rng.choice([0, 1, 2, 4, 7, 11, 20, 40, 55, 300])))
        else:
            o, e = int(rng.choice([0, 1, 3, 5, 10, 12, 25, 50, 200])), int(rng.choice([0, 1, 2, 3, 5, 10, 30]))
            gaps = dict(gap_open=o, gap_extend=e)
        regime = case % 4
        n = int(rng.integers(40, 160))
        if regime == 0:
            lens = rng.integers(1, 12, n)          # very short: many terminators in flight
        elif regime == 1:
            lens = rng.integers(50, 140, n)        # the benchmark regime
        elif regime == 2:
            lens = rng.integers(1, 400, n)         # ragged, several classes in one job
        else:
            lens = np.where(rng.random(n) < 0.1, rng.integers(600, 1300, n), rng.integers(20, 90, n))  # a few > 1024
        scoring = sa.Scoring.from_names(method, matrix, **gaps)
        try:
            check(sa, oracle, rand_seqs(rng, n, lens, AMINO20), scoring)
        except AssertionError as exc:
            raise AssertionError(f"case {case}: {method} {matrix} {gaps} regime {regime}: {exc}") from None


def test_subranges_and_stream_lengths(sa, oracle, torch_cuda=None):
    """any packed sub-range, however it cuts columns and whatever stream length the planner picks for it"""
    import torch
    rng = np.random.default_rng(102)
    seqs = rand_seqs(rng, 700, rng.integers(60, 130, 700), AMINO20)
    store = sa.SequenceStore.from_sequences(seqs)
    for method, gaps in (("nw", dict(gap_pen=4)), ("ga", dict(gap_open=10, gap_extend=1)), ("sw", dict(gap_open=10, gap_extend=1))):
        scoring = sa.Scoring.from_names(method, "blosum62", **gaps)
        want = oracle.align(store, scoring, triangular=True)
        with sa.Context(store, scoring, 0) as ctx:
            stream = torch.cuda.current_stream().cuda_stream
            for _ in range(12):
                a = int(rng.integers(0, ctx.pairs - 1))
                b = int(min(ctx.pairs, a + rng.integers(1, 60_000)))
                out = torch.full((b - a,), -12345, dtype=torch.int32, device="cuda")
                ctx.align_range(a, b - a, out.data_ptr(), stream)
                torch.cuda.synchronize()
                assert np.array_equal(out.cpu().numpy(), want[a:b]), (method, a, b)


@pytest.mark.parametrize("method,gaps", [("nw", dict(gap_pen=4)), ("ga", dict(gap_open=10, gap_extend=1)),
                                         ("sw", dict(gap_open=10, gap_extend=1)), ("ga", dict(gap_open=11, gap_extend=11))])
def test_strip_mined_long_columns(method, gaps, sa, oracle):
    """column sequences of 1 .. 5 strips (1025 .. 5000 residues) mixed with short ones: every strip count, strip
    boundaries (n = 1024k, 1024k+1), rows both short and long, enough rows for several stream tiles"""
    rng = np.random.default_rng(103)
    lens = [1024, 1025, 2047, 2048, 2049, 3000, 4096, 4097, 5000, 1500]
    lens = list(rng.integers(1, 200, 60)) + lens + list(rng.integers(900, 1300, 8)) + list(rng.integers(1, 60, 30)) + [2500, 1030]
    order = rng.permutation(len(lens))
    seqs = rand_seqs(rng, len(lens), [lens[k] for k in order], AMINO20)
    check(sa, oracle, seqs, sa.Scoring.from_names(method, "blosum62", **gaps))


@pytest.mark.parametrize("method,gaps", [("nw", dict(gap_pen=4)), ("ga", dict(gap_open=10, gap_extend=1)),
                                         ("sw", dict(gap_open=10, gap_extend=1)), ("sw", dict(gap_open=5, gap_extend=0))])
def test_tile_geometry(method, gaps, sa, oracle):
    """row counts around whole wave-tiles (8 lane groups x 32 sequences) and degenerate streams: all sequences of
    length 1, length-1 sequences between long ones (terminators two steps apart), identical sequences"""
    rng = np.random.default_rng(104)
    scoring = sa.Scoring.from_names(method, "blosum62", **gaps)
    for n in (2, 3, 9, 255, 256, 257, 300, 513):
        check(sa, oracle, rand_seqs(rng, n, rng.integers(4, 22, n), AMINO20), scoring)
    check(sa, oracle, rand_seqs(rng, 300, np.ones(300, dtype=int), AMINO20), scoring)
    lens = np.where(np.arange(200) % 2 == 0, 1, rng.integers(90, 130, 200))
    check(sa, oracle, rand_seqs(rng, 200, lens, AMINO20), scoring)
    one = rand_seqs(rng, 1, [77], AMINO20)[0]
    check(sa, oracle, [one] * 70, scoring)


def test_stamps_diagnostics_do_not_change_results(sa, oracle, tmp_path):
    """SA_HIP_STAMPS=1 (per-tile timeline of a bundle launch) in a child process: same scores, and the
    timeline line is printed -- the diagnostic is part of the evidence trail and must not rot"""
    import os
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np, sequencealigner_amd as sa\n"
        "from tests.synth import make_protein_set\n"
        "from tests.oracle_binding import Oracle\n"
        "store = sa.SequenceStore.from_sequences(make_protein_set(700, 40, 130, 5))\n"
        "sc = sa.Scoring.from_names('ga', 'blosum62', gap_open=10, gap_extend=1)\n"
        "got = sa.hip_align(store, sc, triangular=True)\n"
        "assert np.array_equal(got, Oracle().align(store, sc, triangular=True, threads=8))\n"
        "print('STAMPS_RUN_OK')\n" % str(__import__("pathlib").Path(__file__).resolve().parents[1]))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, SA_HIP_STAMPS="1"))
    assert res.returncode == 0 and "STAMPS_RUN_OK" in res.stdout, res.stdout[-1000:] + res.stderr[-3000:]
    assert "[stamps] sa_k_systolic_pk_bundle<ga" in res.stderr and "active per 5% of the launch" in res.stderr
