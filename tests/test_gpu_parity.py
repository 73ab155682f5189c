"""GPU (-m gpu): parity of the HIP path, called through the C ABI, against
  * the reference-generated golden vectors (tests/golden/),
  * the CPU oracle on seeded random inputs,
  * size-independent properties at BASELINE.json sizes (sampled oracle comparison, range additivity,
    symmetry of the expanded matrix, row-permutation consistency).
Bar: bit-exact (s32)."""
import numpy as np
import os

import pytest

from tests.golden_util import golden_cases, load_case, tri_to_full
from tests.synth import make_config, make_dna_set, make_protein_set

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU test run without a visible HIP device"
    return torch


@pytest.mark.parametrize("name", golden_cases())
def test_golden_through_c_abi(name, sa):
    store, scoring, expected, full = load_case(name)
    got = sa.hip_align(store, scoring, triangular=True)
    assert np.array_equal(got, expected)
    got_full = sa.hip_align(store, scoring, triangular=False)
    assert np.array_equal(got_full, tri_to_full(expected, store.num))
    if full is not None:
        assert np.array_equal(got_full, full)


RANDOM_CASES = [
    ("nw", "blosum62", dict(gap_pen=4), "protein", 200, 1, 150),
    ("nw", "blosum62", dict(gap_pen=0), "protein", 100, 1, 90),
    ("nw", "pam250", dict(gap_pen=13), "protein", 100, 60, 200),
    ("ga", "blosum62", dict(gap_open=10, gap_extend=1), "protein", 200, 1, 150),
    ("ga", "blosum62", dict(gap_open=2, gap_extend=9), "protein", 100, 1, 100),
    ("ga", "blosum45", dict(gap_open=0, gap_extend=3), "protein", 100, 1, 100),
    ("sw", "blosum62", dict(gap_open=10, gap_extend=1), "protein", 200, 1, 150),
    ("sw", "blosum62", dict(gap_open=0, gap_extend=0), "protein", 100, 1, 100),
    ("sw", "nuc44", dict(gap_open=10, gap_extend=1), "dna", 150, 100, 200),
    ("nw", "dnafull", dict(gap_pen=4), "dna", 100, 1, 300),
    ("ga", "nuc44", dict(gap_open=12, gap_extend=3), "dna", 100, 120, 180),
    ("nw", "blosum62", dict(gap_pen=100000), "protein", 60, 1, 120),
    ("sw", "blosum62", dict(gap_open=50000, gap_extend=70000), "protein", 60, 1, 120),
]


@pytest.mark.parametrize("method,matrix,gaps,kind,n,lo,hi", RANDOM_CASES)
def test_random_sets_against_oracle(method, matrix, gaps, kind, n, lo, hi, sa, oracle):
    seqs = make_protein_set(n, lo, hi, 101) if kind == "protein" else make_dna_set(n, lo, hi, 102, iupac=True)
    store = sa.SequenceStore.from_sequences(seqs)
    scoring = sa.Scoring.from_names(method, matrix, **gaps)
    assert np.array_equal(sa.hip_align(store, scoring, triangular=True), oracle.align(store, scoring, triangular=True))


def test_edge_shapes(sa, oracle):
    """minimum job (2 sequences of length 1), ragged lengths, identical sequences, very long vs very short"""
    scoring_all = [sa.Scoring.from_names("nw", "blosum62", gap_pen=4),
                   sa.Scoring.from_names("ga", "blosum62", gap_open=10, gap_extend=1),
                   sa.Scoring.from_names("sw", "blosum62", gap_open=10, gap_extend=1)]
    sets = [
        [b"A", b"W"],
        [b"A", b"A"],
        [b"ARNDCQEGHILKMFPSTWYVBZX*"] * 5,
        [b"W" * 1, b"W" * 2000, b"A" * 3, b"ARND" * 300, b"W" * 65, b"W" * 64, b"W" * 63],
        make_protein_set(3, 2500, 3000, 5) + [b"M"],
    ]
    for seqs in sets:
        store = sa.SequenceStore.from_sequences(seqs)
        for sc in scoring_all:
            assert np.array_equal(sa.hip_align(store, sc, triangular=True), oracle.align(store, sc, triangular=True))
            assert np.array_equal(sa.hip_align(store, sc, triangular=False), oracle.align(store, sc, triangular=False))


def test_align_phase_timer(sa):
    """sa_hip_last_align_seconds(): the launch/copy phase of the last call -- positive and inside the call's wall time"""
    import time
    store = sa.SequenceStore.from_sequences(make_protein_set(600, 50, 150, 9))
    scoring = sa.Scoring.from_names("nw", "blosum62", gap_pen=4)
    for triangular in (True, False):
        t0 = time.perf_counter()
        sa.hip_align(store, scoring, triangular=triangular)
        wall = time.perf_counter() - t0
        phase = sa.last_align_seconds()
        assert 0.0 < phase <= wall


def test_multi_device_driver_path(sa, oracle, monkeypatch):
    """sa_hip_align's several-devices-in-one-process path (work-balanced slices, one host thread per slice,
    slices delivered straight into the host matrix), exercised on one GPU through SA_HIP_SPLIT."""
    store = sa.SequenceStore.from_sequences(make_protein_set(400, 10, 200, 23))
    for method, gaps in (("nw", dict(gap_pen=4)), ("sw", dict(gap_open=10, gap_extend=1))):
        scoring = sa.Scoring.from_names(method, "blosum62", **gaps)
        want_tri = oracle.align(store, scoring, triangular=True)
        want_full = oracle.align(store, scoring, triangular=False)
        for split in ("3", "8"):
            monkeypatch.setenv("SA_HIP_SPLIT", split)
            assert np.array_equal(sa.hip_align(store, scoring, triangular=True), want_tri)
            assert np.array_equal(sa.hip_align(store, scoring, triangular=False), want_full)
        monkeypatch.delenv("SA_HIP_SPLIT")


def test_device_filter_matches_reference_semantics(sa, oracle):
    """sa_hip_filter (relation on the device, greedy resolution on the host) == the reference's filter run with
    one thread: golden keep list, oracle on planted near-duplicates, ragged lengths, thresholds 0 / 1."""
    import json
    from tests.golden_util import GOLDEN_DIR
    from tests.synth import make_near_duplicates
    z = np.load(GOLDEN_DIR / "filter_f0.9.npz")
    meta = np.ascontiguousarray(z["meta"], np.int32)
    store = sa.SequenceStore(blob=np.ascontiguousarray(z["blob"]), meta=meta, num=meta.shape[0], max=int(meta[:, 1].max()))
    thr = json.loads(str(z["params"]))["threshold"]
    assert list(np.nonzero(sa.hip_filter(store, thr))[0]) == list(z["kept"])
    for n, lo, hi, dup, seed in ((700, 30, 60, 0.3, 9), (3000, 80, 400, 0.2, 10), (130, 1, 9, 0.5, 12)):
        seqs = make_near_duplicates(make_protein_set(n, lo, hi, seed), dup, 0.05, seed)
        st = sa.SequenceStore.from_sequences(seqs)
        for t in (0.9, 0.5, 1.0):
            assert np.array_equal(sa.hip_filter(st, t), oracle.filter(st, t)), (n, t)
        assert sa.hip_filter(st, 0.0).all()


def test_device_filter_late_matches_and_exact_thresholds(sa, oracle):
    """the relation kernel leaves a tile once no pair can reach the threshold any more (matches so far + positions left):
    pairs that disagree on a long prefix and agree on everything after it -- similar only through their LAST positions --
    and pairs that sit exactly on / one residue below the threshold must come out as the reference's float test says"""
    rng = np.random.default_rng(77)
    letters = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", np.uint8)
    def other(a):  # a residue different from a
        return letters[(int(np.nonzero(letters == a)[0][0]) + 1 + int(rng.integers(0, 18))) % 20]
    seqs = []
    for length in (40, 64, 65, 100, 127, 128, 129, 200, 333):
        base = letters[rng.integers(0, 20, length)]
        seqs.append(base.tobytes())
        for mism in (0, 1, length // 10 - 1, length // 10, length // 10 + 1, length // 4, length // 2):
            if mism < 0 or mism > length:
                continue
            v = base.copy()  # mismatches packed into the FIRST positions: the pair looks hopeless for as long as possible
            for q in range(mism):
                v[q] = other(base[q])
            seqs.append(v.tobytes())
            w = base.copy()  # ... and into the last ones
            for q in range(mism):
                w[length - 1 - q] = other(base[length - 1 - q])
            seqs.append(w.tobytes())
            seqs.append(v.tobytes()[:length - 7])  # a shorter partner: min(len) decides
    seqs += [letters[rng.integers(0, 20, int(n))].tobytes() for n in rng.integers(30, 300, 150)]  # unrelated company
    order = rng.permutation(len(seqs))
    st = sa.SequenceStore.from_sequences([seqs[k] for k in order])
    for t in (0.9, 0.75, 0.5, 0.95, 0.1, 1.0):
        got, want = sa.hip_filter(st, t), oracle.filter(st, t)
        assert np.array_equal(got, want), (t, np.nonzero(got != want)[0][:10])
        assert not want.all() or t == 1.0


def test_error_behaviour(sa):
    sc = sa.Scoring.from_names("nw", "blosum62", gap_pen=4)
    with pytest.raises(sa.AlignError, match="Not enough sequences"):
        sa.hip_align(sa.SequenceStore.from_sequences([b"ARND"]), sc)
    with pytest.raises(sa.AlignError, match="Invalid character"):
        sa.hip_align(sa.SequenceStore.from_sequences([b"ARND", b"AJND"]), sc)  # J is not in the amino alphabet
    with pytest.raises(sa.AlignError, match="overflows 32-bit"):
        sa.hip_align(sa.SequenceStore.from_sequences([b"ARND" * 100, b"ARNW" * 100]),
                     sa.Scoring.from_names("nw", "blosum62", gap_pen=2**31 - 1))
    assert sa.hip_align(sa.SequenceStore.from_sequences([b"ARND", b"ARNW"]), sc, write=False) is None  # -W
    assert sa.hip_memory(1 << 20) is True
    assert sa.hip_memory(1 << 50) is False


def test_device_resident_ranges(sa, oracle, torch_cuda):
    """sa_ctx_align_range == the reference's kernel(scores, start, batch): any split of the packed index
    gives the same scores; expand_full gives the symmetric zero-diagonal matrix."""
    torch = torch_cuda
    store = sa.SequenceStore.from_sequences(make_protein_set(300, 20, 130, 7))
    for method, gaps in (("nw", dict(gap_pen=4)), ("ga", dict(gap_open=10, gap_extend=1)), ("sw", dict(gap_open=10, gap_extend=1))):
        scoring = sa.Scoring.from_names(method, "blosum62", **gaps)
        want = oracle.align(store, scoring, triangular=True)
        with sa.Context(store, scoring, 0) as ctx:
            assert ctx.pairs == want.size
            out = torch.full((ctx.pairs,), -7, dtype=torch.int32, device="cuda")
            stream = torch.cuda.current_stream().cuda_stream
            cuts = [0, 1, 2, 65, 4097, 20000, ctx.pairs]
            for a, b in zip(cuts, cuts[1:]):
                ctx.align_range(a, b - a, out.data_ptr() + 4 * a, stream)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), want)
            full = torch.full((store.num, store.num), -9, dtype=torch.int32, device="cuda")
            ctx.expand_full(out.data_ptr(), full.data_ptr(), stream)
            torch.cuda.synchronize()
            f = full.cpu().numpy()
            assert np.array_equal(f, oracle.align(store, scoring, triangular=False))
            assert np.array_equal(f, f.T) and not np.diag(f).any()
            bounds = ctx.partition(4)
            parts = []
            for a, b in zip(bounds, bounds[1:]):
                t = torch.empty(b - a, dtype=torch.int32, device="cuda")
                ctx.align_range(a, b - a, t.data_ptr(), stream)
                parts.append(t)
            torch.cuda.synchronize()
            assert np.array_equal(torch.cat(parts).cpu().numpy(), want)


@pytest.mark.parametrize("cfg_name,n", [("cfg2", 10_000), ("cfg3", 10_000), ("cfg4", 6_000), ("cfg5", 6_000)])
def test_baseline_sizes_sampled(cfg_name, n, sa, oracle, torch_cuda):
    """BASELINE.json shapes (cfg 2/3 at full size; cfg 4/5 sequence shapes on a 6k subset that one GPU does in
    seconds): 200k sampled pairs + one full column + the first 300-sequence block against the oracle, and a
    checksum-of-checksums of a split run against the single-range run."""
    torch = torch_cuda
    seqs, cfg = make_config(cfg_name, n)
    store = sa.SequenceStore.from_sequences(seqs)
    scoring = sa.Scoring.from_names(cfg["method"], cfg["matrix"], **cfg["gaps"])
    with sa.Context(store, scoring, 0) as ctx:
        out = torch.empty(ctx.pairs, dtype=torch.int32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        ctx.align_range(0, ctx.pairs, out.data_ptr(), stream)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        # (1) random sample
        rng = np.random.default_rng(5)
        idx = np.sort(rng.integers(0, ctx.pairs, 200_000))
        assert np.array_equal(got[idx], oracle.align_pairs(store, scoring, idx))
        # (2) last column (j = n-1) and first 300x300 block
        j = store.num - 1
        col = np.arange(j * (j - 1) // 2, j * (j - 1) // 2 + j)
        assert np.array_equal(got[col], oracle.align_pairs(store, scoring, col))
        blk = 300 * 299 // 2
        assert np.array_equal(got[:blk], oracle.align_range(store, scoring, 0, blk))
        # (3) split into 8 work-balanced ranges == single range (what the multi-GPU path relies on)
        bounds = ctx.partition(8)
        out2 = torch.empty_like(out)
        for a, b in zip(bounds, bounds[1:]):
            ctx.align_range(a, b - a, out2.data_ptr() + 4 * a, stream)
        torch.cuda.synchronize()
        assert torch.equal(out, out2)
        sums = [int(out2[a:b].to(torch.int64).sum()) for a, b in zip(bounds, bounds[1:])]
        assert sum(sums) == int(got.astype(np.int64).sum())


@pytest.mark.parametrize("cfg_name", ["cfg4", "cfg5"])
def test_baseline_full_size_cfg4_cfg5(cfg_name, sa, oracle, torch_cuda):
    """BASELINE.json configs 4 and 5 at their FULL sizes on one GPU (50 000 DNA x ~150 bp SW/NUC.4.4 = 1.25e9 pairs;
    100 000 protein x ~120 aa NW with the -f 0.9 filter applied first = up to 5e9 pairs), result kept in HBM:
    300k randomly sampled pairs + the last column against the oracle, and every score inside the bounds the
    method allows (a property no sample needs)."""
    torch = torch_cuda
    seqs, cfg = make_config(cfg_name)
    if cfg_name == "cfg5":  # `-f 0.9`: planted near-duplicates are dropped before alignment -- sa_hip_filter at full size
        from tests.host_binding import Host
        amino = sa.Scoring.from_names("nw", "blosum62", gap_pen=4).lut
        keep = sa.hip_filter(sa.SequenceStore.from_sequences(seqs), 0.9)  # relation on the device, 100 000 sequences
        kept = [s for s, k in zip(seqs, keep) if k]
        assert 0.85 * len(seqs) < len(kept) < 0.95 * len(seqs)
        assert kept == Host().filter(seqs, amino, 0.9)                    # == the exact blocked CPU implementation
        assert oracle.filter(sa.SequenceStore.from_sequences(kept[:3000]), 0.9).all()  # survivors are mutually dissimilar
        seqs = kept
    store = sa.SequenceStore.from_sequences(seqs)
    scoring = sa.Scoring.from_names(cfg["method"], cfg["matrix"], **cfg["gaps"])
    with sa.Context(store, scoring, 0) as ctx:
        out = torch.empty(ctx.pairs, dtype=torch.int32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        step = 1 << 30  # the reference batches too (seqalign_cuda.c:136); any split must give the same vector
        for a in range(0, ctx.pairs, step):
            ctx.align_range(a, min(step, ctx.pairs - a), out.data_ptr() + 4 * a, stream)
        torch.cuda.synchronize()
        rng = np.random.default_rng(11)
        idx = np.sort(rng.integers(0, ctx.pairs, 300_000))
        j = store.num - 1
        idx = np.concatenate([idx, np.arange(j * (j - 1) // 2, j * (j - 1) // 2 + j)])
        got = out[torch.from_numpy(idx).cuda()].cpu().numpy()
        assert np.array_equal(got, oracle.align_pairs(store, scoring, idx))
        lens = store.meta[:, 1].astype(np.int64)
        smax = int(scoring.sub.max())
        lo, hi = int(out.min()), int(out.max())
        if cfg["method"] == "sw":
            assert lo >= 0 and hi <= smax * int(lens.max())
        else:
            g = -scoring.gap_pen
            assert hi <= smax * int(lens.max()) and lo >= -g * 2 * int(lens.max()) + int(scoring.sub.min()) * int(lens.max())


def test_int16_exchange_format(sa, oracle, torch_cuda):
    """sa_ctx_align_range16 + sa_hip_widen16 == sa_ctx_align_range when the score bound fits int16; refused otherwise"""
    torch = torch_cuda
    store = sa.SequenceStore.from_sequences(make_protein_set(500, 30, 260, 31))
    st = torch.cuda.current_stream().cuda_stream
    for method, gaps in (("nw", dict(gap_pen=4)), ("ga", dict(gap_open=10, gap_extend=1)), ("sw", dict(gap_open=10, gap_extend=1))):
        scoring = sa.Scoring.from_names(method, "blosum62", **gaps)
        want = oracle.align(store, scoring, triangular=True)
        with sa.Context(store, scoring, 0) as ctx:
            assert ctx.scores_fit16
            a, b = 12345, ctx.pairs - 777          # a range that starts and ends inside columns, odd offsets
            out16 = torch.full((b - a + 3,), -7, dtype=torch.int16, device="cuda")
            ctx.align_range16(a, b - a, out16[3:].data_ptr(), st)   # deliberately not 16-byte aligned
            out32 = torch.empty(b - a, dtype=torch.int32, device="cuda")
            ctx.widen16(out16[3:].data_ptr(), out32.data_ptr(), b - a, st)
            torch.cuda.synchronize()
            assert np.array_equal(out32.cpu().numpy(), want[a:b])
            assert int(out16[0]) == -7 and int(out16[2]) == -7
            al16 = torch.empty(b - a, dtype=torch.int16, device="cuda")      # aligned: the vector path of the widening
            ctx.align_range16(a, b - a, al16.data_ptr(), st)
            ctx.widen16(al16.data_ptr(), out32.data_ptr(), b - a, st)
            torch.cuda.synchronize()
            assert np.array_equal(out32.cpu().numpy(), want[a:b])
    os.environ["SA_HIP_FORCE_GENERIC"] = "1"      # the always-applicable kernels honour the exchange format too
    try:
        scoring = sa.Scoring.from_names("ga", "blosum62", gap_open=10, gap_extend=1)
        small = store.prefix(120)
        want = oracle.align(small, scoring, triangular=True)
        with sa.Context(small, scoring, 0) as ctx:
            o16 = torch.empty(ctx.pairs, dtype=torch.int16, device="cuda")
            ctx.align_range16(0, ctx.pairs, o16.data_ptr(), st)
            torch.cuda.synchronize()
            assert np.array_equal(o16.cpu().numpy().astype(np.int32), want)
    finally:
        del os.environ["SA_HIP_FORCE_GENERIC"]
    big = sa.SequenceStore.from_sequences(make_protein_set(20, 1500, 2500, 32))
    with sa.Context(big, sa.Scoring.from_names("nw", "blosum62", gap_pen=4), 0) as ctx:
        assert not ctx.scores_fit16          # 2500 * 11 + 2 * 2500 * 4 > 32767
        buf = torch.empty(ctx.pairs, dtype=torch.int16, device="cuda")
        with pytest.raises(sa.AlignError):
            ctx.align_range16(0, ctx.pairs, buf.data_ptr(), st)
