"""GPU (-m gpu): the device-side DEFLATE encoder (sa_zjob_*, csrc/sa_deflate.hip; the -z option, reference
src/io/format/hdf5.c:91-95 + :148-194) by its round trip: every tile it returns is a complete zlib stream that stock
zlib inflates to exactly the bytes H5Dwrite would have handed libhdf5's deflate filter for that chunk -- the full
symmetric matrix in row-major int32 LE, zero diagonal, zeros beyond N.  Bit-exact, no tolerance."""
import zlib

import numpy as np
import pytest

from tests.golden_util import tri_to_full

pytestmark = pytest.mark.gpu


def expected_tiles(full: np.ndarray, chunk: int):
    n = full.shape[0]
    nc = (n + chunk - 1) // chunk
    pad = np.zeros((nc * chunk, nc * chunk), np.int32)
    pad[:n, :n] = full
    return nc, [[pad[r * chunk:(r + 1) * chunk, c * chunk:(c + 1) * chunk].astype("<i4").tobytes() for c in range(nc)] for r in range(nc)]


def check_job(sa, full: np.ndarray, chunk: int, d_packed=0, d_full=0, min_ratio=None):
    n = full.shape[0]
    nc, want = expected_tiles(full, chunk)
    raw = out = 0
    with sa.DeflateJob(n, chunk, d_packed_ptr=d_packed, d_full_ptr=d_full) as job:
        assert job.tiles_per_row == nc
        for r in range(nc):
            streams = job.tile_row(r)
            for c, z in enumerate(streams):
                assert z[:2] == b"\x78\x9c"
                got = zlib.decompress(z)  # checks the Adler-32 as well
                assert got == want[r][c], f"tile ({r},{c}) of {nc}x{nc}, chunk {chunk}"
                raw += len(got)
                out += len(z)
        st = job.stats()
        assert st["raw_bytes"] == raw and st["out_bytes"] == out
    if min_ratio is not None:
        assert raw / out >= min_ratio, (raw, out)
    return raw / out


def packed_of(full: np.ndarray) -> np.ndarray:
    n = full.shape[0]
    out = np.zeros(n * (n - 1) // 2, np.int32)
    for j in range(1, n):
        out[j * (j - 1) // 2: j * (j - 1) // 2 + j] = full[:j, j]
    return out


@pytest.mark.parametrize("n,chunk", [(300, 256), (700, 256), (1500, 512), (2100, 2048)])
def test_packed_matrix_round_trip(n, chunk, sa):
    import torch
    rng = np.random.default_rng(n)
    tri = rng.integers(-150, 110, size=n * (n - 1) // 2, dtype=np.int32)
    full = tri_to_full(tri, n)
    d = torch.from_numpy(tri).cuda()
    ratio = check_job(sa, full, chunk, d_packed=d.data_ptr())
    assert ratio > 2.0  # a sign-and-low-byte matrix: literal-only Huffman would stay below 2.7, stored at 1.0


def test_real_scores_ratio_and_round_trip(sa):
    """NW / BLOSUM62 scores from the device, through the packed path with the product's chunk rule (4096 is the
    clamp: 2 x 2 tiles of 64 MB, 1024 segments each)."""
    import torch
    from tests.synth import make_protein_set
    store = sa.SequenceStore.from_sequences(make_protein_set(4500, 96, 144, 5))
    scoring = sa.Scoring.from_names("nw", "blosum62", gap_pen=4)
    d = torch.empty(store.pairs, dtype=torch.int32, device="cuda")
    with sa.Context(store, scoring, 0) as ctx:
        ctx.align_range(0, store.pairs, d.data_ptr())
        torch.cuda.synchronize()
    tri = d.cpu().numpy()
    check_job(sa, tri_to_full(tri, store.num), 4096, d_packed=d.data_ptr(), min_ratio=2.7)


@pytest.mark.parametrize("kind", ["zeros", "constant", "full_range", "positive_small", "alternating", "big", "two_values"])
def test_full_matrix_adversarial_contents(kind, sa):
    """the d_full entry with contents that stress the code construction: one symbol only, every byte value (incompressible:
    the stream may grow, within the job's bound), code lengths at the 15-bit limit."""
    import torch
    n, chunk = 600, 256
    rng = np.random.default_rng(7)
    if kind == "zeros":
        full = np.zeros((n, n), np.int32)
    elif kind == "constant":
        full = np.full((n, n), -77, np.int32)
    elif kind == "full_range":
        full = rng.integers(-2**31, 2**31 - 1, size=(n, n), dtype=np.int64).astype(np.int32)
    elif kind == "positive_small":
        full = rng.integers(0, 40, size=(n, n), dtype=np.int32)
    elif kind == "alternating":
        full = (np.indices((n, n)).sum(0) % 2 * 2 - 1).astype(np.int32) * rng.integers(1, 300, size=(n, n), dtype=np.int32)
    elif kind == "big":
        full = rng.integers(-70000, 70000, size=(n, n), dtype=np.int32)
    else:  # a geometric histogram: code lengths run into the limit
        e = np.minimum(rng.geometric(0.5, size=(n, n)), 40)
        full = (np.int64(1) << (e % 31)).astype(np.int64).astype(np.int32)
    d = torch.from_numpy(np.ascontiguousarray(full)).cuda()
    check_job(sa, full, chunk, d_full=d.data_ptr())


def test_bad_arguments(sa):
    import torch
    d = torch.zeros(10, dtype=torch.int32, device="cuda")
    with pytest.raises(sa.AlignError):
        sa.DeflateJob(5, 100, d_packed_ptr=d.data_ptr())  # not a power of two
    with pytest.raises(sa.AlignError):
        sa.DeflateJob(5, 256)  # no matrix
    with sa.DeflateJob(5, 256, d_packed_ptr=d.data_ptr()) as job:
        with pytest.raises(sa.AlignError):
            job.tile_row(1)


def test_level_zero_returns_the_tiles_as_they_are(sa):
    """level 0 (a chunked dataset without filters): the same walk, tiles = chunk x chunk int32 LE, unencoded"""
    import torch
    n, chunk = 900, 256
    rng = np.random.default_rng(3)
    tri = rng.integers(-300, 300, size=n * (n - 1) // 2, dtype=np.int32)
    full = tri_to_full(tri, n)
    d = torch.from_numpy(tri).cuda()
    nc, want = expected_tiles(full, chunk)
    with sa.DeflateJob(n, chunk, d_packed_ptr=d.data_ptr(), level=0) as job:
        for r in range(nc):
            assert job.tile_row(r) == want[r]


@pytest.mark.parametrize("level", [6, 0])
@pytest.mark.parametrize("n,chunk,method", [(2100, 512, "nw"), (1300, 256, "sw"), (300, 256, "ga")])
def test_shells_while_the_alignment_runs(n, chunk, method, level, sa, oracle):
    """sa_hip_tiles_begin / sa_zjob_next: the alignment runs column block by column block and the tiles come out shell by
    shell (those whose larger tile index is b need exactly block b) -- every tile exactly once, each the bytes of the oracle's
    matrix (zero diagonal, zero padding), deflated or raw."""
    from tests.synth import make_protein_set
    store = sa.SequenceStore.from_sequences(make_protein_set(n, 20, 90, n))
    gaps = dict(gap_pen=4) if method == "nw" else dict(gap_open=10, gap_extend=1)
    scoring = sa.Scoring.from_names(method, "blosum62", **gaps)
    full = tri_to_full(oracle.align(store, scoring, triangular=True), n)
    nc, want = expected_tiles(full, chunk)
    seen = {}
    with sa.DeflateJob.begin(store, scoring, chunk, level=level) as job:
        assert job.tiles_per_row == nc
        order = []
        while True:
            batch = job.next()
            if not batch:
                break
            assert len(batch) <= nc
            for r, c, z in batch:
                assert (r, c) not in seen
                seen[(r, c)] = zlib.decompress(z) if level else z
                order.append(max(r, c))
        assert job.next() == []
        assert order == sorted(order)  # shell after shell
        assert job.align_seconds > 0
    assert len(seen) == nc * nc
    for (r, c), got in seen.items():
        assert got == want[r][c], (r, c)


@pytest.mark.parametrize("parts,level", [(2, 6), (3, 6), (5, 0), (9, 6)])
def test_shells_dealt_over_several_jobs(parts, level, sa, oracle, monkeypatch):
    """sa_hip_tiles_begin with several devices deals the column blocks over them (block b and its shell -> device b mod n; no
    exchange).  A one-GPU box folds n such jobs onto device 0 (SA_HIP_TILES_SPLIT): the same code, every job with its own
    context, matrix, streams and buffers; the tiles still arrive shell after shell, every one exactly once."""
    from tests.synth import make_protein_set
    n, chunk = 1700, 256  # 7 column blocks: fewer than nine jobs, more than two
    store = sa.SequenceStore.from_sequences(make_protein_set(n, 20, 70, 5))
    scoring = sa.Scoring.from_names("ga", "blosum62", gap_open=10, gap_extend=1)
    full = tri_to_full(oracle.align(store, scoring, triangular=True), n)
    nc, want = expected_tiles(full, chunk)
    monkeypatch.setenv("SA_HIP_TILES_SPLIT", str(parts))
    seen, order = {}, []
    with sa.DeflateJob.begin(store, scoring, chunk, level=level) as job:
        while True:
            batch = job.next()
            if not batch:
                break
            for r, c, z in batch:
                assert (r, c) not in seen
                seen[(r, c)] = zlib.decompress(z) if level else z
                order.append(max(r, c))
        assert order == sorted(order) and job.align_seconds > 0
        st = job.stats()
        assert st["raw_bytes"] == nc * nc * chunk * chunk * 4
    assert len(seen) == nc * nc and all(seen[k] == want[k[0]][k[1]] for k in seen)
