"""CPU: the C host around the device boundary (cli/sa_host.c) -- FASTA/DSV parsing, similarity filter,
HDF5 writer -- against the reference's documented behaviour, its golden filter result, and (where
oracle/_ref exists) the reference's own HDF5 writer via h5diff."""
import json
import subprocess

import numpy as np
import pytest

from tests.golden_util import GOLDEN_DIR, load_case, tri_to_full
from tests.host_binding import H5DIFF, H5DUMP, ROOT, Host, HostError, h5_matrix, h5_sequences
from tests.oracle_binding import RefLib, ref_available


@pytest.fixture(scope="module")
def host():
    return Host()


@pytest.fixture(scope="module")
def amino_lut(sa):
    return sa.Scoring.from_names("nw", "blosum62", gap_pen=4).lut


@pytest.fixture(scope="module")
def dna_lut(sa):
    return sa.Scoring.from_names("nw", "nuc44", gap_pen=4).lut


def test_fasta_parsing(host, amino_lut):
    data = b">one desc\r\nARND\r\ncqeg\r\n>two\nHIL K\n\n>three\nW\n"
    assert host.parse_sequences(data, "fasta", amino_lut) == [b"ARNDCQEG", b"HILK", b"W"]
    for ext in ("fa", "FAS", "fna", "ffn", "faa", "frn", "mpfa"):  # fasta.c:12-14, case-insensitive
        assert host.parse_sequences(b">a\nAR\n>b\nND\n", ext, amino_lut) == [b"AR", b"ND"]
    for bad, msg in ((b"ARND\n>a\nAR\n", "Data before first header"), (b">a\nAR\n>b\n", "Last header has no data"),
                     (b">a\nAR\n>b\n>c\nAR\n", "Sequence #2 is empty"), (b">a\nAR\n>b\nAJ\n", "Sequence #2 is invalid"),
                     (b">a\nAR\n", "Not enough sequences"), (b">a\nA\xe9\n>b\nA\n", "Sequence #1 is corrupted")):
        with pytest.raises(HostError, match=msg):
            host.parse_sequences(bad, "fasta", amino_lut)
    with pytest.raises(HostError, match="Unsupported file format"):
        host.parse_sequences(b">a\nAR\n>b\nND\n", "txt", amino_lut)
    # length limit len <= ((2^31-2)/2)/gap (io/input.c:15-19)
    with pytest.raises(HostError, match="exceeds length limits"):
        host.parse_sequences(b">a\n" + b"A" * 50 + b"\n>b\nA\n", "fasta", amino_lut, gap=-(2**26))


def test_dsv_parsing(host, amino_lut, dna_lut):
    csv = b'id,Sequence,note\n1,arnd,"x, y"\n2,"CQ EG",z\r\n3,W,"say ""hi"""\n'
    assert host.parse_sequences(csv, "csv", amino_lut) == [b"ARND", b"CQEG", b"W"]
    assert host.parse_sequences(b"name\tdna\nx\tACGT\ny\tGGN\n", "tsv", dna_lut) == [b"ACGT", b"GGN"]
    assert host.parse_sequences(b"seq;k\nAR;1\nND;2\n", "ssv", amino_lut) == [b"AR", b"ND"]
    assert host.parse_sequences(b"k|peptide\n1|AR\n2|ND\n", "psv", amino_lut) == [b"AR", b"ND"]
    # no recognised header name: needs the answer the reference would prompt for (dsv.c:139-151)
    with pytest.raises(HostError, match="No sequence column"):
        host.parse_sequences(b"a,b\n1,AR\n2,ND\n", "csv", amino_lut)
    assert host.parse_sequences(b"a,b\n1,AR\n2,ND\n", "csv", amino_lut, column=1) == [b"AR", b"ND"]
    assert host.parse_sequences(b"1,AR\n2,ND\n", "csv", amino_lut, column=1, has_header=False) == [b"AR", b"ND"]
    for bad, msg in ((b"id,seq\n1,AR\n2\n", "no sequence column"), (b"id,seq,x\n1,AR\n", "too few columns"),
                     (b"id,seq\n1,AR,9\n2,ND\n", "too many columns"), (b"id,seq,x\n1,,9\n2,ND,8\n", "Sequence #1 is empty"),
                     (b"id,seq\n1,AR\n2,N1\n", "Sequence #2 is invalid")):
        with pytest.raises(HostError, match=msg):
            host.parse_sequences(bad, "csv", amino_lut)


def test_filter_matches_reference_golden_and_oracle(host, oracle, sa, amino_lut):
    z = np.load(GOLDEN_DIR / "filter_f0.9.npz")
    meta = np.ascontiguousarray(z["meta"], np.int32)
    store = sa.SequenceStore(blob=np.ascontiguousarray(z["blob"]), meta=meta, num=meta.shape[0], max=int(meta[:, 1].max()))
    seqs = [store.sequence(k) for k in range(store.num)]
    thr = json.loads(str(z["params"]))["threshold"]
    want = [seqs[k] for k in z["kept"]]
    for threads in (1, 4):
        assert host.filter(seqs, amino_lut, thr, threads) == want
    # block boundaries of the blocked implementation: > 256 sequences, duplicates straddling blocks
    from tests.synth import make_near_duplicates, make_protein_set
    big = make_near_duplicates(make_protein_set(700, 30, 60, 9), 0.3, 0.05, 9)
    st = sa.SequenceStore.from_sequences(big)
    keep = oracle.filter(st, 0.9)
    assert 0 < keep.sum() < len(big)
    assert host.filter(big, amino_lut, 0.9, 3) == [s for s, k in zip(big, keep) if k]
    assert host.filter(big, amino_lut, 0.0) == big


def test_hdf5_chunk_rule(host):
    # hdf5.c:70-84: contiguous up to 256; else clamp(largest 64*2^k <= N, 256, 4096)
    assert [host.chunk_dim(n) for n in (2, 100, 256, 257, 300, 511, 512, 1000, 1024, 5000, 8192, 100000)] == \
        [2, 100, 256, 256, 256, 256, 512, 512, 1024, 4096, 4096, 4096]


@pytest.mark.skipif(not H5DUMP.exists(), reason="h5dump not available")
def test_hdf5_roundtrip_full_and_triangular(host, tmp_path, amino_lut):
    store, scoring, expected, full = load_case("cfg1_nw_blosum62_p4")
    seqs = [store.sequence(k) for k in range(store.num)]
    want = tri_to_full(expected, store.num)
    for tri, z in ((False, 0), (True, 0), (True, 6)):
        path = tmp_path / f"out_{int(tri)}_{z}.h5"
        host.write_hdf5(path, seqs, amino_lut, expected if tri else want, tri, z)
        assert np.array_equal(h5_matrix(path, store.num), want)
        assert h5_sequences(path) == seqs
    # chunked + deflate path (N > 256), packed input, diagonal must be written as zeros
    n = 300
    rng = np.random.default_rng(3)
    tri = rng.integers(-50, 50, n * (n - 1) // 2, dtype=np.int32)
    seqs300 = [bytes(rng.choice(list(b"ARNDCQEG"), 5).astype(np.uint8)) for _ in range(n)]
    path = tmp_path / "big.h5"
    host.write_hdf5(path, seqs300, amino_lut, tri, True, 6)
    assert np.array_equal(h5_matrix(path, n), tri_to_full(tri, n))
    hdr = subprocess.run([str(H5DUMP), "-H", "-p", str(path)], capture_output=True, text=True).stdout
    assert "CHUNKED ( 256, 256 )" in hdr and "DEFLATE { LEVEL 6 }" in hdr and "H5T_STD_I32LE" in hdr
    # 2 x 2 tiles of 512 with ragged edges, deflated in parallel and handed over with H5Dwrite_chunk (cli/sa_host.c:
    # write_deflated_tiles), packed and full input; cross-check: libhdf5's own filter in one thread (SA_HOST_SERIAL_DEFLATE)
    n = 700
    tri = rng.integers(-500, 500, n * (n - 1) // 2, dtype=np.int32)
    seqs700 = [bytes(rng.choice(list(b"ARNDCQEG"), 4).astype(np.uint8)) for _ in range(n)]
    want = tri_to_full(tri, n)
    for packed in (True, False):
        path = tmp_path / f"par_{int(packed)}.h5"
        host.write_hdf5(path, seqs700, amino_lut, tri if packed else want, packed, 6)
        assert np.array_equal(h5_matrix(path, n), want)
        assert h5_sequences(path) == seqs700
        hdr = subprocess.run([str(H5DUMP), "-H", "-p", str(path)], capture_output=True, text=True).stdout
        assert "CHUNKED ( 512, 512 )" in hdr and "DEFLATE { LEVEL 6 }" in hdr
    import os
    os.environ["SA_HOST_SERIAL_DEFLATE"] = "1"
    try:
        path = tmp_path / "lib_filter.h5"
        host.write_hdf5(path, seqs700, amino_lut, tri, True, 6)
    finally:
        del os.environ["SA_HOST_SERIAL_DEFLATE"]
    assert np.array_equal(h5_matrix(path, n), want)
    if H5DIFF.exists():
        res = subprocess.run([str(H5DIFF), str(path), str(tmp_path / "par_1.h5")], capture_output=True, text=True)
        assert res.returncode == 0, res.stdout + res.stderr
    # tiles of 2048 x 2048 (16 MB): four 4 MB segments per tile, deflated independently and stitched into one zlib stream
    # (full-flush cuts, one header, combined Adler-32) -- libhdf5's own inflate must read it back
    n = 2100
    tri = rng.integers(-300, 80, n * (n - 1) // 2, dtype=np.int32)
    seqs_n = [b"ARND"] * n
    want = tri_to_full(tri, n)
    path = tmp_path / "segments.h5"
    host.write_hdf5(path, seqs_n, amino_lut, tri, True, 2)
    assert np.array_equal(h5_matrix(path, n), want)
    hdr = subprocess.run([str(H5DUMP), "-H", "-p", str(path)], capture_output=True, text=True).stdout
    assert "CHUNKED ( 2048, 2048 )" in hdr and "DEFLATE { LEVEL 2 }" in hdr
    os.environ["SA_HOST_SERIAL_DEFLATE"] = "1"
    try:
        ref_path = tmp_path / "segments_lib.h5"
        host.write_hdf5(ref_path, seqs_n, amino_lut, want, False, 2)
    finally:
        del os.environ["SA_HOST_SERIAL_DEFLATE"]
    if H5DIFF.exists():
        res = subprocess.run([str(H5DIFF), str(ref_path), str(path)], capture_output=True, text=True)
        assert res.returncode == 0, res.stdout + res.stderr


@pytest.mark.skipif(not (ref_available() and H5DIFF.exists()), reason="needs oracle/_ref and h5diff")
@pytest.mark.parametrize("n,z", [(100, 0), (300, 6), (700, 9), (700, 1)])  # (N > 256 with -z: tiles deflated by all cores, H5Dwrite_chunk)
def test_hdf5_equals_reference_writer(host, tmp_path, amino_lut, sa, oracle, n, z):
    from tests.synth import make_protein_set
    seqs = make_protein_set(n, 5, 20, 13)
    store = sa.SequenceStore.from_sequences(seqs)
    scoring = sa.Scoring.from_names("nw", "blosum62", gap_pen=4)
    full = oracle.align(store, scoring, triangular=False)
    ref = RefLib("nw", "blosum62", gap_pen=4, compression=z)
    try:
        ref_path = ref.flush_hdf5(store, full, triangular=False)
        mine = tmp_path / "mine.h5"
        host.write_hdf5(mine, seqs, amino_lut, full, False, z)
        res = subprocess.run([str(H5DIFF), str(ref_path), str(mine)], capture_output=True, text=True)
        assert res.returncode == 0, res.stdout + res.stderr
        props = lambda p: [l.strip() for l in subprocess.run([str(H5DUMP), "-H", "-p", str(p)], capture_output=True, text=True).stdout.splitlines()
                           if any(k in l for k in ("CHUNKED", "DEFLATE", "CONTIGUOUS", "DATATYPE", "DATASPACE"))]
        assert props(ref_path) == props(mine)
    finally:
        ref.close()


def test_matrix_allocator_file_backed_branch(host, tmp_path, monkeypatch):
    """output_load's temporary-file storage (reference src/io/output.c:36-55, src/system/os.c:112-125): once the full
    matrix exceeds 3/4 of MemAvailable the (triangular) matrix is a MAP_SHARED mapping of an unnamed file.  The
    threshold is forced through SA_HOST_MEM_AVAILABLE."""
    import ctypes as C
    lib = host.lib
    lib.sa_host_matrix_alloc.argtypes = [C.c_size_t, C.c_bool, C.c_bool]
    lib.sa_host_matrix_alloc.restype = C.POINTER(C.c_int32)
    lib.sa_host_matrix_free.argtypes = [C.POINTER(C.c_int32), C.c_size_t, C.c_bool]
    lib.sa_host_matrix_needs_file.argtypes = [C.c_size_t]
    lib.sa_host_matrix_needs_file.restype = C.c_bool
    lib.sa_host_available_memory.restype = C.c_size_t
    n = 300
    assert lib.sa_host_available_memory() > 0 and not lib.sa_host_matrix_needs_file(n)
    monkeypatch.setenv("SA_HOST_MEM_AVAILABLE", str(4 * n * n))  # full matrix = 4/3 of "3/4 of available"
    assert lib.sa_host_available_memory() == 4 * n * n and lib.sa_host_matrix_needs_file(n)
    monkeypatch.setenv("SA_HOST_MEM_AVAILABLE", str(2 * 4 * n * n))
    assert not lib.sa_host_matrix_needs_file(n)
    monkeypatch.setenv("TMPDIR", str(tmp_path))

    def mapping_of(addr):
        for line in open("/proc/self/maps"):
            lo, hi = (int(x, 16) for x in line.split()[0].split("-"))
            if lo <= addr < hi:
                return line
        return ""

    for file_backed in (False, True):
        m = lib.sa_host_matrix_alloc(n, True, file_backed)
        assert m, host._err()
        elems = n * (n - 1) // 2
        arr = np.ctypeslib.as_array(m, shape=(elems,))
        assert not arr.any()  # zero-filled like the anonymous mapping: the diagonal is never written (output.c:76-81)
        arr[:] = np.arange(elems, dtype=np.int32)
        assert int(arr[-1]) == elems - 1
        line = mapping_of(C.addressof(m.contents))
        if file_backed:  # a shared mapping of a deleted (unnamed) file under $TMPDIR
            assert " rw-s " in line and str(tmp_path) in line and "(deleted)" in line, line
        else:
            assert " rw-p " in line and str(tmp_path) not in line, line
        del arr
        lib.sa_host_matrix_free(m, n, True)
    assert list(tmp_path.iterdir()) == []  # nothing left behind


@pytest.mark.skipif(not H5DIFF.exists(), reason="h5diff not available")
def test_hdf5_writer_under_sanitizers(tmp_path):
    """cli/sa_host.c linked into tests/host_c/writer_check.c with -fsanitize=address,undefined: the segment-parallel
    deflate path (tiles of 256 KB .. 64 MB, ragged edges, several segments per tile) against libhdf5's own filter path"""
    exe = tmp_path / "writer_check"
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-g", "-fopenmp", "-Wall", "-Wextra", "-I/opt/conda/include",
                           "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", str(exe),
                           str(ROOT / "tests" / "host_c" / "writer_check.c"), str(ROOT / "cli" / "sa_host.c"),
                           "-L/opt/conda/lib", "-lhdf5", "-lz", "-Wl,-rpath,/opt/conda/lib"])
    import os
    for n, z in ((300, 6), (700, 9), (2100, 2), (4200, 1)):
        a, b = tmp_path / f"p_{n}.h5", tmp_path / f"s_{n}.h5"
        res = subprocess.run([str(exe), str(n), str(z), str(a), str(b)], capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
        assert res.returncode == 0 and "runtime error" not in res.stderr and "AddressSanitizer" not in res.stderr, res.stderr[-3000:]
        diff = subprocess.run([str(H5DIFF), str(a), str(b)], capture_output=True, text=True)
        assert diff.returncode == 0, diff.stdout + diff.stderr
        a.unlink(), b.unlink()


@pytest.mark.parametrize("z", [6, 0])
def test_hdf5_from_finished_tiles(host, tmp_path, amino_lut, z):
    """sa_host_write_hdf5_streams: the tiles arrive finished (zlib streams for -z, raw tiles without) in batches of at most
    ceil(N / chunk) tiles, in ANY order -- here in the shell order of the device walk (include/seqalign_hip.h: sa_zjob_next) --
    and go to H5Dwrite_chunk as they are; the file reads back through libhdf5 as the matrix, with the dataset layout of
    sa_host_write_hdf5.  A source that reports an error, or delivers too few tiles, fails the call."""
    import zlib
    n = 700
    rng = np.random.default_rng(11)
    tri = rng.integers(-300, 200, n * (n - 1) // 2, dtype=np.int32)
    want = tri_to_full(tri, n)
    seqs = [bytes(rng.choice(list(b"ARNDCQEG"), 4).astype(np.uint8)) for _ in range(n)]
    chunk = host.chunk_dim(n)
    nc = -(-n // chunk)
    pad = np.zeros((nc * chunk, nc * chunk), np.int32)
    pad[:n, :n] = want

    def tile(r, c):
        raw = pad[r * chunk:(r + 1) * chunk, c * chunk:(c + 1) * chunk].astype("<i4").tobytes()
        return (r, c, zlib.compress(raw, 1) if z else raw)
    shells = []
    for b in range(nc):
        shells.append([tile(b, c) for c in range(b + 1)])
        if b:
            shells.append([tile(r, b) for r in range(b)])
    path = tmp_path / "tiles.h5"
    host.write_hdf5_streams(path, seqs, amino_lut, z, shells)
    assert np.array_equal(h5_matrix(path, n), want)
    assert h5_sequences(path) == seqs
    hdr = subprocess.run([str(H5DUMP), "-H", "-p", str(path)], capture_output=True, text=True).stdout
    assert f"CHUNKED ( {chunk}, {chunk} )" in hdr and ("DEFLATE { LEVEL 6 }" in hdr) == bool(z)
    ref = tmp_path / "ref.h5"
    host.write_hdf5(ref, seqs, amino_lut, tri, True, z)
    assert subprocess.run([str(H5DIFF), str(path), str(ref)], capture_output=True).returncode == 0
    with pytest.raises(HostError, match="tiles"):
        host.write_hdf5_streams(tmp_path / "short.h5", seqs, amino_lut, z, shells[:-1])
    with pytest.raises(HostError, match="encode"):
        host.write_hdf5_streams(tmp_path / "bad.h5", seqs, amino_lut, z, shells[:1] + [None])
    with pytest.raises(HostError, match="chunked"):
        host.write_hdf5_streams(tmp_path / "small.h5", seqs[:100], amino_lut, z, [])
