"""GPU (-m gpu): the `seqalign` tool end to end (FASTA/DSV in -> HIP alignment -> HDF5 out) against the
reference-generated golden matrices."""
import json
import subprocess

import numpy as np
import pytest

from tests.golden_util import GOLDEN_DIR, load_case, tri_to_full
from tests.host_binding import ROOT, h5_matrix, h5_sequences

pytestmark = pytest.mark.gpu
CLI = ROOT / "cli" / "seqalign"


@pytest.fixture(scope="module", autouse=True)
def built_cli():
    if not CLI.exists():
        subprocess.check_call(["make", "-s", "-C", str(ROOT / "cli")])


def run(*args, check=True, env=None):
    import os
    res = subprocess.run([str(CLI), *map(str, args)], capture_output=True, text=True, timeout=600,
                         env=None if env is None else dict(os.environ, **env))
    if check:
        assert res.returncode == 0, res.stdout + res.stderr
    return res


def write_fasta(path, seqs):
    path.write_bytes(b"".join(b">seq%d\n" % k + s[:60] + b"\n" + s[60:] + b"\n" for k, s in enumerate(seqs)))


@pytest.mark.parametrize("case,flags", [
    ("cfg1_nw_blosum62_p4", ["-a", "nw", "-m", "blosum62", "-p", 4]),
    ("p64_ga_blosum62_10_1", ["-a", "Gotoh", "-m", "BLOSUM62", "-s", 10, "-e", 1]),
    ("d64_sw_nuc44_10_1", ["-a", "sw", "-m", "nuc44", "-s", 10, "-e", 1, "-z", 6]),
    ("ext_nw_blosum62_p4", ["-a", "nw", "-m", "blosum62", "--gap-penalty=4"]),
    ("edge_ga_4_4_becomes_nw", ["-a", "ga", "-m", "blosum62", "-s", 4, "-e", 4]),
])
def test_cli_matches_reference_golden(case, flags, tmp_path):
    store, scoring, expected, _ = load_case(case)
    seqs = [store.sequence(k) for k in range(store.num)]
    fasta, out = tmp_path / "in.fasta", tmp_path / "out.h5"
    write_fasta(fasta, seqs)
    res = run("-i", fasta, "-o", out, *flags, "-F", "-B")
    assert "Alignments per second" in res.stdout
    assert np.array_equal(h5_matrix(out, store.num), tri_to_full(expected, store.num))
    assert h5_sequences(out) == seqs


@pytest.mark.parametrize("case,flags", [
    ("cfg1_nw_blosum62_p4", ["-a", "nw", "-m", "blosum62", "-p", 4]),
    ("p64_sw_blosum62_10_1", ["-a", "sw", "-m", "blosum62", "-s", 10, "-e", 1, "-z", 3]),
])
def test_cli_through_the_all_gather_path(case, flags, tmp_path):
    """the tool on the schedule it takes with several GPUs -- dense shares, RCCL all-gather, placement, shells to the host
    (csrc/sa_gather.hip) -- forced onto the one device of this box (a one-rank communicator): the HDF5 it writes is the
    reference's golden, and -B names the schedule"""
    store, scoring, expected, _ = load_case(case)
    seqs = [store.sequence(k) for k in range(store.num)]
    fasta, out = tmp_path / "in.fasta", tmp_path / "out.h5"
    write_fasta(fasta, seqs)
    res = run("-i", fasta, "-o", out, *flags, "-F", "-B", env={"SA_HIP_GATHER": "1"})
    assert "RCCL all-gather" in res.stdout, res.stdout
    assert np.array_equal(h5_matrix(out, store.num), tri_to_full(expected, store.num))
    assert h5_sequences(out) == seqs


def test_cli_dsv_filter_and_flags(tmp_path):
    z = np.load(GOLDEN_DIR / "filter_f0.9.npz")
    meta = z["meta"]
    blob = z["blob"]
    seqs = [blob[o:o + l].tobytes() for o, l in meta]
    csv = tmp_path / "in.csv"
    csv.write_bytes(b"id,sequence\n" + b"".join(b"%d,%s\n" % (k, s) for k, s in enumerate(seqs)))
    out = tmp_path / "f.h5"
    run("-i", csv, "-o", out, "-a", "nw", "-m", "blosum62", "-p", 4, "-f", 0.9, "-QF")
    kept = [seqs[k] for k in z["kept"]]
    assert h5_sequences(out) == kept
    m = h5_matrix(out, len(kept))
    assert np.array_equal(m, m.T) and not np.diag(m).any()
    # existing output without -F is refused (non-interactive default "no"), -W computes without writing
    res = run("-i", csv, "-o", out, "-a", "nw", "-m", "blosum62", "-p", 4, check=False)
    assert res.returncode == 1 and "will not be overwritten" in res.stderr
    run("-i", csv, "-W", "-a", "nw", "-m", "blosum62", "-p", 4, "-Q")
    for bad in (["-a", "nw", "-m", "blosum62", "-s", 1, "-e", 1], ["-a", "sw", "-m", "blosum62", "-p", 1],
                ["-a", "nw", "-m", "nope", "-p", 1], ["-a", "xx", "-m", "blosum62", "-p", 1]):
        res = run("-i", csv, "-W", *bad, check=False)
        assert res.returncode == 1 and "usage information" in res.stderr


def test_cli_progress_line(tmp_path):
    """without -P the tool reports progress on stderr like the reference (ppercent / pproportc, seqalign_cuda.c:181,286-293);
    -P and -Q switch it off"""
    from tests.synth import make_dna_set
    seqs = make_dna_set(3000, 120, 180, 4)
    fasta = tmp_path / "in.fasta"
    write_fasta(fasta, seqs)
    res = run("-i", fasta, "-W", "-a", "sw", "-m", "nuc44", "-s", 10, "-e", 1, "-F")
    assert "Aligning sequences:   0%" in res.stderr and "Aligning sequences: 100%" in res.stderr
    for flag in ("-P", "-Q"):
        res = run("-i", fasta, "-W", "-a", "sw", "-m", "nuc44", "-s", 10, "-e", 1, "-F", flag)
        assert "Aligning sequences" not in res.stderr


@pytest.mark.parametrize("n,method,flags", [
    (700, "nw", ["-a", "nw", "-m", "blosum62", "-p", 4]),
    (1300, "sw", ["-a", "sw", "-m", "blosum62", "-s", 10, "-e", 1]),
])
def test_cli_z_deflates_on_the_device(n, method, flags, tmp_path, sa):
    """-z on a chunked dataset (N > 256): the tiles leave the GPU as zlib streams (csrc/sa_deflate.hip) and go to
    H5Dwrite_chunk; the file reads back -- through libhdf5's own inflate -- as the same matrix the uncompressed path and the
    all-cores zlib path (SA_HOST_CPU_DEFLATE=1) write, with the same dataset layout (h5diff, h5dump -p)."""
    from tests.host_binding import H5DUMP
    from tests.synth import make_protein_set
    seqs = make_protein_set(n, 30, 150, 11)
    fasta = tmp_path / "in.fasta"
    write_fasta(fasta, seqs)
    dev, cpu, plain = tmp_path / "dev.h5", tmp_path / "cpu.h5", tmp_path / "plain.h5"
    res = run("-i", fasta, "-o", dev, *flags, "-z", 6, "-B", "-F", "-V")
    assert "tiles deflated on the device" in res.stdout and "Deflated on the device" in res.stdout + res.stderr
    run("-i", fasta, "-o", cpu, *flags, "-z", 6, "-F", "-Q", env={"SA_HOST_CPU_DEFLATE": "1"})
    run("-i", fasta, "-o", plain, *flags, "-F", "-Q")
    gaps = {"gap_pen": 4} if method == "nw" else {"gap_open": 10, "gap_extend": 1}
    want = tri_to_full(sa.hip_align(sa.SequenceStore.from_sequences(seqs), sa.Scoring.from_names(method, "blosum62", **gaps),
                                    triangular=True), n)
    for path in (dev, cpu, plain):
        assert np.array_equal(h5_matrix(path, n), want), path
    assert h5_sequences(dev) == [s for s in seqs]
    h5diff = H5DUMP.with_name("h5diff")
    assert subprocess.run([str(h5diff), str(dev), str(cpu)], capture_output=True).returncode == 0
    props = subprocess.run([str(H5DUMP), "-p", "-H", "-d", "/similarity_matrix", str(dev)], capture_output=True, text=True).stdout
    assert "COMPRESSION DEFLATE { LEVEL 6 }" in props and "CHUNKED" in props
    assert dev.stat().st_size < 0.6 * plain.stat().st_size


def test_cli_tiles_without_z_come_from_the_device(tmp_path, sa):
    """N > 256 without -z on one device: the chunks are tiled on the device and written with H5Dwrite_chunk; the file is
    the one the host-matrix path (SA_HOST_MATRIX=1: sa_hip_align + H5Dwrite) writes"""
    from tests.host_binding import H5DUMP
    from tests.synth import make_dna_set
    n = 1100
    seqs = make_dna_set(n, 60, 160, 9)
    fasta = tmp_path / "in.fasta"
    write_fasta(fasta, seqs)
    dev, host = tmp_path / "dev.h5", tmp_path / "host.h5"
    flags = ["-a", "sw", "-m", "nuc44", "-s", 10, "-e", 1]
    res = run("-i", fasta, "-o", dev, *flags, "-B", "-F", "-Q")
    if sa.device_count() == 1:
        assert "tiles delivered as HDF5 chunks" in res.stdout
    run("-i", fasta, "-o", host, *flags, "-F", "-Q", env={"SA_HOST_MATRIX": "1"})
    want = tri_to_full(sa.hip_align(sa.SequenceStore.from_sequences(seqs), sa.Scoring.from_names("sw", "nuc44", gap_open=10, gap_extend=1),
                                    triangular=True), n)
    assert np.array_equal(h5_matrix(dev, n), want) and np.array_equal(h5_matrix(host, n), want)
    assert subprocess.run([str(H5DUMP.with_name("h5diff")), str(dev), str(host)], capture_output=True).returncode == 0
