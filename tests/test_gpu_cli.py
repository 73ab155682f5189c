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
