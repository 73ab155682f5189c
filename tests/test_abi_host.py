"""CPU: the C-ABI library loads, exports every symbol include/seqalign_hip.h declares, and its host-only
logic (option tables, validation, planning) behaves like the reference's.  No compute calls."""
import pathlib
import re

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]


def declared_symbols() -> list[str]:
    text = (ROOT / "include" / "seqalign_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sa_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(sa):
    from sequencealigner_amd.binding import ABI_SYMBOLS
    lib = sa.load_library()
    declared = declared_symbols()
    assert declared, "header parse failed"
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in include/seqalign_hip.h but not exported"
    assert sorted(ABI_SYMBOLS) == declared
    assert lib.sa_abi_version() == 4


def test_product_library_does_not_link_the_oracle():
    import subprocess
    import sequencealigner_amd as sa
    out = subprocess.run(["ldd", str(sa.library_path())], capture_output=True, text=True).stdout
    assert "oracle" not in out and "seqalign_ref" not in out
    for py in (ROOT / "sequencealigner_amd").rglob("*.py"):
        assert "oracle" not in py.read_text().replace("no oracle", ""), f"{py} mentions the oracle"


def test_matrix_table(sa):
    names = sa.matrix_names()
    assert len(names) == 67 and names[8] == "blosum62" and names[-2:] == ["dnafull", "nuc44"]
    s = sa.Scoring.from_names("nw", "BLOSUM62", gap_pen=4)  # case-insensitive (bio/matrices.c:47)
    sub = s.sub.reshape(24, 24)
    assert np.array_equal(sub, sub.T)
    order = "ARNDCQEGHILKMFPSTWYVBZX*"
    assert [int(s.lut[ord(c)]) for c in order] == list(range(24))
    assert s.lut[ord("a")] == -1 and s.lut[ord("J")] == -1 and s.lut[0] == -1
    assert sub[order.index("W"), order.index("W")] == 11 and sub[order.index("A"), order.index("R")] == -1
    assert sub.min() == -4 and sub.max() == 11
    n = sa.Scoring.from_names("sw", "nuc44", gap_open=10, gap_extend=1)
    nsub = n.sub.reshape(24, 24)
    assert [int(n.lut[ord(c)]) for c in "ATGCSWRYKMBVHDN*"] == list(range(16))
    assert nsub[0, 0] == 5 and nsub[0, 1] == -4 and not nsub[16:, :].any() and not nsub[:, 16:].any()
    with pytest.raises(sa.AlignError):
        sa.Scoring.from_names("nw", "blosum63", gap_pen=4)


def test_method_aliases_and_gap_rules(sa):
    # parse_align: long and short aliases, case-insensitive (bio/align.c:87-96)
    for alias, m in (("nw", 0), ("Needleman-Wunsch", 0), ("NEEDLEMAN-WUNSCH", 0), ("ga", 1), ("gotoh", 1), ("SW", 2),
                     ("smith-waterman", 2)):
        kw = dict(gap_pen=4) if m == 0 else dict(gap_open=10, gap_extend=1)
        assert sa.Scoring.from_names(alias, "blosum62", **kw).method == m
    with pytest.raises(sa.AlignError):
        sa.Scoring.from_names("needleman", "blosum62", gap_pen=4)
    # stored negated (bio/align.c:127)
    s = sa.Scoring.from_names("ga", "blosum62", gap_open=10, gap_extend=1)
    assert (s.gap_opn, s.gap_ext) == (-10, -1)
    # -p with an affine method / -s -e with a linear one (bio/align.c:130-142)
    with pytest.raises(sa.AlignError):
        sa.Scoring.from_names("sw", "blosum62", gap_pen=4)
    with pytest.raises(sa.AlignError):
        sa.Scoring.from_names("nw", "blosum62", gap_open=10, gap_extend=1)
    with pytest.raises(sa.AlignError):
        sa.Scoring.from_names("ga", "blosum62", gap_open=10)
    with pytest.raises(sa.AlignError):
        sa.Scoring.from_names("nw", "blosum62", gap_pen=-3)
    # Gotoh with equal gaps becomes NW (validate_ga, bio/method/ga.c:70-88)
    s = sa.Scoring.from_names("ga", "blosum62", gap_open=4, gap_extend=4)
    assert s.method == 0 and s.gap_pen == -4 and s.gap_opn == s.gap_ext == -(1 << 30)
    s = sa.Scoring.from_names("sw", "blosum62", gap_open=4, gap_extend=4)
    assert s.method == 2


def test_sequence_store_layout(sa):
    st = sa.SequenceStore.from_sequences(["a", "arndw", b"WWWW*"])
    assert st.blob.tobytes() == b"A\0ARNDW\0WWWW*\0"
    assert st.meta.tolist() == [[0, 1], [2, 5], [8, 5]] and st.max == 5 and st.num == 3
    assert st.sequence(1) == b"ARNDW" and st.pairs == 3


def test_pair_planning(sa, oracle):
    from tests.synth import make_protein_set
    seqs = make_protein_set(500, 10, 300, 21)
    st = sa.SequenceStore.from_sequences(seqs)
    lens = st.meta[:, 1].astype(np.int64)
    total = int(sum(int(lens[j]) * int(lens[:j].sum()) for j in range(st.num)))
    assert st.cells() == total
    # arbitrary sub-range against brute force through the oracle's index mapping
    start, count = 12345, 4321
    brute = 0
    for p in range(start, start + count):
        i, j = oracle.unpack(p)
        brute += int(lens[i]) * int(lens[j])
    assert st.cells(start, count) == brute
    for parts in (1, 2, 3, 8):
        b = st.partition(parts)
        assert b[0] == 0 and b[-1] == st.pairs and all(x <= y for x, y in zip(b, b[1:]))
        work = [st.cells(b[k], b[k + 1] - b[k]) for k in range(parts)]
        assert sum(work) == total
        assert max(work) - min(work) <= 2 * int(lens.max()) ** 2  # within one pair of equal
    with pytest.raises(sa.AlignError):
        st.cells(0, st.pairs + 1)


def test_no_device_fails_loudly(sa):
    if sa.device_count() > 0:
        pytest.skip("a HIP device is visible")
    st = sa.SequenceStore.from_sequences(["ARND", "ARNW"])
    with pytest.raises(sa.AlignError, match="No HIP devices"):
        sa.hip_align(st, sa.Scoring.from_names("nw", "blosum62", gap_pen=4))
    with pytest.raises(sa.AlignError):
        sa.Context(st, sa.Scoring.from_names("nw", "blosum62", gap_pen=4))
    assert sa.hip_memory(1 << 20) is False


_ALLOC_FAILURE = r"""
import ctypes as C, resource, sys
import numpy as np
import sequencealigner_amd as sa
lib = sa.load_library()
n = 40_000_000                                  # sa_pairs_partition needs two int64 prefix arrays: 640 MB
meta = np.zeros((n, 2), np.int32); meta[:, 1] = 50
bounds = (C.c_int64 * 3)()
vm = int(next(l for l in open("/proc/self/status") if l.startswith("VmSize")).split()[1]) * 1024
soft, hard = resource.getrlimit(resource.RLIMIT_AS)
resource.setrlimit(resource.RLIMIT_AS, (vm + (300 << 20), hard))   # soft limit only: it can be lifted again
rc = lib.sa_pairs_partition(meta.ctypes.data, n, 2, bounds)   # std::bad_alloc behind the extern "C" boundary
print("rc", rc, "| error:", lib.sa_last_error().decode())
cells = lib.sa_pairs_cells(meta.ctypes.data, n, 0, 10)
print("cells", cells, "| error:", lib.sa_last_error().decode())
resource.setrlimit(resource.RLIMIT_AS, (soft, hard))
rc = lib.sa_pairs_partition(meta.ctypes.data, n, 2, bounds)   # ... and the library is still usable afterwards
print("again", rc, bounds[1] > 0, bounds[2] == n * (n - 1) // 2)
"""


def test_allocation_failure_behind_the_abi_is_an_error_not_an_abort():
    """The reference's failure model is `perr` + `return false` (src/interface/seqalign_cuda.c:23-30).  The host side of
    this library is C++: a std::bad_alloc (or any other exception) raised behind an extern "C" entry point must come back
    the same way -- failure value + sa_last_error() -- never as std::terminate / SIGABRT of the host process.  Forced
    here with RLIMIT_AS in a child process: the two prefix arrays of sa_pairs_partition cannot be allocated."""
    import subprocess
    import sys
    res = subprocess.run([sys.executable, "-c", _ALLOC_FAILURE], capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert res.returncode == 0, f"child died with {res.returncode} (an abort is -6):\n{res.stdout}\n{res.stderr[-3000:]}"
    lines = res.stdout.strip().splitlines()
    assert lines[0].startswith("rc 1 | error: sa_pairs_partition: out of host memory"), res.stdout
    assert lines[1].startswith("cells -1 | error: sa_pairs_cells: out of host memory"), res.stdout
    assert lines[2] == "again 0 True True", res.stdout


def test_every_entry_point_with_a_body_sits_behind_the_exception_barrier():
    """source-level guard: every extern "C" definition in csrc/ either runs its body through sa_guard / sa_guard_void or is
    on the short list of entry points that cannot throw (no allocation, no container, no std::string)"""
    nothrow = {"sa_last_error", "sa_abi_version", "sa_matrix_count", "sa_matrix_name", "sa_matrix_is_nucleotide", "sa_matrix_load",
               "sa_method_parse", "sa_method_name", "sa_method_gap_kind", "sa_hip_device_count", "sa_hip_device_name",
               "sa_ctx_pairs", "sa_ctx_scores_fit16", "sa_ctx_leave_room", "sa_hip_widen16", "sa_ctx_expand_full",
               "sa_hip_set_progress", "sa_hip_last_align_seconds", "sa_hip_last_align_path", "sa_hip_last_align_breakdown",
               "sa_hip_host_register", "sa_hip_host_unregister", "sa_ctx_destroy", "sa_ctx_timing", "sa_zjob_tiles_per_row", "sa_zjob_stats"}
    guarded, plain = set(), set()
    for src in sorted((ROOT / "sequencealigner_amd" / "csrc").glob("*")):
        if src.suffix not in (".hip", ".cpp"):
            continue
        text = src.read_text()
        for m in re.finditer(r'extern "C" [^\n;{]*?\b(sa_[a-z0-9_]+)\s*\([^;{]*\)\s*\{', text, flags=re.S):
            body = text[m.end():text.index("\n}\n", m.end())]
            (guarded if "sa_guard" in body else plain).add(m.group(1))
    assert guarded >= {"sa_hip_align", "sa_hip_memory", "sa_hip_filter", "sa_ctx_create", "sa_ctx_align_range", "sa_ctx_align_range16",
                       "sa_ctx_align_host", "sa_ctx_share_elems", "sa_ctx_align_share", "sa_ctx_place_shares", "sa_ctx_timing_read",
                       "sa_pairs_cells", "sa_pairs_partition", "sa_zjob_create", "sa_zjob_tile_row", "sa_zjob_next", "sa_zjob_align_seconds", "sa_zjob_destroy", "sa_hip_tiles_begin"}, guarded
    assert plain <= nothrow, f"entry points without a barrier that are not on the no-throw list: {sorted(plain - nothrow)}"
    assert sorted(guarded | plain) == declared_symbols()
