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
    assert lib.sa_abi_version() == 3


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
