#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE ITSELF (oracle/_ref/libseqalign_ref.so = the
reference's unmodified sources, see oracle/Makefile).  Run in the build container only
(needs /root/reference); the fixtures are data: inputs + the scores the reference produced.

Every case stores: blob/meta (the reference's `struct input`), params (json) and `expected`
(packed triangular s32, pair i<j at j(j-1)/2+i) obtained through the reference's own
align() + output_fill() with out.triangular = true; `expected_full` for the small sets is the
full-layout matrix (zero diagonal) from the same driver."""
import json
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

import sequencealigner_amd as sa  # noqa: E402  (host-side tables only; no GPU needed)
from tests.oracle_binding import RefLib  # noqa: E402
from tests.synth import make_dna_set, make_protein_set, splitmix64, AMINO20  # noqa: E402

OUT = ROOT / "tests" / "golden"

EDGE = [b"A", b"A", b"W", b"ARNDW", b"WWWW*", b"BZX", b"HEAGAWGHEE", b"PAWHEAE"]


def extreme_lengths(seed):
    lens = [1, 2, 3, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1024]
    out = []
    for k, ln in enumerate(lens):
        r = splitmix64(np.arange(ln), seed + k) % np.uint64(20)
        out.append(np.frombuffer(AMINO20, np.uint8)[r.astype(np.int64)].tobytes())
    return out


def filter_set(seed):
    bases = make_protein_set(30, 90, 110, seed)
    out = []
    r = splitmix64(np.arange(300 * 40), seed + 1000)
    for k in range(300):
        b = bytearray(bases[k % 30])
        nsub = int(r[40 * k]) % 13
        for t in range(nsub):
            b[int(r[40 * k + 1 + 2 * t]) % len(b)] = AMINO20[int(r[40 * k + 2 + 2 * t]) % 20]
        out.append(bytes(b))
    return out


def emit(name, seqs, method, matrix, gaps, full=False):
    store = sa.SequenceStore.from_sequences(seqs)
    ref = RefLib(method, matrix, **gaps)
    params = ref.params()
    expected = ref.align(store, triangular=True)
    data = dict(blob=store.blob, meta=store.meta, expected=expected,
                params=np.array(json.dumps(dict(method=method, matrix=matrix, gaps=gaps,
                                                stored=dict(method=params["method"], gap_pen=params["gap_pen"],
                                                            gap_opn=params["gap_opn"], gap_ext=params["gap_ext"])))))
    if full:
        data["expected_full"] = ref.align(store, triangular=False)
    ref.close()
    np.savez_compressed(OUT / f"{name}.npz", **data)
    print(f"{name}: {store.num} seqs, {expected.size} pairs, method={params['method']}")


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    P4 = dict(gap_pen=4)
    A101 = dict(gap_open=10, gap_extend=1)
    # (1) SURVEY.md §8(c) edge set x 5 parameter sets (+ the ga->nw swap)
    emit("edge_nw_p4", EDGE, "nw", "blosum62", P4, full=True)
    emit("edge_nw_p0", EDGE, "nw", "blosum62", dict(gap_pen=0), full=True)
    emit("edge_ga_10_1", EDGE, "ga", "blosum62", A101, full=True)
    emit("edge_sw_10_1", EDGE, "sw", "blosum62", A101, full=True)
    emit("edge_sw_0_0", EDGE, "sw", "blosum62", dict(gap_open=0, gap_extend=0), full=True)
    emit("edge_ga_4_4_becomes_nw", EDGE, "ga", "blosum62", dict(gap_open=4, gap_extend=4), full=True)
    # (2) cfg 1
    cfg1 = make_protein_set(100, 40, 60, 1)
    emit("cfg1_nw_blosum62_p4", cfg1, "nw", "blosum62", P4, full=True)
    # (3) 64 x U[80,120] protein, affine methods (+ unusual open<extend, + other matrices)
    p64 = make_protein_set(64, 80, 120, 3)
    emit("p64_ga_blosum62_10_1", p64, "ga", "blosum62", A101)
    emit("p64_sw_blosum62_10_1", p64, "sw", "blosum62", A101)
    emit("p64_nw_blosum62_p4", p64, "nw", "blosum62", P4)
    emit("p64_ga_blosum62_3_7", p64, "ga", "blosum62", dict(gap_open=3, gap_extend=7))
    emit("p64_sw_blosum62_2_5", p64, "sw", "blosum62", dict(gap_open=2, gap_extend=5))
    emit("p64_nw_pam250_p11", p64, "nw", "pam250", dict(gap_pen=11))
    emit("p64_ga_blosum45_12_2", p64, "ga", "blosum45", dict(gap_open=12, gap_extend=2))
    emit("p64_sw_pam30_9_3", p64, "sw", "pam30", dict(gap_open=9, gap_extend=3))
    emit("p64_nw_blosum100_p1", p64, "nw", "blosum100", dict(gap_pen=1))
    emit("p64_ga_blosum62_0_0", p64, "ga", "blosum62", dict(gap_open=0, gap_extend=0))
    # (4) DNA with IUPAC codes
    d64 = make_dna_set(64, 120, 180, 4, iupac=True)
    emit("d64_sw_nuc44_10_1", d64, "sw", "nuc44", A101)
    emit("d64_nw_nuc44_p4", d64, "nw", "nuc44", P4)
    emit("d64_ga_dnafull_16_4", d64, "ga", "dnafull", dict(gap_open=16, gap_extend=4))
    # (5) length extremes 1..1024
    ext = extreme_lengths(50)
    emit("ext_nw_blosum62_p4", ext, "nw", "blosum62", P4)
    emit("ext_ga_blosum62_10_1", ext, "ga", "blosum62", A101)
    emit("ext_sw_blosum62_10_1", ext, "sw", "blosum62", A101)
    # (6) similarity filter, sequential semantics (-T 1)
    fs = filter_set(60)
    store = sa.SequenceStore.from_sequences(fs)
    ref = RefLib("nw", "blosum62", gap_pen=4, threads=1, filter_threshold=0.9)
    kept = ref.filter(store)
    ref.close()
    # surviving sequences -> indices (first occurrence order is preserved by the compaction)
    idx, pos = [], 0
    for k, s in enumerate(fs):
        if pos < len(kept) and s == kept[pos]:
            idx.append(k)
            pos += 1
    assert pos == len(kept), "could not map survivors back to indices"
    np.savez_compressed(OUT / "filter_f0.9.npz", blob=store.blob, meta=store.meta, kept=np.array(idx, np.int32),
                        params=np.array(json.dumps(dict(threshold=0.9, threads=1))))
    print(f"filter_f0.9: {len(fs)} -> {len(kept)} kept")


if __name__ == "__main__":
    main()
