"""Deterministic synthetic inputs: identical bytes on every box, no RNG state.

splitmix64 over a counter, vectorised with numpy; lengths ~ U{lo..hi}, residues iid uniform."""
from __future__ import annotations

import numpy as np

AMINO20 = b"ARNDCQEGHILKMFPSTWYV"
DNA4 = b"ACGT"
IUPAC_EXTRA = b"SWRYKMBVHDN"

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(counter: np.ndarray, seed: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (counter.astype(np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _make(n: int, lo: int, hi: int, seed: int, alphabet: bytes) -> list[bytes]:
    lens = (lo + splitmix64(np.arange(n), seed * 2 + 1) % np.uint64(hi - lo + 1)).astype(np.int64)
    total = int(lens.sum())
    r = splitmix64(np.arange(total), seed * 2 + 2) % np.uint64(len(alphabet))
    letters = np.frombuffer(alphabet, dtype=np.uint8)[r.astype(np.int64)]
    out, pos = [], 0
    for ln in lens:
        out.append(letters[pos:pos + ln].tobytes())
        pos += int(ln)
    return out


def make_protein_set(n: int, lo: int, hi: int, seed: int) -> list[bytes]:
    return _make(n, lo, hi, seed, AMINO20)


def make_dna_set(n: int, lo: int, hi: int, seed: int, iupac: bool = False) -> list[bytes]:
    seqs = _make(n, lo, hi, seed, DNA4)
    if iupac:  # sprinkle a few ambiguity codes (deterministic positions)
        out = []
        for k, s in enumerate(seqs):
            b = bytearray(s)
            for t in range(0, len(b), 37):
                b[(t + k) % len(b)] = IUPAC_EXTRA[(t + k) % len(IUPAC_EXTRA)]
            out.append(bytes(b))
        seqs = out
    return seqs


def make_near_duplicates(base: list[bytes], frac_dup: float, max_sub_frac: float, seed: int, alphabet: bytes = AMINO20) -> list[bytes]:
    """cfg 5: overwrite a fraction of the set with copies of an EARLIER sequence carrying <= max_sub_frac
    substitutions, so that `-f 0.9` really removes sequences."""
    out = list(base)
    n = len(out)
    r = splitmix64(np.arange(3 * n), seed * 2 + 77)
    for k in range(1, n):
        if (int(r[3 * k]) % 10_000) >= int(frac_dup * 10_000):
            continue
        src = int(r[3 * k + 1]) % k
        b = bytearray(out[src])
        nsub = int(len(b) * max_sub_frac)
        rr = splitmix64(np.arange(2 * nsub + 1), int(r[3 * k + 2]) & 0x7FFFFFFF)
        for t in range(nsub):
            b[int(rr[2 * t]) % len(b)] = alphabet[int(rr[2 * t + 1]) % len(alphabet)]
        out[k] = bytes(b)
    return out


# BASELINE.json configs (sizes), SURVEY.md §8(d)
CONFIGS = {
    "cfg1": dict(kind="protein", n=100, lo=40, hi=60, seed=1, method="nw", matrix="blosum62", gaps=dict(gap_pen=4)),
    "cfg2": dict(kind="protein", n=10_000, lo=80, hi=120, seed=2, method="nw", matrix="blosum62", gaps=dict(gap_pen=4)),
    "cfg3": dict(kind="protein", n=10_000, lo=80, hi=120, seed=2, method="ga", matrix="blosum62", gaps=dict(gap_open=10, gap_extend=1)),
    "cfg4": dict(kind="dna", n=50_000, lo=120, hi=180, seed=4, method="sw", matrix="nuc44", gaps=dict(gap_open=10, gap_extend=1)),
    # not a BASELINE config: the mixed-length leg of bench.py (few equal lengths per arranged block)
    "mixed": dict(kind="protein", n=8_000, lo=20, hi=190, seed=7, method="nw", matrix="blosum62", gaps=dict(gap_pen=4)),
    "cfg5": dict(kind="protein", n=100_000, lo=96, hi=144, seed=5, method="nw", matrix="blosum62", gaps=dict(gap_pen=4), dup=0.10),
}


def make_config(name: str, n: int | None = None) -> tuple[list[bytes], dict]:
    cfg = dict(CONFIGS[name])
    n = cfg["n"] if n is None else n
    if cfg["kind"] == "protein":
        seqs = make_protein_set(n, cfg["lo"], cfg["hi"], cfg["seed"])
        if cfg.get("dup"):
            seqs = make_near_duplicates(seqs, cfg["dup"], 0.05, cfg["seed"])
    else:
        seqs = make_dna_set(n, cfg["lo"], cfg["hi"], cfg["seed"])
    return seqs, cfg
