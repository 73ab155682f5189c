"""development: where do the packed kernels differ from the oracle?  usage: pk_debug.py method n lo hi [seed] [kind]"""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
import sequencealigner_amd as sa
from tests.oracle_binding import Oracle
from tests.synth import make_protein_set, make_dna_set

method, n, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 3
kind = sys.argv[6] if len(sys.argv) > 6 else "protein"
gaps = dict(gap_pen=4) if method == "nw" else dict(gap_open=10, gap_extend=1)
seqs = make_protein_set(n, lo, hi, seed) if kind == "protein" else make_dna_set(n, lo, hi, seed, iupac=True)
store = sa.SequenceStore.from_sequences(seqs)
sc = sa.Scoring.from_names(method, "blosum62" if kind == "protein" else "nuc44", **gaps)
got = sa.hip_align(store, sc, triangular=True)
want = Oracle().align(store, sc, triangular=True, threads=16)
bad = np.nonzero(got != want)[0]
print(f"{method} {kind} n={n} len {lo}..{hi}: {bad.size} of {got.size} differ")
lens = store.meta[:, 1]
for p in bad[:30]:
    j = int((1 + np.sqrt(1 + 8 * p)) / 2)
    while j * (j - 1) // 2 > p: j -= 1
    while (j + 1) * j // 2 <= p: j += 1
    i = p - j * (j - 1) // 2
    print(f"  pair ({i},{j}) len_i={lens[i]} len_j={lens[j]} K={-(-lens[j]//8)} pad={8*(-(-lens[j]//8))-lens[j]}: got {got[p]} want {want[p]} diff {got[p]-want[p]}")
if bad.size:
    js = sorted({int((1 + np.sqrt(1 + 8 * p)) / 2) for p in bad})
    print("  columns with errors:", js[:50])
