"""Small SW parity probe with mismatch listing (development helper, run through gpurun)."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
import sequencealigner_amd as sa
from tests.oracle_binding import Oracle

o = Oracle()
rng = np.random.default_rng(5)
AA = "ARNDCQEGHILKMFPSTWYV"
for method in ("sw", "nw"):
    for lo, hi, n in ((5, 14, 40), (20, 40, 60), (90, 110, 80)):
        seqs = ["".join(rng.choice(list(AA), rng.integers(lo, hi + 1))) for _ in range(n)]
        store = sa.SequenceStore.from_sequences(seqs)
        sc = sa.Scoring.from_names(method, "blosum62", gap_open=10, gap_extend=1) if method == "sw" else sa.Scoring.from_names("nw", "blosum62", gap_pen=4)
        got = sa.hip_align(store, sc, triangular=True)
        want = o.align(store, sc, triangular=True) if hasattr(o, "align") else o.align_pairs(store, sc, np.arange(got.size))
        bad = np.nonzero(got != want)[0]
        print(method, lo, hi, n, "mismatches", bad.size, "of", got.size)
        for p in bad[:12]:
            j = int((1 + np.sqrt(1 + 8 * p)) // 2)
            while j * (j - 1) // 2 > p: j -= 1
            while (j + 1) * j // 2 <= p: j += 1
            i = int(p - j * (j - 1) // 2)
            print("   pair", i, j, "len", len(seqs[i]), len(seqs[j]), "got", got[p], "want", want[p])
