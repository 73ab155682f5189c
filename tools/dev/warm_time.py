"""development: what does the first context of a process cost, by kernel family needed?  usage: warm_time.py <maxlen>"""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
t0 = time.perf_counter()
import numpy as np
import sequencealigner_amd as sa
from tests.synth import make_protein_set
t1 = time.perf_counter()
maxlen = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seqs = make_protein_set(200, 80, 120, 3) + ([make_protein_set(1, maxlen, maxlen, 4)[0]] if maxlen > 120 else [])
store = sa.SequenceStore.from_sequences(seqs)
sc = sa.Scoring.from_names(sys.argv[2] if len(sys.argv) > 2 else "nw", "blosum62", gap_pen=4) if (len(sys.argv) <= 2 or sys.argv[2] == "nw") else sa.Scoring.from_names(sys.argv[2], "blosum62", gap_open=10, gap_extend=1)
t2 = time.perf_counter()
ctx = sa.Context(store, sc, 0)
t3 = time.perf_counter()
ctx2 = sa.Context(store, sc, 0)
t4 = time.perf_counter()
print(f"maxlen {maxlen}: import {t1-t0:.3f} s, first context {t3-t2:.3f} s, second context {t4-t3:.3f} s")
