import sys, time, pathlib, os
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
import sequencealigner_amd as sa
from tests.synth import make_config
seqs, cfg = make_config("cfg2")
store = sa.SequenceStore.from_sequences(seqs)
sc = sa.Scoring.from_names(cfg["method"], cfg["matrix"], **cfg["gaps"])
sa.hip_align(store.prefix(500), sc, triangular=True)
for pin in ("", "1"):
    if pin: os.environ["SA_HIP_NO_PIN"] = "1"
    for tri in (True, False):
        ts = []
        for _ in range(3):
            t = time.time(); m = sa.hip_align(store, sc, triangular=tri); ts.append(time.time() - t)
        print(f"NO_PIN={pin or 0} triangular={tri}: {min(ts)*1e3:.0f} ms (best of 3)", flush=True)
