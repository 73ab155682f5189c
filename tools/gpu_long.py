"""Long-protein regime (development helper): 1200 x U[1000,3000] aa, fast path (strip-mined) vs generic kernels."""
import sys, time, pathlib, os, subprocess
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np, torch
import sequencealigner_amd as sa
from tests.oracle_binding import Oracle
from tests.synth import make_protein_set
o = Oracle()
store = sa.SequenceStore.from_sequences(make_protein_set(1200, 1000, 3000, 78))
for method, gaps in (("nw", dict(gap_pen=4)), ("ga", dict(gap_open=10, gap_extend=1)), ("sw", dict(gap_open=10, gap_extend=1))):
    sc = sa.Scoring.from_names(method, "blosum62", **gaps)
    with sa.Context(store, sc, 0) as ctx:
        buf = torch.empty(ctx.pairs, dtype=torch.int32, device="cuda"); st = torch.cuda.current_stream().cuda_stream
        ctx.align_range(0, ctx.pairs, buf.data_ptr(), st); torch.cuda.synchronize()
        t = time.time(); ctx.align_range(0, ctx.pairs, buf.data_ptr(), st); torch.cuda.synchronize(); dt = time.time() - t
        got = buf.cpu().numpy()
    idx = np.sort(np.random.default_rng(1).integers(0, got.size, 3000))
    print(f"{method}: {got.size/dt:.3e} pairs/s {store.cells()/dt/1e9:.0f} GCUPS parity={np.array_equal(got[idx], o.align_pairs(store, sc, idx))} generic={os.environ.get('SA_HIP_FORCE_GENERIC', '0')}", flush=True)
