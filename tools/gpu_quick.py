"""Quick GPU sanity + throughput probe (development helper, run through gpurun)."""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import sequencealigner_amd as sa
from tests.oracle_binding import Oracle
from tests.synth import make_protein_set

print(sa.device_name(0), flush=True)
o = Oracle()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seqs = make_protein_set(n, 80, 120, 2)
store = sa.SequenceStore.from_sequences(seqs)
for method, gaps in (("nw", dict(gap_pen=4)), ("ga", dict(gap_open=10, gap_extend=1)), ("sw", dict(gap_open=10, gap_extend=1))):
    sc = sa.Scoring.from_names(method, "blosum62", **gaps)
    with sa.Context(store, sc, 0) as ctx:
        out = torch.empty(ctx.pairs, dtype=torch.int32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        ctx.align_range(0, ctx.pairs, out.data_ptr(), st); torch.cuda.synchronize()
        t = time.time()
        ctx.align_range(0, ctx.pairs, out.data_ptr(), st); torch.cuda.synchronize()
        dt = time.time() - t
        cells = ctx.cells()
        got = out.cpu().numpy()
    idx = np.sort(np.random.default_rng(1).integers(0, got.size, 20000))
    ok = np.array_equal(got[idx], o.align_pairs(store, sc, idx))
    print(f"{method}: n={n} pairs={got.size} {dt*1e3:.1f} ms  {got.size/dt:.3e} pairs/s  {cells/dt/1e9:.1f} GCUPS  parity={ok}", flush=True)
