"""Throughput per length regime and method (development helper, run through gpurun):
python tools/gpu_len_sweep.py [lo hi n]..."""
import sys, time, pathlib, json
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np, torch
import sequencealigner_amd as sa
from tests.oracle_binding import Oracle
from tests.synth import make_protein_set

o = Oracle()
regimes = [(int(a), int(b), int(c)) for a, b, c in zip(sys.argv[1::3], sys.argv[2::3], sys.argv[3::3])] or [(300, 500, 3000), (150, 250, 5000), (600, 1000, 1500)]
for lo, hi, n in regimes:
    seqs = make_protein_set(n, lo, hi, 11)
    store = sa.SequenceStore.from_sequences(seqs)
    for method, gaps in (("nw", dict(gap_pen=4)), ("ga", dict(gap_open=10, gap_extend=1)), ("sw", dict(gap_open=10, gap_extend=1))):
        sc = sa.Scoring.from_names(method, "blosum62", **gaps)
        with sa.Context(store, sc, 0) as ctx:
            buf = torch.empty(ctx.pairs, dtype=torch.int32, device="cuda")
            st = torch.cuda.current_stream().cuda_stream
            ctx.align_range(0, ctx.pairs, buf.data_ptr(), st); torch.cuda.synchronize()
            t = time.time()
            for _ in range(2):
                ctx.align_range(0, ctx.pairs, buf.data_ptr(), st)
            torch.cuda.synchronize()
            dt = (time.time() - t) / 2
            got = buf.cpu().numpy()
        idx = np.sort(np.random.default_rng(1).integers(0, got.size, 3000))
        ok = bool(np.array_equal(got[idx], o.align_pairs(store, sc, idx)))
        print(f"{n} x U[{lo},{hi}] {method}: {dt*1e3:8.1f} ms  {store.cells()/dt/1e12:6.2f} TCUPS  parity={ok}", flush=True)
