"""Throughput of the three methods on the BASELINE.json shapes (development helper, run through gpurun)."""
import sys, time, pathlib, json
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np, torch
import sequencealigner_amd as sa
from tests.oracle_binding import Oracle
from tests.synth import make_config

o = Oracle()
out = {}
for cfg_name, n in (("cfg2", 10000), ("cfg3", 10000), ("cfg4", 12000)):
    seqs, cfg = make_config(cfg_name, n)
    store = sa.SequenceStore.from_sequences(seqs)
    sc = sa.Scoring.from_names(cfg["method"], cfg["matrix"], **cfg["gaps"])
    with sa.Context(store, sc, 0) as ctx:
        buf = torch.empty(ctx.pairs, dtype=torch.int32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        ctx.align_range(0, ctx.pairs, buf.data_ptr(), st); torch.cuda.synchronize()
        ctx.timing(True)
        t = time.time()
        for _ in range(3):
            ctx.align_range(0, ctx.pairs, buf.data_ptr(), st)
        torch.cuda.synchronize()
        dt = (time.time() - t) / 3
        tm = ctx.timing_read(); ctx.timing(False)
        got = buf.cpu().numpy()
    idx = np.sort(np.random.default_rng(1).integers(0, got.size, 50000))
    ok = bool(np.array_equal(got[idx], o.align_pairs(store, sc, idx)))
    out[cfg_name] = dict(n=n, method=cfg["method"], pairs_per_s=got.size / dt, gcups=store.cells() / dt / 1e9, ms=dt * 1e3,
                         dominant=tm["kernel"], parity=ok)
    print(cfg_name, json.dumps(out[cfg_name]), flush=True)
