"""Timings of the non-alignment device pieces (development helper): filter on cfg5, expand_full on cfg2."""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np, torch
import sequencealigner_amd as sa
from tests.synth import make_config

seqs, cfg = make_config("cfg5")
store = sa.SequenceStore.from_sequences(seqs)
t = time.time(); keep = sa.hip_filter(store, 0.9); dt = time.time() - t
print(f"filter cfg5: {store.num} seqs -> {int(keep.sum())} kept in {dt:.2f} s ({store.pairs/dt:.3e} pair-compares/s)", flush=True)

seqs, cfg = make_config("cfg2")
store = sa.SequenceStore.from_sequences(seqs)
sc = sa.Scoring.from_names(cfg["method"], cfg["matrix"], **cfg["gaps"])
with sa.Context(store, sc, 0) as ctx:
    packed = torch.empty(ctx.pairs, dtype=torch.int32, device="cuda")
    full = torch.empty((store.num, store.num), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    ctx.align_range(0, ctx.pairs, packed.data_ptr(), st)
    ctx.expand_full(packed.data_ptr(), full.data_ptr(), st); torch.cuda.synchronize()
    t = time.time()
    for _ in range(5):
        ctx.expand_full(packed.data_ptr(), full.data_ptr(), st)
    torch.cuda.synchronize(); dt = (time.time() - t) / 5
    print(f"expand_full cfg2: {dt*1e3:.2f} ms  ({(packed.numel()*4 + full.numel()*4)/dt/1e9:.0f} GB/s of algorithmic traffic)", flush=True)
for tri in (True, False):
    t = time.time(); m = sa.hip_align(store, sc, triangular=tri); dt = time.time() - t
    print(f"hip_align cfg2 triangular={tri}: {dt*1e3:.0f} ms", flush=True)
