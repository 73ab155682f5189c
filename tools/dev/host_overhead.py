"""development: where does a host-delivered step spend its time?  wall per sa_ctx_align_host call, its inner phase, and
the device-resident step, with and without per-launch timing events"""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import sequencealigner_amd as sa
from tests.synth import make_config

seqs, cfg = make_config(sys.argv[1] if len(sys.argv) > 1 else "cfg2")
store = sa.SequenceStore.from_sequences(seqs)
sc = sa.Scoring.from_names(cfg["method"], cfg["matrix"], **cfg["gaps"])
ctx = sa.Context(store, sc, 0)
dest = sa.PinnedMatrix(store.pairs)
packed = torch.empty(store.pairs, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for timing in (False, True):
    ctx.timing(timing)
    for _ in range(3): ctx.align_host(dest.array, triangular=True)
    torch.cuda.synchronize(); t0 = time.perf_counter(); ph = 0.0
    for _ in range(20): ph += ctx.align_host(dest.array, triangular=True)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 20
    for _ in range(3): ctx.align_range(0, store.pairs, packed.data_ptr(), s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ctx.align_range(0, store.pairs, packed.data_ptr(), s)
    torch.cuda.synchronize(); res = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.align_range(0, store.pairs, packed.data_ptr(), s); torch.cuda.synchronize()
    res_sync = (time.perf_counter() - t0) / 20
    if timing: ctx.timing_read()
    print(f"timing events {timing}: host-delivered wall {wall*1e3:.3f} ms (inner phase {ph/20*1e3:.3f}), resident back-to-back {res*1e3:.3f}, resident with a sync per step {res_sync*1e3:.3f}")
