"""development: kernel time of every rank's share (tile-interleaved), one after the other on one GPU. usage: rank_times.py [world] [config]"""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import sequencealigner_amd as sa
from tests.synth import make_config
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
seqs, cfg = make_config(sys.argv[2] if len(sys.argv) > 2 else "cfg2")
store = sa.SequenceStore.from_sequences(seqs)
sc = sa.Scoring.from_names(cfg["method"], cfg["matrix"], **cfg["gaps"])
ctx = sa.Context(store, sc, 0)
s = torch.cuda.current_stream().cuda_stream
host = sa.PinnedMatrix(store.pairs)
e = ctx.share_elems(0, store.pairs, world, True)
buf = torch.empty(e, dtype=torch.int16, device="cuda")
def bench(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
for rep in range(2):
    ts = [bench(lambda: ctx.align_share(0, store.pairs, world, r, buf.data_ptr(), True, s, host.ptr)) for r in range(world)]
    print("ms per rank:", " ".join(f"{t*1e3:.3f}" for t in ts), f" max/min {max(ts)/min(ts):.3f}")
host.close()
