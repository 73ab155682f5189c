"""development: one rank's share time against the ideal 1/world for several world sizes (fixed overhead vs proportional)"""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import sequencealigner_amd as sa
from tests.synth import make_config

cfgname = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
seqs, cfg = make_config(cfgname)
store = sa.SequenceStore.from_sequences(seqs)
sc = sa.Scoring.from_names(cfg["method"], cfg["matrix"], **cfg["gaps"])
ctx = sa.Context(store, sc, 0)
s = torch.cuda.current_stream().cuda_stream
full = torch.empty(store.pairs, dtype=torch.int32, device="cuda")
host = sa.PinnedMatrix(store.pairs)

def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

t_full = bench(lambda: ctx.align_range(0, store.pairs, full.data_ptr(), s), 10)
print(f"{cfgname} whole range: {t_full*1e3:.3f} ms")
for to_host in (False, True):
    for world in (1, 2, 4, 8, 16, 32):
        e = ctx.share_elems(0, store.pairs, world, to_host)
        buf = torch.empty(e, dtype=torch.int16, device="cuda")
        r = world - 1
        t = bench(lambda: ctx.align_share(0, store.pairs, world, r, buf.data_ptr(), True, s, host.ptr if to_host else 0))
        print(f"to_host {int(to_host)} world {world:2d}: share {t*1e3:7.3f} ms, ideal {t_full/world*1e3:7.3f}, over {1e3*(t - t_full/world):6.3f} ms = {t_full/world/t*100:5.1f} %")
host.close()
