"""development: where one rank's share of cfg2 loses time against the ideal 1/world of the whole-range step.
usage: share_probe.py [world] [config]   (SA_HIP_CHUNK as in slice_time.py)"""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import sequencealigner_amd as sa
from tests.synth import make_config

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfgname = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
seqs, cfg = make_config(cfgname)
store = sa.SequenceStore.from_sequences(seqs)
sc = sa.Scoring.from_names(cfg["method"], cfg["matrix"], **cfg["gaps"])
ctx = sa.Context(store, sc, 0)
s = torch.cuda.current_stream().cuda_stream
full = torch.empty(store.pairs, dtype=torch.int32, device="cuda")

def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

t_full = bench(lambda: ctx.align_range(0, store.pairs, full.data_ptr(), s), 10)
print(f"whole range: {t_full*1e3:.3f} ms, ideal share {t_full/world*1e3:.3f} ms")
# fixed cost of a call: a range of a few columns (one tile per class at most)
j = store.num - 1
lo = j * (j - 1) // 2
t_small = bench(lambda: ctx.align_range(lo, j, full.data_ptr(), s), 50)
print(f"last column only ({j} pairs): {t_small*1e3:.3f} ms per call")
e = ctx.share_elems(0, store.pairs, world, True)
host = sa.PinnedMatrix(store.pairs)
buf = torch.empty(e, dtype=torch.int16, device="cuda")
for rank in (0, world - 1):
    t = bench(lambda: ctx.align_share(0, store.pairs, world, rank, buf.data_ptr(), True, s, host.ptr))
    print(f"share of rank {rank}/{world}: {t*1e3:.3f} ms = {t_full/world/t*100:.1f} % of ideal; share_elems {e} ({e*world/store.pairs:.4f} x pairs)")
# all ranks' shares back to back on one GPU = the whole job in shares
def all_shares():
    for r in range(world):
        ctx.align_share(0, store.pairs, world, r, buf.data_ptr(), True, s, host.ptr)
t = bench(all_shares, 5)
print(f"all {world} shares back to back: {t*1e3:.3f} ms ({t/world*1e3:.3f} per share)")

# the placement alone: the gathered shares of all ranks (stale data for the others: same traffic) -> packed order
shares = torch.zeros(world * e, dtype=torch.int16, device="cuda")
t = bench(lambda: ctx.place_shares(0, store.pairs, world, shares.data_ptr(), True, full.data_ptr(), s, True))
print(f"placement of {world} shares alone: {t*1e3:.3f} ms ({(2 * world * e + 4 * store.pairs) / t / 1e12:.2f} TB/s of shares read + matrix written)")
del shares
# the whole step of one rank without the fabric: my kernels, place of all shares, host copy of my piece (solo rehearsal)
from sequencealigner_amd.distributed import HipShares, TiledGatherStep
for chunks in (1, 2, 3):
    step = TiledGatherStep(HipShares(ctx, True, host), store.num, world, world - 1, chunks, None, solo=True)
    t = bench(step)
    print(f"solo step of rank {world-1}/{world}, {chunks} super-chunks (kernels + place + host copy, no all-gather): {t*1e3:.3f} ms = {t_full/world/t*100:.1f} % of ideal")
    del step
