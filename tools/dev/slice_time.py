"""development: kernel time of ONE rank's share of a strong-scaling run, measured on one GPU.
  tiled (default): sa_ctx_align_share of the whole job / `chunks` geometric super-chunks, rank's tiles only
  range:           the contiguous-range partition (ChunkedGather), for A/B
SA_HIP_CHUNK overrides the planner's stream length.  usage: slice_time.py [world] [chunks] [config] [tiled|range]"""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import sequencealigner_amd as sa
from sequencealigner_amd.distributed import ChunkedGather, column_chunks
from tests.synth import make_config

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfgname = sys.argv[3] if len(sys.argv) > 3 else "cfg2"
mode = sys.argv[4] if len(sys.argv) > 4 else "tiled"
seqs, cfg = make_config(cfgname)
store = sa.SequenceStore.from_sequences(seqs)
sc = sa.Scoring.from_names(cfg["method"], cfg["matrix"], **cfg["gaps"])
ctx = sa.Context(store, sc, 0)
s = torch.cuda.current_stream().cuda_stream
full = torch.empty(store.pairs, dtype=torch.int32, device="cuda")
def whole():
    ctx.align_range(0, store.pairs, full.data_ptr(), s)
for _ in range(3): whole()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): whole()
torch.cuda.synchronize(); t_full = (time.perf_counter() - t0) / 10
print(f"{cfgname} whole range on one GPU: {t_full*1e3:.3f} ms  -> ideal share at world {world}: {t_full/world*1e3:.3f} ms")
ranges = column_chunks(store.num, chunks)
host = sa.PinnedMatrix(store.pairs)
for rank in (0, world // 2, world - 1, 0):  # (rank 0 twice: the first measurement after the idle set-up runs on ramping clocks)
    if mode == "range":
        sched = ChunkedGather(store.pairs, world, rank, chunks)
        buf = torch.empty(store.pairs // world + 16, dtype=torch.int32, device="cuda")
        def step():
            for c in range(chunks):
                lo, hi = sched.slice_range(c)
                ctx.align_range(lo, hi - lo, buf.data_ptr(), s)
    else:
        bufs = [torch.empty(ctx.share_elems(lo, cnt, world, True), dtype=torch.int16, device="cuda") for lo, cnt in ranges]
        def step():
            for (lo, cnt), b in zip(ranges, bufs):
                ctx.align_share(lo, cnt, world, rank, b.data_ptr(), True, s, host.ptr)
    for _ in range(10): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20
    print(f"{cfgname} {mode} world {world} chunks {chunks} rank {rank}: {t*1e3:.3f} ms per step-share = {t_full/world/t*100:.1f} % of the ideal share")
