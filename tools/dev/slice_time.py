"""development: kernel time of ONE rank's share of a strong-scaling run (cfg2 / world ranks / chunks super-chunks);
SA_HIP_CHUNK overrides the planner's stream length.  usage: slice_time.py [world] [chunks] [config]"""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import sequencealigner_amd as sa
from sequencealigner_amd.distributed import ChunkedGather
from tests.synth import make_config

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfgname = sys.argv[3] if len(sys.argv) > 3 else "cfg2"
seqs, cfg = make_config(cfgname)
store = sa.SequenceStore.from_sequences(seqs)
sc = sa.Scoring.from_names(cfg["method"], cfg["matrix"], **cfg["gaps"])
ctx = sa.Context(store, sc, 0)
buf = torch.empty(store.pairs // world + 16, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for rank in (0, world // 2, world - 1):
    sched = ChunkedGather(store.pairs, world, rank, chunks)
    def step():
        for c in range(chunks):
            lo, hi = sched.slice_range(c)
            ctx.align_range(lo, hi - lo, buf.data_ptr(), s)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20
    print(f"{cfgname} world {world} chunks {chunks} rank {rank}: {t*1e3:.3f} ms per step-share  ({store.pairs / world / t / 1e9:.3f} G pairs/s per rank, ideal share of full-range rate = x{world})")
