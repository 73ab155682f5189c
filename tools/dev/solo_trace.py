"""development: a few solo steps of one rank (TiledGatherStep, no fabric) for a rocprofv3 timeline.
usage: solo_trace.py [world] [chunks] [config]"""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import sequencealigner_amd as sa
from sequencealigner_amd.distributed import HipShares, TiledGatherStep
from tests.synth import make_config

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 3
seqs, cfg = make_config(sys.argv[3] if len(sys.argv) > 3 else "cfg2")
store = sa.SequenceStore.from_sequences(seqs)
sc = sa.Scoring.from_names(cfg["method"], cfg["matrix"], **cfg["gaps"])
ctx = sa.Context(store, sc, 0)
host = sa.PinnedMatrix(store.pairs)
step = TiledGatherStep(HipShares(ctx, True, host), store.num, world, world - 1, chunks, None, solo=True)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(6): step()
torch.cuda.synchronize()
print(f"solo step world {world} chunks {chunks}: {(time.perf_counter()-t0)/6*1e3:.3f} ms")
