"""development: SA_HIP_STAMPS=1 timeline of one share (or the whole range with world 0). usage: stamps_run.py [world] [config]"""
import sys, pathlib
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
if world == 0:
    full = torch.empty(store.pairs, dtype=torch.int32, device="cuda")
    for _ in range(3): ctx.align_range(0, store.pairs, full.data_ptr(), s)
else:
    e = ctx.share_elems(0, store.pairs, world, False)
    buf = torch.empty(e, dtype=torch.int16, device="cuda")
    for _ in range(3): ctx.align_share(0, store.pairs, world, world - 1, buf.data_ptr(), True, s)
torch.cuda.synchronize()
