import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import sequencealigner_amd as sa
from tests.synth import make_config
seqs, cfg = make_config("cfg5")
store = sa.SequenceStore.from_sequences(seqs)
scoring = sa.Scoring.from_names(cfg["method"], cfg["matrix"], **cfg["gaps"])
with sa.Context(store, scoring, 0) as ctx:
    out = torch.empty(ctx.pairs, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    step = 1 << 30
    for a in range(0, ctx.pairs, step):
        print("range", a, flush=True)
        ctx.align_range(a, min(step, ctx.pairs - a), out.data_ptr() + 4 * a, stream)
        torch.cuda.synchronize()
        print("done", a, flush=True)
print("ok")
