"""development: create / use / destroy many contexts in one process (resource leak hunt)"""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import sequencealigner_amd as sa
from tests.synth import make_protein_set
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
store = sa.SequenceStore.from_sequences(make_protein_set(300, 20, 230, 3))  # two bundles -> side streams are used
sc = sa.Scoring.from_names("nw", "blosum62", gap_pen=4)
out = torch.empty(store.pairs, dtype=torch.int32, device="cuda")
for k in range(n):
    with sa.Context(store, sc, 0) as ctx:
        ctx.align_range(0, store.pairs, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    if k % 100 == 0:
        print("contexts", k, flush=True)
print("ok")
