import sys, time; sys.path.insert(0, "/root/repo")
import sequencealigner_amd as sa
from tests.synth import make_config
seqs, cfg = make_config("cfg5")
store = sa.SequenceStore.from_sequences(seqs)
for _ in range(3):
    t = time.perf_counter(); keep = sa.hip_filter(store, 0.9); print("filter", round(time.perf_counter() - t, 3), "s kept", int(keep.sum()), flush=True)
