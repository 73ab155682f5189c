"""tools/dev/deflate_time.py N [chunk] -- the device-side DEFLATE encoder over the scores of N cfg5-shaped proteins:
encode and copy time per tile row, ratio, GB/s of raw matrix bytes; the first tile of every row is inflated with zlib
and compared with the expected bytes."""
import sys
import time
import zlib

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", ".."))
import sequencealigner_amd as sa  # noqa: E402
from tests.synth import make_protein_set  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
store = sa.SequenceStore.from_sequences(make_protein_set(n, 96, 144, 5))
scoring = sa.Scoring.from_names("nw", "blosum62", gap_pen=4)
d = torch.empty(store.pairs, dtype=torch.int32, device="cuda")
with sa.Context(store, scoring, 0) as ctx:
    ctx.align_range(0, store.pairs, d.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.align_range(0, store.pairs, d.data_ptr())
    torch.cuda.synchronize()
    print(f"alignment of {store.pairs} pairs: {time.perf_counter() - t0:.3f} s", flush=True)
with sa.DeflateJob(n, chunk, d_packed_ptr=d.data_ptr()) as job:
    nc = job.tiles_per_row
    t0 = time.perf_counter()
    total = 0
    first = []
    for r in range(nc):
        streams = job.tile_row(r)
        total += sum(len(z) for z in streams)
        first.append(streams[0])
    wall = time.perf_counter() - t0
    st = job.stats()
    raw = st["raw_bytes"]
    print(f"{nc} x {nc} tiles of {chunk}: raw {raw / 1e9:.2f} GB -> {total / 1e9:.2f} GB ({raw / total:.3f} : 1), wall {wall:.3f} s "
          f"(encode wait {st['encode_ms']:.1f} ms, gather + copy {st['copy_ms']:.1f} ms) = {raw / wall / 1e9:.1f} GB/s of matrix", flush=True)
# check the first tile of a few rows
tri = d.cpu().numpy()
for r in sorted(set([0, nc // 2, nc - 1])):
    got = np.frombuffer(zlib.decompress(first[r]), "<i4").reshape(chunk, chunk)
    i = np.arange(r * chunk, (r + 1) * chunk)[:, None]
    j = np.arange(0, chunk)[None, :]
    hi, lo = np.maximum(i, j).astype(np.int64), np.minimum(i, j).astype(np.int64)
    ok = (i < n) & (j < n) & (i != j)
    idx = np.where(ok, hi * (hi - 1) // 2 + lo, 0)
    want = np.where(ok, tri[idx], 0)
    assert np.array_equal(got, want), r
print("tiles checked")
