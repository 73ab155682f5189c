"""tools/dev/deflate_pmc.py [N] -- the encoder's kernels alone, for a counter pass: score-like random values straight into a
device matrix (no alignment kernels under the profiler), one walk over the tile rows."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import sequencealigner_amd as sa  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
g = torch.Generator(device="cuda").manual_seed(7)
pairs = n * (n - 1) // 2
d = (torch.randn(pairs, device="cuda", generator=g) * 25 - 60).to(torch.int32)
torch.cuda.synchronize()
with sa.DeflateJob(n, 4096, d_packed_ptr=d.data_ptr()) as job:
    total = 0
    for r in range(job.tiles_per_row):
        total += sum(len(z) for z in job.tile_row(r))
    st = job.stats()
print(f"{job.tiles_per_row} rows of {job.tiles_per_row} tiles: {st['raw_bytes'] / 1e6:.1f} MB -> {total / 1e6:.1f} MB", flush=True)
