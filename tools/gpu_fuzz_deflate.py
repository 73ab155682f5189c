"""tools/gpu_fuzz_deflate.py [seed] [cases] -- random matrices through the device-side tile / DEFLATE path (csrc/sa_deflate.hip):
random N, the product's chunk rule or a random power of two, score distributions from benign to adversarial, packed and full
device matrices, level 6 (zlib streams, inflated with stock zlib) and level 0 (raw tiles); every tile of every case against the
full symmetric matrix built on the host.  Then sa_hip_tiles_begin / sa_zjob_next (the walk in shells while the alignment runs)
on random small stores against the oracle."""
import os
import sys
import time
import zlib

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sequencealigner_amd as sa  # noqa: E402
from tests.golden_util import tri_to_full  # noqa: E402
from tests.oracle_binding import Oracle  # noqa: E402
from tests.synth import make_dna_set, make_protein_set  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(seed)


def chunk_rule(n):  # cli/sa_host.c: sa_host_hdf5_chunk_dim
    c = 64
    while c < n:
        c *= 2
    if c > n:
        c //= 2
    return min(max(c, 256), 4096)


def draw(n, kind):
    m = n * (n - 1) // 2
    if kind == 0:
        return rng.integers(-200, 120, m, dtype=np.int32)
    if kind == 1:
        return rng.integers(0, 60, m, dtype=np.int32)
    if kind == 2:
        return rng.integers(-2**31, 2**31 - 1, m, dtype=np.int64).astype(np.int32)
    if kind == 3:
        return np.full(m, int(rng.integers(-5, 5)), np.int32)
    if kind == 4:
        return (rng.integers(0, 4, m, dtype=np.int32) << 8) * 1021 + rng.integers(0, 256, m, dtype=np.int32)
    if kind == 5:
        return (np.int64(1) << (np.minimum(rng.geometric(0.5, m), 40) % 31)).astype(np.int32)
    return rng.normal(-60, 25, m).astype(np.int32)


bad = tiles = 0
raw = out = 0
t0 = time.time()
for case in range(cases):
    n = int(rng.integers(257, 2600))
    chunk = chunk_rule(n) if rng.random() < 0.6 else int(2 ** rng.integers(6, 11))
    kind = int(rng.integers(0, 7))
    level = 6 if rng.random() < 0.75 else 0
    tri = draw(n, kind)
    full = tri_to_full(tri, n)
    nc = -(-n // chunk)
    pad = np.zeros((nc * chunk, nc * chunk), np.int32)
    pad[:n, :n] = full
    use_full = rng.random() < 0.3
    d = torch.from_numpy(np.ascontiguousarray(full) if use_full else tri).cuda()
    kw = dict(d_full_ptr=d.data_ptr()) if use_full else dict(d_packed_ptr=d.data_ptr())
    with sa.DeflateJob(n, chunk, level=level, **kw) as job:
        for r in range(nc):
            for c, z in enumerate(job.tile_row(r)):
                got = zlib.decompress(z) if level else z
                want = pad[r * chunk:(r + 1) * chunk, c * chunk:(c + 1) * chunk].astype("<i4").tobytes()
                tiles += 1
                raw += len(want)
                out += len(z)
                if got != want:
                    bad += 1
                    print(f"MISMATCH case {case}: n {n} chunk {chunk} kind {kind} level {level} full {use_full} tile ({r},{c})", flush=True)
print(f"matrices: {cases} cases, {tiles} tiles, {raw / 1e9:.2f} GB -> {out / 1e9:.2f} GB, {bad} mismatches, {time.time() - t0:.1f} s", flush=True)

oracle = Oracle()
bad2 = tiles2 = 0
for case in range(max(4, cases // 6)):
    n = int(rng.integers(257, 1500))
    dna = rng.random() < 0.3
    seqs = make_dna_set(n, 20, 120, 1000 + case) if dna else make_protein_set(n, 10, 110, 1000 + case)
    method = ("nw", "ga", "sw")[int(rng.integers(0, 3))]
    gaps = dict(gap_pen=int(rng.integers(1, 9))) if method == "nw" else dict(gap_open=int(rng.integers(4, 14)), gap_extend=int(rng.integers(1, 4)))
    store = sa.SequenceStore.from_sequences(seqs)
    scoring = sa.Scoring.from_names(method, "nuc44" if dna else "blosum62", **gaps)
    chunk = chunk_rule(n)
    level = 6 if rng.random() < 0.7 else 0
    full = tri_to_full(oracle.align(store, scoring, triangular=True, threads=16), n)
    nc = -(-n // chunk)
    pad = np.zeros((nc * chunk, nc * chunk), np.int32)
    pad[:n, :n] = full
    seen = set()
    with sa.DeflateJob.begin(store, scoring, chunk, level=level) as job:
        while True:
            batch = job.next()
            if not batch:
                break
            for r, c, z in batch:
                got = zlib.decompress(z) if level else z
                tiles2 += 1
                if (r, c) in seen or got != pad[r * chunk:(r + 1) * chunk, c * chunk:(c + 1) * chunk].astype("<i4").tobytes():
                    bad2 += 1
                    print(f"MISMATCH shells case {case}: n {n} {method} {gaps} level {level} tile ({r},{c})", flush=True)
                seen.add((r, c))
    if len(seen) != nc * nc:
        bad2 += 1
        print(f"MISSING TILES shells case {case}: {len(seen)} of {nc * nc}", flush=True)
print(f"shells: {max(4, cases // 6)} alignments, {tiles2} tiles against the oracle, {bad2} mismatches", flush=True)
sys.exit(1 if bad or bad2 else 0)
