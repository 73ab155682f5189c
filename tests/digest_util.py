"""Per-column digests of a packed score vector (pair i<j at j(j-1)/2+i): for every column j the triple
(sum as int64, xor as int32, crc32 of the column's little-endian s32 bytes).  tools/make_digests.py computes them
from the REFERENCE's output for the full-size bench workloads (fixtures tests/golden/digest_*.npz, ~160 KB each);
tests/test_gpu_digests.py computes them from the host-delivered matrix of the benchmarked path and compares every
column -- a bit-exact pin of every pair that needs neither sampling nor the reference on the GPU box."""
from __future__ import annotations

import zlib

import numpy as np


def column_digests(packed: np.ndarray, n: int) -> dict:
    packed = np.ascontiguousarray(packed, dtype="<i4")
    assert packed.size == n * (n - 1) // 2
    sums = np.zeros(n, np.int64)
    xors = np.zeros(n, np.int32)
    crcs = np.zeros(n, np.uint32)
    tri = 0
    for j in range(1, n):
        col = packed[tri:tri + j]
        sums[j] = int(col.sum(dtype=np.int64))
        xors[j] = np.bitwise_xor.reduce(col)
        crcs[j] = zlib.crc32(col.tobytes())
        tri += j
    return dict(sum=sums, xor=xors, crc32=crcs)
