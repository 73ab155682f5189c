"""CPU, world_size 2 over gloo: the N>1 path of bench.py (range sharding + in-place all-gather of the
packed slices) reassembles exactly the single-rank result.  The per-rank "device" here is the CPU
oracle -- this test covers the sharding/collective logic, not the kernels."""
import os
import pathlib
import socket
import subprocess
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[1]

WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
import sequencealigner_amd as sa
from sequencealigner_amd.distributed import ChunkedGather, gather_packed, rank_range
from tests.oracle_binding import Oracle
from tests.synth import make_protein_set

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
store = sa.SequenceStore.from_sequences(make_protein_set(90, 10, 80, 17))
scoring = sa.Scoring.from_names("ga", "blosum62", gap_open=10, gap_extend=1)
o = Oracle()
per, lo, hi = rank_range(store.pairs, world, rank)
packed = torch.zeros(per * world, dtype=torch.int32)
packed[rank * per: rank * per + (hi - lo)] = torch.from_numpy(o.align_range(store, scoring, lo, hi - lo, threads=2))
gather_packed(dist, packed, rank, per)
full = o.align(store, scoring, triangular=True, threads=2)
assert np.array_equal(packed[:store.pairs].numpy(), full), f"rank {rank}: gathered vector differs"
# bench.py's overlapped schedule: chunk-major / rank-minor slices, one in-place all-gather per super-chunk
sched = ChunkedGather(store.pairs, world, rank, 3)
buf = torch.zeros(sched.total, dtype=torch.int32)
works = []
for c in range(sched.chunks):
    lo, hi = sched.slice_range(c)
    if hi > lo:
        sched.my_slice(buf, c)[:hi - lo] = torch.from_numpy(o.align_range(store, scoring, lo, hi - lo, threads=2))
    works.append(dist.all_gather_into_tensor(sched.super_chunk(buf, c), sched.my_slice(buf, c).clone(), async_op=True))
for w in works:
    w.wait()
assert np.array_equal(buf[:store.pairs].numpy(), full), f"rank {rank}: chunked gather differs"
# the int16 exchange format of bench.py: slices gathered as bytes (uint8 views), then widened
buf16 = torch.zeros(sched.total, dtype=torch.int16)
works = []
for c in range(sched.chunks):
    lo, hi = sched.slice_range(c)
    if hi > lo:
        sched.my_slice(buf16, c)[:hi - lo] = torch.from_numpy(o.align_range(store, scoring, lo, hi - lo, threads=2).astype(np.int16))
    works.append(dist.all_gather_into_tensor(sched.super_chunk(buf16, c).view(torch.uint8),
                                             sched.my_slice(buf16, c).clone().view(torch.uint8), async_op=True))
for w in works:
    w.wait()
assert np.array_equal(buf16[:store.pairs].to(torch.int32).numpy(), full), f"rank {rank}: int16 exchange differs"
# bench.py's default schedule at N > 1: TiledGatherStep -- tiles dealt over the ranks, dense shares, one all-gather per
# geometric super-chunk, place into packed order, per-rank host pieces.  The "device" is the oracle behind the same
# backend interface HipShares implements with sa_ctx_share_elems / sa_ctx_align_share / sa_ctx_place_shares.
from sequencealigner_amd.distributed import TiledGatherStep, tri

class OracleShares:
    device = "cpu"
    def __init__(self, dtype, host):
        self.dtype, self.host = dtype, host  # host: the packed host matrix, ONE shared mapping both ranks attach
    def _tiles(self, start, count, world):
        # -> [(owner, share_offset, packed_lo, n)]: (column, block of 16 rows) tiles of the range, dealt round-robin
        tiles, fill, k = [], [0] * world, 0
        j = 1
        while tri(j + 1) <= start:
            j += 1
        p = start
        while p < start + count:
            hi = min(start + count, tri(j + 1))
            while p < hi:
                n = min(16, hi - p)
                r = k % world
                tiles.append((r, fill[r], p, n))
                fill[r] += n
                k += 1
                p += n
            j += 1
        return tiles, max(fill)
    def share_elems(self, start, count, world):
        return self._tiles(start, count, world)[1]
    def align_share(self, start, count, world, r, share, stream, leave_room=False):
        for owner, off, p, n in self._tiles(start, count, world)[0]:
            if owner == r:
                sc = o.align_range(store, scoring, p, n, threads=1)
                share[off:off + n] = torch.from_numpy(sc).to(self.dtype)
                self.host[p:p + n] = sc  # what sa_ctx_align_share does with host_packed: my scores, packed order
    def place(self, start, count, world, shares, packed_range, stream):
        tiles, e = self._tiles(start, count, world)
        for owner, off, p, n in tiles:
            packed_range[p - start:p - start + n] = shares[owner * e + off:owner * e + off + n].to(torch.int32)

shm = sys.argv[2]
for chunks, dtype in ((1, torch.int32), (3, torch.int16)):
    if rank == 0:
        np.full(store.pairs, -(2 ** 31), np.int32).tofile(shm)
    dist.barrier()
    host = np.memmap(shm, dtype=np.int32, mode="r+", shape=(store.pairs,))
    step = TiledGatherStep(OracleShares(dtype, host), store.num, world, rank, chunks, dist)
    step()
    step()
    assert np.array_equal(step.packed.numpy(), full), f"rank {rank}: tiled step, {chunks} chunks: placed vector differs"
    host.flush()
    dist.barrier()
    # the ranks' direct stores fill the ONE host matrix exactly: every element written, every element right
    assert np.array_equal(np.fromfile(shm, dtype=np.int32), full), f"rank {rank}: the shared host matrix is not the packed matrix"
    dist.barrier()
    del host
# work-balanced cut points (the general driver's rule) cover the index exactly once
b = store.partition(world)
assert b[0] == 0 and b[-1] == store.pairs
dist.barrier()
if rank == 0:
    print("MULTIRANK_OK", world, per)
dist.destroy_process_group()
"""


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_gloo_sharding(tmp_path, oracle, sa):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script), str(ROOT), str(tmp_path / "host_matrix.bin")]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "MULTIRANK_OK 2" in res.stdout


def test_chunked_schedule_covers_index_once():
    from sequencealigner_amd.distributed import ChunkedGather
    for pairs in (1, 7, 4950, 49_995_000):
        for world in (1, 2, 8):
            for chunks in (1, 3, 4):
                seen = []
                for c in range(chunks):
                    for r in range(world):
                        lo, hi = ChunkedGather(pairs, world, r, chunks).slice_range(c)
                        seen.append((lo, hi))
                pos = 0
                for lo, hi in seen:  # chunk-major, rank-minor order == packed order
                    assert lo == min(pairs, pos) and hi >= lo
                    pos = hi if hi > lo else pos
                assert pos == pairs


def test_column_chunks_are_column_aligned_geometric_and_cover_the_index():
    from sequencealigner_amd.distributed import column_chunks, tri
    for n in (2, 3, 17, 1100, 10_000, 100_000):
        for chunks in (1, 2, 3, 4, 7):
            r = column_chunks(n, chunks)
            assert 1 <= len(r) <= chunks
            pos = 0
            for lo, cnt in r:
                assert lo == pos and cnt > 0
                j = int((1 + (1 + 8 * lo) ** 0.5) / 2)
                assert any(tri(x) == lo for x in (j - 1, j, j + 1)), "super-chunks start at column starts"
                pos += cnt
            assert pos == tri(n)
            if n >= 1000 and 1 < len(r) <= 4:
                assert all(a[1] > 2 * b[1] for a, b in zip(r, r[1:])), "each super-chunk is much smaller than the one before"


def test_rank_ranges_cover_index_once():
    from sequencealigner_amd.distributed import rank_range
    for pairs in (1, 7, 4950, 49_995_000):
        for world in (1, 2, 3, 4, 8):
            seen = 0
            for r in range(world):
                per, lo, hi = rank_range(pairs, world, r)
                assert lo == min(pairs, r * per) and lo <= hi <= pairs and hi - lo <= per
                seen += hi - lo
            assert seen == pairs and per * world >= pairs
