"""Sharding of the packed pair index across ranks (one process per GPU) -- SURVEY.md §8(e).

The units (pairs) are independent, the inputs (<= a few MB) are replicated on every GPU, so each rank
scores one contiguous range of the packed index -- the reference's own batch abstraction
`kernel(scores, start, batch)` (src/bio/align.h:48) -- and a single all-gather (RCCL over xGMI when the
process group is "nccl") assembles the packed vector on every rank.  Equal-count ranges are padded to a
common length so the gather lands in place in one buffer of world*per elements.

`ChunkedGather` is the index arithmetic (no torch); `GatherStep` is the overlapped multi-stream step that
bench.py times and tests/test_gpu_gather_step.py checks on the GPU."""
from __future__ import annotations


def rank_range(pairs: int, world: int, rank: int) -> tuple[int, int, int]:
    """-> (per, lo, hi): common padded slice length and this rank's packed range [lo, hi)."""
    per = (pairs + world - 1) // world
    lo = min(pairs, rank * per)
    hi = min(pairs, lo + per)
    return per, lo, hi


def gather_packed(dist, packed, rank: int, per: int):
    """In-place all-gather of every rank's slice packed[r*per:(r+1)*per] (torch tensors, any backend)."""
    mine = packed[rank * per:(rank + 1) * per]
    try:
        dist.all_gather_into_tensor(packed, mine)
    except (RuntimeError, NotImplementedError):  # backends without the flat variant
        world = packed.numel() // per
        dist.all_gather([packed[r * per:(r + 1) * per] for r in range(world)], mine.clone())
    return packed


class ChunkedGather:
    """Strong-scaling schedule that overlaps the RCCL all-gather with the kernels.

    The packed index is cut into `chunks` super-chunks and every super-chunk into `world` equal slices
    (chunk-major, rank-minor), so super-chunk c of the result is exactly the concatenation over ranks of
    what they computed for it: `all_gather_into_tensor(buffer[c], my_slice)` lands in place in packed
    order.  Rank r scores slice (c, r) on the compute stream; the gather of super-chunk c is issued on a
    side stream as soon as its slice is done, while the kernels of super-chunk c+1 run."""

    def __init__(self, pairs: int, world: int, rank: int, chunks: int):
        self.pairs, self.world, self.rank, self.chunks = pairs, world, rank, max(1, chunks)
        self.sub = (pairs + world * self.chunks - 1) // (world * self.chunks)
        self.total = self.sub * world * self.chunks

    def slice_range(self, c: int, r: int | None = None) -> tuple[int, int]:
        """packed range [lo, hi) of slice (c, r); empty past the end of the pair space"""
        r = self.rank if r is None else r
        lo = min(self.pairs, (c * self.world + r) * self.sub)
        return lo, min(self.pairs, lo + self.sub)

    def super_chunk(self, buffer, c: int):
        return buffer[c * self.world * self.sub:(c + 1) * self.world * self.sub]

    def my_slice(self, buffer, c: int):
        o = (c * self.world + self.rank) * self.sub
        return buffer[o:o + self.sub]


class GatherStep:
    """One whole-job step on `world` ranks: score my slices, all-gather them (RCCL), deliver to the host.

    Per super-chunk c (streams of this rank; nothing below synchronises the host):
      compute[c]  kernels of slice (c, rank)                       -> my_slice(c)   (int16 or s32)
      comm        all_gather_into_tensor(super_chunk(c), my_slice) -> every GPU holds super-chunk c
      deliver     int16 exchange only: widen super-chunk c to the reference's s32;
                  device->host copy of THIS rank's 1/world share of the finished s32 super-chunk into pinned
                  host memory (every rank drives its own PCIe link; on one node the shares make up the matrix)
    The step returns with the main stream waiting for everything, so `torch.cuda.synchronize()` (or an event on
    the main stream) marks: packed s32 vector complete on every GPU AND this rank's share on the host.

    `dist=None` runs the same schedule on one rank without a process group (slice = whole super-chunk)."""

    def __init__(self, ctx, pairs: int, world: int, rank: int, chunks: int, dist=None, use16: bool = False, to_host: bool = True):
        import torch

        self.torch, self.ctx, self.dist = torch, ctx, dist
        self.sched = ChunkedGather(pairs, world, rank, chunks)
        self.use16 = bool(use16)
        self.to_host = bool(to_host)
        n = self.sched.total
        self.packed = torch.zeros(n, dtype=torch.int32, device="cuda")
        self.packed16 = torch.zeros(n, dtype=torch.int16, device="cuda") if self.use16 else None
        self.main = torch.cuda.current_stream()
        self.compute = [torch.cuda.Stream() for _ in range(self.sched.chunks)]
        self.comm = torch.cuda.Stream()
        self.deliver = torch.cuda.Stream()
        # host share of super-chunk c: elements [c*W*sub + rank*share, +share) of the packed vector, share = sub
        # (the slice this rank computed is also the slice it delivers: W*sub/W = sub elements per super-chunk)
        self.host = torch.zeros(self.sched.sub * self.sched.chunks, dtype=torch.int32).pin_memory() if self.to_host else None

    def host_ranges(self):
        """[(packed_lo, packed_hi, host_offset)] of what this rank delivers (clipped to the pair space)"""
        out = []
        for c in range(self.sched.chunks):
            lo, hi = self.sched.slice_range(c)
            out.append((lo, hi, c * self.sched.sub))
        return out

    def __call__(self):
        torch, sched, dist = self.torch, self.sched, self.dist
        start = torch.cuda.Event()
        start.record(self.main)
        buf = self.packed16 if self.use16 else self.packed
        for c in range(sched.chunks):
            lo, hi = sched.slice_range(c)
            cs = self.compute[c]
            cs.wait_event(start)  # ordered after the previous step
            mine = sched.my_slice(buf, c)
            if self.use16:
                self.ctx.align_range16(lo, hi - lo, mine.data_ptr(), cs.cuda_stream)
            else:
                self.ctx.align_range(lo, hi - lo, mine.data_ptr(), cs.cuda_stream)
            done = torch.cuda.Event()
            done.record(cs)
            gathered = done
            if dist is not None:
                with torch.cuda.stream(self.comm):
                    self.comm.wait_event(done)
                    # (the process group moves bytes; int16 is not among its dtypes, uint8 is)
                    dist.all_gather_into_tensor(sched.super_chunk(buf, c).view(torch.uint8), mine.view(torch.uint8))
                    gathered = torch.cuda.Event()
                    gathered.record(self.comm)
            self.deliver.wait_event(gathered)
            if self.use16:
                o = c * sched.world * sched.sub
                self.ctx.widen16(self.packed16.data_ptr() + 2 * o, self.packed.data_ptr() + 4 * o,
                                 sched.world * sched.sub, self.deliver.cuda_stream)
            if self.to_host:
                with torch.cuda.stream(self.deliver):
                    self.host[c * sched.sub:(c + 1) * sched.sub].copy_(sched.my_slice(self.packed, c), non_blocking=True)
        fin = torch.cuda.Event()
        fin.record(self.deliver)
        self.main.wait_event(fin)
