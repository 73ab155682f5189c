"""Sharding of the pair space across ranks (one process per GPU) -- SURVEY.md §8(e).

The units (pairs) are independent and the inputs (<= a few MB) are replicated on every GPU; only the output is
sharded, and one all-gather per super-chunk (RCCL over xGMI when the process group is "nccl") assembles it.

`TiledGatherStep` (what bench.py times at N > 1) -- north_star's "pair space tiled across the GPUs": the launch plan
of a super-chunk (its largest-first list of workgroup-tiles) is the same on every rank, the tiles are dealt over the
ranks by DP work (sa_ctx_align_share), every rank stores its tiles densely in tile order, the all-gather moves those
dense shares, and sa_ctx_place_shares widens and places them into the reference's packed order on every GPU.  Every
rank keeps full-size tiles and whole arranged row blocks, whatever the world size.

`GatherStep` / `ChunkedGather` -- the contiguous-range variant (the reference's own batch abstraction
`kernel(scores, start, batch)`, src/bio/align.h:48): rank r scores one contiguous range per super-chunk, the gather
lands in place in packed order.  Kept for A/B runs (`SA_BENCH_PARTITION=range`); at 8 ranks its short ranges force
quarter-size tiles (85-88 % of the ideal share on cfg 2)."""
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


# ---- tile-interleaved sharding -------------------------------------------------------------------------------

def tri(j: int) -> int:
    return j * (j - 1) // 2


def column_chunks(n: int, chunks: int, ratio: float = 3.0) -> list[tuple[int, int]]:
    """Column-aligned super-chunks [(start, count)] of the packed index of n sequences, GEOMETRIC in pairs (each
    `ratio` times smaller than the one before): the all-gather + place + host copy of a super-chunk hides behind the
    kernels of the next one, and what nothing hides -- the tail of the last one -- is small."""
    pairs = tri(n)
    chunks = max(1, min(int(chunks), n - 1))
    w = [ratio ** (chunks - 1 - c) for c in range(chunks)]
    cuts, acc = [1], 0.0  # column 0 holds no pair; cuts are column indices
    for c in range(chunks - 1):
        acc += w[c] / sum(w)
        j = int(round((1 + (1 + 8 * acc * pairs) ** 0.5) / 2))  # tri(j) ~ acc * pairs
        cuts.append(min(n - 1, max(cuts[-1] + 1, j)))
    cuts.append(n)
    out = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b > a:
            out.append((tri(a), tri(b) - tri(a)))
    return out


class HipShares:
    """The device side of TiledGatherStep: sa_ctx_share_elems / sa_ctx_align_share / sa_ctx_place_shares on torch
    tensors and streams.  `host`: a binding.PinnedMatrix holding the WHOLE packed host matrix (private, or one shared
    mapping all ranks attach), or None for no host delivery."""

    device = "cuda"

    def __init__(self, ctx, use16: bool, host=None):
        import torch
        self.torch, self.ctx, self.use16, self.host = torch, ctx, bool(use16), host
        self.dtype = torch.int16 if use16 else torch.int32

    @property
    def to_host(self):
        return self.host is not None

    def share_elems(self, start, count, world):
        return self.ctx.share_elems(start, count, world, self.to_host)

    def align_share(self, start, count, world, rank, share, stream, leave_room=False):
        self.ctx.leave_room(leave_room)
        try:
            self.ctx.align_share(start, count, world, rank, share.data_ptr(), self.use16, stream.cuda_stream,
                                 self.host.ptr if self.to_host else 0)
        finally:
            self.ctx.leave_room(False)

    def place(self, start, count, world, shares, packed_range, stream):
        self.ctx.place_shares(start, count, world, shares.data_ptr(), self.use16, packed_range.data_ptr(), stream.cuda_stream,
                              self.to_host)


class _Inline:
    """stream / event stand-ins for a CPU backend (tests/test_multirank_gloo.py): everything runs in program order"""

    def wait_event(self, ev):
        pass

    def record(self, stream=None):
        pass


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class TiledGatherStep:
    """One whole-job step on `world` ranks with tile-interleaved shares (see the module docstring).

    Per super-chunk c (geometric column ranges; streams of this rank, nothing below synchronises the host):
      compute     sa_ctx_align_share: the kernels of MY tiles of super-chunk c -> my dense share (int16 or s32) (from the
                  second super-chunk on with sa_ctx_leave_room: the collective of the one before needs LDS beside them), and,
                  when the backend has a host matrix, the same scores straight into it in packed order (every rank
                  stores what it computed: together the ranks fill the host matrix exactly once -- no copy pass,
                  no waiting for the gather; the reference's device->host copies, seqalign_cuda.c:266-283)
      comm        all_gather_into_tensor(shares[c], my share)                   -> every GPU holds all shares
      deliver     sa_ctx_place_shares: widen + place into packed[start_c ...]   -> the packed s32 vector on every GPU
    The step returns with the main stream waiting for everything: an event on the main stream marks "packed s32
    vector complete on every GPU AND this rank's scores in the host matrix".

    `dist=None` with world > 1 emulates the other ranks on this device (their shares are computed here too, in place of
    the all-gather): tests.  `solo=True`: only this rank's share is computed and nothing is gathered (the other shares
    hold stale data): timing rehearsals of one rank's step on one GPU."""

    def __init__(self, backend, n: int, world: int, rank: int, chunks: int, dist=None, solo: bool = False):
        import torch

        self.torch, self.be, self.dist = torch, backend, dist
        self.world, self.rank, self.solo = world, rank, bool(solo)
        self._chained = False  # (the first step has waited for the caller's stream)
        self.pairs = tri(n)
        self.ranges = column_chunks(n, chunks)
        self.chunks = len(self.ranges)
        cuda = backend.device == "cuda"
        self.elems = [backend.share_elems(lo, cnt, world) for lo, cnt in self.ranges]
        self.shares = [torch.zeros(world * e, dtype=backend.dtype, device=backend.device) for e in self.elems]
        self.packed = torch.zeros(self.pairs, dtype=torch.int32, device=backend.device)
        if cuda:
            self.main = torch.cuda.current_stream()
            # ONE compute stream: the super-chunks' kernels run one after the other (on streams of their own they would all
            # start together and finish together -- nothing for the gather of the first one to hide behind)
            self.compute = torch.cuda.Stream()
            # the placement follows the all-gather on the SAME stream: one cross-stream hop (~15 us each) fewer per super-chunk;
            # with ONE super-chunk nothing overlaps anything, so kernels, gather and placement all run on the compute stream:
            # no cross-stream event between them at all (a hop costs 20-40 us of a 2 ms step)
            self.comm = torch.cuda.Stream() if self.chunks > 1 else self.compute
            self.deliver = self.comm
            self._event = torch.cuda.Event
        else:
            self.main = self.compute = self.comm = self.deliver = _Inline()
            self._event = _Inline

    def my_share(self, c: int, r: int | None = None):
        r = self.rank if r is None else r
        return self.shares[c][r * self.elems[c]:(r + 1) * self.elems[c]]

    def __call__(self):
        torch, dist, be = self.torch, self.dist, self.be
        cuda = be.device == "cuda"
        cs = self.compute
        if not (self._chained and self.comm is cs):
            # ordered after whatever the caller has on its stream (and, with several streams of our own, after the previous
            # step).  One super-chunk runs on the compute stream alone: that stream is in order by itself, and bouncing every
            # step through the caller's stream (its wait for the previous step's end, then this event) puts two cross-queue
            # hand-overs -- 20-50 us each -- between consecutive steps; only the first step waits for the caller.
            start = self._event()
            start.record(self.main)
            cs.wait_event(start)
            self._chained = True
        for c, (lo, cnt) in enumerate(self.ranges):
            ranks = [self.rank] if (dist is not None or self.world == 1 or self.solo) else range(self.world)
            for r in ranks:  # (more than one only when the other ranks are emulated here)
                # from the second super-chunk on, the gather and the placement of the one before run beside these kernels
                be.align_share(lo, cnt, self.world, r, self.my_share(c, r), cs, leave_room=c > 0)
            one_stream = self.comm is cs
            done = self._event()
            if not one_stream:
                done.record(cs)
            gathered = None if one_stream else done
            if dist is not None and self.world > 1:
                if not one_stream:
                    self.comm.wait_event(done)
                with (torch.cuda.stream(self.comm) if cuda else _null()):
                    # (the process group moves bytes; int16 is not among its dtypes, uint8 is)
                    dist.all_gather_into_tensor(self.shares[c].view(torch.uint8), self.my_share(c).view(torch.uint8))
                gathered = None  # (same stream as the placement below)
            if gathered is not None:
                self.deliver.wait_event(gathered)
            be.place(lo, cnt, self.world, self.shares[c], self.packed[lo:lo + cnt], self.deliver)
        fin = self._event()
        fin.record(self.deliver)
        self.main.wait_event(fin)
        if self.deliver is not cs:
            last = self._event()
            last.record(cs)  # (the host stores of the last kernels are complete when the kernels are)
            self.main.wait_event(last)
