"""Sharding of the packed pair index across ranks (one process per GPU) -- SURVEY.md §8(e).

The units (pairs) are independent, the inputs (<= a few MB) are replicated on every GPU, so each rank
scores one contiguous range of the packed index -- the reference's own batch abstraction
`kernel(scores, start, batch)` (src/bio/align.h:48) -- and a single all-gather (RCCL over xGMI when the
process group is "nccl") assembles the packed vector on every rank.  Equal-count ranges are padded to a
common length so the gather lands in place in one buffer of world*per elements."""
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
