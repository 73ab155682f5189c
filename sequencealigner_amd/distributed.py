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
