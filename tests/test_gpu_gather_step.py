"""GPU (-m gpu): the overlapped multi-stream step bench.py times at N > 1 (sequencealigner_amd/distributed.py:
GatherStep -- several sa_ctx_align_range calls in flight on different streams of ONE context, in-place RCCL
all-gather per super-chunk, int16 exchange + widen, per-rank device->host delivery) against the ORACLE, driven the
way bench.py's 1-rank rehearsal drives it: a real "nccl" (= RCCL) process group of world size 1."""
import os

import numpy as np
import pytest

from tests.synth import make_protein_set

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl_group():
    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("method,gaps", [("nw", dict(gap_pen=4)), ("sw", dict(gap_open=10, gap_extend=1))])
def test_gather_step_matches_oracle(method, gaps, sa, oracle, rccl_group):
    import torch
    from sequencealigner_amd.distributed import GatherStep

    seqs = make_protein_set(1100, 15, 140, 41)  # 604 450 pairs, mixed length classes
    store = sa.SequenceStore.from_sequences(seqs)
    scoring = sa.Scoring.from_names(method, "blosum62", **gaps)
    want = oracle.align(store, scoring, triangular=True, threads=16)
    pairs = store.pairs
    with sa.Context(store, scoring, 0) as ctx:
        assert ctx.scores_fit16
        for chunks in (1, 2, 4):
            for use16 in (False, True):
                for group in (rccl_group, None):
                    step = GatherStep(ctx, pairs, 1, 0, chunks, group, use16)
                    for _ in range(3):  # back-to-back steps: counter-ring slots, stream re-use, buffer re-use
                        step()
                    torch.cuda.synchronize()
                    got = step.packed[:pairs].cpu().numpy()
                    assert np.array_equal(got, want), (chunks, use16, group is not None)
                    host = np.concatenate([step.host[ho:ho + hi - lo].numpy() for lo, hi, ho in step.host_ranges()])
                    assert np.array_equal(host, want), (chunks, use16, "host share")


def test_gather_step_emulated_ranks_cover_the_pair_space(sa, oracle):
    """world = 3 without a process group: every emulated rank computes and delivers its slices; together the host
    shares are the packed matrix in natural order (what the N-rank run assembles on one node)."""
    import torch
    from sequencealigner_amd.distributed import GatherStep

    seqs = make_protein_set(500, 40, 90, 43)
    store = sa.SequenceStore.from_sequences(seqs)
    scoring = sa.Scoring.from_names("ga", "blosum62", gap_open=10, gap_extend=1)
    want = oracle.align(store, scoring, triangular=True, threads=16)
    out = np.full(store.pairs, np.iinfo(np.int32).min, np.int32)
    with sa.Context(store, scoring, 0) as ctx:
        for rank in range(3):
            step = GatherStep(ctx, store.pairs, 3, rank, 2, None, use16=True)
            step()
            torch.cuda.synchronize()
            for lo, hi, ho in step.host_ranges():
                out[lo:hi] = step.host[ho:ho + hi - lo].numpy()
    assert np.array_equal(out, want)


# ---- tile-interleaved shares (what bench.py times at N > 1): sa_ctx_align_share + all-gather + sa_ctx_place_shares ----

@pytest.mark.parametrize("method,gaps", [("nw", dict(gap_pen=4)), ("ga", dict(gap_open=10, gap_extend=1)),
                                         ("sw", dict(gap_open=10, gap_extend=1))])
def test_tiled_step_matches_oracle_through_rccl(method, gaps, sa, oracle, rccl_group):
    """world = 1 with a real "nccl" (= RCCL) group, and without a group: dense share -> place == the oracle's matrix,
    and the host matrix the kernels stored into directly == the oracle's matrix"""
    import torch
    from sequencealigner_amd.distributed import HipShares, TiledGatherStep

    seqs = make_protein_set(1100, 15, 140, 41)
    store = sa.SequenceStore.from_sequences(seqs)
    scoring = sa.Scoring.from_names(method, "blosum62", **gaps)
    want = oracle.align(store, scoring, triangular=True, threads=16)
    host = sa.PinnedMatrix(store.pairs)
    try:
        with sa.Context(store, scoring, 0) as ctx:
            for chunks in (1, 3):
                for use16 in (False, True):
                    for group in (rccl_group, None):
                        for to_host in (True, False):
                            host.array[:] = -1
                            step = TiledGatherStep(HipShares(ctx, use16, host if to_host else None), store.num, 1, 0, chunks, group)
                            for _ in range(3):
                                step()
                            torch.cuda.synchronize()
                            assert np.array_equal(step.packed.cpu().numpy(), want), (chunks, use16, group is not None, to_host)
                            if to_host:
                                assert np.array_equal(host.array, want), (chunks, use16, "host matrix")
    finally:
        host.close()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_tiled_step_emulated_ranks_place_the_oracle_matrix(world, sa, oracle):
    """every rank's share computed on this device in place of the all-gather: the placed vector is the oracle's matrix,
    and the ranks' direct host stores fill the host matrix exactly (every element written, every element right)"""
    import torch
    from sequencealigner_amd.distributed import HipShares, TiledGatherStep

    seqs = make_protein_set(1500, 30, 190, 47)  # 1.1 M pairs, all 8-lane packed classes
    store = sa.SequenceStore.from_sequences(seqs)
    host = sa.PinnedMatrix(store.pairs)
    try:
        for method, gaps in (("nw", dict(gap_pen=4)), ("ga", dict(gap_open=10, gap_extend=1))):
            scoring = sa.Scoring.from_names(method, "blosum62", **gaps)
            want = oracle.align(store, scoring, triangular=True, threads=16)
            with sa.Context(store, scoring, 0) as ctx:
                for chunks, use16 in ((1, True), (2, False), (3, True)):
                    host.array[:] = np.iinfo(np.int32).min
                    step = TiledGatherStep(HipShares(ctx, use16, host), store.num, world, world // 2, chunks, None)
                    step()
                    step()
                    torch.cuda.synchronize()
                    assert np.array_equal(step.packed.cpu().numpy(), want), (method, world, chunks)
                    assert np.array_equal(host.array, want), (method, world, chunks, "host matrix")
                # one rank alone stores only its own tiles: the others' elements stay untouched, its own are right
                host.array[:] = np.iinfo(np.int32).min
                step = TiledGatherStep(HipShares(ctx, True, host), store.num, world, 0, 2, None, solo=True)
                step()
                torch.cuda.synchronize()
                mine = host.array != np.iinfo(np.int32).min
                assert 0.8 / world < mine.mean() < 1.25 / world, (world, mine.mean())
                assert np.array_equal(host.array[mine], want[mine])
    finally:
        host.close()


def test_share_sizes_and_balance(sa):
    """every rank's share has the same padded length, and the dense shares of a world are within a few per cent of
    pairs / world elements (tile-order padding only)"""
    seqs = make_protein_set(3000, 80, 120, 2)
    store = sa.SequenceStore.from_sequences(seqs)
    scoring = sa.Scoring.from_names("nw", "blosum62", gap_pen=4)
    with sa.Context(store, scoring, 0) as ctx:
        for world in (1, 2, 4, 8):
            for to_host in (False, True):
                e = ctx.share_elems(0, store.pairs, world, to_host)
                assert e * world >= store.pairs
                assert e * world <= 1.10 * store.pairs + 4096 * world, (world, e, store.pairs)


def test_tiled_shares_cover_every_kernel_family(sa, oracle):
    """columns of 8-lane and 16-lane packed classes, s32 classes, the strip-mined class and (Gotoh with |open| < |extend|)
    the pair-per-wave kernels in ONE store: shares of 3 ranks, placed, equal the oracle -- and so does the host matrix
    they stored into"""
    import torch

    seqs = (make_protein_set(40, 20, 150, 5) + make_protein_set(14, 200, 630, 6) + make_protein_set(6, 700, 1000, 7)
            + make_protein_set(2, 1100, 1400, 8) + make_protein_set(10, 30, 90, 9))
    store = sa.SequenceStore.from_sequences(seqs)
    host = sa.PinnedMatrix(store.pairs)
    try:
        for method, gaps in (("nw", dict(gap_pen=4)), ("sw", dict(gap_open=10, gap_extend=1)), ("ga", dict(gap_open=3, gap_extend=7))):
            scoring = sa.Scoring.from_names(method, "blosum62", **gaps)
            want = oracle.align(store, scoring, triangular=True, threads=16)
            with sa.Context(store, scoring, 0) as ctx:
                for world, lo, to_host in ((3, 0, True), (2, 421, False), (2, 421, True)):  # whole job; a range that starts inside a column
                    cnt = store.pairs - lo
                    e = ctx.share_elems(lo, cnt, world, to_host)
                    shares = torch.full((world * e,), -7, dtype=torch.int32, device="cuda")
                    packed = torch.full((cnt,), -9, dtype=torch.int32, device="cuda")
                    host.array[:] = -11
                    s = torch.cuda.current_stream().cuda_stream
                    for r in range(world):
                        ctx.align_share(lo, cnt, world, r, shares.data_ptr() + 4 * r * e, False, s, host.ptr if to_host else 0)
                    ctx.place_shares(lo, cnt, world, shares.data_ptr(), False, packed.data_ptr(), s, to_host)
                    torch.cuda.synchronize()
                    assert np.array_equal(packed.cpu().numpy(), want[lo:]), (method, world, lo)
                    if to_host:
                        assert np.array_equal(host.array[lo:], want[lo:]), (method, world, lo, "host matrix")
                        assert (host.array[:lo] == -11).all()
    finally:
        host.close()


def test_share_host_matrix_must_be_page_locked(sa):
    import torch
    seqs = make_protein_set(64, 30, 60, 3)
    store = sa.SequenceStore.from_sequences(seqs)
    scoring = sa.Scoring.from_names("nw", "blosum62", gap_pen=4)
    pageable = np.zeros(store.pairs, np.int32)
    with sa.Context(store, scoring, 0) as ctx:
        e = ctx.share_elems(0, store.pairs, 1, True)
        share = torch.zeros(e, dtype=torch.int32, device="cuda")
        with pytest.raises(sa.AlignError, match="page-locked"):
            ctx.align_share(0, store.pairs, 1, 0, share.data_ptr(), False, 0, pageable.ctypes.data)


@pytest.mark.parametrize("n", [2, 3, 9, 40])
def test_tiled_step_on_tiny_stores(n, sa, oracle):
    """fewer pairs than ranks, ranks without a tile, super-chunks of single columns: the step still assembles the matrix"""
    import torch
    from sequencealigner_amd.distributed import HipShares, TiledGatherStep

    seqs = make_protein_set(n, 5, 70, 90 + n)
    store = sa.SequenceStore.from_sequences(seqs)
    host = sa.PinnedMatrix(store.pairs)
    try:
        for method, gaps in (("nw", dict(gap_pen=4)), ("sw", dict(gap_open=10, gap_extend=1))):
            scoring = sa.Scoring.from_names(method, "blosum62", **gaps)
            want = oracle.align(store, scoring, triangular=True)
            with sa.Context(store, scoring, 0) as ctx:
                for world, chunks in ((8, 1), (8, 3), (3, 2)):
                    host.array[:] = -5
                    step = TiledGatherStep(HipShares(ctx, True, host), store.num, world, world - 1, chunks, None)
                    step()
                    torch.cuda.synchronize()
                    assert np.array_equal(step.packed.cpu().numpy(), want), (n, method, world, chunks)
                    assert np.array_equal(host.array, want), (n, method, world, chunks, "host")
    finally:
        host.close()
