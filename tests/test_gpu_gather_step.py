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
