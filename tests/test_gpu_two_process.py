"""GPU (-m gpu): TWO PROCESSES on the one GPU of the box -- the multi-process side of bench.py's N > 1 path that an
emulation inside one process cannot show: two ranks with their own contexts and kernels on the same device, ONE host
matrix (a shared file mapping) that both page-lock and both store into from their kernels, a real inter-process
all-gather of the dense shares, placement on both ranks.  The collective runs over gloo (RCCL refuses two ranks on one
device), staged through host memory; everything else is the code path of `bench.py --gpus N`."""
import os
import pathlib
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parents[1]

WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
import sequencealigner_amd as sa
from sequencealigner_amd.distributed import HipShares, TiledGatherStep
from tests.oracle_binding import Oracle
from tests.synth import make_protein_set

torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
shm = sys.argv[2]

class GlooViaHost:  # gloo moves host tensors: stage the device shares through the host, on the step's collective stream
    def all_gather_into_tensor(self, out, inp):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(o, inp.cpu())
        out.copy_(o)

store = sa.SequenceStore.from_sequences(make_protein_set(1500, 30, 190, 47))
for method, gaps in (("nw", dict(gap_pen=4)), ("sw", dict(gap_open=10, gap_extend=1))):
    scoring = sa.Scoring.from_names(method, "blosum62", **gaps)
    want = Oracle().align(store, scoring, triangular=True, threads=8)
    for chunks, use16 in ((1, True), (3, False)):
        if rank == 0:
            np.full(store.pairs, -(2 ** 31), np.int32).tofile(shm)
        dist.barrier()
        host = sa.PinnedMatrix(store.pairs, shared=shm, create=False)
        with sa.Context(store, scoring, 0) as ctx:
            step = TiledGatherStep(HipShares(ctx, use16, host), store.num, world, rank, chunks, GlooViaHost())
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            assert np.array_equal(step.packed.cpu().numpy(), want), f"rank {rank}: placed vector differs ({method}, {chunks} chunks)"
            dist.barrier()
            # both ranks' kernels stored into the ONE host matrix: complete and right, seen from either process
            assert np.array_equal(host.array, want), f"rank {rank}: shared host matrix differs ({method}, {chunks} chunks)"
            dist.barrier()
        host.close()
        del host
if rank == 0:
    print("TWO_PROCESS_OK", world)
dist.destroy_process_group()
"""


def test_two_ranks_share_one_gpu_and_one_host_matrix(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="8", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script), str(ROOT), str(tmp_path / "host_matrix.bin")]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "TWO_PROCESS_OK 2" in res.stdout


@pytest.mark.parametrize("world", [2, 4])
def test_bench_multi_rank_code_path_on_one_gpu(world):
    """bench.py's own N > 1 code with world = 2 and 4 (SA_BENCH_ONE_GPU_REHEARSAL: every rank on device 0, gloo instead of
    RCCL): shared host matrix under /dev/shm, the untimed trial over partition x super-chunks, timed steps, per-rank
    verification, rank 0's CPU leg while the others wait on the flag file -- the JSON line must report the gathered matrix
    and the host matrix verified on every rank, a cpu_baseline and a parity block without mismatches"""
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SA_BENCH_ONE_GPU_REHEARSAL="1", OMP_NUM_THREADS="8",
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--nseq", "3000",
           "--cpu-seconds", "2"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == world and line["scaling"] == "strong" and line["value"] > 0
    assert line["config"]["gathered_and_host_result_verified_on_every_rank"] is True
    trial = line["config"]["super_chunk_trial_ms"]
    assert {"tiled x1", "tiled x2", "tiled x3", "range x1", "range x2"} <= set(trial)
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["kind"] in ("reference", "port")
    assert line["parity"]["mismatches"] == 0 and line["parity"]["pairs_compared"] > 10_000
    assert line["dtype"] == "u16x2"
