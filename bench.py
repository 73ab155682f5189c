#!/usr/bin/env python3
"""bench.py -- all-vs-all alignment throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE pass of the hot path over the whole workload: every pair i<j of the synthetic
sequence set is scored (NW/BLOSUM62/gap 4 for the headline config) into the packed
upper-triangular vector resident in HBM.  With N>1 ranks the packed pair index is cut into N
contiguous ranges (strong scaling: total work fixed), every rank scores its range, and ONE RCCL
all-gather over xGMI assembles the full vector on every GPU -- that all-gather is inside the timed
step.  Inputs are resident in HBM before the timed region.

Printed JSON (rank 0): metric/value per the driver contract, plus
  roofline     -- dominant kernel vs the HBM roof, live HIP-event timing on the launch stream
  cpu_baseline -- the reference's own CPU path (oracle/_ref, kind "reference") or our C restatement
                  (oracle/, kind "port") timed on this host's cores on a bounded sample
  gcups, valu  -- the bound that actually constrains this integer DP
"""
from __future__ import annotations

import argparse
import json
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import sequencealigner_amd as sa  # noqa: E402
from sequencealigner_amd.distributed import ChunkedGather  # noqa: E402
from tests.synth import CONFIGS, make_config  # noqa: E402

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_LANE_OPS = 256 * 4 * 32 * 2.4e9  # 256 CU x 4 SIMD-32 x 2.4 GHz = 7.86e13 s32 lane-ops/s
OPS_PER_CELL = {"nw": 5, "ga": 9, "sw": 11}  # reference op counts (nw.c:29-35, ga.c:47-62, sw.c:39-57)
# SIMD issue cycles one wave64 DP cell costs at the very least with this kernel's instruction selection, from the
# measured per-opcode rates at 8 waves/SIMD (profiles/r01_microbench_valu_rates.txt, r01f_microbench_vgpr_banks.txt):
# SDWA add / v_max3 2.3, v_max_i32 1.77, v_add_u32 1.1   -> nw 2 ops, ga 5 ops, sw 9 ops per cell
MIN_ISSUE_CYCLES_PER_CELL = {"nw": 2 * 2.3, "ga": 2 * 2.3 + 2 * 1.77 + 1.1, "sw": 2 * 2.3 + 4 * 1.77 + 3 * 1.1}
SIMDS, SHADER_HZ = 256 * 4, 2.4e9


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--n", "--nseq", dest="n", type=int, default=None, help="override sequence count (parity/debug runs)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline duration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-boundary", action="store_true", help="skip the untimed sa_hip_align end-to-end check (profiling runs)")
    ap.add_argument("--chunks", type=int, default=None, help="super-chunks per step for compute/all-gather overlap (N>1; default: pick 1, 2 or 4 by an untimed trial)")
    return ap.parse_args()


def cpu_baseline(seqs, cfg, target_s: float) -> dict:
    """Reference CPU path on a bounded prefix of the same workload (pairs/s, all host cores)."""
    from tests.oracle_binding import Oracle, RefLib, ref_available

    ncores = os.cpu_count() or 1
    oracle = Oracle()
    threads = min(ncores, oracle.max_threads)
    scoring = sa.Scoring.from_names(cfg["method"], cfg["matrix"], **cfg["gaps"])

    def run(n, use_ref):
        store = sa.SequenceStore.from_sequences(seqs[:n])
        if use_ref:
            ref = RefLib(cfg["method"], cfg["matrix"], threads=threads, **cfg["gaps"])
            try:
                t = time.perf_counter()
                ref.align(store, triangular=True)
                return time.perf_counter() - t, store
            finally:
                ref.close()
        t = time.perf_counter()
        oracle.align(store, scoring, triangular=True, threads=threads)
        return time.perf_counter() - t, store

    use_ref = False
    if ref_available():
        try:
            run(64, True)
            use_ref = True
        except Exception:
            use_ref = False
    n0 = min(len(seqs), 600)
    t0, _ = run(n0, use_ref)
    rate = (n0 * (n0 - 1) / 2) / max(t0, 1e-6)
    n = int(min(len(seqs), max(n0, (2 * rate * target_s) ** 0.5)))
    t, store = run(n, use_ref)
    pairs = n * (n - 1) // 2
    return {
        "value": pairs / t, "unit": "pair-alignments/s", "cores": threads,
        "kind": "reference" if use_ref else "port",
        "sample": f"first {n} sequences of the workload = {pairs} pairs, {store.cells()} cells, {t:.2f} s",
        "gcups": store.cells() / t / 1e9,
    }


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("SA_BENCH_FORCE_DIST"):  # the env switch rehearses the RCCL path on one GPU
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    seqs, cfg = make_config(args.config, args.n)
    store = sa.SequenceStore.from_sequences(seqs)
    scoring = sa.Scoring.from_names(cfg["method"], cfg["matrix"], **cfg["gaps"])
    pairs = store.pairs
    cells = store.cells()

    # strong scaling: total work fixed.  The packed index is cut chunk-major / rank-minor (ChunkedGather) so that
    # the all-gather of super-chunk c overlaps the kernels of super-chunk c+1 and lands in place in packed order.
    use_dist = dist is not None
    ctx = sa.Context(store, scoring, local_rank)
    main = torch.cuda.current_stream()
    comm = torch.cuda.Stream() if use_dist else None
    side = []

    # exchange format of the all-gather: int16 when every score of this workload provably fits (half the bytes over
    # xGMI); the gathered vector is widened to the reference's s32 on every GPU inside the timed step
    use16 = use_dist and ctx.scores_fit16 and not os.environ.get("SA_BENCH_GATHER32")

    def schedule(chunks):
        """(sched, step) for `chunks` super-chunks per step; one compute stream per super-chunk, so the kernels of
        consecutive super-chunks may overlap while the gather of each starts as soon as its own kernels finish."""
        sched = ChunkedGather(pairs, world, rank, chunks)
        while use_dist and len(side) < chunks:
            side.append(torch.cuda.Stream())
        streams = side[:chunks] if use_dist else [main]

        def step():
            works = []
            if use_dist:
                start = torch.cuda.Event()
                start.record(main)
            for c in range(sched.chunks):
                lo, hi = sched.slice_range(c)
                cs = streams[c]
                if use_dist:
                    cs.wait_event(start)  # ordered after the previous step
                buf = packed16 if use16 else packed
                if use16:
                    ctx.align_range16(lo, hi - lo, sched.my_slice(buf, c).data_ptr(), cs.cuda_stream)
                else:
                    ctx.align_range(lo, hi - lo, sched.my_slice(buf, c).data_ptr(), cs.cuda_stream)
                if use_dist:
                    done = torch.cuda.Event()
                    done.record(cs)
                    with torch.cuda.stream(comm):
                        comm.wait_event(done)
                        # (the process group moves bytes; int16 is not among its dtypes, uint8 is)
                        works.append(dist.all_gather_into_tensor(sched.super_chunk(buf, c).view(torch.uint8),
                                                                 sched.my_slice(buf, c).view(torch.uint8), async_op=True))
            for w in works:
                w.wait()  # the main stream waits for the gathers (and therefore the kernels) of this step
            if use16:
                ctx.widen16(packed16.data_ptr(), packed.data_ptr(), sched.total, main.cuda_stream)
        return sched, step

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # super-chunks per step: more of them hide more of the all-gather behind the kernels but shorten the kernels'
    # row streams; the trade depends on the fabric, so (unless --chunks fixes it) it is measured before the warmup,
    # untimed, and every rank takes the same decision from the max-over-ranks time
    candidates = [args.chunks] if (args.chunks or not use_dist) else [1, 2, 4]
    if not use_dist:
        candidates = [1]
    packed = torch.zeros(max(ChunkedGather(pairs, world, rank, c).total for c in candidates), dtype=torch.int32, device="cuda")
    packed16 = torch.zeros(packed.numel(), dtype=torch.int16, device="cuda") if use16 else None
    tuned = {}
    if len(candidates) > 1:
        for c in candidates:
            _, trial = schedule(c)
            trial(); trial()
            fence()
            t0 = time.perf_counter()
            for _ in range(4):
                trial()
            fence()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            tuned[c] = float(t.item()) / 4 * 1e3
        candidates = [min(tuned, key=tuned.get)]
    sched, step = schedule(candidates[0])
    my_pairs = sum(hi - lo for lo, hi in (sched.slice_range(c) for c in range(sched.chunks)))
    my_cells = sum(store.cells(lo, hi - lo) for lo, hi in (sched.slice_range(c) for c in range(sched.chunks)))

    for _ in range(args.warmup):
        step()
    fence()
    ctx.timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    tm = ctx.timing_read()
    ctx.timing(False)

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # N>1: every rank re-scores a few random windows of the gathered vector with its own kernels (untimed) and
    # compares -- the gathered result of the last step must be the packed matrix in natural order on every GPU
    gather_ok = None
    if use_dist:
        vrng = np.random.default_rng(1234 + rank)
        okflag = 1
        for _ in range(6):
            w = int(min(pairs, 65536))
            a0 = int(vrng.integers(0, pairs - w + 1))
            chk = torch.empty(w, dtype=torch.int32, device="cuda")
            ctx.align_range(a0, w, chk.data_ptr(), main.cuda_stream)
            torch.cuda.synchronize()
            if not torch.equal(chk, packed[a0:a0 + w]):
                okflag = 0
        t = torch.tensor([okflag], dtype=torch.int32, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        gather_ok = bool(t.item())

    # end-to-end through the host boundary (upload + kernels + D2H of the packed result), rank 0, N=1 only
    e2e = None
    if world == 1 and not args.no_host_boundary:
        t1 = time.perf_counter()
        host = sa.hip_align(store, scoring, triangular=True)
        e2e = time.perf_counter() - t1
        e2e_phase = sa.last_align_seconds()
        assert np.array_equal(host, packed[:pairs].cpu().numpy()), "host-boundary result differs from device-resident result"

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = pairs * args.steps / elapsed
        # dominant kernel: algorithmic HBM bytes of ONE launch (4 B per pair written + the
        # sequence store and its offsets read once) / that kernel's average launch duration (HIP events)
        launches = max(tm["launches"], 1)
        k_pairs = tm["pairs"] // launches
        k_cells = tm["cells"] // launches
        alg_bytes = 4 * k_pairs + int(store.blob.size) + 8 * store.num
        avg_ms = tm["ms"] / launches
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # the class kernels of a range run concurrently (side streams), so a single kernel's duration includes time it
        # shared the chip with its siblings; the VALU view therefore uses the whole step (all kernels, this rank)
        kernel_gcups = my_cells / (elapsed / args.steps) / 1e9
        kname = tm["kernel"]
        # HBM traffic of that kernel per launch from the committed PMC passes (profiles/*_traffic.json), if the
        # workload and kernel match; null otherwise (it cannot be measured inside this process)
        traffic = None
        for tf in sorted((ROOT / "profiles").glob("*_traffic.json")):
            tj = json.loads(tf.read_text())
            if tj.get("kernel") == kname and tj.get("workload") == args.config and world == 1 and args.n is None:
                traffic = tj["traffic_bytes_per_launch"]
        out = {
            "metric": "pair-alignments/sec", "value": value, "unit": "pair-alignments/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "s32",
            "data": "synthetic",
            "config": {"workload": f"{args.config}: {store.num} {cfg['kind']} seqs x U[{cfg['lo']},{cfg['hi']}], "
                                   f"{cfg['method']} {cfg['matrix']} {cfg['gaps']}, all-vs-all packed triangular",
                       "pairs": pairs, "cells": cells, "parallelism": f"pair-range x{world}" + (f" + RCCL all-gather ({'int16 exchange, widened to s32 on device' if use16 else 's32'}), {sched.chunks} overlapped super-chunks" if use_dist else ""),
                       **({"super_chunk_trial_ms": tuned} if tuned else {}),
                       **({"gathered_result_verified_on_every_rank": gather_ok} if gather_ok is not None else {})},
            "gcups": cells * args.steps / elapsed / 1e9,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": kname,
                         "kernel_avg_ms": avg_ms, "launches": tm["launches"], "algorithmic_bytes_per_launch": alg_bytes,
                         "pairs_per_launch": k_pairs, "sum_of_kernel_ms_per_step": tm["all_kernels_ms"] / args.steps,
                         "note": "per-class kernels of one range overlap on side streams; durations are per launch as rocprofv3 reports them"},
            "valu": {"gcups_this_rank": kernel_gcups, "reference_ops_per_cell": OPS_PER_CELL[scoring.method_name],
                     "peak_lane_ops_per_s": VALU_LANE_OPS,
                     "frac_of_valu_peak_at_reference_op_count": kernel_gcups * 1e9 * OPS_PER_CELL[scoring.method_name] / VALU_LANE_OPS,
                     # the bound that actually applies: wave-cells x minimal issue cycles per cell / SIMD cycles available
                     "min_issue_cycles_per_wave_cell": MIN_ISSUE_CYCLES_PER_CELL[scoring.method_name],
                     "frac_of_valu_issue_bound": (my_cells / 64) * MIN_ISSUE_CYCLES_PER_CELL[scoring.method_name]
                                                 / ((elapsed / args.steps) * SIMDS * SHADER_HZ)},
            "device": sa.device_name(local_rank),
        }
        if e2e is not None:
            out["host_boundary"] = {"seconds": e2e, "pairs_per_s": pairs / e2e,
                                    "align_phase_seconds": e2e_phase, "align_phase_pairs_per_s": pairs / e2e_phase if e2e_phase > 0 else None,
                                    "note": "sa_hip_align: encode+upload+kernels+D2H into pageable host memory (PCIe-inclusive); "
                                            "align_phase = its launch/copy loop only, the phase the reference times (SURVEY 8d)"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(seqs, cfg, args.cpu_seconds)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
