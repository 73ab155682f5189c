#!/bin/bash
# round-2 first GPU batch: tests, bench, microbench calibration, PMC of the GA / SW kernels
set -o pipefail
mkdir -p gpurun_out/r02a
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02a/pytest.txt
tail -5 gpurun_out/r02a/pytest.txt
timeout -k 10 400 python3 bench.py > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/r02a/bench.json
timeout -k 10 60 python3 bench.py --gpus 2 > gpurun_out/r02a/bench_gpus2.txt 2>&1; echo "bench --gpus 2 rc=$?"; tail -2 gpurun_out/r02a/bench_gpus2.txt
SA_BENCH_FORCE_DIST=1 timeout -k 10 120 python3 bench.py --steps 5 --warmup 2 > gpurun_out/r02a/bench_rehearsal.json 2> gpurun_out/r02a/bench_rehearsal.err; echo "rehearsal rc=$?"; tail -c 1500 gpurun_out/r02a/bench_rehearsal.json
timeout -k 10 120 tools/microbench/valu_rates > gpurun_out/r02a/valu_rates.txt 2>&1; echo "valu rc=$?"; tail -3 gpurun_out/r02a/valu_rates.txt
timeout -k 10 300 bash tools/pmc_sq.sh r02a_cfg3 --config cfg3 > gpurun_out/r02a/pmc_cfg3.txt 2>&1; echo "pmc cfg3 rc=$?"
timeout -k 10 300 bash tools/pmc_sq.sh r02a_cfg4 --config cfg4 --n 12000 > gpurun_out/r02a/pmc_cfg4.txt 2>&1; echo "pmc cfg4 rc=$?"
timeout -k 10 300 bash tools/pmc_sq.sh r02a_cfg2 --config cfg2 > gpurun_out/r02a/pmc_cfg2.txt 2>&1; echo "pmc cfg2 rc=$?"
