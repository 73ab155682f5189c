#!/bin/bash
# development: arranged row streams -- parity at sizes where they engage, then with/without timing
set -o pipefail
O=gpurun_out/sort_try; mkdir -p $O; export TMPDIR=/tmp
for a in "nw 3000 80 120" "ga 3000 80 120" "sw 3000 120 180 4 dna" "nw 4500 20 190" "sw 2600 1 60 7 dna"; do
  timeout -k 10 300 python3 tools/dev/pk_debug.py $a 2>&1 | head -8 || exit 1
done
timeout -k 10 600 python -m pytest tests/test_gpu_pk.py tests/test_gpu_parity.py tests/test_gpu_host_delivery.py tests/test_gpu_gather_step.py -q -x > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for cfg in cfg2 cfg3; do
for v in 1 0; do
  if [ $v = 1 ]; then export SA_HIP_NO_SORT=1; else unset SA_HIP_NO_SORT; fi
  timeout -k 10 300 python3 bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $O/bench_${cfg}_nosort$v.json 2> $O/bench_${cfg}_nosort$v.err || { tail -5 $O/bench_${cfg}_nosort$v.err; exit 1; }
  python3 tools/show_bench.py $O/bench_${cfg}_nosort$v.json | head -12
done; done
