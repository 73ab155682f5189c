#!/bin/bash
# full-size cfg4 / cfg5 (one GPU): bench.py --config cfgN --steps 2 --warmup 1, no CPU baseline, no extras  (usage: gpu_fullsize.sh <tag>)
set -o pipefail
O=gpurun_out/${1:-fullsize}; mkdir -p $O; export TMPDIR=/tmp
for c in cfg4 cfg5; do
  timeout -k 10 500 python3 bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-host-boundary > $O/bench_${c}_full_size.json 2> $O/bench_${c}.err || { tail -5 $O/bench_${c}.err; exit 1; }
  python3 tools/show_bench.py $O/bench_${c}_full_size.json
done
