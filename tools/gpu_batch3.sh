#!/bin/bash
# tests + quick device-resident bench of cfg2 / cfg3 (development loop of the packed kernels)
set -o pipefail
O=gpurun_out/r02c; mkdir -p $O
export TMPDIR=/tmp
step() { local name=$1 t=$2; shift 2
	timeout -k 10 $t "$@" > $O/$name.txt 2> $O/$name.err; local rc=$?
	echo "$name rc=$rc"
	if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout in $name: stopping"; exit 1; fi
	return 0; }
step pytest 900 python -m pytest tests -m gpu -x -q; tail -15 $O/pytest.txt
step bench_cfg2 200 python3 bench.py --no-extra --no-cpu-baseline --no-host-boundary --steps 10 --warmup 3; python3 tools/show_bench.py $O/bench_cfg2.txt | head -3
step bench_cfg3 200 python3 bench.py --config cfg3 --no-extra --no-cpu-baseline --no-host-boundary --steps 5 --warmup 2; python3 tools/show_bench.py $O/bench_cfg3.txt | head -3
tail -3 $O/bench_cfg2.err $O/bench_cfg3.err
