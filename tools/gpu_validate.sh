#!/bin/bash
# tests + fuzz + the default bench line
set -o pipefail
O=gpurun_out/${1:-validate}; mkdir -p $O
export TMPDIR=/tmp
step() { local name=$1 t=$2; shift 2
	timeout -k 10 $t "$@" > $O/$name.txt 2> $O/$name.err; local rc=$?
	echo "$name rc=$rc"
	if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout in $name: stopping"; exit 1; fi
	return 0; }
step pytest 1000 python -m pytest tests -m gpu -q; tail -4 $O/pytest.txt
step fuzz 400 python3 tools/gpu_fuzz.py 240 2024; tail -3 $O/fuzz.txt
step bench 500 python3 bench.py; python3 tools/show_bench.py $O/bench.txt
