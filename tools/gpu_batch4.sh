#!/bin/bash
# tests + the default bench line + profiles of the packed kernels
set -o pipefail
O=gpurun_out/r02d; mkdir -p $O
export TMPDIR=/tmp
step() { local name=$1 t=$2; shift 2
	timeout -k 10 $t "$@" > $O/$name.txt 2> $O/$name.err; local rc=$?
	echo "$name rc=$rc"
	if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout in $name: stopping"; exit 1; fi
	return 0; }
step pytest 900 python -m pytest tests -m gpu -q; tail -4 $O/pytest.txt
step bench 500 python3 bench.py; python3 tools/show_bench.py $O/bench.txt
