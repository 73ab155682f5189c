#!/bin/bash
# round-2 second GPU batch: tests, bench (direct host writes), residency census, dependency-shape microbench
set -o pipefail
O=gpurun_out/r02b; mkdir -p $O
export TMPDIR=/tmp
step() { # name, timeout, command...: stop the whole batch when a GPU step is killed by its timeout
	local name=$1 t=$2; shift 2
	timeout -k 10 $t "$@" > $O/$name.txt 2> $O/$name.err; local rc=$?
	echo "$name rc=$rc"
	if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout in $name: stopping"; exit 1; fi
	return 0
}
step pytest 900 python -m pytest tests -m gpu -q; tail -4 $O/pytest.txt
step bench 400 python3 bench.py; tail -c 1500 $O/bench.txt | head -c 1500; echo
step bench_nodirect 200 env SA_HIP_NO_DIRECT=1 python3 bench.py --no-extra --no-cpu-baseline --no-host-boundary --steps 10 --warmup 3; python3 tools/show_bench.py $O/bench_nodirect.txt 2>/dev/null | head -5
for cfg in "8 256 32 0" "8 256 64 0" "32 64 32 0" "32 64 64 0" "32 64 96 0" "32 64 96 9700" "32 64 64 9700" "32 64 64 5000" "16 128 64 10000"; do
	step census_$(echo $cfg | tr ' ' '_') 60 tools/microbench/census $cfg
done
cat $O/census_*.txt
step nw_chain 300 tools/microbench/nw_chain; cat $O/nw_chain.txt
