#!/bin/bash
# final profiles of the round: rocprofv3 kernel stats + HBM traffic (cfg2, cfg3, cfg4 shape) + SQ counters
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/${1:-round}
timeout -k 10 400 bash tools/profile_gpu.sh ${1:-round}_cfg2 --no-extra --cpu-full-seconds 0 > gpurun_out/${1:-round}/prof_cfg2.log 2>&1; echo "prof cfg2 rc=$?"
timeout -k 10 400 bash tools/profile_gpu.sh ${1:-round}_cfg3 --config cfg3 --no-extra --cpu-seconds 4 --cpu-full-seconds 0 > gpurun_out/${1:-round}/prof_cfg3.log 2>&1; echo "prof cfg3 rc=$?"
timeout -k 10 500 bash tools/profile_gpu.sh ${1:-round}_cfg4 --config cfg4 --n 12000 --no-extra --cpu-seconds 4 --cpu-full-seconds 0 > gpurun_out/${1:-round}/prof_cfg4.log 2>&1; echo "prof cfg4 rc=$?"
for c in "cfg2" "cfg3" "cfg4 --n 12000"; do t=$(echo $c | cut -d" " -f1); timeout -k 10 300 bash tools/pmc_sq.sh ${1:-round}_$t --config $c > gpurun_out/${1:-round}/pmc_$t.txt 2>&1; echo "pmc $t rc=$?"; done
