#!/bin/bash
# development: WRITE_SIZE of the class kernels with and without arranged row streams
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/write_size; mkdir -p $O; export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-host-boundary --no-extra --device-resident-only"
cd /tmp
for v in sort nosort; do
  if [ $v = nosort ]; then export SA_HIP_NO_SORT=1; else unset SA_HIP_NO_SORT; fi
  rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VMEM_WR TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/$v -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $O/$v.err || { tail -5 $O/$v.err; exit 1; }
done
cd $ROOT
python3 - $O <<'PY'
import csv, glob, sys, os
from collections import defaultdict
for v in ("sort", "nosort"):
    acc = defaultdict(float); n = defaultdict(set)
    for f in glob.glob(os.path.join(sys.argv[1], v, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "systolic" in r["Kernel_Name"]:
                k = (r["Kernel_Name"].split("sa_k_systolic")[1][:14], r["Counter_Name"]); acc[k] += float(r["Counter_Value"]); n[k[0]].add(r["Dispatch_Id"])
    for k in sorted(acc):
        print(v, k, f"{acc[k] / len(n[k[0]]):.4g} per launch")
PY
