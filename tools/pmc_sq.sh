#!/bin/bash
# SQ counter passes over a short bench run (development helper; run through gpurun)
TAG=${1:-x}; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-host-boundary --no-extra --device-resident-only $@"
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p1 -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/p1.err || tail -3 $OUT/p1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/p2 -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/p2.err || tail -3 $OUT/p2.err
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/p3 -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/p3.err || tail -3 $OUT/p3.err
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, os
from collections import defaultdict
d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for f in glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "systolic" not in k: continue
        k = k.split("sa_k_systolic")[1][:16]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]): print(f"   {c:28s} {acc[k][c]:.4g}")
PY
