#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_ic; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-host-boundary --n 6000"
cd /tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQC_TC_STALL --output-format csv -d $OUT/p1 -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/p1.err || tail -3 $OUT/p1.err
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU2 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/p2 -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/p2.err || tail -3 $OUT/p2.err
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, os
from collections import defaultdict
d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
for f in glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "systolic" not in k: continue
        acc[k.split("sa_k_systolic")[1][:12]][r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(acc):
    print(k, "  ".join(f"{c}={v:.4g}" for c, v in sorted(acc[k].items())))
PY
