#!/bin/bash
# development: start/end of every kernel of the last bench step (kernel trace)
set -o pipefail
O=$(pwd)/gpurun_out/timeline; mkdir -p $O; export TMPDIR=/tmp; ROOT=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-boundary --no-extra --device-resident-only "$@" > $O/bench.json 2> $O/trace.err || { tail -5 $O/trace.err; exit 1; }
cd $ROOT
python3 - $O <<'PY'
import csv, glob, sys, os
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Workgroup_Size_X", 0) or 0), int(r.get("Grid_Size_X", 0) or 0)))
rows.sort()
sysr = [r for r in rows if "systolic" in r[2]]
# last step = last group of kernels separated by a gap
last = [sysr[-1]]
for r in reversed(sysr[:-1]):
    if min(x[0] for x in last) - r[1] > 200_000: break
    last.append(r)
last.sort()
t0 = last[0][0]
for s, e, n, wg, grid in last:
    n = n.split("sa_k_")[1][:40] if "sa_k_" in n else n[:40]
    print(f"{(s-t0)/1e6:8.3f} .. {(e-t0)/1e6:8.3f} ms  ({(e-s)/1e6:7.3f})  grid {grid//max(wg,1):5d} wgs  {n}")
print(f"step span {(max(x[1] for x in last)-t0)/1e6:.3f} ms")
PY
