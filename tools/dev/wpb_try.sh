#!/bin/bash
# development: quick parity + timing of the current build (cfg2, cfg3, cfg4 shape), device-resident and host-delivered
set -o pipefail
O=gpurun_out/wpb_try; mkdir -p $O; export TMPDIR=/tmp
for a in "nw 3000 80 120" "ga 3000 80 120" "sw 3000 120 180 4 dna"; do
  timeout -k 10 300 python3 tools/dev/pk_debug.py $a 2>&1 | grep -v amdgpu.ids | head -4 || exit 1
done
for cfg in "cfg2" "cfg3" "cfg4 --n 12000"; do
  t=$(echo $cfg | cut -d" " -f1)
  timeout -k 10 300 python3 bench.py --config $cfg --steps 8 --warmup 2 --no-cpu-baseline --no-extra --no-host-boundary > $O/bench_$t.json 2> $O/bench_$t.err || { tail -5 $O/bench_$t.err; exit 1; }
  python3 - $O/bench_$t.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{j['config']['workload'][:40]:40s} host {j['ms_per_step']:.2f} ms  resident {j['device_resident']['ms_per_step']:.2f} ms  frac {j['valu']['frac_of_valu_issue_bound']:.3f}")
PY
done
