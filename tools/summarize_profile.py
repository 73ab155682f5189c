#!/usr/bin/env python3
"""Condense a tools/profile_gpu.sh output directory into the summary that is committed under profiles/."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]
print(f"# profile summary of {os.path.basename(d)}")
for name in ("bench.json",):
    p = os.path.join(d, name)
    if os.path.exists(p):
        line = open(p).read().strip().splitlines()[-1]
        j = json.loads(line)
        print("bench:", json.dumps({k: j[k] for k in ("value", "ms_per_step", "gcups", "roofline") if k in j}))

# kernel stats
for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("\n## rocprofv3 --kernel-trace --stats :", os.path.relpath(f, d))
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print("  {Name:60s} calls={Calls:>5s} total_ns={TotalDurationNs:>14s} avg_ns={AverageNs:>14s} pct={Percentage}".format(**r))

# the same run seen by bench.py (HIP events around the dominant kernel) next to rocprofv3's per-dispatch durations:
# the first dispatch of every kernel belongs to the untimed warmup step (bench runs with --warmup 1 here)
tb = os.path.join(d, "trace_bench.json")
if os.path.exists(tb):
    j = json.loads(open(tb).read().strip().splitlines()[-1])
    roof = j["roofline"]
    print(f"\n## dominant kernel under rocprofv3: bench.py (HIP events, timed steps) {roof['kernel']} avg {roof['kernel_avg_ms']:.3f} ms")
    import re
    m = re.match(r"sa_k_systolic<(\w+),G(\d+),K(\d+)>", roof["kernel"])
    mp = re.match(r"sa_k_systolic_pk_bundle<(\w+),(\d+),(\d+),(\w+)>", roof["kernel"])
    if m or mp:
        meth = {"nw": "0", "ga": "1", "sw": "2"}[(m or mp).group(1)]
        pat = (f"sa_k_systolic<{meth}, {m.group(2)}, {m.group(3)}, false>" if m else
               f"sa_k_systolic_pk_bundle<{meth}, {mp.group(2)}, {mp.group(3)}, {mp.group(4)}>")
        for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
            durs = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
                    for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
            durs = [x for _, x in sorted(durs)]
            if len(durs) > 1:
                print("   rocprofv3 dispatch durations (ms): " + "  ".join(f"{x:.2f}" for x in durs)
                      + f"   -> timed dispatches avg {sum(durs[1:]) / len(durs[1:]):.3f} ms (all {sum(durs) / len(durs):.3f})")

# per-dispatch counters, aggregated per kernel
def agg(sub, counters):
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == counters[0]:
                cnt[k] += 1
    return acc, cnt

for sub, counters in (("pmc_fetch", ["FETCH_SIZE"]), ("pmc_write", ["WRITE_SIZE"]),
                      ("pmc_sq", ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES",
                                  "SQ_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"])):
    acc, cnt = agg(sub, counters)
    if not acc:
        continue
    print(f"\n## rocprofv3 --pmc {' '.join(counters)}  (sum over dispatches; n = dispatches)")
    for k in sorted(acc, key=lambda k: -sum(acc[k].values()))[:10]:
        vals = "  ".join(f"{c}={acc[k].get(c, 0):.4g}" for c in counters)
        print(f"  {k:70s} n={cnt[k]:<4d} {vals}")
