#!/bin/bash
# copy the judged summaries of a tools/gpu_profile_round.sh run from gpurun_out/ into profiles/  (usage: save_profiles.sh <tag>)
T=${1:?tag}
for c in cfg2 cfg3 cfg4; do
  d=gpurun_out/prof_${T}_$c
  [ -d $d ] || continue
  cp $d/summary.txt profiles/${T}_${c}_summary.txt
  cp $d/trace_bench.json profiles/${T}_${c}_bench_under_rocprof.json
  cp $(ls $d/trace/*/*_kernel_stats.csv | head -1) profiles/${T}_${c}_kernel_stats.csv
  cp gpurun_out/$T/pmc_$c.txt profiles/${T}_${c}_pmc_sq.txt
done
python3 tools/make_traffic_json.py gpurun_out/prof_${T}_cfg2 $T cfg2
python3 tools/make_traffic_json.py gpurun_out/prof_${T}_cfg3 $T cfg3
python3 tools/make_traffic_json.py gpurun_out/prof_${T}_cfg4 $T cfg4
ls profiles | grep "^$T" | wc -l
