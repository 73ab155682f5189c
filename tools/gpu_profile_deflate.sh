#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 evidence for the device-side tile / DEFLATE path.
#  1. kernel-trace stats of cli/seqalign on BASELINE config 5 end to end (the overlapped walk in shells),
#  2. kernel-trace stats of the encoder alone (tools/dev/deflate_time.py 40000),
#  3. HBM traffic counters of the encoder's kernels (FETCH_SIZE, WRITE_SIZE: a pass each), per launch.
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/deflate_prof; mkdir -p $O; export TMPDIR=/tmp
python3 - <<'PY'
import sys; sys.path.insert(0, ".")
from tests.synth import make_config
seqs, cfg = make_config("cfg5", 100000)
open("/tmp/cfg5.fasta", "wb").write(b"".join(b">s%d\n" % k + s + b"\n" for k, s in enumerate(seqs)))
PY
cd /tmp
export SA_CLI_CLEAN_EXIT=1 # (the tool leaves through _exit: the profiler writes its files from an exit handler)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cli -o cli -- $ROOT/cli/seqalign -i /tmp/cfg5.fasta -o /tmp/prof_out.h5 -a nw -m blosum62 -p 4 -f 0.9 -z 6 -B -F -Q > $O/cli_stdout.txt 2> $O/cli.err || { tail -5 $O/cli.err; exit 1; }
rm -f /tmp/prof_out.h5 /tmp/cfg5.fasta
rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc -o enc -- python3 $ROOT/tools/dev/deflate_time.py 40000 > $O/enc_stdout.txt 2> $O/enc.err || { tail -5 $O/enc.err; exit 1; }
# HBM traffic of the encoder's kernels: FETCH_SIZE and WRITE_SIZE in passes of their OWN (together they exceed what the
# hardware collects at once: "error code 38"), on score-like random values (no alignment kernels under the profiler)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python3 $ROOT/tools/dev/deflate_pmc.py 16384 > $O/pmc_${c}_stdout.txt 2> $O/pmc_$c.err || { tail -5 $O/pmc_$c.err; exit 1; }
done
cd $ROOT
python3 - $O <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(float); n = defaultdict(set)
for f in glob.glob(os.path.join(sys.argv[1], "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "sa_k_" in name:
            k = name.split("sa_k_")[1].split("(")[0]
            acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
print("# FETCH_SIZE / WRITE_SIZE per launch as the counters report them (KB); N = 16384: a launch = one tile row = 4 tiles = 268.4 MB of matrix")
for k in sorted(acc):
    print(k[0], k[1], f"{acc[k] / len(n[k]):.6g} per launch over {len(n[k])} launches")
PY
