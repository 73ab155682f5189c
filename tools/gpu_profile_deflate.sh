#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 evidence for the device-side tile / DEFLATE path.
#  1. kernel-trace stats of cli/seqalign on BASELINE config 5 end to end (the overlapped walk in shells),
#  2. kernel-trace stats of the encoder alone (tools/dev/deflate_time.py 40000),
#  (3. HBM traffic counters of the encoder's kernels -- `--pmc FETCH_SIZE WRITE_SIZE` on deflate_time.py -- aborted inside
#      rocprofv3 on this pool, 'caught signal 6', and then sat until the silence guard: only with SA_PROF_PMC=1.)
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
[ -n "$SA_PROF_PMC" ] || exit 0
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $O/pmc -o pmc -- python3 $ROOT/tools/dev/deflate_time.py 20000 > $O/pmc_stdout.txt 2> $O/pmc.err || { tail -5 $O/pmc.err; exit 1; }
cd $ROOT
python3 - $O <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(float); n = defaultdict(set)
for f in glob.glob(os.path.join(sys.argv[1], "pmc", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "sa_k_" in name and "systolic" not in name:
            k = name.split("sa_k_")[1].split("(")[0]
            acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
print("# FETCH_SIZE / WRITE_SIZE per launch (KB as the counters report them; 5 x 5 tiles of 4096: 5 tiles = 335.5 MB of matrix per launch)")
for k in sorted(acc):
    print(k[0], k[1], f"{acc[k] / len(n[k[0]]):.6g} per launch over {len(n[k[0]])} launches")
PY
