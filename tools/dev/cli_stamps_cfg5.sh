#!/bin/bash
# stage stamps (SA_CLI_TIMES) of cli/seqalign on BASELINE config 5 end to end: where the cold process spends its time
python - <<'PY'
import sys; sys.path.insert(0, ".")
from tests.synth import make_config
seqs, cfg = make_config("cfg5", 100000)
open("/tmp/cfg5.fasta", "wb").write(b"".join(b">s%d\n" % k + s + b"\n" for k, s in enumerate(seqs)))
PY
export SA_CLI_TIMES=1
TIMEFORMAT="real %R s"
time (cli/seqalign -i /tmp/cfg5.fasta -o /tmp/o.h5 -a nw -m blosum62 -p 4 -f 0.9 -z 6 -B -F -Q 2>&1 | grep -v "^Aligning")
rm -f /tmp/o.h5 /tmp/cfg5.fasta
