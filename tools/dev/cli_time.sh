#!/bin/bash
# development: where the wall time of a cold `seqalign` run goes (cfg 2, FASTA -> N x N HDF5). run through gpurun.
python - <<'PY'
import sys; sys.path.insert(0, ".")
from tests.synth import make_config
seqs, cfg = make_config("cfg2")
open("/tmp/cfg2.fasta", "wb").write(b"".join(b">s%d\n" % k + s + b"\n" for k, s in enumerate(seqs)))
PY
for i in 1 2 3; do
  rm -f /tmp/out.h5
  time env SA_HIP_VERBOSE=1 SA_CLI_TIMES=1 cli/seqalign -i /tmp/cfg2.fasta -o /tmp/out.h5 -a nw -m blosum62 -p 4 -B -F -Q 2>&1 | grep -v amdgpu.ids
done
