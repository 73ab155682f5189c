#!/bin/bash
# development: cold wall time of `seqalign` on cfg 2 (FASTA -> N x N HDF5) under runtime knobs. run through gpurun.
python - <<'PY'
import sys; sys.path.insert(0, ".")
from tests.synth import make_config
seqs, cfg = make_config("cfg2")
open("/tmp/cfg2.fasta", "wb").write(b"".join(b">s%d\n" % k + s + b"\n" for k, s in enumerate(seqs)))
PY
run() {
  echo "== $*"
  for i in 1 2 3; do
    rm -f /tmp/out.h5
    ( time env "$@" SA_CLI_TIMES=1 cli/seqalign -i /tmp/cfg2.fasta -o /tmp/out.h5 -a nw -m blosum62 -p 4 -B -F -Q ) 2>&1 | grep -E "real|runtime up|page-locked|returned|HDF5 written" | tr '\n' ' '; echo
  done
}
run A=1
run HSA_ENABLE_SDMA=0
run -W_mode=1
echo "== -W (no output): what is left without the matrix"
for i in 1 2; do ( time cli/seqalign -i /tmp/cfg2.fasta -W -a nw -m blosum62 -p 4 -B -F -Q ) 2>&1 | grep -E "real" ; done
echo "== process that only loads the library and counts devices"
python3 - <<'PY'
import subprocess, time, ctypes, sys
t=time.time(); subprocess.run(["cli/seqalign","-l"],capture_output=True); print("seqalign -l (no runtime):", round(time.time()-t,3),"s")
PY
