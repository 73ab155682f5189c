#!/bin/bash
# end to end through cli/seqalign with the options of BASELINE config 5 (-f 0.9 -z 6) on the first N sequences of the cfg 5 set:
# FASTA -> device filter -> NW -> N x N HDF5 deflated at level 6; tiles deflated by all cores (default) vs libhdf5's filter in
# the writing thread (SA_HOST_SERIAL_DEFLATE=1, what the reference's writer does).  usage: gpu_cli_z6.sh [N=30000]
N=${1:-30000}
MODES=${2:-parallel serial}   # (the full 100 000 sequences: parallel only -- the serial writer needs ~27 minutes)
df -h /tmp | tail -1
python - $N <<'PY'
import sys; sys.path.insert(0, ".")
from tests.synth import make_config
seqs, cfg = make_config("cfg5", int(sys.argv[1]))
open("/tmp/cfg5.fasta", "wb").write(b"".join(b">s%d\n" % k + s + b"\n" for k, s in enumerate(seqs)))
PY
nproc
for mode in $MODES; do
  rm -f /tmp/out_$mode.h5
  if [ $mode = serial ]; then export SA_HOST_SERIAL_DEFLATE=1; else unset SA_HOST_SERIAL_DEFLATE; fi
  echo "== $mode"
  time (cli/seqalign -i /tmp/cfg5.fasta -o /tmp/out_$mode.h5 -a nw -m blosum62 -p 4 -f 0.9 -z 6 -B -F -Q 2>&1 | grep -v amdgpu.ids)
  ls -la /tmp/out_$mode.h5
done
[ -f /tmp/out_serial.h5 ] && /opt/conda/bin/h5diff /tmp/out_parallel.h5 /tmp/out_serial.h5 && echo "h5diff: identical contents"
/opt/conda/bin/h5dump -H -p /tmp/out_parallel.h5 | grep -E "DATASPACE|CHUNKED|DEFLATE|SIZE" | head -8
rm -f /tmp/out_parallel.h5 /tmp/out_serial.h5 /tmp/cfg5.fasta
