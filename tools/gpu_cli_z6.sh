#!/bin/bash
# end to end through cli/seqalign with the options of BASELINE config 5 (-f 0.9 -z 6) on the first N sequences of the cfg 5 set:
# FASTA -> device filter -> NW -> N x N HDF5 deflated: device = tiles deflated on the GPU (the default, csrc/sa_deflate.hip),
# parallel = zlib level 6 on all cores (SA_HOST_CPU_DEFLATE=1), serial = libhdf5's filter in the writing thread
# (SA_HOST_SERIAL_DEFLATE=1, what the reference's writer does).  usage: gpu_cli_z6.sh [N=30000] ["device parallel serial"]
N=${1:-30000}
MODES=${2:-device parallel serial}   # (the full 100 000 sequences: not serial -- that writer needs ~27 minutes)
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
  unset SA_HOST_SERIAL_DEFLATE SA_HOST_CPU_DEFLATE
  [ $mode = serial ] && export SA_HOST_SERIAL_DEFLATE=1
  [ $mode = parallel ] && export SA_HOST_CPU_DEFLATE=1
  echo "== $mode"
  time (cli/seqalign -i /tmp/cfg5.fasta -o /tmp/out_$mode.h5 -a nw -m blosum62 -p 4 -f 0.9 -z 6 -B -F -V 2>&1 | grep -v '^Aligning' | grep -v amdgpu.ids)
  ls -la /tmp/out_$mode.h5
done
first=""
for mode in $MODES; do
  [ -z "$first" ] && first=$mode && continue
  /opt/conda/bin/h5diff /tmp/out_$first.h5 /tmp/out_$mode.h5 && echo "h5diff $first vs $mode: identical contents"
done
/opt/conda/bin/h5dump -H -p /tmp/out_$first.h5 | grep -E "DATASPACE|CHUNKED|DEFLATE|SIZE" | head -8
rm -f /tmp/out_device.h5 /tmp/out_parallel.h5 /tmp/out_serial.h5 /tmp/cfg5.fasta
