#!/bin/bash
# BASELINE config 4 end to end through cli/seqalign on one MI355X box: 50 000 DNA reads, SW / NUC.4.4, FASTA -> N x N HDF5 without -z.
# device = the tiles come from the device as HDF5 chunks (the default on one device), host = sa_hip_align into a host matrix +
# H5Dwrite (SA_HOST_MATRIX=1, the reference's flow).  usage: gpu_cli_cfg4.sh [N=50000] ["device host"]
N=${1:-50000}
MODES=${2:-device host}
df -h /tmp | tail -1
python - $N <<'PY'
import sys; sys.path.insert(0, ".")
from tests.synth import make_config
seqs, cfg = make_config("cfg4", int(sys.argv[1]))
open("/tmp/cfg4.fasta", "wb").write(b"".join(b">s%d\n" % k + s + b"\n" for k, s in enumerate(seqs)))
PY
for mode in $MODES; do
  rm -f /tmp/out_$mode.h5
  unset SA_HOST_MATRIX
  [ $mode = host ] && export SA_HOST_MATRIX=1
  echo "== $mode"
  time (cli/seqalign -i /tmp/cfg4.fasta -o /tmp/out_$mode.h5 -a sw -m nuc44 -s 10 -e 1 -B -F -V 2>&1 | grep -v '^Aligning' | grep -v amdgpu.ids)
  ls -la /tmp/out_$mode.h5
done
first=""
for mode in $MODES; do
  [ -z "$first" ] && first=$mode && continue
  /opt/conda/bin/h5diff /tmp/out_$first.h5 /tmp/out_$mode.h5 && echo "h5diff $first vs $mode: identical contents"
done
rm -f /tmp/out_device.h5 /tmp/out_host.h5 /tmp/cfg4.fasta
