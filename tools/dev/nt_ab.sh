#!/bin/bash
# development: rebuild the NW packed kernels with plain stores on the GPU box and compare WRITE_SIZE / time
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/nt_ab; mkdir -p $O; export TMPDIR=/tmp
cd sequencealigner_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DSA_PK_PLAIN_STORES -x hip -c csrc/sa_systolic_pk_nw.hip -o lib/sa_systolic_pk_nw.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC lib/sa_tables.o lib/sa_generic.o lib/sa_systolic.o lib/sa_systolic_nw.o lib/sa_systolic_ga.o lib/sa_systolic_sw.o lib/sa_systolic_pk_nw.o lib/sa_systolic_pk_ga.o lib/sa_systolic_pk_sw.o lib/sa_filter.o lib/sa_driver.o -o lib/libseqalign_hip.so || exit 1
cd $ROOT
echo "== plain stores"; tools/dev/write_size.sh | grep "8, 14"
timeout -k 10 250 python3 tools/dev/host_overhead.py | tail -2
