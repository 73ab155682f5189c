#!/bin/bash
# timing-only ablations of the systolic kernel's per-step overhead (results are WRONG by construction)
for abl in 8 10 9 11; do
  SA_EXTRA_HIPCC_FLAGS="-DSA_ABL=$abl" python3 -c "import __graft_entry__ as g; g.build_hip(force=True)" > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "ABL=$abl $(SA_HIP_STAMPS=1 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --n 6000 2>&1 | grep 'stamps.*K14' | tail -1)"
done
