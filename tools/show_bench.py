#!/usr/bin/env python3
"""Print the interesting numbers of a bench.py JSON line (development helper)."""
import json, sys
for path in sys.argv[1:]:
    j = json.loads(open(path).read().strip().splitlines()[-1])
    r = j["roofline"]
    print(f"{path}: {j['value']:.4g} pairs/s  {j['ms_per_step']:.2f} ms/step  {j['gcups']:.0f} GCUPS | dominant {r['kernel']} "
          f"{r['kernel_avg_ms']:.2f} ms x{r['launches']} (rank {j['valu'].get('gcups_this_rank', 0):.0f} GCUPS), sum of kernels {r.get('sum_of_kernel_ms_per_step', 0):.2f} ms/step"
          + (f" | host boundary {j['host_boundary']['seconds']*1e3:.0f} ms" if 'host_boundary' in j else ""))
