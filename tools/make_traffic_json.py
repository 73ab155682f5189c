"""profiles/<run>_<kernel>_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_gpu.sh.

usage: python tools/make_traffic_json.py gpurun_out/prof_<run> <run> [workload [n_sequences]]
FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 reports half of the coalesced read bytes; an upper estimate for
the byte-granular loads here), WRITE_SIZE is taken as is; both are KB summed over the dispatches of a kernel."""
import csv, glob, json, os, re, sys
from collections import defaultdict

d, run = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "cfg2"
nseq = int(sys.argv[4]) if len(sys.argv) > 4 else None  # sequence count when the workload was run at a reduced size
METH = {"0": "nw", "1": "ga", "2": "sw"}


def collect(sub):
    acc, n = defaultdict(float), defaultdict(set)
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"sa_k_systolic<(\d), (\d+), (\d+), (true|false)>", r["Kernel_Name"])
            mp = re.search(r"sa_k_systolic_pk<(\d), (\d+), (\d+)[,>]", r["Kernel_Name"])
            if mp:
                name = f"sa_k_systolic_pk{'16' if mp.group(2) == '16' else ''}<{METH[mp.group(1)]},K{mp.group(3)}>"
            elif m:
                name = f"sa_k_systolic<{METH[m.group(1)]},G{m.group(2)},K{m.group(3)}>" + (" strips" if m.group(4) == "true" else "")
            else:
                continue
            acc[name] += float(r["Counter_Value"])
            n[name].add(r["Dispatch_Id"])
    return acc, {k: len(v) for k, v in n.items()}


fetch, nf = collect("pmc_fetch")
write, nw = collect("pmc_write")
for k in sorted(fetch):
    if k not in write or nf[k] != nw[k]:
        continue
    f_kb, w_kb = fetch[k] / nf[k], write[k] / nw[k]
    out = {"kernel": k, "run": run, "workload": workload, **({"n_sequences": nseq} if nseq else {}), "dispatches": nf[k],
           "FETCH_SIZE_kb_per_launch": round(f_kb, 1), "WRITE_SIZE_kb_per_launch": round(w_kb, 1),
           "note": "separate --pmc passes (tools/profile_gpu.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports "
                   "half of coalesced read bytes; the byte-granular loads here are uncalibrated, so this is an upper "
                   "estimate), WRITE_SIZE as is",
           "traffic_bytes_per_launch": (2 * f_kb + w_kb) * 1024}
    tag = ("pk_" if "_pk<" in k else "") + re.sub(r"[^A-Za-z0-9]+", "_", k.split("<")[1]).strip("_")
    path = os.path.join("profiles", f"{run}_{tag}_traffic.json")
    json.dump(out, open(path, "w"), indent=1)
    print(path, out["traffic_bytes_per_launch"])
