"""profiles/<run>_<kernel>_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_gpu.sh.

usage: python tools/make_traffic_json.py gpurun_out/prof_<run> <run> [workload]
The bench line of the same profiling run (<dir>/trace_bench.json) supplies what a later bench.py run must match before
it quotes the number: the kernel's name with the classes it walked, pairs per launch and the sequence count.
FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 reports half of the coalesced read bytes; an upper estimate for
the byte-granular loads here), WRITE_SIZE is taken as is; both are KB summed over the dispatches of a kernel."""
import csv, glob, json, os, re, sys
from collections import defaultdict

d, run = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "cfg2"
METH = {"0": "nw", "1": "ga", "2": "sw"}
bench = json.loads(open(os.path.join(d, "trace_bench.json")).read().strip().splitlines()[-1])
roof = bench["roofline"]
nseq = int(re.search(r": (\d+) ", bench["config"]["workload"]).group(1))


def collect(sub):
    acc, n = defaultdict(float), defaultdict(set)
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"sa_k_systolic<(\d), (\d+), (\d+), (true|false)>", r["Kernel_Name"])
            mp = re.search(r"sa_k_systolic_pk_bundle<(\d), (\d+), (\d+), (true|false)>", r["Kernel_Name"])
            if mp:
                name = f"sa_k_systolic_pk_bundle<{METH[mp.group(1)]},{mp.group(2)},{mp.group(3)},{mp.group(4)}>"
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
    if not roof["kernel"].startswith(k):  # only the kernel the bench line of this run describes
        continue
    out = {"kernel": roof["kernel"], "run": run, "workload": workload, "n_sequences": nseq,
           "pairs_per_launch": roof["pairs_per_launch"], "dispatches": nf[k],
           "FETCH_SIZE_kb_per_launch": round(f_kb, 1), "WRITE_SIZE_kb_per_launch": round(w_kb, 1),
           "note": "separate --pmc passes (tools/profile_gpu.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports "
                   "half of coalesced read bytes; the byte-granular loads here are uncalibrated, so this is an upper "
                   "estimate), WRITE_SIZE as is",
           "traffic_bytes_per_launch": (2 * f_kb + w_kb) * 1024}
    tag = ("pk_" if "_pk" in k else "") + re.sub(r"[^A-Za-z0-9]+", "_", roof["kernel"].split("<")[1]).strip("_")
    path = os.path.join("profiles", f"{run}_{tag}_traffic.json")
    json.dump(out, open(path, "w"), indent=1)
    print(path, out["traffic_bytes_per_launch"])
