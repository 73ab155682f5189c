"""development: where the mixed-length workload (8000 x U[20,190]) loses against cfg 2 -- TCUPS of length sub-ranges (one bundle
each), of the whole mix with the bundles side by side and one after the other.  usage: mixed_probe.py [method]"""
import os, sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import sequencealigner_amd as sa
from tests.synth import make_protein_set

method = sys.argv[1] if len(sys.argv) > 1 else "nw"
gaps = dict(gap_pen=4) if method == "nw" else dict(gap_open=10, gap_extend=1)
sc = sa.Scoring.from_names(method, "blosum62", **gaps)
s = torch.cuda.current_stream().cuda_stream


def run(tag, seqs, env=None):
    for k, v in (env or {}).items():
        os.environ[k] = v
    try:
        store = sa.SequenceStore.from_sequences(seqs)
        ctx = sa.Context(store, sc, 0)
    finally:
        for k in (env or {}):
            del os.environ[k]
    out = torch.empty(store.pairs, dtype=torch.int32, device="cuda")
    for _ in range(3):
        ctx.align_range(0, store.pairs, out.data_ptr(), s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        ctx.align_range(0, store.pairs, out.data_ptr(), s)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 8
    print(f"{tag:44s} {store.num:6d} seqs {t * 1e3:8.3f} ms  {store.cells() / t / 1e12:6.2f} TCUPS", flush=True)
    ctx.close()
    return t


run("cfg2-like U[80,120] x 8000", make_protein_set(8000, 80, 120, 2))
run("U[20,190] x 8000 (three bundles one after the other)", make_protein_set(8000, 20, 190, 7))
run("U[20,190] x 8000, bundles side by side", make_protein_set(8000, 20, 190, 7), {"SA_HIP_CONCURRENT_CLASSES": "1"})
run("U[20,190] x 8000, store order (no arrangement)", make_protein_set(8000, 20, 190, 7), {"SA_HIP_NO_SORT": "1"})
run("U[20,64] x 8000   (K 3..8)", make_protein_set(8000, 20, 64, 7))
run("U[65,128] x 8000  (K 9..16)", make_protein_set(8000, 65, 128, 7))
run("U[129,190] x 8000 (K 17..24)", make_protein_set(8000, 129, 190, 7))
run("U[100,100] x 8000 (one class, all rounds pure)", make_protein_set(8000, 100, 100, 7))
run("U[57,64] x 8000   (K 8 only)", make_protein_set(8000, 57, 64, 7))
run("U[185,192] x 8000 (K 24 only)", make_protein_set(8000, 185, 192, 7))
# two bundles (cfg 4 / cfg 5 length ranges): side by side vs one after the other
for tag, lo, hi in (("U[120,180] x 12000 (cfg 4 lengths, K 15..23)", 120, 180), ("U[96,144] x 12000 (cfg 5 lengths, K 12..18)", 96, 144)):
    run(tag + " one after the other", make_protein_set(12000, lo, hi, 4))
    run(tag + " side by side", make_protein_set(12000, lo, hi, 4), {"SA_HIP_CONCURRENT_CLASSES": "1"})
