"""Randomised differential run against the oracle (development helper, run through gpurun):
python tools/gpu_fuzz.py <seconds> [seed]   -- prints one line per failing case and a summary."""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
import sequencealigner_amd as sa
from tests.oracle_binding import Oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
o = Oracle()
AA = np.frombuffer(b"ARNDCQEGHILKMFPSTWYVBZX", dtype=np.uint8)
NT = np.frombuffer(b"ACGTN", dtype=np.uint8)
MATS = ["blosum62", "blosum45", "blosum80", "blosum100", "blosum30", "pam30", "pam70", "pam250", "pam500"]
t0, cases, bad, generic = time.time(), 0, 0, 0
last_note = t0
while time.time() - t0 < budget:
    method = ["nw", "ga", "sw"][int(rng.integers(0, 3))]
    dna = rng.random() < 0.2
    matrix = "nuc44" if dna else MATS[int(rng.integers(0, len(MATS)))]
    if method == "nw":
        gaps = dict(gap_pen=int(rng.choice([0, 1, 2, 3, 4, 5, 8, 12, 20, 40, 57, 60, 100])))
    else:
        gaps = dict(gap_open=int(rng.choice([0, 1, 2, 5, 10, 11, 12, 16, 25, 60, 120])), gap_extend=int(rng.choice([0, 1, 2, 3, 5, 10, 20])))
    n = int(rng.integers(2, 400))
    regime = int(rng.integers(0, 8))
    if regime == 0:
        lens = rng.integers(1, 10, n)
    elif regime == 1:
        lens = rng.integers(40, 400, n)
    elif regime == 2:
        lens = rng.integers(1, 600, min(n, 120))
    elif regime == 3:
        lens = np.where(rng.random(n) < 0.5, 1, rng.integers(100, 140, n))
    elif regime == 4:
        lens = np.full(n, int(rng.integers(1, 260)))
    elif regime == 7:  # the wide 16-lane classes (columns 641..1024) with rows long enough that they really run packed
        nn = min(n, 48)
        lens = np.where(rng.random(nn) < 0.5, rng.integers(16, 130, nn), rng.integers(600, 1031, nn))
    elif regime == 6:  # the frame budget follows the shortest sequence: short rows of a chosen length against wide columns
        m = int(rng.choice([2, 3, 5, 8, 14, 15, 16, 31, 63]))
        nn = min(n, 150)
        lens = np.where(rng.random(nn) < 0.5, m + rng.integers(0, 3, nn), rng.integers(150, 660, nn))
    else:
        lens = np.where(rng.random(min(n, 80)) < 0.1, rng.integers(1000, 2300, min(n, 80)), rng.integers(1, 150, min(n, 80)))
    alpha = NT if dna else AA
    seqs = [alpha[rng.integers(0, len(alpha), int(l))].tobytes() for l in lens]
    if rng.random() < 0.15:  # near-duplicates and exact duplicates
        seqs = [seqs[int(rng.integers(0, len(seqs)))] if rng.random() < 0.5 else s for s in seqs]
    try:
        sc = sa.Scoring.from_names(method, matrix, **gaps)
    except Exception:
        continue
    store = sa.SequenceStore.from_sequences(seqs)
    want = o.align(store, sc, triangular=True)
    got = sa.hip_align(store, sc, triangular=True)
    cases += 1
    if time.time() - last_note > 30:
        last_note = time.time()
        print(f"... {cases} cases, {bad} mismatching, {time.time() - t0:.0f} s", flush=True)
    if rng.random() < 0.3 and store.pairs >= 8:  # the multi-GPU data path on the same case: shares of a random world, placed
        import torch
        world, to_host = int(rng.choice([2, 3, 5, 8])), bool(rng.random() < 0.5)
        with sa.Context(store, sc, 0) as ctx:
            lo = int(rng.integers(0, store.pairs // 2)) if rng.random() < 0.3 else 0
            cnt = store.pairs - lo
            use16 = bool(ctx.scores_fit16 and rng.random() < 0.5)
            e = ctx.share_elems(lo, cnt, world, to_host)
            shares = torch.zeros(world * e, dtype=torch.int16 if use16 else torch.int32, device="cuda")
            packed = torch.full((cnt,), -7, dtype=torch.int32, device="cuda")
            host = sa.PinnedMatrix(store.pairs) if to_host else None
            st = torch.cuda.current_stream().cuda_stream
            for r in range(world):
                ctx.align_share(lo, cnt, world, r, shares.data_ptr() + (2 if use16 else 4) * r * e, use16, st, host.ptr if to_host else 0)
            ctx.place_shares(lo, cnt, world, shares.data_ptr(), use16, packed.data_ptr(), st, to_host)
            torch.cuda.synchronize()
            got_sh = packed.cpu().numpy()
            ok_sh = np.array_equal(got_sh, want[lo:]) and (not to_host or np.array_equal(host.array[lo:], want[lo:]))
            if host is not None:
                host.close()
        if not ok_sh:
            bad += 1
            print(f"MISMATCH (shares) case {cases}: {method} {matrix} {gaps} regime {regime} n={len(seqs)} world {world} to_host {to_host} lo {lo} use16 {use16}", flush=True)
    if not np.array_equal(got, want):
        bad += 1
        k = int(np.nonzero(got != want)[0][0])
        print(f"MISMATCH case {cases}: {method} {matrix} {gaps} regime {regime} n={len(seqs)} first at {k}: got {got[k]} want {want[k]}", flush=True)
print(f"fuzz: {cases} cases, {bad} mismatching, {time.time() - t0:.0f} s, seed {seed}")
sys.exit(1 if bad else 0)
