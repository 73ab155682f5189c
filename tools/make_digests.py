#!/usr/bin/env python3
"""Full-size reference digests: runs the REFERENCE ITSELF (oracle/_ref/libseqalign_ref.so, its unmodified sources)
over a whole bench workload in the build container and commits per-column (sum, xor, crc32) triples of the packed
score matrix under tests/golden/digest_<name>.npz (tests/digest_util.py).  Minutes of CPU per workload.

    python tools/make_digests.py cfg2 cfg3 cfg4:12000 [--threads 6]
"""
import json
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

import sequencealigner_amd as sa  # noqa: E402  (host-side tables only; no GPU needed)
from tests.digest_util import column_digests  # noqa: E402
from tests.oracle_binding import RefLib  # noqa: E402
from tests.synth import make_config  # noqa: E402


def main():
    threads = 0
    names = []
    args = sys.argv[1:]
    while args:
        a = args.pop(0)
        if a == "--threads":
            threads = int(args.pop(0))
        else:
            names.append(a)
    for spec in names:
        name, _, n = spec.partition(":")
        n = int(n) if n else None
        seqs, cfg = make_config(name, n)
        store = sa.SequenceStore.from_sequences(seqs)
        ref = RefLib(cfg["method"], cfg["matrix"], threads=threads, **cfg["gaps"])
        t0 = time.perf_counter()
        packed = ref.align(store, triangular=True)
        dt = time.perf_counter() - t0
        ref.close()
        d = column_digests(packed, store.num)
        tag = name if n is None else f"{name}_n{n}"
        meta = dict(config=name, n=store.num, method=cfg["method"], matrix=cfg["matrix"], gaps=cfg["gaps"],
                    pairs=int(packed.size), total_sum=int(packed.sum(dtype=np.int64)), source="oracle/_ref (reference sources)",
                    reference_seconds=dt, reference_threads=threads)
        np.savez_compressed(ROOT / "tests" / "golden" / f"digest_{tag}.npz", params=np.array(json.dumps(meta)), **d)
        print(f"digest_{tag}: {store.num} seqs, {packed.size} pairs, reference took {dt:.1f} s", flush=True)


if __name__ == "__main__":
    main()
