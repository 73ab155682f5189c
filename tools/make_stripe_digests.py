#!/usr/bin/env python3
"""Reference-held fixtures for the two 8-GPU configs AT THEIR FULL SIZE (BASELINE.json configs 4 and 5).

Running the reference's CPU path over all 1.25e9 / 4e9 pairs would take hours here, so the fixtures pin column
STRIPES of the full-size matrices, computed by the reference itself (oracle/_ref/libseqalign_ref.so: its unmodified
sources; ref_columns() in oracle/ref_shim.c calls ALIGN->method per pair the way src/bio/align.c:44-58 does):

  cfg4  50 000 DNA reads, SW / NUC.4.4 / 10 / 1      : the last 256 columns (12.8 M pairs)
  cfg5  100 000 proteins, NW / BLOSUM62 / 4, -f 0.9  : the reference's OWN filter (src/bio/filter.c:14-89, run with one
        thread: its threaded version is racy) on all 100 000 sequences -> keep mask; then, on the kept
        store, the last 256 columns and 128 columns straddling every power-of-two packed index 2^31, 2^32 the
        matrix reaches (the places where 32-bit index arithmetic would break).

Per listed column the triple (sum int64, xor int32, crc32) of tests/digest_util.py.  Output:
tests/golden/stripes_cfg4.npz, tests/golden/stripes_cfg5.npz.

    python tools/make_stripe_digests.py cfg4 cfg5 [--threads 6]
"""
import hashlib
import json
import pathlib
import sys
import time
import zlib

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

import ctypes as C  # noqa: E402

import sequencealigner_amd as sa  # noqa: E402  (host-side tables only; no GPU needed)
from tests.oracle_binding import RefLib  # noqa: E402
from tests.synth import make_config  # noqa: E402


def tri(j: int) -> int:
    return j * (j - 1) // 2


def stripe_columns(n: int) -> list[tuple[int, int]]:
    """[j_lo, j_hi) ranges: the last 256 columns and 128 columns around every 2^31 / 2^32 crossing"""
    out = []
    for p in (1 << 31, 1 << 32):
        j = int((1 + (1 + 8 * p) ** 0.5) / 2)
        while tri(j) > p:
            j -= 1
        while tri(j + 1) <= p:
            j += 1
        if j + 65 <= n:  # column j holds packed index p
            out.append((j - 63, j + 65))
    out.append((n - 256, n))
    return out


def ref_stripe(ref: RefLib, store, j_lo: int, j_hi: int) -> np.ndarray:
    ref.lib.ref_columns.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    out = np.zeros(tri(j_hi) - tri(j_lo), np.int32)
    rc = ref.lib.ref_columns(store.blob.ctypes.data, store.meta.ctypes.data, store.num, store.max, j_lo, j_hi, out.ctypes.data)
    assert rc == 0
    return out


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
    for name in names:
        seqs, cfg = make_config(name)
        extra = {}
        meta = dict(config=name, n_input=len(seqs), method=cfg["method"], matrix=cfg["matrix"], gaps=cfg["gaps"],
                    source="oracle/_ref (reference sources): ref_columns / ref_filter")
        if name == "cfg5":
            t0 = time.perf_counter()
            fref = RefLib(cfg["method"], cfg["matrix"], threads=1, filter_threshold=0.9, **cfg["gaps"])
            kept = fref.filter(sa.SequenceStore.from_sequences(seqs))
            fref.close()
            mask = np.zeros(len(seqs), np.uint8)
            k = 0
            for idx, s in enumerate(seqs):  # kept is a subsequence of seqs, in order
                if k < len(kept) and kept[k] == s:
                    mask[idx] = 1
                    k += 1
            assert k == len(kept)
            extra["keep_packed"] = np.packbits(mask)
            meta.update(filter_threshold=0.9, kept=len(kept), keep_sha256=hashlib.sha256(mask.tobytes()).hexdigest(),
                        filter_seconds=time.perf_counter() - t0)
            print(f"{name}: reference filter kept {len(kept)} of {len(seqs)} in {meta['filter_seconds']:.0f} s", flush=True)
            seqs = kept
        store = sa.SequenceStore.from_sequences(seqs)
        n = store.num
        ref = RefLib(cfg["method"], cfg["matrix"], threads=threads, **cfg["gaps"])
        cols, sums, xors, crcs = [], [], [], []
        t0 = time.perf_counter()
        for j_lo, j_hi in stripe_columns(n):
            t1 = time.perf_counter()
            scores = ref_stripe(ref, store, j_lo, j_hi)
            for j in range(j_lo, j_hi):
                col = scores[tri(j) - tri(j_lo):tri(j + 1) - tri(j_lo)]
                assert col.size == j
                cols.append(j)
                sums.append(int(col.sum(dtype=np.int64)))
                xors.append(int(np.bitwise_xor.reduce(col)))
                crcs.append(zlib.crc32(np.ascontiguousarray(col, "<i4").tobytes()))
            print(f"{name}: columns [{j_lo}, {j_hi}) = {scores.size} pairs in {time.perf_counter() - t1:.0f} s", flush=True)
        ref.close()
        meta.update(n=n, pairs=tri(n), stripes=stripe_columns(n), reference_seconds=time.perf_counter() - t0, reference_threads=threads)
        np.savez_compressed(ROOT / "tests" / "golden" / f"stripes_{name}.npz", params=np.array(json.dumps(meta)),
                            cols=np.array(cols, np.int32), sum=np.array(sums, np.int64), xor=np.array(xors, np.int64).astype(np.int32),
                            crc32=np.array(crcs, np.uint32), **extra)
        print(f"stripes_{name}: {n} sequences, {len(cols)} columns pinned", flush=True)


if __name__ == "__main__":
    main()
