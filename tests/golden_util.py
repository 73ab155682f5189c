"""Loading of tests/golden/*.npz (inputs + scores produced by the reference itself, tools/make_golden.py)."""
from __future__ import annotations

import json
import pathlib

import numpy as np

GOLDEN_DIR = pathlib.Path(__file__).resolve().parent / "golden"


def golden_cases() -> list[str]:
    return sorted(p.stem for p in GOLDEN_DIR.glob("*.npz") if not p.stem.startswith(("filter_", "digest_", "stripes_")))


def load_case(name: str):
    import sequencealigner_amd as sa
    z = np.load(GOLDEN_DIR / f"{name}.npz")
    params = json.loads(str(z["params"]))
    meta = np.ascontiguousarray(z["meta"], np.int32)
    store = sa.SequenceStore(blob=np.ascontiguousarray(z["blob"], np.uint8), meta=meta, num=int(meta.shape[0]),
                             max=int(meta[:, 1].max()))
    scoring = sa.Scoring.from_names(params["method"], params["matrix"], **params["gaps"])
    # the reference's own globals after ITS option parser ran must equal what ours derived
    stored = params["stored"]
    assert scoring.method_name == stored["method"]
    if scoring.method == 0:
        assert scoring.gap_pen == stored["gap_pen"]
    else:
        assert (scoring.gap_opn, scoring.gap_ext) == (stored["gap_opn"], stored["gap_ext"])
    full = z["expected_full"] if "expected_full" in z.files else None
    return store, scoring, np.asarray(z["expected"], np.int32), full


def tri_to_full(tri: np.ndarray, n: int) -> np.ndarray:
    """packed (pair i<j at j(j-1)/2+i) -> full symmetric, zero diagonal (reference io/output.c:76-83)."""
    full = np.zeros((n, n), np.int32)
    j, i = np.triu_indices(n, 1)[::-1]  # placeholder, replaced below
    jj = np.repeat(np.arange(n), np.arange(n))
    ii = np.concatenate([np.arange(k) for k in range(n)]) if n > 1 else np.zeros(0, int)
    full[ii, jj] = tri
    full[jj, ii] = tri
    return full
