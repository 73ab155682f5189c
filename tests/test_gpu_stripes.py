"""GPU (-m gpu): BASELINE.json configs 4 and 5 AT THEIR FULL SIZE against fixtures the REFERENCE produced, through the
benchmarked path.

tests/golden/stripes_cfg{4,5}.npz (tools/make_stripe_digests.py, oracle/_ref = the reference's unmodified sources) hold
per-column (sum, xor, crc32) digests of column stripes of the full matrices -- cfg 4: the last 256 columns of the
50 000-read SW matrix; cfg 5: the reference's own `-f 0.9` keep mask over the 100 000 sequences
(src/bio/filter.c:14-89, one thread), then on the 89 994 kept sequences the last 256 columns and the 128 columns
straddling packed index 2^31.  Here the whole workload is delivered by sa_ctx_align_host into a page-locked packed HOST
matrix (the non-temporal row-order stores bench.py and the CLI use; 5 GB / 16 GB) and those columns must match digest
for digest; sa_hip_filter's keep mask must equal the reference's bit for bit."""
import hashlib
import json
import pathlib
import zlib

import numpy as np
import pytest

from tests.synth import make_config

pytestmark = pytest.mark.gpu

GOLDEN = pathlib.Path(__file__).resolve().parent / "golden"


def tri(j: int) -> int:
    return j * (j - 1) // 2


def _check_columns(packed: np.ndarray, z, what: str) -> int:
    bad = []
    pairs = 0
    for k, j in enumerate(z["cols"].tolist()):
        col = packed[tri(j):tri(j) + j]
        got = (int(col.sum(dtype=np.int64)), int(np.bitwise_xor.reduce(col)), zlib.crc32(np.ascontiguousarray(col, "<i4").tobytes()))
        want = (int(z["sum"][k]), int(z["xor"][k]), int(z["crc32"][k]))
        if got != want:
            bad.append((j, got, want))
        pairs += j
    assert not bad, f"{what}: {len(bad)} of {len(z['cols'])} pinned columns differ from the reference, first: {bad[0]}"
    return pairs


@pytest.mark.parametrize("name", ["cfg4", "cfg5"])
def test_full_size_host_delivered_columns_match_the_reference(name, sa):
    z = np.load(GOLDEN / f"stripes_{name}.npz")
    meta = json.loads(str(z["params"]))
    seqs, cfg = make_config(name)
    assert len(seqs) == meta["n_input"]
    if name == "cfg5":
        keep = sa.hip_filter(sa.SequenceStore.from_sequences(seqs), 0.9)  # the relation on the device, 100 000 sequences
        want_keep = np.unpackbits(z["keep_packed"])[:len(seqs)].astype(bool)
        assert int(keep.sum()) == meta["kept"], f"sa_hip_filter keeps {int(keep.sum())}, the reference {meta['kept']}"
        assert hashlib.sha256(keep.astype(np.uint8).tobytes()).hexdigest() == meta["keep_sha256"]
        assert np.array_equal(keep, want_keep)
        seqs = [s for s, k in zip(seqs, keep) if k]
    store = sa.SequenceStore.from_sequences(seqs)
    assert store.num == meta["n"] and store.pairs == meta["pairs"]
    scoring = sa.Scoring.from_names(cfg["method"], cfg["matrix"], **cfg["gaps"])
    dest = sa.PinnedMatrix(store.pairs)  # 5.0 GB / 16.2 GB of page-locked host memory: the kernels store straight into it
    try:
        with sa.Context(store, scoring, 0) as ctx:
            seconds = ctx.align_host(dest.array, triangular=True)
        assert seconds > 0
        pinned = _check_columns(dest.array, z, f"{name} host-delivered, full size")
        assert pinned > 8_000_000
        if name == "cfg5":  # the stripe that straddles packed index 2^31 really does
            cols = z["cols"].tolist()
            assert any(tri(j) < (1 << 31) <= tri(j + 1) for j in cols)
    finally:
        dest.close()
