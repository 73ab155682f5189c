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


@pytest.mark.parametrize("name", ["cfg4", "cfg5"])
def test_full_size_tool_output_matches_the_reference(name, tmp_path):
    """the same fixtures through the PRODUCT: cli/seqalign on the whole config -- cfg 5 with its options `-f 0.9 -z 6` -- writes
    the N x N HDF5 file (device filter, alignment column block by column block, the tiles of every shell deflated / tiled on the
    device and written meanwhile: DESIGN.md 4.8), and the pinned columns are read back out of that file through libhdf5
    (h5dump: its inflate for cfg 5) -- column j of the packed matrix is row j of the file up to the diagonal."""
    import os
    import subprocess
    from tests.host_binding import H5DUMP, ROOT
    z = np.load(GOLDEN / f"stripes_{name}.npz")
    meta = json.loads(str(z["params"]))
    seqs, cfg = make_config(name)
    import shutil
    base = pathlib.Path(os.environ.get("SA_TEST_SCRATCH", "/tmp"))
    if shutil.disk_usage(base).free < 30 * 2**30:
        pytest.skip(f"needs 30 GB of scratch space under {base} for the 11 GB HDF5 file and its read-back")
    scratch = base / f"sa_tool_{name}_{os.getpid()}"
    scratch.mkdir(parents=True, exist_ok=True)
    fasta, out = scratch / "in.fasta", scratch / "out.h5"
    try:
        fasta.write_bytes(b"".join(b">s%d\n" % k + s + b"\n" for k, s in enumerate(seqs)))
        gaps = cfg["gaps"]
        flags = ["-p", gaps["gap_pen"]] if "gap_pen" in gaps else ["-s", gaps["gap_open"], "-e", gaps["gap_extend"]]
        if name == "cfg5":
            flags += ["-f", "0.9", "-z", "6"]
        res = subprocess.run([str(ROOT / "cli" / "seqalign"), "-i", str(fasta), "-o", str(out), "-a", cfg["method"], "-m", cfg["matrix"],
                              *map(str, flags), "-B", "-F", "-Q"], capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, res.stdout + res.stderr
        assert "tiles" in res.stdout and "on the device" in res.stdout or "delivered as HDF5 chunks" in res.stdout, res.stdout
        n = meta["n"]
        cols = z["cols"].tolist()
        # the stored chunks of the tile rows that hold the pinned columns (tests/host_c/chunk_read.c: H5Dread_chunk), inflated here
        reader = scratch / "chunk_read"
        subprocess.check_call(["gcc", "-std=c11", "-O2", "-I/opt/conda/include", str(ROOT / "tests" / "host_c" / "chunk_read.c"), "-o", str(reader),
                               "-L/opt/conda/lib", "-lhdf5", "-Wl,-rpath,/opt/conda/lib"])
        bad, pairs = [], 0
        chunk = None
        for tile_row in sorted({j // 4096 for j in cols}):
            mine = [(k, j) for k, j in enumerate(cols) if j // 4096 == tile_row]
            tiles = max(j for _, j in mine) // 4096 + 1
            raw = scratch / "tiles.bin"
            dims = subprocess.run([str(reader), str(out), str(tile_row), str(tiles), str(raw)], capture_output=True, text=True, timeout=900)
            assert dims.returncode == 0, dims.stdout + dims.stderr
            chunk = int(dims.stdout.split()[0])
            assert chunk == 4096
            blob = raw.read_bytes()
            at = 0
            strips = []  # the pinned rows of every tile of this tile row
            for c in range(tiles):
                size = int(np.frombuffer(blob, "<u8", 1, at)[0])
                stored = blob[at + 8:at + 8 + size]
                at += 8 + size
                tile = np.frombuffer(zlib.decompress(stored) if name == "cfg5" else stored, "<i4").reshape(chunk, chunk)
                strips.append(tile[[j - tile_row * chunk for _, j in mine]].copy())
                print(f"{name}: tile ({tile_row}, {c}) read back", flush=True)
            rows = np.concatenate(strips, axis=1)
            for t, (k, j) in enumerate(mine):
                col = rows[t, :j]
                got = (int(col.sum(dtype=np.int64)), int(np.bitwise_xor.reduce(col)), zlib.crc32(np.ascontiguousarray(col, "<i4").tobytes()))
                want = (int(z["sum"][k]), int(z["xor"][k]), int(z["crc32"][k]))
                if got != want:
                    bad.append((j, got, want))
                assert rows[t, j] == 0  # the diagonal is written as 0
                pairs += j
        assert not bad, f"{name} through the tool: {len(bad)} of {len(cols)} pinned columns differ from the reference, first: {bad[0]}"
        assert pairs > 8_000_000 and n == meta["n"]
    finally:
        for p in scratch.glob("*"):
            p.unlink()
        scratch.rmdir()
