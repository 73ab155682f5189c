"""CPU: the product's launch planner (sequencealigner_amd/csrc/sa_plan.cpp, sa_limits.cpp -- pure host code, no HIP)
compiled with g++ -fsanitize=address,undefined into tests/plan_host/plan_check and run on the geometry of the BASELINE
configs.  plan_check rebuilds every plan exactly as sa_ctx_align_range / sa_ctx_align_share do and checks it: every
pair of the range in exactly one tile, tile lists and dense shares inside their buffers, placement segments writing
every element once, arranged copies being block-local permutations.  Any heap error, overflow or UB in the planner
for these shapes fails the test (-fno-sanitize-recover).

cfg 5 is the geometry of the one abort this project has on record (round 3, test_baseline_full_size_cfg4_cfg5[cfg5]:
ranges [k 2^30, +2^30) of the 89 994-sequence post-filter store)."""
import json
import pathlib
import subprocess

import numpy as np
import pytest

from tests.synth import make_config

ROOT = pathlib.Path(__file__).resolve().parents[1]
HARNESS = ROOT / "tests" / "plan_host"
CSRC = ROOT / "sequencealigner_amd" / "csrc"
G30 = 1 << 30


@pytest.fixture(scope="session")
def plan_check(tmp_path_factory):
    exe = tmp_path_factory.mktemp("plan_host") / "plan_check"
    srcs = [HARNESS / "plan_check.cpp", CSRC / "sa_plan.cpp", CSRC / "sa_limits.cpp"]
    # (the matrix tables are data: compiled without instrumentation, which is most of their build time)
    tables = exe.with_name("sa_tables.o")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-c", str(CSRC / "sa_tables.cpp"), "-o", str(tables)])
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-Wall", "-Wextra", "-Wno-unused-parameter", *map(str, srcs), str(tables), "-o", str(exe)])
    return exe


def lens_file(tmp_path, name: str, lens: np.ndarray) -> pathlib.Path:
    p = tmp_path / f"lens_{name}.i32"
    np.asarray(lens, np.int32).tofile(p)
    return p


def config_lens(name: str) -> tuple[np.ndarray, dict]:
    seqs, cfg = make_config(name)
    lens = np.array([len(s) for s in seqs], np.int32)
    if name == "cfg5":  # `-f 0.9` first: the reference's own keep mask (tests/golden/stripes_cfg5.npz)
        d = np.load(ROOT / "tests" / "golden" / "stripes_cfg5.npz")
        lens = lens[np.unpackbits(d["keep_packed"])[:len(seqs)].astype(bool)]
        assert lens.size == json.loads(str(d["params"]))["kept"]
    return lens, cfg


def run(exe, lens_path, method, matrix, gaps, plans, cus=256, timeout=900):
    argv = [str(exe), str(lens_path), method, matrix, str(gaps.get("gap_pen", 0)), str(gaps.get("gap_open", 0)),
            str(gaps.get("gap_extend", 0)), str(cus)]
    for p in plans:
        argv += [str(int(v)) for v in p]
    res = subprocess.run(argv, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, f"{' '.join(argv)}\n{res.stdout[-3000:]}\n{res.stderr[-6000:]}"
    assert "ERROR: AddressSanitizer" not in res.stderr and "runtime error" not in res.stderr, res.stderr[-6000:]
    return res.stdout


def test_cfg2_every_schedule(plan_check, tmp_path):
    """10 000 x ~100 aa: the one-GPU whole-range launch, ranges that start and end inside columns, the world-8 / world-3
    share plans with and without host delivery -- placement checked element by element"""
    lens, cfg = config_lens("cfg2")
    pairs = lens.size * (lens.size - 1) // 2
    out = run(plan_check, lens_file(tmp_path, "cfg2", lens), cfg["method"], cfg["matrix"], cfg["gaps"],
              [(0, -1, 0, 1), (0, -1, 0, 0), (12345, pairs - 12345 - 777, 0, 0), (pairs // 3, pairs // 8, 0, 0),
               (0, -1, 8, 1), (0, -1, 8, 0), (0, -1, 3, 1), (0, -1, 1, 0), (pairs // 2, pairs // 5, 8, 0)])
    assert out.count("plan [") == 9 and "refused" not in out
    assert "world 8 host 1" in out and "/16/4," in out  # two tile sizes in a short launch


def test_cfg3_and_cfg4_shapes(plan_check, tmp_path):
    lens, cfg = config_lens("cfg3")
    out = run(plan_check, lens_file(tmp_path, "cfg3", lens), cfg["method"], cfg["matrix"], cfg["gaps"], [(0, -1, 0, 1), (0, -1, 4, 1)])
    assert out.count("plan [") == 2
    lens, cfg = config_lens("cfg4")  # 50 000 reads, SW: two bundles (K = 15..23), 2^30-pair ranges as the parity test issues them
    out = run(plan_check, lens_file(tmp_path, "cfg4", lens), cfg["method"], cfg["matrix"], cfg["gaps"],
              [(0, G30, 0, 0), (G30, -1, 0, 0), (0, -1, 0, 1)])
    assert out.count("plan [") == 3 and "2 bundles" in out


def test_cfg5_geometry_of_the_recorded_abort(plan_check, tmp_path):
    """the exact calls of tests/test_gpu_parity.py::test_baseline_full_size_cfg4_cfg5[cfg5] -- four ranges of 2^30 pairs
    over the post-filter store, the second to fourth starting inside a column and beyond packed index 2^31 -- and the
    whole 4.05e9-pair range as sa_ctx_align_host plans it"""
    lens, cfg = config_lens("cfg5")
    pairs = int(lens.size) * (int(lens.size) - 1) // 2
    assert lens.size == 89994 and (1 << 31) < pairs < (1 << 32)
    out = run(plan_check, lens_file(tmp_path, "cfg5", lens), cfg["method"], cfg["matrix"], cfg["gaps"],
              [(0, G30, 0, 0), (G30, G30, 0, 0), (2 * G30, G30, 0, 0), (3 * G30, -1, 0, 0), (0, -1, 0, 1)])
    assert out.count("plan [") == 5 and "refused" not in out


def test_every_kernel_family_in_one_store(plan_check, tmp_path):
    """lengths 1 .. 2600: 8- and 16-lane packed classes, s32 classes, strip-mined columns; Gotoh with |open| < |extend|
    sends everything to the pair-per-wave runs; a world that does not divide anything"""
    rng = np.random.default_rng(5)
    lens = np.concatenate([rng.integers(1, 200, 700), rng.integers(200, 1024, 250), rng.integers(1025, 2600, 50), [1, 1, 2, 1023, 1024, 1025]])
    rng.shuffle(lens)
    f = lens_file(tmp_path, "mixed", lens)
    pairs = lens.size * (lens.size - 1) // 2
    plans = [(0, -1, 0, 0), (0, -1, 0, 1), (0, -1, 3, 0), (0, -1, 7, 1), (1000, pairs - 5000, 5, 0), (pairs - 10, 10, 2, 0), (0, 1, 0, 0)]
    for method, matrix, gaps in (("nw", "blosum62", dict(gap_pen=4)), ("sw", "blosum62", dict(gap_open=10, gap_extend=1)),
                                 ("ga", "pam250", dict(gap_open=12, gap_extend=2)), ("ga", "blosum62", dict(gap_open=1, gap_extend=5)),
                                 ("nw", "blosum62", dict(gap_pen=300))):
        out = run(plan_check, f, method, matrix, gaps, plans)
        assert out.count("plan [") == len(plans) and "refused" not in out
        if gaps.get("gap_extend") == 5 or gaps.get("gap_pen") == 300:
            assert f"generic {pairs} pairs" in out  # nothing the systolic families reproduce exactly


def test_tiny_stores_and_many_ranks(plan_check, tmp_path):
    for n in (2, 3, 9):
        lens = np.full(n, 30, np.int32)
        out = run(plan_check, lens_file(tmp_path, f"tiny{n}", lens), "nw", "blosum62", dict(gap_pen=4),
                  [(0, -1, 0, 0), (0, -1, 8, 0), (0, -1, 8, 1), (0, 1, 2, 0)])
        assert out.count("plan [") == 4


def test_oversized_range_is_refused_not_crashed(plan_check, tmp_path):
    """1.1e6 sequences of 8 residues: 6e11 pairs are more workgroup-tiles than one launch can number -- the planner
    says so (sa_last_error) instead of overflowing; the caller splits the range"""
    lens = np.full(1_100_000, 8, np.int32)
    out = run(plan_check, lens_file(tmp_path, "huge", lens), "nw", "blosum62", dict(gap_pen=4), [(0, -1, 0, 0), (0, G30, 0, 0)])
    assert "refused: packed range too large for one launch" in out
    assert out.count("plan [") == 2 and out.count("refused") == 1  # the 2^30-pair sub-range plans fine
