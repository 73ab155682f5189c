"""CPU: the serial core of the device-side DEFLATE encoder (sequencealigner_amd/csrc/sa_deflate_core.h -- match choice,
histograms, length-limited Huffman codes, dynamic block header, element bits, segment end, Adler-32) compiled with
g++ -fsanitize=address,undefined into tests/host_c/deflate_core_test and run on the host: the kernel
(csrc/sa_deflate.hip) executes these same functions in one thread of a workgroup.  The streams must inflate, with
stock zlib, to exactly the input bytes (the -z option's contract: reference src/io/format/hdf5.c:91-95 hands the
chunks to libhdf5's deflate filter, any reader inflates them)."""
import pathlib
import subprocess
import zlib

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = tmp_path_factory.mktemp("deflate_core") / "deflate_core_test"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-Wall", "-Wextra", str(ROOT / "tests" / "host_c" / "deflate_core_test.cpp"), "-o", str(exe)])
    return exe


def test_length_limited_codes_are_complete(harness):
    """Fibonacci / power-of-two / one-heavy-symbol histograms: every code within 15 (7) bits, Kraft sum exactly 1"""
    res = subprocess.run([str(harness), "--kraft"], capture_output=True, text=True)
    assert res.returncode == 0 and "kraft ok" in res.stdout, res.stdout + res.stderr


def contents(kind: str, rng) -> np.ndarray:
    n = 40000
    if kind == "scores":  # sign bytes + a low byte: what a tile of NW scores looks like
        return rng.integers(-150, 110, size=n, dtype=np.int32)
    if kind == "zeros":
        return np.zeros(n, np.int32)
    if kind == "constant":
        return np.full(n, -12345, np.int32)
    if kind == "full_range":
        return rng.integers(-2**31, 2**31 - 1, size=n, dtype=np.int64).astype(np.int32)
    if kind == "positive_small":
        return rng.integers(0, 30, size=n, dtype=np.int32)
    if kind == "geometric":  # code lengths run into the 15-bit limit
        return (np.int64(1) << (np.minimum(rng.geometric(0.5, size=n), 40) % 31)).astype(np.int32)
    if kind == "one_element":
        return np.array([-7], np.int32)
    if kind == "high_parts":  # few distinct high parts, far apart: matches at every distance and misses
        return (rng.integers(0, 5, size=n, dtype=np.int32) << 8) * 977 + rng.integers(0, 256, size=n, dtype=np.int32)
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["scores", "zeros", "constant", "full_range", "positive_small", "geometric", "one_element", "high_parts"])
@pytest.mark.parametrize("segment,group", [(16384, 1), (777, 1), (2048, 16)])
def test_streams_inflate_to_the_input(kind, segment, group, harness, tmp_path):
    data = contents(kind, np.random.default_rng(len(kind) * 1000 + segment))
    src, dst = tmp_path / "in.i32", tmp_path / "out.zz"
    data.astype("<i4").tofile(src)
    res = subprocess.run([str(harness), str(src), str(dst), str(segment), str(group)], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "ERROR: AddressSanitizer" not in res.stderr and "runtime error" not in res.stderr, res.stderr
    z = dst.read_bytes()
    assert zlib.decompress(z) == data.astype("<i4").tobytes()
    if kind == "scores" and segment == 16384:
        assert data.nbytes / len(z) > 2.4  # (a uniform low byte: 9 + 1 + ~2.5 bits per element; literal-only Huffman: 2.2)
    if kind == "full_range":
        assert len(z) < 1.15 * data.nbytes  # incompressible input grows by the code's redundancy only
