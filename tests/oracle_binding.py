"""ctypes access to the TEST ORACLES.  Imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by sequencealigner_amd/.

Oracle : oracle/libsa_oracle.so  (our CPU restatement, oracle/sa_oracle.c)
RefLib : oracle/_ref/libseqalign_ref.so (the reference's own sources + oracle/ref_shim.c), optional."""
from __future__ import annotations

import ctypes as C
import os
import pathlib
import shutil
import subprocess
import tempfile

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[1]
ORACLE_SO = ROOT / "oracle" / "libsa_oracle.so"
REF_SO = ROOT / "oracle" / "_ref" / "libseqalign_ref.so"


class _Params(C.Structure):
    _fields_ = [("method", C.c_int32), ("gap_pen", C.c_int32), ("gap_opn", C.c_int32), ("gap_ext", C.c_int32),
                ("lut", C.c_int32 * 128), ("sub", C.c_int32 * 576)]


def _params(scoring) -> _Params:
    p = _Params(scoring.method, scoring.gap_pen, scoring.gap_opn, scoring.gap_ext)
    C.memmove(p.lut, np.ascontiguousarray(scoring.lut, np.int32).ctypes.data, 512)
    C.memmove(p.sub, np.ascontiguousarray(scoring.sub, np.int32).ctypes.data, 2304)
    return p


class Oracle:
    def __init__(self):
        if not ORACLE_SO.exists():
            subprocess.check_call(["make", "-s", "-C", str(ROOT / "oracle"), "oracle"])
        lib = C.CDLL(str(ORACLE_SO))
        lib.sa_oracle_pair.restype = C.c_int32
        lib.sa_oracle_pair.argtypes = [C.POINTER(_Params), C.c_char_p, C.c_int32, C.c_char_p, C.c_int32]
        lib.sa_oracle_align.argtypes = [C.POINTER(_Params), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int, C.c_int]
        lib.sa_oracle_align_range.argtypes = [C.POINTER(_Params), C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_int]
        lib.sa_oracle_align_pairs.argtypes = [C.POINTER(_Params), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
        lib.sa_oracle_filter.restype = C.c_int32
        lib.sa_oracle_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_void_p]
        lib.sa_oracle_unpack_index.argtypes = [C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        self.lib = lib

    @property
    def max_threads(self) -> int:
        return int(self.lib.sa_oracle_max_threads())

    def pair(self, scoring, seq1: bytes, seq2: bytes) -> int:
        """seq1 = column sequence (the reference's pre-indexed one), seq2 = row sequence."""
        p = _params(scoring)
        return int(self.lib.sa_oracle_pair(C.byref(p), seq1, len(seq1), seq2, len(seq2)))

    def align(self, store, scoring, triangular: bool = False, threads: int = 0) -> np.ndarray:
        n = store.num
        m = np.zeros(n * (n - 1) // 2 if triangular else n * n, np.int32)
        p = _params(scoring)
        rc = self.lib.sa_oracle_align(C.byref(p), store.blob.ctypes.data, store.meta.ctypes.data, n, m.ctypes.data,
                                      int(triangular), threads)
        assert rc == 0
        return m if triangular else m.reshape(n, n)

    def align_range(self, store, scoring, start: int, count: int, threads: int = 0) -> np.ndarray:
        out = np.zeros(count, np.int32)
        p = _params(scoring)
        rc = self.lib.sa_oracle_align_range(C.byref(p), store.blob.ctypes.data, store.meta.ctypes.data, store.num,
                                            start, count, out.ctypes.data, threads)
        assert rc == 0
        return out

    def align_pairs(self, store, scoring, idx: np.ndarray, threads: int = 0) -> np.ndarray:
        idx = np.ascontiguousarray(idx, np.int64)
        out = np.zeros(idx.size, np.int32)
        p = _params(scoring)
        rc = self.lib.sa_oracle_align_pairs(C.byref(p), store.blob.ctypes.data, store.meta.ctypes.data, store.num,
                                            idx.ctypes.data, idx.size, out.ctypes.data, threads)
        assert rc == 0
        return out

    def filter(self, store, threshold: float) -> np.ndarray:
        keep = np.zeros(store.num, np.uint8)
        self.lib.sa_oracle_filter(store.blob.ctypes.data, store.meta.ctypes.data, store.num, C.c_float(threshold), keep.ctypes.data)
        return keep.astype(bool)

    def unpack(self, p: int) -> tuple[int, int]:
        i, j = C.c_int32(), C.c_int32()
        self.lib.sa_oracle_unpack_index(p, C.byref(i), C.byref(j))
        return i.value, j.value


def ref_available() -> bool:
    return REF_SO.exists()


class RefLib:
    """One configured instance of the reference library (fresh private copy of the .so, because the
    reference's option parser keeps per-process state)."""

    def __init__(self, method: str, matrix: str, gap_pen=None, gap_open=None, gap_extend=None, threads: int = 0,
                 filter_threshold: float | None = None, compression: int | None = None):
        if not REF_SO.exists():
            raise FileNotFoundError(REF_SO)
        self._tmpdir = tempfile.mkdtemp(prefix="saref_")
        so = pathlib.Path(self._tmpdir) / "libseqalign_ref_copy.so"
        shutil.copy(REF_SO, so)
        fasta = pathlib.Path(self._tmpdir) / "dummy.fasta"
        fasta.write_bytes(b">a\nAAA\n>b\nAAC\n")
        # no -W: with it the reference's output_fill() is a no-op (src/io/output.c:70-71)
        argv = ["seqalign", "-i", str(fasta), "-o", str(pathlib.Path(self._tmpdir) / "out.h5"), "-a", method,
                "-m", matrix, "-Q", "-P", "-F"]
        if gap_pen is not None:
            argv += ["-p", str(gap_pen)]
        if gap_open is not None:
            argv += ["-s", str(gap_open)]
        if gap_extend is not None:
            argv += ["-e", str(gap_extend)]
        if threads:
            argv += ["-T", str(threads)]
        if filter_threshold is not None:
            argv += ["-f", repr(float(filter_threshold))]
        if compression is not None:
            argv += ["-z", str(compression)]
        self.output_path = pathlib.Path(self._tmpdir) / "out.h5"
        self.lib = C.CDLL(str(so))
        arr = (C.c_char_p * len(argv))(*[a.encode() for a in argv])
        self._argv = arr
        rc = self.lib.ref_configure(len(argv), arr)
        if rc:
            raise RuntimeError(f"reference option parser rejected {argv} (rc={rc})")
        self.lib.ref_pair.restype = C.c_int32
        self.lib.ref_pair.argtypes = [C.c_char_p, C.c_int32, C.c_char_p, C.c_int32]
        self.lib.ref_align.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int]
        self.lib.ref_filter.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]

    def params(self) -> dict:
        gp, go, ge, aff = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        lut = (C.c_int32 * 128)()
        sub = (C.c_int32 * 576)()
        name = C.create_string_buffer(64)
        self.lib.ref_get_params(C.byref(gp), C.byref(go), C.byref(ge), C.byref(aff), lut, sub, name, 64)
        return dict(gap_pen=gp.value, gap_opn=go.value, gap_ext=ge.value, affine=bool(aff.value),
                    lut=np.array(lut, np.int32), sub=np.array(sub, np.int32), method=name.value.decode())

    def pair(self, seq1: bytes, seq2: bytes) -> int:
        return int(self.lib.ref_pair(seq1, len(seq1), seq2, len(seq2)))

    def align(self, store, triangular: bool = False) -> np.ndarray:
        n = store.num
        m = np.zeros(n * (n - 1) // 2 if triangular else n * n, np.int32)
        blob = store.blob.copy()
        meta = store.meta.copy()
        rc = self.lib.ref_align(blob.ctypes.data, meta.ctypes.data, n, store.max, m.ctypes.data, int(triangular))
        assert rc == 0
        return m if triangular else m.reshape(n, n)

    def filter(self, store) -> list[bytes]:
        blob = store.blob.copy()
        meta = store.meta.copy()
        num, mx = C.c_int32(store.num), C.c_int32(store.max)
        kept = self.lib.ref_filter(blob.ctypes.data, meta.ctypes.data, C.byref(num), C.byref(mx))
        assert kept >= 0
        return [blob[meta[k, 0]:meta[k, 0] + meta[k, 1]].tobytes() for k in range(kept)]

    def flush_hdf5(self, store, matrix: np.ndarray, triangular: bool) -> pathlib.Path:
        """The reference's own HDF5 writer on a given result matrix -> path of the file it wrote."""
        blob, meta = store.blob.copy(), store.meta.copy()
        m = np.ascontiguousarray(matrix, np.int32).reshape(-1).copy()
        self.lib.ref_flush.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int]
        rc = self.lib.ref_flush(blob.ctypes.data, meta.ctypes.data, store.num, m.ctypes.data, int(triangular))
        assert rc == 0
        return self.output_path

    def close(self):
        shutil.rmtree(self._tmpdir, ignore_errors=True)
