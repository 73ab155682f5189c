"""ctypes access to cli/libsa_host.so (the C host: parsers, filter, HDF5 writer) for the tests."""
from __future__ import annotations

import ctypes as C
import pathlib
import subprocess

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[1]
HOST_SO = ROOT / "cli" / "libsa_host.so"
H5DUMP = pathlib.Path("/opt/conda/bin/h5dump")
H5DIFF = pathlib.Path("/opt/conda/bin/h5diff")


class _Meta(C.Structure):
    _fields_ = [("off", C.c_int32), ("len", C.c_int32)]


class _Input(C.Structure):
    _fields_ = [("seqs", C.POINTER(C.c_uint8)), ("meta", C.POINTER(_Meta)), ("max", C.c_int32), ("num", C.c_int32)]


class _Store(C.Structure):
    _fields_ = [("inp", _Input), ("blob_bytes", C.c_size_t)]


class HostError(RuntimeError):
    pass


class Host:
    def __init__(self):
        if not HOST_SO.exists():
            subprocess.check_call(["make", "-s", "-C", str(ROOT / "cli"), str(HOST_SO)])
        lib = C.CDLL(str(HOST_SO))
        lib.sa_host_error.restype = C.c_char_p
        lib.sa_host_parse.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_void_p, C.c_int32, C.c_int, C.c_int, C.POINTER(_Store)]
        lib.sa_host_load.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int, C.c_int, C.POINTER(_Store)]
        lib.sa_host_filter.argtypes = [C.POINTER(_Store), C.c_float, C.c_int]
        lib.sa_host_filter.restype = C.c_int32
        lib.sa_host_store_free.argtypes = [C.POINTER(_Store)]
        lib.sa_host_write_hdf5.argtypes = [C.c_char_p, C.POINTER(_Store), C.c_void_p, C.c_bool, C.c_uint]
        self.TILES_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t))
        lib.sa_host_write_hdf5_streams.argtypes = [C.c_char_p, C.POINTER(_Store), C.c_uint, self.TILES_FN, C.c_void_p]
        lib.sa_host_hdf5_chunk_dim.argtypes = [C.c_size_t]
        lib.sa_host_hdf5_chunk_dim.restype = C.c_size_t
        self.lib = lib

    def _err(self):
        return (self.lib.sa_host_error() or b"").decode()

    @staticmethod
    def _seqs(st: _Store) -> list[bytes]:
        out = []
        for k in range(st.inp.num):
            m = st.inp.meta[k]
            out.append(bytes(st.inp.seqs[m.off:m.off + m.len]))
        return out

    def parse(self, data: bytes, ext: str, lut: np.ndarray, gap: int = 0, column: int = -1, has_header: bool = True):
        st = _Store()
        lut = np.ascontiguousarray(lut, np.int32)
        if self.lib.sa_host_parse(data, len(data), ext.encode(), lut.ctypes.data, gap, column, int(has_header), C.byref(st)):
            raise HostError(self._err())
        return st

    def parse_sequences(self, data: bytes, ext: str, lut: np.ndarray, **kw) -> list[bytes]:
        st = self.parse(data, ext, lut, **kw)
        try:
            # layout contract: tight NUL-separated blob (reference src/io/input.c:68-81)
            off = 0
            for k in range(st.inp.num):
                assert st.inp.meta[k].off == off
                off += st.inp.meta[k].len + 1
                assert st.inp.seqs[off - 1] == 0
            assert st.blob_bytes == off
            assert st.inp.max == max(st.inp.meta[k].len for k in range(st.inp.num))
            return self._seqs(st)
        finally:
            self.lib.sa_host_store_free(C.byref(st))

    def filter(self, seqs: list[bytes], lut: np.ndarray, threshold: float, threads: int = 0) -> list[bytes]:
        data = b"".join(b">s\n" + s + b"\n" for s in seqs)
        st = self.parse(data, "fasta", lut)
        try:
            if self.lib.sa_host_filter(C.byref(st), C.c_float(threshold), threads) < 0:
                raise HostError(self._err())
            return self._seqs(st)
        finally:
            self.lib.sa_host_store_free(C.byref(st))

    def write_hdf5(self, path, seqs: list[bytes], lut: np.ndarray, matrix: np.ndarray, triangular: bool, compression: int = 0):
        data = b"".join(b">s\n" + s + b"\n" for s in seqs)
        st = self.parse(data, "fasta", lut)
        try:
            m = np.ascontiguousarray(matrix, np.int32).reshape(-1)
            if self.lib.sa_host_write_hdf5(str(path).encode(), C.byref(st), m.ctypes.data, bool(triangular), compression):
                raise HostError(self._err())
        finally:
            self.lib.sa_host_store_free(C.byref(st))

    def write_hdf5_streams(self, path, seqs: list[bytes], lut: np.ndarray, compression: int, batches):
        """sa_host_write_hdf5_streams with `batches` (an iterable of lists of (tile row, tile column, bytes)) as the tile source --
        what sa_zjob_next is for the tool; a batch of None makes the source report an error"""
        data = b"".join(b">s\n" + s + b"\n" for s in seqs)
        st = self.parse(data, "fasta", lut)
        it = iter(batches)
        keep = []

        def source(user, rows, cols, streams, sizes):
            batch = next(it, [])
            if batch is None:
                return -1
            keep.clear()
            for t, (r, c, z) in enumerate(batch):
                buf = C.create_string_buffer(z, len(z))
                keep.append(buf)
                rows[t], cols[t], streams[t], sizes[t] = r, c, C.addressof(buf), len(z)
            return len(batch)
        try:
            if self.lib.sa_host_write_hdf5_streams(str(path).encode(), C.byref(st), compression, self.TILES_FN(source), None):
                raise HostError(self._err())
        finally:
            self.lib.sa_host_store_free(C.byref(st))

    def chunk_dim(self, dim: int) -> int:
        return int(self.lib.sa_host_hdf5_chunk_dim(dim))


def h5_matrix(path, n: int) -> np.ndarray:
    """/similarity_matrix of an HDF5 file as int32 [n, n] (through h5dump -b LE, h5py is not installed)."""
    out = pathlib.Path(str(path) + ".bin")
    subprocess.check_call([str(H5DUMP), "-d", "/similarity_matrix", "-b", "LE", "-o", str(out), str(path)],
                          stdout=subprocess.DEVNULL)
    return np.fromfile(out, dtype="<i4").reshape(n, n)


def h5_sequences(path) -> list[bytes]:
    txt = subprocess.run([str(H5DUMP), "-d", "/sequences", "-y", "-w", "0", str(path)], capture_output=True, text=True, check=True).stdout
    body = txt[txt.index("DATA {") + 6: txt.rindex("}")]
    body = body[: body.rindex("}")] if body.strip().endswith("}") else body
    import re
    return [m.encode() for m in re.findall(r'"([^"]*)"', body)]
