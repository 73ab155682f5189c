"""ctypes binding of include/seqalign_hip.h (no torch dependency; torch is only plumbing for callers)."""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import pathlib
import sys
from dataclasses import dataclass, field
from typing import Iterable, Optional, Sequence

import numpy as np

_LIB_PATH = pathlib.Path(__file__).resolve().parent / "lib" / "libseqalign_hip.so"
_lib: Optional[C.CDLL] = None

LUT_SIZE = 128
SUB_DIM = 24
SCORE_MIN = -(1 << 30)  # reference src/bio/align.h:19

METHOD_NW, METHOD_GA, METHOD_SW = 0, 1, 2
GAP_LINEAR, GAP_AFFINE = 0, 1


class AlignError(RuntimeError):
    """A call through the C ABI returned failure (message = sa_last_error())."""


class _Meta(C.Structure):  # struct sa_meta  <- reference src/bio/align.h:6-9
    _fields_ = [("off", C.c_int32), ("len", C.c_int32)]


class _Input(C.Structure):  # struct sa_input <- reference src/io/input.h:6-11
    _fields_ = [("seqs", C.c_void_p), ("meta", C.c_void_p), ("max", C.c_int32), ("num", C.c_int32)]


class _Output(C.Structure):  # struct sa_output <- reference src/io/output.h:10-15
    _fields_ = [("matrix", C.c_void_p), ("seqs", C.c_void_p), ("dim", C.c_size_t), ("triangular", C.c_bool)]


class _Scoring(C.Structure):  # struct sa_scoring
    _fields_ = [("method", C.c_int32), ("gap_pen", C.c_int32), ("gap_opn", C.c_int32), ("gap_ext", C.c_int32),
                ("lut", C.c_int32 * LUT_SIZE), ("sub", C.c_int32 * (SUB_DIM * SUB_DIM))]


#: every symbol include/seqalign_hip.h declares (tests check the .so exports exactly these)
ABI_SYMBOLS = (
    "sa_hip_memory", "sa_hip_align", "sa_hip_filter", "sa_ctx_create", "sa_ctx_destroy", "sa_ctx_pairs", "sa_pairs_cells",
    "sa_ctx_align_range", "sa_ctx_align_range16", "sa_ctx_scores_fit16", "sa_hip_widen16", "sa_ctx_expand_full", "sa_pairs_partition", "sa_ctx_timing", "sa_ctx_timing_read",
    "sa_matrix_load", "sa_matrix_count", "sa_matrix_name", "sa_matrix_is_nucleotide", "sa_method_parse",
    "sa_method_name", "sa_method_gap_kind", "sa_hip_device_count", "sa_hip_device_name", "sa_last_error",
    "sa_abi_version",
    "sa_hip_last_align_seconds", "sa_ctx_align_host", "sa_hip_host_register", "sa_hip_host_unregister",
    "sa_ctx_share_elems", "sa_ctx_align_share", "sa_ctx_place_shares", "sa_hip_last_align_breakdown", "sa_ctx_leave_room", "sa_hip_set_progress",
    "sa_hip_last_align_path",
    "sa_zjob_create", "sa_zjob_destroy", "sa_zjob_tiles_per_row", "sa_zjob_tile_row", "sa_zjob_stats", "sa_zjob_next", "sa_zjob_align_seconds", "sa_hip_tiles_begin",
)


def library_path() -> pathlib.Path:
    return _LIB_PATH


def _share_hip_runtime_with_torch() -> None:
    """One HIP runtime per process.  The PyTorch-ROCm wheel ships a private libamdhip64.so that its
    libraries request by the un-versioned file name, so it does not unify with /opt/rocm's copy by
    soname; two runtimes in one process leave the second without devices.  When torch is installed
    but not imported yet, bring ITS runtime in first so that libseqalign_hip.so (NEEDED
    libamdhip64.so.7) and a later `import torch` resolve to the same object.  A plain C host links
    /opt/rocm's runtime and never sees this."""
    if "torch" in sys.modules or os.environ.get("SA_HIP_NO_TORCH_RUNTIME"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    cand = pathlib.Path(list(spec.submodule_search_locations)[0]) / "lib" / "libamdhip64.so"
    if cand.exists():
        C.CDLL(str(cand), mode=C.RTLD_GLOBAL)


def load_library() -> C.CDLL:
    """Load libseqalign_hip.so (built in-tree by __graft_entry__.build()).  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise AlignError(f"{_LIB_PATH} is missing: run `python __graft_entry__.py` (hipcc, gfx950) first; "
                         "there is no CPU fallback")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(str(_LIB_PATH))
    lib.sa_hip_memory.argtypes = [C.c_size_t]
    lib.sa_hip_memory.restype = C.c_bool
    lib.sa_hip_align.argtypes = [_Input, _Output, C.POINTER(_Scoring)]
    lib.sa_hip_align.restype = C.c_bool
    lib.sa_hip_filter.argtypes = [_Input, C.c_float, C.c_void_p]
    lib.sa_hip_filter.restype = C.c_int32
    lib.sa_ctx_create.argtypes = [C.c_int, _Input, C.POINTER(_Scoring)]
    lib.sa_ctx_create.restype = C.c_void_p
    lib.sa_ctx_destroy.argtypes = [C.c_void_p]
    lib.sa_ctx_destroy.restype = None
    lib.sa_ctx_pairs.argtypes = [C.c_void_p]
    lib.sa_ctx_pairs.restype = C.c_int64
    lib.sa_pairs_cells.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int64]
    lib.sa_pairs_cells.restype = C.c_int64
    lib.sa_ctx_align_range.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    lib.sa_ctx_align_range.restype = C.c_int
    lib.sa_ctx_align_range16.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    lib.sa_ctx_align_range16.restype = C.c_int
    lib.sa_ctx_scores_fit16.argtypes = [C.c_void_p]
    lib.sa_ctx_scores_fit16.restype = C.c_int
    lib.sa_hip_widen16.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.sa_hip_widen16.restype = C.c_int
    lib.sa_ctx_expand_full.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.sa_ctx_expand_full.restype = C.c_int
    lib.sa_pairs_partition.argtypes = [C.c_void_p, C.c_int32, C.c_int, C.POINTER(C.c_int64)]
    lib.sa_pairs_partition.restype = C.c_int
    lib.sa_ctx_timing.argtypes = [C.c_void_p, C.c_int]
    lib.sa_ctx_timing.restype = None
    lib.sa_ctx_timing_read.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                       C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    lib.sa_ctx_timing_read.restype = C.c_int
    lib.sa_matrix_load.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.sa_matrix_load.restype = C.c_int
    lib.sa_matrix_count.restype = C.c_int
    lib.sa_matrix_name.argtypes = [C.c_int]
    lib.sa_matrix_name.restype = C.c_char_p
    lib.sa_matrix_is_nucleotide.argtypes = [C.c_int]
    lib.sa_matrix_is_nucleotide.restype = C.c_int
    lib.sa_method_parse.argtypes = [C.c_char_p]
    lib.sa_method_parse.restype = C.c_int
    lib.sa_method_name.argtypes = [C.c_int]
    lib.sa_method_name.restype = C.c_char_p
    lib.sa_method_gap_kind.argtypes = [C.c_int]
    lib.sa_method_gap_kind.restype = C.c_int
    lib.sa_hip_device_count.restype = C.c_int
    lib.sa_hip_device_name.argtypes = [C.c_int]
    lib.sa_hip_device_name.restype = C.c_char_p
    lib.sa_last_error.restype = C.c_char_p
    lib.sa_abi_version.restype = C.c_int
    lib.sa_hip_last_align_seconds.restype = C.c_double
    lib.sa_hip_last_align_path.restype = C.c_int
    lib.sa_ctx_align_host.argtypes = [C.c_void_p, C.c_int64, C.c_int64, _Output, C.POINTER(C.c_double)]
    lib.sa_ctx_align_host.restype = C.c_int
    lib.sa_hip_host_register.argtypes = [C.c_void_p, C.c_size_t]
    lib.sa_hip_host_register.restype = C.c_int
    lib.sa_hip_host_unregister.argtypes = [C.c_void_p]
    lib.sa_hip_host_unregister.restype = C.c_int
    lib.sa_hip_last_align_breakdown.argtypes = [C.POINTER(C.c_double), C.c_int]
    lib.sa_hip_last_align_breakdown.restype = C.c_int
    lib.sa_hip_set_progress.argtypes = [C.c_void_p, C.c_void_p]
    lib.sa_hip_set_progress.restype = None
    lib.sa_ctx_leave_room.argtypes = [C.c_void_p, C.c_int]
    lib.sa_ctx_leave_room.restype = None
    lib.sa_ctx_share_elems.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int]
    lib.sa_ctx_share_elems.restype = C.c_int64
    lib.sa_ctx_align_share.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.sa_ctx_align_share.restype = C.c_int
    lib.sa_ctx_place_shares.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.sa_ctx_place_shares.restype = C.c_int
    lib.sa_zjob_create.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_size_t, C.c_int]
    lib.sa_zjob_create.restype = C.c_void_p
    lib.sa_zjob_destroy.argtypes = [C.c_void_p]
    lib.sa_zjob_destroy.restype = None
    lib.sa_zjob_tiles_per_row.argtypes = [C.c_void_p]
    lib.sa_zjob_tiles_per_row.restype = C.c_size_t
    lib.sa_zjob_tile_row.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    lib.sa_zjob_tile_row.restype = C.c_int
    lib.sa_zjob_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.sa_zjob_stats.restype = None
    lib.sa_zjob_next.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    lib.sa_zjob_next.restype = C.c_int
    lib.sa_zjob_align_seconds.argtypes = [C.c_void_p]
    lib.sa_zjob_align_seconds.restype = C.c_double
    lib.sa_hip_tiles_begin.argtypes = [_Input, C.POINTER(_Scoring), C.c_size_t, C.c_int]
    lib.sa_hip_tiles_begin.restype = C.c_void_p
    _lib = lib
    return lib


def _err() -> str:
    return (load_library().sa_last_error() or b"").decode(errors="replace")


def device_count() -> int:
    return int(load_library().sa_hip_device_count())


def last_align_seconds() -> float:
    """launch/copy phase of the last hip_align call (the reference's bench_align bracket, seqalign_cuda.c:182,292)"""
    return float(load_library().sa_hip_last_align_seconds())


def last_align_path() -> str:
    """schedule of the last hip_align call: "slices" (every device delivers a contiguous slice; one device: the whole
    range) or "gather" (dense shares + RCCL all-gather + placement on every device)"""
    return {0: "none", 1: "slices", 2: "gather"}[int(load_library().sa_hip_last_align_path())]


def last_align_breakdown() -> dict:
    """milliseconds of the last hip_align call by stage (enum sa_breakdown): set-up itemised, then the phase"""
    names = ("encode_ms", "device_ms", "upload_ms", "code_objects_ms", "pin_ms", "plan_ms", "arrange_ms", "phase_ms", "total_ms")
    buf = (C.c_double * len(names))()
    n = load_library().sa_hip_last_align_breakdown(buf, len(names))
    return {names[k]: float(buf[k]) for k in range(n)}


PROGRESS_FN = C.CFUNCTYPE(None, C.c_double, C.c_void_p)
_progress_keepalive = None


def set_progress(fn) -> None:
    """fn(fraction) while hip_align / Context.align_host waits for the device (the reference's progress side channel,
    seqalign_cuda.c:286-289); None switches it off"""
    global _progress_keepalive
    lib = load_library()
    if fn is None:
        lib.sa_hip_set_progress(None, None)
        _progress_keepalive = None
        return
    _progress_keepalive = PROGRESS_FN(lambda fraction, _user: fn(float(fraction)))
    lib.sa_hip_set_progress(C.cast(_progress_keepalive, C.c_void_p), None)


def device_name(device: int = 0) -> str:
    name = load_library().sa_hip_device_name(device)
    if not name:
        raise AlignError(f"no HIP device {device}")
    return name.decode()


def matrix_names() -> list[str]:
    lib = load_library()
    return [lib.sa_matrix_name(k).decode() for k in range(lib.sa_matrix_count())]


def method_names() -> list[str]:
    lib = load_library()
    return [lib.sa_method_name(k).decode() for k in range(3)]


def pair_count(n: int) -> int:
    """reference src/util/macros.h:13 `alignments(n)`"""
    return n * (n - 1) // 2


# --------------------------------------------------------------------------------------------
@dataclass
class SequenceStore:
    """The reference's `struct input` (src/io/input.h:6-11): one NUL-separated uppercase blob + meta[]."""
    blob: np.ndarray            # uint8, sum(len+1) bytes
    meta: np.ndarray            # int32 [num, 2] = (off, len)
    num: int
    max: int

    @classmethod
    def from_sequences(cls, seqs: Iterable[bytes | str]) -> "SequenceStore":
        # what input_load builds after a parser ran (src/io/input.c:68-81): sequences back to back,
        # each NUL-terminated; parsers upper-case residues (src/io/source/fasta.c:51)
        items = [(s.encode() if isinstance(s, str) else bytes(s)).upper() for s in seqs]
        lens = np.fromiter((len(s) for s in items), dtype=np.int64, count=len(items))
        total = int(lens.sum()) + len(items)
        if total > np.iinfo(np.int32).max:
            raise AlignError("sequence store exceeds 2 GiB")
        blob = np.frombuffer(b"\0".join(items) + b"\0", dtype=np.uint8).copy()
        offs = np.zeros(len(items), dtype=np.int64)
        if len(items):
            offs[1:] = np.cumsum(lens[:-1] + 1)
        meta = np.stack([offs, lens], axis=1).astype(np.int32)
        return cls(blob=blob, meta=np.ascontiguousarray(meta), num=len(items), max=int(lens.max()) if len(items) else 0)

    def sequence(self, k: int) -> bytes:
        off, ln = self.meta[k]
        return self.blob[off:off + ln].tobytes()

    def select(self, keep: Sequence[int]) -> "SequenceStore":
        return SequenceStore.from_sequences(self.sequence(int(k)) for k in keep)

    def prefix(self, n: int) -> "SequenceStore":
        return self.select(range(n))

    def _as_c(self) -> _Input:
        return _Input(self.blob.ctypes.data, self.meta.ctypes.data, self.max, self.num)

    @property
    def pairs(self) -> int:
        return pair_count(self.num)

    def cells(self, start: int = 0, count: Optional[int] = None) -> int:
        """DP cells (sum len_i*len_j) of packed pair range [start, start+count) -- GCUPS numerator."""
        count = self.pairs - start if count is None else count
        v = int(load_library().sa_pairs_cells(self.meta.ctypes.data, self.num, start, count))
        if v < 0:
            raise AlignError("bad pair range")
        return v

    def partition(self, parts: int) -> list[int]:
        """Cut points of `parts` contiguous packed-index ranges of near-equal DP work (multi-GPU sharding)."""
        b = (C.c_int64 * (parts + 1))()
        if load_library().sa_pairs_partition(self.meta.ctypes.data, self.num, parts, b):
            raise AlignError(_err())
        return list(b)


@dataclass
class Scoring:
    """The reference's scoring globals (src/bio/align.h:11-19); gaps in STORED (negated) form."""
    method: int
    gap_pen: int = 0
    gap_opn: int = 0
    gap_ext: int = 0
    lut: np.ndarray = field(default_factory=lambda: np.full(LUT_SIZE, -1, np.int32))
    sub: np.ndarray = field(default_factory=lambda: np.zeros(SUB_DIM * SUB_DIM, np.int32))
    matrix_name: str = ""

    @classmethod
    def from_names(cls, method: str, matrix: str, gap_pen: Optional[int] = None, gap_open: Optional[int] = None,
                   gap_extend: Optional[int] = None, equal_affine_to_nw: bool = True) -> "Scoring":
        """`-a METHOD -m MATRIX (-p N | -s N -e N)` with the reference's validation rules.

        parse_align src/bio/align.c:87-96; parse_matrix src/bio/matrices.c:44-58; gap values are given
        positive and stored negated (src/bio/align.c:127-128); -p only with a linear method, -s/-e only with
        an affine one and both required (src/bio/align.c:130-142,170-201); Gotoh with open == extend becomes
        NW with that penalty (validate_ga, src/bio/method/ga.c:70-88, the -F answer)."""
        lib = load_library()
        m = lib.sa_method_parse(method.encode())
        if m < 0:
            raise AlignError("Invalid alignment method")
        lut = np.empty(LUT_SIZE, np.int32)
        sub = np.empty(SUB_DIM * SUB_DIM, np.int32)
        if lib.sa_matrix_load(matrix.encode(), lut.ctypes.data_as(C.POINTER(C.c_int32)),
                              sub.ctypes.data_as(C.POINTER(C.c_int32))):
            raise AlignError("Invalid substitution matrix name")
        for name, v in (("gap_pen", gap_pen), ("gap_open", gap_open), ("gap_extend", gap_extend)):
            if v is not None and not (0 <= int(v) <= 2**31 - 1):
                raise AlignError("Gap values must be positive integers")
        kind = lib.sa_method_gap_kind(m)
        if kind == GAP_LINEAR:
            if gap_open is not None or gap_extend is not None:
                raise AlignError("Affine gaps cannot be set for non-affine methods")
            if gap_pen is None:
                raise AlignError("Linear gap penalty (-p) is required")
            return cls(m, gap_pen=-int(gap_pen), lut=lut, sub=sub, matrix_name=matrix.lower())
        if gap_pen is not None:
            raise AlignError("Gap penalty cannot be set for non-linear methods")
        if gap_open is None or gap_extend is None:
            raise AlignError("Affine gap open (-s) and extend (-e) are required")
        if m == METHOD_GA and gap_open == gap_extend and equal_affine_to_nw:
            return cls(METHOD_NW, gap_pen=-int(gap_open), gap_opn=SCORE_MIN, gap_ext=SCORE_MIN, lut=lut, sub=sub,
                       matrix_name=matrix.lower())
        return cls(m, gap_opn=-int(gap_open), gap_ext=-int(gap_extend), lut=lut, sub=sub, matrix_name=matrix.lower())

    @property
    def method_name(self) -> str:
        return ("nw", "ga", "sw")[self.method]

    def _as_c(self) -> _Scoring:
        s = _Scoring(self.method, self.gap_pen, self.gap_opn, self.gap_ext)
        C.memmove(s.lut, np.ascontiguousarray(self.lut, np.int32).ctypes.data, 4 * LUT_SIZE)
        C.memmove(s.sub, np.ascontiguousarray(self.sub, np.int32).ctypes.data, 4 * SUB_DIM * SUB_DIM)
        return s


# --------------------------------------------------------------------------------------------
def hip_memory(nbytes: int) -> bool:
    """`cuda_memory` replacement (reference src/interface/seqalign_cuda.c:71-93)."""
    return bool(load_library().sa_hip_memory(int(nbytes)))


def hip_align(store: SequenceStore, scoring: Scoring, triangular: bool = False, write: bool = True) -> Optional[np.ndarray]:
    """`cuda_align` replacement (reference src/interface/seqalign_cuda.c:95-296).

    Returns the host matrix the reference would hand to its writer: packed triangular
    (pair i<j at j(j-1)/2+i) or full N x N symmetric with zero diagonal; None when write=False
    (the reference's -W: compute, copy nothing)."""
    lib = load_library()
    n = store.num
    matrix = None
    if write:
        # an anonymous zero-filled mapping like output_load's (output.c:55) -- page-aligned and not malloc's, so the library may
        # page-lock it for the call (it never locks memory malloc manages: DESIGN.md 9)
        import mmap
        elements = pair_count(n) if triangular else n * n
        matrix = np.frombuffer(mmap.mmap(-1, 4 * max(elements, 1)), dtype=np.int32)[:elements]
    out = _Output(matrix.ctypes.data if matrix is not None else None, None, n, bool(triangular))
    sc = scoring._as_c()
    if not lib.sa_hip_align(store._as_c(), out, C.byref(sc)):
        raise AlignError(_err())
    if matrix is None:
        return None
    return matrix if triangular else matrix.reshape(n, n)


class PinnedMatrix:
    """A host result matrix page-locked once (what a C host does in output_load with sa_hip_host_register), so that
    repeated deliveries into it are pure DMA / direct stores.  `.array` is the flat int32 numpy view.

    `shared=path`: the matrix is a shared file mapping (e.g. under /dev/shm) that every rank of a node attaches and
    page-locks -- one host matrix for a one-process-per-GPU run, the reference's single mmap-ed result
    (src/io/output.c:55) -- created by the rank that passes create=True, zero-filled."""

    def __init__(self, elements: int, shared: Optional[str] = None, create: bool = True):
        self._lib = load_library()
        if shared is not None:
            if create:
                with open(shared, "wb") as f:
                    f.truncate(4 * max(int(elements), 1))
            self.array = np.memmap(shared, dtype=np.int32, mode="r+", shape=(max(int(elements), 1),))[:int(elements)]
        else:
            # an anonymous mapping of its own, never malloc's heap: the library refuses to page-lock memory that malloc hands
            # out again, and a C host's matrix is a mapping anyway (output_load, src/io/output.c:55)
            import mmap
            self._map = mmap.mmap(-1, 4 * max(int(elements), 1))
            self.array = np.frombuffer(self._map, dtype=np.int32)[:int(elements)]
        self.path = shared
        self._registered = False
        if elements and self._lib.sa_hip_host_register(C.c_void_p(self.array.ctypes.data), self.array.nbytes):
            raise AlignError(_err())
        self._registered = bool(elements)

    @property
    def ptr(self) -> int:
        return int(self.array.ctypes.data)

    def close(self) -> None:
        if self._registered:
            self._registered = False
            try:
                self._lib.sa_hip_host_unregister(C.c_void_p(self.array.ctypes.data))
            except ImportError:  # interpreter shutting down: the process's mappings go with it
                pass

    __del__ = close


def hip_filter(store: SequenceStore, threshold: float) -> np.ndarray:
    """`filter()` replacement (reference src/bio/filter.c:14-89): boolean keep mask with the sequential semantics
    of `-f threshold`; the similarity relation is computed on the device."""
    keep = np.ones(store.num, dtype=np.uint8)
    rc = load_library().sa_hip_filter(store._as_c(), C.c_float(threshold), keep.ctypes.data)
    if rc < 0:
        raise AlignError(_err())
    return keep.astype(bool)


class Context:
    """Device-resident layer (sa_ctx_*): sequences + scoring uploaded once, ranges of the packed pair
    index computed into device buffers the caller owns (e.g. torch tensors)."""

    def __init__(self, store: SequenceStore, scoring: Scoring, device: int = 0):
        self._lib = load_library()
        self._store = store  # keep host arrays alive
        sc = scoring._as_c()
        self._h = self._lib.sa_ctx_create(int(device), store._as_c(), C.byref(sc))
        if not self._h:
            raise AlignError(_err())
        self.num = store.num
        self.device = int(device)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.sa_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def pairs(self) -> int:
        return int(self._lib.sa_ctx_pairs(self._h))

    def cells(self, start: int = 0, count: Optional[int] = None) -> int:
        return self._store.cells(start, count)

    def partition(self, parts: int) -> list[int]:
        return self._store.partition(parts)

    def align_range(self, start: int, count: int, d_scores_ptr: int, stream: int = 0) -> None:
        if self._lib.sa_ctx_align_range(self._h, start, count, C.c_void_p(d_scores_ptr), C.c_void_p(stream)):
            raise AlignError(_err())

    @property
    def scores_fit16(self) -> bool:
        """every score of this store under this scoring provably fits int16 (exchange format of the all-gather)"""
        return bool(self._lib.sa_ctx_scores_fit16(self._h))

    def align_range16(self, start: int, count: int, d_scores16_ptr: int, stream: int = 0) -> None:
        """as align_range, into an int16 device array; raises when scores_fit16 is False"""
        if self._lib.sa_ctx_align_range16(self._h, start, count, C.c_void_p(d_scores16_ptr), C.c_void_p(stream)):
            raise AlignError(_err())

    def widen16(self, d_src16_ptr: int, d_dst32_ptr: int, count: int, stream: int = 0) -> None:
        """int16 exchange format -> the reference's s32, on the device"""
        if self._lib.sa_hip_widen16(C.c_void_p(d_src16_ptr), C.c_void_p(d_dst32_ptr), count, C.c_void_p(stream)):
            raise AlignError(_err())

    # ---- tile-interleaved sharding (one process per GPU; sequencealigner_amd/distributed.py: TiledGatherStep) ----
    def leave_room(self, on: bool) -> None:
        """three instead of four persistent workgroups per CU from now on: room for a concurrent collective / placement"""
        self._lib.sa_ctx_leave_room(self._h, int(on))

    def share_elems(self, start: int, count: int, world: int, to_host: bool = False) -> int:
        """elements of one rank's dense share of the packed range (the same on every rank)"""
        v = int(self._lib.sa_ctx_share_elems(self._h, start, count, world, int(to_host)))
        if v < 0:
            raise AlignError(_err())
        return v

    def align_share(self, start: int, count: int, world: int, rank: int, d_share_ptr: int, elem16: bool, stream: int = 0,
                    host_packed_ptr: int = 0) -> None:
        """scores of `rank`'s tiles of the range, densely in tile order (int16 or s32 elements); with host_packed_ptr (the
        WHOLE page-locked packed host matrix) the same scores also go straight to host_packed[p]"""
        if self._lib.sa_ctx_align_share(self._h, start, count, world, rank, C.c_void_p(d_share_ptr), int(elem16),
                                        C.c_void_p(host_packed_ptr or None), C.c_void_p(stream)):
            raise AlignError(_err())

    def place_shares(self, start: int, count: int, world: int, d_shares_ptr: int, elem16: bool, d_packed_ptr: int, stream: int = 0,
                     to_host: bool = False) -> None:
        """gathered shares (rank-major) -> d_packed[p - start], the reference's packed order, widened to s32"""
        if self._lib.sa_ctx_place_shares(self._h, start, count, world, int(to_host), C.c_void_p(d_shares_ptr), int(elem16),
                                         C.c_void_p(d_packed_ptr), C.c_void_p(stream)):
            raise AlignError(_err())

    def align_host(self, matrix: Optional[np.ndarray], triangular: bool, start: int = 0, count: Optional[int] = None) -> float:
        """The reference's launch/copy loop (seqalign_cuda.c:182-292) on this context: scores of the packed range into
        the host matrix (a flat int32 array: packed N(N-1)/2 or full N*N; None = the reference's -W).  Returns the
        seconds the loop took (uploads, allocations and page-locking are outside it, as in the reference)."""
        count = self.pairs - start if count is None else count
        out = _Output(matrix.ctypes.data if matrix is not None else None, None, self.num, bool(triangular))
        phase = C.c_double()
        if self._lib.sa_ctx_align_host(self._h, start, count, out, C.byref(phase)):
            raise AlignError(_err())
        return float(phase.value)

    def expand_full(self, d_packed_ptr: int, d_full_ptr: int, stream: int = 0) -> None:
        if self._lib.sa_ctx_expand_full(self._h, C.c_void_p(d_packed_ptr), C.c_void_p(d_full_ptr), C.c_void_p(stream)):
            raise AlignError(_err())

    def timing(self, enable: bool) -> None:
        self._lib.sa_ctx_timing(self._h, int(enable))

    def timing_read(self) -> dict:
        """Dominant kernel since timing(True): name, launches, total ms, pairs and cells it covered."""
        name = C.create_string_buffer(256)
        launches, pairs, cells = C.c_int64(), C.c_int64(), C.c_int64()
        ms, all_ms = C.c_double(), C.c_double()
        if self._lib.sa_ctx_timing_read(self._h, name, 256, C.byref(launches), C.byref(ms), C.byref(pairs),
                                        C.byref(cells), C.byref(all_ms)):
            raise AlignError(_err())
        return dict(kernel=name.value.decode(), launches=int(launches.value), ms=float(ms.value),
                    pairs=int(pairs.value), cells=int(cells.value), all_kernels_ms=float(all_ms.value))


class DeflateJob:
    """Device-side DEFLATE of a device-resident result matrix (sa_zjob_*, the -z option): the tiles (HDF5 chunks) of
    the full symmetric matrix as zlib streams (level > 0) or as they are (level 0).  d_packed_ptr: scores by packed pair
    index; or d_full_ptr: N x N."""

    def __init__(self, num: int, chunk_dim: int, d_packed_ptr: int = 0, d_full_ptr: int = 0, device: int = 0, level: int = 6, _handle=None):
        self._lib = load_library()
        self._h = _handle or self._lib.sa_zjob_create(int(device), C.c_void_p(d_packed_ptr or None), C.c_void_p(d_full_ptr or None),
                                                      int(num), int(chunk_dim), int(level))
        if not self._h:
            raise AlignError(_err())
        self.tiles_per_row = int(self._lib.sa_zjob_tiles_per_row(self._h))

    @classmethod
    def begin(cls, store: "SequenceStore", scoring: "Scoring", chunk_dim: int, level: int = 6) -> "DeflateJob":
        """sa_hip_tiles_begin: the alignment of `store` runs on device 0 while next() hands out the finished tiles shell by shell"""
        lib = load_library()
        sc = scoring._as_c()
        h = lib.sa_hip_tiles_begin(store._as_c(), C.byref(sc), int(chunk_dim), int(level))
        if not h:
            raise AlignError(_err())
        job = cls(store.num, chunk_dim, _handle=h)
        job._store = store  # keep the host arrays alive
        return job

    def next(self) -> list[tuple[int, int, bytes]]:
        """the next batch of finished tiles as (tile row, tile column, bytes); [] when every tile has been handed out"""
        n = self.tiles_per_row
        rows, cols = (C.c_uint32 * n)(), (C.c_uint32 * n)()
        ptrs, sizes = (C.c_void_p * n)(), (C.c_size_t * n)()
        got = self._lib.sa_zjob_next(self._h, rows, cols, ptrs, sizes)
        if got < 0:
            raise AlignError(_err())
        return [(int(rows[t]), int(cols[t]), C.string_at(ptrs[t], sizes[t])) for t in range(got)]

    @property
    def align_seconds(self) -> float:
        return float(self._lib.sa_zjob_align_seconds(self._h))

    def tile_row(self, row: int) -> list[bytes]:
        """the zlib streams of tile row `row` (copied out of the job's page-locked buffer)"""
        ptrs = (C.c_void_p * self.tiles_per_row)()
        sizes = (C.c_size_t * self.tiles_per_row)()
        if self._lib.sa_zjob_tile_row(self._h, int(row), ptrs, sizes):
            raise AlignError(_err())
        return [C.string_at(ptrs[t], sizes[t]) for t in range(self.tiles_per_row)]

    def stats(self) -> dict:
        e, c, r, o = C.c_double(), C.c_double(), C.c_uint64(), C.c_uint64()
        self._lib.sa_zjob_stats(self._h, C.byref(e), C.byref(c), C.byref(r), C.byref(o))
        return {"encode_ms": e.value, "copy_ms": c.value, "raw_bytes": r.value, "out_bytes": o.value}

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.sa_zjob_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
