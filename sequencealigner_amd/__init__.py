"""sequencealigner_amd -- MI355X-native all-vs-all pairwise alignment (NW / Gotoh / SW score matrices).

Python mirror of the reference's device interface (jakovdev/SequenceAligner,
src/interface/seqalign_cuda.h) on top of the C ABI in include/seqalign_hip.h:

    reference (C)                                this package
    ------------------------------------------   ------------------------------------------
    struct input  {seqs, meta, max, num}         SequenceStore        (io/input.h:6-11)
    struct output {matrix, seqs, dim, triangular} numpy matrix + `triangular` flag (io/output.h:10-15)
    globals ALIGN/GAP_*/SEQ_LUT/SUB_MAT          Scoring              (bio/align.h:11-19)
    -a/-m/-p/-s/-e parse+validate                Scoring.from_names   (bio/align.c:87-142, bio/matrices.c:44-58)
    bool cuda_memory(size_t)                     hip_memory(bytes)    (seqalign_cuda.h:7)
    bool cuda_align(struct input, struct output) hip_align(store, scoring, triangular)  (seqalign_cuda.h:9)
    bool filter(struct input *)                  hip_filter(store, threshold)           (bio/filter.c:14)
    kernel(scores, start, batch)                 Context.align_range  (bio/align.h:48)

All compute happens in libseqalign_hip.so (hand-written HIP for gfx950).  There is no
Python/CPU fallback: a missing library or device raises.
"""
from .binding import (  # noqa: F401
    AlignError,
    Context,
    DeflateJob,
    PinnedMatrix,
    Scoring,
    SequenceStore,
    device_count,
    device_name,
    last_align_breakdown,
    last_align_path,
    last_align_seconds,
    hip_align,
    hip_filter,
    hip_memory,
    library_path,
    load_library,
    matrix_names,
    method_names,
    pair_count,
    set_progress,
)

__all__ = [
    "AlignError", "Context", "PinnedMatrix", "Scoring", "SequenceStore", "device_count", "device_name", "last_align_breakdown", "last_align_path", "last_align_seconds", "hip_align", "hip_filter",
    "hip_memory", "library_path", "load_library", "matrix_names", "method_names", "pair_count", "set_progress",
]
