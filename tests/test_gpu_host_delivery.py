"""GPU (-m gpu): sa_ctx_align_host -- the launch/copy loop of cuda_align (reference src/interface/seqalign_cuda.c:182-292)
on a ready context -- against the oracle: packed and full destinations, page-locked and pageable, sub-ranges,
the shell schedule of the full layout and its host-scatter fallback, -W (no destination), several devices."""
import numpy as np
import pytest

from tests.golden_util import tri_to_full
from tests.synth import make_dna_set, make_protein_set

pytestmark = pytest.mark.gpu

CASES = [("nw", "blosum62", dict(gap_pen=4)), ("ga", "blosum62", dict(gap_open=10, gap_extend=1)),
         ("sw", "blosum62", dict(gap_open=10, gap_extend=1))]


def tri(j):
    return j * (j - 1) // 2


@pytest.mark.parametrize("method,matrix,gaps", CASES)
def test_packed_and_full_delivery_match_oracle(method, matrix, gaps, sa, oracle):
    seqs = make_protein_set(900, 20, 170, 31)  # 404 550 pairs: several shrinking batches would need > 3 Mi; see below
    store = sa.SequenceStore.from_sequences(seqs)
    scoring = sa.Scoring.from_names(method, matrix, **gaps)
    want = oracle.align(store, scoring, triangular=True)
    n = store.num
    with sa.Context(store, scoring, 0) as ctx:
        for pinned in (True, False):
            dest = sa.PinnedMatrix(store.pairs) if pinned else None
            arr = dest.array if pinned else np.zeros(store.pairs, np.int32)
            phase = ctx.align_host(arr, triangular=True)
            assert phase > 0 and np.array_equal(arr, want)
            arr[:] = -7  # a sub-range touches nothing outside it
            lo, cnt = 12345, 200000
            ctx.align_host(arr, triangular=True, start=lo, count=cnt)
            assert np.array_equal(arr[lo:lo + cnt], want[lo:lo + cnt]) and (arr[:lo] == -7).all() and (arr[lo + cnt:] == -7).all()
            if dest is not None:
                dest.close()
        full = np.full(n * n, -1, np.int32)
        ctx.align_host(full, triangular=False)  # shell schedule: diagonal written as 0
        assert np.array_equal(full.reshape(n, n), tri_to_full(want, n))
        # a column-aligned slice of the full layout (what one device of several delivers): only its shell is written
        ja, jb = 300, 701
        full[:] = -1
        ctx.align_host(full, triangular=False, start=tri(ja), count=tri(jb) - tri(ja))
        f = full.reshape(n, n)
        ref = tri_to_full(want, n)
        shell = np.zeros((n, n), bool)
        shell[ja:jb, :jb] = True
        shell[:ja, ja:jb] = True
        assert np.array_equal(f[shell], ref[shell]) and (f[~shell] == -1).all()
        # a range that is not column-aligned: staged batches scattered by the host (output_fill order)
        full[:] = 0
        lo, cnt = 1000, 300001
        ctx.align_host(full, triangular=False, start=lo, count=cnt)
        part = np.zeros(store.pairs, np.int32)
        part[lo:lo + cnt] = want[lo:lo + cnt]
        assert np.array_equal(full.reshape(n, n), tri_to_full(part, n))
        assert ctx.align_host(None, triangular=True) > 0  # -W: compute, copy nothing


def test_many_shrinking_batches_and_shell_fallback(sa, oracle, monkeypatch):
    """8 M pairs: the delivery loop runs several geometrically shrinking batches (packed) / column shells (full);
    SA_HIP_NO_SHELLS forces the full layout through the host-scatter path."""
    seqs = make_dna_set(4000, 8, 24, 77)
    store = sa.SequenceStore.from_sequences(seqs)
    scoring = sa.Scoring.from_names("sw", "nuc44", gap_open=10, gap_extend=1)
    want = oracle.align(store, scoring, triangular=True, threads=16)
    n = store.num
    with sa.Context(store, scoring, 0) as ctx:
        arr = np.zeros(store.pairs, np.int32)
        ctx.align_host(arr, triangular=True)
        assert np.array_equal(arr, want)
        full = np.full(n * n, -1, np.int32)
        ctx.align_host(full, triangular=False)
        assert np.array_equal(full.reshape(n, n), tri_to_full(want, n))
    monkeypatch.setenv("SA_HIP_NO_SHELLS", "1")
    with sa.Context(store, scoring, 0) as ctx:
        full = np.zeros(n * n, np.int32)
        ctx.align_host(full, triangular=False)
        assert np.array_equal(full.reshape(n, n), tri_to_full(want, n))


def test_two_physical_devices_when_visible(sa, oracle):
    """sa_hip_align over >= 2 hipSetDevice targets (one host thread and one context per device, each delivering its
    slice / its column shells straight into the host matrix).  One-GPU boxes skip; SA_HIP_SPLIT covers the same code
    with every slice folded onto device 0 (test_gpu_parity.py::test_multi_device_driver_path)."""
    if sa.device_count() < 2:
        pytest.skip("needs two visible devices")
    store = sa.SequenceStore.from_sequences(make_protein_set(1500, 30, 200, 5))
    for method, matrix, gaps in CASES:
        scoring = sa.Scoring.from_names(method, matrix, **gaps)
        want = oracle.align(store, scoring, triangular=True, threads=16)
        assert np.array_equal(sa.hip_align(store, scoring, triangular=True), want)
        assert np.array_equal(sa.hip_align(store, scoring, triangular=False), tri_to_full(want, store.num))
    assert sa.hip_memory(1 << 20)


def test_progress_side_channel(sa, oracle):
    """sa_hip_set_progress: fractions in [0, 1], never decreasing, while the result stays the reference's (the single-launch
    direct-store path polls the launches' tile counters; the batched paths report per batch)"""
    from tests.synth import make_dna_set
    seqs = make_dna_set(9000, 120, 180, 4)  # ~100 ms of SW: a few 50 ms polls
    store = sa.SequenceStore.from_sequences(seqs)
    scoring = sa.Scoring.from_names("sw", "nuc44", gap_open=10, gap_extend=1)
    seen = []
    sa.set_progress(seen.append)
    try:
        dest = sa.PinnedMatrix(store.pairs)
        try:
            with sa.Context(store, scoring, 0) as ctx:
                ctx.align_host(dest.array, triangular=True)
            direct = list(seen)
            idx = np.sort(np.random.default_rng(3).integers(0, store.pairs, 20000))
            assert np.array_equal(dest.array[idx], oracle.align_pairs(store, scoring, idx, threads=16))
        finally:
            dest.close()
        del seen[:]
        small = store.prefix(1500)
        full = sa.hip_align(small, scoring, triangular=False)  # full layout: shells, per batch
        assert np.array_equal(full, full.T)
    finally:
        sa.set_progress(None)
    assert direct, "no progress was reported during a ~100 ms launch"
    for fr in (direct, seen):
        assert all(0.0 <= f <= 1.0 for f in fr) and all(a <= b for a, b in zip(fr, fr[1:]))


def test_progress_polling_stays_out_of_the_timed_phase(sa):
    """A listener must not change what it observes: the phase (the reference's bench_align bracket) of a SHORT job with a
    progress callback set stays within ~1.5 ms of the phase without one.  (Round 3's poll slept 50 ms inside the phase:
    cfg 2's 15 ms loop reported -- and took -- 50 ms whenever the CLI showed its progress line.)"""
    from tests.synth import make_config
    seqs, cfg = make_config("cfg2", 6000)  # ~5 ms of NW
    store = sa.SequenceStore.from_sequences(seqs)
    scoring = sa.Scoring.from_names(cfg["method"], cfg["matrix"], **cfg["gaps"])
    dest = sa.PinnedMatrix(store.pairs)
    full = sa.PinnedMatrix(store.num * store.num)
    try:
        with sa.Context(store, scoring, 0) as ctx:
            def best(matrix, triangular):
                return min(ctx.align_host(matrix, triangular=triangular) for _ in range(7))
            ctx.align_host(dest.array, triangular=True)
            ctx.align_host(full.array, triangular=False)
            quiet, quiet_full = best(dest.array, True), best(full.array, False)
            calls = []
            sa.set_progress(calls.append)
            try:
                loud, loud_full = best(dest.array, True), best(full.array, False)
            finally:
                sa.set_progress(None)
        assert loud < quiet + 1.5e-3, f"direct stores: {quiet * 1e3:.2f} ms without a listener, {loud * 1e3:.2f} ms with one"
        assert loud_full < quiet_full + 2.5e-3, f"shells: {quiet_full * 1e3:.2f} ms without a listener, {loud_full * 1e3:.2f} ms with one"
        assert all(0.0 <= f <= 1.0 for f in calls)
    finally:
        dest.close()
        full.close()


def test_destination_registered_in_pieces_with_a_hole_is_not_stored_into_directly(sa, oracle):
    """a packed destination whose first and last bytes are page-locked but which is NOT one registration (two
    registered pieces, an unregistered hole between them) must not take the direct-store path -- the kernels would fault
    on the hole.  The range check asks the runtime for the registration that holds the first byte and requires it to reach
    past the last one; the call falls back to staged copies and still delivers the reference's scores."""
    import ctypes as C
    from tests.synth import make_protein_set
    store = sa.SequenceStore.from_sequences(make_protein_set(2500, 40, 120, 61))
    scoring = sa.Scoring.from_names("nw", "blosum62", gap_pen=4)
    lib = sa.load_library()
    page = 4096
    import mmap
    backing = mmap.mmap(-1, 4 * (store.pairs + 2 * page))  # (a mapping of its own: the library refuses to page-lock malloc's heap)
    raw = np.frombuffer(backing, np.int32)
    start = (-raw.ctypes.data) % page // 4  # a page-aligned window inside the buffer
    matrix = raw[start:start + store.pairs]
    nbytes = matrix.nbytes
    third = nbytes // 3 // page * page
    tail_off = (nbytes - third) // page * page  # (both pieces start on a page boundary: the library locks nothing else)
    head, tail = matrix.ctypes.data, matrix.ctypes.data + tail_off
    assert lib.sa_hip_host_register(C.c_void_p(head), third) == 0
    assert lib.sa_hip_host_register(C.c_void_p(tail), third) == 0
    try:
        with sa.Context(store, scoring, 0) as ctx:
            ctx.align_host(matrix, triangular=True)
        idx = np.sort(np.random.default_rng(5).integers(0, store.pairs, 30000))
        assert np.array_equal(matrix[idx], oracle.align_pairs(store, scoring, idx, threads=16))
        lo, hi = third // 4 - 1000, tail_off // 4 + 1000  # across both edges of the hole
        assert np.array_equal(matrix[lo:lo + 2000], oracle.align_range(store, scoring, lo, 2000))
        assert np.array_equal(matrix[hi - 2000:hi], oracle.align_range(store, scoring, hi - 2000, 2000))
    finally:
        lib.sa_hip_host_unregister(C.c_void_p(head))
        lib.sa_hip_host_unregister(C.c_void_p(tail))


def test_memory_of_the_malloc_heap_is_never_page_locked(sa, oracle):
    """both GPU memory faults on record hit an address inside the process's brk heap.  The library no longer
    registers such memory -- sa_hip_host_register refuses it, and a destination there is delivered through the staging
    buffers (same scores)."""
    import ctypes as C
    from tests.synth import make_protein_set
    lib = sa.load_library()
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    libc.free.argtypes = [C.c_void_p]
    store = sa.SequenceStore.from_sequences(make_protein_set(150, 30, 90, 77))
    scoring = sa.Scoring.from_names("nw", "blosum62", gap_pen=4)
    nbytes = 4 * store.pairs  # 44 700 bytes: below every mmap threshold, so malloc serves it from the heap
    p = libc.malloc(nbytes)
    try:
        # (late in a long-lived process malloc may serve it from a heap extension it mapped itself, without the "[heap]" label:
        # either way the block does not start on a page boundary, which is what the library goes by for those)
        assert p % 4096 != 0
        assert lib.sa_hip_host_register(C.c_void_p(p), nbytes) != 0
        page = (p + 4095) // 4096 * 4096  # ... and a page-aligned piece of the heap proper is refused by its address
        heap = [tuple(int(x, 16) for x in ln.split()[0].split("-")) for ln in open("/proc/self/maps") if "[heap]" in ln]
        if any(lo <= page and page + 4096 <= hi for lo, hi in heap) and page + 4096 <= p + nbytes:
            assert lib.sa_hip_host_register(C.c_void_p(page), 4096) != 0
        assert "malloc heap" in sa.binding._err()
        matrix = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int32)), shape=(store.pairs,))
        with sa.Context(store, scoring, 0) as ctx:
            ctx.align_host(matrix, triangular=True)
        assert np.array_equal(matrix, oracle.align(store, scoring, triangular=True))
    finally:
        libc.free(p)
