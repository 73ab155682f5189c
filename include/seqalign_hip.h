/*
 * include/seqalign_hip.h -- C ABI of libseqalign_hip.so (MI355X / gfx950).
 *
 * Drop-in device boundary for the all-vs-all alignment hot path of
 * jakovdev/SequenceAligner.  Plain C: pointers, sizes and PODs only, no C++ or
 * torch types.  Every entry point names the reference interface it replaces
 * (paths relative to the reference repository root).
 *
 * The reference keeps scoring state in process globals (GAP_PEN/GAP_OPN/GAP_EXT,
 * SEQ_LUT, SUB_MAT, ALIGN -- src/bio/align.h:11-19,28-42) and pushes it to the
 * device through the `pC` symbol (src/bio/kernels.cuh:12-25).  Here the same
 * state travels explicitly in `struct sa_scoring`; INTEGRATION.md shows the
 * five-line adapter that fills it from the reference's globals.
 *
 * Failure model: like the reference's CALLR/perr (src/interface/seqalign_cuda.c:23-30)
 * every call reports failure through its return value and leaves a human
 * readable message retrievable with sa_last_error() (also printed on stderr
 * when SA_HIP_VERBOSE is set).  There is NO CPU fallback anywhere behind this
 * ABI: no HIP device (or a missing code object) is an error, never a silent
 * host computation.
 */
#ifndef SEQALIGN_HIP_H
#define SEQALIGN_HIP_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SA_ABI_VERSION 4 /* 4: the sa_zjob_* / sa_hip_tiles_begin entry points */

/* ---- data types shared with the reference ------------------------------- */

/* src/bio/align.h:6-9  `struct meta` (byte offset into the blob, length w/o NUL) */
struct sa_meta {
	int32_t off;
	int32_t len;
};

/* src/io/input.h:6-11  `struct input`: one blob of uppercase residues, every
 * sequence NUL-terminated; meta[k] locates sequence k; max = longest len. */
struct sa_input {
	uint8_t *seqs;
	struct sa_meta *meta;
	int32_t max;
	int32_t num;
};

/* src/io/output.h:10-15  `struct output`.  matrix may be NULL (-W: compute,
 * copy nothing; src/interface/seqalign_cuda.c:233,268,279).
 * triangular: pair (i<j) at matrix[j*(j-1)/2 + i] (src/io/output.c:83);
 * otherwise full dim x dim, matrix[dim*i+j] = matrix[dim*j+i] = score and the
 * diagonal is left untouched (src/io/output.c:76-81). */
struct sa_output {
	int32_t *matrix;
	const char **seqs;
	size_t dim;
	bool triangular;
};

/* Alignment methods = entries of the reference's `aligns` registry
 * (src/bio/method/nw.c:46-51, ga.c:92-98, sw.c:66-71). */
enum sa_method { SA_METHOD_NW = 0, SA_METHOD_GA = 1, SA_METHOD_SW = 2, SA_METHOD_COUNT = 3 };
/* src/bio/align.h:34-37 */
enum sa_gap_kind { SA_GAP_LINEAR = 0, SA_GAP_AFFINE = 1 };

#define SA_LUT_SIZE 128 /* src/bio/align.h:11  SEQ_LUT_SIZE */
#define SA_SUB_DIM 24   /* src/bio/align.h:13  SUB_MAT_DIM  */
#define SA_SCORE_MIN (INT32_MIN / 2) /* src/bio/align.h:19 */

/* Replaces the globals read by cuda_align (src/interface/seqalign_cuda.c:115-123,170).
 * Gap values are the STORED form, i.e. already negated (src/bio/align.c:127-128):
 * `-p 4` -> gap_pen = -4; `-s 10 -e 1` -> gap_opn = -10, gap_ext = -1. */
struct sa_scoring {
	int32_t method;                        /* enum sa_method                   (ALIGN)    */
	int32_t gap_pen;                       /* linear gap, used by NW           (GAP_PEN)  */
	int32_t gap_opn;                       /* affine open, used by GA and SW   (GAP_OPN)  */
	int32_t gap_ext;                       /* affine extend, used by GA and SW (GAP_EXT)  */
	int32_t lut[SA_LUT_SIZE];              /* ASCII -> 0..23, -1 invalid       (SEQ_LUT)  */
	int32_t sub[SA_SUB_DIM * SA_SUB_DIM];  /* row-major 24x24                  (SUB_MAT)  */
};

/* ---- the two reference entry points ------------------------------------- */

/* Replaces `bool cuda_memory(size_t bytes)` (src/interface/seqalign_cuda.h:7,
 * src/interface/seqalign_cuda.c:71-93): true iff the device has bytes*4/3 free --
 * on EVERY device sa_hip_align will use (all visible ones, or the first SA_HIP_DEVICES).
 * Caller: output_load (src/io/output.c:37) to choose full vs triangular. */
bool sa_hip_memory(size_t bytes);

/* Replaces `bool cuda_align(struct input, struct output)`
 * (src/interface/seqalign_cuda.h:9, src/interface/seqalign_cuda.c:95-296).
 * Aligns every pair i<j of `in` and fills out.matrix in the layout out.triangular
 * selects.  Caller owns in/out for the duration of the call; all device memory
 * is allocated and released inside.  Uses every visible device when
 * SA_HIP_DEVICES is unset, or the first n with SA_HIP_DEVICES=n.  With several
 * devices the pair space is tiled over them (the job-wide tile list dealt by DP
 * work), the dense shares are all-gathered over RCCL so that every device holds
 * the whole matrix, and the host matrix is written once over all PCIe links;
 * when RCCL cannot be bound or the matrix does not fit a device, every device
 * delivers a contiguous slice instead (sa_hip_last_align_path tells which).
 * Never throws and never aborts the host process: any failure -- a C++
 * exception behind the boundary included -- is `false` + sa_last_error(). */
bool sa_hip_align(struct sa_input in, struct sa_output out, const struct sa_scoring *sc);

/* Device-assisted replacement of `bool filter(struct input *)` (src/bio/filter.c:14-89, the `-f` option):
 * keep[k] (in.num bytes) receives 1 for sequences that survive, 0 for dropped ones, with the reference's
 * SEQUENTIAL semantics (for j ascending, j is dropped iff some kept i<j has
 * matches(first min(len))/min(len) >= threshold in float).  The O(N^2 L) relation is computed on the device
 * as a bit matrix, the order-dependent keep/drop on the host.  Returns the number kept, <0 on error.
 * threshold <= 0 keeps everything (filter.c:16-17).  The caller compacts its store (filter.c:66-79). */
int32_t sa_hip_filter(struct sa_input in, float threshold, uint8_t *keep);

/* ---- device-resident layer (what sa_hip_align is built from) -------------
 * Used by multi-process drivers (one process per GPU + RCCL all-gather of the
 * packed slices, bench.py) and by callers that keep results in HBM. */

typedef struct sa_ctx sa_ctx;

/* Uploads the sequence store + scoring tables to `device` and plans the run
 * (src/interface/seqalign_cuda.c:115-132,168).  NULL on failure. */
sa_ctx *sa_ctx_create(int device, struct sa_input in, const struct sa_scoring *sc);
void sa_ctx_destroy(sa_ctx *ctx);

/* N(N-1)/2 (src/util/macros.h:13 `alignments`) */
int64_t sa_ctx_pairs(const sa_ctx *ctx);

/* Scores of packed pair indices [start, start+count) into DEVICE memory
 * d_scores[0..count) (= the reference's `kernel(scores, start, batch)`,
 * src/bio/align.h:48, src/bio/kernels.cu:32-40,73).  Asynchronous on `stream`
 * (a hipStream_t, NULL = default stream). 0 on success. */
int sa_ctx_align_range(sa_ctx *ctx, int64_t start, int64_t count, int32_t *d_scores, void *stream);

/* The launch/copy loop of cuda_align (src/interface/seqalign_cuda.c:182-292) on a ready context: scores of the
 * packed range [start, start+count) delivered into the HOST matrix `out` (the WHOLE matrix: packed element p
 * at out.matrix[p], or full dim x dim; out.matrix == NULL computes and copies nothing, the reference's -W).
 * Packed destination: double-buffered batches, each batch's device->host copy overlapping the next batch's
 * kernels.  Full destination with a column-aligned range and N^2 ints of free HBM: the L-shaped shell of every
 * column batch is expanded on the device and copied while the next batch computes; otherwise batches are
 * scattered by the host like output_fill (src/io/output.c:76-81).  Device buffers, streams and the page-locking
 * of a destination that is not yet page-locked are set up BEFORE the timed phase (the reference's allocations and
 * uploads are outside bench_align_start..end as well); *phase_seconds receives the duration of the loop itself.
 * sa_hip_align = sa_ctx_create + sa_ctx_align_host per device + sa_ctx_destroy.  0 on success. */
int sa_ctx_align_host(sa_ctx *ctx, int64_t start, int64_t count, struct sa_output out, double *phase_seconds);

/* Page-locks / releases a host range for device->host DMA (hipHostRegister).  A host that allocates its result
 * matrix once (output_load, src/io/output.c:55) registers it there; sa_ctx_align_host / sa_hip_align detect a
 * registered destination and skip their own temporary registration.  0 on success.  The range must be a mapping of
 * the caller's own -- start on a page boundary, outside the malloc heap, as output_load's mmap does: memory that malloc
 * manages is refused here and never page-locked by the library itself (a destination there is filled through the library's
 * pinned staging buffers); page-locking blocks that share pages with their heap neighbours is what both GPU memory faults
 * on this project's record have in common. */
int sa_hip_host_register(void *p, size_t bytes);
int sa_hip_host_unregister(void *p);

/* Exchange format for the multi-GPU all-gather (no reference counterpart: SURVEY 8e): the same scores as
 * int16 when they provably fit -- every |score| <= max_len * max|S| + 2 * max_len * max|gap| <= 32767 for this
 * store and scoring (sa_ctx_scores_fit16) -- so that the collective moves half the bytes; sa_hip_widen16 turns
 * the gathered vector back into the reference's s32 on the device.  sa_ctx_align_range16 fails when the bound
 * does not hold. */
int sa_ctx_scores_fit16(const sa_ctx *ctx);
int sa_ctx_align_range16(sa_ctx *ctx, int64_t start, int64_t count, int16_t *d_scores, void *stream);
int sa_hip_widen16(const int16_t *d_src, int32_t *d_dst, int64_t count, void *stream);

/* Tile-interleaved sharding for one process per GPU (no reference counterpart: SURVEY 8e; replaces the contiguous
 * range abstraction `kernel(scores, start, batch)` of src/bio/kernels.cu:32-40 in the multi-GPU path).  The launch
 * plan of the packed range [start, start+count) -- the largest-first list of workgroup-tiles of every column-length
 * class -- is the same on every rank; its tiles are dealt over `world` ranks by accumulated DP work, so every rank
 * keeps full-size tiles and whole arranged row blocks.  Rank `rank` scores its tiles and stores them densely, in tile
 * order, into d_share: sa_ctx_share_elems() elements (the same on every rank) of int16 (elem16 != 0; needs
 * sa_ctx_scores_fit16) or s32.  After an all-gather of the shares (rank-major, world x share_elems elements),
 * sa_ctx_place_shares widens and places them: d_packed[p - start] = score of pair p, the reference's packed order
 * (src/io/output.c:83).  Host delivery (the device->host copies inside the reference's timed loop,
 * src/interface/seqalign_cuda.c:266-283): with host_packed != NULL -- the WHOLE packed host matrix, page-locked
 * (sa_hip_host_register), e.g. one shared mapping all ranks of a node attach -- the kernels of a rank also store its
 * own scores straight into host_packed[p], as sa_ctx_align_host does for one device: together the ranks fill the
 * matrix exactly once, with no copy pass and without waiting for the gather.  `to_host` of the other two calls must
 * say whether the shares are computed that way (the plan differs: tiles are then their own arranged row blocks).
 * world = 1 is allowed (one share holding every tile).  All asynchronous on `stream`. */
int64_t sa_ctx_share_elems(sa_ctx *ctx, int64_t start, int64_t count, int world, int to_host);
int sa_ctx_align_share(sa_ctx *ctx, int64_t start, int64_t count, int world, int rank, void *d_share, int elem16,
		       int32_t *host_packed, void *stream);
int sa_ctx_place_shares(sa_ctx *ctx, int64_t start, int64_t count, int world, int to_host, const void *d_shares, int elem16,
			int32_t *d_packed, void *stream);

/* on != 0: the launches this context issues from now on run three instead of four persistent workgroups per CU, leaving
 * LDS and wave slots for kernels of OTHER streams to run beside them -- the RCCL all-gather and sa_ctx_place_shares of the
 * previous super-chunk in an overlapped multi-GPU schedule.  (Four per CU fill the LDS: a concurrent kernel then waits
 * for the launch to end.)  Costs the NW kernels ~8 %, Gotoh / SW ~1 %; off by default. */
void sa_ctx_leave_room(sa_ctx *ctx, int on);

/* Packed triangular (device) -> full symmetric dim x dim with zero diagonal
 * (device), the layout of src/io/output.c:76-81.  Asynchronous on `stream`. */
int sa_ctx_expand_full(sa_ctx *ctx, const int32_t *d_packed, int32_t *d_full, void *stream);

/* ---- compressed output on the device (the -z option) ----------------------
 * The reference asks libhdf5 for DEFLATE on the chunked /similarity_matrix (src/io/format/hdf5.c:91-95) and H5Dwrite
 * (:148-194) then deflates every chunk in the one writing thread -- for BASELINE config 5 (89 994 sequences, 484 chunks
 * of 4096 x 4096, level 6) 41 CPU-minutes behind a two-second alignment.  Here the chunks ("tiles") are encoded where
 * the scores are: a job walks the tile rows of the full symmetric matrix (zero diagonal, zeros beyond N: exactly what
 * H5Dwrite hands the filter) over a DEVICE-resident matrix -- packed by pair index (d_packed) or full N x N (d_full,
 * used when d_packed is NULL) -- and returns every tile as a complete zlib stream (RFC 1950 / 1951) in page-locked host
 * memory, ready for H5Dwrite_chunk (filter mask 0); any inflate reads them.  The parse is fixed: the ratio
 * on score matrices sits between zlib -1 and -6, whatever level > 0 the dataset's property list names.  Level 0 returns the
 * tiles as they are (chunk_dim^2 int32 each), for a chunked dataset without filters: the same H5Dwrite_chunk loop then
 * replaces H5Dwrite's gather of every tile out of N-wide rows in the writing thread (hdf5.c:148-194).
 *   sa_zjob_create       buffers for one line of tiles; chunk_dim: a power of two in [64, 4096] (sa_host_hdf5_chunk_dim)
 *   sa_zjob_tile_row     streams[t] / sizes[t] for the tiles_per_row tiles of row `tile_row`, valid until the next call;
 *                        rows are asked for in order, each once: the next row is encoded while the caller writes
 *   sa_hip_tiles_begin   sa_ctx_create + the packed matrix in device memory + a job that walks the tiles in
 *                        SHELLS while the alignment is still running: the tiles whose larger tile index is b need exactly
 *                        the columns [b * chunk_dim, (b + 1) * chunk_dim), so they are encoded and handed out while the
 *                        device aligns the next column blocks (the alignment phase and the reference's output phase,
 *                        src/main.c:31-34, overlap).  The first blocks are on their way when it returns.  With several
 *                        devices (all visible ones, or the first SA_HIP_DEVICES) block b and its shell belong to device
 *                        b mod n: a shell needs nothing but its own block, so the devices exchange nothing and every one
 *                        drives its own PCIe link; the shells still come in ascending order.
 *   sa_zjob_next         the next batch of finished tiles (at most tiles_per_row): rows[t], cols[t] = tile coordinates,
 *                        streams[t] / sizes[t] valid until the next call; returns their number, 0 when every tile has been
 *                        handed out, < 0 on error.  Works on a sa_zjob_create job as well (then: row after row).
 *   sa_zjob_align_seconds  device time of the alignment so far (sum over the finished column blocks: the bracket of
 *                        sa_hip_last_align_seconds).
 * NULL / non-zero + sa_last_error on failure. */
typedef struct sa_zjob sa_zjob;
sa_zjob *sa_zjob_create(int device, const int32_t *d_packed, const int32_t *d_full, int32_t num, size_t chunk_dim, int level);
size_t sa_zjob_tiles_per_row(const sa_zjob *job);
int sa_zjob_tile_row(sa_zjob *job, size_t tile_row, const uint8_t **streams, size_t *sizes);
int sa_zjob_next(sa_zjob *job, uint32_t *rows, uint32_t *cols, const uint8_t **streams, size_t *sizes);
void sa_zjob_stats(const sa_zjob *job, double *encode_ms, double *copy_ms, uint64_t *raw_bytes, uint64_t *out_bytes);
double sa_zjob_align_seconds(const sa_zjob *job);
void sa_zjob_destroy(sa_zjob *job);
sa_zjob *sa_hip_tiles_begin(struct sa_input in, const struct sa_scoring *sc, size_t chunk_dim, int level);

/* ---- pair-space planning (host only, no device needed) -------------------
 * DP cells (sum of len_i*len_j) of the packed pair range [start, start+count),
 * the numerator of GCUPS; -1 on a bad range. */
int64_t sa_pairs_cells(const struct sa_meta *meta, int32_t num, int64_t start, int64_t count);
/* Splits [0, N(N-1)/2) into `parts` contiguous ranges of near-equal DP work
 * (cells); bounds[parts+1] receives the cut points (bounds[0]=0,
 * bounds[parts]=pairs).  Multi-GPU sharding rule, SURVEY.md §8(e): the
 * reference's own batch abstraction kernel(scores, start, batch) is a range. */
int sa_pairs_partition(const struct sa_meta *meta, int32_t num, int parts, int64_t *bounds);

/* Instrumentation for bench.py.  While enabled every kernel launch of
 * sa_ctx_align_range is bracketed by a HIP-event pair on the launch stream.
 * sa_ctx_timing_read reports the DOMINANT kernel (largest accumulated time)
 * since the last enable: its name, launch count, accumulated milliseconds and
 * the pairs / DP cells those launches covered, plus the sum over all kernels. */
void sa_ctx_timing(sa_ctx *ctx, int enable);
int sa_ctx_timing_read(sa_ctx *ctx, char *kernel_name, int cap, int64_t *launches, double *total_ms,
		       int64_t *pairs, int64_t *cells, double *all_kernels_ms);

/* ---- option tables (host only, no device needed) ------------------------ */

/* Replaces parse_matrix (src/bio/matrices.c:44-58): case-insensitive name ->
 * SEQ_LUT + SUB_MAT.  0 on success. */
int sa_matrix_load(const char *name, int32_t lut[SA_LUT_SIZE], int32_t sub[SA_SUB_DIM * SA_SUB_DIM]);
/* -l / list_matrices (src/bio/matrices.c:27-33) */
int sa_matrix_count(void);
const char *sa_matrix_name(int index);
int sa_matrix_is_nucleotide(int index);

/* Replaces parse_align (src/bio/align.c:87-96): alias ("nw", "Needleman-Wunsch",
 * ... case-insensitive) -> enum sa_method, -1 if unknown. */
int sa_method_parse(const char *alias);
const char *sa_method_name(int method);   /* long alias, e.g. "Gotoh"       */
int sa_method_gap_kind(int method);       /* enum sa_gap_kind               */

/* ---- misc ---------------------------------------------------------------- */
/* Seconds the launch/copy loop of the last successful sa_hip_align() took: the phase the reference
 * brackets with bench_align_start()/bench_align_end() (src/interface/seqalign_cuda.c:182,292 --
 * uploads, allocations and context set-up are outside it, the device->host copies inside). */
double sa_hip_last_align_seconds(void);
/* Which schedule the last successful sa_hip_align() took: 1 = every device delivers a contiguous slice of the packed
 * index straight into the host matrix (one device: the whole range); 2 = tile-interleaved dense shares + RCCL
 * all-gather + placement on every device, then delivery (several devices); 0 = no call yet. */
int sa_hip_last_align_path(void);

/* Progress of a running sa_hip_align / sa_ctx_align_host, the reference's ppercent / pproportc side channel
 * (src/interface/seqalign_cuda.c:181,286-289,293): fn(fraction in [0,1], user) is called while the host waits for the
 * device -- per batch on the batched paths, every 50 ms from the launches' tile counters on the single-launch path --
 * from the calling thread when one device is used, from ONE worker thread of the library (the first slice's) when
 * sa_hip_align spreads the job over several: calls never overlap, but the callback must not assume the caller's thread.
 * NULL (the default) reports nothing and polls nothing; the polling waits in 250 us slices and never past completion. */
typedef void (*sa_progress_fn)(double fraction, void *user);
void sa_hip_set_progress(sa_progress_fn fn, void *user);

/* Where the time of the last successful sa_hip_align() went, in milliseconds (its first slice): ms[k] for k of
 * enum sa_breakdown; returns the number of entries written.  Everything but SA_BREAKDOWN_PHASE is set-up the reference
 * keeps outside bench_align_start()/bench_align_end() as well (src/interface/seqalign_cuda.c:115-168). */
enum sa_breakdown {
	SA_BREAKDOWN_ENCODE = 0,   /* validate + residue -> index (the reference uploads SEQ_LUT instead)        */
	SA_BREAKDOWN_DEVICE,       /* hipSetDevice / runtime bring-up                                            */
	SA_BREAKDOWN_UPLOAD,       /* device allocations, uploads, streams and events                            */
	SA_BREAKDOWN_CODE_OBJECTS, /* loading the kernel families this store needs                               */
	SA_BREAKDOWN_PIN,          /* page-locking the destination (skipped when the caller registered it)       */
	SA_BREAKDOWN_PLAN,         /* launch plan of the range (tile lists)                                      */
	SA_BREAKDOWN_ARRANGE,      /* arranged copies of the row store                                           */
	SA_BREAKDOWN_PHASE,        /* the launch/copy loop = sa_hip_last_align_seconds()                         */
	SA_BREAKDOWN_TOTAL,        /* create + align_host of the slice                                           */
	SA_BREAKDOWN_COUNT
};
int sa_hip_last_align_breakdown(double *ms, int n);

int sa_hip_device_count(void);
const char *sa_hip_device_name(int device);
const char *sa_last_error(void);
int sa_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SEQALIGN_HIP_H */
