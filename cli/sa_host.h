/*
 * cli/sa_host.h -- host side of the `seqalign` command line tool (plain C, no device code).
 *
 * Mirrors the reference's input/output layer around the device boundary:
 *   sa_host_load      <- input_load        src/io/input.c:28-93  (+ parsers src/io/source/{fasta,dsv}.c)
 *   sa_host_filter    <- filter            src/bio/filter.c:14-89 (sequential semantics)
 *   sa_host_matrix_*  <- output_load/free  src/io/output.c:16-66,101-106
 *   sa_host_write_hdf5<- flush_hdf5        src/io/format/hdf5.c:14-202
 * Everything returns 0 on success and leaves a message in sa_host_error() otherwise.
 */
#ifndef SA_HOST_H
#define SA_HOST_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#include "../include/seqalign_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

const char *sa_host_error(void);

/* The reference's `struct input` plus ownership (io/input.h:6-11). */
struct sa_host_store {
	struct sa_input in; /* seqs: NUL-separated uppercase blob, meta[k] = {off,len} */
	size_t blob_bytes;
};

/* Parse a whole file image already in memory (used by tests) or a file on disk.
 * `ext` selects the parser like the reference does (extension without the dot):
 * fasta fa fas fna ffn faa frn mpfa | csv tsv ssv psv.  `lut` is SEQ_LUT (residue validation,
 * fasta.c:57-63).  DSV: the sequence column is found by header name (dsv.c:21-24); when no
 * header matches, `dsv_column` (0-based) is used if >= 0, with `dsv_has_header` telling whether
 * the first row is a header -- the non-interactive form of the reference's prompt (dsv.c:128-151);
 * -1 makes that situation an error. */
int sa_host_parse(const uint8_t *data, size_t size, const char *ext, const int32_t lut[SA_LUT_SIZE],
		  int32_t gap_for_length_limit, int dsv_column, int dsv_has_header, struct sa_host_store *out);
int sa_host_load(const char *path, const int32_t lut[SA_LUT_SIZE], int32_t gap_for_length_limit,
		 int dsv_column, int dsv_has_header, struct sa_host_store *out);
void sa_host_store_free(struct sa_host_store *s);

/* -f threshold: for j ascending drop j if some KEPT i<j has matches(first min(len))/min(len) >= thr;
 * compacts blob + meta in place (filter.c:66-79).  Returns kept count (<0 on error). */
int32_t sa_host_filter(struct sa_host_store *s, float threshold, int threads);

/* Compacts blob + meta in place keeping the sequences with keep[k] != 0 (filter.c:66-79); returns the
 * number kept (<0 when fewer than 2 remain). */
int32_t sa_host_compact(struct sa_host_store *s, const uint8_t *keep);

/* Result matrix (output.c:16-66, os.c:32-141): zero-filled mapping of 4*N*N or 4*N(N-1)/2 bytes -- anonymous, or of an
 * unnamed temporary file when `file_backed` (what the reference does once the full matrix exceeds 3/4 of
 * MemAvailable, sa_host_matrix_needs_file; it then also stores the matrix triangular). */
int32_t *sa_host_matrix_alloc(size_t num, bool triangular, bool file_backed);
void sa_host_matrix_free(int32_t *m, size_t num, bool triangular);
size_t sa_host_available_memory(void); /* MemAvailable, os.c:262-295; SA_HOST_MEM_AVAILABLE=<bytes> overrides (tests) */
bool sa_host_matrix_needs_file(size_t num);

/* HDF5: /sequences (N vlen C strings) + /similarity_matrix (N x N I32LE, symmetric, zero diagonal);
 * chunked only when N > 256, chunk = clamp(largest 64*2^k <= N, 256, 4096), deflate level z on the
 * chunked dataset; libver latest, 4 KiB alignment (hdf5.c:16-18,70-89).  A packed triangular matrix
 * is expanded in row blocks (diagonal written as 0). */
int sa_host_write_hdf5(const char *path, const struct sa_host_store *s, const int32_t *matrix, bool triangular,
		       unsigned compression);
size_t sa_host_hdf5_chunk_dim(size_t dim); /* exposed for tests */
/* ... and the same file from tiles that arrive finished -- zlib streams from the device-side encoder of the -z option, or
 * (compression 0) the raw tiles; include/seqalign_hip.h: sa_zjob_next has exactly this signature:
 * next(user, rows, cols, streams, sizes) fills the next batch of at most ceil(N / chunk) tiles (tile coordinates + bytes,
 * valid until its next call) and returns their number, 0 at the end, < 0 on error; the tiles go to H5Dwrite_chunk unchanged,
 * in whatever order they come. */
typedef int (*sa_host_tiles_fn)(void *user, uint32_t *rows, uint32_t *cols, const uint8_t **streams, size_t *sizes);
int sa_host_write_hdf5_streams(const char *path, const struct sa_host_store *s, unsigned compression, sa_host_tiles_fn next,
			       void *user);

#ifdef __cplusplus
}
#endif
#endif
