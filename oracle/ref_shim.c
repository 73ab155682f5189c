/*
 * oracle/ref_shim.c -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
 *
 * Thin driver that is compiled TOGETHER WITH the reference's own, unmodified
 * sources where they lie under /root/reference (see oracle/Makefile, target
 * `_ref`), producing oracle/_ref/libseqalign_ref.so.  It contains no alignment
 * arithmetic of its own: every score it returns is produced by the reference's
 * functions
 *     ALIGN->method  = align_nw / align_ga / align_sw   (src/bio/method/{nw,ga,sw}.c)
 *     align()                                            (src/bio/align.c:21-72)
 *     output_fill()                                      (src/io/output.c:68-84)
 *     filter()                                           (src/bio/filter.c:14-89)
 * configured by the reference's own option parser (args_parse/args_validate,
 * third_party/clix/args.h:1798,1839) fed with a synthetic argv, so that gap
 * negation (src/bio/align.c:127), matrix selection (src/bio/matrices.c:44-58)
 * and the ga->nw swap (src/bio/method/ga.c:70-88) behave exactly as in the CLI.
 *
 * The library is used (a) by tools/make_golden.py to generate tests/golden/,
 * (b) by tests to pin oracle/sa_oracle.c against the real reference when
 * /root/reference is present, and (c) optionally as bench.py's cpu_baseline.
 */
#include <string.h>
#include <stdlib.h>
#include <stdbool.h>

#include <args.h>

#include "bio/align.h"
#include "io/input.h"
#include "io/output.h"
#include "system/os.h"

extern size_t TABLE_SIZE;
bool align(struct input, struct output);
bool filter(struct input *);

/* Runs the reference option parser.  One configuration per loaded library
 * instance (clix keeps "already set" state), callers load a fresh copy. */
int ref_configure(int argc, char **argv)
{
	if (!args_parse(argc, argv))
		return 1;
	if (!args_validate())
		return 2;
	return 0;
}

/* Current state of the reference's globals after configuration. */
void ref_get_params(s32 *gap_pen, s32 *gap_opn, s32 *gap_ext, s32 *affine,
		    s32 lut[SEQ_LUT_SIZE], s32 sub[SUB_MAT_DIM * SUB_MAT_DIM],
		    char *method_name, int method_name_cap)
{
	*gap_pen = GAP_PEN;
	*gap_opn = GAP_OPN;
	*gap_ext = GAP_EXT;
	*affine = ALIGN->gap == GAP_AFFINE;
	memcpy(lut, SEQ_LUT, sizeof(SEQ_LUT));
	memcpy(sub, SUB_MAT, sizeof(SUB_MAT));
	strncpy(method_name, ALIGN->aliases[1], (size_t)method_name_cap - 1);
	method_name[method_name_cap - 1] = 0;
}

/* One pair through the reference's per-pair kernel (bio/align.h:29-30):
 * seq1 is pre-indexed through SEQ_LUT like align.c:49-50 does. */
s32 ref_pair(const uchar *seq1, s32 len1, const uchar *seq2, s32 len2)
{
	s32 mx = len1 > len2 ? len1 : len2;
	TABLE_SIZE = (size_t)(mx + 1) * (size_t)(mx + 1);
	size_t mult = ALIGN->gap == GAP_AFFINE ? 3 : 1;
	s32 *table = malloc(sizeof(*table) * TABLE_SIZE * mult);
	s32 *ind = malloc(sizeof(*ind) * (size_t)len1);
	for (s32 k = 0; k < len1; k++)
		ind[k] = SEQ_LUT[seq1[k]];
	s32 r = ALIGN->method(len1, len2, seq2, ind, table);
	free(ind);
	free(table);
	return r;
}

/* Whole all-vs-all run through the reference driver align() + output_fill().
 * `matrix` is caller-owned and zero-initialised (mmap zero-fill in the
 * reference, io/output.c:55): N*N s32 (full) or N(N-1)/2 s32 (triangular). */
int ref_align(uchar *seqs, struct meta *meta, s32 num, s32 max, s32 *matrix,
	      int triangular)
{
	struct input in = { .seqs = seqs, .meta = meta, .max = max, .num = num };
	struct output out = { .matrix = matrix,
			      .seqs = NULL,
			      .dim = (size_t)num,
			      .triangular = triangular != 0 };
	return align(in, out) ? 0 : 1;
}

/* Reference HDF5 writer (output_flush -> flush_hdf5, src/io/output.c:89-99, src/io/format/hdf5.c:14-202)
 * into the -o path given at configuration time; compression level from -z. */
int ref_flush(uchar *seqs, struct meta *meta, s32 num, s32 *matrix, int triangular)
{
	const char **ptrs = malloc(sizeof(*ptrs) * (size_t)num);
	for (s32 i = 0; i < num; i++)
		ptrs[i] = (const char *)(seqs + meta[i].off);
	struct output out = { .matrix = matrix, .seqs = ptrs, .dim = (size_t)num, .triangular = triangular != 0 };
	bool ok = output_flush(&out);
	free(ptrs);
	return ok ? 0 : 1;
}

/* Reference similarity filter (-f), threshold comes from the parsed argv.
 * Compacts seqs/meta in place, returns the surviving count (or -1). */
int ref_filter(uchar *seqs, struct meta *meta, s32 *num, s32 *max)
{
	struct input in = { .seqs = seqs, .meta = meta, .max = *max, .num = *num };
	if (!filter(&in))
		return -1;
	*num = in.num;
	*max = in.max;
	return in.num;
}

/* A stripe of whole columns [j_lo, j_hi) of the all-vs-all run, each column through the reference's per-pair kernel
 * exactly as its driver calls it (src/bio/align.c:44-58: column sequence j indexed once through SEQ_LUT, then
 * ALIGN->method(len_j, len_i, seq_i, ind_j, table) for every i < j).  out receives column j_lo's j_lo scores, then
 * column j_lo+1's, ...: the packed layout of src/io/output.c:83 restricted to the stripe.  Used by
 * tools/make_digests.py for the full-size cfg 4 / cfg 5 stripes, where running align() over the whole matrix would
 * take hours of CPU. */
int ref_columns(const uchar *seqs, const struct meta *meta, s32 num, s32 max, s32 j_lo, s32 j_hi, s32 *out)
{
	if (j_lo < 1 || j_hi > num || j_lo > j_hi)
		return 1;
	TABLE_SIZE = (size_t)(max + 1) * (size_t)(max + 1);
	const size_t mult = ALIGN->gap == GAP_AFFINE ? 3 : 1;
	int failed = 0;
#pragma omp parallel
	{
		s32 *table = malloc(sizeof(*table) * TABLE_SIZE * mult);
		s32 *ind = malloc(sizeof(*ind) * (size_t)max);
		if (!table || !ind)
			failed = 1;
#pragma omp barrier
		if (!failed) {
#pragma omp for schedule(dynamic) collapse(1)
			for (s32 j = j_lo; j < j_hi; j++) {
				const struct meta m1 = meta[j];
				for (s32 k = 0; k < m1.len; k++)
					ind[k] = SEQ_LUT[seqs[m1.off + k]];
				s32 *col = out + ((size_t)j * (size_t)(j - 1) / 2 - (size_t)j_lo * (size_t)(j_lo - 1) / 2);
				for (s32 i = 0; i < j; i++)
					col[i] = ALIGN->method(m1.len, meta[i].len, seqs + meta[i].off, ind, table);
			}
		}
		free(ind);
		free(table);
	}
	return failed;
}
