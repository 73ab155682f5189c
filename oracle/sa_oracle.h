/*
 * oracle/sa_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's all-vs-all alignment hot path
 * (jakovdev/SequenceAligner, src/bio).  It is the checker for the HIP path:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  The product library (libseqalign_hip.so) never links, loads or
 * calls anything in oracle/.
 *
 * Parity status: PINNED.  The restatement is checked against
 *   (1) tests/golden/ vectors generated here from the reference's own sources
 *       compiled unmodified (oracle/_ref, see oracle/Makefile + tools/make_golden.py),
 *   (2) the known-answer table recorded in SURVEY.md §8(c),
 *   (3) live, against oracle/_ref/libseqalign_ref.so whenever /root/reference
 *       is present (tests/test_oracle_vs_ref.py).
 *
 * Arithmetic: 32-bit signed with two's-complement wrap-around (the reference
 * computes in s32; inside the reference's own validity range -- length limit
 * io/input.c:15-19 -- nothing wraps and results are identical; outside it the
 * reference is undefined behaviour and this oracle defines the wrap so that
 * the GPU path has something exact to be compared with).
 */
#ifndef SA_ORACLE_H
#define SA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bio/align.h:6-9 */
typedef struct sa_o_meta {
	int32_t off;
	int32_t len;
} sa_o_meta;

enum { SA_O_NW = 0, SA_O_GA = 1, SA_O_SW = 2 };

/* bio/align.h:19 */
#define SA_O_SCORE_MIN (INT32_MIN / 2)

/* Scoring parameters exactly as the reference keeps them in globals
 * (bio/align.h:11-19): gaps are stored NEGATED (bio/align.c:127). */
typedef struct sa_o_params {
	int32_t method;        /* SA_O_NW / SA_O_GA / SA_O_SW */
	int32_t gap_pen;       /* GAP_PEN  (<= 0), used by NW */
	int32_t gap_opn;       /* GAP_OPN  (<= 0), used by GA, SW */
	int32_t gap_ext;       /* GAP_EXT  (<= 0), used by GA, SW */
	int32_t lut[128];      /* SEQ_LUT  ASCII -> 0..23, -1 invalid */
	int32_t sub[24 * 24];  /* SUB_MAT  row-major 24x24 */
} sa_o_params;

/* One pair, same argument meaning as `ALIGN->method` (bio/align.h:29-30):
 * seq1/len1 = the column sequence (reference pre-indexes it into ind[],
 * align.c:49-50), seq2/len2 = the row sequence. */
int32_t sa_oracle_pair(const sa_o_params *p, const uint8_t *seq1, int32_t len1,
		       const uint8_t *seq2, int32_t len2);

/* align() + output_fill() (bio/align.c:21-72, io/output.c:68-84):
 * matrix must be zero-initialised by the caller; full N*N symmetric with an
 * untouched (zero) diagonal, or packed triangular with pair (i<j) at
 * j(j-1)/2+i.  threads<=0 -> all.  Returns 0 on success. */
int sa_oracle_align(const sa_o_params *p, const uint8_t *seqs,
		    const sa_o_meta *meta, int32_t num, int32_t *matrix,
		    int triangular, int threads);

/* Packed-index sub-range [start, start+count) of the triangular result,
 * out[k] = score of pair #start+k (the reference's device batch abstraction,
 * bio/kernels.cu:32-40).  Used for sampled parity at full benchmark sizes. */
int sa_oracle_align_range(const sa_o_params *p, const uint8_t *seqs,
			  const sa_o_meta *meta, int32_t num, int64_t start,
			  int64_t count, int32_t *out, int threads);

/* Scores for an explicit list of packed pair indices (random sampling). */
int sa_oracle_align_pairs(const sa_o_params *p, const uint8_t *seqs,
			  const sa_o_meta *meta, int32_t num,
			  const int64_t *pair_idx, int64_t count, int32_t *out,
			  int threads);

/* packed index -> (i,j), i<j  (bio/kernels.cu:17-30,38-40) */
void sa_oracle_unpack_index(int64_t p, int32_t *i, int32_t *j);

/* bio/filter.c:14-89 with SEQUENTIAL semantics (-T 1): for j ascending, drop j
 * if some kept i<j has matches(first min(len))/min(len) >= thr (float).
 * keep[k] (size num) receives 1/0.  Returns the number kept. */
int32_t sa_oracle_filter(const uint8_t *seqs, const sa_o_meta *meta,
			 int32_t num, float threshold, uint8_t *keep);

int sa_oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif /* SA_ORACLE_H */
