/*
 * oracle/sa_oracle.c -- TEST INFRASTRUCTURE ONLY (see sa_oracle.h).
 *
 * CPU restatement of the reference hot path.  Each function cites the
 * reference lines whose behaviour it restates.  The storage differs on
 * purpose (rolling rows instead of the reference's full (len+1)^2 tables --
 * bit-exactness is in the arithmetic, not the storage).
 */
#include "sa_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* s32 add with defined two's-complement wrap (see header). */
static inline int32_t wadd(int32_t a, int32_t b)
{
	return (int32_t)((uint32_t)a + (uint32_t)b);
}
static inline int32_t wmul(int32_t a, int32_t b)
{
	return (int32_t)((uint32_t)a * (uint32_t)b);
}
static inline int32_t max32(int32_t a, int32_t b)
{
	return a > b ? a : b;
}

/* Needleman-Wunsch, linear gap: bio/method/nw.c:5-42.
 * Borders nw.c:14-20 (H[0][c]=c*g, H[r][0]=r*g), cell nw.c:28-38,
 * lookup SUB_MAT[row_res][col_res] nw.c:23,29, result nw.c:41. */
static int32_t o_nw(const sa_o_params *p, const int32_t *ind, int32_t len1,
		    const uint8_t *seq2, int32_t len2, int32_t *row)
{
	const int32_t g = p->gap_pen;
	for (int32_t c = 0; c <= len1; c++)
		row[c] = wmul(c, g);
	for (int32_t r = 1; r <= len2; r++) {
		const int32_t *sub = p->sub + 24 * p->lut[seq2[r - 1]];
		int32_t diag = row[0];
		int32_t left = wmul(r, g);
		row[0] = left;
		for (int32_t c = 1; c <= len1; c++) {
			int32_t up = row[c];
			int32_t v = wadd(diag, sub[ind[c - 1]]);
			v = max32(wadd(up, g), v);
			v = max32(wadd(left, g), v);
			diag = up;
			row[c] = v;
			left = v;
		}
	}
	return row[len1];
}

/* Gotoh global affine: bio/method/ga.c:10-68.
 * Corner ga.c:23-24, top border ga.c:26-30, left border ga.c:32-38,
 * cell ga.c:45-64 (lookup SUB_MAT[col_res][row_res] ga.c:46), result ga.c:67.
 * SW (local affine): bio/method/sw.c:5-62; borders sw.c:18-30, cell sw.c:38-58
 * with the zero floor sw.c:53 and running maximum sw.c:57, result sw.c:61. */
static int32_t o_affine(const sa_o_params *p, int local, const int32_t *ind,
			int32_t len1, const uint8_t *seq2, int32_t len2,
			int32_t *m, int32_t *y)
{
	const int32_t o = p->gap_opn, e = p->gap_ext;
	/* row 0 */
	m[0] = 0;
	y[0] = SA_O_SCORE_MIN;
	if (local) {
		for (int32_t c = 1; c <= len1; c++) {
			m[c] = 0;
			y[c] = SA_O_SCORE_MIN;
		}
	} else {
		int32_t x = SA_O_SCORE_MIN; /* gap_x[0] */
		for (int32_t c = 1; c <= len1; c++) {
			x = max32(wadd(m[c - 1], o), wadd(x, e));
			m[c] = x;
			y[c] = SA_O_SCORE_MIN;
		}
	}
	int32_t score = 0;
	/* column-0 state of the previous row */
	int32_t m0_prev = m[0];
	int32_t y0_prev = y[0];
	for (int32_t r = 1; r <= len2; r++) {
		const int32_t c2 = p->lut[seq2[r - 1]];
		int32_t m0, y0, x_left;
		if (local) {
			m0 = 0;
			y0 = SA_O_SCORE_MIN;
		} else {
			y0 = max32(wadd(m0_prev, o), wadd(y0_prev, e));
			m0 = y0;
		}
		x_left = SA_O_SCORE_MIN;      /* gap_x[row][0] */
		int32_t m_left = m0;          /* match[row][c-1] */
		int32_t m_diag = m0_prev;     /* match[row-1][c-1] */
		for (int32_t c = 1; c <= len1; c++) {
			int32_t sim = p->sub[24 * ind[c - 1] + c2];
			int32_t m_up = m[c], y_up = y[c];
			int32_t sd = wadd(m_diag, sim);
			int32_t gx = max32(wadd(m_left, o), wadd(x_left, e));
			int32_t gy = max32(wadd(m_up, o), wadd(y_up, e));
			int32_t best = local ? max32(sd, 0) : sd;
			best = max32(gx, best);
			best = max32(gy, best);
			if (local)
				score = max32(score, best);
			m_diag = m_up;
			m[c] = best;
			y[c] = gy;
			m_left = best;
			x_left = gx;
		}
		m0_prev = m0;
		y0_prev = y0;
		m[0] = m0;
		y[0] = y0;
	}
	return local ? score : m[len1];
}

static int32_t pair_idx(const sa_o_params *p, const int32_t *ind, int32_t len1,
			const uint8_t *seq2, int32_t len2, int32_t *buf)
{
	switch (p->method) {
	case SA_O_NW:
		return o_nw(p, ind, len1, seq2, len2, buf);
	case SA_O_GA:
		return o_affine(p, 0, ind, len1, seq2, len2, buf, buf + len1 + 1);
	default:
		return o_affine(p, 1, ind, len1, seq2, len2, buf, buf + len1 + 1);
	}
}

int32_t sa_oracle_pair(const sa_o_params *p, const uint8_t *seq1, int32_t len1,
		       const uint8_t *seq2, int32_t len2)
{
	int32_t *ind = malloc(sizeof(*ind) * (size_t)(len1 + 1));
	int32_t *buf = malloc(sizeof(*buf) * 2 * (size_t)(len1 + 1));
	for (int32_t k = 0; k < len1; k++) /* align.c:49-50 */
		ind[k] = p->lut[seq1[k]];
	int32_t r = pair_idx(p, ind, len1, seq2, len2, buf);
	free(buf);
	free(ind);
	return r;
}

static int32_t max_len(const sa_o_meta *meta, int32_t num)
{
	int32_t mx = 0;
	for (int32_t k = 0; k < num; k++)
		mx = meta[k].len > mx ? meta[k].len : mx;
	return mx;
}

int sa_oracle_max_threads(void)
{
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}

static int pick_threads(int threads)
{
	int mx = sa_oracle_max_threads();
	return (threads <= 0 || threads > mx) ? mx : threads;
}

/* bio/align.c:21-72: parallel over column j (dynamic schedule), seq j indexed
 * once (align.c:49-50), every i<j aligned (align.c:51-56), then the finished
 * column is scattered by output_fill (io/output.c:68-84). */
int sa_oracle_align(const sa_o_params *p, const uint8_t *seqs,
		    const sa_o_meta *meta, int32_t num, int32_t *matrix,
		    int triangular, int threads)
{
	const int32_t mx = max_len(meta, num);
	int fail = 0;
#pragma omp parallel num_threads(pick_threads(threads))
	{
		int32_t *ind = malloc(sizeof(*ind) * (size_t)(mx + 1));
		int32_t *buf = malloc(sizeof(*buf) * 2 * (size_t)(mx + 1));
		int32_t *cols = malloc(sizeof(*cols) * (size_t)num);
		if (!ind || !buf || !cols) {
#pragma omp atomic write
			fail = 1;
		}
#pragma omp barrier
		if (!fail) {
#pragma omp for schedule(dynamic)
			for (int32_t j = 1; j < num; j++) {
				const uint8_t *s1 = seqs + meta[j].off;
				const int32_t l1 = meta[j].len;
				for (int32_t k = 0; k < l1; k++)
					ind[k] = p->lut[s1[k]];
				for (int32_t i = 0; i < j; i++)
					cols[i] = pair_idx(p, ind, l1,
							   seqs + meta[i].off,
							   meta[i].len, buf);
				if (!matrix)
					continue;
				if (triangular) { /* output.c:83 */
					size_t base = (size_t)j * (size_t)(j - 1) / 2;
					memcpy(matrix + base, cols,
					       sizeof(*cols) * (size_t)j);
				} else { /* output.c:76-81 */
					size_t dim = (size_t)num;
					for (size_t row = 0; row < (size_t)j; row++) {
						matrix[dim * row + (size_t)j] = cols[row];
						matrix[dim * (size_t)j + row] = cols[row];
					}
				}
			}
		}
		free(cols);
		free(buf);
		free(ind);
	}
	return fail;
}

/* Largest j with j(j-1)/2 <= p  (bio/kernels.cu:17-30), then i = p - j(j-1)/2. */
void sa_oracle_unpack_index(int64_t p, int32_t *i, int32_t *j)
{
	int64_t lo = 1, hi = (int64_t)1 << 31;
	while (lo < hi) {
		int64_t mid = lo + (hi - lo) / 2;
		if (mid * (mid - 1) / 2 <= p)
			lo = mid + 1;
		else
			hi = mid;
	}
	int64_t jj = lo - 1;
	*j = (int32_t)jj;
	*i = (int32_t)(p - jj * (jj - 1) / 2);
}

int sa_oracle_align_pairs(const sa_o_params *p, const uint8_t *seqs,
			  const sa_o_meta *meta, int32_t num,
			  const int64_t *pidx, int64_t count, int32_t *out,
			  int threads)
{
	const int32_t mx = max_len(meta, num);
	const int64_t total = (int64_t)num * (num - 1) / 2;
	int fail = 0;
#pragma omp parallel num_threads(pick_threads(threads))
	{
		int32_t *ind = malloc(sizeof(*ind) * (size_t)(mx + 1));
		int32_t *buf = malloc(sizeof(*buf) * 2 * (size_t)(mx + 1));
		int32_t cached_j = -1;
#pragma omp for schedule(dynamic, 64)
		for (int64_t k = 0; k < count; k++) {
			int64_t pi = pidx ? pidx[k] : k;
			if (pi < 0 || pi >= total) {
#pragma omp atomic write
				fail = 1;
				continue;
			}
			int32_t i, j;
			sa_oracle_unpack_index(pi, &i, &j);
			if (j != cached_j) {
				const uint8_t *s1 = seqs + meta[j].off;
				for (int32_t t = 0; t < meta[j].len; t++)
					ind[t] = p->lut[s1[t]];
				cached_j = j;
			}
			out[k] = pair_idx(p, ind, meta[j].len, seqs + meta[i].off,
					  meta[i].len, buf);
		}
		free(buf);
		free(ind);
	}
	return fail;
}

int sa_oracle_align_range(const sa_o_params *p, const uint8_t *seqs,
			  const sa_o_meta *meta, int32_t num, int64_t start,
			  int64_t count, int32_t *out, int threads)
{
	const int64_t total = (int64_t)num * (num - 1) / 2;
	if (start < 0 || count < 0 || start + count > total)
		return 1;
	int64_t *idx = malloc(sizeof(*idx) * (size_t)(count ? count : 1));
	if (!idx)
		return 1;
	for (int64_t k = 0; k < count; k++)
		idx[k] = start + k;
	int r = sa_oracle_align_pairs(p, seqs, meta, num, idx, count, out, threads);
	free(idx);
	return r;
}

/* bio/filter.c:38-55, sequential order. */
int32_t sa_oracle_filter(const uint8_t *seqs, const sa_o_meta *meta,
			 int32_t num, float threshold, uint8_t *keep)
{
	int32_t kept = 0;
	for (int32_t j = 0; j < num; j++)
		keep[j] = 1;
	if (threshold <= 0.0f) /* filter.c:16-17 */
		return num;
	for (int32_t j = 1; j < num; j++) {
		const uint8_t *s1 = seqs + meta[j].off;
		for (int32_t i = 0; i < j; i++) {
			if (!keep[i])
				continue;
			const uint8_t *s2 = seqs + meta[i].off;
			int32_t ml = meta[j].len < meta[i].len ? meta[j].len : meta[i].len;
			int32_t matches = 0;
			for (int32_t k = 0; k < ml; k++)
				matches += s1[k] == s2[k];
			if ((float)matches / (float)ml >= threshold) {
				keep[j] = 0;
				break;
			}
		}
	}
	for (int32_t j = 0; j < num; j++)
		kept += keep[j];
	return kept;
}
