/*
 * sa_filter.hip -- device part of the similarity filter (`-f`, reference src/bio/filter.c:14-89).
 *
 * The reference drops sequence j when some KEPT i<j has  matches(first min(len) positions)/min(len) >= thr
 * (float division, >=; filter.c:47-54).  Whether a pair is "similar" does not depend on the order, only
 * the keep/drop decision does.  So the O(N^2 L) part -- the boolean relation R[i][j] -- is computed here
 * as a bit matrix, and the host resolves the greedy keep/drop sequentially over j with word-wide ANDs
 * (sa_hip_filter in sa_abi.hip): bit-identical to the reference run with one thread.
 *
 * Mapping: a workgroup takes a 64(j) x 64(i) tile of pairs, stages the 128 sequences in LDS in pieces of
 * PIECE positions, every wave handles 16 rows j, lane = i.  matches accumulate in a VGPR per (lane, row);
 * the 64 results of a row are packed with one ballot = one 64-bit word of R, row-major per j.
 */
#include "sa_internal.h"

namespace {

constexpr int TILE = 64;
constexpr int PIECE = 32;           /* positions staged per pass (a multiple of 4); after every pass the tile asks whether any of its
                                     * pairs can still reach the threshold, see below */
constexpr int ROWB = PIECE + 4;     /* LDS row stride: odd number of dwords -> conflict-free lane-per-row reads */

/* 64-bit words of the relation before row j: sum_{j'<j} ceil(j'/64) */
__host__ __device__ inline long long row_offset(long long j)
{
	if (j <= 0)
		return 0;
	const long long jm = j - 1, blocks = jm / 64;
	return 64 * blocks * (blocks + 1) / 2 + (jm - 64 * blocks) * (blocks + 1);
}

__global__ __launch_bounds__(256) void sa_k_filter_relation(const uint8_t *__restrict__ codes, const int32_t *__restrict__ off,
							     int32_t num, float threshold, unsigned long long *__restrict__ rel,
							     int32_t jt0, long long band_base)
{
	__shared__ __attribute__((aligned(16))) uint8_t s_i[TILE * ROWB];
	__shared__ __attribute__((aligned(16))) uint8_t s_j[TILE * ROWB];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	/* tiles on or below the block diagonal: (jt, it) with it <= jt, enumerated row by row */
	int32_t jt = jt0 + blockIdx.y, it = blockIdx.x;
	if (it > jt)
		return;
	const int32_t j0 = jt * TILE, i0 = it * TILE;
	const int32_t my_i = i0 + lane;
	const int32_t len_i = my_i < num ? off[my_i + 1] - off[my_i] - 1 : 0;
	int32_t maxlen = 0;
	for (int r = 0; r < 16; r++) {
		const int32_t j = j0 + wv * 16 + r;
		const int32_t lj = j < num ? off[j + 1] - off[j] - 1 : 0;
		maxlen = lj > maxlen ? lj : maxlen;
	}
	/* block-wide longest min(len) bound: longest j row of the tile (cheap upper bound) */
	__shared__ int32_t s_max[4];
	if (lane == 0)
		s_max[wv] = maxlen;
	__syncthreads();
	maxlen = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));

	int matches[16];
#pragma unroll
	for (int r = 0; r < 16; r++)
		matches[r] = 0;

	for (int32_t p0 = 0; p0 < maxlen; p0 += PIECE) {
		__syncthreads();
		/* stage PIECE positions of the 64 i-rows and 64 j-rows; out-of-sequence bytes are two different
		 * fillers so they never match each other */
		for (int k = threadIdx.x; k < TILE * PIECE; k += 256) {
			const int row = k / PIECE, pos = k - row * PIECE;
			const int32_t si = i0 + row, sj = j0 + row;
			const int32_t li = si < num ? off[si + 1] - off[si] - 1 : 0;
			const int32_t lj = sj < num ? off[sj + 1] - off[sj] - 1 : 0;
			s_i[row * ROWB + pos] = p0 + pos < li ? codes[off[si] + p0 + pos] : (uint8_t)0xFE;
			s_j[row * ROWB + pos] = p0 + pos < lj ? codes[off[sj] + p0 + pos] : (uint8_t)0xFD;
		}
		__syncthreads();
		const uint32_t *mine = reinterpret_cast<const uint32_t *>(s_i + lane * ROWB);
#pragma unroll 1
		for (int w = 0; w < PIECE / 4; w++) {
			const uint32_t a = mine[w];
#pragma unroll
			for (int r = 0; r < 16; r++) {
				const uint32_t b = *reinterpret_cast<const uint32_t *>(s_j + (wv * 16 + r) * ROWB + 4 * w);
				/* equal bytes of two words: a byte of x = a ^ b is zero iff the top bit of that byte stays clear in
				 * ((x & 0x7f..) + 0x7f..) | x -- six instructions per four residues (xor, and, add, or3, not, bcnt with
				 * accumulate) instead of a mask, a compare and an add per byte (round 4: 126 -> see DESIGN 4.7) */
				const uint32_t x = a ^ b;
				const uint32_t t = ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu;
				matches[r] += __builtin_popcount(~t);
			}
		}
		/* Early exit (round 4).  A pair that has m matches after `done` positions ends with at most m + (ml - done): when not
		 * even that reaches the threshold for ANY pair of the tile, the remaining passes cannot set a bit -- and unrelated
		 * sequences are out after the first 32 positions (a twentieth of the positions match by chance; cfg 5 needs nine
		 * tenths).  The bound is one match on the generous side of the final float test, so no similar pair is cut short. */
		const int32_t done = p0 + PIECE;
		int alive = 0;
#pragma unroll
		for (int r = 0; r < 16; r++) {
			const int32_t j = j0 + wv * 16 + r;
			const int32_t lj = j < num ? off[j + 1] - off[j] - 1 : 0;
			const int32_t ml = len_i < lj ? len_i : lj;
			const int32_t rem = ml > done ? ml - done : 0;
			alive |= my_i < j && j < num && ml > 0 && (float)(matches[r] + rem + 1) / (float)ml >= threshold;
		}
		if (!__syncthreads_or(alive))
			break;
	}
	/* fillers never match, so `matches` counts exactly the equal residues inside min(len_i, len_j) */
#pragma unroll
	for (int r = 0; r < 16; r++) {
		const int32_t j = j0 + wv * 16 + r;
		const int32_t lj = j < num ? off[j + 1] - off[j] - 1 : 0;
		const int32_t ml = len_i < lj ? len_i : lj;
		bool similar = false;
		if (my_i < j && j < num && ml > 0)
			similar = (float)matches[r] / (float)ml >= threshold; /* filter.c:51, same float expression */
		const unsigned long long word = __ballot(similar);
		/* row j of R has ceil(j/64) words; the rows of a band are stored back to back */
		if (lane == 0 && j < num && it < (j + 63) / 64)
			rel[row_offset(j) - band_base + it] = word;
	}
}

} // namespace

long long sa_filter_row_offset(long long j) { return row_offset(j); }

/* relation rows [64*jt0, 64*(jt0+tile_rows)) into rel[0 ...), row-major per j */
hipError_t sa_launch_filter_relation(const uint8_t *codes, const int32_t *off, int32_t num, float threshold,
				      unsigned long long *rel, int32_t jt0, int32_t tile_rows, hipStream_t s)
{
	const int tiles_x = jt0 + tile_rows; /* i-tiles that can be <= the last j-tile of the band */
	hipLaunchKernelGGL(sa_k_filter_relation, dim3((unsigned)tiles_x, (unsigned)tile_rows), dim3(256), 0, s, codes, off, num,
			   threshold, rel, jt0, row_offset((long long)jt0 * TILE));
	return hipGetLastError();
}
