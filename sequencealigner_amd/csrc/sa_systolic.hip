/*
 * sa_systolic.hip -- streaming systolic wave kernels (gfx950, wave64): the fast path.
 *
 * Mapping.  A wavefront is cut into NG = 64/G lane groups of G lanes (G = 16: one DPP row,
 * 32 or 64).  All groups of a wave hold the SAME column sequence j, right-aligned over the
 * group's W = G*K column slots (K consecutive DP columns per lane, held in VGPRs).  The ROWS
 * are not one sequence but a STREAM: the encoded sequence store itself, i.e. the residues of
 * sequences i0, i0+1, ... back to back, each followed by its terminator.  One residue enters
 * lane 0 of the group per step and travels one lane per step (DPP row_shr / wave_shr), so at
 * step t lane l computes row t-l of the anti-diagonal sweep for its K columns; the left / diagonal
 * dependency crosses lanes with one or two more DPP moves.  Because the stream never drains
 * between sequences, the systolic ramp (G-1 steps) is paid once per 64 sequences instead of once
 * per pair.
 *
 * Sequence boundaries cost no per-cell work.  All recurrences are kept relative to a baseline B
 * that is raised by DELTA (> the largest possible score growth inside one sequence) whenever a
 * terminator passes: the terminator row injects the next baseline at lane 0 and every stale value
 * of the previous sequence loses every later max() against it, which is exactly the reference's
 * border initialisation (nw.c:14-20, ga.c:23-38, sw.c:18-30).  The score of sequence e leaves the
 * pipeline at lane G-1 when its terminator arrives there and is un-biased in the epilogue.
 *
 * Recurrences (bit-exact integer re-associations of the reference's; S = substitution score):
 *   NW  nw.c:29-35   H' = H - (r+c)g        H'  = max3(H'diag + (S-2g), H'up, H'left)               2 ops/cell
 *   GA  ga.c:46-63   N  = M - (r+c)e + q    Y' = max(Nup, Y'up); M' = max3(Ndiag + (S-e-o), X', Y')
 *                    q = o - e <= 0         N = M' + q; X'next = max(N, X')                          5 ops/cell
 *   SW  sw.c:39-57   No = M + o             Y = max(No_up, Yup+e); M = max(max3(No_diag + (S-o), X, Y), B)
 *                                           No = M + o; Xnext = max(No, X+e); best = max(best, M)    9 ops/cell
 * The per-residue scores (S + constant) of the lane's K columns are a query profile of s8 values in
 * LDS: row = residue code, one private 4/8/16-byte slot per lane of a 32-lane half, so the
 * ds_read is conflict-free by construction; bytes are consumed by SDWA adds (no unpack op).
 *
 * Validity (checked on the host, sa_driver.hip: systolic_ok): profile values fit s8, q <= 0 for GA,
 * 65*DELTA < 2^30, column length <= largest W.  Everything else runs on sa_generic.hip.
 */
#include "sa_internal.h"

namespace {

constexpr int CH = SA_SYS_CHUNK; /* sequences per group stream */
constexpr int32_t NEG = INT32_MIN / 2;

__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imax3(int a, int b, int c) { return imax(imax(a, b), c); }

__device__ __forceinline__ int dpp_row_shr1(int old, int src)
{
	return __builtin_amdgcn_update_dpp(old, src, 0x111, 0xf, 0xf, false);
}
__device__ __forceinline__ int dpp_wave_shr1(int old, int src)
{
	return __builtin_amdgcn_update_dpp(old, src, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ int dpp_row_shl1(int v)
{
	return __builtin_amdgcn_update_dpp(v, v, 0x101, 0xf, 0xf, false);
}

/* value of lane-1 inside the group; the group's lane 0 receives `lead` instead */
template <int G> __device__ __forceinline__ int shift_in(int lead, int src, bool leader)
{
	if (G == 16)
		return dpp_row_shr1(lead, src);
	if (G == 64)
		return dpp_wave_shr1(lead, src);
	const int v = dpp_wave_shr1(lead, src);
	return leader ? lead : v;
}

template <int RB> struct Slot;
template <> struct Slot<4> { using type = uint32_t; };
template <> struct Slot<8> { using type = uint2; };
template <> struct Slot<16> { using type = uint4; };

__device__ __forceinline__ int slot_byte(const uint32_t w, int q) { return (int)(int8_t)(w >> (8 * q)); }
__device__ __forceinline__ int slot_byte(const uint2 &w, int q)
{
	return (int)(int8_t)((q < 4 ? w.x : w.y) >> (8 * (q & 3)));
}
__device__ __forceinline__ int slot_byte(const uint4 &w, int q)
{
	const uint32_t v = q < 4 ? w.x : q < 8 ? w.y : q < 12 ? w.z : w.w;
	return (int)(int8_t)(v >> (8 * (q & 3)));
}

template <int METHOD, int G, int K>
__global__ __launch_bounds__(64) void sa_k_systolic(SaSysArgs A)
{
	constexpr int NG = 64 / G;
	constexpr int W = G * K;
	constexpr int RB = K <= 4 ? 4 : K <= 8 ? 8 : 16;  /* bytes per profile slot                    */
	constexpr int NT = G == 64 ? 2 : 1;               /* one table per 32 distinct column holders  */
	constexpr int ROWSTRIDE = 32 * RB;
	constexpr int SH = RB == 4 ? 7 : RB == 8 ? 8 : 9;
	using slot_t = typename Slot<RB>::type;

	constexpr int TBLSTRIDE = 32 * ROWSTRIDE;        /* power of two: address = tok<<SH | slot_off */
	__shared__ __attribute__((aligned(16))) uint8_t s_prof[NT == 2 ? 2 * TBLSTRIDE : SA_CODE_ROWS * ROWSTRIDE];
	__shared__ int32_t s_out[NG * CH];
	__shared__ int8_t s_sub[SA_SUB_DIM * SA_SUB_DIM];

	const int lane = threadIdx.x;
	const int lig = lane & (G - 1);
	const int grp = lane / G;
	const bool leader = lig == 0;

	/* ---- which wave-tile: (column j, block of NG*CH rows) ---- */
	const int32_t t = blockIdx.x;
	int32_t lo = 0, hi = A.ncols; /* largest k with tprefix[k] <= t */
	while (hi - lo > 1) {
		const int32_t mid = (lo + hi) >> 1;
		if (A.tprefix[mid] <= t)
			lo = mid;
		else
			hi = mid;
	}
	const int32_t j = A.jlist[lo];
	const int32_t chunk = t - A.tprefix[lo];
	const int64_t tri = (int64_t)j * (j - 1) / 2;
	const int64_t ia64 = A.start > tri ? A.start - tri : 0;
	const int64_t ib64 = A.end - tri < j ? A.end - tri : j;
	const int32_t i_begin = (int32_t)ia64 + chunk * (NG * CH);
	const int32_t i_count = (int32_t)ib64 - i_begin < NG * CH ? (int32_t)ib64 - i_begin : NG * CH;
	const int32_t offj = A.off[j];
	const int32_t n = A.off[j + 1] - offj - 1;

	for (int k = lane; k < SA_SUB_DIM * SA_SUB_DIM; k += 64)
		s_sub[k] = A.sub8[k];
	__syncthreads();

	/* ---- query profile of column sequence j for this lane's K column slots ---- */
	{
		int bq[K];
#pragma unroll
		for (int q = 0; q < K; q++) {
			const int c0 = lig * K + q - (W - n);
			bq[q] = c0 >= 0 ? (int)A.codes[offj + c0] : -1;
		}
		const int slot = lane & 31;
		const int a_lo = NT == 2 ? 0 : (lane >> 5) * (SA_CODE_ROWS / 2);
		const int a_hi = NT == 2 ? SA_CODE_ROWS : a_lo + SA_CODE_ROWS / 2;
		uint8_t *tbl = s_prof + (NT == 2 ? (lane >> 5) * TBLSTRIDE : 0);
		for (int a = a_lo; a < a_hi; a++) {
			uint32_t w[RB / 4];
#pragma unroll
			for (int k = 0; k < RB / 4; k++)
				w[k] = 0x80808080u;
#pragma unroll
			for (int q = 0; q < K; q++) {
				int v = -128;
				if (a < SA_SUB_DIM && bq[q] >= 0) {
					v = (int)s_sub[a * SA_SUB_DIM + bq[q]] + A.pconst;
					/* Gotoh: the first real column sits right of a padding column whose N is
					 * one q too low (see header): compensate in its diagonal term */
					if (METHOD == SA_METHOD_GA && W != n && lig * K + q == W - n)
						v -= A.q;
				}
				w[q >> 2] = (w[q >> 2] & ~(0xffu << (8 * (q & 3)))) | ((uint32_t)(v & 0xff) << (8 * (q & 3)));
			}
			uint32_t *dst = reinterpret_cast<uint32_t *>(tbl + a * ROWSTRIDE + slot * RB);
#pragma unroll
			for (int k = 0; k < RB / 4; k++)
				dst[k] = w[k];
		}
	}
	__syncthreads();

	/* ---- row streams: group g streams sequences [ib_g, ib_g + cnt_g) ---- */
	const int32_t ib_g = i_begin + grp * CH;
	int32_t cnt_g = i_count - grp * CH;
	cnt_g = cnt_g < 0 ? 0 : cnt_g > CH ? CH : cnt_g;
	const int32_t sbeg = A.off[ib_g < A.num ? ib_g : 0];
	const int32_t slen = cnt_g > 0 ? A.off[ib_g + cnt_g] - sbeg : 0;
	int32_t smax = 0; /* longest stream of the wave */
#pragma unroll
	for (int g = 0; g < NG; g++) {
		int32_t c = i_count - g * CH;
		c = c < 0 ? 0 : c > CH ? CH : c;
		const int32_t b = i_begin + g * CH;
		const int32_t l = c > 0 ? A.off[b + c] - A.off[b] : 0;
		smax = l > smax ? l : smax;
	}
	const int32_t steps = smax + G - 1;
	const int32_t nblk = (steps + 15) >> 4;
	const uint8_t *stream = A.codes + sbeg;
	auto load_block = [&](int32_t blk) -> int {
		const int32_t pos = (blk << 4) + (lane & 15);
		return pos < slen ? (int)stream[pos] : (int)SA_CODE_NOP;
	};

	const uint32_t slot_off = (uint32_t)((lane & 31) * RB + (NT == 2 ? (lane >> 5) * TBLSTRIDE : 0));
	const int32_t delta = A.delta;
	const int32_t gq = A.q, go = A.gap_o, ge = A.gap_e;

	/* ---- DP state ---- */
	int V[K];        /* NW: H'   GA: N     SW: No                               */
	int Y[K];        /* GA: Y'   SW: Y                                          */
	int vprev;       /* value of the column left of V[0], previous row (diag)   */
	int xout;        /* GA/SW: X of the column right of V[K-1], current row     */
	int injn;        /* boundary value injected at lane 0 on an ordinary row    */
	int floorB = 0;  /* SW: current baseline = zero floor of the local alignment */
	int best = 0, carry = NEG;
	int nsep = 0;
	if (METHOD == SA_METHOD_NW) {
		injn = 0;
		vprev = 0;
		xout = 0;
#pragma unroll
		for (int q = 0; q < K; q++)
			V[q] = 0, Y[q] = 0;
	} else if (METHOD == SA_METHOD_GA) {
		injn = 2 * gq;
		vprev = leader ? gq : 2 * gq;
		xout = gq;
#pragma unroll
		for (int q = 0; q < K; q++)
			V[q] = 2 * gq, Y[q] = 2 * gq;
	} else {
		injn = go;
		vprev = go;
		xout = go;
#pragma unroll
		for (int q = 0; q < K; q++)
			V[q] = go, Y[q] = go;
	}
	const int cspecial = METHOD == SA_METHOD_GA ? -gq : 0;

	int tok = SA_CODE_NOP;
	int tokq = load_block(0);
	for (int32_t blk = 0; blk < nblk; blk++) {
		const int tokn = load_block(blk + 1);
#pragma unroll
		for (int s = 0; s < 16; s++) {
			tok = shift_in<G>(tokq, tok, leader);
			tokq = dpp_row_shl1(tokq);
			const slot_t pw = *reinterpret_cast<const slot_t *>(s_prof + (((uint32_t)tok << SH) | slot_off));
			int cin = 0;
			if (METHOD == SA_METHOD_SW)
				cin = shift_in<G>(NEG, carry, leader);
			int inj = injn;
			if (tok >= SA_CODE_SEP) {
				if (tok == SA_CODE_SEP) {
					int res = V[K - 1];
					if (METHOD == SA_METHOD_SW) {
						carry = imax(cin, best);
						res = carry;
						floorB += delta;
					}
					if (lig == G - 1)
						s_out[grp * CH + nsep] = res;
					nsep++;
					injn += delta;
				}
				inj = injn + cspecial;
			}
			const int vleft = shift_in<G>(inj, V[K - 1], leader);
			int d[K];
			d[0] = vprev + slot_byte(pw, 0);
#pragma unroll
			for (int q = 1; q < K; q++)
				d[q] = V[q - 1] + slot_byte(pw, q);
			if (METHOD == SA_METHOD_NW) {
				V[0] = imax3(d[0], V[0], vleft);
#pragma unroll
				for (int q = 1; q < K; q++)
					V[q] = imax3(d[q], V[q], V[q - 1]);
			} else if (METHOD == SA_METHOD_GA) {
				int x = shift_in<G>(inj, xout, leader);
#pragma unroll
				for (int q = 0; q < K; q++) {
					const int y = imax(V[q], Y[q]);
					const int m = imax3(d[q], x, y);
					Y[q] = y;
					V[q] = m + gq;
					x = imax(V[q], x);
				}
				xout = x;
			} else {
				int x = shift_in<G>(inj, xout, leader);
#pragma unroll
				for (int q = 0; q < K; q++) {
					const int y = imax(V[q], Y[q] + ge);
					const int m = imax(imax3(d[q], x, y), floorB);
					Y[q] = y;
					V[q] = m + go;
					x = imax(V[q], x + ge);
					best = imax(best, m);
				}
				xout = x;
			}
			vprev = vleft;
		}
		tokq = tokn;
	}
	__syncthreads();

	/* ---- epilogue: un-bias and store, 64 consecutive packed indices per group ---- */
#pragma unroll
	for (int g = 0; g < NG; g++) {
		int32_t c = i_count - g * CH;
		c = c < 0 ? 0 : c > CH ? CH : c;
		if (lane < c) {
			const int32_t i = i_begin + g * CH + lane;
			const int32_t m = A.off[i + 1] - A.off[i] - 1;
			const int32_t raw = s_out[g * CH + lane] - lane * delta;
			int32_t score;
			if (METHOD == SA_METHOD_NW)
				score = raw + (m + n) * A.gap_g;
			else if (METHOD == SA_METHOD_GA)
				score = raw - gq + (m + n) * ge;
			else
				score = raw;
			A.out[tri + i - A.start] = score;
		}
	}
}

template <int METHOD> hipError_t launch_method(int cls, const SaSysArgs &a, int tiles, hipStream_t s)
{
#define SA_CASE(IDX, G_, K_)                                                                              \
	case IDX:                                                                                         \
		hipLaunchKernelGGL((sa_k_systolic<METHOD, G_, K_>), dim3(tiles), dim3(64), 0, s, a);     \
		break;
	switch (cls) {
		SA_SYS_CLASS_LIST(SA_CASE)
	default:
		return hipErrorInvalidValue;
	}
#undef SA_CASE
	return hipGetLastError();
}

} // namespace

hipError_t sa_launch_systolic(int method, int cls, const SaSysArgs &a, int tiles, hipStream_t s)
{
	switch (method) {
	case SA_METHOD_NW:
		return launch_method<SA_METHOD_NW>(cls, a, tiles, s);
	case SA_METHOD_GA:
		return launch_method<SA_METHOD_GA>(cls, a, tiles, s);
	case SA_METHOD_SW:
		return launch_method<SA_METHOD_SW>(cls, a, tiles, s);
	default:
		return hipErrorInvalidValue;
	}
}
