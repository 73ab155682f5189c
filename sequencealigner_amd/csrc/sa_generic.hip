/*
 * sa_generic.hip -- "pair per wavefront" anti-diagonal kernels (gfx950, wave64).
 *
 * The always-applicable path: exact s32 restatement of the reference recurrences
 *   NW  src/bio/method/nw.c:14-41      GA  src/bio/method/ga.c:23-67      SW  src/bio/method/sw.c:18-61
 * with no assumption on gap values or sequence length.  One (i,j) pair per
 * wavefront: lane l owns column c = 64*strip + l + 1 of the DP matrix (seq j),
 * rows (seq i) enter at lane 0 and travel one lane per step, so step t of the
 * sweep computes anti-diagonal t of the strip.  The left/diagonal dependency
 * crosses lanes with a wave shift, the strip's last column is parked in a
 * per-wave boundary array so the next 64-column strip can continue from it.
 *
 * It is the fallback of the systolic streaming kernels (sa_systolic.hip), which
 * need bounded scores and len <= their column budget; this one does not.
 */
#include <algorithm>
#include "sa_internal.h"

namespace {

constexpr int32_t SCORE_MIN = INT32_MIN / 2; /* reference src/bio/align.h:19 */

__device__ __forceinline__ int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }

/* packed index -> (i,j), i<j : largest j with j(j-1)/2 <= p
 * (what the reference does by binary search, src/bio/kernels.cu:17-30) */
__device__ __forceinline__ void unpack_pair(int64_t p, int32_t &i, int32_t &j)
{
	int64_t jj = (int64_t)((1.0 + sqrt(1.0 + 8.0 * (double)p)) * 0.5);
	while (jj * (jj - 1) / 2 > p)
		--jj;
	while ((jj + 1) * jj / 2 <= p)
		++jj;
	j = (int32_t)jj;
	i = (int32_t)(p - jj * (jj - 1) / 2);
}

/* value shifted in from lane-1 (lane 0 keeps its own and overrides it afterwards) */
__device__ __forceinline__ int32_t from_left(int32_t v)
{
	/* DPP wave_shr:1 -- full 64-lane shift, bound_ctrl off: lane 0 keeps `v` */
	return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false);
}

__device__ __forceinline__ int32_t wave_max(int32_t v)
{
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1)
		v = imax(v, __shfl_xor(v, d, 64));
	return v;
}

template <int METHOD>
__global__ __launch_bounds__(256) void sa_k_pair_per_wave(SaGenericArgs A)
{
	__shared__ int32_t s_sub[SA_SUB_DIM * SA_SUB_DIM];
	for (int k = threadIdx.x; k < SA_SUB_DIM * SA_SUB_DIM; k += blockDim.x)
		s_sub[k] = A.sub[k];
	__syncthreads();

	const int lane = threadIdx.x & 63;
	const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
	int32_t *bndM = A.scratch + wave * A.scratch_stride;
	int32_t *bndX = bndM + (A.scratch_stride >> 1);

	const int32_t g = A.gap_pen, o = A.gap_opn, e = A.gap_ext;
	/* closed forms of the reference's iteratively filled borders:
	 * NW nw.c:16-20: k*g.  GA ga.c:26-38: B(1) = max(0+o, SCORE_MIN+e), B(k) = B(k-1)+max(o,e). */
	const int32_t ga_b1 = imax(o, SCORE_MIN + e);
	const int32_t ga_w = imax(o, e);
	auto border = [&](int32_t k) -> int32_t {
		if (METHOD == SA_METHOD_NW)
			return k * g;
		if (METHOD == SA_METHOD_GA)
			return k == 0 ? 0 : ga_b1 + (k - 1) * ga_w;
		return 0;
	};

	for (int64_t q = wave; q < A.count; q += nwaves) {
		int32_t i, j;
		unpack_pair(A.start + q, i, j);
		const int32_t m = A.st.meta[i].len, offi = A.st.meta[i].off; /* rows: seq i */
		const int32_t n = A.st.meta[j].len, offj = A.st.meta[j].off; /* cols: seq j */
		const uint8_t *ci = A.st.codes + offi;
		const uint8_t *cj = A.st.codes + offj;
		const int32_t nstrips = (n + 63) >> 6;
		int32_t best = 0; /* SW running maximum, sw.c:33,57 */
		int32_t h = 0;

		for (int32_t s = 0; s < nstrips; s++) {
			const int32_t c = (s << 6) + lane + 1;
			const bool colvalid = c <= n;
			const int32_t b = colvalid ? cj[c - 1] : 0;
			const int32_t width = (n - (s << 6)) < 64 ? (n - (s << 6)) : 64;
			const bool last_strip = s + 1 == nstrips;
			h = border(c);            /* M[0][c]                                   */
			int32_t y = SCORE_MIN;    /* gap_y[0][c]          ga.c:29, sw.c:22     */
			int32_t x = SCORE_MIN;    /* gap_x[r][c] of the row just computed      */
			int32_t diag = border(c - 1); /* M[0][c-1] until the first row arrives */
			const int32_t steps = m + width - 1;

			for (int32_t t = 0; t < steps; t++) {
				const int32_t r = t - lane + 1;
				int32_t lm = from_left(h);
				int32_t lx = SCORE_MIN;
				if (METHOD != SA_METHOD_NW)
					lx = from_left(x);
				if (lane == 0) {
					if (s == 0) {
						lm = border(r);   /* M[r][0]   nw.c:19, ga.c:32-37, sw.c:26-29 */
						lx = SCORE_MIN;   /* gap_x[r][0]                               */
					} else if (r <= m) {
						lm = bndM[r];
						if (METHOD != SA_METHOD_NW)
							lx = bndX[r];
					}
				}
				const bool valid = colvalid && r >= 1 && r <= m;
				const int32_t a = valid ? ci[r - 1] : 0;
				int32_t nm, nx = SCORE_MIN, ny = SCORE_MIN;
				if (METHOD == SA_METHOD_NW) {
					/* nw.c:29-35 */
					const int32_t match = diag + s_sub[a * SA_SUB_DIM + b];
					const int32_t del = h + g;
					const int32_t ins = lm + g;
					nm = imax(ins, imax(del, match));
				} else {
					/* ga.c:46-63 / sw.c:39-57 */
					const int32_t sd = diag + s_sub[b * SA_SUB_DIM + a];
					nx = imax(lm + o, lx + e);
					ny = imax(h + o, y + e);
					nm = (METHOD == SA_METHOD_SW) ? imax(sd, 0) : sd;
					nm = imax(nx, nm);
					nm = imax(ny, nm);
				}
				diag = lm;
				if (valid) {
					h = nm;
					x = nx;
					y = ny;
					if (METHOD == SA_METHOD_SW)
						best = imax(best, nm);
					if (lane == 63 && !last_strip) {
						bndM[r] = nm;
						if (METHOD != SA_METHOD_NW)
							bndX[r] = nx;
					}
				}
			}
			if (!last_strip)
				__threadfence_block(); /* boundary column visible to this wave's next strip */
		}

		int32_t score;
		if (METHOD == SA_METHOD_SW)
			score = wave_max(best);
		else
			score = __shfl(h, (n - 1) & 63, 64); /* M[m][n] sits in the lane owning column n */
		if (lane == 0) {
			if (A.out16)
				reinterpret_cast<int16_t *>(A.out)[q] = (int16_t)score;
			else
				A.out[q] = score;
			if (A.host_out)
				__builtin_nontemporal_store(score, &A.host_out[q]);
		}
	}
}

/* Packed triangular -> full symmetric (reference layout src/io/output.c:76-81), zero diagonal.
 * One thread per element of the full matrix: reads are gathers from the packed vector
 * (contiguous along i for fixed j), writes are fully coalesced rows. */
__global__ __launch_bounds__(256) void sa_k_expand_full(const int32_t *__restrict__ packed,
							  int32_t *__restrict__ full, int32_t num)
{
	const int64_t total = (int64_t)num * num;
	for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
	     e += (int64_t)gridDim.x * blockDim.x) {
		const int64_t r = e / num, c = e - r * num;
		int32_t v = 0;
		if (r != c) {
			const int64_t lo = r < c ? r : c, hi = r < c ? c : r;
			v = packed[hi * (hi - 1) / 2 + lo];
		}
		full[e] = v;
	}
}

/* The L-shaped shell of the full matrix owned by columns [ja, jb): rows [ja, jb) x cols [0, jb) (region A, row-major
 * first) and rows [0, ja) x cols [ja, jb) (region B).  Every element's pair has its larger index in [ja, jb), i.e.
 * belongs to the packed range tri(ja) .. tri(jb) that was just computed; packed[p - pbase] holds pair p. */
__global__ __launch_bounds__(256) void sa_k_expand_shell(const int32_t *__restrict__ packed, int64_t pbase,
							   int32_t *__restrict__ full, int32_t num, int32_t ja, int32_t jb)
{
	const int64_t na = (int64_t)(jb - ja) * jb, nb = (int64_t)ja * (jb - ja);
	for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < na + nb;
	     e += (int64_t)gridDim.x * blockDim.x) {
		int64_t r, c;
		if (e < na) {
			r = ja + e / jb;
			c = e % jb;
		} else {
			const int64_t f = e - na;
			r = f / (jb - ja);
			c = ja + f % (jb - ja);
		}
		int32_t v = 0;
		if (r != c) {
			const int64_t lo = r < c ? r : c, hi = r < c ? c : r;
			v = packed[hi * (hi - 1) / 2 + lo - pbase];
		}
		full[r * num + c] = v;
	}
}

} // namespace

hipError_t sa_launch_generic(int method, const SaGenericArgs &a, int blocks, hipStream_t s)
{
	switch (method) {
	case SA_METHOD_NW:
		hipLaunchKernelGGL(sa_k_pair_per_wave<SA_METHOD_NW>, dim3(blocks), dim3(256), 0, s, a);
		break;
	case SA_METHOD_GA:
		hipLaunchKernelGGL(sa_k_pair_per_wave<SA_METHOD_GA>, dim3(blocks), dim3(256), 0, s, a);
		break;
	case SA_METHOD_SW:
		hipLaunchKernelGGL(sa_k_pair_per_wave<SA_METHOD_SW>, dim3(blocks), dim3(256), 0, s, a);
		break;
	default:
		return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

const char *sa_generic_kernel_name(int method)
{
	static const char *names[] = { "sa_k_pair_per_wave<nw>", "sa_k_pair_per_wave<ga>", "sa_k_pair_per_wave<sw>" };
	return (method >= 0 && method < 3) ? names[method] : "?";
}

/* int16 exchange format back to the reference's s32 (sign extension), 8 scores per thread and pass */
__global__ __launch_bounds__(256) void sa_k_widen16(const int16_t *__restrict__ src, int32_t *__restrict__ dst, int64_t count,
						     int64_t nvec /* 16-byte groups (0 when a pointer is not 16-byte aligned) */)
{
	const int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
		const uint4 w = reinterpret_cast<const uint4 *>(src)[v];
		int4 lo, hi;
		lo.x = (int16_t)(w.x & 0xffffu), lo.y = (int16_t)(w.x >> 16), lo.z = (int16_t)(w.y & 0xffffu), lo.w = (int16_t)(w.y >> 16);
		hi.x = (int16_t)(w.z & 0xffffu), hi.y = (int16_t)(w.z >> 16), hi.z = (int16_t)(w.w & 0xffffu), hi.w = (int16_t)(w.w >> 16);
		reinterpret_cast<int4 *>(dst)[2 * v] = lo;
		reinterpret_cast<int4 *>(dst)[2 * v + 1] = hi;
	}
	for (int64_t k = (nvec << 3) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += stride)
		dst[k] = src[k];
}

hipError_t sa_launch_widen16(const int16_t *src, int32_t *dst, int64_t count, hipStream_t s)
{
	if (count <= 0)
		return hipSuccess;
	int64_t blocks = ((count >> 3) + 255) / 256;
	blocks = blocks < 1 ? 1 : blocks > 256 * 16 ? 256 * 16 : blocks;
	const bool aligned = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0;
	hipLaunchKernelGGL(sa_k_widen16, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, count, aligned ? count >> 3 : (int64_t)0);
	return hipGetLastError();
}

hipError_t sa_launch_expand_shell(const int32_t *packed, int64_t pbase, int32_t *full, int32_t num, int32_t ja, int32_t jb,
				  hipStream_t s)
{
	const int64_t total = (int64_t)(jb - ja) * jb + (int64_t)ja * (jb - ja);
	if (total <= 0)
		return hipSuccess;
	int64_t blocks = (total + 255) / 256;
	if (blocks > 256 * 16)
		blocks = 256 * 16;
	hipLaunchKernelGGL(sa_k_expand_shell, dim3((unsigned)blocks), dim3(256), 0, s, packed, pbase, full, num, ja, jb);
	return hipGetLastError();
}

hipError_t sa_launch_expand_full(const int32_t *packed, int32_t *full, int32_t num, hipStream_t s)
{
	const int64_t total = (int64_t)num * num;
	int64_t blocks = (total + 255) / 256;
	if (blocks > 256 * 16)
		blocks = 256 * 16;
	hipLaunchKernelGGL(sa_k_expand_full, dim3((unsigned)blocks), dim3(256), 0, s, packed, full, num);
	return hipGetLastError();
}

/* Widen-and-place (tile-interleaved sharding, sa_ctx_place_shares): the gathered dense shares -- every rank's scores in
 * tile order, int16 or s32 -- into the reference's packed order.  One workgroup per run of a tile (SaPlaceSeg): reads are
 * contiguous, writes land inside one arranged block of the column (<= 2048 rows = an 8 KB window the L2 merges), in row
 * order where the tile streamed in store order. */
template <typename T>
__global__ __launch_bounds__(256) void sa_k_place(const SaPlaceSeg *__restrict__ segs, int32_t nsegs, const T *__restrict__ shares,
						   int32_t *__restrict__ packed)
{
	/* One WAVE per run and pass (a run is at most a tile's rows; a workgroup per run was 240 000 workgroups of one
	 * iteration each for cfg 2 at 8 ranks -- 0.45 ms of dispatch).  A tile that is its own arranged block holds a
	 * permutation of the rows [pos0, pos0 + count): it is placed OUTPUT-driven -- a lane takes four consecutive rows,
	 * looks their positions up (posmap: one 16-byte load), gathers the four scores from the run (a 1-2 KB window) and
	 * stores 16 contiguous bytes.  Position-driven scattering made the L2 write partial lines (1.2 TB/s for the pass). */
	const int lane = threadIdx.x & 63;
	const int32_t wave = (int32_t)((blockIdx.x * 256 + threadIdx.x) >> 6), nwaves = (int32_t)(gridDim.x * 4);
	typedef int32_t i32x4 __attribute__((ext_vector_type(4), aligned(4)));
	constexpr int WIN = 2048; /* elements of a wave's LDS window = rows of the largest own-block tile (eight waves x 8 streams x 32) */
	__shared__ __attribute__((aligned(16))) T s_win[4 * WIN];
	for (int32_t k = wave; k < nsegs; k += nwaves) {
		const SaPlaceSeg sg = segs[k];
		const T *src = shares + sg.src;
		int32_t *dst = packed + sg.dst;
		if (sg.flags & 1) { /* map = posmap (row -> position), rows [pos0, pos0 + count) */
			/* A run is one tile's rows (<= 1024).  The run is copied into this wave's LDS window with coalesced 16-byte loads
			 * (runs are padded to multiples of 8 elements: SA_SHARE_PAD), the positions of four groups of four rows per lane
			 * are looked up while that copy is in flight, and the scores are then gathered from LDS: the texture path sees
			 * only whole cache lines (2-byte gathers from global memory kept the pass at 2.2 TB/s). */
			T *const win = s_win + (threadIdx.x >> 6) * WIN;
			if (sg.count > WIN) { /* (no tile is that large today: plain gathers) */
				for (int32_t q = lane; q < sg.count; q += 64)
					if (sg.pos0 + q >= sg.ia && sg.pos0 + q < sg.ib)
						dst[sg.pos0 + q] = (int32_t)src[sg.map[sg.pos0 + q] - sg.pos0];
				continue;
			}
			{
				constexpr int32_t q0 = 0;
				const int32_t n = sg.count, npad = (n + 7) & ~7;
				constexpr int PER = 16 / (int)sizeof(T); /* elements per 16-byte load */
				typedef T tvec __attribute__((ext_vector_type(16 / sizeof(T)), aligned(16)));
				for (int32_t e = PER * lane; e < npad; e += PER * 64)
					*reinterpret_cast<tvec *>(win + e) = *reinterpret_cast<const tvec *>(src + q0 + e);
				i32x4 pp[4], v[4];
				bool whole[4];
#pragma unroll
				for (int g = 0; g < 4; g++) {
					const int32_t q = q0 + 256 * g + 4 * lane;
					whole[g] = q + 4 <= sg.count;
					if (whole[g])
						pp[g] = *reinterpret_cast<const i32x4 *>(sg.map + sg.pos0 + q);
				}
				/* (a tile that is its own block: the positions of its rows are its own positions, pos0 + [0, count)) */
				const int32_t off = sg.pos0 + q0;
#pragma unroll
				for (int g = 0; g < 4; g++)
					if (whole[g])
						v[g] = i32x4{ (int32_t)win[pp[g].x - off], (int32_t)win[pp[g].y - off], (int32_t)win[pp[g].z - off],
							      (int32_t)win[pp[g].w - off] };
#pragma unroll
				for (int g = 0; g < 4; g++) {
					const int32_t q = q0 + 256 * g + 4 * lane, r = sg.pos0 + q;
					if (whole[g] && r >= sg.ia && r + 4 <= sg.ib) {
						*reinterpret_cast<i32x4 *>(dst + r) = v[g];
						continue;
					}
					for (int e = 0; e < 4 && q + e < sg.count; e++) /* a column whose range ends inside these four rows; the run's tail */
						if (r + e >= sg.ia && r + e < sg.ib)
							dst[r + e] = (int32_t)src[sg.map[r + e] - sg.pos0];
				}
			}
		} else { /* map = rowmap (position -> row) or null (store order) */
			for (int32_t p = lane; p < sg.count; p += 64) {
				const int32_t r = sg.map ? sg.map[sg.pos0 + p] : sg.pos0 + p;
				if (r >= sg.ia && r < sg.ib)
					dst[r] = (int32_t)src[p];
			}
		}
	}
}

hipError_t sa_launch_place(const SaPlaceSeg *segs, int32_t nsegs, const void *shares, int elem16, int32_t *packed, hipStream_t s)
{
	if (nsegs <= 0)
		return hipSuccess;
	const unsigned blocks = (unsigned)std::min<int64_t>(((int64_t)nsegs + 3) / 4, 256 * 8);
	if (elem16)
		hipLaunchKernelGGL(sa_k_place<int16_t>, dim3(blocks), dim3(256), 0, s, segs, nsegs, static_cast<const int16_t *>(shares), packed);
	else
		hipLaunchKernelGGL(sa_k_place<int32_t>, dim3(blocks), dim3(256), 0, s, segs, nsegs, static_cast<const int32_t *>(shares), packed);
	return hipGetLastError();
}

/* see sa_warm_kernels: the pair-per-wave, expand, widen and place kernels live in this translation unit */
hipError_t sa_warm_generic(void)
{
	hipFuncAttributes attr;
	return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&sa_k_expand_full));
}
