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
 * Row tokens do not travel through registers.  Each lane group keeps the recent part of its stream
 * in a small LDS ring as pre-shifted profile-row addresses; every 16 steps a lane fetches the 16
 * tokens it will meet (stream positions t-l ... t-l+15) with immediate-offset ds_reads, so the
 * systolic "shift" of the row is pure addressing.  Terminators only matter to the group's first lane
 * (inject the next baseline) and last lane (a score leaves the pipeline): a ballot taken when a block
 * of tokens is written to the ring gives a wave-uniform 16-bit event mask per block, tested with a
 * scalar branch per step -- the common step executes no exec-masked code at all.
 *
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

/* lane i of a DPP row receives v of lane i+n of the same row (n is a constant after unrolling) */
__device__ __forceinline__ int dpp_row_shl(int v, int n)
{
	switch (n) {
#define SA_SHL(N) case N: return __builtin_amdgcn_mov_dpp(v, 0x100 + N, 0xf, 0xf, true); /* bound_ctrl: no `old` to initialise */
		SA_SHL(1) SA_SHL(2) SA_SHL(3) SA_SHL(4) SA_SHL(5) SA_SHL(6) SA_SHL(7) SA_SHL(8)
		SA_SHL(9) SA_SHL(10) SA_SHL(11) SA_SHL(12) SA_SHL(13) SA_SHL(14) SA_SHL(15)
#undef SA_SHL
	default:
		return v;
	}
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

/* one wave-tile: column j = A.jlist[..] against 64/G streams of up to 64 row sequences.
 * LONG (G = 64, K = 16 only): column sequences longer than W = 1024 are processed in ceil(n/W) strips of W
 * columns, one full pass over the row stream per strip.  The strip's last column (and, for the affine
 * methods, the X value leaving it) is parked per row in an HBM scratch line of the workgroup and becomes
 * the first lane's injected boundary of the next strip -- the same role the baseline plays in strip 0. */
template <int METHOD, int G, int K, bool LONG>
__device__ __forceinline__ void systolic_tile(const SaSysArgs &A, const int32_t t_raw, const int32_t ntiles)
{
	static_assert(!LONG || (G == 64 && K == 16), "strip mining is instantiated for the widest class only");
	constexpr int WPB = SA_SYS_WPB; /* waves per workgroup, each wave owns one wave-tile and its own LDS */
	/* Single strip: the boundary value the group's first lane injects is constant between terminators (the
	 * baseline), so it simply LIVES in the first lane's copy of the shift register: the row_shr/wave_shr DPP
	 * never writes that lane (no source), and a terminator entering the group raises it on the rare event
	 * path.  Gotoh injects a different value on the terminator row itself (the corner, one q higher): its
	 * event path sets that and restores the plain value two steps later, when the register is used again.
	 * Only the strip-mined kernel (per-row boundaries from scratch) picks a per-block boundary vector with a
	 * second DPP instead. */
	constexpr bool REGINJ = !LONG;
	constexpr int NG = 64 / G;
	constexpr int W = G * K;
	constexpr int RB = K <= 4 ? 4 : K <= 8 ? 8 : 16;  /* bytes per profile slot                    */
	/* Profile table: row = residue code, one slot per DISTINCT column holder.  G = 16: the 16 lanes of
	 * a row (lanes l and l+16 of a ds_read lane group share a slot: same address when their residues
	 * agree, otherwise at worst a 2-way bank conflict -- LDS has ample slack, and the smaller table is
	 * what lets 8 waves/SIMD fit).  G = 32: 32 slots.  G = 64: two 32-slot tables. */
	constexpr int NSLOT = G == 16 ? 16 : 32;
	constexpr int NT = G == 64 ? 2 : 1;
	constexpr int ROWSTRIDE = NSLOT * RB;
	constexpr int SH = (RB == 4 ? 2 : RB == 8 ? 3 : 4) + (NSLOT == 16 ? 4 : 5);
	using slot_t = typename Slot<RB>::type;

	constexpr int TBLSTRIDE = 32 * ROWSTRIDE;        /* power of two */
	constexpr int PROF_BYTES = NT == 2 ? 2 * TBLSTRIDE : SA_CODE_ROWS * ROWSTRIDE;
	__shared__ __attribute__((aligned(16))) uint8_t s_prof_all[WPB * PROF_BYTES];
	/* s_out (scores leaving the pipeline) is only written after the profile build, which is the only
	 * reader of the staged substitution matrix: they share storage */
	constexpr int OUT_INTS = (NG * CH * 4 > SA_SUB_DIM * SA_SUB_DIM ? NG * CH : SA_SUB_DIM * SA_SUB_DIM / 4) +
				 (LONG ? SA_SUB_DIM * SA_SUB_DIM / 4 : 0); /* LONG: the matrix keeps its own words */
	__shared__ int32_t s_out_all[WPB * OUT_INTS];
	/* token ring per lane group: RING stream positions as u16.  Stored twice (index i and i+RING) so a
	 * run of 16 consecutive positions never wraps, and in two copies skewed by one position so that
	 * every lane's run starts on a 4-byte boundary (odd lanes read copy 1): the run is fetched with 8
	 * aligned ds_read_b32.  (Unaligned wide LDS reads serialize: SQ_LDS_UNALIGNED_STALL.) */
	constexpr int RING = G == 16 ? 64 : 128;
	/* bank placement (u16 units; 2 u16 = one 4-byte bank): in one ds_read_b32 lane group (32 lanes) the
	 * even lanes read 8 (G=16: per row) or 16 consecutive dwords of copy 0 and the odd lanes the same
	 * dwords of copy 1, and with G=16 two rows = two groups are in flight: copy 1 is displaced by 8/16
	 * banks and the next group by 16 banks, so the 32 lanes always hit 32 distinct banks. */
	constexpr int COPY1 = 2 * RING + (G == 16 ? 16 : 32);
	constexpr int GSTRIDE = G == 16 ? 288 : COPY1 + 2 * RING;
	__shared__ __attribute__((aligned(16))) uint16_t s_ring_all[WPB * NG * GSTRIDE];

	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	uint8_t *s_prof = s_prof_all + wv * PROF_BYTES;
	int32_t *s_out = s_out_all + wv * OUT_INTS;
	int8_t *s_sub = reinterpret_cast<int8_t *>(LONG ? s_out + NG * CH : s_out);
	uint16_t *s_ring = s_ring_all + wv * (NG * GSTRIDE);
	const int lig = lane & (G - 1);
	const int grp = lane / G;
	const bool leader = lig == 0;

	/* ---- which wave-tile: (column j, block of NG*CH rows) ---- */
	const bool active = t_raw < ntiles;          /* surplus waves of the last workgroup recompute */
	const int32_t t = active ? t_raw : ntiles - 1; /* the last tile and store nothing              */
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
	const int32_t ch = A.chunk; /* sequences per group stream (<= CH): shorter tiles when the range is small */
	const int32_t i_begin = (int32_t)ia64 + chunk * (NG * ch);
	const int32_t i_count = (int32_t)ib64 - i_begin < NG * ch ? (int32_t)ib64 - i_begin : NG * ch;
	const int32_t offj = A.off[j];
	const int32_t n = A.off[j + 1] - offj - 1;
	/* column slots: nstrips*W of them, the sequence right-aligned (padding only left of column 1) */
	const int32_t nstrips = LONG ? (n + W - 1) / W : 1;
	const int32_t pad = nstrips * W - n;
	/* LONG: per-workgroup scratch lines holding, per stream position, what leaves a strip's last column */
	int32_t *const vb = LONG ? A.long_scratch + (size_t)blockIdx.x * (size_t)A.long_stride : nullptr;
	int32_t *const xb = LONG ? vb + (A.long_stride >> 1) : nullptr;

	for (int k = lane; k < SA_SUB_DIM * SA_SUB_DIM; k += 64)
		s_sub[k] = A.sub8[k];
	__syncthreads();

	/* this lane's slot of profile row 0 (a per-lane LDS address: a token's row is one add away) */
	const uint8_t *const lane_prof = s_prof + (uint32_t)((lane & (NSLOT - 1)) * RB + (NT == 2 ? (lane >> 5) * TBLSTRIDE : 0));
	const int32_t delta = A.delta;
	const int32_t gq = A.q, go = A.gap_o, ge = A.gap_e;
	unsigned long long st_c = 0, st_r = 0, st_steps = 0;
	for (int32_t strip_i = 0; strip_i < nstrips; strip_i++) {
		const int32_t strip = LONG ? strip_i : 0; /* a literal 0 in the single-strip kernels */
		if (strip > 0) {
			__syncthreads();
		}
		/* ---- query profile of column sequence j for this lane's K column slots ---- */
		{
			int bq[K];
	#pragma unroll
			for (int q = 0; q < K; q++) {
				const int c0 = strip * W + lig * K + q - pad;
				bq[q] = c0 >= 0 ? (int)A.codes[offj + c0] : -1;
			}
			/* lanes holding the same columns split the table rows between them */
			constexpr int SHARE = NT == 2 ? 1 : 64 / NSLOT;          /* builders per slot: 4, 2 or 1 */
			constexpr int ROWS_EACH = (SA_CODE_ROWS + SHARE - 1) / SHARE;
			const int slot = lane & (NSLOT - 1);
			const int a_lo = NT == 2 ? 0 : (lane / NSLOT) * ROWS_EACH;
			const int a_hi = NT == 2 ? SA_CODE_ROWS : (a_lo + ROWS_EACH < SA_CODE_ROWS ? a_lo + ROWS_EACH : SA_CODE_ROWS);
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
						if (METHOD == SA_METHOD_GA && pad != 0 && strip * W + lig * K + q == pad)
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
		const int32_t ib_g = i_begin + grp * ch;
		int32_t cnt_g = i_count - grp * ch;
		cnt_g = cnt_g < 0 ? 0 : cnt_g > ch ? ch : cnt_g;
		const int32_t sbeg = A.off[ib_g < A.num ? ib_g : 0];
		const int32_t slen = cnt_g > 0 ? A.off[ib_g + cnt_g] - sbeg : 0;
		int32_t smax = 0; /* longest stream of the wave */
	#pragma unroll
		for (int g = 0; g < NG; g++) {
			int32_t c = i_count - g * ch;
			c = c < 0 ? 0 : c > ch ? ch : c;
			const int32_t b = i_begin + g * ch;
			const int32_t l = c > 0 ? A.off[b + c] - A.off[b] : 0;
			smax = l > smax ? l : smax;
		}
		const int32_t steps = smax + G - 1;
		const int32_t nblk = (steps + 15) >> 4;
		st_steps += (unsigned long long)nblk * 16;
		const uint8_t *stream = A.codes + sbeg;
		/* token prefetch: 16 stream bytes per DPP row and block, fetched two blocks ahead of their use.
		 * The load is unconditional (clamped address); the out-of-stream select happens when the block is
		 * converted, so the vmcnt wait lands a full block after the issue. */
		const int32_t last = slen > 0 ? slen - 1 : 0;
		const int r16 = lane & 15;
		auto load_block = [&](int32_t blk) -> int {
			const int32_t pos = (blk << 4) + r16;
			return (int)stream[pos < last ? pos : last];
		};
		constexpr uint32_t SEPWORD = (uint32_t)SA_CODE_SEP << SH;
		constexpr uint32_t NOPWORD = (uint32_t)SA_CODE_NOP << SH;
		auto block_word = [&](int32_t blk, int raw) -> uint32_t { /* token -> profile row address */
			const int32_t pos = (blk << 4) + r16;
			return pos < slen ? (uint32_t)raw << SH : NOPWORD;
		};
		/* every DPP row writes the ring of the group it belongs to; only the group's first row carries
		 * the stream (rows 1.. of a 32/64-lane group write a private scratch half that is never read) */
		uint16_t *ring = s_ring + grp * GSTRIDE; /* copy 0 at [0, 2*RING), copy 1 at [COPY1, COPY1 + 2*RING) */
		const bool feeder = (lane & (G - 1)) < 16;
		auto ring_write = [&](int32_t blk, uint32_t word) {
			if (feeder) {
				const int p = (blk << 4) + r16;
				const int i0 = p & (RING - 1), i1 = (p + 1) & (RING - 1);
				ring[i0] = (uint16_t)word;
				ring[i0 + RING] = (uint16_t)word;
				ring[COPY1 + i1] = (uint16_t)word;
				ring[COPY1 + i1 + RING] = (uint16_t)word;
			}
		};
		/* terminator positions of a block of tokens: wave ballot (bit = lane) */
		auto sep_ballot = [&](uint32_t word) -> unsigned long long { return __ballot(feeder && word == SEPWORD); };
		/* any group: wave-uniform 16-bit mask of the block's terminator positions */
		auto fold16 = [&](unsigned long long m) -> uint32_t {
			m |= m >> 32;
			m |= m >> 16;
			return (uint32_t)m & 0xffffu;
		};

		/* ---- DP state ---- */
		int V[K];        /* NW: H'   GA: N     SW: No                               */
		int Y[K];        /* GA: Y'   SW: Y                                          */
		int vprev;       /* value of the column left of V[0], previous row (diag)   */
		int vl[2];       /* REGINJ: left-neighbour shift registers of even / odd steps (vprev = the other one) */
		int xout;        /* GA/SW: X of the column right of V[K-1], current row     */
		int fl = 0;      /* SW: floor of the row this lane is processing (travels with the row) */
		int flead = 0;   /* SW, REGINJ: the floor the group's first lane starts a row with (the current baseline) */
		int vbase;       /* GA, REGINJ: the plain boundary value of the current sequence (first lane) */
		int best = 0, carry = NEG;
		int nsep = 0;
		/* boundary value the group's first lane injects on an ordinary row of the FIRST sequence:
		 * NW  B,  GA  B + 2q,  SW  B + o  (B = 0); a terminator row injects cspecial more (GA: B' + q) */
		const int inj0 = METHOD == SA_METHOD_NW ? 0 : METHOD == SA_METHOD_GA ? 2 * gq : go;
		const int cspecial = METHOD == SA_METHOD_GA ? -gq : 0;
		vl[0] = vl[1] = vbase = inj0;
		if (METHOD == SA_METHOD_NW) {
			vprev = 0;
			xout = 0;
	#pragma unroll
			for (int q = 0; q < K; q++)
				V[q] = 0, Y[q] = 0;
		} else if (METHOD == SA_METHOD_GA) {
			vprev = (leader && strip == 0) ? gq : 2 * gq;
			if (REGINJ && leader)
				vl[1] = gq; /* the corner of the first sequence (restored to the plain value at step 1) */
			xout = gq;
	#pragma unroll
			for (int q = 0; q < K; q++)
				V[q] = 2 * gq, Y[q] = 2 * gq;
		} else {
			vprev = go;
			xout = go;
	#pragma unroll
			for (int q = 0; q < K; q++)
				V[q] = go, Y[q] = go;
		}

		/* Injection vector of a block: feeder lane k holds the boundary value of the row at stream
		 * position 16*blk + k, i.e. inj0 + DELTA * (terminators at positions <= that one) (+ cspecial on
		 * the terminator row itself).  The group's first lane picks entry s at step s with a constant
		 * row_shl:s DPP, so baseline raises need no branch and no per-step state. */
		int seps_before = 0; /* terminators of this lane's row-stream in earlier blocks */
		/* LONG, strips after the first: the boundary a row injects is what the previous strip's last column
		 * left for that row (scratch), not the baseline; the baseline vector is still needed for the SW floor */
		auto boundary_vec = [&](const int32_t *line, int32_t blk, int fallback) -> int {
			const int32_t pos = (blk << 4) + r16;
			return (LONG && strip > 0) ? line[pos < last ? pos : last] : fallback;
		};
		auto inject_vector = [&](unsigned long long m, uint32_t word) -> int {
			const uint32_t half = (lane & 32) ? (uint32_t)(m >> 32) : (uint32_t)m;
			const uint32_t seg = (lane & 16) ? half >> 16 : half & 0xffffu;
			const int incl = __builtin_popcount(seg & ((2u << r16) - 1u));
			const int v = inj0 + delta * (seps_before + incl) + (word == SEPWORD ? cspecial : 0);
			seps_before += __builtin_popcount(seg);
			return v;
		};

		/* the 16 tokens a lane meets in block blk: stream positions 16*blk - lig + s, two per dword.  They are
		 * kept in ONE rolling 8-dword window: dword k (steps 2k, 2k+1) is refilled with the next block's
		 * tokens at step 2k+2, right after its last use, so no second buffer is live */
		auto ring_ptr = [&](int32_t blk) -> const uint32_t * {
			const int phase = lig & 1;
			return reinterpret_cast<const uint32_t *>(
				ring + phase * COPY1 + (((blk << 4) - lig + phase) & (RING - 1)));
		};
		auto tok_of = [&](const uint32_t (&two)[8], int s) -> uint32_t {
			return (s & 1) ? two[s >> 1] >> 16 : two[s >> 1] & 0xffffu;
		};
		auto prof_row = [&](uint32_t word) -> slot_t {
			return *reinterpret_cast<const slot_t *>(lane_prof + word);
		};

		/* ---- prologue: empty ring, block 0 in the ring, block 1 in flight ---- */
		for (int k = lane; k < NG * GSTRIDE; k += 64)
			s_ring[k] = (uint16_t)NOPWORD;
		__syncthreads();
		/* terminator bits seen by the LAST lane of a group: position p reaches it G-1 steps late.
		 * hi bit k = position t0+k (current block), lo bit 64-d = position t0-d */
		/* Gotoh, REGINJ: the first sequence's corner sits in vl[1] like after a terminator at position -1 */
		unsigned long long ev_lo = (REGINJ && METHOD == SA_METHOD_GA) ? 1ull << 63 : 0ull, ev_hi;
		int basevec = 0, injvec = 0, xinjvec = 0;
		{
			const uint32_t w0 = block_word(0, load_block(0));
			ring_write(0, w0);
			const unsigned long long m0 = sep_ballot(w0);
			ev_hi = fold16(m0);
			if (!REGINJ) {
				basevec = inject_vector(m0, w0);
				injvec = boundary_vec(vb, 0, basevec);
				xinjvec = boundary_vec(xb, 0, basevec);
			}
		}
		int raw_next = load_block(1);
		__syncthreads();
		uint32_t w2[8]; /* rolling token window */
		{
			const uint32_t *rp0 = ring_ptr(0);
	#pragma unroll
			for (int k = 0; k < 8; k++)
				w2[k] = rp0[k];
		}
		constexpr int PD = 4; /* profile rows are requested PD steps ahead of their use */
		slot_t pq[PD];
	#pragma unroll
		for (int s = 0; s < PD; s++)
			pq[s] = prof_row(tok_of(w2, s));

		if (A.stamps) {
			st_c = __builtin_amdgcn_s_memtime();
			st_r = __builtin_amdgcn_s_memrealtime();
		}
		for (int32_t blk = 0; blk < nblk; blk++) {
			/* steps of this block at which the last lane of some group meets a terminator */
			/* ... and (REGINJ) steps at which a terminator enters the first lane of some group */
			const unsigned long long ev_in = !REGINJ ? 0ull
				: METHOD == SA_METHOD_GA ? (ev_hi | (ev_hi << 2) | (ev_lo >> 62)) /* + the restore steps */
							 : ev_hi;
			const uint32_t ev = (uint32_t)(((ev_lo >> (64 - (G - 1))) | (ev_hi << (G - 1)) | ev_in) & 0xffffu);
			/* next block's tokens go into the ring while this block computes */
			const uint32_t wn = block_word(blk + 1, raw_next);
			ring_write(blk + 1, wn);
			const unsigned long long mn = sep_ballot(wn);
			const int basevec_next = REGINJ ? 0 : inject_vector(mn, wn);
			const int injvec_next = REGINJ ? 0 : boundary_vec(vb, blk + 1, basevec_next);
			const int xinjvec_next = REGINJ ? 0 : boundary_vec(xb, blk + 1, basevec_next);
			raw_next = load_block(blk + 2);
			const uint32_t *rpn = ring_ptr(blk + 1);

	#pragma unroll
			for (int s = 0; s < 16; s++) {
				const slot_t pw = pq[s % PD];
				if (s >= 2 && (s & 1) == 0)
					w2[(s - 2) >> 1] = rpn[(s - 2) >> 1];
				pq[s % PD] = prof_row(tok_of(w2, (s + PD) & 15));
				/* boundary value of this row for the group's first lane */
				const int inj = REGINJ ? 0 : dpp_row_shl(injvec, s);
				const int xinj = (LONG && METHOD != SA_METHOD_NW) ? dpp_row_shl(xinjvec, s) : inj;
				const int binj = (LONG && METHOD == SA_METHOD_SW) ? dpp_row_shl(basevec, s) : inj;
				int &vcur = vl[s & 1];
				if (METHOD == SA_METHOD_SW) {
					carry = imax(shift_in<G>(NEG, carry, leader), best);
				}
				if (__builtin_expect((ev >> s) & 1u, 0)) { /* wave-uniform, rare: a score leaves the pipeline */
					/* the empty volatile statement keeps this a scalar branch (s_bitcmp + s_cbranch_scc):
					 * without it the uniform test is folded into the per-lane one and every step pays
					 * v_cmp + s_and_saveexec + s_cbranch_execz (measured: +200 cycles per step and wave) */
					asm volatile("" ::: "memory");
					if (REGINJ && METHOD == SA_METHOD_GA) {
						if (leader) { /* terminator row: the corner; any other event step: the plain value */
							const bool sep = tok_of(w2, s) == SEPWORD;
							vbase += sep ? delta : 0;
							if (sep)
								vl[(s + 1) & 1] = vbase;
							vcur = sep ? vbase + cspecial : vbase;
						}
					} else if (REGINJ && leader && tok_of(w2, s) == SEPWORD) { /* a new sequence starts: raise the baseline */
						vl[0] += delta;
						vl[1] += delta;
						if (METHOD == SA_METHOD_SW)
							flead += delta;
					}
					if (lig == G - 1 && tok_of(w2, s) == SEPWORD) {
						if (METHOD == SA_METHOD_SW) /* local: best over all strips of the sequence's columns */
							s_out[grp * CH + nsep] = (LONG && strip > 0) ? imax(s_out[grp * CH + nsep], carry) : carry;
						else if (!LONG || strip == nstrips - 1) /* global: the sequence's last column */
							s_out[grp * CH + nsep] = V[K - 1];
						nsep++;
					}
				}
				if (REGINJ)
					vprev = vl[(s + 1) & 1];
				int d[K];
				d[0] = vprev + slot_byte(pw, 0);
	#pragma unroll
				for (int q = 1; q < K; q++)
					d[q] = V[q - 1] + slot_byte(pw, q);
				/* the diagonal adds go first: the DPP move below reads V[K-1], written by the last instruction
				 * of the previous step, and would otherwise need s_nop wait states at the head of the step */
				if (REGINJ && K >= 3)
					__builtin_amdgcn_sched_barrier(0);
				const int vleft = REGINJ ? (vcur = shift_in<G>(vcur, V[K - 1], leader)) : shift_in<G>(inj, V[K - 1], leader);
				if (METHOD == SA_METHOD_NW) {
					V[0] = imax3(d[0], V[0], vleft);
	#pragma unroll
					for (int q = 1; q < K; q++)
						V[q] = imax3(d[q], V[q], V[q - 1]);
				} else if (METHOD == SA_METHOD_GA) {
					int x = shift_in<G>(REGINJ ? vleft : xinj, xout, leader);
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
					int x = shift_in<G>(REGINJ ? vleft : xinj, xout, leader);
					fl = REGINJ ? shift_in<G>(flead, fl, leader) /* the first lane: the current baseline */
						    : shift_in<G>(binj - go, fl, leader); /* floor = baseline of the row's sequence */
	#pragma unroll
					for (int q = 0; q < K; q++) {
						const int y = imax(V[q], Y[q] + ge);
						const int m = imax(imax3(d[q], x, y), fl);
						Y[q] = y;
						V[q] = m + go;
						x = imax(V[q], x + ge);
						best = imax(best, m);
					}
					xout = x;
				}
				if (!REGINJ)
					vprev = vleft;
				if (LONG && strip + 1 < nstrips && lig == G - 1) { /* park this row's strip boundary */
					const int32_t pos = (blk << 4) + s - (G - 1);
					if (pos >= 0 && pos < slen) {
						vb[pos] = V[K - 1];
						if (METHOD != SA_METHOD_NW)
							xb[pos] = xout;
					}
				}
			}
			w2[7] = rpn[7];
			basevec = basevec_next;
			injvec = injvec_next;
			xinjvec = xinjvec_next;
			ev_lo = (ev_lo >> 16) | (ev_hi << 48);
			ev_hi = fold16(mn);
		}
	if (LONG) {
		__threadfence_block(); /* parked boundaries visible to this wave's next strip */
		__syncthreads();
	}
	} /* strips */
	if (A.stamps && lane == 0) {
		A.stamps[3 * (size_t)t + 0] = __builtin_amdgcn_s_memtime() - st_c;
		A.stamps[3 * (size_t)t + 1] = __builtin_amdgcn_s_memrealtime() - st_r;
		A.stamps[3 * (size_t)t + 2] = (unsigned long long)st_steps;
	}
	__syncthreads();

	/* ---- epilogue: un-bias and store, 64 consecutive packed indices per group ---- */
#pragma unroll
	for (int g = 0; g < NG; g++) {
		int32_t c = i_count - g * ch;
		c = c < 0 ? 0 : c > ch ? ch : c;
		if (lane < c && active) {
			const int32_t i = i_begin + g * ch + lane;
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

/* Persistent launch: every wave pulls wave-tiles from a device counter until the class is done, so a
 * finished tile is followed by the next one without a workgroup relaunch and the launch drains with at
 * most one tile of imbalance.  (Plain one-tile-per-workgroup grids left ~19 % of the wave slots idle.) */
template <int METHOD, int G, int K, bool LONG = false>
__global__ __launch_bounds__(64 * SA_SYS_WPB) void sa_k_systolic(SaSysArgs A)
{
	__shared__ int32_t s_next;
	const int32_t ntiles = A.tprefix[A.ncols];
	for (;;) {
		if (threadIdx.x == 0)
			s_next = (int32_t)atomicAdd(A.counter, (unsigned)SA_SYS_WPB);
		__syncthreads();
		/* readfirstlane: the tile index is wave-uniform; telling the compiler so keeps the whole tile
		 * geometry (binary search, offsets, ranges) in SGPRs instead of ~35 VGPRs */
		const int32_t base = __builtin_amdgcn_readfirstlane(s_next);
		__syncthreads();
		if (base >= ntiles)
			break;
		systolic_tile<METHOD, G, K, LONG>(A, base + __builtin_amdgcn_readfirstlane((int32_t)(threadIdx.x >> 6)), ntiles);
	}
}

/* SA_HIP_LDS_PAD=bytes: extra (unused) dynamic LDS per workgroup, to study occupancy (development switch) */
static unsigned lds_pad()
{
	static const unsigned pad = [] {
		const char *e = getenv("SA_HIP_LDS_PAD");
		return e ? (unsigned)atoi(e) : 0u;
	}();
	return pad;
}

template <int METHOD> hipError_t launch_method(int cls, const SaSysArgs &a, int tiles, hipStream_t s)
{
#define SA_CASE(IDX, G_, K_)                                                                              \
	case IDX:                                                                                         \
		hipLaunchKernelGGL((sa_k_systolic<METHOD, G_, K_>), dim3(tiles), dim3(64 * SA_SYS_WPB), lds_pad(), s, a);  \
		break;
	switch (cls) {
		SA_SYS_CLASS_LIST(SA_CASE)
	case SA_SYS_CLASS_LONG: /* strip-mined: column sequences longer than the widest class */
		hipLaunchKernelGGL((sa_k_systolic<METHOD, 64, 16, true>), dim3(tiles), dim3(64 * SA_SYS_WPB), 0, s, a);
		break;
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
