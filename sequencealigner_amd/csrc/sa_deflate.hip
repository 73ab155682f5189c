/*
 * sa_deflate.hip -- the `-z` option on the device: the tiles (HDF5 chunks) of /similarity_matrix leave the GPU as finished
 * zlib streams, built from the packed scores where they were computed.
 *
 * Reference: src/io/format/hdf5.c:91-95 sets H5Pset_deflate on the chunked dataset and H5Dwrite (:148-194) runs libhdf5's
 * deflate filter over every chunk in the one writing thread.  For BASELINE config 5 (89 994 sequences after `-f 0.9`,
 * 484 chunks of 4096 x 4096 int32 = 32.5 GB, level 6) that is 41 CPU-minutes behind a 2-second alignment
 * (profiles/r04_cli_cfg5_full_size_end_to_end.txt: Output 157 s with every core of the box deflating).  DEFLATE over int32
 * scores with a fixed parse (sa_deflate_core.h) is byte work without a serial dependency between elements: it belongs
 * where the scores are.
 *
 * A tile is cut into SEGMENTS of ZSEG elements (64 KB of its row-major bytes), each an independent dynamic-Huffman block
 * that ends byte-aligned; ZGROUP consecutive segments (1 MB) share one set of Huffman codes.  Per tile row:
 *   sa_k_deflate_hist     one workgroup per segment: its elements straight from the packed triangle (lower part:
 *                         contiguous; upper part: the mirrored element; diagonal and the padding beyond N: 0) into LDS,
 *                         match choice, histograms (LDS atomics, then one global atomic per used symbol into the
 *                         group's histogram), Adler sums.
 *   sa_k_deflate_codes    one workgroup per group: 512-key bitonic sort of the literal alphabet, then ONE thread builds the
 *                         three length-limited codes and the block header (sa_deflate_core.h: the code the host harness
 *                         tests) -- ~0.4 ms of dependent LDS round trips, which is why it runs once per megabyte and
 *                         not in every segment's workgroup (first version: 460 us of a segment's 500 were this).
 *   sa_k_deflate_encode   one workgroup per segment: elements into LDS again, the group's code tables and header, then
 *                         rounds of 256 x ZE elements: bits per thread, workgroup scan, ds_or into an LDS bit stage, full
 *                         words out to the segment's slot.
 *   sa_k_deflate_offsets  one workgroup per tile: where each segment goes in the tile's stream, Adler-32 of the tile.
 *   sa_k_deflate_rowbase  where each tile goes in the row's compact buffer.
 *   sa_k_deflate_gather   one workgroup per segment: its bytes to their place (dword copies with a funnel shift
 *                         between the byte alignment of slot and stream); the first one adds the tile's header and end.
 * Host: a job hands out batches of tiles -- tile rows over a finished matrix, or, while the alignment is still running, the
 * arms of SHELLS (see SaZBatch); the next batches are encoded and copied while the caller hands this one to H5Dwrite_chunk
 * (the copy's length is not known when it is enqueued: it takes this batch's bytes per tile plus 3 %, and what that
 * misses -- normally nothing -- follows when the sizes are there).
 */
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#include "sa_ctx.h"
#include "sa_deflate_core.h"

namespace {

constexpr int ZT = 256;     /* threads of a workgroup                                  */
constexpr int ZE = 4;       /* elements per thread and round of the encoder            */
constexpr int ZSEG = 16384; /* elements per segment: 64 KB of the tile                 */
constexpr int ZGROUP = 16;  /* segments that share one set of codes: 1 MB of the tile  */
constexpr int ZHDR_WORDS = (SA_Z_HEADER_BITS + 31) / 32 + 2;
/* the LDS bit stage of the encoder holds the block header and one round */
constexpr int ZSTAGE_WORDS = (SA_Z_HEADER_BITS + ZT * ZE * SA_Z_ELEM_BITS + 31) / 32 + 8;
/* a segment's slot: header + 63 bits per element at the very worst, in words, a multiple of four */
constexpr int ZSLOT_WORDS = 2 * ZSEG + 256;
constexpr int ZHIST = 320; /* literal / length counters [0, 288), distance counters [288, 320) */

struct SaZGroup { /* what sa_k_deflate_codes leaves for the segments of a group */
	uint32_t lcode[288], dcode[32];
	uint32_t hdr_bits, pad_[3];
	uint32_t hdr[ZHDR_WORDS];
};

struct SaZArgs {
	const int32_t *packed; /* scores by packed pair index j (j - 1) / 2 + i, i < j (src/io/output.c:76-83)     */
	const int32_t *full;   /* ... or the full num x num matrix (packed == nullptr)                              */
	int32_t num, chunk, chunk_shift;
	int32_t tile_r0, tile_c0, tile_dr, tile_dc; /* tile t of the batch is (tile_r0 + t * tile_dr, tile_c0 + t * tile_dc) */
	int32_t nseg, ngrp;    /* segments / code groups per tile                                                   */
	uint32_t *slots;       /* [tile][segment][ZSLOT_WORDS]                                                      */
	uint32_t *seg_bytes, *seg_s1, *seg_s2, *seg_off; /* [tile][segment]                                        */
	uint32_t *ghist;       /* [tile][group][ZHIST]                                                              */
	SaZGroup *groups;      /* [tile][group]                                                                     */
	unsigned long long *tile_bytes, *tile_base; /* [tile], [tile + 1]: length of a tile's stream, its place in the row */
	uint32_t *tile_adler;  /* [tile]                                                                            */
	uint8_t *out;          /* the row's streams, compact (every tile starts on a 64-byte boundary)              */
	uint32_t *raw;         /* [tile][chunk][chunk]: the row's tiles as they are (sa_k_tiles_raw)                 */
};

__device__ __forceinline__ uint32_t z_fetch(const SaZArgs &A, int64_t i, int64_t j)
{
	if (i >= A.num || j >= A.num)
		return 0u;
	if (!A.packed)
		return (uint32_t)A.full[i * A.num + j];
	if (i == j)
		return 0u; /* the diagonal is never computed (src/io/output.c:76-81) and written as 0 */
	const int64_t hi = i > j ? i : j, lo = i > j ? j : i;
	return (uint32_t)A.packed[hi * (hi - 1) / 2 + lo];
}

/* Four consecutive elements of a segment and the eight before them, from the row's raw tiles: three 16-byte loads per
 * lane, coalesced (the two older quads are the neighbouring lanes' own: cache hits).  e[8 + t] is element 4 q + t. */
__device__ __forceinline__ void z_load_quad(const uint4 *seg4, uint32_t q, uint32_t (&e)[12])
{
	const uint4 z = { 0u, 0u, 0u, 0u };
	const uint4 c = seg4[q], p1 = q >= 1 ? seg4[q - 1] : z, p2 = q >= 2 ? seg4[q - 2] : z;
	e[0] = p2.x, e[1] = p2.y, e[2] = p2.z, e[3] = p2.w;
	e[4] = p1.x, e[5] = p1.y, e[6] = p1.z, e[7] = p1.w;
	e[8] = c.x, e[9] = c.y, e[10] = c.z, e[11] = c.w;
}
/* sa_z_match (sa_deflate_core.h) on registers: which of the eight elements before element k = 4 q + t has its bytes 1..3 */
__device__ __forceinline__ int z_match_quad(const uint32_t (&e)[12], uint32_t k, int t)
{
	const uint32_t hi = e[8 + t] >> 8;
	uint32_t m = 0;
#pragma unroll
	for (int j = 1; j <= SA_Z_J; j++)
		m |= ((e[8 + t - j] >> 8) == hi ? 1u : 0u) << (j - 1);
	if (k < (uint32_t)SA_Z_J) /* elements of the segment only */
		m &= (1u << k) - 1u;
	return m ? __ffs((int)m) : 0;
}
__device__ __forceinline__ const uint4 *z_segment(const SaZArgs &A, int seg, int tile, uint32_t *n)
{
	const uint32_t tile_elems = (uint32_t)A.chunk * (uint32_t)A.chunk;
	const uint32_t e0 = (uint32_t)seg * ZSEG;
	*n = tile_elems - e0 < (uint32_t)ZSEG ? tile_elems - e0 : (uint32_t)ZSEG; /* (a multiple of 4: tiles are at least 64 x 64) */
	return reinterpret_cast<const uint4 *>(A.raw + (size_t)tile * tile_elems + e0);
}

__device__ __forceinline__ void z_or_bits(uint32_t *stage, uint32_t pos, uint64_t bits, uint32_t n)
{
	if (!n)
		return;
	const uint32_t w = pos >> 5, s = pos & 31u;
	const uint32_t lo = (uint32_t)bits, hi = (uint32_t)(bits >> 32);
	const uint32_t x0 = lo << s;
	const uint32_t x1 = s ? (lo >> (32u - s)) | (hi << s) : hi;
	const uint32_t x2 = s ? hi >> (32u - s) : 0u;
	if (x0)
		atomicOr(&stage[w], x0);
	if (x1)
		atomicOr(&stage[w + 1], x1);
	if (x2)
		atomicOr(&stage[w + 2], x2);
}

__global__ __launch_bounds__(ZT) void sa_k_deflate_hist(SaZArgs A)
{
	__shared__ uint32_t hist[ZHIST];
	__shared__ unsigned long long s_s1, s_s2;
	const int tid = threadIdx.x, lane = tid & 63;
	const int seg = blockIdx.x, tile = blockIdx.y;
	uint32_t n;
	const uint4 *const seg4 = z_segment(A, seg, tile, &n);
	for (int s = tid; s < ZHIST; s += ZT)
		hist[s] = 0;
	if (tid == 0)
		s_s1 = s_s2 = 0ull;
	__syncthreads();

	unsigned long long s1 = 0, s2 = 0;
	const unsigned long long len = 4ull * n;
	for (uint32_t q0 = 0; q0 < n / 4; q0 += ZT) {
		const uint32_t q = q0 + tid;
		const bool live = q < n / 4;
		uint32_t e[12] = {};
		if (live)
			z_load_quad(seg4, q, e);
#pragma unroll
		for (int t = 0; t < 4; t++) {
			const uint32_t k = 4u * q + (uint32_t)t, v = e[8 + t];
			int j = -1;
			if (live) {
				j = z_match_quad(e, k, t);
				atomicAdd(&hist[v & 255u], 1u);
				if (!j) {
					atomicAdd(&hist[(v >> 8) & 255u], 1u);
					atomicAdd(&hist[(v >> 16) & 255u], 1u);
					atomicAdd(&hist[v >> 24], 1u);
				} else if (j > 1) {
					atomicAdd(&hist[288 + sa_z_dcode(j)], 1u);
				}
				const uint32_t b0 = v & 255u, b1 = (v >> 8) & 255u, b2 = (v >> 16) & 255u, b3 = v >> 24;
				const uint32_t sum = b0 + b1 + b2 + b3;
				s1 += sum;
				s2 += (len - 4ull * k) * sum - (b1 + 2u * b2 + 3u * b3);
			}
			/* (the two symbols nearly every element has -- length 3, distance 4 -- are counted per wave, not per lane: 64
			 * lanes adding to one LDS word are served one after the other) */
			const unsigned long long m_any = __ballot(j > 0), m_one = __ballot(j == 1);
			if (lane == 0) {
				if (m_any)
					atomicAdd(&hist[SA_Z_LEN3], (uint32_t)__popcll(m_any));
				if (m_one)
					atomicAdd(&hist[288 + sa_z_dcode(1)], (uint32_t)__popcll(m_one));
			}
		}
	}
	for (int d = 32; d > 0; d >>= 1) {
		s1 += ((unsigned long long)(uint32_t)__shfl_down((int)(s1 >> 32), d) << 32) | (uint32_t)__shfl_down((int)(uint32_t)s1, d);
		s2 += ((unsigned long long)(uint32_t)__shfl_down((int)(s2 >> 32), d) << 32) | (uint32_t)__shfl_down((int)(uint32_t)s2, d);
	}
	if (lane == 0) {
		atomicAdd(&s_s1, s1);
		atomicAdd(&s_s2, s2);
	}
	__syncthreads();
	uint32_t *const gh = A.ghist + ((size_t)tile * (size_t)A.ngrp + (size_t)(seg / ZGROUP)) * ZHIST;
	for (int s = tid; s < ZHIST; s += ZT) {
		const uint32_t c = hist[s] + (s == SA_Z_EOB ? 1u : 0u); /* every segment ends its block */
		if (c)
			atomicAdd(&gh[s], c);
	}
	if (tid == 0) {
		const size_t at = (size_t)tile * (size_t)A.nseg + (size_t)seg;
		A.seg_s1[at] = (uint32_t)(s_s1 % 65521ull);
		A.seg_s2[at] = (uint32_t)(s_s2 % 65521ull);
	}
}

__global__ __launch_bounds__(ZT) void sa_k_deflate_codes(SaZArgs A)
{
	__shared__ SaZWork W;
	__shared__ uint32_t hdr[ZHDR_WORDS];
	__shared__ uint32_t s_hdr_bits;
	const int tid = threadIdx.x;
	const size_t g = (size_t)blockIdx.y * (size_t)A.ngrp + (size_t)blockIdx.x;
	const uint32_t *const gh = A.ghist + g * ZHIST;
	for (int s = tid; s < 288; s += ZT)
		W.lfreq[s] = gh[s];
	if (tid < 32)
		W.dfreq[tid] = gh[288 + tid];
	for (int s = tid; s < ZHDR_WORDS; s += ZT)
		hdr[s] = 0;
	/* ---- literal alphabet by ascending weight: 512 keys (weight << 9 | symbol), bitonic, in the tree's storage.  A group
	 * holds at most ZGROUP * (4 * ZSEG + 1) symbols: the weight fits 23 bits ---- */
	uint32_t *const keys = W.w;
	static_assert(sizeof(W.w) >= 512 * sizeof(uint32_t), "the sort keys fit the tree storage");
	static_assert((uint64_t)ZGROUP * (4ull * ZSEG + 1ull) < (1ull << 23), "weight << 9 | symbol fits 32 bits");
	__syncthreads();
	for (int s = tid; s < 512; s += ZT)
		keys[s] = s < SA_Z_NLIT && W.lfreq[s] ? (W.lfreq[s] << 9) | (uint32_t)s : 0xffffffffu;
	__syncthreads();
	for (int k = 2; k <= 512; k <<= 1) {
		for (int j = k >> 1; j > 0; j >>= 1) {
			for (int i = tid; i < 512; i += ZT) {
				const int p = i ^ j;
				if (p > i) {
					const uint32_t a = keys[i], b = keys[p];
					if ((a > b) == ((i & k) == 0)) {
						keys[i] = b;
						keys[p] = a;
					}
				}
			}
			__syncthreads();
		}
	}
	for (int s = tid; s < 512; s += ZT) {
		const uint32_t key = keys[s];
		if (key != 0xffffffffu) {
			if (s < 288)
				W.order[s] = (uint16_t)(key & 511u);
			if (s == 511 || keys[s + 1] == 0xffffffffu)
				W.used = (uint32_t)s + 1u;
		}
	}
	__syncthreads();
	/* ---- one thread: code lengths, codes, block header (sa_deflate_core.h) ---- */
	if (tid == 0) {
		sa_z_alphabet(W, W.lfreq, SA_Z_NLIT, 15, W.llen, W.lcode, (int)W.used, true);
		sa_z_alphabet(W, W.dfreq, SA_Z_NDIST, 15, W.dlen, W.dcode, 0, false);
		SaZBits b{ hdr, 0u };
		sa_z_header(W, b, false);
		s_hdr_bits = b.pos;
	}
	__syncthreads();
	SaZGroup &G = A.groups[g];
	for (int s = tid; s < 288; s += ZT)
		G.lcode[s] = s < SA_Z_NLIT ? W.lcode[s] : 0u;
	if (tid < 32)
		G.dcode[tid] = tid < SA_Z_NDIST ? W.dcode[tid] : 0u;
	for (int s = tid; s < ZHDR_WORDS; s += ZT)
		G.hdr[s] = hdr[s];
	if (tid == 0)
		G.hdr_bits = s_hdr_bits;
}

__global__ __launch_bounds__(ZT) void sa_k_deflate_encode(SaZArgs A)
{
	__shared__ uint32_t lcode[288], dcode[32], stage[ZSTAGE_WORDS];
	__shared__ uint32_t wave_sum[ZT / 64];
	__shared__ uint32_t s_tail_bytes;
	static_assert(ZE == 4, "a thread encodes one quad per round");

	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int seg = blockIdx.x, tile = blockIdx.y;
	uint32_t n;
	const uint4 *const seg4 = z_segment(A, seg, tile, &n);
	const SaZGroup &G = A.groups[(size_t)tile * (size_t)A.ngrp + (size_t)(seg / ZGROUP)];
	for (int s = tid; s < 288; s += ZT)
		lcode[s] = G.lcode[s];
	if (tid < 32)
		dcode[tid] = G.dcode[tid];
	for (int s = tid; s < ZSTAGE_WORDS; s += ZT)
		stage[s] = s < ZHDR_WORDS ? G.hdr[s] : 0u;
	uint32_t bitbase = __builtin_amdgcn_readfirstlane(G.hdr_bits), wbase = 0;
	__syncthreads();

	/* ---- rounds of ZT * ZE elements: bits, scan, ds_or into the stage, full words out ---- */
	uint32_t *const out = A.slots + ((size_t)tile * (size_t)A.nseg + (size_t)seg) * ZSLOT_WORDS;
	for (uint32_t r0 = 0; r0 < n; r0 += ZT * ZE) {
		uint64_t bits[ZE];
		uint32_t nb[ZE], mine = 0;
		{
			const uint32_t q = r0 / 4 + (uint32_t)tid;
			uint32_t el[12] = {};
			const bool live = q < n / 4;
			if (live)
				z_load_quad(seg4, q, el);
#pragma unroll
			for (int e = 0; e < ZE; e++) {
				nb[e] = 0;
				bits[e] = 0;
				if (live)
					nb[e] = sa_z_element(lcode, dcode, el[8 + e], z_match_quad(el, 4u * q + (uint32_t)e, e), &bits[e]);
				mine += nb[e];
			}
		}
		uint32_t incl = mine;
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
			if (lane >= d)
				incl += up;
		}
		if (lane == 63)
			wave_sum[wave] = incl;
		__syncthreads();
		uint32_t before = 0, total = 0;
#pragma unroll
		for (int w = 0; w < ZT / 64; w++) {
			const uint32_t ws = wave_sum[w];
			before += w < wave ? ws : 0u;
			total += ws;
		}
		uint32_t pos = bitbase + before + incl - mine;
#pragma unroll
		for (int e = 0; e < ZE; e++) {
			z_or_bits(stage, pos, bits[e], nb[e]);
			pos += nb[e];
		}
		__syncthreads();
		const uint32_t tot = bitbase + total, nfull = tot >> 5;
		for (uint32_t w = tid; w < nfull; w += ZT) {
			out[wbase + w] = stage[w];
			stage[w] = 0;
		}
		if (tid == 0 && nfull > 0) { /* the partly filled word moves to the front (word 0 is this thread's own to clear) */
			const uint32_t carry = stage[nfull];
			stage[nfull] = 0;
			stage[0] = carry;
		}
		__syncthreads();
		wbase += nfull;
		bitbase = tot & 31u;
	}
	/* ---- end of block, empty stored block; the rest of the stage out ---- */
	if (tid == 0) {
		SaZBits b{ stage, bitbase };
		s_tail_bytes = sa_z_finish_segment(lcode, b);
	}
	__syncthreads();
	const uint32_t tail_words = (s_tail_bytes + 3u) >> 2;
	for (uint32_t w = tid; w < tail_words; w += ZT)
		out[wbase + w] = stage[w];
	if (tid == 0)
		A.seg_bytes[(size_t)tile * (size_t)A.nseg + (size_t)seg] = wbase * 4u + s_tail_bytes;
}

/* where the segments of a tile go: [78 9c][segment 0]...[segment nseg-1][01 00 00 ff ff][Adler-32, big endian] */
__global__ __launch_bounds__(ZT) void sa_k_deflate_offsets(SaZArgs A)
{
	__shared__ uint32_t s_len[1024], s_s1[1024], s_s2[1024];
	const int tile = blockIdx.x, tid = threadIdx.x;
	const size_t base = (size_t)tile * (size_t)A.nseg;
	const uint32_t tile_elems = (uint32_t)A.chunk * (uint32_t)A.chunk;
	uint32_t at = 2, a = 1, b = 0; /* (a tile's stream stays far below 4 GB: <= 2 x 64 MB) */
	for (int s0 = 0; s0 < A.nseg; s0 += 1024) {
		const int cnt = A.nseg - s0 < 1024 ? A.nseg - s0 : 1024;
		__syncthreads();
		for (int s = tid; s < cnt; s += ZT) {
			s_len[s] = A.seg_bytes[base + s0 + s];
			s_s1[s] = A.seg_s1[base + s0 + s];
			s_s2[s] = A.seg_s2[base + s0 + s];
		}
		__syncthreads();
		if (tid == 0) {
			for (int s = 0; s < cnt; s++) {
				A.seg_off[base + s0 + s] = at;
				at += s_len[s];
				const uint32_t e0 = (uint32_t)(s0 + s) * ZSEG;
				const uint32_t n = tile_elems - e0 < (uint32_t)ZSEG ? tile_elems - e0 : (uint32_t)ZSEG;
				sa_z_adler_append(a, b, s_s1[s], s_s2[s], 4ull * n);
			}
		}
	}
	if (tid == 0) {
		A.tile_bytes[tile] = (unsigned long long)at + 9ull;
		A.tile_adler[tile] = b << 16 | a;
	}
}

__global__ void sa_k_deflate_rowbase(SaZArgs A, int ntiles)
{
	if (threadIdx.x == 0 && blockIdx.x == 0) {
		unsigned long long at = 0;
		for (int t = 0; t < ntiles; t++) {
			A.tile_base[t] = at;
			at += (A.tile_bytes[t] + 63ull) & ~63ull;
		}
		A.tile_base[ntiles] = at;
	}
}

__global__ __launch_bounds__(ZT) void sa_k_deflate_gather(SaZArgs A)
{
	const int seg = blockIdx.x, tile = blockIdx.y, tid = threadIdx.x;
	const size_t at = (size_t)tile * (size_t)A.nseg + (size_t)seg;
	const uint32_t *const srcw = A.slots + at * ZSLOT_WORDS;
	const uint8_t *const srcb = reinterpret_cast<const uint8_t *>(srcw);
	const uint32_t len = A.seg_bytes[at];
	uint8_t *const o = A.out + A.tile_base[tile];
	if (seg == 0 && tid == 0) {
		unsigned long long e = A.tile_bytes[tile] - 9ull;
		const uint32_t adler = A.tile_adler[tile];
		o[0] = 0x78;
		o[1] = 0x9c;
		o[e++] = 0x01;
		o[e++] = 0x00;
		o[e++] = 0x00;
		o[e++] = 0xff;
		o[e++] = 0xff;
		o[e++] = (uint8_t)(adler >> 24);
		o[e++] = (uint8_t)(adler >> 16);
		o[e++] = (uint8_t)(adler >> 8);
		o[e++] = (uint8_t)adler;
	}
	uint8_t *const dst = o + A.seg_off[at];
	uint32_t head = (uint32_t)((4u - (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 3u)) & 3u);
	if (head > len)
		head = len;
	if ((uint32_t)tid < head)
		dst[tid] = srcb[tid];
	const uint32_t nd = (len - head) >> 2, sh = 8u * (head & 3u);
	uint32_t *const dstw = reinterpret_cast<uint32_t *>(dst + head);
	for (uint32_t q = tid; q < nd; q += ZT) /* source byte head + 4 q: word q, shifted */
		dstw[q] = sh ? (srcw[q] >> sh) | (srcw[q + 1] << (32u - sh)) : srcw[q];
	const uint32_t done = head + 4u * nd;
	if ((uint32_t)tid < len - done)
		dst[done + tid] = srcb[done + tid];
}

/* The tiles of a row as they are -- full symmetric matrix, zero diagonal, zeros beyond N -- from the packed triangle, in
 * blocks of 64 x 64: one workgroup per block.  Below the diagonal a matrix row IS a run of the packed index (coalesced
 * along j); above it the mirrored element sits in column j's run, contiguous along i: the block is read along i and
 * turned in LDS, so both halves move whole cache lines (reading the upper half element by element along j fetched a
 * line for every four bytes: the encoder's two passes spent half their time there).  Level 0 returns these tiles; the
 * encoder reads them twice. */
__global__ __launch_bounds__(ZT) void sa_k_tiles_raw(SaZArgs A)
{
	__shared__ uint32_t turn[64][65];
	const int tile = blockIdx.z, tid = threadIdx.x;
	const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
	const int64_t i0 = (int64_t)(A.tile_r0 + tile * A.tile_dr) * A.chunk + r0, j0 = (int64_t)(A.tile_c0 + tile * A.tile_dc) * A.chunk + c0;
	uint32_t *const o = A.raw + (size_t)tile * (size_t)A.chunk * (size_t)A.chunk + (size_t)r0 * (size_t)A.chunk + (size_t)c0;
	const int x = tid & 63, y4 = tid >> 6;
	if (A.packed && j0 >= i0 + 64) { /* strictly above the diagonal: element (i, j) = packed[tri(j) + i] */
		for (int s = 0; s < 64; s += 4) {
			const int64_t j = j0 + s + y4, i = i0 + x;
			turn[s + y4][x] = i < A.num && j < A.num ? (uint32_t)A.packed[j * (j - 1) / 2 + i] : 0u;
		}
		__syncthreads();
		for (int s = 0; s < 64; s += 4)
			o[(size_t)(s + y4) * (size_t)A.chunk + x] = turn[x][s + y4];
		return;
	}
	for (int s = 0; s < 64; s += 4) /* on or below the diagonal, or a full matrix: along j */
		o[(size_t)(s + y4) * (size_t)A.chunk + x] = z_fetch(A, i0 + s + y4, j0 + x);
}

} // namespace

/* ---- host side ---------------------------------------------------------------------------------------------------- */
/* A BATCH is what one pass of the kernels encodes: up to nc tiles in a line -- a tile row (the walk over a finished matrix,
 * sa_zjob_tile_row), or, while the alignment is still running (sa_hip_tiles_begin / sa_zjob_next), the two arms of a SHELL:
 * the tiles whose larger tile index is b need exactly the columns of column block b, so they can be encoded and written while
 * the device aligns block b + 1:  batch 2 b - 1 = (b, 0) .. (b, b),  batch 2 b = (0, b) .. (b - 1, b)  (b = 0: the tile (0, 0)). */
struct SaZBatch {
	int32_t r0 = 0, c0 = 0, dr = 0, dc = 0, nt = 0;
	int32_t block = -1; /* the column block whose alignment it waits for, as an index into the job's own blocks (-1: none) */
};
/* Several devices: column block b -- its alignment and both arms of its shell -- belongs to device b mod n.  A shell needs
 * nothing but its own block, so the devices never exchange a byte: the pair space shards into independent units and every
 * device drives its own PCIe link; the caller still gets the shells in ascending order, from the devices in turn. */

struct sa_zjob {
	int device = 0;
	int32_t num = 0, chunk = 0, chunk_shift = 0, nc = 0, nseg = 0, ngrp = 0;
	bool stored = false; /* level 0: raw tiles */
	const int32_t *d_packed = nullptr, *d_full = nullptr;
	int32_t *d_owned = nullptr; /* the packed matrix, when the job made it (sa_hip_tiles_begin) */
	sa_ctx *ctx = nullptr;
	uint32_t *d_slots = nullptr, *d_seg = nullptr; /* d_seg: bytes, s1, s2, offset, [nc * nseg] each */
	uint32_t *d_ghist = nullptr, *d_tile_adler = nullptr;
	SaZGroup *d_groups = nullptr;
	/* what a batch leaves for its copy, twice (batch id uses [id & 1]): the next batch is encoded while this one's copy runs */
	uint8_t *d_out[2] = {};
	unsigned long long *d_info[2] = {}; /* tile_bytes[nc], tile_base[nc + 1] */
	uint32_t *d_raw = nullptr; /* the batch's raw tiles (level 0: d_out itself) */
	size_t tile_bound = 0, row_bound = 0;
	unsigned long long *h_info[2] = {};
	uint8_t *h_buf[2] = {};
	size_t h_cap[2] = {};
	size_t host_estimate = 0; /* bytes of the largest batch to come, as far as known */
	bool two_buffers = false;
	bool trace = false; /* SA_HIP_ZTRACE (sa_env.h) */
	/* ONE in-order stream carries the alignment's launches and the encoder's kernels -- the alignment's persistent workgroups
	 * fill every CU for the whole of a launch (all VGPRs at four waves per SIMD, or all LDS), so on a stream of its own the
	 * encoder only ever ran when a launch drained, a column block late; in order it runs between two launches, alone, for a few
	 * milliseconds.  The copies have their own stream. */
	hipStream_t stream = nullptr, copy_stream = nullptr;
	hipEvent_t enc_done[2] = {};
	int64_t enc_enqueued = -1;  /* batches [0, enc_enqueued] have their kernels on the stream   */
	int64_t copy_enqueued = -1; /* the batch whose copy is on the copy stream (-1: none)        */
	size_t copy_bytes = 0;      /* bytes of it that the enqueued copy covers                    */
	/* shells: the alignment runs block by block, one to two blocks ahead of the encoder */
	bool shells = false;
	std::vector<int32_t> own_blocks;   /* the column blocks of this job (this device), ascending */
	std::vector<SaZBatch> batches;     /* their arms, in the order they are handed out            */
	std::vector<hipEvent_t> block_start, block_done; /* [own block] */
	int32_t blocks_launched = 0;
	int64_t next_batch = 0;
	/* the job the caller holds, with several devices: the other devices' jobs and whose turn it is */
	std::vector<sa_zjob *> peers;
	std::vector<std::pair<int32_t, int64_t>> order; /* (0: this job, k: peers[k - 1]; batch of that job) */
	size_t order_next = 0;
	double encode_ms = 0, copy_ms = 0;
	uint64_t raw_bytes = 0, out_bytes = 0, late_bytes = 0;
};

static void zjob_free(sa_zjob *z)
{
	if (!z)
		return;
	for (sa_zjob *p : z->peers)
		zjob_free(p);
	z->peers.clear();
	(void)hipSetDevice(z->device);
	for (hipStream_t st : { z->copy_stream, z->stream })
		if (st) {
			(void)hipStreamSynchronize(st);
			(void)hipStreamDestroy(st);
		}
	for (hipEvent_t e : z->block_start)
		(void)hipEventDestroy(e);
	for (hipEvent_t e : z->block_done)
		(void)hipEventDestroy(e);
	(void)hipFree(z->d_slots);
	(void)hipFree(z->d_seg);
	(void)hipFree(z->d_ghist);
	(void)hipFree(z->d_tile_adler);
	(void)hipFree(z->d_groups);
	if (!z->stored)
		(void)hipFree(z->d_raw);
	(void)hipFree(z->d_owned);
	for (int k = 0; k < 2; k++) {
		(void)hipFree(z->d_out[k]);
		(void)hipFree(z->d_info[k]);
		if (z->enc_done[k])
			(void)hipEventDestroy(z->enc_done[k]);
		if (z->h_info[k])
			(void)hipHostFree(z->h_info[k]);
		if (z->h_buf[k])
			(void)hipHostFree(z->h_buf[k]);
	}
	if (z->ctx)
		sa_ctx_destroy(z->ctx);
	delete z;
}

static SaZBatch zjob_batch(const sa_zjob *z, int64_t id)
{
	if (z->shells)
		return z->batches[(size_t)id];
	SaZBatch b; /* tile row `id` */
	b.r0 = (int32_t)id, b.dc = 1, b.nt = z->nc;
	return b;
}
static int64_t zjob_batches(const sa_zjob *z) { return z->shells ? (int64_t)z->batches.size() : (int64_t)z->nc; }
/* the arms of the job's own column blocks: (b, 0) .. (b, b), then (0, b) .. (b - 1, b) */
static void zjob_plan_shells(sa_zjob *z)
{
	z->batches.clear();
	for (size_t k = 0; k < z->own_blocks.size(); k++) {
		const int32_t blk = z->own_blocks[k];
		SaZBatch row;
		row.block = (int32_t)k, row.r0 = blk, row.dc = 1, row.nt = blk + 1;
		z->batches.push_back(row);
		if (blk > 0) {
			SaZBatch col;
			col.block = (int32_t)k, col.c0 = blk, col.dr = 1, col.nt = blk;
			z->batches.push_back(col);
		}
	}
}

static SaZArgs zjob_args(const sa_zjob *z, const SaZBatch &b, int par)
{
	const size_t segs = (size_t)z->nc * (size_t)z->nseg;
	SaZArgs a{};
	a.packed = z->d_packed;
	a.full = z->d_full;
	a.num = z->num;
	a.chunk = z->chunk;
	a.chunk_shift = z->chunk_shift;
	a.tile_r0 = b.r0, a.tile_c0 = b.c0, a.tile_dr = b.dr, a.tile_dc = b.dc;
	a.nseg = z->nseg;
	a.ngrp = z->ngrp;
	a.slots = z->d_slots;
	a.seg_bytes = z->d_seg;
	a.seg_s1 = z->d_seg ? z->d_seg + segs : nullptr;
	a.seg_s2 = z->d_seg ? z->d_seg + 2 * segs : nullptr;
	a.seg_off = z->d_seg ? z->d_seg + 3 * segs : nullptr;
	a.ghist = z->d_ghist;
	a.groups = z->d_groups;
	a.tile_bytes = z->d_info[par];
	a.tile_base = z->d_info[par] ? z->d_info[par] + z->nc : nullptr;
	a.tile_adler = z->d_tile_adler;
	a.out = z->d_out[par];
	a.raw = z->stored ? reinterpret_cast<uint32_t *>(z->d_out[par]) : z->d_raw;
	return a;
}

static bool zjob_host_buffer(sa_zjob *z, int which, size_t bytes)
{
	if (bytes <= z->h_cap[which])
		return true;
	if (z->h_buf[which])
		(void)hipHostFree(z->h_buf[which]);
	z->h_buf[which] = nullptr;
	z->h_cap[which] = 0;
	/* Page-locking is the expensive part (~0.1-0.2 ms per MB): a buffer is made once, for the largest batch to come -- nc tiles at
	 * the bytes per tile seen so far plus a tenth (z->host_estimate; batches grow along a walk in shells: growing the buffer with
	 * them locked 10 GB in all for config 5's 0.5 GB batches, 2 s of system time) -- and only a surprise makes it again. */
	const size_t want = std::min(z->row_bound, std::max(bytes + bytes / 16 + (1 << 20), z->host_estimate));
	SA_HIP_CHECK(hipHostMalloc(&z->h_buf[which], want, hipHostMallocDefault), return false);
	z->h_cap[which] = want;
	return true;
}

/* the alignment of column block `blk` (columns [blk * chunk, (blk + 1) * chunk)): one packed range, onto the stream */
static bool zjob_align_block(sa_zjob *z, int32_t own)
{
	const int32_t blk = z->own_blocks[(size_t)own];
	const int64_t ja = (int64_t)blk * z->chunk, jb = std::min<int64_t>((int64_t)z->num, ja + z->chunk);
	const int64_t start = ja * (ja - 1) / 2, end = jb * (jb - 1) / 2;
	const auto t0 = std::chrono::steady_clock::now();
	SA_HIP_CHECK(hipEventRecord(z->block_start[(size_t)own], z->stream), return false);
	if (end > start && sa_ctx_align_range(z->ctx, start, end - start, z->d_owned + start, z->stream) != 0)
		return false;
	SA_HIP_CHECK(hipEventRecord(z->block_done[(size_t)own], z->stream), return false);
	if (z->trace)
		fprintf(stderr, "[zjob] block %d: %lld pairs launched in %.2f ms of host time\n", blk, (long long)(end - start), sa_ms_since(t0));
	return true;
}
static bool zjob_align_upto(sa_zjob *z, int32_t own)
{
	while (z->blocks_launched < (int32_t)z->own_blocks.size() && z->blocks_launched <= own) {
		if (!zjob_align_block(z, z->blocks_launched))
			return false;
		z->blocks_launched++;
	}
	return true;
}

/* the kernels of batch `id` onto the stream (raw tiles, encoding, sizes into d_info[id & 1], streams into d_out[id & 1]); the
 * caller has seen batch id - 2 arrive, so both are free */
static bool zjob_enqueue_encode(sa_zjob *z, int64_t id)
{
	const SaZBatch b = zjob_batch(z, id);
	const int par = (int)(id & 1);
	if (b.block >= 0 && !zjob_align_upto(z, b.block)) /* (in order: the batch's column block is on the stream before it) */
		return false;
	const SaZArgs a = zjob_args(z, b, par);
	const unsigned nt = (unsigned)b.nt;
	const dim3 per_seg((unsigned)z->nseg, nt), per_grp((unsigned)z->ngrp, nt);
	hipLaunchKernelGGL(sa_k_tiles_raw, dim3((unsigned)z->chunk / 64, (unsigned)z->chunk / 64, nt), dim3(ZT), 0, z->stream, a);
	SA_HIP_CHECK(hipGetLastError(), return false);
	if (!z->stored) {
		SA_HIP_CHECK(hipMemsetAsync(z->d_ghist, 0, sizeof(uint32_t) * (size_t)nt * (size_t)z->ngrp * ZHIST, z->stream), return false);
		hipLaunchKernelGGL(sa_k_deflate_hist, per_seg, dim3(ZT), 0, z->stream, a);
		SA_HIP_CHECK(hipGetLastError(), return false);
		hipLaunchKernelGGL(sa_k_deflate_codes, per_grp, dim3(ZT), 0, z->stream, a);
		SA_HIP_CHECK(hipGetLastError(), return false);
		hipLaunchKernelGGL(sa_k_deflate_encode, per_seg, dim3(ZT), 0, z->stream, a);
		SA_HIP_CHECK(hipGetLastError(), return false);
		hipLaunchKernelGGL(sa_k_deflate_offsets, dim3(nt), dim3(ZT), 0, z->stream, a);
		SA_HIP_CHECK(hipGetLastError(), return false);
		hipLaunchKernelGGL(sa_k_deflate_rowbase, dim3(1), dim3(64), 0, z->stream, a, (int)nt);
		SA_HIP_CHECK(hipGetLastError(), return false);
		hipLaunchKernelGGL(sa_k_deflate_gather, per_seg, dim3(ZT), 0, z->stream, a);
		SA_HIP_CHECK(hipGetLastError(), return false);
	}
	SA_HIP_CHECK(hipEventRecord(z->enc_done[par], z->stream), return false);
	z->enc_enqueued = id;
	return true;
}

/* the copy of batch `id` onto the copy stream: sizes, and the first `bytes` of its compact buffer into host buffer `which` */
static bool zjob_enqueue_copy(sa_zjob *z, int64_t id, int which, size_t bytes)
{
	const int par = (int)(id & 1);
	SA_HIP_CHECK(hipStreamWaitEvent(z->copy_stream, z->enc_done[par], 0), return false);
	if (!z->stored) {
		SA_HIP_CHECK(hipMemcpyAsync(z->h_info[par], z->d_info[par], sizeof(unsigned long long) * (size_t)(2 * z->nc + 1), hipMemcpyDeviceToHost,
					    z->copy_stream),
			     return false);
	}
	if (bytes) {
		SA_HIP_CHECK(hipMemcpyAsync(z->h_buf[which], z->d_out[par], bytes, hipMemcpyDeviceToHost, z->copy_stream), return false);
	}
	z->copy_enqueued = id;
	z->copy_bytes = bytes;
	return true;
}

static sa_zjob *zjob_make(int device, const int32_t *d_packed, const int32_t *d_full, int32_t num, size_t chunk_dim, bool stored)
{
	if (num < 2 || (!d_packed && !d_full)) {
		sa_set_error("sa_zjob: no matrix");
		return nullptr;
	}
	int shift = 0;
	while (((size_t)1 << shift) < chunk_dim)
		shift++;
	if (((size_t)1 << shift) != chunk_dim || chunk_dim < 64 || chunk_dim > 4096) {
		/* (src/io/format/hdf5.c:70-89 only ever produces 256 .. 4096, a power of two) */
		sa_set_error("sa_zjob: the chunk dimension must be a power of two in [64, 4096], got %zu", chunk_dim);
		return nullptr;
	}
	if (!sa_device_ready(device))
		return nullptr;
	sa_zjob *z = new sa_zjob;
	z->device = device;
	z->trace = sa_env_read().ztrace;
	z->num = num;
	z->chunk = (int32_t)chunk_dim;
	z->chunk_shift = shift;
	z->nc = (int32_t)(((size_t)num + chunk_dim - 1) / chunk_dim);
	z->nseg = (int32_t)((chunk_dim * chunk_dim + ZSEG - 1) / ZSEG);
	z->ngrp = (z->nseg + ZGROUP - 1) / ZGROUP;
	z->d_packed = d_packed;
	z->d_full = d_packed ? nullptr : d_full;
	z->stored = stored;
	const size_t segs = (size_t)z->nc * (size_t)z->nseg;
	/* a tile's stream at its very worst (header, 63 bits per element, the segments' ends), on a 64-byte boundary */
	z->tile_bound = stored ? chunk_dim * chunk_dim * sizeof(int32_t) : (((size_t)z->nseg * ZSLOT_WORDS * 4 + 64) + 255) & ~(size_t)255;
	z->row_bound = (size_t)z->nc * z->tile_bound;
	/* a second host buffer lets batch k + 1 be copied while the caller writes batch k; page-locking it costs ~0.1-0.2 ms per MB,
	 * the copies it hides ~0.02 ms per MB and row: worth it from eight tile rows on -- for streams.  Raw tiles are three
	 * times the bytes per row: config 4 (13 rows of 872 MB) measured 1.50 s with one buffer, 1.83 s with two. */
	z->two_buffers = !stored && z->nc >= 8;
	if (stored)
		z->host_estimate = z->row_bound;
	bool ok = false;
	do {
		SA_HIP_CHECK(hipStreamCreateWithFlags(&z->stream, hipStreamNonBlocking), break);
		SA_HIP_CHECK(hipStreamCreateWithFlags(&z->copy_stream, hipStreamNonBlocking), break);
		bool both = true;
		for (int k = 0; k < 2 && both; k++) {
			both = false;
			SA_HIP_CHECK(hipEventCreateWithFlags(&z->enc_done[k], hipEventDisableTiming), break);
			SA_HIP_CHECK(hipMalloc(&z->d_out[k], z->row_bound), break);
			if (!stored) {
				SA_HIP_CHECK(hipMalloc(&z->d_info[k], sizeof(unsigned long long) * (size_t)(2 * z->nc + 1)), break);
				SA_HIP_CHECK(hipHostMalloc(&z->h_info[k], sizeof(unsigned long long) * (size_t)(2 * z->nc + 1), hipHostMallocDefault), break);
			}
			both = true;
		}
		if (!both)
			break;
		if (stored) {
			ok = true;
			break;
		}
		SA_HIP_CHECK(hipMalloc(&z->d_raw, (size_t)z->nc * chunk_dim * chunk_dim * sizeof(uint32_t)), break);
		SA_HIP_CHECK(hipMalloc(&z->d_slots, segs * ZSLOT_WORDS * sizeof(uint32_t)), break);
		SA_HIP_CHECK(hipMalloc(&z->d_seg, 4 * segs * sizeof(uint32_t)), break);
		SA_HIP_CHECK(hipMalloc(&z->d_ghist, sizeof(uint32_t) * (size_t)z->nc * (size_t)z->ngrp * ZHIST), break);
		SA_HIP_CHECK(hipMalloc(&z->d_groups, sizeof(SaZGroup) * (size_t)z->nc * (size_t)z->ngrp), break);
		SA_HIP_CHECK(hipMalloc(&z->d_tile_adler, sizeof(uint32_t) * (size_t)z->nc), break);
		ok = true;
	} while (0);
	if (!ok) {
		zjob_free(z);
		return nullptr;
	}
	return z;
}

/* The tiles of batch `id` into page-locked memory: tile t at streams[t], sizes[t] bytes.  Returns their number, < 0 on error.
 * Before it returns, the kernels of the batches that follow -- in shells: both arms of the next column block, then the
 * alignment of the block after it -- and (two host buffers) the next batch's copy are enqueued: they run while the caller
 * writes this batch. */
static int zjob_fetch(sa_zjob *z, int64_t id, const uint8_t **streams, size_t *sizes)
{
	SA_HIP_CHECK(hipSetDevice(z->device), return -1);
	const SaZBatch b = zjob_batch(z, id);
	const int par = (int)(id & 1), which = z->two_buffers ? par : 0;
	const size_t raw_row = (size_t)b.nt * z->tile_bound;
	const auto t0 = std::chrono::steady_clock::now();
	if (id > z->enc_enqueued) { /* (the first batch; batches are handed out in order) */
		if (id != z->enc_enqueued + 1) {
			sa_set_error("sa_zjob: batches are handed out in order");
			return -1;
		}
		if (!zjob_enqueue_encode(z, id))
			return -1;
	} else if (id < z->enc_enqueued - 1) { /* (its buffers belong to batch id + 2 by now) */
		sa_set_error("sa_zjob: batch %lld is gone (batches are handed out in order, once)", (long long)id);
		return -1;
	}
	if (z->copy_enqueued != id) {
		size_t bytes = 0;
		if (z->stored) {
			bytes = raw_row;
			if (!zjob_host_buffer(z, which, bytes))
				return -1;
		}
		if (!zjob_enqueue_copy(z, id, which, bytes))
			return -1;
	}
	SA_HIP_CHECK(hipStreamSynchronize(z->copy_stream), return -1);
	z->encode_ms += sa_ms_since(t0);
	const auto t1 = std::chrono::steady_clock::now();
	size_t total = raw_row;
	if (!z->stored) {
		const unsigned long long *info = z->h_info[par];
		total = (size_t)info[z->nc + b.nt];
		if (total > z->row_bound) {
			sa_set_error("sa_zjob: a batch of tiles outgrew its bound (%zu > %zu)", total, z->row_bound);
			return -1;
		}
		if (!z->host_estimate)
			z->host_estimate = (size_t)((double)total / (double)b.nt * (double)z->nc * 1.1) + (1 << 20);
		if (total > z->copy_bytes) { /* what the enqueued copy did not cover (the first batch: everything) */
			size_t have = z->copy_bytes;
			if (total > z->h_cap[which]) { /* (a new buffer: the part already copied is copied again) */
				if (!zjob_host_buffer(z, which, total))
					return -1;
				have = 0;
			}
			SA_HIP_CHECK(hipMemcpyAsync(z->h_buf[which] + have, z->d_out[par] + have, total - have, hipMemcpyDeviceToHost, z->copy_stream),
				     return -1);
			SA_HIP_CHECK(hipStreamSynchronize(z->copy_stream), return -1);
			z->late_bytes += total - have;
		}
		for (int t = 0; t < b.nt; t++) {
			sizes[t] = (size_t)info[t];
			streams[t] = z->h_buf[which] + (size_t)info[z->nc + t];
			z->out_bytes += sizes[t];
		}
	} else {
		for (int t = 0; t < b.nt; t++) {
			sizes[t] = z->tile_bound;
			streams[t] = z->h_buf[which] + (size_t)t * z->tile_bound;
			z->out_bytes += sizes[t];
		}
	}
	z->raw_bytes += (uint64_t)b.nt * (uint64_t)z->chunk * (uint64_t)z->chunk * 4u;
	z->copy_ms += sa_ms_since(t1);
	z->copy_enqueued = -1;
	if (z->trace)
		fprintf(stderr, "[zjob] batch %lld (%d tiles, block %d): waited %.2f ms, copy %.2f ms, %zu bytes\n", (long long)id, b.nt, b.block,
			std::chrono::duration<double, std::milli>(t1 - t0).count(), sa_ms_since(t1), total);
	/* ---- what runs while the caller writes this batch ---- */
	const int64_t last = zjob_batches(z) - 1;
	if (id < last) {
		if (z->enc_enqueued == id) { /* nothing ahead: the kernels of the next batch -- in shells of the next BLOCK: both arms -- and
					       * then the alignment of one more block, so that the stream never runs dry while the host writes */
			if (!zjob_enqueue_encode(z, id + 1))
				return -1;
			if (z->shells) {
				const SaZBatch nb = zjob_batch(z, id + 1);
				if (id + 2 <= last && zjob_batch(z, id + 2).block == nb.block && !zjob_enqueue_encode(z, id + 2))
					return -1;
				if (!zjob_align_upto(z, nb.block + 1))
					return -1;
			}
		}
		if (z->two_buffers) { /* (one host buffer: a copy would overwrite what the caller is about to read) */
			const SaZBatch nb = zjob_batch(z, id + 1);
			const int next = (int)((id + 1) & 1);
			/* (the length is not known yet: this batch's bytes per tile, plus 3 %) */
			const size_t guess = (size_t)((double)total / (double)b.nt * (double)nb.nt * 1.03) + (1 << 16);
			const size_t bytes = std::min((size_t)nb.nt * z->tile_bound, guess);
			if (!zjob_host_buffer(z, next, bytes) || !zjob_enqueue_copy(z, id + 1, next, bytes))
				return -1;
		}
	}
	return b.nt;
}

extern "C" sa_zjob *sa_zjob_create(int device, const int32_t *d_packed, const int32_t *d_full, int32_t num, size_t chunk_dim, int level)
{
	return sa_guard("sa_zjob_create", (sa_zjob *)nullptr, [&] { return zjob_make(device, d_packed, d_full, num, chunk_dim, level == 0); });
}

extern "C" void sa_zjob_destroy(sa_zjob *job)
{
	sa_guard_void("sa_zjob_destroy", [&] { zjob_free(job); });
}

extern "C" size_t sa_zjob_tiles_per_row(const sa_zjob *job) { return job ? (size_t)job->nc : 0; }

extern "C" int sa_zjob_tile_row(sa_zjob *z, size_t tile_row, const uint8_t **streams, size_t *sizes)
{
	return sa_guard("sa_zjob_tile_row", 1, [&]() -> int {
		if (!z || !streams || !sizes || tile_row >= (size_t)z->nc || z->shells) {
			sa_set_error("sa_zjob_tile_row: bad arguments");
			return 1;
		}
		return zjob_fetch(z, (int64_t)tile_row, streams, sizes) < 0 ? 1 : 0;
	});
}

extern "C" int sa_zjob_next(sa_zjob *z, uint32_t *rows, uint32_t *cols, const uint8_t **streams, size_t *sizes)
{
	return sa_guard("sa_zjob_next", -1, [&]() -> int {
		if (!z || !rows || !cols || !streams || !sizes) {
			sa_set_error("sa_zjob_next: bad arguments");
			return -1;
		}
		sa_zjob *job = z;
		int64_t id = z->next_batch;
		if (!z->order.empty()) { /* several devices: the shells in ascending order, from the devices in turn */
			if (z->order_next >= z->order.size())
				return 0;
			const auto &turn = z->order[z->order_next];
			job = turn.first == 0 ? z : z->peers[(size_t)turn.first - 1];
			id = turn.second;
		} else if (id >= zjob_batches(z)) {
			return 0;
		}
		const SaZBatch b = zjob_batch(job, id);
		const int n = zjob_fetch(job, id, streams, sizes);
		if (n < 0)
			return -1;
		for (int t = 0; t < n; t++) {
			rows[t] = (uint32_t)(b.r0 + t * b.dr);
			cols[t] = (uint32_t)(b.c0 + t * b.dc);
		}
		if (!z->order.empty())
			z->order_next++;
		else
			z->next_batch++;
		return n;
	});
}

extern "C" void sa_zjob_stats(const sa_zjob *z, double *encode_ms, double *copy_ms, uint64_t *raw_bytes, uint64_t *out_bytes)
{
	if (!z)
		return;
	double e = z->encode_ms, c = z->copy_ms;
	uint64_t r = z->raw_bytes, o = z->out_bytes;
	for (const sa_zjob *p : z->peers)
		e += p->encode_ms, c += p->copy_ms, r += p->raw_bytes, o += p->out_bytes;
	if (encode_ms)
		*encode_ms = e;
	if (copy_ms)
		*copy_ms = c;
	if (raw_bytes)
		*raw_bytes = r;
	if (out_bytes)
		*out_bytes = o;
}

static double zjob_align_ms(const sa_zjob *z)
{
	double ms = 0.0;
	for (int32_t k = 0; k < z->blocks_launched; k++) {
		float one = 0.f;
		if (hipEventElapsedTime(&one, z->block_start[(size_t)k], z->block_done[(size_t)k]) == hipSuccess)
			ms += one;
	}
	(void)hipGetLastError();
	return ms;
}

/* seconds the device spent aligning so far: the sum over the column blocks whose kernels have finished; several devices
 * align side by side: the longest of their sums */
extern "C" double sa_zjob_align_seconds(const sa_zjob *z)
{
	return sa_guard("sa_zjob_align_seconds", 0.0, [&]() -> double {
		if (!z || !z->shells)
			return 0.0;
		(void)hipSetDevice(z->device);
		double ms = zjob_align_ms(z);
		for (const sa_zjob *p : z->peers) {
			(void)hipSetDevice(p->device);
			ms = std::max(ms, zjob_align_ms(p));
		}
		return ms * 1e-3;
	});
}

/* one device's part of a walk in shells: context, the packed matrix (whole: a block's place in it is its own), a job over
 * the column blocks `first`, `first + step`, ...; the first two of them are on their way when this returns */
static sa_zjob *zjob_begin_on(int device, struct sa_input in, const struct sa_scoring *sc, size_t chunk_dim, bool stored, int first, int step)
{
	sa_ctx *ctx = sa_ctx_create(device, in, sc);
	if (!ctx)
		return nullptr;
	int32_t *d_packed = nullptr;
	sa_zjob *z = nullptr;
	bool ok = false;
	do {
		const int64_t pairs = sa_ctx_pairs(ctx);
		SA_HIP_CHECK(hipSetDevice(device), break);
		SA_HIP_CHECK(hipMalloc(&d_packed, sizeof(int32_t) * (size_t)pairs), break);
		z = zjob_make(device, d_packed, nullptr, in.num, chunk_dim, stored);
		if (!z)
			break;
		z->shells = true;
		for (int32_t b = first; b < z->nc; b += step)
			z->own_blocks.push_back(b);
		zjob_plan_shells(z);
		z->block_start.assign(z->own_blocks.size(), nullptr);
		z->block_done.assign(z->own_blocks.size(), nullptr);
		bool ev = true;
		for (size_t k = 0; k < z->own_blocks.size() && ev; k++)
			ev = hipEventCreate(&z->block_start[k]) == hipSuccess && hipEventCreate(&z->block_done[k]) == hipSuccess;
		if (!ev) {
			sa_set_error("sa_hip_tiles_begin: hipEventCreate failed");
			break;
		}
		ok = true;
	} while (0);
	if (z) { /* (the job owns context and matrix from here on, whatever happens) */
		z->d_owned = d_packed;
		z->ctx = ctx;
	}
	/* the alignment starts now: the first blocks are small, the caller's file set-up runs beside them */
	if (ok)
		ok = zjob_align_upto(z, 1);
	if (!ok) {
		if (z)
			zjob_free(z);
		else {
			(void)hipFree(d_packed);
			sa_ctx_destroy(ctx);
		}
		return nullptr;
	}
	return z;
}

/* Context(s) + the packed matrix in device memory + a job that walks it in shells while the alignment runs (see SaZBatch).
 * Every device in use (all visible ones, or the first SA_HIP_DEVICES) takes the column blocks b = device (mod devices);
 * SA_HIP_TILES_SPLIT=n (testing) makes n such jobs share device 0. */
extern "C" sa_zjob *sa_hip_tiles_begin(struct sa_input in, const struct sa_scoring *sc, size_t chunk_dim, int level)
{
	return sa_guard("sa_hip_tiles_begin", (sa_zjob *)nullptr, [&]() -> sa_zjob * {
		const SaEnv env = sa_env_read();
		int parts = sa_devices_in_use(env);
		if (parts <= 0) {
			sa_set_error("No HIP devices available; libseqalign_hip has no CPU fallback");
			return nullptr;
		}
		const bool folded = env.tiles_split > 0;
		if (folded)
			parts = env.tiles_split;
		const int nc = chunk_dim ? (int)(((size_t)in.num + chunk_dim - 1) / chunk_dim) : 1;
		parts = std::max(1, std::min(parts, nc)); /* (a device without a column block would have nothing to do) */
		sa_zjob *z = zjob_begin_on(0, in, sc, chunk_dim, level == 0, 0, parts);
		if (!z)
			return nullptr;
		for (int k = 1; k < parts; k++) {
			sa_zjob *p = zjob_begin_on(folded ? 0 : k, in, sc, chunk_dim, level == 0, k, parts);
			if (!p) {
				zjob_free(z);
				return nullptr;
			}
			z->peers.push_back(p);
		}
		if (parts > 1) { /* whose turn it is: block b belongs to job b mod parts, its arms follow each other */
			std::vector<int64_t> at((size_t)parts, 0);
			for (int32_t b = 0; b < z->nc; b++)
				for (int arm = 0; arm < (b > 0 ? 2 : 1); arm++)
					z->order.emplace_back(b % parts, at[(size_t)(b % parts)]++);
		}
		return z;
	});
}
