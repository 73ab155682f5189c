/*
 * sa_deflate.hip -- the `-z` option on the device: the tiles (HDF5 chunks) of /similarity_matrix leave the GPU as finished
 * zlib streams, built from the packed scores where they were computed.
 *
 * Reference: src/io/format/hdf5.c:91-95 sets H5Pset_deflate on the chunked dataset and H5Dwrite (:148-194) runs libhdf5's
 * deflate filter over every chunk in the one writing thread.  For BASELINE config 5 (89 994 sequences after `-f 0.9`,
 * 484 chunks of 4096 x 4096 int32 = 32.4 GB, level 6) that is 41 CPU-minutes behind a 2-second alignment
 * (profiles/r04_cli_cfg5_full_size_end_to_end.txt: Output 157 s with every core of the box deflating).  DEFLATE over int32
 * scores is HBM-bound byte work with a fixed parse (sa_deflate_core.h): it belongs where the scores are.
 *
 *   sa_k_deflate_segments   one workgroup per SEGMENT (64 KB of a tile: ZSEG elements in the tile's row-major order).
 *                           Elements come straight from the packed triangle (lower part: contiguous; upper part: the
 *                           mirrored element; diagonal and the padding beyond N: 0) into LDS; match choice + histograms
 *                           (LDS atomics) + Adler sums; 512-key bitonic sort; one thread builds the three Huffman codes
 *                           and writes the block header (sa_deflate_core.h: the code the host harness tests); then rounds
 *                           of 256 x ZE elements: bits per thread, workgroup scan, ds_or into an LDS bit stage, full
 *                           words flushed to the segment's slot.
 *   sa_k_deflate_offsets    one workgroup per tile: where each segment goes in the tile's stream, header, final block,
 *                           Adler-32 of the tile from the segments' sums.
 *   sa_k_deflate_gather     one workgroup per segment: its bytes to their place (dword copies with a funnel shift
 *                           between the byte alignment of slot and stream).
 * Host: a job walks the tile rows; row r + 1 is encoded while the host hands row r to H5Dwrite_chunk.
 */
#include <algorithm>
#include <chrono>
#include <vector>

#include "sa_ctx.h"
#include "sa_deflate_core.h"

namespace {

constexpr int ZT = 256;     /* threads of a workgroup                                  */
constexpr int ZE = 2;       /* elements per thread and round                           */
constexpr int ZSEG = 16384; /* elements per segment: 64 KB of the tile                 */
/* the LDS bit stage holds the block header (<= 3 + 14 + 19 * 3 + 316 * 14 bits) and one round */
constexpr int ZSTAGE_WORDS = (4600 + ZT * ZE * SA_Z_ELEM_BITS + 31) / 32 + 8;
/* a segment's slot: header + 63 bits per element at the very worst, in words, a multiple of four */
constexpr int ZSLOT_WORDS = 2 * ZSEG + 256;
constexpr size_t ZLDS_BYTES = sizeof(uint32_t) * ZSEG + sizeof(SaZWork) + sizeof(uint32_t) * ZSTAGE_WORDS + 64;

struct SaZArgs {
	const int32_t *packed; /* scores by packed pair index j (j - 1) / 2 + i, i < j (src/io/output.c:76-83)     */
	const int32_t *full;   /* ... or the full num x num matrix (packed == nullptr)                              */
	int32_t num, chunk, chunk_shift;
	int32_t tile_row, tile_col0;
	int32_t nseg;          /* segments per tile                                                                 */
	uint32_t *slots;       /* [tile][segment][ZSLOT_WORDS]                                                      */
	uint32_t *seg_bytes, *seg_s1, *seg_s2; /* [tile][segment]                                                  */
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

__global__ __launch_bounds__(ZT) void sa_k_deflate_segments(SaZArgs A)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t z_lds[];
	uint32_t *const el = reinterpret_cast<uint32_t *>(z_lds);
	SaZWork &W = *reinterpret_cast<SaZWork *>(z_lds + sizeof(uint32_t) * ZSEG);
	uint32_t *const stage = reinterpret_cast<uint32_t *>(z_lds + sizeof(uint32_t) * ZSEG + sizeof(SaZWork));
	__shared__ uint32_t wave_sum[ZT / 64];

	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int seg = blockIdx.x, tile = blockIdx.y;
	const int64_t i0 = (int64_t)A.tile_row * A.chunk, j0 = (int64_t)(A.tile_col0 + tile) * A.chunk;
	const uint32_t tile_elems = (uint32_t)A.chunk * (uint32_t)A.chunk;
	const uint32_t e0 = (uint32_t)seg * ZSEG;
	const uint32_t n = tile_elems - e0 < (uint32_t)ZSEG ? tile_elems - e0 : (uint32_t)ZSEG;

	/* ---- the segment's elements, histograms cleared ---- */
	for (uint32_t k = tid; k < n; k += ZT) {
		const uint32_t e = e0 + k;
		el[k] = z_fetch(A, i0 + (e >> A.chunk_shift), j0 + (e & ((uint32_t)A.chunk - 1u)));
	}
	for (int s = tid; s < 288; s += ZT)
		W.lfreq[s] = 0;
	if (tid < 32)
		W.dfreq[tid] = 0;
	for (int s = tid; s < ZSTAGE_WORDS; s += ZT)
		stage[s] = 0;
	if (tid == 0)
		W.s1 = W.s2 = 0ull;
	__syncthreads();

	/* ---- match choice, histograms, Adler sums ---- */
	{
		unsigned long long s1 = 0, s2 = 0;
		const unsigned long long len = 4ull * n;
		for (uint32_t k0 = 0; k0 < n; k0 += ZT) {
			const uint32_t k = k0 + tid;
			int j = -1;
			uint32_t v = 0;
			if (k < n) {
				v = el[k];
				j = sa_z_match(el, k);
				atomicAdd(&W.lfreq[v & 255u], 1u);
				if (!j) {
					atomicAdd(&W.lfreq[(v >> 8) & 255u], 1u);
					atomicAdd(&W.lfreq[(v >> 16) & 255u], 1u);
					atomicAdd(&W.lfreq[v >> 24], 1u);
				} else if (j > 1) {
					atomicAdd(&W.dfreq[sa_z_dcode(j)], 1u);
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
					atomicAdd(&W.lfreq[SA_Z_LEN3], (uint32_t)__popcll(m_any));
				if (m_one)
					atomicAdd(&W.dfreq[sa_z_dcode(1)], (uint32_t)__popcll(m_one));
			}
		}
		for (int d = 32; d > 0; d >>= 1) {
			s1 += ((unsigned long long)(uint32_t)__shfl_down((int)(s1 >> 32), d) << 32) | (uint32_t)__shfl_down((int)(uint32_t)s1, d);
			s2 += ((unsigned long long)(uint32_t)__shfl_down((int)(s2 >> 32), d) << 32) | (uint32_t)__shfl_down((int)(uint32_t)s2, d);
		}
		if (lane == 0) {
			atomicAdd(&W.s1, s1);
			atomicAdd(&W.s2, s2);
		}
	}
	__syncthreads();
	if (tid == 0)
		W.lfreq[SA_Z_EOB] = 1;

	/* ---- literal alphabet by ascending weight: 512 keys (weight << 9 | symbol), bitonic, in the tree's storage ---- */
	uint32_t *const keys = W.w;
	static_assert(sizeof(W.w) >= 512 * sizeof(uint32_t), "the sort keys fit the tree storage");
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
	__shared__ uint32_t s_hdr_bits;
	if (tid == 0) {
		sa_z_alphabet(W, W.lfreq, SA_Z_NLIT, 15, W.llen, W.lcode, (int)W.used, true);
		sa_z_alphabet(W, W.dfreq, SA_Z_NDIST, 15, W.dlen, W.dcode, 0, false);
		SaZBits b{ stage, 0u };
		sa_z_header(W, b, false);
		s_hdr_bits = b.pos;
	}
	__syncthreads();

	/* ---- rounds of ZT * ZE elements: bits, scan, ds_or into the stage, full words out ---- */
	uint32_t *const out = A.slots + ((size_t)tile * (size_t)A.nseg + (size_t)seg) * ZSLOT_WORDS;
	uint32_t bitbase = s_hdr_bits, wbase = 0;
	for (uint32_t r0 = 0; r0 < n; r0 += ZT * ZE) {
		uint64_t bits[ZE];
		uint32_t nb[ZE], mine = 0;
#pragma unroll
		for (int e = 0; e < ZE; e++) {
			const uint32_t k = r0 + (uint32_t)tid * ZE + (uint32_t)e;
			nb[e] = 0;
			bits[e] = 0;
			if (k < n)
				nb[e] = sa_z_element(W, el[k], sa_z_match(el, k), &bits[e]);
			mine += nb[e];
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
	__shared__ uint32_t s_tail_bytes;
	if (tid == 0) {
		SaZBits b{ stage, bitbase };
		s_tail_bytes = sa_z_finish_segment(W, b);
	}
	__syncthreads();
	const uint32_t tail_words = (s_tail_bytes + 3u) >> 2;
	for (uint32_t w = tid; w < tail_words; w += ZT)
		out[wbase + w] = stage[w];
	if (tid == 0) {
		const size_t at = (size_t)tile * (size_t)A.nseg + (size_t)seg;
		A.seg_bytes[at] = wbase * 4u + s_tail_bytes;
		A.seg_s1[at] = (uint32_t)(W.s1 % 65521ull);
		A.seg_s2[at] = (uint32_t)(W.s2 % 65521ull);
	}
}

/* where the segments of a tile go: [78 9c][segment 0]...[segment nseg-1][01 00 00 ff ff][Adler-32, big endian] */
__global__ __launch_bounds__(ZT) void sa_k_deflate_offsets(SaZArgs A, uint32_t *seg_off, uint8_t *outb, size_t tile_bound,
							   unsigned long long *tile_bytes)
{
	__shared__ uint32_t s_len[1024], s_s1[1024], s_s2[1024];
	const int tile = blockIdx.x, tid = threadIdx.x;
	const size_t base = (size_t)tile * (size_t)A.nseg;
	const uint32_t tile_elems = (uint32_t)A.chunk * (uint32_t)A.chunk;
	uint8_t *const o = outb + (size_t)tile * tile_bound;
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
				seg_off[base + s0 + s] = at;
				at += s_len[s];
				const uint32_t e0 = (uint32_t)(s0 + s) * ZSEG;
				const uint32_t n = tile_elems - e0 < (uint32_t)ZSEG ? tile_elems - e0 : (uint32_t)ZSEG;
				sa_z_adler_append(a, b, s_s1[s], s_s2[s], 4ull * n);
			}
		}
	}
	if (tid == 0) {
		o[0] = 0x78;
		o[1] = 0x9c;
		o[at++] = 0x01;
		o[at++] = 0x00;
		o[at++] = 0x00;
		o[at++] = 0xff;
		o[at++] = 0xff;
		o[at++] = (uint8_t)(b >> 8);
		o[at++] = (uint8_t)b;
		o[at++] = (uint8_t)(a >> 8);
		o[at++] = (uint8_t)a;
		tile_bytes[tile] = at;
	}
}

__global__ __launch_bounds__(ZT) void sa_k_deflate_gather(SaZArgs A, const uint32_t *seg_off, uint8_t *outb, size_t tile_bound)
{
	const int seg = blockIdx.x, tile = blockIdx.y, tid = threadIdx.x;
	const size_t at = (size_t)tile * (size_t)A.nseg + (size_t)seg;
	const uint32_t *const srcw = A.slots + at * ZSLOT_WORDS;
	const uint8_t *const srcb = reinterpret_cast<const uint8_t *>(srcw);
	const uint32_t len = A.seg_bytes[at];
	uint8_t *const dst = outb + (size_t)tile * tile_bound + seg_off[at];
	uint32_t head = (uint32_t)((4u - (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 3u)) & 3u);
	if (head > len)
		head = len;
	if ((uint32_t)tid < head)
		dst[tid] = srcb[tid];
	const uint32_t nd = (len - head) >> 2, sh = 8u * (head & 3u);
	uint32_t *const dstw = reinterpret_cast<uint32_t *>(dst + head);
	for (uint32_t q = tid; q < nd; q += ZT) {
		const uint32_t w0 = (head >> 2) + q; /* source byte head + 4 q */
		dstw[q] = sh ? (srcw[w0] >> sh) | (srcw[w0 + 1] << (32u - sh)) : srcw[w0];
	}
	const uint32_t done = head + 4u * nd;
	if ((uint32_t)tid < len - done)
		dst[done + tid] = srcb[done + tid];
}

/* level 0: the tiles as they are (a chunked dataset without filters takes them through H5Dwrite_chunk just the same): one
 * workgroup per row of a tile */
__global__ __launch_bounds__(ZT) void sa_k_tiles_stored(SaZArgs A, uint32_t *out)
{
	const int r = blockIdx.x, tile = blockIdx.y;
	const int64_t i = (int64_t)A.tile_row * A.chunk + r, j0 = (int64_t)(A.tile_col0 + tile) * A.chunk;
	uint32_t *const o = out + ((size_t)tile * (size_t)A.chunk + (size_t)r) * (size_t)A.chunk;
	for (int c = threadIdx.x; c < A.chunk; c += ZT)
		o[c] = z_fetch(A, i, j0 + c);
}

} // namespace

/* ---- host side ---------------------------------------------------------------------------------------------------- */
struct sa_zjob {
	int device = 0;
	int32_t num = 0, chunk = 0, chunk_shift = 0, nc = 0, nseg = 0;
	bool stored = false; /* level 0: raw tiles */
	const int32_t *d_packed = nullptr, *d_full = nullptr;
	int32_t *d_owned = nullptr; /* the packed matrix, when the job made it (sa_hip_deflate_begin) */
	sa_ctx *ctx = nullptr;
	uint32_t *d_slots = nullptr, *d_seg_bytes = nullptr, *d_s1 = nullptr, *d_s2 = nullptr, *d_seg_off = nullptr;
	uint8_t *d_out = nullptr;
	size_t tile_bound = 0;
	unsigned long long *d_tile_bytes = nullptr, *h_tile_bytes = nullptr;
	uint8_t *h_buf = nullptr;
	size_t h_cap = 0;
	hipStream_t stream = nullptr;
	int64_t encoded_row = -1; /* the tile row whose segments and offsets are in the device buffers (or on their way) */
	double encode_ms = 0, copy_ms = 0;
	uint64_t raw_bytes = 0, out_bytes = 0;
};

static void zjob_free(sa_zjob *z)
{
	if (!z)
		return;
	(void)hipSetDevice(z->device);
	if (z->stream) {
		(void)hipStreamSynchronize(z->stream);
		(void)hipStreamDestroy(z->stream);
	}
	(void)hipFree(z->d_slots);
	(void)hipFree(z->d_seg_bytes);
	(void)hipFree(z->d_s1);
	(void)hipFree(z->d_s2);
	(void)hipFree(z->d_seg_off);
	(void)hipFree(z->d_out);
	(void)hipFree(z->d_tile_bytes);
	(void)hipFree(z->d_owned);
	if (z->h_tile_bytes)
		(void)hipHostFree(z->h_tile_bytes);
	if (z->h_buf)
		(void)hipHostFree(z->h_buf);
	if (z->ctx)
		sa_ctx_destroy(z->ctx);
	delete z;
}

static SaZArgs zjob_args(const sa_zjob *z, int64_t row)
{
	SaZArgs a{};
	a.packed = z->d_packed;
	a.full = z->d_full;
	a.num = z->num;
	a.chunk = z->chunk;
	a.chunk_shift = z->chunk_shift;
	a.tile_row = (int32_t)row;
	a.tile_col0 = 0;
	a.nseg = z->nseg;
	a.slots = z->d_slots;
	a.seg_bytes = z->d_seg_bytes;
	a.seg_s1 = z->d_s1;
	a.seg_s2 = z->d_s2;
	return a;
}

/* segments + offsets of a tile row onto the job's stream; the sizes follow into page-locked memory */
static bool zjob_encode(sa_zjob *z, int64_t row)
{
	const SaZArgs a = zjob_args(z, row);
	if (z->stored) {
		hipLaunchKernelGGL(sa_k_tiles_stored, dim3((unsigned)z->chunk, (unsigned)z->nc), dim3(ZT), 0, z->stream, a,
				   reinterpret_cast<uint32_t *>(z->d_out));
		SA_HIP_CHECK(hipGetLastError(), return false);
		z->encoded_row = row;
		return true;
	}
	hipLaunchKernelGGL(sa_k_deflate_segments, dim3((unsigned)z->nseg, (unsigned)z->nc), dim3(ZT), ZLDS_BYTES, z->stream, a);
	SA_HIP_CHECK(hipGetLastError(), return false);
	hipLaunchKernelGGL(sa_k_deflate_offsets, dim3((unsigned)z->nc), dim3(ZT), 0, z->stream, a, z->d_seg_off, z->d_out, z->tile_bound,
			   z->d_tile_bytes);
	SA_HIP_CHECK(hipGetLastError(), return false);
	SA_HIP_CHECK(hipMemcpyAsync(z->h_tile_bytes, z->d_tile_bytes, sizeof(unsigned long long) * (size_t)z->nc, hipMemcpyDeviceToHost,
				    z->stream),
		     return false);
	z->encoded_row = row;
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
	z->num = num;
	z->chunk = (int32_t)chunk_dim;
	z->chunk_shift = shift;
	z->nc = (int32_t)(((size_t)num + chunk_dim - 1) / chunk_dim);
	z->nseg = (int32_t)((chunk_dim * chunk_dim + ZSEG - 1) / ZSEG);
	z->d_packed = d_packed;
	z->d_full = d_packed ? nullptr : d_full;
	z->stored = stored;
	const size_t segs = (size_t)z->nc * (size_t)z->nseg;
	/* a tile's stream at its very worst: header, 63 bits per element, the segments' ends */
	z->tile_bound = (((size_t)z->nseg * ZSLOT_WORDS * 4 + 64) + 255) & ~(size_t)255;
	if (stored)
		z->tile_bound = chunk_dim * chunk_dim * sizeof(int32_t);
	bool ok = false;
	do {
		SA_HIP_CHECK(hipStreamCreateWithFlags(&z->stream, hipStreamNonBlocking), break);
		SA_HIP_CHECK(hipMalloc(&z->d_out, (size_t)z->nc * z->tile_bound), break);
		if (stored) {
			ok = true;
			break;
		}
		static std::atomic<unsigned long long> raised{ 0 }; /* bit d: more than 64 KB of dynamic LDS opted into on device d */
		const unsigned long long bit = 1ull << (device & 63);
		if (!(raised.load() & bit)) {
			SA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_k_deflate_segments),
							 hipFuncAttributeMaxDynamicSharedMemorySize, (int)ZLDS_BYTES),
				     break);
			raised.fetch_or(bit);
		}
		SA_HIP_CHECK(hipMalloc(&z->d_slots, segs * ZSLOT_WORDS * sizeof(uint32_t)), break);
		SA_HIP_CHECK(hipMalloc(&z->d_seg_bytes, segs * sizeof(uint32_t)), break);
		SA_HIP_CHECK(hipMalloc(&z->d_s1, segs * sizeof(uint32_t)), break);
		SA_HIP_CHECK(hipMalloc(&z->d_s2, segs * sizeof(uint32_t)), break);
		SA_HIP_CHECK(hipMalloc(&z->d_seg_off, segs * sizeof(uint32_t)), break);
		SA_HIP_CHECK(hipMalloc(&z->d_tile_bytes, sizeof(unsigned long long) * (size_t)z->nc), break);
		SA_HIP_CHECK(hipHostMalloc(&z->h_tile_bytes, sizeof(unsigned long long) * (size_t)z->nc, hipHostMallocDefault), break);
		ok = true;
	} while (0);
	if (!ok) {
		zjob_free(z);
		return nullptr;
	}
	return z;
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
		if (!z || !streams || !sizes || tile_row >= (size_t)z->nc) {
			sa_set_error("sa_zjob_tile_row: bad arguments");
			return 1;
		}
		SA_HIP_CHECK(hipSetDevice(z->device), return 1);
		const auto t0 = std::chrono::steady_clock::now();
		if (z->encoded_row != (int64_t)tile_row && !zjob_encode(z, (int64_t)tile_row))
			return 1;
		SA_HIP_CHECK(hipStreamSynchronize(z->stream), return 1); /* the sizes are here */
		z->encode_ms += sa_ms_since(t0);
		const auto t1 = std::chrono::steady_clock::now();
		size_t total = 0;
		for (int t = 0; t < z->nc; t++) {
			sizes[t] = z->stored ? z->tile_bound : (size_t)z->h_tile_bytes[t];
			if (sizes[t] > z->tile_bound) {
				sa_set_error("sa_zjob_tile_row: a tile's stream outgrew its bound (%zu > %zu)", sizes[t], z->tile_bound);
				return 1;
			}
			total += (sizes[t] + 63) & ~(size_t)63;
		}
		if (total > z->h_cap) {
			if (z->h_buf)
				(void)hipHostFree(z->h_buf);
			z->h_buf = nullptr;
			z->h_cap = 0;
			const size_t want = total + total / 8 + (1 << 20);
			SA_HIP_CHECK(hipHostMalloc(&z->h_buf, want, hipHostMallocDefault), return 1);
			z->h_cap = want;
		}
		if (!z->stored) {
			const SaZArgs a = zjob_args(z, (int64_t)tile_row);
			hipLaunchKernelGGL(sa_k_deflate_gather, dim3((unsigned)z->nseg, (unsigned)z->nc), dim3(ZT), 0, z->stream, a, z->d_seg_off,
					   z->d_out, z->tile_bound);
			SA_HIP_CHECK(hipGetLastError(), return 1);
		}
		size_t at = 0;
		for (int t = 0; t < z->nc; t++) {
			SA_HIP_CHECK(hipMemcpyAsync(z->h_buf + at, z->d_out + (size_t)t * z->tile_bound, sizes[t], hipMemcpyDeviceToHost, z->stream),
				     return 1);
			streams[t] = z->h_buf + at;
			at += (sizes[t] + 63) & ~(size_t)63;
			z->out_bytes += sizes[t];
		}
		z->raw_bytes += (uint64_t)z->nc * (uint64_t)z->chunk * (uint64_t)z->chunk * 4u;
		SA_HIP_CHECK(hipStreamSynchronize(z->stream), return 1);
		z->copy_ms += sa_ms_since(t1);
		/* the next row is encoded while the caller writes this one (its sizes land in h_tile_bytes, which the caller no
		 * longer needs: sizes[] is its own copy) */
		if (tile_row + 1 < (size_t)z->nc && !zjob_encode(z, (int64_t)tile_row + 1))
			return 1;
		return 0;
	});
}

extern "C" void sa_zjob_stats(const sa_zjob *z, double *encode_ms, double *copy_ms, uint64_t *raw_bytes, uint64_t *out_bytes)
{
	if (!z)
		return;
	if (encode_ms)
		*encode_ms = z->encode_ms;
	if (copy_ms)
		*copy_ms = z->copy_ms;
	if (raw_bytes)
		*raw_bytes = z->raw_bytes;
	if (out_bytes)
		*out_bytes = z->out_bytes;
}

/* the alignment into device memory (all pairs, packed) and a job over it; *align_seconds = the launch loop's time */
extern "C" sa_zjob *sa_hip_deflate_begin(struct sa_input in, const struct sa_scoring *sc, size_t chunk_dim, int level, double *align_seconds)
{
	return sa_guard("sa_hip_deflate_begin", (sa_zjob *)nullptr, [&]() -> sa_zjob * {
		sa_ctx *ctx = sa_ctx_create(0, in, sc);
		if (!ctx)
			return nullptr;
		int32_t *d_packed = nullptr;
		sa_zjob *z = nullptr;
		bool ok = false;
		do {
			const int64_t pairs = sa_ctx_pairs(ctx);
			SA_HIP_CHECK(hipMalloc(&d_packed, sizeof(int32_t) * (size_t)pairs), break);
			if (!sa_prepare_range(ctx, 0, pairs, false))
				break;
			SA_HIP_CHECK(hipDeviceSynchronize(), break);
			const auto t0 = std::chrono::steady_clock::now();
			if (sa_ctx_align_range(ctx, 0, pairs, d_packed, nullptr) != 0)
				break;
			SA_HIP_CHECK(hipDeviceSynchronize(), break);
			if (align_seconds)
				*align_seconds = sa_ms_since(t0) * 1e-3;
			z = zjob_make(0, d_packed, nullptr, in.num, chunk_dim, level == 0);
			if (!z)
				break;
			ok = true;
		} while (0);
		if (!ok) {
			(void)hipFree(d_packed);
			sa_ctx_destroy(ctx);
			return nullptr;
		}
		z->d_owned = d_packed;
		z->ctx = ctx;
		return z;
	});
}
