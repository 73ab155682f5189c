// Microbenchmark (development tool): v_pk_maximum3_f16 as a packed UNSIGNED 3-way max on gfx950.
// For bit patterns 0x0000..0x7c00 (non-negative f16, zero .. +inf) the f16 order is the u16 order, so the packed
// 3-operand f16 maximum is a two-lane max3_u16 -- if the instruction neither flushes denormals nor touches payloads.
// Part 1 checks that exhaustively over all pairs (every operand position, both halves); part 2 measures its issue cost
// next to v_pk_max_u16 and the shape the NW chain would use (add + max3 per register).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t max3f(uint32_t a, uint32_t b, uint32_t c)
{
	uint32_t d;
	asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
	return d;
}

__global__ void check(unsigned long long *bad, uint32_t limit)
{
	const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
	if (a > limit) return;
	unsigned long long nb = 0;
	for (uint32_t b = 0; b <= limit; b++) {
		const uint32_t m = a > b ? a : b;
		const uint32_t lo_hi = a | (b << 16), hi_lo = b | (a << 16), z = 0, mm = m | (m << 16);
		/* halves are independent: low = a vs b, high = b vs a */
		nb += max3f(lo_hi, hi_lo, z) != mm;
		nb += max3f(lo_hi, z, hi_lo) != mm;
		nb += max3f(z, lo_hi, hi_lo) != mm;
		nb += max3f(lo_hi, hi_lo, lo_hi) != mm;
		/* three distinct operands: c = (a + b) / 2 lies between */
		const uint32_t c = (a + b) >> 1;
		nb += max3f(c | (c << 16), lo_hi, hi_lo) != mm;
	}
	if (nb) atomicAdd(bad, nb);
}

constexpr int REP = 1500;
template <int MODE> __global__ __launch_bounds__(256) void rate(uint32_t *out, int seed)
{
	uint32_t v[8], d[8], w = threadIdx.x * 0x00010001u + seed, x = seed * 3;
	for (int q = 0; q < 8; q++) v[q] = (threadIdx.x + q * seed) & 0x3fff3fffu, d[q] = v[q] ^ 0x11;
	for (int r = 0; r < REP; r++) {
#pragma unroll
		for (int u = 0; u < 4; u++) {
#pragma unroll
			for (int q = 0; q < 8; q++) {
				if (MODE == 0) asm volatile("v_pk_max_u16 %0, %1, %0" : "+v"(v[q]) : "v"(w));
				if (MODE == 1) asm volatile("v_pk_maximum3_f16 %0, %1, %0, %2" : "+v"(v[q]) : "v"(w), "v"(x));
				if (MODE == 2) asm volatile("v_max3_u16 %0, %1, %0, %2" : "+v"(v[q]) : "v"(w), "v"(x));
				if (MODE == 3) { /* NW chain shape, u16: add, max, max (dependent chain through v[q-1]) */
					asm volatile("v_add_u32_e32 %0, %1, %2" : "=v"(d[q]) : "v"(w), "v"(v[(q + 7) & 7]));
					asm volatile("v_pk_max_u16 %0, %1, %0" : "+v"(v[q]) : "v"(d[q]));
					asm volatile("v_pk_max_u16 %0, %1, %0" : "+v"(v[q]) : "v"(v[(q + 7) & 7]));
				}
				if (MODE == 4) { /* NW chain shape, max3: add, max3 */
					asm volatile("v_add_u32_e32 %0, %1, %2" : "=v"(d[q]) : "v"(w), "v"(v[(q + 7) & 7]));
					asm volatile("v_pk_maximum3_f16 %0, %1, %0, %2" : "+v"(v[q]) : "v"(d[q]), "v"(v[(q + 7) & 7]));
				}
				if (MODE == 5) { /* max3 chain with the adds hoisted (8 adds, then 8 dependent max3) */
					asm volatile("v_add_u32_e32 %0, %1, %2" : "=v"(d[q]) : "v"(w), "v"(v[(q + 7) & 7]));
				}
			}
			if (MODE == 5) {
#pragma unroll
				for (int q = 0; q < 8; q++)
					asm volatile("v_pk_maximum3_f16 %0, %1, %0, %2" : "+v"(v[q]) : "v"(d[q]), "v"(v[(q + 7) & 7]));
			}
		}
	}
	uint32_t s = 0;
	for (int q = 0; q < 8; q++) s += v[q] + d[q];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> int run(const char *name, double per)
{
	uint32_t *out;
	CHECK(hipMalloc(&out, 4 * 256 * 8 * 256));
	printf("%-44s", name);
	for (int bpc : { 1, 2, 4, 8 }) {
		hipEvent_t e0, e1;
		CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
		hipLaunchKernelGGL(rate<MODE>, dim3(256 * bpc), dim3(256), 0, 0, out, 3);
		CHECK(hipEventRecord(e0, 0));
		hipLaunchKernelGGL(rate<MODE>, dim3(256 * bpc), dim3(256), 0, 0, out, 3);
		CHECK(hipEventRecord(e1, 0));
		CHECK(hipDeviceSynchronize());
		float ms = 0.f;
		CHECK(hipEventElapsedTime(&ms, e0, e1));
		printf("  %6.3f ns", (double)ms * 1e6 / ((double)REP * 4 * 8 * per * bpc));
	}
	printf("\n");
	(void)hipFree(out);
	return 0;
}

int main()
{
	unsigned long long *bad, h = 0;
	CHECK(hipMalloc(&bad, 8));
	CHECK(hipMemset(bad, 0, 8));
	const uint32_t limit = 0x7c00;
	hipLaunchKernelGGL(check, dim3((limit + 256) / 256), dim3(256), 0, 0, bad, limit);
	CHECK(hipDeviceSynchronize());
	CHECK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
	printf("v_pk_maximum3_f16 vs unsigned max over all pairs of 0x0000..0x%04x, 5 operand arrangements, both halves: %llu mismatches\n", limit, h);
	printf("host-timed ns per wave64 instruction (or per register step of a chain) and SIMD; columns = 1,2,4,8 waves/SIMD\n");
	run<0>("v_pk_max_u16", 1);
	run<1>("v_pk_maximum3_f16", 1);
	run<2>("v_max3_u16", 1);
	run<3>("NW register step: add + 2 v_pk_max_u16", 1);
	run<4>("NW register step: add + v_pk_maximum3_f16", 1);
	run<5>("NW register step: adds hoisted, max3 chain", 1);
	return h != 0;
}
