// Microbenchmark (development tool): the exact steady-state NW step of sa_k_systolic<nw,16,7>
// (7 SDWA adds + 7 v_max3 + v_mov + v_mov_dpp, 16 steps unrolled) with nothing else around it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITERS = 4000;
__device__ __forceinline__ int imax3(int a, int b, int c) { int m = a > b ? a : b; return m > c ? m : c; }

template <int K, int UNROLL, bool DPP> __global__ __launch_bounds__(256) void k(int *out, unsigned long long *cyc, const int *seed)
{
	const int lane = threadIdx.x & 63;
	int V[K], vprev = lane, inj = seed[1];
	for (int q = 0; q < K; q++) V[q] = lane * q;
	uint2 pw[16];
	for (int s = 0; s < 16; s++) { pw[s].x = (lane * 7 + s * 13 + seed[2]) * 0x01030507u; pw[s].y = pw[s].x * 3; }
	unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < ITERS * (16 / UNROLL); it++) {
#pragma unroll
		for (int s = 0; s < UNROLL; s++) {
			const uint2 p = pw[s];
			const int vleft = DPP ? __builtin_amdgcn_update_dpp(inj, V[K - 1], 0x111, 0xf, 0xf, false) : inj;
			int d[K];
			d[0] = vprev + (int)(int8_t)(p.x);
#pragma unroll
			for (int q = 1; q < K; q++) {
				const unsigned w = q < 4 ? p.x : p.y;
				d[q] = V[q - 1] + (int)(int8_t)(w >> (8 * (q & 3)));
			}
			V[0] = imax3(d[0], V[0], vleft);
#pragma unroll
			for (int q = 1; q < K; q++) V[q] = imax3(d[q], V[q], V[q - 1]);
			vprev = vleft;
		}
	}
	unsigned long long t1 = __builtin_amdgcn_s_memtime();
	int sum = vprev;
	for (int q = 0; q < K; q++) sum += V[q];
	out[blockIdx.x * 256 + threadIdx.x] = sum;
	if (lane == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}

template <int K, int UNROLL, bool DPP> int run(const char *name, int bpc, int *out, unsigned long long *cyc, int *seed)
{
	const int blocks = 256 * bpc;
	for (int it = 0; it < 2; it++) hipLaunchKernelGGL((k<K, UNROLL, DPP>), dim3(blocks), dim3(256), 0, 0, out, cyc, seed);
	CHECK(hipDeviceSynchronize());
	std::vector<unsigned long long> hc(blocks * 4);
	CHECK(hipMemcpy(hc.data(), cyc, 8 * hc.size(), hipMemcpyDeviceToHost));
	double c = 0; for (auto v : hc) c += v; c /= hc.size();
	printf("%-34s waves/SIMD=%d  cycles/step/wave=%7.1f  SIMD-cycles/step=%6.1f  per cell=%5.2f\n", name, bpc, c / (ITERS * 16.0),
	       c / (ITERS * 16.0) / bpc, c / (ITERS * 16.0) / bpc / K);
	return 0;
}

int main()
{
	int *out, *seed; unsigned long long *cyc;
	CHECK(hipMalloc(&out, 4 * 256 * 8 * 256)); CHECK(hipMalloc(&cyc, 8 * 256 * 8 * 4)); CHECK(hipMalloc(&seed, 16));
	int hs[4] = {1, 2, 3, 4}; CHECK(hipMemcpy(seed, hs, 16, hipMemcpyHostToDevice));
	for (int b : {2, 4, 6, 8}) {
		run<7, 16, true>("K=7 unroll 16 (real shape)", b, out, cyc, seed);
		run<7, 2, true>("K=7 unroll 2", b, out, cyc, seed);
		run<7, 16, false>("K=7 unroll 16, no DPP", b, out, cyc, seed);
		run<14, 16, true>("K=14 unroll 16", b, out, cyc, seed);
	}
	return 0;
}
