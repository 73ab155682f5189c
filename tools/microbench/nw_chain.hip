// Microbenchmark (development tool): what bounds the NW systolic step -- per-wave dependent-issue latency or SIMD
// issue bandwidth?  Bare steady-state step (K SDWA adds + K v_max3 + DPP move), VGPRs capped at 64 so that up to
// 8 waves/SIMD are really resident, in three dependency shapes:
//   chain1  one chain of K max3 per step (the kernel's shape: V[q] needs V[q-1] of the same step)
//   chain2  the lane's columns split in two half-chains that work on consecutive rows (software skew): the two
//           chains of a step are independent and are issued interleaved
//   chain4  four quarter-chains
// Reports s_memtime ticks per step and wave, per SIMD, per cell, and the host-timed wall clock of the launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITERS = 3000;
// residency record per wave: hardware slot + [start, end] on the constant 100 MHz clock (host: co-residency per SIMD)
struct Rec { unsigned hwid, xcc; unsigned long long r0, r1; };
__device__ Rec g_rec[256 * 8 * 4];
#define REC_BEGIN const unsigned long long rec_r0 = __builtin_amdgcn_s_memrealtime();
#define REC_END if (lane == 0) { unsigned hwid, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid)); \
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); \
	g_rec[blockIdx.x * 4 + threadIdx.x / 64] = Rec{ hwid, xcc, rec_r0, __builtin_amdgcn_s_memrealtime() }; }
#include <map>
#include <algorithm>
static void residency(int waves)
{
	std::vector<Rec> h(waves);
	if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_rec), sizeof(Rec) * waves) != hipSuccess) return;
	std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
	unsigned long long tmin = ~0ull, tmax = 0; double life = 0;
	for (auto &r : h) {
		const unsigned long long key = ((unsigned long long)(r.xcc & 0xf) << 32) | (r.hwid & 0xff30u);
		ev[key].push_back({ r.r0, +1 }); ev[key].push_back({ r.r1, -1 });
		tmin = std::min(tmin, r.r0); tmax = std::max(tmax, r.r1); life += (double)(r.r1 - r.r0);
	}
	int mx_all = 0, mn_all = 99;
	for (auto &kv : ev) {
		auto &v = kv.second;
		std::sort(v.begin(), v.end(), [](auto &a, auto &b) { return a.first < b.first || (a.first == b.first && a.second < b.second); });
		int cur = 0, mx = 0; for (auto &e : v) { cur += e.second; mx = std::max(mx, cur); }
		mx_all = std::max(mx_all, mx); mn_all = std::min(mn_all, mx);
	}
	printf("      residency: %zu SIMDs, max co-resident waves/SIMD %d..%d, span %.0f us, mean wave lifetime %.0f us, mean resident waves/SIMD %.2f\n",
	       ev.size(), mn_all, mx_all, (tmax - tmin) / 100.0, life / waves / 100.0, life / ((double)ev.size() * (double)(tmax - tmin)));
}

__device__ __forceinline__ int imax3(int a, int b, int c) { int m = a > b ? a : b; return m > c ? m : c; }
__device__ __forceinline__ int sbyte(const uint4 &w, int q)
{
	const unsigned v = q < 4 ? w.x : q < 8 ? w.y : q < 12 ? w.z : w.w;
	return (int)(int8_t)(v >> (8 * (q & 3)));
}

template <int K, int CHAINS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k(int *out, unsigned long long *cyc, const int *seed)
{
	const int lane = threadIdx.x & 63;
	constexpr int H = K / CHAINS; // columns per chain
	int V[K], head[CHAINS], inj = seed[1];
	for (int q = 0; q < K; q++) V[q] = lane * q;
	for (int c = 0; c < CHAINS; c++) head[c] = lane + c;
	uint4 pw[2];
	for (int s = 0; s < 2; s++) {
		pw[s].x = (lane * 7 + s * 13 + seed[2]) * 0x01030507u; pw[s].y = pw[s].x * 3; pw[s].z = pw[s].x * 5; pw[s].w = pw[s].x * 7;
	}
	REC_BEGIN
	unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < ITERS; it++) {
#pragma unroll
		for (int s = 0; s < 16; s++) {
			uint4 p = pw[s & 1];
			asm volatile("" : "+v"(p.x), "+v"(p.y), "+v"(p.z), "+v"(p.w)); // opaque: the real kernel reads a fresh profile row per step
			// chain c covers columns [c*H, (c+1)*H); its left/diagonal input is the previous chain's last column as it
			// was BEFORE this step (the previous row in the skewed schedule), chain 0 takes the lane shift
			int left[CHAINS], diag[CHAINS];
			left[0] = __builtin_amdgcn_update_dpp(inj, V[K - 1], 0x111, 0xf, 0xf, false);
#pragma unroll
			for (int c = 1; c < CHAINS; c++) left[c] = V[c * H - 1];
#pragma unroll
			for (int c = 0; c < CHAINS; c++) diag[c] = head[c];
			int d[K];
#pragma unroll
			for (int q = 0; q < K; q++) d[q] = ((q % H) ? V[q - 1] : diag[q / H]) + sbyte(p, q);
			__builtin_amdgcn_sched_barrier(0);
			// interleave the chains: column i of every chain, then column i+1 ...
#pragma unroll
			for (int i = 0; i < H; i++) {
#pragma unroll
				for (int c = 0; c < CHAINS; c++) {
					const int q = c * H + i;
					V[q] = imax3(d[q], V[q], i ? V[q - 1] : left[c]);
				}
			}
#pragma unroll
			for (int c = 0; c < CHAINS; c++) head[c] = left[c];
		}
	}
	unsigned long long t1 = __builtin_amdgcn_s_memtime();
	int sum = 0;
	for (int q = 0; q < K; q++) sum += V[q];
	for (int c = 0; c < CHAINS; c++) sum += head[c];
	out[blockIdx.x * 256 + threadIdx.x] = sum;
	if (lane == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}

// packed variant: every register holds the same column of TWO column sequences as s16 halves (K registers = 2K cells):
// v_pk_add_i16 + 2 v_pk_max_i16 per register, the score pairs come as K dwords per step (s16 profile)
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pkmax(unsigned a, unsigned b) { unsigned r; asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ unsigned pkadd(unsigned a, unsigned b) { unsigned r; asm("v_pk_add_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <int K, int CHAINS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void kpk(int *out, unsigned long long *cyc, const int *seed)
{
	const int lane = threadIdx.x & 63;
	constexpr int H = K / CHAINS;
	unsigned V[K], head[CHAINS], inj = seed[1];
	for (int q = 0; q < K; q++) V[q] = lane * q;
	for (int c = 0; c < CHAINS; c++) head[c] = lane + c;
	unsigned pw[2][K];
	for (int s = 0; s < 2; s++)
		for (int q = 0; q < K; q++) pw[s][q] = (lane * 7 + s * 13 + q + seed[2]) * 0x00030005u;
	REC_BEGIN
	unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < ITERS; it++) {
#pragma unroll
		for (int s = 0; s < 16; s++) {
			unsigned p[K];
#pragma unroll
			for (int q = 0; q < K; q++) { p[q] = pw[s & 1][q]; asm volatile("" : "+v"(p[q])); }
			unsigned left[CHAINS], diag[CHAINS];
			left[0] = __builtin_amdgcn_update_dpp(inj, V[K - 1], 0x111, 0xf, 0xf, false);
#pragma unroll
			for (int c = 1; c < CHAINS; c++) left[c] = V[c * H - 1];
#pragma unroll
			for (int c = 0; c < CHAINS; c++) diag[c] = head[c];
			unsigned d[K];
#pragma unroll
			for (int q = 0; q < K; q++) d[q] = pkadd((q % H) ? V[q - 1] : diag[q / H], p[q]);
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int i = 0; i < H; i++) {
#pragma unroll
				for (int c = 0; c < CHAINS; c++) {
					const int q = c * H + i;
					V[q] = pkmax(pkmax(d[q], V[q]), i ? V[q - 1] : left[c]);
				}
			}
#pragma unroll
			for (int c = 0; c < CHAINS; c++) head[c] = left[c];
		}
	}
	unsigned long long t1 = __builtin_amdgcn_s_memtime();
	unsigned sum = 0;
	for (int q = 0; q < K; q++) sum += V[q];
	for (int c = 0; c < CHAINS; c++) sum += head[c];
	out[blockIdx.x * 256 + threadIdx.x] = (int)sum;
	if (lane == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
	REC_END
}

template <int K, int CHAINS> int runpk(const char *name, int bpc, int *out, unsigned long long *cyc, int *seed)
{
	const int blocks = 256 * bpc;
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	hipLaunchKernelGGL((kpk<K, CHAINS>), dim3(blocks), dim3(256), 0, 0, out, cyc, seed);
	CHECK(hipEventRecord(e0, 0));
	hipLaunchKernelGGL((kpk<K, CHAINS>), dim3(blocks), dim3(256), 0, 0, out, cyc, seed);
	CHECK(hipEventRecord(e1, 0));
	CHECK(hipDeviceSynchronize());
	float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
	std::vector<unsigned long long> hc(blocks * 4);
	CHECK(hipMemcpy(hc.data(), cyc, 8 * hc.size(), hipMemcpyDeviceToHost));
	double c = 0; for (auto v : hc) c += v; c /= hc.size();
	const double steps = ITERS * 16.0;
	printf("%-22s waves/SIMD=%d  ticks/step/wave=%7.1f  ticks/step/SIMD=%6.1f  per cell=%5.2f  per VALU inst=%5.2f   wall: %6.3f ns/cell/SIMD\n",
	       name, bpc, c / steps, c / steps / bpc, c / steps / bpc / (2 * K), c / steps / bpc / (3 * K + 1 + (CHAINS - 1)),
	       (double)ms * 1e6 / (steps * bpc * 2 * K));
	residency(blocks * 4);
	return 0;
}

template <int K, int CHAINS> int run(const char *name, int bpc, int *out, unsigned long long *cyc, int *seed)
{
	const int blocks = 256 * bpc;
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	hipLaunchKernelGGL((k<K, CHAINS>), dim3(blocks), dim3(256), 0, 0, out, cyc, seed);
	CHECK(hipEventRecord(e0, 0));
	hipLaunchKernelGGL((k<K, CHAINS>), dim3(blocks), dim3(256), 0, 0, out, cyc, seed);
	CHECK(hipEventRecord(e1, 0));
	CHECK(hipDeviceSynchronize());
	float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
	std::vector<unsigned long long> hc(blocks * 4);
	CHECK(hipMemcpy(hc.data(), cyc, 8 * hc.size(), hipMemcpyDeviceToHost));
	double c = 0; for (auto v : hc) c += v; c /= hc.size();
	const double steps = ITERS * 16.0;
	printf("%-22s waves/SIMD=%d  ticks/step/wave=%7.1f  ticks/step/SIMD=%6.1f  per cell=%5.2f  per VALU inst=%5.2f   wall: %6.3f ns/cell/SIMD\n",
	       name, bpc, c / steps, c / steps / bpc, c / steps / bpc / K, c / steps / bpc / (2 * K + 1 + (CHAINS - 1)),
	       (double)ms * 1e6 / (steps * bpc * K));
	residency(blocks * 4);
	return 0;
}

int main()
{
	int *out, *seed; unsigned long long *cyc;
	CHECK(hipMalloc(&out, 4 * 256 * 8 * 256)); CHECK(hipMalloc(&cyc, 8 * 256 * 8 * 4)); CHECK(hipMalloc(&seed, 16));
	int hs[4] = {1, 2, 3, 4}; CHECK(hipMemcpy(seed, hs, 16, hipMemcpyHostToDevice));
	for (int b : {1, 2, 4, 8}) {
		run<8, 1>("K=8  chain1", b, out, cyc, seed);
		run<8, 2>("K=8  chain2", b, out, cyc, seed);
		run<16, 1>("K=16 chain1", b, out, cyc, seed);
		run<16, 2>("K=16 chain2", b, out, cyc, seed);
		run<16, 4>("K=16 chain4", b, out, cyc, seed);
		runpk<8, 1>("K=8x2 pk chain1", b, out, cyc, seed);
		runpk<16, 1>("K=16x2 pk chain1", b, out, cyc, seed);
		runpk<16, 2>("K=16x2 pk chain2", b, out, cyc, seed);
	}
	return 0;
}
