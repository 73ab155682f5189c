// Microbenchmark (development tool): how many waves are REALLY co-resident per SIMD?  Every wave records the
// hardware slot it ran on (HW_REG_HW_ID, HW_REG_XCC_ID) and its [start, end] on the 100 MHz s_memrealtime clock;
// the host counts the maximum number of overlapping intervals per (XCC, SE, SH, CU, SIMD).
// Usage: census <blocks per CU> <threads per block> <vgprs: 32|64|96|128> <lds bytes per block>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <map>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Rec { unsigned hwid, xcc; unsigned long long t0, t1; };

template <int NV> __global__ void k(Rec *rec, int iters)
{
	extern __shared__ int lds[];
	int v[NV];
	for (int q = 0; q < NV; q++) v[q] = threadIdx.x + q;
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int q = 0; q < NV; q++) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(v[q]) : "v"(v[(q + 1) % NV]));
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
	int s = 0;
	for (int q = 0; q < NV; q++) s += v[q];
	if (s == 0x7fffffff) lds[threadIdx.x] = s;
	if ((threadIdx.x & 63) == 0) {
		unsigned hwid, xcc;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
		Rec r{ hwid, xcc, t0, t1 };
		rec[(size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = r;
	}
}

int main(int argc, char **argv)
{
	const int bpc = argc > 1 ? atoi(argv[1]) : 8, threads = argc > 2 ? atoi(argv[2]) : 256, nv = argc > 3 ? atoi(argv[3]) : 32;
	const int lds = argc > 4 ? atoi(argv[4]) : 0;
	const int blocks = 256 * bpc, wpb = threads / 64;
	Rec *d; CHECK(hipMalloc(&d, sizeof(Rec) * blocks * wpb));
	const int iters = 200000 / nv;
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	for (int rep = 0; rep < 2; rep++) {
		CHECK(hipEventRecord(e0, 0));
		if (nv == 32) hipLaunchKernelGGL(k<24>, dim3(blocks), dim3(threads), lds, 0, d, iters);
		else if (nv == 64) hipLaunchKernelGGL(k<56>, dim3(blocks), dim3(threads), lds, 0, d, iters);
		else if (nv == 96) hipLaunchKernelGGL(k<88>, dim3(blocks), dim3(threads), lds, 0, d, iters);
		else hipLaunchKernelGGL(k<120>, dim3(blocks), dim3(threads), lds, 0, d, iters);
		CHECK(hipEventRecord(e1, 0));
		CHECK(hipDeviceSynchronize());
	}
	float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
	std::vector<Rec> h(blocks * wpb);
	CHECK(hipMemcpy(h.data(), d, sizeof(Rec) * h.size(), hipMemcpyDeviceToHost));
	std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev; // per SIMD: (time, +1/-1)
	unsigned long long tmin = ~0ull, tmax = 0; double dur = 0;
	for (auto &r : h) {
		const unsigned long long key = ((unsigned long long)(r.xcc & 0xf) << 32) | (r.hwid & 0xff30u);
		ev[key].push_back({ r.t0, +1 }); ev[key].push_back({ r.t1, -1 });
		tmin = std::min(tmin, r.t0); tmax = std::max(tmax, r.t1); dur += (double)(r.t1 - r.t0);
	}
	std::map<int, int> hist; int simds = 0;
	for (auto &kv : ev) {
		auto &v = kv.second;
		std::sort(v.begin(), v.end(), [](auto &a, auto &b) { return a.first < b.first || (a.first == b.first && a.second < b.second); });
		int cur = 0, mx = 0; for (auto &e : v) { cur += e.second; mx = std::max(mx, cur); }
		hist[mx]++; simds++;
	}
	printf("blocks/CU=%d threads=%d vgpr-class=%d lds=%d: %zu waves on %d distinct SIMDs; kernel %.1f us (events), first start..last end %.1f us, mean wave lifetime %.1f us\n",
	       bpc, threads, nv, lds, h.size(), simds, ms * 1e3, (tmax - tmin) / 100.0, dur / h.size() / 100.0);
	printf("  max co-resident waves per SIMD -> number of SIMDs:");
	for (auto &kv : hist) printf("  %d:%d", kv.first, kv.second);
	printf("\n");
	return 0;
}
