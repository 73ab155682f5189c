// Microbenchmark (development tool): cost of one LDS profile read per systolic step next to the NW VALU mix.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITERS = 4000; // x16 steps

__device__ __forceinline__ int imax3(int a, int b, int c) { int m = a > b ? a : b; return m > c ? m : c; }

// MODE 0: no LDS read (profile word constant); 1: one ds_read_b64 per step, all 16 issued at block start;
// 2: ds_read_b64 per step issued one step ahead; 3: like 1 but conflict-free 32-slot layout; 4: two ds_read_b32 per step
template <int MODE> __global__ __launch_bounds__(256) void k(int *out, unsigned long long *cyc, const int *seed)
{
	__shared__ __attribute__((aligned(16))) uint8_t prof[32 * 256];
	const int lane = threadIdx.x & 63;
	for (int i = threadIdx.x; i < 32 * 256 / 4; i += 256) ((int *)prof)[i] = i * 0x01030507 + seed[0];
	__syncthreads();
	int V[7], vprev = lane, inj = seed[1];
	for (int q = 0; q < 7; q++) V[q] = lane * q;
	unsigned tokens[16];
	for (int s = 0; s < 16; s++) tokens[s] = ((lane * 7 + s * 13 + seed[2]) % 24);
	const unsigned slot_off = MODE == 3 ? (lane & 31) * 8 : (lane & 15) * 8;
	const int SH = MODE == 3 ? 8 : 7;
	unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < ITERS; it++) {
		uint2 pws[16];
		if (MODE == 1 || MODE == 3) {
#pragma unroll
			for (int s = 0; s < 16; s++) pws[s] = *(const uint2 *)(prof + ((tokens[s] << SH) | slot_off));
		}
#pragma unroll
		for (int s = 0; s < 16; s++) {
			uint2 pw;
			if (MODE == 0) { pw.x = tokens[s] * 0x01010101u; pw.y = tokens[s] * 0x02020202u; }
			else if (MODE == 1 || MODE == 3) pw = pws[s];
			else if (MODE == 2) pw = *(const uint2 *)(prof + ((tokens[s] << SH) | slot_off));
			else { pw.x = *(const unsigned *)(prof + ((tokens[s] << SH) | slot_off)); pw.y = *(const unsigned *)(prof + ((tokens[s] << SH) | slot_off) + 4); }
			const int vleft = __builtin_amdgcn_update_dpp(inj, V[6], 0x111, 0xf, 0xf, false);
			int d[7];
			d[0] = vprev + (int)(int8_t)(pw.x);
			d[1] = V[0] + (int)(int8_t)(pw.x >> 8);
			d[2] = V[1] + (int)(int8_t)(pw.x >> 16);
			d[3] = V[2] + ((int)pw.x >> 24);
			d[4] = V[3] + (int)(int8_t)(pw.y);
			d[5] = V[4] + (int)(int8_t)(pw.y >> 8);
			d[6] = V[5] + (int)(int8_t)(pw.y >> 16);
			V[0] = imax3(d[0], V[0], vleft);
#pragma unroll
			for (int q = 1; q < 7; q++) V[q] = imax3(d[q], V[q], V[q - 1]);
			vprev = vleft;
			tokens[s] = (tokens[s] + (V[6] & 1) + 1) % 24; // data-dependent next token (keeps the reads in the loop)
		}
	}
	unsigned long long t1 = __builtin_amdgcn_s_memtime();
	int sum = vprev;
	for (int q = 0; q < 7; q++) sum += V[q];
	out[blockIdx.x * 256 + threadIdx.x] = sum;
	if (lane == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}

template <int MODE> int run(const char *name, int bpc, int *out, unsigned long long *cyc, int *seed)
{
	const int blocks = 256 * bpc;
	for (int it = 0; it < 2; it++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, seed);
	CHECK(hipDeviceSynchronize());
	std::vector<unsigned long long> hc(blocks * 4);
	CHECK(hipMemcpy(hc.data(), cyc, 8 * hc.size(), hipMemcpyDeviceToHost));
	double c = 0; for (auto v : hc) c += v; c /= hc.size();
	printf("%-44s waves/SIMD=%d  cycles/step/wave=%7.1f  SIMD-cycles/step=%6.1f\n", name, bpc, c / (ITERS * 16.0), c / (ITERS * 16.0) / bpc);
	return 0;
}

int main()
{
	int *out, *seed; unsigned long long *cyc;
	CHECK(hipMalloc(&out, 4 * 256 * 8 * 256)); CHECK(hipMalloc(&cyc, 8 * 256 * 8 * 4)); CHECK(hipMalloc(&seed, 16));
	int hs[4] = {1, 2, 3, 4}; CHECK(hipMemcpy(seed, hs, 16, hipMemcpyHostToDevice));
	for (int b : {4, 6, 8}) {
		run<0>("no LDS read (VALU mix + token update)", b, out, cyc, seed);
		run<1>("ds_read_b64/step, 16 issued per block, 16 slots", b, out, cyc, seed);
		run<3>("ds_read_b64/step, 16 issued per block, 32 slots", b, out, cyc, seed);
		run<2>("ds_read_b64/step, issued at use", b, out, cyc, seed);
		run<4>("2x ds_read_b32/step, issued at use", b, out, cyc, seed);
	}
	return 0;
}
