// Microbenchmark (development tool): steady-state NW step with TWO column sequences packed as s16 halves.
//   P1: scores as s16 pairs in registers: v_pk_add_i16 + 2 v_pk_max_i16 per column pair
//   P2: scores as s8 pairs (SDWA):        2 v_add_u16_sdwa + 2 v_pk_max_i16 per column pair
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITERS = 4000;
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pkmax(unsigned a, unsigned b)
{
	unsigned r; asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
}
__device__ __forceinline__ unsigned pkadd(unsigned a, unsigned b)
{
	unsigned r; asm("v_pk_add_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
}
template <int BYTE> __device__ __forceinline__ unsigned sdwa_pair(unsigned p, unsigned v)
{
	unsigned r;
	if constexpr (BYTE == 0) {
		asm("v_add_u16_sdwa %0, sext(%1), %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:WORD_0\n\t"
		    "v_add_u16_sdwa %0, sext(%1), %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:WORD_1" : "=&v"(r) : "v"(p), "v"(v));
	} else {
		asm("v_add_u16_sdwa %0, sext(%1), %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:WORD_0\n\t"
		    "v_add_u16_sdwa %0, sext(%1), %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:WORD_1" : "=&v"(r) : "v"(p), "v"(v));
	}
	return r;
}

// P3: one column per 32-bit register, value in ONE 16-bit half (op_sel picks it), s16 score pairs:
//     v_add_u32_e32 (full rate, the other half is garbage) + v_max3_i16 op_sel
template <int HS0, int HS1, int HS2, int HD> __device__ __forceinline__ void max3h(unsigned &dst, unsigned a, unsigned b, unsigned c)
{
	if constexpr (HS0 == 0 && HS1 == 0 && HS2 == 0 && HD == 0) asm("v_max3_i16 %0, %1, %2, %3 op_sel:[0,0,0,0]" : "+v"(dst) : "v"(a), "v"(b), "v"(c));
	else if constexpr (HS0 == 0 && HS1 == 0 && HS2 == 1 && HD == 0) asm("v_max3_i16 %0, %1, %2, %3 op_sel:[0,0,1,0]" : "+v"(dst) : "v"(a), "v"(b), "v"(c));
	else if constexpr (HS0 == 1 && HS1 == 1 && HS2 == 0 && HD == 1) asm("v_max3_i16 %0, %1, %2, %3 op_sel:[1,1,0,1]" : "+v"(dst) : "v"(a), "v"(b), "v"(c));
	else asm("v_max3_i16 %0, %1, %2, %3 op_sel:[0,0,0,0]" : "+v"(dst) : "v"(a), "v"(b), "v"(c));
}
template <int K> __global__ __launch_bounds__(256) void k3(unsigned *out, unsigned long long *cyc, const int *seed)
{
	const int lane = threadIdx.x & 63;
	unsigned V[K], vprev = lane, inj = seed[1];
	for (int q = 0; q < K; q++) V[q] = (lane * q) << ((q & 1) ? 0 : 16);
	unsigned pw[16][(K + 1) / 2];
	for (int s = 0; s < 16; s++)
		for (int q = 0; q < (K + 1) / 2; q++) pw[s][q] = (lane * 7 + s * 13 + q + seed[2]) * 0x01030507u;
	unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < ITERS; it++) {
#pragma unroll
		for (int s = 0; s < 16; s++) {
			const unsigned vleft = __builtin_amdgcn_update_dpp(inj, V[K - 1], 0x111, 0xf, 0xf, false);
			unsigned d[K];
#pragma unroll
			for (int q = 0; q < K; q++) d[q] = (q ? V[q - 1] : vprev) + pw[s][q / 2];
			/* halves: even q -> hi, odd q -> lo, except the last column (lo); vleft/vprev lo */
#pragma unroll
			for (int q = 0; q < K; q++) {
				const unsigned left = q ? V[q - 1] : vleft;
				if (q == 0) max3h<1, 1, 0, 1>(V[q], d[q], V[q], left);          /* d hi, self hi, left lo -> hi */
				else if (q == K - 1 || (q & 1)) max3h<0, 0, 1, 0>(V[q], d[q], V[q], left);  /* d lo, self lo, left hi -> lo */
				else max3h<1, 1, 0, 1>(V[q], d[q], V[q], left);
			}
			vprev = vleft;
		}
	}
	unsigned long long t1 = __builtin_amdgcn_s_memtime();
	unsigned sum = vprev;
	for (int q = 0; q < K; q++) sum += V[q];
	out[blockIdx.x * 256 + threadIdx.x] = sum;
	if (lane == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}
template <int K> int run3(const char *name, int bpc, unsigned *out, unsigned long long *cyc, int *seed)
{
	const int blocks = 256 * bpc;
	for (int it = 0; it < 2; it++) hipLaunchKernelGGL((k3<K>), dim3(blocks), dim3(256), 0, 0, out, cyc, seed);
	if (hipDeviceSynchronize() != hipSuccess) return 1;
	std::vector<unsigned long long> hc(blocks * 4);
	if (hipMemcpy(hc.data(), cyc, 8 * hc.size(), hipMemcpyDeviceToHost) != hipSuccess) return 1;
	double c = 0; for (auto v : hc) c += v; c /= hc.size();
	printf("%-34s waves/SIMD=%d  cycles/step/wave=%7.1f  SIMD-cycles/step=%6.1f  per cell=%5.2f\n", name, bpc, c / (ITERS * 16.0),
	       c / (ITERS * 16.0) / bpc, c / (ITERS * 16.0) / bpc / K);
	return 0;
}

template <int K, int MODE> __global__ __launch_bounds__(256) void k(unsigned *out, unsigned long long *cyc, const int *seed)
{
	const int lane = threadIdx.x & 63;
	unsigned V[K], vprev = lane, inj = seed[1];
	for (int q = 0; q < K; q++) V[q] = lane * q;
	unsigned pw[16][MODE == 1 ? K : (K + 1) / 2];
	for (int s = 0; s < 16; s++)
		for (int q = 0; q < (MODE == 1 ? K : (K + 1) / 2); q++) pw[s][q] = (lane * 7 + s * 13 + q + seed[2]) * 0x01030507u;
	unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < ITERS; it++) {
#pragma unroll
		for (int s = 0; s < 16; s++) {
			const unsigned vleft = __builtin_amdgcn_update_dpp(inj, V[K - 1], 0x111, 0xf, 0xf, false);
			unsigned d[K];
#pragma unroll
			for (int q = 0; q < K; q++) {
				const unsigned src = q ? V[q - 1] : vprev;
				if constexpr (MODE == 1) d[q] = pkadd(src, pw[s][q]);
				else d[q] = (q & 1) ? sdwa_pair<1>(pw[s][q / 2], src) : sdwa_pair<0>(pw[s][q / 2], src);
			}
			V[0] = pkmax(pkmax(d[0], V[0]), vleft);
#pragma unroll
			for (int q = 1; q < K; q++) V[q] = pkmax(pkmax(d[q], V[q]), V[q - 1]);
			vprev = vleft;
		}
	}
	unsigned long long t1 = __builtin_amdgcn_s_memtime();
	unsigned sum = vprev;
	for (int q = 0; q < K; q++) sum += V[q];
	out[blockIdx.x * 256 + threadIdx.x] = sum;
	if (lane == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}

template <int K, int MODE> int run(const char *name, int bpc, unsigned *out, unsigned long long *cyc, int *seed)
{
	const int blocks = 256 * bpc;
	for (int it = 0; it < 2; it++) hipLaunchKernelGGL((k<K, MODE>), dim3(blocks), dim3(256), 0, 0, out, cyc, seed);
	CHECK(hipDeviceSynchronize());
	std::vector<unsigned long long> hc(blocks * 4);
	CHECK(hipMemcpy(hc.data(), cyc, 8 * hc.size(), hipMemcpyDeviceToHost));
	double c = 0; for (auto v : hc) c += v; c /= hc.size();
	printf("%-34s waves/SIMD=%d  cycles/step/wave=%7.1f  SIMD-cycles/step=%6.1f  per cell=%5.2f\n", name, bpc, c / (ITERS * 16.0),
	       c / (ITERS * 16.0) / bpc, c / (ITERS * 16.0) / bpc / (2 * K));
	return 0;
}

int main()
{
	unsigned *out; int *seed; unsigned long long *cyc;
	CHECK(hipMalloc(&out, 4 * 256 * 8 * 256)); CHECK(hipMalloc(&cyc, 8 * 256 * 8 * 4)); CHECK(hipMalloc(&seed, 16));
	int hs[4] = {1, 2, 3, 4}; CHECK(hipMemcpy(seed, hs, 16, hipMemcpyHostToDevice));
	for (int b : {2, 4, 6, 8}) {
		run<7, 1>("K=7 pk_add s16 regs (P1)", b, out, cyc, seed);
		run<7, 2>("K=7 2xSDWA add s8 pairs (P2)", b, out, cyc, seed);
		run<4, 2>("K=4 2xSDWA add s8 pairs (P2)", b, out, cyc, seed);
		run3<7>("K=7 add_u32 + max3_i16 op_sel (P3)", b, out, cyc, seed);
	}
	return 0;
}
