// Microbenchmark (development tool): sustained issue cost of VALU forms on gfx950.
// B blocks of 256 threads per CU (=> B waves per SIMD), every wave runs REP x 4 x 8 copies of one
// instruction on 8 independent registers; cycles from s_memtime, clock from s_memrealtime (100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int REP = 1500;

#define OP8(ASM, ...) _Pragma("unroll") for (int q = 0; q < 8; q++) asm volatile(ASM : "+v"(v[q]) : __VA_ARGS__)

template <int MODE> __global__ __launch_bounds__(256) void k(int *out, unsigned long long *cyc, unsigned long long *rt, int seed)
{
	int v[8], w = threadIdx.x * 0x01010101 + seed, acc = seed, z = seed * 3;
	for (int q = 0; q < 8; q++) v[q] = threadIdx.x + q * seed;
	unsigned long long t0 = __builtin_amdgcn_s_memtime();
	unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
	for (int r = 0; r < REP; r++) {
#pragma unroll
		for (int u = 0; u < 4; u++) {
			if (MODE == 0) OP8("v_add_u32_e32 %0, %1, %0", "v"(w));
			if (MODE == 1) OP8("v_add_u32_sdwa %0, sext(%1), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD", "v"(w));
			if (MODE == 2) OP8("v_max3_i32 %0, %1, %0, %2", "v"(w), "v"(acc));
			if (MODE == 3) OP8("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", "v"(w));
			if (MODE == 4) OP8("v_max_i32_e32 %0, %1, %0", "v"(w));
			if (MODE == 5) OP8("v_max_u32_e32 %0, %1, %0", "v"(w));
			if (MODE == 6) OP8("v_max_f32_e32 %0, %1, %0", "v"(w));
			if (MODE == 7) OP8("v_max3_f32 %0, %1, %0, %2", "v"(w), "v"(acc));
			if (MODE == 8) OP8("v_add_f32_e32 %0, %1, %0", "v"(w));
			if (MODE == 9) OP8("v_and_b32_e32 %0, %1, %0", "v"(w));
			if (MODE == 10) OP8("v_sub_u32_e32 %0, %1, %0", "v"(w));
			if (MODE == 11) OP8("v_cndmask_b32_e32 %0, %1, %0, vcc", "v"(w));
			if (MODE == 12) OP8("v_lshl_add_u32 %0, %1, 2, %0", "v"(w));
			if (MODE == 13) OP8("v_add3_u32 %0, %1, %0, %2", "v"(w), "v"(acc));
			if (MODE == 14) OP8("v_mad_i32_i24 %0, %1, %2, %0", "v"(w), "v"(acc));
			if (MODE == 15) OP8("v_med3_i32 %0, %1, %0, %2", "v"(w), "v"(acc));
			if (MODE == 16) OP8("v_pk_add_i16 %0, %1, %0", "v"(w));
			if (MODE == 17) OP8("v_pk_max_i16 %0, %1, %0", "v"(w));
			if (MODE == 18) OP8("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf", "v"(w));
			if (MODE == 19) OP8("v_cvt_f32_ubyte1_e32 %0, %1", "v"(w));
			if (MODE == 20) OP8("v_dot4c_i32_i8_e32 %0, %1, %2", "v"(w), "v"(acc));
			if (MODE == 21) OP8("v_max_i32_e64 %0, %1, %0", "v"(w));
			if (MODE == 22) OP8("v_add_u32_e64 %0, %1, %0", "v"(w));
			if (MODE == 23) OP8("v_max_i32_e32 %0, 7, %0", "v"(w));
			if (MODE == 24) OP8("v_max_i16_e32 %0, %1, %0", "v"(w));
			if (MODE == 25) OP8("v_min_i32_e32 %0, %1, %0", "v"(w));
			if (MODE == 26) OP8("v_or_b32_e32 %0, %1, %0", "v"(w));
			if (MODE == 27) OP8("v_xor_b32_e32 %0, %1, %0", "v"(w));
			if (MODE == 28) OP8("v_add_co_u32_e32 %0, vcc, %1, %0", "v"(w));
			if (MODE == 30) OP8("v_fma_f32 %0, %1, %2, %0", "v"(w), "v"(acc));
			if (MODE == 31) OP8("v_max_f16_e32 %0, %1, %0", "v"(w));
			if (MODE == 32) OP8("v_pk_max_f16 %0, %1, %0", "v"(w));
			if (MODE == 33) OP8("v_mov_b32_e32 %0, %1", "v"(w));
			if (MODE == 34) OP8("v_bfe_i32 %0, %1, 8, 8", "v"(w));
			if (MODE == 35) OP8("v_perm_b32 %0, %1, %0, %2", "v"(w), "v"(acc));
			if (MODE == 36) OP8("v_cmp_lt_i32_e32 vcc, %1, %0\n\tv_cndmask_b32_e32 %0, %1, %0, vcc", "v"(w) : "vcc");
			if (MODE == 37) OP8("v_cmp_lt_i32_e64 s[20:21], %1, %0\n\tv_cndmask_b32_e64 %0, %1, %0, s[20:21]", "v"(w) : "s20", "s21");
			if (MODE == 38) OP8("v_cmp_lt_i32_e32 vcc, %1, %0", "v"(w) : "vcc");
			if (MODE == 39) OP8("v_sub_u32_e32 %0, %1, %0\n\tv_ashrrev_i32_e32 %0, 31, %0\n\tv_and_b32_e32 %0, %1, %0\n\tv_max_u32_e32 %0, %1, %0", "v"(w));
		}
	}
	unsigned long long t1 = __builtin_amdgcn_s_memtime();
	unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
	int s = z;
	for (int q = 0; q < 8; q++) s += v[q];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if ((threadIdx.x & 63) == 0) {
		cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
		rt[blockIdx.x * 4 + threadIdx.x / 64] = r1 - r0;
	}
}

static int *g_out; static unsigned long long *g_cyc, *g_rt;

/* calibration accumulators: s_memtime ticks, s_memrealtime ticks and host-timed nanoseconds of the same kernels */
static double g_ticks, g_rticks, g_wall_ns;

template <int MODE> int run(const char *name, int bpc)
{
	const int blocks = 256 * bpc;
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, g_out, g_cyc, g_rt, 3); /* warm */
	CHECK(hipEventRecord(e0, 0));
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, g_out, g_cyc, g_rt, 3);
	CHECK(hipEventRecord(e1, 0));
	CHECK(hipDeviceSynchronize());
	float ms = 0.f;
	CHECK(hipEventElapsedTime(&ms, e0, e1));
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	std::vector<unsigned long long> hc(blocks * 4), hr(blocks * 4);
	CHECK(hipMemcpy(hc.data(), g_cyc, 8 * hc.size(), hipMemcpyDeviceToHost));
	CHECK(hipMemcpy(hr.data(), g_rt, 8 * hr.size(), hipMemcpyDeviceToHost));
	double c = 0, r = 0;
	for (size_t i = 0; i < hc.size(); i++) { c += hc[i]; r += hr[i]; }
	c /= hc.size(); r /= hr.size();
	const double insts = (double)REP * 4 * 8;
	/* wall: every SIMD executes insts*bpc wave-instructions during the launch (all blocks co-resident: 256 CUs x bpc
	 * blocks, one wave per SIMD each); the launch also contains ~10 us of ramp, visible as wall > ticks/clock */
	printf("  %5.2f (%5.3f ns)", c / insts / bpc, (double)ms * 1e6 / (insts * bpc));
	g_ticks += c; g_rticks += r; g_wall_ns += (double)ms * 1e6;
	(void)name;
	return 0;
}

#define ROW(M, NAME) do { printf("\n%-28s", NAME); for (int b : {1, 2, 4, 8}) run<M>("", b); } while (0)

int main()
{
	CHECK(hipMalloc(&g_out, sizeof(int) * 256 * 8 * 256));
	CHECK(hipMalloc(&g_cyc, 8 * 256 * 8 * 4));
	CHECK(hipMalloc(&g_rt, 8 * 256 * 8 * 4));
	printf("Per wave64 instruction, all waves of the SIMD together: s_memtime ticks (host-timed ns of the whole launch per instruction); columns = 1,2,4,8 waves/SIMD");
	ROW(0, "v_add_u32_e32"); ROW(22, "v_add_u32_e64"); ROW(10, "v_sub_u32_e32"); ROW(28, "v_add_co_u32_e32");
	ROW(9, "v_and_b32"); ROW(26, "v_or_b32"); ROW(27, "v_xor_b32"); ROW(33, "v_mov_b32");
	ROW(4, "v_max_i32_e32"); ROW(21, "v_max_i32_e64"); ROW(23, "v_max_i32 inline const"); ROW(25, "v_min_i32"); ROW(5, "v_max_u32");
	ROW(24, "v_max_i16"); ROW(6, "v_max_f32"); ROW(31, "v_max_f16"); ROW(8, "v_add_f32"); ROW(30, "v_fma_f32"); 
	ROW(2, "v_max3_i32"); ROW(7, "v_max3_f32"); ROW(15, "v_med3_i32"); ROW(13, "v_add3_u32"); ROW(12, "v_lshl_add_u32"); ROW(14, "v_mad_i32_i24");
	ROW(1, "v_add_u32_sdwa"); ROW(18, "v_add_u32_dpp"); ROW(3, "v_mov_b32_dpp"); ROW(11, "v_cndmask_b32");
	ROW(16, "v_pk_add_i16"); ROW(17, "v_pk_max_i16"); ROW(32, "v_pk_max_f16"); ROW(19, "v_cvt_f32_ubyte1"); ROW(20, "v_dot4c_i32_i8");
	ROW(34, "v_bfe_i32"); ROW(35, "v_perm_b32");
	ROW(36, "v_cmp + v_cndmask (vcc), per PAIR"); ROW(37, "v_cmp + v_cndmask (sgpr pair), per PAIR"); ROW(38, "v_cmp_lt_i32 alone");
	ROW(39, "sub+ashr+and+max select, per FOUR");
	printf("\n\ncalibration over all launches above: s_memtime %.1f ticks/us of s_memrealtime (100 MHz) = %.0f MHz;  "
	       "s_memtime %.1f ticks per host-timed us (includes launch ramp, lower bound of the clock);  "
	       "=> one s_memtime tick is one shader clock at the frequency the chip held (~%.2f GHz), "
	       "a column entry of 2.00 ticks = %.3f ns per wave64 instruction and SIMD\n",
	       g_ticks / g_rticks * 100.0, g_ticks / g_rticks * 100.0, g_ticks / (g_wall_ns * 1e-3),
	       g_ticks / g_rticks * 0.1, 2.0 / (g_ticks / g_rticks * 0.1));
	return 0;
}
