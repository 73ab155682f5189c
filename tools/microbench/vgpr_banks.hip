// Microbenchmark (development tool): do VGPR bank conflicts (register index mod 4) cost issue cycles on gfx950?
// Same instruction stream, only the physical source registers differ.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int REP = 2000;
#define R8(X) X X X X X X X X
template <int MODE> __global__ __launch_bounds__(256) void k(int *out, unsigned long long *cyc, int seed)
{
	unsigned long long t0, t1;
	asm volatile("v_mov_b32 v20, %0\n v_mov_b32 v21, %0\n v_mov_b32 v22, %0\n v_mov_b32 v23, %0\n v_mov_b32 v24, %0\n v_mov_b32 v25, %0\n"
		     "v_mov_b32 v26, %0\n v_mov_b32 v27, %0\n v_mov_b32 v28, %0\n v_mov_b32 v29, %0\n v_mov_b32 v30, %0\n v_mov_b32 v31, %0\n v_mov_b32 v32, %0\n v_mov_b32 v36, %0\n"
		     :: "v"(seed + (int)threadIdx.x) : "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v36");
	t0 = __builtin_amdgcn_s_memtime();
	for (int r = 0; r < REP; r++) {
		if (MODE == 0) /* max3: three sources in three banks */
			asm volatile(R8("v_max3_i32 v20, v21, v22, v23\n v_max3_i32 v24, v25, v26, v27\n v_max3_i32 v28, v29, v30, v31\n v_max3_i32 v21, v22, v23, v24\n")
				     ::: "v20","v21","v24","v28");
		if (MODE == 1) /* max3: two sources share a bank */
			asm volatile(R8("v_max3_i32 v20, v21, v25, v23\n v_max3_i32 v24, v22, v26, v27\n v_max3_i32 v28, v29, v30, v26\n v_max3_i32 v21, v22, v23, v27\n")
				     ::: "v20","v21","v24","v28");
		if (MODE == 2) /* max3: all three sources in one bank */
			asm volatile(R8("v_max3_i32 v20, v21, v25, v29\n v_max3_i32 v24, v22, v26, v30\n v_max3_i32 v28, v23, v27, v31\n v_max3_i32 v21, v24, v28, v32\n")
				     ::: "v20","v21","v24","v28");
		if (MODE == 3) /* sdwa add: two banks */
			asm volatile(R8("v_add_u32_sdwa v20, v21, sext(v22) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
					"v_add_u32_sdwa v24, v25, sext(v22) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"
					"v_add_u32_sdwa v28, v29, sext(v22) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n"
					"v_add_u32_sdwa v21, v23, sext(v22) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n")
				     ::: "v20","v21","v24","v28");
		if (MODE == 4) /* sdwa add: same bank */
			asm volatile(R8("v_add_u32_sdwa v20, v26, sext(v22) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
					"v_add_u32_sdwa v24, v30, sext(v22) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"
					"v_add_u32_sdwa v28, v26, sext(v22) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n"
					"v_add_u32_sdwa v21, v30, sext(v22) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n")
				     ::: "v20","v21","v24","v28");
		if (MODE == 5) /* dependent chain like the kernel: add then max3 on its result and the previous max3 */
			asm volatile(R8("v_add_u32_sdwa v20, v21, sext(v22) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
					"v_max3_i32 v24, v20, v25, v28\n"
					"v_add_u32_sdwa v20, v25, sext(v22) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"
					"v_max3_i32 v28, v20, v29, v24\n")
				     ::: "v20","v24","v28");
		if (MODE == 6) /* same with the adds hoisted (independent), max3 chain behind */
			asm volatile(R8("v_add_u32_sdwa v20, v21, sext(v22) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
					"v_add_u32_sdwa v36, v25, sext(v22) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"
					"v_max3_i32 v24, v20, v25, v28\n"
					"v_max3_i32 v28, v36, v29, v24\n")
				     ::: "v20","v24","v28","v36");
	}
	t1 = __builtin_amdgcn_s_memtime();
	int r;
	asm volatile("v_add_u32 %0, v20, v24\n v_add_u32 %0, %0, v28\n v_add_u32 %0, %0, v21" : "=v"(r));
	out[blockIdx.x * 256 + threadIdx.x] = r;
	if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}
template <int MODE> int run(const char *name, int *out, unsigned long long *cyc)
{
	printf("%-52s", name);
	for (int bpc : {1, 2, 4, 8}) {
		const int blocks = 256 * bpc;
		for (int it = 0; it < 2; it++) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, cyc, it);
		CHECK(hipDeviceSynchronize());
		std::vector<unsigned long long> hc(blocks * 4);
		CHECK(hipMemcpy(hc.data(), cyc, 8 * hc.size(), hipMemcpyDeviceToHost));
		double c = 0; for (auto v : hc) c += v; c /= hc.size();
		printf(" %7.2f", c / (REP * 32.0) / bpc);
	}
	printf("\n");
	return 0;
}
int main()
{
	int *out; unsigned long long *cyc;
	CHECK(hipMalloc(&out, 4 * 256 * 8 * 256)); CHECK(hipMalloc(&cyc, 8 * 256 * 8 * 4));
	printf("SIMD cycles per wave64 instruction at 1,2,4,8 waves/SIMD\n");
	run<0>("v_max3_i32, sources in 3 banks", out, cyc);
	run<1>("v_max3_i32, two sources share a bank", out, cyc);
	run<2>("v_max3_i32, three sources in one bank", out, cyc);
	run<3>("v_add_u32_sdwa, sources in 2 banks", out, cyc);
	run<4>("v_add_u32_sdwa, sources in one bank", out, cyc);
	run<5>("sdwa add -> dependent max3 chain (kernel order)", out, cyc);
	run<6>("adds hoisted, max3 chain", out, cyc);
	return 0;
}
