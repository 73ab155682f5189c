// Microbenchmark (development tool): what do a process's first HIP calls cost on this box, without this repository's
// code objects in the process?  (the first sa_ctx_create spends 100-150 ms in its first hipMemcpy)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k(int *p) { p[threadIdx.x] = threadIdx.x; }
int main()
{
	auto t0 = std::chrono::steady_clock::now();
	auto ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
	int n = 0;
	hipGetDeviceCount(&n);
	hipSetDevice(0);
	printf("device ready at %.1f ms\n", ms());
	void *d = nullptr;
	hipMalloc(&d, 1 << 20);
	printf("hipMalloc at %.1f ms\n", ms());
	std::vector<char> h(1 << 20, 1);
	hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
	printf("first hipMemcpy (1 MB, pageable) at %.1f ms\n", ms());
	hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
	printf("second hipMemcpy at %.1f ms\n", ms());
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, (int *)d);
	hipDeviceSynchronize();
	printf("first kernel at %.1f ms\n", ms());
	return 0;
}
