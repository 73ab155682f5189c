/* sa_systolic.hip -- method dispatch of the systolic launches (kernels: sa_systolic_kernel.inc) */
#include "sa_internal.h"

hipError_t sa_launch_systolic_nw(int cls, const SaSysArgs &a, int tiles, hipStream_t s);
hipError_t sa_launch_systolic_ga(int cls, const SaSysArgs &a, int tiles, hipStream_t s);
hipError_t sa_launch_systolic_sw(int cls, const SaSysArgs &a, int tiles, hipStream_t s);

hipError_t sa_launch_systolic(int method, int cls, const SaSysArgs &a, int tiles, hipStream_t s)
{
	switch (method) {
	case SA_METHOD_NW:
		return sa_launch_systolic_nw(cls, a, tiles, s);
	case SA_METHOD_GA:
		return sa_launch_systolic_ga(cls, a, tiles, s);
	case SA_METHOD_SW:
		return sa_launch_systolic_sw(cls, a, tiles, s);
	default:
		return hipErrorInvalidValue;
	}
}

hipError_t sa_launch_systolic_pk_nw(int g, int k, const SaSysArgs &a, int tiles, hipStream_t s);
hipError_t sa_launch_systolic_pk_ga(int g, int k, const SaSysArgs &a, int tiles, hipStream_t s);
hipError_t sa_launch_systolic_pk_sw(int g, int k, const SaSysArgs &a, int tiles, hipStream_t s);

hipError_t sa_launch_systolic_pk(int method, int g, int k, const SaSysArgs &a, int tiles, hipStream_t s)
{
	switch (method) {
	case SA_METHOD_NW:
		return sa_launch_systolic_pk_nw(g, k, a, tiles, s);
	case SA_METHOD_GA:
		return sa_launch_systolic_pk_ga(g, k, a, tiles, s);
	case SA_METHOD_SW:
		return sa_launch_systolic_pk_sw(g, k, a, tiles, s);
	default:
		return hipErrorInvalidValue;
	}
}

hipError_t sa_warm_systolic_pk_nw(void);
hipError_t sa_warm_systolic_pk_ga(void);
hipError_t sa_warm_systolic_pk_sw(void);
hipError_t sa_warm_systolic_nw(void);
hipError_t sa_warm_systolic_ga(void);
hipError_t sa_warm_systolic_sw(void);
hipError_t sa_warm_generic(void);

hipError_t sa_warm_kernels(int method)
{
	hipError_t e = method == SA_METHOD_NW ? sa_warm_systolic_nw() : method == SA_METHOD_GA ? sa_warm_systolic_ga() : sa_warm_systolic_sw();
	if (e == hipSuccess && method == SA_METHOD_NW)
		e = sa_warm_systolic_pk_nw();
	if (e == hipSuccess && method == SA_METHOD_GA)
		e = sa_warm_systolic_pk_ga();
	if (e == hipSuccess && method == SA_METHOD_SW)
		e = sa_warm_systolic_pk_sw();
	return e != hipSuccess ? e : sa_warm_generic();
}
