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

#define SA_PK_DECL(M)                                                                                  \
	hipError_t sa_launch_systolic_pk_##M(int g, int klo, int f16, const SaSysArgs &a, int wgs, unsigned lds, hipStream_t s);   \
	hipError_t sa_launch_systolic_pk16_##M(int g, int klo, int f16, const SaSysArgs &a, int wgs, unsigned lds, hipStream_t s); \
	hipError_t sa_launch_systolic_pk16hi_##M(int g, int klo, int f16, const SaSysArgs &a, int wgs, unsigned lds, hipStream_t s); \
	hipError_t sa_warm_systolic_pk16hi_##M(void);                                                       \
	hipError_t sa_warm_systolic_pk_##M(void);                                                           \
	hipError_t sa_warm_systolic_pk16_##M(void);                                                         \
	hipError_t sa_warm_systolic_##M(void);
SA_PK_DECL(nw)
SA_PK_DECL(ga)
SA_PK_DECL(sw)
#undef SA_PK_DECL
hipError_t sa_warm_generic(void);

hipError_t sa_launch_systolic_pk(int method, int g, int klo, int f16, const SaSysArgs &a, int wgs, unsigned lds, hipStream_t s)
{
	switch (method) {
	case SA_METHOD_NW:
		return g == 8 ? sa_launch_systolic_pk_nw(g, klo, f16, a, wgs, lds, s) : (klo >= 45 ? sa_launch_systolic_pk16hi_nw(g, klo, f16, a, wgs, lds, s) : sa_launch_systolic_pk16_nw(g, klo, f16, a, wgs, lds, s));
	case SA_METHOD_GA:
		return g == 8 ? sa_launch_systolic_pk_ga(g, klo, f16, a, wgs, lds, s) : (klo >= 45 ? sa_launch_systolic_pk16hi_ga(g, klo, f16, a, wgs, lds, s) : sa_launch_systolic_pk16_ga(g, klo, f16, a, wgs, lds, s));
	case SA_METHOD_SW:
		return g == 8 ? sa_launch_systolic_pk_sw(g, klo, f16, a, wgs, lds, s) : (klo >= 45 ? sa_launch_systolic_pk16hi_sw(g, klo, f16, a, wgs, lds, s) : sa_launch_systolic_pk16_sw(g, klo, f16, a, wgs, lds, s));
	default:
		return hipErrorInvalidValue;
	}
}

/* Loads the code objects of the kernel families a store will use (SA_WARM_*), outside any timed phase.  A family that is
 * not warmed still works: its code object loads at its first launch. */
hipError_t sa_warm_kernels(int method, int families)
{
	hipError_t e = sa_warm_generic(); /* pair-per-wave fallback, expand, widen, filter: small */
#define SA_WARM(M)                                                      \
	do {                                                            \
		if (e == hipSuccess && (families & SA_WARM_S32))        \
			e = sa_warm_systolic_##M();                     \
		if (e == hipSuccess && (families & SA_WARM_PK8))        \
			e = sa_warm_systolic_pk_##M();                  \
		if (e == hipSuccess && (families & SA_WARM_PK16))       \
			e = sa_warm_systolic_pk16_##M();                \
		if (e == hipSuccess && (families & SA_WARM_PK16HI))     \
			e = sa_warm_systolic_pk16hi_##M();              \
	} while (0)
	if (method == SA_METHOD_NW)
		SA_WARM(nw);
	else if (method == SA_METHOD_GA)
		SA_WARM(ga);
	else
		SA_WARM(sw);
#undef SA_WARM
	return e;
}
