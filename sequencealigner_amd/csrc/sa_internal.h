/* sa_internal.h -- internal declarations shared by the translation units of libseqalign_hip.so */
#ifndef SA_INTERNAL_H
#define SA_INTERNAL_H

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdarg>
#include <string>
#include <vector>

#include "../../include/seqalign_hip.h"

/* residue codes of the encoded sequence store (device side):
 * 0..23 = index from sa_scoring.lut, SA_CODE_SEP = the NUL terminator of every sequence,
 * SA_CODE_NOP = pipeline bubble fed by the systolic kernels. */
enum : int { SA_CODE_SEP = 24, SA_CODE_NOP = 25, SA_CODE_ROWS = 26 };

void sa_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

/* `onfail` runs in the caller's scope (so `break`/`return`/`goto` act on the caller's loop). */
#define SA_HIP_CHECK(call, onfail)                                                  \
	if (hipError_t err__ = (call); err__ != hipSuccess) {                       \
		sa_set_error("%s failed: %s", #call, hipGetErrorString(err__));      \
		onfail;                                                              \
	} else                                                                      \
		(void)0

/* ---- kernel argument blocks --------------------------------------------- */
struct SaSeqStore {
	const uint8_t *codes;         /* encoded blob, same offsets as the host blob            */
	const struct sa_meta *meta;   /* N x {off,len}                                          */
	int32_t num;
};

struct SaGenericArgs {
	SaSeqStore st;
	const int32_t *sub;           /* s32[24*24] substitution matrix in HBM                  */
	int32_t gap_pen, gap_opn, gap_ext;
	int64_t start, count;         /* packed pair range                                      */
	int32_t *out;                 /* out[q] = score of pair start+q                         */
	int32_t *scratch;             /* per-wave strip boundary columns (M and X)              */
	int64_t scratch_stride;       /* ints per wave = 2*(max_len+2)                          */
};

/* ---- systolic streaming kernels (sa_systolic.hip) ------------------------ */
#define SA_SYS_CHUNK 64 /* sequences streamed per lane group and wave-tile */
#ifndef SA_SYS_WPB
#define SA_SYS_WPB 1    /* waves per workgroup (one wave-tile each)            */
#endif
/* kernel classes: (index, lanes per group G, columns per lane K); column budget W = G*K */
#define SA_SYS_CLASS_LIST(X)                                                                        \
	X(0, 16, 1) X(1, 16, 2) X(2, 16, 3)                                                         \
	X(3, 16, 4) X(4, 16, 5) X(5, 16, 6) X(6, 16, 7) X(7, 16, 8)                                 \
	X(8, 32, 5) X(9, 32, 6) X(10, 32, 7) X(11, 32, 8)                                           \
	X(12, 64, 5) X(13, 64, 6) X(14, 64, 7) X(15, 64, 8)                                         \
	X(16, 64, 10) X(17, 64, 12) X(18, 64, 14) X(19, 64, 16)
struct SaSysClass {
	int G, K;
};
static const SaSysClass SA_SYS_CLASSES[] = {
#define SA_SYS_ENTRY(IDX, G_, K_) { G_, K_ },
	SA_SYS_CLASS_LIST(SA_SYS_ENTRY)
#undef SA_SYS_ENTRY
};
enum : int { SA_SYS_NCLASSES = (int)(sizeof(SA_SYS_CLASSES) / sizeof(SA_SYS_CLASSES[0])),
	     SA_SYS_CLASS_LONG = SA_SYS_NCLASSES, /* strip-mined launch of the widest class (G=64, K=16) */
	     SA_SYS_LONG_W = 1024 };

struct SaSysArgs {
	const uint8_t *codes;    /* encoded store, tight layout: sequence k at off[k], terminator after it */
	const int32_t *off;      /* num+1 offsets; len_k = off[k+1]-off[k]-1                              */
	const int8_t *sub8;      /* s8[24*24] substitution matrix                                         */
	const int32_t *jlist;    /* columns (ascending) handled by this launch                            */
	const int32_t *tprefix;  /* wave-tiles before jlist[k]; ncols+1 entries                           */
	int32_t ncols, num;
	int64_t start, end;      /* packed pair range [start, end) being computed                         */
	int32_t *out;            /* out[p - start]                                                        */
	int32_t pconst;          /* constant folded into the profile: NW -2g, GA -e-o, SW -o              */
	int32_t q;               /* GA: o - e (<= 0); else 0                                              */
	int32_t gap_g, gap_o, gap_e;
	int32_t delta;           /* baseline raise per sequence                                           */
	int32_t chunk;           /* sequences per group stream of a wave-tile, 1..SA_SYS_CHUNK            */
	int32_t *long_scratch;   /* strip-mined launch: per workgroup 2 lines of long_stride/2 ints       */
	int64_t long_stride;     /* ints per workgroup (>= 2 * longest row stream of a tile)              */
	unsigned *counter;       /* next unclaimed wave-tile of this launch (zeroed by the host)            */
	unsigned long long *stamps; /* diagnostics only (SA_HIP_STAMPS=1): per wave-tile {cycles, 100MHz ticks,
	                             * steps} of the main loop; nullptr in production                        */
};

/* `workgroups` persistent workgroups pull the launch's wave-tiles from a.counter */
hipError_t sa_launch_systolic(int method, int cls, const SaSysArgs &a, int workgroups, hipStream_t s);

/* ---- launchers implemented in the .hip files ---------------------------- */
hipError_t sa_launch_generic(int method, const SaGenericArgs &a, int blocks, hipStream_t s);
const char *sa_generic_kernel_name(int method);

/* similarity filter relation (sa_filter.hip) */
long long sa_filter_row_offset(long long j);
hipError_t sa_launch_filter_relation(const uint8_t *codes, const int32_t *off, int32_t num, float threshold,
				      unsigned long long *rel, int32_t jt0, int32_t tile_rows, hipStream_t s);

hipError_t sa_launch_expand_full(const int32_t *packed, int32_t *full, int32_t num, hipStream_t s);

#endif /* SA_INTERNAL_H */
