/* sa_internal.h -- internal declarations shared by the translation units of libseqalign_hip.so */
#ifndef SA_INTERNAL_H
#define SA_INTERNAL_H

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdarg>
#include <string>
#include <vector>


#include "sa_shapes.h"

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
	int32_t *host_out;            /* != nullptr: the scores also go to host_out[q] (s32), see SaSysArgs::host_out */
	const int32_t *sub;           /* s32[24*24] substitution matrix in HBM                  */
	int32_t gap_pen, gap_opn, gap_ext;
	int64_t start, count;         /* packed pair range                                      */
	int32_t *out;                 /* out[q] = score of pair start+q                         */
	int32_t out16;                /* != 0: out is an int16_t array (exchange format)        */
	int32_t *scratch;             /* per-wave strip boundary columns (M and X)              */
	int64_t scratch_stride;       /* ints per wave = 2*(max_len+2)                          */
};

struct SaSysArgs {
	const uint8_t *codes;    /* encoded store, tight layout: sequence k at off[k], terminator after it */
	const int32_t *off;      /* num+1 offsets; len_k = off[k+1]-off[k]-1                              */
	const int8_t *sub8;      /* s8[24*24] substitution matrix                                         */
	const int32_t *jlist;    /* columns (ascending) handled by this launch                            */
	const int32_t *tprefix;  /* wave-tiles before jlist[k]; ncols+1 entries                           */
	int32_t ncols, num;
	int64_t start, end;      /* packed pair range [start, end) being computed                         */
	int32_t *out;            /* out[p - start]                                                        */
	int32_t out16;           /* != 0: out is an int16_t array (scores proven to fit; exchange format) */
	int32_t pconst;          /* constant folded into the profile: NW -2g, GA -e-o, SW -o              */
	int32_t q;               /* GA: o - e (<= 0); else 0                                              */
	int32_t gap_g, gap_o, gap_e;
	int32_t delta;           /* baseline raise per sequence                                           */
	int32_t chunk;           /* sequences per group stream of a wave-tile, 1..SA_SYS_CHUNK            */
	int32_t *long_scratch;   /* strip-mined launch: per wave 2 lines of long_stride/2 ints            */
	int64_t long_stride;     /* ints per wave (>= 2 * longest row stream of a wave)                   */
	int32_t pk_base;         /* packed kernels: the constant baseline BASE                                     */
	/* packed kernels: arranged copies of the store (sa_plan.cpp: sa_arrange_rows), largest block first; rows = 0: none */
	const SaArranged *lvp;   /* packed kernels: the SA_PK_SORT_LEVELS arranged copies offered to the class of the tile being run */
	int32_t npart;           /* packed kernels: partial tiles, their column pairs listed behind tprefix         */
	int32_t out_nt;          /* packed kernels: out is host memory, store non-temporally                        */
	int32_t pk_f16;          /* packed kernels: the class's values fit SA_PK_F16_MAX (three-way f16 maxima)     */
	/* tile-interleaved sharding (sa_ctx_align_share): this launch runs the tiles tlist[0..nlocal) of the class's tile
	 * list and stores the scores of tile t densely at out[dense_off[t] ...] in tile order -- packed kernels: two runs of
	 * SA_SHARE_PAD(rows) elements (column A, column B) in POSITION order of the row stream; s32 kernels: the tile's rows.
	 * tlist == nullptr: every tile of the list; dense_off == nullptr: packed order, out[p - start]. */
	const int32_t *tlist;
	const int64_t *dense_off;
	int32_t *host_out;        /* dense shares only: page-locked packed host matrix (device-visible address of element
	                           * `start`) that receives the same scores in packed order, or nullptr          */
	int32_t nlocal;
	const SaPkClassArgs *pkc; /* packed bundle launch: its classes; nlocal = tiles of the whole launch */
	const uint32_t *ulist;    /* ... and its tiles in walking order: two words per tile, the code (SA_PK_UTILE_BITS)
	                           * and the tile's column pair in its class                                      */
	int32_t tile_pair;        /* (kernel-internal: the column pair of the tile being run)                     */
	int32_t npkc;
	unsigned *counter;       /* [0] next unclaimed tile of this launch, [1] workgroups that have left; both zero before
	                          * the launch and put back to zero by its last workgroup                           */
	int32_t rotate_prio;        /* packed bundle: rotate the waves' issue priority tile by tile (sa_systolic_pk.inc)     */
	int32_t stagger;            /* packed bundle: sleep periods (8128 clocks) per wave slot before the first tile      */
	int32_t stamp_u;            /* diagnostics: launch-tile index of the tile being run (packed bundle)                */
	unsigned long long *stamps; /* diagnostics only (SA_HIP_STAMPS=1): per wave-tile {cycles, 100MHz ticks,
	                             * steps} of the main loop; nullptr in production                        */
};

hipError_t sa_launch_place(const SaPlaceSeg *segs, int32_t nsegs, const void *shares, int elem16, int32_t *packed, hipStream_t s);

/* `workgroups` persistent workgroups pull the launch's wave-tiles from a.counter */
hipError_t sa_launch_systolic(int method, int cls, const SaSysArgs &a, int workgroups, hipStream_t s);
/* packed-u16 kernels: the bundle of classes a.pkc[0 .. a.npkc) (all of lane-group width g, K in [klo, klo + SA_PK_BUNDLE),
 * all three-way (f16) or not); lds_bytes = sa_pk_lds_bytes(method, g, largest K of the launch) */
hipError_t sa_launch_systolic_pk(int method, int g, int klo, int f16, const SaSysArgs &a, int workgroups, unsigned lds_bytes,
				 hipStream_t s);
/* forces the code objects of the method's kernels onto the current device (module load outside any timed phase) */
enum : int { SA_WARM_S32 = 1, SA_WARM_PK8 = 2, SA_WARM_PK16 = 4, SA_WARM_PK16HI = 8 }; /* kernel families (one code object each per method) */
hipError_t sa_warm_kernels(int method, int families);

/* ---- launchers implemented in the .hip files ---------------------------- */
hipError_t sa_launch_generic(int method, const SaGenericArgs &a, int blocks, hipStream_t s);
const char *sa_generic_kernel_name(int method);

/* similarity filter relation (sa_filter.hip) */
long long sa_filter_row_offset(long long j);
hipError_t sa_launch_filter_relation(const uint8_t *codes, const int32_t *off, int32_t num, float threshold,
				      unsigned long long *rel, int32_t jt0, int32_t tile_rows, hipStream_t s);

hipError_t sa_launch_expand_full(const int32_t *packed, int32_t *full, int32_t num, hipStream_t s);
hipError_t sa_launch_expand_shell(const int32_t *packed, int64_t pbase, int32_t *full, int32_t num, int32_t ja, int32_t jb,
				  hipStream_t s);
hipError_t sa_launch_widen16(const int16_t *src, int32_t *dst, int64_t count, hipStream_t s);

#endif /* SA_INTERNAL_H */
