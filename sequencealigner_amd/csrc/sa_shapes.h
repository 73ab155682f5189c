/* sa_shapes.h -- shapes and constants the HOST planner and the kernels must agree on: kernel classes, tile geometry,
 * arranged-copy levels, the layout of a dense share.  Pure C++ (no HIP include), so that the planner (sa_plan.cpp) also
 * builds with a plain host compiler -- tests/ runs it under AddressSanitizer / UBSan. */
#ifndef SA_SHAPES_H
#define SA_SHAPES_H

#include <cstdint>

#include "../../include/seqalign_hip.h"

#if defined(__HIPCC__)
#define SA_HD __host__ __device__
#else
#define SA_HD
#endif

/* residue codes of the encoded sequence store (device side):
 * 0..23 = index from sa_scoring.lut, SA_CODE_SEP = the NUL terminator of every sequence,
 * SA_CODE_NOP = pipeline bubble fed by the systolic kernels. */
enum : int { SA_CODE_SEP = 24, SA_CODE_NOP = 25, SA_CODE_ROWS = 26 };

void sa_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));


/* ---- systolic streaming kernels (sa_systolic.hip) ------------------------ */
#define SA_SYS_CHUNK 32 /* sequences streamed per lane group and wave-tile (at most) */
/* waves per workgroup of a class launch.  The wide groups need a large query profile (26 rows x W bytes: 13 KB
 * at W = 512, 27 KB at W = 1024), so four waves share one: a workgroup-tile is one column against 4 x (64/G) row
 * streams, each wave streaming its own groups (the strip-mined launch included: scratch lines are per wave). */
#define SA_SYS_WPB(G, LONG) ((G) >= 32 ? 4 : 1)
/* kernel classes: (index, lanes per group G, columns per lane K); column budget W = G*K, ascending.
 * Every per-step cost of the wave (token, profile read, shifts, event test) is shared by the lane's K
 * columns, so the classes use the narrowest group that reaches W with K <= 16: W = 8..128 in steps of 8
 * (G = 8: two interleaved groups per DPP row), 144..256 (G = 16), 288..512 (G = 32), 576..1024 (G = 64). */
#define SA_SYS_CLASS_LIST(X) \
	X(0, 8, 1) X(1, 8, 2) X(2, 8, 3) X(3, 8, 4) \
	X(4, 8, 5) X(5, 8, 6) X(6, 8, 7) X(7, 8, 8) \
	X(8, 8, 9) X(9, 8, 10) X(10, 8, 11) X(11, 8, 12) \
	X(12, 8, 13) X(13, 8, 14) X(14, 8, 15) X(15, 8, 16) \
	X(16, 16, 9) X(17, 16, 10) X(18, 16, 11) X(19, 16, 12) \
	X(20, 16, 13) X(21, 16, 14) X(22, 16, 15) X(23, 16, 16) \
	X(24, 32, 9) X(25, 32, 10) X(26, 32, 11) X(27, 32, 12) \
	X(28, 32, 13) X(29, 32, 14) X(30, 32, 15) X(31, 32, 16) \
	X(32, 64, 9) X(33, 64, 10) X(34, 64, 11) X(35, 64, 12) \
	X(36, 64, 13) X(37, 64, 14) X(38, 64, 15) X(39, 64, 16)
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

/* packed-u16 kernels (sa_systolic_pk.inc): 8-lane groups, K = 1..SA_PK_KMAX columns per lane (W = 8 K <= 192), two
 * column sequences per register, SA_PK_WPB waves per workgroup sharing the column pair's profile */
#define SA_PK_WPB 4
/* ... and EIGHT waves per workgroup (round 4)
 *  - where the profile is so large (16-lane groups, K >= 45: > 80 KB) that only one workgroup fits a CU: eight waves
 *    sharing it put two waves on every SIMD instead of one (600-1000 aa NW 19.7 -> 24.6 TCUPS).  The split
 *    coincides with the translation units (sa_systolic_pk16hi_*.hip: K >= 45) and with the bundle blocks (KLO = 45, 53, 61);
 *  - for NW with 8-lane groups and K = 17..24 (the bundle KLO = 17): 53 KB per four-wave workgroup are three workgroups
 *    = three waves per SIMD, and NW wants four (three cost it 8 %); eight waves around one profile are 67 KB = two
 *    workgroups = four waves per SIMD.  NW only: its 119 VGPRs allow four waves, Gotoh's 140 and SW's 149 do not. */
#define SA_PK_K16_WIDE 45
SA_HD constexpr int sa_pk_wpb(int method, int g, int k)
{
	return (g == 16 && k >= SA_PK_K16_WIDE) || (method == SA_METHOD_NW && g == 8 && k >= 17) ? 2 * SA_PK_WPB : SA_PK_WPB;
}
#define SA_PK_F16_MAX 0x7bff /* largest value of the 8-lane packed kernels: the largest finite f16 bit pattern */
#define SA_PK_ROWS_OWN_BLOCK 1024 /* tiles of at least this many rows are their own arranged block */
#define SA_PK_SORT_LEVELS 4 /* block sizes SA_PK_SORT_ROWS >> level offered to a launch whose tiles are smaller */
#define SA_PK_SORT_ROWS 2048 /* rows per arranged block of the row store when a tile is smaller (sa_plan.cpp: sa_arrange_rows) */
#define SA_PK_KMAX 24
#define SA_PK_K_LIST(X) \
	X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) \
	X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24)
/* ... and 16-lane groups with K = 13..64 columns per lane for 193..1024 columns (one group per DPP row: the 16 lanes of a
 * ds_read_b128 phase are 16 distinct slots, so one profile copy is conflict-free).  Up to K = 40 the token of a row is
 * its byte offset in the profile (u16); beyond, the offset in units of 256 bytes -- one shift-add instead of one add
 * per step.  K = 64: a 106 KB profile, one workgroup per CU.  The three-way (f16-ordered) form exists up to
 * SA_PK16_F16_KMAX: past ~800 columns no common scoring keeps two frames inside 0x7bff. */
#define SA_PK_K16_MIN 13
#define SA_PK16_KMAX 64
#define SA_PK16_F16_KMAX 52
#define SA_PK_K16_LIST(X) \
	X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) \
	X(27) X(28) X(29) X(30) X(31) X(32) X(33) X(34) X(35) X(36) X(37) X(38) X(39) X(40) \
	X(41) X(42) X(43) X(44) X(45) X(46) X(47) X(48) X(49) X(50) X(51) X(52) X(53) X(54) \
	X(55) X(56) X(57) X(58) X(59) X(60) X(61) X(62) X(63) X(64)
/* class index space of a plan: [0, SA_SYS_NCLASSES) s32 classes, SA_SYS_CLASS_LONG, then SA_PK_CLASS0 + K (8-lane groups),
 * then SA_PK16_CLASS0 + K (16-lane groups) */
enum : int { SA_PK_CLASS0 = SA_SYS_NCLASSES + 1, SA_PK16_CLASS0 = SA_PK_CLASS0 + SA_PK_KMAX + 1,
	     SA_PK_CLASSES_END = SA_PK16_CLASS0 + SA_PK16_KMAX + 1,
	     /* ... and every packed class once more, + SA_PK_SMALL: the columns of the class that a plan runs in its SMALL tiles (the
	      * lowest columns of the range, put at the end of the launch so that it tapers off: sa_plan.cpp: sa_plan_host) */
	     SA_PK_SMALL = SA_PK_CLASSES_END - SA_PK_CLASS0,
	     SA_PLAN_NCLASSES = SA_PK_CLASSES_END + SA_PK_SMALL };

/* The packed kernels are launched as BUNDLES: one persistent kernel (sa_k_systolic_pk_bundle<METHOD, G, KLO, F16>) walks
 * the tiles of up to SA_PK_BUNDLE consecutive K classes -- one launch on the caller's stream, one tail
 * for the whole range instead of one per class, and no cross-stream fork / join (six class launches on side streams ran
 * as two rounds of three on the runtime's hardware queues and cost ~70 us of events and barriers per range:
 * profiles/r03a_*).  8-lane groups: KLO = 1, 9, 17; 16-lane groups: KLO = 13, 21, ... 61.  The registers of a bundle are
 * those of its largest K (K <= 16: <= 127 VGPRs for every method, four waves per SIMD as before). */
#define SA_PK_BUNDLE 8
struct SaArranged { /* one arranged copy of the row store (sa_plan.cpp: sa_arrange_rows); rows = 0: none */
	const uint8_t *codes;
	const int32_t *off;    /* num+1 offsets into codes by position */
	const int32_t *rowmap; /* position -> row                      */
	const int32_t *posmap; /* row -> position                      */
	int32_t rows;          /* sequences per arranged block         */
};
struct SaPkClassArgs { /* one class of a bundle launch, in device memory */
	const int32_t *jlist;     /* columns (ascending) of the class                                        */
	const int32_t *tprefix;   /* full tiles before each column pair, then the pairs of the partial tiles */
	const int64_t *dense_off; /* share plans: element offset of tile t in its owner's dense share        */
	int32_t ncols, npart, k;
	int32_t delta, pk_base;
	int32_t chunk;            /* sequences per row stream of a full tile of this class (SaSysArgs::chunk)  */
	SaArranged lv[SA_PK_SORT_LEVELS]; /* arranged copies of the row store for this tile shape              */
};
/* a bundle launch walks ulist[0 .. nlocal): (class of the launch << SA_PK_UTILE_BITS) | tile of the class -- the full
 * tiles class after class (largest K first), then the partial tiles of ALL classes by decreasing size, so that the
 * launch tapers off on its cheapest tiles; share plans: the tiles of one rank in that order */
#define SA_PK_UTILE_BITS 28
/* LDS of a packed workgroup: scores leaving the pipeline, token rings, then the profile of the column pair (the only
 * part that depends on K): a launch asks for the bytes of its largest K as dynamic LDS */
SA_HD constexpr int sa_pk_lds_fixed(int g, int wpb) { return 416 * wpb * (64 / g); } /* per wave: 64/g groups x (32 scores x 2 halves x 2 B + a 144-entry u16 ring) */
SA_HD constexpr int sa_pk_lds_bytes(int method, int g, int k) { return sa_pk_lds_fixed(g, sa_pk_wpb(method, g, k)) + SA_CODE_ROWS * ((k + 3) / 4) * 256; }
inline int sa_pk_bundle_klo(int g, int k) { return g == 8 ? 1 + (k - 1) / SA_PK_BUNDLE * SA_PK_BUNDLE : SA_PK_K16_MIN + (k - SA_PK_K16_MIN) / SA_PK_BUNDLE * SA_PK_BUNDLE; }


/* rows of a tile's run in a dense share, padded so that every run starts 16-byte aligned in int16 and s32 alike */
#define SA_SHARE_PAD(rows) (((rows) + 7) & ~7)

/* Which arranged copy of the row store a packed tile streams (kernel and host must agree: the host builds the
 * placement of a dense share from it).  lvrows[l] = rows per block of level l (0: level not offered), largest first;
 * ra, rb = row range of the column pair, i_begin = the tile's first position, tile_rows = rows of a FULL tile.
 * A tile takes the largest offered block that lies inside [ra, rb) around it; -1: store order. */
SA_HD inline int sa_pk_pick_level(const int32_t *lvrows, int32_t ra, int32_t rb, int32_t i_begin, int32_t tile_rows)
{
	int pick = -1;
	if (ra % tile_rows != 0)
		return pick;
	for (int l = SA_PK_SORT_LEVELS - 1; l >= 0; l--) { /* smallest first: a larger fitting block overrides */
		const int32_t sr = lvrows[l];
		if (sr <= 0)
			continue;
		const int32_t blk0 = i_begin / sr * sr;
		if (blk0 >= ra && blk0 + sr <= rb)
			pick = l;
	}
	return pick;
}

/* one run of a dense share and where it goes in the packed matrix (sa_k_place): element p of the run is the score of
 * row r = map ? map[pos0 + p] : pos0 + p, stored at packed[dst + r] if ia <= r < ib.  A run whose rows are exactly a
 * permutation of [pos0, pos0 + count) (flags bit 0: an arranged tile that is its whole block) carries the INVERSE map
 * instead (posmap: row -> position) and is placed row by row. */
struct SaPlaceSeg {
	int64_t src;           /* element offset in the gathered shares (rank-major)                 */
	int64_t dst;           /* tri(j) - start of the placed range; generic runs: run start - start */
	const int32_t *map;    /* arranged tiles: position -> row; flags bit 0: row -> position       */
	int32_t count, pos0, ia, ib;
	int32_t flags;         /* bit 0: the run's rows are a permutation of [pos0, pos0 + count)     */
	int32_t pad_;
};

#endif /* SA_SHAPES_H */
