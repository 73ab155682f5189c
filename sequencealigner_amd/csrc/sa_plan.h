/* sa_plan.h -- launch planning of a packed pair range, HOST ONLY (no HIP types, no device pointers).
 *
 * The planner turns (sequence lengths, scoring-derived limits, packed range, world) into plain vectors: per
 * column-length class the column list and the tile prefix, the bundle launches of the packed kernels with their tile
 * lists, and -- for share plans -- the dealing of the tiles over the ranks, the layout of the dense shares and the
 * placement segments.  sa_launch.hip uploads the result and resolves the arranged copies it names (SaArrKey) to device
 * pointers.  Keeping this free of the runtime lets tests/ compile it with g++ -fsanitize=address,undefined and run the
 * geometry of every BASELINE config on the CPU (tests/test_plan_host.py).
 *
 * Replaces, in the reference, the arithmetic around `kernel(scores, start, batch)`: batch sizing
 * src/interface/seqalign_cuda.c:136-166 and the index -> (i, j) search d_find_j src/bio/kernels.cu:17-30. */
#ifndef SA_PLAN_H
#define SA_PLAN_H

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "sa_shapes.h"

/* j of packed index p: largest j with j(j-1)/2 <= p */
int32_t sa_column_of(int64_t p);
inline int64_t sa_tri(int64_t j) { return j * (j - 1) / 2; }

/* prefix sums over the pair space (DP cells of everything before a packed index) */
struct SaPairPlan {
	int32_t num = 0;
	int64_t pairs = 0;
	const sa_meta *meta = nullptr;
	std::vector<int64_t> len_prefix;  /* P[k] = sum_{i<k} len_i            (k = 0..N) */
	std::vector<int64_t> cell_prefix; /* C[j] = sum_{j'<j} len_j' * P[j']  (j = 0..N) */
	SaPairPlan(const sa_meta *m, int32_t n);
	int64_t cells_before(int64_t p) const; /* DP cells of all pairs with packed index < p */
};

/* one arranged copy of the row store: streams per wave, sequences per stream, rows per arranged block (0: none) */
struct SaArrKey {
	int ng = 0, ch = 0;
	int32_t block = 0;
	bool operator==(const SaArrKey &o) const { return ng == o.ng && ch == o.ch && block == o.block; }
};

/* frame shifts a value of the packed kernels can see before its last use: its own terminator entering the
 * group plus one per later terminator entering while its last rows travel through the remaining G - 1 lanes */
inline int sa_pk_live(int32_t min_len, int g) { return 1 + (g - 1) / ((min_len > 1 ? min_len : 1) + 1); }

/* everything the planner needs to know about a context */
struct SaPlanInputs {
	int32_t num = 0;
	const sa_meta *meta = nullptr; /* .len is what counts (offsets are not read) */
	int32_t min_len = 1;
	int method = 0;
	int32_t gap_ext = 0;
	bool sys_ok = false;           /* the s32 systolic classes reproduce this scoring (else: pair-per-wave runs) */
	int pk_kmax = 0, pk16_kmax = 0, pk16_f16_kmax = 0, pk_chunk_cap = SA_SYS_CHUNK;
	int32_t pk_q = 0, pk_floor = 0;
	int64_t pk_gain = 0, pk_slack = 0;
	int persistent_wgs = 256 * 32; /* 32 x CUs */
	/* development switches (sa_env.h) */
	int env_chunk = 0;
	bool no_sort = false, one_tile_size = false;
	int small_below = 16, small_div = 4, small_frac = 5;
};

/* Which kernel families reproduce the reference exactly for a scoring and a store's length range, and their constants
 * (sa_limits.cpp): the s32 systolic classes (sys_*), the packed-u16 classes K = 1..pk_kmax (8-lane groups) and
 * SA_PK_K16_MIN..pk16_kmax (16-lane groups; up to pk16_f16_kmax in the three-way f16-ordered form) */
struct SaKernelLimits {
	bool sys_ok = false;
	int32_t sys_pconst = 0, sys_q = 0;
	int64_t sys_gain = 0, sys_slack = 0;
	int pk_kmax = 0, pk16_kmax = 0, pk16_f16_kmax = 0, pk_chunk_cap = SA_SYS_CHUNK;
	int32_t pk_pconst = 0, pk_q = 0, pk_floor = 0;
	int64_t pk_gain = 0, pk_slack = 0, pk_extra = 0;
};
SaKernelLimits sa_kernel_limits(const sa_scoring &sc, int32_t max_len, int32_t min_len, bool force_generic, bool no_pk, bool no_pk16);

int32_t sa_pk_delta(const SaPlanInputs &in, int g, int k);
int32_t sa_pk_base(const SaPlanInputs &in, int g, int k);

/* a packed class of a plan: lane-group width, columns per lane, and whether it is the small-tile copy of the class */
struct SaPkCls {
	bool small;
	int g, k;
};
SaPkCls sa_pk_decode(int cls);
/* smallest s32 kernel class whose column budget W = G*K holds a column of length n (SA_SYS_CLASS_LONG beyond) */
int sa_systolic_class_for(int32_t n);

/* Does the row store get an arranged copy for this tile shape?  (no full block: nothing to arrange) */
bool sa_arranged_exists(int32_t num, const SaArrKey &key);
/* The arranged copies offered to the tiles of one packed class (largest block first); returns how many, lv[l].block = 0
 * beyond.  host_out: scores (also) leave straight to host memory -- a block is then one tile (row-order epilogue). */
int sa_pk_arranged_keys(const SaPlanInputs &in, int pk_g, int pk_k, int32_t chunk_pk, bool host_out, SaArrKey (&lv)[SA_PK_SORT_LEVELS]);
/* The permutation of one arranged copy: rowmap[position] = row (DESIGN 4.2 "arranged row streams") */
void sa_arrange_rows(const sa_meta *meta, int32_t num, const SaArrKey &key, std::vector<int32_t> &rowmap);

struct SaHostClass {
	int cls = 0;
	int32_t ncols = 0, ntiles = 0;
	int32_t npart = 0; /* packed classes: partial tiles among ntiles (listed behind the tile prefix) */
	int32_t chunk = 0; /* packed classes: sequences per row stream of a full tile */
	int64_t pairs = 0, cells = 0;
	std::vector<int32_t> jlist, tprefix;
	std::vector<int32_t> part_rows;  /* packed classes: rows of the partial tiles, in tile order (decreasing)  */
	/* share plans (tile-interleaved sharding): the tiles of rank 0, rank 1, ... back to back; rank r runs
	 * tlist[rank_first[r] .. rank_first[r + 1]) and stores tile t at doff[t] of its dense share */
	std::vector<int32_t> tlist;
	std::vector<int64_t> doff;
	std::vector<int16_t> owner;      /* rank of every tile */
	std::vector<int32_t> rank_first;
	std::vector<int64_t> rank_pairs, rank_cells;
};

struct SaHostSeg { /* SaPlaceSeg with the map named instead of pointed to */
	int64_t src, dst;
	int32_t count, pos0, ia, ib, flags;
	int map_kind; /* 0: none, 1: rowmap of `key`, 2: posmap of `key` */
	SaArrKey key;
};

struct SaHostPkArgs { /* SaPkClassArgs without the device pointers */
	int cls_index = 0; /* index into SaHostPlan::classes */
	int32_t ncols = 0, npart = 0, k = 0, delta = 0, pk_base = 0, chunk = 0;
	SaArrKey lv[SA_PK_SORT_LEVELS];
};

struct SaHostBundle {
	int g = 8, klo = 1, f16 = 1, kmax = 1;
	std::vector<int> cls;            /* indices into SaHostPlan::classes, walking order                */
	std::vector<SaHostPkArgs> args;  /* cls.size() entries                                              */
	std::vector<uint32_t> ulist;     /* the tiles in walking order, rank after rank (two words per tile) */
	std::vector<int64_t> ufirst;     /* max(world, 1) + 1 offsets into ulist                            */
	std::vector<int32_t> nlocal;     /* tiles of the launch, per rank (one entry when world == 0)       */
	std::vector<int64_t> pairs, cells;
};

struct SaGenericShare {
	int64_t start, count, doff;
};

struct SaHostPlan {
	int64_t start = -1, count = -1;
	int32_t chunk = SA_SYS_CHUNK;    /* sequences per group stream chosen for this range */
	int32_t chunk_pk = SA_SYS_CHUNK; /* ... for the packed classes (their tiles are workgroup-tiles of two columns) */
	int32_t chunk_pk_small = 0;      /* ... and for their columns below j_small: the tiles that end the launch (0: none) */
	int32_t j_small = 0;
	std::vector<SaHostClass> classes;
	std::vector<std::pair<int64_t, int64_t>> generic; /* (start, count) runs for the pair-per-wave kernels */
	int world = 0;
	bool share_host = false;
	int64_t share_elems = 0;
	std::vector<SaHostSeg> segs;
	std::vector<SaHostBundle> bundles;
	std::vector<std::vector<SaGenericShare>> generic_share; /* [rank]: sub-runs of the generic runs */
};

/* world = 0: the plan of sa_ctx_align_range (every tile, packed order).  world >= 1: a SHARE plan -- the same tile lists
 * dealt over `world` ranks, dense tile-order output, placement segments.  false + sa_set_error on a range that does not
 * fit one launch. */
bool sa_plan_host(const SaPlanInputs &in, int64_t start, int64_t count, int world, bool share_host, SaHostPlan &plan);

#endif /* SA_PLAN_H */
