/*
 * sa_driver.hip -- host side of libseqalign_hip.so: device context, planning, batching.
 *
 * Replaces the reference's device driver src/interface/seqalign_cuda.c:
 *   cuda_device_init :48-69   -> device_ready()
 *   cuda_memory      :71-93   -> sa_hip_memory()
 *   cuda_align       :95-296  -> sa_hip_align() on top of sa_ctx_create()/sa_ctx_align_range()
 * Differences by design: scoring state is passed explicitly (struct sa_scoring) instead of
 * process globals; sequences are uploaded pre-encoded (residue index per byte) so no kernel
 * ever touches the ASCII->index table; there is no CPU fallback.
 */
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

#include "sa_internal.h"
#include <atomic>
#include <csignal>
#include <execinfo.h>
#include <fcntl.h>
#include <unistd.h>
#include <chrono>

/* SA_HIP_ABORT_TRACE=<file> (diagnostics): a C backtrace of whoever raises SIGABRT in this process, appended to the file */
static char g_abort_trace_path[512];
static void sa_abort_trace(int)
{
	void *frames[64];
	const int n = backtrace(frames, 64);
	const int fd = open(g_abort_trace_path, O_WRONLY | O_CREAT | O_APPEND, 0644);
	if (fd >= 0) {
		backtrace_symbols_fd(frames, n, fd);
		close(fd);
	}
	signal(SIGABRT, SIG_DFL);
	raise(SIGABRT);
}

__attribute__((constructor)) static void sa_runtime_knobs(void)
{
	if (const char *p = getenv("SA_HIP_ABORT_TRACE")) {
		snprintf(g_abort_trace_path, sizeof(g_abort_trace_path), "%s", p);
		signal(SIGABRT, sa_abort_trace);
	}
}

struct sa_ctx {
	int device = 0;
	int32_t num = 0, max_len = 0, min_len = 0;
	int64_t pairs = 0;
	sa_scoring sc{};
	std::vector<sa_meta> meta;     /* device-side (tight) layout: off[k] = sum_{i<k}(len_i+1)          */
	std::vector<int32_t> off;      /* num+1 tight offsets                                                */
	std::vector<uint8_t> codes;    /* host copy of the encoded store (arranged copies are made from it)  */
	uint8_t *d_codes = nullptr;
	sa_meta *d_meta = nullptr;
	int32_t *d_off = nullptr;
	int32_t *d_sub = nullptr;
	int8_t *d_sub8 = nullptr;
	int32_t *d_scratch = nullptr;
	int64_t scratch_stride = 0;
	int generic_blocks = 0;
	/* tile counters of the persistent launches: one slot of (classes + 1) counters per sa_ctx_align_range
	 * call, taken round-robin from a ring so that ranges issued back to back on DIFFERENT streams (an
	 * overlapped multi-chunk schedule) never share a counter */
	enum { COUNTER_SLOTS = 256, COUNTERS_PER_SLOT = 2 * SA_PLAN_NCLASSES }; /* (next tile, workgroups done) per class */
	unsigned *d_counters = nullptr;
	uint64_t call_no = 0;
	hipEvent_t slot_done[COUNTER_SLOTS] = {}; /* recorded after the launches that used a slot: its next user waits */
	int32_t *d_long_scratch = nullptr; /* strip boundaries of the strip-mined launch, per workgroup   */
	int64_t long_stride = 0;           /* ints per workgroup                                          */
	int long_wgs = 0;
	int persistent_wgs = 0;         /* workgroups of a persistent systolic launch  */
	/* a range with SEVERAL launches (more than one packed bundle, s32 classes beside packed ones) runs them concurrently
	 * on side streams forked from / joined into the caller's stream; created on first use */
	enum { NSIDE = 8 };
	hipStream_t side[NSIDE] = {};
	hipEvent_t fork_ev = nullptr, join_ev[NSIDE] = {};
	/* systolic fast path: parameters and validity (see systolic_setup) */
	bool sys_ok = false;
	int32_t sys_pconst = 0, sys_q = 0;
	int64_t sys_gain = 0, sys_slack = 0;
	/* packed-u16 kernels (sa_systolic_pk.inc): column classes K = 1..pk_kmax run there (0: none), see pk_setup */
	int pk_kmax = 0, pk16_kmax = 0; /* 8-lane groups: K = 1..pk_kmax; 16-lane groups: K = SA_PK_K16_MIN..pk16_kmax */
	int pk16_f16_kmax = 0;          /* 16-lane groups: classes up to this K fit the f16 range (three-way maxima)      */
	int pk_chunk_cap = SA_SYS_CHUNK; /* longest row stream (sequences) the packed classes may be given (SW: bounds the drift) */
	int32_t pk_pconst = 0, pk_q = 0, pk_floor = 0; /* pk_floor: margin below the lowest legitimate value (part of BASE) */
	int64_t pk_gain = 0, pk_slack = 0, pk_extra = 0;
	/* arranged copies of the store for the packed kernels' row streams (arranged_store), one per tile shape */
	struct Arranged {
		int ng = 0, ch = 0;      /* streams per wave, sequences per stream */
		int32_t block = 0;       /* rows per arranged block                 */
		uint8_t *d_codes = nullptr;
		int32_t *d_off = nullptr, *d_rowmap = nullptr, *d_posmap = nullptr;
	};
	std::vector<Arranged> arranged;
	bool env_no_sort = false;
	bool out_is_host = false; /* the range being launched stores straight into host memory (sa_ctx_align_host) */
	/* launch plans of recently used packed ranges (callers loop over the same few ranges) */
	struct ClassLaunch {
		int cls = 0;
		int32_t ncols = 0, ntiles = 0;
		int32_t npart = 0; /* packed classes: partial tiles among ntiles (listed behind the tile prefix) */
		int32_t chunk = 0; /* packed classes: sequences per row stream of a full tile */
		int64_t pairs = 0, cells = 0;
		int32_t *d_jlist = nullptr, *d_tprefix = nullptr;
		/* share plans (tile-interleaved sharding): the tiles of rank 0, rank 1, ... back to back; rank r runs
		 * d_tlist[rank_first[r] .. rank_first[r + 1]) and stores tile t at d_doff[t] of its dense share */
		int32_t *d_tlist = nullptr;
		int64_t *d_doff = nullptr;
		std::vector<int32_t> part_rows;  /* packed classes: rows of the partial tiles, in tile order (decreasing)  */
		std::vector<int16_t> owner;      /* share plans: rank of every tile                                        */
		std::vector<int32_t> rank_first;
		std::vector<int64_t> rank_pairs, rank_cells;
	};
	struct Plan {
		int64_t start = -1, count = -1;
		int32_t chunk = SA_SYS_CHUNK; /* sequences per group stream chosen for this range */
		int32_t chunk_pk = SA_SYS_CHUNK; /* ... for the packed classes (their tiles are workgroup-tiles of two columns) */
		int32_t chunk_pk_small = 0;      /* ... and for their columns below j_small: the tiles that end the launch (0: none) */
		int32_t j_small = 0;
		std::vector<ClassLaunch> classes;
		std::vector<std::pair<int64_t, int64_t>> generic; /* (start, count) runs for the generic kernels */
		uint64_t stamp = 0;
		/* share plans: world > 0.  Every rank's dense share is share_elems elements long (the longest one's length);
		 * segs places the gathered shares (rank-major) into packed order */
		int world = 0;
		bool share_host = false; /* share plans: the ranks also deliver to a host matrix (tiles are their own arranged blocks) */
		int64_t share_elems = 0;
		SaPlaceSeg *d_segs = nullptr;
		int32_t nsegs = 0;
		struct GenericShare {
			int64_t start, count, doff;
		};
		/* the packed classes are launched in bundles (sa_internal.h: SaPkClassArgs): classes of one lane-group width,
		 * one block of SA_PK_BUNDLE consecutive K and one maxima form, walked by decreasing K */
		struct PkLaunch {
			int g = 8, klo = 1, f16 = 1, kmax = 1;
			std::vector<int> cls;          /* indices into `classes`, walking order                       */
			SaPkClassArgs *d_args = nullptr; /* cls.size() entries                                          */
			uint32_t *d_ulist = nullptr;   /* the tiles in walking order, rank after rank                 */
			std::vector<int64_t> ufirst;   /* max(world, 1) + 1 offsets into d_ulist                      */
			std::vector<int32_t> nlocal;   /* tiles of the launch, per rank (one entry when world == 0)   */
			std::vector<int64_t> pairs, cells;
		};
		std::vector<PkLaunch> pk_launches;
		std::vector<std::vector<GenericShare>> generic_share; /* [rank]: sub-runs of the generic runs */
	};
	std::vector<Plan> plans;  /* small LRU cache */
	Plan *plan = nullptr;     /* plan of the current sa_ctx_align_range call */
	uint64_t plan_clock = 0;
	/* instrumentation: one HIP-event pair per kernel launch, keyed by kernel name */
	bool timing = false;
	struct Timed {
		std::string name;
		hipEvent_t e0, e1;
		int64_t pairs, cells;
	};
	std::vector<Timed> events;
	/* development switches, read once when the context is created */
	bool env_serial_classes = false, env_stamps = false, env_no_pin = false, env_no_shells = false, env_no_direct = false;
	int env_chunk = 0, env_stagger = 0, env_pk_wgs = 0, env_rotate_prio = -1; /* (-1: by method) */
	bool leave_room = false; /* sa_ctx_leave_room */
	/* progress reporting (sa_hip_set_progress): the tile counters of the launches of the last sa_ctx_align_range call */
	struct ProgItem {
		const unsigned *counter;
		int64_t tiles;
	};
	std::vector<ProgItem> prog_items;
	/* where the set-up time of this context went, milliseconds (sa_hip_last_align_breakdown) */
	struct SetupMs {
		double encode = 0, device = 0, upload = 0, code_objects = 0, pin = 0, plan = 0, arrange = 0;
	} setup;
	/* host delivery (sa_ctx_align_host): streams, events and buffers, created on first use and kept */
	struct Deliver {
		hipStream_t compute = nullptr, copy = nullptr;
		hipEvent_t done[2] = {}, copied[2] = {};
		int32_t *d_buf[2] = {};   /* double-buffered batches of packed scores                  */
		int64_t buf_elems[2] = {};
		int32_t *h_stage[2] = {}; /* pinned staging for the host-scattered full layout          */
		int64_t stage_elems[2] = {};
		int32_t *d_packed = nullptr, *d_full = nullptr; /* full layout, shell schedule          */
		int64_t packed_elems = 0, full_elems = 0;
	} dl;
};

namespace {
void deliver_release(sa_ctx *ctx);
}

static std::atomic<sa_progress_fn> g_progress_fn{ nullptr };
static std::atomic<void *> g_progress_user{ nullptr };

extern "C" void sa_hip_set_progress(sa_progress_fn fn, void *user)
{
	g_progress_user.store(user);
	g_progress_fn.store(fn);
}

/* (sa_hip_align on several devices runs one thread per slice: the first slice speaks for the job) */
static thread_local bool t_progress_here = true;

static void report_progress(double fraction)
{
	if (!t_progress_here)
		return;
	if (sa_progress_fn fn = g_progress_fn.load())
		fn(fraction < 0 ? 0 : fraction > 1 ? 1 : fraction, g_progress_user.load());
}
static double ms_since(std::chrono::steady_clock::time_point t0)
{
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

/* is the byte at p page-locked and known to the HIP runtime (hipHostMalloc / hipHostRegister)?  Callers that hand a
 * RANGE to the kernels probe its first and its last element: a destination registered only in part must not take the
 * direct-store path (a GPU page fault instead of a fallback). */
static bool host_range_is_pinned(const void *p)
{
	hipPointerAttribute_t attr;
	if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
		(void)hipGetLastError();
		return false;
	}
	return attr.type == hipMemoryTypeHost;
}

static void plan_free(sa_ctx::Plan &pl)
{
	for (auto &c : pl.classes) {
		(void)hipFree(c.d_jlist);
		(void)hipFree(c.d_tprefix);
		(void)hipFree(c.d_tlist);
		(void)hipFree(c.d_doff);
	}
	(void)hipFree(pl.d_segs);
	for (auto &b : pl.pk_launches) {
		(void)hipFree(b.d_args);
		(void)hipFree(b.d_ulist);
	}
	pl = sa_ctx::Plan();
}

static void plan_release(sa_ctx *ctx)
{
	for (auto &pl : ctx->plans)
		plan_free(pl);
	ctx->plans.clear();
	ctx->plan = nullptr;
}

static bool device_ready(int device)
{
	int count = 0;
	hipError_t err = hipGetDeviceCount(&count);
	if (err != hipSuccess || count <= 0) {
		sa_set_error("No HIP devices available (%s); libseqalign_hip has no CPU fallback",
			     err == hipSuccess ? "device count is 0" : hipGetErrorString(err));
		return false;
	}
	if (device < 0 || device >= count) {
		sa_set_error("HIP device %d out of range (%d visible)", device, count);
		return false;
	}
	SA_HIP_CHECK(hipSetDevice(device), return false);
	return true;
}

extern "C" int sa_hip_device_count(void)
{
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess)
		return 0;
	return count;
}

extern "C" const char *sa_hip_device_name(int device)
{
	static thread_local char name[256];
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess)
		return nullptr;
	snprintf(name, sizeof(name), "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
	return name;
}

/* ---- context ------------------------------------------------------------- */

static bool validate_and_encode(const sa_input &in, const sa_scoring &sc, std::vector<uint8_t> &codes,
				std::vector<int32_t> &off, int32_t &max_len)
{
	if (!in.seqs || !in.meta) {
		sa_set_error("sa_input: null sequence store");
		return false;
	}
	if (in.num < 2) { /* reference src/bio/align.h:21, src/io/input.c:63 */
		sa_set_error("Not enough sequences: %d (min: 2)", in.num);
		return false;
	}
	if (sc.method < 0 || sc.method >= SA_METHOD_COUNT) {
		sa_set_error("Invalid alignment method id %d", sc.method);
		return false;
	}
	/* device layout is always tight (sequence, terminator, next sequence ...) whatever the
	 * caller's offsets are: the systolic kernels stream this blob as their row input */
	int64_t end = 0;
	max_len = 0;
	off.assign((size_t)in.num + 1, 0);
	for (int32_t k = 0; k < in.num; k++) {
		const sa_meta m = in.meta[k];
		if (m.len < 1 || m.off < 0) { /* src/bio/align.h:22,25 */
			sa_set_error("Sequence #%d has invalid offset/length (%d/%d)", k + 1, m.off, m.len);
			return false;
		}
		off[(size_t)k] = (int32_t)end;
		end += (int64_t)m.len + 1;
		max_len = std::max(max_len, m.len);
		if (end > INT32_MAX) { /* src/io/source/fasta.c:73 */
			sa_set_error("Sequence store exceeds 2 GiB");
			return false;
		}
	}
	off[(size_t)in.num] = (int32_t)end;
	codes.assign((size_t)end, (uint8_t)SA_CODE_SEP);
	for (int32_t k = 0; k < in.num; k++) {
		const sa_meta m = in.meta[k];
		const uint8_t *s = in.seqs + m.off;
		for (int32_t p = 0; p < m.len; p++) {
			const uint8_t ch = s[p];
			const int32_t idx = ch < SA_LUT_SIZE ? sc.lut[ch] : -1;
			if (idx < 0 || idx >= SA_SUB_DIM) { /* parsers reject these: src/io/source/fasta.c:57-63 */
				sa_set_error("Invalid character 0x%02x in sequence #%d at position %d", ch, k + 1, p + 1);
				return false;
			}
			codes[(size_t)off[(size_t)k] + p] = (uint8_t)idx;
		}
		if (s[m.len] != 0) {
			sa_set_error("Sequence #%d is not NUL-terminated at its recorded length", k + 1);
			return false;
		}
	}
	/* 32-bit safety: the reference computes in s32 and is undefined once a border, a sentinel
	 * plus a gap, or a cell wraps.  Refuse such parameter/length combinations loudly. */
	int64_t amax = 0, smax = 0;
	if (sc.method == SA_METHOD_NW)
		amax = std::llabs((int64_t)sc.gap_pen);
	else
		amax = std::max(std::llabs((int64_t)sc.gap_opn), std::llabs((int64_t)sc.gap_ext));
	for (int k = 0; k < SA_SUB_DIM * SA_SUB_DIM; k++)
		smax = std::max<int64_t>(smax, std::llabs((int64_t)sc.sub[k]));
	const __int128 bound = (__int128)(2 * (int64_t)max_len + 3) * amax + (__int128)max_len * smax;
	if (bound >= ((__int128)1 << 30)) {
		sa_set_error("Gap penalty %lld with sequence length %d overflows 32-bit scores "
			     "(undefined in the reference as well)", (long long)amax, max_len);
		return false;
	}
	return true;
}

/* Decides whether the systolic streaming kernels (sa_systolic.hip) reproduce the reference exactly
 * for this scoring, and derives their constants.  Conditions (see that file's header):
 *   - profile entries S + pconst (and the Gotoh first-column tweak) fit s8, -128 is reserved,
 *   - Gotoh: q = open - extend <= 0 (|open| >= |extend|),
 *   - the per-sequence baseline raise DELTA times the stream length stays far inside s32. */
static void systolic_setup(sa_ctx *ctx)
{
	const sa_scoring &sc = ctx->sc;
	int64_t smax = INT32_MIN, smin = INT32_MAX;
	for (int k = 0; k < SA_SUB_DIM * SA_SUB_DIM; k++) {
		smax = std::max<int64_t>(smax, sc.sub[k]);
		smin = std::min<int64_t>(smin, sc.sub[k]);
	}
	ctx->sys_ok = false;
	if (smin < -127 || smax > 127 || getenv("SA_HIP_FORCE_GENERIC"))
		return;
	int64_t pconst, q = 0, pmax, gain, slack;
	const int64_t g = sc.gap_pen, o = sc.gap_opn, e = sc.gap_ext;
	switch (sc.method) {
	case SA_METHOD_NW:
		pconst = -2 * g;
		pmax = smax + pconst;
		gain = std::max<int64_t>(0, smax - 2 * g);
		slack = 2;
		break;
	case SA_METHOD_GA:
		q = o - e;
		if (q > 0)
			return;
		pconst = -e - o;
		pmax = smax + pconst - q; /* first real column carries -q on top */
		gain = std::max<int64_t>(0, smax - 2 * e);
		slack = 2 * (-q) + 2;
		break;
	default:
		pconst = -o;
		pmax = smax + pconst;
		gain = std::max<int64_t>(0, smax);
		slack = -o - e + 2;
		break;
	}
	if (pmax > 127 || smin + pconst < -127)
		return;
	/* widest column the fast path will see: padded to whole strips when longer than the widest class */
	const int64_t wmax = ((int64_t)ctx->max_len + SA_SYS_LONG_W - 1) / SA_SYS_LONG_W * SA_SYS_LONG_W;
	/* SW works in a per-row shifted domain: values additionally drift by |e| per stream position of a tile */
	const int64_t drift = sc.method == SA_METHOD_SW ? (int64_t)SA_SYS_CHUNK * ((int64_t)ctx->max_len + 1) * std::llabs(e) : 0;
	if ((gain * wmax + slack) * (SA_SYS_CHUNK + 2) + drift >= ((int64_t)1 << 29))
		return;
	ctx->sys_ok = true;
	ctx->sys_pconst = (int32_t)pconst;
	ctx->sys_q = (int32_t)q;
	ctx->sys_gain = gain;
	ctx->sys_slack = slack;
}

/* smallest kernel class whose column budget W = G*K holds a column sequence of length n; longer columns go
 * to the strip-mined launch of the widest class */
static int systolic_class_for(int32_t n)
{
	for (int c = 0; c < SA_SYS_NCLASSES; c++)
		if (SA_SYS_CLASSES[c].G * SA_SYS_CLASSES[c].K >= n)
			return c;
	return SA_SYS_CLASS_LONG;
}

/* a packed class of a plan: lane-group width, columns per lane, and whether it is the small-tile copy of the class */
struct PkCls {
	bool small;
	int g, k;
};
static PkCls pk_decode(int cls)
{
	PkCls r;
	r.small = cls >= SA_PK_CLASSES_END;
	const int c = r.small ? cls - SA_PK_SMALL : cls;
	r.g = c >= SA_PK16_CLASS0 ? 16 : 8;
	r.k = c - (r.g == 16 ? SA_PK16_CLASS0 : SA_PK_CLASS0);
	return r;
}

/* frame shifts a value of the packed kernels can see before its last use: its own terminator entering the group plus
 * one per later terminator entering while its last rows travel through the remaining G - 1 lanes.  Terminators are
 * min_len + 1 stream positions apart at least (every sequence is a row of some stream): 4 / 8 shifts for 8- / 16-lane
 * groups when the store holds a sequence of length 1, one shift when its shortest sequence has >= G - 1 residues. */
static int pk_live(const sa_ctx *ctx, int g) { return 1 + (g - 1) / (std::max(ctx->min_len, 1) + 1); }

static int32_t pk_delta(const sa_ctx *ctx, int g, int k) { return (int32_t)(ctx->pk_gain * g * k + ctx->pk_slack); }
static int32_t pk_base(const sa_ctx *ctx, int g, int k)
{
	/* (Gotoh: values reach BASE + 3q; SW: the lanes start up to G |e| below the baseline) */
	return pk_live(ctx, g) * pk_delta(ctx, g, k) + ctx->pk_floor + 4 * std::abs(ctx->pk_q) + 4 +
	       (ctx->sc.method == SA_METHOD_SW ? g * std::abs(ctx->sc.gap_ext) : 0);
}

/* Decides which column classes the packed-u16 kernels reproduce exactly (see sa_systolic_pk.inc): profile entries
 * S + const (and the Gotoh first-column tweak) are >= 0, Gotoh q <= 0, and for class K every value of a register stays
 * inside [floor, limit]: BASE + DELTA + the largest profile entry must fit.  The largest such K is pk_kmax.  limit =
 * 65535 for the 16-lane groups; the 8-lane kernels take their three-way maxima with v_pk_maximum3_f16, whose order on
 * u16 bit patterns is the unsigned order up to 0x7c00: limit = SA_PK_F16_MAX. */
static void pk_setup(sa_ctx *ctx)
{
	const sa_scoring &sc = ctx->sc;
	ctx->pk_kmax = ctx->pk16_kmax = ctx->pk16_f16_kmax = 0;
	ctx->pk_chunk_cap = SA_SYS_CHUNK;
	if (!ctx->sys_ok || getenv("SA_HIP_NO_PK"))
		return;
	int64_t smax = INT32_MIN, smin = INT32_MAX;
	for (int k = 0; k < SA_SUB_DIM * SA_SUB_DIM; k++) {
		smax = std::max<int64_t>(smax, sc.sub[k]);
		smin = std::min<int64_t>(smin, sc.sub[k]);
	}
	const int64_t g = sc.gap_pen, o = sc.gap_opn, e = sc.gap_ext;
	int64_t pconst, q = 0, pmax, gain, slack, floor_v, extra = 0;
	if (sc.method == SA_METHOD_SW) {
		/* row-shifted domain: V = m - |o - e| and X = max(V, X) - |e| are plain subtractions (o <= e <= 0), values drift
		 * up by |e| per stream position of a tile and the lanes start up to G |e| below the baseline */
		if (o > e)
			return;
		pconst = -o;
		pmax = smax + pconst;
		gain = std::max<int64_t>(1, smax);
		slack = -o - e + 2;
		floor_v = -o - 2 * e + 2;
		q = o + e; /* (only its magnitude is used below: margins) */
		/* the drift of a tile grows with the length of its row streams: long sequences get shorter streams (a stream of
		 * 8 sequences of 1000 residues is 8000 steps -- the per-tile costs are long amortised), so that the drift stays a
		 * fraction of the u16 range and SW keeps the packed kernels whatever the longest sequence is */
		int cap = SA_SYS_CHUNK;
		while (cap > 4 && (int64_t)cap * ((int64_t)ctx->max_len + 1) * (-e) > 12000)
			cap >>= 1;
		ctx->pk_chunk_cap = cap;
		extra = (int64_t)cap * ((int64_t)ctx->max_len + 1) * (-e) + 16 * (-e);
	} else if (sc.method == SA_METHOD_NW) {
		pconst = -2 * g;
		pmax = smax + pconst;
		gain = std::max<int64_t>(1, pmax);
		slack = 2;
		floor_v = 0;
	} else {
		q = o - e;
		if (q > 0)
			return;
		pconst = -e - o;
		pmax = smax + pconst - q; /* the first real column carries -q on top */
		gain = std::max<int64_t>(1, smax - 2 * e);
		slack = 2 * (-q) + 2;
		floor_v = 2 * (-q) + 2;
	}
	if (smin + pconst < 0 || pmax > 4096 || -q > 4096)
		return;
	ctx->pk_pconst = (int32_t)pconst;
	ctx->pk_q = (int32_t)q;
	ctx->pk_extra = extra;
	ctx->pk_gain = gain;
	ctx->pk_slack = slack;
	ctx->pk_floor = (int32_t)floor_v;
	const int64_t fixed = floor_v + 4 * (-q) + 4 + pmax + (-q) + extra;
	for (int k = 1; k <= SA_PK_KMAX; k++) {
		if ((pk_live(ctx, 8) + 1) * (gain * 8 * k + slack) + fixed > SA_PK_F16_MAX) /* (8-lane groups: f16-ordered halves) */
			break;
		ctx->pk_kmax = k;
	}
	if (ctx->pk_kmax == SA_PK_KMAX && !getenv("SA_HIP_NO_PK16")) /* wider columns: 16-lane groups, twice as many shifts in flight */
		for (int k = SA_PK_K16_MIN; k <= SA_PK16_KMAX; k++) {
			const int64_t top = (pk_live(ctx, 16) + 1) * (gain * 16 * k + slack) + fixed;
			if (top > 65535)
				break;
			ctx->pk16_kmax = k;
			if (top <= SA_PK_F16_MAX && k <= SA_PK16_F16_KMAX)
				ctx->pk16_f16_kmax = k;
		}
}

extern "C" sa_ctx *sa_ctx_create(int device, struct sa_input in, const struct sa_scoring *sc)
{
	if (!sc) {
		sa_set_error("sa_ctx_create: null scoring");
		return nullptr;
	}
	std::vector<uint8_t> codes;
	std::vector<int32_t> off;
	int32_t max_len = 0;
	const auto t_encode = std::chrono::steady_clock::now();
	if (!validate_and_encode(in, *sc, codes, off, max_len))
		return nullptr;
	const double encode_ms = ms_since(t_encode);
	const bool verbose = getenv("SA_HIP_VERBOSE") != nullptr;
	const auto t_create = std::chrono::steady_clock::now();
	auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_create).count(); };
	if (!device_ready(device))
		return nullptr;
	const double device_ms = since();
	if (verbose)
		fprintf(stderr, "[seqalign_hip] sa_ctx_create: device ready at %.1f ms\n", since());

	sa_ctx *ctx = new sa_ctx();
	ctx->setup.encode = encode_ms;
	ctx->setup.device = device_ms;
	ctx->device = device;
	ctx->num = in.num;
	ctx->max_len = max_len;
	ctx->min_len = max_len;
	for (int32_t k = 0; k < in.num; k++)
		ctx->min_len = std::min<int32_t>(ctx->min_len, (int32_t)in.meta[k].len);
	ctx->pairs = (int64_t)in.num * (in.num - 1) / 2;
	ctx->sc = *sc;
	ctx->meta.resize((size_t)in.num);
	for (int32_t k = 0; k < in.num; k++)
		ctx->meta[(size_t)k] = sa_meta{ off[(size_t)k], in.meta[k].len };
	ctx->off = off;
	ctx->codes = codes;
	ctx->env_serial_classes = getenv("SA_HIP_SERIAL_CLASSES") != nullptr;
	ctx->env_stamps = getenv("SA_HIP_STAMPS") != nullptr;
	ctx->env_no_pin = getenv("SA_HIP_NO_PIN") != nullptr;
	ctx->env_no_shells = getenv("SA_HIP_NO_SHELLS") != nullptr;
	ctx->env_no_direct = getenv("SA_HIP_NO_DIRECT") != nullptr;
	ctx->env_no_sort = getenv("SA_HIP_NO_SORT") != nullptr;
	if (const char *e = getenv("SA_HIP_ROTATE_PRIO")) /* development switch: 0 = every wave at priority 0 (oldest first) */
		ctx->env_rotate_prio = atoi(e) != 0;
	if (const char *e = getenv("SA_HIP_PK_WGS")) /* development switch: persistent workgroups of a packed launch */
		ctx->env_pk_wgs = std::max(0, atoi(e));
	if (const char *e = getenv("SA_HIP_STAGGER"))
		ctx->env_stagger = std::max(0, std::min(64, atoi(e)));
	if (const char *e = getenv("SA_HIP_CHUNK")) /* development switch: fixed stream length */
		ctx->env_chunk = std::max(1, std::min(SA_SYS_CHUNK, atoi(e)));
	systolic_setup(ctx);
	pk_setup(ctx);
	int8_t sub8[SA_SUB_DIM * SA_SUB_DIM];
	for (int k = 0; k < SA_SUB_DIM * SA_SUB_DIM; k++)
		sub8[k] = (int8_t)std::max(-128, std::min(127, sc->sub[k]));

	bool ok = false;
	do {
		SA_HIP_CHECK(hipMalloc(&ctx->d_codes, codes.size()), break);
		SA_HIP_CHECK(hipMalloc(&ctx->d_meta, sizeof(sa_meta) * (size_t)in.num), break);
		SA_HIP_CHECK(hipMalloc(&ctx->d_sub, sizeof(sc->sub)), break);
		if (verbose)
			fprintf(stderr, "[seqalign_hip] sa_ctx_create: first allocations at %.1f ms\n", since());
		SA_HIP_CHECK(hipMemcpy(ctx->d_codes, codes.data(), codes.size(), hipMemcpyHostToDevice), break);
		if (verbose)
			fprintf(stderr, "[seqalign_hip] sa_ctx_create: first upload at %.1f ms\n", since());
		SA_HIP_CHECK(hipMemcpy(ctx->d_meta, ctx->meta.data(), sizeof(sa_meta) * (size_t)in.num, hipMemcpyHostToDevice), break);
		SA_HIP_CHECK(hipMemcpy(ctx->d_sub, sc->sub, sizeof(sc->sub), hipMemcpyHostToDevice), break);
		SA_HIP_CHECK(hipMalloc(&ctx->d_off, sizeof(int32_t) * off.size()), break);
		SA_HIP_CHECK(hipMemcpy(ctx->d_off, off.data(), sizeof(int32_t) * off.size(), hipMemcpyHostToDevice), break);
		SA_HIP_CHECK(hipMalloc(&ctx->d_sub8, sizeof(sub8)), break);
		SA_HIP_CHECK(hipMemcpy(ctx->d_sub8, sub8, sizeof(sub8), hipMemcpyHostToDevice), break);
		if (verbose)
			fprintf(stderr, "[seqalign_hip] sa_ctx_create: uploads at %.1f ms\n", since());

		/* strip-boundary scratch of the pair-per-wave kernels: 2*(max+2) ints per resident wave */
		hipDeviceProp_t prop;
		SA_HIP_CHECK(hipGetDeviceProperties(&prop, device), break);
		ctx->scratch_stride = 2 * ((int64_t)max_len + 2);
		int64_t blocks = (int64_t)prop.multiProcessorCount * 8;
		const int64_t budget = (int64_t)1 << 30; /* 1 GiB */
		const int64_t per_block = ctx->scratch_stride * 4 * (int64_t)sizeof(int32_t);
		blocks = std::max<int64_t>(prop.multiProcessorCount, std::min(blocks, budget / per_block));
		ctx->generic_blocks = (int)blocks;
		SA_HIP_CHECK(hipMalloc(&ctx->d_scratch, (size_t)(blocks * per_block)), break);
		SA_HIP_CHECK(hipMalloc(&ctx->d_counters, sizeof(unsigned) * sa_ctx::COUNTER_SLOTS * sa_ctx::COUNTERS_PER_SLOT), break);
		SA_HIP_CHECK(hipMemset(ctx->d_counters, 0, sizeof(unsigned) * sa_ctx::COUNTER_SLOTS * sa_ctx::COUNTERS_PER_SLOT), break);
		if (verbose)
			fprintf(stderr, "[seqalign_hip] sa_ctx_create: scratch at %.1f ms\n", since());
		ctx->persistent_wgs = prop.multiProcessorCount * 32;
		/* (the side streams of a range with several launches are created when the first such range comes: a store whose
		 * columns fall into one bundle -- the usual case -- never needs them, and eight streams cost ~70 ms of set-up) */
		/* code objects are loaded lazily at the first launch: do it here, with the other set-up */
		/* code objects of the kernel families this store's columns fall into (plan_build's class choice) */
		int families = 0;
		for (int32_t k = 0; k < in.num; k++) {
			const int n = in.meta[k].len, k8 = (n + 7) / 8, k16 = (n + 15) / 16;
			families |= k8 <= ctx->pk_kmax ? SA_WARM_PK8
				    : (k16 >= SA_PK_K16_MIN && k16 <= ctx->pk16_kmax) ? (sa_pk_bundle_klo(16, k16) >= 45 ? SA_WARM_PK16HI : SA_WARM_PK16)
										       : SA_WARM_S32;
		}
		if (verbose)
			fprintf(stderr, "[seqalign_hip] sa_ctx_create: buffers, streams and events at %.1f ms\n", since());
		ctx->setup.upload = since() - device_ms; /* allocations, uploads, streams and events */
		const double t_warm = since();
		SA_HIP_CHECK(sa_warm_kernels(sc->method, families), break);
		ctx->setup.code_objects = since() - t_warm;
		if (verbose)
			fprintf(stderr, "[seqalign_hip] sa_ctx_create: code objects of families %d loaded at %.1f ms\n", families, since());
		ok = true;
	} while (0);
	if (!ok) {
		sa_ctx_destroy(ctx);
		return nullptr;
	}
	return ctx;
}

extern "C" void sa_ctx_destroy(sa_ctx *ctx)
{
	if (!ctx)
		return;
	(void)hipSetDevice(ctx->device);
	for (auto &ev : ctx->events) {
		(void)hipEventDestroy(ev.e0);
		(void)hipEventDestroy(ev.e1);
	}
	plan_release(ctx);
	deliver_release(ctx);
	for (auto &ar : ctx->arranged) {
		(void)hipFree(ar.d_codes);
		(void)hipFree(ar.d_off);
		(void)hipFree(ar.d_rowmap);
		(void)hipFree(ar.d_posmap);
	}
	(void)hipFree(ctx->d_codes);
	(void)hipFree(ctx->d_meta);
	(void)hipFree(ctx->d_off);
	(void)hipFree(ctx->d_sub);
	(void)hipFree(ctx->d_sub8);
	(void)hipFree(ctx->d_scratch);
	(void)hipFree(ctx->d_counters);
	for (hipEvent_t ev : ctx->slot_done)
		if (ev)
			(void)hipEventDestroy(ev);
	(void)hipFree(ctx->d_long_scratch);
	for (int k = 0; k < sa_ctx::NSIDE; k++) {
		if (ctx->side[k])
			(void)hipStreamDestroy(ctx->side[k]);
		if (ctx->join_ev[k])
			(void)hipEventDestroy(ctx->join_ev[k]);
	}
	if (ctx->fork_ev)
		(void)hipEventDestroy(ctx->fork_ev);
	delete ctx;
}

extern "C" int64_t sa_ctx_pairs(const sa_ctx *ctx) { return ctx ? ctx->pairs : -1; }

/* Arranged copy of the store for the row streams of the packed kernels (sa_systolic_pk.inc).
 *
 * A workgroup-tile streams SA_PK_WPB waves x ng streams x ch sequences.  What a terminator costs a wave is the event
 * step it causes (the frame shift of every register, the capture of a score pair), and the step is shared by all
 * streams of the wave whose terminators pass at the same stream position; streams of different lengths also leave
 * bubbles at the end of a tile.  So the rows of the matrix are cut into aligned blocks of `block` sequences (a multiple
 * of the tile's rows) and every block is re-ordered for this tile shape:
 *   - sequences of equal length are taken ng at a time: a PURE round, one sequence for each stream of a wave;
 *   - what is left over (< ng per length) is sorted by length and cut into MIXED rounds of ng neighbours;
 *   - the rounds, longest first and the mixed ones last, are dealt over the block's wave slots boustrophedon, one per
 *     slot and pass: every wave gets ch rounds of nearly the same total length, its pure rounds first.
 * As long as a wave is in its pure rounds all its streams are in step.  Rows past the last full block keep their order.
 * The copy is (codes, off) in position order plus rowmap (position -> row) and posmap (row -> position). */
static bool arranged_store(sa_ctx *ctx, int ng, int ch, int32_t block, const sa_ctx::Arranged **out)
{
	*out = nullptr;
	for (const auto &ar : ctx->arranged)
		if (ar.ng == ng && ar.ch == ch && ar.block == block) {
			*out = &ar;
			return true;
		}
	const int32_t num = ctx->num;
	const int32_t wave_rows = ng * ch;
	if (block <= 0 || block % (wave_rows * SA_PK_WPB) != 0 || block > num)
		return true; /* no full block: nothing to arrange */
	const auto t_arr = std::chrono::steady_clock::now();
	struct Acc {
		sa_ctx *c;
		std::chrono::steady_clock::time_point t;
		~Acc() { c->setup.arrange += ms_since(t); }
	} acc{ ctx, t_arr };
	std::vector<int32_t> rowmap((size_t)num), posmap((size_t)num), off_s((size_t)num + 1);
	for (int32_t i = 0; i < num; i++)
		rowmap[(size_t)i] = i;
	const int32_t slots = block / wave_rows;
	std::vector<int32_t> idx((size_t)block), rounds, rest;
	for (int32_t b0 = 0; b0 + block <= num; b0 += block) {
		for (int32_t k = 0; k < block; k++)
			idx[(size_t)k] = b0 + k;
		std::stable_sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b) { return ctx->meta[(size_t)a].len > ctx->meta[(size_t)b].len; });
		rounds.clear();
		rest.clear();
		for (int32_t k = 0; k < block;) {
			int32_t e = k;
			while (e < block && ctx->meta[(size_t)idx[(size_t)e]].len == ctx->meta[(size_t)idx[(size_t)k]].len)
				e++;
			const int32_t pure = (e - k) / ng * ng;
			rounds.insert(rounds.end(), idx.begin() + k, idx.begin() + k + pure);
			rest.insert(rest.end(), idx.begin() + k + pure, idx.begin() + e);
			k = e;
		}
		rounds.insert(rounds.end(), rest.begin(), rest.end()); /* (block and the pure part are multiples of ng) */
		for (int32_t r = 0; r < block / ng; r++) {
			const int32_t pass = r / slots, w = r % slots;
			const int32_t slot = (pass & 1) ? slots - 1 - w : w;
			for (int g = 0; g < ng; g++) {
				/* mixed rounds alternate their direction, so that the streams of a wave even out */
				const int gg = (pass & 1) ? ng - 1 - g : g;
				rowmap[(size_t)(b0 + slot * wave_rows + gg * ch + pass)] = rounds[(size_t)(r * ng + g)];
			}
		}
	}
	std::vector<uint8_t> codes_s(ctx->codes.size());
	off_s[0] = 0;
	for (int32_t p = 0; p < num; p++) {
		const int32_t i = rowmap[(size_t)p];
		posmap[(size_t)i] = p;
		const int32_t n = ctx->off[(size_t)i + 1] - ctx->off[(size_t)i];
		memcpy(codes_s.data() + off_s[(size_t)p], ctx->codes.data() + ctx->off[(size_t)i], (size_t)n);
		off_s[(size_t)p + 1] = off_s[(size_t)p] + n;
	}
	sa_ctx::Arranged ar;
	ar.ng = ng;
	ar.ch = ch;
	ar.block = block;
	bool ok = false;
	do {
		SA_HIP_CHECK(hipMalloc(&ar.d_codes, codes_s.size()), break);
		SA_HIP_CHECK(hipMalloc(&ar.d_off, sizeof(int32_t) * off_s.size()), break);
		SA_HIP_CHECK(hipMalloc(&ar.d_rowmap, sizeof(int32_t) * rowmap.size()), break);
		SA_HIP_CHECK(hipMalloc(&ar.d_posmap, sizeof(int32_t) * posmap.size()), break);
		SA_HIP_CHECK(hipMemcpy(ar.d_codes, codes_s.data(), codes_s.size(), hipMemcpyHostToDevice), break);
		SA_HIP_CHECK(hipMemcpy(ar.d_off, off_s.data(), sizeof(int32_t) * off_s.size(), hipMemcpyHostToDevice), break);
		SA_HIP_CHECK(hipMemcpy(ar.d_rowmap, rowmap.data(), sizeof(int32_t) * rowmap.size(), hipMemcpyHostToDevice), break);
		SA_HIP_CHECK(hipMemcpy(ar.d_posmap, posmap.data(), sizeof(int32_t) * posmap.size(), hipMemcpyHostToDevice), break);
		ok = true;
	} while (0);
	if (!ok) {
		(void)hipFree(ar.d_codes);
		(void)hipFree(ar.d_off);
		(void)hipFree(ar.d_rowmap);
		(void)hipFree(ar.d_posmap);
		return false;
	}
	ctx->arranged.push_back(ar);
	*out = &ctx->arranged.back();
	return true;
}

/* j of packed index p: largest j with j(j-1)/2 <= p */
static int32_t column_of(int64_t p)
{
	int64_t j = (int64_t)((1.0 + std::sqrt(1.0 + 8.0 * (double)p)) * 0.5);
	while (j * (j - 1) / 2 > p)
		--j;
	while ((j + 1) * j / 2 <= p)
		++j;
	return (int32_t)j;
}

/* ---- pair-space planning (host only) ---------------------------------------------------- */
namespace {
struct PairPlan {
	int32_t num = 0;
	int64_t pairs = 0;
	const sa_meta *meta = nullptr;
	std::vector<int64_t> len_prefix;  /* P[k] = sum_{i<k} len_i            (k = 0..N) */
	std::vector<int64_t> cell_prefix; /* C[j] = sum_{j'<j} len_j' * P[j']  (j = 0..N) */

	PairPlan(const sa_meta *m, int32_t n) : num(n), pairs((int64_t)n * (n - 1) / 2), meta(m)
	{
		len_prefix.assign((size_t)n + 1, 0);
		cell_prefix.assign((size_t)n + 1, 0);
		for (int32_t k = 0; k < n; k++) {
			len_prefix[(size_t)k + 1] = len_prefix[(size_t)k] + m[k].len;
			cell_prefix[(size_t)k + 1] = cell_prefix[(size_t)k] + (int64_t)m[k].len * len_prefix[(size_t)k];
		}
	}
	/* DP cells of all pairs with packed index < p */
	int64_t cells_before(int64_t p) const
	{
		if (p <= 0)
			return 0;
		if (p >= pairs)
			return cell_prefix[(size_t)num];
		const int32_t j = column_of(p);
		const int64_t i = p - (int64_t)j * (j - 1) / 2;
		return cell_prefix[(size_t)j] + (int64_t)meta[j].len * len_prefix[(size_t)i];
	}
};
} // namespace

extern "C" int64_t sa_pairs_cells(const struct sa_meta *meta, int32_t num, int64_t start, int64_t count)
{
	if (!meta || num < 2)
		return -1;
	PairPlan plan(meta, num);
	if (start < 0 || count < 0 || start + count > plan.pairs)
		return -1;
	return plan.cells_before(start + count) - plan.cells_before(start);
}

extern "C" int sa_pairs_partition(const struct sa_meta *meta, int32_t num, int parts, int64_t *bounds)
{
	if (!meta || num < 2 || parts < 1 || !bounds) {
		sa_set_error("sa_pairs_partition: bad arguments");
		return 1;
	}
	PairPlan plan(meta, num);
	const int64_t total = plan.cell_prefix[(size_t)num];
	bounds[0] = 0;
	for (int k = 1; k < parts; k++) {
		const int64_t target = (int64_t)((__int128)total * k / parts);
		int64_t lo = bounds[k - 1], hi = plan.pairs; /* first p with cells_before(p) >= target */
		while (lo < hi) {
			const int64_t mid = lo + (hi - lo) / 2;
			if (plan.cells_before(mid) >= target)
				hi = mid;
			else
				lo = mid + 1;
		}
		bounds[k] = lo;
	}
	bounds[parts] = plan.pairs;
	return 0;
}

extern "C" void sa_ctx_timing(sa_ctx *ctx, int enable)
{
	if (!ctx)
		return;
	(void)hipSetDevice(ctx->device);
	for (auto &ev : ctx->events) {
		(void)hipEventDestroy(ev.e0);
		(void)hipEventDestroy(ev.e1);
	}
	ctx->events.clear();
	ctx->timing = enable != 0;
}

extern "C" int sa_ctx_timing_read(sa_ctx *ctx, char *kernel_name, int cap, int64_t *launches, double *total_ms,
				  int64_t *pairs, int64_t *cells, double *all_kernels_ms)
{
	if (!ctx)
		return 1;
	(void)hipSetDevice(ctx->device);
	struct Acc {
		double ms = 0;
		int64_t n = 0, pairs = 0, cells = 0;
	};
	std::vector<std::pair<std::string, Acc>> acc;
	double all = 0.0;
	for (auto &ev : ctx->events) {
		SA_HIP_CHECK(hipEventSynchronize(ev.e1), return 1);
		float t = 0.f;
		SA_HIP_CHECK(hipEventElapsedTime(&t, ev.e0, ev.e1), return 1);
		all += t;
		auto it = std::find_if(acc.begin(), acc.end(), [&](const auto &p) { return p.first == ev.name; });
		if (it == acc.end()) {
			acc.emplace_back(ev.name, Acc());
			it = acc.end() - 1;
		}
		it->second.ms += t;
		it->second.n++;
		it->second.pairs += ev.pairs;
		it->second.cells += ev.cells;
	}
	const std::pair<std::string, Acc> *dom = nullptr;
	for (auto &p : acc)
		if (!dom || p.second.ms > dom->second.ms)
			dom = &p;
	if (kernel_name && cap > 0) {
		strncpy(kernel_name, dom ? dom->first.c_str() : "", (size_t)cap - 1);
		kernel_name[cap - 1] = 0;
	}
	if (launches)
		*launches = dom ? dom->second.n : 0;
	if (total_ms)
		*total_ms = dom ? dom->second.ms : 0.0;
	if (pairs)
		*pairs = dom ? dom->second.pairs : 0;
	if (cells)
		*cells = dom ? dom->second.cells : 0;
	if (all_kernels_ms)
		*all_kernels_ms = all;
	return 0;
}

/* ---- launch planning for a packed range ------------------------------------------------- */

static bool pk_arranged_levels(sa_ctx *ctx, int pk_g, int32_t chunk_pk, bool host_out, SaArranged (&lv)[SA_PK_SORT_LEVELS]);

/* world = 0: the plan of sa_ctx_align_range (every tile, packed order).  world >= 1: a SHARE plan -- the same tile lists
 * dealt over `world` ranks, dense tile-order output, placement segments (sa_ctx_align_share / sa_ctx_place_shares).
 * share_host: the scores (also) go straight to host memory -- arranged tiles are then their own blocks and leave in row
 * order; the arranged copies a plan's classes point to depend on it, so it is part of the plan's key. */
static bool plan_build(sa_ctx *ctx, int64_t start, int64_t count, int world, bool share_host)
{
	for (auto &pl : ctx->plans)
		if (pl.start == start && pl.count == count && pl.world == world && pl.share_host == share_host) {
			pl.stamp = ++ctx->plan_clock;
			ctx->plan = &pl;
			return true;
		}
	const auto t_plan = std::chrono::steady_clock::now();
	const double arrange_before = ctx->setup.arrange;
	struct Acc {
		sa_ctx *c;
		std::chrono::steady_clock::time_point t;
		double a0;
		~Acc() { c->setup.plan += ms_since(t) - (c->setup.arrange - a0); }
	} acc{ ctx, t_plan, arrange_before };
	constexpr size_t MAX_PLANS = 32;
	if (ctx->plans.capacity() < MAX_PLANS)
		ctx->plans.reserve(MAX_PLANS); /* pointers into the vector stay valid */
	if (ctx->plans.size() >= MAX_PLANS) { /* evict the least recently used (hipFree waits for its users) */
		size_t victim = 0;
		for (size_t k = 1; k < ctx->plans.size(); k++)
			if (ctx->plans[k].stamp < ctx->plans[victim].stamp)
				victim = k;
		plan_free(ctx->plans[victim]);
		ctx->plans.erase(ctx->plans.begin() + (long)victim);
	}
	ctx->plan = nullptr;
	const int64_t end = start + count;
	std::vector<std::vector<int32_t>> jl((size_t)SA_PLAN_NCLASSES), tp((size_t)SA_PLAN_NCLASSES);
	std::vector<int64_t> cpairs((size_t)SA_PLAN_NCLASSES, 0), ccells((size_t)SA_PLAN_NCLASSES, 0);
	std::vector<std::vector<std::pair<int32_t, int32_t>>> rows_of((size_t)SA_PLAN_NCLASSES); /* packed classes: [ia, ib) per column */
	std::vector<int64_t> lenpre((size_t)ctx->num + 1, 0);
	for (int32_t k = 0; k < ctx->num; k++)
		lenpre[(size_t)k + 1] = lenpre[(size_t)k] + ctx->meta[(size_t)k].len;
	sa_ctx::Plan plan;
	/* Tile granularity: a wave-tile streams `chunk` sequences per lane group.  64 amortizes the per-tile
	 * setup best; small ranges (one rank's share at 8 GPUs, a super-chunk of an overlapped schedule) get
	 * shorter streams so that there are still several tiles per wave slot. */
	{
		const int64_t want_tiles = (int64_t)ctx->persistent_wgs * 6;
		const int64_t mine = world > 1 ? count / world : count; /* pairs one rank runs */
		int32_t chunk = SA_SYS_CHUNK;
		while (chunk > 8 && mine / (4 * chunk) < want_tiles)
			chunk >>= 1;
		if (ctx->env_chunk)
			chunk = ctx->env_chunk;
		plan.chunk = chunk;
		/* packed classes: a workgroup-tile covers 2 columns x SA_PK_WPB * 8 streams of `chunk` sequences and ~4
		 * workgroups are resident per CU: keep >= 5 tiles per resident workgroup so that a small range (one rank's
		 * share at 8 GPUs, a super-chunk) still drains evenly */
		/* (measured, cfg 2: quarter tiles cost 5 % more than full ones -- 15.37 against 14.61 ms for the whole range -- and
		 * half tiles 1 %; one rank's share at 8 ranks takes 2.06 ms in quarter tiles, 1.98 in half or full tiles) */
		const int64_t want_pk = (int64_t)ctx->persistent_wgs * 5 / 8; /* 32 x CUs x 5 / 8 = 5 x (4 workgroups per CU) */
		int32_t cpk = SA_SYS_CHUNK;
		while (cpk > 4 && mine / ((int64_t)2 * SA_PK_WPB * 8 * cpk) < want_pk)
			cpk >>= 1;
		if (ctx->env_chunk)
			cpk = ctx->env_chunk;
		cpk = std::min(cpk, ctx->pk_chunk_cap);
		plan.chunk_pk = cpk;
		/* Two tile sizes for a launch that gives a workgroup slot fewer than 16 tiles (one rank's share of a multi-GPU
		 * run; a super-chunk): when the tiles run out the slots finish their last ones over a whole tile's duration, and
		 * only work in small units can fill that triangle.  So the bulk runs in tiles as large as leave >= 2.5 per slot,
		 * and the lowest columns of the range -- a fifth of its pairs -- in tiles a quarter of that size, which the
		 * launch order puts last (small tiles cost more per row, +5 % at a quarter of the full size, so not everywhere). */
		const int64_t slots = (int64_t)ctx->persistent_wgs / 8;
		const int64_t below = getenv("SA_HIP_SMALL_BELOW") ? atoi(getenv("SA_HIP_SMALL_BELOW")) : 16; /* (experiments) */
		if (!ctx->env_chunk && !getenv("SA_HIP_ONE_TILE_SIZE") && mine / ((int64_t)2 * SA_PK_WPB * 8 * cpk) < below * slots) {
			int32_t big = SA_SYS_CHUNK;
			while (big > 16 && 2 * mine / ((int64_t)2 * SA_PK_WPB * 8 * big) < 5 * slots)
				big >>= 1;
			big = std::min(std::max(big, cpk), ctx->pk_chunk_cap);
			if (big >= 16) {
				plan.chunk_pk = big;
				const int sdiv = getenv("SA_HIP_SMALL_DIV") ? std::max(2, atoi(getenv("SA_HIP_SMALL_DIV"))) : 4;   /* (experiments) */
				const int sfrac = getenv("SA_HIP_SMALL_FRAC") ? std::max(2, atoi(getenv("SA_HIP_SMALL_FRAC"))) : 5;
				plan.chunk_pk_small = std::max(4, big / sdiv);
				const int64_t jlo = column_of(start), jhi = column_of(end - 1) + 1;
				int64_t lo = jlo, hi = jhi; /* smallest column with >= a fifth of the range's pairs below it */
				while (lo < hi) {
					const int64_t mid = (lo + hi) / 2;
					if (mid * (mid - 1) / 2 - start >= count / sfrac)
						hi = mid;
					else
						lo = mid + 1;
				}
				plan.j_small = (int32_t)lo;
			}
		}
	}
	const int32_t j0 = column_of(start), j1 = column_of(end - 1);
	for (int32_t j = j0; j <= j1; j++) {
		const int64_t tri = (int64_t)j * (j - 1) / 2;
		const int64_t ia = std::max<int64_t>(0, start - tri), ib = std::min<int64_t>(j, end - tri);
		if (ib <= ia)
			continue;
		const int32_t n = ctx->meta[(size_t)j].len;
		const int k8 = (n + 7) / 8, k16 = (n + 15) / 16;
		if (k8 <= ctx->pk_kmax || (k16 >= SA_PK_K16_MIN && k16 <= ctx->pk16_kmax)) {
			/* packed-u16 class K = ceil(n / 8) (8-lane groups) or ceil(n / 16) (16-lane groups): tiles are counted per
			 * column PAIR below */
			const size_t pc = (size_t)(k8 <= ctx->pk_kmax ? SA_PK_CLASS0 + k8 : SA_PK16_CLASS0 + k16) + (j < plan.j_small ? SA_PK_SMALL : 0);
			jl[pc].push_back(j);
			rows_of[pc].emplace_back((int32_t)ia, (int32_t)ib);
			cpairs[pc] += ib - ia;
			ccells[pc] += (int64_t)n * (lenpre[(size_t)ib] - lenpre[(size_t)ia]);
			continue;
		}
		const int cls = ctx->sys_ok ? systolic_class_for(n) : -1;
		if (cls < 0) {
			if (!plan.generic.empty() && plan.generic.back().first + plan.generic.back().second == tri + ia)
				plan.generic.back().second += ib - ia;
			else
				plan.generic.emplace_back(tri + ia, ib - ia);
			continue;
		}
		/* strip-mined tiles stream one group of at most 16 sequences (their scratch lines are per position) */
		const int rows = cls == SA_SYS_CLASS_LONG ? SA_SYS_WPB(64, true) * std::min(plan.chunk, 16)
							   : SA_SYS_WPB(SA_SYS_CLASSES[cls].G, false) * (64 / SA_SYS_CLASSES[cls].G) * plan.chunk;
		if (tp[(size_t)cls].empty())
			tp[(size_t)cls].push_back(0);
		const int64_t tiles = (ib - ia + rows - 1) / rows;
		if ((int64_t)tp[(size_t)cls].back() + tiles > INT32_MAX) {
			sa_set_error("packed range too large for one launch; split it into smaller ranges");
			return false;
		}
		jl[(size_t)cls].push_back(j);
		tp[(size_t)cls].push_back(tp[(size_t)cls].back() + (int32_t)tiles);
		cpairs[(size_t)cls] += ib - ia;
		ccells[(size_t)cls] += (int64_t)n * (lenpre[(size_t)ib] - lenpre[(size_t)ia]);
	}
	/* packed classes: consecutive columns of a class share a tile (sa_systolic_pk.inc): rows = the union of their row
	 * ranges, SA_PK_WPB * 8 streams of `chunk` sequences per workgroup-tile.  The prefix counts the full tiles; the
	 * partial last tiles follow them in the tile numbering, largest first (their pairs are appended to the prefix). */
	std::vector<int32_t> nparts((size_t)SA_PLAN_NCLASSES, 0);
	std::vector<std::vector<int32_t>> part_rows((size_t)SA_PLAN_NCLASSES);
	for (int cls = SA_PK_CLASS0; cls < SA_PLAN_NCLASSES; cls++) {
		const auto &rw = rows_of[(size_t)cls];
		if (rw.empty())
			continue;
		const PkCls pc = pk_decode(cls);
		const int64_t rows = (int64_t)SA_PK_WPB * (64 / pc.g) * (pc.small ? plan.chunk_pk_small : plan.chunk_pk);
		tp[(size_t)cls].push_back(0);
		std::vector<std::pair<int32_t, int32_t>> parts; /* (rows, pair) */
		for (size_t c = 0; c < rw.size(); c += 2) {
			const auto &a = rw[c], &b = rw[c + 1 < rw.size() ? c + 1 : c];
			const int64_t span = std::max(a.second, b.second) - std::min(a.first, b.first);
			const int64_t tiles = span / rows;
			if ((int64_t)tp[(size_t)cls].back() + tiles + (int64_t)parts.size() + 1 > INT32_MAX) {
				sa_set_error("packed range too large for one launch; split it into smaller ranges");
				return false;
			}
			tp[(size_t)cls].push_back(tp[(size_t)cls].back() + (int32_t)tiles);
			if (span % rows)
				parts.emplace_back((int32_t)(span % rows), (int32_t)(c / 2));
		}
		std::stable_sort(parts.begin(), parts.end(), [](const auto &x, const auto &y) { return x.first > y.first; });
		for (const auto &pt : parts) {
			tp[(size_t)cls].push_back(pt.second);
			part_rows[(size_t)cls].push_back(pt.first);
		}
		nparts[(size_t)cls] = (int32_t)parts.size();
	}
	bool ok = true;
	for (int cls = 0; cls < SA_PLAN_NCLASSES && ok; cls++) {
		if (jl[(size_t)cls].empty())
			continue;
		sa_ctx::ClassLaunch cl;
		cl.cls = cls;
		cl.ncols = (int32_t)jl[(size_t)cls].size();
		cl.npart = nparts[(size_t)cls];
		cl.ntiles = cl.npart ? tp[(size_t)cls][tp[(size_t)cls].size() - 1 - (size_t)cl.npart] + cl.npart : tp[(size_t)cls].back();
		cl.pairs = cpairs[(size_t)cls];
		cl.cells = ccells[(size_t)cls];
		cl.part_rows = part_rows[(size_t)cls];
		cl.chunk = cls >= SA_PK_CLASS0 ? (pk_decode(cls).small ? plan.chunk_pk_small : plan.chunk_pk) : 0;
		if (cl.ntiles > (1 << SA_PK_UTILE_BITS) && cls >= SA_PK_CLASS0) {
			sa_set_error("packed range too large for one launch; split it into smaller ranges");
			ok = false;
		}
		SA_HIP_CHECK(hipMalloc(&cl.d_jlist, sizeof(int32_t) * jl[(size_t)cls].size()), ok = false);
		if (ok) {
			SA_HIP_CHECK(hipMalloc(&cl.d_tprefix, sizeof(int32_t) * tp[(size_t)cls].size()), ok = false);
		}
		if (ok) {
			SA_HIP_CHECK(hipMemcpy(cl.d_jlist, jl[(size_t)cls].data(), sizeof(int32_t) * jl[(size_t)cls].size(),
					       hipMemcpyHostToDevice), ok = false);
		}
		if (ok) {
			SA_HIP_CHECK(hipMemcpy(cl.d_tprefix, tp[(size_t)cls].data(), sizeof(int32_t) * tp[(size_t)cls].size(),
					       hipMemcpyHostToDevice), ok = false);
		}
		plan.classes.push_back(cl);
	}
	/* order of the classes (s32 launches; the share plans' dealing order): the class with the most DP work first */
	std::stable_sort(plan.classes.begin(), plan.classes.end(),
			 [](const sa_ctx::ClassLaunch &x, const sa_ctx::ClassLaunch &y) { return x.cells > y.cells; });
	plan.start = start;
	plan.count = count;
	plan.world = world;
	plan.share_host = share_host;
	plan.stamp = ++ctx->plan_clock;
	/* ---- share plan: deal the tiles of every class over the ranks, lay out the dense shares, list the placement ---- */
	if (ok && world >= 1) {
		/* One list for the whole job, the same on every rank: classes in order, inside a class the launch's own tile
		 * order (full tiles, then the partial ones by decreasing size).  A tile goes to the rank with the least
		 * accumulated work so far (cost = row residues x per-step instruction weight of the class): the shares end
		 * within one small partial tile of each other, every rank keeps full-size tiles and whole arranged blocks. */
		std::vector<int64_t> load((size_t)world, 0), fill((size_t)world, 0);
		struct Geo {
			int32_t cls_idx, t, owner;
			int32_t j[2], ia[2], ib[2], i_begin, i_count;
			bool dup, own; /* own: an arranged tile that is its whole block (its rows are a permutation of its positions) */
			const int32_t *rowmap, *posmap;
			int64_t doff;
		};
		std::vector<Geo> geo;
		for (size_t ci = 0; ci < plan.classes.size() && ok; ci++) {
			auto &cl = plan.classes[ci];
			const int cls = cl.cls;
			const bool is_pk = cls >= SA_PK_CLASS0;
			const int pk_g = is_pk ? pk_decode(cls).g : 8;
			const int G = is_pk ? pk_g : cls == SA_SYS_CLASS_LONG ? 64 : SA_SYS_CLASSES[cls].G;
			const int K = is_pk ? pk_decode(cls).k : cls == SA_SYS_CLASS_LONG ? 16 : SA_SYS_CLASSES[cls].K;
			const int64_t weight = (int64_t)(K + 6) * (G / 8);
			const auto &J = jl[(size_t)cls];
			const auto &T = tp[(size_t)cls];
			std::vector<Geo> tiles((size_t)cl.ntiles);
			if (is_pk) {
				SaArranged lv[SA_PK_SORT_LEVELS];
				if (!pk_arranged_levels(ctx, pk_g, cl.chunk, share_host, lv)) {
					ok = false;
					break;
				}
				const int32_t lvrows[SA_PK_SORT_LEVELS] = { lv[0].rows, lv[1].rows, lv[2].rows, lv[3].rows };
				const int32_t rows = SA_PK_WPB * (64 / pk_g) * cl.chunk;
				const auto &rw = rows_of[(size_t)cls];
				const int32_t npairs = (cl.ncols + 1) / 2, nfull = T[(size_t)npairs];
				for (int32_t t = 0; t < cl.ntiles; t++) {
					int32_t lo;
					if (t < nfull)
						lo = (int32_t)(std::upper_bound(T.begin(), T.begin() + npairs + 1, t) - T.begin()) - 1;
					else
						lo = T[(size_t)(npairs + 1 + (t - nfull))];
					const size_t c0 = (size_t)2 * lo, c1 = c0 + 1 < rw.size() ? c0 + 1 : c0;
					Geo &g = tiles[(size_t)t];
					g.dup = c1 == c0;
					g.j[0] = J[c0], g.j[1] = J[c1];
					g.ia[0] = rw[c0].first, g.ib[0] = rw[c0].second;
					g.ia[1] = rw[c1].first, g.ib[1] = rw[c1].second;
					const int32_t ra = std::min(g.ia[0], g.ia[1]), rb = std::max(g.ib[0], g.ib[1]);
					const int32_t chunk = t < nfull ? t - T[(size_t)lo] : T[(size_t)lo + 1] - T[(size_t)lo];
					g.i_begin = ra + chunk * rows;
					g.i_count = std::min(rows, rb - g.i_begin);
					const int l = sa_pk_pick_level(lvrows, ra, rb, g.i_begin, rows);
					g.rowmap = l >= 0 ? lv[l].rowmap : nullptr;
					g.posmap = l >= 0 ? lv[l].posmap : nullptr;
					g.own = l >= 0 && lv[l].rows == g.i_count;
				}
			} else {
				const int rows = cls == SA_SYS_CLASS_LONG ? SA_SYS_WPB(64, true) * std::min(plan.chunk, 16)
									   : SA_SYS_WPB(G, false) * (64 / G) * plan.chunk;
				for (int32_t k = 0; k < cl.ncols; k++) {
					const int32_t j = J[(size_t)k];
					const int64_t tri = (int64_t)j * (j - 1) / 2;
					const int32_t ia = (int32_t)std::max<int64_t>(0, start - tri), ib = (int32_t)std::min<int64_t>(j, end - tri);
					for (int32_t t = T[(size_t)k]; t < T[(size_t)k + 1]; t++) {
						Geo &g = tiles[(size_t)t];
						g.dup = true; /* one column, one run */
						g.j[0] = g.j[1] = j;
						g.ia[0] = g.ia[1] = ia, g.ib[0] = g.ib[1] = ib;
						g.i_begin = ia + (t - T[(size_t)k]) * rows;
						g.i_count = std::min(rows, ib - g.i_begin);
						g.rowmap = g.posmap = nullptr;
						g.own = false;
					}
				}
			}
			std::vector<std::vector<int32_t>> mine((size_t)world);
			std::vector<int64_t> doff((size_t)cl.ntiles, 0);
			cl.rank_pairs.assign((size_t)world, 0);
			cl.rank_cells.assign((size_t)world, 0);
			cl.owner.assign((size_t)cl.ntiles, 0);
			for (int32_t t = 0; t < cl.ntiles; t++) {
				Geo &g = tiles[(size_t)t];
				/* (arranged tiles hold a permutation of their block's rows: the residue count of the position range is that
				 * of the rows only when the tile is its whole block -- close enough for a load estimate) */
				const int64_t res = lenpre[(size_t)(g.i_begin + g.i_count)] - lenpre[(size_t)g.i_begin] + g.i_count;
				int r = 0;
				for (int q = 1; q < world; q++)
					if (load[(size_t)q] < load[(size_t)r])
						r = q;
				load[(size_t)r] += res * weight;
				g.cls_idx = (int32_t)ci;
				g.t = t;
				g.owner = r;
				cl.owner[(size_t)t] = (int16_t)r;
				g.doff = fill[(size_t)r];
				doff[(size_t)t] = g.doff;
				fill[(size_t)r] += (int64_t)(g.dup ? 1 : 2) * SA_SHARE_PAD(g.i_count);
				mine[(size_t)r].push_back(t);
				for (int h = 0; h < (g.dup ? 1 : 2); h++) {
					const int64_t lo_i = std::max(g.ia[h], g.i_begin), hi_i = std::min(g.ib[h], g.i_begin + g.i_count);
					if (hi_i > lo_i) {
						cl.rank_pairs[(size_t)r] += hi_i - lo_i;
						cl.rank_cells[(size_t)r] += (int64_t)ctx->meta[(size_t)g.j[h]].len * (lenpre[(size_t)hi_i] - lenpre[(size_t)lo_i]);
					}
				}
				geo.push_back(g);
			}
			std::vector<int32_t> tl;
			cl.rank_first.assign((size_t)world + 1, 0);
			for (int r = 0; r < world; r++) {
				tl.insert(tl.end(), mine[(size_t)r].begin(), mine[(size_t)r].end());
				cl.rank_first[(size_t)r + 1] = (int32_t)tl.size();
			}
			SA_HIP_CHECK(hipMalloc(&cl.d_tlist, sizeof(int32_t) * std::max<size_t>(tl.size(), 1)), ok = false);
			if (ok) {
				SA_HIP_CHECK(hipMalloc(&cl.d_doff, sizeof(int64_t) * std::max<size_t>(doff.size(), 1)), ok = false);
			}
			if (ok && !tl.empty()) {
				SA_HIP_CHECK(hipMemcpy(cl.d_tlist, tl.data(), sizeof(int32_t) * tl.size(), hipMemcpyHostToDevice), ok = false);
			}
			if (ok && !doff.empty()) {
				SA_HIP_CHECK(hipMemcpy(cl.d_doff, doff.data(), sizeof(int64_t) * doff.size(), hipMemcpyHostToDevice), ok = false);
			}
		}
		/* what no systolic class covers: every run of the pair-per-wave kernels is cut into `world` equal pieces */
		plan.generic_share.assign((size_t)world, {});
		for (const auto &run : plan.generic) {
			const int64_t per = (run.second + world - 1) / world;
			for (int r = 0; r < world; r++) {
				const int64_t lo_p = std::min(run.second, (int64_t)r * per), hi_p = std::min(run.second, lo_p + per);
				if (hi_p <= lo_p)
					continue;
				plan.generic_share[(size_t)r].push_back({ run.first + lo_p, hi_p - lo_p, fill[(size_t)r] });
				fill[(size_t)r] += SA_SHARE_PAD(hi_p - lo_p);
			}
		}
		plan.share_elems = std::max<int64_t>(8, *std::max_element(fill.begin(), fill.end()));
		/* placement: one segment per run of a tile, pieces of the generic sub-runs */
		std::vector<SaPlaceSeg> segs;
		for (const Geo &g : geo)
			for (int h = 0; h < (g.dup ? 1 : 2); h++) {
				SaPlaceSeg sg{};
				sg.src = (int64_t)g.owner * plan.share_elems + g.doff + (int64_t)h * SA_SHARE_PAD(g.i_count);
				sg.dst = (int64_t)g.j[h] * (g.j[h] - 1) / 2 - start;
				sg.map = g.own ? g.posmap : g.rowmap;
				sg.count = g.i_count;
				sg.pos0 = g.i_begin;
				sg.ia = g.ia[h];
				sg.ib = g.ib[h];
				sg.flags = g.own ? 1 : 0;
				segs.push_back(sg);
			}
		for (int r = 0; r < world; r++)
			for (const auto &gs : plan.generic_share[(size_t)r])
				for (int64_t o = 0; o < gs.count; o += 8192) {
					SaPlaceSeg sg{};
					sg.src = (int64_t)r * plan.share_elems + gs.doff + o;
					sg.dst = gs.start + o - start;
					sg.map = nullptr;
					sg.count = (int32_t)std::min<int64_t>(8192, gs.count - o);
					sg.pos0 = 0;
					sg.ia = 0;
					sg.ib = sg.count;
					segs.push_back(sg);
				}
		plan.nsegs = (int32_t)segs.size();
		if (ok && !segs.empty()) {
			SA_HIP_CHECK(hipMalloc(&plan.d_segs, sizeof(SaPlaceSeg) * segs.size()), ok = false);
			if (ok) {
				SA_HIP_CHECK(hipMemcpy(plan.d_segs, segs.data(), sizeof(SaPlaceSeg) * segs.size(), hipMemcpyHostToDevice), ok = false);
			}
		}
	}
	/* ---- packed classes -> bundle launches ---- */
	if (ok) {
		const int nranks = std::max(world, 1);
		std::vector<int> order; /* packed classes by decreasing K inside their bundle: the launch ends on its cheapest tiles */
		for (size_t ci = 0; ci < plan.classes.size(); ci++)
			if (plan.classes[ci].cls >= SA_PK_CLASS0)
				order.push_back((int)ci);
		/* (the small-tile copies of the classes sort behind all the others: their full tiles end the full tiles) */
		std::sort(order.begin(), order.end(), [&](int x, int y) { return plan.classes[(size_t)x].cls > plan.classes[(size_t)y].cls; });
		std::stable_partition(order.begin(), order.end(), [&](int x) { return !pk_decode(plan.classes[(size_t)x].cls).small; });
		for (int ci : order) {
			const auto &cl = plan.classes[(size_t)ci];
			const int g = pk_decode(cl.cls).g;
			const int k = pk_decode(cl.cls).k;
			const int klo = sa_pk_bundle_klo(g, k);
			const int f16 = g == 8 || k <= ctx->pk16_f16_kmax ? 1 : 0;
			sa_ctx::Plan::PkLaunch *b = nullptr;
			for (auto &x : plan.pk_launches)
				if (x.g == g && x.klo == klo && x.f16 == f16)
					b = &x;
			if (!b) {
				plan.pk_launches.emplace_back();
				b = &plan.pk_launches.back();
				b->g = g, b->klo = klo, b->f16 = f16, b->kmax = k;
			}
			b->kmax = std::max(b->kmax, k);
			b->cls.push_back(ci);
		}
		for (auto &b : plan.pk_launches) {
			std::vector<SaPkClassArgs> args(b.cls.size());
			for (size_t x = 0; x < b.cls.size(); x++) {
				const auto &cl = plan.classes[(size_t)b.cls[x]];
				const int k = pk_decode(cl.cls).k;
				SaPkClassArgs &a = args[x];
				if (!pk_arranged_levels(ctx, b.g, cl.chunk, share_host, a.lv)) {
					ok = false;
					break;
				}
				a.chunk = cl.chunk;
				a.jlist = cl.d_jlist;
				a.tprefix = cl.d_tprefix;
				a.dense_off = world >= 1 ? cl.d_doff : nullptr;
				a.ncols = cl.ncols;
				a.npart = cl.npart;
				a.k = k;
				a.delta = pk_delta(ctx, b.g, k);
				a.pk_base = pk_base(ctx, b.g, k);
			}
			if (!ok)
				break;
			/* walking order, per rank: the large full tiles class after class, then everything smaller -- the full tiles of
			 * the small-tile classes and the partial tiles of all classes -- by decreasing work (rows x per-step weight) */
			std::vector<uint32_t> ul;
			b.ufirst.assign((size_t)nranks + 1, 0);
			b.nlocal.assign((size_t)nranks, 0);
			b.pairs.assign((size_t)nranks, 0);
			b.cells.assign((size_t)nranks, 0);
			struct Part {
				int64_t work;
				uint32_t code, pair;
			};
			std::vector<Part> parts;
			for (int r = 0; r < nranks; r++) {
				parts.clear();
				for (size_t x = 0; x < b.cls.size(); x++) {
					const auto &cl = plan.classes[(size_t)b.cls[x]];
					const int32_t nfull = cl.ntiles - cl.npart;
					const auto &T = tp[(size_t)cl.cls];
					const int32_t npairs = (cl.ncols + 1) / 2;
					int32_t pair_of_full = 0; /* (tiles ascend: the pair index only moves forward) */
					for (int32_t t = 0; t < cl.ntiles; t++) {
						uint32_t pair;
						if (t < nfull) {
							while (pair_of_full + 1 < npairs && T[(size_t)pair_of_full + 1] <= t)
								pair_of_full++;
							pair = (uint32_t)pair_of_full;
						} else {
							pair = (uint32_t)T[(size_t)(npairs + 1 + (t - nfull))];
						}
						if (world >= 1 && cl.owner[(size_t)t] != r)
							continue;
						const uint32_t code = ((uint32_t)x << SA_PK_UTILE_BITS) | (uint32_t)t;
						const int64_t full_rows = (int64_t)SA_PK_WPB * (64 / b.g) * cl.chunk;
						if (t < nfull && !pk_decode(cl.cls).small) {
							ul.push_back(code);
							ul.push_back(pair);
						} else /* small full tiles and every partial tile: by decreasing work, after the large full tiles */
							parts.push_back({ (t < nfull ? full_rows : (int64_t)cl.part_rows[(size_t)(t - nfull)]) * (args[x].k + 4), code, pair });
					}
					b.pairs[(size_t)r] += world >= 1 ? cl.rank_pairs[(size_t)r] : cl.pairs;
					b.cells[(size_t)r] += world >= 1 ? cl.rank_cells[(size_t)r] : cl.cells;
				}
				std::stable_sort(parts.begin(), parts.end(), [](const Part &p, const Part &q) { return p.work > q.work; });
				for (const Part &pt : parts) {
					ul.push_back(pt.code);
					ul.push_back(pt.pair);
				}
				b.ufirst[(size_t)r + 1] = (int64_t)ul.size(); /* (words: two per tile) */
				const int64_t n = (b.ufirst[(size_t)r + 1] - b.ufirst[(size_t)r]) / 2;
				if (n > INT32_MAX) {
					sa_set_error("packed range too large for one launch; split it into smaller ranges");
					ok = false;
					break;
				}
				b.nlocal[(size_t)r] = (int32_t)n;
			}
			if (ok) {
				SA_HIP_CHECK(hipMalloc(&b.d_args, sizeof(SaPkClassArgs) * args.size()), ok = false);
			}
			if (ok) {
				SA_HIP_CHECK(hipMemcpy(b.d_args, args.data(), sizeof(SaPkClassArgs) * args.size(), hipMemcpyHostToDevice), ok = false);
			}
			if (ok) {
				SA_HIP_CHECK(hipMalloc(&b.d_ulist, sizeof(uint32_t) * std::max<size_t>(ul.size(), 1)), ok = false);
			}
			if (ok && !ul.empty()) {
				SA_HIP_CHECK(hipMemcpy(b.d_ulist, ul.data(), sizeof(uint32_t) * ul.size(), hipMemcpyHostToDevice), ok = false);
			}
			if (!ok)
				break;
		}
	}
	if (!ok) {
		plan_free(plan);
		return false;
	}
	ctx->plans.push_back(plan);
	ctx->plan = &ctx->plans.back();
	return true;
}

static const char *const METHOD_TAG[] = { "nw", "ga", "sw" };

/* share: world >= 1 runs the tiles of `rank` only and stores them densely (sa_ctx_align_share); 0: the whole range */
static int align_range_impl(sa_ctx *ctx, int64_t start, int64_t count, int32_t *d_scores, void *stream, bool out16,
			    int world = 0, int rank = 0, int32_t *host_out = nullptr);

extern "C" int sa_ctx_align_range(sa_ctx *ctx, int64_t start, int64_t count, int32_t *d_scores, void *stream)
{
	return align_range_impl(ctx, start, count, d_scores, stream, false);
}

/* Largest |score| any pair of this store can reach under this scoring, from lengths, matrix extremes and gaps */
static int64_t score_magnitude_bound(const sa_ctx *ctx)
{
	int64_t smax = 0;
	for (int k = 0; k < SA_SUB_DIM * SA_SUB_DIM; k++)
		smax = std::max<int64_t>(smax, std::llabs((long long)ctx->sc.sub[k]));
	const int64_t L = ctx->max_len;
	const int64_t gap = std::max<int64_t>(std::llabs((long long)ctx->sc.gap_pen),
					      std::max<int64_t>(std::llabs((long long)ctx->sc.gap_opn), std::llabs((long long)ctx->sc.gap_ext)));
	/* an alignment path has at most L substitution steps and at most 2L gap steps, each gap step costing at
	 * most `gap` (an opened gap costs open OR extend per position in this model) */
	return L * smax + 2 * L * gap;
}

extern "C" int sa_ctx_scores_fit16(const sa_ctx *ctx)
{
	return ctx && score_magnitude_bound(ctx) <= 32767 ? 1 : 0;
}

extern "C" int sa_ctx_align_range16(sa_ctx *ctx, int64_t start, int64_t count, int16_t *d_scores, void *stream)
{
	if (!sa_ctx_scores_fit16(ctx)) {
		sa_set_error("sa_ctx_align_range16: scores of this store and scoring are not provably within int16");
		return 1;
	}
	return align_range_impl(ctx, start, count, reinterpret_cast<int32_t *>(d_scores), stream, true);
}

extern "C" int sa_hip_widen16(const int16_t *d_src, int32_t *d_dst, int64_t count, void *stream)
{
	if (count < 0 || (count && (!d_src || !d_dst))) {
		sa_set_error("sa_hip_widen16: bad arguments");
		return 1;
	}
	SA_HIP_CHECK(sa_launch_widen16(d_src, d_dst, count, (hipStream_t)stream), return 1);
	return 0;
}

/* ---- tile-interleaved sharding: one process per GPU, dense shares, all-gather, widen-and-place ---- */
static bool share_args_ok(sa_ctx *ctx, int64_t start, int64_t count, int world, const char *who)
{
	if (!ctx || start < 0 || count <= 0 || start + count > ctx->pairs || world < 1 || world > 1024) {
		sa_set_error("%s: bad range [%lld,+%lld) of %lld pairs or world %d", who, (long long)start, (long long)count,
			     ctx ? (long long)ctx->pairs : -1LL, world);
		return false;
	}
	return true;
}

extern "C" void sa_ctx_leave_room(sa_ctx *ctx, int on)
{
	if (ctx)
		ctx->leave_room = on != 0;
}

extern "C" int64_t sa_ctx_share_elems(sa_ctx *ctx, int64_t start, int64_t count, int world, int to_host)
{
	if (!share_args_ok(ctx, start, count, world, "sa_ctx_share_elems"))
		return -1;
	SA_HIP_CHECK(hipSetDevice(ctx->device), return -1);
	if (!plan_build(ctx, start, count, world, to_host != 0))
		return -1;
	return ctx->plan->share_elems;
}

extern "C" int sa_ctx_align_share(sa_ctx *ctx, int64_t start, int64_t count, int world, int rank, void *d_share, int elem16,
				   int32_t *host_packed, void *stream)
{
	if (!share_args_ok(ctx, start, count, world, "sa_ctx_align_share"))
		return 1;
	if (rank < 0 || rank >= world || !d_share) {
		sa_set_error("sa_ctx_align_share: rank %d of %d, share buffer %p", rank, world, d_share);
		return 1;
	}
	if (elem16 && !sa_ctx_scores_fit16(ctx)) {
		sa_set_error("sa_ctx_align_share: scores of this store and scoring are not provably within int16");
		return 1;
	}
	int32_t *host_dev = nullptr;
	if (host_packed) { /* the kernels store through the device-visible alias of the page-locked matrix */
		SA_HIP_CHECK(hipSetDevice(ctx->device), return 1);
		void *dp = nullptr;
		if (!host_range_is_pinned(host_packed + start) || !host_range_is_pinned(host_packed + start + count - 1) ||
		    hipHostGetDevicePointer(&dp, host_packed + start, 0) != hipSuccess) {
			(void)hipGetLastError();
			sa_set_error("sa_ctx_align_share: the host matrix is not page-locked over the range (sa_hip_host_register)");
			return 1;
		}
		host_dev = static_cast<int32_t *>(dp);
	}
	return align_range_impl(ctx, start, count, static_cast<int32_t *>(d_share), stream, elem16 != 0, world, rank, host_dev);
}

extern "C" int sa_ctx_place_shares(sa_ctx *ctx, int64_t start, int64_t count, int world, int to_host, const void *d_shares,
				    int elem16, int32_t *d_packed, void *stream)
{
	if (!share_args_ok(ctx, start, count, world, "sa_ctx_place_shares"))
		return 1;
	if (!d_shares || !d_packed) {
		sa_set_error("sa_ctx_place_shares: null buffer");
		return 1;
	}
	SA_HIP_CHECK(hipSetDevice(ctx->device), return 1);
	if (!plan_build(ctx, start, count, world, to_host != 0))
		return 1;
	SA_HIP_CHECK(sa_launch_place(ctx->plan->d_segs, ctx->plan->nsegs, d_shares, elem16, d_packed, (hipStream_t)stream), return 1);
	return 0;
}

/* Arranged row streams of one packed launch of the current plan (builds the copies it needs on first use).  Scores
 * stored straight into host memory must leave in row order: there a block is one tile; in device memory a block may
 * span several tiles (their stores scatter inside it). */
static bool pk_arranged_levels(sa_ctx *ctx, int pk_g, int32_t chunk_pk, bool host_out, SaArranged (&lv)[SA_PK_SORT_LEVELS])
{
	for (auto &x : lv)
		x = SaArranged{};
	if (ctx->env_no_sort)
		return true;
	const int ng = 64 / pk_g;
	const int32_t rows = SA_PK_WPB * ng * chunk_pk;
	int nl = 0;
	for (int l = 0; l < SA_PK_SORT_LEVELS; l++) {
		const int32_t block = SA_PK_SORT_ROWS >> l;
		/* (a tile of SA_PK_ROWS_OWN_BLOCK rows has enough equal lengths of its own, and storing in row order keeps
		 * the HBM write traffic at the algorithmic 4 bytes per pair) */
		if (host_out || rows >= SA_PK_ROWS_OWN_BLOCK || block <= rows || block % rows != 0)
			continue;
		const sa_ctx::Arranged *ar = nullptr;
		if (!arranged_store(ctx, ng, chunk_pk, block, &ar))
			return false;
		if (ar)
			lv[nl++] = { ar->d_codes, ar->d_off, ar->d_rowmap, ar->d_posmap, ar->block };
	}
	if (nl < SA_PK_SORT_LEVELS) { /* the tile itself as a block: its scores leave in row order */
		const sa_ctx::Arranged *ar = nullptr;
		if (!arranged_store(ctx, ng, chunk_pk, rows, &ar))
			return false;
		if (ar)
			lv[nl++] = { ar->d_codes, ar->d_off, ar->d_rowmap, ar->d_posmap, ar->block };
	}
	return true;
}

/* Host-side preparation of a range's launches -- the plan and the arranged copies of the store -- so that a caller
 * that times the launch/copy loop (sa_ctx_align_host) can do it with its other set-up: like the uploads, it is input
 * preparation, not alignment.  Idempotent; align_range_impl does the same on demand. */
static bool prepare_range(sa_ctx *ctx, int64_t start, int64_t count, bool host_out)
{
	if (count <= 0)
		return true;
	if (!plan_build(ctx, start, count, 0, host_out))
		return false;
	const bool was = ctx->out_is_host;
	ctx->out_is_host = host_out;
	bool ok = true;
	for (const auto &cl : ctx->plan->classes) {
		if (cl.cls < SA_PK_CLASS0)
			continue;
		SaArranged lv[SA_PK_SORT_LEVELS];
		ok = pk_arranged_levels(ctx, pk_decode(cl.cls).g, cl.chunk, host_out, lv);
		if (!ok)
			break;
	}
	ctx->out_is_host = was;
	return ok;
}

static int align_range_impl(sa_ctx *ctx, int64_t start, int64_t count, int32_t *d_scores, void *stream, bool out16,
			    int world, int rank, int32_t *host_out)
{
	if (!ctx || start < 0 || count < 0 || start + count > ctx->pairs || (!d_scores && count)) {
		sa_set_error("sa_ctx_align_range: bad range [%lld,+%lld) of %lld pairs", (long long)start,
			     (long long)count, ctx ? (long long)ctx->pairs : -1LL);
		return 1;
	}
	if (count == 0)
		return 0;
	SA_HIP_CHECK(hipSetDevice(ctx->device), return 1);
	hipStream_t s = (hipStream_t)stream;
	const bool share = world >= 1;
	if (!plan_build(ctx, start, count, world, share ? host_out != nullptr : ctx->out_is_host))
		return 1;

	auto timed_begin = [&](hipEvent_t &e0, hipEvent_t &e1) -> bool {
		if (!ctx->timing)
			return true;
		SA_HIP_CHECK(hipEventCreate(&e0), return false);
		SA_HIP_CHECK(hipEventCreate(&e1), return false);
		SA_HIP_CHECK(hipEventRecord(e0, s), return false);
		return true;
	};
	auto timed_end = [&](const std::string &name, hipEvent_t e0, hipEvent_t e1, int64_t pairs, int64_t cells) -> bool {
		if (!ctx->timing)
			return true;
		SA_HIP_CHECK(hipEventRecord(e1, s), return false);
		ctx->events.push_back(sa_ctx::Timed{ name, e0, e1, pairs, cells });
		return true;
	};

	/* Systolic streaming kernels.  Packed classes: one persistent launch per BUNDLE (normally one for the whole range).
	 * s32 classes: one persistent launch per class.  A single launch goes to the caller's stream; several go to side
	 * streams forked from / joined back into it so that they run concurrently. */
	const int rk = share ? rank : 0;
	struct Item {
		int bundle, cls; /* index into plan->pk_launches, or into plan->classes (s32 classes) */
	};
	std::vector<Item> items;
	for (size_t bi = 0; bi < ctx->plan->pk_launches.size(); bi++)
		if (ctx->plan->pk_launches[bi].nlocal[(size_t)rk] > 0)
			items.push_back({ (int)bi, -1 });
	for (size_t ci = 0; ci < ctx->plan->classes.size(); ci++) {
		const auto &cl = ctx->plan->classes[ci];
		if (cl.cls >= SA_PK_CLASS0)
			continue;
		if (share && cl.rank_first[(size_t)rank + 1] == cl.rank_first[(size_t)rank])
			continue;
		items.push_back({ -1, (int)ci });
	}
	const bool fan_out = items.size() > 1 && !ctx->env_serial_classes;
	const size_t slot = (size_t)(ctx->call_no++ % sa_ctx::COUNTER_SLOTS);
	unsigned *const counters = ctx->d_counters + slot * sa_ctx::COUNTERS_PER_SLOT;
	if (ctx->slot_done[slot]) { /* 256 calls ago, possibly on another stream: normally long complete */
		SA_HIP_CHECK(hipStreamWaitEvent(s, ctx->slot_done[slot], 0), return 1);
	} else {
		SA_HIP_CHECK(hipEventCreateWithFlags(&ctx->slot_done[slot], hipEventDisableTiming), return 1);
	}
	/* (the counters are zero: set so once at context creation, and every launch's last workgroup puts its own back) */
	if (fan_out) {
		if (!ctx->fork_ev) {
			SA_HIP_CHECK(hipEventCreateWithFlags(&ctx->fork_ev, hipEventDisableTiming), return 1);
		}
		for (int k = 0; k < sa_ctx::NSIDE && k < (int)items.size(); k++)
			if (!ctx->side[k]) {
				SA_HIP_CHECK(hipStreamCreateWithFlags(&ctx->side[k], hipStreamNonBlocking), return 1);
				SA_HIP_CHECK(hipEventCreateWithFlags(&ctx->join_ev[k], hipEventDisableTiming), return 1);
			}
		SA_HIP_CHECK(hipEventRecord(ctx->fork_ev, s), return 1);
	}
	hipStream_t caller = s;
	int launch_no = 0;
	ctx->prog_items.clear();
	for (const Item &it : items) {
		const int side_k = launch_no++ % sa_ctx::NSIDE;
		if (fan_out) {
			s = ctx->side[side_k];
			SA_HIP_CHECK(hipStreamWaitEvent(s, ctx->fork_ev, 0), return 1);
		}
		const bool is_pk = it.bundle >= 0;
		const sa_ctx::Plan::PkLaunch *pb = is_pk ? &ctx->plan->pk_launches[(size_t)it.bundle] : nullptr;
		const sa_ctx::ClassLaunch *clp = is_pk ? nullptr : &ctx->plan->classes[(size_t)it.cls];
		const int cls = is_pk ? -1 : clp->cls;
		const bool is_long = cls == SA_SYS_CLASS_LONG;
		SaSysArgs a{};
		a.codes = ctx->d_codes;
		a.off = ctx->d_off;
		a.sub8 = ctx->d_sub8;
		a.num = ctx->num;
		a.start = start;
		a.end = start + count;
		a.out = d_scores;
		a.out16 = out16 ? 1 : 0;
		a.gap_g = ctx->sc.gap_pen;
		a.gap_o = ctx->sc.gap_opn;
		a.gap_e = ctx->sc.gap_ext;
		a.host_out = host_out;
		int32_t ntiles_here;
		int64_t pairs_here, cells_here;
		char name[96];
		if (is_pk) {
			a.pconst = ctx->pk_pconst;
			a.q = ctx->sc.method == SA_METHOD_SW ? 0 : ctx->pk_q;
			a.out_nt = ctx->out_is_host && !share ? 1 : 0;
			a.pk_f16 = pb->f16;
			a.chunk = ctx->plan->chunk_pk; /* (the kernel takes chunk and arranged copies of every tile from its class block) */
			a.stagger = ctx->env_stagger;
			/* (measured: Gotoh's share of cfg 3 at 8 ranks 5.08 -> 5.01 ms, at 4 ranks 97.0 -> 98.0 % of ideal; NW needs all four
			 * waves to fill a SIMD, nobody starves, and the rotation costs it 1 %) */
			a.rotate_prio = ctx->env_rotate_prio >= 0 ? ctx->env_rotate_prio : ctx->sc.method != SA_METHOD_NW;
			a.pkc = pb->d_args;
			a.ulist = pb->d_ulist + pb->ufirst[(size_t)rk];
			a.npkc = (int32_t)pb->cls.size();
			a.nlocal = ntiles_here = pb->nlocal[(size_t)rk];
			pairs_here = pb->pairs[(size_t)rk];
			cells_here = pb->cells[(size_t)rk];
			a.counter = counters + 2 * (SA_PK_CLASS0 + it.bundle);
			int klo_seen = pb->kmax;
			for (int ci : pb->cls)
				klo_seen = std::min(klo_seen, pk_decode(ctx->plan->classes[(size_t)ci].cls).k);
			/* template arguments as rocprofv3 prints them, then the classes this launch walks */
			snprintf(name, sizeof(name), "sa_k_systolic_pk_bundle<%s,%d,%d,%s>[K%d-%d]", METHOD_TAG[ctx->sc.method], pb->g, pb->klo,
				 pb->f16 ? "true" : "false", klo_seen, pb->kmax);
		} else {
			const auto &cl = *clp;
			const int64_t W = is_long ? ((int64_t)ctx->max_len + SA_SYS_LONG_W - 1) / SA_SYS_LONG_W * SA_SYS_LONG_W
						  : SA_SYS_CLASSES[cls].G * SA_SYS_CLASSES[cls].K;
			a.jlist = cl.d_jlist;
			a.tprefix = cl.d_tprefix;
			a.npart = cl.npart;
			a.ncols = cl.ncols;
			a.pconst = ctx->sys_pconst;
			a.q = ctx->sys_q;
			a.delta = (int32_t)(ctx->sys_gain * W + ctx->sys_slack);
			a.counter = counters + 2 * cls;
			a.chunk = is_long ? std::min(ctx->plan->chunk, 16) : ctx->plan->chunk;
			ntiles_here = cl.ntiles;
			pairs_here = cl.pairs;
			cells_here = cl.cells;
			if (share) {
				a.tlist = cl.d_tlist + cl.rank_first[(size_t)rank];
				a.nlocal = ntiles_here = cl.rank_first[(size_t)rank + 1] - cl.rank_first[(size_t)rank];
				a.dense_off = cl.d_doff;
				pairs_here = cl.rank_pairs[(size_t)rank];
				cells_here = cl.rank_cells[(size_t)rank];
			}
			if (is_long) {
				/* scratch: two lines (V and X) of a wave's longest possible row stream, for every wave of as many
				 * workgroups as fit a 4 GiB budget */
				if (!ctx->d_long_scratch) {
					ctx->long_stride = 2 * (16 * ((int64_t)ctx->max_len + 1) + 64);
					const int64_t budget_ints = ((int64_t)4 << 30) / 4;
					const int64_t wpb = SA_SYS_WPB(64, true); /* one pair of lines per wave */
					ctx->long_wgs = (int)std::max<int64_t>(16, std::min<int64_t>(ctx->persistent_wgs / 4, budget_ints / (ctx->long_stride * wpb)));
					SA_HIP_CHECK(hipMalloc(&ctx->d_long_scratch, sizeof(int32_t) * (size_t)(ctx->long_stride * ctx->long_wgs * wpb)), return 1);
				}
				a.long_scratch = ctx->d_long_scratch;
				a.long_stride = ctx->long_stride;
				snprintf(name, sizeof(name), "sa_k_systolic<%s,G64,K16,strips>", METHOD_TAG[ctx->sc.method]);
			} else {
				snprintf(name, sizeof(name), "sa_k_systolic<%s,G%d,K%d>", METHOD_TAG[ctx->sc.method],
					 SA_SYS_CLASSES[cls].G, SA_SYS_CLASSES[cls].K);
			}
		}
		/* diagnostics: SA_HIP_STAMPS=1 makes every launch synchronous and prints the main-loop
		 * cycles per step and the shader clock the chip held (never enabled in timed runs) */
		unsigned long long *d_stamps = nullptr;
		const size_t nstamp = is_pk ? (size_t)ntiles_here : (size_t)clp->ntiles; /* (packed bundle: 7 words per tile) */
		const size_t stamp_words = is_pk ? 7 * nstamp : 3 * nstamp;
		if (ctx->env_stamps) {
			SA_HIP_CHECK(hipMalloc(&d_stamps, sizeof(unsigned long long) * stamp_words), return 1);
			SA_HIP_CHECK(hipMemset(d_stamps, 0, sizeof(unsigned long long) * stamp_words), return 1);
			a.stamps = d_stamps;
		}
		ctx->prog_items.push_back({ a.counter, ntiles_here });
		hipEvent_t e0 = nullptr, e1 = nullptr;
		if (!timed_begin(e0, e1))
			return 1;
		/* packed bundle: four workgroups per CU fill its LDS (4 x 39.9 KB at K = 13..16) and saturate the SIMDs (measured:
		 * a grid of 4 per CU = 8 per CU; 3 per CU: NW -8 %, Gotoh -1 %).  leave_room: three per CU, so that kernels of other
		 * streams -- an RCCL collective, the placement of the previous super-chunk -- find LDS and wave slots beside them
		 * (beside four they wait for the launch to end: a 0.05 ms placement took 0.73 ms and held the next kernel up) */
		const int pk_wgs = ctx->env_pk_wgs ? ctx->env_pk_wgs : ctx->persistent_wgs / 32 * (ctx->leave_room ? 3 : 4);
		const int wgs = (int)std::min<int64_t>(is_pk ? pk_wgs : is_long ? ctx->long_wgs : ctx->persistent_wgs, ntiles_here);
		if (is_pk) {
			SA_HIP_CHECK(sa_launch_systolic_pk(ctx->sc.method, pb->g, pb->klo, pb->f16, a, wgs,
							   (unsigned)sa_pk_lds_bytes(pb->g, pb->kmax), s), return 1);
		} else {
			SA_HIP_CHECK(sa_launch_systolic(ctx->sc.method, cls, a, wgs, s), return 1);
		}
		if (d_stamps) {
			std::vector<unsigned long long> h(stamp_words);
			SA_HIP_CHECK(hipStreamSynchronize(s), return 1);
			SA_HIP_CHECK(hipMemcpy(h.data(), d_stamps, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost), return 1);
			(void)hipFree(d_stamps);
			if (is_pk) {
				/* packed bundle: when every tile ran (100 MHz ticks) -> makespan, busy time, and how many workgroups were
				 * working in each twentieth of the launch: ramp, plateau and tail at a glance */
				unsigned long long t_lo = ~0ull, t_hi = 0;
				double busy = 0;
				for (size_t k = 0; k < nstamp; k++) {
					t_lo = std::min(t_lo, h[3 * k]);
					t_hi = std::max(t_hi, h[3 * k + 1]);
					busy += (double)(h[3 * k + 1] - h[3 * k]);
				}
				const double span = (double)(t_hi - t_lo);
				constexpr int BINS = 20;
				double act[BINS] = {};
				for (size_t k = 0; k < nstamp; k++) {
					const double a0 = (double)(h[3 * k] - t_lo) / span * BINS, a1 = (double)(h[3 * k + 1] - t_lo) / span * BINS;
					for (int b = (int)a0; b < BINS && b <= (int)a1; b++)
						act[b] += std::min(a1, (double)b + 1) - std::max(a0, (double)b);
				}
				if (const char *dump = getenv("SA_HIP_STAMPS_DUMP")) { /* raw words for offline analysis (7 per tile, see the kernel) */
					if (FILE *f = fopen(dump, "wb")) {
						fwrite(h.data(), sizeof(h[0]), h.size(), f);
						fclose(f);
					}
				}
				double clocks = 0, pro = 0, loop = 0, epi = 0, steps = 0;
				for (size_t k = 0; k < nstamp; k++) {
					const unsigned long long c0 = h[3 * k + 2], c1 = h[3 * nstamp + 3 * k], c2 = h[3 * nstamp + 3 * k + 1], c3 = h[6 * nstamp + k];
					clocks += (double)(c3 - c0);
					pro += (double)(c1 - c0);
					loop += (double)(c2 - c1);
					epi += (double)(c3 - c2);
					steps += (double)h[3 * nstamp + 3 * k + 2];
				}
				fprintf(stderr, "[stamps] per tile: prologue %.0f clocks, main loop %.0f (%.0f steps, %.1f clocks per step), epilogue %.0f\n",
					pro / (double)nstamp, loop / (double)nstamp, steps / (double)nstamp, loop / steps, epi / (double)nstamp);
				fprintf(stderr, "[stamps] %s: %zu tiles, makespan %.1f us, mean tile %.1f us = %.0f shader clocks (%.0f MHz), mean active workgroups %.0f; active per 5%% of the launch:",
					name, nstamp, span / 100.0, busy / (double)nstamp / 100.0, clocks / (double)nstamp, clocks / busy * 100.0, busy / span);
				for (int b = 0; b < BINS; b++)
					fprintf(stderr, " %.0f", act[b]);
				/* ... and the last tenth of the launch in hundredths, with the mean rows of the tiles that END there */
				double fine[10] = {}, rows_end[10] = {}, n_end[10] = {};
				for (size_t k = 0; k < nstamp; k++) {
					const double a0 = ((double)(h[3 * k] - t_lo) / span - 0.9) * 100.0, a1 = ((double)(h[3 * k + 1] - t_lo) / span - 0.9) * 100.0;
					for (int b = std::max(0, (int)a0); b < 10 && b <= (int)a1; b++)
						fine[b] += std::min(a1, (double)b + 1) - std::max(a0, (double)b);
					if (a1 >= 0) {
						const int b = std::min(9, (int)a1);
						rows_end[b] += (double)h[3 * nstamp + 3 * k + 2];
						n_end[b] += 1;
					}
				}
				fprintf(stderr, "; per 1%% of the last tenth (mean steps of the tiles ending there):");
				for (int b = 0; b < 10; b++)
					fprintf(stderr, " %.0f(%.0f)", fine[b], n_end[b] ? rows_end[b] / n_end[b] : 0.0);
				fprintf(stderr, "\n");
			} else {
				double cyc = 0, rt = 0, steps = 0;
				for (size_t k = 0; k < nstamp; k++) {
					cyc += (double)h[3 * k];
					rt += (double)h[3 * k + 1];
					steps += (double)h[3 * k + 2];
				}
				fprintf(stderr, "[stamps] %s: %zu wave-tiles, %.1f cycles/step per wave, clock %.0f MHz, %.0f steps/tile\n",
					name, nstamp, cyc / steps, cyc / rt * 100.0, steps / (double)nstamp);
			}
		}
		if (!timed_end(name, e0, e1, pairs_here, cells_here))
			return 1;
		if (fan_out) {
			SA_HIP_CHECK(hipEventRecord(ctx->join_ev[side_k], s), return 1);
			SA_HIP_CHECK(hipStreamWaitEvent(caller, ctx->join_ev[side_k], 0), return 1);
		}
	}
	s = caller;
	SA_HIP_CHECK(hipEventRecord(ctx->slot_done[slot], s), return 1);

	/* everything the fast path does not cover: pair-per-wave kernels on contiguous packed runs */
	/* (share: this rank's pieces of those runs, each at its offset of the dense share) */
	std::vector<std::pair<int64_t, int64_t>> runs = ctx->plan->generic;
	std::vector<int64_t> run_out;
	if (share) {
		runs.clear();
		for (const auto &gs : ctx->plan->generic_share[(size_t)rank]) {
			runs.emplace_back(gs.start, gs.count);
			run_out.push_back(gs.doff);
		}
	}
	for (size_t ri = 0; ri < runs.size(); ri++) {
		const auto &run = runs[ri];
		const int64_t out_at = share ? run_out[ri] : run.first - start;
		SaGenericArgs a{};
		a.st.codes = ctx->d_codes;
		a.st.meta = ctx->d_meta;
		a.st.num = ctx->num;
		a.sub = ctx->d_sub;
		a.gap_pen = ctx->sc.gap_pen;
		a.gap_opn = ctx->sc.gap_opn;
		a.gap_ext = ctx->sc.gap_ext;
		a.start = run.first;
		a.count = run.second;
		a.out = out16 ? reinterpret_cast<int32_t *>(reinterpret_cast<int16_t *>(d_scores) + out_at) : d_scores + out_at;
		a.out16 = out16 ? 1 : 0;
		a.host_out = host_out ? host_out + (run.first - start) : nullptr;
		a.scratch = ctx->d_scratch;
		a.scratch_stride = ctx->scratch_stride;
		const int blocks = (int)std::min<int64_t>(ctx->generic_blocks, (run.second + 3) / 4);
		hipEvent_t e0 = nullptr, e1 = nullptr;
		if (!timed_begin(e0, e1))
			return 1;
		SA_HIP_CHECK(sa_launch_generic(ctx->sc.method, a, blocks, s), return 1);
		int64_t run_cells = 0;
		if (ctx->timing) {
			PairPlan pp(ctx->meta.data(), ctx->num);
			run_cells = pp.cells_before(run.first + run.second) - pp.cells_before(run.first);
		}
		if (!timed_end(sa_generic_kernel_name(ctx->sc.method), e0, e1, run.second, run_cells))
			return 1;
	}
	return 0;
}

extern "C" int sa_ctx_expand_full(sa_ctx *ctx, const int32_t *d_packed, int32_t *d_full, void *stream)
{
	if (!ctx || !d_packed || !d_full) {
		sa_set_error("sa_ctx_expand_full: null argument");
		return 1;
	}
	SA_HIP_CHECK(hipSetDevice(ctx->device), return 1);
	SA_HIP_CHECK(sa_launch_expand_full(d_packed, d_full, ctx->num, (hipStream_t)stream), return 1);
	return 0;
}

/* ---- sa_hip_filter: device-assisted similarity filter (reference src/bio/filter.c:14-89) ---- */
extern "C" int32_t sa_hip_filter(struct sa_input in, float threshold, uint8_t *keep)
{
	if (!in.seqs || !in.meta || !keep || in.num < 1) {
		sa_set_error("sa_hip_filter: bad arguments");
		return -1;
	}
	const int32_t num = in.num;
	for (int32_t k = 0; k < num; k++)
		keep[k] = 1;
	if (threshold <= 0.0f)
		return num;
	if (!device_ready(0))
		return -1;
	/* tight copy of the raw residues (the filter compares bytes, filter.c:49) */
	std::vector<int32_t> off((size_t)num + 1, 0);
	int64_t end = 0;
	for (int32_t k = 0; k < num; k++) {
		if (in.meta[k].len < 1 || in.meta[k].off < 0) {
			sa_set_error("Sequence #%d has invalid offset/length", k + 1);
			return -1;
		}
		off[(size_t)k] = (int32_t)end;
		end += (int64_t)in.meta[k].len + 1;
		if (end > INT32_MAX) {
			sa_set_error("Sequence store exceeds 2 GiB");
			return -1;
		}
	}
	off[(size_t)num] = (int32_t)end;
	std::vector<uint8_t> blob((size_t)end, 0);
	for (int32_t k = 0; k < num; k++)
		memcpy(blob.data() + off[(size_t)k], in.seqs + in.meta[k].off, (size_t)in.meta[k].len);

	uint8_t *d_blob = nullptr;
	int32_t *d_off = nullptr;
	unsigned long long *d_rel = nullptr, *h_rel = nullptr;
	int32_t kept = -1;
	do {
		SA_HIP_CHECK(hipMalloc(&d_blob, blob.size()), break);
		SA_HIP_CHECK(hipMalloc(&d_off, sizeof(int32_t) * off.size()), break);
		SA_HIP_CHECK(hipMemcpy(d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice), break);
		SA_HIP_CHECK(hipMemcpy(d_off, off.data(), sizeof(int32_t) * off.size(), hipMemcpyHostToDevice), break);
		/* bands of 64-row tiles, at most ~256 MiB of relation words in flight */
		const int32_t tiles = (num + 63) / 64;
		const long long budget_words = (256LL << 20) / 8;
		const long long widest = (num + 63) / 64; /* words of the longest row */
		int32_t band_tiles = (int32_t)std::max<long long>(1, budget_words / (64 * widest));
		band_tiles = std::min(band_tiles, tiles);
		const size_t band_words = (size_t)(64LL * band_tiles * widest);
		SA_HIP_CHECK(hipMalloc(&d_rel, sizeof(unsigned long long) * band_words), break);
		SA_HIP_CHECK(hipHostMalloc(&h_rel, sizeof(unsigned long long) * band_words), break);
		std::vector<unsigned long long> keptbits((size_t)widest + 1, 0ULL);
		keptbits[0] = 1ULL; /* sequence 0 is always kept */
		bool failed = false;
		for (int32_t jt0 = 0; jt0 < tiles && !failed; jt0 += band_tiles) {
			const int32_t rows_t = std::min(band_tiles, tiles - jt0);
			const long long j_lo = 64LL * jt0, j_hi = std::min<long long>(num, 64LL * (jt0 + rows_t));
			const long long base = sa_filter_row_offset(j_lo), words = sa_filter_row_offset(j_hi) - base;
			SA_HIP_CHECK(sa_launch_filter_relation(d_blob, d_off, num, threshold, d_rel, jt0, rows_t, nullptr), failed = true; break);
			SA_HIP_CHECK(hipMemcpy(h_rel, d_rel, sizeof(unsigned long long) * (size_t)words, hipMemcpyDeviceToHost), failed = true; break);
			/* greedy keep/drop in sequence order (filter.c:38-55 run with one thread) */
			for (long long j = std::max<long long>(j_lo, 1); j < j_hi; j++) {
				const unsigned long long *row = h_rel + (sa_filter_row_offset(j) - base);
				const long long nw = (j + 63) / 64;
				bool lost = false;
				for (long long w = 0; w < nw && !lost; w++)
					lost = (row[w] & keptbits[(size_t)w]) != 0;
				if (lost)
					keep[j] = 0;
				else
					keptbits[(size_t)(j / 64)] |= 1ULL << (j % 64);
			}
		}
		if (failed)
			break;
		kept = 0;
		for (int32_t k = 0; k < num; k++)
			kept += keep[k];
	} while (0);
	(void)hipFree(d_blob);
	(void)hipFree(d_off);
	(void)hipFree(d_rel);
	if (h_rel)
		(void)hipHostFree(h_rel);
	return kept;
}

/* ---- host delivery: the launch/copy loop of cuda_align (src/interface/seqalign_cuda.c:182-292) ---- */

namespace {

constexpr int64_t BATCH_PAIRS = (int64_t)64 << 20; /* reference batch: src/interface/seqalign_cuda.c:136 */
constexpr int64_t FINAL_BATCH_PAIRS = (int64_t)3 << 20; /* the batch whose device->host copy nothing overlaps */

int64_t tri_of(int64_t j) { return j * (j - 1) / 2; }

/* scatter a packed slice into the full symmetric host matrix (what output_fill does per column,
 * reference src/io/output.c:76-81) */
void host_scatter_full(int32_t *matrix, size_t dim, const int32_t *slice, int64_t start, int64_t count)
{
	int64_t p = start;
	int64_t j = column_of(p);
	int64_t i = p - j * (j - 1) / 2;
	for (int64_t k = 0; k < count; k++) {
		matrix[dim * (size_t)i + (size_t)j] = slice[k];
		matrix[dim * (size_t)j + (size_t)i] = slice[k];
		if (++i == j) {
			i = 0;
			++j;
		}
	}
}


/* MemAvailable of the host: a destination larger than half of it (a file-backed matrix, src/io/output.c:36) is not
 * page-locked, its copies are staged by the runtime instead */
size_t host_available_bytes()
{
	size_t kb = 0;
	if (FILE *f = fopen("/proc/meminfo", "r")) {
		char line[256];
		while (fgets(line, sizeof(line), f))
			if (sscanf(line, "MemAvailable: %zu kB", &kb) == 1)
				break;
		fclose(f);
	}
	return kb * 1024;
}

bool deliver_resources(sa_ctx *ctx)
{
	auto &d = ctx->dl;
	if (d.compute)
		return true;
	SA_HIP_CHECK(hipStreamCreateWithFlags(&d.compute, hipStreamNonBlocking), return false);
	SA_HIP_CHECK(hipStreamCreateWithFlags(&d.copy, hipStreamNonBlocking), return false);
	for (int k = 0; k < 2; k++) {
		SA_HIP_CHECK(hipEventCreateWithFlags(&d.done[k], hipEventDisableTiming), return false);
		SA_HIP_CHECK(hipEventCreateWithFlags(&d.copied[k], hipEventDisableTiming), return false);
	}
	return true;
}

template <typename T> bool grow(T *&ptr, int64_t &have, int64_t want, bool host)
{
	if (have >= want)
		return true;
	if (ptr) {
		if (host)
			(void)hipHostFree(ptr);
		else
			(void)hipFree(ptr);
		ptr = nullptr;
		have = 0;
	}
	if (host) {
		SA_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&ptr), sizeof(T) * (size_t)want), return false);
	} else {
		SA_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&ptr), sizeof(T) * (size_t)want), return false);
	}
	have = want;
	return true;
}

void deliver_release(sa_ctx *ctx)
{
	auto &d = ctx->dl;
	for (int k = 0; k < 2; k++) {
		if (d.d_buf[k])
			(void)hipFree(d.d_buf[k]);
		if (d.h_stage[k])
			(void)hipHostFree(d.h_stage[k]);
		if (d.done[k])
			(void)hipEventDestroy(d.done[k]);
		if (d.copied[k])
			(void)hipEventDestroy(d.copied[k]);
	}
	if (d.d_packed)
		(void)hipFree(d.d_packed);
	if (d.d_full)
		(void)hipFree(d.d_full);
	if (d.compute)
		(void)hipStreamDestroy(d.compute);
	if (d.copy)
		(void)hipStreamDestroy(d.copy);
	d = sa_ctx::Deliver();
}

/* Full layout, column-aligned range [tri(j0), tri(j1)), everything resident: the packed scores of the range stay
 * on the device, every batch of columns [ja, jb) is expanded into its L-shaped shell of the full matrix --
 * rows [ja, jb) x cols [0, jb) and rows [0, ja) x cols [ja, jb), all of whose pairs belong to the batch -- and
 * the two rectangles go to the host as strided copies while the next batch computes.  The shells of all batches
 * tile the matrix exactly once, so the device->host traffic is N^2 elements and nearly all of it is overlapped. */
bool deliver_full_shells(sa_ctx *ctx, int64_t j0, int64_t j1, int32_t *matrix, double &phase_seconds)
{
	auto &d = ctx->dl;
	const int64_t start = tri_of(j0);
	const size_t dim = (size_t)ctx->num;
	/* batches cut at column starts.  The copies run ~8x faster than the kernels produce data, so only the LAST
	 * batch's copy is exposed: the batches shrink geometrically (half of what is left, at most one reference batch)
	 * down to a small final one, which keeps the launches few and long (short launches run at a lower rate) */
	std::vector<int64_t> cuts{ j0 };
	for (int64_t j = j0; j < j1;) {
		const int64_t left = tri_of(j1) - tri_of(j);
		const int64_t want = left <= FINAL_BATCH_PAIRS ? left : std::min<int64_t>(BATCH_PAIRS, left / 2);
		int64_t jn = column_of(std::min(tri_of(j) + want, tri_of(j1) - 1)) + 1;
		jn = std::min(std::max(jn, j + 1), j1);
		cuts.push_back(jn);
		j = jn;
	}
	const auto t_phase = std::chrono::steady_clock::now();
	for (size_t b = 0; b + 1 < cuts.size(); b++) {
		const int64_t ja = cuts[b], jb = cuts[b + 1];
		const int64_t lo = tri_of(ja), cnt = tri_of(jb) - lo;
		if (cnt > 0 && sa_ctx_align_range(ctx, lo, cnt, d.d_packed + (lo - start), d.compute))
			return false;
		SA_HIP_CHECK(sa_launch_expand_shell(d.d_packed, start, d.d_full, ctx->num, (int32_t)ja, (int32_t)jb, d.compute),
			     return false);
		SA_HIP_CHECK(hipEventRecord(d.done[0], d.compute), return false);
		SA_HIP_CHECK(hipStreamWaitEvent(d.copy, d.done[0], 0), return false);
		/* rows [ja, jb) x cols [0, jb) */
		SA_HIP_CHECK(hipMemcpy2DAsync(matrix + (size_t)ja * dim, dim * sizeof(int32_t), d.d_full + (size_t)ja * dim,
					      dim * sizeof(int32_t), (size_t)jb * sizeof(int32_t), (size_t)(jb - ja),
					      hipMemcpyDeviceToHost, d.copy), return false);
		if (ja > 0) { /* rows [0, ja) x cols [ja, jb) */
			SA_HIP_CHECK(hipMemcpy2DAsync(matrix + (size_t)ja, dim * sizeof(int32_t), d.d_full + (size_t)ja,
						      dim * sizeof(int32_t), (size_t)(jb - ja) * sizeof(int32_t), (size_t)ja,
						      hipMemcpyDeviceToHost, d.copy), return false);
		}
		if (g_progress_fn.load() && b + 2 < cuts.size()) { /* (batches issued so far; the host runs ahead of the device by one) */
			SA_HIP_CHECK(hipEventSynchronize(d.done[0]), return false);
			report_progress((double)(tri_of(jb) - start) / (double)std::max<int64_t>(1, tri_of(j1) - start));
		}
	}
	SA_HIP_CHECK(hipStreamSynchronize(d.compute), return false);
	SA_HIP_CHECK(hipStreamSynchronize(d.copy), return false);
	phase_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_phase).count();
	return true;
}

/* General path: double-buffered batches of packed scores; a job that would fit one reference-size batch is still
 * cut in ~10 pieces so that the device->host copy of a piece overlaps the kernels of the next.  Packed
 * destination: straight into the caller's matrix; full destination (range not column-aligned, or the device
 * cannot hold N^2): staged in pinned memory and scattered by the host like output_fill. */
bool deliver_batches(sa_ctx *ctx, int64_t lo, int64_t total, const sa_output &out, int64_t batch, bool shrink,
		     double &phase_seconds)
{
	auto &d = ctx->dl;
	const size_t dim = (size_t)ctx->num;
	const bool stage = out.matrix && !out.triangular;
	const auto t_phase = std::chrono::steady_clock::now();
	int64_t issued = 0;
	int nb = 0;
	struct Pending {
		int64_t start, count;
		int buf;
	} pend[2];
	int npend = 0;
	auto deliver_oldest = [&]() -> bool {
		const Pending pd = pend[0];
		SA_HIP_CHECK(hipEventSynchronize(d.copied[pd.buf]), return false);
		if (stage)
			host_scatter_full(out.matrix, dim, d.h_stage[pd.buf], pd.start, pd.count);
		pend[0] = pend[1];
		npend--;
		report_progress((double)(pd.start + pd.count - lo) / (double)std::max<int64_t>(1, total));
		return true;
	};
	while (issued < total) {
		const int b = nb & 1;
		if (npend == 2 && !deliver_oldest())
			return false;
		/* `shrink`: half of what is left (see deliver_full_shells), else fixed-size batches */
		const int64_t left = total - issued;
		const int64_t cnt = !shrink || left <= FINAL_BATCH_PAIRS ? std::min(batch, left) : std::min(batch, (left + 1) / 2);
		if (sa_ctx_align_range(ctx, lo + issued, cnt, d.d_buf[b], d.compute))
			return false;
		SA_HIP_CHECK(hipEventRecord(d.done[b], d.compute), return false);
		SA_HIP_CHECK(hipStreamWaitEvent(d.copy, d.done[b], 0), return false);
		if (out.matrix) {
			int32_t *dst = out.triangular ? out.matrix + lo + issued : d.h_stage[b];
			SA_HIP_CHECK(hipMemcpyAsync(dst, d.d_buf[b], sizeof(int32_t) * (size_t)cnt, hipMemcpyDeviceToHost, d.copy),
				     return false);
		}
		SA_HIP_CHECK(hipEventRecord(d.copied[b], d.copy), return false);
		/* the next kernel that reuses this buffer must wait for its copy-out */
		SA_HIP_CHECK(hipStreamWaitEvent(d.compute, d.copied[b], 0), return false);
		pend[npend++] = Pending{ lo + issued, cnt, b };
		issued += cnt;
		nb++;
	}
	while (npend)
		if (!deliver_oldest())
			return false;
	SA_HIP_CHECK(hipStreamSynchronize(d.compute), return false);
	SA_HIP_CHECK(hipStreamSynchronize(d.copy), return false);
	phase_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_phase).count();
	return true;
}

} // namespace

extern "C" int sa_hip_host_register(void *p, size_t bytes)
{
	if (!p || !bytes) {
		sa_set_error("sa_hip_host_register: null range");
		return 1;
	}
	/* (the calling thread's current device is left alone: in a one-process-per-GPU host it is the rank's own device, and a
	 * portable registration serves every device anyway) */
	if (sa_hip_device_count() <= 0) {
		sa_set_error("No HIP devices available; libseqalign_hip has no CPU fallback");
		return 1;
	}
	SA_HIP_CHECK(hipHostRegister(p, bytes, hipHostRegisterPortable), return 1);
	return 0;
}

extern "C" int sa_hip_host_unregister(void *p)
{
	if (!p)
		return 0;
	SA_HIP_CHECK(hipHostUnregister(p), return 1);
	return 0;
}

extern "C" int sa_ctx_align_host(sa_ctx *ctx, int64_t start, int64_t count, struct sa_output out, double *phase_seconds)
{
	if (phase_seconds)
		*phase_seconds = 0.0;
	if (!ctx || start < 0 || count < 0 || start + count > ctx->pairs) {
		sa_set_error("sa_ctx_align_host: bad range [%lld,+%lld) of %lld pairs", (long long)start, (long long)count,
			     ctx ? (long long)ctx->pairs : -1LL);
		return 1;
	}
	if (out.matrix && out.dim != (size_t)ctx->num) {
		sa_set_error("sa_ctx_align_host: output dim %zu does not match %d sequences", out.dim, ctx->num);
		return 1;
	}
	if (count == 0)
		return 0;
	SA_HIP_CHECK(hipSetDevice(ctx->device), return 1);
	if (!deliver_resources(ctx))
		return 1;
	auto &d = ctx->dl;
	const size_t dim = (size_t)ctx->num;
	const int64_t total = count;

	/* ---- set-up, outside the timed phase like the reference's allocations (seqalign_cuda.c:125-168) ---- */
	/* full layout: the shell schedule needs a column-aligned range and N^2 + the range's packed scores in HBM */
	bool shells = false;
	int64_t j0 = 0, j1 = 0;
	if (out.matrix && !out.triangular && !ctx->env_no_shells) {
		j0 = start == 0 ? 0 : column_of(start);
		j1 = column_of(start + count - 1) + 1;
		if (tri_of(j0) == start && tri_of(j1) == start + count) {
			size_t free_b = 0, total_b = 0;
			SA_HIP_CHECK(hipMemGetInfo(&free_b, &total_b), return 1);
			const long double have = (long double)free_b + sizeof(int32_t) * ((long double)d.packed_elems + (long double)d.full_elems);
			const long double need = sizeof(int32_t) * ((long double)dim * dim + (long double)total);
			shells = need * 4 / 3 < have;
		}
	}
	/* Page-lock the destination so that the copies are true DMA and overlap the kernels (a pageable destination is
	 * staged through a bounce buffer and serialises).  A caller that allocated the matrix with sa_hip_host_register /
	 * hipHostMalloc has done this already.  Best effort: if registration fails the copies still work, only slower. */
	void *pinned_here = nullptr;
	if (out.matrix && !ctx->env_no_pin && (out.triangular || shells)) {
		int32_t *base = out.triangular ? out.matrix + start : out.matrix;
		const size_t bytes = sizeof(int32_t) * (out.triangular ? (size_t)total : dim * dim);
		const size_t avail = host_available_bytes();
		if (!host_range_is_pinned(base) && (!avail || bytes <= avail / 2)) {
			const auto t_pin = std::chrono::steady_clock::now();
			if (hipHostRegister(base, bytes, hipHostRegisterDefault) == hipSuccess)
				pinned_here = base;
			else
				(void)hipGetLastError();
			ctx->setup.pin += ms_since(t_pin);
		}
	}
	/* Packed destination that is page-locked: the kernels store their scores straight into it.  The epilogue's
	 * coalesced 128-byte runs leave the chip as posted PCIe writes while the next tiles compute -- ~6 GB/s at
	 * cfg 2's rate against a link that moves ~50 -- so there is no copy pass, no batching and a single launch tail. */
	int32_t *direct = nullptr;
	if (out.matrix && out.triangular && !ctx->env_no_direct && host_range_is_pinned(out.matrix + start) &&
	    host_range_is_pinned(out.matrix + start + total - 1)) {
		void *dp = nullptr;
		if (hipHostGetDevicePointer(&dp, out.matrix + start, 0) == hipSuccess)
			direct = static_cast<int32_t *>(dp);
		else
			(void)hipGetLastError();
	}
	int64_t batch = 0;
	if (direct) {
		/* no staging buffers at all */
	} else if (shells) {
		if (!grow(d.d_packed, d.packed_elems, total, false) || !grow(d.d_full, d.full_elems, (int64_t)(dim * dim), false))
			return 1;
	} else {
		batch = std::min<int64_t>(std::max<int64_t>(total, 1), BATCH_PAIRS);
		if (out.matrix && total > FINAL_BATCH_PAIRS)
			batch = std::min<int64_t>(batch, (total + 1) / 2);
		for (int k = 0; k < 2; k++)
			if (!grow(d.d_buf[k], d.buf_elems[k], batch, false))
				return 1;
		if (out.matrix && !out.triangular)
			for (int k = 0; k < 2; k++)
				if (!grow(d.h_stage[k], d.stage_elems[k], batch, true))
					return 1;
	}
	if (direct && !prepare_range(ctx, start, total, true))
		return 1;
	SA_HIP_CHECK(hipDeviceSynchronize(), return 1);

	/* ---- the launch/copy loop: what the reference brackets with bench_align_start/end ---- */
	double phase = 0.0;
	bool ok;
	if (direct) {
		const auto t_phase = std::chrono::steady_clock::now();
		ctx->out_is_host = true;
		ok = sa_ctx_align_range(ctx, start, total, direct, d.compute) == 0;
		ctx->out_is_host = false;
		if (ok && t_progress_here && g_progress_fn.load() && !ctx->prog_items.empty()) {
			/* Progress (the reference's pproportc in its batch loop, src/interface/seqalign_cuda.c:286-289): one launch does
			 * the whole range here, so the host reads the launches' tile counters every 50 ms while it waits.  A counter
			 * goes back to zero when its launch ends: fractions only ever grow. */
			unsigned *h_cnt = nullptr;
			const size_t nit = ctx->prog_items.size();
			int64_t all = 0;
			for (const auto &it : ctx->prog_items)
				all += it.tiles;
			double shown = 0.0;
			if (hipHostMalloc(reinterpret_cast<void **>(&h_cnt), sizeof(unsigned) * nit) == hipSuccess) {
				std::vector<int64_t> seen(nit, 0);
				while (hipStreamQuery(d.compute) == hipErrorNotReady) {
					std::this_thread::sleep_for(std::chrono::milliseconds(50));
					bool got = true;
					for (size_t k = 0; k < nit && got; k++)
						got = hipMemcpyAsync(&h_cnt[k], ctx->prog_items[k].counter, sizeof(unsigned), hipMemcpyDeviceToHost, d.copy) == hipSuccess;
					if (!got || hipStreamSynchronize(d.copy) != hipSuccess)
						break;
					int64_t done = 0;
					for (size_t k = 0; k < nit; k++) {
						seen[k] = std::max(seen[k], std::min<int64_t>(h_cnt[k], ctx->prog_items[k].tiles));
						done += seen[k];
					}
					const double f = all ? (double)done / (double)all : 0.0;
					if (f > shown)
						report_progress(shown = f);
				}
				(void)hipGetLastError();
				(void)hipHostFree(h_cnt);
			}
		}
		if (ok) {
			SA_HIP_CHECK(hipStreamSynchronize(d.compute), ok = false);
		}
		phase = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_phase).count();
	} else {
		ok = shells ? deliver_full_shells(ctx, j0, j1, out.matrix, phase)
			    : deliver_batches(ctx, start, total, out, batch, out.matrix != nullptr, phase);
	}
	if (!ok) { /* leave nothing in flight that still targets the caller's memory */
		(void)hipStreamSynchronize(d.compute);
		(void)hipStreamSynchronize(d.copy);
	}
	if (pinned_here)
		(void)hipHostUnregister(pinned_here);
	if (phase_seconds)
		*phase_seconds = phase;
	return ok ? 0 : 1;
}

/* ---- sa_hip_align: the cuda_align replacement (host buffers in, host matrix out) ---------- */

/* launch/copy phase of the last successful sa_hip_align call (the reference's bench_align_start..end bracket) */
static std::atomic<double> g_last_align_seconds{ 0.0 };
static std::mutex g_breakdown_mutex;
static double g_breakdown[SA_BREAKDOWN_COUNT] = {};

/* devices sa_hip_align spreads a job over: all visible ones, or the first SA_HIP_DEVICES */
static int devices_in_use(void)
{
	int ndev = sa_hip_device_count();
	if (const char *env = getenv("SA_HIP_DEVICES")) {
		const int want = atoi(env);
		if (want >= 1 && want < ndev)
			ndev = want;
	}
	return ndev;
}

/* reference src/interface/seqalign_cuda.c:71-93; with several devices the answer must hold on each of them */
extern "C" bool sa_hip_memory(size_t bytes)
{
	const int ndev = devices_in_use();
	if (ndev <= 0) {
		sa_set_error("No HIP devices available; libseqalign_hip has no CPU fallback");
		return false;
	}
	const long double need = (long double)bytes * 4.0L / 3.0L;
	for (int dev = 0; dev < ndev; dev++) {
		if (!device_ready(dev))
			return false;
		size_t free_b = 0, total_b = 0;
		SA_HIP_CHECK(hipMemGetInfo(&free_b, &total_b), return false);
		if ((long double)free_b < need) {
			sa_set_error("%.2f GiB exceeds available GPU memory (%.2f GiB free on device %d)",
				     (double)(need / (1 << 30)), (double)free_b / (double)(1 << 30), dev);
			(void)hipSetDevice(0);
			return false;
		}
	}
	(void)hipSetDevice(0);
	return true;
}

extern "C" bool sa_hip_align(struct sa_input in, struct sa_output out, const struct sa_scoring *sc)
{
	if (!sc) {
		sa_set_error("sa_hip_align: null scoring");
		return false;
	}
	if (out.matrix && out.dim != (size_t)in.num) {
		sa_set_error("sa_hip_align: output dim %zu does not match %d sequences", out.dim, in.num);
		return false;
	}
	int ndev = devices_in_use();
	if (ndev <= 0) {
		sa_set_error("No HIP devices available; libseqalign_hip has no CPU fallback");
		return false;
	}
	const int64_t pairs = (int64_t)in.num * (in.num - 1) / 2;
	/* SA_HIP_SPLIT=n (testing aid): run the multi-device code path with n work-balanced slices even when
	 * fewer devices are visible -- slice k goes to device k mod visible */
	const int nvisible = ndev;
	int split = 0;
	if (const char *env = getenv("SA_HIP_SPLIT"))
		split = atoi(env);
	if (split >= 2 && pairs >= split)
		ndev = split;
	else if (pairs < (int64_t)ndev * 4096)
		ndev = 1;
	/* One device per slice of the packed index, cut by DP work (sa_pairs_partition); every device delivers its slice
	 * straight into the host matrix -- the host is the destination, so no device-to-device exchange is needed
	 * (DESIGN.md 6).  Full layout: the cuts are moved to column starts so that every slice is a set of whole
	 * columns and can use the shell schedule. */
	std::vector<int64_t> bounds((size_t)ndev + 1, 0);
	bounds[(size_t)ndev] = pairs;
	if (ndev > 1) {
		if (sa_pairs_partition(in.meta, in.num, ndev, bounds.data()))
			return false;
		if (out.matrix && !out.triangular)
			for (int k = 1; k < ndev; k++)
				bounds[(size_t)k] = std::max(bounds[(size_t)k - 1], tri_of(column_of(bounds[(size_t)k])));
	}
	/* several slices share one destination: page-lock it once for all of them (set-up, outside the timed phase) */
	void *pinned_here = nullptr;
	double pin_all_ms = 0.0;
	if (ndev > 1 && out.matrix && !getenv("SA_HIP_NO_PIN") && device_ready(0) && !host_range_is_pinned(out.matrix)) {
		const size_t n = (size_t)in.num;
		const size_t bytes = sizeof(int32_t) * (out.triangular ? (size_t)pairs : n * n);
		const auto t_pin = std::chrono::steady_clock::now();
		if (hipHostRegister(out.matrix, bytes, hipHostRegisterPortable) == hipSuccess)
			pinned_here = out.matrix;
		else
			(void)hipGetLastError();
		pin_all_ms = ms_since(t_pin);
	}
	std::vector<std::string> errs((size_t)ndev);
	std::vector<char> oks((size_t)ndev, 0);
	std::vector<double> phases((size_t)ndev, 0.0);
	auto run = [&](int k) {
		const int64_t lo = bounds[(size_t)k], hi = bounds[(size_t)k + 1];
		if (hi <= lo && k > 0) { /* (slice 0 always builds its context: that is where the input is validated) */
			oks[(size_t)k] = 1;
			return;
		}
		const auto t_slice = std::chrono::steady_clock::now();
		t_progress_here = k == 0;
		sa_ctx *ctx = sa_ctx_create(k % nvisible, in, sc);
		if (ctx && sa_ctx_align_host(ctx, lo, hi - lo, out, &phases[(size_t)k]) == 0)
			oks[(size_t)k] = 1;
		else
			errs[(size_t)k] = sa_last_error();
		if (ctx && k == 0) { /* the first slice speaks for the call */
			std::lock_guard<std::mutex> g(g_breakdown_mutex);
			const double v[SA_BREAKDOWN_COUNT] = { ctx->setup.encode, ctx->setup.device, ctx->setup.upload, ctx->setup.code_objects,
							       ctx->setup.pin + pin_all_ms, ctx->setup.plan, ctx->setup.arrange, phases[0] * 1e3,
							       ms_since(t_slice) };
			memcpy(g_breakdown, v, sizeof(v));
		}
		sa_ctx_destroy(ctx);
		t_progress_here = true;
	};
	if (ndev == 1) {
		run(0);
	} else {
		std::vector<std::thread> threads;
		for (int k = 0; k < ndev; k++)
			threads.emplace_back(run, k);
		for (auto &t : threads)
			t.join();
	}
	if (pinned_here) {
		(void)hipSetDevice(0);
		(void)hipHostUnregister(pinned_here);
	}
	for (int k = 0; k < ndev; k++)
		if (!oks[(size_t)k]) {
			if (ndev > 1)
				sa_set_error("device %d: %s", k % nvisible, errs[(size_t)k].c_str());
			else
				sa_set_error("%s", errs[(size_t)k].c_str());
			return false;
		}
	g_last_align_seconds.store(*std::max_element(phases.begin(), phases.end()));
	return true;
}

extern "C" int sa_hip_last_align_breakdown(double *ms, int n)
{
	if (!ms || n < 0)
		return 0;
	std::lock_guard<std::mutex> g(g_breakdown_mutex);
	const int m = std::min(n, (int)SA_BREAKDOWN_COUNT);
	for (int k = 0; k < m; k++)
		ms[k] = g_breakdown[k];
	return m;
}

extern "C" double sa_hip_last_align_seconds(void)
{
	return g_last_align_seconds.load();
}
