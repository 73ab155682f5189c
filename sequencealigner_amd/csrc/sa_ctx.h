/* sa_ctx.h -- the device context behind the C ABI and what its translation units share.
 *
 *   sa_context.hip : sa_ctx_create / destroy, input validation + encoding, kernel-family limits, arranged copies
 *   sa_plan.cpp    : launch planning, host only (sa_plan.h)
 *   sa_launch.hip  : plan cache + upload, sa_ctx_align_range, shares and placement
 *   sa_deliver.hip : sa_ctx_align_host (the launch/copy loop towards host memory)
 *   sa_gather.hip  : several devices of one process through dense shares + RCCL all-gather
 *   sa_abi.hip     : sa_hip_align, sa_hip_memory, sa_hip_filter, progress, side channels
 */
#ifndef SA_CTX_H
#define SA_CTX_H

#include <atomic>
#include <chrono>
#include <deque>
#include <exception>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "sa_env.h"
#include "sa_guard.h"
#include "sa_internal.h"
#include "sa_plan.h"

struct sa_ctx {
	int device = 0;
	int32_t num = 0, max_len = 0, min_len = 0;
	int64_t pairs = 0;
	sa_scoring sc{};
	SaEnv env;                     /* the switches, as they were when the context was created             */
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
	/* Tile counters of the persistent launches: one slot of COUNTERS_PER_SLOT counters per sa_ctx_align_range call, taken
	 * round-robin from a ring so that ranges issued back to back on DIFFERENT streams never share a counter.
	 * INVARIANT: a slot's counters are zero whenever no launch is using them -- zeroed at context creation, and every
	 * launch's last workgroup puts its own pair back.  A launch that does not run to completion (a failed call, a fault)
	 * breaks that: `slot_dirty` marks the slot of any call that returned an error after it may have launched, and its
	 * next user zeroes the counters on its stream first. */
	enum { COUNTER_SLOTS = 256, COUNTERS_PER_SLOT = 2 * SA_PLAN_NCLASSES }; /* (next tile, workgroups done) per class */
	unsigned *d_counters = nullptr;
	uint64_t call_no = 0;
	hipEvent_t slot_done[COUNTER_SLOTS] = {}; /* recorded after the launches that used a slot: its next user waits */
	bool slot_dirty[COUNTER_SLOTS] = {};
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
	/* arranged copies of the store for the packed kernels' row streams (arranged_store), one per tile shape;
	 * a deque: references handed out stay valid when another copy is added */
	struct Arranged {
		SaArrKey key;
		uint8_t *d_codes = nullptr;
		int32_t *d_off = nullptr, *d_rowmap = nullptr, *d_posmap = nullptr;
	};
	std::deque<Arranged> arranged;
	bool out_is_host = false; /* the range being launched stores straight into host memory (sa_ctx_align_host) */
	/* launch plans of recently used packed ranges (callers loop over the same few ranges): the host plan (sa_plan.h) and
	 * its device copies */
	struct Plan {
		SaHostPlan h;
		struct DevClass {
			int32_t *d_jlist = nullptr, *d_tprefix = nullptr, *d_tlist = nullptr;
			int64_t *d_doff = nullptr;
		};
		struct DevBundle {
			SaPkClassArgs *d_args = nullptr;
			uint32_t *d_ulist = nullptr;
		};
		std::vector<DevClass> dc; /* [h.classes.size()] */
		std::vector<DevBundle> db; /* [h.bundles.size()] */
		SaPlaceSeg *d_segs = nullptr;
		int32_t nsegs = 0;
		uint64_t stamp = 0;
	};
	std::vector<std::unique_ptr<Plan>> plans; /* small LRU cache */
	Plan *plan = nullptr;                     /* plan of the current sa_ctx_align_range call */
	uint64_t plan_clock = 0;
	/* instrumentation: one HIP-event pair per kernel launch, keyed by kernel name */
	bool timing = false;
	struct Timed {
		std::string name;
		hipEvent_t e0, e1;
		int64_t pairs, cells;
	};
	std::vector<Timed> events;
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
		unsigned *h_counters = nullptr; /* page-locked mirror of the progress counters           */
		int64_t h_counters_n = 0;
	} dl;
};

/* ---- shared between the translation units ------------------------------------------------------------------------ */
static inline double sa_ms_since(std::chrono::steady_clock::time_point t0)
{
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

/* sa_context.hip */
bool sa_device_ready(int device);
/* the arranged copy for `key` (built and uploaded on first use); *out = nullptr when the store has no full block */
bool sa_arranged_store(sa_ctx *ctx, const SaArrKey &key, const sa_ctx::Arranged **out);
SaPlanInputs sa_plan_inputs(const sa_ctx *ctx);

/* sa_launch.hip */
void sa_plan_release(sa_ctx *ctx);
/* plan (cached) of a range -> ctx->plan; world / share_host as in sa_plan_host */
bool sa_plan_get(sa_ctx *ctx, int64_t start, int64_t count, int world, bool share_host);
/* plan and arranged copies of a range, ahead of a timed launch loop */
bool sa_prepare_range(sa_ctx *ctx, int64_t start, int64_t count, bool host_out);
int sa_align_range_impl(sa_ctx *ctx, int64_t start, int64_t count, int32_t *d_scores, void *stream, bool out16,
			int world = 0, int rank = 0, int32_t *host_out = nullptr);

/* sa_deliver.hip */
void sa_deliver_release(sa_ctx *ctx);
/* Is [p, p + bytes) page-locked host memory known to the HIP runtime, as ONE allocation / registration?  (A destination
 * registered in pieces with a hole between them must not take the direct-store path: a GPU page fault, not a fallback.) */
bool sa_host_range_is_pinned(const void *p, size_t bytes);
size_t sa_host_available_bytes();
/* does the range touch the malloc heap ([heap])?  Such memory is never page-locked here */
bool sa_host_range_in_malloc_heap(const void *p, size_t bytes);

/* sa_abi.hip: progress side channel (sa_hip_set_progress) */
bool sa_progress_wanted();          /* a callback is set and this thread speaks for the job */
void sa_report_progress(double fraction);
void sa_progress_speaker(bool here); /* sa_hip_align on several devices: only the first slice's thread reports */
int sa_devices_in_use(const SaEnv &env);

/* sa_gather.hip: the multi-device body of sa_hip_align that follows north_star literally -- dense shares per device, RCCL
 * all-gather, placement, delivery.  false + sa_last_error on failure; *phase_seconds like sa_ctx_align_host. */
bool sa_align_gathered(struct sa_input in, struct sa_output out, const struct sa_scoring *sc, const int *devices, int ndev,
		       double *phase_seconds, double *breakdown_ms);
bool sa_rccl_available(std::string *why);

#endif /* SA_CTX_H */
