/*
 * sa_launch.hip -- from a packed pair range to kernel launches: the plan cache (host plan of sa_plan.cpp + its device
 * copies), sa_ctx_align_range and the int16 exchange format, tile-interleaved shares and their placement.
 *
 * Replaces the reference's `kernel(scores, start, batch)` abstraction and its launch loop
 * (src/bio/align.h:48, src/bio/kernels.cu:32-40, src/interface/seqalign_cuda.c:170-264).
 */
#include <algorithm>
#include <cstdio>
#include <cstring>

#include "sa_ctx.h"

/* ---- plan cache ---------------------------------------------------------------------------------------------------- */

static void plan_free_device(sa_ctx::Plan &pl)
{
	for (auto &c : pl.dc) {
		(void)hipFree(c.d_jlist);
		(void)hipFree(c.d_tprefix);
		(void)hipFree(c.d_tlist);
		(void)hipFree(c.d_doff);
	}
	(void)hipFree(pl.d_segs);
	for (auto &b : pl.db) {
		(void)hipFree(b.d_args);
		(void)hipFree(b.d_ulist);
	}
	pl.dc.clear();
	pl.db.clear();
	pl.d_segs = nullptr;
}

void sa_plan_release(sa_ctx *ctx)
{
	for (auto &pl : ctx->plans)
		plan_free_device(*pl);
	ctx->plans.clear();
	ctx->plan = nullptr;
}

template <typename T> static bool upload(T *&dst, const std::vector<T> &v)
{
	SA_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&dst), sizeof(T) * std::max<size_t>(v.size(), 1)), return false);
	if (!v.empty()) {
		SA_HIP_CHECK(hipMemcpy(dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice), return false);
	}
	return true;
}

/* the arranged copies a class's tiles choose from, as the kernels see them (builds and uploads what is missing) */
static bool resolve_levels(sa_ctx *ctx, const SaArrKey (&keys)[SA_PK_SORT_LEVELS], SaArranged (&lv)[SA_PK_SORT_LEVELS])
{
	for (int l = 0; l < SA_PK_SORT_LEVELS; l++) {
		lv[l] = SaArranged{};
		if (!keys[l].block)
			continue;
		const sa_ctx::Arranged *ar = nullptr;
		if (!sa_arranged_store(ctx, keys[l], &ar))
			return false;
		if (ar)
			lv[l] = { ar->d_codes, ar->d_off, ar->d_rowmap, ar->d_posmap, ar->key.block };
	}
	return true;
}

static bool plan_upload(sa_ctx *ctx, sa_ctx::Plan &pl)
{
	const SaHostPlan &h = pl.h;
	pl.dc.resize(h.classes.size());
	for (size_t ci = 0; ci < h.classes.size(); ci++) {
		const auto &cl = h.classes[ci];
		auto &d = pl.dc[ci];
		if (!upload(d.d_jlist, cl.jlist) || !upload(d.d_tprefix, cl.tprefix))
			return false;
		if (h.world >= 1 && (!upload(d.d_tlist, cl.tlist) || !upload(d.d_doff, cl.doff)))
			return false;
	}
	if (h.world >= 1 && !h.segs.empty()) {
		std::vector<SaPlaceSeg> segs(h.segs.size());
		for (size_t k = 0; k < h.segs.size(); k++) {
			const SaHostSeg &hs = h.segs[k];
			SaPlaceSeg &sg = segs[k];
			sg = SaPlaceSeg{};
			sg.src = hs.src;
			sg.dst = hs.dst;
			sg.count = hs.count;
			sg.pos0 = hs.pos0;
			sg.ia = hs.ia;
			sg.ib = hs.ib;
			sg.flags = hs.flags;
			sg.map = nullptr;
			if (hs.map_kind) {
				const sa_ctx::Arranged *ar = nullptr;
				if (!sa_arranged_store(ctx, hs.key, &ar))
					return false;
				if (!ar) {
					sa_set_error("plan: placement names an arranged copy the store does not have");
					return false;
				}
				sg.map = hs.map_kind == 2 ? ar->d_posmap : ar->d_rowmap;
			}
		}
		if (!upload(pl.d_segs, segs))
			return false;
		pl.nsegs = (int32_t)segs.size();
	}
	pl.db.resize(h.bundles.size());
	for (size_t bi = 0; bi < h.bundles.size(); bi++) {
		const auto &b = h.bundles[bi];
		std::vector<SaPkClassArgs> args(b.args.size());
		for (size_t x = 0; x < b.args.size(); x++) {
			const SaHostPkArgs &ha = b.args[x];
			SaPkClassArgs &a = args[x];
			a = SaPkClassArgs{};
			if (!resolve_levels(ctx, ha.lv, a.lv))
				return false;
			const auto &d = pl.dc[(size_t)ha.cls_index];
			a.jlist = d.d_jlist;
			a.tprefix = d.d_tprefix;
			a.dense_off = h.world >= 1 ? d.d_doff : nullptr;
			a.ncols = ha.ncols;
			a.npart = ha.npart;
			a.k = ha.k;
			a.delta = ha.delta;
			a.pk_base = ha.pk_base;
			a.chunk = ha.chunk;
		}
		if (!upload(pl.db[bi].d_args, args) || !upload(pl.db[bi].d_ulist, b.ulist))
			return false;
	}
	return true;
}

bool sa_plan_get(sa_ctx *ctx, int64_t start, int64_t count, int world, bool share_host)
{
	for (auto &pl : ctx->plans)
		if (pl->h.start == start && pl->h.count == count && pl->h.world == world && pl->h.share_host == share_host) {
			pl->stamp = ++ctx->plan_clock;
			ctx->plan = pl.get();
			return true;
		}
	const auto t_plan = std::chrono::steady_clock::now();
	struct Acc { /* the arranged copies built on the way are booked under their own heading */
		sa_ctx *c;
		std::chrono::steady_clock::time_point t;
		double a0;
		~Acc() { c->setup.plan += sa_ms_since(t) - (c->setup.arrange - a0); }
	} acc{ ctx, t_plan, ctx->setup.arrange };
	constexpr size_t MAX_PLANS = 160; /* (a walk in shells -- sa_deflate.hip -- holds one plan per column block: 74 for 300 000 sequences) */
	ctx->plan = nullptr;
	if (ctx->plans.size() >= MAX_PLANS) { /* evict the least recently used (hipFree waits for its users) */
		size_t victim = 0;
		for (size_t k = 1; k < ctx->plans.size(); k++)
			if (ctx->plans[k]->stamp < ctx->plans[victim]->stamp)
				victim = k;
		plan_free_device(*ctx->plans[victim]);
		ctx->plans.erase(ctx->plans.begin() + (long)victim);
	}
	std::unique_ptr<sa_ctx::Plan> pl(new sa_ctx::Plan());
	if (!sa_plan_host(sa_plan_inputs(ctx), start, count, world, share_host, pl->h))
		return false;
	struct Undo { /* a half-uploaded plan (failed call, exception) gives its device memory back */
		sa_ctx::Plan *p;
		~Undo()
		{
			if (p)
				plan_free_device(*p);
		}
	} undo{ pl.get() };
	if (!plan_upload(ctx, *pl))
		return false;
	pl->stamp = ++ctx->plan_clock;
	ctx->plans.push_back(std::move(pl));
	undo.p = nullptr;
	ctx->plan = ctx->plans.back().get();
	return true;
}

bool sa_prepare_range(sa_ctx *ctx, int64_t start, int64_t count, bool host_out)
{
	if (count <= 0)
		return true;
	return sa_plan_get(ctx, start, count, 0, host_out); /* (uploading a plan builds the arranged copies it names) */
}

/* ---- launches ------------------------------------------------------------------------------------------------------ */

static const char *const METHOD_TAG[] = { "nw", "ga", "sw" };

/* diagnostics (SA_HIP_STAMPS=1): what the per-tile clocks of one launch say */
static void print_stamps(const sa_ctx *ctx, const char *name, bool is_pk, const std::vector<unsigned long long> &h, size_t nstamp)
{
	if (!nstamp)
		return;
	if (!is_pk) {
		double cyc = 0, rt = 0, steps = 0;
		for (size_t k = 0; k < nstamp; k++) {
			cyc += (double)h[3 * k];
			rt += (double)h[3 * k + 1];
			steps += (double)h[3 * k + 2];
		}
		fprintf(stderr, "[stamps] %s: %zu wave-tiles, %.1f cycles/step per wave, clock %.0f MHz, %.0f steps/tile\n",
			name, nstamp, cyc / steps, cyc / rt * 100.0, steps / (double)nstamp);
		return;
	}
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
	if (ctx->env.stamps_dump) { /* raw words for offline analysis (7 per tile, see the kernel) */
		if (FILE *f = fopen(ctx->env.stamps_dump, "wb")) {
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
}

/* share: world >= 1 runs the tiles of `rank` only and stores them densely (sa_ctx_align_share); 0: the whole range */
static int align_range_launches(sa_ctx *ctx, int64_t start, int64_t count, int32_t *d_scores, hipStream_t s, bool out16,
				int world, int rank, int32_t *host_out, size_t slot)
{
	const bool share = world >= 1;
	const sa_ctx::Plan &pl = *ctx->plan;
	const SaHostPlan &hp = pl.h;

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
	 * s32 classes: one persistent launch per class.  Several launches of a range run ONE AFTER THE OTHER on the caller's
	 * stream, the bundle of the largest K first: every one of them fills the chip by itself, and side by side (round 3: side
	 * streams forked from / joined into the caller's) their workgroups -- different LDS sizes, different code -- crowd each
	 * other out of the CUs: cfg 4's two bundles 49.6 -> 46.1 ms, cfg 5's 30.8 -> 29.7 ms, three bundles of a mixed-length
	 * store 12.35 -> 11.54 ms (profiles/r04_bundles_side_by_side_vs_serial.txt).  SA_HIP_CONCURRENT_CLASSES=1: side by side. */
	const int rk = share ? rank : 0;
	struct Item {
		int bundle, cls; /* index into the plan's bundles, or into its classes (s32 classes) */
	};
	std::vector<Item> items;
	for (size_t bi = 0; bi < hp.bundles.size(); bi++)
		if (hp.bundles[bi].nlocal[(size_t)rk] > 0)
			items.push_back({ (int)bi, -1 });
	for (size_t ci = 0; ci < hp.classes.size(); ci++) {
		const auto &cl = hp.classes[ci];
		if (cl.cls >= SA_PK_CLASS0)
			continue;
		if (share && cl.rank_first[(size_t)rank + 1] == cl.rank_first[(size_t)rank])
			continue;
		items.push_back({ -1, (int)ci });
	}
	const bool fan_out = items.size() > 1 && ctx->env.concurrent_classes;
	unsigned *const counters = ctx->d_counters + slot * sa_ctx::COUNTERS_PER_SLOT;
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
		const SaHostBundle *pb = is_pk ? &hp.bundles[(size_t)it.bundle] : nullptr;
		const SaHostClass *clp = is_pk ? nullptr : &hp.classes[(size_t)it.cls];
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
			const auto &db = pl.db[(size_t)it.bundle];
			a.pconst = ctx->pk_pconst;
			a.q = ctx->sc.method == SA_METHOD_SW ? 0 : ctx->pk_q;
			a.out_nt = ctx->out_is_host && !share ? 1 : 0;
			a.pk_f16 = pb->f16;
			a.chunk = hp.chunk_pk; /* (the kernel takes chunk and arranged copies of every tile from its class block) */
			a.stagger = ctx->env.stagger;
			/* (measured: Gotoh's share of cfg 3 at 8 ranks 5.08 -> 5.01 ms, at 4 ranks 97.0 -> 98.0 % of ideal; NW needs all four
			 * waves to fill a SIMD, nobody starves, and the rotation costs it 1 %) */
			a.rotate_prio = ctx->env.rotate_prio >= 0 ? ctx->env.rotate_prio : ctx->sc.method != SA_METHOD_NW;
			a.pkc = db.d_args;
			a.ulist = db.d_ulist + pb->ufirst[(size_t)rk];
			a.npkc = (int32_t)pb->cls.size();
			a.nlocal = ntiles_here = pb->nlocal[(size_t)rk];
			pairs_here = pb->pairs[(size_t)rk];
			cells_here = pb->cells[(size_t)rk];
			a.counter = counters + 2 * (SA_PK_CLASS0 + it.bundle);
			int klo_seen = pb->kmax;
			for (int ci : pb->cls)
				klo_seen = std::min(klo_seen, sa_pk_decode(hp.classes[(size_t)ci].cls).k);
			/* template arguments as rocprofv3 prints them, then the classes this launch walks */
			snprintf(name, sizeof(name), "sa_k_systolic_pk_bundle<%s,%d,%d,%s>[K%d-%d]", METHOD_TAG[ctx->sc.method], pb->g, pb->klo,
				 pb->f16 ? "true" : "false", klo_seen, pb->kmax);
		} else {
			const auto &cl = *clp;
			const auto &dc = pl.dc[(size_t)it.cls];
			const int64_t W = is_long ? ((int64_t)ctx->max_len + SA_SYS_LONG_W - 1) / SA_SYS_LONG_W * SA_SYS_LONG_W
						  : SA_SYS_CLASSES[cls].G * SA_SYS_CLASSES[cls].K;
			a.jlist = dc.d_jlist;
			a.tprefix = dc.d_tprefix;
			a.npart = cl.npart;
			a.ncols = cl.ncols;
			a.pconst = ctx->sys_pconst;
			a.q = ctx->sys_q;
			a.delta = (int32_t)(ctx->sys_gain * W + ctx->sys_slack);
			a.counter = counters + 2 * cls;
			a.chunk = is_long ? std::min(hp.chunk, 16) : hp.chunk;
			ntiles_here = cl.ntiles;
			pairs_here = cl.pairs;
			cells_here = cl.cells;
			if (share) {
				a.tlist = dc.d_tlist + cl.rank_first[(size_t)rank];
				a.nlocal = ntiles_here = cl.rank_first[(size_t)rank + 1] - cl.rank_first[(size_t)rank];
				a.dense_off = dc.d_doff;
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
		if (ctx->env.stamps) {
			SA_HIP_CHECK(hipMalloc(&d_stamps, sizeof(unsigned long long) * std::max<size_t>(stamp_words, 1)), return 1);
			SA_HIP_CHECK(hipMemset(d_stamps, 0, sizeof(unsigned long long) * std::max<size_t>(stamp_words, 1)), return 1);
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
		const int pk_wgs = ctx->env.pk_wgs ? ctx->env.pk_wgs : ctx->persistent_wgs / 32 * (ctx->leave_room ? 3 : 4);
		const int wgs = (int)std::min<int64_t>(is_pk ? pk_wgs : is_long ? ctx->long_wgs : ctx->persistent_wgs, ntiles_here);
		if (is_pk) {
			SA_HIP_CHECK(sa_launch_systolic_pk(ctx->sc.method, pb->g, pb->klo, pb->f16, a, wgs,
							   (unsigned)sa_pk_lds_bytes(ctx->sc.method, pb->g, pb->kmax), s), return 1);
		} else {
			SA_HIP_CHECK(sa_launch_systolic(ctx->sc.method, cls, a, wgs, s), return 1);
		}
		if (d_stamps) {
			std::vector<unsigned long long> h(stamp_words);
			SA_HIP_CHECK(hipStreamSynchronize(s), return 1);
			SA_HIP_CHECK(hipMemcpy(h.data(), d_stamps, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost), return 1);
			(void)hipFree(d_stamps);
			print_stamps(ctx, name, is_pk, h, nstamp);
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
	std::vector<std::pair<int64_t, int64_t>> runs = hp.generic;
	std::vector<int64_t> run_out;
	if (share) {
		runs.clear();
		for (const auto &gs : hp.generic_share[(size_t)rank]) {
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
			SaPairPlan pp(ctx->meta.data(), ctx->num);
			run_cells = pp.cells_before(run.first + run.second) - pp.cells_before(run.first);
		}
		if (!timed_end(sa_generic_kernel_name(ctx->sc.method), e0, e1, run.second, run_cells))
			return 1;
	}
	return 0;
}

int sa_align_range_impl(sa_ctx *ctx, int64_t start, int64_t count, int32_t *d_scores, void *stream, bool out16, int world, int rank,
			int32_t *host_out)
{
	if (!ctx || start < 0 || count < 0 || start > ctx->pairs - count || (!d_scores && count)) {
		sa_set_error("sa_ctx_align_range: bad range [%lld,+%lld) of %lld pairs", (long long)start,
			     (long long)count, ctx ? (long long)ctx->pairs : -1LL);
		return 1;
	}
	if (count == 0)
		return 0;
	SA_HIP_CHECK(hipSetDevice(ctx->device), return 1);
	hipStream_t s = (hipStream_t)stream;
	const bool share = world >= 1;
	if (!sa_plan_get(ctx, start, count, world, share ? host_out != nullptr : ctx->out_is_host))
		return 1;
	/* this call's counter slot (see sa_ctx: the invariant and the dirty flag) */
	const size_t slot = (size_t)(ctx->call_no++ % sa_ctx::COUNTER_SLOTS);
	if (ctx->slot_done[slot]) { /* 256 calls ago, possibly on another stream: normally long complete */
		SA_HIP_CHECK(hipStreamWaitEvent(s, ctx->slot_done[slot], 0), return 1);
	} else {
		SA_HIP_CHECK(hipEventCreateWithFlags(&ctx->slot_done[slot], hipEventDisableTiming), return 1);
	}
	if (ctx->slot_dirty[slot]) { /* (a launch of the failed call may still be running, on whatever stream: wait it out first) */
		SA_HIP_CHECK(hipDeviceSynchronize(), return 1);
		SA_HIP_CHECK(hipMemsetAsync(ctx->d_counters + slot * sa_ctx::COUNTERS_PER_SLOT, 0,
					    sizeof(unsigned) * sa_ctx::COUNTERS_PER_SLOT, s), return 1);
		ctx->slot_dirty[slot] = false;
	}
	int rc = 1;
	struct Mark { /* any way out other than success (error return, exception) may leave a launch behind that never counts out */
		sa_ctx *c;
		size_t slot;
		int *rc;
		~Mark()
		{
			if (*rc)
				c->slot_dirty[slot] = true;
		}
	} mark{ ctx, slot, &rc };
	rc = align_range_launches(ctx, start, count, d_scores, s, out16, world, rank, host_out, slot);
	return rc;
}

extern "C" int sa_ctx_align_range(sa_ctx *ctx, int64_t start, int64_t count, int32_t *d_scores, void *stream)
{
	return sa_guard("sa_ctx_align_range", 1, [&] { return sa_align_range_impl(ctx, start, count, d_scores, stream, false); });
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
	return sa_guard("sa_ctx_align_range16", 1,
			[&] { return sa_align_range_impl(ctx, start, count, reinterpret_cast<int32_t *>(d_scores), stream, true); });
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

/* ---- tile-interleaved sharding: one process (or one device of a process) per GPU, dense shares, all-gather, place ---- */
static bool share_args_ok(sa_ctx *ctx, int64_t start, int64_t count, int world, const char *who)
{
	if (!ctx || start < 0 || count <= 0 || start > ctx->pairs - count || world < 1 || world > 1024) {
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
	return sa_guard("sa_ctx_share_elems", (int64_t)-1, [&]() -> int64_t {
		if (!share_args_ok(ctx, start, count, world, "sa_ctx_share_elems"))
			return -1;
		SA_HIP_CHECK(hipSetDevice(ctx->device), return -1);
		if (!sa_plan_get(ctx, start, count, world, to_host != 0))
			return -1;
		return ctx->plan->h.share_elems;
	});
}

extern "C" int sa_ctx_align_share(sa_ctx *ctx, int64_t start, int64_t count, int world, int rank, void *d_share, int elem16,
				   int32_t *host_packed, void *stream)
{
	return sa_guard("sa_ctx_align_share", 1, [&] {
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
			if (!sa_host_range_is_pinned(host_packed + start, sizeof(int32_t) * (size_t)count) ||
			    hipHostGetDevicePointer(&dp, host_packed + start, 0) != hipSuccess) {
				(void)hipGetLastError();
				sa_set_error("sa_ctx_align_share: the host matrix is not page-locked over the range (sa_hip_host_register)");
				return 1;
			}
			host_dev = static_cast<int32_t *>(dp);
		}
		return sa_align_range_impl(ctx, start, count, static_cast<int32_t *>(d_share), stream, elem16 != 0, world, rank, host_dev);
	});
}

extern "C" int sa_ctx_place_shares(sa_ctx *ctx, int64_t start, int64_t count, int world, int to_host, const void *d_shares,
				    int elem16, int32_t *d_packed, void *stream)
{
	return sa_guard("sa_ctx_place_shares", 1, [&] {
		if (!share_args_ok(ctx, start, count, world, "sa_ctx_place_shares"))
			return 1;
		if (!d_shares || !d_packed) {
			sa_set_error("sa_ctx_place_shares: null buffer");
			return 1;
		}
		SA_HIP_CHECK(hipSetDevice(ctx->device), return 1);
		if (!sa_plan_get(ctx, start, count, world, to_host != 0))
			return 1;
		SA_HIP_CHECK(sa_launch_place(ctx->plan->d_segs, ctx->plan->nsegs, d_shares, elem16, d_packed, (hipStream_t)stream), return 1);
		return 0;
	});
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
