/*
 * sa_context.hip -- the device context: input validation and encoding, what each kernel family can reproduce for a
 * scoring (systolic_setup, pk_setup), uploads, arranged copies of the row store, instrumentation.
 *
 * Replaces, in the reference's device driver src/interface/seqalign_cuda.c:
 *   cuda_device_init :48-69   -> sa_sa_device_ready()
 *   uploads          :115-132 -> sa_ctx_create()
 * Differences by design: scoring state is passed explicitly (struct sa_scoring) instead of process globals; sequences
 * are uploaded pre-encoded (residue index per byte) so no kernel ever touches the ASCII->index table; there is no CPU
 * fallback.
 */
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "sa_ctx.h"

bool sa_device_ready(int device)
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

SaPlanInputs sa_plan_inputs(const sa_ctx *ctx)
{
	SaPlanInputs in;
	in.num = ctx->num;
	in.meta = ctx->meta.data();
	in.min_len = ctx->min_len;
	in.method = ctx->sc.method;
	in.gap_ext = ctx->sc.gap_ext;
	in.sys_ok = ctx->sys_ok;
	in.pk_kmax = ctx->pk_kmax;
	in.pk16_kmax = ctx->pk16_kmax;
	in.pk16_f16_kmax = ctx->pk16_f16_kmax;
	in.pk_chunk_cap = ctx->pk_chunk_cap;
	in.pk_q = ctx->pk_q;
	in.pk_floor = ctx->pk_floor;
	in.pk_gain = ctx->pk_gain;
	in.pk_slack = ctx->pk_slack;
	in.persistent_wgs = ctx->persistent_wgs;
	in.env_chunk = ctx->env.chunk;
	in.no_sort = ctx->env.no_sort;
	in.one_tile_size = ctx->env.one_tile_size;
	in.small_below = ctx->env.small_below;
	in.small_div = ctx->env.small_div;
	in.small_frac = ctx->env.small_frac;
	return in;
}

namespace {
struct CtxDeleter {
	void operator()(sa_ctx *c) const { sa_ctx_destroy(c); }
};
} // namespace

static sa_ctx *ctx_create_impl(int device, struct sa_input in, const struct sa_scoring *sc)
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
	const double encode_ms = sa_ms_since(t_encode);
	const SaEnv env = sa_env_read();
	const bool verbose = env.verbose;
	const auto t_create = std::chrono::steady_clock::now();
	auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_create).count(); };
	if (!sa_device_ready(device))
		return nullptr;
	const double device_ms = since();
	if (verbose)
		fprintf(stderr, "[seqalign_hip] sa_ctx_create: device ready at %.1f ms\n", since());

	std::unique_ptr<sa_ctx, CtxDeleter> holder(new sa_ctx()); /* (released on every early exit, exceptions included) */
	sa_ctx *ctx = holder.get();
	ctx->env = env;
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
	{ /* what each kernel family reproduces exactly for this scoring and these lengths (sa_limits.cpp) */
		const SaKernelLimits L = sa_kernel_limits(*sc, max_len, ctx->min_len, env.force_generic, env.no_pk, env.no_pk16);
		ctx->sys_ok = L.sys_ok;
		ctx->sys_pconst = L.sys_pconst;
		ctx->sys_q = L.sys_q;
		ctx->sys_gain = L.sys_gain;
		ctx->sys_slack = L.sys_slack;
		ctx->pk_kmax = L.pk_kmax;
		ctx->pk16_kmax = L.pk16_kmax;
		ctx->pk16_f16_kmax = L.pk16_f16_kmax;
		ctx->pk_chunk_cap = L.pk_chunk_cap;
		ctx->pk_pconst = L.pk_pconst;
		ctx->pk_q = L.pk_q;
		ctx->pk_floor = L.pk_floor;
		ctx->pk_gain = L.pk_gain;
		ctx->pk_slack = L.pk_slack;
		ctx->pk_extra = L.pk_extra;
	}
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
	if (!ok)
		return nullptr;
	return holder.release();
}

extern "C" sa_ctx *sa_ctx_create(int device, struct sa_input in, const struct sa_scoring *sc)
{
	return sa_guard("sa_ctx_create", (sa_ctx *)nullptr, [&] { return ctx_create_impl(device, in, sc); });
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
	sa_plan_release(ctx);
	sa_deliver_release(ctx);
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
/* Arranged copy of the store for the row streams of the packed kernels (sa_systolic_pk.inc): the permutation is
 * sa_arrange_rows (sa_plan.cpp); here the copy is built -- (codes, off) in position order plus rowmap
 * (position -> row) and posmap (row -> position) -- and uploaded, once per tile shape and context. */
bool sa_arranged_store(sa_ctx *ctx, const SaArrKey &key, const sa_ctx::Arranged **out)
{
	*out = nullptr;
	for (const auto &ar : ctx->arranged)
		if (ar.key == key) {
			*out = &ar;
			return true;
		}
	const int32_t num = ctx->num;
	if (!sa_arranged_exists(num, key))
		return true; /* no full block: nothing to arrange */
	const auto t_arr = std::chrono::steady_clock::now();
	struct Acc {
		sa_ctx *c;
		std::chrono::steady_clock::time_point t;
		~Acc() { c->setup.arrange += sa_ms_since(t); }
	} acc{ ctx, t_arr };
	std::vector<int32_t> rowmap, posmap((size_t)num), off_s((size_t)num + 1);
	sa_arrange_rows(ctx->meta.data(), num, key, rowmap);
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
	ar.key = key;
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
		ctx->arranged.push_back(ar);
		ok = true;
	} while (0);
	if (!ok) {
		(void)hipFree(ar.d_codes);
		(void)hipFree(ar.d_off);
		(void)hipFree(ar.d_rowmap);
		(void)hipFree(ar.d_posmap);
		return false;
	}
	*out = &ctx->arranged.back();
	return true;
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

static int timing_read_impl(sa_ctx *ctx, char *kernel_name, int cap, int64_t *launches, double *total_ms,
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

extern "C" int sa_ctx_timing_read(sa_ctx *ctx, char *kernel_name, int cap, int64_t *launches, double *total_ms,
				  int64_t *pairs, int64_t *cells, double *all_kernels_ms)
{
	return sa_guard("sa_ctx_timing_read", 1,
			[&] { return timing_read_impl(ctx, kernel_name, cap, launches, total_ms, pairs, cells, all_kernels_ms); });
}
