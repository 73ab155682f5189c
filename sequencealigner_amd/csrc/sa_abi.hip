/*
 * sa_abi.hip -- the entry points a host program binds: sa_hip_align / sa_hip_memory / sa_hip_filter (the reference's
 * cuda_align / cuda_memory / filter), the pair-space arithmetic, progress and timing side channels.
 *
 * Replaces src/interface/seqalign_cuda.c:
 *   cuda_memory :71-93   -> sa_hip_memory()
 *   cuda_align  :95-296  -> sa_hip_align() on top of sa_ctx_create() / sa_ctx_align_host()
 * Every function with a body that can allocate host memory runs it behind sa_guard (sa_ctx.h): failures -- exceptions
 * included -- come back as the failure value + sa_last_error(), like the reference's perr + return false (:23-30).
 */
#include <algorithm>
#include <atomic>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <execinfo.h>
#include <fcntl.h>
#include <mutex>
#include <thread>
#include <unistd.h>

#include "sa_ctx.h"

/* SA_HIP_ABORT_TRACE=<file> (diagnostics): a C backtrace of whoever raises SIGABRT in this process, appended to the file;
 * then the handler that was installed before (Python's faulthandler, the default action) takes over.  Why a file: a test
 * runner that captures fd 2 swallows what the runtime, the allocator or libstdc++ print before they abort. */
static char g_abort_trace_path[512];
static struct sigaction g_abort_prev;
static void sa_abort_trace(int sig)
{
	void *frames[64];
	const int n = backtrace(frames, 64);
	const int fd = open(g_abort_trace_path, O_WRONLY | O_CREAT | O_APPEND, 0644);
	if (fd >= 0) {
		backtrace_symbols_fd(frames, n, fd);
		close(fd);
	}
	sigaction(SIGABRT, &g_abort_prev, nullptr);
	raise(sig);
}

__attribute__((constructor)) static void sa_runtime_knobs(void)
{
	if (const char *p = getenv("SA_HIP_ABORT_TRACE")) {
		snprintf(g_abort_trace_path, sizeof(g_abort_trace_path), "%s", p);
		void *warm[4];
		(void)backtrace(warm, 4); /* (loads libgcc now: the first call allocates, which a signal handler must not) */
		struct sigaction sa;
		memset(&sa, 0, sizeof(sa));
		sa.sa_handler = sa_abort_trace;
		sigemptyset(&sa.sa_mask);
		sa.sa_flags = SA_NODEFER;
		sigaction(SIGABRT, &sa, &g_abort_prev);
	}
}

static std::atomic<sa_progress_fn> g_progress_fn{ nullptr };
static std::atomic<void *> g_progress_user{ nullptr };

extern "C" void sa_hip_set_progress(sa_progress_fn fn, void *user)
{
	g_progress_user.store(user);
	g_progress_fn.store(fn);
}

/* (sa_hip_align on several devices runs one thread per slice: the first slice's thread speaks for the job, so with more
 * than one device the callback comes from a worker thread of the library, one at a time) */
static thread_local bool t_progress_here = true;

void sa_progress_speaker(bool here) { t_progress_here = here; }
bool sa_progress_wanted() { return t_progress_here && g_progress_fn.load() != nullptr; }

void sa_report_progress(double fraction)
{
	if (!t_progress_here)
		return;
	if (sa_progress_fn fn = g_progress_fn.load())
		fn(fraction < 0 ? 0 : fraction > 1 ? 1 : fraction, g_progress_user.load());
}

/* ---- sa_hip_filter: device-assisted similarity filter (reference src/bio/filter.c:14-89) ---- */
static int32_t filter_impl(struct sa_input in, float threshold, uint8_t *keep)
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
	if (!sa_device_ready(0))
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

	struct Bufs { /* (released whatever way the function is left) */
		uint8_t *d_blob = nullptr;
		int32_t *d_off = nullptr;
		unsigned long long *d_rel = nullptr, *h_rel = nullptr;
		~Bufs()
		{
			(void)hipFree(d_blob);
			(void)hipFree(d_off);
			(void)hipFree(d_rel);
			if (h_rel)
				(void)hipHostFree(h_rel);
		}
	} bufs;
	uint8_t *&d_blob = bufs.d_blob;
	int32_t *&d_off = bufs.d_off;
	unsigned long long *&d_rel = bufs.d_rel, *&h_rel = bufs.h_rel;
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
		const bool verbose = sa_env_read().verbose;
		double ms_kernel = 0, ms_copy = 0, ms_host = 0;
		keptbits[0] = 1ULL; /* sequence 0 is always kept */
		bool failed = false;
		for (int32_t jt0 = 0; jt0 < tiles && !failed; jt0 += band_tiles) {
			const int32_t rows_t = std::min(band_tiles, tiles - jt0);
			const long long j_lo = 64LL * jt0, j_hi = std::min<long long>(num, 64LL * (jt0 + rows_t));
			const long long base = sa_filter_row_offset(j_lo), words = sa_filter_row_offset(j_hi) - base;
			const auto t_band = std::chrono::steady_clock::now();
			SA_HIP_CHECK(sa_launch_filter_relation(d_blob, d_off, num, threshold, d_rel, jt0, rows_t, nullptr), failed = true; break);
			if (verbose) {
				SA_HIP_CHECK(hipDeviceSynchronize(), failed = true; break);
				ms_kernel += sa_ms_since(t_band);
			}
			const auto t_copy = std::chrono::steady_clock::now();
			SA_HIP_CHECK(hipMemcpy(h_rel, d_rel, sizeof(unsigned long long) * (size_t)words, hipMemcpyDeviceToHost), failed = true; break);
			ms_copy += sa_ms_since(t_copy);
			const auto t_host = std::chrono::steady_clock::now();
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
			ms_host += sa_ms_since(t_host);
		}
		if (verbose)
			fprintf(stderr, "[seqalign_hip] sa_hip_filter: %d sequences: relation kernels %.1f ms, download %.1f ms, greedy pass on the host %.1f ms\n",
				num, ms_kernel, ms_copy, ms_host);
		if (failed)
			break;
		kept = 0;
		for (int32_t k = 0; k < num; k++)
			kept += keep[k];
	} while (0);
	return kept;
}

extern "C" int32_t sa_hip_filter(struct sa_input in, float threshold, uint8_t *keep)
{
	return sa_guard("sa_hip_filter", (int32_t)-1, [&] { return filter_impl(in, threshold, keep); });
}

/* ---- sa_hip_align: the cuda_align replacement (host buffers in, host matrix out) ---------- */

/* launch/copy phase of the last successful sa_hip_align call (the reference's bench_align_start..end bracket) */
static std::atomic<double> g_last_align_seconds{ 0.0 };
static std::mutex g_breakdown_mutex;
static double g_breakdown[SA_BREAKDOWN_COUNT] = {};
static std::atomic<int> g_last_align_path{ 0 };

/* devices sa_hip_align spreads a job over: all visible ones, or the first SA_HIP_DEVICES */
int sa_devices_in_use(const SaEnv &env)
{
	int ndev = sa_hip_device_count();
	if (env.devices >= 1 && env.devices < ndev)
		ndev = env.devices;
	return ndev;
}

/* reference src/interface/seqalign_cuda.c:71-93; with several devices the answer must hold on each of them */
extern "C" bool sa_hip_memory(size_t bytes)
{
	return sa_guard("sa_hip_memory", false, [&] {
		const int ndev = sa_devices_in_use(sa_env_read());
		if (ndev <= 0) {
			sa_set_error("No HIP devices available; libseqalign_hip has no CPU fallback");
			return false;
		}
		const long double need = (long double)bytes * 4.0L / 3.0L;
		for (int dev = 0; dev < ndev; dev++) {
			if (!sa_device_ready(dev))
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
	});
}

/* Several devices, every one delivering a contiguous slice of the packed index straight into the host matrix: one host
 * thread and one context per slice, cut by DP work (sa_pairs_partition).  The alternative to the all-gather path
 * (sa_gather.hip) for jobs whose matrix does not fit a device three times over, for hosts without RCCL, and for slices
 * folded onto fewer devices (SA_HIP_SPLIT). */
static bool align_slices(struct sa_input in, struct sa_output out, const struct sa_scoring *sc, const SaEnv &env, int ndev, int nvisible)
{
	const int64_t pairs = (int64_t)in.num * (in.num - 1) / 2;
	std::vector<int64_t> bounds((size_t)ndev + 1, 0);
	bounds[(size_t)ndev] = pairs;
	if (ndev > 1) {
		if (sa_pairs_partition(in.meta, in.num, ndev, bounds.data()))
			return false;
		/* full layout: the cuts move to column starts so that every slice is a set of whole columns (shell schedule) */
		if (out.matrix && !out.triangular)
			for (int k = 1; k < ndev; k++)
				bounds[(size_t)k] = std::max(bounds[(size_t)k - 1], sa_tri(sa_column_of(bounds[(size_t)k])));
	}
	/* several slices share one destination: page-lock it once for all of them (set-up, outside the timed phase) */
	struct Pin {
		void *p = nullptr;
		~Pin()
		{
			if (p) {
				(void)hipSetDevice(0);
				(void)hipHostUnregister(p);
			}
		}
	} pin;
	double pin_all_ms = 0.0;
	if (ndev > 1 && out.matrix && !env.no_pin && sa_device_ready(0)) {
		const size_t n = (size_t)in.num;
		const size_t bytes = sizeof(int32_t) * (out.triangular ? (size_t)pairs : n * n);
		const size_t avail = sa_host_available_bytes();
		if (!sa_host_range_is_pinned(out.matrix, bytes) && (!avail || bytes <= avail / 2) && !sa_host_range_in_malloc_heap(out.matrix, bytes)) {
			const auto t_pin = std::chrono::steady_clock::now();
			if (hipHostRegister(out.matrix, bytes, hipHostRegisterPortable) == hipSuccess)
				pin.p = out.matrix;
			else
				(void)hipGetLastError();
			pin_all_ms = sa_ms_since(t_pin);
		}
	}
	std::vector<std::string> errs((size_t)ndev);
	std::vector<char> oks((size_t)ndev, 0);
	std::vector<double> phases((size_t)ndev, 0.0);
	auto run = [&](int k) noexcept { /* a thread body: nothing may escape it */
		sa_guard_void("sa_hip_align", [&] {
			const int64_t lo = bounds[(size_t)k], hi = bounds[(size_t)k + 1];
			if (hi <= lo && k > 0) { /* (slice 0 always builds its context: that is where the input is validated) */
				oks[(size_t)k] = 1;
				return;
			}
			const auto t_slice = std::chrono::steady_clock::now();
			sa_progress_speaker(k == 0);
			sa_ctx *ctx = sa_ctx_create(k % nvisible, in, sc);
			if (ctx && sa_ctx_align_host(ctx, lo, hi - lo, out, &phases[(size_t)k]) == 0)
				oks[(size_t)k] = 1;
			if (ctx && k == 0) { /* the first slice speaks for the call */
				std::lock_guard<std::mutex> g(g_breakdown_mutex);
				const double v[SA_BREAKDOWN_COUNT] = { ctx->setup.encode, ctx->setup.device, ctx->setup.upload, ctx->setup.code_objects,
								       ctx->setup.pin + pin_all_ms, ctx->setup.plan, ctx->setup.arrange, phases[0] * 1e3,
								       sa_ms_since(t_slice) };
				memcpy(g_breakdown, v, sizeof(v));
			}
			sa_ctx_destroy(ctx);
		});
		if (!oks[(size_t)k]) {
			try {
				errs[(size_t)k] = sa_last_error(); /* (the error text is per thread) */
			} catch (...) {
			}
		}
		sa_progress_speaker(true);
	};
	if (ndev == 1) {
		run(0);
	} else {
		std::vector<std::thread> threads;
		threads.reserve((size_t)ndev);
		std::string spawn_error;
		for (int k = 0; k < ndev; k++) {
			try {
				threads.emplace_back(run, k);
			} catch (const std::exception &e) { /* no thread for this slice: the ones already running are joined below */
				spawn_error = e.what();
				break;
			}
		}
		for (auto &t : threads)
			t.join();
		if (!spawn_error.empty()) {
			sa_set_error("sa_hip_align: cannot start a host thread per device: %s", spawn_error.c_str());
			return false;
		}
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

/* does every device have room for the all-gather schedule?  (share + gathered shares + placed matrix, + N^2 for the full layout) */
static bool gather_fits(const struct sa_input &in, const struct sa_output &out, int ndev)
{
	const long double pairs = (long double)in.num * (in.num - 1) / 2;
	const long double n2 = (long double)in.num * in.num;
	long double need = 4.0L * pairs * (1.02L + 1.02L / ndev) + 4.0L * pairs + (out.matrix && !out.triangular ? 4.0L * n2 : 0.0L);
	need = need * 4 / 3 + ((long double)(1 << 30));
	for (int dev = 0; dev < ndev; dev++) {
		if (hipSetDevice(dev) != hipSuccess)
			return false;
		size_t free_b = 0, total_b = 0;
		if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || (long double)free_b < need) {
			(void)hipGetLastError();
			return false;
		}
	}
	return true;
}

static bool align_impl(struct sa_input in, struct sa_output out, const struct sa_scoring *sc)
{
	if (!sc) {
		sa_set_error("sa_hip_align: null scoring");
		return false;
	}
	if (out.matrix && out.dim != (size_t)in.num) {
		sa_set_error("sa_hip_align: output dim %zu does not match %d sequences", out.dim, in.num);
		return false;
	}
	const SaEnv env = sa_env_read();
	int ndev = sa_devices_in_use(env);
	if (ndev <= 0) {
		sa_set_error("No HIP devices available; libseqalign_hip has no CPU fallback");
		return false;
	}
	const int64_t pairs = (int64_t)in.num * (in.num - 1) / 2;
	/* SA_HIP_SPLIT=n (testing aid): run the slice path with n work-balanced slices even when fewer devices are
	 * visible -- slice k goes to device k mod visible */
	const int nvisible = ndev;
	const bool split = env.split >= 2 && pairs >= env.split;
	if (split)
		ndev = env.split;
	else if (pairs < (int64_t)ndev * 4096)
		ndev = 1;
	/* Several devices: dense shares + RCCL all-gather + placement (sa_gather.hip) whenever RCCL can be bound and
	 * the matrix fits the devices; SA_HIP_GATHER=0 / 1 forces the choice (1 also with a single device: a one-rank
	 * communicator, the rehearsal the one-GPU tests run). */
	bool gather = !split && (env.gather == 1 || (env.gather != 0 && ndev > 1));
	if (gather && env.gather != 1 && !(sa_rccl_available(nullptr) && gather_fits(in, out, ndev)))
		gather = false;
	if (gather) {
		std::vector<int> devs((size_t)ndev);
		for (int k = 0; k < ndev; k++)
			devs[(size_t)k] = k;
		double phase = 0.0, bd[SA_BREAKDOWN_COUNT] = {};
		if (!sa_align_gathered(in, out, sc, devs.data(), ndev, &phase, bd))
			return false;
		{
			std::lock_guard<std::mutex> g(g_breakdown_mutex);
			memcpy(g_breakdown, bd, sizeof(bd));
		}
		g_last_align_seconds.store(phase);
		g_last_align_path.store(2);
		return true;
	}
	if (!align_slices(in, out, sc, env, ndev, nvisible))
		return false;
	g_last_align_path.store(1);
	return true;
}

extern "C" bool sa_hip_align(struct sa_input in, struct sa_output out, const struct sa_scoring *sc)
{
	return sa_guard("sa_hip_align", false, [&] { return align_impl(in, out, sc); });
}

extern "C" int sa_hip_last_align_path(void) { return g_last_align_path.load(); }

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
