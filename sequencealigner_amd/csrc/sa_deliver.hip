/*
 * sa_deliver.hip -- host delivery: the launch/copy loop of cuda_align (src/interface/seqalign_cuda.c:182-292) on a ready
 * context.  Three schedules: direct stores into a page-locked packed matrix, L-shaped shells for the full
 * layout, double-buffered batches for everything else.
 */
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <thread>

#include "sa_ctx.h"

/* MemAvailable of the host: a destination larger than half of it (a file-backed matrix, src/io/output.c:36) is not
 * page-locked, its copies are staged by the runtime instead */
size_t sa_host_available_bytes()
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

static bool byte_is_pinned(const void *p)
{
	hipPointerAttribute_t attr;
	if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
		(void)hipGetLastError();
		return false;
	}
	return attr.type == hipMemoryTypeHost;
}

bool sa_host_range_is_pinned(const void *p, size_t bytes)
{
	if (!p || !bytes || !byte_is_pinned(p))
		return false;
	const char *lo = static_cast<const char *>(p), *hi = lo + bytes;
	/* the allocation / registration the first byte belongs to must reach past the last one */
	void *base = nullptr;
	size_t size = 0;
	if (hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t *>(&base), &size, const_cast<void *>(p)) == hipSuccess && base && size)
		return static_cast<const char *>(base) <= lo && hi <= static_cast<const char *>(base) + size;
	(void)hipGetLastError();
	/* (the runtime would not say: every page-table granule of the range must be locked -- 10 000 probes for 20 GB) */
	constexpr size_t STEP = (size_t)2 << 20;
	for (const char *q = lo + STEP; q < hi; q += STEP)
		if (!byte_is_pinned(q))
			return false;
	return byte_is_pinned(hi - 1);
}

/* May [p, p + bytes) be memory that malloc manages?  It is when it touches the brk segment ("[heap]" in /proc/self/maps), and it
 * is taken to be when it does not start on a page boundary: a mapping of the caller's own does (output_load's mmap,
 * src/io/output.c:55), a malloc block never does -- neither one in the heap, nor one malloc mapped by itself (16 bytes of header
 * in front), nor one in a heap extension that malloc mapped because brk was blocked (no "[heap]" label there).  Such a range is
 * never page-locked by this library: registering and unregistering memory that malloc goes on to hand out again is the one thing
 * both GPU memory faults on record have in common -- a fault address inside the heap, hit later by a pageable copy of the
 * runtime.  A destination there is served through the library's own pinned staging buffers. */
bool sa_host_range_in_malloc_heap(const void *p, size_t bytes)
{
	if (!p || !bytes)
		return false;
	if (reinterpret_cast<uintptr_t>(p) % 4096u != 0)
		return true;
	FILE *f = fopen("/proc/self/maps", "r");
	if (!f)
		return false;
	const uintptr_t lo = reinterpret_cast<uintptr_t>(p), hi = lo + bytes;
	bool hit = false;
	char line[512];
	while (fgets(line, sizeof(line), f)) {
		if (!strstr(line, "[heap]"))
			continue;
		unsigned long long a = 0, b = 0;
		if (sscanf(line, "%llx-%llx", &a, &b) == 2 && lo < (uintptr_t)b && hi > (uintptr_t)a)
			hit = true;
	}
	fclose(f);
	return hit;
}

/* some of [p, p + bytes) is page-locked, but not the range as one registration (pieces, a hole, a registration that ends
 * inside it): the runtime refuses a copy that crosses a registration boundary, and the range cannot be registered as a
 * whole either -- such a destination is filled by the host from the library's own pinned staging buffers */
static bool host_range_is_partly_pinned(const void *p, size_t bytes)
{
	if (!p || !bytes || sa_host_range_is_pinned(p, bytes))
		return false;
	const char *lo = static_cast<const char *>(p), *hi = lo + bytes;
	constexpr size_t STEP = (size_t)2 << 20;
	for (const char *q = lo; q < hi; q += STEP)
		if (byte_is_pinned(q))
			return true;
	return byte_is_pinned(hi - 1);
}

namespace {

constexpr int64_t BATCH_PAIRS = (int64_t)64 << 20; /* reference batch: src/interface/seqalign_cuda.c:136 */
constexpr int64_t FINAL_BATCH_PAIRS = (int64_t)3 << 20; /* the batch whose device->host copy nothing overlaps */

int64_t tri_of(int64_t j) { return sa_tri(j); }

/* scatter a packed slice into the full symmetric host matrix (what output_fill does per column,
 * reference src/io/output.c:76-81) */
void host_scatter_full(int32_t *matrix, size_t dim, const int32_t *slice, int64_t start, int64_t count)
{
	int64_t p = start;
	int64_t j = sa_column_of(p);
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

void release_impl(sa_ctx *ctx)
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
	if (d.h_counters)
		(void)hipHostFree(d.h_counters);
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
		int64_t jn = sa_column_of(std::min(tri_of(j) + want, tri_of(j1) - 1)) + 1;
		jn = std::min(std::max(jn, j + 1), j1);
		cuts.push_back(jn);
		j = jn;
	}
	/* progress: one event per batch, looked at (never waited for) while the host issues the next batches */
	const bool progress = sa_progress_wanted();
	std::vector<hipEvent_t> batch_done;
	struct Events {
		std::vector<hipEvent_t> &v;
		~Events()
		{
			for (hipEvent_t e : v)
				(void)hipEventDestroy(e);
		}
	} events_guard{ batch_done };
	if (progress)
		for (size_t b = 0; b + 1 < cuts.size(); b++) {
			hipEvent_t e = nullptr;
			SA_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming), return false);
			batch_done.push_back(e);
		}
	size_t reported = 0;
	auto report_finished = [&]() {
		while (reported < batch_done.size() && hipEventQuery(batch_done[reported]) == hipSuccess) {
			reported++;
			sa_report_progress((double)(tri_of(cuts[reported]) - start) / (double)std::max<int64_t>(1, tri_of(j1) - start));
		}
		(void)hipGetLastError();
	};
	const auto t_phase = std::chrono::steady_clock::now();
	for (size_t b = 0; b + 1 < cuts.size(); b++) {
		const int64_t ja = cuts[b], jb = cuts[b + 1];
		const int64_t lo = tri_of(ja), cnt = tri_of(jb) - lo;
		if (cnt > 0 && sa_align_range_impl(ctx, lo, cnt, d.d_packed + (lo - start), d.compute, false))
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
		if (progress) {
			SA_HIP_CHECK(hipEventRecord(batch_done[b], d.compute), return false);
			report_finished();
		}
	}
	if (progress) /* wait in short slices so that the bar keeps moving; never past completion */
		while (hipStreamQuery(d.compute) == hipErrorNotReady) {
			std::this_thread::sleep_for(std::chrono::microseconds(250));
			report_finished();
		}
	(void)hipGetLastError();
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
		     bool stage_packed, double &phase_seconds)
{
	auto &d = ctx->dl;
	const size_t dim = (size_t)ctx->num;
	const bool stage = out.matrix && (!out.triangular || stage_packed);
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
		if (stage && out.triangular)
			memcpy(out.matrix + pd.start, d.h_stage[pd.buf], sizeof(int32_t) * (size_t)pd.count);
		else if (stage)
			host_scatter_full(out.matrix, dim, d.h_stage[pd.buf], pd.start, pd.count);
		pend[0] = pend[1];
		npend--;
		sa_report_progress((double)(pd.start + pd.count - lo) / (double)std::max<int64_t>(1, total));
		return true;
	};
	while (issued < total) {
		const int b = nb & 1;
		if (npend == 2 && !deliver_oldest())
			return false;
		/* `shrink`: half of what is left (see deliver_full_shells), else fixed-size batches */
		const int64_t left = total - issued;
		const int64_t cnt = !shrink || left <= FINAL_BATCH_PAIRS ? std::min(batch, left) : std::min(batch, (left + 1) / 2);
		if (sa_align_range_impl(ctx, lo + issued, cnt, d.d_buf[b], d.compute, false))
			return false;
		SA_HIP_CHECK(hipEventRecord(d.done[b], d.compute), return false);
		SA_HIP_CHECK(hipStreamWaitEvent(d.copy, d.done[b], 0), return false);
		if (out.matrix) {
			int32_t *dst = stage ? d.h_stage[b] : out.matrix + lo + issued;
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

void sa_deliver_release(sa_ctx *ctx) { release_impl(ctx); }

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
	if (sa_host_range_in_malloc_heap(p, bytes)) {
		sa_set_error("sa_hip_host_register: the range lies in the malloc heap or does not start on a page boundary; memory that malloc "
			     "manages is not page-locked (allocate the matrix with mmap, as output_load does: src/io/output.c:55)");
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

static int align_host_impl(sa_ctx *ctx, int64_t start, int64_t count, struct sa_output out, double *phase_seconds)
{
	if (phase_seconds)
		*phase_seconds = 0.0;
	if (!ctx || start < 0 || count < 0 || start > ctx->pairs - count) {
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
	if (out.matrix && !out.triangular && !ctx->env.no_shells) {
		j0 = start == 0 ? 0 : sa_column_of(start);
		j1 = sa_column_of(start + count - 1) + 1;
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
	/* (a destination that is page-locked only in part -- pieces, a hole: see host_range_is_partly_pinned -- is neither
	 * registered here nor handed to the runtime as a copy target: the host fills it from pinned staging buffers) */
	const bool partly = out.matrix && host_range_is_partly_pinned(out.triangular ? out.matrix + start : out.matrix,
								      sizeof(int32_t) * (out.triangular ? (size_t)total : dim * dim));
	if (partly)
		shells = false;
	void *pinned_here = nullptr;
	if (out.matrix && !partly && !ctx->env.no_pin && (out.triangular || shells)) {
		int32_t *base = out.triangular ? out.matrix + start : out.matrix;
		const size_t bytes = sizeof(int32_t) * (out.triangular ? (size_t)total : dim * dim);
		const size_t avail = sa_host_available_bytes();
		if (!sa_host_range_is_pinned(base, bytes) && (!avail || bytes <= avail / 2) && !sa_host_range_in_malloc_heap(base, bytes)) {
			const auto t_pin = std::chrono::steady_clock::now();
			if (hipHostRegister(base, bytes, hipHostRegisterDefault) == hipSuccess)
				pinned_here = base;
			else
				(void)hipGetLastError();
			ctx->setup.pin += sa_ms_since(t_pin);
		}
	}
	/* Packed destination that is page-locked: the kernels store their scores straight into it.  The epilogue's
	 * coalesced 128-byte runs leave the chip as posted PCIe writes while the next tiles compute -- ~6 GB/s at
	 * cfg 2's rate against a link that moves ~50 -- so there is no copy pass, no batching and a single launch tail. */
	int32_t *direct = nullptr;
	if (out.matrix && out.triangular && !ctx->env.no_direct &&
	    sa_host_range_is_pinned(out.matrix + start, sizeof(int32_t) * (size_t)total)) {
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
		if (out.matrix && (!out.triangular || partly))
			for (int k = 0; k < 2; k++)
				if (!grow(d.h_stage[k], d.stage_elems[k], batch, true))
					return 1;
	}
	if (direct && !sa_prepare_range(ctx, start, total, true))
		return 1;
	unsigned *h_cnt = nullptr; /* page-locked mirror of the tile counters, only when somebody listens */
	if (direct && sa_progress_wanted()) {
		if (!d.h_counters && hipHostMalloc(reinterpret_cast<void **>(&d.h_counters), sizeof(unsigned) * 64) == hipSuccess)
			d.h_counters_n = 64;
		(void)hipGetLastError();
		h_cnt = d.h_counters;
	}
	SA_HIP_CHECK(hipDeviceSynchronize(), return 1);

	/* ---- the launch/copy loop: what the reference brackets with bench_align_start/end ---- */
	double phase = 0.0;
	bool ok;
	if (direct) {
		const auto t_phase = std::chrono::steady_clock::now();
		ctx->out_is_host = true;
		ok = sa_align_range_impl(ctx, start, total, direct, d.compute, false) == 0;
		ctx->out_is_host = false;
		if (ok && h_cnt && !ctx->prog_items.empty()) {
			/* Progress (the reference's pproportc in its batch loop, src/interface/seqalign_cuda.c:286-289): one launch does
			 * the whole range here, so the host looks at the launches' tile counters every 50 ms.  It waits in 250 us
			 * slices and leaves the moment the stream is done: the poll adds nothing measurable to the phase (a 50 ms sleep
			 * in this loop once turned cfg 2's 15 ms phase into 50).  A counter goes back to zero when its launch ends:
			 * fractions only ever grow. */
			const size_t nit = std::min<size_t>(ctx->prog_items.size(), (size_t)d.h_counters_n);
			int64_t all = 0;
			for (size_t k = 0; k < nit; k++)
				all += ctx->prog_items[k].tiles;
			double shown = 0.0;
			std::vector<int64_t> seen(nit, 0);
			auto next_look = std::chrono::steady_clock::now() + std::chrono::milliseconds(50);
			while (hipStreamQuery(d.compute) == hipErrorNotReady) {
				std::this_thread::sleep_for(std::chrono::microseconds(250));
				if (std::chrono::steady_clock::now() < next_look)
					continue;
				next_look += std::chrono::milliseconds(50);
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
					sa_report_progress(shown = f);
			}
			(void)hipGetLastError();
		}
		if (ok) {
			SA_HIP_CHECK(hipStreamSynchronize(d.compute), ok = false);
		}
		phase = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_phase).count();
	} else {
		ok = shells ? deliver_full_shells(ctx, j0, j1, out.matrix, phase)
			    : deliver_batches(ctx, start, total, out, batch, out.matrix != nullptr, partly && out.triangular, phase);
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

extern "C" int sa_ctx_align_host(sa_ctx *ctx, int64_t start, int64_t count, struct sa_output out, double *phase_seconds)
{
	return sa_guard("sa_ctx_align_host", 1, [&] { return align_host_impl(ctx, start, count, out, phase_seconds); });
}
