/*
 * sa_gather.hip -- several devices of ONE host process, the way BASELINE.json's north_star words it: "the pair space is
 * tiled across the GPUs with an RCCL all-gather over xGMI to assemble the full similarity matrix before HDF5 write".
 *
 * The reference has a single device (src/interface/seqalign_cuda.c:65) and goes align -> write (src/main.c:31-34); this
 * is what stands between the two on a node with several MI355X:
 *
 *   set-up (outside the timed phase, like the reference's allocations :125-168)
 *     ncclCommInitAll over the devices in use; one context per device (inputs replicated); per device the dense share,
 *     the gathered shares, the placed packed matrix (and N^2 ints for the full layout); the host matrix page-locked once
 *   phase (what bench_align_start..end brackets, :182-292: launches AND device->host copies)
 *     1. sa_ctx_align_share on every device: its tiles of the job-wide tile list, dense, int16 when the scores fit;
 *        packed destination: the same kernels store the rank's scores straight into the host matrix
 *     2. one grouped ncclAllGather of the shares (ndev x share_elems elements on every device)
 *     3. sa_ctx_place_shares on every device: the reference's packed order, the whole matrix resident on every GPU
 *     4. full destination: every device expands and copies the L-shaped shell of ITS column range (equal areas), so the
 *        N^2 host matrix is written once, over all the PCIe links
 *
 * RCCL is bound at run time (dlopen of librccl.so.1): a host process that already carries an RCCL -- a PyTorch process
 * does -- shares it, and a single-GPU user of libseqalign_hip.so never loads the 500 MB library.
 */
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <mutex>

#include <rccl/rccl.h> /* types and prototypes only: nothing here links against librccl */

#include "sa_ctx.h"

namespace {

struct Rccl {
	void *lib = nullptr;
	decltype(&ncclCommInitAll) CommInitAll = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	decltype(&ncclGroupStart) GroupStart = nullptr;
	decltype(&ncclGroupEnd) GroupEnd = nullptr;
	decltype(&ncclAllGather) AllGather = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
	std::string why;
};

Rccl &rccl()
{
	static Rccl r;
	static std::once_flag once;
	std::call_once(once, [] {
		const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
		for (const char *n : names)
			if ((r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL)))
				break;
		if (!r.lib) {
			r.why = std::string("librccl.so.1 not loadable: ") + (dlerror() ? dlerror() : "?");
			return;
		}
		auto sym = [&](const char *s) -> void * {
			void *p = dlsym(r.lib, s);
			if (!p && r.why.empty())
				r.why = std::string("librccl lacks ") + s;
			return p;
		};
		r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
		r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
		r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
		r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
		r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
		r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
	});
	return r;
}

#define SA_NCCL_CHECK(call, onfail)                                                                        \
	if (ncclResult_t res__ = (call); res__ != ncclSuccess) {                                           \
		sa_set_error("%s failed: %s", #call, R.GetErrorString ? R.GetErrorString(res__) : "?");     \
		onfail;                                                                                    \
	} else                                                                                             \
		(void)0

/* everything one device holds for the job; released in reverse by the destructor whatever way the call ends */
struct Dev {
	int device = -1;
	sa_ctx *ctx = nullptr;
	hipStream_t stream = nullptr, copy = nullptr;
	hipEvent_t placed = nullptr;
	void *d_share = nullptr, *d_gathered = nullptr;
	int32_t *d_packed = nullptr, *d_full = nullptr;
	ncclComm_t comm = nullptr;
	int64_t ja = 0, jb = 0; /* full layout: the columns whose shell this device expands and copies */
};

struct Job {
	std::vector<Dev> devs;
	Rccl *R = nullptr;
	void *pinned_here = nullptr;
	~Job()
	{
		for (auto &d : devs) {
			if (d.device < 0)
				continue;
			(void)hipSetDevice(d.device);
			if (d.stream)
				(void)hipStreamSynchronize(d.stream);
			if (d.copy)
				(void)hipStreamSynchronize(d.copy);
			if (d.comm && R && R->CommDestroy)
				(void)R->CommDestroy(d.comm);
			(void)hipFree(d.d_share);
			(void)hipFree(d.d_gathered);
			(void)hipFree(d.d_packed);
			(void)hipFree(d.d_full);
			if (d.placed)
				(void)hipEventDestroy(d.placed);
			if (d.stream)
				(void)hipStreamDestroy(d.stream);
			if (d.copy)
				(void)hipStreamDestroy(d.copy);
			sa_ctx_destroy(d.ctx);
		}
		if (pinned_here)
			(void)hipHostUnregister(pinned_here);
		(void)hipGetLastError();
	}
};

} // namespace

bool sa_rccl_available(std::string *why)
{
	Rccl &R = rccl();
	const bool ok = R.lib && R.CommInitAll && R.CommDestroy && R.GroupStart && R.GroupEnd && R.AllGather;
	if (!ok && why)
		*why = R.why;
	return ok;
}

bool sa_align_gathered(struct sa_input in, struct sa_output out, const struct sa_scoring *sc, const int *devices, int ndev,
		       double *phase_seconds, double *breakdown_ms)
{
	Rccl &R = rccl();
	std::string why;
	if (!sa_rccl_available(&why)) {
		sa_set_error("RCCL all-gather path: %s", why.c_str());
		return false;
	}
	if (ndev < 1 || !devices || !sc) {
		sa_set_error("sa_align_gathered: bad arguments");
		return false;
	}
	const int64_t pairs = (int64_t)in.num * (in.num - 1) / 2;
	const size_t dim = (size_t)in.num;
	Job job;
	job.R = &R;
	job.devs.resize((size_t)ndev);
	const auto t_all = std::chrono::steady_clock::now();

	/* ---- set-up -------------------------------------------------------------------------------------------------- */
	for (int k = 0; k < ndev; k++) {
		Dev &d = job.devs[(size_t)k];
		d.ctx = sa_ctx_create(devices[k], in, sc); /* (validates the input: the first failure speaks for the call) */
		if (!d.ctx)
			return false;
		d.device = devices[k];
		SA_HIP_CHECK(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking), return false);
		SA_HIP_CHECK(hipStreamCreateWithFlags(&d.copy, hipStreamNonBlocking), return false);
		SA_HIP_CHECK(hipEventCreateWithFlags(&d.placed, hipEventDisableTiming), return false);
	}
	{ /* one communicator per device, all of this process (several slices folded onto one device cannot share a clique) */
		std::vector<int> sorted(devices, devices + ndev);
		std::sort(sorted.begin(), sorted.end());
		if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) {
			sa_set_error("RCCL all-gather path: every slice needs a device of its own");
			return false;
		}
		std::vector<ncclComm_t> comms((size_t)ndev, nullptr);
		SA_NCCL_CHECK(R.CommInitAll(comms.data(), ndev, devices), return false);
		for (int k = 0; k < ndev; k++)
			job.devs[(size_t)k].comm = comms[(size_t)k];
	}
	const bool elem16 = sa_ctx_scores_fit16(job.devs[0].ctx) != 0;
	const size_t esz = elem16 ? sizeof(int16_t) : sizeof(int32_t);
	const bool want_host = out.matrix != nullptr;
	const bool full = want_host && !out.triangular;
	/* the host matrix, page-locked once for all devices */
	int32_t *host_packed = nullptr; /* packed destination: the share kernels deliver their own scores */
	if (want_host && !job.devs[0].ctx->env.no_pin) {
		const size_t bytes = sizeof(int32_t) * (out.triangular ? (size_t)pairs : dim * dim);
		const auto t_pin = std::chrono::steady_clock::now();
		(void)hipSetDevice(devices[0]);
		if (!sa_host_range_is_pinned(out.matrix, bytes)) {
			const size_t avail = sa_host_available_bytes();
			if ((!avail || bytes <= avail / 2) && !sa_host_range_in_malloc_heap(out.matrix, bytes) &&
			    hipHostRegister(out.matrix, bytes, hipHostRegisterPortable) == hipSuccess)
				job.pinned_here = out.matrix;
			else
				(void)hipGetLastError();
		}
		if (out.triangular && !job.devs[0].ctx->env.no_direct && sa_host_range_is_pinned(out.matrix, bytes))
			host_packed = out.matrix;
		if (breakdown_ms)
			breakdown_ms[SA_BREAKDOWN_PIN] += sa_ms_since(t_pin);
	}
	int64_t share_elems = -1;
	for (int k = 0; k < ndev; k++) {
		Dev &d = job.devs[(size_t)k];
		const int64_t e = sa_ctx_share_elems(d.ctx, 0, pairs, ndev, host_packed != nullptr);
		if (e < 0)
			return false;
		if (share_elems >= 0 && e != share_elems) {
			sa_set_error("RCCL all-gather path: the devices disagree on the share size (%lld vs %lld)", (long long)e, (long long)share_elems);
			return false;
		}
		share_elems = e;
		SA_HIP_CHECK(hipSetDevice(d.device), return false);
		SA_HIP_CHECK(hipMalloc(&d.d_share, esz * (size_t)share_elems), return false);
		SA_HIP_CHECK(hipMalloc(&d.d_gathered, esz * (size_t)share_elems * (size_t)ndev), return false);
		SA_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&d.d_packed), sizeof(int32_t) * (size_t)pairs), return false);
		if (full) {
			SA_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&d.d_full), sizeof(int32_t) * dim * dim), return false);
		}
	}
	if (full) { /* shells of equal area: the shell of columns [ja, jb) has jb^2 - ja^2 elements */
		for (int k = 0; k < ndev; k++) {
			const double lo = std::sqrt((double)k / ndev) * (double)dim, hi = std::sqrt((double)(k + 1) / ndev) * (double)dim;
			job.devs[(size_t)k].ja = k == 0 ? 0 : (int64_t)lo;
			job.devs[(size_t)k].jb = k == ndev - 1 ? (int64_t)dim : (int64_t)hi;
		}
		for (int k = 1; k < ndev; k++)
			job.devs[(size_t)k].ja = job.devs[(size_t)k - 1].jb;
	}
	for (auto &d : job.devs) {
		SA_HIP_CHECK(hipSetDevice(d.device), return false);
		SA_HIP_CHECK(hipDeviceSynchronize(), return false);
	}
	if (breakdown_ms) {
		const auto &s0 = job.devs[0].ctx->setup;
		breakdown_ms[SA_BREAKDOWN_ENCODE] = s0.encode;
		breakdown_ms[SA_BREAKDOWN_DEVICE] = s0.device;
		breakdown_ms[SA_BREAKDOWN_UPLOAD] = s0.upload;
		breakdown_ms[SA_BREAKDOWN_CODE_OBJECTS] = s0.code_objects;
		breakdown_ms[SA_BREAKDOWN_PLAN] = s0.plan;
		breakdown_ms[SA_BREAKDOWN_ARRANGE] = s0.arrange;
	}

	/* ---- the phase ----------------------------------------------------------------------------------------------- */
	const auto t_phase = std::chrono::steady_clock::now();
	for (int k = 0; k < ndev; k++) {
		Dev &d = job.devs[(size_t)k];
		if (sa_ctx_align_share(d.ctx, 0, pairs, ndev, k, d.d_share, elem16, host_packed, d.stream))
			return false;
	}
	SA_NCCL_CHECK(R.GroupStart(), return false);
	for (int k = 0; k < ndev; k++) {
		Dev &d = job.devs[(size_t)k];
		/* (bytes: the collective does not care what the elements are, and RCCL has no 16-bit integer type) */
		SA_NCCL_CHECK(R.AllGather(d.d_share, d.d_gathered, esz * (size_t)share_elems, ncclInt8, d.comm, d.stream), (void)R.GroupEnd(); return false);
	}
	SA_NCCL_CHECK(R.GroupEnd(), return false);
	for (int k = 0; k < ndev; k++) {
		Dev &d = job.devs[(size_t)k];
		if (sa_ctx_place_shares(d.ctx, 0, pairs, ndev, host_packed != nullptr, d.d_gathered, elem16, d.d_packed, d.stream))
			return false;
		SA_HIP_CHECK(hipSetDevice(d.device), return false);
		if (full) {
			SA_HIP_CHECK(sa_launch_expand_shell(d.d_packed, 0, d.d_full, in.num, (int32_t)d.ja, (int32_t)d.jb, d.stream), return false);
			SA_HIP_CHECK(hipEventRecord(d.placed, d.stream), return false);
			SA_HIP_CHECK(hipStreamWaitEvent(d.copy, d.placed, 0), return false);
			const int64_t ja = d.ja, jb = d.jb;
			if (jb > ja) {
				SA_HIP_CHECK(hipMemcpy2DAsync(out.matrix + (size_t)ja * dim, dim * sizeof(int32_t), d.d_full + (size_t)ja * dim,
							      dim * sizeof(int32_t), (size_t)jb * sizeof(int32_t), (size_t)(jb - ja),
							      hipMemcpyDeviceToHost, d.copy), return false);
				if (ja > 0) {
					SA_HIP_CHECK(hipMemcpy2DAsync(out.matrix + (size_t)ja, dim * sizeof(int32_t), d.d_full + (size_t)ja,
								      dim * sizeof(int32_t), (size_t)(jb - ja) * sizeof(int32_t), (size_t)ja,
								      hipMemcpyDeviceToHost, d.copy), return false);
				}
			}
		} else if (want_host && !host_packed) { /* packed destination the kernels could not store into: 1/ndev each */
			const int64_t lo = pairs * k / ndev, hi = pairs * (k + 1) / ndev;
			SA_HIP_CHECK(hipEventRecord(d.placed, d.stream), return false);
			SA_HIP_CHECK(hipStreamWaitEvent(d.copy, d.placed, 0), return false);
			if (hi > lo) {
				SA_HIP_CHECK(hipMemcpyAsync(out.matrix + lo, d.d_packed + lo, sizeof(int32_t) * (size_t)(hi - lo), hipMemcpyDeviceToHost, d.copy),
					     return false);
			}
		}
	}
	for (auto &d : job.devs) {
		SA_HIP_CHECK(hipSetDevice(d.device), return false);
		SA_HIP_CHECK(hipStreamSynchronize(d.stream), return false);
		SA_HIP_CHECK(hipStreamSynchronize(d.copy), return false);
	}
	const double phase = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_phase).count();
	sa_report_progress(1.0);
	if (phase_seconds)
		*phase_seconds = phase;
	if (breakdown_ms) {
		breakdown_ms[SA_BREAKDOWN_PHASE] = phase * 1e3;
		breakdown_ms[SA_BREAKDOWN_TOTAL] = sa_ms_since(t_all);
	}
	return true;
}
