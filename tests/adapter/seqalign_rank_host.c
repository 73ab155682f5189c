/*
 * tests/adapter/seqalign_rank_host.c -- INTEGRATION.md section 3 as a translation unit that compiles, links and runs:
 * the C calls ONE RANK of a one-process-per-GPU host makes, with RCCL (ncclAllGather) between them.
 *
 *   seqalign_rank_host <world> <rank> <id-file> <method> <matrix> <p|-> <s|-> <e|-> <in.fasta> <out.i32> [host]
 *
 * Any launcher starts `world` of these (rank r on GPU r of the node).  Rank 0 creates the ncclUniqueId and leaves it in
 * <id-file>; the others wait for the file.  Every rank parses the input (inputs are replicated: <= 12 MB), computes the
 * dense share of its tiles of the job-wide tile list, all-gathers the shares, places them -- after which EVERY GPU
 * holds the whole packed matrix in the reference's order (src/io/output.c:83) -- and rank 0 writes it to <out.i32>
 * (raw little-endian s32) for the test to compare with the reference's goldens.  With the trailing argument "host" the
 * ranks' kernels also deliver their own scores straight into one shared, page-locked host matrix (a file mapping
 * under /dev/shm named <out.i32>.shm that every rank attaches), which is then what rank 0 writes.
 *
 * The reference has no counterpart (single device, src/interface/seqalign_cuda.c:65); this sits between its
 * cuda_align and output_flush (src/main.c:31-34).  Test: tests/test_adapter_build.py (world 1 on the one-GPU box).
 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "../../cli/sa_host.h"
#include "../../include/seqalign_hip.h"

#define DIE(...)                              \
	do {                                  \
		fprintf(stderr, __VA_ARGS__); \
		fprintf(stderr, "\n");        \
		exit(1);                      \
	} while (0)
#define HIP(call)                                                                      \
	do {                                                                           \
		hipError_t e__ = (call);                                               \
		if (e__ != hipSuccess)                                                 \
			DIE("%s: %s", #call, hipGetErrorString(e__));                  \
	} while (0)
#define NCCL(call)                                                                     \
	do {                                                                           \
		ncclResult_t r__ = (call);                                             \
		if (r__ != ncclSuccess)                                                \
			DIE("%s: %s", #call, ncclGetErrorString(r__));                 \
	} while (0)
#define SA(call)                                                                       \
	do {                                                                           \
		if (call)                                                              \
			DIE("%s: %s", #call, sa_last_error());                         \
	} while (0)

static int32_t gap_arg(const char *s) { return strcmp(s, "-") ? -atoi(s) : 0; }

int main(int argc, char **argv)
{
	if (argc < 11)
		DIE("usage: %s world rank id-file method matrix p s e in.fasta out.i32 [host]", argv[0]);
	const int world = atoi(argv[1]), rank = atoi(argv[2]);
	const char *idfile = argv[3];
	const int to_host = argc > 11 && !strcmp(argv[11], "host");

	struct sa_scoring sc;
	memset(&sc, 0, sizeof(sc));
	sc.method = sa_method_parse(argv[4]);
	if (sc.method < 0 || sa_matrix_load(argv[5], sc.lut, sc.sub))
		DIE("%s", sa_last_error());
	sc.gap_pen = gap_arg(argv[6]);
	sc.gap_opn = gap_arg(argv[7]);
	sc.gap_ext = gap_arg(argv[8]);
	struct sa_host_store store;
	if (sa_host_load(argv[9], sc.lut, sc.gap_pen, -1, 0, &store)) /* (gap in its stored, negated form: src/io/input.c:15-19) */
		DIE("%s", sa_host_error());
	const int64_t pairs = (int64_t)store.in.num * (store.in.num - 1) / 2;

	/* ---- the communicator: one rank per GPU of the node ---- */
	int ndev = 0;
	HIP(hipGetDeviceCount(&ndev));
	const int device = rank % (ndev > 0 ? ndev : 1);
	HIP(hipSetDevice(device));
	ncclUniqueId id;
	if (rank == 0) {
		NCCL(ncclGetUniqueId(&id));
		char tmp[4096];
		snprintf(tmp, sizeof(tmp), "%s.tmp", idfile);
		FILE *f = fopen(tmp, "wb");
		if (!f || fwrite(&id, sizeof(id), 1, f) != 1)
			DIE("cannot write %s", tmp);
		fclose(f);
		rename(tmp, idfile);
	} else {
		FILE *f = NULL;
		for (int tries = 0; tries < 600 && !(f = fopen(idfile, "rb")); tries++)
			usleep(100000);
		if (!f || fread(&id, sizeof(id), 1, f) != 1)
			DIE("cannot read %s", idfile);
		fclose(f);
	}
	ncclComm_t comm;
	NCCL(ncclCommInitRank(&comm, world, id, rank));
	hipStream_t compute, comm_stream;
	HIP(hipStreamCreateWithFlags(&compute, hipStreamNonBlocking));
	HIP(hipStreamCreateWithFlags(&comm_stream, hipStreamNonBlocking));
	hipEvent_t computed;
	HIP(hipEventCreateWithFlags(&computed, hipEventDisableTiming));

	/* ---- the shared host matrix (optional) ---- */
	int32_t *host_matrix = NULL;
	char shm[4096];
	snprintf(shm, sizeof(shm), "%s.shm", argv[10]);
	if (to_host) {
		const int fd = open(shm, O_RDWR | O_CREAT, 0600);
		if (fd < 0 || ftruncate(fd, (off_t)(sizeof(int32_t) * (size_t)pairs)))
			DIE("cannot create %s", shm);
		host_matrix = mmap(NULL, sizeof(int32_t) * (size_t)pairs, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
		close(fd);
		if (host_matrix == MAP_FAILED)
			DIE("mmap %s", shm);
		SA(sa_hip_host_register(host_matrix, sizeof(int32_t) * (size_t)pairs));
	}

	/* ---- one super-chunk = the whole packed index ---- */
	const int64_t start = 0, count = pairs;
	sa_ctx *ctx = sa_ctx_create(device, store.in, &sc);                    /* inputs replicated: <= 12 MB            */
	if (!ctx)
		DIE("sa_ctx_create: %s", sa_last_error());
	const int elem16 = sa_ctx_scores_fit16(ctx);
	const size_t elem_bytes = elem16 ? sizeof(int16_t) : sizeof(int32_t);
	const int64_t e = sa_ctx_share_elems(ctx, start, count, world, to_host); /* elements of ONE rank's dense share     */
	if (e < 0)
		DIE("sa_ctx_share_elems: %s", sa_last_error());
	char *d_shares = NULL;                                                 /* world * e elements; mine at rank * e   */
	int32_t *d_packed = NULL;
	HIP(hipMalloc((void **)&d_shares, elem_bytes * (size_t)e * (size_t)world));
	HIP(hipMalloc((void **)&d_packed, sizeof(int32_t) * (size_t)pairs));
	SA(sa_ctx_align_share(ctx, start, count, world, rank, d_shares + elem_bytes * (size_t)e * (size_t)rank, elem16, host_matrix, compute));
	HIP(hipEventRecord(computed, compute));
	HIP(hipStreamWaitEvent(comm_stream, computed, 0));
	NCCL(ncclAllGather(d_shares + elem_bytes * (size_t)e * (size_t)rank, d_shares, elem_bytes * (size_t)e, ncclChar, comm, comm_stream)); /* in place */
	SA(sa_ctx_place_shares(ctx, start, count, world, to_host, d_shares, elem16, d_packed + start, comm_stream));
	HIP(hipStreamSynchronize(comm_stream));
	HIP(hipStreamSynchronize(compute));
	/* d_packed[p] = score of pair p on every GPU; host_matrix[p] written by the rank that computed p */

	if (rank == 0) {
		int32_t *result = host_matrix;
		if (!to_host) {
			result = malloc(sizeof(int32_t) * (size_t)pairs);
			if (!result)
				DIE("out of memory");
			HIP(hipMemcpy(result, d_packed, sizeof(int32_t) * (size_t)pairs, hipMemcpyDeviceToHost));
		} else if (world > 1) { /* the other ranks' stores: they synchronise before they leave; a real host would barrier here */
			usleep(200000);
		}
		FILE *f = fopen(argv[10], "wb");
		if (!f || fwrite(result, sizeof(int32_t), (size_t)pairs, f) != (size_t)pairs)
			DIE("cannot write %s", argv[10]);
		fclose(f);
		if (!to_host)
			free(result);
	}
	if (host_matrix) {
		sa_hip_host_unregister(host_matrix);
		munmap(host_matrix, sizeof(int32_t) * (size_t)pairs);
		if (rank == 0)
			unlink(shm);
	}
	sa_ctx_destroy(ctx);
	ncclCommDestroy(comm);
	sa_host_store_free(&store);
	printf("rank %d of %d on device %d: %lld pairs, share %lld x %zu bytes\n", rank, world, device, (long long)pairs, (long long)e, elem_bytes);
	return 0;
}
