/*
 * tests/adapter/seqalign_hip_adapter.c -- the reference-side binding of libseqalign_hip.so, COMPILED.
 *
 * This is the translation unit INTEGRATION.md §1 shows: what a maintainer of jakovdev/SequenceAligner adds as
 * src/interface/seqalign_hip.c and builds INSTEAD OF src/interface/seqalign_cuda.c.  It exports the reference's two
 * device entry points (src/interface/seqalign_cuda.h:7-9) on top of include/seqalign_hip.h and owns the -C option
 * object that other reference TUs order against (src/system/os.c:480, src/io/output.c:108).
 *
 * oracle/Makefile (`make _ref_hip`) compiles it against the reference's own headers where they lie and links it with
 * the reference's own main.c and every other reference TU -> oracle/_ref/seqalign_ref_hip: the reference CLI running
 * on the HIP library.  tests/test_adapter_build.py runs that binary (CPU: -h/-l; -m gpu: FASTA -> HDF5 vs goldens).
 * The static_asserts below are the layout proof the boundary rests on.
 */
#include "interface/seqalign_cuda.h" /* the reference's prototypes: cuda_memory, cuda_align */

#include <args.h>
#include <print.h>
#include <stddef.h>
#include <string.h>

#include "bio/align.h" /* GAP_PEN GAP_OPN GAP_EXT SEQ_LUT SUB_MAT ALIGN struct meta */
#include "io/input.h"
#include "io/output.h"
#include "util/benchmark.h"

#include <seqalign_hip.h> /* this repository: include/seqalign_hip.h */

/* ---- layout compatibility: struct input/output/meta are passed BY VALUE across the boundary ---- */
#define SAME_MEMBER(ref_t, hip_t, m)                                                        \
	static_assert(offsetof(ref_t, m) == offsetof(hip_t, m), #m ": offset differs");     \
	static_assert(sizeof(((ref_t *)0)->m) == sizeof(((hip_t *)0)->m), #m ": size differs")
static_assert(sizeof(struct meta) == sizeof(struct sa_meta), "struct meta");
SAME_MEMBER(struct meta, struct sa_meta, off);
SAME_MEMBER(struct meta, struct sa_meta, len);
static_assert(sizeof(struct input) == sizeof(struct sa_input), "struct input");
SAME_MEMBER(struct input, struct sa_input, seqs);
SAME_MEMBER(struct input, struct sa_input, meta);
SAME_MEMBER(struct input, struct sa_input, max);
SAME_MEMBER(struct input, struct sa_input, num);
static_assert(sizeof(struct output) == sizeof(struct sa_output), "struct output");
SAME_MEMBER(struct output, struct sa_output, matrix);
SAME_MEMBER(struct output, struct sa_output, seqs);
SAME_MEMBER(struct output, struct sa_output, dim);
SAME_MEMBER(struct output, struct sa_output, triangular);
static_assert(sizeof(SEQ_LUT) == sizeof(((struct sa_scoring *)0)->lut), "SEQ_LUT");
static_assert(sizeof(SUB_MAT) == sizeof(((struct sa_scoring *)0)->sub), "SUB_MAT");
static_assert(SCORE_MIN == SA_SCORE_MIN, "SCORE_MIN");
static_assert(SEQ_LUT_SIZE == SA_LUT_SIZE && SUB_MAT_DIM == SA_SUB_DIM, "table dimensions");

[[gnu::nonnull]]
bool align(struct input, struct output); /* the reference's CPU driver, src/bio/align.c:21 */

static bool no_cuda; /* -C: stay on the CPU path, exactly as in the reference */

static bool device_present(void)
{
	static bool asked;
	if (no_cuda || asked)
		return true;
	asked = true;
	if (sa_hip_device_count() > 0)
		return true;
	pwarn("No HIP devices available"); /* src/interface/seqalign_cuda.c:55-61 */
	if (!print_Yn("Would you like to switch to non-CUDA (CPU)?"))
		return false;
	no_cuda = true;
	return true;
}

bool cuda_memory(size_t bytes) /* src/interface/seqalign_cuda.c:71-93 */
{
	if (!device_present())
		return false;
	if (no_cuda)
		return true;
	return sa_hip_memory(bytes);
}

/* the reference's progress bar in its launch loop (src/interface/seqalign_cuda.c:181,286-289,293), fed by the library */
static void hip_progress(double fraction, void *user)
{
	(void)user;
	pproportc(fraction, "Aligning sequences");
}

bool cuda_align(struct input in, struct output out) /* src/interface/seqalign_cuda.c:95-296 */
{
	if (!device_present())
		return false;
	if (no_cuda)
		return align(in, out);

	struct sa_scoring sc = { .method = sa_method_parse(ALIGN->aliases[1]), /* "nw" | "ga" | "sw": bio/method/ *.c */
				 .gap_pen = GAP_PEN, .gap_opn = GAP_OPN, .gap_ext = GAP_EXT };
	memcpy(sc.lut, SEQ_LUT, sizeof(sc.lut)); /* s32[128]    bio/align.h:11-12 */
	memcpy(sc.sub, SUB_MAT, sizeof(sc.sub)); /* s32[24][24] bio/align.h:13-14 */
	if (sc.method < 0) {
		perr("Alignment method %s has no HIP kernels", ALIGN->aliases[0]);
		return false;
	}
	pinfo("Using HIP device: %s", sa_hip_device_name(0));

	const struct sa_input hin = { in.seqs, (struct sa_meta *)in.meta, in.max, in.num };
	const struct sa_output hout = { out.matrix, out.seqs, out.dim, out.triangular };
	ppercent(0, "Aligning sequences");
	sa_hip_set_progress(hip_progress, NULL);
	bench_align_start(); /* same bracket as the reference; sa_hip_last_align_seconds() is the loop alone */
	const bool ok = sa_hip_align(hin, hout, &sc);
	bench_align_end();
	sa_hip_set_progress(NULL, NULL);
	if (ok)
		ppercent(100, "Aligning sequences");
	if (!ok) {
		perr("%s", sa_last_error()); /* CALLR-style reporting, src/interface/seqalign_cuda.c:23-30 */
		return false;
	}
	bench_align_print();
	return true;
}

static void print_no_cuda(void)
{
	pinfom("HIP: Enabled");
}

ARG_EXTERN(compression);
ARG_EXTERN(threads);

ARGUMENT(disable_cuda) = {
	.opt = 'C',
	.lopt = "no-cuda",
	.help = "Disable the device path (HIP)",
	.set = &no_cuda,
	.action_callback = print_no_cuda,
	.action_phase = ARG_CALLBACK_IF_UNSET,
	.action_order = ARG_ORDER_AFTER(ARG(compression)),
	.help_order = ARG_ORDER_AFTER(ARG(threads)),
};
