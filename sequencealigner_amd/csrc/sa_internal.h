/* sa_internal.h -- internal declarations shared by the translation units of libseqalign_hip.so */
#ifndef SA_INTERNAL_H
#define SA_INTERNAL_H

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdarg>
#include <string>
#include <vector>

#include "../../include/seqalign_hip.h"

/* residue codes of the encoded sequence store (device side):
 * 0..23 = index from sa_scoring.lut, SA_CODE_SEP = the NUL terminator of every sequence,
 * SA_CODE_NOP = pipeline bubble fed by the systolic kernels. */
enum : int { SA_CODE_SEP = 24, SA_CODE_NOP = 25, SA_CODE_ROWS = 26 };

void sa_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

/* `onfail` runs in the caller's scope (so `break`/`return`/`goto` act on the caller's loop). */
#define SA_HIP_CHECK(call, onfail)                                                  \
	if (hipError_t err__ = (call); err__ != hipSuccess) {                       \
		sa_set_error("%s failed: %s", #call, hipGetErrorString(err__));      \
		onfail;                                                              \
	} else                                                                      \
		(void)0

/* ---- kernel argument blocks --------------------------------------------- */
struct SaSeqStore {
	const uint8_t *codes;         /* encoded blob, same offsets as the host blob            */
	const struct sa_meta *meta;   /* N x {off,len}                                          */
	int32_t num;
};

struct SaGenericArgs {
	SaSeqStore st;
	const int32_t *sub;           /* s32[24*24] substitution matrix in HBM                  */
	int32_t gap_pen, gap_opn, gap_ext;
	int64_t start, count;         /* packed pair range                                      */
	int32_t *out;                 /* out[q] = score of pair start+q                         */
	int32_t *scratch;             /* per-wave strip boundary columns (M and X)              */
	int64_t scratch_stride;       /* ints per wave = 2*(max_len+2)                          */
};

/* ---- launchers implemented in the .hip files ---------------------------- */
hipError_t sa_launch_generic(int method, const SaGenericArgs &a, int blocks, hipStream_t s);
const char *sa_generic_kernel_name(int method);

hipError_t sa_launch_expand_full(const int32_t *packed, int32_t *full, int32_t num, hipStream_t s);

#endif /* SA_INTERNAL_H */
