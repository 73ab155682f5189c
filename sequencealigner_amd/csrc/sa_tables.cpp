/* sa_tables.cpp -- host-only option tables behind the C ABI: substitution matrices,
 * method registry, last-error string.  No device code, no HIP calls. */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <strings.h>

#include <cstdarg>
#include "sa_shapes.h"
#include "sa_matrix_tables.inc"

static thread_local char g_err[1024] = "";

void sa_set_error(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	if (getenv("SA_HIP_VERBOSE"))
		fprintf(stderr, "seqalign_hip: %s\n", g_err);
}

extern "C" const char *sa_last_error(void) { return g_err; }
extern "C" int sa_abi_version(void) { return SA_ABI_VERSION; }

/* ---- matrices: replaces parse_matrix / list_matrices (reference src/bio/matrices.c:27-58) ---- */
extern "C" int sa_matrix_count(void) { return SA_MATRIX_COUNT; }

extern "C" const char *sa_matrix_name(int index)
{
	return (index < 0 || index >= SA_MATRIX_COUNT) ? nullptr : SA_MATRIX_INDEX[index].name;
}

extern "C" int sa_matrix_is_nucleotide(int index)
{
	return (index < 0 || index >= SA_MATRIX_COUNT) ? -1 : SA_MATRIX_INDEX[index].family;
}

extern "C" int sa_matrix_load(const char *name, int32_t lut[SA_LUT_SIZE], int32_t sub[SA_SUB_DIM * SA_SUB_DIM])
{
	if (!name || !lut || !sub) {
		sa_set_error("sa_matrix_load: null argument");
		return 1;
	}
	for (int k = 0; k < SA_MATRIX_COUNT; k++) {
		if (strcasecmp(name, SA_MATRIX_INDEX[k].name) != 0)
			continue;
		const char *alpha = SA_ALPHABETS[SA_MATRIX_INDEX[k].family];
		const int dim = SA_MATRIX_INDEX[k].dim;
		const int8_t *m = SA_MATRIX_BLOB + SA_MATRIX_INDEX[k].offset;
		for (int c = 0; c < SA_LUT_SIZE; c++)
			lut[c] = -1;
		for (int a = 0; alpha[a]; a++)
			lut[(unsigned char)alpha[a]] = a;
		memset(sub, 0, sizeof(int32_t) * SA_SUB_DIM * SA_SUB_DIM);
		for (int a = 0; a < dim; a++)
			for (int b = 0; b < dim; b++)
				sub[SA_SUB_DIM * a + b] = m[dim * a + b];
		return 0;
	}
	sa_set_error("Invalid substitution matrix name: %s", name);
	return 1;
}

/* ---- methods: replaces the `aligns` registry + parse_align (reference src/bio/align.c:87-96) ---- */
static const struct {
	const char *long_name, *short_name;
	int gap;
} SA_METHODS[SA_METHOD_COUNT] = {
	{ "Needleman-Wunsch", "nw", SA_GAP_LINEAR },
	{ "Gotoh", "ga", SA_GAP_AFFINE },
	{ "Smith-Waterman", "sw", SA_GAP_AFFINE },
};

extern "C" int sa_method_parse(const char *alias)
{
	if (!alias)
		return -1;
	for (int m = 0; m < SA_METHOD_COUNT; m++)
		if (!strcasecmp(alias, SA_METHODS[m].long_name) || !strcasecmp(alias, SA_METHODS[m].short_name))
			return m;
	sa_set_error("Invalid alignment method: %s", alias);
	return -1;
}

extern "C" const char *sa_method_name(int method)
{
	return (method < 0 || method >= SA_METHOD_COUNT) ? nullptr : SA_METHODS[method].long_name;
}

extern "C" int sa_method_gap_kind(int method)
{
	return (method < 0 || method >= SA_METHOD_COUNT) ? -1 : SA_METHODS[method].gap;
}
