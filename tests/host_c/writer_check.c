/* tests/host_c/writer_check.c -- TEST HARNESS: cli/sa_host.c (parsers, HDF5 writer) linked into a plain executable that
 * tests/test_cli_host.py builds with gcc -fsanitize=address,undefined.  Writes a random N x N similarity matrix (packed
 * triangle in, level z) twice -- tiles deflated segment by segment on all cores, and through libhdf5's own filter
 * (SA_HOST_SERIAL_DEFLATE) -- for the test to h5diff.   writer_check <N> <z> <out_parallel.h5> <out_serial.h5> */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../cli/sa_host.h"

int main(int argc, char **argv)
{
	if (argc < 5)
		return 2;
	const size_t n = (size_t)atol(argv[1]);
	const unsigned z = (unsigned)atoi(argv[2]);
	struct sa_host_store st;
	memset(&st, 0, sizeof(st));
	st.in.num = (int32_t)n;
	st.in.max = 4;
	st.blob_bytes = 5 * n;
	st.in.seqs = malloc(5 * n);
	st.in.meta = malloc(sizeof(*st.in.meta) * n);
	for (size_t k = 0; k < n; k++) {
		memcpy(st.in.seqs + 5 * k, "ARND", 5);
		st.in.meta[k].off = (int32_t)(5 * k);
		st.in.meta[k].len = 4;
	}
	const size_t pairs = n * (n - 1) / 2;
	int32_t *tri = malloc(sizeof(int32_t) * (pairs ? pairs : 1));
	unsigned long long x = 88172645463325252ull;
	for (size_t p = 0; p < pairs; p++) { /* xorshift: scores in [-300, 80) */
		x ^= x << 13, x ^= x >> 7, x ^= x << 17;
		tri[p] = (int32_t)(x % 380) - 300;
	}
	if (sa_host_write_hdf5(argv[3], &st, tri, true, z)) {
		fprintf(stderr, "%s\n", sa_host_error());
		return 1;
	}
	setenv("SA_HOST_SERIAL_DEFLATE", "1", 1);
	if (sa_host_write_hdf5(argv[4], &st, tri, true, z)) {
		fprintf(stderr, "%s\n", sa_host_error());
		return 1;
	}
	free(tri);
	free(st.in.meta);
	free(st.in.seqs);
	return 0;
}
