/* tests/host_c/chunk_read.c -- the chunks of /similarity_matrix as they are stored (H5Dread_chunk: no filter pipeline, no
 * hyperslab machinery): `chunk_read file.h5 <tile row> <tiles> out.bin` writes, for the tiles (row, 0) .. (row, tiles - 1),
 * a u64 length and the stored bytes -- a zlib stream when the dataset is deflated, the raw tile otherwise.  The tests
 * inflate them themselves (h5dump re-inflates every 64 MB chunk of a tile row once per row of a hyperslab). */
#include <hdf5.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int main(int argc, char **argv)
{
	if (argc != 5)
		return 2;
	const hsize_t row = (hsize_t)atoll(argv[2]), tiles = (hsize_t)atoll(argv[3]);
	hid_t file = H5Fopen(argv[1], H5F_ACC_RDONLY, H5P_DEFAULT);
	hid_t set = file < 0 ? -1 : H5Dopen2(file, "/similarity_matrix", H5P_DEFAULT);
	if (set < 0)
		return 1;
	hid_t plist = H5Dget_create_plist(set);
	hsize_t cd[2] = { 0, 0 };
	if (H5Pget_chunk(plist, 2, cd) != 2)
		return 1;
	FILE *out = fopen(argv[4], "wb");
	if (!out)
		return 1;
	for (hsize_t c = 0; c < tiles; c++) {
		hsize_t at[2] = { row * cd[0], c * cd[1] }, bytes = 0;
		uint32_t mask = 0;
		if (H5Dget_chunk_storage_size(set, at, &bytes) < 0 || !bytes)
			return 1;
		void *buf = malloc(bytes);
		if (!buf || H5Dread_chunk(set, H5P_DEFAULT, at, &mask, buf) < 0 || mask != 0)
			return 1;
		const uint64_t n = bytes;
		fwrite(&n, sizeof(n), 1, out);
		fwrite(buf, 1, bytes, out);
		free(buf);
	}
	fclose(out);
	printf("%llu %llu\n", (unsigned long long)cd[0], (unsigned long long)cd[1]);
	H5Pclose(plist);
	H5Dclose(set);
	H5Fclose(file);
	return 0;
}
