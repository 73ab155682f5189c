/* tests/host_c/deflate_core_test.cpp -- the serial core of the device-side DEFLATE encoder
 * (sequencealigner_amd/csrc/sa_deflate_core.h) run on the host: a tile of int32 elements is cut into segments exactly as
 * the kernel cuts it, every segment goes through the SAME functions (match choice, histograms, code lengths, canonical
 * codes, block header, element bits, segment end), the pieces are stitched into one zlib stream and written out.  The
 * Python test inflates it with zlib and compares with the input; this program itself checks the Kraft sums.
 *
 *   deflate_core_test <in.i32> <out.zz> <segment elements> [group]  encode a file of little-endian int32; group = segments
 *                                                                   that share one set of codes (default 1)
 *   deflate_core_test --kraft                                    length limiting on adversarial histograms
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../sequencealigner_amd/csrc/sa_deflate_core.h"

static int kraft_ok(const uint8_t *len, int n, int maxbits, bool complete)
{
	uint64_t sum = 0;
	for (int s = 0; s < n; s++) {
		if (len[s] > maxbits)
			return 0;
		if (len[s])
			sum += 1ull << (maxbits - len[s]);
	}
	return complete ? sum == 1ull << maxbits : sum <= 1ull << maxbits;
}

static int kraft_cases()
{
	SaZWork W;
	/* Fibonacci weights: the deepest tree there is; powers of two; one heavy symbol; all equal; two symbols */
	for (int shape = 0; shape < 6; shape++) {
		for (int n : { 2, 3, 19, 30, 60, 286 }) {
			uint32_t freq[288] = {};
			uint64_t a = 1, b = 1;
			for (int s = 0; s < n; s++) {
				switch (shape) {
				case 0: freq[s] = (uint32_t)(a > 60000 ? 60000 + s : a); { const uint64_t c = a + b; a = b; b = c; } break;
				case 1: freq[s] = 1u << (s % 17); break;
				case 2: freq[s] = s == 0 ? 65000 : 1; break;
				case 3: freq[s] = 7; break;
				case 4: freq[s] = (uint32_t)((s * 2654435761u) % 65536u) + 1; break;
				default: freq[s] = s % 3 == 0 ? 0 : s + 1; break;
				}
			}
			for (int maxbits : { 7, 15 }) {
				if (maxbits == 7 && n > 19)
					continue;
				uint8_t len[288];
				uint32_t code[288];
				sa_z_alphabet(W, freq, n, maxbits, len, code, 0, false);
				if (!kraft_ok(len, n, maxbits, true)) {
					fprintf(stderr, "Kraft sum wrong: shape %d n %d maxbits %d\n", shape, n, maxbits);
					return 1;
				}
				for (int s = 0; s < n; s++)
					if ((freq[s] != 0) != (len[s] != 0)) {
						fprintf(stderr, "symbol %d: weight %u length %d (shape %d n %d)\n", s, freq[s], len[s], shape, n);
						return 1;
					}
			}
		}
	}
	printf("kraft ok\n");
	return 0;
}

int main(int argc, char **argv)
{
	if (argc == 2 && !strcmp(argv[1], "--kraft"))
		return kraft_cases();
	if (argc != 4 && argc != 5) {
		fprintf(stderr, "usage: %s in.i32 out.zz segment_elements [segments per code group] | --kraft\n", argv[0]);
		return 2;
	}
	FILE *f = fopen(argv[1], "rb");
	if (!f)
		return 2;
	fseek(f, 0, SEEK_END);
	const long bytes = ftell(f);
	fseek(f, 0, SEEK_SET);
	std::vector<uint32_t> e((size_t)bytes / 4);
	if (fread(e.data(), 4, e.size(), f) != e.size())
		return 2;
	fclose(f);
	const size_t seg = (size_t)atol(argv[3]);
	const size_t group = argc > 4 ? (size_t)atol(argv[4]) : 1; /* segments that share one set of codes (the kernel: ZGROUP) */
	std::vector<uint8_t> out;
	out.push_back(0x78);
	out.push_back(0x9c);
	uint32_t a = 1, b = 0;
	SaZWork W;
	for (size_t g0 = 0; g0 < e.size(); g0 += seg * group) {
		const size_t gend = g0 + seg * group < e.size() ? g0 + seg * group : e.size();
		/* histograms of the group's segments (matches never reach across a segment's start) */
		memset(&W, 0, sizeof(W));
		for (size_t s0 = g0; s0 < gend; s0 += seg) {
			const size_t n = s0 + seg <= gend ? seg : gend - s0;
			const uint32_t *el = e.data() + s0;
			for (size_t k = 0; k < n; k++) {
				const uint32_t v = el[k];
				const int j = sa_z_match(el, (uint32_t)k);
				W.lfreq[v & 255]++;
				if (j) {
					W.lfreq[SA_Z_LEN3]++;
					W.dfreq[sa_z_dcode(j)]++;
				} else {
					for (int t = 1; t < 4; t++)
						W.lfreq[(v >> (8 * t)) & 255]++;
				}
			}
			W.lfreq[SA_Z_EOB]++;
		}
		sa_z_alphabet(W, W.lfreq, SA_Z_NLIT, 15, W.llen, W.lcode, 0, false);
		sa_z_alphabet(W, W.dfreq, SA_Z_NDIST, 15, W.dlen, W.dcode, 0, false);
		if (!kraft_ok(W.llen, SA_Z_NLIT, 15, true) || !kraft_ok(W.dlen, SA_Z_NDIST, 15, true)) {
			fprintf(stderr, "group at %zu: incomplete code\n", g0);
			return 1;
		}
		std::vector<uint32_t> header(256, 0u);
		SaZBits hb{ header.data(), 0 };
		sa_z_header(W, hb, false);
		if (!kraft_ok(W.clen, SA_Z_NCL, 7, true) || hb.pos > SA_Z_HEADER_BITS) {
			fprintf(stderr, "group at %zu: bad block header (%u bits)\n", g0, hb.pos);
			return 1;
		}
		for (size_t s0 = g0; s0 < gend; s0 += seg) {
			const size_t n = s0 + seg <= gend ? seg : gend - s0;
			const uint32_t *el = e.data() + s0;
			uint64_t s1 = 0, s2 = 0;
			const uint64_t len = 4 * (uint64_t)n;
			std::vector<uint32_t> words(2 * n + 1024, 0u);
			for (size_t k = 0; k < header.size(); k++)
				words[k] = header[k];
			SaZBits bw{ words.data(), hb.pos };
			for (size_t k = 0; k < n; k++) {
				uint64_t bits;
				const uint32_t nb = sa_z_element(W.lcode, W.dcode, el[k], sa_z_match(el, (uint32_t)k), &bits);
				if (nb > SA_Z_ELEM_BITS)
					return 1;
				sa_z_put(bw, (uint32_t)bits, nb > 32 ? 32 : nb);
				if (nb > 32)
					sa_z_put(bw, (uint32_t)(bits >> 32), nb - 32);
				for (int t = 0; t < 4; t++) {
					const uint64_t byte = (el[k] >> (8 * t)) & 255;
					s1 += byte;
					s2 += (len - (4 * k + t)) * byte;
				}
			}
			const uint32_t nbytes = sa_z_finish_segment(W.lcode, bw);
			const uint8_t *p = reinterpret_cast<const uint8_t *>(words.data());
			out.insert(out.end(), p, p + nbytes);
			sa_z_adler_append(a, b, (uint32_t)(s1 % 65521u), (uint32_t)(s2 % 65521u), len);
		}
	}
	const uint8_t fin[5] = { 0x01, 0x00, 0x00, 0xff, 0xff };
	out.insert(out.end(), fin, fin + 5);
	const uint32_t adler = b << 16 | a;
	for (int t = 3; t >= 0; t--)
		out.push_back((uint8_t)(adler >> (8 * t)));
	f = fopen(argv[2], "wb");
	if (!f || fwrite(out.data(), 1, out.size(), f) != out.size())
		return 2;
	fclose(f);
	printf("%ld -> %zu bytes (%.3f : 1)\n", bytes, out.size(), (double)bytes / (double)out.size());
	return 0;
}
