/* sa_deflate_core.h -- the serial pieces of the device-side DEFLATE encoder (RFC 1951, dynamic-Huffman blocks; the `-z`
 * option, reference src/io/format/hdf5.c:91-95: H5Pset_deflate on the chunked /similarity_matrix).  Pure functions over
 * plain arrays: the kernel (sa_deflate.hip) runs them in ONE thread of a workgroup on LDS arrays, the host harness
 * (tests/host_c/deflate_core_test.cpp) runs the same code single-threaded and inflates the result with zlib -- the
 * format logic is tested without a device.
 *
 * What is encoded.  A tile of the matrix is int32 little endian scores: a low byte that looks random and three high
 * bytes that repeat the sign (00 00 00 or ff ff ff, or the high part of a neighbour's value).  A literal-only Huffman
 * coder cannot get below one bit for each of those three bytes (2.6 : 1 on NW / BLOSUM62 scores, zlib -6: 3.1); what
 * LZ77 has to offer for them is a length-3 match onto the high bytes of an EARLIER element with the same high part.  So
 * every element is
 *     literal(low byte)  +  match(length 3, distance 4 j)     j = 1..8: the nearest earlier element with equal bytes 1..3
 *     literal(low byte)  +  three literals                    when none of the eight has them
 * -- a fixed parse (no hash chains, no lazy matching: every element decides by itself from its eight predecessors, which
 * is what makes it a data-parallel kernel) under Huffman codes built for each segment: 2.87 : 1 on the same scores, between
 * zlib -1 (2.78) and -6 (3.13).  Segments of a tile are independent dynamic blocks that end byte-aligned on an empty stored
 * block (the "sync flush" marker 00 00 ff ff), so that workgroups write them side by side; a tile's stream is the zlib
 * header, its segments, a final empty stored block and the Adler-32 of the raw tile.  Any inflate reads it. */
#ifndef SA_DEFLATE_CORE_H
#define SA_DEFLATE_CORE_H

#include <cstdint>

#include "sa_shapes.h"

enum : int {
	SA_Z_NLIT = 286,  /* literal / length alphabet                                   */
	SA_Z_NDIST = 30,  /* distance alphabet                                           */
	SA_Z_NCL = 19,    /* code-length alphabet of the block header                    */
	SA_Z_EOB = 256,   /* end of block                                                */
	SA_Z_LEN3 = 257,  /* match length 3 (no extra bits)                              */
	SA_Z_J = 8,       /* elements a match may reach back (distance 4 .. 32 bytes)    */
	SA_Z_ELEM_BITS = 63, /* an element never takes more: 4 x 15, or 15 + 15 + 15 + 3   */
	SA_Z_HEADER_BITS = 4544 /* a block header never takes more: 17 + 19 x 3 + 316 x (7 + 7) */
};

/* distance 4 j, j = 1..8: code (RFC 1951 3.2.5), number of extra bits, their value -- a nibble per j */
SA_HD inline uint32_t sa_z_dcode(int j) { return (0x99887653u >> (4 * (j - 1))) & 15u; }
SA_HD inline uint32_t sa_z_dext_bits(int j) { return (0x33332210u >> (4 * (j - 1))) & 15u; }
SA_HD inline uint32_t sa_z_dext_val(int j) { return (0x73733310u >> (4 * (j - 1))) & 15u; }

/* everything the serial path touches, one per workgroup (LDS) */
struct SaZWork {
	uint32_t lfreq[288], dfreq[32], cfreq[20];
	uint32_t lcode[288], dcode[32], ccode[20]; /* bit-reversed code | length << 16 (0: unused symbol) */
	uint8_t llen[288], dlen[32], clen[20];
	uint16_t order[288];   /* used symbols by ascending weight                      */
	uint32_t w[576];       /* tree weights, then depths                             */
	uint16_t par[576];
	uint16_t bl[16], next[16];
	uint8_t seq[320], seqx[320]; /* header: code-length symbols and their extra bits */
	uint32_t nseq;
	uint32_t used;              /* literal alphabet: symbols in use (kernel: after the parallel sort) */
};

/* bit writer of the one thread that writes the block header and the block's end (LSB first; the words are zero before) */
struct SaZBits {
	uint32_t *w;
	uint32_t pos;
};
SA_HD inline void sa_z_put(SaZBits &b, uint32_t v, uint32_t n)
{
	const uint64_t x = (uint64_t)v << (b.pos & 31u);
	b.w[b.pos >> 5] |= (uint32_t)x;
	if ((uint32_t)(x >> 32))
		b.w[(b.pos >> 5) + 1] |= (uint32_t)(x >> 32);
	b.pos += n;
}

/* used symbols of freq[0..n) by (weight, symbol) ascending -> order[]; returns their number.  Insertion sort: for the
 * small alphabets (distances, code lengths) and the host harness; the kernel sorts the literal alphabet in parallel */
SA_HD inline int sa_z_sort_small(const uint32_t *freq, int n, uint16_t *order)
{
	int used = 0;
	for (int s = 0; s < n; s++) {
		if (!freq[s])
			continue;
		int at = used++;
		while (at > 0 && freq[order[at - 1]] > freq[s]) {
			order[at] = order[at - 1];
			at--;
		}
		order[at] = (uint16_t)s;
	}
	return used;
}

/* Code lengths (<= maxbits) of a Huffman code over order[0..used) (ascending weight, used >= 2); len[] of every other
 * symbol is left alone.  Two queues (the sorted leaves, the internal nodes in the order they are made: both ascending),
 * depths from the root down; leaves deeper than maxbits are counted at maxbits, which oversubscribes the code by a whole
 * number of units of 2^-maxbits, and the count of codes per length is repaired one unit at a time (below); lengths are
 * then dealt out again, longest to the lightest symbol (Kraft sum exactly 1, checked by the harness). */
SA_HD inline void sa_z_lengths(const uint32_t *freq, const uint16_t *order, int used, int maxbits, uint8_t *len, uint32_t *w,
			       uint16_t *par, uint16_t *bl)
{
	for (int i = 0; i < used; i++)
		w[i] = freq[order[i]];
	int leaf = 0, inner = used, made = used;
	for (int k = 0; k + 1 < used; k++) {
		int pick[2];
		for (int t = 0; t < 2; t++)
			pick[t] = leaf < used && (inner >= made || w[leaf] <= w[inner]) ? leaf++ : inner++;
		w[made] = w[pick[0]] + w[pick[1]];
		par[pick[0]] = par[pick[1]] = (uint16_t)made;
		made++;
	}
	w[made - 1] = 0; /* from here on w[] of an internal node is its depth */
	for (int k = made - 2; k >= used; k--)
		w[k] = w[par[k]] + 1;
	for (int b = 0; b <= maxbits; b++)
		bl[b] = 0;
	uint32_t kraft = 0; /* in units of 2^-maxbits */
	for (int i = 0; i < used; i++) {
		int d = (int)w[par[i]] + 1;
		if (d > maxbits)
			d = maxbits;
		bl[d]++;
		kraft += 1u << (maxbits - d);
	}
	/* one step: the deepest leaf above the limit goes one level down (- 2^(maxbits-b-1)) and a leaf of the last level
	 * comes up beside it (+ 2^(maxbits-b-1) - 1): one unit less */
	for (uint32_t over = kraft - (1u << maxbits); over > 0; over--) {
		int b = maxbits - 1;
		while (bl[b] == 0)
			b--;
		bl[b]--;
		bl[b + 1] += 2;
		bl[maxbits]--;
	}
	int at = 0;
	for (int b = maxbits; b >= 1; b--)
		for (int c = 0; c < bl[b]; c++)
			len[order[at++]] = (uint8_t)b;
}

SA_HD inline uint32_t sa_z_reverse(uint32_t c, int n)
{
	uint32_t r = 0;
	for (int b = 0; b < n; b++)
		r |= ((c >> b) & 1u) << (n - 1 - b);
	return r;
}

/* canonical codes (RFC 1951 3.2.2) of len[0..n), bit-reversed for an LSB-first writer: code[s] = reversed | len << 16 */
SA_HD inline void sa_z_codes(const uint8_t *len, int n, uint32_t *code, uint16_t *bl, uint16_t *next)
{
	for (int b = 0; b < 16; b++)
		bl[b] = 0;
	for (int s = 0; s < n; s++)
		bl[len[s]]++;
	bl[0] = 0;
	uint32_t c = 0;
	for (int b = 1; b < 16; b++) {
		c = (c + bl[b - 1]) << 1;
		next[b] = (uint16_t)c;
	}
	for (int s = 0; s < n; s++) {
		const int l = len[s];
		code[s] = l ? sa_z_reverse(next[l]++, l) | (uint32_t)l << 16 : 0u;
	}
}

/* lengths + codes of one alphabet from its histogram.  inflate wants every code complete except a lone code of one bit
 * (and the code-length code complete always): an alphabet with fewer than two used symbols gets a second symbol, so
 * that both have one bit. */
SA_HD inline void sa_z_alphabet(SaZWork &W, uint32_t *freq, int n, int maxbits, uint8_t *len, uint32_t *code, int used,
				bool sorted)
{
	for (int s = 0; s < n; s++)
		len[s] = 0;
	if (!sorted)
		used = sa_z_sort_small(freq, n, W.order);
	if (used < 2) {
		const int have = used ? W.order[0] : -1;
		const int extra = have == 0 ? 1 : 0;
		freq[extra] = 1; /* (never coded: costs nothing) */
		if (!used)
			freq[1] = 1;
		used = sa_z_sort_small(freq, n, W.order);
	}
	sa_z_lengths(freq, W.order, used, maxbits, len, W.w, W.par, W.bl);
	sa_z_codes(len, n, code, W.bl, W.next);
}

/* The header of a dynamic block (RFC 1951 3.2.7) for W.llen / W.dlen: the two length tables as one run-length coded
 * sequence (16: repeat the previous length 3..6 times, 17 / 18: 3..10 / 11..138 zeros), Huffman-coded itself.
 * Returns the bits the block's symbols will take (for the caller's size check), header excluded. */
SA_HD inline void sa_z_header(SaZWork &W, SaZBits &b, bool final_block)
{
	int hlit = SA_Z_NLIT, hdist = SA_Z_NDIST;
	while (hlit > 257 && !W.llen[hlit - 1])
		hlit--;
	while (hdist > 1 && !W.dlen[hdist - 1])
		hdist--;
	const int total = hlit + hdist;
	auto at = [&](int k) -> int { return k < hlit ? W.llen[k] : W.dlen[k - hlit]; };
	for (int s = 0; s < SA_Z_NCL; s++)
		W.cfreq[s] = 0;
	uint32_t ns = 0;
	for (int k = 0; k < total;) {
		const int l = at(k);
		int run = 1;
		while (k + run < total && at(k + run) == l)
			run++;
		if (l == 0 && run >= 3) {
			const int take = run > 138 ? 138 : run;
			W.seq[ns] = take >= 11 ? 18 : 17;
			W.seqx[ns] = (uint8_t)(take >= 11 ? take - 11 : take - 3);
			k += take;
		} else if (l != 0 && run >= 4) { /* the length itself, then repeats of it */
			W.seq[ns] = (uint8_t)l;
			W.seqx[ns] = 0;
			W.cfreq[l]++;
			ns++;
			const int take = run - 1 > 6 ? 6 : run - 1;
			W.seq[ns] = 16;
			W.seqx[ns] = (uint8_t)(take - 3);
			k += 1 + take;
		} else {
			W.seq[ns] = (uint8_t)l;
			W.seqx[ns] = 0;
			k++;
		}
		W.cfreq[W.seq[ns]]++;
		ns++;
	}
	W.nseq = ns;
	sa_z_alphabet(W, W.cfreq, SA_Z_NCL, 7, W.clen, W.ccode, 0, false);
	const uint8_t perm[SA_Z_NCL] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
	int hclen = SA_Z_NCL;
	while (hclen > 4 && !W.clen[perm[hclen - 1]])
		hclen--;
	sa_z_put(b, final_block ? 1u : 0u, 1);
	sa_z_put(b, 2u, 2);
	sa_z_put(b, (uint32_t)(hlit - 257), 5);
	sa_z_put(b, (uint32_t)(hdist - 1), 5);
	sa_z_put(b, (uint32_t)(hclen - 4), 4);
	for (int k = 0; k < hclen; k++)
		sa_z_put(b, W.clen[perm[k]], 3);
	for (uint32_t k = 0; k < ns; k++) {
		const uint32_t s = W.seq[k], c = W.ccode[s];
		sa_z_put(b, c & 0xffffu, c >> 16);
		if (s >= 16)
			sa_z_put(b, W.seqx[k], s == 16 ? 2 : s == 17 ? 3 : 7);
	}
}

/* the end of a segment: end-of-block, then an empty stored block -- three header bits, padding to the byte, 00 00 ff ff.
 * Returns the segment's length in bytes. */
SA_HD inline uint32_t sa_z_finish_segment(const uint32_t *lcode, SaZBits &b)
{
	const uint32_t c = lcode[SA_Z_EOB];
	sa_z_put(b, c & 0xffffu, c >> 16);
	sa_z_put(b, 0u, 3);
	b.pos = (b.pos + 7u) & ~7u;
	sa_z_put(b, 0xffff0000u, 32);
	return b.pos >> 3;
}

/* which of the eight elements before k has the bytes 1..3 of v = e[k] (0: none); elements of the SEGMENT only */
SA_HD inline int sa_z_match(const uint32_t *e, uint32_t k)
{
	const uint32_t hi = e[k] >> 8;
	const int reach = k < (uint32_t)SA_Z_J ? (int)k : SA_Z_J;
	for (int j = 1; j <= reach; j++)
		if ((e[k - j] >> 8) == hi)
			return j;
	return 0;
}

/* the bits of one element (LSB first) under the segment's codes; returns their number (<= SA_Z_ELEM_BITS) */
SA_HD inline uint32_t sa_z_element(const uint32_t *lcode, const uint32_t *dcode, uint32_t v, int j, uint64_t *bits)
{
	uint32_t c = lcode[v & 255u];
	uint64_t acc = c & 0xffffu;
	uint32_t n = c >> 16;
	if (j) {
		c = lcode[SA_Z_LEN3];
		acc |= (uint64_t)(c & 0xffffu) << n;
		n += c >> 16;
		c = dcode[sa_z_dcode(j)];
		acc |= (uint64_t)(c & 0xffffu) << n;
		n += c >> 16;
		acc |= (uint64_t)sa_z_dext_val(j) << n;
		n += sa_z_dext_bits(j);
	} else {
		for (int t = 1; t < 4; t++) {
			c = lcode[(v >> (8 * t)) & 255u];
			acc |= (uint64_t)(c & 0xffffu) << n;
			n += c >> 16;
		}
	}
	*bits = acc;
	return n;
}

/* Adler-32 of a tile from its segments' sums: s1 = sum of the bytes, s2 = sum of (len - index) * byte, both mod 65521;
 * (a, b) = the running checksum (1, 0 at the start of the tile) */
SA_HD inline void sa_z_adler_append(uint32_t &a, uint32_t &b, uint32_t s1, uint32_t s2, uint64_t len)
{
	b = (uint32_t)((b + (len % 65521u) * a + s2) % 65521u);
	a = (a + s1) % 65521u;
}

#endif /* SA_DEFLATE_CORE_H */
