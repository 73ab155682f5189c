/* cli/sa_host.c -- see sa_host.h.  Plain C11 + OpenMP + libhdf5. */
#define _GNU_SOURCE
#include "sa_host.h"

#include <ctype.h>
#include <errno.h>
#include <fcntl.h>
#include <limits.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <hdf5.h>
#include <zlib.h>

static _Thread_local char g_err[512];

static int fail(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	return 1;
}

const char *sa_host_error(void) { return g_err; }

/* ---- length limit: reference src/io/input.c:15-19, src/bio/align.h:23 -------------------- */
static bool length_ok(int64_t len, int32_t gap_stored)
{
	const int64_t seq_len_max = (INT32_MAX - 1) / 2;
	const int64_t gap = -(int64_t)gap_stored;
	return gap ? len <= seq_len_max / gap : len <= seq_len_max;
}

struct builder {
	uint8_t *w;       /* write cursor (compacts in place over the file image) */
	uint8_t *base;
	int32_t num, max;
	int64_t sum;
};

/* appends one residue-validated sequence taken from [p, p+n) skipping CR/LF/space(/quote) */
static int add_sequence(struct builder *b, const uint8_t *p, size_t n, bool skip_quote, const int32_t *lut,
			int32_t gap, int32_t index1)
{
	int32_t slen = 0;
	uint8_t *start = b->w;
	for (size_t k = 0; k < n; k++) {
		const unsigned c = (unsigned)toupper(p[k]);
		if (c == '\r' || c == '\n' || c == ' ' || (skip_quote && c == '"'))
			continue;
		if (c == 0 || c > SCHAR_MAX)
			return fail("Sequence #%d is corrupted", index1);
		if (lut[c] < 0)
			return fail("Sequence #%d is invalid", index1);
		*b->w++ = (uint8_t)c;
		slen++;
	}
	(void)start;
	if (!slen)
		return fail("Sequence #%d is empty", index1);
	if (!length_ok(slen, gap))
		return fail("Sequence #%d exceeds length limits", index1);
	if (b->sum + slen + 1 > INT32_MAX)
		return fail("Length overflow after %d sequences", index1);
	b->max = slen > b->max ? slen : b->max;
	b->sum += slen + 1;
	*b->w++ = 0;
	b->num++;
	return 0;
}

/* ---- FASTA (reference src/io/source/fasta.c:16-85) ---------------------------------------- */
static int parse_fasta(uint8_t *file, const uint8_t *fend, const int32_t *lut, int32_t gap, struct builder *b)
{
	const uint8_t *p = file;
	if (p >= fend || *p != '>')
		return fail("Data before first header");
	while (p < fend) {
		while (p < fend && *p != '\n' && *p != '\r') /* header line */
			p++;
		while (p < fend && (*p == '\n' || *p == '\r'))
			p++;
		if (p >= fend)
			return fail("Last header has no data");
		const uint8_t *body = p;
		while (p < fend && *p != '>')
			p++;
		/* the body is consumed before the write cursor can reach it: w <= body always */
		if (add_sequence(b, body, (size_t)(p - body), false, lut, gap, b->num + 1))
			return 1;
	}
	return 0;
}

/* ---- DSV (behaviour of reference src/io/source/dsv.c:26-230) --------------------------------
 * Grammar:  record = field *( DELIM field ) EOL        EOL = CR | LF | end of input
 *           field  = *( bare-char | quoted-run )       quoted-run = '"' *( char | '""' ) '"'
 * A delimiter or line break inside a quoted run is data.  A field that begins and ends with a quote is
 * delivered without that outer pair (inner "" stay as they are -- residues never contain quotes, and the
 * header match is on whole names).  One scanner serves the header walk, the column count and the rows. */
enum dsv_stop { DSV_AT_DELIM, DSV_AT_EOL };

struct dsv_token {
	const uint8_t *text;
	int32_t len;
	enum dsv_stop stop; /* what ended the field; the cursor is past a delimiter, ON a line break / the limit */
};

static struct dsv_token dsv_next(const uint8_t **cursor, const uint8_t *limit, uint8_t delim)
{
	enum { BARE, QUOTED, QUOTE_IN_QUOTED } state = BARE;
	const uint8_t *const first = *cursor;
	const uint8_t *p = first;
	struct dsv_token tok = { first, 0, DSV_AT_EOL };
	for (; p < limit; p++) {
		const uint8_t ch = *p;
		if (state == QUOTE_IN_QUOTED) { /* the quote before this byte: half of an escaped pair, or the closing one */
			if (ch == '"') {
				state = QUOTED;
				continue;
			}
			state = BARE;
		}
		if (state == QUOTED) {
			if (ch == '"')
				state = QUOTE_IN_QUOTED;
			continue;
		}
		if (ch == '"') {
			state = QUOTED;
		} else if (ch == delim) {
			tok.stop = DSV_AT_DELIM;
			break;
		} else if (ch == '\n' || ch == '\r') {
			break;
		}
	}
	tok.len = (int32_t)(p - first);
	if (tok.len >= 2 && first[0] == '"' && first[tok.len - 1] == '"') {
		tok.text = first + 1;
		tok.len -= 2;
	}
	*cursor = tok.stop == DSV_AT_DELIM ? p + 1 : p;
	return tok;
}

static const uint8_t *dsv_skip_breaks(const uint8_t *p, const uint8_t *limit)
{
	while (p < limit && (*p == '\n' || *p == '\r'))
		p++;
	return p;
}

static bool dsv_is_sequence_header(const struct dsv_token *t)
{
	static const char *const NAMES[] = { "sequence", "seq", "protein", "dna", "rna", "amino", "peptide", "chain" };
	for (size_t k = 0; k < sizeof(NAMES) / sizeof(NAMES[0]); k++)
		if ((size_t)t->len == strlen(NAMES[k]) && !strncasecmp((const char *)t->text, NAMES[k], (size_t)t->len))
			return true;
	return false;
}

static int parse_dsv(uint8_t *file, const uint8_t *fend, uint8_t delim, const int32_t *lut, int32_t gap,
		     int dsv_column, int dsv_has_header, struct builder *b)
{
	/* first record: number of columns, and the first column whose name says "sequences" (dsv.c:21-24) */
	const uint8_t *p = file;
	int32_t cols = 0, seq_col = -1;
	for (;;) {
		const struct dsv_token t = dsv_next(&p, fend, delim);
		if (!t.len)
			return fail("First row has empty column");
		if (seq_col < 0 && dsv_is_sequence_header(&t))
			seq_col = cols;
		cols++;
		if (t.stop != DSV_AT_DELIM)
			break;
	}
	if (seq_col < 0) {
		/* the reference asks the user here (dsv.c:139-151) */
		if (dsv_column < 0 || dsv_column >= cols)
			return fail("No sequence column found in the header (sequence, seq, protein, dna, rna, amino, "
				    "peptide, chain); pass --column N [--no-header]");
		seq_col = dsv_column;
		if (!dsv_has_header)
			p = file; /* the first record is data */
	}
	for (p = dsv_skip_breaks(p, fend); p < fend; p = dsv_skip_breaks(p, fend)) {
		const int32_t row = b->num + 1;
		struct dsv_token t = { NULL, 0, DSV_AT_DELIM };
		int32_t col = 0;
		for (; col < cols && t.stop == DSV_AT_DELIM; col++) {
			t = dsv_next(&p, fend, delim);
			if (col != seq_col)
				continue;
			if (!t.len)
				return fail("Sequence #%d is empty", row);
			if (add_sequence(b, t.text, (size_t)t.len, true, lut, gap, row))
				return 1;
		}
		if (col <= seq_col)
			return fail("DSV row #%d has no sequence column", row);
		if (col < cols)
			return fail("DSV row #%d has too few columns", row);
		if (t.stop == DSV_AT_DELIM)
			return fail("DSV row #%d has too many columns", row);
	}
	return 0;
}

/* ---- input_load (reference src/io/input.c:28-93) ------------------------------------------ */
static int build_store(uint8_t *file, size_t size, const char *ext, const int32_t *lut, int32_t gap, int dsv_column,
		       int dsv_has_header, struct sa_host_store *out)
{
	static const char *FASTA_EXT[] = { "fasta", "fa", "fas", "fna", "ffn", "faa", "frn", "mpfa", NULL };
	static const struct {
		const char *ext;
		uint8_t delim;
	} DSV_EXT[] = { { "csv", ',' }, { "tsv", '\t' }, { "ssv", ';' }, { "psv", '|' }, { NULL, 0 } };
	struct builder b = { .w = file, .base = file };
	int rc = -1;
	for (const char **e = FASTA_EXT; *e && rc < 0; e++)
		if (!strcasecmp(*e, ext))
			rc = parse_fasta(file, file + size, lut, gap, &b);
	for (int k = 0; DSV_EXT[k].ext && rc < 0; k++)
		if (!strcasecmp(DSV_EXT[k].ext, ext))
			rc = parse_dsv(file, file + size, DSV_EXT[k].delim, lut, gap, dsv_column, dsv_has_header, &b);
	if (rc < 0)
		return fail("Unsupported file format: .%s", ext);
	if (rc)
		return 1;
	if (b.num < 2)
		return fail("Not enough sequences: %d (min: 2)", b.num);
	struct sa_meta *meta = NULL;
	if (posix_memalign((void **)&meta, 64, sizeof(*meta) * (size_t)b.num))
		return fail("Out of memory for %d sequences", b.num);
	const uint8_t *p = file;
	for (int32_t k = 0; k < b.num; k++) {
		const int32_t len = (int32_t)strlen((const char *)p);
		meta[k] = (struct sa_meta){ (int32_t)(p - file), len };
		p += len + 1;
	}
	out->in = (struct sa_input){ file, meta, b.max, b.num };
	out->blob_bytes = (size_t)(p - file);
	return 0;
}

int sa_host_parse(const uint8_t *data, size_t size, const char *ext, const int32_t lut[SA_LUT_SIZE],
		  int32_t gap, int dsv_column, int dsv_has_header, struct sa_host_store *out)
{
	memset(out, 0, sizeof(*out));
	uint8_t *copy = NULL;
	if (size > INT32_MAX)
		return fail("Input larger than 2 GiB");
	if (posix_memalign((void **)&copy, 64, size + 64))
		return fail("Out of memory reading input");
	memcpy(copy, data, size);
	memset(copy + size, 0, 64);
	if (build_store(copy, size, ext, lut, gap, dsv_column, dsv_has_header, out)) {
		free(copy);
		return 1;
	}
	return 0;
}

int sa_host_load(const char *path, const int32_t lut[SA_LUT_SIZE], int32_t gap, int dsv_column, int dsv_has_header,
		 struct sa_host_store *out)
{
	memset(out, 0, sizeof(*out));
	const char *name = strrchr(path, '/');
	name = name ? name + 1 : path;
	const char *dot = strrchr(name, '.');
	if (!dot || dot == name)
		return fail("File extension not found: %s", name);
	FILE *f = fopen(path, "rb");
	if (!f)
		return fail("Failed to open %s: %s", name, strerror(errno));
	fseek(f, 0, SEEK_END);
	const long sz = ftell(f);
	fseek(f, 0, SEEK_SET);
	if (sz < 0 || sz > INT32_MAX) {
		fclose(f);
		return fail("Input larger than 2 GiB: %s", name);
	}
	uint8_t *buf = NULL;
	if (posix_memalign((void **)&buf, 64, (size_t)sz + 64)) {
		fclose(f);
		return fail("Out of memory reading %s", name);
	}
	if (fread(buf, 1, (size_t)sz, f) != (size_t)sz) {
		fclose(f);
		free(buf);
		return fail("Failed to read %s", name);
	}
	fclose(f);
	memset(buf + sz, 0, 64);
	if (build_store(buf, (size_t)sz, dot + 1, lut, gap, dsv_column, dsv_has_header, out)) {
		free(buf);
		return 1;
	}
	return 0;
}

void sa_host_store_free(struct sa_host_store *s)
{
	free(s->in.meta);
	free(s->in.seqs);
	memset(s, 0, sizeof(*s));
}

/* ---- similarity filter (reference src/bio/filter.c:14-89, sequential semantics) ---------- */
int32_t sa_host_filter(struct sa_host_store *s, float threshold, int threads)
{
	if (threshold <= 0.0f)
		return s->in.num;
	const int32_t num = s->in.num;
	const uint8_t *seqs = s->in.seqs;
	struct sa_meta *meta = s->in.meta;
	uint8_t *lost = calloc((size_t)num, 1);
	if (!lost) {
		fail("Out of memory during sequence filtering");
		return -1;
	}
#ifdef _OPENMP
	if (threads > 0)
		omp_set_num_threads(threads);
#else
	(void)threads;
#endif
	/* Blocked so that the result equals the sequential loop exactly: a block of later sequences is
	 * first tested in parallel against every survivor BEFORE the block (their fate is final), then
	 * the block is resolved against itself in order. */
	const int32_t BLOCK = 256;
	for (int32_t b0 = 1; b0 < num; b0 += BLOCK) {
		const int32_t b1 = b0 + BLOCK < num ? b0 + BLOCK : num;
#pragma omp parallel for schedule(dynamic, 4)
		for (int32_t j = b0; j < b1; j++) {
			const uint8_t *s1 = seqs + meta[j].off;
			for (int32_t i = 0; i < b0 && !lost[j]; i++) {
				if (lost[i])
					continue;
				const uint8_t *s2 = seqs + meta[i].off;
				const int32_t ml = meta[j].len < meta[i].len ? meta[j].len : meta[i].len;
				int32_t matches = 0;
				for (int32_t k = 0; k < ml; k++)
					matches += s1[k] == s2[k];
				if ((float)matches / (float)ml >= threshold)
					lost[j] = 1;
			}
		}
		for (int32_t j = b0; j < b1; j++) {
			if (lost[j])
				continue;
			const uint8_t *s1 = seqs + meta[j].off;
			for (int32_t i = b0; i < j; i++) {
				if (lost[i])
					continue;
				const uint8_t *s2 = seqs + meta[i].off;
				const int32_t ml = meta[j].len < meta[i].len ? meta[j].len : meta[i].len;
				int32_t matches = 0;
				for (int32_t k = 0; k < ml; k++)
					matches += s1[k] == s2[k];
				if ((float)matches / (float)ml >= threshold) {
					lost[j] = 1;
					break;
				}
			}
		}
	}
	for (int32_t r = 0; r < num; r++)
		lost[r] = !lost[r]; /* -> keep mask */
	const int32_t kept = sa_host_compact(s, lost);
	free(lost);
	return kept;
}

int32_t sa_host_compact(struct sa_host_store *s, const uint8_t *keep)
{
	struct sa_meta *meta = s->in.meta;
	const int32_t num = s->in.num;
	int32_t kept = 0, used = 0, mx = 0;
	for (int32_t r = 0; r < num; r++) {
		if (!keep[r])
			continue;
		struct sa_meta m = meta[r];
		if (used != m.off)
			memmove(s->in.seqs + used, s->in.seqs + m.off, (size_t)m.len + 1);
		m.off = used;
		used += m.len + 1;
		meta[kept++] = m;
		mx = m.len > mx ? m.len : mx;
	}
	s->in.num = kept;
	s->in.max = mx;
	s->blob_bytes = (size_t)used;
	if (kept < 2) {
		fail("Not enough sequences: %d (min: 2)", kept);
		return -1;
	}
	return kept;
}

/* ---- result matrix (reference src/io/output.c:16-66, src/system/os.c:32-141,262-295) ------ */
static size_t matrix_bytes(size_t num, bool triangular)
{
	return sizeof(int32_t) * (triangular ? num * (num - 1) / 2 : num * num);
}

/* Anonymous zero-filled mapping, or -- when the caller found the matrix too large for RAM (output.c:36) -- a
 * mapping of an unnamed temporary file that disappears with the process (os.c:112-125: O_TMPFILE under /tmp;
 * $TMPDIR is honoured here, and a file system without O_TMPFILE gets an unlinked mkstemp file instead). */
int32_t *sa_host_matrix_alloc(size_t num, bool triangular, bool file_backed)
{
	const size_t bytes = matrix_bytes(num, triangular) ? matrix_bytes(num, triangular) : 1;
	int fd = -1;
	if (file_backed) {
		const char *dir = getenv("TMPDIR");
		if (!dir || !*dir)
			dir = "/tmp";
		fd = open(dir, O_TMPFILE | O_RDWR, S_IRUSR | S_IWUSR);
		if (fd < 0) {
			char path[4096];
			snprintf(path, sizeof(path), "%s/seqalign_matrix_XXXXXX", dir);
			fd = mkstemp(path);
			if (fd >= 0)
				unlink(path);
		}
		if (fd < 0) {
			fail("Could not create a temporary file in %s", dir);
			return NULL;
		}
		if (ftruncate(fd, (off_t)bytes) != 0) {
			fail("Could not create %zu byte temporary file", bytes);
			close(fd);
			return NULL;
		}
	}
	void *p = mmap(NULL, bytes, PROT_READ | PROT_WRITE, fd >= 0 ? MAP_SHARED : MAP_PRIVATE | MAP_ANONYMOUS, fd, 0);
	if (fd >= 0)
		close(fd);
	if (p == MAP_FAILED) {
		fail("Failed to allocate %.2f GiB for the similarity matrix", (double)bytes / (double)(1 << 30));
		return NULL;
	}
	madvise(p, bytes, MADV_HUGEPAGE);
	madvise(p, bytes, MADV_DONTDUMP);
	return p;
}

void sa_host_matrix_free(int32_t *m, size_t num, bool triangular)
{
	if (m)
		munmap(m, matrix_bytes(num, triangular) ? matrix_bytes(num, triangular) : 1);
}

/* MemAvailable (os.c:262-295).  SA_HOST_MEM_AVAILABLE=<bytes> overrides the probe: the test hook of the
 * temporary-file branch. */
size_t sa_host_available_memory(void)
{
	const char *forced = getenv("SA_HOST_MEM_AVAILABLE");
	if (forced && *forced)
		return (size_t)strtoull(forced, NULL, 10);
	FILE *f = fopen("/proc/meminfo", "r");
	if (!f)
		return 0;
	char line[256];
	size_t kb = 0;
	while (fgets(line, sizeof(line), f))
		if (sscanf(line, "MemAvailable: %zu kB", &kb) == 1)
			break;
	fclose(f);
	return kb * 1024;
}

/* output.c:36: a full matrix larger than 3/4 of the available memory goes to temporary file storage (and is then
 * stored triangular) */
bool sa_host_matrix_needs_file(size_t num)
{
	const size_t avail = sa_host_available_memory();
	return avail && matrix_bytes(num, false) > avail / 4 * 3;
}

/* ---- HDF5 writer (reference src/io/format/hdf5.c:14-202) ---------------------------------- */
size_t sa_host_hdf5_chunk_dim(size_t dim)
{
	if (dim <= 256) /* H5_MIN_CHUNK_SIZE: contiguous layout, -z ignored (hdf5.c:71) */
		return dim;
	size_t c = 64;
	while (c < dim) /* the reference's byte target never updates: largest 64*2^k <= dim */
		c *= 2;
	if (c > dim)
		c /= 2;
	if (c < 256)
		c = 256;
	if (c > 4096)
		c = 4096;
	return c < dim ? c : dim;
}

/* Compressed output (-z): the deflate filter of libhdf5 runs in the one thread that calls H5Dwrite -- for cfg 5 (89 994
 * sequences, 484 tiles of 4096 x 4096, level 6) that is minutes of one core while the alignment took two seconds.  Tiles are
 * independent, and so are SEGMENTS of a tile: a zlib stream may be a chain of raw deflate pieces that each end on a full
 * flush (byte-aligned, no back-references across the cut -- what pigz does), under one header and the Adler-32 of the whole.
 * So every core deflates one 4 MB segment of one tile; a batch of tiles is gathered (zero beyond the matrix; the packed
 * triangle is expanded on the way, diagonal 0), deflated, stitched and handed to H5Dwrite_chunk, one writer at a time.
 * Same dataset, same chunk shape, same filter pipeline as the reference's file (src/io/format/hdf5.c:70-112); any HDF5
 * reader inflates it; h5diff-equal by test. */
enum { SA_DEFLATE_SEGMENT = 4 << 20 };

static int write_deflated_tiles(hid_t mset, size_t chunk, size_t dim, const int32_t *matrix, bool triangular, unsigned level)
{
	const size_t nc = (dim + chunk - 1) / chunk, ntiles = nc * nc, tile_bytes = chunk * chunk * sizeof(int32_t);
	const size_t nseg = (tile_bytes + SA_DEFLATE_SEGMENT - 1) / SA_DEFLATE_SEGMENT;
	const size_t seg_bound = compressBound(SA_DEFLATE_SEGMENT) + 16, out_bound = 2 + nseg * seg_bound + 4;
	int threads = omp_get_max_threads();
	/* tiles per batch: enough segments for every thread, within a quarter of the available memory */
	size_t batch = ((size_t)threads + nseg - 1) / nseg;
	const size_t avail = sa_host_available_memory();
	while (batch > 1 && avail && batch * (tile_bytes + out_bound) > avail / 4)
		batch--;
	if (batch > ntiles)
		batch = ntiles;
	uint8_t *tiles = malloc(batch * tile_bytes), *outs = malloc(batch * out_bound);
	size_t *seg_len = malloc(sizeof(size_t) * batch * nseg);
	uLong *seg_adler = malloc(sizeof(uLong) * batch * nseg);
	int rc = 0;
	if (!tiles || !outs || !seg_len || !seg_adler)
		rc = 2;
	for (size_t c0 = 0; c0 < ntiles && !rc; c0 += batch) {
		const size_t nb = c0 + batch <= ntiles ? batch : ntiles - c0;
		/* gather: one tile row per unit of work */
#pragma omp parallel for schedule(static, 16)
		for (size_t u = 0; u < nb * chunk; u++) {
			const size_t c = c0 + u / chunk, r = u % chunk, cy = c / nc, cx = c % nc, j0 = cx * chunk;
			const size_t i = cy * chunk + r;
			int32_t *dst = (int32_t *)(tiles + (u / chunk) * tile_bytes) + r * chunk;
			const size_t w = i < dim ? (j0 + chunk <= dim ? chunk : dim - j0) : 0;
			if (!triangular) {
				if (w)
					memcpy(dst, matrix + i * dim + j0, w * sizeof(int32_t));
			} else {
				for (size_t q = 0; q < w; q++) {
					const size_t j = j0 + q;
					dst[q] = j < i ? matrix[i * (i - 1) / 2 + j] : j == i ? 0 : matrix[j * (j - 1) / 2 + i];
				}
			}
			if (w < chunk)
				memset(dst + w, 0, (chunk - w) * sizeof(int32_t));
		}
		/* deflate: one segment per unit of work, raw streams that end on a full flush (the last one finishes) */
		int bad = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(| : bad)
		for (size_t u = 0; u < nb * nseg; u++) {
			const size_t t = u / nseg, sgm = u % nseg;
			const size_t off = sgm * SA_DEFLATE_SEGMENT, len = off + SA_DEFLATE_SEGMENT <= tile_bytes ? SA_DEFLATE_SEGMENT : tile_bytes - off;
			const uint8_t *src = tiles + t * tile_bytes + off;
			uint8_t *dst = outs + t * out_bound + 2 + sgm * seg_bound;
			z_stream zs;
			memset(&zs, 0, sizeof(zs));
			if (deflateInit2(&zs, (int)level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) {
				bad |= 1;
				continue;
			}
			zs.next_in = (Bytef *)src;
			zs.avail_in = (uInt)len;
			zs.next_out = dst;
			zs.avail_out = (uInt)seg_bound;
			const int want = sgm + 1 == nseg ? Z_FINISH : Z_FULL_FLUSH;
			const int zrc = deflate(&zs, want);
			if ((want == Z_FINISH ? zrc != Z_STREAM_END : zrc != Z_OK) || zs.avail_in)
				bad |= 1;
			seg_len[u] = (size_t)zs.total_out;
			seg_adler[u] = adler32(adler32(0L, Z_NULL, 0), src, (uInt)len);
			deflateEnd(&zs);
		}
		if (bad) {
			rc = 1;
			break;
		}
		/* stitch and write, one tile after the other */
		for (size_t t = 0; t < nb && !rc; t++) {
			uint8_t *o = outs + t * out_bound;
			const unsigned flevel = level < 2 ? 0 : level < 6 ? 1 : level == 6 ? 2 : 3;
			unsigned flg = flevel << 6;
			flg += 31 - ((0x78u << 8 | flg) % 31);
			o[0] = 0x78;
			o[1] = (uint8_t)flg;
			size_t at = 2;
			uLong adler = adler32(0L, Z_NULL, 0);
			for (size_t sgm = 0; sgm < nseg; sgm++) {
				const size_t off = sgm * SA_DEFLATE_SEGMENT, len = off + SA_DEFLATE_SEGMENT <= tile_bytes ? SA_DEFLATE_SEGMENT : tile_bytes - off;
				if (at != 2 + sgm * seg_bound)
					memmove(o + at, o + 2 + sgm * seg_bound, seg_len[t * nseg + sgm]);
				at += seg_len[t * nseg + sgm];
				adler = sgm ? adler32_combine(adler, seg_adler[t * nseg + sgm], (z_off_t)len) : seg_adler[t * nseg + sgm];
			}
			o[at++] = (uint8_t)(adler >> 24);
			o[at++] = (uint8_t)(adler >> 16);
			o[at++] = (uint8_t)(adler >> 8);
			o[at++] = (uint8_t)adler;
			const size_t c = c0 + t;
			hsize_t pos[2] = { (c / nc) * chunk, (c % nc) * chunk };
			if (H5Dwrite_chunk(mset, H5P_DEFAULT, 0, pos, at, o) < 0)
				rc = 1;
		}
	}
	free(seg_adler);
	free(seg_len);
	free(outs);
	free(tiles);
	if (rc == 2)
		return fail("Out of memory during HDF5 conversion");
	return rc ? fail("Failed to write chunk to HDF5") : 0;
}

/* the file, /sequences and the (empty) /similarity_matrix dataset: everything of flush_hdf5 (src/io/format/hdf5.c:14-112) up to
 * the matrix data.  0 on success: *file_out, *mset_out open, *chunk_out = the chunk dimension (dim when contiguous). */
static int open_output(const char *path, const struct sa_host_store *s, unsigned compression, hid_t *file_out, hid_t *mset_out,
		       size_t *chunk_out)
{
	const size_t dim = (size_t)s->in.num;
	hid_t fapl = H5Pcreate(H5P_FILE_ACCESS);
	H5Pset_libver_bounds(fapl, H5F_LIBVER_LATEST, H5F_LIBVER_LATEST);
	H5Pset_alignment(fapl, 4096, 4096);
	hid_t file = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, fapl);
	H5Pclose(fapl);
	if (file < 0)
		return fail("Failed to create HDF5 file: %s", path);

	/* /sequences: 1-D, N variable-length C strings */
	const char **strs = malloc(sizeof(*strs) * dim);
	if (!strs) {
		H5Fclose(file);
		return fail("Out of memory allocating output sequence data");
	}
	for (size_t k = 0; k < dim; k++)
		strs[k] = (const char *)(s->in.seqs + s->in.meta[k].off);
	hsize_t sd[1] = { dim };
	hid_t sspace = H5Screate_simple(1, sd, NULL);
	hid_t stype = H5Tcopy(H5T_C_S1);
	H5Tset_size(stype, H5T_VARIABLE);
	hid_t sset = H5Dcreate2(file, "/sequences", stype, sspace, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
	herr_t st = sset < 0 ? -1 : H5Dwrite(sset, stype, H5S_ALL, H5S_ALL, H5P_DEFAULT, strs);
	if (sset >= 0)
		H5Dclose(sset);
	H5Sclose(sspace);
	H5Tclose(stype);
	free(strs);
	if (st < 0) {
		H5Fclose(file);
		return fail("Failed to write sequence data to HDF5 dataset");
	}

	/* /similarity_matrix: N x N int32 little endian */
	hsize_t md[2] = { dim, dim };
	hid_t mspace = H5Screate_simple(2, md, NULL);
	hid_t plist = H5Pcreate(H5P_DATASET_CREATE);
	const size_t chunk = sa_host_hdf5_chunk_dim(dim);
	if (dim > 256) {
		hsize_t cd[2] = { chunk, chunk };
		H5Pset_chunk(plist, 2, cd);
		if (compression)
			H5Pset_deflate(plist, compression);
	}
	hid_t mset = H5Dcreate2(file, "/similarity_matrix", H5T_STD_I32LE, mspace, H5P_DEFAULT, plist, H5P_DEFAULT);
	H5Pclose(plist);
	H5Sclose(mspace);
	if (mset < 0) {
		H5Fclose(file);
		return fail("Failed to create HDF5 dataset for Similarity Matrix");
	}
	*file_out = file;
	*mset_out = mset;
	*chunk_out = chunk;
	return 0;
}

/* Output whose tiles arrive finished (sa_zjob_tile_row of include/seqalign_hip.h): zlib streams from the device-side encoder
 * when `compression` > 0, the raw tiles when 0.  Same file, dataset, chunk shape and filter pipeline as sa_host_write_hdf5 --
 * the tiles go to H5Dwrite_chunk as they are, tile row after tile row. */
int sa_host_write_hdf5_streams(const char *path, const struct sa_host_store *s, unsigned compression, sa_host_tiles_fn next,
			       void *user)
{
	const size_t dim = (size_t)s->in.num;
	if (dim <= 256)
		return fail("Tiles need a chunked dataset (more than 256 sequences)");
	hid_t file, mset;
	size_t chunk;
	if (open_output(path, s, compression, &file, &mset, &chunk))
		return 1;
	const size_t nc = (dim + chunk - 1) / chunk;
	const uint8_t **streams = malloc(sizeof(*streams) * nc);
	size_t *sizes = malloc(sizeof(*sizes) * nc);
	uint32_t *rows = malloc(sizeof(*rows) * nc), *cols = malloc(sizeof(*cols) * nc);
	int rc = streams && sizes && rows && cols ? 0 : fail("Out of memory during HDF5 conversion");
	size_t written = 0;
	while (!rc) {
		const int n = next(user, rows, cols, streams, sizes);
		if (n < 0)
			rc = fail("Failed to encode tiles of the Similarity Matrix");
		if (n <= 0)
			break;
		for (int t = 0; t < n && !rc; t++) {
			hsize_t pos[2] = { (hsize_t)rows[t] * chunk, (hsize_t)cols[t] * chunk };
			if (rows[t] >= nc || cols[t] >= nc || H5Dwrite_chunk(mset, H5P_DEFAULT, 0, pos, sizes[t], streams[t]) < 0)
				rc = fail("Failed to write chunk to HDF5");
		}
		written += (size_t)n;
	}
	if (!rc && written != nc * nc)
		rc = fail("The Similarity Matrix has %zu tiles, %zu were delivered", nc * nc, written);
	free(cols);
	free(rows);
	free(sizes);
	free(streams);
	H5Dclose(mset);
	H5Fclose(file);
	return rc;
}

int sa_host_write_hdf5(const char *path, const struct sa_host_store *s, const int32_t *matrix, bool triangular,
		       unsigned compression)
{
	const size_t dim = (size_t)s->in.num;
	int rc = 1;
	herr_t st;
	hid_t file, mset;
	size_t chunk;
	if (open_output(path, s, compression, &file, &mset, &chunk))
		return 1;
	if (dim > 256 && compression && !getenv("SA_HOST_SERIAL_DEFLATE")) {
		rc = write_deflated_tiles(mset, chunk, dim, matrix, triangular, compression);
	} else if (!triangular) {
		st = H5Dwrite(mset, H5T_NATIVE_INT32, H5S_ALL, H5S_ALL, H5P_DEFAULT, matrix);
		rc = st < 0 ? fail("Failed to write Similarity Matrix to HDF5") : 0;
	} else {
		/* expand `rows` rows at a time; unlike hdf5.c:152-162 the diagonal is written as 0 */
		size_t rows = chunk > 4 ? chunk : 4;
		const size_t avail = sa_host_available_memory();
		if (avail && rows * dim * 4 * 4 > avail && avail / (16 * dim) > 4)
			rows = avail / (16 * dim);
		int32_t *buf = malloc(sizeof(int32_t) * rows * dim);
		hid_t fspace = H5Dget_space(mset);
		rc = 0;
		if (!buf)
			rc = fail("Out of memory during HDF5 conversion");
		for (size_t off = 0; off < dim && !rc; off += rows) {
			const size_t end = off + rows < dim ? off + rows : dim;
#pragma omp parallel for schedule(static)
			for (size_t i = off; i < end; i++) {
				int32_t *row = buf + dim * (i - off);
				for (size_t j = 0; j < i; j++)
					row[j] = matrix[i * (i - 1) / 2 + j];
				row[i] = 0;
				for (size_t j = i + 1; j < dim; j++)
					row[j] = matrix[j * (j - 1) / 2 + i];
			}
			hsize_t start[2] = { off, 0 }, count[2] = { end - off, dim };
			H5Sselect_hyperslab(fspace, H5S_SELECT_SET, start, NULL, count, NULL);
			hid_t mem = H5Screate_simple(2, count, NULL);
			st = H5Dwrite(mset, H5T_NATIVE_INT32, mem, fspace, H5P_DEFAULT, buf);
			H5Sclose(mem);
			if (st < 0)
				rc = fail("Failed to write chunk to HDF5");
		}
		if (fspace >= 0)
			H5Sclose(fspace);
		free(buf);
	}
	H5Dclose(mset);
	H5Fclose(file);
	return rc;
}
