/*
 * cli/seqalign.c -- the `seqalign` command line tool on top of libseqalign_hip.so.
 *
 * Same surface as the reference binary (src/main.c:9-39 and the per-TU option tables, README "Usage"):
 *   required  -i/--input FILE   -o/--output FILE (or -W)   -m/--matrix NAME   -a/--align METHOD
 *             -p/--gap-penalty N   |   -s/--gap-open N  -e/--gap-extend N
 *   optional  -l/--list-matrices  -f/--filter FLOAT  -z/--compression N  -B/--benchmark  -T/--threads N
 *             -C/--no-cuda  -W/--no-write  -P/--no-progress  -D/--no-detail  -F/--force-proceed
 *             -Q/--quiet  -V/--verbose  -h/--help
 *   added     --column N  --no-header   (non-interactive answers to the reference's DSV column prompt)
 * Flow: parse+validate -> load (FASTA/DSV) -> filter -> allocate matrix -> sa_hip_align -> HDF5 -> -B report.
 * Exit code 1 with a usage hint on any failure (src/main.c:11-14).
 */
#define _GNU_SOURCE
#include <errno.h>
#include <omp.h>
#include <stdarg.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "sa_host.h"

static bool quiet, verbose, force_yes, no_progress;

static void info(const char *fmt, ...)
{
	if (quiet)
		return;
	va_list ap;
	va_start(ap, fmt);
	vprintf(fmt, ap);
	va_end(ap);
	putchar('\n');
}

static void verb(const char *fmt, ...)
{
	if (quiet || !verbose)
		return;
	va_list ap;
	va_start(ap, fmt);
	vprintf(fmt, ap);
	va_end(ap);
	putchar('\n');
}

static void err(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	fputs("error: ", stderr);
	vfprintf(stderr, fmt, ap);
	va_end(ap);
	fputc('\n', stderr);
}

/* y/n prompt; -F answers yes (third_party/clix/print.h:585-603), a non-interactive stdin takes the default */
static bool ask(const char *question, bool dflt)
{
	if (force_yes)
		return true;
	if (!isatty(STDIN_FILENO))
		return dflt;
	printf("%s [%s] ", question, dflt ? "Y/n" : "y/N");
	fflush(stdout);
	char line[16];
	if (!fgets(line, sizeof(line), stdin) || line[0] == '\n')
		return dflt;
	return line[0] == 'y' || line[0] == 'Y';
}

static double now(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

struct options {
	const char *input, *output, *matrix, *align;
	long gap_pen, gap_open, gap_ext; /* -1 = not given */
	float filter;
	unsigned compression;
	int threads;
	bool benchmark, no_device, no_write, list;
	int dsv_column, dsv_has_header;
};

static void usage(const char *argv0)
{
	printf("Usage: %s -i FILE -o FILE -m MATRIX -a METHOD (-p N | -s N -e N) [options]\n"
	       "  -i, --input FILE         Input file path: FASTA, DSV format\n"
	       "  -o, --output FILE        Output file path: HDF5 format\n"
	       "  -m, --matrix MATRIX      Substitution matrix (use -l to list)\n"
	       "  -a, --align METHOD       Needleman-Wunsch: nw | Gotoh: ga | Smith-Waterman: sw\n"
	       "  -p, --gap-penalty N      Linear gap penalty\n"
	       "  -s, --gap-open N         Affine gap open penalty\n"
	       "  -e, --gap-extend N       Affine gap extend penalty\n"
	       "  -l, --list-matrices      List available substitution matrices\n"
	       "  -f, --filter FLOAT       Filter sequences with similarity above threshold [0.0-1.0]\n"
	       "  -z, --compression N      Compression level for HDF5 datasets [0-9]\n"
	       "  -B, --benchmark          Enable timing of various steps\n"
	       "  -T, --threads N          Number of host threads (0 = auto)\n"
	       "  -C, --no-cuda            Not available: this build has no CPU alignment path\n"
	       "  -W, --no-write           Disable writing to output file\n"
	       "  -P, --no-progress        Disable progress bars\n"
	       "  -D, --no-detail          Disable detailed printing (accepted, no-op)\n"
	       "  -F, --force-proceed      Force proceed without user prompts (for CI)\n"
	       "  -Q, --quiet              Suppress all non-error printing\n"
	       "  -V, --verbose            Enable verbose printing\n"
	       "      --column N           DSV: 1-based sequence column when no header names it\n"
	       "      --no-header          DSV: with --column, the first row is data\n"
	       "  -h, --help               Display this help message\n",
	       argv0);
}

static bool parse_long(const char *s, long lo, long hi, long *out)
{
	char *end;
	errno = 0;
	const long v = strtol(s, &end, 10);
	if (errno || end == s || *end || v < lo || v > hi)
		return false;
	*out = v;
	return true;
}

/* returns 0 ok, 1 error, 2 handled-and-exit-success */
static int parse_args(int argc, char **argv, struct options *o)
{
	static const struct {
		const char *lname;
		char sname;
		bool takes;
	} OPTS[] = { { "input", 'i', true }, { "output", 'o', true }, { "matrix", 'm', true }, { "align", 'a', true },
		     { "gap-penalty", 'p', true }, { "gap-open", 's', true }, { "gap-extend", 'e', true },
		     { "list-matrices", 'l', false }, { "filter", 'f', true }, { "compression", 'z', true },
		     { "benchmark", 'B', false }, { "threads", 'T', true }, { "no-cuda", 'C', false },
		     { "no-write", 'W', false }, { "no-progress", 'P', false }, { "no-detail", 'D', false },
		     { "force-proceed", 'F', false }, { "quiet", 'Q', false }, { "verbose", 'V', false },
		     { "help", 'h', false }, { "column", 1, true }, { "no-header", 2, false }, { NULL, 0, false } };
	*o = (struct options){ .gap_pen = -1, .gap_open = -1, .gap_ext = -1, .dsv_column = -1, .dsv_has_header = 1 };
	for (int k = 1; k < argc; k++) {
		const char *arg = argv[k];
		if (arg[0] != '-' || !arg[1]) {
			err("Unexpected argument: %s", arg);
			return 1;
		}
		/* one long option, or a bundle of short ones (-BVW, third_party/clix/args.h:1651-1695) */
		const char *bundle = arg + 1;
		const bool is_long = arg[1] == '-';
		do {
			int idx = -1;
			const char *inline_val = NULL;
			if (is_long) {
				const char *name = arg + 2, *eq = strchr(name, '=');
				const size_t nlen = eq ? (size_t)(eq - name) : strlen(name);
				for (int t = 0; OPTS[t].lname; t++)
					if (strlen(OPTS[t].lname) == nlen && !strncmp(OPTS[t].lname, name, nlen))
						idx = t;
				inline_val = eq ? eq + 1 : NULL;
			} else {
				for (int t = 0; OPTS[t].lname; t++)
					if (OPTS[t].sname == *bundle)
						idx = t;
			}
			if (idx < 0) {
				err("Unknown option: %s", arg);
				return 1;
			}
			const char *val = NULL;
			if (OPTS[idx].takes) {
				if (inline_val)
					val = inline_val;
				else if (!is_long && bundle[1])
					val = bundle + 1; /* -p4 */
				else if (k + 1 < argc)
					val = argv[++k];
				else {
					err("Option --%s requires a parameter", OPTS[idx].lname);
					return 1;
				}
			}
			long v;
			switch (OPTS[idx].sname) {
			case 'i': o->input = val; break;
			case 'o': o->output = val; break;
			case 'm': o->matrix = val; break;
			case 'a': o->align = val; break;
			case 'p':
			case 's':
			case 'e':
				/* src/bio/align.c:127-128 */
				if (!parse_long(val, 0, INT32_MAX, &v)) {
					err("Gap values must be positive integers");
					return 1;
				}
				*(OPTS[idx].sname == 'p' ? &o->gap_pen : OPTS[idx].sname == 's' ? &o->gap_open : &o->gap_ext) = v;
				break;
			case 'l': o->list = true; break;
			case 'f': {
				char *end;
				o->filter = strtof(val, &end);
				if (end == val || *end || o->filter < 0.0f || o->filter > 1.0f) {
					err("Filter threshold must be between 0.0 and 1.0");
					return 1;
				}
				break;
			}
			case 'z':
				if (!parse_long(val, 0, 9, &v)) {
					err("Compression level must be between 0-9");
					return 1;
				}
				o->compression = (unsigned)v;
				break;
			case 'B': o->benchmark = true; break;
			case 'T':
				if (!parse_long(val, 0, 1024, &v)) {
					err("Invalid thread count");
					return 1;
				}
				o->threads = (int)v;
				break;
			case 'C': o->no_device = true; break;
			case 'W': o->no_write = true; break;
			case 'P': no_progress = true; break;
			case 'D': break;
			case 'F': force_yes = true; break;
			case 'Q': quiet = true; break;
			case 'V': verbose = true; break;
			case 'h': usage(argv[0]); return 2;
			case 1:
				if (!parse_long(val, 1, 1 << 20, &v)) {
					err("Invalid column number");
					return 1;
				}
				o->dsv_column = (int)v - 1;
				break;
			case 2: o->dsv_has_header = 0; break;
			}
			if (is_long || OPTS[idx].takes)
				break;
		} while (*++bundle);
	}
	return 0;
}

/* SA_CLI_TIMES=1 (diagnostics): wall-clock stamps of the tool's stages on stderr */
static double t_process0;
static void stamp(const char *what)
{
	static int on = -1;
	if (on < 0)
		on = getenv("SA_CLI_TIMES") != NULL;
	if (on)
		fprintf(stderr, "[cli %8.1f ms] %s\n", (now() - t_process0) * 1e3, what);
}

static void progress_line(double fraction, void *user)
{
	(void)user;
	fprintf(stderr, "\rAligning sequences: %3d%%", (int)(fraction * 100.0));
	fflush(stderr);
}

/* the tiles of the device walk as they arrive (sa_zjob_next), with the progress line fed from them: tile (r, c) of shell
 * b = max(r, c) arrives when column block b is aligned, and the pairs up to there are (b + 1)^2 of nc^2 */
struct tile_feed {
	sa_zjob *job;
	size_t nc;
	bool show;
};
static int next_tiles(void *user, uint32_t *rows, uint32_t *cols, const uint8_t **streams, size_t *sizes)
{
	struct tile_feed *f = user;
	const int n = sa_zjob_next(f->job, rows, cols, streams, sizes);
	if (n > 0 && f->show) {
		const uint32_t b = rows[n - 1] > cols[n - 1] ? rows[n - 1] : cols[n - 1];
		progress_line((double)(b + 1) * (double)(b + 1) / ((double)f->nc * (double)f->nc), NULL);
	}
	return n;
}

int main(int argc, char **argv)
{
	t_process0 = now();
	struct options o;
	const int prc = parse_args(argc, argv, &o);
	if (prc == 2)
		return 0;
	struct sa_scoring sc;
	memset(&sc, 0, sizeof(sc));
	bool ok = prc == 0;
	if (ok && o.list) { /* -l: src/bio/matrices.c:27-33 */
		printf("\nListing available substitution matrices\n");
		for (int fam = 0; fam < 2; fam++) {
			printf("\n%s Matrices:\n  ", fam ? "Nucleotide" : "Amino");
			for (int k = 0, col = 0; k < sa_matrix_count(); k++)
				if (sa_matrix_is_nucleotide(k) == fam)
					printf("%-10s%s", sa_matrix_name(k), ++col % 5 ? "" : "\n  ");
			putchar('\n');
		}
		return 0;
	}
	/* ---- validation, in the reference's terms (src/bio/align.c:130-201, ga.c:70-88, output.c:126-148) */
	if (ok && !o.input)
		ok = (err("Missing required option: -i, --input"), false);
	if (ok && !o.matrix)
		ok = (err("Missing required option: -m, --matrix"), false);
	if (ok && !o.align)
		ok = (err("Missing required option: -a, --align"), false);
	if (ok && o.output && o.no_write)
		ok = (err("Options -o, --output and -W, --no-write conflict"), false);
	if (ok && !o.output && !o.no_write)
		ok = (err("Missing required option: -o, --output"), false);
	if (ok && sa_matrix_load(o.matrix, sc.lut, sc.sub))
		ok = (err("Invalid substitution matrix name"), false);
	if (ok && (sc.method = sa_method_parse(o.align)) < 0)
		ok = (err("Invalid alignment method"), false);
	if (ok && o.gap_pen >= 0 && (o.gap_open >= 0 || o.gap_ext >= 0))
		ok = (err("Options -p and -s/-e conflict"), false);
	if (ok && sa_method_gap_kind(sc.method) == SA_GAP_LINEAR) {
		if (o.gap_open >= 0 || o.gap_ext >= 0)
			ok = (err("Affine gaps cannot be set for non-affine methods"), false);
		else if (o.gap_pen < 0)
			ok = (err("Missing required option: -p, --gap-penalty"), false);
	} else if (ok) {
		if (o.gap_pen >= 0)
			ok = (err("Gap penalty cannot be set for non-linear methods"), false);
		else if (o.gap_open < 0 || o.gap_ext < 0)
			ok = (err("Missing required option: -s, --gap-open and -e, --gap-extend"), false);
	}
	if (ok) {
		sc.gap_pen = o.gap_pen >= 0 ? -(int32_t)o.gap_pen : 0;
		sc.gap_opn = o.gap_open >= 0 ? -(int32_t)o.gap_open : 0;
		sc.gap_ext = o.gap_ext >= 0 ? -(int32_t)o.gap_ext : 0;
		if (sc.method == SA_METHOD_GA && sc.gap_opn == sc.gap_ext &&
		    ask("Equal affine gaps found, switch to Needleman-Wunsch?", true)) {
			sc.method = SA_METHOD_NW;
			sc.gap_pen = sc.gap_opn;
			sc.gap_opn = sc.gap_ext = SA_SCORE_MIN;
		}
	}
	if (ok && o.no_device)
		ok = (err("-C/--no-cuda: this build has no CPU alignment path (HIP device required)"), false);
	if (ok && o.output && access(o.output, F_OK) == 0) {
		if (!ask("Output file already exists. Do you want to DELETE it?", false))
			ok = (err("Output file exists and will not be overwritten"), false);
		else if (remove(o.output) != 0)
			ok = (err("Failed to delete existing output file"), false);
	}
	if (!ok) {
		fprintf(stderr, "Use %s -h, --help for usage information\n", argv[0]);
		return 1;
	}

	/* -T: host threads (validate_threads, src/system/os.c:466-473).  The alignment itself runs on the device; the
	 * host's parallel loops are the triangular->full expansion of the HDF5 writer and the CPU filter */
	if (o.threads > 0)
		omp_set_num_threads(o.threads);

	info("SEQUENCE ALIGNER (MI355X / HIP)");
	info("Input: %s", o.input);
	if (o.output)
		info("Output: %s", o.output);
	info("Matrix: %s", o.matrix);
	info("Method: %s", sa_method_name(sc.method));
	if (sc.method == SA_METHOD_NW)
		info("Gap penalty: %d", sc.gap_pen);
	else
		info("Gap open: %d, extend: %d", sc.gap_opn, sc.gap_ext);
	if (o.filter > 0.0f)
		info("Filter threshold: %.1f%%", (double)o.filter * 100.0);

	double t_in = 0, t_filter = 0, t_align = 0, t_out = 0, t0;
	stamp("options parsed");
	struct sa_host_store store;
	t0 = now();
	if (sa_host_load(o.input, sc.lut, sc.gap_pen, o.dsv_column, o.dsv_has_header, &store)) {
		err("%s", sa_host_error());
		return 1;
	}
	t_in = now() - t0;
	stamp("input loaded");
	t0 = now();
	const int32_t before = store.in.num;
	if (o.filter > 0.0f) { /* similarity relation on the device, greedy keep/drop in sequence order (filter.c:14-89) */
		uint8_t *keep = malloc((size_t)store.in.num);
		if (!keep || sa_hip_filter(store.in, o.filter, keep) < 0) {
			err("%s", keep ? sa_last_error() : "Out of memory during sequence filtering");
			return 1;
		}
		if (sa_host_compact(&store, keep) < 0) {
			err("%s", sa_host_error());
			return 1;
		}
		free(keep);
	}
	t_filter = now() - t0;
	if (o.filter > 0.0f)
		info("Filtered out %d sequences", before - store.in.num);
	info("Loaded %d sequences", store.in.num);
	info("Average sequence length: %.2f", (double)store.blob_bytes / (double)store.in.num - 1.0);

	/* output_load (src/io/output.c:35-55): a full matrix that exceeds 3/4 of the available RAM goes to temporary
	 * file storage and is stored triangular; so is one the device(s) cannot hold */
	const size_t n = (size_t)store.in.num;
	struct sa_output out = { NULL, NULL, n, false };
	bool pinned = false;
	/* -z on a chunked dataset: the tiles are deflated on the device, from the packed scores where they were computed
	 * (include/seqalign_hip.h: sa_hip_tiles_begin / sa_zjob_next) -- no host matrix at all.  libhdf5's filter in the
	 * one writing thread (what flush_hdf5 leaves to H5Dwrite, src/io/format/hdf5.c:148-194) takes 41 CPU-minutes for config 5.
	 * SA_HOST_CPU_DEFLATE=1 keeps zlib at exactly the level asked for (all cores, sa_host_write_hdf5). */
	const long long npairs = (long long)n * ((long long)n - 1) / 2;
	const size_t zchunk = sa_host_hdf5_chunk_dim(n);
	bool device_deflate = !o.no_write && n > 256 && !getenv("SA_HOST_MATRIX") &&
			      (o.compression > 0 ? !getenv("SA_HOST_CPU_DEFLATE") && !getenv("SA_HOST_SERIAL_DEFLATE")
						 /* without -z the same walk returns the tiles as they are: H5Dwrite_chunk instead of H5Dwrite's
						  * gather of every tile out of N-wide rows.  On several devices the plain path stays with sa_hip_align
						  * (tiles dealt by DP work, RCCL all-gather: the whole matrix on every device and in host memory);
						  * with -z the walk itself runs on all of them (sa_hip_tiles_begin: block b -> device b mod n) */
						 : sa_hip_device_count() == 1);
	if (device_deflate) {
		/* the packed scores + one tile row: raw, or raw + worst-case slots and streams (1 + 2 x 2.02 x the row's raw bytes) */
		const size_t row_raw = ((n + zchunk - 1) / zchunk) * zchunk * zchunk * sizeof(int32_t);
		device_deflate = sa_hip_memory(sizeof(int32_t) * (size_t)npairs + (o.compression ? 6 : 1) * row_raw);
		stamp("device memory probed (runtime up)");
	}
	if (!o.no_write && !device_deflate) {
		const size_t full_bytes = sizeof(int32_t) * n * n;
		const bool tmpf = sa_host_matrix_needs_file(n);
		out.triangular = tmpf || !sa_hip_memory(full_bytes);
		stamp("device memory probed (runtime up)");
		info("Similarity Matrix dimensions: %zu x %zu%s", n, n, out.triangular ? " (stored triangular)" : "");
		if (tmpf)
			info("Similarity Matrix size exceeds memory limits, creating temporary file storage");
		t0 = now();
		out.matrix = sa_host_matrix_alloc(n, out.triangular, tmpf);
		if (!out.matrix) {
			err("%s", sa_host_error());
			return 1;
		}
		stamp("matrix allocated");
		/* page-lock it for the device->host copies while it is being set up (a file-backed matrix is larger than
		 * RAM by definition and stays pageable: the library stages those copies) */
		if (!tmpf) {
			const size_t bytes = sizeof(int32_t) * (out.triangular ? n * (n - 1) / 2 : n * n);
			/* (the library's own rule, sa_ctx_align_host: never lock more than half of what is available -- a
			 * registration of 70 % of free RAM thrashes or meets the OOM killer instead of failing cleanly) */
			const size_t avail = sa_host_available_memory();
			pinned = bytes && (!avail || bytes <= avail / 2) && sa_hip_host_register(out.matrix, bytes) == 0;
		}
		t_out += now() - t0;
		stamp("matrix page-locked");
	}

	const long long pairs = npairs;
	info("Performing %lld pairwise alignments", pairs);
	/* progress (ppercent / pproportc in the reference's launch loop, src/interface/seqalign_cuda.c:181,286-289,293) */
	const bool show_progress = !no_progress && !quiet;
	if (show_progress) {
		sa_hip_set_progress(progress_line, NULL);
		progress_line(0.0, NULL);
	}
	verb("Devices: %d (%s)", sa_hip_device_count(), sa_hip_device_name(0) ? sa_hip_device_name(0) : "none");
	double t_setup = 0;
	int schedule = 0;
	if (device_deflate) {
		info("Similarity Matrix dimensions: %zu x %zu (%s on the device, tile by tile)", n, n, o.compression ? "deflated" : "tiled");
		/* the alignment runs on the device while the finished tiles are written: the tiles whose larger tile index is b
		 * need exactly column block b (include/seqalign_hip.h: sa_hip_tiles_begin).  The reference's phases (src/main.c:31-34,
		 * bench_align / bench_io) overlap here: "Alignment" below is the device's alignment time, "Output" the rest of
		 * the wall time of this section. */
		t0 = now();
		sa_zjob *job = sa_hip_tiles_begin(store.in, &sc, zchunk, (int)o.compression);
		if (!job) {
			err("%s", sa_last_error());
			return 1;
		}
		t_setup = now() - t0;
		stamp("sa_hip_tiles_begin returned");
		t0 = now();
		struct tile_feed feed = { job, sa_zjob_tiles_per_row(job), show_progress };
		if (sa_host_write_hdf5_streams(o.output, &store, o.compression, next_tiles, &feed)) {
			err("%s (%s)", sa_host_error(), sa_last_error());
			return 1;
		}
		if (show_progress) {
			progress_line(1.0, NULL);
			fputc('\n', stderr);
			sa_hip_set_progress(NULL, NULL);
		}
		t_align = sa_zjob_align_seconds(job);
		const double section = now() - t0;
		t_out += section > t_align ? section - t_align : 0.0;
		double enc_ms = 0, copy_ms = 0;
		uint64_t raw = 0, outb = 0;
		sa_zjob_stats(job, &enc_ms, &copy_ms, &raw, &outb);
		verb("%s on the device: %.2f GB -> %.2f GB (%.2f : 1); the writer waited %.0f ms for the encoder, %.0f ms for gather + copy",
		     o.compression ? "Deflated" : "Tiled", (double)raw / 1e9, (double)outb / 1e9, outb ? (double)raw / (double)outb : 0.0, enc_ms,
		     copy_ms);
		sa_zjob_destroy(job);
		stamp("HDF5 written");
	} else {
		t0 = now();
		if (!sa_hip_align(store.in, out, &sc)) {
			err("%s", sa_last_error());
			return 1;
		}
		t_align = now() - t0;
		stamp("sa_hip_align returned");
		if (show_progress) {
			progress_line(1.0, NULL);
			fputc('\n', stderr);
			sa_hip_set_progress(NULL, NULL);
		}
		/* the reference times the launch/copy loop only (bench_align_start..end inside cuda_align,
		 * src/interface/seqalign_cuda.c:182,292): device set-up and uploads are not part of "Alignment" */
		t_setup = t_align - sa_hip_last_align_seconds();
		t_align = sa_hip_last_align_seconds();
		schedule = sa_hip_last_align_path();

		if (!o.no_write) {
			t0 = now();
			if (sa_host_write_hdf5(o.output, &store, out.matrix, out.triangular, o.compression)) {
				err("%s", sa_host_error());
				return 1;
			}
			t_out += now() - t0;
			stamp("HDF5 written");
		}
	}
	if (o.benchmark) { /* -B: src/util/benchmark.c:50-64 */
		const double total = t_in + t_filter + t_align + t_out;
		printf("Timing breakdown:\n  Input: %.3f sec\n  Filter: %.3f sec\n  Alignment: %.3f sec\n  Output: %.3f sec\n"
		       "  Total: %.3f sec\n",
		       t_in, t_filter, t_align, t_out, total);
		printf("  (device set-up and upload, outside the phases as in the reference: %.3f sec)\n", t_setup);
		printf("  (schedule: %s)\n", device_deflate ? (o.compression ? "column blocks into device memory, their tiles deflated on the device and written meanwhile"
								       : "column blocks into device memory, their tiles delivered as HDF5 chunks meanwhile")
				       : schedule == 2
					       ? "tiles dealt over the devices, RCCL all-gather of the dense shares, placement on every device"
					       : "every device delivers its slice of the packed index straight into the host matrix");
		printf("Alignments per second: %.2f\n", t_align > 0 ? (double)pairs / t_align : 0.0);
	}
	if (pinned)
		sa_hip_host_unregister(out.matrix);
	sa_host_matrix_free(out.matrix, n, out.triangular);
	sa_host_store_free(&store);
	stamp("released");
	/* everything is written and closed: leave without the runtime's teardown (device reset, queue and signal
	 * destruction: ~0.15 s that nothing waits for) */
	fflush(NULL);
	if (getenv("SA_CLI_CLEAN_EXIT")) /* (a profiler that writes its files from an exit handler: rocprofv3) */
		return 0;
	_exit(0);
}
