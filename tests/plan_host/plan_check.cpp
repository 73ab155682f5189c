/* tests/plan_host/plan_check.cpp -- TEST HARNESS (not part of the product library).
 *
 * Links the product's host-only planner (sequencealigner_amd/csrc/sa_plan.cpp, sa_limits.cpp, sa_tables.cpp) into a plain
 * executable that tests/test_plan_host.py builds with g++ -fsanitize=address,undefined and runs on the CPU: the launch
 * plan of a packed pair range is built exactly as sa_ctx_align_range / sa_ctx_align_share build it, and then CHECKED --
 * every pair of the range is covered by exactly one tile, tile lists and dense-share layouts stay inside their
 * buffers, placement segments cover the range once, arranged copies are permutations that stay inside their blocks.
 *
 *   plan_check <lens.i32> <method> <matrix> <gap_pen> <gap_open> <gap_ext> <CUs> <start> <count> <world> <share_host> ...
 *
 * (start, count, world, share_host) may repeat; count = -1 means "to the end", start/count accept k*2^30 style via
 * plain integers only.  Prints one line per plan; exit status 0 = every check held. */
#include <algorithm>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "../../sequencealigner_amd/csrc/sa_plan.h"

static int g_fail = 0;
static int g_method = 0; /* the scoring's method: the workgroup size of a packed class depends on it (sa_pk_wpb) */
#define CHECK(cond, ...)                                        \
	do {                                                    \
		if (!(cond)) {                                  \
			fprintf(stderr, "CHECK FAILED %s:%d: ", __FILE__, __LINE__); \
			fprintf(stderr, __VA_ARGS__);           \
			fprintf(stderr, "\n");                  \
			if (++g_fail > 20)                      \
				exit(1);                        \
		}                                               \
	} while (0)

struct TileGeo {
	int32_t j[2], ia[2], ib[2], i_begin, i_count;
	bool dup;
};

/* the rows and columns of tile t of a class, derived from the class's lists alone (what the kernels do) */
static TileGeo tile_geo(const SaHostPlan &pl, const SaHostClass &cl, int32_t t)
{
	TileGeo g{};
	const int64_t start = pl.start, end = pl.start + pl.count;
	auto rows_of = [&](int32_t j, int32_t &ia, int32_t &ib) {
		const int64_t tri = sa_tri(j);
		ia = (int32_t)std::max<int64_t>(0, start - tri);
		ib = (int32_t)std::min<int64_t>(j, end - tri);
	};
	if (cl.cls >= SA_PK_CLASS0) {
		const SaPkCls pc = sa_pk_decode(cl.cls);
		const int32_t rows = sa_pk_wpb(g_method, pc.g, pc.k) * (64 / pc.g) * cl.chunk;
		const int32_t npairs = (cl.ncols + 1) / 2, nfull = cl.tprefix[(size_t)npairs];
		int32_t lo;
		if (t < nfull)
			lo = (int32_t)(std::upper_bound(cl.tprefix.begin(), cl.tprefix.begin() + npairs + 1, t) - cl.tprefix.begin()) - 1;
		else
			lo = cl.tprefix[(size_t)(npairs + 1 + (t - nfull))];
		const size_t c0 = (size_t)2 * (size_t)lo, c1 = c0 + 1 < cl.jlist.size() ? c0 + 1 : c0;
		g.dup = c0 == c1;
		g.j[0] = cl.jlist[c0], g.j[1] = cl.jlist[c1];
		rows_of(g.j[0], g.ia[0], g.ib[0]);
		rows_of(g.j[1], g.ia[1], g.ib[1]);
		const int32_t ra = std::min(g.ia[0], g.ia[1]), rb = std::max(g.ib[0], g.ib[1]);
		const int32_t chunk = t < nfull ? t - cl.tprefix[(size_t)lo] : cl.tprefix[(size_t)lo + 1] - cl.tprefix[(size_t)lo];
		g.i_begin = ra + chunk * rows;
		g.i_count = std::min(rows, rb - g.i_begin);
	} else {
		const int G = cl.cls == SA_SYS_CLASS_LONG ? 64 : SA_SYS_CLASSES[cl.cls].G;
		const int rows = cl.cls == SA_SYS_CLASS_LONG ? SA_SYS_WPB(64, true) * std::min(pl.chunk, 16) : SA_SYS_WPB(G, false) * (64 / G) * pl.chunk;
		const int32_t k = (int32_t)(std::upper_bound(cl.tprefix.begin(), cl.tprefix.end(), t) - cl.tprefix.begin()) - 1;
		g.dup = true;
		g.j[0] = g.j[1] = cl.jlist[(size_t)k];
		rows_of(g.j[0], g.ia[0], g.ib[0]);
		g.ia[1] = g.ia[0], g.ib[1] = g.ib[0];
		g.i_begin = g.ia[0] + (t - cl.tprefix[(size_t)k]) * rows;
		g.i_count = std::min(rows, g.ib[0] - g.i_begin);
	}
	return g;
}

static void check_plan(const SaPlanInputs &in, const SaHostPlan &pl)
{
	const int64_t start = pl.start, end = pl.start + pl.count;
	const int32_t j0 = sa_column_of(start), j1 = sa_column_of(end - 1);
	std::vector<int64_t> covered((size_t)(j1 - j0 + 1), 0);
	int64_t tiles_total = 0;
	for (const auto &cl : pl.classes) {
		CHECK(cl.ncols == (int32_t)cl.jlist.size() && cl.ntiles >= 0, "class %d: ncols", cl.cls);
		CHECK(std::is_sorted(cl.jlist.begin(), cl.jlist.end()), "class %d: columns not ascending", cl.cls);
		const bool is_pk = cl.cls >= SA_PK_CLASS0;
		if (is_pk) {
			const int32_t npairs = (cl.ncols + 1) / 2;
			CHECK((int32_t)cl.tprefix.size() == npairs + 1 + cl.npart, "class %d: prefix length %zu vs %d pairs + %d partial", cl.cls,
			      cl.tprefix.size(), npairs, cl.npart);
			CHECK(cl.ntiles == cl.tprefix[(size_t)npairs] + cl.npart, "class %d: tile count", cl.cls);
			CHECK((int32_t)cl.part_rows.size() == cl.npart, "class %d: part_rows", cl.cls);
			CHECK(cl.chunk >= 1 && cl.chunk <= SA_SYS_CHUNK, "class %d: chunk %d", cl.cls, cl.chunk);
			const SaPkCls pc = sa_pk_decode(cl.cls);
			for (int32_t j : cl.jlist) {
				const int32_t n = in.meta[j].len;
				CHECK(n <= pc.g * pc.k && n > pc.g * (pc.k - 1), "class %d (G %d K %d): column %d of length %d", cl.cls, pc.g, pc.k, j, n);
			}
		} else {
			CHECK((int32_t)cl.tprefix.size() == cl.ncols + 1 && cl.ntiles == cl.tprefix.back(), "s32 class %d: prefix", cl.cls);
		}
		tiles_total += cl.ntiles;
		int64_t pairs = 0;
		for (int32_t t = 0; t < cl.ntiles; t++) {
			const TileGeo g = tile_geo(pl, cl, t);
			CHECK(g.i_count > 0, "class %d tile %d: empty", cl.cls, t);
			if (is_pk && t >= cl.ntiles - cl.npart)
				CHECK(g.i_count == cl.part_rows[(size_t)(t - (cl.ntiles - cl.npart))], "class %d tile %d: partial rows %d vs %d", cl.cls, t,
				      g.i_count, cl.part_rows[(size_t)(t - (cl.ntiles - cl.npart))]);
			for (int h = 0; h < (g.dup ? 1 : 2); h++) {
				CHECK(g.j[h] >= j0 && g.j[h] <= j1, "class %d tile %d: column %d outside the range", cl.cls, t, g.j[h]);
				const int64_t lo = std::max(g.ia[h], g.i_begin), hi = std::min(g.ib[h], g.i_begin + g.i_count);
				if (hi > lo) {
					covered[(size_t)(g.j[h] - j0)] += hi - lo;
					pairs += hi - lo;
				}
			}
		}
		CHECK(pairs == cl.pairs, "class %d: tiles cover %" PRId64 " pairs, class claims %" PRId64, cl.cls, pairs, cl.pairs);
	}
	int64_t generic_pairs = 0;
	for (const auto &run : pl.generic) {
		CHECK(run.first >= start && run.first + run.second <= end && run.second > 0, "generic run outside the range");
		generic_pairs += run.second;
		int64_t p = run.first, left = run.second;
		while (left > 0) {
			const int32_t j = sa_column_of(p);
			const int64_t n = std::min<int64_t>(left, sa_tri(j + 1) - p);
			covered[(size_t)(j - j0)] += n;
			p += n;
			left -= n;
		}
	}
	int64_t sum = 0;
	for (int32_t j = j0; j <= j1; j++) {
		const int64_t tri = sa_tri(j);
		const int64_t want = std::min<int64_t>(j, end - tri) - std::max<int64_t>(0, start - tri);
		CHECK(covered[(size_t)(j - j0)] == std::max<int64_t>(want, 0), "column %d: %" PRId64 " pairs covered, %" PRId64 " in the range", j,
		      covered[(size_t)(j - j0)], want);
		sum += covered[(size_t)(j - j0)];
	}
	CHECK(sum == pl.count, "plan covers %" PRId64 " of %" PRId64 " pairs", sum, pl.count);

	/* bundles: every packed tile appears exactly once in exactly one launch list (per rank: its owner's) */
	const int nranks = std::max(pl.world, 1);
	std::vector<std::vector<uint8_t>> seen(pl.classes.size());
	for (size_t ci = 0; ci < pl.classes.size(); ci++)
		seen[ci].assign((size_t)pl.classes[ci].ntiles, 0);
	for (const auto &b : pl.bundles) {
		CHECK(b.args.size() == b.cls.size() && (int)b.ufirst.size() == nranks + 1 && (int)b.nlocal.size() == nranks, "bundle shape");
		CHECK(b.cls.size() <= ((size_t)1 << (32 - SA_PK_UTILE_BITS)), "bundle of %zu classes overflows the tile code", b.cls.size());
		CHECK(sa_pk_lds_bytes(g_method, b.g, b.kmax) <= 160 * 1024, "bundle K %d: LDS", b.kmax);
		for (int r = 0; r < nranks; r++) {
			CHECK(b.ufirst[(size_t)r + 1] - b.ufirst[(size_t)r] == 2 * (int64_t)b.nlocal[(size_t)r], "bundle: ufirst / nlocal disagree");
			CHECK(b.ufirst[(size_t)r + 1] <= (int64_t)b.ulist.size(), "bundle: ulist too short");
			for (int64_t u = b.ufirst[(size_t)r]; u < b.ufirst[(size_t)r + 1]; u += 2) {
				const uint32_t code = b.ulist[(size_t)u], pair = b.ulist[(size_t)u + 1];
				const uint32_t x = code >> SA_PK_UTILE_BITS, t = code & ((1u << SA_PK_UTILE_BITS) - 1);
				CHECK(x < b.cls.size(), "bundle: class index %u of %zu", x, b.cls.size());
				if (x >= b.cls.size())
					continue;
				const auto &cl = pl.classes[(size_t)b.cls[x]];
				CHECK((int32_t)t < cl.ntiles, "bundle: tile %u of %d", t, cl.ntiles);
				if ((int32_t)t >= cl.ntiles)
					continue;
				CHECK(pair < (uint32_t)((cl.ncols + 1) / 2), "bundle: pair %u", pair);
				const TileGeo g = tile_geo(pl, cl, (int32_t)t);
				CHECK(g.j[0] == cl.jlist[(size_t)2 * pair], "bundle: tile %u names pair %u but lies in the pair of column %d", t, pair, g.j[0]);
				CHECK(!seen[(size_t)b.cls[x]][t], "bundle: tile %u listed twice", t);
				seen[(size_t)b.cls[x]][t] = 1;
				if (pl.world >= 1)
					CHECK(cl.owner[t] == r, "bundle: tile %u in rank %d's list, owner %d", t, r, (int)cl.owner[t]);
				const auto &a = b.args[x];
				CHECK(a.k == sa_pk_decode(cl.cls).k && a.chunk == cl.chunk && a.ncols == cl.ncols && a.npart == cl.npart, "bundle args");
				CHECK(a.pk_base + a.delta <= 65535 && a.delta > 0, "bundle: base %d delta %d", a.pk_base, a.delta); /* one frame above BASE fits u16 */
			}
		}
	}
	for (size_t ci = 0; ci < pl.classes.size(); ci++)
		if (pl.classes[ci].cls >= SA_PK_CLASS0)
			for (int32_t t = 0; t < pl.classes[ci].ntiles; t++)
				CHECK(seen[ci][(size_t)t], "packed class %d tile %d in no launch list", pl.classes[ci].cls, t);

	if (pl.world >= 1) {
		/* dense shares: per rank the runs must not overlap and must stay inside share_elems */
		std::vector<std::vector<std::pair<int64_t, int64_t>>> runs((size_t)pl.world);
		for (const auto &cl : pl.classes) {
			CHECK((int32_t)cl.owner.size() == cl.ntiles && (int32_t)cl.doff.size() == cl.ntiles && (int)cl.rank_first.size() == pl.world + 1,
			      "share class %d: shape", cl.cls);
			CHECK((int32_t)cl.tlist.size() == cl.ntiles, "share class %d: tlist", cl.cls);
			for (int r = 0; r < pl.world; r++)
				for (int32_t x = cl.rank_first[(size_t)r]; x < cl.rank_first[(size_t)r + 1]; x++)
					CHECK(cl.owner[(size_t)cl.tlist[(size_t)x]] == r, "share class %d: tlist / owner", cl.cls);
			for (int32_t t = 0; t < cl.ntiles; t++) {
				const TileGeo g = tile_geo(pl, cl, t);
				const int r = cl.owner[(size_t)t];
				CHECK(r >= 0 && r < pl.world, "owner");
				runs[(size_t)r].emplace_back(cl.doff[(size_t)t], (int64_t)(g.dup ? 1 : 2) * SA_SHARE_PAD(g.i_count));
			}
		}
		for (int r = 0; r < pl.world; r++)
			for (const auto &gs : pl.generic_share[(size_t)r])
				runs[(size_t)r].emplace_back(gs.doff, SA_SHARE_PAD(gs.count));
		for (int r = 0; r < pl.world; r++) {
			auto &v = runs[(size_t)r];
			std::sort(v.begin(), v.end());
			int64_t at = 0;
			for (const auto &run : v) {
				CHECK(run.first >= at, "rank %d: share runs overlap at %" PRId64, r, run.first);
				CHECK(run.first % 8 == 0, "rank %d: run not 16-byte aligned in int16", r);
				at = run.first + run.second;
			}
			CHECK(at <= pl.share_elems, "rank %d: share of %" PRId64 " elements, share_elems %" PRId64, r, at, pl.share_elems);
		}
		/* placement: the segments write every element of the range exactly once (element = dst + row); counted always,
		 * checked element by element with a bitmap when the range is small enough for one (<= 2^28 pairs) */
		const bool exact = pl.count <= ((int64_t)1 << 28);
		std::vector<uint8_t> bits(exact ? (size_t)((pl.count + 7) / 8) : 0, 0);
		int64_t placed_sum = 0;
		std::map<std::tuple<int, int, int32_t>, std::vector<int32_t>> rowmaps;
		auto place = [&](const SaHostSeg &sg, int64_t r) {
			const int64_t e = sg.dst + r;
			CHECK(e >= 0 && e < pl.count, "segment writes element %" PRId64 " outside the range", e);
			if (exact && e >= 0 && e < pl.count) {
				CHECK(!(bits[(size_t)(e >> 3)] & (1u << (e & 7))), "element %" PRId64 " placed twice", e);
				bits[(size_t)(e >> 3)] |= (uint8_t)(1u << (e & 7));
			}
			placed_sum++;
		};
		for (const auto &sg : pl.segs) {
			CHECK(sg.src >= 0 && sg.src + sg.count <= (int64_t)pl.world * pl.share_elems, "segment source outside the gathered shares");
			CHECK(sg.count > 0 && sg.pos0 >= 0 && sg.pos0 + sg.count <= std::max<int64_t>(in.num, sg.count), "segment shape");
			if (sg.map_kind) {
				CHECK(sa_arranged_exists(in.num, sg.key), "segment names an arranged copy that does not exist");
				CHECK(sg.pos0 / sg.key.block == (sg.pos0 + sg.count - 1) / sg.key.block, "arranged segment crosses a block");
				CHECK(sg.pos0 + sg.count <= in.num / sg.key.block * sg.key.block, "arranged segment past the last full block");
				CHECK((sg.flags & 1) == (sg.map_kind == 2), "flags / map kind");
				if (sg.flags & 1)
					CHECK(sg.count == sg.key.block && sg.pos0 % sg.key.block == 0, "own-block segment is not a whole block");
				auto key = std::make_tuple(sg.key.ng, sg.key.ch, sg.key.block);
				auto it = rowmaps.find(key);
				if (it == rowmaps.end()) {
					std::vector<int32_t> rm;
					sa_arrange_rows(in.meta, in.num, sg.key, rm);
					it = rowmaps.emplace(key, std::move(rm)).first;
				}
				for (int32_t q = sg.pos0; q < sg.pos0 + sg.count; q++) {
					const int32_t r = it->second[(size_t)q];
					if (r >= sg.ia && r < sg.ib)
						place(sg, r);
				}
			} else {
				CHECK(sg.flags == 0, "flags without a map");
				for (int64_t r = std::max<int64_t>(sg.ia, sg.pos0); r < std::min<int64_t>(sg.ib, (int64_t)sg.pos0 + sg.count); r++)
					place(sg, r);
			}
		}
		CHECK(placed_sum == pl.count, "placement writes %" PRId64 " of %" PRId64 " elements", placed_sum, pl.count);
	}
	printf("plan [%" PRId64 ",+%" PRId64 ") world %d host %d: %zu classes, %zu bundles, %" PRId64 " tiles, chunk %d/%d/%d, %zu segs, share %" PRId64
	       ", generic %" PRId64 " pairs\n",
	       pl.start, pl.count, pl.world, (int)pl.share_host, pl.classes.size(), pl.bundles.size(), tiles_total, pl.chunk, pl.chunk_pk,
	       pl.chunk_pk_small, pl.segs.size(), pl.share_elems, generic_pairs);
}

static void check_arranged(const SaPlanInputs &in, const SaHostPlan &pl)
{
	std::vector<SaArrKey> keys;
	for (const auto &b : pl.bundles)
		for (const auto &a : b.args)
			for (const auto &k : a.lv)
				if (k.block && std::find(keys.begin(), keys.end(), k) == keys.end())
					keys.push_back(k);
	for (const auto &key : keys) {
		std::vector<int32_t> rm;
		sa_arrange_rows(in.meta, in.num, key, rm);
		CHECK((int32_t)rm.size() == in.num, "rowmap size");
		std::vector<uint8_t> hit((size_t)in.num, 0);
		for (int32_t p = 0; p < in.num; p++) {
			const int32_t r = rm[(size_t)p];
			CHECK(r >= 0 && r < in.num && !hit[(size_t)r], "arranged copy (%d,%d,%d): not a permutation at %d", key.ng, key.ch, key.block, p);
			if (r >= 0 && r < in.num)
				hit[(size_t)r] = 1;
			CHECK(p / key.block == r / key.block || p >= in.num / key.block * key.block, "arranged copy: row %d left its block", r);
		}
	}
}

int main(int argc, char **argv)
{
	if (argc < 12 || (argc - 8) % 4 != 0) {
		fprintf(stderr, "usage: plan_check lens.i32 method matrix gap_pen gap_open gap_ext CUs (start count world share_host)...\n");
		return 2;
	}
	FILE *f = fopen(argv[1], "rb");
	if (!f) {
		perror(argv[1]);
		return 2;
	}
	fseek(f, 0, SEEK_END);
	const long bytes = ftell(f);
	fseek(f, 0, SEEK_SET);
	std::vector<int32_t> lens((size_t)bytes / 4);
	if (fread(lens.data(), 4, lens.size(), f) != lens.size())
		return 2;
	fclose(f);
	const int32_t num = (int32_t)lens.size();
	std::vector<sa_meta> meta((size_t)num);
	int32_t max_len = 0, min_len = INT32_MAX;
	int64_t off = 0;
	for (int32_t k = 0; k < num; k++) {
		meta[(size_t)k] = sa_meta{ (int32_t)off, lens[(size_t)k] };
		off += lens[(size_t)k] + 1;
		max_len = std::max(max_len, lens[(size_t)k]);
		min_len = std::min(min_len, lens[(size_t)k]);
	}
	sa_scoring sc{};
	sc.method = sa_method_parse(argv[2]);
	if (sc.method < 0 || sa_matrix_load(argv[3], sc.lut, sc.sub)) {
		fprintf(stderr, "%s\n", sa_last_error());
		return 2;
	}
	g_method = sc.method;
	sc.gap_pen = -atoi(argv[4]);
	sc.gap_opn = -atoi(argv[5]);
	sc.gap_ext = -atoi(argv[6]);
	const int cus = atoi(argv[7]);
	const SaKernelLimits L = sa_kernel_limits(sc, max_len, min_len, false, false, false);
	SaPlanInputs in;
	in.num = num;
	in.meta = meta.data();
	in.min_len = min_len;
	in.method = sc.method;
	in.gap_ext = sc.gap_ext;
	in.sys_ok = L.sys_ok;
	in.pk_kmax = L.pk_kmax;
	in.pk16_kmax = L.pk16_kmax;
	in.pk16_f16_kmax = L.pk16_f16_kmax;
	in.pk_chunk_cap = L.pk_chunk_cap;
	in.pk_q = L.pk_q;
	in.pk_floor = L.pk_floor;
	in.pk_gain = L.pk_gain;
	in.pk_slack = L.pk_slack;
	in.persistent_wgs = cus * 32;
	printf("store: %d sequences, lengths %d..%d; limits: sys_ok %d pk_kmax %d pk16_kmax %d (f16 up to %d) chunk cap %d\n", num, min_len,
	       max_len, (int)L.sys_ok, L.pk_kmax, L.pk16_kmax, L.pk16_f16_kmax, L.pk_chunk_cap);
	const int64_t pairs = (int64_t)num * (num - 1) / 2;
	for (int a = 8; a + 3 < argc; a += 4) {
		const int64_t start = atoll(argv[a]);
		int64_t count = atoll(argv[a + 1]);
		if (count < 0)
			count = pairs - start;
		count = std::min(count, pairs - start);
		const int world = atoi(argv[a + 2]);
		const bool share_host = atoi(argv[a + 3]) != 0;
		SaHostPlan pl;
		if (!sa_plan_host(in, start, count, world, share_host, pl)) {
			printf("plan [%" PRId64 ",+%" PRId64 ") world %d: refused: %s\n", start, count, world, sa_last_error());
			continue;
		}
		check_plan(in, pl);
		check_arranged(in, pl);
	}
	if (g_fail) {
		fprintf(stderr, "%d checks failed\n", g_fail);
		return 1;
	}
	return 0;
}
