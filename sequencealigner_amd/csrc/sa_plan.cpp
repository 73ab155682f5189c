/* sa_plan.cpp -- launch planning of a packed pair range: pure host arithmetic (see sa_plan.h).
 * No HIP call and no device pointer in this file; tests/plan_host/plan_check.cpp links it under ASan / UBSan. */
#include "sa_plan.h"
#include "sa_guard.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

int32_t sa_column_of(int64_t p)
{
	int64_t j = (int64_t)((1.0 + std::sqrt(1.0 + 8.0 * (double)p)) * 0.5);
	while (j * (j - 1) / 2 > p)
		--j;
	while ((j + 1) * j / 2 <= p)
		++j;
	return (int32_t)j;
}

SaPairPlan::SaPairPlan(const sa_meta *m, int32_t n) : num(n), pairs((int64_t)n * (n - 1) / 2), meta(m)
{
	len_prefix.assign((size_t)n + 1, 0);
	cell_prefix.assign((size_t)n + 1, 0);
	for (int32_t k = 0; k < n; k++) {
		len_prefix[(size_t)k + 1] = len_prefix[(size_t)k] + m[k].len;
		cell_prefix[(size_t)k + 1] = cell_prefix[(size_t)k] + (int64_t)m[k].len * len_prefix[(size_t)k];
	}
}

int64_t SaPairPlan::cells_before(int64_t p) const
{
	if (p <= 0)
		return 0;
	if (p >= pairs)
		return cell_prefix[(size_t)num];
	const int32_t j = sa_column_of(p);
	const int64_t i = p - (int64_t)j * (j - 1) / 2;
	return cell_prefix[(size_t)j] + (int64_t)meta[j].len * len_prefix[(size_t)i];
}

int sa_systolic_class_for(int32_t n)
{
	for (int c = 0; c < SA_SYS_NCLASSES; c++)
		if (SA_SYS_CLASSES[c].G * SA_SYS_CLASSES[c].K >= n)
			return c;
	return SA_SYS_CLASS_LONG;
}

SaPkCls sa_pk_decode(int cls)
{
	SaPkCls r;
	r.small = cls >= SA_PK_CLASSES_END;
	const int c = r.small ? cls - SA_PK_SMALL : cls;
	r.g = c >= SA_PK16_CLASS0 ? 16 : 8;
	r.k = c - (r.g == 16 ? SA_PK16_CLASS0 : SA_PK_CLASS0);
	return r;
}

int32_t sa_pk_delta(const SaPlanInputs &in, int g, int k) { return (int32_t)(in.pk_gain * g * k + in.pk_slack); }

int32_t sa_pk_base(const SaPlanInputs &in, int g, int k)
{
	/* (Gotoh: values reach BASE + 3q; SW: the lanes start up to G |e| below the baseline) */
	return sa_pk_live(in.min_len, g) * sa_pk_delta(in, g, k) + in.pk_floor + 4 * std::abs(in.pk_q) + 4 +
	       (in.method == SA_METHOD_SW ? g * std::abs(in.gap_ext) : 0);
}

bool sa_arranged_exists(int32_t num, const SaArrKey &key)
{
	const int64_t wave_rows = (int64_t)key.ng * key.ch;
	return key.block > 0 && wave_rows > 0 && key.block % (wave_rows * SA_PK_WPB) == 0 && key.block <= num;
}

int sa_pk_arranged_keys(const SaPlanInputs &in, int pk_g, int pk_k, int32_t chunk_pk, bool host_out, SaArrKey (&lv)[SA_PK_SORT_LEVELS])
{
	for (auto &x : lv)
		x = SaArrKey{};
	if (in.no_sort || chunk_pk <= 0)
		return 0;
	const int ng = 64 / pk_g;
	const int32_t rows = sa_pk_wpb(in.method, pk_g, pk_k) * ng * chunk_pk;
	int nl = 0;
	for (int l = 0; l < SA_PK_SORT_LEVELS; l++) {
		const int32_t block = SA_PK_SORT_ROWS >> l;
		/* (a tile of SA_PK_ROWS_OWN_BLOCK rows has enough equal lengths of its own, and storing in row order keeps
		 * the HBM write traffic at the algorithmic 4 bytes per pair) */
		if (host_out || rows >= SA_PK_ROWS_OWN_BLOCK || block <= rows || block % rows != 0)
			continue;
		const SaArrKey key{ ng, chunk_pk, block };
		if (sa_arranged_exists(in.num, key))
			lv[nl++] = key;
	}
	if (nl < SA_PK_SORT_LEVELS) { /* the tile itself as a block: its scores leave in row order */
		const SaArrKey key{ ng, chunk_pk, rows };
		if (sa_arranged_exists(in.num, key))
			lv[nl++] = key;
	}
	return nl;
}

/* A workgroup-tile streams SA_PK_WPB waves x ng streams x ch sequences.  What a terminator costs a wave is the event
 * step it causes (the frame shift of every register, the capture of a score pair), and the step is shared by all
 * streams of the wave whose terminators pass at the same stream position; streams of different lengths also leave
 * bubbles at the end of a tile.  So the rows of the matrix are cut into aligned blocks of `block` sequences (a multiple
 * of the tile's rows) and every block is re-ordered for this tile shape:
 *   - sequences of equal length are taken ng at a time: a PURE round, one sequence for each stream of a wave;
 *   - what is left over (< ng per length) is sorted by length and cut into MIXED rounds of ng neighbours;
 *   - the rounds, longest first and the mixed ones last, are dealt over the block's wave slots boustrophedon, one per
 *     slot and pass: every wave gets ch rounds of nearly the same total length, its pure rounds first.
 * As long as a wave is in its pure rounds all its streams are in step.  Rows past the last full block keep their order. */
void sa_arrange_rows(const sa_meta *meta, int32_t num, const SaArrKey &key, std::vector<int32_t> &rowmap)
{
	rowmap.resize((size_t)num);
	for (int32_t i = 0; i < num; i++)
		rowmap[(size_t)i] = i;
	if (!sa_arranged_exists(num, key))
		return;
	const int ng = key.ng, ch = key.ch;
	const int32_t block = key.block;
	const int32_t wave_rows = ng * ch;
	const int32_t slots = block / wave_rows;
	std::vector<int32_t> idx((size_t)block), rounds, rest;
	for (int32_t b0 = 0; b0 + block <= num; b0 += block) {
		for (int32_t k = 0; k < block; k++)
			idx[(size_t)k] = b0 + k;
		std::stable_sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b) { return meta[a].len > meta[b].len; });
		rounds.clear();
		rest.clear();
		for (int32_t k = 0; k < block;) {
			int32_t e = k;
			while (e < block && meta[idx[(size_t)e]].len == meta[idx[(size_t)k]].len)
				e++;
			const int32_t pure = (e - k) / ng * ng;
			rounds.insert(rounds.end(), idx.begin() + k, idx.begin() + k + pure);
			rest.insert(rest.end(), idx.begin() + k + pure, idx.begin() + e);
			k = e;
		}
		rounds.insert(rounds.end(), rest.begin(), rest.end()); /* (block and the pure part are multiples of ng) */
		for (int32_t r = 0; r < block / ng; r++) {
			const int32_t pass = r / slots, w = r % slots;
			const int32_t slot = (pass & 1) ? slots - 1 - w : w;
			for (int g = 0; g < ng; g++) {
				/* mixed rounds alternate their direction, so that the streams of a wave even out */
				const int gg = (pass & 1) ? ng - 1 - g : g;
				rowmap[(size_t)(b0 + slot * wave_rows + gg * ch + pass)] = rounds[(size_t)(r * ng + g)];
			}
		}
	}
}

static const char *const TOO_LARGE = "packed range too large for one launch; split it into smaller ranges";

bool sa_plan_host(const SaPlanInputs &in, int64_t start, int64_t count, int world, bool share_host, SaHostPlan &plan)
{
	plan = SaHostPlan();
	const int32_t num = in.num;
	const int64_t all_pairs = (int64_t)num * (num - 1) / 2;
	if (num < 2 || !in.meta || start < 0 || count <= 0 || start > all_pairs - count || world < 0 || world > 1024) {
		sa_set_error("plan: bad range [%lld,+%lld) of %lld pairs or world %d", (long long)start, (long long)count,
			     (long long)all_pairs, world);
		return false;
	}
	const int64_t end = start + count;
	std::vector<std::vector<int32_t>> jl((size_t)SA_PLAN_NCLASSES), tp((size_t)SA_PLAN_NCLASSES);
	std::vector<int64_t> cpairs((size_t)SA_PLAN_NCLASSES, 0), ccells((size_t)SA_PLAN_NCLASSES, 0);
	std::vector<std::vector<std::pair<int32_t, int32_t>>> rows_of((size_t)SA_PLAN_NCLASSES); /* packed classes: [ia, ib) per column */
	std::vector<int64_t> lenpre((size_t)num + 1, 0);
	for (int32_t k = 0; k < num; k++)
		lenpre[(size_t)k + 1] = lenpre[(size_t)k] + in.meta[k].len;
	/* Tile granularity: a wave-tile streams `chunk` sequences per lane group.  The longest streams amortize the per-tile
	 * setup best; small ranges (one rank's share at 8 GPUs, a super-chunk of an overlapped schedule) get shorter streams
	 * so that there are still several tiles per wave slot. */
	{
		const int64_t want_tiles = (int64_t)in.persistent_wgs * 6;
		const int64_t mine = world > 1 ? count / world : count; /* pairs one rank runs */
		int32_t chunk = SA_SYS_CHUNK;
		while (chunk > 8 && mine / (4 * chunk) < want_tiles)
			chunk >>= 1;
		if (in.env_chunk)
			chunk = in.env_chunk;
		plan.chunk = chunk;
		/* packed classes: a workgroup-tile covers 2 columns x SA_PK_WPB * 8 streams of `chunk` sequences and ~4
		 * workgroups are resident per CU: keep >= 5 tiles per resident workgroup so that a small range still drains evenly
		 * (measured, cfg 2: quarter tiles cost 5 % more than full ones, half tiles 1 %) */
		const int64_t want_pk = (int64_t)in.persistent_wgs * 5 / 8; /* 32 x CUs x 5 / 8 = 5 x (4 workgroups per CU) */
		int32_t cpk = SA_SYS_CHUNK;
		while (cpk > 4 && mine / ((int64_t)2 * SA_PK_WPB * 8 * cpk) < want_pk)
			cpk >>= 1;
		if (in.env_chunk)
			cpk = in.env_chunk;
		cpk = std::min(cpk, in.pk_chunk_cap);
		plan.chunk_pk = cpk;
		/* Two tile sizes for a launch that gives a workgroup slot fewer than `small_below` tiles (one rank's share of a
		 * multi-GPU run; a super-chunk): when the tiles run out the slots finish their last ones over a whole tile's
		 * duration, and only work in small units can fill that triangle.  So the bulk runs in tiles as large as leave
		 * >= 2.5 per slot, and the lowest columns of the range -- a fifth of its pairs -- in tiles a quarter of that size,
		 * which the launch order puts last (small tiles cost more per row, so not everywhere). */
		const int64_t slots = (int64_t)in.persistent_wgs / 8;
		if (!in.env_chunk && !in.one_tile_size && mine / ((int64_t)2 * SA_PK_WPB * 8 * cpk) < (int64_t)in.small_below * slots) {
			int32_t big = SA_SYS_CHUNK;
			while (big > 16 && 2 * mine / ((int64_t)2 * SA_PK_WPB * 8 * big) < 5 * slots)
				big >>= 1;
			big = std::min(std::max(big, cpk), in.pk_chunk_cap);
			if (big >= 16) {
				plan.chunk_pk = big;
				plan.chunk_pk_small = std::max(4, big / std::max(2, in.small_div));
				const int64_t jlo = sa_column_of(start), jhi = sa_column_of(end - 1) + 1;
				int64_t lo = jlo, hi = jhi; /* smallest column with >= a fifth of the range's pairs below it */
				while (lo < hi) {
					const int64_t mid = (lo + hi) / 2;
					if (mid * (mid - 1) / 2 - start >= count / std::max(2, in.small_frac))
						hi = mid;
					else
						lo = mid + 1;
				}
				plan.j_small = (int32_t)lo;
			}
		}
	}
	const int32_t j0 = sa_column_of(start), j1 = sa_column_of(end - 1);
	for (int32_t j = j0; j <= j1; j++) {
		const int64_t tri = (int64_t)j * (j - 1) / 2;
		const int64_t ia = std::max<int64_t>(0, start - tri), ib = std::min<int64_t>(j, end - tri);
		if (ib <= ia)
			continue;
		const int32_t n = in.meta[j].len;
		const int k8 = (n + 7) / 8, k16 = (n + 15) / 16;
		if (k8 <= in.pk_kmax || (k16 >= SA_PK_K16_MIN && k16 <= in.pk16_kmax)) {
			/* packed-u16 class K = ceil(n / 8) (8-lane groups) or ceil(n / 16) (16-lane groups): tiles are counted per
			 * column PAIR below */
			const size_t pc = (size_t)(k8 <= in.pk_kmax ? SA_PK_CLASS0 + k8 : SA_PK16_CLASS0 + k16) + (j < plan.j_small ? SA_PK_SMALL : 0);
			jl[pc].push_back(j);
			rows_of[pc].emplace_back((int32_t)ia, (int32_t)ib);
			cpairs[pc] += ib - ia;
			ccells[pc] += (int64_t)n * (lenpre[(size_t)ib] - lenpre[(size_t)ia]);
			continue;
		}
		const int cls = in.sys_ok ? sa_systolic_class_for(n) : -1;
		if (cls < 0) {
			if (!plan.generic.empty() && plan.generic.back().first + plan.generic.back().second == tri + ia)
				plan.generic.back().second += ib - ia;
			else
				plan.generic.emplace_back(tri + ia, ib - ia);
			continue;
		}
		/* strip-mined tiles stream one group of at most 16 sequences (their scratch lines are per position) */
		const int rows = cls == SA_SYS_CLASS_LONG ? SA_SYS_WPB(64, true) * std::min(plan.chunk, 16)
							   : SA_SYS_WPB(SA_SYS_CLASSES[cls].G, false) * (64 / SA_SYS_CLASSES[cls].G) * plan.chunk;
		if (tp[(size_t)cls].empty())
			tp[(size_t)cls].push_back(0);
		const int64_t tiles = (ib - ia + rows - 1) / rows;
		if ((int64_t)tp[(size_t)cls].back() + tiles > INT32_MAX) {
			sa_set_error("%s", TOO_LARGE);
			return false;
		}
		jl[(size_t)cls].push_back(j);
		tp[(size_t)cls].push_back(tp[(size_t)cls].back() + (int32_t)tiles);
		cpairs[(size_t)cls] += ib - ia;
		ccells[(size_t)cls] += (int64_t)n * (lenpre[(size_t)ib] - lenpre[(size_t)ia]);
	}
	/* packed classes: consecutive columns of a class share a tile (sa_systolic_pk.inc): rows = the union of their row
	 * ranges, SA_PK_WPB * 64/G streams of `chunk` sequences per workgroup-tile.  The prefix counts the full tiles; the
	 * partial last tiles follow them in the tile numbering, largest first (their pairs are appended to the prefix). */
	std::vector<int32_t> nparts((size_t)SA_PLAN_NCLASSES, 0);
	std::vector<std::vector<int32_t>> part_rows((size_t)SA_PLAN_NCLASSES);
	for (int cls = SA_PK_CLASS0; cls < SA_PLAN_NCLASSES; cls++) {
		const auto &rw = rows_of[(size_t)cls];
		if (rw.empty())
			continue;
		const SaPkCls pc = sa_pk_decode(cls);
		const int64_t rows = (int64_t)sa_pk_wpb(in.method, pc.g, pc.k) * (64 / pc.g) * (pc.small ? plan.chunk_pk_small : plan.chunk_pk);
		tp[(size_t)cls].push_back(0);
		std::vector<std::pair<int32_t, int32_t>> parts; /* (rows, pair) */
		for (size_t c = 0; c < rw.size(); c += 2) {
			const auto &a = rw[c], &b = rw[c + 1 < rw.size() ? c + 1 : c];
			const int64_t span = std::max(a.second, b.second) - std::min(a.first, b.first);
			const int64_t tiles = span / rows;
			if ((int64_t)tp[(size_t)cls].back() + tiles + (int64_t)parts.size() + 1 > ((int64_t)1 << SA_PK_UTILE_BITS)) {
				sa_set_error("%s", TOO_LARGE);
				return false;
			}
			tp[(size_t)cls].push_back(tp[(size_t)cls].back() + (int32_t)tiles);
			if (span % rows)
				parts.emplace_back((int32_t)(span % rows), (int32_t)(c / 2));
		}
		std::stable_sort(parts.begin(), parts.end(), [](const auto &x, const auto &y) { return x.first > y.first; });
		for (const auto &pt : parts) {
			tp[(size_t)cls].push_back(pt.second);
			part_rows[(size_t)cls].push_back(pt.first);
		}
		nparts[(size_t)cls] = (int32_t)parts.size();
	}
	for (int cls = 0; cls < SA_PLAN_NCLASSES; cls++) {
		if (jl[(size_t)cls].empty())
			continue;
		SaHostClass cl;
		cl.cls = cls;
		cl.ncols = (int32_t)jl[(size_t)cls].size();
		cl.npart = nparts[(size_t)cls];
		cl.ntiles = cl.npart ? tp[(size_t)cls][tp[(size_t)cls].size() - 1 - (size_t)cl.npart] + cl.npart : tp[(size_t)cls].back();
		cl.pairs = cpairs[(size_t)cls];
		cl.cells = ccells[(size_t)cls];
		cl.part_rows = std::move(part_rows[(size_t)cls]);
		cl.chunk = cls >= SA_PK_CLASS0 ? (sa_pk_decode(cls).small ? plan.chunk_pk_small : plan.chunk_pk) : 0;
		if (cls >= SA_PK_CLASS0 && cl.ntiles > (1 << SA_PK_UTILE_BITS)) {
			sa_set_error("%s", TOO_LARGE);
			return false;
		}
		cl.jlist = std::move(jl[(size_t)cls]);
		cl.tprefix = std::move(tp[(size_t)cls]);
		plan.classes.push_back(std::move(cl));
	}
	/* order of the classes (s32 launches; the share plans' dealing order): the class with the most DP work first */
	std::stable_sort(plan.classes.begin(), plan.classes.end(),
			 [](const SaHostClass &x, const SaHostClass &y) { return x.cells > y.cells; });
	plan.start = start;
	plan.count = count;
	plan.world = world;
	plan.share_host = share_host;
	/* ---- share plan: deal the tiles of every class over the ranks, lay out the dense shares, list the placement ---- */
	if (world >= 1) {
		/* One list for the whole job, the same on every rank: classes in order, inside a class the launch's own tile
		 * order (full tiles, then the partial ones by decreasing size).  A tile goes to the rank with the least
		 * accumulated work so far (cost = row residues x per-step instruction weight of the class): the shares end
		 * within one small partial tile of each other, every rank keeps full-size tiles and whole arranged blocks. */
		std::vector<int64_t> load((size_t)world, 0), fill((size_t)world, 0);
		struct Geo {
			int32_t owner;
			int32_t j[2], ia[2], ib[2], i_begin, i_count;
			bool dup, own; /* own: an arranged tile that is its whole block (its rows are a permutation of its positions) */
			SaArrKey key;  /* block = 0: store order */
			int64_t doff;
		};
		std::vector<Geo> geo;
		for (size_t ci = 0; ci < plan.classes.size(); ci++) {
			auto &cl = plan.classes[ci];
			const int cls = cl.cls;
			const bool is_pk = cls >= SA_PK_CLASS0;
			const int pk_g = is_pk ? sa_pk_decode(cls).g : 8;
			const int G = is_pk ? pk_g : cls == SA_SYS_CLASS_LONG ? 64 : SA_SYS_CLASSES[cls].G;
			const int K = is_pk ? sa_pk_decode(cls).k : cls == SA_SYS_CLASS_LONG ? 16 : SA_SYS_CLASSES[cls].K;
			const int64_t weight = (int64_t)(K + 6) * (G / 8);
			const auto &J = cl.jlist;
			const auto &T = cl.tprefix;
			std::vector<Geo> tiles((size_t)cl.ntiles);
			if (is_pk) {
				SaArrKey lv[SA_PK_SORT_LEVELS];
				sa_pk_arranged_keys(in, pk_g, K, cl.chunk, share_host, lv);
				const int32_t lvrows[SA_PK_SORT_LEVELS] = { lv[0].block, lv[1].block, lv[2].block, lv[3].block };
				const int32_t rows = sa_pk_wpb(in.method, pk_g, K) * (64 / pk_g) * cl.chunk;
				const auto &rw = rows_of[(size_t)cls];
				const int32_t npairs = (cl.ncols + 1) / 2, nfull = T[(size_t)npairs];
				for (int32_t t = 0; t < cl.ntiles; t++) {
					int32_t lo;
					if (t < nfull)
						lo = (int32_t)(std::upper_bound(T.begin(), T.begin() + npairs + 1, t) - T.begin()) - 1;
					else
						lo = T[(size_t)(npairs + 1 + (t - nfull))];
					const size_t c0 = (size_t)2 * (size_t)lo, c1 = c0 + 1 < rw.size() ? c0 + 1 : c0;
					Geo &g = tiles[(size_t)t];
					g.dup = c1 == c0;
					g.j[0] = J[c0], g.j[1] = J[c1];
					g.ia[0] = rw[c0].first, g.ib[0] = rw[c0].second;
					g.ia[1] = rw[c1].first, g.ib[1] = rw[c1].second;
					const int32_t ra = std::min(g.ia[0], g.ia[1]), rb = std::max(g.ib[0], g.ib[1]);
					const int32_t chunk = t < nfull ? t - T[(size_t)lo] : T[(size_t)lo + 1] - T[(size_t)lo];
					g.i_begin = ra + chunk * rows;
					g.i_count = std::min(rows, rb - g.i_begin);
					const int l = sa_pk_pick_level(lvrows, ra, rb, g.i_begin, rows);
					g.key = l >= 0 ? lv[l] : SaArrKey{};
					g.own = l >= 0 && lv[l].block == g.i_count;
				}
			} else {
				const int rows = cls == SA_SYS_CLASS_LONG ? SA_SYS_WPB(64, true) * std::min(plan.chunk, 16)
									   : SA_SYS_WPB(G, false) * (64 / G) * plan.chunk;
				for (int32_t k = 0; k < cl.ncols; k++) {
					const int32_t j = J[(size_t)k];
					const int64_t tri = (int64_t)j * (j - 1) / 2;
					const int32_t ia = (int32_t)std::max<int64_t>(0, start - tri), ib = (int32_t)std::min<int64_t>(j, end - tri);
					for (int32_t t = T[(size_t)k]; t < T[(size_t)k + 1]; t++) {
						Geo &g = tiles[(size_t)t];
						g.dup = true; /* one column, one run */
						g.j[0] = g.j[1] = j;
						g.ia[0] = g.ia[1] = ia, g.ib[0] = g.ib[1] = ib;
						g.i_begin = ia + (t - T[(size_t)k]) * rows;
						g.i_count = std::min(rows, ib - g.i_begin);
						g.key = SaArrKey{};
						g.own = false;
					}
				}
			}
			std::vector<std::vector<int32_t>> mine((size_t)world);
			cl.doff.assign((size_t)cl.ntiles, 0);
			cl.rank_pairs.assign((size_t)world, 0);
			cl.rank_cells.assign((size_t)world, 0);
			cl.owner.assign((size_t)cl.ntiles, 0);
			for (int32_t t = 0; t < cl.ntiles; t++) {
				Geo &g = tiles[(size_t)t];
				/* (arranged tiles hold a permutation of their block's rows: the residue count of the position range is that
				 * of the rows only when the tile is its whole block -- close enough for a load estimate) */
				const int64_t res = lenpre[(size_t)(g.i_begin + g.i_count)] - lenpre[(size_t)g.i_begin] + g.i_count;
				int r = 0;
				for (int q = 1; q < world; q++)
					if (load[(size_t)q] < load[(size_t)r])
						r = q;
				load[(size_t)r] += res * weight;
				g.owner = r;
				cl.owner[(size_t)t] = (int16_t)r;
				g.doff = fill[(size_t)r];
				cl.doff[(size_t)t] = g.doff;
				fill[(size_t)r] += (int64_t)(g.dup ? 1 : 2) * SA_SHARE_PAD(g.i_count);
				mine[(size_t)r].push_back(t);
				for (int h = 0; h < (g.dup ? 1 : 2); h++) {
					const int64_t lo_i = std::max(g.ia[h], g.i_begin), hi_i = std::min(g.ib[h], g.i_begin + g.i_count);
					if (hi_i > lo_i) {
						cl.rank_pairs[(size_t)r] += hi_i - lo_i;
						cl.rank_cells[(size_t)r] += (int64_t)in.meta[g.j[h]].len * (lenpre[(size_t)hi_i] - lenpre[(size_t)lo_i]);
					}
				}
				geo.push_back(g);
			}
			cl.rank_first.assign((size_t)world + 1, 0);
			for (int r = 0; r < world; r++) {
				cl.tlist.insert(cl.tlist.end(), mine[(size_t)r].begin(), mine[(size_t)r].end());
				cl.rank_first[(size_t)r + 1] = (int32_t)cl.tlist.size();
			}
		}
		/* what no systolic class covers: every run of the pair-per-wave kernels is cut into `world` equal pieces */
		plan.generic_share.assign((size_t)world, {});
		for (const auto &run : plan.generic) {
			const int64_t per = (run.second + world - 1) / world;
			for (int r = 0; r < world; r++) {
				const int64_t lo_p = std::min(run.second, (int64_t)r * per), hi_p = std::min(run.second, lo_p + per);
				if (hi_p <= lo_p)
					continue;
				plan.generic_share[(size_t)r].push_back({ run.first + lo_p, hi_p - lo_p, fill[(size_t)r] });
				fill[(size_t)r] += SA_SHARE_PAD(hi_p - lo_p);
			}
		}
		plan.share_elems = std::max<int64_t>(8, *std::max_element(fill.begin(), fill.end()));
		/* placement: one segment per run of a tile, pieces of the generic sub-runs */
		for (const Geo &g : geo)
			for (int h = 0; h < (g.dup ? 1 : 2); h++) {
				SaHostSeg sg{};
				sg.src = (int64_t)g.owner * plan.share_elems + g.doff + (int64_t)h * SA_SHARE_PAD(g.i_count);
				sg.dst = (int64_t)g.j[h] * (g.j[h] - 1) / 2 - start;
				sg.map_kind = g.key.block ? (g.own ? 2 : 1) : 0;
				sg.key = g.key;
				sg.count = g.i_count;
				sg.pos0 = g.i_begin;
				sg.ia = g.ia[h];
				sg.ib = g.ib[h];
				sg.flags = g.own ? 1 : 0;
				plan.segs.push_back(sg);
			}
		for (int r = 0; r < world; r++)
			for (const auto &gs : plan.generic_share[(size_t)r])
				for (int64_t o = 0; o < gs.count; o += 8192) {
					SaHostSeg sg{};
					sg.src = (int64_t)r * plan.share_elems + gs.doff + o;
					sg.dst = gs.start + o - start;
					sg.map_kind = 0;
					sg.count = (int32_t)std::min<int64_t>(8192, gs.count - o);
					sg.pos0 = 0;
					sg.ia = 0;
					sg.ib = sg.count;
					plan.segs.push_back(sg);
				}
		if (plan.segs.size() > (size_t)INT32_MAX) {
			sa_set_error("%s", TOO_LARGE);
			return false;
		}
	}
	/* ---- packed classes -> bundle launches ---- */
	const int nranks = std::max(world, 1);
	std::vector<int> order; /* packed classes by decreasing K inside their bundle: the launch ends on its cheapest tiles */
	for (size_t ci = 0; ci < plan.classes.size(); ci++)
		if (plan.classes[ci].cls >= SA_PK_CLASS0)
			order.push_back((int)ci);
	/* (the small-tile copies of the classes sort behind all the others: their full tiles end the full tiles) */
	std::sort(order.begin(), order.end(), [&](int x, int y) { return plan.classes[(size_t)x].cls > plan.classes[(size_t)y].cls; });
	std::stable_partition(order.begin(), order.end(), [&](int x) { return !sa_pk_decode(plan.classes[(size_t)x].cls).small; });
	for (int ci : order) {
		const auto &cl = plan.classes[(size_t)ci];
		const int g = sa_pk_decode(cl.cls).g;
		const int k = sa_pk_decode(cl.cls).k;
		const int klo = sa_pk_bundle_klo(g, k);
		const int f16 = g == 8 || k <= in.pk16_f16_kmax ? 1 : 0;
		SaHostBundle *b = nullptr;
		for (auto &x : plan.bundles)
			if (x.g == g && x.klo == klo && x.f16 == f16)
				b = &x;
		if (!b) {
			plan.bundles.emplace_back();
			b = &plan.bundles.back();
			b->g = g, b->klo = klo, b->f16 = f16, b->kmax = k;
		}
		b->kmax = std::max(b->kmax, k);
		b->cls.push_back(ci);
	}
	for (auto &b : plan.bundles) {
		b.args.resize(b.cls.size());
		for (size_t x = 0; x < b.cls.size(); x++) {
			const auto &cl = plan.classes[(size_t)b.cls[x]];
			const int k = sa_pk_decode(cl.cls).k;
			SaHostPkArgs &a = b.args[x];
			sa_pk_arranged_keys(in, b.g, k, cl.chunk, share_host, a.lv);
			a.cls_index = b.cls[x];
			a.chunk = cl.chunk;
			a.ncols = cl.ncols;
			a.npart = cl.npart;
			a.k = k;
			a.delta = sa_pk_delta(in, b.g, k);
			a.pk_base = sa_pk_base(in, b.g, k);
		}
		/* walking order, per rank: the large full tiles class after class, then everything smaller -- the full tiles of
		 * the small-tile classes and the partial tiles of all classes -- by decreasing work (rows x per-step weight) */
		auto &ul = b.ulist;
		b.ufirst.assign((size_t)nranks + 1, 0);
		b.nlocal.assign((size_t)nranks, 0);
		b.pairs.assign((size_t)nranks, 0);
		b.cells.assign((size_t)nranks, 0);
		struct Part {
			int64_t work;
			uint32_t code, pair;
		};
		std::vector<Part> parts;
		for (int r = 0; r < nranks; r++) {
			parts.clear();
			for (size_t x = 0; x < b.cls.size(); x++) {
				const auto &cl = plan.classes[(size_t)b.cls[x]];
				const int32_t nfull = cl.ntiles - cl.npart;
				const auto &T = cl.tprefix;
				const int32_t npairs = (cl.ncols + 1) / 2;
				const bool small = sa_pk_decode(cl.cls).small;
				const int64_t full_rows = (int64_t)sa_pk_wpb(in.method, b.g, sa_pk_decode(cl.cls).k) * (64 / b.g) * cl.chunk;
				int32_t pair_of_full = 0; /* (tiles ascend: the pair index only moves forward) */
				for (int32_t t = 0; t < cl.ntiles; t++) {
					uint32_t pair;
					if (t < nfull) {
						while (pair_of_full + 1 < npairs && T[(size_t)pair_of_full + 1] <= t)
							pair_of_full++;
						pair = (uint32_t)pair_of_full;
					} else {
						pair = (uint32_t)T[(size_t)(npairs + 1 + (t - nfull))];
					}
					if (world >= 1 && cl.owner[(size_t)t] != r)
						continue;
					const uint32_t code = ((uint32_t)x << SA_PK_UTILE_BITS) | (uint32_t)t;
					if (t < nfull && !small) {
						ul.push_back(code);
						ul.push_back(pair);
					} else /* small full tiles and every partial tile: by decreasing work, after the large full tiles */
						parts.push_back({ (t < nfull ? full_rows : (int64_t)cl.part_rows[(size_t)(t - nfull)]) * (b.args[x].k + 4), code, pair });
				}
				b.pairs[(size_t)r] += world >= 1 ? cl.rank_pairs[(size_t)r] : cl.pairs;
				b.cells[(size_t)r] += world >= 1 ? cl.rank_cells[(size_t)r] : cl.cells;
			}
			std::stable_sort(parts.begin(), parts.end(), [](const Part &p, const Part &q) { return p.work > q.work; });
			for (const Part &pt : parts) {
				ul.push_back(pt.code);
				ul.push_back(pt.pair);
			}
			b.ufirst[(size_t)r + 1] = (int64_t)ul.size(); /* (words: two per tile) */
			const int64_t n = (b.ufirst[(size_t)r + 1] - b.ufirst[(size_t)r]) / 2;
			if (n > INT32_MAX) {
				sa_set_error("%s", TOO_LARGE);
				return false;
			}
			b.nlocal[(size_t)r] = (int32_t)n;
		}
	}
	return true;
}

/* ---- C ABI: pair-space arithmetic (host only, no device needed) ----------------------------------------------------- */
extern "C" int64_t sa_pairs_cells(const struct sa_meta *meta, int32_t num, int64_t start, int64_t count)
{
	return sa_guard("sa_pairs_cells", (int64_t)-1, [&]() -> int64_t {
		if (!meta || num < 2)
			return -1;
		SaPairPlan plan(meta, num);
		if (start < 0 || count < 0 || start > plan.pairs - count)
			return -1;
		return plan.cells_before(start + count) - plan.cells_before(start);
	});
}

extern "C" int sa_pairs_partition(const struct sa_meta *meta, int32_t num, int parts, int64_t *bounds)
{
	return sa_guard("sa_pairs_partition", 1, [&] {
		if (!meta || num < 2 || parts < 1 || !bounds) {
			sa_set_error("sa_pairs_partition: bad arguments");
			return 1;
		}
		SaPairPlan plan(meta, num);
		const int64_t total = plan.cell_prefix[(size_t)num];
		bounds[0] = 0;
		for (int k = 1; k < parts; k++) {
			const int64_t target = (int64_t)((__int128)total * k / parts);
			int64_t lo = bounds[k - 1], hi = plan.pairs; /* first p with cells_before(p) >= target */
			while (lo < hi) {
				const int64_t mid = lo + (hi - lo) / 2;
				if (plan.cells_before(mid) >= target)
					hi = mid;
				else
					lo = mid + 1;
			}
			bounds[k] = lo;
		}
		bounds[parts] = plan.pairs;
		return 0;
	});
}
