/* sa_limits.cpp -- what each kernel family can reproduce exactly for a scoring (pure host arithmetic, see sa_plan.h) */
#include "sa_plan.h"

#include <algorithm>
#include <cstdlib>

/* Decides whether the systolic streaming kernels (sa_systolic.hip) reproduce the reference exactly
 * for this scoring, and derives their constants.  Conditions (see that file's header):
 *   - profile entries S + pconst (and the Gotoh first-column tweak) fit s8, -128 is reserved,
 *   - Gotoh: q = open - extend <= 0 (|open| >= |extend|),
 *   - the per-sequence baseline raise DELTA times the stream length stays far inside s32. */
static void systolic_limits(SaKernelLimits &L, const sa_scoring &sc, int32_t max_len, bool force_generic)
{
	int64_t smax = INT32_MIN, smin = INT32_MAX;
	for (int k = 0; k < SA_SUB_DIM * SA_SUB_DIM; k++) {
		smax = std::max<int64_t>(smax, sc.sub[k]);
		smin = std::min<int64_t>(smin, sc.sub[k]);
	}
	L.sys_ok = false;
	if (smin < -127 || smax > 127 || force_generic)
		return;
	int64_t pconst, q = 0, pmax, gain, slack;
	const int64_t g = sc.gap_pen, o = sc.gap_opn, e = sc.gap_ext;
	switch (sc.method) {
	case SA_METHOD_NW:
		pconst = -2 * g;
		pmax = smax + pconst;
		gain = std::max<int64_t>(0, smax - 2 * g);
		slack = 2;
		break;
	case SA_METHOD_GA:
		q = o - e;
		if (q > 0)
			return;
		pconst = -e - o;
		pmax = smax + pconst - q; /* first real column carries -q on top */
		gain = std::max<int64_t>(0, smax - 2 * e);
		slack = 2 * (-q) + 2;
		break;
	default:
		pconst = -o;
		pmax = smax + pconst;
		gain = std::max<int64_t>(0, smax);
		slack = -o - e + 2;
		break;
	}
	if (pmax > 127 || smin + pconst < -127)
		return;
	/* widest column the fast path will see: padded to whole strips when longer than the widest class */
	const int64_t wmax = ((int64_t)max_len + SA_SYS_LONG_W - 1) / SA_SYS_LONG_W * SA_SYS_LONG_W;
	/* SW works in a per-row shifted domain: values additionally drift by |e| per stream position of a tile */
	const int64_t drift = sc.method == SA_METHOD_SW ? (int64_t)SA_SYS_CHUNK * ((int64_t)max_len + 1) * std::llabs(e) : 0;
	if ((gain * wmax + slack) * (SA_SYS_CHUNK + 2) + drift >= ((int64_t)1 << 29))
		return;
	L.sys_ok = true;
	L.sys_pconst = (int32_t)pconst;
	L.sys_q = (int32_t)q;
	L.sys_gain = gain;
	L.sys_slack = slack;
}

/* Decides which column classes the packed-u16 kernels reproduce exactly (see sa_systolic_pk.inc): profile entries
 * S + const (and the Gotoh first-column tweak) are >= 0, Gotoh q <= 0, and for class K every value of a register stays
 * inside [floor, limit]: BASE + DELTA + the largest profile entry must fit.  The largest such K is pk_kmax.  limit =
 * 65535 for the 16-lane groups; the 8-lane kernels take their three-way maxima with v_pk_maximum3_f16, whose order on
 * u16 bit patterns is the unsigned order up to 0x7c00: limit = SA_PK_F16_MAX. */
static void pk_limits(SaKernelLimits &L, const sa_scoring &sc, int32_t max_len, int32_t min_len, bool no_pk, bool no_pk16)
{
	L.pk_kmax = L.pk16_kmax = L.pk16_f16_kmax = 0;
	L.pk_chunk_cap = SA_SYS_CHUNK;
	if (!L.sys_ok || no_pk)
		return;
	int64_t smax = INT32_MIN, smin = INT32_MAX;
	for (int k = 0; k < SA_SUB_DIM * SA_SUB_DIM; k++) {
		smax = std::max<int64_t>(smax, sc.sub[k]);
		smin = std::min<int64_t>(smin, sc.sub[k]);
	}
	const int64_t g = sc.gap_pen, o = sc.gap_opn, e = sc.gap_ext;
	int64_t pconst, q = 0, pmax, gain, slack, floor_v, extra = 0;
	if (sc.method == SA_METHOD_SW) {
		/* row-shifted domain: V = m - |o - e| and X = max(V, X) - |e| are plain subtractions (o <= e <= 0), values drift
		 * up by |e| per stream position of a tile and the lanes start up to G |e| below the baseline */
		if (o > e)
			return;
		pconst = -o;
		pmax = smax + pconst;
		gain = std::max<int64_t>(1, smax);
		slack = -o - e + 2;
		floor_v = -o - 2 * e + 2;
		q = o + e; /* (only its magnitude is used below: margins) */
		/* the drift of a tile grows with the length of its row streams: long sequences get shorter streams (a stream of
		 * 8 sequences of 1000 residues is 8000 steps -- the per-tile costs are long amortised), so that the drift stays a
		 * fraction of the u16 range and SW keeps the packed kernels whatever the longest sequence is */
		int cap = SA_SYS_CHUNK;
		while (cap > 4 && (int64_t)cap * ((int64_t)max_len + 1) * (-e) > 12000)
			cap >>= 1;
		L.pk_chunk_cap = cap;
		extra = (int64_t)cap * ((int64_t)max_len + 1) * (-e) + 16 * (-e);
	} else if (sc.method == SA_METHOD_NW) {
		pconst = -2 * g;
		pmax = smax + pconst;
		gain = std::max<int64_t>(1, pmax);
		slack = 2;
		floor_v = 0;
	} else {
		q = o - e;
		if (q > 0)
			return;
		pconst = -e - o;
		pmax = smax + pconst - q; /* the first real column carries -q on top */
		gain = std::max<int64_t>(1, smax - 2 * e);
		slack = 2 * (-q) + 2;
		floor_v = 2 * (-q) + 2;
	}
	if (smin + pconst < 0 || pmax > 4096 || -q > 4096)
		return;
	L.pk_pconst = (int32_t)pconst;
	L.pk_q = (int32_t)q;
	L.pk_extra = extra;
	L.pk_gain = gain;
	L.pk_slack = slack;
	L.pk_floor = (int32_t)floor_v;
	const int64_t fixed = floor_v + 4 * (-q) + 4 + pmax + (-q) + extra;
	for (int k = 1; k <= SA_PK_KMAX; k++) {
		if ((sa_pk_live(min_len, 8) + 1) * (gain * 8 * k + slack) + fixed > SA_PK_F16_MAX) /* (8-lane groups: f16-ordered halves) */
			break;
		L.pk_kmax = k;
	}
	if (L.pk_kmax == SA_PK_KMAX && !no_pk16) /* wider columns: 16-lane groups, twice as many shifts in flight */
		for (int k = SA_PK_K16_MIN; k <= SA_PK16_KMAX; k++) {
			const int64_t top = (sa_pk_live(min_len, 16) + 1) * (gain * 16 * k + slack) + fixed;
			if (top > 65535)
				break;
			L.pk16_kmax = k;
			if (top <= SA_PK_F16_MAX && k <= SA_PK16_F16_KMAX)
				L.pk16_f16_kmax = k;
		}
}

SaKernelLimits sa_kernel_limits(const sa_scoring &sc, int32_t max_len, int32_t min_len, bool force_generic, bool no_pk, bool no_pk16)
{
	SaKernelLimits L;
	systolic_limits(L, sc, max_len, force_generic);
	pk_limits(L, sc, max_len, min_len, no_pk, no_pk16);
	return L;
}
