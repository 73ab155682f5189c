/* sa_env.cpp -- the one place that reads the process environment (see sa_env.h for what each switch does) */
#include "sa_env.h"

#include <algorithm>
#include <cstdlib>

static bool flag(const char *name) { return getenv(name) != nullptr; }
static int number(const char *name, int fallback, int lo, int hi)
{
	const char *v = getenv(name);
	if (!v || !*v)
		return fallback;
	return std::max(lo, std::min(hi, atoi(v)));
}

SaEnv sa_env_read()
{
	SaEnv e;
	e.force_generic = flag("SA_HIP_FORCE_GENERIC");
	e.no_pk = flag("SA_HIP_NO_PK");
	e.no_pk16 = flag("SA_HIP_NO_PK16");
	e.no_sort = flag("SA_HIP_NO_SORT");
	e.concurrent_classes = flag("SA_HIP_CONCURRENT_CLASSES");
	e.one_tile_size = flag("SA_HIP_ONE_TILE_SIZE");
	e.chunk = number("SA_HIP_CHUNK", 0, 1, 32);
	e.pk_wgs = number("SA_HIP_PK_WGS", 0, 0, 1 << 20);
	e.stagger = number("SA_HIP_STAGGER", 0, 0, 64);
	e.rotate_prio = getenv("SA_HIP_ROTATE_PRIO") ? (atoi(getenv("SA_HIP_ROTATE_PRIO")) != 0) : -1;
	e.small_below = number("SA_HIP_SMALL_BELOW", 16, 0, 1 << 20);
	e.small_div = number("SA_HIP_SMALL_DIV", 4, 2, 64);
	e.small_frac = number("SA_HIP_SMALL_FRAC", 5, 2, 1 << 20);
	e.no_pin = flag("SA_HIP_NO_PIN");
	e.no_direct = flag("SA_HIP_NO_DIRECT");
	e.no_shells = flag("SA_HIP_NO_SHELLS");
	e.devices = number("SA_HIP_DEVICES", 0, 0, 1024);
	e.split = number("SA_HIP_SPLIT", 0, 0, 1024);
	e.gather = getenv("SA_HIP_GATHER") ? (atoi(getenv("SA_HIP_GATHER")) != 0) : -1;
	e.tiles_split = number("SA_HIP_TILES_SPLIT", 0, 0, 64);
	e.verbose = flag("SA_HIP_VERBOSE");
	e.stamps = flag("SA_HIP_STAMPS");
	e.stamps_dump = getenv("SA_HIP_STAMPS_DUMP");
	e.ztrace = flag("SA_HIP_ZTRACE");
	return e;
}
