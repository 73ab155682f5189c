/* sa_guard.h -- the exception barrier of every extern "C" entry point (pure C++, no HIP) */
#ifndef SA_GUARD_H
#define SA_GUARD_H

#include <exception>
#include <new>

#include "sa_shapes.h"

/* ---- failure model -----------------------------------------------------------------------------------------------
 * Every device failure of the reference is `perr` + `return false` (src/interface/seqalign_cuda.c:23-30).  Here the host
 * side is C++: an allocation failure or any other exception raised behind an entry point must come back the same way --
 * failure value + sa_last_error() -- never as std::terminate in the host's process.  Every extern "C" function with a
 * body that can allocate runs it through sa_guard. */
template <class R, class F> static inline R sa_guard(const char *who, R fail, F &&body) noexcept
{
	try {
		return body();
	} catch (const std::bad_alloc &) {
		sa_set_error("%s: out of host memory", who);
	} catch (const std::exception &e) {
		sa_set_error("%s: %s", who, e.what());
	} catch (...) {
		sa_set_error("%s: unknown C++ exception", who);
	}
	return fail;
}
template <class F> static inline void sa_guard_void(const char *who, F &&body) noexcept
{
	try {
		body();
	} catch (const std::bad_alloc &) {
		sa_set_error("%s: out of host memory", who);
	} catch (const std::exception &e) {
		sa_set_error("%s: %s", who, e.what());
	} catch (...) {
		sa_set_error("%s: unknown C++ exception", who);
	}
}

#endif /* SA_GUARD_H */
