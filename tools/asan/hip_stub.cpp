// hip_stub.cpp — the three symbols the host sources take from search.hip, so that builder.cpp, makedb.cpp and
// host_search.cpp link into a CPU-only library for the sanitized build (tools/asan/Makefile).  Test infrastructure.
#include <cstdarg>
#include <cstdio>

#include "../../kaamer_amd/csrc/kaamer_internal.h"

static thread_local char g_err[512] = "";

int kaamer_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char *kaamer_last_error(void) { return g_err; }
extern "C" int kaamer_abi_version(void) { return KAAMER_ABI_VERSION; }
