// core.hip -- ABI version + per-thread error string of libcosa_hip.so.
#include "common.hpp"
#include <cstring>

namespace cosa {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace cosa

extern "C" int cosa_abi_version(void) { return 1; }
extern "C" const char *cosa_last_error(void) { return cosa::g_err; }
