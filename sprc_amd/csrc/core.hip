// core.hip -- version, error reporting
#include <stdarg.h>

#include "common.hpp"

namespace sprc {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace sprc

extern "C" int sprc_version(void) { return SPRC_ABI_VERSION; }
extern "C" const char* sprc_last_error(void) { return sprc::g_err; }
