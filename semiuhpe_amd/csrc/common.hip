#include <stdarg.h>

#include "common.h"

namespace suhpe {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
static thread_local const char* g_last_kernel = "";
void note_kernel(const char* name) { g_last_kernel = name ? name : ""; }
const char* last_kernel() { return g_last_kernel; }
}  // namespace suhpe

extern "C" const char* suhpe_last_error(void) { return suhpe::g_err; }
extern "C" int suhpe_abi_version(void) { return 1; }
