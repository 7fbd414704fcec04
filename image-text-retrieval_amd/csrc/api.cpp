// Error string + ABI version of libitr_hip.so (host only).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/itr_hip.h"

namespace itr {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace itr

extern "C" const char *itr_last_error(void) { return itr::g_err; }
extern "C" int itr_abi_version(void) { return 30; }
