// Library-level entry points: ABI version, architecture string, error reporting.
#include <stdarg.h>
#include <string.h>
#include "fsw_common.h"

namespace fsw {
static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}
}  // namespace fsw

extern "C" int fsw_abi_version(void) { return FSW_ABI_VERSION; }
extern "C" const char* fsw_arch(void) { return "gfx950"; }
extern "C" const char* fsw_last_error(void) { return fsw::g_error; }
