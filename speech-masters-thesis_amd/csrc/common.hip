// Error reporting + ABI version for libsmt_hip.so.
#include <stdarg.h>

#include "smt_common.h"

namespace smt {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace smt

extern "C" const char* smt_last_error(void) { return smt::g_err; }
extern "C" int smt_abi_version(void) { return 3; }
