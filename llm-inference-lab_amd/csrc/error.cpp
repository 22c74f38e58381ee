// Thread-local error string behind sd_last_error() (include/specdec_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include <string>

#include "../../include/specdec_hip.h"

namespace sd {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

void clear_error() { g_err.clear(); }

}  // namespace sd

extern "C" const char* sd_last_error(void) { return sd::g_err.c_str(); }

extern "C" int sd_abi_version(void) { return SD_ABI_VERSION; }
