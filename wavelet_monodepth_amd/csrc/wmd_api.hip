// Library-level entry points: version, status strings, thread-local error text.
#include <string.h>
#include "wmd_internal.h"

namespace wmd {

static thread_local char g_last_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

}  // namespace wmd

extern "C" int wmd_version(void) { return WMD_VERSION; }

extern "C" const char* wmd_last_error(void) { return wmd::g_last_error; }

extern "C" const char* wmd_status_string(int status) {
    switch (status) {
        case WMD_OK: return "ok";
        case WMD_ERR_BAD_ARG: return "bad argument";
        case WMD_ERR_BAD_SHAPE: return "bad shape";
        case WMD_ERR_UNSUPPORTED: return "unsupported";
        case WMD_ERR_HIP: return "HIP error";
        case WMD_ERR_WORKSPACE: return "workspace too small";
        case WMD_ERR_COMM: return "RCCL error";
        default: return "unknown status";
    }
}
