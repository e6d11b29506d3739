// Library-wide entry points: error string, version, device probe.
#include "mz_common.h"
#include <cstring>

namespace mz {
static thread_local char g_last_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}
}  // namespace mz

extern "C" {

const char* mz_last_error(void) { return mz::g_last_error; }

int mz_version(void) { return MZHIP_VERSION; }

int mz_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

}  // extern "C"
