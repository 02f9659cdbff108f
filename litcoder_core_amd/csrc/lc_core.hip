// Library-level entry points: version, per-thread error string, device check.
#include "lc_common.h"
#include <cstring>

namespace lc {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
}  // namespace lc

extern "C" int lc_version(void) { return 100; }

extern "C" const char* lc_last_error(void) { return lc::g_err; }

extern "C" int lc_check_device(int dev) {
    hipDeviceProp_t prop;
    LC_HIP(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return lc::fail(LC_E_ARCH, "device %d is %s; this library carries gfx950 code only", dev, prop.gcnArchName);
    return LC_OK;
}
