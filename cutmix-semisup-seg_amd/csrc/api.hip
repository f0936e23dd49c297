// Library-level entry points of libcutmixseg_hip.so: version, thread-local error text, device-props cache.
#include "common.hpp"
#include <mutex>
#include <string.h>

namespace cms {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace cms

extern "C" int cms_version(void) { return CMS_VERSION; }

extern "C" const char* cms_last_error(void) { return cms::g_err; }

extern "C" int cms_device_info(int* n_cu, char* name_out, size_t name_cap) {
    static std::once_flag once;
    static hipDeviceProp_t prop;
    static hipError_t err = hipSuccess;
    std::call_once(once, [] {
        int dev = 0;
        err = hipGetDevice(&dev);
        if (err == hipSuccess) err = hipGetDeviceProperties(&prop, dev);
    });
    if (err != hipSuccess) {
        cms::set_error("cms_device_info: %s", hipGetErrorString(err));
        return CMS_ELAUNCH;
    }
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (name_out && name_cap) {
        strncpy(name_out, prop.gcnArchName, name_cap - 1);
        name_out[name_cap - 1] = 0;
    }
    return CMS_OK;
}
