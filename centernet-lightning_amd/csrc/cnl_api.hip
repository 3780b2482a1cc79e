// libcenternet_gfx950.so — version / error plumbing of the C ABI (include/centernet_gfx950.h).
#include "cnl_common.h"

namespace cnl {

char* last_error_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

int cu_count(int dev, int* n_cu) {
    static std::atomic<int> cache[64];          // zero-initialised; one slot per device ordinal
    int v = cache[dev & 63].load(std::memory_order_relaxed);
    if (v <= 0) {
        hipDeviceProp_t prop;
        hipError_t e_ = hipGetDeviceProperties(&prop, dev);
        if (e_ != hipSuccess) return fail(CNL_E_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e_));
        v = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        cache[dev & 63].store(v, std::memory_order_relaxed);
    }
    *n_cu = v;
    return CNL_OK;
}

}  // namespace cnl

extern "C" int cnl_version(void) { return CNL_ABI_VERSION; }

extern "C" size_t cnl_last_error(char* buf, size_t n) {
    const char* s = cnl::last_error_buf();
    size_t len = strlen(s);
    if (buf && n) {
        size_t c = len < n - 1 ? len : n - 1;
        memcpy(buf, s, c);
        buf[c] = 0;
    }
    return len;
}
