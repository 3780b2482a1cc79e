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
