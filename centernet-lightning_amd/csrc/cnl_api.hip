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
extern "C" int cnl_absmax_stride(void) { return CNL_ABSMAX_STRIDE; }
// sizeof of the parameter structs as THIS library was compiled (0 conv, 1 decode, 2 deconv): a binder checks its own struct layout against it
extern "C" size_t cnl_sizeof_params(int32_t which) {
    return which == 0 ? sizeof(cnl_conv_params) : which == 1 ? sizeof(cnl_decode_params) : which == 2 ? sizeof(cnl_deconv_params) : 0;
}

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

// Page-locked host memory mapped into the device's address space (coherent): kernels read and write it through the host pointer, so a
// small per-frame record (the tracker's costs, its index lists) crosses PCIe as the kernel's own loads / stores — no copy-engine
// packet, no second stream operation to wait for.
extern "C" int cnl_host_alloc(size_t bytes, void** ptr) {
    CNL_REQUIRE(ptr && bytes > 0, CNL_E_BAD_ARG, "cnl_host_alloc: null pointer / zero bytes");
    void* h = nullptr;
    hipError_t e_ = hipHostMalloc(&h, bytes, hipHostMallocMapped | hipHostMallocCoherent);
    if (e_ != hipSuccess) return cnl::fail(CNL_E_HIP, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e_));
    void* d = nullptr;
    e_ = hipHostGetDevicePointer(&d, h, 0);
    if (e_ != hipSuccess || d != h) {
        (void)hipHostFree(h);
        return cnl::fail(CNL_E_HIP, "cnl_host_alloc: the device does not see the allocation at the host address (%s)", hipGetErrorString(e_));
    }
    *ptr = h;
    return CNL_OK;
}

extern "C" int cnl_host_free(void* ptr) {
    if (!ptr) return CNL_OK;
    hipError_t e_ = hipHostFree(ptr);
    if (e_ != hipSuccess) return cnl::fail(CNL_E_HIP, "hipHostFree: %s", hipGetErrorString(e_));
    return CNL_OK;
}
