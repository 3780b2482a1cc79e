// Shared host-side helpers for libcenternet_gfx950.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include "centernet_gfx950.h"

// cache policy (aux) of the conv kernels' output stores: 2 = nt.  A layer's output is far larger than the L2 (4 MB per XCD) and is
// next touched by the following launch; written with the default policy it only evicts the input patches and weights that the
// running kernel re-reads.
#ifndef CNL_NT_STORES
#define CNL_NT_STORES 2
#endif

namespace cnl {

// Thread-local last-error text, exported through cnl_last_error().
char* last_error_buf();
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CNL_E_HIP, "%s: %s", what, hipGetErrorString(e));
    return CNL_OK;
}

#define CNL_REQUIRE(cond, code, ...)                      \
    do {                                                  \
        if (!(cond)) return ::cnl::fail(code, __VA_ARGS__); \
    } while (0)

#define CNL_HIP(call)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) return ::cnl::fail(CNL_E_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// XCD-aware, bijective remap of a 1-D block id: the hardware places block b on XCD b % 8; give each
// XCD a contiguous chunk of logical tile ids so neighbouring tiles share that XCD's private L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

}  // namespace cnl
