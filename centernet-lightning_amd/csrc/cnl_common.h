// Shared host-side helpers for libcenternet_gfx950.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
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
// Floats between the per-image slots of an x_absmax / y_absmax array (cnl_absmax_stride()): 32 = one 128-byte line per image.  Device-scope
// atomics are resolved line by line at the memory side; with the N slots of a launch packed into one line (stride 1, rounds 1-3) every
// report of every image queued behind every other (profiles/r04_ymax_atomics.txt).
#ifndef CNL_ABSMAX_STRIDE
#define CNL_ABSMAX_STRIDE 32
#endif
constexpr int AMS = CNL_ABSMAX_STRIDE;

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

// One-time, PER-DEVICE launch setup of a kernel (one process may drive several devices: hipFuncSetAttribute applies to the current
// device only, so a process-wide `static bool done` would leave every later device with the default dynamic-LDS limit and failing
// launches).  A bit per device ordinal in an atomic word: lock-free, and a lost race merely repeats an idempotent call.  The C ABI
// returns error codes, which std::call_once cannot carry out of its callable without exceptions — hence the atomic form.
struct DeviceOnce {
    std::atomic<unsigned long long> done{0};
};
int cu_count(int dev, int* n_cu);      // cnl_api.hip: cached multiProcessorCount of device `dev`
inline int kernel_setup(DeviceOnce& once, const void* fn, int lds_bytes, int* n_cu = nullptr) {
    int dev = 0;
    hipError_t e_ = hipGetDevice(&dev);
    if (e_ != hipSuccess) return fail(CNL_E_HIP, "hipGetDevice: %s", hipGetErrorString(e_));
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(once.done.load(std::memory_order_acquire) & bit)) {
        e_ = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e_ != hipSuccess) return fail(CNL_E_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize, %d): %s", lds_bytes, hipGetErrorString(e_));
        once.done.fetch_or(bit, std::memory_order_release);
    }
    return n_cu ? cu_count(dev, n_cu) : CNL_OK;
}

// XCD-aware, bijective remap of a 1-D block id: the hardware places block b on XCD b % 8; give each
// XCD a contiguous chunk of logical tile ids so neighbouring tiles share that XCD's private L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// max |y| hand-over (cnl_conv_params.y_absmax): every wave of a launch folds its maximum into N floats — ONE or two cache lines.  Device-scope
// atomics are resolved at the memory side and serialise per address: unconditional reports cost a 64-channel layer a quarter of its time
// (profiles/r04_ymax_atomics.txt: layer1 107 -> 83 us, layer3 82 -> 72 us without them).  So a wave looks first — an agent-scope load of the slot,
// requested early where the kernel can (peek_max) — and only RAISES the slot: after an image's first tiles almost every report is a no-op.
// (A stale, lower peek only costs a redundant atomic: the result is exact either way.)
__device__ __forceinline__ unsigned peek_max(const unsigned* slot) {
#ifdef CNL_NO_PEEK      // A/B builds (tools/ymax_ab.sh): every report is an atomic, as in rounds 1-3
    return 0u;
#else
    return __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void raise_max(unsigned* slot, float m, unsigned seen) {
    if (m > 0.f && __float_as_uint(m) > seen) atomicMax(slot, __float_as_uint(m));
}
__device__ __forceinline__ void report_max(unsigned* slot, float m) {      // peek and raise in one place (waits for the load)
    if (m > 0.f && __float_as_uint(m) > peek_max(slot)) atomicMax(slot, __float_as_uint(m));
}

// The heatmap's sigmoid (reference centernet.py:205) in conv epilogues: 1 / (1 + 2^(-x log2 e)) on the hardware's v_exp_f32 and v_rcp_f32 (1 ulp each)
// instead of expf + an IEEE divide (~40 instructions per value: 33-36 us of the 170-us heatmap conv at C1, profiles/r04_experiments.txt r6b).
// Absolute error <= 3e-7 on [0, 1] (the argument's rounding: |x| 2^-24 ln 2 relative on the exponential, times its weight <= 1/4 in the
// quotient), monotone in x; +-inf and the saturated ends give exactly 0 / 1.  The path's bar is 1e-4.
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}

// Maximum of a NON-NEGATIVE value over the wave / over each 32-lane half, in every lane: DPP inside the rows of 16 lanes, then four
// v_readlane.  (A __shfl_xor butterfly computes its ds_bpermute lane addresses from the lane id; the compiler hoists those five or six
// registers to kernel entry, where they stay live — or spill, every reload a vmcnt(0) — across a kernel's main loop.)
__device__ __forceinline__ float row16_max_nonneg(float v) {
#define CNL_DPP_F(ctrl_) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), (ctrl_), 0xF, 0xF, false))
    v = fmaxf(v, CNL_DPP_F(0xB1));        // quad_perm [1, 0, 3, 2]
    v = fmaxf(v, CNL_DPP_F(0x4E));        // quad_perm [2, 3, 0, 1]
    v = fmaxf(v, CNL_DPP_F(0x141));       // row_half_mirror
    v = fmaxf(v, CNL_DPP_F(0x140));       // row_mirror
#undef CNL_DPP_F
    return v;
}
__device__ __forceinline__ float wave_max_nonneg(float v) {
    const int b = __builtin_bit_cast(int, row16_max_nonneg(v));
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// first[0] = maximum over lanes 0..31, first[1] = over lanes 32..63 (both uniform)
__device__ __forceinline__ void half_max_nonneg(float v, float (&out)[2]) {
    const int b = __builtin_bit_cast(int, row16_max_nonneg(v));
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    out[0] = fmaxf(r0, r1); out[1] = fmaxf(r2, r3);
}

}  // namespace cnl
