// mfma_power_mix.hip — what the instructions that accompany the matrix instructions in winograd9's chunk cost in SUSTAINED CLOCK under the power limit.
// Base: the loop of mfma_order.hip (order 1), one 4-wave workgroup per CU.  Per 32 MFMAs the variants add, in winograd9's proportions (per 144 MFMAs:
// 280 VALU, 40 ds_read_b128 + 11 ds_write_b128, 23 global loads of 16 B per lane): V = 62 v_fma_f32, L = 9 ds_read_b128 + 2 ds_write_b128,
// G = 5 global_load_dwordx4 that hit L2 (an 8 MB buffer walked with a large stride), and all of them.  The figure of merit is MFMA TFLOP/s.
// Build: hipcc --offload-arch=gfx950 -O3 tools/experiments/mfma_power_mix.hip -o tools/mfma_power_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mf(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <bool V, bool L, bool G>
__global__ __launch_bounds__(256, 1) void loop(float* out, const unsigned* __restrict__ rnd, const f32x4* __restrict__ big, int iters, long long* clk) {
    __shared__ f32x4 lds[2048];                       // 32 KB
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4 A[4], B[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 4; ++e) {
            A[i][e] = rnd[(threadIdx.x * 37 + i * 4 + e) & 65535];
            B[i][e] = rnd[(blockIdx.x * 256 + threadIdx.x * 41 + 16 + i * 4 + e) & 65535];
        }
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = f32x4{__uint_as_float(rnd[i & 65535] & 0x3FFFFFFFu), __uint_as_float(rnd[(i + 7) & 65535] & 0x3FFFFFFFu), 1.f, 2.f};
    __syncthreads();
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = __uint_as_float(0x3F800000u | (rnd[(threadIdx.x + i * 256) & 65535] & 0x7FFFFFu));      // random mantissas in [1, 2)
    const float c1 = __uint_as_float(0x3F7FF123u), c2 = __uint_as_float(0x38D1B717u | (threadIdx.x << 3));
    f32x4 lv = lds[threadIdx.x], gv = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned gidx = blockIdx.x * 2048u + threadIdx.x;                  // f32x4 index into an 8 MB buffer (512 K elements)
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int a = (u >> 2) & 3, b = u & 3;
            acc[u & 7] = mf(A[a], B[b], acc[u & 7]);
            if (V) {                                               // 62 per 32 MFMAs
                x[(2 * u) & 7] = __builtin_fmaf(x[(2 * u) & 7], c1, c2);
                if (u < 30) x[(2 * u + 1) & 7] = __builtin_fmaf(x[(2 * u + 1) & 7], c1, c2);
            }
            if (L && (u % 7) == 3 && u < 32) {                     // u = 3, 10, 17, 24, 31 -> 5; plus u = 0, 6, 13, 20 below -> 9 reads
                f32x4 t = lds[(threadIdx.x * 5 + it * 3 + u * 64) & 2047];
                lv[0] += t[0];
            }
            if (L && (u == 0 || u == 6 || u == 13 || u == 20)) {
                f32x4 t = lds[(threadIdx.x * 3 + it + u * 32) & 2047];
                lv[1] += t[1];
            }
            if (L && (u == 8 || u == 24)) lds[(threadIdx.x + (it & 7) * 256) & 2047] = lv;     // 2 writes
            if (G && (u == 1 || u == 7 || u == 14 || u == 21 || u == 28)) {                     // 5 loads of 16 B per lane, L2 hits
                const f32x4 t = big[gidx & 524287u];
                gidx += 256u * 53u;
                gv[0] += t[0];
            }
        }
        if ((it & 255) == 255) for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] *= 1e-6f;
    }
    const long long c1_ = clock64(), w1 = wall_clock64();
    float s = lv[0] + lv[1] + gv[0];
    for (int i = 0; i < 8; ++i) { s += x[i]; for (int r = 0; r < 16; ++r) s += acc[i][r]; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1_ - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}
template <bool V, bool L, bool G> static void run(const unsigned* rnd, const f32x4* big, const char* name) {
    const int blocks = 256, iters = 30000;
    float* out; long long* clk;
    (void)hipMalloc(&out, blocks * 256 * 4); (void)hipMalloc(&clk, blocks * 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    loop<V, L, G><<<blocks, 256>>>(out, rnd, big, 3000, clk);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    loop<V, L, G><<<blocks, 256>>>(out, rnd, big, iters, clk);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 2);
    (void)hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
    double ghz = 0, cyc = 0; for (int b = 0; b < blocks; ++b) { ghz += (double)h[b * 2] / ((double)h[b * 2 + 1] / 100e6) / 1e9; cyc += (double)h[b * 2]; }
    const double flops = (double)blocks * 4 * iters * 32.0 * 32768.0;
    printf("%-46s %8.2f ms  %7.1f MFMA TFLOP/s  clock %.3f GHz  %.1f cycles per MFMA\n", name, ms, flops / ms / 1e9, ghz / blocks, cyc / blocks / iters / 32.0);
    (void)hipFree(out); (void)hipFree(clk);
}
int main() {
    std::vector<unsigned> h(65536);
    unsigned s = 12345u;
    for (auto& v : h) {
        unsigned w = 0;
        for (int k = 0; k < 2; ++k) { s = s * 1664525u + 1013904223u; const unsigned m = (s >> 9) & 0x3FF, e = 0x3C + ((s >> 20) & 1), sg = (s >> 25) & 1; w |= ((sg << 15) | (e << 10) | m) << (16 * k); }
        v = w;
    }
    unsigned* rnd; (void)hipMalloc(&rnd, h.size() * 4); (void)hipMemcpy(rnd, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> hb(524288 * 4);
    for (auto& v : hb) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) * (1.0f / 16777216.0f); }
    f32x4* big; (void)hipMalloc(&big, hb.size() * 4); (void)hipMemcpy(big, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<false, false, false>(rnd, big, "MFMA only");
        run<true, false, false>(rnd, big, "+ 62 v_fma_f32 per 32 MFMAs");
        run<false, true, false>(rnd, big, "+ 9 ds_read_b128 + 2 ds_write_b128");
        run<false, false, true>(rnd, big, "+ 5 global_load_dwordx4 (L2 hits)");
        run<true, true, true>(rnd, big, "+ all three (winograd9's mix)");
    }
    return 0;
}
