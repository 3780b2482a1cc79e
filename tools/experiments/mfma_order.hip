// mfma_order.hip — does the ORDER of the matrix instructions (which operand registers consecutive v_mfma_f32_32x32x16_f16 share) change
// what the power limit lets the chip sustain?  Random fp16 operands in registers, one 4-wave workgroup per CU, 8 accumulator tiles per wave,
// the same multiset of (A, B, C) triples per iteration in three orders.  Build: hipcc --offload-arch=gfx950 -O3 tools/experiments/mfma_order.hip -o mfma_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mf(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// ORDER 0: A changes every MFMA, B changes every MFMA (nothing shared between neighbours)
// ORDER 1: A fixed over 4 consecutive MFMAs (B and C change)           — "row-major over a tile block"
// ORDER 2: A fixed over 8, B alternates between two                     — maximal sharing
template <int ORDER>
__global__ __launch_bounds__(256, 1) void loop(float* out, const unsigned* __restrict__ rnd, int iters, long long* clk) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4 A[4], B[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 4; ++e) {
            A[i][e] = rnd[(threadIdx.x * 37 + i * 4 + e) & 65535];
            B[i][e] = rnd[(blockIdx.x * 256 + threadIdx.x * 41 + 16 + i * 4 + e) & 65535];
        }
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        // 32 MFMAs per iteration: every (a, b) pair of 4 x 4 twice, accumulator k = (a * 4 + b) & 7
        if (ORDER == 0) {
#pragma unroll
            for (int u = 0; u < 32; ++u) { const int a = u & 3, b = (u + (u >> 2)) & 3; acc[u & 7] = mf(A[a], B[b], acc[u & 7]); }
        } else if (ORDER == 1) {
#pragma unroll
            for (int u = 0; u < 32; ++u) { const int a = (u >> 2) & 3, b = u & 3; acc[u & 7] = mf(A[a], B[b], acc[u & 7]); }
        } else {
#pragma unroll
            for (int u = 0; u < 32; ++u) { const int a = (u >> 3) & 3, b = (u & 1) + 2 * ((u >> 4) & 1); acc[u & 7] = mf(A[a], B[b], acc[u & 7]); }
        }
        if ((it & 255) == 255) for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] *= 1e-6f;
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}
template <int ORDER> static void run(const unsigned* rnd, const char* name) {
    const int blocks = 256, iters = 40000;
    float* out; long long* clk;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    loop<ORDER><<<blocks, 256>>>(out, rnd, 2000, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    loop<ORDER><<<blocks, 256>>>(out, rnd, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 2);
    hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
    double ghz = 0; for (int b = 0; b < blocks; ++b) ghz += (double)h[b * 2] / ((double)h[b * 2 + 1] / 100e6) / 1e9;
    const double flops = (double)blocks * 4 * iters * 32.0 * 32768.0;
    printf("%-44s %8.2f ms  %7.1f TFLOP/s  s_memtime/s_memrealtime %.3f GHz\n", name, ms, flops / ms / 1e9, ghz / blocks);
    hipFree(out); hipFree(clk);
}
int main() {
    std::vector<unsigned> h(65536);
    unsigned s = 12345u;
    for (auto& v : h) {              // two random fp16 in [-2, 2) per dword: exponent 0x3C..0x3F region, random mantissa and sign
        unsigned w = 0;
        for (int k = 0; k < 2; ++k) { s = s * 1664525u + 1013904223u; const unsigned m = (s >> 9) & 0x3FF, e = 0x3C + ((s >> 20) & 1), sg = (s >> 25) & 1; w |= ((sg << 15) | (e << 10) | m) << (16 * k); }
        v = w;
    }
    unsigned* rnd; hipMalloc(&rnd, h.size() * 4); hipMemcpy(rnd, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        run<0>(rnd, "order 0: A and B change every MFMA");
        run<1>(rnd, "order 1: A shared by 4 neighbours");
        run<2>(rnd, "order 2: A shared by 8, B alternates");
    }
    return 0;
}
