// mfma_shape.hip — what does the power limit let the chip sustain on v_mfma_f32_16x16x32_f16 against v_mfma_f32_32x32x16_f16 (the same
// flops per cycle by the book; a 16x16 tile moves half the accumulator bytes and twice the operand bytes per flop), and how much does
// the operand DATA matter (random mantissas / half of the B values zero, as behind a ReLU / small-magnitude "lo" pieces)?
// One 4-wave workgroup per CU, 128 accumulator registers per lane in both shapes, operands resident in registers.
// Build: hipcc --offload-arch=gfx950 -O3 tools/experiments/mfma_shape.hip -o mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mf32(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mf16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <int SHAPE>
__global__ __launch_bounds__(256, 1) void loop(float* out, const unsigned* __restrict__ rnd, int iters, long long* clk, unsigned bmask) {
    u32x4 A[4], B[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 4; ++e) {
            A[i][e] = rnd[(threadIdx.x * 37 + i * 4 + e) & 65535];
            B[i][e] = rnd[(blockIdx.x * 256 + threadIdx.x * 41 + 16 + i * 4 + e) & 65535] & (((threadIdx.x * 7 + i + e) & 1) ? bmask : 0xFFFFFFFFu);
        }
    float s = 0;
    long long c0, w0, c1, w1;
    if (SHAPE == 32) {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        c0 = clock64(), w0 = wall_clock64();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) { const int a = (u >> 2) & 3, b = u & 3; acc[u & 7] = mf32(A[a], B[b], acc[u & 7]); }
            if ((it & 255) == 255) for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] *= 1e-6f;
        }
        c1 = clock64(), w1 = wall_clock64();
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        f32x4 acc[32];
        for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        c0 = clock64(), w0 = wall_clock64();
        for (int it = 0; it < iters; ++it) {
            // 64 MFMAs of half the flops each = the same flops per iteration
#pragma unroll
            for (int u = 0; u < 64; ++u) { const int a = (u >> 2) & 3, b = u & 3; acc[u & 31] = mf16(A[a], B[b], acc[u & 31]); }
            if ((it & 255) == 255) for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) acc[i][r] *= 1e-6f;
        }
        c1 = clock64(), w1 = wall_clock64();
        for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}
template <int SHAPE> static void run(const unsigned* rnd, unsigned bmask, const char* name) {
    const int blocks = 256, iters = 40000;
    float* out; long long* clk;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    loop<SHAPE><<<blocks, 256>>>(out, rnd, 2000, clk, bmask);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    loop<SHAPE><<<blocks, 256>>>(out, rnd, iters, clk, bmask);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 2);
    hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
    double ghz = 0; for (int b = 0; b < blocks; ++b) ghz += (double)h[b * 2] / ((double)h[b * 2 + 1] / 100e6) / 1e9;
    const double flops = (double)blocks * 4 * iters * 32.0 * 32768.0;
    printf("%-64s %8.2f ms  %7.1f TFLOP/s  %.3f GHz\n", name, ms, flops / ms / 1e9, ghz / blocks);
    hipFree(out); hipFree(clk);
}
static std::vector<unsigned> make(int kind) {
    std::vector<unsigned> h(65536);
    unsigned s = 12345u;
    for (auto& v : h) {
        unsigned w = 0;
        for (int k = 0; k < 2; ++k) {
            s = s * 1664525u + 1013904223u;
            unsigned m = (s >> 9) & 0x3FF, e = 0x3C + ((s >> 20) & 1), sg = (s >> 25) & 1;
            if (kind == 1) { m &= 0x3E0; }                 // 5 significant mantissa bits
            if (kind == 2) { e = 0x10 + ((s >> 20) & 7); } // small magnitudes (a "lo" piece)
            w |= ((sg << 15) | (e << 10) | m) << (16 * k);
        }
        v = w;
    }
    return h;
}
int main() {
    unsigned* rnd; hipMalloc(&rnd, 65536 * 4);
    const char* kinds[3] = {"random 10-bit mantissas", "5-bit mantissas", "small exponents"};
    for (int rep = 0; rep < 2; ++rep)
        for (int kind = 0; kind < 3; ++kind) {
            auto h = make(kind);
            hipMemcpy(rnd, h.data(), h.size() * 4, hipMemcpyHostToDevice);
            char nm[128];
            snprintf(nm, sizeof nm, "32x32x16  %s", kinds[kind]); run<32>(rnd, 0xFFFFFFFFu, nm);
            snprintf(nm, sizeof nm, "16x16x32  %s", kinds[kind]); run<16>(rnd, 0xFFFFFFFFu, nm);
            if (kind == 0) {
                snprintf(nm, sizeof nm, "32x32x16  %s, half of B's dwords zero", kinds[kind]); run<32>(rnd, 0u, nm);
                snprintf(nm, sizeof nm, "16x16x32  %s, half of B's dwords zero", kinds[kind]); run<16>(rnd, 0u, nm);
            }
        }
    return 0;
}
