// mfma_lo_bits.hip — the order of the six matrix instructions of one (input row, kernel row) segment of the row-Winograd kernels
// (3 split terms x 2 cout halves; A = weight piece of a cout half, B = V piece) under the chip's power limit:
//   CUR   n0t0 n1t0 n0t1 n1t1 n0t2 n1t2   the kernels' order: accumulators alternate, 3 of 6 neighbours share an operand (B)
//   ALT   n0t1 n1t2 n0t2 n1t1 n0t0 n1t0   accumulators alternate, 4 of 6 share
//   GRAY  n0t0 n0t2 n0t1 n1t1 n1t2 n1t0   every neighbour shares A or B, but three back-to-back MFMAs into ONE accumulator block
// t0 = hi lo', t1 = lo hi', t2 = hi hi'.  8 accumulator blocks x 2 halves per wave (256 registers, as in winograd9), 24 segments per iteration with
// the kernel's reuse pattern (a V row serves three segments, a weight set of a kernel row serves eight).  Random fp16 operands in registers.
// Build: hipcc --offload-arch=gfx950 -O3 tools/experiments/mfma_lo_bits.hip -o mfma_seg_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mf(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
constexpr int SEG_ROW[24] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5, 6, 6, 6, 7, 7, 8, 7, 8, 9};
constexpr int SEG_KY[24] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 1, 1, 2, 2, 2};
constexpr int ORD_T[3][6] = {{0, 0, 1, 1, 2, 2}, {1, 2, 2, 1, 0, 0}, {0, 2, 1, 1, 2, 0}};
constexpr int ORD_N[3][6] = {{0, 1, 0, 1, 0, 1}, {0, 1, 0, 1, 0, 1}, {0, 0, 0, 1, 1, 1}};
template <int ORDER>
__global__ __launch_bounds__(256, 1) void loop(float* out, const unsigned* __restrict__ rnd, int iters, long long* clk, unsigned lomask, unsigned bzero) {
    f32x16 acc[8][2];
    for (int i = 0; i < 8; ++i) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;
    u32x4 fb[3][2][2], vf[4][2];       // [ky][cout half][piece], [row % 4][piece]
    for (int k = 0; k < 3; ++k) for (int n = 0; n < 2; ++n) for (int p = 0; p < 2; ++p) for (int e = 0; e < 4; ++e)
        fb[k][n][p][e] = rnd[(threadIdx.x * 37 + ((k * 2 + n) * 2 + p) * 4 + e) & 65535] & (p ? lomask : 0xFFFFFFFFu);
    for (int k = 0; k < 4; ++k) for (int p = 0; p < 2; ++p) for (int e = 0; e < 4; ++e)
        vf[k][p][e] = rnd[(blockIdx.x * 256 + threadIdx.x * 41 + 64 + (k * 2 + p) * 4 + e) & 65535] & (p ? lomask : 0xFFFFFFFFu) & (((threadIdx.x * 7 + k + e) & 3) < bzero ? 0u : 0xFFFFFFFFu);
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 144; ++s) {
            const int seg = s / 6, r = SEG_ROW[seg], ky = SEG_KY[seg], t = ORD_T[ORDER][s % 6], n = ORD_N[ORDER][s % 6];
            const int ku = t == 1 ? 1 : 0, kv = t == 0 ? 1 : 0;
            acc[r - ky][n] = mf(fb[ky][n][ku], vf[r & 3][kv], acc[r - ky][n]);
        }
        if ((it & 63) == 63) for (int i = 0; i < 8; ++i) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) acc[i][n][r] *= 1e-6f;
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) s += acc[i][n][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}
template <int ORDER> static void run(const unsigned* rnd, const char* name, unsigned lomask, unsigned bzero) {
    const int blocks = 256, iters = 9000;
    float* out; long long* clk;
    (void)hipMalloc(&out, blocks * 256 * 4); (void)hipMalloc(&clk, blocks * 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    loop<ORDER><<<blocks, 256>>>(out, rnd, 500, clk, lomask, bzero);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    loop<ORDER><<<blocks, 256>>>(out, rnd, iters, clk, lomask, bzero);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 2);
    (void)hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
    double ghz = 0, cyc = 0; for (int b = 0; b < blocks; ++b) { ghz += (double)h[b * 2] / ((double)h[b * 2 + 1] / 100e6) / 1e9; cyc += (double)h[b * 2]; }
    const double flops = (double)blocks * 4 * iters * 144.0 * 32768.0;
    printf("%-44s %8.2f ms  %7.1f TFLOP/s  %.3f GHz  %.1f cycles / MFMA\n", name, ms, flops / ms / 1e9, ghz / blocks, cyc / blocks / iters / 144.0);
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
    // lo pieces (weights and V) with the low mantissa bits cleared: 10 (all) / 8 / 6 / 4 / 2 explicit mantissa bits kept; V with a quarter / half of its dwords zero (post-ReLU)
    run<0>(rnd, "warm-up", 0xFFFFFFFFu, 0);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(rnd, "lo pieces: 10 mantissa bits (as shipped)", 0xFFFFFFFFu, 0);
        run<0>(rnd, "lo pieces: 8 mantissa bits", 0xFFFCFFFCu, 0);
        run<0>(rnd, "lo pieces: 6 mantissa bits", 0xFFF0FFF0u, 0);
        run<0>(rnd, "lo pieces: 4 mantissa bits", 0xFFC0FFC0u, 0);
        run<0>(rnd, "lo pieces: 2 mantissa bits", 0xFF00FF00u, 0);
        run<0>(rnd, "10 bits, a quarter of V's dwords zero", 0xFFFFFFFFu, 1);
        run<0>(rnd, "10 bits, half of V's dwords zero", 0xFFFFFFFFu, 2);
        run<0>(rnd, "6 bits, half of V's dwords zero", 0xFFF0FFF0u, 2);
    }
    return 0;
}
