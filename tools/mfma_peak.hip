// mfma_peak.hip — on-box measurement of the fp32 MFMA ceiling (v_mfma_f32_32x32x2_f32) and the shader clock it
// sustains, plus a device-copy HBM ceiling.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void mfma_loop(float* out, int iters, long long* clk) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc[3], 0, 0, 0);
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

// same loop with RANDOM operands held in 16 registers: data toggling raises power and the chip clocks down (DVFS)
__global__ __launch_bounds__(256, 2) void mfma_loop_random(float* out, const float* __restrict__ rnd, int iters, long long* clk) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = rnd[(threadIdx.x * 16 + i) & 65535]; b[i] = rnd[(blockIdx.x * 256 + threadIdx.x * 16 + 8 + i) & 65535]; }
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u * 4 + 0) & 7], b[(u * 3 + 1) & 7], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u * 4 + 1) & 7], b[(u * 3 + 2) & 7], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u * 4 + 2) & 7], b[(u * 3 + 3) & 7], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u * 4 + 3) & 7], b[(u * 3 + 4) & 7], acc[3], 0, 0, 0);
        }
        // keep accumulators bounded so values stay "random-looking" instead of overflowing to inf
        if ((it & 63) == 63) for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] *= 1e-3f;
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

__global__ __launch_bounds__(256) void copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = src[i];
}

int main() {
    const int blocks_per_cu[2] = {1, 2};
    for (int bi = 0; bi < 2; ++bi) {
        const int blocks = 256 * blocks_per_cu[bi], iters = 20000 / blocks_per_cu[bi];
        float* out; long long* clk;
        hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        mfma_loop<<<blocks, 256>>>(out, 100, clk);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        mfma_loop<<<blocks, 256>>>(out, iters, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(blocks * 2);
        hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
        double flops = (double)blocks * 4 /*waves*/ * iters * 64.0 * 4096.0;
        double ghz = 0; for (int b = 0; b < blocks; ++b) ghz += (double)h[b * 2] / ((double)h[b * 2 + 1] / 100e6) / 1e9;   // wall clock = 100 MHz
        printf("mfma_f32_32x32x2: %d blocks/CU  %.2f ms  %.1f TFLOP/s  shader clock %.3f GHz (clock64/wall_clock64)\n",
               blocks_per_cu[bi], ms, flops / ms / 1e9, ghz / blocks);
        hipFree(out); hipFree(clk);
    }
    {
        const int blocks = 512, iters = 10000;
        float *out, *rnd; long long* clk;
        hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16); hipMalloc(&rnd, 65536 * 4);
        std::vector<float> hr(65536);
        unsigned x = 12345u;
        for (auto& v : hr) { x = x * 1664525u + 1013904223u; v = ((x >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
        hipMemcpy(rnd, hr.data(), 65536 * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        mfma_loop_random<<<blocks, 256>>>(out, rnd, 100, clk);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        mfma_loop_random<<<blocks, 256>>>(out, rnd, iters, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(blocks * 2);
        hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
        double flops = (double)blocks * 4 * iters * 64.0 * 4096.0, ghz = 0;
        for (int b = 0; b < blocks; ++b) ghz += (double)h[b * 2] / ((double)h[b * 2 + 1] / 100e6) / 1e9;
        printf("mfma_f32_32x32x2 RANDOM operands: 2 blocks/CU  %.2f ms  %.1f TFLOP/s  shader clock %.3f GHz\n", ms, flops / ms / 1e9, ghz / blocks);
    }
    const long n = 1l << 26;   // 1 GiB of float4
    f32x4 *a, *b; hipMalloc(&a, n * 16); hipMalloc(&b, n * 16);
    hipMemset(a, 1, n * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    copy_kernel<<<256 * 16, 256>>>(a, b, n); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) copy_kernel<<<256 * 16, 256>>>(a, b, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("device copy 1 GiB: %.3f ms/iter  %.2f TB/s (read+write)\n", ms / 5, 2.0 * n * 16 / (ms / 5) / 1e9);
    return 0;
}
