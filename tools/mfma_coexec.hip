// mfma_coexec.hip — does ordinary VALU / LDS work issued between MFMAs of the same SIMD overlap with the matrix pipe on gfx950?
// For NV in {0,1,2,4,8} v_pk_add_f32 per v_mfma_f32_32x32x2_f32 (and NL ds_read_b64), with 1 and 2 waves per SIMD, prints shader
// cycles per MFMA per SIMD (64 = the pipe is the only limit).  Build on the box: hipcc --offload-arch=gfx950 -O3 tools/mfma_coexec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NV, int NL>
__global__ __launch_bounds__(512) void k(float* out, int iters, long long* clk) {
    __shared__ f32x2 lds[1024];
    lds[threadIdx.x] = f32x2{1.f, 2.f};
    lds[threadIdx.x + 512] = f32x2{3.f, 4.f};
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f32x2{(float)threadIdx.x, (float)i};
    f32x2 l[4] = {};
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
    const f32x2* lp = lds + (threadIdx.x & 511);
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j)
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[(u * NV + j) & 7]) : "v"(v[(u * NV + j + 3) & 7]));
#pragma unroll
            for (int j = 0; j < NL; ++j)
                asm volatile("ds_read_b64 %0, %1" : "=v"(l[(u * NL + j) & 3]) : "v"((unsigned)(size_t)lp) : "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long c1 = clock64();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
    for (int i = 0; i < 4; ++i) s += l[i].x;
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { clk[blockIdx.x * 16 + (threadIdx.x >> 6)] = c0; clk[blockIdx.x * 16 + 8 + (threadIdx.x >> 6)] = c1; }
}

template <int NV, int NL>
void run(int threads) {
    const int blocks = 256, iters = 2000;
    float* out; long long* clk;
    hipMalloc(&out, blocks * 512 * 4); hipMalloc(&clk, blocks * 128);
    k<NV, NL><<<blocks, threads>>>(out, 50, clk);
    hipDeviceSynchronize();
    k<NV, NL><<<blocks, threads>>>(out, iters, clk);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * 16);
    hipMemcpy(h.data(), clk, blocks * 128, hipMemcpyDeviceToHost);
    const int waves = threads / 64;
    double cyc = 0;          // per workgroup: first wave's start to last wave's end (the two waves of a SIMD need not start together)
    for (int b = 0; b < blocks; ++b) {
        long long lo = h[b * 16], hi = h[b * 16 + 8];
        for (int w = 1; w < waves; ++w) { lo = std::min(lo, h[b * 16 + w]); hi = std::max(hi, h[b * 16 + 8 + w]); }
        cyc += (double)(hi - lo);
    }
    cyc /= blocks;
    const double mfma_per_simd = (double)iters * 16 * (waves / 4);
    printf("NV=%d NL=%d waves/SIMD=%d : %.1f cycles per MFMA per SIMD\n", NV, NL, waves / 4, cyc / mfma_per_simd);
    hipFree(out); hipFree(clk);
}

// per-instruction cost of other instruction kinds issued by the SAME wave between two MFMAs (1 wave per SIMD)
template <int KIND, int NI>
__global__ __launch_bounds__(256) void k2(float* out, int iters, long long* clk) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)i;
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
    f32x2 w2 = {a, b};
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 l4[4] = {};
    unsigned sacc = blockIdx.x;
    float vm = a;
    const unsigned la = (unsigned)(size_t)(lds + (threadIdx.x & 255) * 4);
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                if (KIND == 1) asm volatile("s_add_u32 %0, %0, 7" : "+s"(sacc));
                if (KIND == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(l4[(u * NI + j) & 3]) : "v"(la) : "memory");
                if (KIND == 3) asm volatile("ds_write_b64 %0, %1" :: "v"(la), "v"(w2) : "memory");
                if (KIND == 4) asm volatile("s_nop 0");
                if (KIND == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(vm) : "v"(a));
                if (KIND == 6) asm volatile("v_add_u32 %0, %0, %1" : "+v"(sacc) : "v"(threadIdx.x));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long c1 = clock64();
    float s = vm + (float)sacc;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 4; ++i) s += l4[i].x;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 8 + (threadIdx.x >> 6)] = c1 - c0;
}

template <int KIND, int NI>
void run2(const char* name) {
    const int blocks = 256, iters = 2000;
    float* out; long long* clk;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 64);
    k2<KIND, NI><<<blocks, 256>>>(out, 50, clk);
    hipDeviceSynchronize();
    k2<KIND, NI><<<blocks, 256>>>(out, iters, clk);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * 8);
    hipMemcpy(h.data(), clk, blocks * 64, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < 4; ++w) cyc += (double)h[b * 8 + w];
    cyc /= blocks * 4;
    printf("%-14s x%d per MFMA: %.1f cycles per MFMA (1 wave/SIMD)\n", name, NI, cyc / (iters * 16.0));
    hipFree(out); hipFree(clk);
}

// dependent-accumulator distance: NA accumulators used round-robin by one wave per SIMD
template <int NA>
__global__ __launch_bounds__(256) void k3(float* out, int iters, long long* clk) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u % NA] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u % NA], 0, 0, 0);
    }
    const long long c1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 8 + (threadIdx.x >> 6)] = c1 - c0;
}
template <int NA>
void run3() {
    const int blocks = 256, iters = 2000;
    float* out; long long* clk;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 64);
    k3<NA><<<blocks, 256>>>(out, 50, clk);
    hipDeviceSynchronize();
    k3<NA><<<blocks, 256>>>(out, iters, clk);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * 8);
    hipMemcpy(h.data(), clk, blocks * 64, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < 4; ++w) cyc += (double)h[b * 8 + w];
    cyc /= blocks * 4;
    printf("%d accumulators round-robin: %.1f cycles per MFMA (1 wave/SIMD)\n", NA, cyc / (iters * 16.0));
    hipFree(out); hipFree(clk);
}

int main() {
    run3<1>(); run3<2>(); run3<4>(); run3<8>();
    run2<1, 1>("s_add_u32"); run2<1, 4>("s_add_u32"); run2<4, 1>("s_nop"); run2<4, 4>("s_nop");
    run2<2, 1>("ds_read_b128"); run2<2, 2>("ds_read_b128"); run2<3, 1>("ds_write_b64"); run2<3, 2>("ds_write_b64");
    run2<5, 1>("v_mov_b32"); run2<5, 4>("v_mov_b32"); run2<6, 1>("v_add_u32"); run2<6, 4>("v_add_u32");
    for (int t : {256, 512}) {
        run<0, 0>(t); run<1, 0>(t); run<2, 0>(t); run<4, 0>(t); run<8, 0>(t);
        run<0, 1>(t); run<0, 2>(t); run<2, 1>(t); run<4, 2>(t);
    }
    return 0;
}
