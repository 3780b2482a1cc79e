// bf16x3_probe.hip — two questions about forming fp32 products on the bf16 matrix cores of gfx950 (16x the fp32 MFMA rate):
//   (A) accuracy: C = A B with A, B fp32, each split EXACTLY into three bf16 pieces (truncation: a = a1 + a2 + a3 bit for bit), the
//       six largest cross products a1b1, a1b2, a2b1, a1b3, a3b1, a2b2 on v_mfma_f32_32x32x16_bf16 with fp32 accumulation — error vs
//       float64 beside that of the fp32 matrix core (v_mfma_f32_32x32x2_f32 = an fmaf chain) on the same data, K = 2304;
//   (B) issue: cycles per bf16 MFMA per SIMD with NV VALU and NL LDS instructions issued between consecutive MFMAs of a wave
//       (2 waves per SIMD) — unlike beside the fp32 MFMA (tools/mfma_coexec.hip), do they hide?
// Build on the box: hipcc --offload-arch=gfx950 -O3 tools/bf16x3_probe.hip -o /tmp/bf16x3_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float v, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned b = __float_as_uint(v);
    const float hf = __uint_as_float(b & 0xFFFF0000u);
    const float r1 = v - hf;                                   // exact
    const unsigned b1 = __float_as_uint(r1);
    const float mf = __uint_as_float(b1 & 0xFFFF0000u);
    const float r2 = r1 - mf;                                  // exact
    h = b >> 16; m = b1 >> 16; l = __float_as_uint(r2) >> 16;  // r2 has <= 8 significant bits: nothing is dropped
}

// one wave: C[32][32] = A[32][K] * B[K][32]; mode 0 = fp32 MFMA, 1 = six-term bf16 split, 2 = three-term (a1b1, a1b2, a2b1)
__global__ __launch_bounds__(64) void acc_kernel(const float* A, const float* B, int K, float* C, int mode) {
    const int lane = threadIdx.x, row = lane & 31, half = lane >> 5;
    f32x16 c = {};
    if (mode == 0) {
        for (int k = 0; k < K; k += 2) c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[row * K + k + half], B[(k + half) * 32 + row], c, 0, 0, 0);
    } else {
        for (int k0 = 0; k0 < K; k0 += 16) {
            unsigned ah[8], am[8], al[8], bh[8], bm[8], bl[8];
            for (int e = 0; e < 8; ++e) {
                split3(A[row * K + k0 + half * 8 + e], ah[e], am[e], al[e]);
                split3(B[(k0 + half * 8 + e) * 32 + row], bh[e], bm[e], bl[e]);
            }
            auto pack = [](const unsigned* p) {
                u32x4 r;
                for (int i = 0; i < 4; ++i) r[i] = p[2 * i] | (p[2 * i + 1] << 16);
                return __builtin_bit_cast(bf16x8, r);
            };
            const bf16x8 A1 = pack(ah), A2 = pack(am), A3 = pack(al), B1 = pack(bh), B2 = pack(bm), B3 = pack(bl);
            if (mode == 1) {
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A3, B1, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B3, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B2, c, 0, 0, 0);
            }
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B1, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B2, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, c, 0, 0, 0);
        }
    }
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + row] = c[r];
}

// fp16 two-way split with power-of-two scales: x * S = hi + lo (hi = RN16(x S), lo = RN16(x S - hi)); terms hi lo', lo hi', hi hi'
__global__ __launch_bounds__(64) void acc16_kernel(const float* A, const float* B, int K, float* C, float sa, float sb, int terms) {
    const int lane = threadIdx.x, row = lane & 31, half = lane >> 5;
    f32x16 c = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        f16x8 ah, al, bh, bl;
        for (int e = 0; e < 8; ++e) {
            const float x = A[row * K + k0 + half * 8 + e] * sa, y = B[(k0 + half * 8 + e) * 32 + row] * sb;
            ah[e] = (_Float16)x; al[e] = (_Float16)(x - (float)ah[e]);
            bh[e] = (_Float16)y; bl[e] = (_Float16)(y - (float)bh[e]);
        }
        if (terms == 4) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
    }
    const float inv = 1.f / (sa * sb);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + row] = c[r] * inv;
}

// time of a pure fp16 MFMA loop whose A operand holds normal numbers (mode 0), subnormals (1) or zeros (2)
__global__ __launch_bounds__(256) void denorm_kernel(float* out, int iters, int mode, long long* clk) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned short bits = mode == 0 ? 0x3C00 : mode == 1 ? (unsigned short)(0x0001 + (threadIdx.x & 0xFF)) : 0;
    typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
    u16x8 au; for (int e = 0; e < 8; ++e) au[e] = bits;
    const f16x8 a = __builtin_bit_cast(f16x8, au);
    f16x8 b; for (int e = 0; e < 8; ++e) b[e] = (_Float16)(1.0f + 0.001f * e);
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u & 3], 0, 0, 0);
    const long long c1 = clock64();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}

// KIND 0: 32-bit VALU mix (v_and_b32, v_sub_f32, v_perm_b32), KIND 1: v_pk_add_f32, KIND 2: v_cvt_pk_bf16_f32
template <int NV, int NL, int KIND>
__global__ __launch_bounds__(512) void issue_kernel(float* out, int iters, long long* clk) {
    __shared__ u32x4 lds[1024];
    lds[threadIdx.x] = u32x4{1, 2, 3, 4};
    lds[threadIdx.x + 512] = u32x4{5, 6, 7, 8};
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f32x2{(float)threadIdx.x + 0.37f, (float)i + 0.11f};
    u32x4 l[4] = {};
    u32x4 au = {threadIdx.x * 77u, 0x3f803f80u, 0x3f813f82u, 0x40004000u}, bu = {0x3c003c00u, blockIdx.x, 0x3d003d00u, 0x3e003e00u};
    const bf16x8 a = __builtin_bit_cast(bf16x8, au), b = __builtin_bit_cast(bf16x8, bu);
    const u32x4* lp = lds + (threadIdx.x & 511);
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int q = (u * NV + j), d = q & 7, s = (q + 3) & 7;
                if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[d]) : "v"(v[s]));
                else if (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(v[d].x) : "v"(v[s].x), "v"(v[s].y));
                else if (q % 3 == 0) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(v[d].x) : "v"(v[s].x));
                else if (q % 3 == 1) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(v[d].y) : "v"(v[s].x), "v"(v[s].y));
                else asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(v[d].x) : "v"(v[s].x), "v"(v[s].y), "v"(0x07060302u));
            }
#pragma unroll
            for (int j = 0; j < NL; ++j)
                asm volatile("ds_read_b128 %0, %1" : "=v"(l[(u * NL + j) & 3]) : "v"((unsigned)(size_t)lp) : "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long c1 = clock64();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
    for (int i = 0; i < 4; ++i) s += (float)l[i].x;
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { clk[blockIdx.x * 16 + (threadIdx.x >> 6)] = c0; clk[blockIdx.x * 16 + 8 + (threadIdx.x >> 6)] = c1; }
}

template <int NV, int NL, int KIND>
void run_issue(int threads = 512) {
    const int blocks = 256, iters = 1000;
    float* out; long long* clk;
    hipMalloc(&out, blocks * 512 * 4); hipMalloc(&clk, blocks * 128);
    issue_kernel<NV, NL, KIND><<<blocks, threads>>>(out, 50, clk);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    issue_kernel<NV, NL, KIND><<<blocks, threads>>>(out, iters, clk);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 16);
    hipMemcpy(h.data(), clk, blocks * 128, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int b = 0; b < blocks; ++b) {
        long long lo = h[b * 16], hi = h[b * 16 + 8];
        for (int w = 1; w < threads / 64; ++w) { lo = std::min(lo, h[b * 16 + w]); hi = std::max(hi, h[b * 16 + 8 + w]); }
        cyc += (double)(hi - lo);
    }
    cyc /= blocks;
    const double mfma_per_simd = (double)iters * 24 * (threads / 256);
    const double tflops = 256.0 * (threads / 64) * iters * 24 * 32768.0 / (ms * 1e-3) / 1e12;
    printf("waves/SIMD=%d kind=%d NV=%d NL=%d : %.1f cycles per bf16 MFMA per SIMD   (%.0f TFLOP/s bf16 = %.1f TFLOP/s fp32-equivalent at 6 MFMA per product)\n",
           threads / 256, KIND, NV, NL, cyc / mfma_per_simd, tflops, tflops / 6);
    hipFree(out); hipFree(clk);
}

int main() {
    {   // (A)
        const int K = 2304;
        std::mt19937 rng(1);
        std::normal_distribution<float> nd(0.f, 1.f);
        std::vector<float> A(32 * K), B(K * 32);
        for (auto& x : A) { const float a = nd(rng), b = nd(rng), c = nd(rng), d = nd(rng); x = std::max(a, 0.f) - std::max(b, 0.f) - std::max(c, 0.f) + std::max(d, 0.f); }
        for (auto& x : B) x = nd(rng) * 0.03f;
        std::vector<double> ref(32 * 32, 0.0);
        for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[m * K + k] * B[k * 32 + n]; ref[m * 32 + n] = s; }
        float *dA, *dB, *dC;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 32 * 32 * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        const char* names[3] = {"fp32 MFMA 32x32x2", "bf16 split, 6 terms", "bf16 split, 3 terms"};
        for (int mode = 0; mode < 3; ++mode) {
            acc_kernel<<<1, 64>>>(dA, dB, K, dC, mode);
            std::vector<float> C(32 * 32);
            hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
            double mx = 0, rms = 0, sc = 0;
            for (int i = 0; i < 1024; ++i) { const double e = C[i] - ref[i]; mx = std::max(mx, std::fabs(e)); rms += e * e; sc = std::max(sc, std::fabs(ref[i])); }
            printf("(A) K=%d  %-22s max |err| %.3e   rms %.3e   (max |C| %.2f)\n", K, names[mode], mx, std::sqrt(rms / 1024), sc);
        }
    }
    {   // (A') fp16 two-way split
        const int K = 2304;
        std::mt19937 rng(1);
        std::normal_distribution<float> nd(0.f, 1.f);
        std::vector<float> A(32 * K), B(K * 32);
        for (auto& x : A) { const float a = nd(rng), b = nd(rng), c = nd(rng), d = nd(rng); x = std::max(a, 0.f) - std::max(b, 0.f) - std::max(c, 0.f) + std::max(d, 0.f); }
        for (auto& x : B) x = nd(rng) * 0.03f;
        std::vector<double> ref(32 * 32, 0.0);
        for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[m * K + k] * B[k * 32 + n]; ref[m * 32 + n] = s; }
        float *dA, *dB, *dC;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 32 * 32 * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        const float scales[4][2] = {{1.f, 1.f}, {64.f, 256.f}, {1024.f, 4096.f}, {1.f / 1024, 1.f}};
        for (int si = 0; si < 4; ++si)
            for (int terms = 3; terms <= 4; ++terms) {
                acc16_kernel<<<1, 64>>>(dA, dB, K, dC, scales[si][0], scales[si][1], terms);
                std::vector<float> C(32 * 32);
                hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
                double mx = 0, rms = 0;
                for (int i = 0; i < 1024; ++i) { const double e = C[i] - ref[i]; mx = std::max(mx, std::fabs(e)); rms += e * e; }
                printf("(A') K=%d fp16 split, %d terms, scales %g x %g : max |err| %.3e   rms %.3e\n", K, terms, scales[si][0], scales[si][1], mx, std::sqrt(rms / 1024));
            }
        float* out; long long* clk;
        hipMalloc(&out, 256 * 256 * 4); hipMalloc(&clk, 256 * 8);
        for (int mode = 0; mode < 3; ++mode) {
            denorm_kernel<<<256, 256>>>(out, 200, mode, clk);
            hipDeviceSynchronize();
            denorm_kernel<<<256, 256>>>(out, 2000, mode, clk);
            hipDeviceSynchronize();
            std::vector<long long> h(256);
            hipMemcpy(h.data(), clk, 256 * 8, hipMemcpyDeviceToHost);
            double c = 0; for (auto v : h) c += v; c /= 256;
            printf("(A') fp16 MFMA with %s A operand: %.1f cycles per MFMA\n", mode == 0 ? "normal" : mode == 1 ? "SUBNORMAL" : "zero", c / (2000.0 * 16));
        }
    }
    // (B)
    run_issue<0, 0, 0>();
    run_issue<2, 0, 0>(); run_issue<4, 0, 0>(); run_issue<5, 0, 0>(); run_issue<6, 0, 0>(); run_issue<8, 0, 0>();
    run_issue<2, 0, 1>(); run_issue<4, 0, 1>();
    run_issue<2, 0, 2>(); run_issue<4, 0, 2>();
    run_issue<0, 1, 0>(); run_issue<0, 2, 0>();
    run_issue<4, 1, 0>(); run_issue<5, 1, 0>(); run_issue<6, 1, 0>(); run_issue<4, 2, 0>();
    // one wave per SIMD: what hides behind the wave's OWN MFMAs?
    run_issue<0, 0, 0>(256); run_issue<1, 0, 0>(256); run_issue<2, 0, 0>(256); run_issue<3, 0, 0>(256); run_issue<4, 0, 0>(256);
    run_issue<5, 0, 0>(256); run_issue<6, 0, 0>(256); run_issue<8, 0, 0>(256);
    run_issue<0, 1, 0>(256); run_issue<0, 2, 0>(256); run_issue<4, 1, 0>(256); run_issue<4, 2, 0>(256); run_issue<2, 0, 2>(256); run_issue<4, 0, 2>(256);
    return 0;
}
