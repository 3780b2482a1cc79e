// winograd.hip — 3x3 / stride 1 / pad 1 convolution by Winograd F(2x2,3x3) on the fp32 matrix cores (gfx950).
//
// Same call sites as conv_mfma.hip (ResNet BasicBlock 3x3 convs, make_conv, GenericHead blocks: reference
// models/meta.py:24-26, models/layers.py:72-77) for the layers that are 3x3 stride-1: 92 % of the conv time.
// conv_mfma.hip already runs at ~98 % MFMA utilisation at the clock the chip sustains under this load
// (2.14 GHz measured, DVFS), so the only lever left is fewer multiplies: F(2x2,3x3) needs 16 instead of 36
// per (2x2 output tile, cin, cout) = 2.25x fewer MFMA flops.  The result is the same function up to fp32
// rounding (|err| ~2.6e-6 at K=2304 vs 1.2e-6 for the direct sum; tolerance 1e-4).
//
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A        d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs
// which is 16 independent GEMMs, one per transform position xi:  M_xi[tile][co] = sum_ci V_xi[tile][ci] U_xi[co][ci].
//
// Workgroup = 4 waves = 8x8 tiles (16x16 output pixels of one image) x 32 output channels x all 16 positions.
// Per chunk of 8 input channels:
//   LDS-DMA   the 18x18x8 input patch (zero halo from the buffer bounds check) and the chunk's U slice
//             ([xi][32 co][8 ci], pre-transformed and pre-packed once per weight load);
//   transform thread = (tile, channel): 16 ds_read_b32 -> 32 adds (B^T d B) -> 16 ds_write_b32 into V[xi][tile][8 ci];
//   MFMA      wave w owns positions 4w..4w+3: per position two 32-tile A fragments + one B fragment, each ONE
//             ds_read_b128 (lanes 0-31 take ci 0-3, lanes 32-63 ci 4-7: four K=2 steps per read), 8 MFMAs.
// Epilogue: accumulators go through LDS ([xi][tile][co]) so that thread = (tile, co) can apply A^T . A, + bias
// (+ residual) (+ ReLU) and store the 2x2 outputs NHWC (32 consecutive channels per 128-byte segment).
// LDS: V 32 KB + U 2 x 16 KB + patch 12 KB = 76 KB -> 2 workgroups per CU: one group's transform phase (VALU + LDS)
// runs under the other's MFMA phase.
#include "cnl_common.h"

namespace cnl_wino {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

struct WinoArgs {
    const float* x;
    const float* u;
    const float* bias;
    const float* res;
    float* y;
    int N, H, W, Cin, Cout, CoutP;
    int ldx, ldy, ldr;
    int CC;                       // Cin / 8
    int nb, bx, by;               // blocks along cout, x, y
    int blocks;
    unsigned x_bytes, u_bytes;
    unsigned flags;
};

constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr int T = 64;                       // tiles per workgroup (8 x 8)
constexpr int PW = 18;                      // patch width / height in pixels
constexpr int V_BYTES = 16 * T * 32;        // 32768
constexpr int U_BYTES = 16 * 32 * 32;       // 16384 per buffer
constexpr int P_BYTES = 3 * 256 * 16;       // 12288 (648 slots used)
constexpr int LDS_BYTES = V_BYTES + 2 * U_BYTES + P_BYTES;   // 77824

__device__ __forceinline__ void dma16(const float* base, unsigned bytes, char* lds_dst, unsigned voffset, unsigned soffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_dst, 16, voffset, soffset, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_zero() {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    return __builtin_amdgcn_mfma_f32_32x32x2f32(0.f, 0.f, z, 0, 0, 0);
}

__global__ __launch_bounds__(256, 2) void winograd_conv_kernel(const WinoArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sV = smem;
    char* sU = smem + V_BYTES;
    char* sP = smem + V_BYTES + 2 * U_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5;

    // block -> (image n, tile-block row/col, cout block); cout fastest so the blocks sharing a patch are neighbours
    unsigned b = cnl::xcd_remap(blockIdx.x, (unsigned)a.blocks);
    const int nbi = b % a.nb; b /= a.nb;
    const int bxi = b % a.bx; b /= a.bx;
    const int byi = b % a.by;
    const int n = b / a.by;
    const int y0 = byi * 16, x0 = bxi * 16, n0 = nbi * 32;

    // ---- per-lane DMA bookkeeping ----
    unsigned p_off[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int s = i * 256 + tid;                  // 16-byte slot of the patch: pixel s>>1, channel half s&1
        const int px = s >> 1, half = s & 1;
        const int py = px / PW, pxx = px - py * PW;
        const int iy = y0 - 1 + py, ix = x0 - 1 + pxx;
        const bool ok = s < PW * PW * 2 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        p_off[i] = ok ? (unsigned)((((n * a.H + iy) * a.W + ix) * a.ldx + half * 4) * 4) : OOB;
    }
    unsigned u_off[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) u_off[p] = (unsigned)((((wave * 4 + p) * a.CoutP + n0) * 8) * 4 + lane * 16);
    const unsigned u_chunk = (unsigned)(16 * a.CoutP * 8 * 4);          // bytes per channel chunk of U

#define WINO_ISSUE(cc_)                                                                                          \
    do {                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                            \
            dma16(a.x, a.x_bytes, sP + (i * 256 + wave * 64) * 16, p_off[i] == OOB ? OOB : p_off[i] + (unsigned)((cc_) * 32), 0); \
        _Pragma("unroll") for (int p = 0; p < 4; ++p)                                                            \
            dma16(a.u, a.u_bytes, sU + ((cc_) & 1) * U_BYTES + (wave * 4 + p) * 1024, u_off[p], (unsigned)(cc_) * u_chunk); \
    } while (0)

    WINO_ISSUE(0);

    f32x16 acc[4][2];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int g = 0; g < 2; ++g) acc[p][g] = mfma_zero();

    // transform item addresses (two items per thread): item idx = tid + 256*it -> (tile = idx >> 3, ch = idx & 7)
    int t_src[2], t_dst[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = tid + 256 * it;
        const int ch = idx & 7, tile = idx >> 3;
        const int ty = tile >> 3, tx = tile & 7;
        t_src[it] = (((2 * ty) * PW + 2 * tx) * 8 + ch) * 4;
        t_dst[it] = (tile * 8 + ch) * 4;
    }
    // fragment addresses
    const int a_frag = ((lane & 31) * 8 + hi * 4) * 4;      // + (xi*64 + g*32) * 32
    const int b_frag = ((lane & 31) * 8 + hi * 4) * 4;      // + xi * 1024

    for (int cc = 0; cc < a.CC; ++cc) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // chunk cc landed everywhere; MFMA phase cc-1 finished -> V free
        // ---- input transform: patch -> V = B^T d B ----
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const char* src = sP + t_src[it];
            float d[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) d[i][j] = *reinterpret_cast<const float*>(src + (i * PW + j) * 32);
            float t[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t[0][j] = d[0][j] - d[2][j];
                t[1][j] = d[1][j] + d[2][j];
                t[2][j] = d[2][j] - d[1][j];
                t[3][j] = d[1][j] - d[3][j];
            }
            char* dst = sV + t_dst[it];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<float*>(dst + (i * 4 + 0) * (T * 32)) = t[i][0] - t[i][2];
                *reinterpret_cast<float*>(dst + (i * 4 + 1) * (T * 32)) = t[i][1] + t[i][2];
                *reinterpret_cast<float*>(dst + (i * 4 + 2) * (T * 32)) = t[i][2] - t[i][1];
                *reinterpret_cast<float*>(dst + (i * 4 + 3) * (T * 32)) = t[i][1] - t[i][3];
            }
        }
        __syncthreads();                                   // V complete; patch buffer free
        if (cc + 1 < a.CC) WINO_ISSUE(cc + 1);
        // ---- 16 position GEMMs: wave owns positions 4*wave .. 4*wave+3 ----
        const char* uB = sU + (cc & 1) * U_BYTES;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int xi = wave * 4 + p;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(sV + (xi * T) * 32 + a_frag);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(sV + (xi * T + 32) * 32 + a_frag);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(uB + xi * 1024 + b_frag);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc[p][0] = mfma32(a0[c], bb[c], acc[p][0]);
                acc[p][1] = mfma32(a1[c], bb[c], acc[p][1]);
            }
        }
    }
#undef WINO_ISSUE

    // ---- epilogue: M (16 positions) -> LDS -> Y = A^T M A -> + bias (+ residual) (ReLU) -> NHWC ----
    const float lo = (a.flags & CNL_RELU) ? 0.f : -__builtin_inff();
    float* sM = reinterpret_cast<float*>(smem);            // [16][32 tiles][32 co] = 64 KB
    const int co = tid & 31;
    const int col = n0 + co;
    const bool col_ok = col < a.Cout;
    const float bv = col_ok ? a.bias[col] : 0.f;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        __syncthreads();                                   // everyone is done reading V/U (g=0) or sM of the previous pass
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int xi = wave * 4 + p;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tl = (r & 3) + 8 * (r >> 2) + 4 * hi;
                sM[(xi * 32 + tl) * 32 + (lane & 31)] = acc[p][g][r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int tl = (tid >> 5) + 8 * it;            // tile inside this 32-tile group
            float m[16];
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) m[xi] = sM[(xi * 32 + tl) * 32 + co];
            float q[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                q[i][0] = m[i * 4 + 0] + m[i * 4 + 1] + m[i * 4 + 2];
                q[i][1] = m[i * 4 + 1] - m[i * 4 + 2] - m[i * 4 + 3];
            }
            float yv[2][2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                yv[0][c] = q[0][c] + q[1][c] + q[2][c];
                yv[1][c] = q[1][c] - q[2][c] - q[3][c];
            }
            const int tile = g * 32 + tl;
            const int oy = y0 + 2 * (tile >> 3), ox = x0 + 2 * (tile & 7);
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int yy = oy + dy, xx = ox + dx;
                    if (col_ok && yy < a.H && xx < a.W) {
                        const size_t pix = ((size_t)n * a.H + yy) * a.W + xx;
                        float v = yv[dy][dx] + bv;
                        if (a.res) v += a.res[pix * a.ldr + col];
                        a.y[pix * a.ldy + col] = fmaxf(v, lo);
                    }
                }
        }
    }
}

// U = G g G^T per (co, ci), packed [Cin/8][16][CoutP][8]; rows co >= Cout are zero.
__global__ __launch_bounds__(256) void winograd_weights_kernel(const float* __restrict__ w, float* __restrict__ u, int Cin, int Cout,
                                                               int CoutP) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)CoutP * Cin) return;
    const int ci = (int)(t % Cin), co = (int)(t / Cin);
    float g[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) g[i][j] = co < Cout ? w[((long)co * 9 + i * 3 + j) * Cin + ci] : 0.f;   // OHWI
    float h[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        h[0][j] = g[0][j];
        h[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
        h[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
        h[3][j] = g[2][j];
    }
    const int cc = ci >> 3, c8 = ci & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float uu[4] = {h[i][0], 0.5f * (h[i][0] + h[i][1] + h[i][2]), 0.5f * (h[i][0] - h[i][1] + h[i][2]), h[i][2]};
#pragma unroll
        for (int j = 0; j < 4; ++j) u[((((long)cc * 16 + (i * 4 + j)) * CoutP + co) * 8) + c8] = uu[j];
    }
}

}  // namespace cnl_wino
using namespace cnl_wino;

extern "C" size_t cnl_winograd_weight_floats(int32_t Cin, int32_t Cout) {
    if (Cin <= 0 || Cout <= 0 || Cin % 8) return 0;
    const size_t CoutP = (size_t)((Cout + 31) / 32) * 32;
    return (size_t)(Cin / 8) * 16 * CoutP * 8;
}

extern "C" int cnl_winograd_transform_weights_f32(const float* w_ohwi, float* u, int32_t Cin, int32_t Cout, void* stream) {
    CNL_REQUIRE(w_ohwi && u, CNL_E_BAD_ARG, "cnl_winograd_transform_weights_f32: null pointer");
    CNL_REQUIRE(Cin > 0 && Cout > 0 && Cin % 8 == 0, CNL_E_UNSUPPORTED, "cnl_winograd_transform_weights_f32: Cin %% 8 != 0");
    const int CoutP = (Cout + 31) / 32 * 32;
    const long total = (long)CoutP * Cin;
    hipLaunchKernelGGL(winograd_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_ohwi, u, Cin,
                       Cout, CoutP);
    return cnl::check_launch("winograd_weights_kernel");
}

extern "C" int cnl_conv3x3_winograd_f32(const cnl_conv_params* p, void* stream) {
    CNL_REQUIRE(p, CNL_E_BAD_ARG, "cnl_conv3x3_winograd_f32: null params");
    CNL_REQUIRE(p->x && p->w && p->bias && p->y, CNL_E_BAD_ARG, "cnl_conv3x3_winograd_f32: null tensor pointer");
    CNL_REQUIRE(p->N > 0 && p->H_in > 0 && p->W_in > 0 && p->Cin > 0 && p->Cout > 0, CNL_E_BAD_ARG,
                "cnl_conv3x3_winograd_f32: non-positive dimension");
    CNL_REQUIRE(p->KH == 3 && p->KW == 3 && p->stride == 1 && p->pad == 1, CNL_E_UNSUPPORTED,
                "cnl_conv3x3_winograd_f32: only 3x3 / stride 1 / pad 1");
    CNL_REQUIRE(!(p->flags & (CNL_UPSAMPLE_IN | CNL_UPSAMPLE_OUT_ADD | CNL_SIGMOID)), CNL_E_UNSUPPORTED,
                "cnl_conv3x3_winograd_f32: upsample / sigmoid flags are handled by cnl_conv2d_nhwc_f32");
    CNL_REQUIRE(p->Cin % 8 == 0 && p->ldx % 4 == 0 && p->ldx >= p->Cin && p->ldy >= p->Cout, CNL_E_UNSUPPORTED,
                "cnl_conv3x3_winograd_f32: Cin %% 8 != 0 or bad pixel stride");
    CNL_REQUIRE(((uintptr_t)p->x & 15) == 0 && ((uintptr_t)p->w & 15) == 0, CNL_E_BAD_ARG, "cnl_conv3x3_winograd_f32: unaligned x / u");
    CNL_REQUIRE(!p->residual || p->ldr >= p->Cout, CNL_E_BAD_ARG, "cnl_conv3x3_winograd_f32: ldr < Cout");
    WinoArgs a;
    a.x = p->x; a.u = p->w; a.bias = p->bias; a.res = p->residual; a.y = p->y;
    a.N = p->N; a.H = p->H_in; a.W = p->W_in; a.Cin = p->Cin; a.Cout = p->Cout;
    a.CoutP = (p->Cout + 31) / 32 * 32;
    a.ldx = p->ldx; a.ldy = p->ldy; a.ldr = p->ldr;
    a.CC = p->Cin / 8;
    a.nb = a.CoutP / 32; a.bx = (p->W_in + 15) / 16; a.by = (p->H_in + 15) / 16;
    const long long blocks = (long long)p->N * a.by * a.bx * a.nb;
    CNL_REQUIRE(blocks < (1ll << 31), CNL_E_UNSUPPORTED, "cnl_conv3x3_winograd_f32: grid too large");
    a.blocks = (int)blocks;
    const unsigned long long xb = (((unsigned long long)p->N * p->H_in * p->W_in - 1) * p->ldx + p->Cin) * 4ull;
    const unsigned long long ub = (unsigned long long)cnl_winograd_weight_floats(p->Cin, p->Cout) * 4ull;
    CNL_REQUIRE(xb < 0xFFFFFF00ull && ub < 0xFFFFFF00ull, CNL_E_UNSUPPORTED, "cnl_conv3x3_winograd_f32: tensor spans >= 4 GiB; split the batch");
    a.x_bytes = (unsigned)xb; a.u_bytes = (unsigned)ub;
    a.flags = p->flags;
    static bool attr_done = false;
    if (!attr_done) {
        CNL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&winograd_conv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    LDS_BYTES));
        attr_done = true;
    }
    hipLaunchKernelGGL(winograd_conv_kernel, dim3((unsigned)blocks), dim3(256), LDS_BYTES, (hipStream_t)stream, a);
    return cnl::check_launch("winograd_conv_kernel");
}
