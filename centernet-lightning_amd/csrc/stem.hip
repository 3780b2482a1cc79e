// stem.hip — ResNet stem for gfx950: Conv2d(3,64,7,s2,p3)+BN(folded)+ReLU on the matrix cores, and
// MaxPool2d(3,s2,p1) on NHWC.
//
// Replaces torchvision resnet.conv1/bn1/relu/maxpool reached through backbone.forward_features
// (reference models/meta.py:42; contract tests/test_backbones.py:60-70: stride-2 feature with 64 ch).
//
// conv: implicit GEMM with K = 7*7*3 = 147, ordered k = ky*22 + t (t = kx*3 + c < 21, t = 21 a zero weight row): 154.
// Cin = 3 makes a global-memory im2col hopeless (12-byte pixels), so the im2col happens on the LDS side: a workgroup
// stages the (2*16+5) x (2*32+5) x 3 input patch of its 16x32 output tile and the whole 154x64 weight
// matrix in LDS; each MFMA A operand is a per-lane ds_read_b32 at  row(ky) * RS + 6*px + t
// (kx and c are contiguous in the patch row, so t is a plain offset).  The even row length makes the K step of lane half
// hi (k = 2s + hi) fall in ONE patch row for both halves, so with the 77 steps fully unrolled every LDS address is
// "lane base + immediate": no VALU instruction in the MFMA loop (on gfx950 VALU work does not overlap the matrix pipe,
// tools/mfma_coexec.hip; the previous per-lane k%21 / k/21 bookkeeping cost 2 VALU per MFMA, a quarter of the loop).
// Each wave computes four output rows (4 x 32 pixels) x 64 channels = eight 32x32 accumulators with v_mfma_f32_32x32x2_f32.
// The weight matrix is copied global -> LDS with coalesced reads (thread e reads w[e]) into a [k][65]
// image (row stride 65 keeps both the transposing write and the fragment reads bank-conflict-free); the first
// version read it with a 147-float stride per lane and spent 3/4 of its time there.
// The input is read through explicit element strides, so NCHW and channels_last callers are zero-copy.
#include "cnl_common.h"
#include <cstdlib>

namespace cnl_stem {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ST_TH = 16, ST_TW = 32;                 // output tile (4 rows per wave)
constexpr int ST_PR = 2 * ST_TH + 5;                  // patch rows
constexpr int ST_PC = 2 * ST_TW + 5;                  // patch cols
constexpr int ST_RS = 256;                            // patch row stride in floats: 207 used; 1 KB = 4 DMA instructions per row
constexpr int ST_KROW = 22;                           // K entries per kernel row: 7 taps x 3 channels + 1 zero pad
constexpr int ST_KP = 7 * ST_KROW;                    // 154
constexpr int ST_W_BYTES = ST_KP * 64 * 4;            // 39424 = 38.5 x 1 KB (the 39th DMA piece's upper lanes are out of bounds: zeros)
constexpr int ST_LDS_BYTES = ST_PR * ST_RS * 4 + 39 * 1024;    // 77824 -> 2 workgroups / CU
constexpr unsigned ST_OOB = 0xFFFFFFF0u;

typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void dma4(const float* base, unsigned bytes, float* lds_dst, unsigned voffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_dst, 4, voffset, 0, 0, 0);
}
__device__ __forceinline__ void dma16(const float* base, unsigned bytes, char* lds_dst, unsigned voffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_dst, 16, voffset, 0, 0, 0);
}

__global__ __launch_bounds__(256, 2) void stem_conv_kernel(const float* __restrict__ x, long sn, int sc, int sh, int sw,
                                                           unsigned x_img_bytes, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, int N, int H,
                                                           int W, int Ho, int Wo, int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* patch = reinterpret_cast<float*>(smem);
    float* wl = patch + ST_PR * ST_RS;                // [148][64], k-major (pre-packed by cnl_stem_pack_weights_f32)

    const int tid = threadIdx.x;
    const int lane = tid & 63, hi = lane >> 5, px = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int oy0 = ty * ST_TH, ox0 = tx * ST_TW;
    const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;

    // ---- staging by LDS-DMA: no VGPR round trip and almost no VALU (this runs beside the co-resident group's MFMA stream) ----
    // weights: 39 pieces of 1 KB, lane-linear
    for (int q = wave; q < (ST_W_BYTES + 1023) / 1024; q += 4)
        dma16(w, (unsigned)ST_W_BYTES, reinterpret_cast<char*>(wl) + q * 1024, (unsigned)(q * 1024 + lane * 16));
    // input patch -> LDS [row][col*3 + c]: 4-byte DMA, lane f of a row fetches pixel col = f/3, channel c = f%3 from wherever
    // the caller's strides put it (NCHW planes or channels_last alike); out-of-image lanes get zeros from the bounds check
    const float* xn = x + (long)n * sn;
    unsigned lane_off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int f = q * 64 + lane;
        const int col = f / 3, c = f - col * 3;
        const int ix = ix0 + col;
        lane_off[q] = (f < ST_PC * 3 && (unsigned)ix < (unsigned)W) ? (unsigned)((c * sc + ix * sw) * 4) : ST_OOB;
    }
    for (int r = wave; r < ST_PR; r += 4) {
        const int iy = iy0 + r;
        const bool row_ok = (unsigned)iy < (unsigned)H;                       // wave-uniform
        const unsigned row_off = (unsigned)(iy * sh * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            dma4(xn, x_img_bytes, patch + r * ST_RS + q * 64, (row_ok && lane_off[q] != ST_OOB) ? lane_off[q] + row_off : ST_OOB);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // lane's A base: output row (wave*4 + i), column px -> patch row 2*(wave*4+i) + ky, col 2*px + kx; K step s covers
    // k = 2s + hi = ky*22 + t with ky = s / 11 and t = 2*(s % 11) + hi: the lane half only shifts the base by one float
    const float* pa = patch + (wave * 8) * ST_RS + px * 6 + hi;  // + (2*i + ky) * ST_RS + 2 * (s % 11)
    const float* pb = wl + hi * 64 + px;                         // + s * 128 (+ 32 for the second cout group)
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
        for (int u = 0; u < ST_KROW / 2; ++u) {
            const int s = ky * (ST_KROW / 2) + u;
            const float b0 = pb[s * 128];
            const float b1 = pb[s * 128 + 32];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float av = pa[(2 * i + ky) * ST_RS + 2 * u];
                if (u == ST_KROW / 2 - 1) av = hi ? 0.f : av;     // t = 21 is the pad entry: its patch slot holds a neighbouring pixel
                                                                  // (zero weight, but 0 x inf would not be 0)
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc[i][1], 0, 0, 0);
            }
        }
    }

    // epilogue: bias + ReLU, NHWC store (col = lane&31 -> channel, rows -> pixels of the row)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int co = j * 32 + px;
        const float bv = bias[co];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int oy = oy0 + wave * 4 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ox = ox0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (oy < Ho && ox < Wo) {
                    const float v = fmaxf(acc[i][j][r] + bv, 0.f);
                    __builtin_nontemporal_store(v, &y[(((size_t)n * Ho + oy) * Wo + ox) * 64 + co]);     // CNL_NT_STORES: 537 MB that only the max-pool reads
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void maxpool_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y, int N, int H, int W,
                                                      int C4, int Ho, int Wo, long total) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C4);
        long p = e / C4;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int n = (int)(p / Ho);
        const float ninf = -__builtin_inff();
        f32x4 m = {ninf, ninf, ninf, ninf};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const f32x4 v = x[(((long)n * H + iy) * W + ix) * C4 + c];
                m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
            }
        }
        y[e] = m;
    }
}

// OHWI [64][7][7][3] (BN folded) -> the kernel's LDS image [154][64]: row k = ky*22 + t holds tap (ky, kx = t/3, c = t%3), t = 21 zeros
__global__ __launch_bounds__(256) void stem_pack_kernel(const float* __restrict__ w, float* __restrict__ wp) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= ST_KP * 64) return;
    const int k = e >> 6, co = e & 63;
    const int ky = k / ST_KROW, t = k - ky * ST_KROW;
    wp[e] = t < 21 ? w[co * 147 + ky * 21 + t] : 0.f;
}

}  // namespace cnl_stem
using namespace cnl_stem;

// stem_f16x2.hip: the same conv on the fp16 matrix cores; its split weights ride behind the fp32 image in the packed buffer
size_t cnl_stem5_extra_floats();
int cnl_stem5_pack(const float* w_ohwi, float* extra, void* stream);
int cnl_stem5_launch(const void* x, bool u8, const float* mean255, const float* inv_std255, long sn, int sc, int sh, int sw, unsigned img_bytes,
                     const float* extra, const float* bias, float* y, float* y_absmax, int N, int H, int W, int Ho, int Wo, int tiles_x, int tiles_y,
                     unsigned blocks, bool pool, void* stream);

extern "C" size_t cnl_stem_packed_weight_floats(void) { return (size_t)ST_KP * 64 + cnl_stem5_extra_floats(); }

extern "C" int cnl_stem_pack_weights_f32(const float* w_ohwi, float* w_packed, void* stream) {
    CNL_REQUIRE(w_ohwi && w_packed, CNL_E_BAD_ARG, "cnl_stem_pack_weights_f32: null pointer");
    hipLaunchKernelGGL(stem_pack_kernel, dim3((ST_KP * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_ohwi, w_packed);
    const int rc = cnl::check_launch("stem_pack_kernel");
    return rc != CNL_OK ? rc : cnl_stem5_pack(w_ohwi, w_packed + ST_KP * 64, stream);
}

static int stem_launch(const char* who, bool pool, const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, const float* w,
                       const float* bias, float* y, float* y_absmax, int32_t N, int32_t H, int32_t W, uint32_t algo, void* stream) {
    CNL_REQUIRE(x && w && bias && y, CNL_E_BAD_ARG, "%s: null tensor pointer", who);
    CNL_REQUIRE(N > 0 && H > 0 && W > 0, CNL_E_BAD_ARG, "%s: non-positive dimension", who);
    CNL_REQUIRE(sc > 0 && sh > 0 && sw > 0 && sn >= 0, CNL_E_UNSUPPORTED, "%s: non-positive strides", who);
    CNL_REQUIRE(((uintptr_t)w & 15) == 0, CNL_E_BAD_ARG, "%s: packed weights must be 16-byte aligned", who);
    const unsigned long long img_bytes = (2ull * sc + (unsigned long long)(H - 1) * sh + (unsigned long long)(W - 1) * sw + 1) * 4ull;
    CNL_REQUIRE(img_bytes < 0xFFFFFF00ull, CNL_E_UNSUPPORTED, "%s: one image spans >= 4 GiB", who);
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    const int tiles_x = (Wo + ST_TW - 1) / ST_TW, tiles_y = (Ho + ST_TH - 1) / ST_TH;
    const long long blocks = (long long)N * tiles_x * tiles_y;
    CNL_REQUIRE(blocks < (1ll << 31), CNL_E_UNSUPPORTED, "%s: grid too large", who);
    CNL_REQUIRE(algo <= CNL_ALGO_F32, CNL_E_BAD_ARG, "%s: unknown algo %u", who, algo);
    if (algo != CNL_ALGO_F32 || pool)           // (the fused max-pool exists in the fp16-split kernel only)
        return cnl_stem5_launch(x, false, nullptr, nullptr, (long)sn, (int)sc, (int)sh, (int)sw, (unsigned)img_bytes, w + ST_KP * 64, bias, y, y_absmax, N, H, W,
                                Ho, Wo, tiles_x, tiles_y, (unsigned)blocks, pool, stream);
    const int lds = ST_LDS_BYTES;
    static cnl::DeviceOnce once;
    const int rc = cnl::kernel_setup(once, (const void*)stem_conv_kernel, lds);
    if (rc != CNL_OK) return rc;
    hipLaunchKernelGGL(stem_conv_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, x, (long)sn, (int)sc,
                       (int)sh, (int)sw, (unsigned)img_bytes, w, bias, y, N, H, W, Ho, Wo, tiles_x, tiles_y);
    return cnl::check_launch("stem_conv_kernel");
}

extern "C" int cnl_stem_conv7x7_f32(const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, const float* w,
                                    const float* bias, float* y, float* y_absmax, int32_t N, int32_t H, int32_t W, uint32_t algo, void* stream) {
    return stem_launch("cnl_stem_conv7x7_f32", false, x, sn, sc, sh, sw, w, bias, y, y_absmax, N, H, W, algo, stream);
}

extern "C" int cnl_stem_conv7x7_maxpool_f32(const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, const float* w,
                                            const float* bias, float* y, float* y_absmax, int32_t N, int32_t H, int32_t W, void* stream) {
    return stem_launch("cnl_stem_conv7x7_maxpool_f32", true, x, sn, sc, sh, sw, w, bias, y, y_absmax, N, H, W, CNL_ALGO_AUTO, stream);
}

// uint8 frames straight into the stem: A.Normalize happens on the staged patch inside the kernel (stem_f16x2.hip)
extern "C" int cnl_stem_conv7x7_u8(const uint8_t* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, const float* mean255,
                                   const float* inv_std255, const float* w, const float* bias, float* y, float* y_absmax, int32_t N, int32_t H, int32_t W,
                                   int32_t fuse_maxpool, void* stream) {
    const char* who = "cnl_stem_conv7x7_u8";
    CNL_REQUIRE(x && w && bias && y && mean255 && inv_std255, CNL_E_BAD_ARG, "%s: null pointer", who);
    CNL_REQUIRE(N > 0 && H > 0 && W > 0, CNL_E_BAD_ARG, "%s: non-positive dimension", who);
    CNL_REQUIRE(sc > 0 && sh > 0 && sw > 0 && sn >= 0, CNL_E_UNSUPPORTED, "%s: non-positive strides", who);
    CNL_REQUIRE(((uintptr_t)w & 15) == 0, CNL_E_BAD_ARG, "%s: packed weights must be 16-byte aligned", who);
    const unsigned long long img_bytes = 2ull * sc + (unsigned long long)(H - 1) * sh + (unsigned long long)(W - 1) * sw + 1;
    CNL_REQUIRE(img_bytes < 0xFFFFFF00ull, CNL_E_UNSUPPORTED, "%s: one image spans >= 4 GiB", who);
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    const int tiles_x = (Wo + ST_TW - 1) / ST_TW, tiles_y = (Ho + ST_TH - 1) / ST_TH;
    const long long blocks = (long long)N * tiles_x * tiles_y;
    CNL_REQUIRE(blocks < (1ll << 31), CNL_E_UNSUPPORTED, "%s: grid too large", who);
    return cnl_stem5_launch(x, true, mean255, inv_std255, (long)sn, (int)sc, (int)sh, (int)sw, (unsigned)img_bytes, w + ST_KP * 64, bias, y, y_absmax, N, H, W, Ho,
                            Wo, tiles_x, tiles_y, (unsigned)blocks, fuse_maxpool != 0, stream);
}

extern "C" int cnl_maxpool3x3s2_nhwc_f32(const float* x, float* y, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    CNL_REQUIRE(x && y, CNL_E_BAD_ARG, "cnl_maxpool3x3s2_nhwc_f32: null tensor pointer");
    CNL_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, CNL_E_BAD_ARG, "cnl_maxpool3x3s2_nhwc_f32: non-positive dimension");
    CNL_REQUIRE(C % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, CNL_E_UNSUPPORTED,
                "cnl_maxpool3x3s2_nhwc_f32: C %% 4 != 0 or unaligned pointers");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long total = (long)N * Ho * Wo * (C / 4);
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const f32x4*)x, (f32x4*)y, N, H,
                       W, C / 4, Ho, Wo, total);
    return cnl::check_launch("maxpool_kernel");
}
