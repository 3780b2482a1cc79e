// conv_mfma.hip — fused Conv2d(+BN folded)+bias(+residual)(+ReLU|sigmoid) on NHWC fp32 for gfx950.
//
// Replaces the ATen/cuDNN conv2d + batch_norm + relu + add (+ upsample / sigmoid) call sites of the
// reference's forward (models/meta.py:21-47, models/layers.py:72-77,99,152,160-177; torchvision
// BasicBlock) — see include/centernet_gfx950.h for the per-call-site list.
//
// Algorithm: implicit GEMM, exact fp32 on the matrix cores.
//   C[m][co] = sum_k A[m][k] * B[co][k],   m = (n,oy,ox),  k = (ky,kx,ci),  K = KH*KW*Cin
//   A[m][k]  = x[n, oy*s+ky-p, ox*s+kx-p, ci]   (zero outside the image; optional nearest-2x source)
//   B[co][k] = w[co][ky][kx][ci]                (OHWI, K contiguous)
// Tiling: workgroup tile BM x BN, BK = 32 floats (one 128-byte row per pixel / output channel);
// each wave owns TM x TN accumulator tiles of 32x32 (v_mfma_f32_32x32x2_f32, 64-wide wavefront).
// Staging: buffer_load_dwordx4 ... lds (LDS-DMA, no VGPR round trip).  The DMA writes LDS
// lane-linearly, so the bank-conflict swizzle is applied on the SOURCE address (which 16-byte slot of
// the 128-byte row a lane fetches) and again on the ds_read_b128 address; out-of-image taps and
// M/Cout tails are produced by the buffer descriptor's bounds check (offset >= num_records -> 0).
// K order inside a BK chunk is permuted (lanes 0-31 take floats 8j..8j+3, lanes 32-63 take 8j+4..8j+7
// of read j) so one ds_read_b128 feeds four MFMAs; A and B use the same permutation, so the sum is
// unchanged up to fp32 summation order.
// Pipeline: 2 LDS stages, one barrier per K chunk, 2 workgroups per CU so one group's DMA wait
// overlaps the other's MFMAs.  Block ids are remapped so each XCD works on a contiguous run of tiles.
#include "cnl_common.h"

namespace cnl_conv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
    const float* x;
    const float* w;
    const float* bias;
    const float* res;
    float* y;
    int N, Hin, Win, Cin, Cout;
    int KH, KW, stride, pad;
    int ldx, ldy, ldr;
    int HL, WL;        // logical input size (2x when CNL_UPSAMPLE_IN)
    int Ho, Wo, M;     // conv output size, M = N*Ho*Wo
    int CC, KT, K;     // Cin/32, KH*KW*CC, KH*KW*Cin
    unsigned x_bytes, w_bytes;
    unsigned flags;
    int tiles_n, tiles;
};

constexpr unsigned OOB = 0xFFFFFFF0u;   // voffset that is always >= num_records -> DMA writes zeros

template <int WM, int WN, int TM, int TN>
struct Cfg {
    static constexpr int NW = WM * WN;
    static constexpr int THREADS = NW * 64;
    static constexpr int BM = WM * TM * 32;
    static constexpr int BN = WN * TN * 32;
    static constexpr int A_INSTR = BM / (NW * 8);   // buffer_load..lds instructions per wave for A
    static constexpr int B_INSTR = BN / (NW * 8);
    static constexpr int STAGE_BYTES = (BM + BN) * 128;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
    static_assert(BM % (NW * 8) == 0 && BN % (NW * 8) == 0, "tile rows must split evenly over waves");
    static_assert(THREADS == 256, "kernel is declared __launch_bounds__(256, 2)");
};

typedef __attribute__((address_space(3))) void lds_void;

// amdgcn builtins are wrapped in NON-template device functions: called with template-dependent arguments
// directly inside the kernel template they make hipcc's host pass silently drop the kernel's host stub.
__device__ __forceinline__ void dma16(const float* base, unsigned bytes, char* lds_dst, unsigned voffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_dst, 16, voffset, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

template <int WM, int WN, int TM, int TN, bool UP_IN>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvArgs a) {
    using C = Cfg<WM, WN, TM, TN>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int hi = lane >> 5;

    const unsigned tile = cnl::xcd_remap(blockIdx.x, (unsigned)a.tiles);
    const int n_tile = tile % a.tiles_n;
    const int m_tile = tile / a.tiles_n;
    const int m0 = m_tile * C::BM;
    const int n0 = n_tile * C::BN;

    // ---- per-lane staging bookkeeping: which rows this lane fetches, and which 16-B slot ----
    const int lrow = lane >> 3;      // row within the 8-row group one DMA instruction covers
    const int pslot = lane & 7;      // physical 16-B slot inside the 128-B LDS row
    int a_iy0[C::A_INSTR], a_ix0[C::A_INSTR], a_pix[C::A_INSTR], a_q[C::A_INSTR];
#pragma unroll
    for (int j = 0; j < C::A_INSTR; ++j) {
        const int r = (j * C::NW + wave) * 8 + lrow;
        const int m = m0 + r;
        a_q[j] = (pslot ^ ((r >> 1) & 7)) * 4;          // logical slot (in floats) fetched into pslot
        if (m < a.M) {
            const int hw = a.Ho * a.Wo;
            const int n = m / hw;
            const int rem = m - n * hw;
            const int oy = rem / a.Wo;
            const int ox = rem - oy * a.Wo;
            a_iy0[j] = oy * a.stride - a.pad;
            a_ix0[j] = ox * a.stride - a.pad;
            a_pix[j] = n * a.Hin * a.Win;
        } else {
            a_iy0[j] = -0x40000000;                     // never valid
            a_ix0[j] = 0;
            a_pix[j] = 0;
        }
    }
    unsigned b_off[C::B_INSTR];
#pragma unroll
    for (int j = 0; j < C::B_INSTR; ++j) {
        const int r = (j * C::NW + wave) * 8 + lrow;
        const int q = (pslot ^ ((r >> 1) & 7)) * 4;
        b_off[j] = (unsigned)(((n0 + r) * a.K + q) * 4);   // rows >= Cout land beyond w_bytes -> zeros
    }

    // LDS-DMA of K chunk (ky,kx,c0) / kbase into `stage`.  (A macro, not a lambda: a lambda holding amdgcn
    // builtins inside a kernel template silently blocks hipcc's host-side instantiation of the kernel stub.)
#define CNL_ISSUE(stage_, ky_, kx_, c0_, kbase_)                                                                 \
    do {                                                                                                         \
        char* sA_ = smem + (stage_) * C::STAGE_BYTES;                                                            \
        char* sB_ = sA_ + C::BM * 128;                                                                           \
        _Pragma("unroll") for (int j = 0; j < C::A_INSTR; ++j) {                                                 \
            const int iy = a_iy0[j] + (ky_);                                                                     \
            const int ix = a_ix0[j] + (kx_);                                                                     \
            const bool ok = (unsigned)iy < (unsigned)a.HL && (unsigned)ix < (unsigned)a.WL;                      \
            const int sy = UP_IN ? (iy >> 1) : iy;                                                               \
            const int sx = UP_IN ? (ix >> 1) : ix;                                                               \
            const unsigned off = (unsigned)(((a_pix[j] + sy * a.Win + sx) * a.ldx + (c0_) + a_q[j]) * 4);        \
            dma16(a.x, a.x_bytes, sA_ + (j * C::NW + wave) * 1024, ok ? off : OOB);                              \
        }                                                                                                        \
        _Pragma("unroll") for (int j = 0; j < C::B_INSTR; ++j) {                                                 \
            dma16(a.w, a.w_bytes, sB_ + (j * C::NW + wave) * 1024, b_off[j] + (unsigned)((kbase_) * 4));         \
        }                                                                                                        \
    } while (0)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read addresses (bytes inside a stage): row * 128 + ((2*jj + hi) ^ swz) * 16
    const int swz = (lane >> 1) & 7;
    const int a_row_byte = (wm * TM * 32 + (lane & 31)) * 128;
    const int b_row_byte = C::BM * 128 + (wn * TN * 32 + (lane & 31)) * 128;

    int ky = 0, kx = 0, cc = 0;     // position of the chunk being ISSUED
    CNL_ISSUE(0, 0, 0, 0, 0);
    for (int kt = 0; kt < a.KT; ++kt) {
        const int stage = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA for chunk kt has landed
        __syncthreads();                                    // everyone's has; stage^1 is free again
        if (kt + 1 < a.KT) {
            if (++cc == a.CC) {
                cc = 0;
                if (++kx == a.KW) { kx = 0; ++ky; }
            }
            CNL_ISSUE(stage ^ 1, ky, kx, cc * 32, (kt + 1) * 32);
        }
        const char* sS = smem + stage * C::STAGE_BYTES;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int slot_byte = (((2 * jj + hi) ^ swz) << 4);
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const f32x4*>(sS + a_row_byte + i * 32 * 128 + slot_byte);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[j] = *reinterpret_cast<const f32x4*>(sS + b_row_byte + j * 32 * 128 + slot_byte);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = mfma32(af[i][c], bf[j][c], acc[i][j]);
        }
    }

#undef CNL_ISSUE
    // ---- epilogue: + bias (+ residual) (ReLU | sigmoid), NHWC store ----
    // C/D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const bool relu = a.flags & CNL_RELU;
    const bool sigm = a.flags & CNL_SIGMOID;
    const bool up_out = a.flags & CNL_UPSAMPLE_OUT_ADD;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
        const bool col_ok = col < a.Cout;
        const float bv = col_ok ? a.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + (wm * TM + i) * 32 + 4 * hi;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                if (!(col_ok && m < a.M)) continue;
                float v = acc[i][j][r] + bv;
                if (!up_out) {
                    if (a.res) v += a.res[(size_t)m * a.ldr + col];
                    if (relu) v = fmaxf(v, 0.f);
                    if (sigm) v = 1.0f / (1.0f + expf(-v));
                    a.y[(size_t)m * a.ldy + col] = v;
                } else {
                    const int hw = a.Ho * a.Wo;
                    const int n = m / hw;
                    const int rem = m - n * hw;
                    const int oy = rem / a.Wo;
                    const int ox = rem - oy * a.Wo;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const size_t pix = ((size_t)n * (2 * a.Ho) + 2 * oy + (d >> 1)) * (2 * a.Wo) + 2 * ox + (d & 1);
                        float u = v + a.res[pix * a.ldr + col];
                        if (relu) u = fmaxf(u, 0.f);
                        a.y[pix * a.ldy + col] = u;
                    }
                }
            }
        }
    }
}

template <int WM, int WN, int TM, int TN, bool UP_IN>
int launch_one(const ConvArgs& a, hipStream_t stream) {
    using C = Cfg<WM, WN, TM, TN>;
    static bool attr_done = false;
    if (!attr_done) {
        CNL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<WM, WN, TM, TN, UP_IN>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, TM, TN, UP_IN>), dim3(a.tiles), dim3(C::THREADS), C::LDS_BYTES, stream, a);
    return cnl::check_launch("conv_mfma_kernel");
}

template <int WM, int WN, int TM, int TN>
int launch_cfg(const ConvArgs& in, hipStream_t stream) {
    using C = Cfg<WM, WN, TM, TN>;
    ConvArgs a = in;
    const int tiles_m = (a.M + C::BM - 1) / C::BM;
    a.tiles_n = (a.Cout + C::BN - 1) / C::BN;
    a.tiles = tiles_m * a.tiles_n;
    if (a.flags & CNL_UPSAMPLE_IN) return launch_one<WM, WN, TM, TN, true>(a, stream);
    return launch_one<WM, WN, TM, TN, false>(a, stream);
}

}  // namespace cnl_conv
using namespace cnl_conv;

extern "C" int cnl_conv2d_out_hw(const cnl_conv_params* p, int32_t* H_out, int32_t* W_out) {
    CNL_REQUIRE(p && H_out && W_out, CNL_E_BAD_ARG, "cnl_conv2d_out_hw: null argument");
    CNL_REQUIRE(p->stride > 0 && p->KH > 0 && p->KW > 0, CNL_E_BAD_ARG, "cnl_conv2d_out_hw: bad kernel/stride");
    const int up = (p->flags & CNL_UPSAMPLE_IN) ? 2 : 1;
    *H_out = (p->H_in * up + 2 * p->pad - p->KH) / p->stride + 1;
    *W_out = (p->W_in * up + 2 * p->pad - p->KW) / p->stride + 1;
    return CNL_OK;
}

extern "C" int cnl_conv2d_nhwc_f32(const cnl_conv_params* p, void* stream) {
    CNL_REQUIRE(p, CNL_E_BAD_ARG, "cnl_conv2d_nhwc_f32: null params");
    CNL_REQUIRE(p->x && p->w && p->bias && p->y, CNL_E_BAD_ARG, "cnl_conv2d_nhwc_f32: null tensor pointer");
    CNL_REQUIRE(p->N > 0 && p->H_in > 0 && p->W_in > 0 && p->Cin > 0 && p->Cout > 0, CNL_E_BAD_ARG,
                "cnl_conv2d_nhwc_f32: non-positive dimension");
    CNL_REQUIRE(p->KH > 0 && p->KW > 0 && p->stride > 0 && p->pad >= 0, CNL_E_BAD_ARG,
                "cnl_conv2d_nhwc_f32: bad kernel/stride/pad");
    CNL_REQUIRE(p->Cin % 32 == 0, CNL_E_UNSUPPORTED,
                "cnl_conv2d_nhwc_f32: Cin=%d is not a multiple of 32 (use cnl_stem_conv7x7_f32 for the RGB stem)", p->Cin);
    CNL_REQUIRE(p->ldx >= p->Cin && p->ldy >= p->Cout && p->ldx % 4 == 0, CNL_E_BAD_ARG,
                "cnl_conv2d_nhwc_f32: pixel strides ldx=%d ldy=%d too small / misaligned", p->ldx, p->ldy);
    CNL_REQUIRE(((uintptr_t)p->x & 15) == 0 && ((uintptr_t)p->w & 15) == 0, CNL_E_BAD_ARG,
                "cnl_conv2d_nhwc_f32: x and w must be 16-byte aligned");
    const bool up_out = p->flags & CNL_UPSAMPLE_OUT_ADD;
    CNL_REQUIRE(!up_out || p->residual, CNL_E_BAD_ARG, "cnl_conv2d_nhwc_f32: CNL_UPSAMPLE_OUT_ADD needs a residual");
    CNL_REQUIRE(!up_out || !(p->flags & CNL_SIGMOID), CNL_E_UNSUPPORTED, "cnl_conv2d_nhwc_f32: sigmoid with UPSAMPLE_OUT_ADD");
    CNL_REQUIRE(!p->residual || p->ldr >= p->Cout, CNL_E_BAD_ARG, "cnl_conv2d_nhwc_f32: ldr=%d < Cout", p->ldr);

    ConvArgs a;
    a.x = p->x; a.w = p->w; a.bias = p->bias; a.res = p->residual; a.y = p->y;
    a.N = p->N; a.Hin = p->H_in; a.Win = p->W_in; a.Cin = p->Cin; a.Cout = p->Cout;
    a.KH = p->KH; a.KW = p->KW; a.stride = p->stride; a.pad = p->pad;
    a.ldx = p->ldx; a.ldy = p->ldy; a.ldr = p->ldr;
    a.flags = p->flags;
    const int up = (p->flags & CNL_UPSAMPLE_IN) ? 2 : 1;
    a.HL = p->H_in * up; a.WL = p->W_in * up;
    a.Ho = (a.HL + 2 * p->pad - p->KH) / p->stride + 1;
    a.Wo = (a.WL + 2 * p->pad - p->KW) / p->stride + 1;
    CNL_REQUIRE(a.Ho > 0 && a.Wo > 0, CNL_E_BAD_ARG, "cnl_conv2d_nhwc_f32: empty output");
    const long long M = (long long)p->N * a.Ho * a.Wo;
    CNL_REQUIRE(M < (1ll << 31) - 512, CNL_E_UNSUPPORTED, "cnl_conv2d_nhwc_f32: N*Ho*Wo too large");
    a.M = (int)M;
    a.CC = p->Cin / 32; a.KT = p->KH * p->KW * a.CC; a.K = p->KH * p->KW * p->Cin;
    const unsigned long long xb = (((unsigned long long)p->N * p->H_in * p->W_in - 1) * p->ldx + p->Cin) * 4ull;
    const unsigned long long wb = (unsigned long long)p->Cout * a.K * 4ull;
    CNL_REQUIRE(xb < 0xFFFFFF00ull, CNL_E_UNSUPPORTED,
                "cnl_conv2d_nhwc_f32: input spans %llu bytes; split the batch so it stays below 4 GiB", xb);
    CNL_REQUIRE(wb + (unsigned long long)256 * a.K * 4ull < 0xFFFFFF00ull, CNL_E_UNSUPPORTED, "cnl_conv2d_nhwc_f32: weight too large");
    a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb;
    a.tiles_n = a.tiles = 0;

    hipStream_t s = (hipStream_t)stream;
    // Tile choice: BN follows Cout; shrink BM when the grid would not fill 256 CUs x 2 workgroups.
    if (a.Cout <= 32) return launch_cfg<4, 1, 2, 1>(a, s);          // 256 x 32
    if (a.Cout <= 64) return launch_cfg<4, 1, 2, 2>(a, s);          // 256 x 64
    const long long tiles128 = ((M + 127) / 128) * ((a.Cout + 127) / 128);
    if (tiles128 < 512) return launch_cfg<2, 2, 1, 2>(a, s);        //  64 x 128
    return launch_cfg<2, 2, 2, 2>(a, s);                            // 128 x 128
}
