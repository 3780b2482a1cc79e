// winograd8.hip — Winograd F(4x4,3x3) on the FP16 matrix cores at fp32-grade accuracy (scaled two-way fp16 split, three MFMAs per product).
//
// winograd5.hip (F(2x2,3x3)) is bound by the VALU work of its input transform + split (292 VALU beside 48 MFMAs per 16-channel chunk)
// and by power: only less work per output converts into time.  F(4x4,3x3) needs 36 transform-domain elements per 16 outputs where
// F(2x2) needs 64: 0.56x the matrix work, 0.56x the elements to split and to move through LDS.  The price is rounding error: the
// transforms are no longer sums of +-1 terms.  With the interpolation points (0, +-5/8, +-3/2, inf) — chosen on a CPU emulation of
// this arithmetic (tools/wino_f4_numerics.py; the textbook points 0, +-1, +-2 are 3.5x worse) — the error against float64 is
// ~1e-6 of the layer's largest output on a K = 2304 layer (F(2x2) split: 2.7e-7, fp32 direct sum: 3.2e-7): inside the path's 1e-4
// bar by two orders of magnitude, but above the fp32 MFMA's, so the kernel is taken only where the plan asks for it
// (cnl_conv_params.algo: CNL_ALGO_AUTO allows it, CNL_ALGO_F2 / CNL_ALGO_F32 do not) and the end-to-end feature-error gate
// (tests/test_gpu_e2e.py) bounds what it may cost.
//
// Arithmetic.  Y = A^T [ sum_ci (G g G^T) . (B^T d B) ] A with, for the points p = (0, a, -a, b, -b) and infinity:
//   B^T rows   [a^2 b^2, 0, -(a^2+b^2), 0, 1, 0]; [0, -+a b^2, -b^2, +-a, 1, 0]; [0, -+a^2 b, -a^2, +-b, 1, 0]; [0, a^2 b^2, 0, -(a^2+b^2), 0, 1]
//   G rows     [1, p, p^2] / prod_{k != j}(p_j - p_k),  [0, 0, 1]          (applied in float64 when the weights are loaded)
//   A^T        [1, p, p^2, p^3] per point,  [0, 0, 0, 1] for infinity
// A +-p pair shares its even and odd part: 12 fused multiply-adds per 6-point transform for any symmetric point set.
// V = B^T d B is scaled per IMAGE by a power of two S (|V| <= 27.9 max |x|; 32 max |x| S in [2^14, 2^15)) and split V S = hi + lo
// (hi = RN16, lo = RZ16 of the exact residual), U likewise with its own scale when the weights are transformed; three terms
// hi lo' + lo hi' + hi hi' on v_mfma_f32_32x32x16_f16, fp32 accumulation, 1 / (S S_u) applied in the epilogue (exact).
//
// Work item: 8 x 4 tiles (32 x 16 output pixels) x 64 couts x 36 positions = 288 accumulator registers per lane: one 4-wave
// workgroup per CU, one wave per SIMD, wave w owns positions 9w .. 9w+8.  Per 16-channel chunk:
//   barrier | all threads: B^T d B of the 18 x 34 x 16 patch (thread = tile x channel quad x row half: 30 ds_read_b128, 288 fma,
//   180 split operations, 36 ds_write_b64) -> V [36][piece][32 tiles][16 ci] fp16 in LDS | barrier | LDS-DMA of the patch two chunks
//   ahead | 54 MFMAs per wave: A fragments from V, B fragments global -> registers two positions ahead.
// Epilogue: each wave reduces its positions along x in registers (a full row of six, or a half row), the eight row parts meet through
// LDS (128 KB per cout group), thread = (tile, 4 couts) finishes A^T . A, adds bias (+ residual), ReLU, 16-byte non-temporal stores.
#include "cnl_common.h"

#pragma clang fp contract(off)

#ifndef W8_NT_Y
#define W8_NT_Y 2
#endif
#ifndef W8_NT_X
#define W8_NT_X 0     /* cache policy (aux) of the patch DMA loads: 2 = nt (streaming: do not displace the weights in L2) */
#endif
#ifndef W8_ORDER
#define W8_ORDER 2    /* work-item order: 0 cout block fastest, 1 cout block slowest inside an image, 2 pairs of cout blocks fastest */
#endif
#ifndef W8_PK
#define W8_PK 1       /* packed fp32 fma in the input transform */
#endif

namespace cnl_wino8 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

struct Args {
    const float* x;
    const void* u8;                   // pre-split, pre-scaled weights (fp16 pieces): [Cin/16][36][2][CoutP][16]
    const float* xmax;                // per-image max |x| (handed over by the producer, or an own pass)
    const float* su;                  // scale of the weights
    unsigned* ymax;                   // optional: per-image max |y| for the consumer (atomic max on the bits)
    const float* bias;
    const float* res;
    float* y;
    int N, H, W, Cin, Cout, CoutP;
    int Hs, Ws;                       // stored input size (H/2, W/2 with CNL_UPSAMPLE_IN)
    int ldx, ldy, ldr;
    int CC;                           // Cin / 16
    int nb, bx, by;                   // blocks along cout (64), x (32 px), y (16 px)
    int blocks;
    unsigned x_bytes, u_bytes, y_bytes, r_bytes;
    unsigned flags;
};

constexpr float PA = 0.625f, PB = 1.5f;                   // interpolation points 0, +-PA, +-PB, infinity
constexpr float A2 = PA * PA, B2 = PB * PB, A2B2 = A2 * B2, SAB = A2 + B2, A3 = A2 * PA, B3 = B2 * PB;

constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr int TX = 8, TY = 4, T = TX * TY;                // tiles per work item
constexpr int BN = 64;
constexpr int NPOS = 36, NP = 2;
constexpr int PH = 4 * TY + 2, PW = 4 * TX + 2;           // 18 x 34 patch
constexpr int PP = 35;                                    // slots per (row, quad) line of the LDS image [py][quad][PP][4 floats]
constexpr int P_SLOTS = 2560;                             // 10 x 256 (PH * 4 * PP = 2520 used)
static_assert(PH * 4 * PP <= P_SLOTS, "patch image exceeds its LDS buffer");
constexpr int P_BYTES = P_SLOTS * 16;                     // 40960 per buffer
constexpr int VPIECE = T * 32;                            // 1024: one (position, piece) plane [32 tiles][16 ci fp16]
constexpr int V_BYTES = NPOS * NP * VPIECE;               // 73728
constexpr int LDS_BYTES = V_BYTES + 2 * P_BYTES;          // 155648
constexpr int Q_BYTES = 8 * 4 * 32 * 32 * 4;              // 131072: the epilogue's exchange buffer (overlays V and the patches)
static_assert(Q_BYTES <= LDS_BYTES, "epilogue exchange buffer must fit");

__device__ __forceinline__ void dma16(const float* base, unsigned bytes, char* lds_dst, unsigned voffset, unsigned soffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_dst, 16, voffset, soffset, 0, W8_NT_X);
}
__device__ __forceinline__ u32x4 buf_load16(const void* base, unsigned bytes, unsigned voffset, unsigned soffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    return (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, soffset, 0);
}
__device__ __forceinline__ void buf_store16(f32x4 v, float* base, unsigned bytes, unsigned voffset, unsigned soffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, voffset, soffset, W8_NT_Y);
}
__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// The same MFMA with its accumulator pinned to ARCHITECTURAL VGPRs.  The kernel holds 18 accumulator tiles = 288 registers per lane, 32
// more than the 256 AGPRs; left to itself the compiler keeps two tiles in VGPRs and copies them through AGPRs around their MFMAs (80
// v_accvgpr moves + result-wait nops per chunk).  The "+v" constraint selects the VGPR form of the instruction for the ninth position of a
// wave.  The operands come from ds_read / buffer_load (s_waitcnt is inserted on registers, also for inline asm); the s_nop covers the
// VALU -> MFMA distance in case the compiler moved an operand with v_mov just before (the hazard recognizer does not look inside asm).
__device__ __forceinline__ void mfma16_vgpr(u32x4 a, u32x4 b, f32x16& c) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// split of a channel pair (winograd5.hip): hi = RN16(v S) packed, r = v S - hi exactly, lo = RZ16(r) packed.  The results go to LDS
// (never straight into an MFMA operand), so the inline asm is outside the VALU -> MFMA hazard window the compiler cannot see into.
__device__ __forceinline__ unsigned split_hi_lo(float v0, float S) {
    unsigned pk;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(pk) : "v"(v0), "v"(S));
    return pk;
}
__device__ __forceinline__ unsigned split_hi_hi(unsigned pk, float v1, float S) {
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(pk) : "v"(v1), "v"(S));
    return pk;
}
__device__ __forceinline__ float split_res_lo(float v, float S, unsigned pk) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(v), "v"(S), "v"(pk));
    return r;
}
__device__ __forceinline__ float split_res_hi(float v, float S, unsigned pk) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(v), "v"(S), "v"(pk));
    return r;
}
__device__ __forceinline__ f32x4 lds_f4(const char* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ u32x4 lds_u4(const char* p) { return *reinterpret_cast<const u32x4*>(p); }
// c * x + y on four channels as two v_pk_fma_f32: the transform phase issues no MFMAs (all four waves transform at the same time), so
// the packed form — an anti-lever beside MFMAs — simply halves the instruction count of a phase that is bound by instruction issue
// (one wave per SIMD issues ~one instruction per 4 cycles whatever its width).  IEEE fma per element: bit-identical to fmaf.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 fma4(float c, f32x4 x, f32x4 y) {
#if W8_PK
    const f32x2 cc = {c, c};
    f32x2 lo, hi;
    const f32x2 xl = {x[0], x[1]}, xh = {x[2], x[3]}, yl = {y[0], y[1]}, yh = {y[2], y[3]};
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(cc), "v"(xl), "v"(yl));
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(cc), "v"(xh), "v"(yh));
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
#else
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = __builtin_fmaf(c, x[e], y[e]);
    return r;
#endif
}

// split four channels of one V element group and store both pieces: dst = this thread's 8 bytes of the (position, piece 0) plane
__device__ __forceinline__ void split_store(const f32x4 v, const float S, char* dst) {
    unsigned h0 = split_hi_lo(v[0], S), h1 = split_hi_lo(v[2], S);
    h0 = split_hi_hi(h0, v[1], S);
    h1 = split_hi_hi(h1, v[3], S);
    const float r0 = split_res_lo(v[0], S, h0), r1 = split_res_hi(v[1], S, h0);
    const float r2 = split_res_lo(v[2], S, h1), r3 = split_res_hi(v[3], S, h1);
    const unsigned l0 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
    const unsigned l1 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r2, r3));
    *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(dst + VPIECE) = u32x2{l0, l1};
}

// six-point transform  out = B^T in  for the symmetric point set (12 fma on four channels each)
__device__ __forceinline__ void bt6(const f32x4 t0, const f32x4 t1, const f32x4 t2, const f32x4 t3, const f32x4 t4, const f32x4 t5,
                                    f32x4 (&v)[6]) {
    const f32x4 ea = fma4(-B2, t2, t4), oa = fma4(-B2, t1, t3);
    const f32x4 eb = fma4(-A2, t2, t4), ob = fma4(-A2, t1, t3);
    v[0] = fma4(A2B2, t0, fma4(-SAB, t2, t4));
    v[1] = fma4(PA, oa, ea);
    v[2] = fma4(-PA, oa, ea);
    v[3] = fma4(PB, ob, eb);
    v[4] = fma4(-PB, ob, eb);
    v[5] = fma4(A2B2, t1, fma4(-SAB, t3, t5));
}

// Input transform of one chunk for this thread's (tile, channel quad): rows 3 HALF .. 3 HALF + 2 of B^T d, all six columns.
// src = the thread's d[0][0] in the patch image, dst = its 8 bytes in the (position 0, piece 0) plane of V.
// (Measured and rejected, profiles/r02_winograd8_variants.txt: the same work as two passes over channel pairs — half the live
// registers, so that weight fragments could be requested DURING the transform — was 13 % slower per transform and bought nothing:
// the weight stream is bound by the L2 -> CU path, not by the loads in flight.)
template <int HALF>
__device__ __forceinline__ void transform_chunk(const char* src, char* dst, const float S) {
    f32x4 t[3][6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const f32x4 d1 = lds_f4(src + (1 * 4 * PP + b) * 16), d2 = lds_f4(src + (2 * 4 * PP + b) * 16);
        const f32x4 d3 = lds_f4(src + (3 * 4 * PP + b) * 16), d4 = lds_f4(src + (4 * 4 * PP + b) * 16);
        if (HALF == 0) {
            const f32x4 d0 = lds_f4(src + b * 16);
            const f32x4 e = fma4(-B2, d2, d4), o = fma4(-B2, d1, d3);
            t[0][b] = fma4(A2B2, d0, fma4(-SAB, d2, d4));
            t[1][b] = fma4(PA, o, e);
            t[2][b] = fma4(-PA, o, e);
        } else {
            const f32x4 d5 = lds_f4(src + (5 * 4 * PP + b) * 16);
            const f32x4 e = fma4(-A2, d2, d4), o = fma4(-A2, d1, d3);
            t[0][b] = fma4(PB, o, e);
            t[1][b] = fma4(-PB, o, e);
            t[2][b] = fma4(A2B2, d1, fma4(-SAB, d3, d5));
        }
    }
#pragma unroll
    for (int ri = 0; ri < 3; ++ri) {
        f32x4 v[6];
        bt6(t[ri][0], t[ri][1], t[ri][2], t[ri][3], t[ri][4], t[ri][5], v);
#pragma unroll
        for (int j = 0; j < 6; ++j) split_store(v[j], S, dst + ((3 * HALF + ri) * 6 + j) * NP * VPIECE);
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void winograd8_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sV = smem;                                  // [36 positions][2 pieces][32 tiles][16 ci] fp16
    char* sP = smem + V_BYTES;                        // [2][18 py][4 quads][35 px][4 ci] fp32 (+ slack)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5;
    const bool up = a.flags & CNL_UPSAMPLE_IN;
    const unsigned u_piece = (unsigned)(a.CoutP * 32);
    const unsigned u_pos = (unsigned)NP * u_piece;
    const unsigned u_chunk = (unsigned)NPOS * u_pos;

    // transform jobs: the 16-lane groups of a ds_read_b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31}, + 32) each hold four neighbouring
    // tile columns x four channel quads; with the 35-slot line pitch their slots cover all 16 residues (no bank conflicts)
    int t_grp, t_r;
    {
        const int l5 = lane & 31;
        if (l5 < 4) { t_grp = 0; t_r = l5; }
        else if (l5 < 12) { t_grp = 1; t_r = l5 - 4; }
        else if (l5 < 16) { t_grp = 0; t_r = l5 - 8; }
        else if (l5 < 20) { t_grp = 1; t_r = l5 - 8; }
        else if (l5 < 28) { t_grp = 0; t_r = l5 - 12; }
        else { t_grp = 1; t_r = l5 - 16; }
    }
    const int t_tx = (t_r & 3) | (t_grp << 2), t_q = t_r >> 2, t_ty = hi | ((wave & 1) << 1);
    const int t_half = wave >> 1;                                         // wave-uniform: rows 0-2 or 3-5 of B^T d
    const int src0 = (((4 * t_ty) * 4 + t_q) * PP + 4 * t_tx) * 16;
    const int t_tile = t_ty * TX + t_tx;
    // V rows are 32 bytes = two 16-byte halves (ci 0-7 | 8-15); rows of odd tile rows store them swapped (conflict-free b128 fragment reads)
    const int dstv = t_tile * 32 + (((t_q >> 1) ^ (t_ty & 1)) << 4) + (t_q & 1) * 8;
    const int fragA = (lane & 31) * 32 + ((hi ^ ((lane >> 3) & 1)) << 4);
    const float lo = (a.flags & CNL_RELU) ? 0.f : -__builtin_inff();
    const float Su = a.su[0];

    float omax = 0.f;
    // ---- work-item state (set by W8_SETUP for the item about to run; the epilogue works on copies) ----
    int n, y0, x0, n0;
    unsigned p_off[10], u_voff;
    float S, inv;
#define W8_SETUP(item_)                                                                                          \
    do {                                                                                                         \
        /* pairs of cout blocks fastest, then the tile, then the pair index (as winograd5.hip): the two workgroups that share an    \
           input patch run side by side; cout-block-slowest orders (one slice of U per XCD at a time) measured 4 % slower */        \
        unsigned b_ = cnl::xcd_remap((item_), (unsigned)a.blocks);                                               \
        int nbi_, bxi_, byi_;                                                                                    \
        if (W8_ORDER == 0 || (W8_ORDER == 2 && (a.nb & 1))) {       /* cout block fastest */                     \
            nbi_ = b_ % a.nb; b_ /= a.nb; bxi_ = b_ % a.bx; b_ /= a.bx; byi_ = b_ % a.by; n = b_ / a.by;         \
        } else if (W8_ORDER == 1) {                                 /* cout block slowest inside an image */     \
            bxi_ = b_ % a.bx; b_ /= a.bx; byi_ = b_ % a.by; b_ /= a.by; nbi_ = b_ % a.nb; n = b_ / a.nb;         \
        } else {                                                    /* pairs of cout blocks fastest */           \
            const int np_ = a.nb / 2;                                                                            \
            const int lo_ = b_ % 2; b_ /= 2; bxi_ = b_ % a.bx; b_ /= a.bx; byi_ = b_ % a.by; b_ /= a.by;         \
            const int pr_ = b_ % np_; n = b_ / np_; nbi_ = pr_ * 2 + lo_;                                        \
        }                                                                                                        \
        y0 = byi_ * (4 * TY); x0 = bxi_ * (4 * TX); n0 = nbi_ * BN;                                              \
        _Pragma("unroll") for (int i = 0; i < 10; ++i) {                                                         \
            const int s_ = i * 256 + tid;              /* 16-byte slot of the patch image: (py*4 + quad)*PP + px */ \
            const int rowq_ = s_ / PP, pxx_ = s_ - rowq_ * PP;                                                   \
            const int py_ = rowq_ >> 2, q_ = rowq_ & 3;                                                          \
            const int iy_ = y0 - 1 + py_, ix_ = x0 - 1 + pxx_;                                                   \
            const bool ok_ = py_ < PH && pxx_ < PW && (unsigned)iy_ < (unsigned)a.H && (unsigned)ix_ < (unsigned)a.W; \
            const int sy_ = up ? (iy_ >> 1) : iy_, sx_ = up ? (ix_ >> 1) : ix_;                                  \
            p_off[i] = ok_ ? (unsigned)((((n * a.Hs + sy_) * a.Ws + sx_) * a.ldx + q_ * 4) * 4) : OOB;           \
        }                                                                                                        \
        u_voff = (unsigned)((n0 + (lane & 31)) * 32 + hi * 16);                                                  \
        S = 1.f;                                                                                                 \
        {                                                                                                        \
            const float mx_ = 32.f * a.xmax[n];                                                                  \
            if (mx_ > 0.f && mx_ < __builtin_inff()) {                                                           \
                int e_;                                                                                          \
                (void)__builtin_frexpf(mx_, &e_);            /* 2^(e-1) <= mx < 2^e */                           \
                e_ = 15 - e_;                                                                                    \
                S = __builtin_ldexpf(1.f, e_ < -100 ? -100 : (e_ > 100 ? 100 : e_));                             \
            }                                                                                                    \
        }                                                                                                        \
        inv = 1.f / (S * Su);                                                                                    \
    } while (0)
#define W8_ISSUE_P(cc_)                                                                                          \
    do {                                                                                                         \
        if ((cc_) < a.CC) {                                                                                      \
            char* d_ = sP + ((cc_) & 1) * P_BYTES;                                                               \
            _Pragma("unroll") for (int i = 0; i < 10; ++i)                                                       \
                dma16(a.x, a.x_bytes, d_ + (i * 256 + wave * 64) * 16, p_off[i], (unsigned)((cc_) * 64));  \
        }                                                                                                        \
    } while (0)
    // the four B fragments of position j_ of chunk cc_ into ring buffer (j_) % 3
#define W8_LOAD_B(cc_, j_)                                                                                       \
    do {                                                                                                         \
        const unsigned so_ = (unsigned)(cc_) * u_chunk + (unsigned)(9 * wave + (j_)) * u_pos;              \
        _Pragma("unroll") for (int g_ = 0; g_ < 2; ++g_)                                                         \
            _Pragma("unroll") for (int kk_ = 0; kk_ < NP; ++kk_)                                                 \
                fb[(j_) % 3][g_][kk_] = buf_load16(a.u8, a.u_bytes, u_voff, so_ + (unsigned)g_ * 1024u + (unsigned)kk_ * u_piece); \
    } while (0)
#define W8_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    unsigned item = blockIdx.x;
    while (true) {
        W8_SETUP(item);
        W8_BARRIER();                                   // the previous item's epilogue is done with the exchange buffer
        W8_ISSUE_P(0);
        W8_ISSUE_P(1);
        f32x16 acc[9][2];
#pragma unroll
        for (int j = 0; j < 9; ++j)
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][g][r] = 0.f;
        u32x4 fb[3][2][NP];                             // B fragments: ring of three positions x cout group x piece
        W8_LOAD_B(0, 0);
        W8_LOAD_B(0, 1);
        // the item's first two patches landed (this wave's part): their DMAs are older than these 8 loads (VMEM returns in order)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        for (int cc = 0; cc < a.CC; ++cc) {
            // patch cc landed (this wave's part): its DMAs were issued in the last slices of chunk cc-2's MFMA phase, before B loads
            // that have been consumed since
            W8_BARRIER();                               // everyone's part; and every wave is done with V of the previous chunk
            {
                const char* src = sP + (cc & 1) * P_BYTES + src0;
                if (t_half == 0) transform_chunk<0>(src, sV + dstv, S);
                else transform_chunk<1>(src, sV + dstv, S);
            }
            W8_BARRIER();                               // V complete; the patch buffer of this chunk is free
            W8_ISSUE_P(cc + 2);                         // the patch two chunks ahead, into the buffer this chunk's transform has read
            // MFMA phase: 9 position slots x 6 slices of ONE MFMA + at most one memory instruction issued in its shadow, fenced so
            // that the compiler keeps the software pipeline (left alone it sinks every B load to just before its first use: one L2
            // round trip per MFMA pair).  B fragments two positions ahead (ring of three; slots 7 / 8 fetch positions 0 / 1 of the
            // NEXT chunk, which then have the whole transform phase to arrive), A fragments one position ahead; the two cout
            // groups alternate so that consecutive MFMAs never share an accumulator.  What the phase waits for is the weight stream
            // itself — 147 KB per chunk and CU at the ~27 B/clk/CU the L2 -> CU path delivers, whatever the number of loads in flight
            // (all nine positions requested at once: slower) and whatever the work-item order (profiles/r02_winograd8_variants.txt).
            u32x4 fa[2][NP];
            {
                {
                    const char* va = sV + (9 * wave) * NP * VPIECE + fragA;
                    fa[0][0] = lds_u4(va);
                    fa[0][1] = lds_u4(va + VPIECE);
                }
#pragma unroll
                for (int j = 0; j < 9; ++j) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        __builtin_amdgcn_sched_barrier(0);
                        const int t = k >> 1, g = k & 1;   // terms hi lo', lo hi', hi hi'
                        if (j == 8) mfma16_vgpr(fa[j & 1][t == 1 ? 1 : 0], fb[j % 3][g][t == 0 ? 1 : 0], acc[j][g]);
                        else acc[j][g] = mfma16(fa[j & 1][t == 1 ? 1 : 0], fb[j % 3][g][t == 0 ? 1 : 0], acc[j][g]);
                        if (k < 4) {        // one B fragment of the position after next
                            const int jn = (j + 2) % 9;
                            const int cn = cc + (j + 2 >= 9 ? 1 : 0);
                            if (j + 2 < 9 || cn < a.CC) {
                                const unsigned so_ = (unsigned)cn * u_chunk + (unsigned)(9 * wave + jn) * u_pos + (unsigned)(k >> 1) * 1024u + (unsigned)(k & 1) * u_piece;
                                fb[jn % 3][k >> 1][k & 1] = buf_load16(a.u8, a.u_bytes, u_voff, so_);
                            }
                        } else if (k >= 4 && j + 1 < 9) {  // one A fragment of the next position
                            fa[(j + 1) & 1][k - 4] = lds_u4(sV + ((9 * wave + j + 1) * NP + (k - 4)) * VPIECE + fragA);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the asm MFMAs' results are read by VALU below: past the XDL write -> VALU read distance

        // ---- epilogue: Y = A^T M A.  Stage 1 (registers): each wave reduces its positions along x — a full row of six or a half row;
        // the eight row parts meet through LDS (one cout group per pass: [8 parts][4 dx][32 tiles][32 couts] = 128 KB over V and the
        // patch buffers); stage 2: thread = (tile, 4 couts) finishes A^T . A.  (Measured and rejected: four 64 KB passes that leave the
        // patch buffers free for the next item's first patches during the epilogue — the extra barriers cost more than the hidden
        // HBM latency saved, profiles/r02_winograd8_variants.txt.) ----
        const int en = n, ey0 = y0, ex0 = x0, en0 = n0;
        const float einv = inv;
        const unsigned next = item + gridDim.x;
        const bool more = next < (unsigned)a.blocks;
        float* sQ = reinterpret_cast<float*>(smem);
        const int e_cq = tid & 7, e_tile = tid >> 3;
        const int oy = ey0 + 4 * (e_tile >> 3), ox = ex0 + 4 * (e_tile & 7);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            W8_BARRIER();                               // V (g = 0) / the previous pass's exchange buffer is no longer read
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tl = (r & 3) + 8 * (r >> 2) + 4 * hi;
                float* q = sQ + ((2 * wave) * 4 * 32 + tl) * 32 + (lane & 31);          // [part][dx][tile][cout]: + part * 4096 + dx * 1024
                if ((wave & 1) == 0) {                  // local 0..5 = a full row, 6..8 = columns 0..2 of the next row
                    const float m0 = acc[0][g][r], m1 = acc[1][g][r], m2 = acc[2][g][r], m3 = acc[3][g][r], m4 = acc[4][g][r], m5 = acc[5][g][r];
                    const float sa = m1 + m2, da = m1 - m2, sb = m3 + m4, db = m3 - m4;
                    q[0 * 1024] = (m0 + sa) + sb;
                    q[1 * 1024] = __builtin_fmaf(PA, da, PB * db);
                    q[2 * 1024] = __builtin_fmaf(A2, sa, B2 * sb);
                    q[3 * 1024] = __builtin_fmaf(A3, da, __builtin_fmaf(B3, db, m5));
                    const float n0_ = acc[6][g][r], n1 = acc[7][g][r], n2 = acc[8][g][r];
                    const float sa2 = n1 + n2, da2 = n1 - n2;
                    q[4 * 1024] = n0_ + sa2;
                    q[5 * 1024] = PA * da2;
                    q[6 * 1024] = A2 * sa2;
                    q[7 * 1024] = A3 * da2;
                } else {                                // local 0..2 = columns 3..5 of a row, 3..8 = the next full row
                    const float n3 = acc[0][g][r], n4 = acc[1][g][r], n5 = acc[2][g][r];
                    const float sb2 = n3 + n4, db2 = n3 - n4;
                    q[0 * 1024] = sb2;
                    q[1 * 1024] = PB * db2;
                    q[2 * 1024] = B2 * sb2;
                    q[3 * 1024] = __builtin_fmaf(B3, db2, n5);
                    const float m0 = acc[3][g][r], m1 = acc[4][g][r], m2 = acc[5][g][r], m3 = acc[6][g][r], m4 = acc[7][g][r], m5 = acc[8][g][r];
                    const float sa = m1 + m2, da = m1 - m2, sb = m3 + m4, db = m3 - m4;
                    q[4 * 1024] = (m0 + sa) + sb;
                    q[5 * 1024] = __builtin_fmaf(PA, da, PB * db);
                    q[6 * 1024] = __builtin_fmaf(A2, sa, B2 * sb);
                    q[7 * 1024] = __builtin_fmaf(A3, da, __builtin_fmaf(B3, db, m5));
                }
            }
            // this thread's output columns, bias and addresses of the pass (requested before the barrier)
            const int col = en0 + g * 32 + e_cq * 4;
            const bool col_ok = col < a.Cout;
            f32x4 bv;
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = col_ok ? a.bias[col + e] : 0.f;
            const unsigned pix = (unsigned)((en * a.H + oy) * a.W + ox);
            const unsigned y_voff = (pix * (unsigned)a.ldy + (unsigned)col) * 4u;
            const unsigned r_voff = (pix * (unsigned)a.ldr + (unsigned)col) * 4u;
            W8_BARRIER();
            // stage 2: rows 1 and 4 arrive in two parts; Y[dy][dx] = sum_i A^T[dy][i] q[i][dx]
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) {
                f32x4 p[8];
#pragma unroll
                for (int pt = 0; pt < 8; ++pt) p[pt] = *reinterpret_cast<const f32x4*>(sQ + ((pt * 4 + dx) * 32 + e_tile) * 32 + e_cq * 4);
                f32x4 yv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float q0 = p[0][e], q1 = p[1][e] + p[2][e], q2 = p[3][e], q3 = p[4][e], q4 = p[5][e] + p[6][e], q5 = p[7][e];
                    const float sa = q1 + q2, da = q1 - q2, sb = q3 + q4, db = q3 - q4;
                    yv[0][e] = (q0 + sa) + sb;
                    yv[1][e] = __builtin_fmaf(PA, da, PB * db);
                    yv[2][e] = __builtin_fmaf(A2, sa, B2 * sb);
                    yv[3][e] = __builtin_fmaf(A3, da, __builtin_fmaf(B3, db, q5));
                }
#pragma unroll
                for (int dy = 0; dy < 4; ++dy) {
                    const bool ok = col_ok && oy + dy < a.H && ox + dx < a.W;
                    const unsigned so = (unsigned)(dy * a.W + dx);
                    f32x4 rv = {0.f, 0.f, 0.f, 0.f};
                    if (a.res) rv = __builtin_bit_cast(f32x4, buf_load16(a.res, a.r_bytes, ok ? r_voff : OOB, so * (unsigned)a.ldr * 4u));
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[e] = fmaxf(yv[dy][e] * einv + bv[e] + rv[e], lo);
                        omax = fmaxf(omax, ok ? fabsf(o[e]) : 0.f);
                    }
                    buf_store16(o, a.y, a.y_bytes, ok ? y_voff : OOB, so * (unsigned)a.ldy * 4u);
                }
            }
        }
        if (a.ymax) {          // max |y| of this item into its image's slot: one atomic per wave and item
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) omax = fmaxf(omax, __shfl_xor(omax, o, 64));
            if (lane == 0 && omax > 0.f) atomicMax(a.ymax + en, __float_as_uint(omax));
            omax = 0.f;
        }
        if (!more) break;
        item = next;
    }
#undef W8_SETUP
#undef W8_ISSUE_P
#undef W8_LOAD_B
#undef W8_BARRIER
}

// U = G g G^T in float64; mode 0: fold max |U| into scal[2] (bit pattern, atomic max); mode 1: scale by S_u = 2^(13 - e), split into two
// fp16 pieces, store [ci/16][36][piece][CoutP][16 ci]; scal[1] receives S_u.
__global__ __launch_bounds__(256) void weights8_kernel(const float* __restrict__ w, unsigned short* __restrict__ u8, float* __restrict__ scal,
                                                       int Cin, int Cout, int CoutP, int mode) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const bool live = t < (long)CoutP * Cin;
    const int ci = live ? (int)(t % Cin) : 0, co = live ? (int)(t / Cin) : 0;
    double g[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) g[i][j] = (live && co < Cout) ? (double)w[((long)co * 9 + i * 3 + j) * Cin + ci] : 0.0;
    const double a_ = PA, b_ = PB, a2 = a_ * a_, b2 = b_ * b_;
    const double n0 = a2 * b2, n1 = 2.0 * a2 * (a2 - b2), n3 = 2.0 * b2 * (b2 - a2);
    const double G[6][3] = {{1.0 / n0, 0.0, 0.0},          {1.0 / n1, a_ / n1, a2 / n1}, {1.0 / n1, -a_ / n1, a2 / n1},
                            {1.0 / n3, b_ / n3, b2 / n3},  {1.0 / n3, -b_ / n3, b2 / n3}, {0.0, 0.0, 1.0}};
    double h[6][3];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) h[i][j] = G[i][0] * g[0][j] + G[i][1] * g[1][j] + G[i][2] * g[2][j];
    if (mode == 0) {
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) m = fmaxf(m, fabsf((float)(h[i][0] * G[j][0] + h[i][1] * G[j][1] + h[i][2] * G[j][2])));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if ((threadIdx.x & 63) == 0 && m > 0.f && m < __builtin_inff()) atomicMax(reinterpret_cast<unsigned*>(scal + 2), __float_as_uint(m));
        return;
    }
    float Su = 1.f;
    {
        const float mx = scal[2];
        if (mx > 0.f && mx < __builtin_inff()) {
            int e_;
            (void)__builtin_frexpf(mx, &e_);
            e_ = 13 - e_;
            Su = __builtin_ldexpf(1.f, e_ < -100 ? -100 : (e_ > 100 ? 100 : e_));
        }
    }
    if (t == 0) scal[1] = Su;
    if (!live) return;
    const int cc = ci >> 4, c16 = ci & 15;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const double x = (h[i][0] * G[j][0] + h[i][1] * G[j][1] + h[i][2] * G[j][2]) * (double)Su;
            const _Float16 hf = (_Float16)(float)x;
            const _Float16 lf = (_Float16)(float)(x - (double)(float)hf);
            const long base = ((((long)cc * NPOS + (i * 6 + j)) * NP) * CoutP + co) * 16 + c16;
            u8[base] = __builtin_bit_cast(unsigned short, hf);
            u8[base + (long)CoutP * 16] = __builtin_bit_cast(unsigned short, lf);
        }
}

}  // namespace cnl_wino8

size_t cnl_wino8_weight_bytes(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0 || Cin % 16) return 0;
    const size_t CoutP = (size_t)((Cout + 63) / 64) * 64;
    return (size_t)(Cin / 16) * cnl_wino8::NPOS * cnl_wino8::NP * CoutP * 32;
}
size_t cnl_wino8_scalar_floats() { return 16; }   // [1] S_u, [2] max |U|

int cnl_wino8_transform_weights(const float* w_ohwi, void* u8, float* scal, int Cin, int Cout, void* stream) {
    using namespace cnl_wino8;
    const int CoutP = (Cout + 63) / 64 * 64;
    CNL_HIP(hipMemsetAsync(scal, 0, 16 * sizeof(float), (hipStream_t)stream));
    const long total = (long)CoutP * Cin;
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(weights8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_ohwi, (unsigned short*)u8, scal, Cin, Cout, CoutP, 0);
    hipLaunchKernelGGL(weights8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_ohwi, (unsigned short*)u8, scal, Cin, Cout, CoutP, 1);
    return cnl::check_launch("weights8_kernel");
}

// does the F(4x4) kernel cover this layer (alignment / shape; the dispatcher adds its own profitability rule)
bool cnl_wino8_eligible(const cnl_conv_params* p) {
    return p->Cin % 16 == 0 && p->Cout % 4 == 0 && p->ldy % 4 == 0 && ((uintptr_t)p->y & 15) == 0 &&
           (!p->residual || (p->ldr % 4 == 0 && ((uintptr_t)p->residual & 15) == 0));
}

// Launch (arguments already validated by cnl_conv3x3_winograd_f32); xmax = per-image max |x| (hint or the caller's own pass).
int cnl_wino8_launch(const cnl_conv_params* p, const void* u8, const float* scal, const float* xmax, void* stream) {
    using namespace cnl_wino8;
    Args a;
    a.x = p->x; a.u8 = u8; a.xmax = xmax; a.su = scal + 1; a.ymax = reinterpret_cast<unsigned*>(p->y_absmax);
    a.bias = p->bias; a.res = p->residual; a.y = p->y;
    const int upf = (p->flags & CNL_UPSAMPLE_IN) ? 2 : 1;
    a.N = p->N; a.Hs = p->H_in; a.Ws = p->W_in; a.H = p->H_in * upf; a.W = p->W_in * upf; a.Cin = p->Cin; a.Cout = p->Cout;
    a.CoutP = (p->Cout + 63) / 64 * 64;
    a.ldx = p->ldx; a.ldy = p->ldy; a.ldr = p->ldr;
    a.CC = p->Cin / 16;
    a.nb = a.CoutP / BN; a.bx = (a.W + 4 * TX - 1) / (4 * TX); a.by = (a.H + 4 * TY - 1) / (4 * TY);
    const long long blocks = (long long)p->N * a.by * a.bx * a.nb;
    CNL_REQUIRE(blocks < (1ll << 31), CNL_E_UNSUPPORTED, "cnl_conv3x3_winograd_f32: grid too large");
    a.blocks = (int)blocks;
    const unsigned long long xb = (((unsigned long long)p->N * p->H_in * p->W_in - 1) * p->ldx + p->Cin) * 4ull;
    const unsigned long long ub = (unsigned long long)cnl_wino8_weight_bytes(p->Cin, p->Cout);
    const unsigned long long Mo = (unsigned long long)p->N * a.H * a.W;
    const unsigned long long yb = ((Mo - 1) * p->ldy + p->Cout) * 4ull;
    const unsigned long long rb = p->residual ? ((Mo - 1) * p->ldr + p->Cout) * 4ull : 0ull;
    const unsigned long long slack = (unsigned long long)(3 * a.W + 4) * 4ull;     // scalar-offset reach of the epilogue stores
    CNL_REQUIRE(xb < 0xFFFFFF00ull && ub < 0xFFFFFF00ull && yb + slack * p->ldy < 0xFFFFFF00ull && rb + slack * (p->residual ? p->ldr : 0) < 0xFFFFFF00ull,
                CNL_E_UNSUPPORTED, "cnl_conv3x3_winograd_f32: tensor spans >= 4 GiB; split the batch");
    a.x_bytes = (unsigned)xb; a.u_bytes = (unsigned)ub; a.y_bytes = (unsigned)yb; a.r_bytes = (unsigned)rb;
    a.flags = p->flags;
    static cnl::DeviceOnce once;
    int n_cu = 0;
    const int rc = cnl::kernel_setup(once, reinterpret_cast<const void*>(&winograd8_kernel), LDS_BYTES, &n_cu);
    if (rc != CNL_OK) return rc;
    const unsigned grid = (unsigned)(blocks < (long long)n_cu ? blocks : (long long)n_cu);
    hipLaunchKernelGGL(winograd8_kernel, dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, a);
    return cnl::check_launch("winograd8_kernel");
}
