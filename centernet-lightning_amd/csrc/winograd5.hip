// winograd5.hip — Winograd F(2x2,3x3) on the FP16 matrix cores at fp32 accuracy: two pieces per operand, three MFMAs per product.
//
// winograd3.hip showed that the matrix core fed with exact bf16 pieces reproduces fp32 products, and that the kernel around it is
// bound by power and instruction issue, i.e. by the amount of work per output.  fp16 pieces carry 11 significand bits instead
// of 8: with a power-of-two scale S per tensor (max |x S| in [2^13, 2^14), exact), x S = hi + lo with hi = RN16(x S),
// lo = RZ16(x S - hi) represents x to 2^-22 relative (down to 2^-17 of the tensor's maximum; below that the ABSOLUTE error stays
// <= 2^-25 / S, i.e. 2^-38 of the maximum), and the three terms  hi lo' + lo hi' + hi hi'  leave out lo lo' <= 2^-22 |x y|.
// Measured on MI355X (tools/bf16x3_probe.hip, K = 2304): max |err| 7.7e-6 / rms 9.2e-7 against 1.27e-5 / 1.36e-6 for the fp32
// MFMA and 8.0e-6 / 1.33e-6 for the six-term bf16 split; fp16 subnormals run at full MFMA rate and are not flushed.
// Against winograd3.hip: 3 instead of 6 MFMAs per product, 4 instead of 6 bytes per operand element through LDS / from L2, and
// 2.5 instead of 5.5 VALU operations per V element for the split — the mixed-precision fma does the scaling, the rounding and
// the exact residual:  hi = v_fma_mixlo/hi_f16(v, S, 0),  r = v_fma_mix_f32(v, S, -hi),  lo = v_cvt_pkrtz_f16_f32(r0, r1).
// The scale of the activations comes from the tensor's maximum magnitude (absmax_kernel, one pass over the input before the
// launch, result left in the layer's weight buffer); the weights carry their own scale, fixed when they are transformed.
// Structure (work item, wave roles, wave-private single-buffered V, patch DMA, one barrier per chunk): winograd3.hip.
#include "cnl_common.h"

#pragma clang fp contract(off)

#ifndef W5_NT_X
#define W5_NT_X 0     /* cache policy (aux) of the patch DMA loads: 2 = nt */
#endif
#ifndef W5_NT_Y
#define W5_NT_Y 2     /* cache policy (aux) of the output stores: nt — a layer's output is far larger than the L2 and would only evict the input patches and weights that ARE re-read (fused first head blocks -9 %) */
#endif
namespace cnl_wino5 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

struct Args {
    const float* x;
    const void* u3;                   // pre-split, pre-scaled weights (fp16 pieces)
    const float* xmax;                // max |x| of this launch's input (absmax_kernel, or handed over by the producer)
    const float* su;                  // scale of the weights
    unsigned* ymax;                   // optional: max |y| of this launch's output, for the consumer (atomic max on the bits)
    const float* bias;
    const float* res;
    float* y;
    int N, H, W, Cin, Cout, CoutP;   // H, W: output (= logical input) size
    int Hs, Ws;                       // stored input size (H/2, W/2 with CNL_UPSAMPLE_IN, else H, W)
    int ldx, ldy, ldr;
    int CC;                           // Cin / 16
    int nb, bx, by;                   // blocks along cout, x (16 px), y (16 px)
    int blocks;
    unsigned x_bytes, u_bytes, y_bytes, r_bytes;
    unsigned flags;
    int order;                        // work-item order (see W5_SETUP)
};

constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr int T = 64;                       // tiles per workgroup: 8 x 8
constexpr int BN = 64;
constexpr int PH = 18, PW = 18;             // patch height / width in pixels
constexpr int PWP = 19;                     // padded patch row of the LDS image [py][quad][PWP][4 floats]
constexpr int IT_STRIDE = 4 * 4 * PWP * 16; // patch bytes between transform items (two tile rows = four patch rows)
constexpr int VPIECE = T * 32;              // 2048: one (position, piece) plane of a wave's V: [64 tiles][16 ci bf16]
constexpr int NP = 2;                       // pieces per operand
constexpr int VW_BYTES = 4 * NP * VPIECE;   // 16384 per wave
constexpr int V_BYTES = 4 * VW_BYTES;       // 65536 (= one epilogue pass)
constexpr int P_SLOTS = 1408;               // 1368 used; 5 x 256 (all waves) + 128 (waves 0-1)
constexpr int P_BYTES = P_SLOTS * 16;       // 22528 per buffer (two buffers)
constexpr int LDS_BYTES = V_BYTES + 2 * P_BYTES;                 // 110592: one workgroup per CU (the accumulators allow no more)

__device__ __forceinline__ void dma16(const float* base, unsigned bytes, char* lds_dst, unsigned voffset, unsigned soffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_dst, 16, voffset, soffset, 0, W5_NT_X);
}
__device__ __forceinline__ u32x4 buf_load16(const void* base, unsigned bytes, unsigned voffset, unsigned soffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    return (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, soffset, 0);
}
__device__ __forceinline__ float buf_load(const float* base, unsigned bytes, unsigned voffset, unsigned soffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voffset, soffset, 0));
}
__device__ __forceinline__ void buf_store(float v, float* base, unsigned bytes, unsigned voffset, unsigned soffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, voffset, soffset, W5_NT_Y);
}
__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// the split of a channel pair (v0, v1), scaled by the power of two S:  hi = RN16(v S) packed, r = v S - hi exactly
__device__ __forceinline__ unsigned split_hi_lo(float v0, float S) {            // RN16(v0 S) in the low half
    unsigned pk;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(pk) : "v"(v0), "v"(S));
    return pk;
}
__device__ __forceinline__ unsigned split_hi_hi(unsigned pk, float v1, float S) {   // ... and RN16(v1 S) in the high half
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
__device__ __forceinline__ f32x16 mfma_zero() {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const u32x4 zz = {0u, 0u, 0u, 0u};
    return mfma16(zz, zz, z);
}
__device__ __forceinline__ f32x4 lds_f4(const char* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ u32x4 lds_u4(const char* p) { return *reinterpret_cast<const u32x4*>(p); }

// Registers of one wave's input-transform pipeline.  A "pass-item" = (item: tile, 4 channels) x (pass P: position pair {2P, 2P+1}
// of the wave's row).  Its VALU operations are indexed 0..39 so that the main loop can place them per MFMA slice:
//   0..11   t[c] = da[c] + sg * db[c]      the wave's row of B^T d; pass 0: columns 0, 1, 2 (1, 2 are kept), pass 1: column 3 only
//   12..19  v[0], v[1]                     the two positions of the pair: (t0 - t2, t1 + t2) or (t2 - t1, t1 - t3)
//   20..39  both v together: 4 x mixlo, 4 x mixhi (packed hi pairs), 8 residuals, 4 x pkrtz (packed lo pairs)
struct Xf {
    f32x4 da[2][3], db[2][3];       // [register set][column]: rows ra / rb of the patch (read one pass-item ahead)
    f32x4 t[3], v[2];
    f32x4 th[4][2];                 // t of patch columns 1, 2 of each item, kept from pass 0 for pass 1
    float r[2][4];
    unsigned pk[2][NP][2];          // [position of the pair][piece][channel pair]
};
constexpr int XOPS = 40;
__device__ __forceinline__ void xop(Xf& s, const int set, const int P, const int op, const float sg, const float S, const int it) {
    if (op < 12) {
        const int c = op >> 2, e = op & 3;
        if (P == 0) {
            const float t_ = __builtin_fmaf(s.db[set][c][e], sg, s.da[set][c][e]);
            if (c == 0) s.t[0][e] = t_; else s.th[it][c - 1][e] = t_;
        } else if (c == 2) s.t[2][e] = __builtin_fmaf(s.db[set][2][e], sg, s.da[set][2][e]);
    } else if (op < 20) {
        const int vi = (op - 12) >> 2, e = op & 3;
        if (P == 0) s.v[vi][e] = vi == 0 ? s.t[0][e] - s.th[it][1][e] : s.th[it][0][e] + s.th[it][1][e];
        else s.v[vi][e] = vi == 0 ? s.th[it][1][e] - s.th[it][0][e] : s.th[it][0][e] - s.t[2][e];
    } else if (op < XOPS) {
        // the two positions and their channel pairs advance side by side: dependent mixed-precision operations (partial register
        // writes, half-register reads) stay >= 3 instructions apart, which saves the hazard nops
        const int w = op - 20;
        if (w < 4) s.pk[w >> 1][0][w & 1] = split_hi_lo(s.v[w >> 1][2 * (w & 1)], S);
        else if (w < 8) s.pk[(w - 4) >> 1][0][w & 1] = split_hi_hi(s.pk[(w - 4) >> 1][0][w & 1], s.v[(w - 4) >> 1][2 * (w & 1) + 1], S);
        else if (w < 16) {
            const int vi = (w - 8) >> 2, e = w & 3;
            s.r[vi][e] = (e & 1) ? split_res_hi(s.v[vi][e], S, s.pk[vi][0][e >> 1]) : split_res_lo(s.v[vi][e], S, s.pk[vi][0][e >> 1]);
        } else {
            const int vi = (w - 16) >> 1, pp = w & 1;
            s.pk[vi][1][pp] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(s.r[vi][2 * pp], s.r[vi][2 * pp + 1]));
        }
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void winograd5_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sV = smem;                                  // [4 waves][4 positions][3 pieces][64 tiles][16 ci] bf16
    char* sP = smem + V_BYTES;                        // [2][18 py][4 quads][19 px][4 ci] fp32 (+ slack)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = transform row i owned by this wave
    const int hi = lane >> 5;
    const int xi0 = wave * 4;
    const bool up = a.flags & CNL_UPSAMPLE_IN;
    const unsigned u_piece = (unsigned)(a.CoutP * 32);              // bytes per (chunk, position, piece) plane of U
    const unsigned u_pos = (unsigned)NP * u_piece;
    const unsigned u_chunk = 16u * u_pos;

    // t[i][*] = d[ra][*] + sg * d[rb][*]:  i = 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 3 ? 3 : (wave == 2 ? 1 : 2);
    const float sg = wave == 1 ? 1.f : -1.f;
    // transform items: lane -> (tile column tx, channel quad q, tile row parity tyl); item it = 0..3 -> tile row 2 it + tyl
    // The quad index is chosen so that each 16-lane group of a ds_read_b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) holds
    // all eight tile columns for two quads whose planes lie an odd number of 16-byte slots apart (PWP = 19): with px = 2 tx the
    // group then covers all 16 slot residues — no bank conflicts on the patch reads.
    const int t_tx = lane & 7, t_tyl = lane >> 5;
    const int t_q = ((((lane >> 2) ^ (lane >> 3) ^ (lane >> 4)) & 1) << 1) | ((lane >> 3) & 1);
    const int src_a = (((2 * t_tyl + ra) * 4 + t_q) * PWP + 2 * t_tx) * 16;      // + it * IT_STRIDE + column * 16
    const int src_b = (((2 * t_tyl + rb) * 4 + t_q) * PWP + 2 * t_tx) * 16;
    // V rows are 32 bytes = two 16-byte halves (ci 0-7 | ci 8-15); rows of odd tile rows store them swapped, which makes the
    // ds_read_b128 fragment reads (16-lane groups, 32-byte row pitch) bank-conflict-free
    const int dstv = wave * VW_BYTES + (t_tyl * 8 + t_tx) * 32 + (((t_q >> 1) ^ t_tyl) << 4) + (t_q & 1) * 8;   // + it*512 + (j*NP+k)*VPIECE
    const int fragA = wave * VW_BYTES + (lane & 31) * 32 + ((hi ^ ((lane >> 3) & 1)) << 4);                    // + (j*NP+k)*VPIECE + tg*1024
    const float lo = (a.flags & CNL_RELU) ? 0.f : -__builtin_inff();
    // power-of-two scale of V, PER IMAGE (an image's result never depends on its batch neighbours — batch invariance, shard == full
    // batch): |V| <= 4 max |x| (sums of four inputs), 4 max |x| S in [2^13, 2^14) — exact, undone in the epilogue together with the
    // scale of the weights.  S / inv_n belong to the item set up last (W5_SETUP), `inv` to the one in the epilogue.
    const float Su = a.su[0];
    float S = 1.f, inv_n = 1.f / Su;
    float omax = 0.f;          // running max |y| of this lane's stores of the current item

    int n, y0, x0, n0;
    unsigned p_off[6], u_voff;
    float bias_n[2];           // bias of the item set up last (the next one, from the epilogue's prefetch on)
#define W5_SETUP(item_)                                                                                          \
    do {                                                                                                         \
        unsigned b_ = cnl::xcd_remap((item_), (unsigned)a.blocks);                                               \
        int nbi_, bxi_, byi_;                                                                                    \
        if (a.order == 0) {          /* cout block fastest: the workgroups sharing an input patch run side by side */ \
            nbi_ = b_ % a.nb; b_ /= a.nb; bxi_ = b_ % a.bx; b_ /= a.bx; byi_ = b_ % a.by; n = b_ / a.by;         \
        } else if (a.order == 1) {   /* cout block slowest inside an image: the CUs of an XCD share one slice of U */ \
            bxi_ = b_ % a.bx; b_ /= a.bx; byi_ = b_ % a.by; b_ /= a.by; nbi_ = b_ % a.nb; n = b_ / a.nb;         \
        } else {                     /* pairs of cout blocks fastest, then the tile, then the pair index */       \
            const int np_ = (a.nb + 1) / 2;                                                                      \
            const int lo_ = b_ % 2; b_ /= 2; bxi_ = b_ % a.bx; b_ /= a.bx; byi_ = b_ % a.by; b_ /= a.by;         \
            const int pr_ = b_ % np_; n = b_ / np_; nbi_ = pr_ * 2 + lo_;                                        \
        }                                                                                                        \
        y0 = byi_ * 16; x0 = bxi_ * 16; n0 = nbi_ * BN;                                                          \
        _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                                          \
            const int s_ = i * 256 + tid;              /* 16-byte slot of the patch image: (py*4 + quad)*PWP + px */ \
            const int rowq_ = s_ / PWP, pxx_ = s_ - rowq_ * PWP;                                                 \
            const int py_ = rowq_ >> 2, q_ = rowq_ & 3;                                                          \
            const int iy_ = y0 - 1 + py_, ix_ = x0 - 1 + pxx_;                                                   \
            const bool ok_ = py_ < PH && pxx_ < PW && (unsigned)iy_ < (unsigned)a.H && (unsigned)ix_ < (unsigned)a.W; \
            const int sy_ = up ? (iy_ >> 1) : iy_, sx_ = up ? (ix_ >> 1) : ix_;   /* nearest-2x upsample folded in */ \
            p_off[i] = ok_ ? (unsigned)((((n * a.Hs + sy_) * a.Ws + sx_) * a.ldx + q_ * 4) * 4) : OOB;           \
        }                                                                                                        \
        /* this lane's B fragments: cout row n0 + (lane & 31) (+ 32 for the second group), channel half hi */    \
        u_voff = (unsigned)((n0 + (lane & 31)) * 32 + hi * 16);                                                  \
        {                                                                                                        \
            const float mx4_ = 4.f * a.xmax[n * AMS];                                                                  \
            S = 1.f;                                                                                             \
            if (mx4_ > 0.f && mx4_ < __builtin_inff()) {                                                         \
                int e_;                                                                                          \
                (void)__builtin_frexpf(mx4_, &e_);            /* 2^(e-1) <= mx4 < 2^e */                         \
                e_ = 14 - e_;                                                                                    \
                S = __builtin_ldexpf(1.f, e_ < -100 ? -100 : (e_ > 100 ? 100 : e_));                             \
            }                                                                                                    \
            inv_n = 1.f / (S * Su);                                                                              \
        }                                                                                                        \
        /* bias of this thread's two epilogue columns: requested now, used after the chunk loop */               \
        _Pragma("unroll") for (int g_ = 0; g_ < 2; ++g_) {                                                       \
            const int col_ = n0 + g_ * 32 + (tid & 31);                                                          \
            bias_n[g_] = col_ < a.Cout ? a.bias[col_] : 0.f;                                                     \
        }                                                                                                        \
    } while (0)
    // the channel-chunk offset rides in the SCALAR offset (the bounds check looks at the vector offset alone, so halo lanes still
    // read zeros); a chunk past the end is not fetched
#define W5_ISSUE_P(cc_)                                                                                          \
    do {                                                                                                         \
        if ((cc_) < a.CC) {                                                                                      \
            char* d_ = sP + ((cc_) & 1) * P_BYTES;                                                               \
            _Pragma("unroll") for (int i = 0; i < 5; ++i)                                                        \
                dma16(a.x, a.x_bytes, d_ + (i * 256 + wave * 64) * 16, p_off[i], (unsigned)((cc_) * 64));        \
            if (wave < 2) dma16(a.x, a.x_bytes, d_ + (1280 + wave * 64) * 16, p_off[5], (unsigned)((cc_) * 64)); \
        }                                                                                                        \
    } while (0)
    // B fragments of position xi0 + j_ of chunk cc_, cout group g_ (three pieces): global -> registers
#define W5_LOAD_B(cc_, j_, buf_, g_)                                                                             \
    do {                                                                                                         \
        if ((cc_) < a.CC) {                                                                                      \
            const unsigned so_ = (unsigned)(cc_) * u_chunk + (unsigned)(xi0 + (j_)) * u_pos + (unsigned)(g_) * 1024u; \
            _Pragma("unroll") for (int kk_ = 0; kk_ < NP; ++kk_)                                                 \
                fb[buf_][g_][kk_] = buf_load16(a.u3, a.u_bytes, u_voff, so_ + (unsigned)kk_ * u_piece);          \
        }                                                                                                        \
    } while (0)
#define W5_READ_A(j_, buf_, g_)                                                                                  \
    _Pragma("unroll") for (int kk_ = 0; kk_ < NP; ++kk_) fa[buf_][g_][kk_] = lds_u4(sV + fragA + ((j_) * NP + kk_) * VPIECE + (g_) * 1024)
    // MFMA s_ (0..11) of a position: cout group s_ / 6 — group 0 first, so that its B registers are free (and refilled for the
    // position after next) from mid-slot on —, then term (s_ % 6) >> 1 in the order (hi lo', lo hi', hi hi'), tile group s_ & 1
#define W5_MFMA(j_, buf_, s_)                                                                                    \
    do {                                                                                                         \
        const int cg_ = (s_) / 6, term_ = ((s_) % 6) >> 1, tg_ = (s_) & 1;                                       \
        const int ka_ = term_ == 1 ? 1 : 0, kb_ = term_ == 0 ? 1 : 0;                                            \
        acc[j_][tg_][cg_] = mfma16(fa[buf_][tg_][ka_], fb[buf_][cg_][kb_], acc[j_][tg_][cg_]);                   \
    } while (0)
    // patch reads of pass-item (P_, it_) into register set set_; pa_ / pb_ = patch buffer + src_a / src_b
#define W5_X_READ1(set_, pa_, pb_, P_, it_, c_, row_)                                                            \
    do {                                                                                                         \
        if (1 && (P_) == 1 && (c_) != 2) break;                                                          \
        if ((row_) == 0) xf.da[set_][c_] = lds_f4((pa_) + (it_) * IT_STRIDE + ((P_) + (c_)) * 16);               \
        else xf.db[set_][c_] = lds_f4((pb_) + (it_) * IT_STRIDE + ((P_) + (c_)) * 16);                           \
    } while (0)
#define W5_X_READ(set_, pa_, pb_, P_, it_)                                                                       \
    _Pragma("unroll") for (int c_ = 0; c_ < 3; ++c_) {                                                           \
        W5_X_READ1(set_, pa_, pb_, P_, it_, c_, 0);                                                              \
        W5_X_READ1(set_, pa_, pb_, P_, it_, c_, 1);                                                              \
    }
#define W5_X_WRITE(P_, it_)                                                                                      \
    _Pragma("unroll") for (int jj_ = 0; jj_ < 2; ++jj_)                                                          \
        _Pragma("unroll") for (int kk_ = 0; kk_ < NP; ++kk_)                                                     \
            *reinterpret_cast<u32x2*>(sV + dstv + (it_) * 512 + ((2 * (P_) + jj_) * NP + kk_) * VPIECE) =        \
                u32x2{xf.pk[jj_][kk_][0], xf.pk[jj_][kk_][1]};
    // workgroup barrier WITHOUT the vmcnt(0) that __syncthreads() adds when LDS-DMA is in flight (own LDS accesses drained)
#define W5_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    // One position slot = 12 MFMAs in 12 slices fenced by sched_barrier(0).
    //   j_ / buf_    position multiplied in this slot and its fragment buffer
    //   (cA_, jA_)   the next position: its A fragments and the B fragments of its cout group 1 are fetched into buf_ ^ 1 (PA_)
    //   (cB_, jB_)   the position after next: B fragments of its cout group 0 go into buf_ from slice 6 on (PB_)
    //   JOBS_        transform the two pass-items (P_, it0_) [slices 0-5, register set 0] and (P_, it0_ + 1) [6-11, set 1]
    //   RDN_         in slices 6-11 read the patch of the NEXT slot's first pass-item (nP_, nIt_) from (npa_, npb_) into set 0
    //   MID_         (slot of position 0) before slice 6: this wave's DMAs of the next patch landed, barrier; slice 7: DMA
    //                of the patch after that (chunk dC_)
#define W5_SLOT(j_, buf_, cA_, jA_, PA_, cB_, jB_, PB_, JOBS_, P_, it0_, RDN_, nP_, nIt_, npa_, npb_, MID_, dC_) \
    do {                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        _Pragma("unroll") for (int k = 0; k < 12; ++k) {                                                         \
            const int pi = k / 6, ks = k % 6;                                                                    \
            if ((MID_) && k == 6) {                                                                              \
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   /* all but the newest 4 B loads: the patch DMA is older */ \
                W5_BARRIER();                                                                   \
                __builtin_amdgcn_sched_barrier(0);                                                               \
            }                                                                                                    \
            W5_MFMA(j_, buf_, k);                                                                                \
            if (PA_) {                                                                                           \
                if (k == 0) W5_LOAD_B(cA_, jA_, (buf_) ^ 1, 1);                                                  \
                if (k >= 2 && k < 6) {                                                            \
                    const int g_ = (k - 2) >> 1, kk_ = (k - 2) & 1;                                              \
                    fa[(buf_) ^ 1][g_][kk_] = lds_u4(sV + fragA + ((jA_) * NP + kk_) * VPIECE + g_ * 1024);      \
                }                                                                                                \
            }                                                                                                    \
            if ((PB_) && k == 6) W5_LOAD_B(cB_, jB_, buf_, 0);                                                   \
            if (JOBS_) {                                                                                         \
                if (k < 6) W5_X_READ1(1, pa, pb, P_, (it0_) + 1, k >> 1, k & 1);                                 \
                _Pragma("unroll") for (int o_ = 0; o_ < 8; ++o_) xop(xf, pi, P_, ks * 8 + o_, sg, S, (it0_) + pi); \
                if (ks == 5) { if (pi == 0) { W5_X_WRITE(P_, it0_); } else { W5_X_WRITE(P_, (it0_) + 1); } }     \
            }                                                                                                    \
            if ((RDN_) && k >= 6) W5_X_READ1(0, npa_, npb_, nP_, nIt_, (k - 6) >> 1, k & 1);                     \
            if ((MID_) && k == 7) W5_ISSUE_P(dC_);                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
        }                                                                                                        \
    } while (0)

    unsigned item = blockIdx.x;
    W5_SETUP(item);
    W5_ISSUE_P(0);
    W5_ISSUE_P(1);
    u32x4 fa[2][2][NP];      // A fragments: [buffer][tile group][piece]
    u32x4 fb[2][2][NP];      // B fragments: [buffer][cout group][piece]
    W5_LOAD_B(0, 0, 0, 0);
    W5_LOAD_B(0, 0, 0, 1);
    W5_LOAD_B(0, 1, 1, 0);
    bool first = true;
    while (true) {
        f32x16 acc[4][2][2];     // [position j of row `wave`][tile group][cout group]
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[j][g >> 1][g & 1] = mfma_zero();
        Xf xf;

        // patches 0 / 1 landed (this wave's parts)?  Their DMAs are followed in this wave's VMEM queue by the 6 B loads of chunk 0
        // and the 32 stores of the previous item's second epilogue pass (its residual loads are older): a counted wait lets
        // those stay in flight
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(38)" ::: "memory");
        first = false;
        W5_BARRIER();                                         // ... and everybody's
        {   // input transform of chunk 0, all four positions (not overlapped with MFMAs): eight pass-items, each read one ahead
            const char* pa = sP + src_a;
            const char* pb = sP + src_b;
            W5_X_READ(0, pa, pb, 0, 0);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int it = g >> 1, P = g & 1, set = g & 1;
                if (g < 7) { W5_X_READ(set ^ 1, pa, pb, (g + 1) & 1, (g + 1) >> 1); }
#pragma unroll
                for (int o = 0; o < XOPS; ++o) xop(xf, set, P, o, sg, S, it);
                W5_X_WRITE(P, it);
            }
        }
        W5_READ_A(0, 0, 0);
        W5_READ_A(0, 0, 1);
        {   // position 0 of chunk 0: no transform work yet; patch 1 is read from slice 6 on, patch 2 requested
            const char* npa = sP + P_BYTES + src_a;
            const char* npb = sP + P_BYTES + src_b;
            const char* pa = npa; const char* pb = npb;       // (unused: JOBS_ = 0)
            W5_SLOT(0, 0, 0, 1, 1, 0, 2, 1, 0, 0, 0, 1, 0, 0, npa, npb, 1, 2);
            (void)pa; (void)pb;
        }
        // chunk n: positions 1..3 of chunk n-1, then position 0 of chunk n; beside them the transform of chunk n
        for (int cn = 1; cn < a.CC; ++cn) {
            const char* pa = sP + (cn & 1) * P_BYTES + src_a;
            const char* pb = sP + (cn & 1) * P_BYTES + src_b;
            const char* npa = sP + ((cn + 1) & 1) * P_BYTES + src_a;
            const char* npb = sP + ((cn + 1) & 1) * P_BYTES + src_b;
            W5_SLOT(1, 1, cn - 1, 2, 1, cn - 1, 3, 1, 1, 0, 0, 1, 0, 2, pa, pb, 0, 0);      // positions {0,1} of chunk cn, items 0-1
            W5_SLOT(2, 0, cn - 1, 3, 1, cn, 0, 1, 1, 0, 2, 1, 1, 0, pa, pb, 0, 0);          //                          items 2-3
            W5_SLOT(3, 1, cn, 0, 1, cn, 1, 1, 1, 1, 0, 1, 1, 2, pa, pb, 0, 0);              // positions {2,3} of chunk cn, items 0-1
            W5_SLOT(0, 0, cn, 1, 1, cn, 2, 1, 1, 1, 2, 1, 0, 0, npa, npb, 1, cn + 2);       //                          items 2-3
        }
        {   // positions 1..3 of the last chunk: MFMAs only
            const char* pa = sP; const char* pb = sP;
            const int cl = a.CC - 1;
            W5_SLOT(1, 1, cl, 2, 1, cl, 3, 1, 0, 0, 0, 0, 0, 0, pa, pb, 0, 0);
            W5_SLOT(2, 0, cl, 3, 1, cl, 0, 0, 0, 0, 0, 0, 0, 0, pa, pb, 0, 0);
            W5_SLOT(3, 1, cl, 0, 0, cl, 0, 0, 0, 0, 0, 0, 0, 0, pa, pb, 0, 0);
            (void)pa; (void)pb;
        }
        // ---- epilogue: Y = A^T M A.  Stage 1 (this wave's row of positions, in registers): q_c = sum_j A^T[c][j] M[i][j]; the
        // four rows meet through LDS, one tile group per pass ([4 i][2 c][2 cout groups][32 tiles][32 co] = 64 KB of the V region) ----
        float* sQ = reinterpret_cast<float*>(smem);
        const int co = tid & 31;
        const int en = n, ey0 = y0, ex0 = x0, en0 = n0;        // this item's coordinates (the setup below moves on to the next)
        const bool full = (y0 + 16 <= a.H) && (x0 + 16 <= a.W) && (n0 + BN <= a.Cout);
        const float bv[2] = {bias_n[0], bias_n[1]};
        const float inv = inv_n;
        const unsigned next = item + gridDim.x;
        const bool more = next < (unsigned)a.blocks;
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) {
            // this pass's output addresses and residual values (requested before stage 1, so their latency is covered)
            unsigned y_voff[2][4], r_voffs[2][4];
            bool ok[2][4][2][2];
            float rv[2][4][2][2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int col = en0 + g * 32 + co;
                const bool col_ok = col < a.Cout;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int tl = (tid >> 5) + 8 * it;            // tile inside the 4 x 8 tile group
                    const int oy = ey0 + 8 * tg + 2 * (tl >> 3), ox = ex0 + 2 * (tl & 7);
                    const unsigned pix = (unsigned)((en * a.H + oy) * a.W + ox);
                    y_voff[g][it] = (pix * (unsigned)a.ldy + (unsigned)col) * 4u;
                    const unsigned r_voff = (pix * (unsigned)a.ldr + (unsigned)col) * 4u;
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx) {
                            ok[g][it][dy][dx] = full || (col_ok && oy + dy < a.H && ox + dx < a.W);
                            rv[g][it][dy][dx] = 0.f;
                        }
                    r_voffs[g][it] = r_voff;
                    if (1 && a.res) {
#pragma unroll
                        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                            for (int dx = 0; dx < 2; ++dx)
                                rv[g][it][dy][dx] = buf_load(a.res, a.r_bytes, ok[g][it][dy][dx] ? r_voff : OOB, (unsigned)((dy * a.W + dx) * a.ldr * 4));
                    }
                }
            }
            W5_BARRIER();                                      // everyone is done reading V / the patches (tg = 0) or sQ
            if (tg == 1 && more) {                             // patch buffers and fragment registers are idle
                W5_SETUP(next);
                W5_ISSUE_P(0);
                W5_ISSUE_P(1);
                W5_LOAD_B(0, 0, 0, 0);
                W5_LOAD_B(0, 0, 0, 1);
                W5_LOAD_B(0, 1, 1, 0);
            }
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int tl = (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const float m0 = acc[0][tg][g][r], m1 = acc[1][tg][g][r], m2 = acc[2][tg][g][r], m3 = acc[3][tg][g][r];
                    sQ[(((wave * 2 + 0) * 2 + g) * 32 + tl) * 32 + (lane & 31)] = m0 + m1 + m2;
                    sQ[(((wave * 2 + 1) * 2 + g) * 32 + tl) * 32 + (lane & 31)] = m1 - m2 - m3;
                }
            W5_BARRIER();
            // Stage 2: thread = (tile, co): Y[a][c] = sum_i A^T[a][i] q[i][c]; 4 tiles x 2 cout groups per thread and pass
#pragma unroll
            for (int g = 0; g < 2; ++g) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int tl = (tid >> 5) + 8 * it;
                    if (!1 && a.res) {
#pragma unroll
                        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                            for (int dx = 0; dx < 2; ++dx)
                                rv[g][it][dy][dx] = buf_load(a.res, a.r_bytes, ok[g][it][dy][dx] ? r_voffs[g][it] : OOB, (unsigned)((dy * a.W + dx) * a.ldr * 4));
                    }
                    float q[4][2];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int c = 0; c < 2; ++c) q[i][c] = sQ[(((i * 2 + c) * 2 + g) * 32 + tl) * 32 + co];
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const float ya = (q[0][c] + q[1][c] + q[2][c]) * inv;
                        const float yb = (q[1][c] - q[2][c] - q[3][c]) * inv;
                        const float oa = fmaxf(ya + bv[g] + rv[g][it][0][c], lo), ob = fmaxf(yb + bv[g] + rv[g][it][1][c], lo);
                        omax = fmaxf(omax, fmaxf(ok[g][it][0][c] ? fabsf(oa) : 0.f, ok[g][it][1][c] ? fabsf(ob) : 0.f));
                        buf_store(oa, a.y, a.y_bytes, ok[g][it][0][c] ? y_voff[g][it] : OOB, (unsigned)(c * a.ldy * 4));
                        buf_store(ob, a.y, a.y_bytes, ok[g][it][1][c] ? y_voff[g][it] : OOB, (unsigned)((a.W + c) * a.ldy * 4));
                    }
                }
            }
        }
        if (a.ymax) {          // max |y| of this item into its image's slot: one atomic per wave and item (no return value awaited)
            omax = cnl::wave_max_nonneg(omax);
            if (lane == 0) cnl::report_max(a.ymax + en * AMS, omax);
            omax = 0.f;
        }
        if (!more) break;
        item = next;
    }
#undef W5_SLOT
#undef W5_MFMA
#undef W5_ISSUE_P
#undef W5_LOAD_B
#undef W5_SETUP
}

// max |x| over [pixels][C] floats with pixel stride ld (C % 4 == 0, 16-byte aligned): non-negative floats order like their bit
// patterns, so the reduction is an unsigned atomicMax; *out must be zeroed first.  NaNs are ignored (they would poison the scale).
// blockIdx.y = image (pixels per image, one result per image): the scale of an image must not depend on its batch neighbours.
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long pixels, int C, int ld, unsigned* __restrict__ out) {
    const int c4 = C >> 2;
    const long total = pixels * c4;
    x += (long)blockIdx.y * pixels * ld;
    out += blockIdx.y * AMS;
    float m = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long px = i / c4;
        const int q = (int)(i - px * c4);
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + px * ld + q * 4);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {                                    // ONE atomic per workgroup
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        if (m > 0.f) atomicMax(out, __float_as_uint(m));
    }
}

// fp32 OHWI 3x3 weights -> U = G g G^T, scaled by S_u = 2^(13 - e) (max |U| = m 2^e: max |U S_u| in [2^12, 2^13)) and split into two
// fp16 pieces: [ci/16][xi][piece][CoutP][16 ci].  scal[2] holds max |U| (absmax_kernel over the fp32 U), scal[1] receives S_u.
__global__ __launch_bounds__(256) void weights5_kernel(const float* __restrict__ w, unsigned short* __restrict__ u5, float* __restrict__ scal,
                                                       int Cin, int Cout, int CoutP) {
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
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t == 0) scal[1] = Su;
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
    const int cc = ci >> 4, c16 = ci & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float uu[4] = {h[i][0], 0.5f * (h[i][0] + h[i][1] + h[i][2]), 0.5f * (h[i][0] - h[i][1] + h[i][2]), h[i][2]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x = uu[j] * Su;
            const _Float16 hf = (_Float16)x;
            const _Float16 lf = (_Float16)(x - (float)hf);
            const long base = ((((long)cc * 16 + (i * 4 + j)) * NP) * CoutP + co) * 16 + c16;
            u5[base] = __builtin_bit_cast(unsigned short, hf);
            u5[base + (long)CoutP * 16] = __builtin_bit_cast(unsigned short, lf);
        }
    }
}

}  // namespace cnl_wino5

// bytes of the fp16-split weights of a layer (0 when this kernel does not apply) and of the scalars behind them
size_t cnl_wino5_weight_bytes(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0 || Cin % 16) return 0;
    const size_t CoutP = (size_t)((Cout + 63) / 64) * 64;
    return (size_t)(Cin / 16) * 16 * cnl_wino5::NP * CoutP * 32;
}
size_t cnl_wino5_scalar_floats() { return 16 + 1024 * AMS; }  // [1] S_u, [2] max |U|, [16 + n AMS] max |x| of image n of the current launch (own pass)

// u_f32 = the fp32 U of the layer (already computed), u5 = destination of the pieces, scal = the layer's scalars
int cnl_wino5_transform_weights(const float* w_ohwi, const float* u_f32, size_t u_f32_floats, void* u5, float* scal, int Cin, int Cout,
                                void* stream) {
    using namespace cnl_wino5;
    const int CoutP = (Cout + 63) / 64 * 64;
    CNL_HIP(hipMemsetAsync(scal, 0, 16 * sizeof(float), (hipStream_t)stream));
    hipLaunchKernelGGL(absmax_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, u_f32, (long)(u_f32_floats / 4), 4, 4,
                       reinterpret_cast<unsigned*>(scal + 2));
    const long total = (long)CoutP * Cin;
    hipLaunchKernelGGL(weights5_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_ohwi,
                       (unsigned short*)u5, scal, Cin, Cout, CoutP);
    return cnl::check_launch("weights5_kernel");
}

extern "C" int cnl_absmax_per_image_f32(const float* x, int32_t N, int64_t pixels, int32_t C, int32_t ld, float* out, void* stream) {
    CNL_REQUIRE(x && out, CNL_E_BAD_ARG, "cnl_absmax_per_image_f32: null pointer");
    CNL_REQUIRE(N > 0 && N <= 65535 && pixels > 0 && C > 0, CNL_E_BAD_ARG, "cnl_absmax_per_image_f32: N in 1..65535, pixels and C positive");
    CNL_REQUIRE(C % 4 == 0 && ld % 4 == 0 && ld >= C && ((uintptr_t)x & 15) == 0, CNL_E_UNSUPPORTED,
                "cnl_absmax_per_image_f32: C=%d, ld=%d must be multiples of 4 (ld >= C) and x 16-byte aligned", C, ld);
    CNL_HIP(hipMemsetAsync(out, 0, sizeof(float) * (size_t)N * AMS, (hipStream_t)stream));
    const long long vec4 = (long long)pixels * (C / 4);
    const long long want = (vec4 + 256 * 16 - 1) / (256 * 16);
    const unsigned mgrid = (unsigned)(want < 1 ? 1 : (want > 64 ? 64 : want));
    hipLaunchKernelGGL(cnl_wino5::absmax_kernel, dim3(mgrid, (unsigned)N), dim3(256), 0, (hipStream_t)stream, x, (long)pixels, C, ld,
                       reinterpret_cast<unsigned*>(out));
    return cnl::check_launch("absmax_kernel");
}

// max |x| per image of p's input into scal[16 + n] (the layer's scratch: callers without the x_absmax hint; one such launch at a
// time per layer — the hint-carrying plan of engine.py never comes here)
int cnl_wino5_own_absmax(const cnl_conv_params* p, float* scal, void* stream) {
    CNL_REQUIRE(p->N <= 1024, CNL_E_UNSUPPORTED, "cnl_conv3x3_winograd_f32: more than 1024 images per launch need x_absmax");
    CNL_HIP(hipMemsetAsync(scal + 16, 0, sizeof(float) * (size_t)p->N * AMS, (hipStream_t)stream));
    const long long vec4 = (long long)p->H_in * p->W_in * (p->Cin / 4);         // per image
    const long long want = (vec4 + 256 * 16 - 1) / (256 * 16);                  // >= 16 float4 per thread
    const unsigned mgrid = (unsigned)(want < 1 ? 1 : (want > 64 ? 64 : want));
    hipLaunchKernelGGL(cnl_wino5::absmax_kernel, dim3(mgrid, (unsigned)p->N), dim3(256), 0, (hipStream_t)stream, p->x, (long)p->H_in * p->W_in,
                       p->Cin, p->ldx, reinterpret_cast<unsigned*>(scal + 16));
    return cnl::check_launch("absmax_kernel");
}

// Launch (arguments already validated by cnl_conv3x3_winograd_f32); u5 = the fp16-split weights, scal = the layer's scalars.
int cnl_wino5_launch(const cnl_conv_params* p, const void* u5, float* scal, void* stream) {
    using namespace cnl_wino5;
    Args a;
    CNL_REQUIRE(p->x_absmax || p->N <= 1024, CNL_E_UNSUPPORTED, "cnl_conv3x3_winograd_f32: more than 1024 images per launch need x_absmax");
    a.x = p->x; a.u3 = u5; a.xmax = p->x_absmax ? p->x_absmax : scal + 16; a.su = scal + 1; a.ymax = reinterpret_cast<unsigned*>(p->y_absmax);
    a.bias = p->bias; a.res = p->residual; a.y = p->y;
    const int upf = (p->flags & CNL_UPSAMPLE_IN) ? 2 : 1;
    a.N = p->N; a.Hs = p->H_in; a.Ws = p->W_in; a.H = p->H_in * upf; a.W = p->W_in * upf; a.Cin = p->Cin; a.Cout = p->Cout;
    a.CoutP = (p->Cout + 63) / 64 * 64;
    a.ldx = p->ldx; a.ldy = p->ldy; a.ldr = p->ldr;
    a.CC = p->Cin / 16;
    a.nb = a.CoutP / BN; a.bx = (a.W + 15) / 16; a.by = (a.H + 15) / 16;
    const long long blocks = (long long)p->N * a.by * a.bx * a.nb;
    CNL_REQUIRE(blocks < (1ll << 31), CNL_E_UNSUPPORTED, "cnl_conv3x3_winograd_f32: grid too large");
    a.blocks = (int)blocks;
    const unsigned long long xb = (((unsigned long long)p->N * p->H_in * p->W_in - 1) * p->ldx + p->Cin) * 4ull;
    const unsigned long long ub = (unsigned long long)cnl_wino5_weight_bytes(p->Cin, p->Cout);
    const unsigned long long Mo = (unsigned long long)p->N * a.H * a.W;
    const unsigned long long yb = ((Mo - 1) * p->ldy + p->Cout) * 4ull;
    const unsigned long long rb = p->residual ? ((Mo - 1) * p->ldr + p->Cout) * 4ull : 0ull;
    const unsigned long long slack = (unsigned long long)(a.W + 2) * 4ull;     // scalar-offset reach of the epilogue stores
    CNL_REQUIRE(xb < 0xFFFFFF00ull && ub < 0xFFFFFF00ull && yb + slack * p->ldy < 0xFFFFFF00ull && rb + slack * (p->residual ? p->ldr : 0) < 0xFFFFFF00ull,
                CNL_E_UNSUPPORTED, "cnl_conv3x3_winograd_f32: tensor spans >= 4 GiB; split the batch");
    a.x_bytes = (unsigned)xb; a.u_bytes = (unsigned)ub; a.y_bytes = (unsigned)yb; a.r_bytes = (unsigned)rb;
    a.flags = p->flags;
    a.order = (a.nb & 1) ? 0 : 2;          
    static cnl::DeviceOnce once;
    int n_cu = 0;                          // persistent workgroups: one per CU, walking the work items with stride gridDim.x
    int rc = cnl::kernel_setup(once, reinterpret_cast<const void*>(&winograd5_kernel), LDS_BYTES, &n_cu);
    if (rc != CNL_OK) return rc;
    // the scale of the activations: max |x| of this launch's input — handed over by the producer (x_absmax), else one pass over it
    // (stream-ordered before the convolution)
    if (!p->x_absmax && (rc = cnl_wino5_own_absmax(p, scal, stream)) != CNL_OK) return rc;
    const unsigned grid = (unsigned)(blocks < (long long)n_cu ? blocks : (long long)n_cu);
    hipLaunchKernelGGL(winograd5_kernel, dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, a);
    return cnl::check_launch("winograd5_kernel");
}
