// winograd3.hip — Winograd F(2x2,3x3) with the 16 position GEMMs on the BF16 matrix cores, at fp32 accuracy.
//
// gfx950's fp32 MFMA (v_mfma_f32_32x32x2_f32, what winograd.hip / winograd2.hip run on) issues at the fp32 VECTOR rate
// (157 TFLOP/s); v_mfma_f32_32x32x16_bf16 is 16x faster.  Every fp32 number is the exact sum of three bf16 numbers
// (a = a1 + a2 + a3, 8 + 8 + 8 significand bits, by truncation — no rounding anywhere), every bf16 x bf16 product is exact in
// fp32, and the matrix core accumulates in fp32.  So V x U is formed from the six largest cross terms
//       a1 b1 + a1 b2 + a2 b1 + a2 b2 + a1 b3 + a3 b1          (dropped: a2 b3, a3 b2, a3 b3 <= 2^-24 |a b|)
// = 6 bf16 MFMAs instead of 16 fp32-rate ones per product: 2.67x less matrix-pipe time, with an error at or below that of the
// fp32 matrix core itself (tools/bf16x3_probe.hip on MI355X, K = 2304: max |err| 8.0e-6 vs 1.27e-5 for v_mfma_f32_32x32x2_f32;
// 3 terms would give 1.5e-4).  Inputs, weights, accumulation and outputs stay fp32; only the multiplier array is bf16.
// Same call sites as winograd.hip (reference models/meta.py:24-26, models/layers.py:72-77: every 3x3 / stride-1 conv).
//
// The split moves the bottleneck from the matrix pipe to everything around it, so the decomposition differs from winograd2.hip:
//   * Workgroup = 4 waves, ONE per SIMD (512 registers each), 8x8 tiles (16x16 output px) x 64 output channels x 16 positions:
//     per input element the split costs 5.5 VALU ops, and only >= 64 tiles x 64 couts per CU keeps that — and the weight
//     stream from L2 (6 bytes per U element, 32 B/clk/CU) — under the matrix time.  Unlike beside the fp32 MFMA, up to ~6 VALU
//     / LDS instructions per MFMA issue for free beside the bf16 MFMA (bf16x3_probe: 32.5 cycles per MFMA with 6 VALU between).
//   * wave i owns transform row i (positions 4i..4i+3) for all 64 tiles and both 32-cout groups: 256 accumulator registers,
//     per position and 16-channel chunk 6 A fragments (3 pieces x 2 tile groups, ds_read_b128) + 6 B fragments (3 pieces x 2
//     cout groups, global -> registers, one position ahead) -> 24 MFMAs; 96 MFMAs per wave and chunk.
//   * V rows of position row i are produced AND consumed by wave i (rows of B^T d are independent): V is wave-private, single
//     buffered (96 KB for the workgroup) and ordered by program order alone — positions {0,1} of chunk n+1 are produced while
//     positions {1,2} of chunk n are multiplied (their fragments are in registers by then), positions {2,3} while {3} and {0 of
//     n+1} are.  lane = (tile, 4 channels): 6 ds_read_b128 of the patch, 12 fma + 8 add + 32 split + 12 pack ops, 6
//     ds_write_b64 per (item, position pair); hand-placed 6 VALU per MFMA slice.
//   * only the input patch (18x18 px x 16 ch, double-buffered, LDS-DMA two chunks ahead) is shared: ONE barrier per chunk.
//   * U is pre-split once per weight load (cnl_winograd_transform_weights_f32): [ci/16][position][piece][cout][16 ci] bf16.
#include "cnl_common.h"
#include <cstdlib>

#pragma clang fp contract(off)

// timing experiments (results are wrong when any is set): drop parts of the chunk loop
#ifndef W3_EXP_NOB
#define W3_EXP_NOB 0
#endif
#ifndef W3_EXP_NOA
#define W3_EXP_NOA 0
#endif
#ifndef W3_EXP_NOXR
#define W3_EXP_NOXR 0
#endif
#ifndef W3_EXP_NOXW
#define W3_EXP_NOXW 0
#endif
#ifndef W3_EXP_NOOPS
#define W3_EXP_NOOPS 0
#endif
#ifndef W3_HOLD_T
#define W3_HOLD_T 1
#endif
#ifndef W3_EPI_HOIST
#define W3_EPI_HOIST 1
#endif

namespace cnl_wino3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

struct Args {
    const float* x;
    const void* u3;                   // pre-split weights (bf16 pieces)
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
    long long* trace;                 // CNL_W3TRACE builds only: per-wave cycle sums of the phases of a work item
};

constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr int T = 64;                       // tiles per workgroup: 8 x 8
constexpr int BN = 64;
constexpr int PH = 18, PW = 18;             // patch height / width in pixels
constexpr int PWP = 19;                     // padded patch row of the LDS image [py][quad][PWP][4 floats]
constexpr int IT_STRIDE = 4 * 4 * PWP * 16; // patch bytes between transform items (two tile rows = four patch rows)
constexpr int VPIECE = T * 32;              // 2048: one (position, piece) plane of a wave's V: [64 tiles][16 ci bf16]
constexpr int VW_BYTES = 4 * 3 * VPIECE;    // 24576 per wave
constexpr int V_BYTES = 4 * VW_BYTES;       // 98304
constexpr int P_SLOTS = 1408;               // 1368 used; 5 x 256 (all waves) + 128 (waves 0-1)
constexpr int P_BYTES = P_SLOTS * 16;       // 22528 per buffer (two buffers)
constexpr int LDS_BYTES = V_BYTES + 2 * P_BYTES;                 // 143360: one workgroup per CU

__device__ __forceinline__ void dma16(const float* base, unsigned bytes, char* lds_dst, unsigned voffset, unsigned soffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_dst, 16, voffset, soffset, 0, 0);
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
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, voffset, soffset, 0);
}
__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_zero() {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const u32x4 zz = {0u, 0u, 0u, 0u};
    return mfma16(zz, zz, z);
}
__device__ __forceinline__ f32x4 lds_f4(const char* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ u32x4 lds_u4(const char* p) { return *reinterpret_cast<const u32x4*>(p); }

// Registers of one wave's input-transform pipeline.  A "pass-item" = (item: tile, 4 channels) x (pass P: position pair {2P, 2P+1}
// of the wave's row).  Its 64 VALU operations are indexed 0..63 so that the main loop can place them six per MFMA slice:
//   0..11   t[c] = da[c] + sg * db[c]      three patch columns (P .. P+2), four channels: the wave's row of B^T d
//   12..19  v[0], v[1]                     the two positions of the pair: (t0 - t2, t1 + t2) or (t2 - t1, t1 - t3)
//   20..63  exact three-way bf16 split of v[0], v[1]: per element  h = x & 0xFFFF0000, r = x - h, m = r & 0xFFFF0000, l = r - m
//           (pieces = the top halves of x, r, l), per channel pair three v_perm_b32 that pack the top halves; 22 per v
struct Xf {
    f32x4 da[2][3], db[2][3];       // [register set][column]: rows ra / rb of the patch (read one pass-item ahead)
    f32x4 t[3], v[2];
    f32x4 th[4][2];                 // t of patch columns 1, 2 of each item, kept from pass 0 for pass 1 (W3_HOLD_T)
    float h[4], r[4], l[4];
    unsigned pk[2][3][2];           // [position of the pair][piece][channel pair]
};
__device__ __forceinline__ void xop(Xf& s, const int set, const int P, const int op, const float sg, const int it = 0) {
    if (op < 12) {
        const int c = op >> 2, e = op & 3;
#if W3_HOLD_T
        // pass 0 computes t of columns 0, 1, 2 and keeps 1, 2; pass 1 computes only column 3 (read into slot 2 of the set)
        if (P == 0) {
            const float t_ = __builtin_fmaf(s.db[set][c][e], sg, s.da[set][c][e]);
            if (c == 0) s.t[0][e] = t_; else s.th[it][c - 1][e] = t_;
        } else if (c == 2) s.t[2][e] = __builtin_fmaf(s.db[set][2][e], sg, s.da[set][2][e]);
#else
        s.t[c][e] = __builtin_fmaf(s.db[set][c][e], sg, s.da[set][c][e]);
#endif
    } else if (op < 20) {
        const int vi = (op - 12) >> 2, e = op & 3;
#if W3_HOLD_T
        if (P == 0) s.v[vi][e] = vi == 0 ? s.t[0][e] - s.th[it][1][e] : s.th[it][0][e] + s.th[it][1][e];
        else s.v[vi][e] = vi == 0 ? s.th[it][1][e] - s.th[it][0][e] : s.th[it][0][e] - s.t[2][e];
#else
        if (P == 0) s.v[vi][e] = vi == 0 ? s.t[0][e] - s.t[2][e] : s.t[1][e] + s.t[2][e];
        else s.v[vi][e] = vi == 0 ? s.t[1][e] - s.t[0][e] : s.t[0][e] - s.t[2][e];
#endif
    } else if (op < 64) {
        // per v[vi] 22 operations, ordered so that consecutive ones are independent (a dependent VALU instruction issues ~4 cycles
        // later than an independent one): the four elements advance through the split side by side
        const int q = op - 20, vi = q / 22, w = q % 22;
        if (w < 4) s.h[w] = __uint_as_float(__float_as_uint(s.v[vi][w]) & 0xFFFF0000u);
        else if (w < 8) s.r[w - 4] = s.v[vi][w - 4] - s.h[w - 4];
        else if (w < 10) s.pk[vi][0][w - 8] = __builtin_amdgcn_perm(__float_as_uint(s.v[vi][2 * (w - 8) + 1]), __float_as_uint(s.v[vi][2 * (w - 8)]), 0x07060302u);
        else if (w < 14) s.h[w - 10] = __uint_as_float(__float_as_uint(s.r[w - 10]) & 0xFFFF0000u);
        else if (w < 18) s.l[w - 14] = s.r[w - 14] - s.h[w - 14];
        else if (w < 20) s.pk[vi][1][w - 18] = __builtin_amdgcn_perm(__float_as_uint(s.r[2 * (w - 18) + 1]), __float_as_uint(s.r[2 * (w - 18)]), 0x07060302u);
        else s.pk[vi][2][w - 20] = __builtin_amdgcn_perm(__float_as_uint(s.l[2 * (w - 20) + 1]), __float_as_uint(s.l[2 * (w - 20)]), 0x07060302u);
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void winograd3_kernel(const Args a) {
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
    const unsigned u_pos = 3u * u_piece;
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
    const int dstv = wave * VW_BYTES + (t_tyl * 8 + t_tx) * 32 + (((t_q >> 1) ^ t_tyl) << 4) + (t_q & 1) * 8;   // + it*512 + (j*3+k)*VPIECE
    const int fragA = wave * VW_BYTES + (lane & 31) * 32 + ((hi ^ ((lane >> 3) & 1)) << 4);                    // + (j*3+k)*VPIECE + tg*1024
    const float lo = (a.flags & CNL_RELU) ? 0.f : -__builtin_inff();

    int n, y0, x0, n0;
    unsigned p_off[6], u_voff;
    float bias_n[2];           // bias of the item set up last (the next one, from the epilogue's prefetch on)
#define W3_SETUP(item_)                                                                                          \
    do {                                                                                                         \
        unsigned b_ = cnl::xcd_remap((item_), (unsigned)a.blocks);                                               \
        const int nbi_ = b_ % a.nb; b_ /= a.nb;                                                                  \
        const int bxi_ = b_ % a.bx; b_ /= a.bx;                                                                  \
        const int byi_ = b_ % a.by;                                                                              \
        n = b_ / a.by; y0 = byi_ * 16; x0 = bxi_ * 16; n0 = nbi_ * BN;                                           \
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
        /* bias of this thread's two epilogue columns: requested now, used after the chunk loop */               \
        _Pragma("unroll") for (int g_ = 0; g_ < 2; ++g_) {                                                       \
            const int col_ = n0 + g_ * 32 + (tid & 31);                                                          \
            bias_n[g_] = col_ < a.Cout ? a.bias[col_] : 0.f;                                                     \
        }                                                                                                        \
    } while (0)
    // the channel-chunk offset rides in the SCALAR offset (the bounds check looks at the vector offset alone, so halo lanes still
    // read zeros); a chunk past the end is not fetched
#define W3_ISSUE_P(cc_)                                                                                          \
    do {                                                                                                         \
        if ((cc_) < a.CC) {                                                                                      \
            char* d_ = sP + ((cc_) & 1) * P_BYTES;                                                               \
            _Pragma("unroll") for (int i = 0; i < 5; ++i)                                                        \
                dma16(a.x, a.x_bytes, d_ + (i * 256 + wave * 64) * 16, p_off[i], (unsigned)((cc_) * 64));        \
            if (wave < 2) dma16(a.x, a.x_bytes, d_ + (1280 + wave * 64) * 16, p_off[5], (unsigned)((cc_) * 64)); \
        }                                                                                                        \
    } while (0)
    // B fragments of position xi0 + j_ of chunk cc_, cout group g_ (three pieces): global -> registers
#define W3_LOAD_B(cc_, j_, buf_, g_)                                                                             \
    do {                                                                                                         \
        if ((cc_) < a.CC) {                                                                                      \
            const unsigned so_ = (unsigned)(cc_) * u_chunk + (unsigned)(xi0 + (j_)) * u_pos + (unsigned)(g_) * 1024u; \
            _Pragma("unroll") for (int kk_ = 0; kk_ < 3; ++kk_)                                                  \
                fb[buf_][g_][kk_] = buf_load16(a.u3, a.u_bytes, u_voff, so_ + (unsigned)kk_ * u_piece);          \
        }                                                                                                        \
    } while (0)
#define W3_READ_A(j_, buf_, g_)                                                                                  \
    _Pragma("unroll") for (int kk_ = 0; kk_ < 3; ++kk_) fa[buf_][g_][kk_] = lds_u4(sV + fragA + ((j_) * 3 + kk_) * VPIECE + (g_) * 1024)
    // MFMA s_ (0..23) of a position: cout group s_ / 12 — group 0 first, so that its B registers are free (and refilled for the
    // position after next) from mid-slot on: every B load is in flight for 1.5 slots —, then term = (s_ % 12) >> 1 in the order
    // (a3 b1, a1 b3, a2 b2, a2 b1, a1 b2, a1 b1) — small terms first —, tile group s_ & 1
#define W3_MFMA(j_, buf_, s_)                                                                                    \
    do {                                                                                                         \
        const int cg_ = (s_) / 12, term_ = ((s_) % 12) >> 1, tg_ = (s_) & 1;                                     \
        const int ka_ = term_ == 0 ? 2 : (term_ == 2 || term_ == 3) ? 1 : 0;                                     \
        const int kb_ = term_ == 1 ? 2 : (term_ == 2 || term_ == 4) ? 1 : 0;                                     \
        acc[j_][tg_][cg_] = mfma16(fa[buf_][tg_][ka_], fb[buf_][cg_][kb_], acc[j_][tg_][cg_]);                   \
    } while (0)
    // patch reads of pass-item (P_, it_) into register set set_; pa_ / pb_ = patch buffer + src_a / src_b
#define W3_X_READ1(set_, pa_, pb_, P_, it_, c_, row_)                                                            \
    do {                                                                                                         \
        if (W3_HOLD_T && (P_) == 1 && (c_) != 2) break;                                                          \
        if ((row_) == 0) xf.da[set_][c_] = lds_f4((pa_) + (it_) * IT_STRIDE + ((P_) + (c_)) * 16);               \
        else xf.db[set_][c_] = lds_f4((pb_) + (it_) * IT_STRIDE + ((P_) + (c_)) * 16);                           \
    } while (0)
#define W3_X_READ(set_, pa_, pb_, P_, it_)                                                                       \
    _Pragma("unroll") for (int c_ = 0; c_ < 3; ++c_) {                                                           \
        W3_X_READ1(set_, pa_, pb_, P_, it_, c_, 0);                                                              \
        W3_X_READ1(set_, pa_, pb_, P_, it_, c_, 1);                                                              \
    }
#define W3_X_WRITE(P_, it_)                                                                                      \
    _Pragma("unroll") for (int jj_ = 0; jj_ < 2; ++jj_)                                                          \
        _Pragma("unroll") for (int kk_ = 0; kk_ < 3; ++kk_)                                                      \
            *reinterpret_cast<u32x2*>(sV + dstv + (it_) * 512 + ((2 * (P_) + jj_) * 3 + kk_) * VPIECE) =         \
                u32x2{xf.pk[jj_][kk_][0], xf.pk[jj_][kk_][1]};
    // workgroup barrier WITHOUT the vmcnt(0) that __syncthreads() adds when LDS-DMA is in flight (own LDS accesses drained)
#define W3_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    // One position slot = 24 MFMAs in 24 slices fenced by sched_barrier(0); every slice carries at most ~6 other instructions.
    //   j_ / buf_    position multiplied in this slot and its fragment buffer
    //   (cA_, jA_)   the next position: its A fragments and the B fragments of its cout group 1 are fetched into buf_ ^ 1 (PA_)
    //   (cB_, jB_)   the position after next: B fragments of its cout group 0 go into buf_ from slice 12 on (PB_)
    //   JOBS_        transform the two pass-items (P_, it0_) [slices 0-11, register set 0] and (P_, it0_ + 1) [12-23, set 1]
    //   RDN_         in slices 12-17 read the patch of the NEXT slot's first pass-item (nP_, nIt_) from (npa_, npb_) into set 0
    //   MID_         (slot of position 0) before slice 12: this wave's DMAs of the next patch landed, barrier; slice 13: DMA
    //                of the patch after that (chunk dC_)
#define W3_SLOT(j_, buf_, cA_, jA_, PA_, cB_, jB_, PB_, JOBS_, P_, it0_, RDN_, nP_, nIt_, npa_, npb_, MID_, dC_) \
    do {                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        _Pragma("unroll") for (int k = 0; k < 24; ++k) {                                                         \
            const int pi = k / 12, ks = k % 12;                                                                  \
            if ((MID_) && k == 12) {                                                                             \
                W3T_MID0();                                                                                      \
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   /* all but the newest 6 B loads: the patch DMA is older */ \
                W3_BARRIER();                                                                                    \
                W3T_MID1();                                                                                      \
                __builtin_amdgcn_sched_barrier(0);                                                               \
            }                                                                                                    \
            W3_MFMA(j_, buf_, k);                                                                                \
            if (PA_) {                                                                                           \
                if (k == 0 && !W3_EXP_NOB) W3_LOAD_B(cA_, jA_, (buf_) ^ 1, 1);                                   \
                if (k >= 6 && k < 12 && !W3_EXP_NOA) {                                                           \
                    const int g_ = (k - 6) / 3, kk_ = (k - 6) % 3;                                               \
                    fa[(buf_) ^ 1][g_][kk_] = lds_u4(sV + fragA + ((jA_) * 3 + kk_) * VPIECE + g_ * 1024);       \
                }                                                                                                \
            }                                                                                                    \
            if ((PB_) && k == 12 && !W3_EXP_NOB) W3_LOAD_B(cB_, jB_, buf_, 0);                                   \
            if (JOBS_) {                                                                                         \
                if (k < 6 && !W3_EXP_NOXR) W3_X_READ1(1, pa, pb, P_, (it0_) + 1, k >> 1, k & 1);                 \
                if (ks < 11) {                                                                                   \
                    if (!W3_EXP_NOOPS) { _Pragma("unroll") for (int o_ = 0; o_ < 6; ++o_) xop(xf, pi, P_, ks * 6 + o_, sg, (it0_) + pi); } \
                } else if (!W3_EXP_NOXW) {                                                                       \
                    if (pi == 0) { W3_X_WRITE(P_, it0_); } else { W3_X_WRITE(P_, (it0_) + 1); }                  \
                }                                                                                                \
            }                                                                                                    \
            if ((RDN_) && k >= 12 && k < 18 && !W3_EXP_NOXR) W3_X_READ1(0, npa_, npb_, nP_, nIt_, (k - 12) >> 1, k & 1); \
            if ((MID_) && k == 13) W3_ISSUE_P(dC_);                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
        }                                                                                                        \
    } while (0)

#ifdef CNL_W3TRACE
    long long tr[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tr_t = clock64(), tr_m = 0;
    const long long tr_start = tr_t;
#define W3T(i_) do { const long long c_ = clock64(); tr[i_] += c_ - tr_t; tr_t = c_; } while (0)
#define W3T_MID0() do { tr_m = clock64(); } while (0)
#define W3T_MID1() do { tr[5] += clock64() - tr_m; } while (0)
#else
#define W3T(i_) do { } while (0)
#define W3T_MID0() do { } while (0)
#define W3T_MID1() do { } while (0)
#endif
    unsigned item = blockIdx.x;
    W3_SETUP(item);
    W3_ISSUE_P(0);
    W3_ISSUE_P(1);
    u32x4 fa[2][2][3];       // A fragments: [buffer][tile group][piece]
    u32x4 fb[2][2][3];       // B fragments: [buffer][cout group][piece]
    W3_LOAD_B(0, 0, 0, 0);
    W3_LOAD_B(0, 0, 0, 1);
    W3_LOAD_B(0, 1, 1, 0);
    bool first = true;
    while (true) {
        f32x16 acc[4][2][2];     // [position j of row `wave`][tile group][cout group]
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[j][g >> 1][g & 1] = mfma_zero();
        Xf xf;
        W3T(0);

        // patches 0 / 1 landed (this wave's parts)?  Their DMAs are followed in this wave's VMEM queue by the 9 B loads of chunk 0
        // and the 32 stores of the previous item's second epilogue pass (its residual loads are older): a counted wait lets
        // those stay in flight.  (The two bias loads of the setup are scalar loads.)
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(41)" ::: "memory");
        first = false;
        W3_BARRIER();                                         // ... and everybody's
        W3T(1);
        {   // input transform of chunk 0, all four positions (not overlapped with MFMAs): eight pass-items, each read one ahead
            const char* pa = sP + src_a;
            const char* pb = sP + src_b;
            W3_X_READ(0, pa, pb, 0, 0);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int it = g >> 1, P = g & 1, set = g & 1;
                if (g < 7) { W3_X_READ(set ^ 1, pa, pb, (g + 1) & 1, (g + 1) >> 1); }
#pragma unroll
                for (int o = 0; o < 64; ++o) xop(xf, set, P, o, sg, it);
                W3_X_WRITE(P, it);
            }
        }
        W3T(2);
        W3_READ_A(0, 0, 0);
        W3_READ_A(0, 0, 1);
        {   // position 0 of chunk 0: no transform work yet; patch 1 is read from slice 12 on, patch 2 requested
            const char* npa = sP + P_BYTES + src_a;
            const char* npb = sP + P_BYTES + src_b;
            const char* pa = npa; const char* pb = npb;       // (unused: JOBS_ = 0)
            W3_SLOT(0, 0, 0, 1, 1, 0, 2, 1, 0, 0, 0, 1, 0, 0, npa, npb, 1, 2);
            (void)pa; (void)pb;
        }
        // chunk n: positions 1..3 of chunk n-1, then position 0 of chunk n; beside them the transform of chunk n
        for (int cn = 1; cn < a.CC; ++cn) {
            const char* pa = sP + (cn & 1) * P_BYTES + src_a;
            const char* pb = sP + (cn & 1) * P_BYTES + src_b;
            const char* npa = sP + ((cn + 1) & 1) * P_BYTES + src_a;
            const char* npb = sP + ((cn + 1) & 1) * P_BYTES + src_b;
            W3_SLOT(1, 1, cn - 1, 2, 1, cn - 1, 3, 1, 1, 0, 0, 1, 0, 2, pa, pb, 0, 0);      // positions {0,1} of chunk cn, items 0-1
            W3_SLOT(2, 0, cn - 1, 3, 1, cn, 0, 1, 1, 0, 2, 1, 1, 0, pa, pb, 0, 0);          //                          items 2-3
            W3_SLOT(3, 1, cn, 0, 1, cn, 1, 1, 1, 1, 0, 1, 1, 2, pa, pb, 0, 0);              // positions {2,3} of chunk cn, items 0-1
            W3_SLOT(0, 0, cn, 1, 1, cn, 2, 1, 1, 1, 2, 1, 0, 0, npa, npb, 1, cn + 2);       //                          items 2-3
        }
        {   // positions 1..3 of the last chunk: MFMAs only
            const char* pa = sP; const char* pb = sP;
            const int cl = a.CC - 1;
            W3_SLOT(1, 1, cl, 2, 1, cl, 3, 1, 0, 0, 0, 0, 0, 0, pa, pb, 0, 0);
            W3_SLOT(2, 0, cl, 3, 1, cl, 0, 0, 0, 0, 0, 0, 0, 0, pa, pb, 0, 0);
            W3_SLOT(3, 1, cl, 0, 0, cl, 0, 0, 0, 0, 0, 0, 0, 0, pa, pb, 0, 0);
            (void)pa; (void)pb;
        }

        W3T(3);
        // ---- epilogue: Y = A^T M A.  Stage 1 (this wave's row of positions, in registers): q_c = sum_j A^T[c][j] M[i][j]; the
        // four rows meet through LDS, one tile group per pass ([4 i][2 c][2 cout groups][32 tiles][32 co] = 64 KB of the V region) ----
        float* sQ = reinterpret_cast<float*>(smem);
        const int co = tid & 31;
        const int en = n, ey0 = y0, ex0 = x0, en0 = n0;        // this item's coordinates (the setup below moves on to the next)
        const bool full = (y0 + 16 <= a.H) && (x0 + 16 <= a.W) && (n0 + BN <= a.Cout);
        const float bv[2] = {bias_n[0], bias_n[1]};
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
                    if (W3_EPI_HOIST && a.res) {
#pragma unroll
                        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                            for (int dx = 0; dx < 2; ++dx)
                                rv[g][it][dy][dx] = buf_load(a.res, a.r_bytes, ok[g][it][dy][dx] ? r_voff : OOB, (unsigned)((dy * a.W + dx) * a.ldr * 4));
                    }
                }
            }
            W3T(8);
            W3_BARRIER();                                      // everyone is done reading V / the patches (tg = 0) or sQ
            W3T(9);
            if (tg == 1 && more) {                             // patch buffers and fragment registers are idle
                W3_SETUP(next);
                W3_ISSUE_P(0);
                W3_ISSUE_P(1);
                W3_LOAD_B(0, 0, 0, 0);
                W3_LOAD_B(0, 0, 0, 1);
                W3_LOAD_B(0, 1, 1, 0);
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
            W3_BARRIER();
            W3T(10);
            // Stage 2: thread = (tile, co): Y[a][c] = sum_i A^T[a][i] q[i][c]; 4 tiles x 2 cout groups per thread and pass
#pragma unroll
            for (int g = 0; g < 2; ++g) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int tl = (tid >> 5) + 8 * it;
                    if (!W3_EPI_HOIST && a.res) {
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
                        const float ya = q[0][c] + q[1][c] + q[2][c];
                        const float yb = q[1][c] - q[2][c] - q[3][c];
                        buf_store(fmaxf(ya + bv[g] + rv[g][it][0][c], lo), a.y, a.y_bytes, ok[g][it][0][c] ? y_voff[g][it] : OOB, (unsigned)(c * a.ldy * 4));
                        buf_store(fmaxf(yb + bv[g] + rv[g][it][1][c], lo), a.y, a.y_bytes, ok[g][it][1][c] ? y_voff[g][it] : OOB, (unsigned)((a.W + c) * a.ldy * 4));
                    }
                }
            }
        }
        W3T(11);
#ifdef CNL_W3TRACE
        tr[4] = tr[8] + tr[9] + tr[10] + tr[11];
        tr[6] += 1;
#endif
        if (!more) break;
        item = next;
    }
#ifdef CNL_W3TRACE
    if (a.trace && lane == 0) {
        long long* t_ = a.trace + ((long)blockIdx.x * 4 + wave) * 12;
        tr[7] = clock64() - tr_start;
        for (int i = 0; i < 12; ++i) t_[i] = tr[i];
    }
#endif
#undef W3_SLOT
#undef W3_MFMA
#undef W3_ISSUE_P
#undef W3_LOAD_B
#undef W3_SETUP
}

// fp32 OHWI 3x3 weights -> U = G g G^T split into three bf16 pieces: [ci/16][xi][piece][CoutP][16 ci]
__global__ __launch_bounds__(256) void weights3_kernel(const float* __restrict__ w, unsigned short* __restrict__ u3, int Cin, int Cout, int CoutP) {
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
    const int cc = ci >> 4, c16 = ci & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float uu[4] = {h[i][0], 0.5f * (h[i][0] + h[i][1] + h[i][2]), 0.5f * (h[i][0] - h[i][1] + h[i][2]), h[i][2]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x = uu[j];
            const float hf = __uint_as_float(__float_as_uint(x) & 0xFFFF0000u);
            const float r1 = x - hf;
            const float mf = __uint_as_float(__float_as_uint(r1) & 0xFFFF0000u);
            const float r2 = r1 - mf;
            const long base = ((((long)cc * 16 + (i * 4 + j)) * 3) * CoutP + co) * 16 + c16;
            u3[base] = (unsigned short)(__float_as_uint(x) >> 16);
            u3[base + (long)CoutP * 16] = (unsigned short)(__float_as_uint(r1) >> 16);
            u3[base + 2l * CoutP * 16] = (unsigned short)(__float_as_uint(r2) >> 16);
        }
    }
}

}  // namespace cnl_wino3

// bytes of the pre-split weights behind the fp32 U of the same layer (0 when this kernel does not apply)
size_t cnl_wino3_weight_bytes(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0 || Cin % 16) return 0;
    const size_t CoutP = (size_t)((Cout + 63) / 64) * 64;
    return (size_t)(Cin / 16) * 16 * 3 * CoutP * 32;
}

int cnl_wino3_transform_weights(const float* w_ohwi, void* u3, int Cin, int Cout, void* stream) {
    const int CoutP = (Cout + 63) / 64 * 64;
    const long total = (long)CoutP * Cin;
    hipLaunchKernelGGL(cnl_wino3::weights3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_ohwi,
                       (unsigned short*)u3, Cin, Cout, CoutP);
    return cnl::check_launch("weights3_kernel");
}

// Launch (arguments already validated by cnl_conv3x3_winograd_f32); u3 = the pre-split weights.
int cnl_wino3_launch(const cnl_conv_params* p, const void* u3, void* stream) {
    using namespace cnl_wino3;
    Args a;
    a.x = p->x; a.u3 = u3; a.bias = p->bias; a.res = p->residual; a.y = p->y;
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
    const unsigned long long ub = (unsigned long long)cnl_wino3_weight_bytes(p->Cin, p->Cout);
    const unsigned long long Mo = (unsigned long long)p->N * a.H * a.W;
    const unsigned long long yb = ((Mo - 1) * p->ldy + p->Cout) * 4ull;
    const unsigned long long rb = p->residual ? ((Mo - 1) * p->ldr + p->Cout) * 4ull : 0ull;
    const unsigned long long slack = (unsigned long long)(a.W + 2) * 4ull;     // scalar-offset reach of the epilogue stores
    CNL_REQUIRE(xb < 0xFFFFFF00ull && ub < 0xFFFFFF00ull && yb + slack * p->ldy < 0xFFFFFF00ull && rb + slack * (p->residual ? p->ldr : 0) < 0xFFFFFF00ull,
                CNL_E_UNSUPPORTED, "cnl_conv3x3_winograd_f32: tensor spans >= 4 GiB; split the batch");
    a.x_bytes = (unsigned)xb; a.u_bytes = (unsigned)ub; a.y_bytes = (unsigned)yb; a.r_bytes = (unsigned)rb;
    a.flags = p->flags;
    a.trace = nullptr;
#ifdef CNL_W3TRACE
    if (const char* e = getenv("CNL_TRACE_PTR")) a.trace = (long long*)strtoull(e, nullptr, 0);
#endif
    static bool attr_done = false;
    if (!attr_done) {
        CNL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&winograd3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_done = true;
    }
    static int n_cu = 0;         // persistent workgroups: one per CU (140 KB of LDS), walking the work items with stride gridDim.x
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        CNL_HIP(hipGetDevice(&dev));
        CNL_HIP(hipGetDeviceProperties(&prop, dev));
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const unsigned grid = (unsigned)(blocks < (long long)n_cu ? blocks : (long long)n_cu);
    hipLaunchKernelGGL(winograd3_kernel, dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, a);
    return cnl::check_launch("winograd3_kernel");
}
