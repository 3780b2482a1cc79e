// winograd4.hip — the bf16-split Winograd kernel of winograd3.hip (same arithmetic: exact three-way bf16 split of V and U, six
// cross terms per product, fp32 accumulation) re-decomposed for TWO waves per SIMD.
//
// winograd3.hip is bound by instruction issue of its single wave per SIMD (~14 cycles per MFMA + 4 per VALU + 8 per LDS / VMEM
// instruction, nothing to cover a wait: ~55 cycles per MFMA against 32 of matrix time).  With two waves per SIMD the issue slots of
// one wave's transform hide behind the other's MFMAs (tools/bf16x3_probe.hip: 6 VALU + 1 LDS per MFMA at 32.5 cycles).  The
// obstacle is registers — 256 per wave, 128 of them accumulators — so the work item is cut along the POSITIONS:
//   workgroup = 8 waves, the same 8x8 tiles x 64 couts x 16 positions (140 KB LDS, one workgroup per CU);
//   wave (i, jp) owns positions {2 jp, 2 jp + 1} of transform row i for all 64 tiles and both cout groups: 2 x 2 x 2 accumulator
//   tiles = 128 registers; per position and 16-channel chunk 6 A fragments (double-buffered) + 6 B fragments (rolling: the cout
//   group 0 MFMAs run first and their registers are refilled for the next position at mid-slot) -> 24 MFMAs, 48 per chunk;
//   V of a position is produced AND consumed by the same wave (private, single-buffered, program order): while position p of
//   chunk n is multiplied, V[p] of chunk n+1 is computed — one (item, position) at a time: 4 patch reads, 8 fma (two columns of
//   B^T d), 4 add, 22 split / pack ops, 3 ds_write_b64 — 34 VALU per MFMA-sextet, placed six per MFMA slice;
//   every U and V element is still loaded / produced by exactly one wave (no duplicated streams); the only shared data is the
//   input patch (double-buffered LDS-DMA, ONE barrier per chunk).
// Weights: the pre-split layout of winograd3.hip.  Same call sites (reference models/meta.py:24-26, models/layers.py:72-77).
#include "cnl_common.h"
#include <cstdlib>

#pragma clang fp contract(off)

namespace cnl_wino4 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

struct Args {
    const float* x;
    const void* u3;                   // pre-split weights (bf16 pieces): [ci/16][position][piece][CoutP][16 ci]
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
};

constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr int T = 64;                       // tiles per workgroup: 8 x 8
constexpr int BN = 64;
constexpr int PH = 18, PW = 18;             // patch height / width in pixels
constexpr int PWP = 19;                     // padded patch row of the LDS image [py][quad][PWP][4 floats]
constexpr int IT_STRIDE = 4 * 4 * PWP * 16; // patch bytes between transform items (two tile rows = four patch rows)
constexpr int VPIECE = T * 32;              // 2048: one (position, piece) plane of a wave's V: [64 tiles][16 ci bf16]
constexpr int VW_BYTES = 2 * 3 * VPIECE;    // 12288 per wave (two positions)
constexpr int V_BYTES = 8 * VW_BYTES;       // 98304
constexpr int P_SLOTS = 1408;               // 1368 used; 2 x 512 (all waves) + 384 (waves 0-5)
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

// Registers of the input transform of ONE (item: tile, 4 channels) x (position j of the wave's row).  V[i][j] = t[A] +- t[B] with
// (A, B) = columns (0,2), (1,2), (2,1), (1,3) of t = (B^T d)[i] for j = 0..3.  The 34 VALU operations are indexed so that the main
// loop can place them six per MFMA slice (consecutive ones independent):
//   0..7    tA, tB = da + sg * db        8..11   v = tA +- tB
//   12..33  exact three-way bf16 split of v: h = x & 0xFFFF0000, r = x - h, m = r & 0xFFFF0000, l = r - m; v_perm_b32 packs the
//           top halves of (x, r, l) of a channel pair
struct Xf {
    f32x4 da[2], db[2];             // rows ra / rb of the patch at columns A, B
    f32x4 t[2], v;
    float h[4], r[4], l[4];
    unsigned pk[3][2];              // [piece][channel pair]
};
__device__ __forceinline__ void xop(Xf& s, const int j, const int op, const float sg) {
    if (op < 8) {
        const int c = op >> 2, e = op & 3;
        s.t[c][e] = __builtin_fmaf(s.db[c][e], sg, s.da[c][e]);
    } else if (op < 12) {
        const int e = op & 3;
        s.v[e] = j == 1 ? s.t[0][e] + s.t[1][e] : s.t[0][e] - s.t[1][e];
    } else if (op < 34) {
        const int w = op - 12;
        if (w < 4) s.h[w] = __uint_as_float(__float_as_uint(s.v[w]) & 0xFFFF0000u);
        else if (w < 8) s.r[w - 4] = s.v[w - 4] - s.h[w - 4];
        else if (w < 10) s.pk[0][w - 8] = __builtin_amdgcn_perm(__float_as_uint(s.v[2 * (w - 8) + 1]), __float_as_uint(s.v[2 * (w - 8)]), 0x07060302u);
        else if (w < 14) s.h[w - 10] = __uint_as_float(__float_as_uint(s.r[w - 10]) & 0xFFFF0000u);
        else if (w < 18) s.l[w - 14] = s.r[w - 14] - s.h[w - 14];
        else if (w < 20) s.pk[1][w - 18] = __builtin_amdgcn_perm(__float_as_uint(s.r[2 * (w - 18) + 1]), __float_as_uint(s.r[2 * (w - 18)]), 0x07060302u);
        else s.pk[2][w - 20] = __builtin_amdgcn_perm(__float_as_uint(s.l[2 * (w - 20) + 1]), __float_as_uint(s.l[2 * (w - 20)]), 0x07060302u);
    }
}

// the whole kernel for the waves with position pair JP (compile-time: the patch columns and signs of the transform are literals)
template <int JP>
__device__ __forceinline__ void run(const Args& a, char* smem) {
    char* sV = smem;                                  // [8 waves][2 positions][3 pieces][64 tiles][16 ci] bf16
    char* sP = smem + V_BYTES;                        // [2][18 py][4 quads][19 px][4 ci] fp32 (+ slack)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1;                         // transform row of this wave (its position pair is JP = wave & 1)
    const int hi = lane >> 5;
    const int xi0 = wi * 4 + 2 * JP;                  // first of this wave's two positions
    const bool up = a.flags & CNL_UPSAMPLE_IN;
    const unsigned u_piece = (unsigned)(a.CoutP * 32);              // bytes per (chunk, position, piece) plane of U
    const unsigned u_pos = 3u * u_piece;
    const unsigned u_chunk = 16u * u_pos;

    // t[i][*] = d[ra][*] + sg * d[rb][*]:  i = 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
    const int ra = wi == 0 ? 0 : (wi == 2 ? 2 : 1);
    const int rb = wi == 3 ? 3 : (wi == 2 ? 1 : 2);
    const float sg = wi == 1 ? 1.f : -1.f;
    // transform items: lane -> (tile column tx, channel quad q, tile row parity tyl); item it = 0..3 -> tile row 2 it + tyl.  The
    // quad index makes every 16-lane group of a ds_read_b128 cover all 16 bank residues (see winograd3.hip)
    const int t_tx = lane & 7, t_tyl = lane >> 5;
    const int t_q = ((((lane >> 2) ^ (lane >> 3) ^ (lane >> 4)) & 1) << 1) | ((lane >> 3) & 1);
    const int src_a = (((2 * t_tyl + ra) * 4 + t_q) * PWP + 2 * t_tx) * 16;      // + it * IT_STRIDE + column * 16
    const int src_b = (((2 * t_tyl + rb) * 4 + t_q) * PWP + 2 * t_tx) * 16;
    const int dstv = wave * VW_BYTES + (t_tyl * 8 + t_tx) * 32 + (((t_q >> 1) ^ t_tyl) << 4) + (t_q & 1) * 8;   // + it*512 + (pl*3+k)*VPIECE
    const int fragA = wave * VW_BYTES + (lane & 31) * 32 + ((hi ^ ((lane >> 3) & 1)) << 4);                    // + (pl*3+k)*VPIECE + tg*1024
    const float lo = (a.flags & CNL_RELU) ? 0.f : -__builtin_inff();

    int n, y0, x0, n0;
    unsigned p_off[3], u_voff;
#define W4_SETUP(item_)                                                                                          \
    do {                                                                                                         \
        unsigned b_ = cnl::xcd_remap((item_), (unsigned)a.blocks);                                               \
        const int nbi_ = b_ % a.nb; b_ /= a.nb;                                                                  \
        const int bxi_ = b_ % a.bx; b_ /= a.bx;                                                                  \
        const int byi_ = b_ % a.by;                                                                              \
        n = b_ / a.by; y0 = byi_ * 16; x0 = bxi_ * 16; n0 = nbi_ * BN;                                           \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                          \
            const int s_ = i * 512 + tid;              /* 16-byte slot of the patch image: (py*4 + quad)*PWP + px */ \
            const int rowq_ = s_ / PWP, pxx_ = s_ - rowq_ * PWP;                                                 \
            const int py_ = rowq_ >> 2, q_ = rowq_ & 3;                                                          \
            const int iy_ = y0 - 1 + py_, ix_ = x0 - 1 + pxx_;                                                   \
            const bool ok_ = py_ < PH && pxx_ < PW && (unsigned)iy_ < (unsigned)a.H && (unsigned)ix_ < (unsigned)a.W; \
            const int sy_ = up ? (iy_ >> 1) : iy_, sx_ = up ? (ix_ >> 1) : ix_;   /* nearest-2x upsample folded in */ \
            p_off[i] = ok_ ? (unsigned)((((n * a.Hs + sy_) * a.Ws + sx_) * a.ldx + q_ * 4) * 4) : OOB;           \
        }                                                                                                        \
        u_voff = (unsigned)((n0 + (lane & 31)) * 32 + hi * 16);                                                  \
    } while (0)
#define W4_ISSUE_P(cc_)                                                                                          \
    do {                                                                                                         \
        if ((cc_) < a.CC) {                                                                                      \
            char* d_ = sP + ((cc_) & 1) * P_BYTES;                                                               \
            dma16(a.x, a.x_bytes, d_ + (wave * 64) * 16, p_off[0], (unsigned)((cc_) * 64));                      \
            dma16(a.x, a.x_bytes, d_ + (512 + wave * 64) * 16, p_off[1], (unsigned)((cc_) * 64));                \
            if (wave < 6) dma16(a.x, a.x_bytes, d_ + (1024 + wave * 64) * 16, p_off[2], (unsigned)((cc_) * 64)); \
        }                                                                                                        \
    } while (0)
    // B fragments (three pieces) of local position pl_ of chunk cc_, cout group g_: global -> registers
#define W4_LOAD_B(cc_, pl_, g_)                                                                                  \
    do {                                                                                                         \
        if ((cc_) < a.CC) {                                                                                      \
            const unsigned so_ = (unsigned)(cc_) * u_chunk + (unsigned)(xi0 + (pl_)) * u_pos + (unsigned)(g_) * 1024u; \
            _Pragma("unroll") for (int kk_ = 0; kk_ < 3; ++kk_)                                                  \
                fb[g_][kk_] = buf_load16(a.u3, a.u_bytes, u_voff, so_ + (unsigned)kk_ * u_piece);                \
        }                                                                                                        \
    } while (0)
#define W4_READ_A(pl_, buf_, g_, kk_) fa[buf_][g_][kk_] = lds_u4(sV + fragA + ((pl_) * 3 + (kk_)) * VPIECE + (g_) * 1024)
    // MFMA s_ (0..23) of a position: cout group s_ / 12, term (s_ % 12) >> 1 in the order (a3 b1, a1 b3, a2 b2, a2 b1, a1 b2, a1 b1),
    // tile group s_ & 1
#define W4_MFMA(pl_, buf_, s_)                                                                                   \
    do {                                                                                                         \
        const int cg_ = (s_) / 12, term_ = ((s_) % 12) >> 1, tg_ = (s_) & 1;                                     \
        const int ka_ = term_ == 0 ? 2 : (term_ == 2 || term_ == 3) ? 1 : 0;                                     \
        const int kb_ = term_ == 1 ? 2 : (term_ == 2 || term_ == 4) ? 1 : 0;                                     \
        acc[pl_][tg_][cg_] = mfma16(fa[buf_][tg_][ka_], fb[cg_][kb_], acc[pl_][tg_][cg_]);                       \
    } while (0)
    // patch reads of (item it_, position j_): columns A, B of rows ra / rb
#define W4_COLA(j_) ((j_) == 0 ? 0 : (j_) == 2 ? 2 : 1)
#define W4_COLB(j_) ((j_) == 2 ? 1 : (j_) == 3 ? 3 : 2)
#define W4_X_READ(pa_, pb_, j_, it_)                                                                             \
    do {                                                                                                         \
        xf.da[0] = lds_f4((pa_) + (it_) * IT_STRIDE + W4_COLA(j_) * 16);                                         \
        xf.db[0] = lds_f4((pb_) + (it_) * IT_STRIDE + W4_COLA(j_) * 16);                                         \
        xf.da[1] = lds_f4((pa_) + (it_) * IT_STRIDE + W4_COLB(j_) * 16);                                         \
        xf.db[1] = lds_f4((pb_) + (it_) * IT_STRIDE + W4_COLB(j_) * 16);                                         \
    } while (0)
#define W4_X_WRITE(pl_, it_)                                                                                     \
    _Pragma("unroll") for (int kk_ = 0; kk_ < 3; ++kk_)                                                          \
        *reinterpret_cast<u32x2*>(sV + dstv + (it_) * 512 + ((pl_) * 3 + kk_) * VPIECE) = u32x2{xf.pk[kk_][0], xf.pk[kk_][1]};
#define W4_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    // One position slot = 24 MFMAs of local position pl_ (fragment buffer pl_) in 24 slices fenced by sched_barrier(0).
    //   (cN_, plN_)  the next position: its A fragments go into buffer plN_ (slices 6-11), its cout-group-0 B fragments into the
    //                registers that this position's group-0 MFMAs release at mid-slot (slice 12); PN_ = there is a next position
    //   slice 0      cout-group-1 B fragments of THIS position (used from slice 12 on)
    //   JOBS_        produce V[pl_] of chunk cc_ + 1 from the patch at (pa, pb): item it in slices 6 it .. 6 it + 5 (34 operations,
    //                writes in the last slice), the reads of item it + 1 in slice 6 it + 3; the reads of item 0 come with slice 0
    //                (FIRST_: first slot after the barrier) or with slice 21 of the previous slot (RDN_)
#define W4_SLOT(cc_, pl_, cN_, plN_, PN_, JOBS_, FIRST_, RDN_)                                                   \
    do {                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        _Pragma("unroll") for (int k = 0; k < 24; ++k) {                                                         \
            const int it_ = k / 6, ks = k % 6;                                                                   \
            W4_MFMA(pl_, pl_, k);                                                                                \
            if (k == 0) W4_LOAD_B(cc_, pl_, 1);                                                                  \
            if ((PN_) && k == 12) W4_LOAD_B(cN_, plN_, 0);                                                       \
            if ((PN_) && k >= 6 && k < 12) { W4_READ_A(plN_, plN_, (k - 6) / 3, (k - 6) % 3); }                  \
            if (JOBS_) {                                                                                         \
                if ((FIRST_) && k == 0) { W4_X_READ(pa, pb, 2 * JP + (pl_), 0); }                                \
                _Pragma("unroll") for (int o_ = 0; o_ < 6; ++o_) xop(xf, 2 * JP + (pl_), ks * 6 + o_, sg);       \
                if (ks == 5) { W4_X_WRITE(pl_, it_); }                                                           \
                if (ks == 3 && it_ < 3) { W4_X_READ(pa, pb, 2 * JP + (pl_), it_ + 1); }                          \
                if ((RDN_) && k == 21) { W4_X_READ(pa, pb, 2 * JP + 1 - (pl_), 0); }                             \
            }                                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
        }                                                                                                        \
    } while (0)

    unsigned item = blockIdx.x;
    W4_SETUP(item);
    W4_ISSUE_P(0);
    W4_ISSUE_P(1);
    u32x4 fa[2][2][3];       // A fragments: [buffer = local position][tile group][piece]
    u32x4 fb[2][3];          // B fragments: [cout group][piece] (rolling)
    float bias_n[2];
    W4_LOAD_B(0, 0, 0);
    bool first = true;
    while (true) {
        f32x16 acc[2][2][2];     // [local position][tile group][cout group]
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j >> 2][(j >> 1) & 1][j & 1] = mfma_zero();
        Xf xf;
        // bias of this thread's epilogue column (cout group 0 / 1), requested now, used after the chunk loop
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int col = n0 + g * 32 + (tid & 31);
            bias_n[g] = col < a.Cout ? a.bias[col] : 0.f;
        }
        // patches 0 / 1 landed (this wave's parts)?  On later items their DMAs are followed in the VMEM queue by the 3 B loads of
        // chunk 0 and the stores of the last epilogue pass (8): a counted wait lets those stay in flight
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
        first = false;
        W4_BARRIER();
        {   // input transform of chunk 0 for this wave's two positions (the other wave of the SIMD covers the latencies)
            const char* pa = sP + src_a;
            const char* pb = sP + src_b;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    W4_X_READ(pa, pb, 2 * JP + pl, it);
#pragma unroll
                    for (int o = 0; o < 34; ++o) xop(xf, 2 * JP + pl, o, sg);
                    W4_X_WRITE(pl, it);
                }
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) { W4_READ_A(0, 0, q / 3, q % 3); }
        // chunk cn: position 0, then position 1 of this wave; beside them V of chunk cn + 1 (from patch cn + 1)
        for (int cn = 0; cn + 1 < a.CC; ++cn) {
            // patch cn + 1 landed everywhere, patch cn consumed everywhere (its buffer receives patch cn + 2)
            // (cn = 0: the chunk-0 transform above read patch 0)
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");       /* all but the newest 6 B loads: the patch DMA is older */
            W4_BARRIER();
            W4_ISSUE_P(cn + 2);
            const char* pa = sP + ((cn + 1) & 1) * P_BYTES + src_a;
            const char* pb = sP + ((cn + 1) & 1) * P_BYTES + src_b;
            W4_SLOT(cn, 0, cn, 1, 1, 1, 1, 1);
            W4_SLOT(cn, 1, cn + 1, 0, 1, 1, 0, 0);
        }
        {   // last chunk: MFMAs only
            const char* pa = sP; const char* pb = sP;
            const int cl = a.CC - 1;
            W4_SLOT(cl, 0, cl, 1, 1, 0, 0, 0);
            W4_SLOT(cl, 1, cl, 0, 0, 0, 0, 0);
            (void)pa; (void)pb;
        }

        // ---- epilogue: Y = A^T M A.  Stage 1 (this wave's two positions of row i, in registers): partial q_c = sum_j A^T[c][j]
        // M[i][j] over its j; the eight partials per (tile, co) meet through LDS, one (tile group, cout group) per pass
        // ([4 i][2 jp][2 c][32 tiles][32 co] = 64 KB of the V region) ----
        float* sQ = reinterpret_cast<float*>(smem);
        const int co = tid & 31;
        const int en = n, ey0 = y0, ex0 = x0, en0 = n0;        // this item's coordinates (the setup below moves on to the next)
        const bool full = (y0 + 16 <= a.H) && (x0 + 16 <= a.W) && (n0 + BN <= a.Cout);
        const float bv[2] = {bias_n[0], bias_n[1]};
        const unsigned next = item + gridDim.x;
        const bool more = next < (unsigned)a.blocks;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int tg = ps >> 1, g = ps & 1;
            // this pass's output addresses and residual values (requested before the exchange, so their latency is covered)
            unsigned y_voff[2];
            bool ok[2][2][2];
            float rv[2][2][2];
            const int col = en0 + g * 32 + co;
            const bool col_ok = col < a.Cout;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int tl = (tid >> 5) + 16 * it;           // tile inside the 4 x 8 tile group
                const int oy = ey0 + 8 * tg + 2 * (tl >> 3), ox = ex0 + 2 * (tl & 7);
                const unsigned pix = (unsigned)((en * a.H + oy) * a.W + ox);
                y_voff[it] = (pix * (unsigned)a.ldy + (unsigned)col) * 4u;
                const unsigned r_voff = (pix * (unsigned)a.ldr + (unsigned)col) * 4u;
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        ok[it][dy][dx] = full || (col_ok && oy + dy < a.H && ox + dx < a.W);
                        rv[it][dy][dx] = 0.f;
                    }
                if (a.res) {
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx)
                            rv[it][dy][dx] = buf_load(a.res, a.r_bytes, ok[it][dy][dx] ? r_voff : OOB, (unsigned)((dy * a.W + dx) * a.ldr * 4));
                }
            }
            W4_BARRIER();                                      // everyone is done reading V / the patches (ps = 0) or sQ
            if (ps == 3 && more) {                             // patch buffers and fragment registers are idle
                W4_SETUP(next);
                W4_ISSUE_P(0);
                W4_ISSUE_P(1);
                W4_LOAD_B(0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tl = (r & 3) + 8 * (r >> 2) + 4 * hi;
                const float ma = acc[0][tg][g][r], mb = acc[1][tg][g][r];      // positions 2 JP, 2 JP + 1
                // A^T = [1 1 1 0; 0 1 -1 -1]:  JP = 0: q0 = m0 + m1, q1 = m1;  JP = 1: q0 = m2, q1 = -m2 - m3
                sQ[(((wi * 2 + JP) * 2 + 0) * 32 + tl) * 32 + (lane & 31)] = JP == 0 ? ma + mb : ma;
                sQ[(((wi * 2 + JP) * 2 + 1) * 32 + tl) * 32 + (lane & 31)] = JP == 0 ? mb : -ma - mb;
            }
            W4_BARRIER();
            // Stage 2: thread = (tile, co): Y[a][c] = sum_i A^T[a][i] (q[i][0][c] + q[i][1][c]); 2 tiles per thread and pass
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int tl = (tid >> 5) + 16 * it;
                float q[4][2];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        q[i][c] = sQ[(((i * 2 + 0) * 2 + c) * 32 + tl) * 32 + co] + sQ[(((i * 2 + 1) * 2 + c) * 32 + tl) * 32 + co];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float ya = q[0][c] + q[1][c] + q[2][c];
                    const float yb = q[1][c] - q[2][c] - q[3][c];
                    buf_store(fmaxf(ya + bv[g] + rv[it][0][c], lo), a.y, a.y_bytes, ok[it][0][c] ? y_voff[it] : OOB, (unsigned)(c * a.ldy * 4));
                    buf_store(fmaxf(yb + bv[g] + rv[it][1][c], lo), a.y, a.y_bytes, ok[it][1][c] ? y_voff[it] : OOB, (unsigned)((a.W + c) * a.ldy * 4));
                }
            }
        }
        if (!more) break;
        item = next;
    }
#undef W4_SLOT
#undef W4_MFMA
#undef W4_ISSUE_P
#undef W4_LOAD_B
#undef W4_SETUP
}

__global__ __launch_bounds__(512) void winograd4_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1) run<1>(a, smem);
    else run<0>(a, smem);
}

}  // namespace cnl_wino4

size_t cnl_wino3_weight_bytes(int Cin, int Cout);      // winograd3.hip

// Launch (arguments already validated by cnl_conv3x3_winograd_f32); u3 = the pre-split weights.
int cnl_wino4_launch(const cnl_conv_params* p, const void* u3, void* stream) {
    using namespace cnl_wino4;
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
    static bool attr_done = false;
    if (!attr_done) {
        CNL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&winograd4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
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
    hipLaunchKernelGGL(winograd4_kernel, dim3(grid), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
    return cnl::check_launch("winograd4_kernel");
}
