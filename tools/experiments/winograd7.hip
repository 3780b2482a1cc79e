// winograd7.hip — the fp16-split Winograd arithmetic of winograd5.hip (scaled two-way fp16 split, three cross terms, fp32
// accumulation; same weights, scales and maximum hand-over) in the TWO-waves-per-SIMD decomposition of winograd4.hip.
//
// winograd5's single wave per SIMD spends 28 % of its cycles at s_waitcnt / barriers and 13 % issue-stalled, with nothing to cover
// them; its clock (2.06 GHz) leaves power headroom that the bf16 kernels did not have (winograd4: same time at a lower clock).  With
// two pieces per operand the non-accumulator registers of the position-split decomposition fit in 128:
//   workgroup = 8 waves, 8x8 tiles x 64 couts x 16 positions; wave (i, jp) owns positions {2 jp, 2 jp + 1} of transform row i for all
//   64 tiles and both cout groups (2 x 2 x 2 accumulator tiles = 128 registers): per position and 16-channel chunk 4 A fragments
//   (double-buffered) + 4 B fragments (double-buffered per cout group, requested 1.5 position slots ahead) -> 12 MFMAs;
//   V of a position is produced and consumed by the same wave (private, single-buffered, program order), one (item, position) at
//   a time: 4 patch reads, 8 fma + 4 add + 10 split operations, one ds_write2st64_b64; the patch is the only shared data (one
//   barrier per chunk); the eight partial inverse-transform sums of a tile meet through LDS in the epilogue.
#include "cnl_common.h"
#include <cstdlib>

#pragma clang fp contract(off)

namespace cnl_wino7 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

struct Args {
    const float* x;
    const void* u3;                   // pre-split, pre-scaled weights (fp16 pieces): [ci/16][position][piece][CoutP][16 ci]
    const float* xmax;                // per-image max |x| (absmax_kernel, or handed over by the producer)
    const float* su;                  // scale of the weights
    unsigned* ymax;                   // optional: per-image max |y| of this launch's output
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
    int order;                        // work-item order (see W7_SETUP)
};

constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr int T = 64;                       // tiles per workgroup: 8 x 8
constexpr int BN = 64;
constexpr int PH = 18, PW = 18;             // patch height / width in pixels
constexpr int PWP = 19;                     // padded patch row of the LDS image [py][quad][PWP][4 floats]
constexpr int IT_STRIDE = 4 * 4 * PWP * 16; // patch bytes between transform items (two tile rows = four patch rows)
constexpr int VPIECE = T * 32;              // 2048: one (position, piece) plane of a wave's V: [64 tiles][16 ci bf16]
constexpr int NP = 2;                       // pieces per operand
constexpr int VW_BYTES = 2 * NP * VPIECE;   // 8192 per wave (two positions)
constexpr int V_BYTES = 8 * VW_BYTES;       // 65536 (= one epilogue pass)
constexpr int P_SLOTS = 1408;               // 1368 used; 2 x 512 (all waves) + 384 (waves 0-5)
constexpr int P_BYTES = P_SLOTS * 16;       // 22528 per buffer (two buffers)
constexpr int LDS_BYTES = V_BYTES + 2 * P_BYTES;                 // 110592: one workgroup per CU

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
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, voffset, soffset, CNL_NT_STORES);
}
__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// the split of a channel pair (v0, v1), scaled by the power of two S (see winograd5.hip): hi = RN16(v S) packed, r = v S - hi exactly
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
__device__ __forceinline__ f32x16 mfma_zero() {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const u32x4 zz = {0u, 0u, 0u, 0u};
    return mfma16(zz, zz, z);
}
__device__ __forceinline__ f32x4 lds_f4(const char* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ u32x4 lds_u4(const char* p) { return *reinterpret_cast<const u32x4*>(p); }

// Registers of the input transform of ONE (item: tile, 4 channels) x (position j of the wave's row).  V[i][j] = t[A] +- t[B] with
// (A, B) = columns (0,2), (1,2), (2,1), (1,3) of t = (B^T d)[i] for j = 0..3.  The 22 VALU operations are indexed so that the main
// loop can place them per MFMA slice:  0..7 tA, tB = da + sg * db;  8..11 v = tA +- tB;  12..21 split: 2 x mixlo, 2 x mixhi (packed hi
// pairs), 4 residuals, 2 x pkrtz (packed lo pairs)
struct Xf {
    f32x4 da[2], db[2];             // rows ra / rb of the patch at columns A, B
    f32x4 t[2], v;
    float r[4];
    unsigned pk[NP][2];             // [piece][channel pair]
};
constexpr int XOPS = 22;
__device__ __forceinline__ void xop(Xf& s, const int j, const int op, const float sg, const float S) {
    if (op < 8) {
        const int c = op >> 2, e = op & 3;
        s.t[c][e] = __builtin_fmaf(s.db[c][e], sg, s.da[c][e]);
    } else if (op < 12) {
        const int e = op & 3;
        s.v[e] = j == 1 ? s.t[0][e] + s.t[1][e] : s.t[0][e] - s.t[1][e];
    } else if (op < XOPS) {
        const int w = op - 12;
        if (w < 2) s.pk[0][w] = split_hi_lo(s.v[2 * w], S);
        else if (w < 4) s.pk[0][w - 2] = split_hi_hi(s.pk[0][w - 2], s.v[2 * (w - 2) + 1], S);
        else if (w < 8) s.r[w - 4] = ((w - 4) & 1) ? split_res_hi(s.v[w - 4], S, s.pk[0][(w - 4) >> 1]) : split_res_lo(s.v[w - 4], S, s.pk[0][(w - 4) >> 1]);
        else s.pk[1][w - 8] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(s.r[2 * (w - 8)], s.r[2 * (w - 8) + 1]));
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
    const unsigned u_pos = (unsigned)NP * u_piece;
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
    const int dstv = wave * VW_BYTES + (t_tyl * 8 + t_tx) * 32 + (((t_q >> 1) ^ t_tyl) << 4) + (t_q & 1) * 8;   // + it*512 + (pl*NP+k)*VPIECE
    const int fragA = wave * VW_BYTES + (lane & 31) * 32 + ((hi ^ ((lane >> 3) & 1)) << 4);                    // + (pl*NP+k)*VPIECE + tg*1024
    const float lo = (a.flags & CNL_RELU) ? 0.f : -__builtin_inff();
    // per-image power-of-two scale of V (see winograd5.hip): S / inv_n belong to the item set up last, `inv` to the one in the epilogue
    const float Su = a.su[0];
    float S = 1.f, inv_n = 1.f / Su;
    float omax = 0.f;

    int n, y0, x0, n0;
    unsigned p_off[3], u_voff;
#define W7_SETUP(item_)                                                                                          \
    do {                                                                                                         \
        unsigned b_ = cnl::xcd_remap((item_), (unsigned)a.blocks);                                               \
        int nbi_, bxi_, byi_;                                                                                    \
        if (a.order == 0) {          /* cout block fastest */                                                    \
            nbi_ = b_ % a.nb; b_ /= a.nb; bxi_ = b_ % a.bx; b_ /= a.bx; byi_ = b_ % a.by; n = b_ / a.by;         \
        } else {                     /* pairs of cout blocks fastest, then the tile, then the pair index */       \
            const int np_ = (a.nb + 1) / 2;                                                                      \
            const int lo_ = b_ % 2; b_ /= 2; bxi_ = b_ % a.bx; b_ /= a.bx; byi_ = b_ % a.by; b_ /= a.by;         \
            const int pr_ = b_ % np_; n = b_ / np_; nbi_ = pr_ * 2 + lo_;                                        \
        }                                                                                                        \
        y0 = byi_ * 16; x0 = bxi_ * 16; n0 = nbi_ * BN;                                                          \
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
        {                                                                                                        \
            const float mx4_ = 4.f * a.xmax[n * AMS];                                                                  \
            S = 1.f;                                                                                             \
            if (mx4_ > 0.f && mx4_ < __builtin_inff()) {                                                         \
                int e_;                                                                                          \
                (void)__builtin_frexpf(mx4_, &e_);                                                               \
                e_ = 14 - e_;                                                                                    \
                S = __builtin_ldexpf(1.f, e_ < -100 ? -100 : (e_ > 100 ? 100 : e_));                             \
            }                                                                                                    \
            inv_n = 1.f / (S * Su);                                                                              \
        }                                                                                                        \
    } while (0)
#define W7_ISSUE_P(cc_)                                                                                          \
    do {                                                                                                         \
        if ((cc_) < a.CC) {                                                                                      \
            char* d_ = sP + ((cc_) & 1) * P_BYTES;                                                               \
            dma16(a.x, a.x_bytes, d_ + (wave * 64) * 16, p_off[0], (unsigned)((cc_) * 64));                      \
            dma16(a.x, a.x_bytes, d_ + (512 + wave * 64) * 16, p_off[1], (unsigned)((cc_) * 64));                \
            if (wave < 6) dma16(a.x, a.x_bytes, d_ + (1024 + wave * 64) * 16, p_off[2], (unsigned)((cc_) * 64)); \
        }                                                                                                        \
    } while (0)
    // B fragments (three pieces) of local position pl_ of chunk cc_, cout group g_: global -> registers
#define W7_LOAD_B(cc_, pl_, g_)                         /* into buffer pl_ */                                    \
    do {                                                                                                         \
        if ((cc_) < a.CC) {                                                                                      \
            const unsigned so_ = (unsigned)(cc_) * u_chunk + (unsigned)(xi0 + (pl_)) * u_pos + (unsigned)(g_) * 1024u; \
            _Pragma("unroll") for (int kk_ = 0; kk_ < NP; ++kk_)                                                 \
                fb[pl_][g_][kk_] = buf_load16(a.u3, a.u_bytes, u_voff, so_ + (unsigned)kk_ * u_piece);           \
        }                                                                                                        \
    } while (0)
#define W7_READ_A(pl_, buf_, g_, kk_) fa[buf_][g_][kk_] = lds_u4(sV + fragA + ((pl_) * NP + (kk_)) * VPIECE + (g_) * 1024)
    // MFMA s_ (0..11) of a position: cout group s_ / 6, term (s_ % 6) >> 1 in the order (hi lo', lo hi', hi hi'), tile group s_ & 1
#define W7_MFMA(pl_, s_)                                                                                         \
    do {                                                                                                         \
        const int cg_ = (s_) / 6, term_ = ((s_) % 6) >> 1, tg_ = (s_) & 1;                                       \
        const int ka_ = term_ == 1 ? 1 : 0, kb_ = term_ == 0 ? 1 : 0;                                            \
        acc[pl_][tg_][cg_] = mfma16(fa[pl_][tg_][ka_], fb[pl_][cg_][kb_], acc[pl_][tg_][cg_]);                   \
    } while (0)
    // patch reads of (item it_, position j_): columns A, B of rows ra / rb
#define W7_COLA(j_) ((j_) == 0 ? 0 : (j_) == 2 ? 2 : 1)
#define W7_COLB(j_) ((j_) == 2 ? 1 : (j_) == 3 ? 3 : 2)
#define W7_X_READ(pa_, pb_, j_, it_)                                                                             \
    do {                                                                                                         \
        xf.da[0] = lds_f4((pa_) + (it_) * IT_STRIDE + W7_COLA(j_) * 16);                                         \
        xf.db[0] = lds_f4((pb_) + (it_) * IT_STRIDE + W7_COLA(j_) * 16);                                         \
        xf.da[1] = lds_f4((pa_) + (it_) * IT_STRIDE + W7_COLB(j_) * 16);                                         \
        xf.db[1] = lds_f4((pb_) + (it_) * IT_STRIDE + W7_COLB(j_) * 16);                                         \
    } while (0)
#define W7_X_WRITE(pl_, it_)                                                                                     \
    _Pragma("unroll") for (int kk_ = 0; kk_ < NP; ++kk_)                                                         \
        *reinterpret_cast<u32x2*>(sV + dstv + (it_) * 512 + ((pl_) * NP + kk_) * VPIECE) = u32x2{xf.pk[kk_][0], xf.pk[kk_][1]};
#define W7_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    // One position slot = 12 MFMAs of local position pl_ (fragment buffers pl_) in 12 slices fenced by sched_barrier(0).
    //   (cN_, plN_)  the next position (the wave's other one): its A fragments go into buffer plN_ (slices 2-5) and the B fragments of
    //                its cout group 1 too (slice 0: that buffer's group-1 registers were released when the previous slot ended)
    //   (cB_)        the position after next = this local position in chunk cB_: B fragments of its cout group 0 into buffer pl_
    //                at slice 6, when this slot's group-0 MFMAs have been issued
    //   JOBS_        produce V[pl_] of chunk cc_ + 1 from the patch at (pa, pb): item it in slices 3 it .. 3 it + 2 (22 operations,
    //                the write in the last slice), the reads of item it + 1 in slice 3 it + 1; the reads of item 0 come with slice 0
    //                (FIRST_: first slot after the barrier) or with slice 10 of the previous slot (RDN_)
#define W7_SLOT(cc_, pl_, cN_, plN_, PN_, cB_, PB_, JOBS_, FIRST_, RDN_)                                         \
    do {                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        _Pragma("unroll") for (int k = 0; k < 12; ++k) {                                                         \
            const int it_ = k / 3, ks = k % 3;                                                                   \
            W7_MFMA(pl_, k);                                                                                     \
            if ((PN_) && k == 0) W7_LOAD_B(cN_, plN_, 1);                                                        \
            if ((PB_) && k == 6) W7_LOAD_B(cB_, pl_, 0);                                                         \
            if ((PN_) && k >= 2 && k < 6) { W7_READ_A(plN_, plN_, (k - 2) >> 1, (k - 2) & 1); }                  \
            if (JOBS_) {                                                                                         \
                if ((FIRST_) && k == 0) { W7_X_READ(pa, pb, 2 * JP + (pl_), 0); }                                \
                _Pragma("unroll") for (int o_ = 0; o_ < 8; ++o_) xop(xf, 2 * JP + (pl_), ks * 8 + o_, sg, S);    \
                if (ks == 2) { W7_X_WRITE(pl_, it_); }                                                           \
                if (ks == 1 && it_ < 3) { W7_X_READ(pa, pb, 2 * JP + (pl_), it_ + 1); }                          \
                if ((RDN_) && k == 10) { W7_X_READ(pa, pb, 2 * JP + 1 - (pl_), 0); }                             \
            }                                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
        }                                                                                                        \
    } while (0)

    unsigned item = blockIdx.x;
    W7_SETUP(item);
    W7_ISSUE_P(0);
    W7_ISSUE_P(1);
    u32x4 fa[2][2][NP];      // A fragments: [buffer = local position][tile group][piece]
    u32x4 fb[2][2][NP];      // B fragments: [buffer = local position][cout group][piece]
    float bias_n[2];
#define W7_LOAD_B_ITEM() do { W7_LOAD_B(0, 0, 0); W7_LOAD_B(0, 0, 1); W7_LOAD_B(0, 1, 0); } while (0)
    W7_LOAD_B_ITEM();
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
        // patches 0 / 1 landed (this wave's parts)?  On later items their DMAs are followed in the VMEM queue by the 6 B loads of
        // chunk 0, the 2 bias loads above and the stores of the last epilogue pass (8): a counted wait lets those stay in flight
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        first = false;
        W7_BARRIER();
        {   // input transform of chunk 0 for this wave's two positions (the other wave of the SIMD covers the latencies)
            const char* pa = sP + src_a;
            const char* pb = sP + src_b;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    W7_X_READ(pa, pb, 2 * JP + pl, it);
#pragma unroll
                    for (int o = 0; o < XOPS; ++o) xop(xf, 2 * JP + pl, o, sg, S);
                    W7_X_WRITE(pl, it);
                }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { W7_READ_A(0, 0, q >> 1, q & 1); }
        // chunk cn: position 0, then position 1 of this wave; beside them V of chunk cn + 1 (from patch cn + 1)
        for (int cn = 0; cn + 1 < a.CC; ++cn) {
            // patch cn + 1 landed everywhere, patch cn consumed everywhere (its buffer receives patch cn + 2)
            // (cn = 0: the chunk-0 transform above read patch 0)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       /* all but the newest 4 B loads: the patch DMA is older */
            W7_BARRIER();
            W7_ISSUE_P(cn + 2);
            const char* pa = sP + ((cn + 1) & 1) * P_BYTES + src_a;
            const char* pb = sP + ((cn + 1) & 1) * P_BYTES + src_b;
            W7_SLOT(cn, 0, cn, 1, 1, cn + 1, 1, 1, 1, 1);
            W7_SLOT(cn, 1, cn + 1, 0, 1, cn + 1, 1, 1, 0, 0);
        }
        {   // last chunk: MFMAs only
            const char* pa = sP; const char* pb = sP;
            const int cl = a.CC - 1;
            W7_SLOT(cl, 0, cl, 1, 1, cl, 0, 0, 0, 0);
            W7_SLOT(cl, 1, cl, 0, 0, cl, 0, 0, 0, 0);
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
        const float inv = inv_n;
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
            W7_BARRIER();                                      // everyone is done reading V / the patches (ps = 0) or sQ
            if (ps == 3 && more) {                             // patch buffers and fragment registers are idle
                W7_SETUP(next);
                W7_ISSUE_P(0);
                W7_ISSUE_P(1);
                W7_LOAD_B_ITEM();
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tl = (r & 3) + 8 * (r >> 2) + 4 * hi;
                const float ma = acc[0][tg][g][r], mb = acc[1][tg][g][r];      // positions 2 JP, 2 JP + 1
                // A^T = [1 1 1 0; 0 1 -1 -1]:  JP = 0: q0 = m0 + m1, q1 = m1;  JP = 1: q0 = m2, q1 = -m2 - m3
                sQ[(((wi * 2 + JP) * 2 + 0) * 32 + tl) * 32 + (lane & 31)] = JP == 0 ? ma + mb : ma;
                sQ[(((wi * 2 + JP) * 2 + 1) * 32 + tl) * 32 + (lane & 31)] = JP == 0 ? mb : -ma - mb;
            }
            W7_BARRIER();
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
                    const float ya = (q[0][c] + q[1][c] + q[2][c]) * inv;
                    const float yb = (q[1][c] - q[2][c] - q[3][c]) * inv;
                    const float oa = fmaxf(ya + bv[g] + rv[it][0][c], lo), ob = fmaxf(yb + bv[g] + rv[it][1][c], lo);
                    omax = fmaxf(omax, fmaxf(ok[it][0][c] ? fabsf(oa) : 0.f, ok[it][1][c] ? fabsf(ob) : 0.f));
                    buf_store(oa, a.y, a.y_bytes, ok[it][0][c] ? y_voff[it] : OOB, (unsigned)(c * a.ldy * 4));
                    buf_store(ob, a.y, a.y_bytes, ok[it][1][c] ? y_voff[it] : OOB, (unsigned)((a.W + c) * a.ldy * 4));
                }
            }
        }
        if (a.ymax) {          // max |y| of this item into its image's slot: one atomic per wave and item
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) omax = fmaxf(omax, __shfl_xor(omax, o, 64));
            if (lane == 0) cnl::report_max(a.ymax + en * AMS, omax);
            omax = 0.f;
        }
        if (!more) break;
        item = next;
    }
#undef W7_SLOT
#undef W7_MFMA
#undef W7_ISSUE_P
#undef W7_LOAD_B
#undef W7_SETUP
}

__global__ __launch_bounds__(512) void winograd7_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1) run<1>(a, smem);
    else run<0>(a, smem);
}

}  // namespace cnl_wino7

size_t cnl_wino5_weight_bytes(int Cin, int Cout);      // winograd5.hip: the weight layout, scales and scalars are shared

namespace cnl_wino7 {
// per-image max |x| (see winograd5.hip): blockIdx.y = image
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long pixels, int C, int ld, unsigned* __restrict__ out) {
    const int c4 = C >> 2;
    const long total = pixels * c4;
    x += (long)blockIdx.y * pixels * ld;
    out += blockIdx.y;
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
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        if (m > 0.f) atomicMax(out, __float_as_uint(m));
    }
}
}  // namespace cnl_wino7

// Launch (arguments already validated by cnl_conv3x3_winograd_f32); u5 = the fp16-split weights, scal = the layer's scalars.
int cnl_wino7_launch(const cnl_conv_params* p, const void* u5, float* scal, void* stream) {
    using namespace cnl_wino7;
    Args a;
    CNL_REQUIRE(p->x_absmax || p->N <= 4096, CNL_E_UNSUPPORTED, "cnl_conv3x3_winograd_f32: more than 4096 images per launch need x_absmax");
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
    static bool attr_done = false;
    if (!attr_done) {
        CNL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&winograd7_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_done = true;
    }
    static int n_cu = 0;         // persistent workgroups: one per CU, walking the work items with stride gridDim.x
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        CNL_HIP(hipGetDevice(&dev));
        CNL_HIP(hipGetDeviceProperties(&prop, dev));
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    if (!p->x_absmax) {          // the per-image scale: one pass over the input (stream-ordered before the convolution)
        CNL_HIP(hipMemsetAsync(scal + 16, 0, sizeof(float) * (size_t)p->N, (hipStream_t)stream));
        const long long vec4 = (long long)p->H_in * p->W_in * (p->Cin / 4);
        const long long want = (vec4 + 256 * 16 - 1) / (256 * 16);
        const unsigned mgrid = (unsigned)(want < 1 ? 1 : (want > 64 ? 64 : want));
        hipLaunchKernelGGL(absmax_kernel, dim3(mgrid, (unsigned)p->N), dim3(256), 0, (hipStream_t)stream, p->x, (long)p->H_in * p->W_in,
                           p->Cin, p->ldx, reinterpret_cast<unsigned*>(scal + 16));
    }
    const unsigned grid = (unsigned)(blocks < (long long)n_cu ? blocks : (long long)n_cu);
    hipLaunchKernelGGL(winograd7_kernel, dim3(grid), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
    return cnl::check_launch("winograd7_kernel");
}
