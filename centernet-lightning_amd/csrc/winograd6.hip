// winograd6.hip — the fp16-split Winograd kernel of winograd5.hip (same arithmetic, same weight layout, same scales) on work
// items of 8x16 output pixels x 128 output channels instead of 16x16 x 64: the accumulators (4 positions x 4 cout groups per wave,
// 256 registers) cover twice the output channels per transformed input tile, so the input transform + split — what winograd5's
// single wave per SIMD spends most of its issue slots on — and the V / patch traffic through LDS are halved per output, at the
// price of streaming the weights twice as often (4 B fragments per position and cout group pair instead of 2).  A timing
// emulation inside winograd5 (half the transform jobs, every weight load doubled) predicted -6 % on the 256 -> 256 head blocks.
//   workgroup = 4 waves (one per SIMD), 4x8 tiles x 128 couts x 16 positions; wave i owns transform row i: per position and
//   16-channel chunk 2 A fragments + 8 B fragments -> 12 MFMAs (cout-group pairs {0,1} then {2,3}: the pair's B registers are
//   refilled for the position after next as soon as its six MFMAs are issued); ONE pass-item (tile, 4 channels, position pair) of
//   transform work per position slot; V wave-private and single-buffered as in winograd5; LDS 88 KB.
// Used where Cout is a multiple of 128 (cnl_conv3x3_winograd_f32's dispatch); everything else as winograd5.hip.
#include "cnl_common.h"

#pragma clang fp contract(off)

#ifndef W6_NT_X
#define W6_NT_X 0     /* cache policy (aux) of the patch DMA loads: 2 = nt */
#endif
#ifndef W6_NT_Y
#define W6_NT_Y 2     /* cache policy (aux) of the output stores: nt — a layer's output is far larger than the L2 and would only evict the input patches and weights that ARE re-read (fused first head blocks -9 %) */
#endif
#ifndef W6_EXP
#define W6_EXP 0     /* timing experiments (wrong results): 1 no B loads, 2 no A reads, 3 no patch reads, 4 no barrier in the loop */
#endif
namespace cnl_wino6 {

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
    int order;                        // work-item order (see W6_SETUP)
    int SW;                           // STACK: W + 2 = image pitch of the virtual row of images
    unsigned mg_sw, sh_sw;            // STACK: magic division by SW (n < 2^31)
};

constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr int T = 32;                       // tiles per workgroup: 4 x 8 (8 x 16 output pixels)
constexpr int BN = 128;                     // output channels per workgroup: four 32-cout groups
constexpr int PH = 10, PW = 18;             // patch height / width in pixels
constexpr int PWP = 19;                     // padded patch row of the LDS image [py][quad][PWP][4 floats]
constexpr int IT_STRIDE = 4 * 4 * PWP * 16; // patch bytes between transform items (two tile rows = four patch rows)
constexpr int VPIECE = T * 32;              // 1024: one (position, piece) plane of a wave's V: [32 tiles][16 ci fp16]
constexpr int NP = 2;                       // pieces per operand
constexpr int VW_BYTES = 4 * NP * VPIECE;   // 8192 per wave
constexpr int V_BYTES = 65536;              // V (4 x 8 KB) at its start; the whole region is one epilogue pass
constexpr int P_SLOTS = 768;                // 760 used; 3 x 256
constexpr int P_BYTES = P_SLOTS * 16;       // 12288 per buffer (two buffers)
constexpr int OFF_BYTES = 3 * 256 * 4;          // per-thread patch offsets of the current item (see W6_SETUP)
constexpr int LDS_BYTES = V_BYTES + 2 * P_BYTES + OFF_BYTES;     // 93184

__device__ __forceinline__ void dma16(const float* base, unsigned bytes, char* lds_dst, unsigned voffset, unsigned soffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_dst, 16, voffset, soffset, 0, W6_NT_X);
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
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, voffset, soffset, W6_NT_Y);
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
    f32x4 th[2][2];                 // t of patch columns 1, 2 of each item, kept from pass 0 for pass 1
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

// n / d for n < 2^31 with host-computed (magic, shift); shift == 0xFF encodes d == 1
__device__ __forceinline__ unsigned fast_div6(unsigned n, unsigned magic, unsigned shift) { return shift == 0xFFu ? n : (__umulhi(n, magic) >> shift); }
__device__ __forceinline__ float pow2_scale_v(float mx4) {        // the power of two S with mx4 S in [2^13, 2^14); 1 for 0 / Inf / NaN
    float S = 1.f;
    if (mx4 > 0.f && mx4 < __builtin_inff()) {
        int e;
        (void)__builtin_frexpf(mx4, &e);
        e = 14 - e;
        S = __builtin_ldexpf(1.f, e < -100 ? -100 : (e > 100 ? 100 : e));
    }
    return S;
}

// STACK: the N images of the launch lie SIDE BY SIDE in one virtual row, W + 2 columns apart (two zero columns between neighbours — the
// zero padding both of them need), and the 16-pixel blocks tile that row: a 34-wide map then costs 36 columns per image instead of 48.
// A work item may span two images, so the image (and with it the per-image scale, the addresses and the max |y| slot) is a property of
// the tile column / patch slot instead of the item; every tile is still computed from its own 4x4 patch in the same order, so the
// results are bit-identical to the plain layout's.  Tiles inside a gap read both neighbours and are never stored.
template <bool STACK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void winograd6_kernel(const Args a) {
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
    const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane((a.flags & CNL_RELU) ? 0 : (int)0xff800000u));      // ReLU floor or -inf (scalar register)
    // power-of-two scale of V, PER IMAGE (an image's result never depends on its batch neighbours — batch invariance, shard == full
    // batch): |V| <= 4 max |x| (sums of four inputs), 4 max |x| S in [2^13, 2^14) — exact, undone in the epilogue together with the
    // scale of the weights.  S / inv_n belong to the item set up last (W6_SETUP), `inv` to the one in the epilogue.
    const float Su = a.su[0];
    float S = 1.f, inv_n = 1.f / Su;
    float omax = 0.f;          // running max |y| of this lane's stores of the current item

    int n, y0, x0, n0;
    int ne_n = 0;              // STACK: image of this thread's epilogue tile column, for the item set up last
    unsigned u_voff;
    // the three patch-piece offsets of a thread live in LDS, not in registers: they are used once per chunk, the register allocator spilled
    // them to scratch, and every reload inside the chunk loop came with a vmcnt(0) — a wait for all the loads in flight
    unsigned* sOff = reinterpret_cast<unsigned*>(smem + V_BYTES + 2 * P_BYTES) + tid;
#define W6_SETUP(item_)                                                                                          \
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
        y0 = byi_ * 8; x0 = bxi_ * 16; n0 = nbi_ * BN;                                                          \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                          \
            const int s_ = i * 256 + tid;              /* 16-byte slot of the patch image: (py*4 + quad)*PWP + px */ \
            const int rowq_ = s_ / PWP, pxx_ = s_ - rowq_ * PWP;                                                 \
            const int py_ = rowq_ >> 2, q_ = rowq_ & 3;                                                          \
            const int iy_ = y0 - 1 + py_;                                                                        \
            int ix_ = x0 - 1 + pxx_, ni_ = n;                                                                    \
            bool okx_ = (unsigned)ix_ < (unsigned)a.W;                                                           \
            if constexpr (STACK) {                  /* virtual column -> (image, column); gap columns read zeros */ \
                ni_ = ix_ >= 0 ? (int)fast_div6((unsigned)ix_, a.mg_sw, a.sh_sw) : 0;                            \
                const int vx_ = ix_;                                                                             \
                ix_ = vx_ - ni_ * a.SW;                                                                          \
                okx_ = vx_ >= 0 && ni_ < a.N && ix_ < a.W;                                                       \
            }                                                                                                    \
            const bool ok_ = py_ < PH && pxx_ < PW && (unsigned)iy_ < (unsigned)a.H && okx_;                     \
            const int sy_ = up ? (iy_ >> 1) : iy_, sx_ = up ? (ix_ >> 1) : ix_;   /* nearest-2x upsample folded in */ \
            sOff[i * 256] = ok_ ? (unsigned)((((ni_ * a.Hs + sy_) * a.Ws + sx_) * a.ldx + q_ * 4) * 4) : OOB;    \
        }                                                                                                        \
        /* this lane's B fragments: cout row n0 + (lane & 31) (+ 32 for the second group), channel half hi */    \
        u_voff = (unsigned)((n0 + (lane & 31)) * 32 + hi * 16);                                                  \
        if constexpr (STACK) {                                                                                   \
            /* the transform lane's tile column and the epilogue thread's tile column each have their own image */ \
            const int nt_ = min((int)fast_div6((unsigned)(x0 + 2 * t_tx), a.mg_sw, a.sh_sw), a.N - 1);           \
            S = pow2_scale_v(4.f * a.xmax[nt_ * AMS]);                                                                 \
            ne_n = (int)fast_div6((unsigned)(x0 + 2 * ((tid >> 5) & 7)), a.mg_sw, a.sh_sw);                      \
            inv_n = 1.f / (pow2_scale_v(4.f * a.xmax[min(ne_n, a.N - 1) * AMS]) * Su);                                 \
        } else {                                                                                                 \
            S = pow2_scale_v(4.f * a.xmax[n * AMS]);                                                                   \
            inv_n = 1.f / (S * Su);                                                                              \
        }                                                                                                        \
    } while (0)
    // the channel-chunk offset rides in the SCALAR offset (the bounds check looks at the vector offset alone, so halo lanes still
    // read zeros); a chunk past the end is not fetched
#define W6_ISSUE_P(cc_)                                                                                          \
    do {                                                                                                         \
        if ((cc_) < a.CC) {                                                                                      \
            char* d_ = sP + ((cc_) & 1) * P_BYTES;                                                               \
            _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                        \
                dma16(a.x, a.x_bytes, d_ + (i * 256 + wave * 64) * 16, sOff[i * 256], (unsigned)((cc_) * 64));   \
        }                                                                                                        \
    } while (0)
    // B fragments of position xi0 + j_ of chunk cc_, cout group g_ (three pieces): global -> registers
#define W6_LOAD_B(cc_, j_, buf_, g_)                                                                             \
    do {                                                                                                         \
        if ((cc_) < a.CC) {                                                                                      \
            const unsigned so_ = (unsigned)(cc_) * u_chunk + (unsigned)(xi0 + (j_)) * u_pos + (unsigned)(g_) * 1024u; \
            _Pragma("unroll") for (int kk_ = 0; kk_ < NP; ++kk_)                                                 \
                fb[buf_][g_][kk_] = buf_load16(a.u3, a.u_bytes, u_voff, so_ + (unsigned)kk_ * u_piece);          \
        }                                                                                                        \
    } while (0)
#define W6_READ_A(j_, buf_)                                                                                      \
    _Pragma("unroll") for (int kk_ = 0; kk_ < NP; ++kk_) fa[buf_][kk_] = lds_u4(sV + fragA + ((j_) * NP + kk_) * VPIECE)
    // MFMA s_ (0..11) of a position: cout-group pair s_ / 6 — pair {0,1} first, so that its B registers are free (and refilled
    // for the position after next) from mid-slot on —, then term (s_ % 6) >> 1 in the order (hi lo', lo hi', hi hi'), group s_ & 1
#define W6_MFMA(j_, buf_, s_)                                                                                    \
    do {                                                                                                         \
        const int cg_ = ((s_) / 6) * 2 + ((s_) & 1), term_ = ((s_) % 6) >> 1;                                    \
        const int ka_ = term_ == 1 ? 1 : 0, kb_ = term_ == 0 ? 1 : 0;                                            \
        acc[j_][cg_] = mfma16(fa[buf_][ka_], fb[buf_][cg_][kb_], acc[j_][cg_]);                                  \
    } while (0)
    // patch reads of pass-item (P_, it_) into register set set_; pa_ / pb_ = patch buffer + src_a / src_b
#define W6_X_READ1(set_, pa_, pb_, P_, it_, c_, row_)                                                            \
    do {                                                                                                         \
        if (1 && (P_) == 1 && (c_) != 2) break;                                                          \
        if ((row_) == 0) xf.da[set_][c_] = lds_f4((pa_) + (it_) * IT_STRIDE + ((P_) + (c_)) * 16);               \
        else xf.db[set_][c_] = lds_f4((pb_) + (it_) * IT_STRIDE + ((P_) + (c_)) * 16);                           \
    } while (0)
#define W6_X_READ(set_, pa_, pb_, P_, it_)                                                                       \
    _Pragma("unroll") for (int c_ = 0; c_ < 3; ++c_) {                                                           \
        W6_X_READ1(set_, pa_, pb_, P_, it_, c_, 0);                                                              \
        W6_X_READ1(set_, pa_, pb_, P_, it_, c_, 1);                                                              \
    }
#define W6_X_WRITE(P_, it_)                                                                                      \
    _Pragma("unroll") for (int jj_ = 0; jj_ < 2; ++jj_)                                                          \
        _Pragma("unroll") for (int kk_ = 0; kk_ < NP; ++kk_)                                                     \
            *reinterpret_cast<u32x2*>(sV + dstv + (it_) * 512 + ((2 * (P_) + jj_) * NP + kk_) * VPIECE) =        \
                u32x2{xf.pk[jj_][kk_][0], xf.pk[jj_][kk_][1]};
    // workgroup barrier WITHOUT the vmcnt(0) that __syncthreads() adds when LDS-DMA is in flight (own LDS accesses drained)
#define W6_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    // One position slot = 12 MFMAs in 12 slices fenced by sched_barrier(0).
    //   j_ / buf_    position multiplied in this slot and its fragment buffer
    //   (cA_, jA_)   the next position: its A fragments and the B fragments of its cout groups 2, 3 are fetched into buf_ ^ 1 (PA_)
    //   (cB_, jB_)   the position after next: B fragments of its cout groups 0, 1 go into buf_ from slice 6 on (PB_)
    //   JOBS_        transform ONE pass-item (P_, it_) [register set set_]: 40 operations in slices 0-9, writes in slice 10
    //   RDN_         in slices 6-11 read the patch of the NEXT slot's pass-item (nP_, nIt_) from (npa_, npb_) into set set_ ^ 1
    //   MID_         (slot of position 0) before slice 6: this wave's DMAs of the next patch landed, barrier; slice 8: DMA of
    //                the patch after that (chunk dC_)
#define W6_SLOT(j_, buf_, cA_, jA_, PA_, cB_, jB_, PB_, JOBS_, P_, it_, set_, RDN_, nP_, nIt_, npa_, npb_, MID_, dC_) \
    do {                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        _Pragma("unroll") for (int k = 0; k < 12; ++k) {                                                         \
            if ((MID_) && k == 6) {                                                                              \
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   /* all but the newest 8 B loads: the patch DMA is older */ \
                W6_BARRIER();                                                                                    \
                __builtin_amdgcn_sched_barrier(0);                                                               \
            }                                                                                                    \
            W6_MFMA(j_, buf_, k);                                                                                \
            if (PA_) {                                                                                           \
                if (k < 2) W6_LOAD_B(cA_, jA_, (buf_) ^ 1, 2 + k);                                               \
                if (k >= 2 && k < 4) fa[(buf_) ^ 1][k - 2] = lds_u4(sV + fragA + ((jA_) * NP + (k - 2)) * VPIECE); \
            }                                                                                                    \
            if ((PB_) && k >= 6 && k < 8) W6_LOAD_B(cB_, jB_, buf_, k - 6);                                      \
            if (JOBS_) {                                                                                         \
                if (k < 10) { _Pragma("unroll") for (int o_ = 0; o_ < 4; ++o_) xop(xf, set_, P_, k * 4 + o_, sg, S, it_); } \
                if (k == 10) { W6_X_WRITE(P_, it_); }                                                            \
            }                                                                                                    \
            if ((RDN_) && k >= 6) W6_X_READ1((set_) ^ 1, npa_, npb_, nP_, nIt_, (k - 6) >> 1, k & 1);            \
            if ((MID_) && k == 8) W6_ISSUE_P(dC_);                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
        }                                                                                                        \
    } while (0)

    unsigned item = blockIdx.x;
    W6_SETUP(item);
    W6_ISSUE_P(0);
    W6_ISSUE_P(1);
    u32x4 fa[2][NP];         // A fragments: [buffer][piece]
    u32x4 fb[2][4][NP];      // B fragments: [buffer][cout group][piece]
#define W6_LOAD_B_ITEM() do { W6_LOAD_B(0, 0, 0, 0); W6_LOAD_B(0, 0, 0, 1); W6_LOAD_B(0, 0, 0, 2); W6_LOAD_B(0, 0, 0, 3); W6_LOAD_B(0, 1, 1, 0); W6_LOAD_B(0, 1, 1, 1); } while (0)
    W6_LOAD_B_ITEM();
    bool first = true;
    while (true) {
        f32x16 acc[4][4];        // [position j of row `wave`][cout group]
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[j][g] = mfma_zero();
        Xf xf;

        // patches 0 / 1 landed (this wave's parts)?  Their DMAs are followed in this wave's VMEM queue by the 12 B loads of chunk 0
        // and the 32 stores of the previous item's second epilogue pass (its residual loads are older): a counted wait lets
        // those stay in flight
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(44)" ::: "memory");
        first = false;
        W6_BARRIER();                                         // ... and everybody's
        {   // input transform of chunk 0, all four positions (not overlapped with MFMAs): four pass-items, each read one ahead
            const char* pa = sP + src_a;
            const char* pb = sP + src_b;
            W6_X_READ(0, pa, pb, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int it = g >> 1, P = g & 1, set = g & 1;
                if (g < 3) { W6_X_READ(set ^ 1, pa, pb, (g + 1) & 1, (g + 1) >> 1); }
#pragma unroll
                for (int o = 0; o < XOPS; ++o) xop(xf, set, P, o, sg, S, it);
                W6_X_WRITE(P, it);
            }
        }
        W6_READ_A(0, 0);
        {   // position 0 of chunk 0: no transform work yet; patch 1 is read from slice 6 on, patch 2 requested
            const char* npa = sP + P_BYTES + src_a;
            const char* npb = sP + P_BYTES + src_b;
            const char* pa = npa; const char* pb = npb;       // (unused: JOBS_ = 0)
            W6_SLOT(0, 0, 0, 1, 1, 0, 2, 1, 0, 0, 0, 1, 1, 0, 0, npa, npb, 1, 2);
            (void)pa; (void)pb;
        }
        // chunk n: positions 1..3 of chunk n-1, then position 0 of chunk n; beside them the transform of chunk n
        for (int cn = 1; cn < a.CC; ++cn) {
            const char* pa = sP + (cn & 1) * P_BYTES + src_a;
            const char* pb = sP + (cn & 1) * P_BYTES + src_b;
            const char* npa = sP + ((cn + 1) & 1) * P_BYTES + src_a;
            const char* npb = sP + ((cn + 1) & 1) * P_BYTES + src_b;
            W6_SLOT(1, 1, cn - 1, 2, 1, cn - 1, 3, 1, 1, 0, 0, 0, 1, 0, 1, pa, pb, 0, 0);      // positions {0,1} of chunk cn, item 0
            W6_SLOT(2, 0, cn - 1, 3, 1, cn, 0, 1, 1, 0, 1, 1, 1, 1, 0, pa, pb, 0, 0);          //                          item 1
            W6_SLOT(3, 1, cn, 0, 1, cn, 1, 1, 1, 1, 0, 0, 1, 1, 1, pa, pb, 0, 0);              // positions {2,3} of chunk cn, item 0
            W6_SLOT(0, 0, cn, 1, 1, cn, 2, 1, 1, 1, 1, 1, 1, 0, 0, npa, npb, 1, cn + 2);       //                          item 1
        }
        {   // positions 1..3 of the last chunk: MFMAs only
            const char* pa = sP; const char* pb = sP;
            const int cl = a.CC - 1;
            W6_SLOT(1, 1, cl, 2, 1, cl, 3, 1, 0, 0, 0, 0, 0, 0, 0, pa, pb, 0, 0);
            W6_SLOT(2, 0, cl, 3, 1, cl, 0, 0, 0, 0, 0, 0, 0, 0, 0, pa, pb, 0, 0);
            W6_SLOT(3, 1, cl, 0, 0, cl, 0, 0, 0, 0, 0, 0, 0, 0, 0, pa, pb, 0, 0);
            (void)pa; (void)pb;
        }
        // ---- epilogue: Y = A^T M A.  Stage 1 (this wave's row of positions, in registers): q_c = sum_j A^T[c][j] M[i][j]; the
        // four rows meet through LDS, one PAIR of cout groups per pass ([4 i][2 c][2 cout groups][32 tiles][32 co] = the 64 KB region) ----
        float* sQ = reinterpret_cast<float*>(smem);
        const int co = tid & 31;
        const int en = STACK ? ne_n : n, ey0 = y0, en0 = n0;   // this item's coordinates (the setup below moves on to the next)
        const int ex0 = STACK ? x0 - ne_n * a.SW : x0;         // STACK: this thread's tile column in its own image's coordinates
        const bool full = !STACK && (y0 + 8 <= a.H) && (x0 + 16 <= a.W) && (n0 + BN <= a.Cout);
        const bool img_ok = !STACK || ne_n < a.N;
        // bias of this thread's epilogue columns: requested here (carried from the item's set-up they were four registers live across the
        // whole chunk loop — the loop's patch offsets spilled instead, each reload a vmcnt(0) inside the loop)
        float bv[4];
#pragma unroll
        for (int g_ = 0; g_ < 4; ++g_) {
            const int col_ = n0 + g_ * 32 + (tid & 31);
            bv[g_] = buf_load(a.bias, (unsigned)a.Cout * 4u, col_ < a.Cout ? (unsigned)col_ * 4u : OOB, 0);
        }
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
                const int col = en0 + tg * 64 + g * 32 + co;      /* tg: here the PAIR of cout groups of this pass */
                const bool col_ok = col < a.Cout;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int tl = (tid >> 5) + 8 * it;            // tile inside the 4 x 8 tile group
                    const int oy = ey0 + 2 * (tl >> 3), ox = ex0 + 2 * (tl & 7);
                    const unsigned pix = (unsigned)((en * a.H + oy) * a.W + ox);
                    y_voff[g][it] = (pix * (unsigned)a.ldy + (unsigned)col) * 4u;
                    const unsigned r_voff = (pix * (unsigned)a.ldr + (unsigned)col) * 4u;
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx) {
                            ok[g][it][dy][dx] = full || (img_ok && col_ok && oy + dy < a.H && ox + dx < a.W);
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
            W6_BARRIER();                                      // everyone is done reading V / the patches (tg = 0) or sQ
            if (tg == 1 && more) {                             // patch buffers and fragment registers are idle
                W6_SETUP(next);
                W6_ISSUE_P(0);
                W6_ISSUE_P(1);
                W6_LOAD_B_ITEM();
            }
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int tl = (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const float m0 = acc[0][tg * 2 + g][r], m1 = acc[1][tg * 2 + g][r], m2 = acc[2][tg * 2 + g][r], m3 = acc[3][tg * 2 + g][r];
                    sQ[(((wave * 2 + 0) * 2 + g) * 32 + tl) * 32 + (lane & 31)] = m0 + m1 + m2;
                    sQ[(((wave * 2 + 1) * 2 + g) * 32 + tl) * 32 + (lane & 31)] = m1 - m2 - m3;
                }
            W6_BARRIER();
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
                        const float oa = fmaxf(ya + bv[tg * 2 + g] + rv[g][it][0][c], lo), ob = fmaxf(yb + bv[tg * 2 + g] + rv[g][it][1][c], lo);
                        omax = fmaxf(omax, fmaxf(ok[g][it][0][c] ? fabsf(oa) : 0.f, ok[g][it][1][c] ? fabsf(ob) : 0.f));
                        buf_store(oa, a.y, a.y_bytes, ok[g][it][0][c] ? y_voff[g][it] : OOB, (unsigned)(c * a.ldy * 4));
                        buf_store(ob, a.y, a.y_bytes, ok[g][it][1][c] ? y_voff[g][it] : OOB, (unsigned)((a.W + c) * a.ldy * 4));
                    }
                }
            }
        }
        if (a.ymax) {          // max |y| of this item into its image's slot: one atomic per wave and item (no return value awaited)
            if constexpr (STACK) {     // a half wave = one tile column = one image
                float hm[2];
                cnl::half_max_nonneg(omax, hm);
                omax = lane < 32 ? hm[0] : hm[1];
                if ((lane & 31) == 0 && img_ok) cnl::report_max(a.ymax + en * AMS, omax);
            } else {
                omax = cnl::wave_max_nonneg(omax);
                if (lane == 0) cnl::report_max(a.ymax + en * AMS, omax);
            }
            omax = 0.f;
        }
        if (!more) break;
        item = next;
    }
#undef W6_SLOT
#undef W6_MFMA
#undef W6_ISSUE_P
#undef W6_LOAD_B
#undef W6_SETUP
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

}  // namespace cnl_wino6

size_t cnl_wino5_weight_bytes(int Cin, int Cout);      // winograd5.hip: the weight layout, scales and scalars are shared
int cnl_wino5_own_absmax(const cnl_conv_params* p, float* scal, void* stream);

// Launch (arguments already validated by cnl_conv3x3_winograd_f32); u5 = the fp16-split weights, scal = the layer's scalars.
int cnl_wino6_launch(const cnl_conv_params* p, const void* u5, float* scal, void* stream) {
    using namespace cnl_wino6;
    Args a;
    CNL_REQUIRE(p->x_absmax || p->N <= 4096, CNL_E_UNSUPPORTED, "cnl_conv3x3_winograd_f32: more than 4096 images per launch need x_absmax");
    a.x = p->x; a.u3 = u5; a.xmax = p->x_absmax ? p->x_absmax : scal + 16; a.su = scal + 1; a.ymax = reinterpret_cast<unsigned*>(p->y_absmax);
    a.bias = p->bias; a.res = p->residual; a.y = p->y;
    const int upf = (p->flags & CNL_UPSAMPLE_IN) ? 2 : 1;
    a.N = p->N; a.Hs = p->H_in; a.Ws = p->W_in; a.H = p->H_in * upf; a.W = p->W_in * upf; a.Cin = p->Cin; a.Cout = p->Cout;
    a.CoutP = (p->Cout + 63) / 64 * 64;
    a.ldx = p->ldx; a.ldy = p->ldy; a.ldr = p->ldr;
    a.CC = p->Cin / 16;
    a.nb = (p->Cout + BN - 1) / BN; a.bx = (a.W + 15) / 16; a.by = (a.H + 7) / 8;
    // images side by side (STACK, see the kernel): where the 16-pixel blocks pad the map's width by more than the two gap columns cost
    a.SW = a.W + 2; a.mg_sw = 0; a.sh_sw = 0xFFu;
    const long long bx_stack = ((long long)p->N * a.SW + 15) / 16;
    const bool stack = upf == 1 && (a.W & 1) == 0 && p->N >= 2 && (long long)p->N * a.SW < (1ll << 30) && bx_stack * 100 <= (long long)p->N * a.bx * 92;
    if (stack) {
        unsigned sft = 0;
        while ((1ull << sft) < (unsigned)a.SW) ++sft;               // ceil(log2 SW); SW >= 4
        a.mg_sw = (unsigned)(((1ull << (31 + sft)) / (unsigned)a.SW) + 1);
        a.sh_sw = sft - 1;
        a.bx = (int)bx_stack;
    }
    const long long blocks = (long long)(stack ? 1 : p->N) * a.by * a.bx * a.nb;
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
    static cnl::DeviceOnce once_stack;
    int rc = stack ? cnl::kernel_setup(once_stack, reinterpret_cast<const void*>(&winograd6_kernel<true>), LDS_BYTES, &n_cu)
                   : cnl::kernel_setup(once, reinterpret_cast<const void*>(&winograd6_kernel<false>), LDS_BYTES, &n_cu);
    if (rc != CNL_OK) return rc;
    if (!p->x_absmax && (rc = cnl_wino5_own_absmax(p, scal, stream)) != CNL_OK) return rc;      // winograd5.hip
    const unsigned grid = (unsigned)(blocks < (long long)n_cu ? blocks : (long long)n_cu);
    if (stack) hipLaunchKernelGGL(winograd6_kernel<true>, dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(winograd6_kernel<false>, dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, a);
    return cnl::check_launch("winograd6_kernel");
}
