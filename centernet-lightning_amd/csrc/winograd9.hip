// winograd9.hip — 3x3 / stride-1 convolution as 1-D Winograd F(2,3) ALONG x with the three kernel rows folded into the reduction,
// every fp32 product formed on the FP16 matrix cores from scaled two-way splits (arithmetic of winograd5.hip: three cross terms,
// fp32 accumulation, error at or below the fp32 MFMA's).
//
//   out[y][2t + {0,1}] = A^T [ sum_{ky, ci} (G g[ky][.][ci]) (.) (B^T d[y + ky - 1][2t - 1 .. 2t + 2][ci]) ]
//
// i.e. FOUR GEMMs (one per transform position p) with K = 3 Cin:  Y_p[co][tile] = sum_{ky,ci} U_p[ky][co][ci] V_p[y + ky - 1][tile][ci],
// out0 = Y0 + Y1 + Y2, out1 = Y1 - Y2 - Y3.  Against the 2-D F(2x2,3x3) kernels (16 positions): 1.5x the matrix work, but
//   * 4 positions instead of 16: the 256 accumulator registers of a lane hold 8 rows x 64 pixels x 64 couts per work item (512 px
//     instead of 256) -> 2.7x fewer weight bytes per output from L2;
//   * V of an INPUT row serves the three output rows around it, and V_p = d_a +- d_b is one add: 2.25 V elements per output pixel
//     and channel instead of 4, 3.5 instead of 4.5 VALU operations each;
//   * the lane that transforms + splits a V fragment is the lane that feeds it to the MFMA (B operand: column = tile, k = 8
//     channels): V never goes through LDS.  Per 16-channel chunk a wave issues 144 MFMAs beside 280 VALU, 40 ds_read_b128, 23
//     global loads (12 weight fragments, 10.5 patch pieces) and 10.5 ds_write_b128 — 2.5 other issues per MFMA where winograd5
//     has 7.8;
//   * rounding error below 2-D F(2x2)'s (the transform adds two numbers, not four).
// Wave p of the 4-wave workgroup owns transform position p for the whole work item (each V_p is produced exactly once per CU, each
// weight fragment is loaded by exactly one wave); the four positions meet through LDS in the epilogue.
// The input patch of a chunk (10 rows x 66 pixels x 16 channels) travels global -> registers (issued early in the chunk before the
// one it is for: a plain buffer load costs the wave a few cycles) -> LDS (ds_write_b128 right after the chunk's barrier).  The
// LDS-DMA form (buffer_load ... lds) of the first version cost ~130 cycles of issue per 1 KB piece — 23 % of the chunk loop
// (timing builds: 6 200 -> 4 800 cycles per chunk without it, against 4 608 of pure MFMA issue).
// Weight fragments are single-buffered: the MFMAs of a chunk run kernel row by kernel row inside an input row (ky-major), so the
// ky = 0 / 1 / 2 fragments die at slices 114 / 126 / 144 of 144 and are reloaded for the next chunk 24-30 slices before their
// first use (L2 hits).
// Weights: U_p[ky] = (G g[ky])_p, scaled PER OUTPUT CHANNEL by a power of two (max |U S_u[co]| in [2^12, 2^13)) and split into two
// fp16 pieces, [ci/16][p][ky][piece][CoutP][16 ci]; the epilogue multiplies by 1/(S_v S_u[co]).
#include "cnl_common.h"
#include <utility>

#pragma clang fp contract(off)

#ifndef W9_NT_Y
#define W9_NT_Y 2     /* cache policy (aux) of the output stores: nt (see winograd5.hip) */
#endif
int cnl_wino_packed_stride(const cnl_conv_params* p);
namespace cnl_wino9 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct Args {
    const float* x;
    const void* u9;                   // pre-split, pre-scaled weights (fp16 pieces); UR launches: the four row-pair sets (weights9_up_kernel)
    const float* xmax;                // max |x| per image of this launch's input
    const float* isu;                 // [CoutP] 1 / S_u[co]
    unsigned* ymax;                   // optional: max |y| per image of this launch's output (atomic max on the bits)
    const float* bias;
    const float* res;
    float* y;
    int N, H, W, Cin, Cout, CoutP;    // H, W: output (= logical input) size
    int Hs, Ws;                       // stored input size (H/2, W/2 with CNL_UPSAMPLE_IN, else H, W)
    int ldx, ldy, ldr;
    int CC;                           // Cin / 16 (even)
    int nb, bx, by;                   // blocks along cout (64), x (64 px), y (8 rows)
    unsigned m_nb, m_bx, m_by;        // floor(2^32 / d) of the three: item index -> coordinates by multiply-high + one correction
    int ipb, lw;                      // images side by side in one 64-pixel block row (W = 32: 2, W = 16: 4; else 1) and log2 W for them; N = groups of ipb images
    int Nimg;                         // images of the launch (N = image groups)
    int pk;                           // packed rows (0: off): the launch's images side by side in ONE virtual row, each in a strip of pk = W + 2 columns
    unsigned m_pk;                    //   (its W pixels + the two columns of zero padding that separate it from the next image); floor(2^32 / pk)
    int blocks;
    unsigned x_bytes, u_bytes, y_bytes, r_bytes, b_bytes;
    unsigned flags;
    const float* fw;                  // optional (cnl_conv_params.fuse_w / fuse_part): a following 1x1 conv of <= 4 channels folded into the epilogue:
    float* fpart;                     //   fw [CoutP][4], fpart [CoutP / 32][N H W][4] partial sums per block of 32 couts
    unsigned fp_bytes, fp_block;      //   bytes of fpart, bytes per block
#ifdef W9_TRACE
    unsigned long long* trace;        // timing build: [item][32] s_memtime stamps of block 0 / thread 0 (16..23: inside the item's last MODE-0 chunk of parity 0)
#endif
};
#ifdef W9_TRACE
#ifndef W9_P0       // the eight slices stamped inside a chunk (-DW9_P0=.. -DW9_P7=.. to look elsewhere; a stamp costs ~60 cycles itself)
#define W9_P0 0
#define W9_P1 28
#define W9_P2 60
#define W9_P3 97
#define W9_P4 99
#define W9_P5 114
#define W9_P6 126
#define W9_P7 143
#endif
#define W9_STAMP(i_) do { if (blockIdx.x == 0 && tid == 0 && tr_item < 64) a.trace[tr_item * 32 + (i_)] = __builtin_readcyclecounter(); } while (0)
#else
#define W9_STAMP(i_) do {} while (0)
#endif

constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr int R = 8;                        // output rows per work item
constexpr int PR = R + 2;                   // patch rows
constexpr int TW = 32;                      // tiles (pixel pairs) per row of a work item: 64 output pixels
constexpr int BN = 64;                      // couts per work item
constexpr int PXH = TW + 1;                 // 33 pixels per parity plane of a patch row (columns x0-1 .. x0+64)
constexpr int QUAD_SLOTS = 2 * PXH;         // 66 16-byte slots per (row, channel quad): [parity][33]
constexpr int ROW_SLOTS = 4 * QUAD_SLOTS;   // 264 per patch row: [quad][parity][33]
constexpr int ROW_BYTES = ROW_SLOTS * 16;   // 4224
constexpr int P_SLOTS = 2688;               // PR * ROW_SLOTS = 2640 used (+ 48 slots that absorb the idle lanes of the last staging piece)
constexpr int P_BYTES = P_SLOTS * 16;       // 43008 per buffer (two buffers)
constexpr int X_BYTES = 65536;              // epilogue exchange: [4 blocks][4 positions][4 quads][64 lanes] x 16 B
constexpr int B_BYTES = 1536;               // the item's 64 bias values and 64 inverse weight scales (staged for the epilogue) + 64 rows of a folded 1x1 conv's weights
constexpr int LDS_BYTES = 2 * P_BYTES + X_BYTES + B_BYTES;     // 152064: one workgroup per CU (the accumulators allow no more)
constexpr int NSLICE = 144;                 // MFMAs per wave and chunk
constexpr int JOB_SLICES = 14;              // one V fragment (28 VALU operations) is produced beside 14 MFMAs
constexpr int JOB0 = 2;                     // job j runs in slices [JOB0 + 14 j, JOB0 + 14 j + 14)
constexpr int BARRIER_SLICE = 98;           // before job 7 (the first to read the next patch)
constexpr int NSTG = 6;                     // staging registers per thread: a patch (10 pieces (row i, pixel tid / 4, quad tid % 4) + 1 for the two last pixel columns) travels in two halves

#ifndef W9_AUX_X
#define W9_AUX_X 0    /* cache policy of the patch loads / the weight loads (A/B builds: 2 = nt, 1 = sc0, 16 = sc1) */
#endif
#ifndef W9_AUX_U
#define W9_AUX_U 0
#endif
template <int AUX = 0>
__device__ __forceinline__ u32x4 buf_load16(const void* base, unsigned bytes, unsigned voffset, unsigned soffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    return (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, soffset, AUX);
}
__device__ __forceinline__ void buf_store16(f32x4 v, float* base, unsigned bytes, unsigned voffset, unsigned soffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, voffset, soffset, W9_NT_Y);
}
__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// the split of a channel pair (v0, v1), scaled by the power of two S:  hi = RN16(v S) packed, r = v S - hi exactly (winograd5.hip)
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
// the lane id from the hardware (2 VALU) on an opaque input: per-lane values derived from it are computed where they are used instead
// of at kernel entry, from where they would stay live across the chunk loop
__device__ __forceinline__ int lane_now() {
    unsigned z = 0;
    asm volatile("" : "+v"(z));
    return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
}
__device__ __forceinline__ f32x4 lds_f4(const char* p) { return *reinterpret_cast<const f32x4*>(p); }
#define W9_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// ---- the static schedule of a chunk -----------------------------------------------------------------------------------------
// 24 segments of 6 MFMAs (3 terms x 2 cout halves), each one (input row r, kernel row ky) -> output row r - ky.  Rows in order,
// ky-major inside a row; the tail interleaves rows 7-9 so that the ky = 0 fragments die at slice 114 and the ky = 1 fragments at 126.
constexpr int SEG_ROW[24] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5, 6, 6, 6, 7, 7, 8, 7, 8, 9};
constexpr int SEG_KY[24] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 1, 1, 2, 2, 2};
constexpr int KY0_DEAD = 114, KY1_DEAD = 126;      // first slices after the last use of the ky = 0 / ky = 1 weight fragments

struct Item {               // per-work-item addressing state (two live: the item being multiplied and the one after it)
    unsigned vcol, vext;     // source offsets: column part of pieces 0..9 (the row is a scalar offset), full offset of piece 10
    unsigned u_voff;         // this lane's row of the weight planes
    unsigned img_base;       // scalar: byte offset of image n
    int y0m1;                // scalar: y0 - 1, first patch row
    float S;                 // power-of-two scale of V for this item's image
};
struct State {
    f32x16 acc[R][2];        // [output row][cout half]: D[cout][tile]
    u32x4 fb[4][2][2];       // weight fragments (A operand): [ky (UR: weight set)][cout half][piece], single-buffered (set 3: UR only)
    u32x4 vf[4][2];          // V fragments (B operand): [(10 chunk + row) % 4][piece]
    f32x4 raw[4];            // patch reads of a job: pixel a quad 0, a quad 1, pixel b quad 0, b quad 1 (consumed by operations 0..7, refilled for the next job right after)
    float v[8];              // transform temporaries of the running job (V, then its residual in place)
    u32x4 stg[NSTG];         // patch pieces on their way global -> LDS
    Item cur, nxt;
    unsigned row_pitch;      // scalar: bytes per stored input row
    const char* pa[2];       // LDS address of this lane's pixel a / b in patch buffer 0 / 1
    const char* pb[2];
    char* wb;                // LDS write address of piece 0 in buffer 0 (piece i: + i rows), and of piece 10
    char* wext;
    float sg;
#ifdef W9_TRACE
    unsigned long long* trp;   // timing build: this item's stamp row (block 0 / thread 0), else null
#endif
    bool idle;               // UR: this wave's transform position is identically zero (wave 2)
    float bst, ist;          // this lane's bias / inverse weight scale of the item (cout n0 + lane), on their way to LDS
    u32x4 fwst;              // FUSE: row n0 + lane of the folded 1x1 conv's weights, on its way to LDS
    char* sB;
};

// VALU operation o (0..27) of the job that builds V fragment `buf` (S: the scale of the image the fragment belongs to)
template <int O>
__device__ __forceinline__ void vop(State& st, const int buf, const float S) {
    if constexpr (O < 8) {
        st.v[O] = __builtin_fmaf(st.raw[2 + (O >> 2)][O & 3], st.sg, st.raw[O >> 2][O & 3]);
    } else if constexpr (O < 12) {
        st.vf[buf][0][O - 8] = split_hi_lo(st.v[2 * (O - 8)], S);
    } else if constexpr (O < 16) {
        st.vf[buf][0][O - 12] = split_hi_hi(st.vf[buf][0][O - 12], st.v[2 * (O - 12) + 1], S);
    } else if constexpr (O < 24) {
        constexpr int e = O - 16;
        st.v[e] = (e & 1) ? split_res_hi(st.v[e], S, st.vf[buf][0][e >> 1]) : split_res_lo(st.v[e], S, st.vf[buf][0][e >> 1]);
    } else {
        constexpr int j = O - 24;
        st.vf[buf][1][j] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(st.v[2 * j], st.v[2 * j + 1]));
        // (pinned: left alone, the compiler sinks the last jobs' packs of an item's last chunk below the epilogue — their results are first
        //  used by the next item — and keeps the 16 unpacked residuals alive across it instead of 8 packed registers)
        asm volatile("" : "+v"(st.vf[buf][1][j]));
    }
}
// LDS read i (0..3) of patch row `row` of buffer `pbuf`
template <int I>
__device__ __forceinline__ void rread(State& st, const int pbuf, const int row) {
    const char* p = (I < 2 ? st.pa[pbuf] : st.pb[pbuf]) + row * ROW_BYTES + (I & 1) * (QUAD_SLOTS * 16);
    st.raw[I] = lds_f4(p);
}
// weight fragment i (cout half i >> 1, piece i & 1) of kernel row KY of chunk cc: global -> registers
template <int KY, int NSET = 3>        // NSET: kernel rows (3) or row-pair weight sets (UR: 4) per (chunk, position)
__device__ __forceinline__ void load_b(State& st, const Args& a, const unsigned u_voff, const int cc, const int i, const unsigned u_plane, const unsigned u_wave) {
    const int nbh = i >> 1, piece = i & 1;
    const unsigned so = (unsigned)cc * ((unsigned)(8 * NSET) * u_plane) + u_wave + (unsigned)(KY * 2 + piece) * u_plane + (unsigned)nbh * 1024u;
    st.fb[KY][nbh][piece] = buf_load16<W9_AUX_U>(a.u9, a.u_bytes, u_voff, so);
}
// The patch of a chunk travels global -> staging registers -> LDS in two halves that share the registers: half A = rows 0..4 + the
// piece with the two last pixel columns of all rows (stg[5]), half B = rows 5..9.  Piece I of half HALF of chunk cc of item `it`:
// always issued (a branch around the load makes hipcc's wait counts conservative: vmcnt(0) at the next weight-fragment use); a row
// outside the image reads out of range -> zeros, the convolution's padding.
// UR: only the six DISTINCT patch rows travel — 0, 1, 3 in half A and 5, 7, 9 in half B (rows 2 / 4 / 6 / 8 repeat the row in front of them and are never read)
constexpr int ur_row(int HALF, int I) { return HALF == 0 ? (I == 0 ? 0 : 2 * I - 1) : 5 + 2 * I; }
template <int HALF, int I, bool UR = false>
__device__ __forceinline__ void pload(State& st, const Item& it, const Args& a, const int cc, const bool up) {
    if constexpr (UR && (I == 3 || I == 4)) return;
    if constexpr (I < 5) {
        const int iy = __builtin_amdgcn_readfirstlane(it.y0m1) + (UR ? ur_row(HALF, I) : 5 * HALF + I);     // (uniform by construction; the copies cur = nxt hide it from the
        const bool ok = (unsigned)iy < (unsigned)a.H;                               //  compiler, which then wraps every load in a waterfall loop)
        const int sy = ok ? (up ? (iy >> 1) : iy) : 0;
        const unsigned so = __builtin_amdgcn_readfirstlane(it.img_base + (unsigned)sy * st.row_pitch + (unsigned)cc * 64u);
        st.stg[I] = buf_load16<W9_AUX_X>(a.x, a.x_bytes, ok ? it.vcol : OOB, so);
    } else {
        static_assert(HALF == 0, "the column piece belongs to half A");
        st.stg[5] = buf_load16<W9_AUX_X>(a.x, a.x_bytes, it.vext, __builtin_amdgcn_readfirstlane(it.img_base + (unsigned)cc * 64u));   // (threads >= 80: out of range -> zeros)
    }
}
// ... staging register -> patch buffer `pbuf`
template <int HALF, int I, bool UR = false>
__device__ __forceinline__ void pwrite(State& st, const int pbuf) {
    if constexpr (UR && (I == 3 || I == 4)) return;
    if constexpr (I < 5) *reinterpret_cast<u32x4*>(st.wb + pbuf * P_BYTES + (UR ? ur_row(HALF, I) : 5 * HALF + I) * ROW_BYTES) = st.stg[I];
    else *reinterpret_cast<u32x4*>(st.wext + pbuf * P_BYTES) = st.stg[5];                        // (threads >= 80: into the slack slots)
}
template <int HALF, bool UR, int... I>
__device__ __forceinline__ void pload_all(State& st, const Item& it, const Args& a, const int cc, const bool up, std::integer_sequence<int, I...>) {
    (pload<HALF, I, UR>(st, it, a, cc, up), ...);
}
template <int HALF, bool UR, int... I>
__device__ __forceinline__ void pwrite_all(State& st, const int pbuf, std::integer_sequence<int, I...>) {
    (pwrite<HALF, I, UR>(st, pbuf), ...);
}

// One slice: MFMA S of the chunk with parity PAR, and what is issued beside it.  The chunk stream runs on across work items:
// MODE 0 = a chunk with two more chunks of its item behind it, 1 = the item's last but one (the patch it requests is chunk 0 of the
// NEXT item), 2 = the item's last (requests chunk 1 of the next item, loads the weights and builds the first two V rows of its chunk 0).
// is slice S the first MFMA of a chunk into its accumulator block (output row, cout half)?  An item's first chunk starts those from
// C = 0 instead of the kernel zeroing all 256 accumulator registers per item (256 v_accvgpr_write = 1 K cycles with nothing beside them)
// UR ("upsampled rows": the launch folds a nearest-2x upsample into its patch gather AND has the row-pair weights, cnl_conv_params.w_up): y0 and H are even, so
// the patch rows (1,2) (3,4) (5,6) (7,8) of an item are the same source row twice, and the three kernel rows of an output row meet only TWO distinct input rows:
//   output row 2m   = row 2m   x g[0]          + row 2m+1 x (g[1] + g[2])         (rows 2m+1, 2m+2 identical)
//   output row 2m+1 = row 2m+1 x (g[0] + g[1]) + row 2m+3 x g[2]                  (rows 2m+1, 2m+2 identical)
// With the four pre-summed weight sets {g0, g0+g1, g1+g2, g2} (weights9_up_kernel) 16 of a chunk's 24 segments remain — 96 matrix instructions instead of 144:
// a segment (r, ky) of an ODD patch row r takes set ky + 1, the (r, 0) segment of an EVEN row takes set 0, its (r, 1) and (r, 2) segments are left out (their
// slices keep everything that is issued beside the MFMA).  Another order of the same fp32 products and two pre-summed weights: results within rounding of the
// general form, not bit-identical to it — a function of the shape and of w_up being given, never of the batch.
constexpr bool seg_skipped(int seg, bool UR) { return UR && (SEG_ROW[seg] & 1) == 0 && SEG_KY[seg] != 0; }
constexpr int seg_wset(int seg, bool UR) { return !UR ? SEG_KY[seg] : ((SEG_ROW[seg] & 1) ? SEG_KY[seg] + 1 : 0); }
constexpr bool first_use(int S, bool UR = false) {
    const int yo = SEG_ROW[S / 6] - SEG_KY[S / 6], nbh = S & 1;
    for (int s = 0; s < S; ++s)
        if (!seg_skipped(s / 6, UR) && SEG_ROW[s / 6] - SEG_KY[s / 6] == yo && (s & 1) == nbh) return false;
    return true;
}
constexpr int UR_SET0_DEAD = 96;      // UR: first slice after the last use of weight set 0 (segment (6, 0)); sets 1 / 2 die where ky = 0 / 1 do, set 3 with the chunk
template <int S, int PAR, int MODE, bool FIRST, bool FUSE = false, bool UR = false>
__device__ __forceinline__ void slice(State& st, const Args& a, const int cn, const bool up, const unsigned u_plane, const unsigned u_wave) {
    constexpr int seg = S / 6;
    constexpr int r = SEG_ROW[seg], ky = SEG_KY[seg];
    constexpr int term = (S % 6) / 2, nbh = S & 1;
    constexpr int ku = term == 1 ? 1 : 0, kv = term == 0 ? 1 : 0;         // terms: hi lo', lo hi', hi hi'
    constexpr int vbuf = ((UR && r >= 2 && (r & 1) == 0 ? r - 1 : r) + 2 * PAR) & 3;      // UR: an even patch row IS the row in front of it — its segment reads that row's fragment
#ifdef W9_TRACE
    if constexpr (PAR == 0 && MODE == 0 && !FIRST && (S == W9_P0 || S == W9_P1 || S == W9_P2 || S == W9_P3 || S == W9_P4 || S == W9_P5 || S == W9_P6 || S == W9_P7)) {
        constexpr int k = S == W9_P0 ? 16 : S == W9_P1 ? 17 : S == W9_P2 ? 18 : S == W9_P3 ? 19 : S == W9_P4 ? 20 : S == W9_P5 ? 21 : S == W9_P6 ? 22 : 23;
        if (st.trp) st.trp[k] = __builtin_readcyclecounter();
    }
#endif
    if constexpr (S == BARRIER_SLICE) {
        // every wave is done reading this chunk's patch, and the next chunk's (written a chunk ago) is complete
        W9_BARRIER();
        __builtin_amdgcn_sched_barrier(0);
    }
#ifdef W9_SKIP4     // timing build (wrong results): every fourth MFMA is left out — what 25 % fewer matrix instructions buy under the chip's power limit
    if constexpr ((S & 3) != 3 || (FIRST && first_use(S, UR))) {
#endif
    constexpr int wset = seg_wset(seg, UR);
    if constexpr (!seg_skipped(seg, UR)) {
        if constexpr (FIRST && first_use(S, UR)) st.acc[r - ky][nbh] = mfma16(st.fb[wset][nbh][ku], st.vf[vbuf][kv], f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f});
        else st.acc[r - ky][nbh] = mfma16(st.fb[wset][nbh][ku], st.vf[vbuf][kv], st.acc[r - ky][nbh]);
    }
#ifdef W9_SKIP4
    }
#endif
    // the MFMA leads its slice: left to the scheduler, a slice's buffer load is issued BEFORE its MFMA, and the ~16 cycles the load's
    // issue takes (1 KB through the address unit) open a bubble in the matrix pipe instead of hiding in the MFMA's 32-cycle shadow
    __builtin_amdgcn_sched_barrier(0);
    // ---- V production: job j builds the fragment of row j + 2 of this chunk (j < 8) or of row j - 8 of the next chunk ----
    if constexpr (S >= JOB0 && S < JOB0 + 10 * JOB_SLICES) {
        constexpr int j = (S - JOB0) / JOB_SLICES, k = (S - JOB0) % JOB_SLICES;
        constexpr int buf = j < 8 ? ((j + 2 + 2 * PAR) & 3) : ((j - 8 + 2 * (PAR ^ 1)) & 3);
        constexpr int prow = j < 8 ? j + 2 : j - 8;                                  // the patch row this job's fragment belongs to
        constexpr bool idle = UR && prow >= 2 && (prow & 1) == 0;                    // UR: rows 2, 4, 6, 8 have no fragment of their own (and no patch row in LDS)
        if constexpr (!idle) {
            const float Sj = (MODE == 2 && j >= 8) ? st.nxt.S : st.cur.S;
            vop<2 * k>(st, buf, Sj);
            vop<2 * k + 1>(st, buf, Sj);
        }
        // raw reads of the next job (j + 1): rows 3..9 of this patch, then rows 0, 1, 2 of the next
        if constexpr (k >= 4 && k <= 7) {
            constexpr int jn = j + 1;
            constexpr int nrow = jn < 8 ? jn + 2 : jn - 8;
            constexpr int npb = jn < 8 ? PAR : (PAR ^ 1);
            constexpr bool nidle = UR && nrow >= 2 && (nrow & 1) == 0;
            if constexpr (!nidle) rread<k - 4>(st, npb, nrow);
        }
    }
    // ---- weight fragments: kernel row 2 of THIS chunk (first used at slice 30), rows 0 / 1 of the next once this chunk is done with them ----
    if constexpr (!UR) {
        if constexpr (S < 4) load_b<2>(st, a, st.cur.u_voff, cn, S, u_plane, u_wave);
        if constexpr (S >= KY0_DEAD && S < KY0_DEAD + 8 && (S - KY0_DEAD) % 2 == 0)
            load_b<0>(st, a, MODE == 2 ? st.nxt.u_voff : st.cur.u_voff, MODE == 2 ? 0 : cn + 1, (S - KY0_DEAD) / 2, u_plane, u_wave);
        if constexpr (S >= KY1_DEAD && S < KY1_DEAD + 8 && (S - KY1_DEAD) % 2 == 0)
            load_b<1>(st, a, MODE == 2 ? st.nxt.u_voff : st.cur.u_voff, MODE == 2 ? 0 : cn + 1, (S - KY1_DEAD) / 2, u_plane, u_wave);
    } else {
        // UR: set 3 (g2: first used by segment (3, 2), slice 48) of THIS chunk; sets 0 / 1 / 2 of the next once this chunk is done with them
        if constexpr (S < 4) load_b<3, 4>(st, a, st.cur.u_voff, cn, S, u_plane, u_wave);
        if constexpr (S >= UR_SET0_DEAD && S < UR_SET0_DEAD + 8 && (S - UR_SET0_DEAD) % 2 == 0)
            load_b<0, 4>(st, a, MODE == 2 ? st.nxt.u_voff : st.cur.u_voff, MODE == 2 ? 0 : cn + 1, (S - UR_SET0_DEAD) / 2, u_plane, u_wave);
        if constexpr (S >= KY0_DEAD && S < KY0_DEAD + 8 && (S - KY0_DEAD) % 2 == 0)
            load_b<1, 4>(st, a, MODE == 2 ? st.nxt.u_voff : st.cur.u_voff, MODE == 2 ? 0 : cn + 1, (S - KY0_DEAD) / 2, u_plane, u_wave);
        if constexpr (S >= KY1_DEAD && S < KY1_DEAD + 8 && (S - KY1_DEAD) % 2 == 0)
            load_b<2, 4>(st, a, MODE == 2 ? st.nxt.u_voff : st.cur.u_voff, MODE == 2 ? 0 : cn + 1, (S - KY1_DEAD) / 2, u_plane, u_wave);
    }
    // ---- patch of the chunk after next (of the next item in MODE 1 / 2), in two halves through the same staging registers:
    //   slices 28..36   half B of the NEXT chunk's patch (requested a chunk ago) -> the other buffer, rows 5..9 (first read a chunk from now)
    //   slices 38..58   request half A        99..109  half A -> this chunk's buffer (dead after the barrier)        111..127  request half B
    if constexpr (S >= 28 && S <= 36 && (S - 28) % 2 == 0) pwrite<1, (S - 28) / 2, UR>(st, PAR ^ 1);
    if constexpr (S >= 38 && S <= 58 && (S - 38) % 4 == 0) {
        if constexpr (MODE == 0) pload<0, (S - 38) / 4, UR>(st, st.cur, a, cn + 2, up);
        else pload<0, (S - 38) / 4, UR>(st, st.nxt, a, MODE - 1, up);
    }
    if constexpr (S >= 99 && S <= 109 && (S - 99) % 2 == 0) pwrite<0, (S - 99) / 2, UR>(st, PAR);
    if constexpr (S >= 111 && S <= 127 && (S - 111) % 4 == 0) {
        if constexpr (MODE == 0) pload<1, (S - 111) / 4, UR>(st, st.cur, a, cn + 2, up);
        else pload<1, (S - 111) / 4, UR>(st, st.nxt, a, MODE - 1, up);
    }
    // ---- the item's bias / weight-scale values (requested at the top of the item) -> LDS for the epilogue; every wave writes the same 64
    //      values (no branch inside the chunk); two chunk barriers lie between this and the first read ----
    if constexpr (FIRST && S == 60) {
        float* sb = reinterpret_cast<float*>(st.sB) + lane_now();
        sb[0] = st.bst;
        sb[64] = st.ist;
    }
    if constexpr (FUSE && FIRST && S == 62) *reinterpret_cast<u32x4*>(st.sB + 512 + lane_now() * 16) = st.fwst;
    __builtin_amdgcn_sched_barrier(0);
}

// ---- the COMPACT chunk of the row-pair form (UR): its 16 segments as 96 slices, six V jobs, seven patch pieces ------------------------------------------------------
// (The first UR version kept the 144-slice schedule with 48 MFMA-less slices: 4 280 cycles per chunk against 3 072 of matrix issue — an empty slice still costs the ~20 cycles
//  of what is issued beside it.)  Segments in issue order: (patch row r, kernel row ky); the V fragment is that of the ODD row (an even row r reads row r - 1's: the same
//  source row), numbered 0..5 for rows 0, 1, 3, 5, 7, 9 and 6, 7 for rows 0, 1 of the NEXT chunk: buffer = (number + 2 PAR) & 3.  The set-0 segment of an even row LEADS its
//  group, so that set 0 dies at slice 54 and is reloaded 42 slices before the next chunk needs it (sets 1 / 2: 30 / 28 slices; set 3 is loaded in the chunk that uses it).
#ifndef W9_UR_COMPACT
#define W9_UR_COMPACT 1
#endif
#ifndef W9_UR_SKIP_ZERO      /* the row-pair form's epilogue leaves the identically-zero transform position (wave 2) out of the exchange (A/B: 0) */
#define W9_UR_SKIP_ZERO 1
#endif
constexpr int NSLICE_UR = 96;
constexpr int URS_ROW[16] = {0, 2, 1, 1, 4, 3, 3, 3, 6, 5, 5, 5, 7, 7, 7, 9};
constexpr int URS_KY[16] = {0, 0, 0, 1, 0, 0, 1, 2, 0, 0, 1, 2, 0, 1, 2, 2};
constexpr int urs_set(int seg) { return (URS_ROW[seg] & 1) ? URS_KY[seg] + 1 : 0; }
constexpr int urs_frag(int seg) { const int r = URS_ROW[seg], ro = (r >= 2 && (r & 1) == 0) ? r - 1 : r; return (ro + 1) / 2; }
constexpr bool urs_first_use(int S) {
    const int yo = URS_ROW[S / 6] - URS_KY[S / 6], nbh = S & 1;
    for (int s = 0; s < S; ++s)
        if (URS_ROW[s / 6] - URS_KY[s / 6] == yo && (s & 1) == nbh) return false;
    return true;
}
constexpr int UR_JOB0 = 2, UR_JOB_STRIDE = 16;        // job j: slices [2 + 16 j, 16 + 16 j): fragments 2..5 of this chunk (rows 3, 5, 7, 9), then 6, 7 (rows 0, 1 of the next)
constexpr int UR_BARRIER = 44;                        // behind the last read of this chunk's patch (row 9: slices 38..41)
constexpr int UR_SET0_LOAD = 54, UR_SET1_LOAD = 78, UR_SET2_LOAD = 86;
// (Tiles start on pixel pairs, so transform position 2 (d2 - d1) of an upsampled row is identically zero: wave 2 multiplies zeros.  Leaving its matrix work out needs a second code path per chunk —
//  a wave-uniform branch around two unrolled chunk bodies: the 256 accumulators then live across a merge and the compiler spilled 2 300-2 700 registers; not kept.  The epilogue leaves position 2 out of the exchange.)
template <int S, int PAR, int MODE, bool FIRST>
__device__ __forceinline__ void slice_ur(State& st, const Args& a, const int cn, const bool up, const unsigned u_plane, const unsigned u_wave) {
    constexpr int seg = S / 6;
    constexpr int r = URS_ROW[seg], ky = URS_KY[seg];
    constexpr int term = (S % 6) / 2, nbh = S & 1;
    constexpr int ku = term == 1 ? 1 : 0, kv = term == 0 ? 1 : 0;         // terms: hi lo', lo hi', hi hi'
    constexpr int vbuf = (urs_frag(seg) + 2 * PAR) & 3, wset = urs_set(seg);
    if constexpr (S == UR_BARRIER) {
        W9_BARRIER();
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (FIRST && urs_first_use(S)) st.acc[r - ky][nbh] = mfma16(st.fb[wset][nbh][ku], st.vf[vbuf][kv], f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f});
    else st.acc[r - ky][nbh] = mfma16(st.fb[wset][nbh][ku], st.vf[vbuf][kv], st.acc[r - ky][nbh]);
    __builtin_amdgcn_sched_barrier(0);
    // ---- V production ----
    if constexpr (S >= UR_JOB0 && (S - UR_JOB0) % UR_JOB_STRIDE < 14) {
        constexpr int j = (S - UR_JOB0) / UR_JOB_STRIDE, k = (S - UR_JOB0) % UR_JOB_STRIDE;
        constexpr int buf = (j + 2 + 2 * PAR) & 3;
        const float Sj = (MODE == 2 && j >= 4) ? st.nxt.S : st.cur.S;
        vop<2 * k>(st, buf, Sj);
        vop<2 * k + 1>(st, buf, Sj);
        if constexpr (k >= 4 && k <= 7) {             // raw reads of the next job: rows 5, 7, 9 of this patch, rows 0, 1 of the next, and row 3 of the next for ITS job 0
            constexpr int jn = j + 1;
            constexpr int nrow = jn < 4 ? 2 * jn + 3 : (jn < 6 ? jn - 4 : 3);
            constexpr int npb = jn < 4 ? PAR : (PAR ^ 1);
            rread<k - 4>(st, npb, nrow);
        }
    }
    // ---- weight fragments: set 3 of THIS chunk (first used at slice 42), sets 0 / 1 / 2 of the next once this chunk is done with them ----
    if constexpr (S < 8 && S % 2 == 0) load_b<3, 4>(st, a, st.cur.u_voff, cn, S / 2, u_plane, u_wave);
    if constexpr (S >= UR_SET0_LOAD && S < UR_SET0_LOAD + 8 && (S - UR_SET0_LOAD) % 2 == 0)
        load_b<0, 4>(st, a, MODE == 2 ? st.nxt.u_voff : st.cur.u_voff, MODE == 2 ? 0 : cn + 1, (S - UR_SET0_LOAD) / 2, u_plane, u_wave);
    if constexpr (S >= UR_SET1_LOAD && S < UR_SET1_LOAD + 8 && (S - UR_SET1_LOAD) % 2 == 0)
        load_b<1, 4>(st, a, MODE == 2 ? st.nxt.u_voff : st.cur.u_voff, MODE == 2 ? 0 : cn + 1, (S - UR_SET1_LOAD) / 2, u_plane, u_wave);
    if constexpr (S >= UR_SET2_LOAD && S < UR_SET2_LOAD + 8 && (S - UR_SET2_LOAD) % 2 == 0)
        load_b<2, 4>(st, a, MODE == 2 ? st.nxt.u_voff : st.cur.u_voff, MODE == 2 ? 0 : cn + 1, (S - UR_SET2_LOAD) / 2, u_plane, u_wave);
    // ---- patch of the chunk after next, the six distinct rows in two halves through the staging registers:
    //   1 / 3 / 5   rows 5, 7, 9 of the NEXT chunk's patch (requested 44 slices ago) -> the other buffer
    //   7 / 9 / 11 / 13   request rows 0, 1, 3 + the column piece        45 / 47 / 49 / 51   ... -> this chunk's buffer (dead behind the barrier: 38 slices for the loads to land)        53 / 55 / 57   request rows 5, 7, 9
    if constexpr (S >= 1 && S <= 5 && S % 2 == 1) pwrite<1, (S - 1) / 2, true>(st, PAR ^ 1);
    if constexpr (S >= 7 && S <= 13 && S % 2 == 1) {
        constexpr int I = S == 13 ? 5 : (S - 7) / 2;
        if constexpr (MODE == 0) pload<0, I, true>(st, st.cur, a, cn + 2, up);
        else pload<0, I, true>(st, st.nxt, a, MODE - 1, up);
    }
    if constexpr ((S >= 45 && S <= 49 && S % 2 == 1) || S == 51) pwrite<0, S == 51 ? 5 : (S - 45) / 2, true>(st, PAR);
    if constexpr (S >= 53 && S <= 57 && S % 2 == 1) {
        if constexpr (MODE == 0) pload<1, (S - 53) / 2, true>(st, st.cur, a, cn + 2, up);
        else pload<1, (S - 53) / 2, true>(st, st.nxt, a, MODE - 1, up);
    }
    // ---- the item's bias / weight-scale values -> LDS for the epilogue ----
    if constexpr (FIRST && S == 32) {
        float* sb = reinterpret_cast<float*>(st.sB) + lane_now();
        sb[0] = st.bst;
        sb[64] = st.ist;
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int PAR, int MODE, bool FIRST, bool FUSE, bool UR, int... S>
__device__ __forceinline__ void chunk_impl(State& st, const Args& a, const int cn, const bool up, const unsigned u_plane, const unsigned u_wave,
                                           std::integer_sequence<int, S...>) {
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (UR && W9_UR_COMPACT) (slice_ur<S, PAR, MODE, FIRST>(st, a, cn, up, u_plane, u_wave), ...);
    else (slice<S, PAR, MODE, FIRST, FUSE, UR>(st, a, cn, up, u_plane, u_wave), ...);
}
template <int PAR, int MODE, bool FIRST = false, bool FUSE = false, bool UR = false>        // FIRST: the first chunk of an item (its accumulators start from zero)
__device__ __forceinline__ void chunk(State& st, const Args& a, const int cn, const bool up, const unsigned u_plane, const unsigned u_wave) {
    chunk_impl<PAR, MODE, FIRST, FUSE, UR>(st, a, cn, up, u_plane, u_wave, std::make_integer_sequence<int, (UR && W9_UR_COMPACT) ? NSLICE_UR : NSLICE>{});
}
template <int... O>
__device__ __forceinline__ void job_all(State& st, const int buf, const float S, std::integer_sequence<int, O...>) {
    (vop<O>(st, buf, S), ...);
}

// RES: the launch adds a residual; FUSE: a following 1x1 conv of <= 4 channels is folded into the epilogue; PK: packed rows (cnl_wino_packed_stride).
// Template parameters, not run-time branches: the plain instantiations are the code of round 4 (the packed-row index arithmetic as uniform
// branches cost the unpacked launches of C1 1.5-2.3 us each: more scalar registers live across the chunk loop — profiles/r05_experiments.txt r5c)
// UR: folded nearest-2x upsample with the row-pair weight sets (see seg_skipped()): 96 matrix instructions per chunk.
template <bool RES, bool FUSE = false, bool PK = false, bool UR = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void winograd9_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sX = smem + 2 * P_BYTES;      // exchange region of the epilogue: two halves of 32 KB

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = transform position p owned by this wave
    const int h = lane >> 5, t = lane & 31;
    const bool up = UR || (a.flags & CNL_UPSAMPLE_IN);
    const unsigned u_plane = (unsigned)(a.CoutP * 32);               // bytes per (chunk, position, ky, piece) plane of U
    const unsigned u_wave = (unsigned)wave * (UR ? 8u : 6u) * u_plane;
    typedef std::make_integer_sequence<int, 6> HalfA;      // rows 0..4 + the column piece
    typedef std::make_integer_sequence<int, 5> HalfB;      // rows 5..9

    State st;
    st.idle = UR && wave == 2;
    st.sB = smem + 2 * P_BYTES + X_BYTES;
    st.row_pitch = (unsigned)(a.Ws * a.ldx * 4);
    // V_p = d[offa] + sg d[offb] over the four pixels 2t-1 .. 2t+2 of a tile:  p = 0: -(d0 - d2) (its weights are stored negated), 1: d1 + d2,
    // 2: d2 - d1, 3: d1 - d3 — so that the two OUTER pixels d0 / d3 are always operand b (see below for images side by side).
    const int offa = wave == 0 ? 2 : (wave == 2 ? 2 : 1);
    const int offb = wave == 0 ? 0 : (wave == 3 ? 3 : (wave == 2 ? 1 : 2));
    const int tpi = (a.ipb > 1) ? (1 << (a.lw - 1)) : 64;            // tiles per image in the block row
    const bool outer_is_neighbour = a.ipb > 1 && ((wave == 0 && (t & (tpi - 1)) == 0) || (wave == 3 && (t & (tpi - 1)) == tpi - 1));
    st.sg = wave == 1 ? 1.f : -1.f;
    const int si_lane = (a.ipb > 1) ? ((2 * t) >> a.lw) : 0;         // this lane's sub-image (V production: its tile's image)
    // patch image [row][quad][parity][33 px] x 16 B: the lanes of a ds_read_b128 group (same h, 16 distinct t mod 16) read
    // consecutive slots of one plane — all 64 banks, no conflicts
    {
        const int sa = (2 * h) * QUAD_SLOTS + (offa & 1) * PXH + t + (offa >> 1);
        // ... except where that outer pixel belongs to the neighbouring image: those lanes read patch column 0 instead (x = -1: outside
        // every image, zero-filled in every row of every chunk) — the zero padding itself, also when the neighbour holds Inf / NaN
        const int sb = outer_is_neighbour ? (2 * h) * QUAD_SLOTS : (2 * h) * QUAD_SLOTS + (offb & 1) * PXH + t + (offb >> 1);
        st.pa[0] = smem + sa * 16; st.pa[1] = smem + P_BYTES + sa * 16;
        st.pb[0] = smem + sb * 16; st.pb[1] = smem + P_BYTES + sb * 16;
    }
    // staging pieces: piece i < 10 = (patch row i, column tid / 4, channel quad tid % 4): 64 bytes per pixel from 4 lanes; piece 10 =
    // the columns 64, 65 of all ten rows (threads 0..79).  8 consecutive lanes (2 pixels x 4 quads) of a ds_write_b128 hit all 32 banks.
    {
        const int q = tid & 3, c = tid >> 2;
        st.wb = smem + (q * QUAD_SLOTS + (c & 1) * PXH + (c >> 1)) * 16;
        const int er = tid >> 3, ec = 64 + ((tid >> 2) & 1);
        st.wext = tid < 80 ? smem + (er * ROW_SLOTS + q * QUAD_SLOTS + (ec & 1) * PXH + (ec >> 1)) * 16 : smem + (PR * ROW_SLOTS + (tid & 31)) * 16;
    }
    const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane((a.flags & CNL_RELU) ? 0 : (int)0xff800000u));      // ReLU floor or -inf (scalar register)

    // coordinates of a work item (scalars) and the per-thread addressing that follows from them
    struct Coord { int n, y0, x0, n0; };
    // q = b / d, r = b % d for uniform b with m = floor(2^32 / d) (0xFFFFFFFF for d = 1): the multiply-high is at most one short
#define W9_DIVMOD(q_, r_, b_, d_, m_)                                                                            \
    do {                                                                                                         \
        unsigned qq_ = __builtin_amdgcn_readfirstlane(__umulhi((b_), (m_)));                                     \
        unsigned rr_ = (b_) - qq_ * (unsigned)(d_);                                                              \
        if (rr_ >= (unsigned)(d_)) { ++qq_; rr_ -= (unsigned)(d_); }                                             \
        (q_) = qq_; (r_) = rr_;                                                                                  \
    } while (0)
    /* the same per lane (packed rows: virtual column -> image, pixel) */                                        
#define W9_VDIVMOD(q_, r_, b_, d_, m_)                                                                           \
    do {                                                                                                         \
        unsigned qq_ = __umulhi((b_), (m_));                                                                     \
        unsigned rr_ = (b_) - qq_ * (unsigned)(d_);                                                              \
        if (rr_ >= (unsigned)(d_)) { ++qq_; rr_ -= (unsigned)(d_); }                                             \
        (q_) = qq_; (r_) = rr_;                                                                                  \
    } while (0)
#define W9_COORD(c_, item_)                                                                                      \
    do {                                                                                                         \
        unsigned b_ = __builtin_amdgcn_readfirstlane(cnl::xcd_remap((item_), (unsigned)a.blocks));               \
        unsigned q_, nbi_, bxi_, byi_;                                                                           \
        W9_DIVMOD(q_, nbi_, b_, a.nb, a.m_nb); b_ = q_;                                                          \
        W9_DIVMOD(q_, bxi_, b_, a.bx, a.m_bx); b_ = q_;                                                          \
        W9_DIVMOD(q_, byi_, b_, a.by, a.m_by);                                                                   \
        (c_).n = (int)q_; (c_).y0 = (int)byi_ * R; (c_).x0 = (int)bxi_ * (2 * TW); (c_).n0 = (int)nbi_ * BN;     \
    } while (0)
#define W9_ITEM(it_, c_)                                                                                         \
    do {                                                                                                         \
        (it_).y0m1 = (c_).y0 - 1;                                                                                \
        (it_).img_base = (unsigned)((c_).n * a.ipb * a.Hs) * st.row_pitch;                                       \
        int tid_ = tid;                                                                                          \
        asm volatile("" : "+v"(tid_));      /* keeps the per-thread decode inside the item loop (hoisted, its values live across the main loop) */ \
        const int q_ = tid_ & 3, ix_ = (c_).x0 - 1 + (tid_ >> 2);                                                \
        const int er_ = tid_ >> 3, ex_ = (c_).x0 + 63 + ((tid_ >> 2) & 1), ey_ = (c_).y0 - 1 + er_;              \
        if (PK) {          /* packed rows: virtual column -> (image, pixel of its strip); the strip's last two columns, a column before the */ \
                           /* first strip or behind the last are the zero padding (out of range -> zeros) */   \
            unsigned si_, px_, esi_, epx_;                                                                       \
            W9_VDIVMOD(si_, px_, (unsigned)ix_, a.pk, a.m_pk);                                                   \
            const bool okc_ = ix_ >= 0 && (int)px_ < a.W && (int)si_ < a.Nimg;                                   \
            const int ush_ = up ? 1 : 0;       /* a folded nearest-2x upsample: logical (row, column) -> stored (row >> 1, column >> 1) */ \
            (it_).vcol = okc_ ? (unsigned)((((int)si_ * a.Hs * a.Ws + ((int)px_ >> ush_)) * a.ldx + q_ * 4) * 4) : OOB; \
            W9_VDIVMOD(esi_, epx_, (unsigned)ex_, a.pk, a.m_pk);                                                 \
            const bool oke_ = tid_ < 80 && (unsigned)ey_ < (unsigned)a.H && (int)epx_ < a.W && (int)esi_ < a.Nimg; \
            (it_).vext = oke_ ? (unsigned)(((((int)esi_ * a.Hs + (ey_ >> ush_)) * a.Ws + ((int)epx_ >> ush_)) * a.ldx + q_ * 4) * 4) : OOB; \
        } else if (a.ipb > 1) {   /* block row = ipb images of width 2^lw side by side (x0 = 0): column -> (sub-image, pixel) */ \
            const int si_ = ix_ >> a.lw, px_ = ix_ & (a.W - 1);                                                  \
            const bool okc_ = (unsigned)ix_ < 64u && (c_).n * a.ipb + si_ < a.Nimg;                              \
            (it_).vcol = okc_ ? (unsigned)(((si_ * a.H * a.W + px_) * a.ldx + q_ * 4) * 4) : OOB;                \
            const int esi_ = ex_ >> a.lw, epx_ = ex_ & (a.W - 1);                                                \
            const bool oke_ = tid_ < 80 && (unsigned)ey_ < (unsigned)a.H && ex_ < 64 && (c_).n * a.ipb + esi_ < a.Nimg; \
            (it_).vext = oke_ ? (unsigned)((((esi_ * a.H + ey_) * a.W + epx_) * a.ldx + q_ * 4) * 4) : OOB;      \
        } else {                                                                                                 \
            const int sx_ = up ? (ix_ >> 1) : ix_;                                                               \
            (it_).vcol = (unsigned)ix_ < (unsigned)a.W ? (unsigned)((sx_ * a.ldx + q_ * 4) * 4) : OOB;           \
            const bool ok_ = tid_ < 80 && (unsigned)ey_ < (unsigned)a.H && (unsigned)ex_ < (unsigned)a.W;        \
            const int esy_ = up ? (ey_ >> 1) : ey_, esx_ = up ? (ex_ >> 1) : ex_;                                \
            (it_).vext = ok_ ? (unsigned)(((esy_ * a.Ws + esx_) * a.ldx + q_ * 4) * 4) : OOB;                    \
        }                                                                                                        \
        (it_).u_voff = (unsigned)(((c_).n0 + (tid_ & 31)) * 32 + ((tid_ >> 5) & 1) * 16);                        \
    } while (0)
    // max |x| of the image a lane's tile (or an epilogue thread's tile) belongs to; images past the end of the batch: 0 -> scale 1
#define W9_XMAX_OF(img_) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(                           \
        __builtin_amdgcn_make_buffer_rsrc((void*)a.xmax, 0, a.Nimg * 4 * AMS, 0x00020000), (unsigned)(img_) * (4u * AMS), 0, 0))
    /* the image of this lane's tile (V production) in work item c_ */
#define W9_IMG_LANE(out_, c_)                                                                                    \
    do {                                                                                                         \
        (out_) = (c_).n * a.ipb + si_lane;                                                                       \
        if (PK) {                                                                                                \
            unsigned q_, r_;                                                                                     \
            W9_VDIVMOD(q_, r_, (unsigned)((c_).x0 + 2 * (lane_now() & 31)), a.pk, a.m_pk);                       \
            (out_) = (int)q_;                                                                                    \
        }                                                                                                        \
    } while (0)
    // (a buffer load, range-checked by the hardware: a branch around a plain load makes the compiler wait for it, vmcnt(0), inside the
    //  branch — at the top of an item that is a wait for the previous item's stores)
    // power-of-two scale of V from the image's maximum: |V| <= 2 max |x|, 2 max |x| S in [2^13, 2^14); es_ = log2 S
#define W9_SCALE_EXP(es_, xmax_)                                                                                 \
    do {                                                                                                         \
        const float mx2_ = 2.f * (xmax_);                                                                        \
        (es_) = 0;                                                                                               \
        if (mx2_ > 0.f && mx2_ < __builtin_inff()) {                                                             \
            int e_;                                                                                              \
            (void)__builtin_frexpf(mx2_, &e_);            /* 2^(e-1) <= mx2 < 2^e */                             \
            e_ = 14 - e_;                                                                                        \
            (es_) = e_ < -100 ? -100 : (e_ > 100 ? 100 : e_);                                                    \
        }                                                                                                        \
    } while (0)

    unsigned item = blockIdx.x;
#ifdef W9_TRACE
    int tr_item = 0;
#endif
    W9_STAMP(0);
    Coord cc_cur, cc_nxt;
    int es_cur;
    W9_COORD(cc_cur, item);
    W9_ITEM(st.cur, cc_cur);
    {
        int img_lane;
        W9_IMG_LANE(img_lane, cc_cur);
        W9_SCALE_EXP(es_cur, W9_XMAX_OF(img_lane));
    }
    st.cur.S = __builtin_ldexpf(1.f, es_cur);
    // the first item's prologue (afterwards the chunk stream itself fetches ahead): patches 0 / 1, weights and first V rows of chunk 0
    {   // all four half-patches are requested before the first is written (the fragment registers are still free: one memory latency, not four)
        u32x4 keep[3][NSTG];
        W9_STAMP(2);
        pload_all<0, UR>(st, st.cur, a, 0, up, HalfA{});
#pragma unroll
        for (int i = 0; i < NSTG; ++i) keep[0][i] = st.stg[i];
        pload_all<1, UR>(st, st.cur, a, 0, up, HalfB{});
#pragma unroll
        for (int i = 0; i < 5; ++i) keep[1][i] = st.stg[i];
        pload_all<0, UR>(st, st.cur, a, 1, up, HalfA{});
#pragma unroll
        for (int i = 0; i < NSTG; ++i) keep[2][i] = st.stg[i];
        pload_all<1, UR>(st, st.cur, a, 1, up, HalfB{});       // (half B of patch 1 stays in the staging registers: chunk 0 writes it at slice 28)
        u32x4 last[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) last[i] = st.stg[i];
#pragma unroll
        for (int i = 0; i < NSTG; ++i) st.stg[i] = keep[0][i];
        pwrite_all<0, UR>(st, 0, HalfA{});
#pragma unroll
        for (int i = 0; i < 5; ++i) st.stg[i] = keep[1][i];
        pwrite_all<1, UR>(st, 0, HalfB{});
#pragma unroll
        for (int i = 0; i < NSTG; ++i) st.stg[i] = keep[2][i];
        pwrite_all<0, UR>(st, 1, HalfA{});
#pragma unroll
        for (int i = 0; i < 5; ++i) st.stg[i] = last[i];
    }
    W9_STAMP(3);
#pragma unroll
    for (int i = 0; i < 4; ++i) load_b<0, UR ? 4 : 3>(st, a, st.cur.u_voff, 0, i, u_plane, u_wave);
#pragma unroll
    for (int i = 0; i < 4; ++i) load_b<1, UR ? 4 : 3>(st, a, st.cur.u_voff, 0, i, u_plane, u_wave);
    if constexpr (UR) {
#pragma unroll
        for (int i = 0; i < 4; ++i) load_b<2, 4>(st, a, st.cur.u_voff, 0, i, u_plane, u_wave);
    }
    W9_BARRIER();
    W9_STAMP(4);
    rread<0>(st, 0, 0); rread<1>(st, 0, 0); rread<2>(st, 0, 0); rread<3>(st, 0, 0);
    job_all(st, 0, st.cur.S, std::make_integer_sequence<int, 28>{});
    rread<0>(st, 0, 1); rread<1>(st, 0, 1); rread<2>(st, 0, 1); rread<3>(st, 0, 1);
    job_all(st, 1, st.cur.S, std::make_integer_sequence<int, 28>{});
    if constexpr (!UR) { rread<0>(st, 0, 2); rread<1>(st, 0, 2); rread<2>(st, 0, 2); rread<3>(st, 0, 2); }      // (UR: row 2 has no job ...)
    else if constexpr (W9_UR_COMPACT) { rread<0>(st, 0, 3); rread<1>(st, 0, 3); rread<2>(st, 0, 3); rread<3>(st, 0, 3); }     // ... the compact chunk's job 0 builds row 3
    while (true) {
        W9_STAMP(1);
#ifdef W9_TRACE
        st.trp = (blockIdx.x == 0 && tid == 0 && tr_item < 64) ? a.trace + tr_item * 32 : nullptr;
#endif
        // the item after this one (the last item of a workgroup names itself: its fetch-ahead then re-reads valid memory into dead buffers)
        const unsigned next = item + gridDim.x;
        const bool more = next < (unsigned)a.blocks;
        W9_COORD(cc_nxt, more ? next : item);
        W9_ITEM(st.nxt, cc_nxt);
        int img_next;
        W9_IMG_LANE(img_next, cc_nxt);
        const float xmax_next = W9_XMAX_OF(img_next);          // requested now, used behind the chunk loop
        {   // this item's bias and inverse weight scales: one value per lane, written to LDS inside the first chunk (slice 60) — the epilogue
            // then reads them with LDS latency instead of waiting ~1.5 K cycles for global loads
            const int co = cc_cur.n0 + lane_now();
            st.bst = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(__builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, (int)a.b_bytes, 0x00020000),
                                                                                    co < a.Cout ? (unsigned)co * 4u : OOB, 0, 0));
            st.ist = a.isu[co];
            if constexpr (FUSE) st.fwst = buf_load16(a.fw, (unsigned)a.CoutP * 16u, (unsigned)co * 16u, 0);
        }
        W9_STAMP(5);

        // the item's first chunk starts the accumulators from zero (first_use); CC = 2: that chunk is also the last but one
        int es_nxt;
        if (a.CC > 2) {
            chunk<0, 0, true, FUSE, UR>(st, a, 0, up, u_plane, u_wave);
            chunk<1, 0, false, false, UR>(st, a, 1, up, u_plane, u_wave);
            for (int cn = 2; cn < a.CC - 2; cn += 2) {
                chunk<0, 0, false, false, UR>(st, a, cn, up, u_plane, u_wave);
                chunk<1, 0, false, false, UR>(st, a, cn + 1, up, u_plane, u_wave);
            }
            // (the next item's maximum is first used HERE, behind the chunks: used at the top of the item it waits for its load, and with it
            //  for the previous item's stores)
            W9_SCALE_EXP(es_nxt, xmax_next);
            st.nxt.S = __builtin_ldexpf(1.f, es_nxt);
            chunk<0, 1, false, false, UR>(st, a, a.CC - 2, up, u_plane, u_wave);
        } else {
            W9_SCALE_EXP(es_nxt, xmax_next);
            st.nxt.S = __builtin_ldexpf(1.f, es_nxt);
            chunk<0, 1, true, FUSE, UR>(st, a, 0, up, u_plane, u_wave);
        }
        chunk<1, 2, false, false, UR>(st, a, a.CC - 1, up, u_plane, u_wave);
        W9_STAMP(6);

        // ---- epilogue: out0 = Y0 + Y1 + Y2, out1 = Y1 - Y2 - Y3; the four positions (waves) meet through LDS.  Pass j = output row j:
        // every wave writes its two blocks (cout halves) of that row into one half of the exchange region (32 KB, the halves alternate: one
        // barrier per pass) as [cout half][position][tile][16-byte piece = 4 couts], pieces XOR-swizzled by the tile (conflict-free both ways);
        // wave w finishes cout half w & 1 of tiles 16 (w >> 1) .. + 15 for all four positions with thread = (tile, piece): the 8 lanes of a
        // tile store one full 128-byte line (32 couts) of each of its two pixels.  (A first version stored 16 bytes per lane with the lanes
        // along the tiles — 64 quarter lines per instruction: the epilogue passes queued behind their own stores, 3-20 K cycles each.)
        // (The patch buffers already hold the next item's first two patches.) ----
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));        // the epilogue's per-lane constants are re-derived here (hoisted, they live across the chunk loop and spill)
        const int t_e = lane_e & 31, h_e = lane_e >> 5;
        const int g_e = wave & 1;
        const int piece_e = lane_e & 7;
        const int cout_e = cc_cur.n0 + g_e * 32 + piece_e * 4;            // this thread's four couts
        const bool cok_e = cout_e < a.Cout;
        const f32x4 bq = lds_f4(st.sB + (g_e * 32 + piece_e * 4) * 4);
        const f32x4 isu_e = lds_f4(st.sB + 256 + (g_e * 32 + piece_e * 4) * 4);
        // writer: lane (h, t) holds piece 2 q + h of tile t;  reader, iteration i: tile 16 (w >> 1) + 8 i + lane / 8, piece lane % 8
        const int wslot0 = t_e * 8, wsw = t_e & 7;
        int rtile[2], rslot[2], rimg[2], rpx[2];
        f32x4 iq[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            rtile[i] = 16 * (wave >> 1) + 8 * i + (lane_e >> 3);
            rslot[i] = rtile[i] * 8 + (piece_e ^ (rtile[i] & 7));
            // the image this tile belongs to (several side by side on narrow maps): its column there, its scale
            const int si = a.ipb > 1 ? ((2 * rtile[i]) >> a.lw) : 0;
            rimg[i] = cc_cur.n * a.ipb + si;
            rpx[i] = a.ipb > 1 ? ((2 * rtile[i]) & (a.W - 1)) : cc_cur.x0 + 2 * rtile[i];
            if constexpr (PK) {   // packed rows: the tile's virtual column -> (image, pixel); a strip's two padding columns (pixel >= W) are not stored
                unsigned q_, r_;
                W9_VDIVMOD(q_, r_, (unsigned)(cc_cur.x0 + 2 * rtile[i]), a.pk, a.m_pk);
                rimg[i] = (int)q_; rpx[i] = (int)r_;
            }
            if (a.ipb > 1 || PK) {        // the tile's image is not the one this lane builds V for: its scale from its maximum
                int es_i;
                W9_SCALE_EXP(es_i, W9_XMAX_OF(rimg[i]));
                iq[i] = isu_e * __builtin_ldexpf(1.f, -es_i);
            } else {              // one image per block row: 1 / S from the exponent of the scale in use (S = 2^e exactly)
                iq[i] = isu_e * __builtin_bit_cast(float, 0x7F000000u - __builtin_bit_cast(unsigned, st.cur.S));
            }
        }
        // the folded 1x1 conv: this thread's four couts x its (up to) four output channels, one row of fw per cout
        f32x4 wq[4];
        unsigned fv0[2];                 // byte offset of each tile's first pixel of row y0 in this thread's block of fpart
        if constexpr (FUSE) {
#pragma unroll
            for (int q = 0; q < 4; ++q) wq[q] = lds_f4(st.sB + 512 + (g_e * 32 + piece_e * 4 + q) * 16);       // (staged in the item's first chunk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                fv0[i] = (unsigned)((cc_cur.n0 >> 5) + g_e) * a.fp_block + (unsigned)((rimg[i] * a.H + cc_cur.y0) * a.W + rpx[i]) * 16u;
        }
        float omax2[2] = {0.f, 0.f};
        unsigned yv0[2];                 // byte offset of each tile's first pixel in output row y0 (row j: + j rows)
#pragma unroll
        for (int i = 0; i < 2; ++i) yv0[i] = ((unsigned)((rimg[i] * a.H + cc_cur.y0) * a.W + rpx[i]) * (unsigned)a.ldy + (unsigned)cout_e) * 4u;
        const unsigned y_row = (unsigned)(a.W * a.ldy) * 4u;
        // what the images' max |y| slots hold so far: requested HERE, ahead of the item's stores, read behind the last pass (cnl::peek_max)
        unsigned yseen[2] = {0u, 0u};
        if (a.ymax) {
#pragma unroll
            for (int i = 0; i < 2; ++i) yseen[i] = cnl::peek_max(a.ymax + (rimg[i] < a.Nimg ? rimg[i] : 0) * AMS);
        }
        // pass j + 1's blocks are written (into the other half) while pass j's are finished: ds_write_b128 costs 13 LDS cycles per wave
        // (MI355X_MICROARCH.md, LDS table) — eight in a row stall the wave behind the LDS queue (measured: 370 of a pass's 1200 cycles); two
        // per quarter of the arithmetic drain beside it
#define W9_XWRITE2(j_, g_, q0_)                                                                                  \
        if (!(UR && W9_UR_COMPACT && W9_UR_SKIP_ZERO) || !st.idle)          /* (the zero position has nothing to hand over) */ \
        _Pragma("unroll") for (int q = (q0_); q < (q0_) + 2; ++q) {                                              \
            const f32x16& A = st.acc[j_][g_];                                                                    \
            *reinterpret_cast<f32x4*>(sX + ((j_) & 1) * (X_BYTES / 2) + (((g_) * 4 + wave) * 256 + wslot0 + ((2 * q + h_e) ^ wsw)) * 16) = \
                f32x4{A[4 * q], A[4 * q + 1], A[4 * q + 2], A[4 * q + 3]};                                       \
        }
#define W9_XWRITE(j_) do { W9_XWRITE2(j_, 0, 0); W9_XWRITE2(j_, 0, 2); W9_XWRITE2(j_, 1, 0); W9_XWRITE2(j_, 1, 2); } while (0)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 iql[2] = {f32x2{iq[0][0], iq[0][1]}, f32x2{iq[1][0], iq[1][1]}}, iqh[2] = {f32x2{iq[0][2], iq[0][3]}, f32x2{iq[1][2], iq[1][3]}};
        const f32x2 bql = {bq[0], bq[1]}, bqh = {bq[2], bq[3]};
        f32x4 rvn[2][2];                 // RES: the residual values of the next pass's row
        if constexpr (RES) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool ok0 = cc_cur.y0 < a.H && cok_e && rimg[i] < a.Nimg;
                const unsigned rvo = ((unsigned)((rimg[i] * a.H + cc_cur.y0) * a.W + rpx[i]) * (unsigned)a.ldr + (unsigned)cout_e) * 4u;
#pragma unroll
                for (int px = 0; px < 2; ++px)
                    rvn[i][px] = __builtin_bit_cast(f32x4, buf_load16(a.res, a.r_bytes, (ok0 && rpx[i] + px < a.W) ? rvo : OOB, (unsigned)(px * a.ldr * 4)));
            }
        }
        W9_XWRITE(0);
        W9_BARRIER();
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const char* X = sX + (j & 1) * (X_BYTES / 2);
            const int oy = cc_cur.y0 + j;
            const bool row_ok = oy < a.H && cok_e;
            unsigned yv[2];
            bool ok[2][2];
            f32x4 rv[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ox = rpx[i];
                yv[i] = yv0[i] + (unsigned)j * y_row;
                ok[i][0] = row_ok && ox < a.W && rimg[i] < a.Nimg; ok[i][1] = row_ok && ox + 1 < a.W && rimg[i] < a.Nimg;
                if constexpr (RES) {
                    // the residual of THIS row was requested a pass ago (round 5: requested at the top of its own pass, every pass waited a memory
                    // latency for it — layer1's conv2 launches 15-20 us behind their conv1 twins); the next row's goes out now
#pragma unroll
                    for (int px = 0; px < 2; ++px) rv[i][px] = rvn[i][px];
                    if (j + 1 < R) {
                        const bool okn = oy + 1 < a.H && cok_e && rimg[i] < a.Nimg;
                        const unsigned rvo = ((unsigned)((rimg[i] * a.H + oy + 1) * a.W + ox) * (unsigned)a.ldr + (unsigned)cout_e) * 4u;
#pragma unroll
                        for (int px = 0; px < 2; ++px)
                            rvn[i][px] = __builtin_bit_cast(f32x4, buf_load16(a.res, a.r_bytes, (okn && ox + px < a.W) ? rvo : OOB, (unsigned)(px * a.ldr * 4)));
                    }
                }
            }
            W9_STAMP(7 + (j & 7));
            f32x4 Y[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    if constexpr (UR && W9_UR_COMPACT && W9_UR_SKIP_ZERO) {
                        if (p == 2) { Y[i][p] = f32x4{0.f, 0.f, 0.f, 0.f}; continue; }
                    }
                    Y[i][p] = lds_f4(X + ((g_e * 4 + p) * 256 + rslot[i]) * 16);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                // packed fp32 arithmetic (no MFMA in flight here): two couts per instruction
                f32x4 o0, o1;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const f32x2 y0 = {Y[i][0][2 * hh], Y[i][0][2 * hh + 1]}, y1 = {Y[i][1][2 * hh], Y[i][1][2 * hh + 1]};
                    const f32x2 y2 = {Y[i][2][2 * hh], Y[i][2][2 * hh + 1]}, y3 = {Y[i][3][2 * hh], Y[i][3][2 * hh + 1]};
                    const f32x2 sc = hh ? iqh[i] : iql[i], bb = hh ? bqh : bql;
                    f32x2 ya = (y0 + y1 + y2) * sc + bb;
                    f32x2 yb = (y1 - y2 - y3) * sc + bb;
                    if constexpr (RES) {
                        ya += f32x2{rv[i][0][2 * hh], rv[i][0][2 * hh + 1]};
                        yb += f32x2{rv[i][1][2 * hh], rv[i][1][2 * hh + 1]};
                    }
                    o0[2 * hh] = fmaxf(ya[0], lo); o0[2 * hh + 1] = fmaxf(ya[1], lo);
                    o1[2 * hh] = fmaxf(yb[0], lo); o1[2 * hh + 1] = fmaxf(yb[1], lo);
                    if (j + 1 < R) {
                        if (i == 0 && hh == 0) { W9_XWRITE2(j + 1, 0, 0); }
                        if (i == 0 && hh == 1) { W9_XWRITE2(j + 1, 0, 2); }
                        if (i == 1 && hh == 0) { W9_XWRITE2(j + 1, 1, 0); }
                        if (i == 1 && hh == 1) { W9_XWRITE2(j + 1, 1, 2); }
                    }
                }
                if (ok[i][0]) omax2[i] = fmaxf(omax2[i], fmaxf(fmaxf(fabsf(o0[0]), fabsf(o0[1])), fmaxf(fabsf(o0[2]), fabsf(o0[3]))));
                if (ok[i][1]) omax2[i] = fmaxf(omax2[i], fmaxf(fmaxf(fabsf(o1[0]), fabsf(o1[1])), fmaxf(fabsf(o1[2]), fabsf(o1[3]))));
                buf_store16(o0, a.y, a.y_bytes, ok[i][0] ? yv[i] : OOB, 0);
                buf_store16(o1, a.y, a.y_bytes, ok[i][1] ? yv[i] : OOB, (unsigned)(a.ldy * 4));
                if constexpr (FUSE) {
                    // d[c] = sum over this thread's four couts of out[co] * fw[co][c], then over the 8 lanes (pieces) of the tile: quad
                    // neighbours, quad pairs, the two quads of the half row — a fixed tree, the same in every lane; lane piece 0 stores the
                    // 32-cout partial sums of both pixels (16 bytes each)
                    float e0[4], e1[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        e0[c] = __builtin_fmaf(o0[3], wq[3][c], __builtin_fmaf(o0[2], wq[2][c], __builtin_fmaf(o0[1], wq[1][c], o0[0] * wq[0][c])));
                        e1[c] = __builtin_fmaf(o1[3], wq[3][c], __builtin_fmaf(o1[2], wq[2][c], __builtin_fmaf(o1[1], wq[1][c], o1[0] * wq[0][c])));
                    }
#define W9_DPP_OF(v_, ctrl_) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v_)), (ctrl_), 0xF, 0xF, false))
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        e0[c] = e0[c] + W9_DPP_OF(e0[c], 0xB1); e1[c] = e1[c] + W9_DPP_OF(e1[c], 0xB1);        // quad_perm [1, 0, 3, 2]
                        e0[c] = e0[c] + W9_DPP_OF(e0[c], 0x4E); e1[c] = e1[c] + W9_DPP_OF(e1[c], 0x4E);        // quad_perm [2, 3, 0, 1]
                        e0[c] = e0[c] + W9_DPP_OF(e0[c], 0x141); e1[c] = e1[c] + W9_DPP_OF(e1[c], 0x141);      // row_half_mirror: the other quad of the 8 lanes
                    }
#undef W9_DPP_OF
                    const f32x4 d0 = {e0[0], e0[1], e0[2], e0[3]}, d1 = {e1[0], e1[1], e1[2], e1[3]};
                    const unsigned fv = fv0[i] + (unsigned)j * (unsigned)(a.W * 16);
                    const bool do_store = piece_e == 0 && oy < a.H && rimg[i] < a.Nimg;      // (also for blocks of couts >= Cout: their weights are zero)
                    // (default cache policy, not nt: eight lanes of a wave fill a 128-byte line of fpart together with their neighbours' stores)
                    {
                        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.fpart, 0, (int)a.fp_bytes, 0x00020000);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, d0), rs, (do_store && rpx[i] < a.W) ? fv : OOB, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, d1), rs, (do_store && rpx[i] + 1 < a.W) ? fv : OOB, 16, 0);
                    }
                }
            }
            if (j + 1 < R) { W9_BARRIER(); }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef W9_XWRITE
#undef W9_XWRITE2
        if (a.ymax) {          // max |y| of this item into its image's slot: the 8 tiles of an iteration lie in one image (tiles per image: 8, 16 or all)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int img = __builtin_amdgcn_readfirstlane(rimg[i]);
                if constexpr (PK) {    // packed rows: the 8 tiles may lie in two neighbouring strips (pk >= 16): the first tile's image and the one behind it
                    const float m1 = cnl::wave_max_nonneg(rimg[i] != img ? omax2[i] : 0.f);
                    if (lane_e == 0 && img + 1 < a.Nimg) cnl::report_max(a.ymax + (img + 1) * AMS, m1);
                    omax2[i] = rimg[i] == img ? omax2[i] : 0.f;
                }
                const float m = cnl::wave_max_nonneg(omax2[i]);
                if (lane_e == 0 && img < a.Nimg) cnl::raise_max(a.ymax + img * AMS, m, (unsigned)__builtin_amdgcn_readfirstlane((int)yseen[i]));
            }
        }
        W9_STAMP(15);
#ifdef W9_TRACE
        ++tr_item;
#endif
        if (!more) break;
        item = next;
        cc_cur = cc_nxt;
        es_cur = es_nxt;
        st.cur = st.nxt;
    }
#undef W9_COORD
#undef W9_DIVMOD
#undef W9_VDIVMOD
#undef W9_IMG_LANE
#undef W9_ITEM
#undef W9_SCALE_EXP
#undef W9_XMAX_OF
}

// fp32 OHWI 3x3 weights -> U_p[ky] = (G g[ky])_p per (co, ci), scaled per OUTPUT CHANNEL by S_u[co] = 2^(13 - e) (max |U[co]| = m 2^e)
// and split into two fp16 pieces: [ci/16][p][ky][piece][CoutP][16 ci]; isu[co] = 1 / S_u[co].  One workgroup per output channel.
__global__ __launch_bounds__(256) void weights9_kernel(const float* __restrict__ w, unsigned short* __restrict__ u9, float* __restrict__ isu,
                                                       int Cin, int Cout, int CoutP) {
    const int co = blockIdx.x;
    __shared__ float wm[4];
    float m = 0.f;
    if (co < Cout)
        for (int ci = threadIdx.x; ci < Cin; ci += 256)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float g0 = w[((long)co * 9 + ky * 3 + 0) * Cin + ci], g1 = w[((long)co * 9 + ky * 3 + 1) * Cin + ci],
                            g2 = w[((long)co * 9 + ky * 3 + 2) * Cin + ci];
                const float u1 = 0.5f * (g0 + g1 + g2), u2 = 0.5f * (g0 - g1 + g2);
                m = fmaxf(m, fmaxf(fmaxf(fabsf(g0), fabsf(g2)), fmaxf(fabsf(u1), fabsf(u2))));
            }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    float Su = 1.f;
    if (m > 0.f && m < __builtin_inff()) {
        int e_;
        (void)__builtin_frexpf(m, &e_);
        e_ = 13 - e_;
        Su = __builtin_ldexpf(1.f, e_ < -100 ? -100 : (e_ > 100 ? 100 : e_));
    }
    if (threadIdx.x == 0) isu[co] = 1.f / Su;
    for (int ci = threadIdx.x; ci < Cin; ci += 256) {
        const int cc = ci >> 4, c16 = ci & 15;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            float g0 = 0.f, g1 = 0.f, g2 = 0.f;
            if (co < Cout) {
                g0 = w[((long)co * 9 + ky * 3 + 0) * Cin + ci]; g1 = w[((long)co * 9 + ky * 3 + 1) * Cin + ci];
                g2 = w[((long)co * 9 + ky * 3 + 2) * Cin + ci];
            }
            const float uu[4] = {-g0, 0.5f * (g0 + g1 + g2), 0.5f * (g0 - g1 + g2), g2};      // position 0 negated: the kernel forms -(d0 - d2) there
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float xs = uu[p] * Su;
                const _Float16 hf = (_Float16)xs;
                const _Float16 lf = (_Float16)(xs - (float)hf);
                const long base = (((((long)cc * 4 + p) * 3 + ky) * 2) * CoutP + co) * 16 + c16;
                u9[base] = __builtin_bit_cast(unsigned short, hf);
                u9[base + (long)CoutP * 16] = __builtin_bit_cast(unsigned short, lf);
            }
        }
    }
}

// The row-pair weight sets of a conv consumed behind a nearest-2x upsample (UR): per (co, ci) the kernel rows g[0], g[0] + g[1], g[1] + g[2], g[2] (pre-summed in fp32),
// each transformed along x like weights9_kernel's, one power-of-two scale per output channel over all four sets:
// [ci/16][p][set][piece][CoutP][16 ci] fp16, isu[co] = 1 / S_u[co].  One workgroup per output channel.
__global__ __launch_bounds__(256) void weights9_up_kernel(const float* __restrict__ w, unsigned short* __restrict__ u9, float* __restrict__ isu,
                                                          int Cin, int Cout, int CoutP) {
    const int co = blockIdx.x;
    __shared__ float wm[4];
    float m = 0.f;
    const auto rows = [&](int ci, float (&c)[4][3]) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const float g0 = w[((long)co * 9 + 0 * 3 + kx) * Cin + ci], g1 = w[((long)co * 9 + 1 * 3 + kx) * Cin + ci], g2 = w[((long)co * 9 + 2 * 3 + kx) * Cin + ci];
            c[0][kx] = g0; c[1][kx] = g0 + g1; c[2][kx] = g1 + g2; c[3][kx] = g2;
        }
    };
    if (co < Cout)
        for (int ci = threadIdx.x; ci < Cin; ci += 256) {
            float c[4][3];
            rows(ci, c);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float u1 = 0.5f * (c[s][0] + c[s][1] + c[s][2]), u2 = 0.5f * (c[s][0] - c[s][1] + c[s][2]);
                m = fmaxf(m, fmaxf(fmaxf(fabsf(c[s][0]), fabsf(c[s][2])), fmaxf(fabsf(u1), fabsf(u2))));
            }
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    float Su = 1.f;
    if (m > 0.f && m < __builtin_inff()) {
        int e_;
        (void)__builtin_frexpf(m, &e_);
        e_ = 13 - e_;
        Su = __builtin_ldexpf(1.f, e_ < -100 ? -100 : (e_ > 100 ? 100 : e_));
    }
    if (threadIdx.x == 0) isu[co] = 1.f / Su;
    for (int ci = threadIdx.x; ci < Cin; ci += 256) {
        const int cc = ci >> 4, c16 = ci & 15;
        float c[4][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
        if (co < Cout) rows(ci, c);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float uu[4] = {-c[s][0], 0.5f * (c[s][0] + c[s][1] + c[s][2]), 0.5f * (c[s][0] - c[s][1] + c[s][2]), c[s][2]};      // position 0 negated (as weights9_kernel)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float xs = uu[p] * Su;
                const _Float16 hf = (_Float16)xs;
                const _Float16 lf = (_Float16)(xs - (float)hf);
                const long base = (((((long)cc * 4 + p) * 4 + s) * 2) * CoutP + co) * 16 + c16;
                u9[base] = __builtin_bit_cast(unsigned short, hf);
                u9[base + (long)CoutP * 16] = __builtin_bit_cast(unsigned short, lf);
            }
        }
    }
}

}  // namespace cnl_wino9

// the row-pair weight sets (UR launches): bytes of the pieces, floats of the per-cout scales behind them (cnl_winograd_up_weight_floats = both)
size_t cnl_wino9_up_weight_bytes(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0 || Cin % 32) return 0;
    const size_t CoutP = (size_t)((Cout + 63) / 64) * 64;
    return (size_t)(Cin / 16) * 4 * 4 * 2 * CoutP * 32;
}
int cnl_wino9_up_transform_weights(const float* w_ohwi, void* u9, float* isu, int Cin, int Cout, void* stream) {
    using namespace cnl_wino9;
    const int CoutP = (Cout + 63) / 64 * 64;
    hipLaunchKernelGGL(weights9_up_kernel, dim3((unsigned)CoutP), dim3(256), 0, (hipStream_t)stream, w_ohwi, (unsigned short*)u9, isu, Cin, Cout, CoutP);
    return cnl::check_launch("weights9_up_kernel");
}

// bytes of this kernel's fp16-split weights (0 when it does not apply) and floats of the per-cout scales behind them
size_t cnl_wino9_weight_bytes(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0 || Cin % 32) return 0;
    const size_t CoutP = (size_t)((Cout + 63) / 64) * 64;
    return (size_t)(Cin / 16) * 4 * 3 * 2 * CoutP * 32;
}
size_t cnl_wino9_scalar_floats(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0 || Cin % 32) return 0;
    return (size_t)((Cout + 63) / 64) * 64;
}

int cnl_wino9_transform_weights(const float* w_ohwi, void* u9, float* isu, int Cin, int Cout, void* stream) {
    using namespace cnl_wino9;
    const int CoutP = (Cout + 63) / 64 * 64;
    hipLaunchKernelGGL(weights9_kernel, dim3((unsigned)CoutP), dim3(256), 0, (hipStream_t)stream, w_ohwi, (unsigned short*)u9, isu, Cin, Cout, CoutP);
    return cnl::check_launch("weights9_kernel");
}

#ifdef W9_TRACE
static unsigned long long* g_w9_trace = nullptr;
extern "C" __attribute__((visibility("default"))) void cnl_w9_set_trace(void* p) { g_w9_trace = (unsigned long long*)p; }
#endif
// can this kernel run the layer at all?  (shape / alignment only)
bool cnl_wino9_eligible(const cnl_conv_params* p) {
    return p->Cin % 32 == 0 && p->Cin >= 32 && p->Cout % 4 == 0 && p->ldy % 4 == 0 && ((uintptr_t)p->y & 15) == 0 &&
           (!p->residual || (p->ldr % 4 == 0 && ((uintptr_t)p->residual & 15) == 0));
}

// Packed rows (winograd9.hip / winograd10.hip): maps whose width is not a multiple of the 64-pixel block row — the 19 x 34, 38 x 68, 76 x 136 and
// 152 x 272 maps of 608 x 1088 frames (reference datasets/utils.py:29-33: the MOT input shape) — would leave up to half of every block row empty.
// There the launch's images are laid side by side in ONE virtual row, image n in the strip [n pk, n pk + W) with pk = W + 2: the two columns
// behind an image ARE its right and the next image's left zero padding (out-of-range loads), so no tile ever sees a neighbour's pixels and
// nothing is masked; a block row is any 64 consecutive virtual columns (W even: strips start on tile boundaries).  One tile in W/2 + 1 is spent
// on the padding columns instead of up to one block in two.  Every output keeps its chain of fp32 additions and its image's scale: the
// result is bit-identical to the unpacked form (tests/test_gpu_conv.py), so — like the work-item shape — the choice may look at N.
// Returns pk, or 0 where the plain block grid is at least as good (W a multiple of 64, the two / four-images-per-block forms of W = 32 / 16).
int cnl_wino_packed_stride(const cnl_conv_params* p) {
    const int upf = (p->flags & CNL_UPSAMPLE_IN) ? 2 : 1, W = p->W_in * upf;       // (a folded nearest-2x upsample: the logical width)
    if (W % 2 || W < 14 || (upf == 1 && (W == 32 || W == 16))) return 0;
    if (p->algo >= CNL_ALGO_FORCE + 32) return 0;          // tests: FORCE + 32 + v = variant v (9 / 10 / 11) on the plain block grid
    const long long pk = W + 2;
    const long long plain = (long long)p->N * ((W + 63) / 64), packed = ((long long)p->N * pk + 63) / 64;
    return packed < plain ? (int)pk : 0;
}

// The kernels address each tensor through 32-bit buffer offsets (< 4 GiB per launch).  A launch whose input, output or residual spans more —
// the first blocks of three heads fused along Cout on 32 frames of 608 x 1088: 32 x 152 x 272 x 768 floats = 4.06 GB — runs as the fewest equal
// groups of images that fit, one kernel launch each (images are independent and keep their own scale: the same bits as one launch).
// Returns the images per launch (N when everything fits), 0 when a single image does not fit.
int cnl_wino_images_per_launch(const cnl_conv_params* p) {
    const int upf = (p->flags & CNL_UPSAMPLE_IN) ? 2 : 1;
    const unsigned long long lim = 0xFFFFFF00ull - 4096ull * 4ull;      // (the span checks of the launchers keep a pixel of slack)
    const unsigned long long xi = (unsigned long long)p->H_in * p->W_in * p->ldx * 4ull;
    const unsigned long long yi = (unsigned long long)p->H_in * upf * p->W_in * upf * p->ldy * 4ull;
    const unsigned long long ri = p->residual ? (unsigned long long)p->H_in * upf * p->W_in * upf * p->ldr * 4ull : 0ull;
    const unsigned long long worst = xi > yi ? (xi > ri ? xi : ri) : (yi > ri ? yi : ri);
    const unsigned long long nmax = lim / (worst + 4ull * (unsigned long long)(p->ldy > p->ldr ? p->ldy : p->ldr));
    if (nmax == 0) return 0;
    if ((unsigned long long)p->N <= nmax) return p->N;
    const unsigned long long parts = ((unsigned long long)p->N + nmax - 1) / nmax;
    return (int)(((unsigned long long)p->N + parts - 1) / parts);
}
// sub-batch [n0, n0 + n) of launch p: pointers moved to its first image (64-bit host arithmetic), per-image maxima with them
void cnl_wino_sub_batch(const cnl_conv_params* p, int n0, int n, cnl_conv_params* q, const float** xmax) {
    const int upf = (p->flags & CNL_UPSAMPLE_IN) ? 2 : 1;
    *q = *p;
    q->N = n;
    q->x = p->x + (size_t)n0 * p->H_in * p->W_in * p->ldx;
    q->y = p->y + (size_t)n0 * p->H_in * upf * p->W_in * upf * p->ldy;
    if (p->residual) q->residual = p->residual + (size_t)n0 * p->H_in * upf * p->W_in * upf * p->ldr;
    if (p->y_absmax) q->y_absmax = p->y_absmax + (size_t)n0 * AMS;
    if (p->x_absmax) q->x_absmax = p->x_absmax + (size_t)n0 * AMS;
    *xmax += (size_t)n0 * AMS;
}

static int wino9_launch_one(const cnl_conv_params* p, const void* u9, const float* isu, const float* xmax, void* stream);
// can this launch take the row-pair form (UR)?  A folded upsample, the row-pair weights given, nothing the UR instantiations lack (residual, folded 1x1)
bool cnl_wino9_up_rows(const cnl_conv_params* p) {
    return (p->flags & CNL_UPSAMPLE_IN) && p->w_up && !p->residual && !p->fuse_w && ((uintptr_t)p->w_up & 15) == 0;
}
// Launch (arguments already validated by cnl_conv3x3_winograd_f32); xmax = N per-image maxima of the input.
int cnl_wino9_launch(const cnl_conv_params* p, const void* u9, const float* isu, const float* xmax, void* stream) {
    const int per = cnl_wino_images_per_launch(p);
    CNL_REQUIRE(per > 0, CNL_E_UNSUPPORTED, "cnl_conv3x3_winograd_f32: one image of a tensor spans >= 4 GiB");
    if (per >= p->N) return wino9_launch_one(p, u9, isu, xmax, stream);
    CNL_REQUIRE(!p->fuse_w, CNL_E_UNSUPPORTED, "cnl_conv3x3_winograd_f32: fuse_w on a launch whose tensors span >= 4 GiB; split the batch");
    for (int n0 = 0; n0 < p->N; n0 += per) {
        cnl_conv_params q;
        const float* xm = xmax;
        cnl_wino_sub_batch(p, n0, p->N - n0 < per ? p->N - n0 : per, &q, &xm);
        const int rc = wino9_launch_one(&q, u9, isu, xm, stream);
        if (rc != CNL_OK) return rc;
    }
    return CNL_OK;
}
static int wino9_launch_one(const cnl_conv_params* p, const void* u9, const float* isu, const float* xmax, void* stream) {
    using namespace cnl_wino9;
    Args a;
    const bool ur = cnl_wino9_up_rows(p);
    if (ur) {               // the row-pair sets and their scales instead of the layer's three kernel rows
        u9 = p->w_up;
        isu = p->w_up + cnl_wino9_up_weight_bytes(p->Cin, p->Cout) / 4;
    }
    a.x = p->x; a.u9 = u9; a.xmax = xmax; a.isu = isu; a.ymax = reinterpret_cast<unsigned*>(p->y_absmax);
    a.bias = p->bias; a.res = p->residual; a.y = p->y;
    const int upf = (p->flags & CNL_UPSAMPLE_IN) ? 2 : 1;
    a.Nimg = p->N; a.Hs = p->H_in; a.Ws = p->W_in; a.H = p->H_in * upf; a.W = p->W_in * upf; a.Cin = p->Cin; a.Cout = p->Cout;
    // narrow maps: 2 (W = 32) or 4 (W = 16) images side by side in one 64-pixel block row (no folded upsample there)
    a.ipb = (upf == 1 && (a.W == 32 || a.W == 16)) ? 64 / a.W : 1;
    a.lw = a.W == 32 ? 5 : 4;
    a.N = (p->N + a.ipb - 1) / a.ipb;
    // other widths that 64-pixel blocks pad: packed rows (cnl_wino_packed_stride) — same arithmetic chain per output, same bits
    a.pk = cnl_wino_packed_stride(p);
    a.m_pk = a.pk ? (unsigned)(0x100000000ull / (unsigned)a.pk) : 0u;
    if (a.pk) a.N = 1;
    a.CoutP = (p->Cout + 63) / 64 * 64;
    a.ldx = p->ldx; a.ldy = p->ldy; a.ldr = p->ldr;
    a.CC = p->Cin / 16;
    a.nb = a.CoutP / BN; a.bx = a.pk ? (int)(((long long)p->N * a.pk + 2 * TW - 1) / (2 * TW)) : (a.W + 2 * TW - 1) / (2 * TW); a.by = (a.H + R - 1) / R;
    const auto magic = [](int d) { return d == 1 ? 0xFFFFFFFFu : (unsigned)(0x100000000ull / (unsigned)d); };
    a.m_nb = magic(a.nb); a.m_bx = magic(a.bx); a.m_by = magic(a.by);
    const long long blocks = (long long)a.N * a.by * a.bx * a.nb;
    CNL_REQUIRE(blocks < (1ll << 31), CNL_E_UNSUPPORTED, "cnl_conv3x3_winograd_f32: grid too large");
    a.blocks = (int)blocks;
    const unsigned long long xb = (((unsigned long long)p->N * p->H_in * p->W_in - 1) * p->ldx + p->Cin) * 4ull;
    const unsigned long long ub = (unsigned long long)(ur ? cnl_wino9_up_weight_bytes(p->Cin, p->Cout) : cnl_wino9_weight_bytes(p->Cin, p->Cout));
    const unsigned long long Mo = (unsigned long long)p->N * a.H * a.W;
    const unsigned long long yb = ((Mo - 1) * p->ldy + p->Cout) * 4ull;
    const unsigned long long rb = p->residual ? ((Mo - 1) * p->ldr + p->Cout) * 4ull : 0ull;
    CNL_REQUIRE(xb < 0xFFFFFF00ull && ub < 0xFFFFFF00ull && yb + 4ull * p->ldy < 0xFFFFFF00ull && rb + 4ull * (p->residual ? p->ldr : 0) < 0xFFFFFF00ull,
                CNL_E_UNSUPPORTED, "cnl_conv3x3_winograd_f32: tensor spans >= 4 GiB; split the batch");
    a.x_bytes = (unsigned)xb; a.u_bytes = (unsigned)ub; a.y_bytes = (unsigned)yb; a.r_bytes = (unsigned)rb; a.b_bytes = (unsigned)p->Cout * 4u;
    a.flags = p->flags;
    a.fw = p->fuse_w; a.fpart = p->fuse_w ? p->fuse_part : nullptr; a.fp_bytes = a.fp_block = 0;
    if (a.fpart) {
        const unsigned long long blk = Mo * 16ull, all = blk * (unsigned long long)(a.CoutP / 32);
        CNL_REQUIRE(all < 0xFFFFFF00ull, CNL_E_UNSUPPORTED, "cnl_conv3x3_winograd_f32: fuse_part spans >= 4 GiB; split the batch");
        a.fp_bytes = (unsigned)all; a.fp_block = (unsigned)blk;
    }
#ifdef W9_TRACE
    a.trace = g_w9_trace;
#endif
    CNL_REQUIRE(!(a.fpart && p->residual), CNL_E_UNSUPPORTED, "cnl_conv3x3_winograd_f32: fuse_w with a residual");
    // the eight instantiations: (plain | residual | folded 1x1) x (plain grid | packed rows) + the row-pair form behind a folded upsample x (plain grid | packed rows)
    static cnl::DeviceOnce once[8];
    const void* const fns[8] = {reinterpret_cast<const void*>(&winograd9_kernel<false, false, false>), reinterpret_cast<const void*>(&winograd9_kernel<true, false, false>),
                                reinterpret_cast<const void*>(&winograd9_kernel<false, true, false>), reinterpret_cast<const void*>(&winograd9_kernel<false, false, true>),
                                reinterpret_cast<const void*>(&winograd9_kernel<true, false, true>), reinterpret_cast<const void*>(&winograd9_kernel<false, true, true>),
                                reinterpret_cast<const void*>(&winograd9_kernel<false, false, false, true>), reinterpret_cast<const void*>(&winograd9_kernel<false, false, true, true>)};
    const int which = ur ? (a.pk ? 7 : 6) : (a.fpart ? 2 : (p->residual ? 1 : 0)) + (a.pk ? 3 : 0);
    int n_cu = 0;                          // persistent workgroups: one per CU, walking the work items with stride gridDim.x
    int rc = cnl::kernel_setup(once[which], fns[which], LDS_BYTES, &n_cu);
    if (rc != CNL_OK) return rc;
#ifdef W9_MAX_CUS      // experiment builds: the persistent grid on fewer CUs (is a power-bound launch any slower on 240 of 256?)
    if (n_cu > W9_MAX_CUS) n_cu = W9_MAX_CUS;
#endif
    const unsigned grid = (unsigned)(blocks < (long long)n_cu ? blocks : (long long)n_cu);
    switch (which) {
    case 0: hipLaunchKernelGGL((winograd9_kernel<false, false, false>), dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, a); break;
    case 1: hipLaunchKernelGGL((winograd9_kernel<true, false, false>), dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, a); break;
    case 2: hipLaunchKernelGGL((winograd9_kernel<false, true, false>), dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, a); break;
    case 3: hipLaunchKernelGGL((winograd9_kernel<false, false, true>), dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, a); break;
    case 4: hipLaunchKernelGGL((winograd9_kernel<true, false, true>), dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, a); break;
    case 5: hipLaunchKernelGGL((winograd9_kernel<false, true, true>), dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, a); break;
    case 6: hipLaunchKernelGGL((winograd9_kernel<false, false, false, true>), dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, a); break;
    default: hipLaunchKernelGGL((winograd9_kernel<false, false, true, true>), dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, a); break;
    }
    return cnl::check_launch("winograd9_kernel");
}
