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
namespace cnl_wino9 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct Args {
    const float* x;
    const void* u9;                   // pre-split, pre-scaled weights (fp16 pieces)
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
    int blocks;
    unsigned x_bytes, u_bytes, y_bytes, r_bytes, b_bytes;
    unsigned flags;
#ifdef W9_TRACE
    unsigned long long* trace;        // timing build: [item][16] s_memtime stamps of block 0 / thread 0
#endif
};
#ifdef W9_TRACE
#define W9_STAMP(i_) do { if (blockIdx.x == 0 && tid == 0 && tr_item < 64) a.trace[tr_item * 16 + (i_)] = __builtin_readcyclecounter(); } while (0)
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
constexpr int LDS_BYTES = 2 * P_BYTES + X_BYTES;     // 151552: one workgroup per CU (the accumulators allow no more)
constexpr int NSLICE = 144;                 // MFMAs per wave and chunk
constexpr int JOB_SLICES = 14;              // one V fragment (28 VALU operations) is produced beside 14 MFMAs
constexpr int JOB0 = 2;                     // job j runs in slices [JOB0 + 14 j, JOB0 + 14 j + 14)
constexpr int BARRIER_SLICE = 98;           // before job 7 (the first to read the next patch)
constexpr int NSTG = 11;                    // staging pieces per thread and patch: 10 x (row i, pixel tid / 4, quad tid % 4) + the two last pixel columns

__device__ __forceinline__ u32x4 buf_load16(const void* base, unsigned bytes, unsigned voffset, unsigned soffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    return (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, soffset, 0);
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
__device__ __forceinline__ f32x4 lds_f4(const char* p) { return *reinterpret_cast<const f32x4*>(p); }
#define W9_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// ---- the static schedule of a chunk -----------------------------------------------------------------------------------------
// 24 segments of 6 MFMAs (3 terms x 2 cout halves), each one (input row r, kernel row ky) -> output row r - ky.  Rows in order,
// ky-major inside a row; the tail interleaves rows 7-9 so that the ky = 0 fragments die at slice 114 and the ky = 1 fragments at 126.
constexpr int SEG_ROW[24] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5, 6, 6, 6, 7, 7, 8, 7, 8, 9};
constexpr int SEG_KY[24] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 1, 1, 2, 2, 2};
constexpr int KY0_DEAD = 114, KY1_DEAD = 126;      // first slices after the last use of the ky = 0 / ky = 1 weight fragments

struct State {
    f32x16 acc[R][2];        // [output row][cout half]: D[cout][tile]
    u32x4 fb[3][2][2];       // weight fragments (A operand): [ky][cout half][piece], single-buffered
    u32x4 vf[4][2];          // V fragments (B operand): [(10 chunk + row) % 4][piece]
    f32x4 raw[4];            // patch reads of a job: pixel a quad 0, a quad 1, pixel b quad 0, b quad 1 (consumed by operations 0..7, refilled for the next job right after)
    float v[8];              // transform temporaries of the running job (V, then its residual in place)
    u32x4 stg[NSTG];         // patch pieces on their way global -> LDS
    unsigned vcol, vext;     // source offsets: column part of pieces 0..9 (the row is a scalar offset), full offset of piece 10
    unsigned u_voff;
    unsigned img_base, row_pitch;   // scalars: byte offset of image n, bytes per stored input row
    int y0m1;                       // y0 - 1: first patch row
    const char* pa[2];       // LDS address of this lane's pixel a / b in patch buffer 0 / 1
    const char* pb[2];
    char* wb;                // LDS write address of piece 0 in buffer 0 (piece i: + i rows), and of piece 10
    char* wext;
    float sg, S;
};

// VALU operation o (0..27) of the job that builds V fragment `buf`
template <int O>
__device__ __forceinline__ void vop(State& st, const int buf) {
    if constexpr (O < 8) {
        st.v[O] = __builtin_fmaf(st.raw[2 + (O >> 2)][O & 3], st.sg, st.raw[O >> 2][O & 3]);
    } else if constexpr (O < 12) {
        st.vf[buf][0][O - 8] = split_hi_lo(st.v[2 * (O - 8)], st.S);
    } else if constexpr (O < 16) {
        st.vf[buf][0][O - 12] = split_hi_hi(st.vf[buf][0][O - 12], st.v[2 * (O - 12) + 1], st.S);
    } else if constexpr (O < 24) {
        constexpr int e = O - 16;
        st.v[e] = (e & 1) ? split_res_hi(st.v[e], st.S, st.vf[buf][0][e >> 1]) : split_res_lo(st.v[e], st.S, st.vf[buf][0][e >> 1]);
    } else {
        constexpr int j = O - 24;
        st.vf[buf][1][j] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(st.v[2 * j], st.v[2 * j + 1]));
    }
}
// LDS read i (0..3) of patch row `row` of buffer `pbuf`
template <int I>
__device__ __forceinline__ void rread(State& st, const int pbuf, const int row) {
    const char* p = (I < 2 ? st.pa[pbuf] : st.pb[pbuf]) + row * ROW_BYTES + (I & 1) * (QUAD_SLOTS * 16);
    st.raw[I] = lds_f4(p);
}
// weight fragment i (cout half i >> 1, piece i & 1) of kernel row KY of chunk cc: global -> registers
template <int KY>
__device__ __forceinline__ void load_b(State& st, const Args& a, const int cc, const int i, const unsigned u_plane, const unsigned u_wave) {
    const int nbh = i >> 1, piece = i & 1;
    const unsigned so = (unsigned)cc * (24u * u_plane) + u_wave + (unsigned)(KY * 2 + piece) * u_plane + (unsigned)nbh * 1024u;
    st.fb[KY][nbh][piece] = buf_load16(a.u9, a.u_bytes, st.u_voff, so);
}
// patch piece I of chunk cc: global -> staging register (rows outside the image: zeros, the conv's padding)
template <int I>
__device__ __forceinline__ void pload(State& st, u32x4 (&stg)[NSTG], const Args& a, const int cc, const bool up, const int wave) {
    if constexpr (I < 10) {
        // always issued (a branch around the load makes hipcc's wait counts conservative: vmcnt(0) at the next weight-fragment use);
        // a row outside the image reads out of range -> zeros
        const int iy = st.y0m1 + I;
        const bool ok = (unsigned)iy < (unsigned)a.H;
        const int sy = ok ? (up ? (iy >> 1) : iy) : 0;
        stg[I] = buf_load16(a.x, a.x_bytes, ok ? st.vcol : OOB, st.img_base + (unsigned)sy * st.row_pitch + (unsigned)cc * 64u);
    } else {
        stg[10] = buf_load16(a.x, a.x_bytes, st.vext, st.img_base + (unsigned)cc * 64u);      // (threads >= 80: out of range -> zeros)
    }
}
// ... staging register -> patch buffer `pbuf`
template <int I>
__device__ __forceinline__ void pwrite(State& st, const u32x4 (&stg)[NSTG], const int pbuf, const int wave) {
    if constexpr (I < 10) {
        *reinterpret_cast<u32x4*>(st.wb + pbuf * P_BYTES + I * ROW_BYTES) = stg[I];
    } else {
        *reinterpret_cast<u32x4*>(st.wext + pbuf * P_BYTES) = stg[10];                           // (threads >= 80: into the slack slots)
    }
}
template <int... I>
__device__ __forceinline__ void pload_all(State& st, u32x4 (&stg)[NSTG], const Args& a, const int cc, const bool up, const int wave, std::integer_sequence<int, I...>) {
    (pload<I>(st, stg, a, cc, up, wave), ...);
}
template <int... I>
__device__ __forceinline__ void pwrite_all(State& st, const u32x4 (&stg)[NSTG], const int pbuf, const int wave, std::integer_sequence<int, I...>) {
    (pwrite<I>(st, stg, pbuf, wave), ...);
}

// One slice: MFMA S of the chunk with parity PAR, and what is issued beside it.
template <int S, int PAR, bool LAST>
__device__ __forceinline__ void slice(State& st, const Args& a, const int cn, const int wave, const bool up, const unsigned u_plane, const unsigned u_wave) {
    constexpr int seg = S / 6;
    constexpr int r = SEG_ROW[seg], ky = SEG_KY[seg];
    constexpr int term = (S % 6) / 2, nbh = S & 1;
    constexpr int ku = term == 1 ? 1 : 0, kv = term == 0 ? 1 : 0;         // terms: hi lo', lo hi', hi hi'
    constexpr int vbuf = (r + 2 * PAR) & 3;
    if constexpr (S == BARRIER_SLICE && !LAST) {
        // every wave is done reading this chunk's patch, and the next chunk's (written a chunk ago) is complete
        W9_BARRIER();
        __builtin_amdgcn_sched_barrier(0);
    }
    st.acc[r - ky][nbh] = mfma16(st.fb[ky][nbh][ku], st.vf[vbuf][kv], st.acc[r - ky][nbh]);
    // ---- V production: job j builds the fragment of row j + 2 of this chunk (j < 8) or of row j - 8 of the next chunk ----
    if constexpr (S >= JOB0 && S < JOB0 + 10 * JOB_SLICES) {
        constexpr int j = (S - JOB0) / JOB_SLICES, k = (S - JOB0) % JOB_SLICES;
        if constexpr (!(LAST && j >= 8)) {
            constexpr int buf = j < 8 ? ((j + 2 + 2 * PAR) & 3) : ((j - 8 + 2 * (PAR ^ 1)) & 3);
            vop<2 * k>(st, buf);
            vop<2 * k + 1>(st, buf);
            // raw reads of the next job (j + 1): rows 3..9 of this patch, then rows 0, 1, 2 of the next
            if constexpr (k >= 4 && k <= 7 && !(LAST && j >= 7)) {
                constexpr int jn = j + 1;
                constexpr int nrow = jn < 8 ? jn + 2 : jn - 8;
                constexpr int npb = jn < 8 ? PAR : (PAR ^ 1);
                rread<k - 4>(st, npb, nrow);
            }
        }
    }
    // ---- weight fragments: kernel row 2 of THIS chunk (first used at slice 30), rows 0 / 1 of the next once this chunk is done with them ----
    if constexpr (S < 4) load_b<2>(st, a, cn, S, u_plane, u_wave);
    if constexpr (!LAST && S >= KY0_DEAD && S < KY0_DEAD + 8 && (S - KY0_DEAD) % 2 == 0) load_b<0>(st, a, cn + 1, (S - KY0_DEAD) / 2, u_plane, u_wave);
    if constexpr (!LAST && S >= KY1_DEAD && S < KY1_DEAD + 8 && (S - KY1_DEAD) % 2 == 0) load_b<1>(st, a, cn + 1, (S - KY1_DEAD) / 2, u_plane, u_wave);
    // ---- patch of the chunk after next: global -> registers early, registers -> this chunk's buffer (dead after the barrier) ----
    if constexpr (!LAST && S >= 4 && S <= 44 && (S - 4) % 4 == 0) {
        pload<(S - 4) / 4>(st, st.stg, a, cn + 2, up, wave);      // (unconditional: past the last chunk it fetches out-of-range zeros / a neighbour's channels into a dead buffer)
    }
    if constexpr (!LAST && S >= 99 && S <= 119 && (S - 99) % 2 == 0) {
        pwrite<(S - 99) / 2>(st, st.stg, PAR, wave);
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int PAR, bool LAST, int... S>
__device__ __forceinline__ void chunk_impl(State& st, const Args& a, const int cn, const int wave, const bool up, const unsigned u_plane,
                                           const unsigned u_wave, std::integer_sequence<int, S...>) {
    __builtin_amdgcn_sched_barrier(0);
    (slice<S, PAR, LAST>(st, a, cn, wave, up, u_plane, u_wave), ...);
}
template <int PAR, bool LAST>
__device__ __forceinline__ void chunk(State& st, const Args& a, const int cn, const int wave, const bool up, const unsigned u_plane, const unsigned u_wave) {
    chunk_impl<PAR, LAST>(st, a, cn, wave, up, u_plane, u_wave, std::make_integer_sequence<int, NSLICE>{});
}
template <int... O>
__device__ __forceinline__ void job_all(State& st, const int buf, std::integer_sequence<int, O...>) {
    (vop<O>(st, buf), ...);
}

template <bool RES>      // RES: the launch adds a residual (32 more registers live through the epilogue passes)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void winograd9_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sX = smem + 2 * P_BYTES;      // dedicated exchange region; pass 1 uses the (idle) patch area instead

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = transform position p owned by this wave
    const int h = lane >> 5, t = lane & 31;
    const bool up = a.flags & CNL_UPSAMPLE_IN;
    const unsigned u_plane = (unsigned)(a.CoutP * 32);               // bytes per (chunk, position, ky, piece) plane of U
    const unsigned u_wave = (unsigned)wave * 6u * u_plane;
    typedef std::make_integer_sequence<int, NSTG> AllPieces;

    State st;
    st.row_pitch = (unsigned)(a.Ws * a.ldx * 4);
    // V_p = d[offa] + sg d[offb] over the four pixels 2t-1 .. 2t+2 of a tile:  p = 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
    const int offa = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int offb = wave == 3 ? 3 : (wave == 2 ? 1 : 2);
    st.sg = wave == 1 ? 1.f : -1.f;
    // patch image [row][quad][parity][33 px] x 16 B: the lanes of a ds_read_b128 group (same h, 16 distinct t mod 16) read
    // consecutive slots of one plane — all 64 banks, no conflicts
    {
        const int sa = (2 * h) * QUAD_SLOTS + (offa & 1) * PXH + t + (offa >> 1);
        const int sb = (2 * h) * QUAD_SLOTS + (offb & 1) * PXH + t + (offb >> 1);
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
    const float lo = (a.flags & CNL_RELU) ? 0.f : -__builtin_inff();
    float inv_n = 1.f;
    float omax = 0.f;

    int n, y0, x0, n0;
#define W9_IMAGE_OF(item_) ((int)(__builtin_amdgcn_readfirstlane(cnl::xcd_remap((item_), (unsigned)a.blocks)) / (unsigned)(a.nb * a.bx * a.by)))
#define W9_SETUP(item_, xmax_)                                                                                   \
    do {                                                                                                         \
        unsigned b_ = __builtin_amdgcn_readfirstlane(cnl::xcd_remap((item_), (unsigned)a.blocks));               \
        const int nbi_ = b_ % a.nb; b_ /= a.nb;                                                                  \
        const int bxi_ = b_ % a.bx; b_ /= a.bx;                                                                  \
        const int byi_ = b_ % a.by; n = b_ / a.by;                                                               \
        y0 = byi_ * R; x0 = bxi_ * (2 * TW); n0 = nbi_ * BN;                                                     \
        st.y0m1 = y0 - 1;                                                                                        \
        st.img_base = (unsigned)(n * a.Hs) * st.row_pitch;                                                       \
        int tid_ = tid;                                                                                          \
        asm volatile("" : "+v"(tid_));      /* keeps the per-thread decode inside the item loop (hoisted, its values live across the main loop) */ \
        {                                                                                                        \
            const int q_ = tid_ & 3, ix_ = x0 - 1 + (tid_ >> 2);                                                 \
            const int sx_ = up ? (ix_ >> 1) : ix_;                                                               \
            st.vcol = (unsigned)ix_ < (unsigned)a.W ? (unsigned)((sx_ * a.ldx + q_ * 4) * 4) : OOB;              \
            const int er_ = tid_ >> 3, ex_ = x0 + 63 + ((tid_ >> 2) & 1), ey_ = y0 - 1 + er_;                    \
            const bool ok_ = tid_ < 80 && (unsigned)ey_ < (unsigned)a.H && (unsigned)ex_ < (unsigned)a.W;        \
            const int esy_ = up ? (ey_ >> 1) : ey_, esx_ = up ? (ex_ >> 1) : ex_;                                \
            st.vext = ok_ ? (unsigned)(((esy_ * a.Ws + esx_) * a.ldx + q_ * 4) * 4) : OOB;                       \
        }                                                                                                        \
        st.u_voff = (unsigned)((n0 + (tid_ & 31)) * 32 + ((tid_ >> 5) & 1) * 16);                                \
        {                                                                                                        \
            const float mx2_ = 2.f * (xmax_);             /* |V| <= 2 max |x| */                                 \
            int es_ = 0;                                                                                         \
            if (mx2_ > 0.f && mx2_ < __builtin_inff()) {                                                         \
                int e_;                                                                                          \
                (void)__builtin_frexpf(mx2_, &e_);            /* 2^(e-1) <= mx2 < 2^e */                         \
                e_ = 14 - e_;                                                                                    \
                es_ = e_ < -100 ? -100 : (e_ > 100 ? 100 : e_);                                                  \
            }                                                                                                    \
            st.S = __builtin_ldexpf(1.f, es_);                                                                   \
            inv_n = __builtin_ldexpf(1.f, -es_);                                                                 \
        }                                                                                                        \
    } while (0)
#define W9_LOAD_B01()                                                                                            \
    do {                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) load_b<0>(st, a, 0, i, u_plane, u_wave);                   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) load_b<1>(st, a, 0, i, u_plane, u_wave);                   \
    } while (0)

    unsigned item = blockIdx.x;
#ifdef W9_TRACE
    int tr_item = 0;
#endif
    W9_STAMP(0);
    W9_SETUP(item, a.xmax[W9_IMAGE_OF(item)]);
    u32x4 stg1[NSTG];                   // patch 1 of the item set up last (patch 0 waits in st.stg)
    pload_all(st, st.stg, a, 0, up, wave, AllPieces{});
    pload_all(st, stg1, a, 1, up, wave, AllPieces{});
    W9_LOAD_B01();
    while (true) {
        W9_STAMP(1);
#pragma unroll
        for (int yo = 0; yo < R; ++yo)
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int e = 0; e < 16; ++e) st.acc[yo][g][e] = 0.f;
        // patches 0 / 1 of this item: registers -> LDS (everybody is past the previous item's last exchange pass: barrier below pass 3)
        pwrite_all(st, st.stg, 0, wave, AllPieces{});
        pwrite_all(st, stg1, 1, wave, AllPieces{});
        W9_BARRIER();
        // fragments of rows 0 and 1 of chunk 0 (not overlapped with MFMAs), raw reads of row 2 for job 0
        rread<0>(st, 0, 0); rread<1>(st, 0, 0); rread<2>(st, 0, 0); rread<3>(st, 0, 0);
        job_all(st, 0, std::make_integer_sequence<int, 28>{});
        rread<0>(st, 0, 1); rread<1>(st, 0, 1); rread<2>(st, 0, 1); rread<3>(st, 0, 1);
        job_all(st, 1, std::make_integer_sequence<int, 28>{});
        rread<0>(st, 0, 2); rread<1>(st, 0, 2); rread<2>(st, 0, 2); rread<3>(st, 0, 2);
        W9_STAMP(5);

        for (int cn = 0; cn < a.CC - 2; cn += 2) {
            chunk<0, false>(st, a, cn, wave, up, u_plane, u_wave);
            chunk<1, false>(st, a, cn + 1, wave, up, u_plane, u_wave);
        }
        chunk<0, false>(st, a, a.CC - 2, wave, up, u_plane, u_wave);     // (a mid-loop exit instead of this second copy sends the register allocator into 700 spills)
        chunk<1, true>(st, a, a.CC - 1, wave, up, u_plane, u_wave);
        W9_STAMP(6);

        // ---- epilogue: out0 = Y0 + Y1 + Y2, out1 = Y1 - Y2 - Y3; the four positions (waves) meet through LDS.  Pass k: output rows
        // 2k, 2k+1 x two cout halves = 4 blocks; every wave writes its 4 blocks, wave w finishes block w = (row 2k + (w >> 1), half w & 1)
        // for all four positions: thread = (tile, 4 couts) x 4 cout quads, 16-byte stores ----
        const int en = n, ey0 = y0, ex0 = x0;
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));        // the epilogue's per-lane constants are re-derived here (hoisted, they live across the chunk loop and spill)
        const int t_e = lane_e & 31, h_e = lane_e >> 5;
        const float inv = inv_n;
        const unsigned next = item + gridDim.x;
        const bool more = next < (unsigned)a.blocks;
        const float xmax_next = more ? a.xmax[W9_IMAGE_OF(next)] : 0.f;     // requested now, used by the prefetch in pass 2
        const int cbase = n0 + (wave & 1) * 32 + 4 * h_e;             // + 8 q: this thread's cout quads
        f32x4 bq[4], iq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = cbase + 8 * q;
            bq[q] = __builtin_bit_cast(f32x4, buf_load16(a.bias, a.b_bytes, c < a.Cout ? (unsigned)c * 4u : OOB, 0));
            const f32x4 s_ = *reinterpret_cast<const f32x4*>(a.isu + c);
            iq[q] = s_ * inv;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            char* X = (k == 1) ? smem : sX;
            const int oy = ey0 + 2 * k + (wave >> 1);
            const int ox = ex0 + 2 * t_e;
            const bool row_ok = oy < a.H;
            const unsigned pix = (unsigned)((en * a.H + oy) * a.W + ox);
            unsigned yv[4];
            bool okc[4];
            f32x4 rv[2][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                okc[q] = row_ok && (cbase + 8 * q) < a.Cout;
                yv[q] = (pix * (unsigned)a.ldy + (unsigned)(cbase + 8 * q)) * 4u;
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    rv[px][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if constexpr (RES) {
                        const unsigned rvo = (pix * (unsigned)a.ldr + (unsigned)(cbase + 8 * q)) * 4u;
                        rv[px][q] = __builtin_bit_cast(f32x4, buf_load16(a.res, a.r_bytes, (okc[q] && ox + px < a.W) ? rvo : OOB,
                                                                         (unsigned)(px * a.ldr * 4)));
                    }
                }
            }
            W9_STAMP(7 + 2 * k);
            if (k == 3) W9_BARRIER();                  // pass 2's readers are done with sX
#pragma unroll
            for (int yy = 0; yy < 2; ++yy)
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x16& A = st.acc[2 * k + yy][g];
                        *reinterpret_cast<f32x4*>(X + ((((yy * 2 + g) * 4 + wave) * 4 + q) * 64 + lane_e) * 16) =
                            f32x4{A[4 * q], A[4 * q + 1], A[4 * q + 2], A[4 * q + 3]};
                    }
            W9_BARRIER();
            W9_STAMP(8 + 2 * k);
            if (k == 2 && more) {                      // the staging and fragment registers are idle: request the next item's first two patches
                W9_SETUP(next, xmax_next);
                pload_all(st, st.stg, a, 0, up, wave, AllPieces{});
                pload_all(st, stg1, a, 1, up, wave, AllPieces{});
            } else if (k == 2) {                       // (no loop-carried staging values: they would stay live through the chunk loop)
#pragma unroll
                for (int i = 0; i < NSTG; ++i) { st.stg[i] = u32x4{0u, 0u, 0u, 0u}; stg1[i] = u32x4{0u, 0u, 0u, 0u}; }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 Y[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) Y[p] = lds_f4(X + (((wave * 4 + p) * 4 + q) * 64 + lane_e) * 16);
                f32x4 o0, o1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ya = (Y[0][e] + Y[1][e] + Y[2][e]) * iq[q][e];
                    const float yb = (Y[1][e] - Y[2][e] - Y[3][e]) * iq[q][e];
                    if constexpr (RES) {
                        o0[e] = fmaxf(ya + bq[q][e] + rv[0][q][e], lo);
                        o1[e] = fmaxf(yb + bq[q][e] + rv[1][q][e], lo);
                    } else {
                        o0[e] = fmaxf(ya + bq[q][e], lo);
                        o1[e] = fmaxf(yb + bq[q][e], lo);
                    }
                }
                const bool ok0 = okc[q] && ox < a.W, ok1 = okc[q] && ox + 1 < a.W;
                if (ok0) omax = fmaxf(omax, fmaxf(fmaxf(fabsf(o0[0]), fabsf(o0[1])), fmaxf(fabsf(o0[2]), fabsf(o0[3]))));
                if (ok1) omax = fmaxf(omax, fmaxf(fmaxf(fabsf(o1[0]), fabsf(o1[1])), fmaxf(fabsf(o1[2]), fabsf(o1[3]))));
                buf_store16(o0, a.y, a.y_bytes, ok0 ? yv[q] : OOB, 0);
                buf_store16(o1, a.y, a.y_bytes, ok1 ? yv[q] : OOB, (unsigned)(a.ldy * 4));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) { W9_LOAD_B01(); }
        if (a.ymax) {          // max |y| of this item into its image's slot: one atomic per wave and item
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) omax = fmaxf(omax, __shfl_xor(omax, o, 64));
            if (lane == 0 && omax > 0.f) atomicMax(a.ymax + en, __float_as_uint(omax));
            omax = 0.f;
        }
        W9_STAMP(15);
#ifdef W9_TRACE
        ++tr_item;
#endif
        if (!more) break;
        item = next;
        W9_BARRIER();          // everybody is done with the exchange regions before the patch buffers are refilled
    }
#undef W9_SETUP
#undef W9_LOAD_B01
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
            const float uu[4] = {g0, 0.5f * (g0 + g1 + g2), 0.5f * (g0 - g1 + g2), g2};
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

}  // namespace cnl_wino9

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
extern "C" void cnl_w9_set_trace(void* p) { g_w9_trace = (unsigned long long*)p; }
#endif
// can this kernel run the layer at all?  (shape / alignment only)
bool cnl_wino9_eligible(const cnl_conv_params* p) {
    return p->Cin % 32 == 0 && p->Cin >= 32 && p->Cout % 4 == 0 && p->ldy % 4 == 0 && ((uintptr_t)p->y & 15) == 0 &&
           (!p->residual || (p->ldr % 4 == 0 && ((uintptr_t)p->residual & 15) == 0));
}

// Launch (arguments already validated by cnl_conv3x3_winograd_f32); xmax = N per-image maxima of the input.
int cnl_wino9_launch(const cnl_conv_params* p, const void* u9, const float* isu, const float* xmax, void* stream) {
    using namespace cnl_wino9;
    Args a;
    a.x = p->x; a.u9 = u9; a.xmax = xmax; a.isu = isu; a.ymax = reinterpret_cast<unsigned*>(p->y_absmax);
    a.bias = p->bias; a.res = p->residual; a.y = p->y;
    const int upf = (p->flags & CNL_UPSAMPLE_IN) ? 2 : 1;
    a.N = p->N; a.Hs = p->H_in; a.Ws = p->W_in; a.H = p->H_in * upf; a.W = p->W_in * upf; a.Cin = p->Cin; a.Cout = p->Cout;
    a.CoutP = (p->Cout + 63) / 64 * 64;
    a.ldx = p->ldx; a.ldy = p->ldy; a.ldr = p->ldr;
    a.CC = p->Cin / 16;
    a.nb = a.CoutP / BN; a.bx = (a.W + 2 * TW - 1) / (2 * TW); a.by = (a.H + R - 1) / R;
    const long long blocks = (long long)p->N * a.by * a.bx * a.nb;
    CNL_REQUIRE(blocks < (1ll << 31), CNL_E_UNSUPPORTED, "cnl_conv3x3_winograd_f32: grid too large");
    a.blocks = (int)blocks;
    const unsigned long long xb = (((unsigned long long)p->N * p->H_in * p->W_in - 1) * p->ldx + p->Cin) * 4ull;
    const unsigned long long ub = (unsigned long long)cnl_wino9_weight_bytes(p->Cin, p->Cout);
    const unsigned long long Mo = (unsigned long long)p->N * a.H * a.W;
    const unsigned long long yb = ((Mo - 1) * p->ldy + p->Cout) * 4ull;
    const unsigned long long rb = p->residual ? ((Mo - 1) * p->ldr + p->Cout) * 4ull : 0ull;
    CNL_REQUIRE(xb < 0xFFFFFF00ull && ub < 0xFFFFFF00ull && yb + 4ull * p->ldy < 0xFFFFFF00ull && rb + 4ull * (p->residual ? p->ldr : 0) < 0xFFFFFF00ull,
                CNL_E_UNSUPPORTED, "cnl_conv3x3_winograd_f32: tensor spans >= 4 GiB; split the batch");
    a.x_bytes = (unsigned)xb; a.u_bytes = (unsigned)ub; a.y_bytes = (unsigned)yb; a.r_bytes = (unsigned)rb; a.b_bytes = (unsigned)p->Cout * 4u;
    a.flags = p->flags;
#ifdef W9_TRACE
    a.trace = g_w9_trace;
#endif
    static cnl::DeviceOnce once;
    int n_cu = 0;                          // persistent workgroups: one per CU, walking the work items with stride gridDim.x
    static cnl::DeviceOnce once_res;
    int rc = p->residual ? cnl::kernel_setup(once_res, reinterpret_cast<const void*>(&winograd9_kernel<true>), LDS_BYTES, &n_cu)
                         : cnl::kernel_setup(once, reinterpret_cast<const void*>(&winograd9_kernel<false>), LDS_BYTES, &n_cu);
    if (rc != CNL_OK) return rc;
    const unsigned grid = (unsigned)(blocks < (long long)n_cu ? blocks : (long long)n_cu);
    if (p->residual) hipLaunchKernelGGL(winograd9_kernel<true>, dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(winograd9_kernel<false>, dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, a);
    return cnl::check_launch("winograd9_kernel");
}
