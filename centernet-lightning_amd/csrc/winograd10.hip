// winograd10.hip — the row-Winograd arithmetic of winograd9.hip (1-D F(2,3) along x, the three kernel rows folded into the reduction,
// every fp32 product formed on the FP16 matrix cores from scaled two-way splits) on HALF-HEIGHT work items, TWO workgroups per CU.
//
//   item = 4 output rows x 64 pixels x 64 couts (winograd9: 8 rows), 128 accumulator registers per lane, 256 registers per wave:
//   two 4-wave workgroups are co-resident on a CU, two waves per SIMD.  What that buys (VERDICT r3 #1a, #3, #5):
//   * one workgroup's item boundary — the eight-pass exchange of winograd9 is four passes here — runs under the OTHER workgroup's chunk
//     loop instead of leaving the matrix pipe idle (winograd9: 12 K cycles of a 34 K-cycle item on the 64-channel layers);
//   * with one wave per SIMD every exposed issue latency (a buffer load costs the wave ~30 cycles, a barrier its skew) is a bubble in
//     the matrix pipe; with two, the partner's MFMAs fill it;
//   * twice the work items of half the size: maps whose 8-row items do not fill 256 CUs (16 x 16 and 32 x 32 maps, one-image batches).
//   What it costs: 6 patch rows per 4 output rows instead of 10 per 8 (V production and patch traffic x 1.2), each weight fragment
//   serves 4 rows instead of 8 (weight fragments from L2 x 2 per MFMA), and 128 instead of 256 non-accumulator registers.
// Same weights as winograd9.hip (its buffer and per-cout scales are used as they are), same scaling rules (weights per output channel,
// activations per image: batch-invariant), and the SAME chain of fp32 additions per accumulator (chunk-major, then input row, kernel row, split
// term): the results are bit-identical to winograd9's wherever both run (tests/test_gpu_conv.py pins that with torch.equal).
// Measured (profiles/r04_w10_trace.txt): the two workgroups of a CU are not equals — the issue arbitration is by age, the
// first-dispatched one runs almost as if alone and the second fills what is left; together they raise the matrix-pipe utilisation of the
// 256-channel head blocks from 0.72 to 0.80-0.85, and the power limit takes it back as clock.  The kernel is dispatched where the half-height
// items are what matters: 16-pixel-wide maps with long channel loops, and the latency class (one-image batches).
//
// Schedule of a 16-channel chunk (72 MFMAs per wave, one per sched_barrier slice; wave p = transform position p):
//   segments (input row r, kernel row ky) -> output row r - ky, rows in order, ky-major inside a row:
//     (0,0) (1,0)(1,1) (2,0)(2,1)(2,2) (3,0)(3,1)(3,2) (4,1)(4,2) (5,2)          6 MFMAs each: 3 split terms x 2 cout halves
//   V fragments live in THREE rotating buffers (row r -> r % 3): job j = slices [6 + 12 j, 17 + 12 j) builds row 2, 3, 4, 5 of this chunk
//   (j = 0..3) and rows 0, 1 of the next (j = 4, 5; job 5 ends in the next chunk's slices 0..4) — each exactly one consumption period
//   ahead of its first use, 28 VALU operations (3 per slice, then 2), its four ds_read_b128 issued during the job before it;
//   weight fragments single-buffered (48 registers): ky = 0 dies at slice 42, ky = 1 at 60 -> reloaded for the next chunk there, ky = 2 of
//   THIS chunk is loaded in slices 0..3 (first use: slice 30);
//   the patch of chunk n + 2 travels global -> 4 staging registers -> LDS in two halves (rows 0..2 + the two last pixel columns, rows
//   3..5); one barrier per chunk (slice 40).
#include "cnl_common.h"
#include <utility>

#pragma clang fp contract(off)

namespace cnl_wino10 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct Args {
    const float* x;
    const void* u9;                   // pre-split, pre-scaled weights of winograd9.hip: [ci/16][p][ky][piece][CoutP][16 ci] fp16
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
    int nb, bx, by;                   // blocks along cout (64), x (64 px), y (4 rows)
    unsigned m_nb, m_bx, m_by;        // floor(2^32 / d) of the three
    int ipb, lw;                      // images side by side in one 64-pixel block row (W = 32: 2, W = 16: 4; else 1) and log2 W for them
    int Nimg;                         // images of the launch (N = image groups)
    int pk;                           // packed rows (0: off): the launch's images side by side in ONE virtual row, each in a strip of pk = W + 2 columns
    unsigned m_pk;                    //   (its W pixels + the two columns of zero padding that separate it from the next image); floor(2^32 / pk)
    int blocks;
    unsigned x_bytes, u_bytes, y_bytes, r_bytes, b_bytes;
    unsigned flags;
#ifdef W10_TRACE
    unsigned long long* trace;        // timing build: [item][16] s_memtime stamps of block 0 / thread 0
#endif
};
#ifdef W10_TRACE
#define W10_STAMP(i_) do { if (blockIdx.x == 0 && tid == 0 && tr_item < 64) a.trace[tr_item * 16 + (i_)] = __builtin_readcyclecounter(); } while (0)
#else
#define W10_STAMP(i_) do {} while (0)
#endif

constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr int R = 4;                        // output rows per work item
constexpr int PR = R + 2;                   // patch rows
constexpr int TW = 32;                      // tiles (pixel pairs) per row of a work item: 64 output pixels
constexpr int PXH = TW + 1;                 // 33 pixels per parity plane of a patch row (columns x0-1 .. x0+64)
constexpr int QUAD_SLOTS = 2 * PXH;         // 66 16-byte slots per (row, channel quad): [parity][33]
constexpr int ROW_SLOTS = 4 * QUAD_SLOTS;   // 264 per patch row: [quad][parity][33]
constexpr int ROW_BYTES = ROW_SLOTS * 16;   // 4224
constexpr int P_SLOTS = PR * ROW_SLOTS + 48;      // 1632 (+ 48 slots that absorb the idle lanes of the column piece)
constexpr int P_BYTES = P_SLOTS * 16;       // 26112 per buffer (two buffers)
constexpr int X_BYTES = 65536;              // epilogue exchange: two halves of [2 cout halves][4 positions][32 tiles][8 pieces] x 16 B — ALIASES the patch buffers
constexpr int B_BYTES = 512;                // the item's 64 bias values and 64 inverse weight scales
constexpr int LDS_BYTES = X_BYTES + B_BYTES;      // 66048: two workgroups per CU
static_assert(2 * P_BYTES <= X_BYTES, "the exchange region covers both patch buffers");
constexpr int NSLICE = 72;                  // MFMAs per wave and chunk
constexpr int JOB0 = 6, JOB_SLICES = 12;    // job j: slices [JOB0 + 12 j, JOB0 + 12 j + 11) (the twelfth slice stays free: distance to the first MFMA that reads the fragment)
constexpr int BARRIER_SLICE = 40;
constexpr int NSTG = 4;

__device__ __forceinline__ u32x4 buf_load16(const void* base, unsigned bytes, unsigned voffset, unsigned soffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    return (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, soffset, 0);
}
__device__ __forceinline__ void buf_store16(f32x4 v, float* base, unsigned bytes, unsigned voffset, unsigned soffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, voffset, soffset, CNL_NT_STORES);
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
__device__ __forceinline__ int lane_now() {
    unsigned z = 0;
    asm volatile("" : "+v"(z));
    return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
}
__device__ __forceinline__ f32x4 lds_f4(const char* p) { return *reinterpret_cast<const f32x4*>(p); }
#define W10_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// ---- the static schedule of a chunk -----------------------------------------------------------------------------------------
constexpr int NSEG = 12;
constexpr int SEG_ROW[NSEG] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 5};
constexpr int SEG_KY[NSEG] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 1, 2, 2};
constexpr int KY0_DEAD = 42, KY1_DEAD = 60;        // first slices after the last use of the ky = 0 / ky = 1 weight fragments

struct Item {               // per-work-item addressing state
    unsigned vcol, vext;     // source offsets: column part of pieces 0..5 (the row is a scalar offset), full offset of the column piece
    unsigned u_voff;         // this lane's row of the weight planes
    unsigned img_base;       // scalar: byte offset of image n
    int y0m1;                // scalar: y0 - 1, first patch row
    float S;                 // power-of-two scale of V for this item's image
};
template <int NBH>           // NBH: cout halves (32 couts each) per work item
struct State {
    f32x16 acc[R][NBH];      // [output row][cout half]: D[cout][tile]
    u32x4 fb[3][NBH][2];     // weight fragments (A operand): [ky][cout half][piece], single-buffered
    u32x4 vf[3][2];          // V fragments (B operand): [row % 3][piece]
    f32x4 raw[4];            // patch reads of a job: pixel a quad 0, a quad 1, pixel b quad 0, b quad 1
    float v[8];              // transform temporaries of the running job (V, then its residual in place)
    u32x4 stg[NSTG];         // patch pieces on their way global -> LDS
    Item cur;
    unsigned row_pitch;      // scalar: bytes per stored input row
    const char* pa;          // LDS address of this lane's pixel a / b in patch buffer 0 (buffer 1: + P_BYTES)
    const char* pb;
    char* wb;                // LDS write address of piece 0 in buffer 0 (piece i: + i rows), and of the column piece
    char* wext;
    float sg;
    float bst, ist;          // this lane's bias / inverse weight scale of the item (cout n0 + lane), on their way to LDS
    char* sB;
};

// VALU operation o (0..27) of the job that builds V fragment `buf`
template <int O, class ST>
__device__ __forceinline__ void vop(ST& st, const int buf) {
    const float S = st.cur.S;
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
        asm volatile("" : "+v"(st.vf[buf][1][j]));      // (pinned in its slice: see winograd9.hip)
    }
}
// LDS read i (0..3) of patch row `row` of buffer `pbuf`
template <int I, class ST>
__device__ __forceinline__ void rread(ST& st, const int pbuf, const int row) {
    const char* p = (I < 2 ? st.pa : st.pb) + pbuf * P_BYTES + row * ROW_BYTES + (I & 1) * (QUAD_SLOTS * 16);
    st.raw[I] = lds_f4(p);
}
// weight fragment i (cout half i >> 1, piece i & 1) of kernel row KY of chunk cc: global -> registers (`ok` false: nothing is fetched)
template <int KY, class ST>
__device__ __forceinline__ void load_b(ST& st, const Args& a, const int cc, const int i, const unsigned u_plane, const unsigned u_wave, const bool ok) {
    const int nbh = i >> 1, piece = i & 1;
    const unsigned so = (unsigned)cc * (24u * u_plane) + u_wave + (unsigned)(KY * 2 + piece) * u_plane + (unsigned)nbh * 1024u;
    st.fb[KY][nbh][piece] = buf_load16(a.u9, a.u_bytes, ok ? st.cur.u_voff : OOB, so);
}
// Patch piece I of half HALF (0: rows 0..2 + the piece with the two last pixel columns of all rows = stg[3]; 1: rows 3..5) of chunk cc:
// always issued (no branch: see winograd9.hip); a row outside the image — or `ok` false: a chunk past the item's last — reads out of range.
template <int HALF, int I, class ST>
__device__ __forceinline__ void pload(ST& st, const Args& a, const int cc, const bool up, const bool ok) {
    if constexpr (I < 3) {
        const int iy = __builtin_amdgcn_readfirstlane(st.cur.y0m1) + 3 * HALF + I;
        const bool okr = ok && (unsigned)iy < (unsigned)a.H;
        const int sy = okr ? (up ? (iy >> 1) : iy) : 0;
        const unsigned so = __builtin_amdgcn_readfirstlane(st.cur.img_base + (unsigned)sy * st.row_pitch + (unsigned)cc * 64u);
        st.stg[I] = buf_load16(a.x, a.x_bytes, okr ? st.cur.vcol : OOB, so);
    } else {
        static_assert(HALF == 0, "the column piece belongs to half A");
        st.stg[3] = buf_load16(a.x, a.x_bytes, ok ? st.cur.vext : OOB, __builtin_amdgcn_readfirstlane(st.cur.img_base + (unsigned)cc * 64u));
    }
}
template <int HALF, int I, class ST>
__device__ __forceinline__ void pwrite(ST& st, const int pbuf) {
    if constexpr (I < 3) *reinterpret_cast<u32x4*>(st.wb + pbuf * P_BYTES + (3 * HALF + I) * ROW_BYTES) = st.stg[I];
    else *reinterpret_cast<u32x4*>(st.wext + pbuf * P_BYTES) = st.stg[3];
}

constexpr bool first_use(int S) {      // is slice S the first MFMA of a chunk into its accumulator block (output row, cout half)?
    const int yo = SEG_ROW[S / 6] - SEG_KY[S / 6], nbh = S & 1;
    for (int s = 0; s < S; ++s)
        if (SEG_ROW[s / 6] - SEG_KY[s / 6] == yo && (s & 1) == nbh) return false;
    return true;
}
// One slice: MFMA S of the chunk with parity PAR (FIRST: the item's first chunk — accumulators start from C = 0, the V rows 0 / 1, the weight
// rows 0 / 1 and both patches come from the item's prologue), and what is issued beside it.
template <int S, int PAR, bool FIRST, int NBH>
__device__ __forceinline__ void slice(State<NBH>& st, const Args& a, const int cn, const bool up, const unsigned u_plane, const unsigned u_wave,
                                      const bool has1, const bool has2) {
    constexpr int seg = S / 6;
    constexpr int r = SEG_ROW[seg], ky = SEG_KY[seg];
    constexpr int term = (S % 6) / 2, nbh = S & 1;
    constexpr int ku = term == 1 ? 1 : 0, kv = term == 0 ? 1 : 0;         // terms: hi lo', lo hi', hi hi'
    constexpr int vbuf = r % 3;
    if constexpr (S == BARRIER_SLICE) {
        // every wave is done reading this chunk's patch, and the next chunk's (written since the previous barrier) is complete
        W10_BARRIER();
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (nbh < NBH) {       // (32-cout items: the odd slices carry no MFMA — the chunk keeps its 72 issue slots for the V jobs, loads and staging)
        if constexpr (FIRST && first_use(S)) st.acc[r - ky][nbh] = mfma16(st.fb[ky][nbh][ku], st.vf[vbuf][kv], f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f});
        else st.acc[r - ky][nbh] = mfma16(st.fb[ky][nbh][ku], st.vf[vbuf][kv], st.acc[r - ky][nbh]);
    }
    __builtin_amdgcn_sched_barrier(0);       // the MFMA leads its slice
    // ---- V production: G = position in the job stream (job 5 wraps into the next chunk's first slices) ----
    constexpr int G = (S + NSLICE - JOB0) % NSLICE;
    constexpr int j = G / JOB_SLICES, k = G % JOB_SLICES;
    if constexpr (!(FIRST && S < JOB0) && k < 11) {
        constexpr int row = (j + 2) % 6;          // rows 2..5 of this chunk, then rows 0, 1 of the next
        constexpr int buf = row % 3;
        if constexpr (k < 6) {
            vop<3 * k>(st, buf); vop<3 * k + 1>(st, buf); vop<3 * k + 2>(st, buf);
        } else {
            vop<18 + 2 * (k - 6)>(st, buf); vop<19 + 2 * (k - 6)>(st, buf);
        }
    }
    // raw reads of the NEXT job (issued during slices k = 4..7 of the running one: its raw registers are free after operation 7, slice k = 2)
    if constexpr (!(FIRST && S < JOB0) && k >= 4 && k <= 7) {
        constexpr int jn = (j + 1) % 6;
        constexpr int nrow = (jn + 2) % 6;
        // the buffer of the chunk that row belongs to, seen from the slice the read is issued in: jobs 1..3 read this chunk's patch, jobs 4 / 5
        // the next chunk's; job 0 (row 2) is read across the chunk boundary — slices 70, 71 (next chunk's buffer) and 0, 1 (by then: this chunk's)
        constexpr int npb = (jn >= 1 && jn <= 3) ? PAR : (jn >= 4 ? (PAR ^ 1) : (S >= JOB0 ? (PAR ^ 1) : PAR));
        rread<k - 4>(st, npb, nrow);
    }
    // ---- weight fragments: kernel row 2 of THIS chunk (first used at slice 30), rows 0 / 1 of the next once this chunk is done with them ----
    if constexpr (S < 2 * NBH) load_b<2>(st, a, cn, S, u_plane, u_wave, true);
    if constexpr (S >= KY0_DEAD && S < KY0_DEAD + 4 * NBH && (S - KY0_DEAD) % 2 == 0) load_b<0>(st, a, cn + 1, (S - KY0_DEAD) / 2, u_plane, u_wave, has1);
    if constexpr (S >= KY1_DEAD && S < KY1_DEAD + 4 * NBH && (S - KY1_DEAD) % 2 == 0) load_b<1>(st, a, cn + 1, (S - KY1_DEAD) / 2, u_plane, u_wave, has1);
    // ---- patch of chunk cn + 2, in two halves through the same staging registers:
    //   slices 10..14   half B of the NEXT chunk's patch (requested a chunk ago) -> the other buffer, rows 3..5 (first read a chunk from now)
    //   slices 16..28   request half A          41..47  half A -> this chunk's buffer (dead after the barrier)        50..58  request half B
    if constexpr (!FIRST && S >= 10 && S <= 14 && (S - 10) % 2 == 0) pwrite<1, (S - 10) / 2>(st, PAR ^ 1);
    if constexpr (S >= 16 && S <= 28 && (S - 16) % 4 == 0) pload<0, (S - 16) / 4>(st, a, cn + 2, up, has2);
    if constexpr (S >= 41 && S <= 47 && (S - 41) % 2 == 0) pwrite<0, (S - 41) / 2>(st, PAR);
    if constexpr (S >= 50 && S <= 58 && (S - 50) % 4 == 0) pload<1, (S - 50) / 4>(st, a, cn + 2, up, has2);
    // ---- the item's bias / weight-scale values -> LDS for the epilogue (every wave writes the same 64 values; the region is not aliased) ----
    if constexpr (FIRST && S == 30) {
        float* sb = reinterpret_cast<float*>(st.sB) + lane_now();
        sb[0] = st.bst;
        sb[64] = st.ist;
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int PAR, bool FIRST, int NBH, int... S>
__device__ __forceinline__ void chunk_impl(State<NBH>& st, const Args& a, const int cn, const bool up, const unsigned u_plane, const unsigned u_wave,
                                           const bool has1, const bool has2, std::integer_sequence<int, S...>) {
    __builtin_amdgcn_sched_barrier(0);
    (slice<S, PAR, FIRST, NBH>(st, a, cn, up, u_plane, u_wave, has1, has2), ...);
}
template <int PAR, bool FIRST = false, int NBH>
__device__ __forceinline__ void chunk(State<NBH>& st, const Args& a, const int cn, const bool up, const unsigned u_plane, const unsigned u_wave) {
    const bool has1 = cn + 1 < a.CC, has2 = cn + 2 < a.CC;
    chunk_impl<PAR, FIRST, NBH>(st, a, cn, up, u_plane, u_wave, has1, has2, std::make_integer_sequence<int, NSLICE>{});
}
template <class ST, int... O>
__device__ __forceinline__ void job_all(ST& st, const int buf, std::integer_sequence<int, O...>) {
    (vop<O>(st, buf), ...);
}

template <bool RES, int NBH, bool PK = false>      // RES: the launch adds a residual; NBH: 64- or 32-cout work items; PK: packed rows (winograd9.hip)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void winograd10_kernel(const Args a) {
    constexpr int BN = 32 * NBH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sX = smem;                    // exchange region of the epilogue (two halves of 32 KB): the patch buffers are dead by then

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = transform position p owned by this wave
    const int h = lane >> 5, t = lane & 31;
    const bool up = a.flags & CNL_UPSAMPLE_IN;
    const unsigned u_plane = (unsigned)(a.CoutP * 32);               // bytes per (chunk, position, ky, piece) plane of U
    const unsigned u_wave = (unsigned)wave * 6u * u_plane;

    State<NBH> st;
    st.sB = smem + X_BYTES;
    st.row_pitch = (unsigned)(a.Ws * a.ldx * 4);
    // V_p = d[offa] + sg d[offb] over the four pixels 2t-1 .. 2t+2 of a tile:  p = 0: -(d0 - d2) (its weights are stored negated), 1: d1 + d2,
    // 2: d2 - d1, 3: d1 - d3 — the two OUTER pixels d0 / d3 are always operand b
    const int offa = wave == 0 ? 2 : (wave == 2 ? 2 : 1);
    const int offb = wave == 0 ? 0 : (wave == 3 ? 3 : (wave == 2 ? 1 : 2));
    const int tpi = (a.ipb > 1) ? (1 << (a.lw - 1)) : 64;            // tiles per image in the block row
    const bool outer_is_neighbour = a.ipb > 1 && ((wave == 0 && (t & (tpi - 1)) == 0) || (wave == 3 && (t & (tpi - 1)) == tpi - 1));
    st.sg = wave == 1 ? 1.f : -1.f;
    const int si_lane = (a.ipb > 1) ? ((2 * t) >> a.lw) : 0;         // this lane's sub-image (V production: its tile's image)
    {
        const int sa = (2 * h) * QUAD_SLOTS + (offa & 1) * PXH + t + (offa >> 1);
        // an outer pixel that belongs to the neighbouring image: patch column 0 instead (x = -1: outside every image, zero-filled)
        const int sb = outer_is_neighbour ? (2 * h) * QUAD_SLOTS : (2 * h) * QUAD_SLOTS + (offb & 1) * PXH + t + (offb >> 1);
        st.pa = smem + sa * 16;
        st.pb = smem + sb * 16;
    }
    // staging pieces: piece i < 6 = (patch row i, column tid / 4, channel quad tid % 4); the column piece = columns 64, 65 of all six rows (threads 0..47)
    {
        const int q = tid & 3, c = tid >> 2;
        st.wb = smem + (q * QUAD_SLOTS + (c & 1) * PXH + (c >> 1)) * 16;
        const int er = tid >> 3, ec = 64 + ((tid >> 2) & 1);
        st.wext = tid < 8 * PR ? smem + (er * ROW_SLOTS + q * QUAD_SLOTS + (ec & 1) * PXH + (ec >> 1)) * 16 : smem + (PR * ROW_SLOTS + (tid & 31)) * 16;
    }
    const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane((a.flags & CNL_RELU) ? 0 : (int)0xff800000u));      // ReLU floor or -inf

#define W10_DIVMOD(q_, r_, b_, d_, m_)                                                                           \
    do {                                                                                                         \
        unsigned qq_ = __builtin_amdgcn_readfirstlane(__umulhi((b_), (m_)));                                     \
        unsigned rr_ = (b_) - qq_ * (unsigned)(d_);                                                              \
        if (rr_ >= (unsigned)(d_)) { ++qq_; rr_ -= (unsigned)(d_); }                                             \
        (q_) = qq_; (r_) = rr_;                                                                                  \
    } while (0)
    // the same per lane (packed rows: virtual column -> image, pixel)
#define W10_VDIVMOD(q_, r_, b_, d_, m_)                                                                          \
    do {                                                                                                         \
        unsigned qq_ = __umulhi((b_), (m_));                                                                     \
        unsigned rr_ = (b_) - qq_ * (unsigned)(d_);                                                              \
        if (rr_ >= (unsigned)(d_)) { ++qq_; rr_ -= (unsigned)(d_); }                                             \
        (q_) = qq_; (r_) = rr_;                                                                                  \
    } while (0)
    // max |x| of image img_ (the one a lane's tile belongs to); images past the end of the batch: 0 -> scale 1
#define W10_XMAX_OF(img_) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(                          \
        __builtin_amdgcn_make_buffer_rsrc((void*)a.xmax, 0, a.Nimg * 4 * AMS, 0x00020000), (unsigned)(img_) * (4u * AMS), 0, 0))
    // power-of-two scale of V from the image's maximum: |V| <= 2 max |x|, 2 max |x| S in [2^13, 2^14); es_ = log2 S
#define W10_SCALE_EXP(es_, xmax_)                                                                                \
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

#ifdef W10_TRACE
    int tr_item = 0;
    // per-workgroup stamps behind the 64 x 16 item stamps: [block][8] = start (s_memtime), end, start (s_memrealtime, 100 MHz), end, HW_ID, XCC_ID, items
    unsigned long long* const trb = a.trace + 64 * 16 + (size_t)blockIdx.x * 8;
    if (tid == 0) {
        trb[0] = __builtin_readcyclecounter();
        trb[2] = __builtin_amdgcn_s_memrealtime();
        trb[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        trb[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
    int tr_n = 0;
#endif
    // Coordinates, per-thread addressing and ALL global requests of a work item (fourteen patch pieces of its chunks 0 / 1 into `keep`, the
    // kernel-row 0 / 1 weight fragments of chunk 0, its image's maximum, its bias / weight scales).  Issued for the first item before the
    // loop and for every later one INSIDE the previous item's epilogue (round 5: behind pass 1, when three of the four accumulator rows are
    // dead), so that an item no longer starts by waiting a memory latency for its first patch (traced: 3.3-4.6 K of an item's 25-30 K
    // cycles on the 64-channel layers, profiles/r04_w10_trace.txt).  `ok` false (no next item): every request reads out of range.
    struct Coord { int n, y0, x0, n0; };
    u32x4 keep[4][NSTG];
    Coord cc;
    float xmax_cur;
    auto request = [&](const unsigned item_, const bool ok_item) __attribute__((always_inline)) {
        // ---- coordinates of the work item (scalars) and the per-thread addressing that follows from them ----
        {
            unsigned b_ = __builtin_amdgcn_readfirstlane(cnl::xcd_remap(item_, (unsigned)a.blocks));
            unsigned q_, nbi_, bxi_, byi_;
            W10_DIVMOD(q_, nbi_, b_, a.nb, a.m_nb); b_ = q_;
            W10_DIVMOD(q_, bxi_, b_, a.bx, a.m_bx); b_ = q_;
            W10_DIVMOD(q_, byi_, b_, a.by, a.m_by);
            cc.n = (int)q_; cc.y0 = (int)byi_ * R; cc.x0 = (int)bxi_ * (2 * TW); cc.n0 = (int)nbi_ * BN;
        }
        {
            st.cur.y0m1 = cc.y0 - 1;
            st.cur.img_base = (unsigned)(cc.n * a.ipb * a.Hs) * st.row_pitch;
            int tid_ = tid;
            asm volatile("" : "+v"(tid_));      // keeps the per-thread decode inside the item loop
            const int q_ = tid_ & 3, ix_ = cc.x0 - 1 + (tid_ >> 2);
            const int er_ = tid_ >> 3, ex_ = cc.x0 + 63 + ((tid_ >> 2) & 1), ey_ = cc.y0 - 1 + er_;
            if constexpr (PK) { // packed rows: virtual column -> (image, pixel of its strip); the strip's last two columns, a column before the
                               // first strip or behind the last are the zero padding (out of range -> zeros)
                unsigned si_, px_, esi_, epx_;
                W10_VDIVMOD(si_, px_, (unsigned)ix_, a.pk, a.m_pk);
                const bool okc_ = ix_ >= 0 && (int)px_ < a.W && (int)si_ < a.Nimg;
                const int ush_ = up ? 1 : 0;       // a folded nearest-2x upsample: logical (row, column) -> stored (row >> 1, column >> 1)
                st.cur.vcol = okc_ ? (unsigned)((((int)si_ * a.Hs * a.Ws + ((int)px_ >> ush_)) * a.ldx + q_ * 4) * 4) : OOB;
                W10_VDIVMOD(esi_, epx_, (unsigned)ex_, a.pk, a.m_pk);
                const bool oke_ = tid_ < 8 * PR && (unsigned)ey_ < (unsigned)a.H && (int)epx_ < a.W && (int)esi_ < a.Nimg;
                st.cur.vext = oke_ ? (unsigned)(((((int)esi_ * a.Hs + (ey_ >> ush_)) * a.Ws + ((int)epx_ >> ush_)) * a.ldx + q_ * 4) * 4) : OOB;
            } else if (a.ipb > 1) {   // block row = ipb images of width 2^lw side by side (x0 = 0): column -> (sub-image, pixel)
                const int si_ = ix_ >> a.lw, px_ = ix_ & (a.W - 1);
                const bool okc_ = (unsigned)ix_ < 64u && cc.n * a.ipb + si_ < a.Nimg;
                st.cur.vcol = okc_ ? (unsigned)(((si_ * a.H * a.W + px_) * a.ldx + q_ * 4) * 4) : OOB;
                const int esi_ = ex_ >> a.lw, epx_ = ex_ & (a.W - 1);
                const bool oke_ = tid_ < 8 * PR && (unsigned)ey_ < (unsigned)a.H && ex_ < 64 && cc.n * a.ipb + esi_ < a.Nimg;
                st.cur.vext = oke_ ? (unsigned)((((esi_ * a.H + ey_) * a.W + epx_) * a.ldx + q_ * 4) * 4) : OOB;
            } else {
                const int sx_ = up ? (ix_ >> 1) : ix_;
                st.cur.vcol = (unsigned)ix_ < (unsigned)a.W ? (unsigned)((sx_ * a.ldx + q_ * 4) * 4) : OOB;
                const bool ok_ = tid_ < 8 * PR && (unsigned)ey_ < (unsigned)a.H && (unsigned)ex_ < (unsigned)a.W;
                const int esy_ = up ? (ey_ >> 1) : ey_, esx_ = up ? (ex_ >> 1) : ex_;
                st.cur.vext = ok_ ? (unsigned)(((esy_ * a.Ws + esx_) * a.ldx + q_ * 4) * 4) : OOB;
            }
            st.cur.u_voff = (unsigned)((cc.n0 + (tid_ & 31)) * 32 + ((tid_ >> 5) & 1) * 16);
        }
        int img_lane = cc.n * a.ipb + si_lane;                 // the image of this lane's tile (V production)
        if constexpr (PK) {
            unsigned q_, r_;
            W10_VDIVMOD(q_, r_, (unsigned)(cc.x0 + 2 * (lane_now() & 31)), a.pk, a.m_pk);
            img_lane = (int)q_;
        }
        xmax_cur = W10_XMAX_OF(img_lane);
        {   // this item's bias and inverse weight scales: one value per lane, written to LDS inside the first chunk (slice 30)
            const int co = cc.n0 + lane_now();
            st.bst = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(__builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, (int)a.b_bytes, 0x00020000),
                                                                                    co < a.Cout ? (unsigned)co * 4u : OOB, 0, 0));
            st.ist = co < a.CoutP ? a.isu[co] : 0.f;
        }
        // patches 0 / 1 (all fourteen pieces requested before the first is written: one memory latency) and the weight rows 0 / 1 of chunk 0
#define W10_PLOAD_HALF(dst_, half_, cc_)                                                                         \
        do {                                                                                                     \
            pload<half_, 0>(st, a, cc_, up, ok_item); pload<half_, 1>(st, a, cc_, up, ok_item); pload<half_, 2>(st, a, cc_, up, ok_item); \
            if constexpr (half_ == 0) pload<0, 3>(st, a, cc_, up, ok_item);                                      \
            _Pragma("unroll") for (int i = 0; i < NSTG; ++i) keep[dst_][i] = st.stg[i];                          \
        } while (0)
        W10_PLOAD_HALF(0, 0, 0);
        W10_PLOAD_HALF(1, 1, 0);
        W10_PLOAD_HALF(2, 0, 1);
        W10_PLOAD_HALF(3, 1, 1);
#undef W10_PLOAD_HALF
#pragma unroll
        for (int i = 0; i < 2 * NBH; ++i) load_b<0>(st, a, 0, i, u_plane, u_wave, ok_item);
#pragma unroll
        for (int i = 0; i < 2 * NBH; ++i) load_b<1>(st, a, 0, i, u_plane, u_wave, ok_item);
    };
    unsigned item = blockIdx.x;
    request(item, true);
    while (true) {
        W10_STAMP(0);
        const Coord ci = cc;                // this item's coordinates (cc is overwritten with the next item's inside the epilogue)
        // ---- prologue: patches 0 / 1 -> LDS, V rows 0 / 1 of chunk 0, the raw reads of row 2 ----
        {
            W10_STAMP(1);
            {
                int es_cur;
                W10_SCALE_EXP(es_cur, xmax_cur);
                st.cur.S = __builtin_ldexpf(1.f, es_cur);
            }
            W10_BARRIER();                  // the previous item's last exchange pass has been read by every wave: the region is free
            W10_STAMP(2);
#pragma unroll
            for (int i = 0; i < NSTG; ++i) st.stg[i] = keep[0][i];
            pwrite<0, 0>(st, 0); pwrite<0, 1>(st, 0); pwrite<0, 2>(st, 0); pwrite<0, 3>(st, 0);
#pragma unroll
            for (int i = 0; i < 3; ++i) st.stg[i] = keep[1][i];
            pwrite<1, 0>(st, 0); pwrite<1, 1>(st, 0); pwrite<1, 2>(st, 0);
#pragma unroll
            for (int i = 0; i < NSTG; ++i) st.stg[i] = keep[2][i];
            pwrite<0, 0>(st, 1); pwrite<0, 1>(st, 1); pwrite<0, 2>(st, 1); pwrite<0, 3>(st, 1);
#pragma unroll
            for (int i = 0; i < 3; ++i) st.stg[i] = keep[3][i];
            pwrite<1, 0>(st, 1); pwrite<1, 1>(st, 1); pwrite<1, 2>(st, 1);
        }
        W10_BARRIER();
        W10_STAMP(3);
        rread<0>(st, 0, 0); rread<1>(st, 0, 0); rread<2>(st, 0, 0); rread<3>(st, 0, 0);
        job_all(st, 0, std::make_integer_sequence<int, 28>{});
        rread<0>(st, 0, 1); rread<1>(st, 0, 1); rread<2>(st, 0, 1); rread<3>(st, 0, 1);
        job_all(st, 1, std::make_integer_sequence<int, 28>{});
        rread<0>(st, 0, 2); rread<1>(st, 0, 2); rread<2>(st, 0, 2); rread<3>(st, 0, 2);

        W10_STAMP(4);
        chunk<0, true, NBH>(st, a, 0, up, u_plane, u_wave);
        W10_STAMP(8);
        chunk<1, false, NBH>(st, a, 1, up, u_plane, u_wave);
        W10_STAMP(9);
        for (int cn = 2; cn < a.CC; cn += 2) {
            chunk<0, false, NBH>(st, a, cn, up, u_plane, u_wave);
            chunk<1, false, NBH>(st, a, cn + 1, up, u_plane, u_wave);
        }
        W10_STAMP(5);
        W10_BARRIER();                      // every wave is done with the patch buffers: the exchange region may overwrite them
        W10_STAMP(6);

        // ---- epilogue (winograd9.hip's, four passes): out0 = Y0 + Y1 + Y2, out1 = Y1 - Y2 - Y3; the four positions (waves) meet through LDS.
        // Pass j = output row j: every wave writes its two blocks (cout halves) of that row into one half of the exchange region as
        // [cout half][position][tile][16-byte piece = 4 couts], pieces XOR-swizzled by the tile; wave w finishes cout half w & 1 of tiles
        // 16 (w >> 1) .. + 15 for all four positions with thread = (tile, piece): the 8 lanes of a tile store one full 128-byte line
        // (32-cout items: one cout half, wave w finishes tiles 8 w .. + 7) ----
        constexpr int NI = NBH;             // groups of 8 tiles a wave finishes per pass
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int t_e = lane_e & 31, h_e = lane_e >> 5;
        const int g_e = NBH == 2 ? (wave & 1) : 0;
        const int piece_e = lane_e & 7;
        const int cout_e = ci.n0 + g_e * 32 + piece_e * 4;            // this thread's four couts
        const bool cok_e = cout_e < a.Cout;
        const f32x4 bq = lds_f4(st.sB + (g_e * 32 + piece_e * 4) * 4);
        const f32x4 isu_e = lds_f4(st.sB + 256 + (g_e * 32 + piece_e * 4) * 4);
        const int wslot0 = t_e * 8, wsw = t_e & 7;
        int rslot[NI], rimg[NI], rpx[NI];
        f32x4 iq[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int rtile = (NBH == 2 ? 16 * (wave >> 1) + 8 * i : 8 * wave) + (lane_e >> 3);
            rslot[i] = rtile * 8 + (piece_e ^ (rtile & 7));
            const int si = a.ipb > 1 ? ((2 * rtile) >> a.lw) : 0;
            rimg[i] = ci.n * a.ipb + si;
            rpx[i] = a.ipb > 1 ? ((2 * rtile) & (a.W - 1)) : ci.x0 + 2 * rtile;
            if constexpr (PK) {   // packed rows: the tile's virtual column -> (image, pixel); a strip's two padding columns (pixel >= W) are not stored
                unsigned q_, r_;
                W10_VDIVMOD(q_, r_, (unsigned)(ci.x0 + 2 * rtile), a.pk, a.m_pk);
                rimg[i] = (int)q_; rpx[i] = (int)r_;
            }
            if (a.ipb > 1 || PK) {        // the tile's image is not the one this lane builds V for: its scale from its maximum
                int es_i;
                W10_SCALE_EXP(es_i, W10_XMAX_OF(rimg[i]));
                iq[i] = isu_e * __builtin_ldexpf(1.f, -es_i);
            } else {              // 1 / S from the exponent of the scale in use (S = 2^e exactly)
                iq[i] = isu_e * __builtin_bit_cast(float, 0x7F000000u - __builtin_bit_cast(unsigned, st.cur.S));
            }
        }
        float omax2[NI];
        unsigned yv0[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) omax2[i] = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) yv0[i] = ((unsigned)((rimg[i] * a.H + ci.y0) * a.W + rpx[i]) * (unsigned)a.ldy + (unsigned)cout_e) * 4u;
        const unsigned y_row = (unsigned)(a.W * a.ldy) * 4u;
        // what the images' max |y| slots hold so far: requested HERE, ahead of the item's stores, read behind the last pass (cnl::peek_max)
        unsigned yseen[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) yseen[i] = a.ymax ? cnl::peek_max(a.ymax + (rimg[i] < a.Nimg ? rimg[i] : 0) * AMS) : 0u;
#define W10_XWRITE2(j_, g_, q0_)                                                                                 \
        _Pragma("unroll") for (int q = (q0_); q < (q0_) + 2; ++q) {                                              \
            const f32x16& A = st.acc[j_][g_];                                                                    \
            *reinterpret_cast<f32x4*>(sX + ((j_) & 1) * (X_BYTES / 2) + (((g_) * 4 + wave) * 256 + wslot0 + ((2 * q + h_e) ^ wsw)) * 16) = \
                f32x4{A[4 * q], A[4 * q + 1], A[4 * q + 2], A[4 * q + 3]};                                       \
        }
#define W10_XWRITE(j_) do { W10_XWRITE2(j_, 0, 0); W10_XWRITE2(j_, 0, 2); if constexpr (NBH == 2) { W10_XWRITE2(j_, NBH - 1, 0); W10_XWRITE2(j_, NBH - 1, 2); } } while (0)
        const unsigned next = item + gridDim.x;
        const bool more = next < (unsigned)a.blocks;
        f32x4 rvl[R - 2][NI][2];           // residual values of rows 2, 3 (RES)
        f32x2 iql[NI], iqh[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) { iql[i] = f32x2{iq[i][0], iq[i][1]}; iqh[i] = f32x2{iq[i][2], iq[i][3]}; }
        const f32x2 bql = {bq[0], bq[1]}, bqh = {bq[2], bq[3]};
        W10_XWRITE(0);
        W10_BARRIER();
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const char* X = sX + (j & 1) * (X_BYTES / 2);
            const int oy = ci.y0 + j;
            const bool row_ok = oy < a.H && cok_e;
            unsigned yv[NI];
            bool ok[NI][2];
            f32x4 rv[NI][2];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int ox = rpx[i];
                yv[i] = yv0[i] + (unsigned)j * y_row;
                ok[i][0] = row_ok && ox < a.W && rimg[i] < a.Nimg; ok[i][1] = row_ok && ox + 1 < a.W && rimg[i] < a.Nimg;
                if constexpr (RES) {
                    if (j < 2) {
                        const unsigned rvo = ((unsigned)((rimg[i] * a.H + oy) * a.W + ox) * (unsigned)a.ldr + (unsigned)cout_e) * 4u;
#pragma unroll
                        for (int px = 0; px < 2; ++px)
                            rv[i][px] = __builtin_bit_cast(f32x4, buf_load16(a.res, a.r_bytes, ok[i][px] ? rvo : OOB, (unsigned)(px * a.ldr * 4)));
                    } else {      // rows 2, 3: requested behind pass 1, AHEAD of the next item's requests (loads return in order)
#pragma unroll
                        for (int px = 0; px < 2; ++px) rv[i][px] = rvl[j - 2][i][px];
                    }
                }
            }
            f32x4 Y[NI][4];
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int p = 0; p < 4; ++p) Y[i][p] = lds_f4(X + ((g_e * 4 + p) * 256 + rslot[i]) * 16);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
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
                        if (i == 0 && hh == 0) { W10_XWRITE2(j + 1, 0, 0); }
                        if (i == 0 && hh == 1) { W10_XWRITE2(j + 1, 0, 2); }
                        if constexpr (NBH == 2) {
                            if (i == 1 && hh == 0) { W10_XWRITE2(j + 1, NBH - 1, 0); }
                            if (i == 1 && hh == 1) { W10_XWRITE2(j + 1, NBH - 1, 2); }
                        }
                    }
                }
                if (ok[i][0]) omax2[i] = fmaxf(omax2[i], fmaxf(fmaxf(fabsf(o0[0]), fabsf(o0[1])), fmaxf(fabsf(o0[2]), fabsf(o0[3]))));
                if (ok[i][1]) omax2[i] = fmaxf(omax2[i], fmaxf(fmaxf(fabsf(o1[0]), fabsf(o1[1])), fmaxf(fabsf(o1[2]), fabsf(o1[3]))));
                buf_store16(o0, a.y, a.y_bytes, ok[i][0] ? yv[i] : OOB, 0);
                buf_store16(o1, a.y, a.y_bytes, ok[i][1] ? yv[i] : OOB, (unsigned)(a.ldy * 4));
            }
            if (j + 1 < R) { W10_BARRIER(); }
            __builtin_amdgcn_sched_barrier(0);
            if (j == 1) {
                // three of the four accumulator rows are dead (row 3 went into the exchange region during this pass... row 2 during pass 0): the
                // next item's requests go out here and fly during passes 2 and 3 — first the residual rows those passes add
                if constexpr (RES) {
#pragma unroll
                    for (int jj = 2; jj < R; ++jj)
#pragma unroll
                        for (int i = 0; i < NI; ++i) {
                            const bool okr = ci.y0 + jj < a.H && cok_e && rimg[i] < a.Nimg;
                            const unsigned rvo = ((unsigned)((rimg[i] * a.H + ci.y0 + jj) * a.W + rpx[i]) * (unsigned)a.ldr + (unsigned)cout_e) * 4u;
#pragma unroll
                            for (int px = 0; px < 2; ++px)
                                rvl[jj - 2][i][px] = __builtin_bit_cast(f32x4, buf_load16(a.res, a.r_bytes, (okr && rpx[i] + px < a.W) ? rvo : OOB, (unsigned)(px * a.ldr * 4)));
                        }
                }
                request(more ? next : item, more);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#undef W10_XWRITE
#undef W10_XWRITE2
        if (a.ymax) {          // max |y| of this item into its image's slot: the 8 tiles of an iteration lie in one image ...
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int img = __builtin_amdgcn_readfirstlane(rimg[i]);
                if constexpr (PK) {    // ... or, in packed rows, in two neighbouring strips (pk >= 16): the first tile's image and the one behind it
                    const float m1 = cnl::wave_max_nonneg(rimg[i] != img ? omax2[i] : 0.f);
                    if (lane_e == 0 && img + 1 < a.Nimg) cnl::report_max(a.ymax + (img + 1) * AMS, m1);
                    omax2[i] = rimg[i] == img ? omax2[i] : 0.f;
                }
                const float m = cnl::wave_max_nonneg(omax2[i]);
                if (lane_e == 0 && img < a.Nimg) cnl::raise_max(a.ymax + img * AMS, m, (unsigned)__builtin_amdgcn_readfirstlane((int)yseen[i]));
            }
        }
        W10_STAMP(7);
#ifdef W10_TRACE
        ++tr_item;
        ++tr_n;
#endif
        if (!more) break;
        item = next;
    }
#ifdef W10_TRACE
    if (tid == 0) {
        trb[1] = __builtin_readcyclecounter();
        trb[3] = __builtin_amdgcn_s_memrealtime();
        trb[6] = tr_n;
    }
#endif
#undef W10_DIVMOD
#undef W10_VDIVMOD
#undef W10_SCALE_EXP
#undef W10_XMAX_OF
}

}  // namespace cnl_wino10

#ifdef W10_TRACE
static unsigned long long* g_w10_trace = nullptr;
extern "C" __attribute__((visibility("default"))) void cnl_w10_set_trace(void* p) { g_w10_trace = (unsigned long long*)p; }
#endif
// can this kernel run the layer at all?  (the conditions of winograd9.hip: it reads that kernel's weights)
bool cnl_wino9_eligible(const cnl_conv_params* p);
size_t cnl_wino9_weight_bytes(int Cin, int Cout);
bool cnl_wino10_eligible(const cnl_conv_params* p) { return cnl_wino9_eligible(p); }
int cnl_wino_packed_stride(const cnl_conv_params* p);

// Launch (arguments already validated by cnl_conv3x3_winograd_f32); u9 / isu: the weight pieces and per-cout scales of winograd9.hip;
// xmax = N per-image maxima of the input.
template <int NBH>
static int wino10_launch(const cnl_conv_params* p, const void* u9, const float* isu, const float* xmax, void* stream) {
    using namespace cnl_wino10;
    constexpr int BN = 32 * NBH;
    Args a;
    a.x = p->x; a.u9 = u9; a.xmax = xmax; a.isu = isu; a.ymax = reinterpret_cast<unsigned*>(p->y_absmax);
    a.bias = p->bias; a.res = p->residual; a.y = p->y;
    const int upf = (p->flags & CNL_UPSAMPLE_IN) ? 2 : 1;
    a.Nimg = p->N; a.Hs = p->H_in; a.Ws = p->W_in; a.H = p->H_in * upf; a.W = p->W_in * upf; a.Cin = p->Cin; a.Cout = p->Cout;
    // narrow maps: 2 (W = 32) or 4 (W = 16) images side by side in one 64-pixel block row (no folded upsample there)
    a.ipb = (upf == 1 && (a.W == 32 || a.W == 16)) ? 64 / a.W : 1;
    a.lw = a.W == 32 ? 5 : 4;
    a.N = (p->N + a.ipb - 1) / a.ipb;
    // other widths that 64-pixel blocks pad: packed rows (cnl_wino_packed_stride, winograd9.hip) — same arithmetic chain per output, same bits
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
    const unsigned long long ub = (unsigned long long)cnl_wino9_weight_bytes(p->Cin, p->Cout);
    const unsigned long long Mo = (unsigned long long)p->N * a.H * a.W;
    const unsigned long long yb = ((Mo - 1) * p->ldy + p->Cout) * 4ull;
    const unsigned long long rb = p->residual ? ((Mo - 1) * p->ldr + p->Cout) * 4ull : 0ull;
    CNL_REQUIRE(xb < 0xFFFFFF00ull && ub < 0xFFFFFF00ull && yb + 4ull * p->ldy < 0xFFFFFF00ull && rb + 4ull * (p->residual ? p->ldr : 0) < 0xFFFFFF00ull,
                CNL_E_UNSUPPORTED, "cnl_conv3x3_winograd_f32: tensor spans >= 4 GiB; split the batch");
    a.x_bytes = (unsigned)xb; a.u_bytes = (unsigned)ub; a.y_bytes = (unsigned)yb; a.r_bytes = (unsigned)rb; a.b_bytes = (unsigned)p->Cout * 4u;
    a.flags = p->flags;
#ifdef W10_TRACE
    a.trace = g_w10_trace;
#endif
    static cnl::DeviceOnce once, once_res;
    int n_cu = 0;                          // persistent workgroups: two per CU, walking the work items with stride gridDim.x
    static cnl::DeviceOnce once_pk, once_res_pk;
    int rc = a.pk ? (p->residual ? cnl::kernel_setup(once_res_pk, reinterpret_cast<const void*>(&winograd10_kernel<true, NBH, true>), LDS_BYTES, &n_cu)
                                 : cnl::kernel_setup(once_pk, reinterpret_cast<const void*>(&winograd10_kernel<false, NBH, true>), LDS_BYTES, &n_cu))
                  : (p->residual ? cnl::kernel_setup(once_res, reinterpret_cast<const void*>(&winograd10_kernel<true, NBH>), LDS_BYTES, &n_cu)
                                 : cnl::kernel_setup(once, reinterpret_cast<const void*>(&winograd10_kernel<false, NBH>), LDS_BYTES, &n_cu));
    if (rc != CNL_OK) return rc;
    const unsigned grid = (unsigned)(blocks < 2ll * n_cu ? blocks : 2ll * n_cu);
    if (a.pk) {
        if (p->residual) hipLaunchKernelGGL((winograd10_kernel<true, NBH, true>), dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((winograd10_kernel<false, NBH, true>), dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, a);
    } else if (p->residual) hipLaunchKernelGGL((winograd10_kernel<true, NBH>), dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((winograd10_kernel<false, NBH>), dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, a);
    return cnl::check_launch("winograd10_kernel");
}
// cout32: 4-row x 64-pixel x 32-cout work items (twice the items of half the size: 16-pixel maps, one-image batches) instead of 64-cout ones
int cnl_wino_images_per_launch(const cnl_conv_params* p);                                      // winograd9.hip: tensors of >= 4 GiB run in groups of images
void cnl_wino_sub_batch(const cnl_conv_params* p, int n0, int n, cnl_conv_params* q, const float** xmax);
int cnl_wino10_launch(const cnl_conv_params* p, const void* u9, const float* isu, const float* xmax, bool cout32, void* stream) {
    const int per = cnl_wino_images_per_launch(p);
    CNL_REQUIRE(per > 0, CNL_E_UNSUPPORTED, "cnl_conv3x3_winograd_f32: one image of a tensor spans >= 4 GiB");
    for (int n0 = 0; n0 < p->N; n0 += per) {
        cnl_conv_params q;
        const float* xm = xmax;
        cnl_wino_sub_batch(p, n0, p->N - n0 < per ? p->N - n0 : per, &q, &xm);
        const int rc = cout32 ? wino10_launch<1>(&q, u9, isu, xm, stream) : wino10_launch<2>(&q, u9, isu, xm, stream);
        if (rc != CNL_OK) return rc;
    }
    return CNL_OK;
}
