// winograd12.hip — the row-Winograd arithmetic of winograd9.hip / winograd10.hip (1-D F(2,3) along x, three kernel rows folded into the reduction,
// fp32 products on the FP16 matrix cores from scaled two-way splits; the same weights, the same chain of fp32 additions per accumulator: the
// same bits) for the SHORT channel loops: Cin = 64 (four 16-channel chunks per work item), where winograd9 spends a third of every work item
// in its epilogue with the matrix pipe idle (12 K of 34 K cycles: the four transform positions meet through LDS in eight dependent passes).
//
//   item = 4 output rows x 64 pixels x 64 couts (winograd10's), ONE workgroup per CU, and TWO sets of 128 accumulator registers per lane:
//   while the MFMAs of item i fill one set, the epilogue of item i - 1 drains the other — its pass j (output row j: exchange writes before
//   the chunk's barrier, exchange reads / output transform / stores behind it) is issued in the MFMA shadows of chunk j of item i, like
//   the V production.  The chunk stream runs on across work items (winograd9's MODE 1 / 2: the last two chunks of an item request the next
//   item's first two patches, its first weight fragments and build its first two V rows), so a workgroup's matrix pipe only idles in
//   the prologue of its first item and the epilogue of its last.
//
// Schedule of a chunk: winograd10.hip's (72 MFMAs per wave, V jobs, weight and patch traffic), plus per chunk j < 4 the epilogue pass j of the
// previous item:  slices 1..8 eight exchange writes (one output row of the other accumulator set), 3..9 the residual requests, 11 a barrier
// of their own, 12..19 eight exchange reads, every other slice of 22..36 / 42..56 the output transform of the two tile groups in eight
// micro-steps each, 37, 38 / 57, 58 the stores.  The exchange region no longer aliases the patch buffers (118 KB of LDS).
//
// EXPERIMENT (round 5, `make -C csrc experiments`, algo = CNL_ALGO_FORCE + 12): correct — bit-identical to winograd9 on every test shape with
// Cin = 64 (tests/test_gpu_conv.py's row-Winograd cases run with variant 12 against the experiment library) — and NOT faster: both schedules of
// the riding epilogue (behind the chunk's barrier; behind a barrier of its own) run a chunk in ~5 K cycles against 2.3 K of matrix issue, so the
// hidden epilogue costs as much as winograd9's exposed one: layer1 64->64 @128^2 x 32  92-95 us (winograd9 87-90), + residual 115 (106-112), first
// head blocks 64->512 behind the upsample 682-693 (624-637), 152 x 272 x 64 -> 64 + residual 140 (148-150), 64 -> 256 there 457 (430-435)
// (profiles/r05_experiments.txt r5f).  Three designs — one workgroup with an exposed epilogue (9), two workgroups sharing a CU (10), one
// workgroup with two accumulator sets (this) — land within 10 % of each other on the 64-channel layers.
#include "cnl_common.h"
#include <utility>

#pragma clang fp contract(off)

int cnl_wino_packed_stride(const cnl_conv_params* p);
namespace cnl_wino12 {

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
    int nb, bx, by;                   // blocks along cout (64), x (64 px), y (4 rows)
    unsigned m_nb, m_bx, m_by;        // floor(2^32 / d) of the three
    int ipb, lw;                      // images side by side in one 64-pixel block row (W = 32: 2, W = 16: 4; else 1) and log2 W for them
    int Nimg;                         // images of the launch (N = image groups)
    int pk;                           // packed rows (winograd9.hip: cnl_wino_packed_stride), 0: off
    unsigned m_pk;
    int blocks;
    unsigned x_bytes, u_bytes, y_bytes, r_bytes, b_bytes;
    unsigned flags;
};

constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr int R = 4;                        // output rows per work item
constexpr int PR = R + 2;                   // patch rows
constexpr int TW = 32;                      // tiles (pixel pairs) per row of a work item: 64 output pixels
constexpr int BN = 64;                      // couts per work item
constexpr int CC = 4;                       // chunks per work item: Cin = 64 (the epilogue pass j of the previous item rides in chunk j)
constexpr int PXH = TW + 1;
constexpr int QUAD_SLOTS = 2 * PXH;
constexpr int ROW_SLOTS = 4 * QUAD_SLOTS;
constexpr int ROW_BYTES = ROW_SLOTS * 16;
constexpr int P_SLOTS = PR * ROW_SLOTS + 48;
constexpr int P_BYTES = P_SLOTS * 16;       // 26112 per patch buffer (two buffers)
constexpr int X_OFF = 2 * P_BYTES;          // exchange region: two halves of [2 cout halves][4 positions][32 tiles][8 pieces] x 16 B
constexpr int X_BYTES = 65536;
constexpr int B_OFF = X_OFF + X_BYTES;      // two sets (item parity) of 64 bias values + 64 inverse weight scales
constexpr int B_SET = 512;
constexpr int LDS_BYTES = B_OFF + 2 * B_SET;        // 118784
constexpr int NSLICE = 72;
constexpr int JOB0 = 6, JOB_SLICES = 12;
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
#define W12_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// ---- the static schedule of a chunk (winograd10.hip's) ----
constexpr int NSEG = 12;
constexpr int SEG_ROW[NSEG] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 5};
constexpr int SEG_KY[NSEG] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 1, 2, 2};
constexpr int KY0_DEAD = 42, KY1_DEAD = 60;

struct Item {               // per-work-item addressing state (two live: the item being multiplied and the one after it)
    unsigned vcol, vext;
    unsigned u_voff;
    unsigned img_base;
    int y0m1;
    float S;
};
struct Coord { int n, y0, x0, n0; };
// what the epilogue of a work item needs once its chunks are done (it runs during the NEXT item's chunks)
struct Epi {
    f32x4 bq;                // bias of this thread's four couts
    f32x4 iq[2];             // 1 / (S_v S_u) of them, per tile group (the group's image)
    int rimg[2], rpx[2];     // image and first pixel column of this thread's tile, per tile group
    unsigned yv0[2];         // byte offset of that pixel in output row y0
    unsigned yseen[2];
    float omax2[2];
    int y0, cout_e;
};
struct State {
    f32x16 acc[2][R][2];     // [set][output row][cout half]: D[cout][tile]
    u32x4 fb[3][2][2];       // weight fragments (A operand): [ky][cout half][piece], single-buffered
    u32x4 vf[3][2];          // V fragments (B operand): [row % 3][piece]
    f32x4 raw[4];
    float v[8];
    u32x4 stg[NSTG];
    Item cur, nxt;
    unsigned row_pitch;
    const char* pa;
    const char* pb;
    char* wb;
    char* wext;
    float sg;
    float bst, ist;
    char* smem;
    // the running epilogue pass
    f32x4 Y[2][4];           // exchange reads: [tile group][position]
    f32x4 rv[2][2];          // residual values: [tile group][pixel]
    f32x4 o0[2], o1[2];      // finished outputs of the two pixels of a tile, per tile group
    unsigned yv[2];
    bool ok[2][2];
};

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
        asm volatile("" : "+v"(st.vf[buf][1][j]));
    }
}
template <int I>
__device__ __forceinline__ void rread(State& st, const int pbuf, const int row) {
    const char* p = (I < 2 ? st.pa : st.pb) + pbuf * P_BYTES + row * ROW_BYTES + (I & 1) * (QUAD_SLOTS * 16);
    st.raw[I] = lds_f4(p);
}
template <int KY>
__device__ __forceinline__ void load_b(State& st, const Args& a, const unsigned u_voff, const int cc, const int i, const unsigned u_plane, const unsigned u_wave,
                                       const bool ok) {
    const int nbh = i >> 1, piece = i & 1;
    const unsigned so = (unsigned)cc * (24u * u_plane) + u_wave + (unsigned)(KY * 2 + piece) * u_plane + (unsigned)nbh * 1024u;
    st.fb[KY][nbh][piece] = buf_load16(a.u9, a.u_bytes, ok ? u_voff : OOB, so);
}
template <int HALF, int I>
__device__ __forceinline__ void pload(State& st, const Item& it, const Args& a, const int cc, const bool up, const bool ok) {
    if constexpr (I < 3) {
        const int iy = __builtin_amdgcn_readfirstlane(it.y0m1) + 3 * HALF + I;
        const bool okr = ok && (unsigned)iy < (unsigned)a.H;
        const int sy = okr ? (up ? (iy >> 1) : iy) : 0;
        const unsigned so = __builtin_amdgcn_readfirstlane(it.img_base + (unsigned)sy * st.row_pitch + (unsigned)cc * 64u);
        st.stg[I] = buf_load16(a.x, a.x_bytes, okr ? it.vcol : OOB, so);
    } else {
        static_assert(HALF == 0, "the column piece belongs to half A");
        st.stg[3] = buf_load16(a.x, a.x_bytes, ok ? it.vext : OOB, __builtin_amdgcn_readfirstlane(it.img_base + (unsigned)cc * 64u));
    }
}
template <int HALF, int I>
__device__ __forceinline__ void pwrite(State& st, const int pbuf) {
    if constexpr (I < 3) *reinterpret_cast<u32x4*>(st.wb + pbuf * P_BYTES + (3 * HALF + I) * ROW_BYTES) = st.stg[I];
    else *reinterpret_cast<u32x4*>(st.wext + pbuf * P_BYTES) = st.stg[3];
}

constexpr bool first_use(int S) {
    const int yo = SEG_ROW[S / 6] - SEG_KY[S / 6], nbh = S & 1;
    for (int s = 0; s < S; ++s)
        if (SEG_ROW[s / 6] - SEG_KY[s / 6] == yo && (s & 1) == nbh) return false;
    return true;
}

// per-lane constants of the epilogue (functions of the lane and the wave only)
struct EpiLane {
    int wslot0, wsw, h_e, g_e;
    int rslot[2];
};

// ---- the epilogue pass J of the item in accumulator set ESET, cut into the pieces a chunk issues in its slices ----
// exchange write q of cout half g (row J): position `wave`, this lane's tile, pieces 2 q + h
template <int ESET, int J, int G, int Q>
__device__ __forceinline__ void epi_xwrite(State& st, const EpiLane& el, const int wave) {
    const f32x16& A = st.acc[ESET][J][G];
    *reinterpret_cast<f32x4*>(st.smem + X_OFF + (J & 1) * (X_BYTES / 2) + ((G * 4 + wave) * 256 + el.wslot0 + ((2 * Q + el.h_e) ^ el.wsw)) * 16) =
        f32x4{A[4 * Q], A[4 * Q + 1], A[4 * Q + 2], A[4 * Q + 3]};
}
template <bool RES, int J>
__device__ __forceinline__ void epi_begin(State& st, const Epi& ep, const Args& a, const bool cok_e) {      // row masks, output offsets (and the residual requests)
    const int oy = ep.y0 + J;
    const bool row_ok = oy < a.H && cok_e;
    const unsigned y_row = (unsigned)(a.W * a.ldy) * 4u;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ox = ep.rpx[i];
        st.yv[i] = ep.yv0[i] + (unsigned)J * y_row;
        st.ok[i][0] = row_ok && ox < a.W && ep.rimg[i] < a.Nimg;
        st.ok[i][1] = row_ok && ox + 1 < a.W && ep.rimg[i] < a.Nimg;
    }
}
template <int J, int I, int PX>
__device__ __forceinline__ void epi_res(State& st, const Epi& ep, const Args& a) {
    const unsigned rvo = ((unsigned)((ep.rimg[I] * a.H + ep.y0 + J) * a.W + ep.rpx[I]) * (unsigned)a.ldr + (unsigned)ep.cout_e) * 4u;
    st.rv[I][PX] = __builtin_bit_cast(f32x4, buf_load16(a.res, a.r_bytes, st.ok[I][PX] ? rvo : OOB, (unsigned)(PX * a.ldr * 4)));
}
template <int J, int I, int P>
__device__ __forceinline__ void epi_yread(State& st, const EpiLane& el) {
    st.Y[I][P] = lds_f4(st.smem + X_OFF + (J & 1) * (X_BYTES / 2) + ((el.g_e * 4 + P) * 256 + el.rslot[I]) * 16);
}
// micro-step T (0..7) of the output transform of tile group I: out0 = Y0 + Y1 + Y2, out1 = Y1 - Y2 - Y3, x 1 / (S_v S_u) + bias (+ residual), ReLU
template <bool RES, int I, int T>
__device__ __forceinline__ void epi_step(State& st, Epi& ep, const float lo) {
    if constexpr (T == 0 || T == 1 || T == 3 || T == 4) {
        constexpr int hh = T >= 3 ? 1 : 0;
        constexpr bool B = (T == 1 || T == 4);
        const f32x2 y0 = {st.Y[I][0][2 * hh], st.Y[I][0][2 * hh + 1]}, y1 = {st.Y[I][1][2 * hh], st.Y[I][1][2 * hh + 1]};
        const f32x2 y2 = {st.Y[I][2][2 * hh], st.Y[I][2][2 * hh + 1]}, y3 = {st.Y[I][3][2 * hh], st.Y[I][3][2 * hh + 1]};
        const f32x2 sc = {ep.iq[I][2 * hh], ep.iq[I][2 * hh + 1]}, bb = {ep.bq[2 * hh], ep.bq[2 * hh + 1]};
        f32x2 r = B ? (y1 - y2 - y3) * sc + bb : (y0 + y1 + y2) * sc + bb;
        if constexpr (RES) r += f32x2{st.rv[I][B ? 1 : 0][2 * hh], st.rv[I][B ? 1 : 0][2 * hh + 1]};
        if constexpr (B) { st.o1[I][2 * hh] = r[0]; st.o1[I][2 * hh + 1] = r[1]; }
        else { st.o0[I][2 * hh] = r[0]; st.o0[I][2 * hh + 1] = r[1]; }
    } else if constexpr (T == 2 || T == 5) {
        constexpr int hh = T == 5 ? 1 : 0;
        st.o0[I][2 * hh] = fmaxf(st.o0[I][2 * hh], lo); st.o0[I][2 * hh + 1] = fmaxf(st.o0[I][2 * hh + 1], lo);
        st.o1[I][2 * hh] = fmaxf(st.o1[I][2 * hh], lo); st.o1[I][2 * hh + 1] = fmaxf(st.o1[I][2 * hh + 1], lo);
    } else if constexpr (T == 6) {
        if (st.ok[I][0]) ep.omax2[I] = fmaxf(ep.omax2[I], fmaxf(fmaxf(fabsf(st.o0[I][0]), fabsf(st.o0[I][1])), fmaxf(fabsf(st.o0[I][2]), fabsf(st.o0[I][3]))));
    } else {
        if (st.ok[I][1]) ep.omax2[I] = fmaxf(ep.omax2[I], fmaxf(fmaxf(fabsf(st.o1[I][0]), fabsf(st.o1[I][1])), fmaxf(fabsf(st.o1[I][2]), fabsf(st.o1[I][3]))));
    }
}
template <int I, int PX>
__device__ __forceinline__ void epi_store(State& st, const Args& a) {
    buf_store16(PX ? st.o1[I] : st.o0[I], a.y, a.y_bytes, st.ok[I][PX] ? st.yv[I] : OOB, PX ? (unsigned)(a.ldy * 4) : 0u);
}

// One slice: MFMA S of the chunk with parity PAR into accumulator set SET, and what is issued beside it.
//   MODE 0: a chunk with two more chunks of its item behind it; 1: the item's last but one (its patch request is chunk 0 of the NEXT item);
//   2: the item's last (requests chunk 1 of the next item, loads its chunk-0 weights, builds its first two V rows)
//   FIRST: the item's first chunk (accumulators start from C = 0);  PEEL: the workgroup's first item (its prologue wrote both patches and built rows 0 / 1)
//   EPI: the epilogue pass EPI of the previous item (set SET ^ 1) rides along, or -1
template <int S, int PAR, int MODE, bool FIRST, bool PEEL, int EPI, int SET, bool RES>
__device__ __forceinline__ void slice(State& st, Epi& ep, const EpiLane& el, const Args& a, const int cn, const bool up, const unsigned u_plane,
                                      const unsigned u_wave, const bool more, const int wave, const float lo, const bool cok_prev) {
    constexpr int seg = S / 6;
    constexpr int r = SEG_ROW[seg], ky = SEG_KY[seg];
    constexpr int term = (S % 6) / 2, nbh = S & 1;
    constexpr int ku = term == 1 ? 1 : 0, kv = term == 0 ? 1 : 0;
    constexpr int vbuf = r % 3;
    if constexpr (S == BARRIER_SLICE || (EPI >= 0 && S == 11)) {      // (11: the exchange writes of the riding epilogue pass are complete)
        W12_BARRIER();
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (FIRST && first_use(S)) st.acc[SET][r - ky][nbh] = mfma16(st.fb[ky][nbh][ku], st.vf[vbuf][kv], f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f});
    else st.acc[SET][r - ky][nbh] = mfma16(st.fb[ky][nbh][ku], st.vf[vbuf][kv], st.acc[SET][r - ky][nbh]);
    __builtin_amdgcn_sched_barrier(0);
    // ---- V production (job 5 wraps into the next chunk's first slices; jobs 4 / 5 of the item's last chunk build the NEXT item's rows 0 / 1) ----
    constexpr int G = (S + NSLICE - JOB0) % NSLICE;
    constexpr int j = G / JOB_SLICES, k = G % JOB_SLICES;
    if constexpr (!(PEEL && FIRST && S < JOB0) && k < 11) {
        constexpr int row = (j + 2) % 6;
        constexpr int buf = row % 3;
        // the scale: rows 0 / 1 built inside an item's last chunk (jobs 4, 5 from slice 54 on) belong to the next item; job 5's tail in a first chunk to this one
        const float Sj = (MODE == 2 && S >= JOB0 + 4 * JOB_SLICES) ? st.nxt.S : st.cur.S;
        if constexpr (k < 6) {
            vop<3 * k>(st, buf, Sj); vop<3 * k + 1>(st, buf, Sj); vop<3 * k + 2>(st, buf, Sj);
        } else {
            vop<18 + 2 * (k - 6)>(st, buf, Sj); vop<19 + 2 * (k - 6)>(st, buf, Sj);
        }
    }
    if constexpr (!(PEEL && FIRST && S < JOB0) && k >= 4 && k <= 7) {
        constexpr int jn = (j + 1) % 6;
        constexpr int nrow = (jn + 2) % 6;
        constexpr int npb = (jn >= 1 && jn <= 3) ? PAR : (jn >= 4 ? (PAR ^ 1) : (S >= JOB0 ? (PAR ^ 1) : PAR));
        rread<k - 4>(st, npb, nrow);
    }
    // ---- weight fragments: kernel row 2 of THIS chunk, rows 0 / 1 of the next chunk (of the next item's chunk 0 in MODE 2) ----
    if constexpr (S < 4) load_b<2>(st, a, st.cur.u_voff, cn, S, u_plane, u_wave, true);
    if constexpr (S >= KY0_DEAD && S < KY0_DEAD + 8 && (S - KY0_DEAD) % 2 == 0)
        load_b<0>(st, a, MODE == 2 ? st.nxt.u_voff : st.cur.u_voff, MODE == 2 ? 0 : cn + 1, (S - KY0_DEAD) / 2, u_plane, u_wave, MODE == 2 ? more : true);
    if constexpr (S >= KY1_DEAD && S < KY1_DEAD + 8 && (S - KY1_DEAD) % 2 == 0)
        load_b<1>(st, a, MODE == 2 ? st.nxt.u_voff : st.cur.u_voff, MODE == 2 ? 0 : cn + 1, (S - KY1_DEAD) / 2, u_plane, u_wave, MODE == 2 ? more : true);
    // ---- patch of the chunk after next (of the next item in MODE 1 / 2), in two halves through the same staging registers ----
    if constexpr (!(PEEL && FIRST) && S >= 10 && S <= 14 && (S - 10) % 2 == 0) pwrite<1, (S - 10) / 2>(st, PAR ^ 1);
    if constexpr (S >= 16 && S <= 28 && (S - 16) % 4 == 0) {
        if constexpr (MODE == 0) pload<0, (S - 16) / 4>(st, st.cur, a, cn + 2, up, true);
        else pload<0, (S - 16) / 4>(st, st.nxt, a, MODE - 1, up, more);
    }
    if constexpr (S >= 41 && S <= 47 && (S - 41) % 2 == 0) pwrite<0, (S - 41) / 2>(st, PAR);
    if constexpr (S >= 50 && S <= 58 && (S - 50) % 4 == 0) {
        if constexpr (MODE == 0) pload<1, (S - 50) / 4>(st, st.cur, a, cn + 2, up, true);
        else pload<1, (S - 50) / 4>(st, st.nxt, a, MODE - 1, up, more);
    }
    // ---- the item's bias / weight-scale values -> LDS (set SET), for its epilogue ----
    if constexpr (FIRST && S == 30) {
        float* sb = reinterpret_cast<float*>(st.smem + B_OFF + SET * B_SET) + lane_now();
        sb[0] = st.bst;
        sb[64] = st.ist;
    }
    // ---- epilogue pass EPI of the previous item (accumulator set SET ^ 1): exchange writes in slices 1..8, a barrier of their own at slice 11
    //      (the chunk's barrier at slice 40 leaves too few slices behind it: the first version, reads at 42..49 and the transform in 16 slices,
    //      ran a chunk in 5 K cycles), reads 12..19, the output transform in micro-steps on every other slice 22..56, stores 37 / 38 / 57 / 58 ----
    if constexpr (EPI >= 0) {
        constexpr int ES = SET ^ 1;
        if constexpr (S == 0) epi_begin<RES, EPI>(st, ep, a, cok_prev);
        if constexpr (S >= 1 && S <= 8) epi_xwrite<ES, EPI, (S - 1) / 4, (S - 1) % 4>(st, el, wave);
        if constexpr (RES && (S == 3 || S == 5 || S == 7 || S == 9)) epi_res<EPI, (S - 3) / 4, ((S - 3) / 2) & 1>(st, ep, a);
        if constexpr (S >= 12 && S <= 19) epi_yread<EPI, (S - 12) / 4, (S - 12) % 4>(st, el);
        if constexpr (S >= 22 && S <= 36 && (S - 22) % 2 == 0) epi_step<RES, 0, (S - 22) / 2>(st, ep, lo);
        if constexpr (S == 37) epi_store<0, 0>(st, a);
        if constexpr (S == 38) epi_store<0, 1>(st, a);
        if constexpr (S >= 42 && S <= 56 && (S - 42) % 2 == 0) epi_step<RES, 1, (S - 42) / 2>(st, ep, lo);
        if constexpr (S == 57) epi_store<1, 0>(st, a);
        if constexpr (S == 58) epi_store<1, 1>(st, a);
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int PAR, int MODE, bool FIRST, bool PEEL, int EPI, int SET, bool RES, int... S>
__device__ __forceinline__ void chunk_impl(State& st, Epi& ep, const EpiLane& el, const Args& a, const int cn, const bool up, const unsigned u_plane,
                                           const unsigned u_wave, const bool more, const int wave, const float lo, const bool cok_prev, std::integer_sequence<int, S...>) {
    __builtin_amdgcn_sched_barrier(0);
    (slice<S, PAR, MODE, FIRST, PEEL, EPI, SET, RES>(st, ep, el, a, cn, up, u_plane, u_wave, more, wave, lo, cok_prev), ...);
}
template <int PAR, int MODE, bool FIRST, bool PEEL, int EPI, int SET, bool RES>
__device__ __forceinline__ void chunk(State& st, Epi& ep, const EpiLane& el, const Args& a, const int cn, const bool up, const unsigned u_plane, const unsigned u_wave,
                                      const bool more, const int wave, const float lo, const bool cok_prev) {
    chunk_impl<PAR, MODE, FIRST, PEEL, EPI, SET, RES>(st, ep, el, a, cn, up, u_plane, u_wave, more, wave, lo, cok_prev, std::make_integer_sequence<int, NSLICE>{});
}
template <int... O>
__device__ __forceinline__ void job_all(State& st, const int buf, const float S, std::integer_sequence<int, O...>) {
    (vop<O>(st, buf, S), ...);
}

template <bool RES, bool PK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void winograd12_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, t = lane & 31;
    const bool up = a.flags & CNL_UPSAMPLE_IN;
    const unsigned u_plane = (unsigned)(a.CoutP * 32);
    const unsigned u_wave = (unsigned)wave * 6u * u_plane;

    State st;
    st.smem = smem;
    st.row_pitch = (unsigned)(a.Ws * a.ldx * 4);
    const int offa = wave == 0 ? 2 : (wave == 2 ? 2 : 1);
    const int offb = wave == 0 ? 0 : (wave == 3 ? 3 : (wave == 2 ? 1 : 2));
    const int tpi = (a.ipb > 1) ? (1 << (a.lw - 1)) : 64;
    const bool outer_is_neighbour = a.ipb > 1 && ((wave == 0 && (t & (tpi - 1)) == 0) || (wave == 3 && (t & (tpi - 1)) == tpi - 1));
    st.sg = wave == 1 ? 1.f : -1.f;
    const int si_lane = (a.ipb > 1) ? ((2 * t) >> a.lw) : 0;
    {
        const int sa = (2 * h) * QUAD_SLOTS + (offa & 1) * PXH + t + (offa >> 1);
        const int sb = outer_is_neighbour ? (2 * h) * QUAD_SLOTS : (2 * h) * QUAD_SLOTS + (offb & 1) * PXH + t + (offb >> 1);
        st.pa = smem + sa * 16;
        st.pb = smem + sb * 16;
    }
    {
        const int q = tid & 3, c = tid >> 2;
        st.wb = smem + (q * QUAD_SLOTS + (c & 1) * PXH + (c >> 1)) * 16;
        const int er = tid >> 3, ec = 64 + ((tid >> 2) & 1);
        st.wext = tid < 8 * PR ? smem + (er * ROW_SLOTS + q * QUAD_SLOTS + (ec & 1) * PXH + (ec >> 1)) * 16 : smem + (PR * ROW_SLOTS + (tid & 31)) * 16;
    }
    const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane((a.flags & CNL_RELU) ? 0 : (int)0xff800000u));
    EpiLane el;
    {
        el.wslot0 = t * 8; el.wsw = t & 7; el.h_e = h; el.g_e = wave & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rtile = 16 * (wave >> 1) + 8 * i + (lane >> 3);
            el.rslot[i] = rtile * 8 + ((lane & 7) ^ (rtile & 7));
        }
    }

#define W12_DIVMOD(q_, r_, b_, d_, m_)                                                                           \
    do {                                                                                                         \
        unsigned qq_ = __builtin_amdgcn_readfirstlane(__umulhi((b_), (m_)));                                     \
        unsigned rr_ = (b_) - qq_ * (unsigned)(d_);                                                              \
        if (rr_ >= (unsigned)(d_)) { ++qq_; rr_ -= (unsigned)(d_); }                                             \
        (q_) = qq_; (r_) = rr_;                                                                                  \
    } while (0)
#define W12_VDIVMOD(q_, r_, b_, d_, m_)                                                                          \
    do {                                                                                                         \
        unsigned qq_ = __umulhi((b_), (m_));                                                                     \
        unsigned rr_ = (b_) - qq_ * (unsigned)(d_);                                                              \
        if (rr_ >= (unsigned)(d_)) { ++qq_; rr_ -= (unsigned)(d_); }                                             \
        (q_) = qq_; (r_) = rr_;                                                                                  \
    } while (0)
#define W12_XMAX_OF(img_) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(                          \
        __builtin_amdgcn_make_buffer_rsrc((void*)a.xmax, 0, a.Nimg * 4 * AMS, 0x00020000), (unsigned)(img_) * (4u * AMS), 0, 0))
#define W12_SCALE_EXP(es_, xmax_)                                                                                \
    do {                                                                                                         \
        const float mx2_ = 2.f * (xmax_);                                                                        \
        (es_) = 0;                                                                                               \
        if (mx2_ > 0.f && mx2_ < __builtin_inff()) {                                                             \
            int e_;                                                                                              \
            (void)__builtin_frexpf(mx2_, &e_);                                                                   \
            e_ = 14 - e_;                                                                                        \
            (es_) = e_ < -100 ? -100 : (e_ > 100 ? 100 : e_);                                                    \
        }                                                                                                        \
    } while (0)
    // coordinates of a work item (scalars) and the per-thread addressing that follows from them
    auto coord_of = [&](const unsigned item_) __attribute__((always_inline)) {
        Coord c;
        unsigned b_ = __builtin_amdgcn_readfirstlane(cnl::xcd_remap(item_, (unsigned)a.blocks));
        unsigned q_, nbi_, bxi_, byi_;
        W12_DIVMOD(q_, nbi_, b_, a.nb, a.m_nb); b_ = q_;
        W12_DIVMOD(q_, bxi_, b_, a.bx, a.m_bx); b_ = q_;
        W12_DIVMOD(q_, byi_, b_, a.by, a.m_by);
        c.n = (int)q_; c.y0 = (int)byi_ * R; c.x0 = (int)bxi_ * (2 * TW); c.n0 = (int)nbi_ * BN;
        return c;
    };
    auto item_of = [&](Item& it, const Coord& c) __attribute__((always_inline)) {
        it.y0m1 = c.y0 - 1;
        it.img_base = (unsigned)(c.n * a.ipb * a.Hs) * st.row_pitch;
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));
        const int q_ = tid_ & 3, ix_ = c.x0 - 1 + (tid_ >> 2);
        const int er_ = tid_ >> 3, ex_ = c.x0 + 63 + ((tid_ >> 2) & 1), ey_ = c.y0 - 1 + er_;
        if constexpr (PK) {
            unsigned si_, px_, esi_, epx_;
            W12_VDIVMOD(si_, px_, (unsigned)ix_, a.pk, a.m_pk);
            const bool okc_ = ix_ >= 0 && (int)px_ < a.W && (int)si_ < a.Nimg;
            const int ush_ = up ? 1 : 0;
            it.vcol = okc_ ? (unsigned)((((int)si_ * a.Hs * a.Ws + ((int)px_ >> ush_)) * a.ldx + q_ * 4) * 4) : OOB;
            W12_VDIVMOD(esi_, epx_, (unsigned)ex_, a.pk, a.m_pk);
            const bool oke_ = tid_ < 8 * PR && (unsigned)ey_ < (unsigned)a.H && (int)epx_ < a.W && (int)esi_ < a.Nimg;
            it.vext = oke_ ? (unsigned)(((((int)esi_ * a.Hs + (ey_ >> ush_)) * a.Ws + ((int)epx_ >> ush_)) * a.ldx + q_ * 4) * 4) : OOB;
        } else if (a.ipb > 1) {
            const int si_ = ix_ >> a.lw, px_ = ix_ & (a.W - 1);
            const bool okc_ = (unsigned)ix_ < 64u && c.n * a.ipb + si_ < a.Nimg;
            it.vcol = okc_ ? (unsigned)(((si_ * a.H * a.W + px_) * a.ldx + q_ * 4) * 4) : OOB;
            const int esi_ = ex_ >> a.lw, epx_ = ex_ & (a.W - 1);
            const bool oke_ = tid_ < 8 * PR && (unsigned)ey_ < (unsigned)a.H && ex_ < 64 && c.n * a.ipb + esi_ < a.Nimg;
            it.vext = oke_ ? (unsigned)((((esi_ * a.H + ey_) * a.W + epx_) * a.ldx + q_ * 4) * 4) : OOB;
        } else {
            const int sx_ = up ? (ix_ >> 1) : ix_;
            it.vcol = (unsigned)ix_ < (unsigned)a.W ? (unsigned)((sx_ * a.ldx + q_ * 4) * 4) : OOB;
            const bool ok_ = tid_ < 8 * PR && (unsigned)ey_ < (unsigned)a.H && (unsigned)ex_ < (unsigned)a.W;
            const int esy_ = up ? (ey_ >> 1) : ey_, esx_ = up ? (ex_ >> 1) : ex_;
            it.vext = ok_ ? (unsigned)(((esy_ * a.Ws + esx_) * a.ldx + q_ * 4) * 4) : OOB;
        }
        it.u_voff = (unsigned)((c.n0 + (tid_ & 31)) * 32 + ((tid_ >> 5) & 1) * 16);
    };
    auto img_lane_of = [&](const Coord& c) __attribute__((always_inline)) {
        int img = c.n * a.ipb + si_lane;
        if constexpr (PK) {
            unsigned q_, r_;
            W12_VDIVMOD(q_, r_, (unsigned)(c.x0 + 2 * (lane_now() & 31)), a.pk, a.m_pk);
            img = (int)q_;
        }
        return img;
    };
    // the epilogue's per-item constants (the item's bias / weight scales were staged in LDS set `set` during its first chunk)
    auto epi_of = [&](Epi& ep, const Coord& c, const int set, const float S_item) __attribute__((always_inline)) {
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int piece_e = lane_e & 7;
        ep.y0 = c.y0;
        ep.cout_e = c.n0 + el.g_e * 32 + piece_e * 4;
        ep.bq = lds_f4(smem + B_OFF + set * B_SET + (el.g_e * 32 + piece_e * 4) * 4);
        const f32x4 isu_e = lds_f4(smem + B_OFF + set * B_SET + 256 + (el.g_e * 32 + piece_e * 4) * 4);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rtile = 16 * (wave >> 1) + 8 * i + (lane_e >> 3);
            const int si = a.ipb > 1 ? ((2 * rtile) >> a.lw) : 0;
            ep.rimg[i] = c.n * a.ipb + si;
            ep.rpx[i] = a.ipb > 1 ? ((2 * rtile) & (a.W - 1)) : c.x0 + 2 * rtile;
            if constexpr (PK) {
                unsigned q_, r_;
                W12_VDIVMOD(q_, r_, (unsigned)(c.x0 + 2 * rtile), a.pk, a.m_pk);
                ep.rimg[i] = (int)q_; ep.rpx[i] = (int)r_;
            }
            if (a.ipb > 1 || PK) {
                int es_i;
                W12_SCALE_EXP(es_i, W12_XMAX_OF(ep.rimg[i]));
                ep.iq[i] = isu_e * __builtin_ldexpf(1.f, -es_i);
            } else {
                ep.iq[i] = isu_e * __builtin_bit_cast(float, 0x7F000000u - __builtin_bit_cast(unsigned, S_item));
            }
            ep.omax2[i] = 0.f;
            ep.yv0[i] = ((unsigned)((ep.rimg[i] * a.H + c.y0) * a.W + ep.rpx[i]) * (unsigned)a.ldy + (unsigned)ep.cout_e) * 4u;
            ep.yseen[i] = a.ymax ? cnl::peek_max(a.ymax + (ep.rimg[i] < a.Nimg ? ep.rimg[i] : 0) * AMS) : 0u;
        }
    };
    // max |y| of a finished item into its images' slots
    auto report = [&](Epi& ep) __attribute__((always_inline)) {
        if (!a.ymax) return;
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int img = __builtin_amdgcn_readfirstlane(ep.rimg[i]);
            if constexpr (PK) {
                const float m1 = cnl::wave_max_nonneg(ep.rimg[i] != img ? ep.omax2[i] : 0.f);
                if (lane_e == 0 && img + 1 < a.Nimg) cnl::report_max(a.ymax + (img + 1) * AMS, m1);
                ep.omax2[i] = ep.rimg[i] == img ? ep.omax2[i] : 0.f;
            }
            const float m = cnl::wave_max_nonneg(ep.omax2[i]);
            if (lane_e == 0 && img < a.Nimg) cnl::raise_max(a.ymax + img * AMS, m, (unsigned)__builtin_amdgcn_readfirstlane((int)ep.yseen[i]));
        }
    };
    // bias / weight-scale requests of an item (one value per lane; written to LDS inside its first chunk)
    auto bias_of = [&](const Coord& c) __attribute__((always_inline)) {
        const int co = c.n0 + lane_now();
        st.bst = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(__builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, (int)a.b_bytes, 0x00020000),
                                                                                co < a.Cout ? (unsigned)co * 4u : OOB, 0, 0));
        st.ist = co < a.CoutP ? a.isu[co] : 0.f;
    };

    // ---- the workgroup's first item: prologue (patches 0 / 1 -> LDS, weight rows 0 / 1 of chunk 0, V rows 0 / 1, the raw reads of row 2) ----
    unsigned item = blockIdx.x;
    Coord cc_cur = coord_of(item), cc_nxt;
    item_of(st.cur, cc_cur);
    {
        const float xmax_cur = W12_XMAX_OF(img_lane_of(cc_cur));
        u32x4 keep[4][NSTG];
#define W12_PLOAD_HALF(dst_, half_, cc_)                                                                         \
        do {                                                                                                     \
            pload<half_, 0>(st, st.cur, a, cc_, up, true); pload<half_, 1>(st, st.cur, a, cc_, up, true); pload<half_, 2>(st, st.cur, a, cc_, up, true); \
            if constexpr (half_ == 0) pload<0, 3>(st, st.cur, a, cc_, up, true);                                 \
            _Pragma("unroll") for (int i = 0; i < NSTG; ++i) keep[dst_][i] = st.stg[i];                          \
        } while (0)
        W12_PLOAD_HALF(0, 0, 0);
        W12_PLOAD_HALF(1, 1, 0);
        W12_PLOAD_HALF(2, 0, 1);
        W12_PLOAD_HALF(3, 1, 1);
#undef W12_PLOAD_HALF
#pragma unroll
        for (int i = 0; i < 4; ++i) load_b<0>(st, a, st.cur.u_voff, 0, i, u_plane, u_wave, true);
#pragma unroll
        for (int i = 0; i < 4; ++i) load_b<1>(st, a, st.cur.u_voff, 0, i, u_plane, u_wave, true);
        {
            int es_cur;
            W12_SCALE_EXP(es_cur, xmax_cur);
            st.cur.S = __builtin_ldexpf(1.f, es_cur);
        }
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
    W12_BARRIER();
    rread<0>(st, 0, 0); rread<1>(st, 0, 0); rread<2>(st, 0, 0); rread<3>(st, 0, 0);
    job_all(st, 0, st.cur.S, std::make_integer_sequence<int, 28>{});
    rread<0>(st, 0, 1); rread<1>(st, 0, 1); rread<2>(st, 0, 1); rread<3>(st, 0, 1);
    job_all(st, 1, st.cur.S, std::make_integer_sequence<int, 28>{});
    rread<0>(st, 0, 2); rread<1>(st, 0, 2); rread<2>(st, 0, 2); rread<3>(st, 0, 2);

    Epi ep;                  // the item whose accumulators are being drained (set = the other one)
    bool cok_prev = false;
    // One work item in accumulator set SET: its four chunks (with the previous item's epilogue riding along unless PEEL), then the hand-over
#define W12_ITEM_BODY(SET_, PEEL_)                                                                               \
    {                                                                                                            \
        const unsigned next = item + gridDim.x;                                                                  \
        const bool more = next < (unsigned)a.blocks;                                                             \
        cc_nxt = coord_of(more ? next : item);                                                                   \
        item_of(st.nxt, cc_nxt);                                                                                 \
        const float xmax_next = W12_XMAX_OF(img_lane_of(cc_nxt));                                                \
        bias_of(cc_cur);                                                                                         \
        constexpr int E0 = PEEL_ ? -1 : 0, E1 = PEEL_ ? -1 : 1, E2 = PEEL_ ? -1 : 2, E3 = PEEL_ ? -1 : 3;        \
        chunk<0, 0, true, PEEL_, E0, SET_, RES>(st, ep, el, a, 0, up, u_plane, u_wave, more, wave, lo, cok_prev); \
        chunk<1, 0, false, PEEL_, E1, SET_, RES>(st, ep, el, a, 1, up, u_plane, u_wave, more, wave, lo, cok_prev); \
        {                                                                                                        \
            int es_nxt;                                                                                          \
            W12_SCALE_EXP(es_nxt, xmax_next);                                                                    \
            st.nxt.S = __builtin_ldexpf(1.f, es_nxt);                                                            \
        }                                                                                                        \
        chunk<0, 1, false, PEEL_, E2, SET_, RES>(st, ep, el, a, 2, up, u_plane, u_wave, more, wave, lo, cok_prev); \
        chunk<1, 2, false, PEEL_, E3, SET_, RES>(st, ep, el, a, 3, up, u_plane, u_wave, more, wave, lo, cok_prev); \
        if (!(PEEL_)) report(ep);                                                                                \
        epi_of(ep, cc_cur, SET_, st.cur.S);                                                                      \
        cok_prev = ep.cout_e < a.Cout;                                                                           \
        if (!more) { last_set = SET_; break; }                                                                   \
        item = next; cc_cur = cc_nxt; st.cur = st.nxt;                                                           \
    }
    int last_set = 0;
    do {
        W12_ITEM_BODY(0, true)
        while (true) {
            W12_ITEM_BODY(1, false)
            W12_ITEM_BODY(0, false)
        }
    } while (false);
#undef W12_ITEM_BODY

    // ---- the last item's epilogue, on its own: four passes through the exchange region (winograd10.hip's) ----
    // (every wave has passed the barrier of the last chunk: the previous item's exchange reads of half 0 are over; half 1 is next written behind the barrier below)
#define W12_LAST_EPILOGUE(ES_)                                                                                   \
    {                                                                                                            \
        epi_xwrite<ES_, 0, 0, 0>(st, el, wave); epi_xwrite<ES_, 0, 0, 1>(st, el, wave); epi_xwrite<ES_, 0, 0, 2>(st, el, wave); epi_xwrite<ES_, 0, 0, 3>(st, el, wave); \
        epi_xwrite<ES_, 0, 1, 0>(st, el, wave); epi_xwrite<ES_, 0, 1, 1>(st, el, wave); epi_xwrite<ES_, 0, 1, 2>(st, el, wave); epi_xwrite<ES_, 0, 1, 3>(st, el, wave); \
        W12_BARRIER();                                                                                           \
        W12_LAST_PASS(ES_, 0) W12_LAST_PASS(ES_, 1) W12_LAST_PASS(ES_, 2) W12_LAST_PASS(ES_, 3)                  \
    }
#define W12_LAST_PASS(ES_, J_)                                                                                   \
    {                                                                                                            \
        epi_begin<RES, J_>(st, ep, a, cok_prev);                                                                 \
        if constexpr (RES) { epi_res<J_, 0, 0>(st, ep, a); epi_res<J_, 0, 1>(st, ep, a); epi_res<J_, 1, 0>(st, ep, a); epi_res<J_, 1, 1>(st, ep, a); } \
        epi_yread<J_, 0, 0>(st, el); epi_yread<J_, 0, 1>(st, el); epi_yread<J_, 0, 2>(st, el); epi_yread<J_, 0, 3>(st, el); \
        epi_yread<J_, 1, 0>(st, el); epi_yread<J_, 1, 1>(st, el); epi_yread<J_, 1, 2>(st, el); epi_yread<J_, 1, 3>(st, el); \
        if constexpr (J_ + 1 < R) {                                                                              \
            epi_xwrite<ES_, (J_ + 1) % R, 0, 0>(st, el, wave); epi_xwrite<ES_, (J_ + 1) % R, 0, 1>(st, el, wave); epi_xwrite<ES_, (J_ + 1) % R, 0, 2>(st, el, wave); epi_xwrite<ES_, (J_ + 1) % R, 0, 3>(st, el, wave); \
            epi_xwrite<ES_, (J_ + 1) % R, 1, 0>(st, el, wave); epi_xwrite<ES_, (J_ + 1) % R, 1, 1>(st, el, wave); epi_xwrite<ES_, (J_ + 1) % R, 1, 2>(st, el, wave); epi_xwrite<ES_, (J_ + 1) % R, 1, 3>(st, el, wave); \
        }                                                                                                        \
        epi_step<RES, 0, 0>(st, ep, lo); epi_step<RES, 0, 1>(st, ep, lo); epi_step<RES, 0, 2>(st, ep, lo); epi_step<RES, 0, 3>(st, ep, lo); \
        epi_step<RES, 0, 4>(st, ep, lo); epi_step<RES, 0, 5>(st, ep, lo); epi_step<RES, 0, 6>(st, ep, lo); epi_step<RES, 0, 7>(st, ep, lo); \
        epi_store<0, 0>(st, a); epi_store<0, 1>(st, a);                                                          \
        epi_step<RES, 1, 0>(st, ep, lo); epi_step<RES, 1, 1>(st, ep, lo); epi_step<RES, 1, 2>(st, ep, lo); epi_step<RES, 1, 3>(st, ep, lo); \
        epi_step<RES, 1, 4>(st, ep, lo); epi_step<RES, 1, 5>(st, ep, lo); epi_step<RES, 1, 6>(st, ep, lo); epi_step<RES, 1, 7>(st, ep, lo); \
        epi_store<1, 0>(st, a); epi_store<1, 1>(st, a);                                                          \
        if constexpr (J_ + 1 < R) { W12_BARRIER(); }                                                             \
    }
    if (last_set == 0) W12_LAST_EPILOGUE(0) else W12_LAST_EPILOGUE(1)
    report(ep);
#undef W12_LAST_PASS
#undef W12_LAST_EPILOGUE
#undef W12_DIVMOD
#undef W12_VDIVMOD
#undef W12_XMAX_OF
#undef W12_SCALE_EXP
}

}  // namespace cnl_wino12

bool cnl_wino9_eligible(const cnl_conv_params* p);
size_t cnl_wino9_weight_bytes(int Cin, int Cout);
int cnl_wino_images_per_launch(const cnl_conv_params* p);
void cnl_wino_sub_batch(const cnl_conv_params* p, int n0, int n, cnl_conv_params* q, const float** xmax);
// this kernel runs exactly four chunks per work item
bool cnl_wino12_eligible(const cnl_conv_params* p) { return cnl_wino9_eligible(p) && p->Cin == 16 * cnl_wino12::CC && !p->fuse_w; }

static int wino12_launch_one(const cnl_conv_params* p, const void* u9, const float* isu, const float* xmax, void* stream) {
    using namespace cnl_wino12;
    Args a;
    a.x = p->x; a.u9 = u9; a.xmax = xmax; a.isu = isu; a.ymax = reinterpret_cast<unsigned*>(p->y_absmax);
    a.bias = p->bias; a.res = p->residual; a.y = p->y;
    const int upf = (p->flags & CNL_UPSAMPLE_IN) ? 2 : 1;
    a.Nimg = p->N; a.Hs = p->H_in; a.Ws = p->W_in; a.H = p->H_in * upf; a.W = p->W_in * upf; a.Cin = p->Cin; a.Cout = p->Cout;
    a.ipb = (upf == 1 && (a.W == 32 || a.W == 16)) ? 64 / a.W : 1;
    a.lw = a.W == 32 ? 5 : 4;
    a.N = (p->N + a.ipb - 1) / a.ipb;
    a.pk = cnl_wino_packed_stride(p);
    a.m_pk = a.pk ? (unsigned)(0x100000000ull / (unsigned)a.pk) : 0u;
    if (a.pk) a.N = 1;
    a.CoutP = (p->Cout + 63) / 64 * 64;
    a.ldx = p->ldx; a.ldy = p->ldy; a.ldr = p->ldr;
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
    static cnl::DeviceOnce once[4];
    const void* const fns[4] = {reinterpret_cast<const void*>(&winograd12_kernel<false, false>), reinterpret_cast<const void*>(&winograd12_kernel<true, false>),
                                reinterpret_cast<const void*>(&winograd12_kernel<false, true>), reinterpret_cast<const void*>(&winograd12_kernel<true, true>)};
    const int which = (p->residual ? 1 : 0) + (a.pk ? 2 : 0);
    int n_cu = 0;
    const int rc = cnl::kernel_setup(once[which], fns[which], LDS_BYTES, &n_cu);
    if (rc != CNL_OK) return rc;
    const unsigned grid = (unsigned)(blocks < (long long)n_cu ? blocks : (long long)n_cu);
    switch (which) {
    case 0: hipLaunchKernelGGL((winograd12_kernel<false, false>), dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, a); break;
    case 1: hipLaunchKernelGGL((winograd12_kernel<true, false>), dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, a); break;
    case 2: hipLaunchKernelGGL((winograd12_kernel<false, true>), dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, a); break;
    default: hipLaunchKernelGGL((winograd12_kernel<true, true>), dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, a); break;
    }
    return cnl::check_launch("winograd12_kernel");
}
int cnl_wino12_launch(const cnl_conv_params* p, const void* u9, const float* isu, const float* xmax, void* stream) {
    const int per = cnl_wino_images_per_launch(p);
    CNL_REQUIRE(per > 0, CNL_E_UNSUPPORTED, "cnl_conv3x3_winograd_f32: one image of a tensor spans >= 4 GiB");
    for (int n0 = 0; n0 < p->N; n0 += per) {
        cnl_conv_params q;
        const float* xm = xmax;
        cnl_wino_sub_batch(p, n0, p->N - n0 < per ? p->N - n0 : per, &q, &xm);
        const int rc = wino12_launch_one(&q, u9, isu, xm, stream);
        if (rc != CNL_OK) return rc;
    }
    return CNL_OK;
}
