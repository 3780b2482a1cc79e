// winograd.hip — 3x3 / stride 1 / pad 1 convolution by Winograd F(2x2,3x3) on the fp32 matrix cores (gfx950).
//
// Same call sites as conv_mfma.hip (ResNet BasicBlock 3x3 convs, make_conv, GenericHead blocks: reference
// models/meta.py:24-26, models/layers.py:72-77) for the layers that are 3x3 stride-1: 92 % of the conv time.
// conv_mfma.hip already runs at ~98 % MFMA utilisation at the clock the chip sustains under this load
// (2.14 GHz measured, DVFS), so the only lever left is fewer multiplies: F(2x2,3x3) needs 16 instead of 36
// per (2x2 output tile, cin, cout) = 2.25x fewer MFMA flops.  The result is the same function up to fp32
// rounding (|err| ~2.6e-6 at K=2304 vs 1.2e-6 for the direct sum; tolerance 1e-4).
//
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A        d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs
// which is 16 independent GEMMs, one per transform position xi = 4i+j:  M_xi[tile][co] = sum_ci V_xi[tile][ci] U_xi[co][ci].
//
// Workgroup = 8 waves = 8x8 tiles (16x16 output pixels of one image) x 64 output channels x all 16 positions; one
// workgroup per CU (86 KB of LDS: V and the input patch, both double-buffered), two waves per SIMD.
// Per chunk of 8 input channels, ONE barrier:
//   LDS-DMA   the 18x18x8 input patch of chunk cc+2 (zero halo from the buffer bounds check; optional nearest-2x source);
//   MFMA      wave w owns transform row i = w>>1 (positions 4i..4i+3) for cout group h = w&1 and both 32-tile groups:
//             per position two A fragments (ONE ds_read_b128 each: lanes 0-31 take ci 0-3, lanes 32-63 ci 4-7, four K=2 steps
//             per read) + one B fragment, 8 MFMAs; 32 MFMAs per wave per chunk.  The B (weight) fragments go global -> registers
//             directly (U = [ci/8][xi][co][8], pre-transformed once per weight load): every U element is used by exactly one
//             wave, so staging it in LDS (as this kernel first did) bought no reuse — dropping it removed 29 % of the LDS
//             traffic and half of the LDS footprint at unchanged speed: LDS bandwidth is not what limits this loop;
//   transform of chunk cc+1: thread = (tile, channel PAIR, output-row half) — packed fp32 adds (v_pk_add_f32), 8-byte LDS
//             accesses; the two waves of a SIMD take output rows {0,1} / {2,3} of B^T d B (no redundant work): 6 ds_read2_b64,
//             16 packed adds, 4 ds_write2st64_b64 per wave and chunk, placed in fenced slices among those MFMAs — the adds in two
//             dense bursts early in the chunk, because on gfx950 VALU work does not overlap the matrix pipe of its SIMD and an
//             MFMA -> VALU switch costs ~13 cycles (tools/mfma_coexec.hip, profiles/r01_mfma_coexec.txt).
// Persistent workgroups (one per CU) walk the work items; the next item's first chunk is fetched during the epilogue.
// Measured and rejected (round 1, details in DESIGN.md 3.1): a wave-specialised variant (8 matrix + 4 producer waves, 5 % slower);
// a software-pipelined chunk loop with the barrier in mid-chunk; s_setprio schedules that equalise the two waves of a SIMD (no
// change); two independent 4-wave workgroups per CU with wave-private V / U (correct, 2.5 % slower: LDS-DMA traffic doubles).
// Where the time goes (tools/wino_trace.py, PMC in profiles/): matrix pipes ~84 % busy; the older wave of each SIMD wins issue
// arbitration and idles ~40 % of a chunk at the barrier while the younger one finishes alone.
// Epilogue: owning a whole transform row lets each wave apply the first half of A^T M A in registers (4 -> 2 matrices);
// the halves meet through LDS ([i][c][tile][co], two passes) where thread = (tile, co) finishes Y, adds bias
// (+ residual) (+ ReLU) and stores the 2x2 outputs NHWC with buffer stores (uniform part of the address in the SGPR
// offset, 64 consecutive channels per 256 bytes).
// (round 2: kept under experiments/ — superseded by winograd2.hip, bit-identical to it; built only by `make experiments`)
#include "cnl_common.h"
#include <cstdlib>

namespace cnl_wino {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;

struct WinoArgs {
    const float* x;
    const float* u;
    const float* bias;
    const float* res;
    float* y;
    int N, H, W, Cin, Cout, CoutP;   // H, W: output (= logical input) size
    int Hs, Ws;                       // stored input size (H/2, W/2 with CNL_UPSAMPLE_IN, else H, W)
    int ldx, ldy, ldr;
    int CC;                           // Cin / 8
    int nb, bx, by;                   // blocks along cout, x, y
    int blocks;
    unsigned x_bytes, u_bytes, y_bytes, r_bytes;
    unsigned flags;
    long long* trace;                 // CNL_WTRACE builds only: per-wave barrier-wait / chunk-body cycle sums
};

constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr int T = 64;                       // tiles per workgroup (8 x 8 -> 16 x 16 output pixels)
constexpr int BN = 64;                      // output channels per workgroup
constexpr int PW = 18;                      // patch width / height in pixels
constexpr int V_BYTES = 16 * T * 32;        // 32768 per buffer
constexpr int PWP = 19;                     // padded patch row (pixels) of the LDS image [py][half][PWP][4 floats]: the 76-float
                                            // half-row stride makes the transform's 4x4 gathers bank-conflict-free
constexpr int P_SLOTS = 704;                // 684 used; 512 (all waves) + 192 (waves 0-2)
constexpr int P_BYTES = P_SLOTS * 16;       // 11264 per buffer
constexpr int LDS_BYTES = 2 * V_BYTES + 2 * P_BYTES;                 // 88064: V and the patch; U never touches LDS (see the kernel)

__device__ __forceinline__ void dma16(const float* base, unsigned bytes, char* lds_dst, unsigned voffset, unsigned soffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_dst, 16, voffset, soffset, 0, 0);
}
// same, 1024 bytes further on in BOTH the global source and the LDS destination (the instruction's immediate offset applies to
// both addresses): saves the VALU add of a second per-lane offset
__device__ __forceinline__ void dma16_plus1k(const float* base, unsigned bytes, char* lds_dst, unsigned voffset, unsigned soffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_dst, 16, voffset, soffset, 1024, 0);
}
__device__ __forceinline__ f32x4 buf_load16(const float* base, unsigned bytes, unsigned voffset, unsigned soffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, soffset, 0));
}
__device__ __forceinline__ float buf_load(const float* base, unsigned bytes, unsigned voffset, unsigned soffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voffset, soffset, 0));
}
__device__ __forceinline__ void buf_store(float v, float* base, unsigned bytes, unsigned voffset, unsigned soffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, voffset, soffset, 0);
}
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_zero() {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    return __builtin_amdgcn_mfma_f32_32x32x2f32(0.f, 0.f, z, 0, 0, 0);
}
__device__ __forceinline__ float lds_f(const char* p) { return *reinterpret_cast<const float*>(p); }
__device__ __forceinline__ f32x4 lds_f4(const char* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x2 lds_f2(const char* p) { return *reinterpret_cast<const f32x2*>(p); }
// packed fp32 add / subtract on a channel pair (the compiler scalarises <2 x float> arithmetic in this kernel)
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// b * s + a with s = +-1 (exact: one rounding, the same result as the add / subtract it stands for)
__device__ __forceinline__ f32x2 pk_fma(f32x2 b, f32x2 s, f32x2 a) {
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(s), "v"(a));
    return r;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__global__ __launch_bounds__(512, 2) void winograd_conv_kernel(const WinoArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sV = smem;                                  // [2][16 xi][64 tiles][8 ci]
    char* sP = smem + 2 * V_BYTES;                    // [2][18 py][2 halves][19 px][4 ci] (+ slack)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5;
    const int wi = wave >> 1, wh = wave & 1;          // transform row and cout group owned by this wave
    const bool up = a.flags & CNL_UPSAMPLE_IN;
    const unsigned u_chunk = (unsigned)(16 * a.CoutP * 8 * 4);          // bytes per channel chunk of U
    const unsigned u_pos = (unsigned)(a.CoutP * 8 * 4);                 // bytes per transform position inside a chunk

    // transform item: thread -> (tile, channel PAIR, half): every add of B^T d B is one v_pk_add_f32 on two channels and every
    // LDS access 8 bytes wide.  The two halves of the workgroup (waves 0-3 / 4-7, one wave of each SIMD) produce output ROWS
    // {0,1} / {2,3} of the 4x4 transform from patch rows {0,1,2} / {1,2,3} — rows of t = B^T d are independent, so the split has no
    // redundant work: 6 ds_read2_b64 + 16 v_pk_add_f32 + 4 ds_write2st64_b64 per wave and chunk (8 + 32 + 8 for one thread per
    // (tile, channel)).
    const int lt = tid & 255;
    const int th = tid >> 8;                                           // transform half of this wave (waves 0-3: 0, 4-7: 1)
    const int t_cp = lt & 3, t_tile = lt >> 2;
    const int t_src = ((((2 * (t_tile >> 3)) * 2 + (t_cp >> 1)) * PWP + 2 * (t_tile & 7)) * 4 + (t_cp & 1) * 2) * 4;
    // rows (r0, r1, r2) of the patch read by this half, chosen so that ONE instruction sequence serves both:
    //   t0 = r0 - r2,  t1 = sg * r1 + r2   with  half 0: (d0, d1, d2), sg = +1   ->  d0 - d2, d1 + d2   (output rows 0, 1)
    //                                            half 1: (d2, d3, d1), sg = -1   ->  d2 - d1, d1 - d3   (output rows 2, 3)
    const int t_r0 = t_src + (th ? 2 : 0) * 2 * PWP * 16, t_r1 = t_src + (th ? 3 : 1) * 2 * PWP * 16, t_r2 = t_src + (th ? 1 : 2) * 2 * PWP * 16;
    const float sgf = th ? -1.f : 1.f;
    const f32x2 sg = {sgf, sgf};
    // V / U rows are 32 bytes = two 16-byte halves (ci 0-3 | ci 4-7); rows with bit 3 set store them swapped, which makes the
    // ds_read_b128 fragment reads (16-lane groups, 32-byte row pitch) bank-conflict-free
    const int t_dst = (t_tile * 8 + (((t_cp >> 1) ^ ((t_tile >> 3) & 1)) << 2) + (t_cp & 1) * 2) * 4 + th * 8 * (T * 32);
    // chunk-0 transform (prologue, all threads): thread -> (tile = tid >> 3, ch = tid & 7)
    const int p_ch = tid & 7, p_tile = tid >> 3;
    const int p_src = ((((2 * (p_tile >> 3)) * 2 + (p_ch >> 2)) * PWP + 2 * (p_tile & 7)) * 4 + (p_ch & 3)) * 4;
    const int p_dst = (p_tile * 8 + (p_ch ^ (((p_tile >> 3) & 1) << 2))) * 4;
    const int hs = hi ^ ((lane >> 3) & 1);                             // physical half holding this lane's logical half
    const int fragA = ((lane & 31) * 8 + hs * 4) * 4;                  // + (xi*64 + g*32) * 32
    const int xi0 = wi * 4;
    const float lo = (a.flags & CNL_RELU) ? 0.f : -__builtin_inff();

    // ---- per-work-item bookkeeping: item -> (image n, tile-block row/col, cout block); cout fastest so that the workgroups
    // sharing an input patch run side by side; per-lane DMA source offsets of the patch and of the U slice ----
    int n, y0, x0, n0;
    unsigned p_off[2], u_off;
#define WINO_SETUP(item_)                                                                                        \
    do {                                                                                                         \
        unsigned b_ = cnl::xcd_remap((item_), (unsigned)a.blocks);                                               \
        const int nbi_ = b_ % a.nb; b_ /= a.nb;                                                                  \
        const int bxi_ = b_ % a.bx; b_ /= a.bx;                                                                  \
        const int byi_ = b_ % a.by;                                                                              \
        n = b_ / a.by; y0 = byi_ * 16; x0 = bxi_ * 16; n0 = nbi_ * BN;                                           \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                          \
            const int s_ = i * 512 + tid;              /* 16-byte slot of the patch image: (py*2 + half)*PWP + px */ \
            const int rowh_ = s_ / PWP, pxx_ = s_ - rowh_ * PWP;                                                 \
            const int py_ = rowh_ >> 1, half_ = rowh_ & 1;                                                       \
            const int iy_ = y0 - 1 + py_, ix_ = x0 - 1 + pxx_;                                                   \
            const bool ok_ = py_ < PW && pxx_ < PW && (unsigned)iy_ < (unsigned)a.H && (unsigned)ix_ < (unsigned)a.W; \
            const int sy_ = up ? (iy_ >> 1) : iy_, sx_ = up ? (ix_ >> 1) : ix_;   /* nearest-2x upsample folded in */ \
            p_off[i] = ok_ ? (unsigned)((((n * a.Hs + sy_) * a.Ws + sx_) * a.ldx + half_ * 4) * 4) : OOB;        \
        }                                                                                                        \
        /* this lane's B fragment of position xi: cout row n0 + wh*32 + (lane & 31), channel half hi */          \
        u_off = (unsigned)((((xi0 * a.CoutP + n0 + wh * 32 + (lane & 31)) * 8) + hi * 4) * 4);                   \
    } while (0)
#define WINO_ISSUE_P(cc_)                                                                                        \
    do {                                                                                                         \
        char* d_ = sP + ((cc_) & 1) * P_BYTES;                                                                   \
        /* the channel-chunk offset rides in the SCALAR offset (no VALU): the bounds check looks at the vector offset alone, so   \
           halo lanes (p_off == OOB) still read zeros; a chunk past the end is simply not fetched (nothing consumes it) */      \
        if ((cc_) < a.CC) {                                                                                      \
            dma16(a.x, a.x_bytes, d_ + (wave * 64) * 16, p_off[0], (unsigned)((cc_) * 32));                      \
            if (wave < 3) dma16(a.x, a.x_bytes, d_ + (512 + wave * 64) * 16, p_off[1], (unsigned)((cc_) * 32));  \
        }                                                                                                        \
    } while (0)
    // B (weight) fragments never touch LDS: every U element is used by exactly one wave (positions x cout half partition U), so
    // each wave loads its own fragments global -> registers, 16 bytes per lane and position, one chunk ahead: the slot of position
    // j is refilled for chunk cc+1 right after position j's last MFMA of chunk cc.  No LDS write + read, no barrier dependency;
    // the compiler orders uses after the loads with counted vmcnt waits of its own (VMEM returns in issue order) — which is why the
    // chunk loop below has ONE body for both transform halves: with two loop bodies the register allocator reused fragment
    // registers of one for other data in the other, and the inserted waits drained vmcnt to 0 in every chunk.
#define WINO_LOAD_U(cc_, j_)                                                                                     \
    do {                                                                                                         \
        if ((cc_) < a.CC) fbU[j_] = buf_load16(a.u, a.u_bytes, u_off, (unsigned)(cc_) * u_chunk + (unsigned)(j_) * u_pos); \
    } while (0)
    // the 8 MFMAs of one position: k = 0..7 -> c = k >> 1, g = k & 1
#define WINO_MFMA8(j_, fa_, k_) acc[j_][(k_) & 1] = mfma32((fa_)[(k_) & 1][(k_) >> 1], fbU[j_][(k_) >> 1], acc[j_][(k_) & 1])
    // workgroup barrier WITHOUT the vmcnt(0) that __syncthreads() adds while VMEM -> LDS transfers are in flight
#define WINO_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    // Persistent workgroups (grid = one per CU): the first chunk of the NEXT work item is fetched while the epilogue of the
    // current one runs, so only the very first item of a launch waits for HBM latency with an idle matrix pipe.
#ifdef CNL_WTRACE
    long long tr_wait = 0, tr_body = 0, tr_n = 0, tr_a = 0, tr_b = 0, tr_vm = 0;
    const long long tr_start = clock64();
#define WTRACE_PRE()  do { tr_a = clock64(); if (tr_b) { tr_body += tr_a - tr_b; ++tr_n; } } while (0)
#define WTRACE_POST() do { tr_b = clock64(); tr_wait += tr_b - tr_a; } while (0)
#define WTRACE_MID()  do { tr_vm += clock64() - tr_a; } while (0)
#else
#define WTRACE_MID()  do { } while (0)
#define WTRACE_PRE()  do { } while (0)
#define WTRACE_POST() do { } while (0)
#endif
    unsigned item = blockIdx.x;
    f32x4 fbU[4];            // B fragments of this wave's four positions (current chunk; refilled in a rolling fashion)
    WINO_SETUP(item);
    WINO_ISSUE_P(0);
    WINO_ISSUE_P(1);
#pragma unroll
    for (int j = 0; j < 4; ++j) WINO_LOAD_U(0, j);
    bool first = true;
    while (true) {
        f32x16 acc[4][2];        // [position j of row wi][tile group]
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 2; ++g) acc[j][g] = mfma_zero();

        // patches 0 / 1 and the B fragments of chunk 0 landed?  They are followed in this wave's VMEM queue by the second half of the
        // previous item's epilogue (16 stores, +16 residual loads): a counted wait lets those stay in flight.
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (a.res) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        first = false;
        WINO_BARRIER();
        {   // input transform of chunk 0 (not overlapped with MFMAs)
            const char* src_ = sP + p_src;
            float d_[4][4], t_[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) d_[i][j] = lds_f(src_ + (i * 2 * PWP + j) * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t_[0][j] = d_[0][j] - d_[2][j];
                t_[1][j] = d_[1][j] + d_[2][j];
                t_[2][j] = d_[2][j] - d_[1][j];
                t_[3][j] = d_[1][j] - d_[3][j];
            }
            char* dst_ = sV + p_dst;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<float*>(dst_ + (i * 4 + 0) * (T * 32)) = t_[i][0] - t_[i][2];
                *reinterpret_cast<float*>(dst_ + (i * 4 + 1) * (T * 32)) = t_[i][1] + t_[i][2];
                *reinterpret_cast<float*>(dst_ + (i * 4 + 2) * (T * 32)) = t_[i][2] - t_[i][1];
                *reinterpret_cast<float*>(dst_ + (i * 4 + 3) * (T * 32)) = t_[i][1] - t_[i][3];
            }
        }

        // steady state: ONE barrier per chunk; MFMAs of chunk cc with the input transform of chunk cc+1 hand-interleaved, one
        // slice = {1 MFMA, 2 VALU | 1 LDS write}, slices fenced by sched_barrier(0): left to itself hipcc emits the whole
        // transform after the last MFMA, where both waves of a SIMD reach it together and the matrix pipe idles.
        // one chunk: 32 MFMAs in 32 fenced slices; DO_T_ adds this wave's share of the next chunk's input transform
// slice schedule of the input transform inside a chunk (slice = one MFMA): WS_RD patch reads; WS_DMA, +1 DMA issue; t adds WS_TN per
// slice from WS_T; v adds WS_VN per slice from WS_V; WS_WN write pairs per slice from WS_W.  A VALU instruction right after an MFMA
// costs the issuing wave ~14 cycles and each further one ~5 (tools/mfma_coexec.hip), so the adds go in few, dense bursts, and early
// in the chunk, where the other wave of the SIMD is certain to have MFMAs to fill the pipe with.
#define WS_RD 0      /* 6 ds_read2_b64 of the patch            */
#define WS_DMA 2     /* patch DMA issue                          */
#define WS_T 4       /* 8 packed adds: t = B^T d                */
#define WS_V 5       /* 8 packed adds: V = t B                  */
#define WS_W 6       /* 2 + 2 ds_write2st64_b64 (slices 6, 7)   */
#define WINO_CHUNK()                                                                                                         \
        do {                                                                                                                 \
            const char* vB = sV + (cc & 1) * V_BYTES + fragA;                                                                \
            const char* pB = sP + ((cc + 1) & 1) * P_BYTES;                                                                  \
            char* dst_ = sV + ((cc + 1) & 1) * V_BYTES + t_dst;                                                              \
            f32x4 fa[2][2];             /* double-buffered A fragments: [buffer][tile group] */                              \
            fa[0][0] = lds_f4(vB + (xi0 * 64) * 32); fa[0][1] = lds_f4(vB + (xi0 * 64 + 32) * 32);                           \
            f32x2 d_[3][4], t_[2][4], v_[2][4];                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                               \
            _Pragma("unroll") for (int k = 0; k < 32; ++k) {                                                                 \
                const int j = k >> 3, kk = k & 7, buf = j & 1;                                                               \
                WINO_MFMA8(j, fa[buf], kk);                                                                                  \
                if (kk == 7) WINO_LOAD_U(cc + 1, j);                    /* position j is done: refill its slot for the next chunk */ \
                if (kk == 2 && j < 3) {                                 /* next position's A fragments, 6 MFMAs ahead of use */ \
                    fa[buf ^ 1][0] = lds_f4(vB + ((xi0 + j + 1) * 64) * 32);                                                 \
                    fa[buf ^ 1][1] = lds_f4(vB + ((xi0 + j + 1) * 64 + 32) * 32);                                            \
                }                                                                                                            \
                if (k == WS_RD) {                                       /* three patch rows of this half-item */             \
                    _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                                       \
                        d_[0][jj] = lds_f2(pB + t_r0 + jj * 16);                                                             \
                        d_[1][jj] = lds_f2(pB + t_r1 + jj * 16);                                                             \
                        d_[2][jj] = lds_f2(pB + t_r2 + jj * 16);                                                             \
                    }                                                                                                        \
                }                                                                                                            \
                if (k == WS_DMA) WINO_ISSUE_P(cc + 2);                                                                       \
                if (k == WS_T) {                                        /* t = B^T d: two output rows, four columns */       \
                    _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                                       \
                        t_[0][jj] = pk_sub(d_[0][jj], d_[2][jj]);                                                            \
                        t_[1][jj] = pk_fma(d_[1][jj], sg, d_[2][jj]);                                                        \
                    }                                                                                                        \
                }                                                                                                            \
                if (k == WS_V) {                                        /* V = t B ... */                                    \
                    _Pragma("unroll") for (int ii = 0; ii < 2; ++ii) {                                                       \
                        v_[ii][0] = pk_sub(t_[ii][0], t_[ii][2]);                                                            \
                        v_[ii][1] = pk_add(t_[ii][1], t_[ii][2]);                                                            \
                        v_[ii][2] = pk_sub(t_[ii][2], t_[ii][1]);                                                            \
                        v_[ii][3] = pk_sub(t_[ii][1], t_[ii][3]);                                                            \
                    }                                                                                                        \
                }                                                                                                            \
                if (k >= WS_W && k < WS_W + 2) {                        /* ... written as ds_write2st64_b64 pairs */         \
                    const int ii = k - WS_W;                                                                                 \
                    _Pragma("unroll") for (int jj = 0; jj < 4; ++jj)                                                         \
                        *reinterpret_cast<f32x2*>(dst_ + (4 * ii + jj) * (T * 32)) = v_[ii][jj];                             \
                }                                                                                                            \
                __builtin_amdgcn_sched_barrier(0);                                                                           \
            }                                                                                                                \
        } while (0)

        // chunk top: this wave's share of patch cc+1 landed (4 newer loads: the B fragments) + barrier: V[cc&1] complete, patch cc+1
        // complete, MFMA phase cc-1 and transform cc done everywhere.  Chunk 0 has nothing to wait for (item-start wait above).
#define WINO_TOP()                                                                  \
        do {                                                                        \
            WTRACE_PRE();                                                           \
            if (cc > 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");            \
            WTRACE_MID();                                                           \
            WINO_BARRIER();                                                         \
            WTRACE_POST();                                                          \
        } while (0)
        int cc = 0;
        for (; cc + 1 < a.CC; ++cc) {
            WINO_TOP();
            WINO_CHUNK();
        }
#undef WINO_CHUNK
        {   // last chunk: MFMAs only; its B fragments were requested during the previous chunk
            WINO_BARRIER();
            const char* vB = sV + (cc & 1) * V_BYTES + fragA;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 fa[1][2];
                fa[0][0] = lds_f4(vB + ((xi0 + j) * 64) * 32); fa[0][1] = lds_f4(vB + ((xi0 + j) * 64 + 32) * 32);
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) WINO_MFMA8(j, fa[0], kk);
            }
        }

        // ---- epilogue: Y = A^T M A.  Stage 1 (row of positions, in registers): q_c = sum_j A^T[c][j] M[i][j] ----
        float* sQ = reinterpret_cast<float*>(smem);            // [4 i][2 c][32 tiles][64 co] = 64 KB per tile group (V buffers)
        const int co = tid & 63;
        const int col = n0 + co;
        const bool col_ok = col < a.Cout;
        const float bv = col_ok ? a.bias[col] : 0.f;
        const bool full = (y0 + 16 <= a.H) && (x0 + 16 <= a.W) && (n0 + BN <= a.Cout);
        const int en = n, ey0 = y0, ex0 = x0;                  // this item's coordinates (the setup below moves on to the next)
        const unsigned next = item + gridDim.x;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            WINO_BARRIER();                                    // done reading V (g = 0) or sQ of the previous pass
            if (g == 1 && next < (unsigned)a.blocks) {         // fetch the next item's first two patches and first B fragments
                WINO_SETUP(next);
                WINO_ISSUE_P(0);
                WINO_ISSUE_P(1);
#pragma unroll
                for (int j = 0; j < 4; ++j) WINO_LOAD_U(0, j);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tl = (r & 3) + 8 * (r >> 2) + 4 * hi;
                const float m0 = acc[0][g][r], m1 = acc[1][g][r], m2 = acc[2][g][r], m3 = acc[3][g][r];
                sQ[((wi * 2 + 0) * 32 + tl) * 64 + wh * 32 + (lane & 31)] = m0 + m1 + m2;
                sQ[((wi * 2 + 1) * 32 + tl) * 64 + wh * 32 + (lane & 31)] = m1 - m2 - m3;
            }
            WINO_BARRIER();
            // Stage 2: thread = (tile, co): Y[a][c] = sum_i A^T[a][i] q[i][c]; 4 items per thread
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int tl = (tid >> 6) + 8 * it;            // tile inside this 32-tile group
                const int tile = g * 32 + tl;
                const int oy = ey0 + 2 * (tile >> 3), ox = ex0 + 2 * (tile & 7);
                const unsigned pix = (unsigned)((en * a.H + oy) * a.W + ox);
                const unsigned y_voff = (pix * (unsigned)a.ldy + (unsigned)col) * 4u;
                const unsigned r_voff = (pix * (unsigned)a.ldr + (unsigned)col) * 4u;
                bool ok[2][2];
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) ok[dy][dx] = full || (col_ok && oy + dy < a.H && ox + dx < a.W);
                float rv[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
                if (a.res) {
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx)
                            rv[dy][dx] = buf_load(a.res, a.r_bytes, ok[dy][dx] ? r_voff : OOB, (unsigned)((dy * a.W + dx) * a.ldr * 4));
                }
                float q[4][2];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int c = 0; c < 2; ++c) q[i][c] = sQ[((i * 2 + c) * 32 + tl) * 64 + co];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float ya = q[0][c] + q[1][c] + q[2][c];
                    const float yb = q[1][c] - q[2][c] - q[3][c];
                    buf_store(fmaxf(ya + bv + rv[0][c], lo), a.y, a.y_bytes, ok[0][c] ? y_voff : OOB, (unsigned)(c * a.ldy * 4));
                    buf_store(fmaxf(yb + bv + rv[1][c], lo), a.y, a.y_bytes, ok[1][c] ? y_voff : OOB, (unsigned)((a.W + c) * a.ldy * 4));
                }
            }
        }
#ifdef CNL_WTRACE
        tr_b = 0;                      // the body of the last loop chunk runs into the last chunk + epilogue: not counted
#endif
        if (next >= (unsigned)a.blocks) break;
        item = next;
    }
#ifdef CNL_WTRACE
    if (a.trace && lane == 0) {
        long long* t = a.trace + ((long)blockIdx.x * 8 + wave) * 4;
        t[0] = tr_wait; t[1] = tr_body; t[2] = tr_n; t[3] = clock64() - tr_start;
        a.trace[256 * 8 * 4 + (long)blockIdx.x * 8 + wave] = tr_vm;
    }
#endif
#undef WINO_MFMA8
#undef WINO_ISSUE_P
#undef WINO_LOAD_U
#undef WINO_SETUP
}

}  // namespace cnl_wino
using namespace cnl_wino;

static size_t wino_f32_floats(int Cin, int Cout) {
    const size_t CoutP = (size_t)((Cout + 63) / 64) * 64;
    return (size_t)(Cin / 8) * 16 * CoutP * 8;
}

// Launch (arguments already validated by cnl_conv3x3_winograd_f32).
int cnl_wino1_launch(const cnl_conv_params* p, void* stream) {
    WinoArgs a;
    a.x = p->x; a.u = p->w; a.bias = p->bias; a.res = p->residual; a.y = p->y;
    const int upf = (p->flags & CNL_UPSAMPLE_IN) ? 2 : 1;
    a.N = p->N; a.Hs = p->H_in; a.Ws = p->W_in; a.H = p->H_in * upf; a.W = p->W_in * upf; a.Cin = p->Cin; a.Cout = p->Cout;
    a.CoutP = (p->Cout + 63) / 64 * 64;
    a.ldx = p->ldx; a.ldy = p->ldy; a.ldr = p->ldr;
    a.CC = p->Cin / 8;
    a.nb = a.CoutP / BN; a.bx = (a.W + 15) / 16; a.by = (a.H + 15) / 16;
    const long long blocks = (long long)p->N * a.by * a.bx * a.nb;
    CNL_REQUIRE(blocks < (1ll << 31), CNL_E_UNSUPPORTED, "cnl_conv3x3_winograd_f32: grid too large");
    a.trace = nullptr;
#ifdef CNL_WTRACE
    if (const char* e = getenv("CNL_TRACE_PTR")) a.trace = (long long*)strtoull(e, nullptr, 0);
#endif
    a.blocks = (int)blocks;
    const unsigned long long xb = (((unsigned long long)p->N * p->H_in * p->W_in - 1) * p->ldx + p->Cin) * 4ull;
    const unsigned long long ub = (unsigned long long)wino_f32_floats(p->Cin, p->Cout) * 4ull;
    const unsigned long long Mo = (unsigned long long)p->N * a.H * a.W;
    const unsigned long long yb = ((Mo - 1) * p->ldy + p->Cout) * 4ull;
    const unsigned long long rb = p->residual ? ((Mo - 1) * p->ldr + p->Cout) * 4ull : 0ull;
    const unsigned long long slack = (unsigned long long)(a.W + 2) * 4ull;     // scalar-offset reach of the epilogue stores
    CNL_REQUIRE(xb < 0xFFFFFF00ull && ub < 0xFFFFFF00ull && yb + slack * p->ldy < 0xFFFFFF00ull && rb + slack * (p->residual ? p->ldr : 0) < 0xFFFFFF00ull,
                CNL_E_UNSUPPORTED, "cnl_conv3x3_winograd_f32: tensor spans >= 4 GiB; split the batch");
    a.x_bytes = (unsigned)xb; a.u_bytes = (unsigned)ub; a.y_bytes = (unsigned)yb; a.r_bytes = (unsigned)rb;
    a.flags = p->flags;
    static cnl::DeviceOnce once;
    int n_cu = 0;
    const int rc = cnl::kernel_setup(once, reinterpret_cast<const void*>(&winograd_conv_kernel), LDS_BYTES, &n_cu);
    if (rc != CNL_OK) return rc;
    const unsigned grid = (unsigned)(blocks < n_cu ? blocks : n_cu);
    hipLaunchKernelGGL(winograd_conv_kernel, dim3(grid), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
    return cnl::check_launch("winograd_conv_kernel");
}
