// winograd2.hip — second work decomposition of the Winograd F(2x2,3x3) convolution: the same arithmetic as winograd.hip in the same
// order (results are bit-identical), on 8x16-pixel blocks instead of 16x16.  cnl_conv3x3_winograd_f32 dispatches here for maps
// that 16-row blocks would pad more (e.g. the 152x272 / 38x68 / 19x34 maps of 608x1088 MOT frames: 5 % / 17 % / 25 % fewer
// padded pixels) — per padded pixel the two run equally fast.
//
// Workgroup = 4 waves = 4x8 tiles (8x16 output pixels of one image) x 64 output channels x all 16 transform positions; TWO
// independent workgroups per CU (45 KB of LDS each), so whatever one waits for — its barrier, its epilogue — is covered by the
// other one's MFMAs (`SQ_WAIT_ANY` 21.6 % -> 10.8 % against winograd.hip; the kernel time is the same: both are clock-limited).
//   wave i owns transform row i (positions 4i..4i+3) for all 32 tiles and both 32-cout groups: per position ONE A fragment (one
//   ds_read_b128 feeding four K=2 steps) and TWO B fragments, 8 MFMAs; 32 MFMAs per wave per 8-channel chunk, 128 accumulator
//   registers — the same matrix work per wave as in winograd.hip.
//   V (transformed input) rows of position row i are produced AND consumed by wave i: rows of t = B^T d are independent, so wave
//   i needs two patch rows per tile and computes t[i][*], V[i][*] = t[i] B with 8 packed ops per (tile, channel pair) — V is
//   wave-private and needs no barrier.
//   B (weight) fragments go global -> registers (every U element is used by exactly one wave), refilled for chunk cc+1 right
//   after position j's last MFMA of chunk cc; the compiler orders their uses with counted vmcnt waits.
//   Only the input patch (10x18 px x 8 ch, double-buffered, fetched two chunks ahead by LDS-DMA) is shared: ONE barrier per chunk.
#include "cnl_common.h"
#include <cstdlib>

namespace cnl_wino2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;

struct Args {
    const float* x;
    const float* u;
    const float* bias;
    const float* res;
    float* y;
    int N, H, W, Cin, Cout, CoutP;   // H, W: output (= logical input) size
    int Hs, Ws;                       // stored input size (H/2, W/2 with CNL_UPSAMPLE_IN, else H, W)
    int ldx, ldy, ldr;
    int CC;                           // Cin / 8
    int nb, bx, by;                   // blocks along cout, x (16 px), y (8 px)
    int blocks;
    unsigned x_bytes, u_bytes, y_bytes, r_bytes;
    unsigned flags;
};

constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr int T = 32;                       // tiles per workgroup: 4 tile rows x 8 tile columns
constexpr int BN = 64;
constexpr int PH = 10, PW = 18;             // patch height / width in pixels
constexpr int PWP = 19;                     // padded patch row of the LDS image [py][half][PWP][4 floats]
constexpr int V_BYTES = 16 * T * 32;        // 16384 per buffer (two buffers)
constexpr int P_SLOTS = 384;                // 380 used; 256 (all waves) + 128 (waves 0-1)
constexpr int P_BYTES = P_SLOTS * 16;       // 6144 per buffer (two buffers)
constexpr int LDS_BYTES = 2 * V_BYTES + 2 * P_BYTES;             // 45056 -> two workgroups per CU (the VGPR budget allows no more)

__device__ __forceinline__ void dma16(const float* base, unsigned bytes, char* lds_dst, unsigned voffset, unsigned soffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_dst, 16, voffset, soffset, 0, 0);
}
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
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, voffset, soffset, CNL_NT_STORES);
}
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma_zero() {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    return __builtin_amdgcn_mfma_f32_32x32x2f32(0.f, 0.f, z, 0, 0, 0);
}
__device__ __forceinline__ f32x4 lds_f4(const char* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x2 lds_f2(const char* p) { return *reinterpret_cast<const f32x2*>(p); }
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a + s * b with s = +-1 (exact: one rounding, the same result as the add / subtract it stands for)
__device__ __forceinline__ f32x2 pk_fma(f32x2 b, f32x2 s, f32x2 a) {
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(s), "v"(a));
    return r;
}

__global__ __launch_bounds__(256, 2) void winograd2_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sV = smem;                                  // [2][16 xi][32 tiles][8 ci]
    char* sP = smem + 2 * V_BYTES;                    // [2][10 py][2 halves][19 px][4 ci] (+ slack)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = transform row i owned by this wave
    const int hi = lane >> 5;
    const int xi0 = wave * 4;
    const bool up = a.flags & CNL_UPSAMPLE_IN;
    const unsigned u_chunk = (unsigned)(16 * a.CoutP * 8 * 4);      // bytes per channel chunk of U
    const unsigned u_pos = (unsigned)(a.CoutP * 8 * 4);             // bytes per position inside a chunk

    // ---- input transform of this wave's row: lane -> (tile column tx, channel pair cp) for tile rows ty = hi, 2 + hi ----
    // t[i][*] = d[ra][*] + sg * d[rb][*]:  i = 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 3 ? 3 : (wave == 2 ? 1 : 2);
    const float sgf = wave == 1 ? 1.f : -1.f;
    const f32x2 sg = {sgf, sgf};
    const int t_tx = (lane >> 2) & 7, t_cp = lane & 3;
    int src_a[2], src_b[2], dst_v[2];                 // byte offsets inside a patch / V buffer, per item q (tile row 2q + hi)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int ty = 2 * q + hi;
        const int col = ((t_cp >> 1) * PWP + 2 * t_tx) * 16 + (t_cp & 1) * 8;
        src_a[q] = ((2 * ty + ra) * 2 * PWP) * 16 + col;
        src_b[q] = ((2 * ty + rb) * 2 * PWP) * 16 + col;
        const int tile = ty * 8 + t_tx;
        // V / U rows are 32 bytes = two 16-byte halves (ci 0-3 | ci 4-7); rows with bit 3 set store them swapped, which makes the
        // ds_read_b128 fragment reads (16-lane groups, 32-byte row pitch) bank-conflict-free
        dst_v[q] = ((xi0 * T + tile) * 8 + (((t_cp >> 1) ^ ((tile >> 3) & 1)) << 2) + (t_cp & 1) * 2) * 4;
    }
    const int hs = hi ^ ((lane >> 3) & 1);                              // physical half holding this lane's logical half
    const int fragA = ((xi0 * T + (lane & 31)) * 8 + hs * 4) * 4;       // + j * T * 32
    const float lo = (a.flags & CNL_RELU) ? 0.f : -__builtin_inff();

    int n, y0, x0, n0;
    unsigned p_off[2], u_voff;
#define W2_SETUP(item_)                                                                                          \
    do {                                                                                                         \
        unsigned b_ = cnl::xcd_remap((item_), (unsigned)a.blocks);                                               \
        const int nbi_ = b_ % a.nb; b_ /= a.nb;                                                                  \
        const int bxi_ = b_ % a.bx; b_ /= a.bx;                                                                  \
        const int byi_ = b_ % a.by;                                                                              \
        n = b_ / a.by; y0 = byi_ * 8; x0 = bxi_ * 16; n0 = nbi_ * BN;                                            \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                          \
            const int s_ = i * 256 + tid;              /* 16-byte slot of the patch image: (py*2 + half)*PWP + px */ \
            const int rowh_ = s_ / PWP, pxx_ = s_ - rowh_ * PWP;                                                 \
            const int py_ = rowh_ >> 1, half_ = rowh_ & 1;                                                       \
            const int iy_ = y0 - 1 + py_, ix_ = x0 - 1 + pxx_;                                                   \
            const bool ok_ = py_ < PH && pxx_ < PW && (unsigned)iy_ < (unsigned)a.H && (unsigned)ix_ < (unsigned)a.W; \
            const int sy_ = up ? (iy_ >> 1) : iy_, sx_ = up ? (ix_ >> 1) : ix_;   /* nearest-2x upsample folded in */ \
            p_off[i] = ok_ ? (unsigned)((((n * a.Hs + sy_) * a.Ws + sx_) * a.ldx + half_ * 4) * 4) : OOB;        \
        }                                                                                                        \
        /* this lane's B fragment: cout row n0 + (lane & 31) (+ 32 for the second group), channel half hi */     \
        u_voff = (unsigned)((((xi0 * a.CoutP + n0 + (lane & 31)) * 8) + hi * 4) * 4);                            \
    } while (0)
    // the channel-chunk offset rides in the SCALAR offset (no VALU; the bounds check looks at the vector offset alone, so halo
    // lanes still read zeros); a chunk past the end is not fetched
#define W2_ISSUE_P(cc_)                                                                                          \
    do {                                                                                                         \
        if ((cc_) < a.CC) {                                                                                      \
            char* d_ = sP + ((cc_) & 1) * P_BYTES;                                                               \
            dma16(a.x, a.x_bytes, d_ + (wave * 64) * 16, p_off[0], (unsigned)((cc_) * 32));                      \
            if (wave < 2) dma16(a.x, a.x_bytes, d_ + (256 + wave * 64) * 16, p_off[1], (unsigned)((cc_) * 32));  \
        }                                                                                                        \
    } while (0)
    // B (weight) fragments of position xi0 + j_ of chunk cc_: global -> registers, 16 bytes per lane and cout group (every U element
    // is used by exactly one wave); the compiler orders their uses with counted vmcnt waits
#define W2_LOAD_U(cc_, j_)                                                                                       \
    do {                                                                                                         \
        if ((cc_) < a.CC) {                                                                                      \
            const unsigned so_ = (unsigned)(cc_) * u_chunk + (unsigned)(j_) * u_pos;                             \
            fbU[j_][0] = buf_load16(a.u, a.u_bytes, u_voff, so_);                                                \
            fbU[j_][1] = buf_load16(a.u, a.u_bytes, u_voff, so_ + 1024u);                                        \
        }                                                                                                        \
    } while (0)
    // the 8 MFMAs of one position: k = 0..7 -> c = k >> 1 (K step), g = k & 1 (cout group)
#define W2_MFMA8(j_, fa_, k_) acc[j_][(k_) & 1] = mfma32((fa_)[(k_) >> 1], fbU[j_][(k_) & 1][(k_) >> 1], acc[j_][(k_) & 1])
#define W2_READ_FRAGS(buf_, vbase_, j_) fa[buf_] = lds_f4((vbase_) + fragA + (j_) * (T * 32))
    // input transform of one item (tile, channel pair) of this wave's row: 8 reads, 8 packed ops, 4 writes
#define W2_T_READ(q_, pbase_)                                                                                    \
    _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                                           \
        da[q_][jj] = lds_f2((pbase_) + src_a[q_] + jj * 16);                                                     \
        db[q_][jj] = lds_f2((pbase_) + src_b[q_] + jj * 16);                                                     \
    }
#define W2_T_MATH(q_)                                                                                            \
    do {                                                                                                         \
        f32x2 t_[4];                                                                                             \
        _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) t_[jj] = pk_fma(db[q_][jj], sg, da[q_][jj]);            \
        vv[q_][0] = pk_sub(t_[0], t_[2]);                                                                        \
        vv[q_][1] = pk_add(t_[1], t_[2]);                                                                        \
        vv[q_][2] = pk_sub(t_[2], t_[1]);                                                                        \
        vv[q_][3] = pk_sub(t_[1], t_[3]);                                                                        \
    } while (0)
#define W2_T_WRITE(q_, vbase_)                                                                                   \
    _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) *reinterpret_cast<f32x2*>((vbase_) + dst_v[q_] + jj * (T * 32)) = vv[q_][jj];

    // workgroup barrier WITHOUT the vmcnt(0) that __syncthreads() adds when LDS-DMA is in flight (own LDS accesses drained)
#define W2_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    unsigned item = blockIdx.x;
    W2_SETUP(item);
    W2_ISSUE_P(0);
    W2_ISSUE_P(1);
    f32x4 fbU[4][2];         // B fragments of this wave's four positions x two cout groups (refilled in a rolling fashion)
#pragma unroll
    for (int j = 0; j < 4; ++j) W2_LOAD_U(0, j);
    bool first = true;
    while (true) {
        f32x16 acc[4][2];        // [position j of row `wave`][cout group]
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 2; ++g) acc[j][g] = mfma_zero();
        f32x4 fa[2];             // double-buffered A fragments
        f32x2 da[2][4], db[2][4], vv[2][4];

        // patch 0/1 and U of chunk 0 landed (this wave's parts)?  Their DMAs are followed in this wave's VMEM queue by the previous
        // item's 32 output stores (+ 32 residual loads): a counted wait lets those stay in flight
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (a.res) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");      /* 64 newer ops: at most one of them is waited for too */
        else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        first = false;
        W2_BARRIER();                                         // ... and everybody's
        {   // input transform of chunk 0 (not overlapped with this workgroup's MFMAs — the other workgroup of the CU has some)
            W2_T_READ(0, sP);
            W2_T_READ(1, sP);
            W2_T_MATH(0);
            W2_T_MATH(1);
            W2_T_WRITE(0, sV);
            W2_T_WRITE(1, sV);
        }
        W2_BARRIER();                                         // patch 0 consumed everywhere: its buffer may be refilled (chunk 0, slice 2)
        W2_READ_FRAGS(0, sV, 0);

        // One chunk = 32 MFMAs in 32 slices fenced by sched_barrier(0).  The vmcnt waits are COUNTED: LDS-DMAs complete in issue
        // order, so "U of position j landed" = "at most n newer DMAs outstanding", n = what this wave issues after it: per steady
        // chunk 2 (U slot 0, slice 1), 1 (patch, slice 2; a second one on waves 0-1, which makes their waits one instruction
        // stricter than necessary), 2, 2, 2 (U slots 1-3, slices 9, 17, 25).  TAIL_ = the chunk before the last, which fetches no
        // patch (there is no chunk CC): one less in the counts.
        //   slice 0      patch reads of the next chunk's transform          slice 4-7  transform math (two bursts) and V writes
        //   slice 1+8j   refill U slot of position j for chunk cc+1          slice 2    patch DMA for chunk cc+2
        //   slice 2+8j   fragments of position j+1 (j < 3)                   slice 26   fragments of position 0 of chunk cc+1
#define W2_WAITV(n_, nt_, TAIL_)                                                                                 \
    do {                                                                                                         \
        if (TAIL_) asm volatile("s_waitcnt vmcnt(" #nt_ ")" ::: "memory");                                       \
        else asm volatile("s_waitcnt vmcnt(" #n_ ")" ::: "memory");                                              \
    } while (0)
#define W2_CHUNK(TAIL_, PAR_, CX_)  /* chunk CX_; PAR_ = CX_ & 1 (a literal in the unrolled loop: LDS addresses = register + immediate) */ \
        do {                                                                                                     \
            const char* vB = sV + (PAR_) * V_BYTES;                                                              \
            char* vN = sV + (1 - (PAR_)) * V_BYTES;                                                              \
            const char* pN = sP + (1 - (PAR_)) * P_BYTES;                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
            _Pragma("unroll") for (int k = 0; k < 32; ++k) {                                                     \
                const int j = k >> 3, kk = k & 7, buf = j & 1;                                                   \
                W2_MFMA8(j, fa[buf], kk);                                                                        \
                if (kk == 7) W2_LOAD_U((CX_) + 1, j);            /* position j is done: refill its fragments for the next chunk */ \
                if (k == 0) { W2_T_READ(0, pN); W2_T_READ(1, pN); }                                              \
                if (kk == 2 && j < 3) {                       /* next position's fragments, 6 MFMAs ahead of use */ \
                    W2_READ_FRAGS(buf ^ 1, vB, j + 1);                                                           \
                }                                                                                                \
                if (k == 2) W2_ISSUE_P((CX_) + 2);                                                                  \
                if (k == 4) { W2_T_MATH(0); }                                                                    \
                if (k == 5) { W2_T_MATH(1); }                                                                    \
                if (k == 6) { W2_T_WRITE(0, vN); }                                                               \
                if (k == 7) { W2_T_WRITE(1, vN); }                                                               \
                if (k == 26) {                                /* first position of the next chunk (own V rows, own U slot) */ \
                    W2_READ_FRAGS(0, vN, 0);                                                                     \
                }                                                                                                \
                __builtin_amdgcn_sched_barrier(0);                                                               \
            }                                                                                                    \
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    /* this wave's part of patch cc+2 landed (8 newer fragment loads) */ \
            W2_BARRIER();                                     /* everybody's; and patch cc+1 consumed everywhere */ \
        } while (0)

        int cc = 0;
        for (; cc + 3 < a.CC; cc += 2) {
            W2_CHUNK(false, 0, cc);
            W2_CHUNK(false, 1, cc + 1);
        }
        if (cc + 2 < a.CC) {                                  // at most one more steady chunk
            W2_CHUNK(false, (cc & 1), cc);
            ++cc;
        }
        if (cc + 1 < a.CC) {
            W2_CHUNK(true, (cc & 1), cc);
            ++cc;
        }
#undef W2_CHUNK
        {   // last chunk: MFMAs only (position 0 is already in registers)
            const char* vB = sV + (cc & 1) * V_BYTES;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int buf = j & 1;
                if (j < 3) W2_READ_FRAGS(buf ^ 1, vB, j + 1);
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) W2_MFMA8(j, fa[buf], kk);
            }
        }

        // ---- epilogue: Y = A^T M A.  Stage 1 (this wave's row of positions, in registers): q_c = sum_j A^T[c][j] M[i][j]; the
        // four rows meet through LDS, one cout group per pass ([4 i][2 c][32 tiles][32 co] = 32 KB = the two V buffers) ----
        float* sQ = reinterpret_cast<float*>(smem);
        const int co = tid & 31;
        const int en = n, ey0 = y0, ex0 = x0, en0 = n0;        // this item's coordinates (the setup below moves on to the next)
        const bool full = (y0 + 8 <= a.H) && (x0 + 16 <= a.W) && (n0 + BN <= a.Cout);
        const unsigned next = item + gridDim.x;
        const bool more = next < (unsigned)a.blocks;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            W2_BARRIER();                                      // everyone is done reading V / the patches (g = 0) or sQ (g = 1)
            if (g == 0 && more) {                              // patch buffers and fragment registers are idle from here on
                W2_SETUP(next);
                W2_ISSUE_P(0);
                W2_ISSUE_P(1);
#pragma unroll
                for (int j = 0; j < 4; ++j) W2_LOAD_U(0, j);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tl = (r & 3) + 8 * (r >> 2) + 4 * hi;
                const float m0 = acc[0][g][r], m1 = acc[1][g][r], m2 = acc[2][g][r], m3 = acc[3][g][r];
                sQ[((wave * 2 + 0) * 32 + tl) * 32 + (lane & 31)] = m0 + m1 + m2;
                sQ[((wave * 2 + 1) * 32 + tl) * 32 + (lane & 31)] = m1 - m2 - m3;
            }
            W2_BARRIER();
            // Stage 2: thread = (tile, co): Y[a][c] = sum_i A^T[a][i] q[i][c]; 4 tiles per thread and pass
            const int col = en0 + g * 32 + co;
            const bool col_ok = col < a.Cout;
            const float bv = col_ok ? a.bias[col] : 0.f;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int tl = (tid >> 5) + 8 * it;            // tile inside the 4 x 8 block
                const int oy = ey0 + 2 * (tl >> 3), ox = ex0 + 2 * (tl & 7);
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
                    for (int c = 0; c < 2; ++c) q[i][c] = sQ[((i * 2 + c) * 32 + tl) * 32 + co];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float ya = q[0][c] + q[1][c] + q[2][c];
                    const float yb = q[1][c] - q[2][c] - q[3][c];
                    buf_store(fmaxf(ya + bv + rv[0][c], lo), a.y, a.y_bytes, ok[0][c] ? y_voff : OOB, (unsigned)(c * a.ldy * 4));
                    buf_store(fmaxf(yb + bv + rv[1][c], lo), a.y, a.y_bytes, ok[1][c] ? y_voff : OOB, (unsigned)((a.W + c) * a.ldy * 4));
                }
            }
        }
        if (!more) break;
        item = next;
    }
#undef W2_MFMA8
#undef W2_ISSUE_P
#undef W2_LOAD_U
#undef W2_SETUP
}

}  // namespace cnl_wino2

// Launch (arguments already validated by cnl_conv3x3_winograd_f32).
int cnl_wino2_launch(const cnl_conv_params* p, size_t u_floats, void* stream) {
    using namespace cnl_wino2;
    Args a;
    a.x = p->x; a.u = p->w; a.bias = p->bias; a.res = p->residual; a.y = p->y;
    const int upf = (p->flags & CNL_UPSAMPLE_IN) ? 2 : 1;
    a.N = p->N; a.Hs = p->H_in; a.Ws = p->W_in; a.H = p->H_in * upf; a.W = p->W_in * upf; a.Cin = p->Cin; a.Cout = p->Cout;
    a.CoutP = (p->Cout + 63) / 64 * 64;
    a.ldx = p->ldx; a.ldy = p->ldy; a.ldr = p->ldr;
    a.CC = p->Cin / 8;
    a.nb = a.CoutP / BN; a.bx = (a.W + 15) / 16; a.by = (a.H + 7) / 8;
    const long long blocks = (long long)p->N * a.by * a.bx * a.nb;
    CNL_REQUIRE(blocks < (1ll << 31), CNL_E_UNSUPPORTED, "cnl_conv3x3_winograd_f32: grid too large");
    a.blocks = (int)blocks;
    const unsigned long long xb = (((unsigned long long)p->N * p->H_in * p->W_in - 1) * p->ldx + p->Cin) * 4ull;
    const unsigned long long ub = (unsigned long long)u_floats * 4ull;
    const unsigned long long Mo = (unsigned long long)p->N * a.H * a.W;
    const unsigned long long yb = ((Mo - 1) * p->ldy + p->Cout) * 4ull;
    const unsigned long long rb = p->residual ? ((Mo - 1) * p->ldr + p->Cout) * 4ull : 0ull;
    const unsigned long long slack = (unsigned long long)(a.W + 2) * 4ull;     // scalar-offset reach of the epilogue stores
    CNL_REQUIRE(xb < 0xFFFFFF00ull && ub < 0xFFFFFF00ull && yb + slack * p->ldy < 0xFFFFFF00ull && rb + slack * (p->residual ? p->ldr : 0) < 0xFFFFFF00ull,
                CNL_E_UNSUPPORTED, "cnl_conv3x3_winograd_f32: tensor spans >= 4 GiB; split the batch");
    a.x_bytes = (unsigned)xb; a.u_bytes = (unsigned)ub; a.y_bytes = (unsigned)yb; a.r_bytes = (unsigned)rb;
    a.flags = p->flags;
    // persistent workgroups: two per CU (45 KB of LDS each), walking the work items with stride gridDim.x
    static cnl::DeviceOnce once;
    int n_cu = 0;
    const int rc = cnl::kernel_setup(once, reinterpret_cast<const void*>(&winograd2_kernel), LDS_BYTES, &n_cu);
    if (rc != CNL_OK) return rc;
    const unsigned grid = (unsigned)(blocks < 2ll * n_cu ? blocks : 2ll * n_cu);
    hipLaunchKernelGGL(winograd2_kernel, dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, a);
    return cnl::check_launch("winograd2_kernel");
}
