// conv_mfma.hip — fused Conv2d(+BN folded)+bias(+residual)(+ReLU|sigmoid) on NHWC fp32 for gfx950.
//
// Replaces the ATen/cuDNN conv2d + batch_norm + relu + add (+ upsample / sigmoid) call sites of the
// reference's forward (models/meta.py:21-47, models/layers.py:72-77,99,152,160-177; torchvision
// BasicBlock) — see include/centernet_gfx950.h for the per-call-site list.
//
// Algorithm: implicit GEMM, exact fp32 on the matrix cores.
//   C[m][co] = sum_k A[m][k] * B[co][k],   m = (n,oy,ox),  k = (ky,kx,ci),  K = KH*KW*Cin
//   A[m][k]  = x[n, oy*s+ky-p, ox*s+kx-p, ci]   (zero outside the image; optional nearest-2x source)
//   B[co][k] = w[co][ky][kx][ci]                (OHWI, K contiguous)
// Tiling: workgroup tile BM x BN, BK = 32 floats (one 128-byte row per pixel / output channel);
// 4 waves, each owning TM x TN accumulator tiles of 32x32 (v_mfma_f32_32x32x2_f32, 64-wide wavefront).
// Staging: buffer_load_dwordx4 ... lds (LDS-DMA, no VGPR round trip).  The DMA writes LDS
// lane-linearly, so the bank-conflict swizzle is applied on the SOURCE address (which 16-byte slot of
// the 128-byte row a lane fetches) and again on the ds_read_b128 address; out-of-image taps and
// M/Cout tails are produced by the buffer descriptor's bounds check (offset >= num_records -> 0).
// K order inside a BK chunk is permuted (lanes 0-31 take floats 8j..8j+3, lanes 32-63 take 8j+4..8j+7
// of read j) so one ds_read_b128 feeds four MFMAs; A and B use the same permutation, so the sum is
// unchanged up to fp32 summation order.
// Pipeline: 2 LDS stages, one barrier per K chunk, the next chunk's DMA (address math included) issued
// under the current chunk's MFMAs, fragments for read j+1 prefetched while read j's MFMAs run,
// 2 workgroups per CU.  Block ids are remapped so each XCD works on a contiguous run of tiles.
//
// Instruction budget outside the MFMA stream.  A workgroup's tile prologue / epilogue runs beside the
// co-resident workgroup's MFMA stream and only gets the issue slots that stream leaves free: measured,
// a ~3000-instruction prologue took 20-35 us there (5.6 us alone) — a quarter of a K=576 tile.  So the
// prologue is kept to a few hundred instructions: magic-number division for (n,oy,ox), compile-time tap
// loops (KS template), accumulators zeroed by 4 MFMAs instead of 64 v_mov, and the epilogue addresses
// ride on the buffer instructions' scalar offset (no per-store VALU address math).
#include "conv_args.h"
#include <cstdlib>

namespace cnl_conv {

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_zero() {        // 16 zeroed accumulator registers from ONE instruction
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    return __builtin_amdgcn_mfma_f32_32x32x2f32(0.f, 0.f, z, 0, 0, 0);
}
// Epilogue for one 32x32 accumulator tile: + bias (+ residual) -> max(.,lo) -> (sigmoid) -> NHWC store.
// C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
// `voff` = byte offset of (row 0 of this lane, col); the row's r-dependent part goes in the scalar offset.
template <bool CHECK, bool RES>
__device__ __forceinline__ void store_tile(const f32x16& acc, float bv, float lo, float hi6, bool sigm, const ConvArgs& a, unsigned y_voff,
                                           unsigned r_voff, int m_base, bool col_ok) {
    float v[16];
    if constexpr (RES) {
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ro = (r & 3) + 8 * (r >> 2);
            const bool ok = !CHECK || (col_ok && m_base + ro < a.M);
            rv[r] = buf_load(a.res, a.r_bytes, ok ? r_voff : OOB, (unsigned)(ro * a.ldr * 4));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[r] + bv + rv[r];
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[r] + bv;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = fminf(fmaxf(v[r], lo), hi6);
    if (sigm) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = cnl::fast_sigmoid(v[r]);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ro = (r & 3) + 8 * (r >> 2);
        const bool ok = !CHECK || (col_ok && m_base + ro < a.M);
        buf_store(v[r], a.y, a.y_bytes, ok ? y_voff : OOB, (unsigned)(ro * a.ldy * 4));
    }
}

// KS: compile-time square kernel size (1 or 3), or 0 = run-time KH x KW (<= 32 taps).
template <int WM, int WN, int TM, int TN, int KS, bool UP_IN>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvArgs a) {
    using C = Cfg<WM, WN, TM, TN>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int hi = lane >> 5;
#ifdef CNL_TRACE
    const long long t_start = wall_clock64();
#endif

    const unsigned tile = cnl::xcd_remap(blockIdx.x, (unsigned)a.tiles);
    const int n_tile = tile % a.tiles_n;
    const int m_tile = tile / a.tiles_n;
    const int m0 = m_tile * C::BM;
    const int n0 = n_tile * C::BN;
    const int KH = KS ? KS : a.KH, KW = KS ? KS : a.KW;

    // ---- per-lane staging bookkeeping (once per tile) ----
    // Row r of the A tile is output pixel m0+r; this lane fetches rows (j*NW+wave)*8 + lane/8, j < A_INSTR, always the
    // same 16-byte slot.  a_mask bit t: tap t=(ky*KW+kx) of that row lies inside the image (and m < M); a_base: byte
    // offset of tap (0,0), channel 0 (+ swizzled slot).  Per chunk a wave-uniform delta is added — in the VECTOR
    // offset: a_base is "negative" (wrapped) for border pixels and the bounds check looks at the vector offset alone.
    const int lrow = lane >> 3;      // row within the 8-row group one DMA instruction covers
    const int pslot = lane & 7;      // physical 16-B slot inside the 128-B LDS row
    unsigned a_mask[C::A_INSTR], a_base[C::A_INSTR];
    int a_iy0[UP_IN ? C::A_INSTR : 1], a_ix0[UP_IN ? C::A_INSTR : 1];
#pragma unroll
    for (int j = 0; j < C::A_INSTR; ++j) {
        const int r = (j * C::NW + wave) * 8 + lrow;
        const int m = m0 + r;
        const int q = (pslot ^ ((r >> 1) & 7)) * 4;          // logical slot (in floats) fetched into pslot
        const unsigned n = fast_div((unsigned)m, a.mg_hw, a.sh_hw);
        const unsigned rem = (unsigned)m - n * (unsigned)(a.Ho * a.Wo);
        const unsigned oy = fast_div(rem, a.mg_w, a.sh_w);
        const unsigned ox = rem - oy * (unsigned)a.Wo;
        const int iy0 = (int)oy * a.stride - a.pad;
        const int ix0 = (int)ox * a.stride - a.pad_x;
        unsigned mask = 0;
        if constexpr (KS != 0) {
            unsigned xb = 0;
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) xb |= ((unsigned)(ix0 + kx) < (unsigned)a.WL) ? (1u << kx) : 0u;
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) mask |= ((unsigned)(iy0 + ky) < (unsigned)a.HL) ? (xb << (ky * KS)) : 0u;
        } else {
            for (int ky = 0; ky < KH; ++ky)
                for (int kx = 0; kx < KW; ++kx)
                    if ((unsigned)(iy0 + ky) < (unsigned)a.HL && (unsigned)(ix0 + kx) < (unsigned)a.WL) mask |= 1u << (ky * KW + kx);
        }
        a_mask[j] = m < a.M ? mask : 0u;
        const int pix = (int)n * a.Hin * a.Win;
        if constexpr (UP_IN) {
            a_iy0[j] = iy0;
            a_ix0[j] = ix0;
            a_base[j] = (unsigned)((pix * a.ldx + q) * 4);
        } else {
            a_base[j] = (unsigned)(((pix + iy0 * a.Win + ix0) * a.ldx + q) * 4);   // may wrap; used only when the tap is valid
        }
    }
    unsigned b_off[C::B_INSTR];
#pragma unroll
    for (int j = 0; j < C::B_INSTR; ++j) {
        const int r = (j * C::NW + wave) * 8 + lrow;
        const int q = (pslot ^ ((r >> 1) & 7)) * 4;
        b_off[j] = (unsigned)(((n0 + r) * a.K + q) * 4);   // rows >= Cout land beyond w_bytes -> zeros
    }

    // Per-lane DMA source offsets of the CURRENT tap (a_voff): recomputed only when the tap changes, i.e. once per CC chunks; the
    // channel offset of a chunk inside the tap is wave-uniform and rides in the scalar offset, so a chunk's DMA issue needs no
    // VALU at all (VALU work does not overlap the matrix pipe on gfx950: tools/mfma_coexec.hip).  The tap part must stay in the
    // VECTOR offset: a_base is "negative" (wrapped) for border pixels and the bounds check looks at the vector offset alone.
    unsigned a_voff[C::A_INSTR];
#define CNL_TAP(tap_, ky_, kx_)                                                                                   \
    do {                                                                                                          \
        const unsigned bit_ = 1u << (tap_);                                                                       \
        const unsigned delta_ = UP_IN ? 0u : (unsigned)((((ky_) * a.Win + (kx_)) * a.ldx) * 4);                   \
        _Pragma("unroll") for (int j = 0; j < C::A_INSTR; ++j) {                                                  \
            unsigned off_ = a_base[j] + delta_;                                                                   \
            if constexpr (UP_IN)                                                                                  \
                off_ += (unsigned)(((((a_iy0[j] + (ky_)) >> 1) * a.Win + ((a_ix0[j] + (kx_)) >> 1)) * a.ldx) * 4); \
            a_voff[j] = (a_mask[j] & bit_) ? off_ : OOB;                                                          \
        }                                                                                                         \
    } while (0)
    // LDS-DMA of K chunk (channels c0_.. of the current tap) into `stage_`.  (A macro, not a lambda: see the note above.)
#define CNL_ISSUE(stage_, c0_, kbase_)                                                                            \
    do {                                                                                                          \
        char* sA_ = smem + (stage_) * C::STAGE_BYTES;                                                             \
        char* sB_ = sA_ + C::BM * 128;                                                                            \
        _Pragma("unroll") for (int j = 0; j < C::A_INSTR; ++j)                                                    \
            dma16(a.x, a.x_bytes, sA_ + (j * C::NW + wave) * 1024, a_voff[j], (unsigned)((c0_) * 4));             \
        _Pragma("unroll") for (int j = 0; j < C::B_INSTR; ++j)                                                    \
            dma16(a.w, a.w_bytes, sB_ + (j * C::NW + wave) * 1024, b_off[j], (unsigned)((kbase_) * 4));           \
    } while (0)

    int tap = 0, ky = 0, kx = 0, cc = 0;     // position of the chunk being ISSUED
#define CNL_ADVANCE()                                   \
    do {                                                \
        if (++cc == a.CC) {                             \
            cc = 0;                                     \
            ++tap;                                      \
            if (++kx == KW) { kx = 0; ++ky; }           \
            CNL_TAP(tap, ky, kx);                       \
        }                                               \
    } while (0)
#ifdef CNL_TRACE
    const long long t_pro = wall_clock64();
    const long long c_pro = clock64();
#endif
    CNL_TAP(0, 0, 0);
    CNL_ISSUE(0, 0, 0);                      // first chunk in flight before anything else

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma_zero();

    // fragment read addresses (bytes inside a stage): row * 128 + ((2*jj + hi) ^ swz) * 16
    const int swz = (lane >> 1) & 7;
    const int a_row_byte = (wm * TM * 32 + (lane & 31)) * 128;
    const int b_row_byte = C::BM * 128 + (wn * TN * 32 + (lane & 31)) * 128;
    f32x4 af[2][TM], bf[2][TN];

#define CNL_READ(buf_, stage_ptr_, jj_)                                                                           \
    do {                                                                                                          \
        const int sb_ = (((2 * (jj_) + hi) ^ swz) << 4);                                                          \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) af[buf_][i] = lds_read16((stage_ptr_) + a_row_byte + i * 32 * 128 + sb_); \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) bf[buf_][j] = lds_read16((stage_ptr_) + b_row_byte + j * 32 * 128 + sb_); \
    } while (0)
#define CNL_MFMA(buf_)                                                                                            \
    do {                                                                                                          \
        _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                             \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                        \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                    \
                    acc[i][j] = mfma32(af[buf_][i][c], bf[buf_][j][c], acc[i][j]);                                \
    } while (0)
    // scheduling pin for "issue the next group's fragment reads right after the first MFMA of this group": hipcc otherwise
    // sinks the ds_reads to just before their first use and exposes the LDS latency four times per chunk
#define CNL_SCHED_RM()                                                     \
    do {                                                                   \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 \
        __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);           \
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN - 1, 0);   \
    } while (0)
    // One K chunk (64 MFMAs per wave).  Fragments are always read one 16-MFMA group ahead of their use, and the
    // hand-over to the next chunk (DMA wait + barrier + first fragment read of chunk kt+1 + DMA issue of chunk kt+2)
    // happens BEFORE the last MFMA group of chunk kt, so neither the barrier nor LDS latency is exposed:
    //   barrier B_kt guarantees (a) every wave has finished reading chunk kt's stage -> it may be refilled with chunk kt+2,
    //                           (b) every wave's DMA of chunk kt+1 has landed        -> it may be read.
#define CNL_CHUNK(kt_, SYNC_, ISSUE_)                                                                             \
    do {                                                                                                          \
        const char* sS = smem + ((kt_) & 1) * C::STAGE_BYTES;                                                     \
        const char* sN = smem + (((kt_) + 1) & 1) * C::STAGE_BYTES;                                               \
        CNL_READ(1, sS, 1);                                                                                       \
        CNL_MFMA(0);                                                                                              \
        CNL_SCHED_RM();                                                                                           \
        CNL_READ(0, sS, 2);                                                                                       \
        CNL_MFMA(1);                                                                                              \
        CNL_SCHED_RM();                                                                                           \
        CNL_READ(1, sS, 3);                                                                                       \
        CNL_MFMA(0);                                                                                              \
        CNL_SCHED_RM();                                                                                           \
        if (SYNC_) {                                                                                              \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                      \
            __syncthreads();                                                                                      \
            if (ISSUE_) {                                                                                         \
                CNL_ADVANCE();                                                                                    \
                CNL_ISSUE((kt_) & 1, cc * 32, ((kt_) + 2) * 32);                                                  \
            }                                                                                                     \
            CNL_READ(0, sN, 0);                                                                                   \
        }                                                                                                         \
        CNL_MFMA(1);                                                                                              \
        if (SYNC_) {                                                                                              \
            __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);   /* next chunk's first fragments at once */ \
            if (ISSUE_) {                                                                                         \
                _Pragma("unroll") for (int g = 0; g < C::A_INSTR + C::B_INSTR; ++g) {                             \
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                            \
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  /* one LDS-DMA per MFMA */                \
                }                                                                                                 \
            }                                                                                                     \
        }                                                                                                         \
    } while (0)

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // chunk 0 landed (this wave) ...
    __syncthreads();                                    // ... and everyone's
    if (a.KT > 1) {
        CNL_ADVANCE();
        CNL_ISSUE(1, cc * 32, 32);
    }
    CNL_READ(0, smem, 0);
    int kt = 0;
    for (; kt + 2 < a.KT; ++kt) CNL_CHUNK(kt, true, true);
    if (kt + 1 < a.KT) {
        CNL_CHUNK(kt, true, false);
        ++kt;
    }
    CNL_CHUNK(kt, false, false);
#undef CNL_CHUNK
#undef CNL_SCHED_RM
#undef CNL_ADVANCE
#undef CNL_TAP
#undef CNL_ISSUE
#undef CNL_READ
#undef CNL_MFMA
#ifdef CNL_TRACE
    const long long t_loop = wall_clock64();
    const long long c_loop = clock64();
#endif

    // ---- epilogue ----
    const float lo = (a.flags & (CNL_RELU | CNL_RELU6)) ? 0.f : -__builtin_inff();
    const float hi6 = (a.flags & CNL_RELU6) ? 6.f : __builtin_inff();
    const bool sigm = a.flags & CNL_SIGMOID;
    const bool full = (m0 + C::BM <= a.M) && (n0 + C::BN <= a.Cout);
    // Fuse epilogue through LDS (round 5): in the MFMA's C layout a lane holds ONE cout of 16 pixels — 4-byte loads of the skip tensor and 4-byte
    // stores, 64 of each per tile and 2x position, and one (n, oy, ox) decomposition per accumulator register.  Every wave turns its tiles
    // through a private part of the (now idle) staging buffers instead: written as [pixel][32 couts] rows (16-byte slots XOR-swizzled by the
    // pixel: conflict-free both ways), read back with 8 lanes per pixel — a lane then holds four consecutive couts: 16-byte loads and
    // stores, every instruction 8 full 128-byte lines, a quarter of the memory instructions.  Same values, same order of additions.
    bool fuse_x = false;
    if constexpr (KS == 1 && !UP_IN) {
        static_assert(TM * TN * 4096 * C::NW <= C::LDS_BYTES, "the wave-private transpose regions live in the staging buffers");
        fuse_x = (a.flags & CNL_UPSAMPLE_OUT_ADD) && !(a.flags & CNL_I_SUBPIXEL) && a.Cout % 4 == 0 && a.ldy % 4 == 0 && a.ldr % 4 == 0 &&
                 (((uintptr_t)a.y | (uintptr_t)a.res) & 15) == 0;
    }
    if (fuse_x) {
        __syncthreads();                                               // every wave is done with the last chunk's operands
        char* const T = smem + wave * (TM * TN * 4096);
        const int cl = lane & 31;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int p = (r & 3) + 8 * (r >> 2) + 4 * hi;
                    *reinterpret_cast<float*>(T + (i * TN + j) * 4096 + p * 128 + (((cl >> 2) ^ (p & 7)) << 4) + (cl & 3) * 4) = acc[i][j][r];
                }
        const unsigned Wo2 = 2u * (unsigned)a.Wo;
        const bool want_max = a.ymax != nullptr;
        const unsigned img0 = fast_div((unsigned)m0, a.mg_hw, a.sh_hw);
        float om0 = 0.f, om1 = 0.f;
        const int q = lane & 7, pr = lane >> 3;
        f32x4 bvj[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const unsigned c = (unsigned)(n0 + (wn * TN + j) * 32 + 4 * q);
            bvj[j] = buf_load4(a.bias, (unsigned)a.Cout * 4u, c < (unsigned)a.Cout ? c * 4u : OOB, 0);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int p = pr + 8 * k;
                const int m = m0 + (wm * TM + i) * 32 + p;
                const bool row_ok = m < a.M;
                const unsigned n = fast_div((unsigned)m, a.mg_hw, a.sh_hw);
                const unsigned rem = (unsigned)m - n * (unsigned)(a.Ho * a.Wo);
                const unsigned oy = fast_div(rem, a.mg_w, a.sh_w);
                const unsigned ox = rem - oy * (unsigned)a.Wo;
                const unsigned pix = (n * 2u * (unsigned)a.Ho + 2u * oy) * Wo2 + 2u * ox;      // < 2^30: checked on the host (4 GiB rule)
                float ov = 0.f;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const unsigned c = (unsigned)(n0 + (wn * TN + j) * 32 + 4 * q);
                    const bool ok = row_ok && c < (unsigned)a.Cout;
                    const unsigned y_v = ok ? (pix * (unsigned)a.ldy + c) * 4u : OOB, r_v = ok ? (pix * (unsigned)a.ldr + c) * 4u : OOB;
                    f32x4 rv[4];
#pragma unroll
                    for (int d = 0; d < 4; ++d) rv[d] = buf_load4(a.res, a.r_bytes, r_v, (((d >> 1) * Wo2 + (d & 1)) * (unsigned)a.ldr) * 4u);
                    const f32x4 v = lds_read16(T + (i * TN + j) * 4096 + p * 128 + ((q ^ (p & 7)) << 4)) + bvj[j];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        f32x4 o = v + rv[d];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            o[e] = fmaxf(o[e], lo);
                            if (ok) ov = fmaxf(ov, fabsf(o[e]));
                        }
                        buf_store4(o, a.y, a.y_bytes, y_v, (((d >> 1) * Wo2 + (d & 1)) * (unsigned)a.ldy) * 4u);
                    }
                }
                if (want_max && row_ok) {
                    if (n == img0) om0 = fmaxf(om0, ov);
                    else if (n == img0 + 1) om1 = fmaxf(om1, ov);
                    else cnl::report_max(a.ymax + n * AMS, ov);
                }
            }
        }
        if (want_max) {
            om0 = cnl::wave_max_nonneg(om0);
            om1 = cnl::wave_max_nonneg(om1);
            if (lane == 0) {
                cnl::report_max(a.ymax + img0 * AMS, om0);
                if (om1 > 0.f) cnl::report_max(a.ymax + (img0 + 1) * AMS, om1);
            }
        }
    } else
    if (!(a.flags & (CNL_UPSAMPLE_OUT_ADD | CNL_I_SUBPIXEL))) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
            const bool col_ok = col < a.Cout;
            const float bv = col_ok ? a.bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mb = m0 + (wm * TM + i) * 32 + 4 * hi;
                const unsigned y_voff = (unsigned)((mb * a.ldy + col) * 4);
                const unsigned r_voff = (unsigned)((mb * a.ldr + col) * 4);
                if (full) {
                    if (a.res) store_tile<false, true>(acc[i][j], bv, lo, hi6, sigm, a, y_voff, r_voff, mb, true);
                    else store_tile<false, false>(acc[i][j], bv, lo, hi6, sigm, a, y_voff, r_voff, mb, true);
                } else {
                    if (a.res) store_tile<true, true>(acc[i][j], bv, lo, hi6, sigm, a, y_voff, r_voff, mb, col_ok);
                    else store_tile<true, false>(acc[i][j], bv, lo, hi6, sigm, a, y_voff, r_voff, mb, col_ok);
                }
            }
        }
    } else {
        // scatter epilogues into a 2x-resolution output:
        //   CNL_UPSAMPLE_OUT_ADD  FPN Fuse: write the four 2x-upsampled positions, adding the skip tensor there (layers.py:160-174)
        //   CNL_I_SUBPIXEL        one phase of a stride-2 transposed conv: write position (sub_dy, sub_dx) only; the optional
        //                         residual is added AFTER the activation (Fuse: skip + resize(top), resize = deconv+BN+ReLU)
        const bool sub = a.flags & CNL_I_SUBPIXEL;
        float bvj[TN];
        bool colj[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
            colj[j] = col < a.Cout;
            bvj[j] = colj[j] ? a.bias[col] : 0.f;
        }
        const unsigned col0 = (unsigned)(n0 + wn * TN * 32 + (lane & 31));
        const unsigned Wo2 = 2u * (unsigned)a.Wo;
        // max |y| hand-over of the Fuse epilogue (the 3x3 output conv behind it takes it as x_absmax instead of a pass over this tensor): a tile's
        // rows lie in its first image or the next one (maps smaller than a tile: the rare rows report on their own)
        const bool want_max = a.ymax && !sub;
        const unsigned img0 = fast_div((unsigned)m0, a.mg_hw, a.sh_hw);
        float om0 = 0.f, om1 = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + (wm * TM + i) * 32 + 4 * hi;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // one (n, oy, ox) decomposition per accumulator row; the four 2x positions and the TN column groups only add
                // wave-uniform offsets, which ride in the buffer instructions' scalar offset
                const int m = mb + (r & 3) + 8 * (r >> 2);
                const bool row_ok = m < a.M;
                const unsigned n = fast_div((unsigned)m, a.mg_hw, a.sh_hw);
                const unsigned rem = (unsigned)m - n * (unsigned)(a.Ho * a.Wo);
                const unsigned oy = fast_div(rem, a.mg_w, a.sh_w);
                const unsigned ox = rem - oy * (unsigned)a.Wo;
                const unsigned pix = (n * 2u * (unsigned)a.Ho + 2u * oy) * Wo2 + 2u * ox;      // < 2^30: checked on the host (4 GiB rule)
                const unsigned y_v = (pix * (unsigned)a.ldy + col0) * 4u, r_v = (pix * (unsigned)a.ldr + col0) * 4u;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const bool ok = row_ok && colj[j];
                    const float v = acc[i][j][r] + bvj[j];
                    if (sub) {
                        const unsigned dp = (unsigned)a.sub_dy * Wo2 + (unsigned)a.sub_dx;
                        float o = fminf(fmaxf(v, lo), hi6);
                        if (a.res) o += buf_load(a.res, a.r_bytes, ok ? r_v : OOB, (dp * (unsigned)a.ldr + j * 32u) * 4u);
                        buf_store(o, a.y, a.y_bytes, ok ? y_v : OOB, (dp * (unsigned)a.ldy + j * 32u) * 4u);
                    } else {
                        float rv[4];
#pragma unroll
                        for (int d = 0; d < 4; ++d)
                            rv[d] = buf_load(a.res, a.r_bytes, ok ? r_v : OOB, (((d >> 1) * Wo2 + (d & 1)) * (unsigned)a.ldr + j * 32u) * 4u);
                        float ov = 0.f;
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const float o = fmaxf(v + rv[d], lo);
                            ov = fmaxf(ov, fabsf(o));
                            buf_store(o, a.y, a.y_bytes, ok ? y_v : OOB, (((d >> 1) * Wo2 + (d & 1)) * (unsigned)a.ldy + j * 32u) * 4u);
                        }
                        if (want_max && ok) {
                            if (n == img0) om0 = fmaxf(om0, ov);
                            else if (n == img0 + 1) om1 = fmaxf(om1, ov);
                            else cnl::report_max(a.ymax + n * AMS, ov);
                        }
                    }
                }
            }
        }
        if (want_max) {
            om0 = cnl::wave_max_nonneg(om0);
            om1 = cnl::wave_max_nonneg(om1);
            if (lane == 0) {
                cnl::report_max(a.ymax + img0 * AMS, om0);
                if (om1 > 0.f) cnl::report_max(a.ymax + (img0 + 1) * AMS, om1);
            }
        }
    }
#ifdef CNL_TRACE
    if (a.trace && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long* t = a.trace + (long)blockIdx.x * 8;
        t[0] = t_start; t[1] = t_pro; t[2] = t_pro; t[3] = t_loop; t[4] = wall_clock64();
        t[5] = c_loop - c_pro;
        t[6] = tile;
    }
#endif
}

template <int WM, int WN, int TM, int TN, int KS, bool UP_IN>
int launch_one(const ConvArgs& a, hipStream_t stream) {
    using C = Cfg<WM, WN, TM, TN>;
    static cnl::DeviceOnce once;            // one per template instantiation
    const int rc = cnl::kernel_setup(once, reinterpret_cast<const void*>(&conv_mfma_kernel<WM, WN, TM, TN, KS, UP_IN>), 160 * 1024);
    if (rc != CNL_OK) return rc;
    hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, TM, TN, KS, UP_IN>), dim3(a.tiles), dim3(C::THREADS), C::LDS_BYTES, stream, a);
    return cnl::check_launch("conv_mfma_kernel");
}

template <int WM, int WN, int TM, int TN>
int launch_cfg(const ConvArgs& in, hipStream_t stream) {
    using C = Cfg<WM, WN, TM, TN>;
    ConvArgs a = in;
    const int tiles_m = (a.M + C::BM - 1) / C::BM;
    a.tiles_n = (a.Cout + C::BN - 1) / C::BN;
    a.tiles = tiles_m * a.tiles_n;
    const bool up = a.flags & CNL_UPSAMPLE_IN;
    const int ks = (a.KH == a.KW && (a.KH == 1 || a.KH == 3)) ? a.KH : 0;
    if (ks == 3) return up ? launch_one<WM, WN, TM, TN, 3, true>(a, stream) : launch_one<WM, WN, TM, TN, 3, false>(a, stream);
    if (ks == 1 && !up) return launch_one<WM, WN, TM, TN, 1, false>(a, stream);
    return up ? launch_one<WM, WN, TM, TN, 0, true>(a, stream) : launch_one<WM, WN, TM, TN, 0, false>(a, stream);
}

// (magic, shift) with  n / d == umulhi(n, magic) >> shift  for every n < 2^31 (d >= 2); d == 1 -> shift 0xFF.
static void magic_u31(unsigned d, unsigned* magic, unsigned* shift) {
    if (d <= 1) { *magic = 0; *shift = 0xFFu; return; }
    unsigned s = 0;
    while ((1ull << s) < d) ++s;                       // ceil(log2 d)
    *magic = (unsigned)(((1ull << (31 + s)) / d) + 1);
    *shift = s - 1;
}

}  // namespace cnl_conv
using namespace cnl_conv;

// Derived fields of ConvArgs (a.Ho / a.Wo already set), the 4 GiB addressing checks, and the tile-shape dispatch.
// out4x: the launch writes into a tensor of 4x the conv's own output pixels (2x-resolution scatter epilogues).
static int finish_and_launch(ConvArgs& a, bool out4x, const char* who, hipStream_t s) {
    CNL_REQUIRE(a.Ho > 0 && a.Wo > 0, CNL_E_BAD_ARG, "%s: empty output", who);
    const long long M = (long long)a.N * a.Ho * a.Wo;
    CNL_REQUIRE(M < (1ll << 31) - 512, CNL_E_UNSUPPORTED, "%s: N*Ho*Wo too large", who);
    a.M = (int)M;
    a.CC = a.Cin / 32; a.KT = a.KH * a.KW * a.CC; a.K = a.KH * a.KW * a.Cin;
    const unsigned long long lim = 0xFFFFFF00ull;
    const unsigned long long xb = (((unsigned long long)a.N * a.Hin * a.Win - 1) * a.ldx + a.Cin) * 4ull;
    const unsigned long long wb = (unsigned long long)a.Cout * a.K * 4ull;
    const unsigned long long Mo = out4x ? 4ull * M : (unsigned long long)M;
    const unsigned long long yb = ((Mo - 1) * a.ldy + a.Cout) * 4ull;
    const unsigned long long rb = a.res ? ((Mo - 1) * a.ldr + a.Cout) * 4ull : 0ull;
    CNL_REQUIRE(xb < lim && yb + 512ull * a.ldy * 4 < lim && rb + 512ull * a.ldr * 4 < lim, CNL_E_UNSUPPORTED,
                "%s: a tensor spans >= 4 GiB (x %llu, y %llu bytes); split the batch", who, xb, yb);
    CNL_REQUIRE(wb + (unsigned long long)256 * a.K * 4ull < lim, CNL_E_UNSUPPORTED, "%s: weight too large", who);
    a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb; a.y_bytes = (unsigned)yb; a.r_bytes = (unsigned)rb;
    a.tiles_n = a.tiles = 0;
    magic_u31((unsigned)(a.Ho * a.Wo), &a.mg_hw, &a.sh_hw);
    magic_u31((unsigned)a.Wo, &a.mg_w, &a.sh_w);
    a.trace = nullptr;
#ifdef CNL_TRACE
    if (const char* e = getenv("CNL_TRACE_PTR")) a.trace = (long long*)strtoull(e, nullptr, 0);
#endif
    if (f16x2_eligible(a)) {                                         // conv_f16x2.hip: the caller handed over max |x|, max |w|
        if (a.ksplit > 1) {                  // split reduction: needs >= 2 chunks per slice and an unscattered output
            const int per = (a.KT + a.ksplit - 1) / a.ksplit;
            const unsigned long long pb = (unsigned long long)a.ksplit * (unsigned long long)M * a.Cout * 4ull;
            if (per >= 2 && !out4x && !(a.flags & CNL_UPSAMPLE_IN) && pb < lim) {
                a.kt_per = per;
                a.ksplit = (a.KT + per - 1) / per;                   // no empty slice
                a.part_bytes = (unsigned)((unsigned long long)a.ksplit * (unsigned long long)M * a.Cout * 4ull);
            } else {
                a.ksplit = 0;
            }
        }
        return f16x2_launch(a, s);
    }
    a.ksplit = 0;
    // Tile choice: BN follows Cout; shrink BM when the grid would not fill 256 CUs x 2 workgroups.
    if (a.Cout <= 32) return launch_cfg<4, 1, 2, 1>(a, s);          // 256 x 32
    if (a.Cout <= 64) return launch_cfg<4, 1, 2, 2>(a, s);          // 256 x 64
    if (a.Cout <= 96) return launch_cfg<4, 1, 1, 3>(a, s);          // 128 x 96 (the 80-class heatmap out_conv: 17 % instead of 37 % idle columns)
    const long long tiles128 = ((M + 127) / 128) * ((a.Cout + 127) / 128);
    if (tiles128 < 512) return launch_cfg<2, 2, 1, 2>(a, s);        //  64 x 128
    return launch_cfg<2, 2, 2, 2>(a, s);                            // 128 x 128
}

extern "C" int cnl_conv2d_out_hw(const cnl_conv_params* p, int32_t* H_out, int32_t* W_out) {
    CNL_REQUIRE(p && H_out && W_out, CNL_E_BAD_ARG, "cnl_conv2d_out_hw: null argument");
    CNL_REQUIRE(p->stride > 0 && p->KH > 0 && p->KW > 0, CNL_E_BAD_ARG, "cnl_conv2d_out_hw: bad kernel/stride");
    const int up = (p->flags & CNL_UPSAMPLE_IN) ? 2 : 1;
    *H_out = (p->H_in * up + 2 * p->pad - p->KH) / p->stride + 1;
    *W_out = (p->W_in * up + 2 * p->pad - p->KW) / p->stride + 1;
    return CNL_OK;
}

extern "C" size_t cnl_conv2d_splitk_scratch_bytes(const cnl_conv_params* p) {
    if (!p || p->splitk <= 1) return 0;
    int32_t ho = 0, wo = 0;
    if (cnl_conv2d_out_hw(p, &ho, &wo) != CNL_OK || ho <= 0 || wo <= 0) return 0;
    return (size_t)p->splitk * (size_t)p->N * (size_t)ho * (size_t)wo * (size_t)p->Cout * 4;
}

extern "C" int cnl_conv2d_kernel(const cnl_conv_params* p) {
    CNL_REQUIRE(p, CNL_E_BAD_ARG, "cnl_conv2d_kernel: null params");
    ConvArgs a;
    a.KH = p->KH; a.KW = p->KW; a.pad = a.pad_x = p->pad; a.flags = p->flags; a.Cout = p->Cout;
    a.xmax = p->x_absmax; a.wmax = p->w_absmax; a.wscale = nullptr; a.res = p->residual; a.algo = p->algo;
    a.wsplit = (p->flags & CNL_W_SPLIT) ? p->w : nullptr;              // (presence only)
    a.ksplit = p->splitk > 1 ? p->splitk : 0;
    int32_t ho = 0, wo = 0;
    const int rc = cnl_conv2d_out_hw(p, &ho, &wo);
    if (rc != CNL_OK) return rc;
    a.Ho = ho; a.Wo = wo;
    return f16x2_eligible(a) ? CNL_CONV_F16X2 : CNL_CONV_F32;
}

extern "C" int cnl_conv2d_nhwc_f32(const cnl_conv_params* p, void* stream) {
    CNL_REQUIRE(p, CNL_E_BAD_ARG, "cnl_conv2d_nhwc_f32: null params");
    CNL_REQUIRE(p->x && p->w && p->bias && p->y, CNL_E_BAD_ARG, "cnl_conv2d_nhwc_f32: null tensor pointer");
    CNL_REQUIRE(p->N > 0 && p->H_in > 0 && p->W_in > 0 && p->Cin > 0 && p->Cout > 0, CNL_E_BAD_ARG,
                "cnl_conv2d_nhwc_f32: non-positive dimension");
    CNL_REQUIRE(p->KH > 0 && p->KW > 0 && p->stride > 0 && p->pad >= 0, CNL_E_BAD_ARG,
                "cnl_conv2d_nhwc_f32: bad kernel/stride/pad");
    CNL_REQUIRE(p->KH * p->KW <= 32, CNL_E_UNSUPPORTED, "cnl_conv2d_nhwc_f32: KH*KW = %d > 32 taps", p->KH * p->KW);
    CNL_REQUIRE(p->Cin % 32 == 0, CNL_E_UNSUPPORTED,
                "cnl_conv2d_nhwc_f32: Cin=%d is not a multiple of 32 (use cnl_stem_conv7x7_f32 for the RGB stem)", p->Cin);
    CNL_REQUIRE(p->ldx >= p->Cin && p->ldy >= p->Cout && p->ldx % 4 == 0, CNL_E_BAD_ARG,
                "cnl_conv2d_nhwc_f32: pixel strides ldx=%d ldy=%d too small / misaligned", p->ldx, p->ldy);
    CNL_REQUIRE(((uintptr_t)p->x & 15) == 0 && ((uintptr_t)p->w & 15) == 0, CNL_E_BAD_ARG,
                "cnl_conv2d_nhwc_f32: x and w must be 16-byte aligned");
    const bool up_out = p->flags & CNL_UPSAMPLE_OUT_ADD;
    CNL_REQUIRE(!up_out || p->residual, CNL_E_BAD_ARG, "cnl_conv2d_nhwc_f32: CNL_UPSAMPLE_OUT_ADD needs a residual");
    CNL_REQUIRE(!up_out || !(p->flags & CNL_SIGMOID), CNL_E_UNSUPPORTED, "cnl_conv2d_nhwc_f32: sigmoid with UPSAMPLE_OUT_ADD");
    CNL_REQUIRE(!p->residual || p->ldr >= p->Cout, CNL_E_BAD_ARG, "cnl_conv2d_nhwc_f32: ldr=%d < Cout", p->ldr);

    ConvArgs a;
    a.x = p->x; a.w = p->w; a.bias = p->bias; a.res = p->residual; a.y = p->y;
    a.N = p->N; a.Hin = p->H_in; a.Win = p->W_in; a.Cin = p->Cin; a.Cout = p->Cout;
    a.KH = p->KH; a.KW = p->KW; a.stride = p->stride; a.pad = a.pad_x = p->pad;
    a.sub_dy = a.sub_dx = 0;
    a.ldx = p->ldx; a.ldy = p->ldy; a.ldr = p->ldr;
    a.flags = p->flags;
    a.xmax = p->x_absmax; a.wmax = p->w_absmax; a.ymax = reinterpret_cast<unsigned*>(p->y_absmax); a.wscale = nullptr; a.algo = p->algo;
    if (p->flags & CNL_W_SPLIT) {      // p->w is a cnl_conv_split_weights_f32 buffer: [fp32 OHWI][the same as scaled fp16 pieces][scale, 3 pad]
        CNL_REQUIRE(p->KH == p->KW && (p->KH == 1 || p->KH == 3), CNL_E_BAD_ARG, "cnl_conv2d_nhwc_f32: CNL_W_SPLIT with a %d x %d kernel", p->KH, p->KW);
        const size_t total = (size_t)p->Cout * p->KH * p->KW * p->Cin;
        a.wsplit = p->w + total;
        a.wscale = p->w + 2 * total;
    }
    const int up = (p->flags & CNL_UPSAMPLE_IN) ? 2 : 1;
    a.HL = p->H_in * up; a.WL = p->W_in * up;
    a.Ho = (a.HL + 2 * p->pad - p->KH) / p->stride + 1;
    a.Wo = (a.WL + 2 * p->pad - p->KW) / p->stride + 1;
    if (p->splitk > 1) {
        CNL_REQUIRE(p->splitk <= 256, CNL_E_BAD_ARG, "cnl_conv2d_nhwc_f32: splitk=%d (at most 256 slices)", p->splitk);
        CNL_REQUIRE(p->splitk_scratch && ((uintptr_t)p->splitk_scratch & 15) == 0 && p->splitk_scratch_bytes >= cnl_conv2d_splitk_scratch_bytes(p),
                    CNL_E_WORKSPACE, "cnl_conv2d_nhwc_f32: splitk_scratch missing, unaligned or smaller than cnl_conv2d_splitk_scratch_bytes()");
        a.ksplit = p->splitk;
        a.part = p->splitk_scratch;
    }
    return finish_and_launch(a, up_out, "cnl_conv2d_nhwc_f32", (hipStream_t)stream);
}

/*
 * 3x3 / stride 1 / pad 1 convolution on the NEAREST-2x UPSAMPLED input (make_upsample(nearest) + ConvBnAct: layers.py:99,72-77;
 * the first head block behind the simple neck) as four sub-pixel phases on the LOW-resolution input: rows 2y and 2y+1 of the
 * upsampled image are both row y, so for output row 2y+a the three kernel rows collapse onto two input rows —
 *   a = 0: rows y-1, y with weights w[0], w[1]+w[2];   a = 1: rows y, y+1 with weights w[0]+w[1], w[2]   (columns alike) —
 * i.e. four 2x2 convolutions with pre-summed weights: 16 instead of 36 multiplies per 2x2 output block (the same 2.25x as
 * Winograd F(2x2,3x3)), each an ordinary implicit GEMM with K = 4 Cin writing one phase of the 2x output grid.
 */
namespace cnl_conv {
__global__ __launch_bounds__(256) void up2_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout) {
    const long total = 16l * Cin * Cout;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int ci = (int)(e % Cin);
        long t = e / Cin;
        const int kx2 = (int)(t & 1), ky2 = (int)((t >> 1) & 1);
        t >>= 2;
        const int co = (int)(t % Cout), ph = (int)(t / Cout);
        const int a = ph >> 1, b = ph & 1;
        // kernel rows / columns that land on tap ky2 / kx2 of phase a / b
        const int ky_lo = a ? (ky2 ? 2 : 0) : (ky2 ? 1 : 0), ky_hi = a ? (ky2 ? 2 : 1) : (ky2 ? 2 : 0);
        const int kx_lo = b ? (kx2 ? 2 : 0) : (kx2 ? 1 : 0), kx_hi = b ? (kx2 ? 2 : 1) : (kx2 ? 2 : 0);
        float acc = 0.f;
        for (int ky = ky_lo; ky <= ky_hi; ++ky)
            for (int kx = kx_lo; kx <= kx_hi; ++kx) acc += w[((long)(co * 3 + ky) * 3 + kx) * Cin + ci];
        wp[e] = acc;
    }
}
// the packed phase weights once more as scaled fp16 pieces in the B-row layout of conv_f16x2.hip's sub-pixel variant: per cout row
// and 32-k chunk 8 slots of 16 bytes, slot (2 (2 g + h) + piece) = the 8 halves lane half h multiplies in 16-k group g
// (k = 16 g + 4 h + e for e < 4, 16 g + 8 + 4 h + e - 4 for e >= 4: the K permutation of the A fragments).  One workgroup: max |w| ->
// S_w = 2^(14 - e) -> hi = RN16(w S_w), lo = RZ16(w S_w - hi).  scal[0] = S_w.
__global__ __launch_bounds__(1024) void up2_split_kernel(const float* __restrict__ wp, unsigned short* __restrict__ ws, float* __restrict__ scal, long total) {
    __shared__ float red[16];
    const int tid = threadIdx.x;
    float mx = 0.f;
    for (long e = tid; e < total; e += 1024) {
        const float v = fabsf(wp[e]);
        mx = (v < __builtin_inff()) ? fmaxf(mx, v) : mx;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = 0.f;
    for (int i = 0; i < 16; ++i) mx = fmaxf(mx, red[i]);
    float Sw = 1.f;
    if (mx > 0.f) {
        int ex;
        (void)__builtin_frexpf(mx, &ex);
        ex = 14 - ex;
        Sw = __builtin_ldexpf(1.f, ex < -60 ? -60 : (ex > 60 ? 60 : ex));
    }
    if (tid == 0) scal[0] = Sw;
    for (long e = tid; e < total; e += 1024) {          // e = (row-chunk index) * 32 + kk over the fp32 layout [..][K] (K % 32 == 0)
        const int kk = (int)(e & 31);
        const long rc = e >> 5;
        const int g = kk >> 4, r16 = kk & 15;
        const int h = (r16 >> 2) & 1, el = (r16 & 3) + ((r16 >> 3) << 2);      // inverse of k = 16 g + 4 h + e (e < 4) / 16 g + 8 + 4 h + e - 4
        const float v = wp[e] * Sw;
        const _Float16 hv = (_Float16)v;
        const float r = v - (float)hv;
        unsigned lo2;
        asm("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(lo2) : "v"(r), "v"(0.f));
        const long base = rc * 64 + (2 * (2 * g + h)) * 8 + el;               // in halves: 64 per row-chunk, 8 per slot
        ws[base] = __builtin_bit_cast(unsigned short, hv);
        ws[base + 8] = (unsigned short)(lo2 & 0xFFFFu);
    }
}
}  // namespace cnl_conv

// [fp32 phase weights 16 Cin Cout][their fp16 split, same size][4 scalars]
extern "C" size_t cnl_up2_weight_floats(int32_t Cin, int32_t Cout) {
    if (Cin <= 0 || Cout <= 0) return 0;
    return (size_t)32 * Cin * Cout + 4;
}

extern "C" int cnl_up2_pack_weights_f32(const float* w_ohwi, float* w_packed, int32_t Cin, int32_t Cout, void* stream) {
    CNL_REQUIRE(w_ohwi && w_packed && Cin > 0 && Cout > 0, CNL_E_BAD_ARG, "cnl_up2_pack_weights_f32: null pointer / non-positive size");
    const long total = 16l * Cin * Cout;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    CNL_REQUIRE(Cin % 32 == 0, CNL_E_UNSUPPORTED, "cnl_up2_pack_weights_f32: Cin=%d is not a multiple of 32", Cin);
    hipLaunchKernelGGL(up2_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w_ohwi, w_packed, Cin, Cout);
    hipLaunchKernelGGL(up2_split_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, w_packed, reinterpret_cast<unsigned short*>(w_packed + total),
                       w_packed + 2 * total, total);
    return cnl::check_launch("up2_pack_kernel / up2_split_kernel");
}

// [fp32 OHWI weights][their scaled fp16 split in conv_f16x2.hip's B-row layout, same size][scale + 3 pad floats]
extern "C" size_t cnl_conv_split_weight_floats(int32_t Cin, int32_t Cout, int32_t KH, int32_t KW) {
    if (Cin <= 0 || Cout <= 0 || KH != KW || (KH != 1 && KH != 3) || Cin % 32) return 0;
    return (size_t)2 * Cout * KH * KW * Cin + 4;
}

extern "C" int cnl_conv_split_weights_f32(const float* w_ohwi, float* w_buf, int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, void* stream) {
    CNL_REQUIRE(w_ohwi && w_buf, CNL_E_BAD_ARG, "cnl_conv_split_weights_f32: null pointer");
    CNL_REQUIRE(cnl_conv_split_weight_floats(Cin, Cout, KH, KW) != 0, CNL_E_UNSUPPORTED,
                "cnl_conv_split_weights_f32: Cin=%d Cout=%d kernel %d x %d (needs Cin %% 32 == 0 and a square 1x1 / 3x3 kernel)", Cin, Cout, KH, KW);
    CNL_REQUIRE(((uintptr_t)w_buf & 15) == 0, CNL_E_BAD_ARG, "cnl_conv_split_weights_f32: w_buf must be 16-byte aligned");
    const long total = (long)Cout * KH * KW * Cin;
    if (w_buf != w_ohwi) {
        const hipError_t e_ = hipMemcpyAsync(w_buf, w_ohwi, (size_t)total * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream);
        if (e_ != hipSuccess) return cnl::fail(CNL_E_HIP, "cnl_conv_split_weights_f32: hipMemcpyAsync: %s", hipGetErrorString(e_));
    }
    hipLaunchKernelGGL(cnl_conv::up2_split_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, w_buf, reinterpret_cast<unsigned short*>(w_buf + total),
                       w_buf + 2 * total, total);
    return cnl::check_launch("up2_split_kernel");
}

static int up2_args(const cnl_conv_params* p, const char* who) {
    CNL_REQUIRE(p, CNL_E_BAD_ARG, "%s: null params", who);
    CNL_REQUIRE(p->x && p->w && p->bias && p->y, CNL_E_BAD_ARG, "%s: null tensor pointer", who);
    CNL_REQUIRE(p->N > 0 && p->H_in > 0 && p->W_in > 0 && p->Cin > 0 && p->Cout > 0, CNL_E_BAD_ARG, "%s: non-positive dimension", who);
    CNL_REQUIRE(p->KH == 3 && p->KW == 3 && p->stride == 1 && p->pad == 1, CNL_E_UNSUPPORTED, "%s: 3x3 / stride 1 / pad 1 only", who);
    CNL_REQUIRE(p->Cin % 32 == 0, CNL_E_UNSUPPORTED, "%s: Cin=%d is not a multiple of 32", who, p->Cin);
    CNL_REQUIRE(p->ldx >= p->Cin && p->ldy >= p->Cout && p->ldx % 4 == 0, CNL_E_BAD_ARG, "%s: pixel strides ldx=%d ldy=%d too small / misaligned",
                who, p->ldx, p->ldy);
    CNL_REQUIRE(((uintptr_t)p->x & 15) == 0 && ((uintptr_t)p->w & 15) == 0, CNL_E_BAD_ARG, "%s: x and w must be 16-byte aligned", who);
    CNL_REQUIRE((p->flags & CNL_UPSAMPLE_IN) && !(p->flags & ~(uint32_t)(CNL_RELU | CNL_RELU6 | CNL_UPSAMPLE_IN)), CNL_E_UNSUPPORTED,
                "%s: flags must be CNL_UPSAMPLE_IN (| CNL_RELU | CNL_RELU6)", who);
    CNL_REQUIRE(!p->residual, CNL_E_UNSUPPORTED, "%s: residual not supported", who);
    return CNL_OK;
}

static void up2_phase(const cnl_conv_params* p, int dy, int dx, ConvArgs& a) {
    const size_t total = (size_t)16 * p->Cin * p->Cout;          // floats of the fp32 phase weights; the fp16 split follows, then S_w
    a.x = p->x; a.w = p->w + (size_t)(dy * 2 + dx) * p->Cout * 4 * p->Cin; a.bias = p->bias; a.res = nullptr; a.y = p->y;
    a.wscale = p->w + 2 * total;
    a.N = p->N; a.Hin = p->H_in; a.Win = p->W_in; a.Cin = p->Cin; a.Cout = p->Cout;
    a.KH = a.KW = 2; a.stride = 1; a.pad = dy ? 0 : 1; a.pad_x = dx ? 0 : 1;
    a.sub_dy = dy; a.sub_dx = dx;
    a.ldx = p->ldx; a.ldy = p->ldy; a.ldr = 0;
    a.flags = (p->flags & (CNL_RELU | CNL_RELU6)) | CNL_I_SUBPIXEL;
    a.xmax = p->x_absmax; a.wmax = p->w_absmax; a.ymax = reinterpret_cast<unsigned*>(p->y_absmax); a.algo = p->algo;
    a.HL = p->H_in; a.WL = p->W_in;
    a.Ho = p->H_in; a.Wo = p->W_in;
    if (f16x2_eligible(a)) a.w += total;                          // the fp16-split kernel multiplies the pre-split copy
}

extern "C" int cnl_conv3x3_up2_kernel(const cnl_conv_params* p) {
    const int rc = up2_args(p, "cnl_conv3x3_up2_kernel");
    if (rc != CNL_OK) return rc;
    ConvArgs a;
    up2_phase(p, 0, 0, a);
    return f16x2_eligible(a) ? CNL_CONV_F16X2 : CNL_CONV_F32;
}

extern "C" int cnl_conv3x3_up2_nhwc_f32(const cnl_conv_params* p, void* stream) {
    const int rc0 = up2_args(p, "cnl_conv3x3_up2_nhwc_f32");
    if (rc0 != CNL_OK) return rc0;
    for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx) {
            ConvArgs a;
            up2_phase(p, dy, dx, a);
            const int rc = finish_and_launch(a, true, "cnl_conv3x3_up2_nhwc_f32", (hipStream_t)stream);
            if (rc != CNL_OK) return rc;
        }
    return CNL_OK;
}

/*
 * Stride-2 transposed convolution as four sub-pixel phases, each an ordinary correlation over the input written to one of the
 * four positions of the 2x output grid (no zero-stuffing, no col2im): out[2y+dy, 2x+dx] = sum over the taps ky == (dy+p) mod 2,
 * kx == (dx+p) mod 2 of in[y + (dy+p-ky)/2, x + (dx+p-kx)/2] . w[ky][kx].
 */
extern "C" int cnl_deconv_phase_geometry(int32_t K, int32_t d, int32_t* taps, int32_t* pad) {
    CNL_REQUIRE(K >= 2 && K <= 4 && (d == 0 || d == 1) && taps && pad, CNL_E_BAD_ARG, "cnl_deconv_phase_geometry: K in 2..4, d in 0..1");
    const int p = (K + K % 2) / 2 - 1;                     // make_upsample: padding = (k + output_padding)//2 - 1 (layers.py:87-88)
    int n = 0, ky_max = -1;
    for (int ky = 0; ky < K; ++ky)
        if (((ky - d - p) & 1) == 0) { ++n; ky_max = ky; }
    *taps = n;
    *pad = -((d + p - ky_max) / 2);                        // minus the smallest input offset (exact division: same parity)
    return CNL_OK;
}

extern "C" size_t cnl_deconv_weight_floats(int32_t Cin, int32_t Cout, int32_t K) {
    if (Cin <= 0 || Cout <= 0 || K < 2 || K > 4) return 0;
    return (size_t)Cin * Cout * K * K;
}

extern "C" int cnl_deconv2x_nhwc_f32(const cnl_deconv_params* p, void* stream) {
    CNL_REQUIRE(p, CNL_E_BAD_ARG, "cnl_deconv2x_nhwc_f32: null params");
    CNL_REQUIRE(p->x && p->w && p->bias && p->y, CNL_E_BAD_ARG, "cnl_deconv2x_nhwc_f32: null tensor pointer");
    CNL_REQUIRE(p->N > 0 && p->H_in > 0 && p->W_in > 0 && p->Cin > 0 && p->Cout > 0, CNL_E_BAD_ARG,
                "cnl_deconv2x_nhwc_f32: non-positive dimension");
    CNL_REQUIRE(p->K >= 2 && p->K <= 4, CNL_E_UNSUPPORTED, "cnl_deconv2x_nhwc_f32: deconv_kernel=%d (supported: 2, 3, 4)", p->K);
    CNL_REQUIRE(p->Cin % 32 == 0, CNL_E_UNSUPPORTED, "cnl_deconv2x_nhwc_f32: Cin=%d is not a multiple of 32", p->Cin);
    CNL_REQUIRE(p->ldx >= p->Cin && p->ldy >= p->Cout && p->ldx % 4 == 0, CNL_E_BAD_ARG,
                "cnl_deconv2x_nhwc_f32: pixel strides ldx=%d ldy=%d too small / misaligned", p->ldx, p->ldy);
    CNL_REQUIRE(((uintptr_t)p->x & 15) == 0 && ((uintptr_t)p->w & 15) == 0, CNL_E_BAD_ARG,
                "cnl_deconv2x_nhwc_f32: x and w must be 16-byte aligned");
    CNL_REQUIRE(!(p->flags & ~(uint32_t)(CNL_RELU | CNL_RELU6)), CNL_E_UNSUPPORTED, "cnl_deconv2x_nhwc_f32: flags other than CNL_RELU / CNL_RELU6");
    CNL_REQUIRE(!p->residual || p->ldr >= p->Cout, CNL_E_BAD_ARG, "cnl_deconv2x_nhwc_f32: ldr=%d < Cout", p->ldr);
    const float* w = p->w;
    for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx) {
            int32_t kh, kw, py, px;
            cnl_deconv_phase_geometry(p->K, dy, &kh, &py);
            cnl_deconv_phase_geometry(p->K, dx, &kw, &px);
            ConvArgs a;
            a.x = p->x; a.w = w; a.bias = p->bias; a.res = p->residual; a.y = p->y;
            a.N = p->N; a.Hin = p->H_in; a.Win = p->W_in; a.Cin = p->Cin; a.Cout = p->Cout;
            a.KH = kh; a.KW = kw; a.stride = 1; a.pad = py; a.pad_x = px;
            a.sub_dy = dy; a.sub_dx = dx;
            a.ldx = p->ldx; a.ldy = p->ldy; a.ldr = p->ldr;
            a.flags = p->flags | CNL_I_SUBPIXEL;
            a.xmax = a.wmax = a.wscale = nullptr; a.ymax = nullptr; a.algo = CNL_ALGO_AUTO;
            a.HL = p->H_in; a.WL = p->W_in;
            a.Ho = p->H_in; a.Wo = p->W_in;
            const int rc = finish_and_launch(a, true, "cnl_deconv2x_nhwc_f32", (hipStream_t)stream);
            if (rc != CNL_OK) return rc;
            w += (size_t)p->Cout * kh * kw * p->Cin;       // next phase block of the packed weights
        }
    return CNL_OK;
}
