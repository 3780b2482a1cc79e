// stem_f16x2.hip — the ResNet stem Conv2d(3,64,7,s2,p3)+BN(folded)+ReLU of stem.hip with every fp32 product formed on the fp16
// matrix cores (torchvision resnet.conv1/bn1/relu through backbone.forward_features: reference models/meta.py:42).
//
// Arithmetic: winograd5.hip's scaled two-way split.  The weights are split once (cnl_stem_pack_weights_f32: S_w from max |w|); the
// input patch of a workgroup is scaled by a power of two S_x derived from the maximum of THAT patch (an output depends on its own
// image only, and on nothing outside the tile's patch: batch-invariant by construction, no pass over the input, no hint);
// x S = hi + lo (hi = RN16, lo = RZ16 of the exact residual), hi lo' + lo hi' + hi hi' accumulated in fp32, epilogue x 1/(S_x S_w).
//
// Structure: stem.hip's LDS-side im2col (the workgroup stages the (2*16+5) x (2*32+5) x 3 patch of its 16x32 output tile by 4-byte
// LDS-DMA from whatever strides the caller has), but K is laid out in groups of 8 for v_mfma_f32_32x32x16_f16: group g = 3 ky + q
// holds t = 8q .. 8q+7 of kernel row ky (t = kx*3 + c; t >= 21 are pads: zero weights, and the A operand is masked there because
// the patch slot holds a neighbouring pixel and 0 x inf would not be 0); 21 groups + one all-zero = 11 steps of 16 instead of 77 of
// 2.  Round 4: the staged patch is split ONCE per workgroup — the scan that finds the patch maximum keeps its values in registers, and
// after the scale is known each thread writes the hi / lo fp16 pieces of its values over the fp32 patch as two planes [row][256] — so a
// lane's operand is 8 consecutive halfs of each plane (two ds_read2_b32 per piece), no VALU.  (Rounds 1-3 split the operand of every
// MFMA in registers: each patch element ~8 times, 10.6 VALU per MFMA — the loop was VALU-bound, 120 of the kernel's 216 us.)  The split
// weights sit in LDS as [piece][group][cout][8] so a lane's operand is one ds_read_b128.  Per step and wave: 4 rows x 2 cout groups x 3
// MFMAs of 32 cycles.
#include "cnl_common.h"

namespace cnl_stem5 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#ifndef S5_NW
#define S5_NW 4       /* waves per workgroup (A/B build: make variant TAG=s5nw8 EXTRA=-DS5_NW=8) */
#endif
constexpr int TH = 16, TW = 32;                       // output tile
constexpr int NW = S5_NW;                             // waves per workgroup; each owns RPW conv rows of the tile.  Round 5 measured 8 waves x 2 rows (the same
constexpr int RPW = TH / NW;                          //   instructions per tile on twice the waves, 4 per SIMD, 102-117 registers): the stem WITHOUT the pool 294 -> 250 us,
                                                      //   the stem + pool of the default plans 170 -> 175.6 (profiles/r05_experiments.txt r5i): 4 x 4 stays
static_assert(NW == 4 || NW == 8, "the pooling epilogue is written for 4 or 2 conv rows per wave");
constexpr int PR = 2 * TH + 5;                        // patch rows (37)
constexpr int PC = 2 * TW + 5;                        // patch cols (69)
constexpr int RS = 256;                               // patch row stride in floats: 207 used (+ 3 pad reads); 1 KB = 4 DMA instructions
constexpr int NG = 21;                                // K groups of 8 that hold weights (7 kernel rows x 3)
constexpr int STEPS = 11;                             // 22 groups / 2 lane halves
constexpr int W_PIECE_BYTES = NG * 64 * 16;           // 21504: one piece (hi or lo) of the split weights
constexpr int W_BYTES = 2 * W_PIECE_BYTES;            // 43008 = 42 x 1 KB
constexpr int PATCH_BYTES = PR * RS * 4;              // 37888: the fp32 patch as staged, then its two fp16 planes (hi at 0, lo at PLANE_BYTES)
constexpr int PLANE_BYTES = PR * RS * 2;              // 18944
constexpr int NT = 64 * NW;                           // threads per workgroup
constexpr int SCAN_IT = (PR * (RS / 4) + NT - 1) / NT; // float4 per thread (5; NW = 4: 10)
constexpr int LDS_BYTES = PATCH_BYTES + W_BYTES + 64; // 80960 -> 2 workgroups / CU
constexpr unsigned OOB = 0xFFFFFFF0u;

typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void dma4(const float* base, unsigned bytes, float* lds_dst, unsigned voffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_dst, 4, voffset, 0, 0, 0);
}
// one BYTE per lane, zero-extended to the lane's dword in LDS (uint8 frames)
__device__ __forceinline__ void dma1(const void* base, unsigned bytes, float* lds_dst, unsigned voffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_dst, 1, voffset, 0, 0, 0);
}
__device__ __forceinline__ void dma16(const void* base, unsigned bytes, char* lds_dst, unsigned voffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_dst, 16, voffset, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// (v0, v1) S -> hi pair (RN16, packed) and lo pair (RZ16 of the exact residuals, packed).  Plain C on purpose (the compiler folds it
// into v_fma_mixlo/mixhi_f16, v_fma_mix_f32, v_cvt_pkrtz): with inline asm the hazard recognizer does not see VALU instructions,
// and this kernel reuses an MFMA's operand registers for the next row's split a few instructions after the MFMA issues.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2(float v0, float v1, float S, unsigned& hi, unsigned& lo) {
    const _Float16 h0 = (_Float16)__builtin_fmaf(v0, S, 0.f), h1 = (_Float16)__builtin_fmaf(v1, S, 0.f);
    const float r0 = __builtin_fmaf(v0, S, -(float)h0), r1 = __builtin_fmaf(v1, S, -(float)h1);
    const f16x2 hv = {h0, h1};
    hi = __builtin_bit_cast(unsigned, hv);
    lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
}
// the power of two that puts a tensor of maximum magnitude mx into [2^13, 2^14)  (1 for 0 / Inf / NaN maxima)
__device__ __forceinline__ float pow2_scale(float mx) {
    float S = 1.f;
    if (mx > 0.f && mx < __builtin_inff()) {
        int e;
        (void)__builtin_frexpf(mx, &e);
        e = 14 - e;
        S = __builtin_ldexpf(1.f, e < -60 ? -60 : (e > 60 ? 60 : e));
    }
    return S;
}

// w_split: [piece][group][cout][8] fp16 (W_BYTES), scal[0] = S_w.
// U8: x is a uint8 image (strides in bytes) and the first thing the workgroup does with its staged patch is A.Normalize in place —
// (float(x) - mean255[c]) * inv_std255[c], two roundings, exactly cnl_normalize_u8_nhwc_f32 — inside the pass that scans the patch
// for its maximum anyway; slots outside the image stay 0 (the conv pads the NORMALISED image).  The fp32 image never exists in HBM.
struct Norm {
    float m[3], r[3];
};
// POOL: y is the output of MaxPool2d(3, s2, p1) applied to the conv output ([N, Hp, Wp, 64], its tile seams ZEROED by the caller's launch
// function): the workgroup pools its 16x32 conv tile through LDS; the 7x15 pooled cells whose 3x3 window lies inside the tile are
// stored, the border cells (whose window continues in a neighbouring tile) are merged with atomic max on the bit pattern — exact
// and order-independent, the values being post-ReLU (>= +0).  The 537 MB conv output never exists.
#ifdef S5_TRACE            // timing build: per workgroup [8] = s_memrealtime (100 MHz) at start / patch staged / planes written / K loop done / end, HW_ID, XCC_ID, -
__device__ unsigned long long* s5_trace_ptr;
#define S5_STAMP(i_) do { if (tid == 0 && s5_trace_ptr) s5_trace_ptr[(size_t)blockIdx.x * 8 + (i_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define S5_STAMP(i_) do {} while (0)
#endif
template <bool POOL, bool U8, bool FULL>
__global__ __launch_bounds__(NT, NW / 2) void stem_f16x2_kernel(const void* __restrict__ x, long sn, int sc, int sh, int sw,
                                                            unsigned x_img_bytes, const void* __restrict__ w_split,
                                                            const float* __restrict__ scal, const float* __restrict__ bias,
                                                            float* __restrict__ y, unsigned* __restrict__ ymax, int N, int H, int W, int Ho, int Wo,
                                                            int tiles_x, int tiles_y, const Norm nrm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* patch = reinterpret_cast<float*>(smem);
    char* wl = smem + PATCH_BYTES;
    float* red = reinterpret_cast<float*>(smem + PATCH_BYTES + W_BYTES);      // [NW] per-wave patch maxima

    const int tid = threadIdx.x;
    const int lane = tid & 63, hi = lane >> 5, px = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
    S5_STAMP(0);
#ifdef S5_TRACE
    if (tid == 0 && s5_trace_ptr) {
        s5_trace_ptr[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        s5_trace_ptr[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif

    // ---- staging by LDS-DMA (see stem.hip) ----
#ifndef S5_EXP
#define S5_EXP 0        // timing builds (results wrong): 1 no patch DMA, 2 no weight DMA, 3 one K step, 4 no stores, 5 no patch scan, 6 / 7 border cells of the pooled map stored plainly / not at all
#endif
    if (S5_EXP != 2)
    for (int q = wave; q < W_BYTES / 1024; q += NW) dma16(w_split, (unsigned)W_BYTES, wl + q * 1024, (unsigned)(q * 1024 + lane * 16));
    constexpr int ES = U8 ? 1 : 4;                                             // bytes per input element
    const char* xn = reinterpret_cast<const char*>(x) + (long)n * sn * ES;
    unsigned lane_off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int f = q * 64 + lane;
        const int col = f / 3, c = f - col * 3;
        const int ix = ix0 + col;
        lane_off[q] = (f < PC * 3 && (unsigned)ix < (unsigned)W) ? (unsigned)((c * sc + ix * sw) * ES) : OOB;
    }
    for (int r = wave; r < (S5_EXP == 1 ? 0 : PR); r += NW) {
        const int iy = iy0 + r;
        const bool row_ok = (unsigned)iy < (unsigned)H;                       // wave-uniform
        const unsigned row_off = (unsigned)(iy * sh * ES);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned off = (row_ok && lane_off[q] != OOB) ? lane_off[q] + row_off : OOB;
            if (U8) dma1(xn, x_img_bytes, patch + r * RS + q * 64, off);
            else dma4(reinterpret_cast<const float*>(xn), x_img_bytes, patch + r * RS + q * 64, off);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    S5_STAMP(1);

    // ---- the patch's scale: max |x| over the staged patch (out-of-image slots hold zeros); the values stay in registers ----
    float mx = 0.f;
    float4 pv[SCAN_IT];
#pragma unroll
    for (int it = 0; it < SCAN_IT; ++it) {
        const int e = tid + it * NT;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < PR * (RS / 4) && (S5_EXP != 5 || it == 0)) {
            v = *reinterpret_cast<const float4*>(patch + e * 4);
            if (U8) {                        // the slots hold zero-extended bytes: A.Normalize (in registers; the fp32 image never exists)
#pragma clang fp contract(off)               // one rounding per operation: bit-parity with cnl_normalize_u8_nhwc_f32
                const int r = e >> 6, f0 = (e & 63) * 4;                           // RS / 4 = 64 float4 per patch row
                const bool row_ok = (unsigned)(iy0 + r) < (unsigned)H;
                float o[4];
                const unsigned u[4] = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int f = f0 + i, col = f / 3, c = f - col * 3;
                    const bool ok = row_ok && f < PC * 3 && (unsigned)(ix0 + col) < (unsigned)W;
                    const float m_ = c == 0 ? nrm.m[0] : (c == 1 ? nrm.m[1] : nrm.m[2]), r_ = c == 0 ? nrm.r[0] : (c == 1 ? nrm.r[1] : nrm.r[2]);
                    o[i] = ok ? ((float)u[i] - m_) * r_ : 0.f;
                }
                v = make_float4(o[0], o[1], o[2], o[3]);
            }
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
        pv[it] = v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red[wave] = mx;
    __syncthreads();                                  // (every thread has read its part of the fp32 patch: the planes may overwrite it)
    float pmx = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) pmx = fmaxf(pmx, red[w]);
    const float Sx = pow2_scale(pmx);
    const float inv = 1.f / (Sx * scal[0]);
    // ---- x S = hi + lo, ONCE per element: two fp16 planes over the patch ----
#pragma unroll
    for (int it = 0; it < SCAN_IT; ++it) {
        const int e = tid + it * NT;
        if (e < PR * (RS / 4)) {
            unsigned h0, l0, h1, l1;
            split2(pv[it].x, pv[it].y, Sx, h0, l0);
            split2(pv[it].z, pv[it].w, Sx, h1, l1);
            *reinterpret_cast<uint2*>(smem + e * 8) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(smem + PLANE_BYTES + e * 8) = make_uint2(l0, l1);
        }
    }
    __syncthreads();
    S5_STAMP(2);

    f32x16 acc[RPW][2];
#pragma unroll
    for (int i = 0; i < RPW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // lane's A base: output row (wave*RPW + i), column px -> patch row 2*(wave*RPW+i) + ky, elements 6*px + 8q .. + 7 of each plane
    const char* pa = smem + ((wave * 2 * RPW) * RS + px * 6) * 2;
    const char* pb = wl + px * 16;                                // + piece * W_PIECE_BYTES + (g * 64 + 32 j) * 16
#pragma unroll
    for (int s = 0; s < (S5_EXP == 3 ? 1 : STEPS); ++s) {
        // group of this lane half: g = 2s + hi; the all-zero group 21 (hi = 1 of the last step) reads group 20's slots and masks all
        const int g0 = 2 * s, g1 = (2 * s + 1 < NG) ? 2 * s + 1 : 2 * s;
        const int ky0 = g0 / 3, q0 = g0 % 3, ky1 = g1 / 3, q1 = g1 % 3;
        const bool dead1 = 2 * s + 1 >= NG;
        const int a_off = hi ? (ky1 * RS + 8 * q1) : (ky0 * RS + 8 * q0);              // floats
        const int b_off = (hi ? g1 : g0) * 64 * 16;                                   // bytes
        // masks of the packed pairs (e4,e5) and (e6,e7): t = 21, 22, 23 are pads of the groups with q == 2
        const unsigned m01 = (hi && dead1) ? 0u : 0xFFFFFFFFu;
        const unsigned m2 = hi ? (dead1 ? 0u : (q1 == 2 ? 0x0000FFFFu : 0xFFFFFFFFu)) : (q0 == 2 ? 0x0000FFFFu : 0xFFFFFFFFu);
        const unsigned m3 = hi ? ((dead1 || q1 == 2) ? 0u : 0xFFFFFFFFu) : (q0 == 2 ? 0u : 0xFFFFFFFFu);
        u32x4 bh[2], bl[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bh[j] = *reinterpret_cast<const u32x4*>(pb + b_off + j * 32 * 16);
            bl[j] = *reinterpret_cast<const u32x4*>(pb + W_PIECE_BYTES + b_off + j * 32 * 16);
            if (dead1) {                                                                // compile-time: last step only
#pragma unroll
                for (int e = 0; e < 4; ++e) { bh[j][e] &= m01; bl[j][e] &= m01; }
            }
        }
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const unsigned* ph = reinterpret_cast<const unsigned*>(pa + ((2 * i) * RS + a_off) * 2);      // (4-byte aligned: 12 px + 16 q)
            const unsigned* pl = reinterpret_cast<const unsigned*>(pa + PLANE_BYTES + ((2 * i) * RS + a_off) * 2);
            unsigned h[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { h[e] = ph[e]; l[e] = pl[e]; }
            if (dead1) { h[0] &= m01; h[1] &= m01; l[0] &= m01; l[1] &= m01; }
            if (dead1 || q0 == 2 || q1 == 2) { h[2] &= m2; h[3] &= m3; l[2] &= m2; l[3] &= m3; }
            const u32x4 ah = {h[0], h[1], h[2], h[3]}, al = {l[0], l[1], l[2], l[3]};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = mfma16(ah, bl[j], acc[i][j]);
                acc[i][j] = mfma16(al, bh[j], acc[i][j]);
                acc[i][j] = mfma16(ah, bh[j], acc[i][j]);
            }
        }
    }

#ifdef S5_TRACE
    asm volatile("s_nop 0" :: "v"(acc[RPW - 1][1][15]), "v"(acc[0][0][0]));      // the stamp waits for the last MFMAs
#endif
    S5_STAMP(3);
    float omax = 0.f;                   // max of what this thread contributes to y (post-ReLU: >= 0)
    if constexpr (!POOL) {
        // epilogue: scale back, bias + ReLU, NHWC store (col = lane&31 -> channel, rows -> pixels of the row)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int co = j * 32 + px;
            const float bv = bias[co];
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                const int oy = oy0 + wave * RPW + i;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ox = ox0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (oy < Ho && ox < Wo && (S5_EXP != 4 || acc[i][j][r] == 12345.f)) {
                        const float v = fmaxf(acc[i][j][r] * inv + bv, 0.f);
                        omax = fmaxf(omax, v);
                        __builtin_nontemporal_store(v, &y[(((size_t)n * Ho + oy) * Wo + ox) * 64 + co]);
                    }
                }
            }
        }
    } else {
        // epilogue with the max-pool, in registers.  A lane holds, per conv row, the columns c0 .. c0+3 of four groups (c0 = 8g + 4hi):
        // pooled column 4g+2hi+1 = max of columns c0+1 .. c0+3 is lane-local, pooled column 4g+2hi = max(c0-1, c0, c0+1) takes c0-1
        // from the other lane half (one xor-32 shuffle per group); a wave holds conv rows 4w .. 4w+3 (RPW = 4): pooled row 2w+1 (rows 4w+1 .. 4w+3)
        // is wave-local, pooled row 2w = max(4w-1, 4w, 4w+1) takes the column-pooled row 4w-1 from the previous wave through LDS.  RPW = 2: a wave
        // holds conv rows 2w, 2w+1 and owns pooled row w = max(2w-1, 2w, 2w+1), row 2w-1 again from the previous wave.
        // Pooled row 0 / column 0 (window continues in the tile above / left) and row 8 / column 16 (only the first row / column of the
        // window is in this tile) are merged across workgroups with atomic max; invalid conv positions (outside the image) count as 0.
        float* X = reinterpret_cast<float*>(smem);                     // [j][wave][9][64 lanes]: column-pooled last conv row of the wave
        static_assert(2 * NW * 9 * 64 * 4 <= PATCH_BYTES, "the hand-over rows live in the patch region");
        const int Hp = (Ho - 1) / 2 + 1, Wp = (Wo - 1) / 2 + 1;        // MaxPool2d(3, 2, 1) output size
        const int rows_ok = Ho - oy0, cols_ok = Wo - ox0;              // conv rows / cols of the tile inside the image
        const int py0 = oy0 >> 1, px0 = ox0 >> 1;
        // column pooling of 16 values (one conv row, or the maximum of several): out[0..3] = pooled columns 4g+2hi, out[4..7] = 4g+2hi+1,
        // out[8] = column 31 alone (lane half 1; pooled column 16)
#define S5_COLPOOL(in_, out_)                                                                                     \
        do {                                                                                                      \
            float s_[4];                                                                                          \
            _Pragma("unroll") for (int g = 0; g < 4; ++g) s_[g] = __shfl_xor((in_)[4 * g + 3], 32, 64);           \
            _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                       \
                const float nb_ = hi ? s_[g] : (g ? s_[g ? g - 1 : 0] : 0.f);                                     \
                (out_)[g] = fmaxf(fmaxf((in_)[4 * g], (in_)[4 * g + 1]), nb_);                                    \
                (out_)[4 + g] = fmaxf(fmaxf((in_)[4 * g + 1], (in_)[4 * g + 2]), (in_)[4 * g + 3]);               \
            }                                                                                                     \
            (out_)[8] = hi ? (in_)[15] : 0.f;                                                                     \
        } while (0)
        __syncthreads();                                               // everyone is done with the LDS images
        // FULL (Ho % 16 == 0 and Wo % 32 == 0: every tile of the launch lies inside the conv map — 512 x 512 and 608 x 1088 inputs): no
        // validity masks, and every address is a wave-uniform base + one per-lane offset (the general path's per-value masks and 64-bit per-lane addresses made 2 500
        // instructions with ~320 SGPR spill moves of this epilogue: 7.5 us of a workgroup's 22, profiles/r04_stem_trace.txt).
        {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float bv = bias[j * 32 + px];
#pragma unroll
            for (int i = 0; i < RPW; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[i][j][r] * inv + bv;
                    const bool ok = FULL || (wave * RPW + i < rows_ok && (r & 3) + 8 * (r >> 2) + 4 * hi < cols_ok);
                    acc[i][j][r] = (ok && v > 0.f) ? v : 0.f;            // ReLU; never -0 or NaN
                    omax = fmaxf(omax, acc[i][j][r]);                  // (every valid conv value lies in some pooling window: max y = max of these)
                }
            float in3[16], p3[9];
#pragma unroll
            for (int r = 0; r < 16; ++r) in3[r] = acc[RPW - 1][j][r];
            S5_COLPOOL(in3, p3);
#pragma unroll
            for (int k = 0; k < 9; ++k) X[((j * NW + wave) * 9 + k) * 64 + lane] = p3[k];
        }
        __syncthreads();
        float* const yimg = y + (size_t)n * Hp * Wp * 64;              // wave-uniform
        const int lane_off = hi * 128 + px;                            // + 32 j: the lane's part of every cell offset (floats)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float in_o[16], in_e[16], po[9], pe[9];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if constexpr (RPW == 4) in_o[r] = fmaxf(fmaxf(acc[1][j][r], acc[2][j][r]), acc[RPW - 1][j][r]);
                in_e[r] = fmaxf(acc[0][j][r], acc[1][j][r]);
            }
            if constexpr (RPW == 4) S5_COLPOOL(in_o, po);
            S5_COLPOOL(in_e, pe);
            if (wave > 0) {
#pragma unroll
                for (int k = 0; k < 9; ++k) pe[k] = fmaxf(pe[k], X[((j * NW + wave - 1) * 9 + k) * 64 + lane]);
            }
            // rows: 0 = the pooled row that takes the previous wave's last conv row (RPW = 4: 2w, RPW = 2: w), 1 = pooled row 2w+1 (RPW = 4 only),
            // 2 = pooled row 8 (the last wave only: its conv row 15 alone)
#pragma unroll
            for (int rw = 0; rw < 3; ++rw) {
                if (rw == 1 && RPW != 4) continue;
                if (rw == 2 && wave != NW - 1) continue;
                const int pr = rw == 2 ? 8 : (RPW == 4 ? 2 * wave + rw : wave);
                const int gy = py0 + pr;
                if (gy >= Hp) continue;
                const bool row_border = pr == 0 || pr == 8;
                float* const yrow = yimg + ((size_t)gy * Wp + px0) * 64 + j * 32;          // wave-uniform
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    // pooled column of the lane: pc = 4 (k & 3) + (k >> 2) + 2 hi; k == 8: column 16 (lane half 1 only)
                    const int pcu = k == 8 ? 14 : (4 * (k & 3) + (k >> 2));                 // the wave-uniform part
                    float m;
                    if (rw == 2) m = X[((j * NW + NW - 1) * 9 + k) * 64 + lane];      // own column-pooled row 15
                    else m = (RPW == 4 && rw) ? po[k] : pe[k];
                    if (k == 8 && !hi) continue;
                    if ((!FULL || k == 8) && px0 + pcu + 2 * hi >= Wp) continue;
                    float* dst = yrow + pcu * 64 + lane_off;
                    const bool col_border = (k == 0 && !hi) || k == 8;          // pc == 0 / pc == 16
                    if (row_border || col_border) {
                        if (S5_EXP == 6) __builtin_nontemporal_store(m, dst);      // timing build: the border cells as plain stores (wrong there)
                        else if (S5_EXP == 7 || S5_EXP == 4) { }                                  // timing build: border cells not written at all
                        else if (m > 0.f) atomicMax(reinterpret_cast<unsigned*>(dst), __float_as_uint(m));
                    } else if (S5_EXP != 4 || m == 12345.f) {
                        __builtin_nontemporal_store(m, dst);
                    }
                }
            }
        }
        }
#undef S5_COLPOOL
    }
    if (ymax) {                          // max |y| of image n (hand-over to the first Winograd layer: cnl_conv_params.x_absmax)
        const float m = cnl::wave_max_nonneg(omax);          // (16 K waves, 32 floats of one cache line: see cnl::report_max)
        if (lane == 0) cnl::report_max(ymax + n * AMS, m);
    }
    S5_STAMP(4);
}

// OHWI [64][7][7][3] (BN folded) -> [piece][group][cout][8] fp16 with the power-of-two scale S_w (scal[0]); one workgroup
__global__ __launch_bounds__(256) void stem_split_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ ws, float* __restrict__ scal) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    float mx = 0.f;
    for (int e = tid; e < 64 * 147; e += 256) {
        const float v = fabsf(w[e]);
        mx = (v < __builtin_inff()) ? fmaxf(mx, v) : mx;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    const float Sw = pow2_scale(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
    if (tid == 0) scal[0] = Sw;
    for (int e = tid; e < NG * 64 * 8; e += 256) {
        const int k = e & 7, co = (e >> 3) & 63, g = e >> 9;
        const int ky = g / 3, t = 8 * (g % 3) + k;
        const float v = t < 21 ? w[co * 147 + ky * 21 + t] * Sw : 0.f;
        const _Float16 hv = (_Float16)v;                                     // round to nearest even
        const float r = v - (float)hv;                                       // exact
        unsigned lo2;
        asm("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(lo2) : "v"(r), "v"(0.f));
        ws[e] = __builtin_bit_cast(unsigned short, hv);
        ws[NG * 64 * 8 + e] = (unsigned short)(lo2 & 0xFFFFu);
    }
}

// The pooled map's cells that several workgroups merge with atomic max — pooled rows gy % 8 == 0 and columns gx % 16 == 0, the seams of the
// 8 x 16-cell tiles — start from +0; one float4 per thread.  (Rounds 1-3 zero-filled the whole map: 134 MB at C1 for 25 MB of seams.)
__global__ __launch_bounds__(256) void stem_zero_borders_kernel(float* __restrict__ y, int N, int Hp, int Wp) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const int R8 = (Hp + 7) / 8, C16 = (Wp + 15) / 16;
    const long rows = (long)N * R8 * Wp * 16;
    long cell;                                  // (n * Hp + gy) * Wp + gx
    int q;
    if (t < rows) {
        q = (int)(t & 15);
        long c = t >> 4;
        const int gx = (int)(c % Wp); c /= Wp;
        const int r = (int)(c % R8); const long n = c / R8;
        cell = (n * Hp + (long)r * 8) * Wp + gx;
    } else {
        long u = t - rows;
        if (u >= (long)N * Hp * C16 * 16) return;
        q = (int)(u & 15);
        u >>= 4;
        const int cix = (int)(u % C16); u /= C16;
        const int gy = (int)(u % Hp); const long n = u / Hp;
        cell = (n * Hp + gy) * Wp + (long)cix * 16;
    }
    *reinterpret_cast<float4*>(y + cell * 64 + q * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
}

}  // namespace cnl_stem5
using namespace cnl_stem5;

#ifdef S5_TRACE
extern "C" __attribute__((visibility("default"))) int cnl_stem5_set_trace(void* p) { return hipMemcpyToSymbol(HIP_SYMBOL(s5_trace_ptr), &p, sizeof(p)) == hipSuccess ? 0 : -1; }
#endif
// floats appended to the fp32 packed weights: the two fp16 pieces + 4 scalars
size_t cnl_stem5_extra_floats() { return (size_t)W_BYTES / 4 + 4; }

int cnl_stem5_pack(const float* w_ohwi, float* extra, void* stream) {
    hipLaunchKernelGGL(stem_split_pack_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, w_ohwi,
                       reinterpret_cast<unsigned short*>(extra), extra + W_BYTES / 4);
    return cnl::check_launch("stem_split_pack_kernel");
}

int cnl_stem5_launch(const void* x, bool u8, const float* mean255, const float* inv_std255, long sn, int sc, int sh, int sw, unsigned img_bytes,
                     const float* extra, const float* bias, float* y, float* y_absmax, int N, int H, int W, int Ho, int Wo, int tiles_x, int tiles_y,
                     unsigned blocks, bool pool, void* stream) {
    Norm nrm = {{0.f, 0.f, 0.f}, {1.f, 1.f, 1.f}};
    if (u8) {
        for (int c = 0; c < 3; ++c) { nrm.m[c] = mean255[c]; nrm.r[c] = inv_std255[c]; }
    }
    static cnl::DeviceOnce once[8];
    const void* fns[8] = {(const void*)stem_f16x2_kernel<false, false, false>, (const void*)stem_f16x2_kernel<true, false, false>,
                          (const void*)stem_f16x2_kernel<false, true, false>,  (const void*)stem_f16x2_kernel<true, true, false>,
                          (const void*)stem_f16x2_kernel<false, false, true>,  (const void*)stem_f16x2_kernel<true, false, true>,
                          (const void*)stem_f16x2_kernel<false, true, true>,   (const void*)stem_f16x2_kernel<true, true, true>};
    const bool full = Ho % TH == 0 && Wo % TW == 0;              // no partial tile in the launch
    const int which = (full ? 4 : 0) + (u8 ? 2 : 0) + (pool ? 1 : 0);
    const int rc = cnl::kernel_setup(once[which], fns[which], LDS_BYTES);
    if (rc != CNL_OK) return rc;
    if (pool) {                  // the border cells of the tiles are merged with atomic max: THEY start from +0 (every other cell is stored once)
        const int Hp = (Ho - 1) / 2 + 1, Wp = (Wo - 1) / 2 + 1;
        const long total = ((long)N * ((Hp + 7) / 8) * Wp + (long)N * Hp * ((Wp + 15) / 16)) * 16;
#ifdef S5_FULL_ZERO              // A/B build: rounds 1-3's fill of the whole map
        CNL_HIP(hipMemsetAsync(y, 0, (size_t)N * Hp * Wp * 64 * sizeof(float), (hipStream_t)stream));
        if (false)
#endif
        hipLaunchKernelGGL(stem_zero_borders_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, N, Hp, Wp);
        const int rcz = cnl::check_launch("stem_zero_borders_kernel");
        if (rcz != CNL_OK) return rcz;
    }
#define S5_LAUNCH(P_, U_, F_)                                                                                                      \
    hipLaunchKernelGGL((stem_f16x2_kernel<P_, U_, F_>), dim3(blocks), dim3(NT), LDS_BYTES, (hipStream_t)stream, x, sn, sc, sh, sw, img_bytes, \
                       (const void*)extra, extra + W_BYTES / 4, bias, y, reinterpret_cast<unsigned*>(y_absmax), N, H, W, Ho, Wo, tiles_x, tiles_y, nrm)
    switch (which) {
        case 0: S5_LAUNCH(false, false, false); break;
        case 1: S5_LAUNCH(true, false, false); break;
        case 2: S5_LAUNCH(false, true, false); break;
        case 3: S5_LAUNCH(true, true, false); break;
        case 4: S5_LAUNCH(false, false, true); break;
        case 5: S5_LAUNCH(true, false, true); break;
        case 6: S5_LAUNCH(false, true, true); break;
        default: S5_LAUNCH(true, true, true); break;
    }
#undef S5_LAUNCH
    return cnl::check_launch("stem_f16x2_kernel");
}
