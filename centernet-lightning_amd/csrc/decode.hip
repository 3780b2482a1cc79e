// decode.hip — fused CenterNet decode for gfx950 (HBM-bound integer/compare work; no MFMA here).
//
// Replaces CenterNet.decode_detections / get_topk_from_heatmap / gather_and_decode_boxes
// (reference models/centernet.py:229-304) and EmbeddingHead.gather_at_indices (models/fairmot.py:63-73):
// the reference runs max_pool2d, eq, mul, max(dim=1), topk and 5-6 gathers as separate full passes.
//
// Stage 1 (peaks_*): ONE streaming pass over the heatmap.  Per pixel: nms x nms max (separable:
//   horizontal max per row, vertical max over a register ring while sliding down a strip of rows),
//   equality mask (plateaus survive, -inf padding), value * mask, then max / first-argmax over classes.
//   Writes score (f32) + label (i32) per pixel to the workspace (8 B/pixel vs 4*C B read).
//   - channel-minor layouts (NHWC, what this library's heads emit): thread = (pixel, VEC channels),
//     16-byte coalesced loads, cross-channel-group reduce through LDS in class order;
//   - any other strides (NCHW tensors from reference-style callers): thread = pixel column, loop
//     over classes, lanes along x.
// Stage 2 (topk_kernel): one workgroup per image.  The scores are read ONCE (keys kept in LDS).  A pruning bound — the minimum over
//   the 16 waves of the ceil(k/16)-th largest per-thread maximum — has at least k elements at or above it, so only the elements >= it
//   (typically 1.5-3 k of H*W) are compacted in index order and sorted: bitonic on (key, ~index) pairs, wave shuffles below distance 64,
//   LDS above -> (score desc, index asc), the canonical order the oracle defines where torch.topk leaves ties unspecified.  When more
//   than 1024 elements pass the bound (plateaus / ties en masse) the exact k-th key comes from a radix select (12 / 10 / 10-bit digits,
//   LDS histograms) instead.  Then the label / ltrb box / embedding gathers and the box decode for the k winners only (the reference
//   transforms the whole 4xHxW map first, centernet.py:282-286).
#include "cnl_common.h"

#pragma clang fp contract(off)   // one rounding per op, like ATen: box decode must be bit-exact

namespace cnl_decode {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int VEC> struct VecT;
template <> struct VecT<4> { typedef f32x4 T; };
template <> struct VecT<2> { typedef f32x2 T; };
template <> struct VecT<1> { typedef float T; };

template <int VEC>
__device__ __forceinline__ void vload(const float* p, float (&v)[VEC]) {
    if constexpr (VEC == 1) {
        v[0] = *p;
    } else {
        const typename VecT<VEC>::T t = *reinterpret_cast<const typename VecT<VEC>::T*>(p);
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] = t[i];
    }
}

struct PeakArgs {
    const float* heat;
    long sn, sc, sh, sw;
    int N, C, H, W;
    int CG, PXB, R;          // channel groups per pixel, pixels per block, rows per strip
    int tiles_x, strips;
    float* ws_score;
    int* ws_label;
};

// ---- stage 1, channel-minor layout (sc == 1) ----
template <int VEC, int P>
__global__ __launch_bounds__(256) void peaks_cminor_kernel(const PeakArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red_v = reinterpret_cast<float*>(smem);                  // [R][PXB][CG]
    int* red_c = reinterpret_cast<int*>(red_v + a.R * a.PXB * a.CG);

    // XCD-aware tile order: each XCD gets a contiguous range of (x tile, strip) ids, so the halo rows / columns a tile shares with
    // its neighbours are re-read from that XCD's L2 instead of HBM (the hardware deals consecutive block ids round-robin over XCDs)
    int b = (int)cnl::xcd_remap(blockIdx.x, gridDim.x);
    const int bx = b % a.tiles_x; b /= a.tiles_x;
    const int by = b % a.strips;
    const int n = b / a.strips;
    const int tid = threadIdx.x;
    const int px = tid / a.CG, g = tid - px * a.CG;
    const int x = bx * a.PXB + px;
    const int y0 = by * a.R;
    const bool active = px < a.PXB && x < a.W;
    const float NINF = -__builtin_inff();

    if (active) {
        const float* base = a.heat + (long)n * a.sn + (long)g * VEC;
        float hm[2 * P + 1][VEC];     // horizontal maxima of the last 2P+1 rows
        float ct[P + 1][VEC];         // centre values of the last P+1 rows
#pragma unroll
        for (int i = 0; i < 2 * P + 1; ++i)
#pragma unroll
            for (int v = 0; v < VEC; ++v) hm[i][v] = NINF;
#pragma unroll
        for (int i = 0; i < P + 1; ++i)
#pragma unroll
            for (int v = 0; v < VEC; ++v) ct[i][v] = NINF;

        const int y_end = min(y0 + a.R, a.H);
        // Small class counts (round 6): the strip's 8 + 2P rows are REQUESTED AT ONCE — (8 + 2P)(2P + 1) unconditional loads from clamped coordinates, all in flight —
        // and reduced afterwards, the pixels outside the image turned into -inf then.  (Row by row, every step was one memory round trip behind a per-lane branch:
        // ten dependent trips per strip, 13.8-14.8 us for the 10.6 MB of C4's maps.)
        constexpr int CM_R = 8;                      // = a.R on this path (cnl_decode_f32)
        if constexpr ((2 * P + 1) * (CM_R + 2 * P) * VEC <= 128) {
            float t[CM_R + 2 * P][2 * P + 1][VEC];
#pragma unroll
            for (int r = 0; r < CM_R + 2 * P; ++r) {
                const float* row = base + (long)min(max(y0 - P + r, 0), a.H - 1) * a.sh;
#pragma unroll
                for (int d = 0; d < 2 * P + 1; ++d) vload<VEC>(row + (long)min(max(x - P + d, 0), a.W - 1) * a.sw, t[r][d]);
            }
            bool col_ok[2 * P + 1];
#pragma unroll
            for (int d = 0; d < 2 * P + 1; ++d) col_ok[d] = (unsigned)(x - P + d) < (unsigned)a.W;
#pragma unroll
            for (int r = 0; r < CM_R + 2 * P; ++r) {
                const int yy = y0 - P + r;
#pragma unroll
                for (int i = 0; i < 2 * P; ++i)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) hm[i][v] = hm[i + 1][v];
#pragma unroll
                for (int i = 0; i < P; ++i)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) ct[i][v] = ct[i + 1][v];
                const bool row_ok = (unsigned)yy < (unsigned)a.H && yy < y_end + P;
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    float h = NINF;
#pragma unroll
                    for (int d = 0; d < 2 * P + 1; ++d) h = fmaxf(h, row_ok && col_ok[d] ? t[r][d][v] : NINF);
                    hm[2 * P][v] = h;
                    ct[P][v] = row_ok ? t[r][P][v] : NINF;      // (x itself is inside the image: the thread is active)
                }
                const int yo = yy - P;
                if (yo >= y0 && yo < y_end) {
                    float bv = 0.f;
                    int bc = 0;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        float m = hm[0][v];
#pragma unroll
                        for (int i = 1; i < 2 * P + 1; ++i) m = fmaxf(m, hm[i][v]);
                        const float cv = ct[0][v];
                        const float val = cv * (cv == m ? 1.0f : 0.0f);      // heatmap * nms_mask
                        if (v == 0 || val > bv) { bv = val; bc = g * VEC + v; }
                    }
                    const int o = ((yo - y0) * a.PXB + px) * a.CG + g;
                    red_v[o] = bv;
                    red_c[o] = bc;
                }
            }
        } else {
            for (int yy = y0 - P; yy < y_end + P; ++yy) {
                // shift the rings
#pragma unroll
                for (int i = 0; i < 2 * P; ++i)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) hm[i][v] = hm[i + 1][v];
#pragma unroll
                for (int i = 0; i < P; ++i)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) ct[i][v] = ct[i + 1][v];
                float h[VEC], c[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) { h[v] = NINF; c[v] = NINF; }
                if ((unsigned)yy < (unsigned)a.H) {
                    const float* row = base + (long)yy * a.sh;
#pragma unroll
                    for (int dx = -P; dx <= P; ++dx) {
                        const int xx = x + dx;
                        if ((unsigned)xx < (unsigned)a.W) {
                            float t[VEC];
                            vload<VEC>(row + (long)xx * a.sw, t);
#pragma unroll
                            for (int v = 0; v < VEC; ++v) {
                                h[v] = fmaxf(h[v], t[v]);
                                if (dx == 0) c[v] = t[v];
                            }
                        }
                    }
                }
#pragma unroll
                for (int v = 0; v < VEC; ++v) { hm[2 * P][v] = h[v]; ct[P][v] = c[v]; }
                const int yo = yy - P;                    // row whose (2P+1)-window is now complete
                if (yo >= y0) {
                    float bv = 0.f;
                    int bc = 0;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        float m = hm[0][v];
#pragma unroll
                        for (int i = 1; i < 2 * P + 1; ++i) m = fmaxf(m, hm[i][v]);
                        const float cv = ct[0][v];
                        const float val = cv * (cv == m ? 1.0f : 0.0f);      // heatmap * nms_mask
                        if (v == 0 || val > bv) { bv = val; bc = g * VEC + v; }
                    }
                    const int o = ((yo - y0) * a.PXB + px) * a.CG + g;
                    red_v[o] = bv;
                    red_c[o] = bc;
                }
            }
        }
    }
    __syncthreads();
    // cross-group reduce in class order (strict '>' keeps the first maximal class: torch.max(dim=1))
    for (int t = tid; t < a.R * a.PXB; t += 256) {
        const int r = t / a.PXB, p = t - r * a.PXB;
        const int xo = bx * a.PXB + p, yo = y0 + r;
        if (xo >= a.W || yo >= a.H) continue;
        const int o = (r * a.PXB + p) * a.CG;
        float bv = red_v[o];
        int bc = red_c[o];
        for (int gg = 1; gg < a.CG; ++gg) {
            const float v = red_v[o + gg];
            if (v > bv) { bv = v; bc = red_c[o + gg]; }
        }
        const long q = (long)n * a.H * a.W + (long)yo * a.W + xo;
        a.ws_score[q] = bv;
        a.ws_label[q] = bc;
    }
}

// ---- stage 1, channel-minor layout with C % 8 == 0 (the 80-class heads): thread = (run of RP = 4 or 2 pixels, 8 channels) ----
// A thread loads the RP + 2P pixels its run needs per row as 2 x 16 bytes each (a pixel's C floats are contiguous: the 10 lanes of a pixel
// read 320 contiguous bytes), so every input vector is loaded 1.5 x (RP = 4, P = 1; 2 x at RP = 2: out of L1) instead of 3 x.  The maximum / first arg-max over the channel groups is an LDS atomic max of (score key, ~class) pairs —
// 8 bytes per output pixel instead of a [pixel][group] array, no second reduction pass.
struct Peak8Args {
    PeakArgs p;
    int CG8, RUNS, TW;       // channel groups of 8 per pixel, pixel runs per block, block width in pixels (RUNS * RP)
};

__device__ __forceinline__ unsigned score_key_fwd(float f) {
    unsigned u = __float_as_uint(f);
    if (u == 0x80000000u) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float score_key_inv(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// Block = 160 threads = 16 runs x 10 channel groups at C = 80 (64 pixels wide); strips of R = 16 rows: measured best at 32 x 128 x 128 x 80
// (back-to-back decode, us: R 8 / 12 / 16 / 24 / 32 = 52 / 62 / 45 / 57 / 68; 80 / 160 / 320 threads = 46.7 / 45.0 / 46.0;
// profiles/r02_decode_variants.txt; with two-pixel runs, round 6: R 8 / 16 / 32 = 30.2 / 28.5 / 33.8 us for stage 1 alone); R = 4 when 16-row strips would leave most CUs
// without a block (small batches).
constexpr int PK8_THREADS = 160;

template <int P, int PK8_R, int RP = 4>      // RP: pixels per run (4; 2 in the P = 1 form: half the registers, twice the waves — the launcher's choice)
__global__ __launch_bounds__(PK8_THREADS) void peaks_c8_kernel(const Peak8Args q) {
    const PeakArgs& a = q.p;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* red = reinterpret_cast<unsigned long long*>(smem);       // [PK8_R][TW]
    int b = (int)cnl::xcd_remap(blockIdx.x, gridDim.x);
    const int bx = b % a.tiles_x; b /= a.tiles_x;
    const int by = b % a.strips;
    const int n = b / a.strips;
    const int tid = threadIdx.x;
    const int run = tid / q.CG8, g = tid - run * q.CG8;
    const int x0 = bx * q.TW + run * RP;
    const int y0 = by * PK8_R;
    const bool active = run < q.RUNS && x0 < a.W;
    const float NINF = -__builtin_inff();
    for (int i = tid; i < PK8_R * q.TW; i += PK8_THREADS) red[i] = 0ull;
    __syncthreads();

    if (active) {
        const float* base = a.heat + (long)n * a.sn + (long)g * 8;
        float hm[2 * P + 1][RP][8];    // horizontal maxima of the last 2P+1 rows
        float ct[P + 1][RP][8];        // centre values of the last P+1 rows
#pragma unroll
        for (int i = 0; i < 2 * P + 1; ++i)
#pragma unroll
            for (int p = 0; p < RP; ++p)
#pragma unroll
                for (int v = 0; v < 8; ++v) hm[i][p][v] = NINF;
#pragma unroll
        for (int i = 0; i < P + 1; ++i)
#pragma unroll
            for (int p = 0; p < RP; ++p)
#pragma unroll
                for (int v = 0; v < 8; ++v) ct[i][p][v] = NINF;
        const int y_end = min(y0 + PK8_R, a.H);
        // Even strips walk down, odd strips walk up: two neighbouring strips then read the two rows they share (each other's halo) at the
        // same moment — the start for one pair of neighbours, the end for the other — and the second read hits L2 instead of fetching the
        // row again (the ring is symmetric in the row order).
        const int dir = (by & 1) ? -1 : 1;
        const int rows = y_end - y0 + 2 * P;
        if constexpr (P == 1) {
            // Round 6: no register shuffling.  The three rows of horizontal maxima are a ring written at step % 3 (their maximum does not care about the order), and
            // the centre values of the previous row are simply the previous step's loads: two load buffers used alternately.  Six steps unrolled make every index
            // static (80 v_mov of 340 VALU instructions per step gone); the class of the maximum is found after the fact (v_max3 tree + 7 compare / select
            // pairs instead of 8 x compare + two selects).
            float H[3][RP][8], T[2][RP + 2][8];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int p = 0; p < RP; ++p)
#pragma unroll
                    for (int v = 0; v < 8; ++v) H[i][p][v] = NINF;
#pragma unroll
            for (int j = 0; j < RP + 2; ++j)
#pragma unroll
                for (int v = 0; v < 8; ++v) T[1][j][v] = NINF;
            auto one = [&](const int step, float (&tc)[RP + 2][8], const float (&tp)[RP + 2][8], float (&hn)[RP][8]) __attribute__((always_inline)) {
                const int yy = dir > 0 ? y0 - 1 + step : y_end - step;
                // The loads are UNCONDITIONAL, from clamped coordinates, and the pixels outside the image become -inf afterwards, behind a wave-uniform branch only
                // the waves at an image border take: a load under a per-lane branch with a -inf fill on the other side makes the compiler wait for it at the
                // join (s_waitcnt vmcnt(1) after every pair: two loads in flight instead of twelve).
                const bool row_ok = (unsigned)yy < (unsigned)a.H;
                const float* row = base + (long)min(max(yy, 0), a.H - 1) * a.sh;
#pragma unroll
                for (int j = 0; j < RP + 2; ++j) {
                    const int xx = min(max(x0 - 1 + j, 0), a.W - 1);
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(row + (long)xx * a.sw);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(row + (long)xx * a.sw + 4);
#pragma unroll
                    for (int v = 0; v < 4; ++v) { tc[j][v] = lo[v]; tc[j][4 + v] = hi[v]; }
                }
                if (__ballot(!(row_ok && x0 >= 1 && x0 + RP + 1 <= a.W)) != 0ull) {
#pragma unroll
                    for (int j = 0; j < RP + 2; ++j) {
                        const bool ok = row_ok && (unsigned)(x0 - 1 + j) < (unsigned)a.W;
#pragma unroll
                        for (int v = 0; v < 8; ++v) tc[j][v] = ok ? tc[j][v] : NINF;
                    }
                }
#pragma unroll
                for (int p = 0; p < RP; ++p)
#pragma unroll
                    for (int v = 0; v < 8; ++v) hn[p][v] = fmaxf(fmaxf(tc[p][v], tc[p + 1][v]), tc[p + 2][v]);
                const int yo = yy - dir;                  // row whose 3 x 3 windows are now complete; its centre values: the previous step's loads
                if (step >= 2) {
#pragma unroll
                    for (int p = 0; p < RP; ++p) {
                        float val[8];
#pragma unroll
                        for (int v = 0; v < 8; ++v) {
                            const float m = fmaxf(fmaxf(H[0][p][v], H[1][p][v]), H[2][p][v]);
                            const float cv = tp[p + 1][v];
                            val[v] = cv * (cv == m ? 1.0f : 0.0f);      // heatmap * nms_mask
                        }
                        // maximum over the 8 classes and the FIRST class that holds it (a strict '>' scan in class order, as torch.max(dim=1))
                        float bv = val[0];
                        int bc = 0;
#pragma unroll
                        for (int v = 1; v < 8; ++v)
                            if (val[v] > bv) { bv = val[v]; bc = v; }
                        if (x0 + p < a.W)            // larger key wins; equal keys: the smaller class
                            atomicMax(&red[(yo - y0) * q.TW + run * RP + p],
                                      ((unsigned long long)score_key_fwd(bv) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)(g * 8 + bc)));
                    }
                }
            };
            for (int step = 0; step < rows; step += 6) {
                one(step, T[0], T[1], H[0]);
                if (step + 1 < rows) one(step + 1, T[1], T[0], H[1]);
                if (step + 2 < rows) one(step + 2, T[0], T[1], H[2]);
                if (step + 3 < rows) one(step + 3, T[1], T[0], H[0]);
                if (step + 4 < rows) one(step + 4, T[0], T[1], H[1]);
                if (step + 5 < rows) one(step + 5, T[1], T[0], H[2]);
            }
        } else {
            for (int step = 0; step < rows; ++step) {
                const int yy = dir > 0 ? y0 - P + step : y_end + P - 1 - step;
#pragma unroll
                for (int i = 0; i < 2 * P; ++i)
#pragma unroll
                    for (int p = 0; p < RP; ++p)
#pragma unroll
                        for (int v = 0; v < 8; ++v) hm[i][p][v] = hm[i + 1][p][v];
#pragma unroll
                for (int i = 0; i < P; ++i)
#pragma unroll
                    for (int p = 0; p < RP; ++p)
#pragma unroll
                        for (int v = 0; v < 8; ++v) ct[i][p][v] = ct[i + 1][p][v];
                float t[RP + 2 * P][8];
                const bool row_ok = (unsigned)yy < (unsigned)a.H;
                const float* row = base + (long)min(max(yy, 0), a.H - 1) * a.sh;      // unconditional loads from clamped coordinates, as in the 3 x 3 form
#pragma unroll
                for (int j = 0; j < RP + 2 * P; ++j) {
                    const int xx = min(max(x0 - P + j, 0), a.W - 1);
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(row + (long)xx * a.sw);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(row + (long)xx * a.sw + 4);
#pragma unroll
                    for (int v = 0; v < 4; ++v) { t[j][v] = lo[v]; t[j][4 + v] = hi[v]; }
                }
                if (__ballot(!(row_ok && x0 >= P && x0 + RP + P <= a.W)) != 0ull) {
#pragma unroll
                    for (int j = 0; j < RP + 2 * P; ++j) {
                        const bool ok = row_ok && (unsigned)(x0 - P + j) < (unsigned)a.W;
#pragma unroll
                        for (int v = 0; v < 8; ++v) t[j][v] = ok ? t[j][v] : NINF;
                    }
                }
#pragma unroll
                for (int p = 0; p < RP; ++p)
#pragma unroll
                    for (int v = 0; v < 8; ++v) {
                        float h = t[p][v];
#pragma unroll
                        for (int d = 1; d <= 2 * P; ++d) h = fmaxf(h, t[p + d][v]);
                        hm[2 * P][p][v] = h;
                        ct[P][p][v] = t[p + P][v];
                    }
                const int yo = yy - dir * P;              // row whose (2P+1)-window is now complete
                if (step >= 2 * P) {
#pragma unroll
                    for (int p = 0; p < RP; ++p) {
                        float bv = 0.f;
                        int bc = 0;
#pragma unroll
                        for (int v = 0; v < 8; ++v) {
                            float m = hm[0][p][v];
#pragma unroll
                            for (int i = 1; i < 2 * P + 1; ++i) m = fmaxf(m, hm[i][p][v]);
                            const float cv = ct[0][p][v];
                            const float val = cv * (cv == m ? 1.0f : 0.0f);      // heatmap * nms_mask
                            if (v == 0 || val > bv) { bv = val; bc = g * 8 + v; }
                        }
                        if (x0 + p < a.W)            // larger key wins; equal keys: the smaller class (torch.max(dim=1) keeps the first)
                            atomicMax(&red[(yo - y0) * q.TW + run * RP + p],
                                      ((unsigned long long)score_key_fwd(bv) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)bc));
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int t = tid; t < PK8_R * q.TW; t += PK8_THREADS) {
        const int r = t / q.TW, p = t - r * q.TW;
        const int xo = bx * q.TW + p, yo = y0 + r;
        if (xo >= a.W || yo >= a.H) continue;
        const unsigned long long c = red[t];
        const long o = (long)n * a.H * a.W + (long)yo * a.W + xo;
        a.ws_score[o] = score_key_inv((unsigned)(c >> 32));
        a.ws_label[o] = (int)(0xFFFFFFFFu - (unsigned)(c & 0xFFFFFFFFull));
    }
}

// ---- stage 1, generic strides (lanes along x, loop over classes) ----
template <int P, int R>
__global__ __launch_bounds__(256) void peaks_generic_kernel(const PeakArgs a) {
    int b = (int)cnl::xcd_remap(blockIdx.x, gridDim.x);
    const int bx = b % a.tiles_x; b /= a.tiles_x;
    const int by = b % a.strips;
    const int n = b / a.strips;
    const int x = bx * 64 + (threadIdx.x & 63);
    const int y0 = (by * 4 + (threadIdx.x >> 6)) * R;
    if (x >= a.W || y0 >= a.H) return;
    const float NINF = -__builtin_inff();
    float best[R];
    int bcls[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { best[r] = 0.f; bcls[r] = 0; }
    const int y_end = min(y0 + R, a.H);
    for (int c = 0; c < a.C; ++c) {
        const float* base = a.heat + (long)n * a.sn + (long)c * a.sc;
        float hm[2 * P + 1], ct[P + 1];
#pragma unroll
        for (int i = 0; i < 2 * P + 1; ++i) hm[i] = NINF;
#pragma unroll
        for (int i = 0; i < P + 1; ++i) ct[i] = NINF;
#pragma unroll
        for (int s = 0; s < R + 2 * P; ++s) {
            const int yy = y0 - P + s;
#pragma unroll
            for (int i = 0; i < 2 * P; ++i) hm[i] = hm[i + 1];
#pragma unroll
            for (int i = 0; i < P; ++i) ct[i] = ct[i + 1];
            float h = NINF, cv = NINF;
            if ((unsigned)yy < (unsigned)a.H && yy < y_end + P) {
                const float* row = base + (long)yy * a.sh;
#pragma unroll
                for (int dx = -P; dx <= P; ++dx) {
                    const int xx = x + dx;
                    if ((unsigned)xx < (unsigned)a.W) {
                        const float t = row[(long)xx * a.sw];
                        h = fmaxf(h, t);
                        if (dx == 0) cv = t;
                    }
                }
            }
            hm[2 * P] = h;
            ct[P] = cv;
            if (s >= 2 * P) {
                const int r = s - 2 * P;                 // output row y0 + r (compile-time index)
                float m = hm[0];
#pragma unroll
                for (int i = 1; i < 2 * P + 1; ++i) m = fmaxf(m, hm[i]);
                const float c0 = ct[0];
                const float val = c0 * (c0 == m ? 1.0f : 0.0f);
                if (c == 0 || val > best[r]) { best[r] = val; bcls[r] = c; }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int yo = y0 + r;
        if (yo < a.H) {
            const long q = (long)n * a.H * a.W + (long)yo * a.W + x;
            a.ws_score[q] = best[r];
            a.ws_label[q] = bcls[r];
        }
    }
}

// ---- stage 1, class planes: contiguous rows (sw == 1) of W % 4 == 0 pixels — NCHW tensors, the reference's own layout — and the 3 x 3 pool (round 6) ----
// The generic kernel above walks class by class, row by row, three 4-byte loads per step behind per-lane branches: 410 us for C1's 168 MB (0.4 TB/s), 14 x the
// channel-minor kernel.  Here a 256-thread workgroup takes a strip of PP_R rows x 64 pixels for ALL classes: thread = (run of 4 pixels, class group cg of 16), classes
// cg, cg + 16, ... (fewer than 16 classes: 8 or 4 groups and strips of 128 / 256 pixels); per class the (PP_R + 2) x (16-byte run + left + right neighbour) loads of the strip are issued AT ONCE, unconditionally, from clamped coordinates
// (the pixels outside the image become -inf afterwards), then reduced; the running (maximum, first class) per pixel stays in registers over the thread's classes (strict
// '>' in ascending class order) and the 16 class groups meet in an LDS atomic max of (score key, ~class) pairs, as in peaks_c8_kernel.  Pools other than 3 x 3 (P = 0, 2, 3:
// 209 / 653 / 914 us on the generic kernel) are instantiations of the same code.
constexpr int PLANES_MIN_C = 4;          // with 2 classes half of the 4 class groups idle: 20.3 us for C4's maps, the generic kernel 18.6
template <int P, int PP_R>               // pool (2P + 1)^2; strips of PP_R rows (8 for P <= 1, 4 above: the strip's (PP_R + 2P) x (4 + 2P) values of a class live in registers)
__global__ __launch_bounds__(256) void peaks_planes_kernel(const PeakArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* red = reinterpret_cast<unsigned long long*>(smem);       // [PP_R][PXB]
    const int RUNS = a.PXB >> 2, CGN = a.CG;             // runs of 4 pixels x class groups = 256 threads (16 x 16 from 16 classes up; 32 x 8, 64 x 4 for fewer)
    int b = (int)cnl::xcd_remap(blockIdx.x, gridDim.x);
    const int bx = b % a.tiles_x; b /= a.tiles_x;
    const int by = b % a.strips;
    const int n = b / a.strips;
    const int tid = threadIdx.x;
    const int run = tid % RUNS, cg = tid / RUNS;
    const int x0 = bx * a.PXB + run * 4;
    const int y0 = by * PP_R;
    const float NINF = -__builtin_inff();
    for (int i = tid; i < PP_R * a.PXB; i += 256) red[i] = 0ull;
    __syncthreads();
    if (x0 < a.W && cg < a.C) {
        float best[PP_R][4];
        int bcls[PP_R][4];
        for (int c = cg; c < a.C; c += CGN) {
            const float* base = a.heat + (long)n * a.sn + (long)c * a.sc;
            float t[PP_R + 2 * P][4 + 2 * P];
#pragma unroll
            for (int r = 0; r < PP_R + 2 * P; ++r) {
                const float* row = base + (long)min(max(y0 - P + r, 0), a.H - 1) * a.sh;
                const f32x4 v = *reinterpret_cast<const f32x4*>(row + x0);
#pragma unroll
                for (int j = 0; j < 4; ++j) t[r][P + j] = v[j];
#pragma unroll
                for (int d = 1; d <= P; ++d) {
                    t[r][P - d] = row[max(x0 - d, 0)];
                    t[r][P + 3 + d] = row[min(x0 + 3 + d, a.W - 1)];
                }
            }
#pragma unroll
            for (int r = 0; r < PP_R + 2 * P; ++r) {
                const bool row_ok = (unsigned)(y0 - P + r) < (unsigned)a.H;          // (block-uniform)
#pragma unroll
                for (int d = 1; d <= P; ++d) {
                    t[r][P - d] = row_ok && x0 - d >= 0 ? t[r][P - d] : NINF;
                    t[r][P + 3 + d] = row_ok && x0 + 3 + d < a.W ? t[r][P + 3 + d] : NINF;
                }
                if (!row_ok) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) t[r][P + j] = NINF;
                }
            }
            float h[PP_R + 2 * P][4];
#pragma unroll
            for (int r = 0; r < PP_R + 2 * P; ++r)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float m = t[r][p];
#pragma unroll
                    for (int j = 1; j <= 2 * P; ++j) m = fmaxf(m, t[r][p + j]);
                    h[r][p] = m;
                }
#pragma unroll
            for (int r = 0; r < PP_R; ++r)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float m = h[r][p];
#pragma unroll
                    for (int i = 1; i <= 2 * P; ++i) m = fmaxf(m, h[r + i][p]);
                    const float cv = t[r + P][p + P];
                    const float val = cv * (cv == m ? 1.0f : 0.0f);      // heatmap * nms_mask
                    if (c == cg || val > best[r][p]) { best[r][p] = val; bcls[r][p] = c; }
                }
        }
#pragma unroll
        for (int r = 0; r < PP_R; ++r)
            if (y0 + r < a.H) {
#pragma unroll
                for (int p = 0; p < 4; ++p)       // larger key wins; equal keys: the smaller class (torch.max(dim=1) keeps the first)
                    atomicMax(&red[r * a.PXB + run * 4 + p], ((unsigned long long)score_key_fwd(best[r][p]) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)bcls[r][p]));
            }
    }
    __syncthreads();
    for (int t = tid; t < PP_R * a.PXB; t += 256) {
        const int r = t / a.PXB, p = t - r * a.PXB;
        const int xo = bx * a.PXB + p, yo = y0 + r;
        if (xo >= a.W || yo >= a.H) continue;
        const unsigned long long c = red[t];
        const long o = (long)n * a.H * a.W + (long)yo * a.W + xo;
        a.ws_score[o] = score_key_inv((unsigned)(c >> 32));
        a.ws_label[o] = (int)(0xFFFFFFFFu - (unsigned)(c & 0xFFFFFFFFull));
    }
}

// ---- box decode shared by the fused path and the standalone gather (centernet.py:278-303) ----
__device__ __forceinline__ void decode_box(const float* bp, long bsc, int xi, int yi, int W, int H, int normalize, int box_log,
                                           float mult, float stride, float* bo) {
    const float cx = (float)xi + 0.5f, cy = (float)yi + 0.5f;
    float g[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v = bp[(long)j * bsc];
        if (box_log) v = expf(v);
        v = v * mult;
        g[j] = fmaxf(v, 0.f);
    }
    float x1 = cx - g[0], y1 = cy - g[1], x2 = cx + g[2], y2 = cy + g[3];
    if (normalize) {
        const float fw = (float)W, fh = (float)H;
        x1 = x1 / fw; x2 = x2 / fw; y1 = y1 / fh; y2 = y2 / fh;
    } else {
        x1 *= stride; y1 *= stride; x2 *= stride; y2 *= stride;
    }
    bo[0] = x1; bo[1] = y1; bo[2] = x2; bo[3] = y2;
}


// ---- stage 2: per-image top-k + gathers ----
struct TopkArgs {
    const float* ws_score;
    const int* ws_label;
    const float* box; long bsn, bsc, bsh, bsw;
    const float* reid; long rsn, rsc, rsh, rsw;
    int HW, W, H, E, k, KP;       // KP = next pow2 >= k
    int keys_in_lds;              // HW * 4 bytes of dynamic LDS hold the image's score keys (read from memory ONCE)
    int keys16_in_lds;            // ... or, where those do not fit, HW * 2 bytes hold their upper halves (the compaction's prefilter)
    int normalize, box_log;
    float mult, stride;
    float* scores; long long* indices; long long* labels; float* boxes; float* emb;
};

// order-preserving float -> uint key (larger float => larger key); -0.0 is folded onto +0.0 so the
// ordering matches float comparison semantics (torch treats them as equal)
__device__ __forceinline__ unsigned score_key(float f) {
    unsigned u = __float_as_uint(f);
    if (u == 0x80000000u) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// inclusive prefix sum over the 64 lanes of a wave, six DPP adds: within the rows of 16 (row_shr 1 / 2 / 4 / 8, zeros shifted in), then row 0 -> 1 and 2 -> 3
// (row_bcast:15), then rows 0 + 1 -> 2, 3 (row_bcast:31)
__device__ __forceinline__ unsigned wave_incl_scan(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return v;
}

constexpr int TK_THREADS = 1024;
#ifdef TK_TIMING
__device__ unsigned long long tk_stamps[64 * 16];
#define TK_STAMP(slot_) do { if (tid == 0 && blockIdx.x < 64) tk_stamps[blockIdx.x * 16 + (slot_)] = __builtin_readcyclecounter(); } while (0)
#else
#define TK_STAMP(slot_) do {} while (0)
#endif

// wave-level helpers (64-wide wavefront)
__device__ __forceinline__ unsigned wave_incl_scan(unsigned v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    return v;
}

// Descending bitonic sort of cand[0 .. S) (S a power of two <= TK_THREADS), one element per thread: exchanges at distances below 64 are
// wave shuffles, only the distances >= 64 go through LDS (S = 256: 36 steps, 3 of them with barriers).  Every thread of the block calls it.
__device__ __forceinline__ void block_sort_desc(unsigned long long* cand, int S, int tid) {
    unsigned long long e = tid < S ? cand[tid] : 0ull;
    for (int size = 2; size <= S; size <<= 1) {
        for (int st = size >> 1; st > 0; st >>= 1) {
            unsigned long long p;
            if (st >= 64) {
                __syncthreads();                       // the previous exchange's readers are done
                if (tid < S) cand[tid] = e;
                __syncthreads();
                p = tid < S ? cand[tid ^ st] : 0ull;
            } else {
                if ((tid & ~63) >= S) continue;        // a wave with no element: nothing to exchange (it still joins the barriers above)
                p = __shfl_xor(e, st);
            }
            const bool take_max = ((tid & st) == 0) == ((tid & size) == 0);
            e = take_max ? (e > p ? e : p) : (e < p ? e : p);
        }
    }
    __syncthreads();
    if (tid < S) cand[tid] = e;
    __syncthreads();
}

template <bool R48>      // R48: the instantiation for maps whose keys a thread keeps in 48 registers (below); the other one carries none of that
__global__ __launch_bounds__(TK_THREADS) void topk_kernel(const TopkArgs a) {
    extern __shared__ unsigned lds_keys[];          // [HW] when a.keys_in_lds
    __shared__ __align__(16) unsigned hist[4096];
    __shared__ unsigned wave_tot[TK_THREADS / 64], wave_max[TK_THREADS / 64];
    __shared__ unsigned long long cand[1024 + 8];
    __shared__ unsigned sh_prefix, sh_need, sh_count;

    const int n = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const float* sc = a.ws_score + (long)n * a.HW;
    const int KCH = (a.HW + TK_THREADS - 1) / TK_THREADS;      // indices per thread

    // --- phase A: the keys go to LDS (when they fit: the only pass over memory) and every thread keeps the maximum of the keys it read.
    // PRUNING BOUND: the m-th largest of a wave's 64 thread maxima, m = ceil(k / 16), has m elements at or above it; the minimum of that
    // over the 16 waves, T_lo, therefore has >= 16 m >= k elements at or above it, so T_lo <= the k-th largest key and every winner (and
    // every tie at the k-th key) is among the elements >= T_lo — typically 1.5-3 k of them instead of H*W.
    TK_STAMP(0);
    constexpr int FAST_CAP = 1024;                           // candidates the rank step takes (O(n^2 / threads)); more -> radix select
    unsigned long long* win = reinterpret_cast<unsigned long long*>(hist);      // [k] winners of the fast path (hist is unused there)
    for (int i = tid; i < FAST_CAP + 8; i += TK_THREADS) cand[i] = 0ull;        // zero padding: never greater than a candidate
    hist[tid] = 0u;                                          // the bins of the bound's second step (below)
    if (tid == 0) sh_count = 0u;
    TK_STAMP(11);
    unsigned tmax = 0u;
    // Maps of at most 16 x 1024 pixels (128 x 128): the thread's 16 keys (indices tid + 1024 j) simply stay in registers — no LDS copy, no
    // index arithmetic; the fallback re-reads the scores from memory.  Larger maps: keys in LDS ([owner thread][odd pitch]) when they fit.
    const bool in_regs = !R48 && KCH <= 16;                 // (the launcher picks the R48 instantiation for 16 < KCH <= 48 only: it carries no kreg)
    const bool klds = a.keys_in_lds && !in_regs;
    // Maps whose 32-bit keys do not fit the LDS budget (the 152 x 272 maps of 608 x 1088 frames: 165 KB) keep the UPPER HALVES of the keys there (round 6):
    // key >> 16 >= bound >> 16 is a superset test, so the second pass over the scores — six rounds of dependent loads, 12 of the kernel's 17 K compaction
    // cycles at that size — becomes 41 two-byte LDS reads per thread; the few elements that pass are re-read and tested exactly.
    const bool k16 = !R48 && a.keys16_in_lds && !in_regs && !klds;      // (maps with H * W % 4 == 0 of up to 48 K pixels run the R48 instantiation: keys in registers, below)
    unsigned short* lds_k16 = reinterpret_cast<unsigned short*>(lds_keys);
    unsigned kreg[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) kreg[j] = 0u;
    if (in_regs) {
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = tid + j * TK_THREADS < a.HW ? sc[tid + j * TK_THREADS] : 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (tid + j * TK_THREADS < a.HW) {
                kreg[j] = score_key(v[j]);
                tmax = tmax > kreg[j] ? tmax : kreg[j];
            }
        }
    }
    // Round 6: 16-byte loads where the image's pixel count allows (any partition of the pixels over the threads serves the bound: the m-th largest of a wave's 64
    // thread maxima has m elements at or above it) — the 41 K-pixel maps of 608 x 1088 frames took three rounds of sixteen 4-byte loads per thread, now eleven
    // 16-byte loads in two rounds.
    const bool vec4 = !in_regs && (a.HW & 3) == 0;
    // ... and maps of up to 48 K pixels whose keys do not fit the LDS keep them in REGISTERS, twelve quads per thread (the kernel runs 1024 threads of at most 128 registers):
    // the compaction's scan is then 48 register compares instead of 41 two-byte LDS reads behind per-element branches (7-8 K of its 12.7 K cycles, r6g)
    const bool reg48 = R48 && vec4 && !klds && KCH <= 48;
    unsigned kq[48];
#pragma unroll
    for (int j = 0; j < 48; ++j) kq[j] = 0u;
    const int nquads = (a.HW >> 2) > tid ? ((a.HW >> 2) - tid + TK_THREADS - 1) / TK_THREADS : 0;      // quads tid, tid + 1024, ... this thread owns
    if (reg48) {
#pragma unroll
        for (int b0 = 0; b0 < 12; b0 += 6) {
            f32x4 v[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) v[j] = b0 + j < nquads ? *reinterpret_cast<const f32x4*>(sc + 4 * (tid + (b0 + j) * TK_THREADS)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (b0 + j < nquads) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        kq[4 * (b0 + j) + c] = score_key(v[j][c]);
                        tmax = tmax > kq[4 * (b0 + j) + c] ? tmax : kq[4 * (b0 + j) + c];
                    }
                }
        }
    }
    for (int q = tid; vec4 && !reg48 && q < (a.HW >> 2); q += 8 * TK_THREADS) {
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = q + j * TK_THREADS < (a.HW >> 2) ? *reinterpret_cast<const f32x4*>(sc + 4 * (q + j * TK_THREADS)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int qq = q + j * TK_THREADS;
            if (qq < (a.HW >> 2)) {
                unsigned key[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    key[c] = score_key(v[j][c]);
                    tmax = tmax > key[c] ? tmax : key[c];
                }
                if (klds) *reinterpret_cast<uint4*>(lds_keys + 4 * qq) = make_uint4(key[0], key[1], key[2], key[3]);
                if (k16) *reinterpret_cast<uint2*>(lds_k16 + 4 * qq) = make_uint2((key[0] >> 16) | (key[1] & 0xFFFF0000u), (key[2] >> 16) | (key[3] & 0xFFFF0000u));
            }
        }
    }
    for (int i = tid; !in_regs && !vec4 && i < a.HW; i += 16 * TK_THREADS) {      // sixteen independent loads in flight per thread, then the bookkeeping
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = i + j * TK_THREADS < a.HW ? sc[i + j * TK_THREADS] : 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int ii = i + j * TK_THREADS;
            if (ii < a.HW) {
                const unsigned key = score_key(v[j]);
                if (klds) lds_keys[ii] = key;             // identity layout: consecutive lanes, consecutive words (round 6: the [owner][odd pitch] image cost a division per key)
                if (k16) lds_k16[ii] = (unsigned short)(key >> 16);
                tmax = tmax > key ? tmax : key;
            }
        }
    }
    {
        // wave bitonic sort of the 64 thread maxima, descending.  Exchange distances 1 / 2: DPP quad permutes (one VALU instruction),
        // 4 / 8 / 16: ds_swizzle xor masks, 32: one bpermute.
        TK_STAMP(9);
        unsigned v = tmax;
#define TK_EXCH(st_) ((st_) == 1 ? (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false)                        \
                    : (st_) == 2 ? (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false)                        \
                    : (st_) == 4 ? (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x1F | (4 << 10))                                \
                    : (st_) == 8 ? (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x1F | (8 << 10))                                \
                    : (st_) == 16 ? (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x1F | (16 << 10))                              \
                    : (unsigned)__shfl_xor((int)v, 32))
#pragma unroll
        for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
            for (int st = size >> 1; st > 0; st >>= 1) {
                const unsigned pv = TK_EXCH(st);
                const bool take_max = ((lane & st) == 0) == ((lane & size) == 0);
                v = take_max ? (v > pv ? v : pv) : (v < pv ? v : pv);
            }
        }
#undef TK_EXCH
        const unsigned vm = __shfl(v, (a.k + TK_THREADS / 64 - 1) / (TK_THREADS / 64) - 1);
        if (lane == 0) { wave_tot[wave] = vm; wave_max[wave] = v; }
        TK_STAMP(10);
    }
    __syncthreads();                                         // + every key is in LDS, cand is zero, the counter is reset
    TK_STAMP(1);
    unsigned Tlo = wave_tot[0];
#pragma unroll
    for (int w = 1; w < TK_THREADS / 64; ++w) Tlo = Tlo < wave_tot[w] ? Tlo : wave_tot[w];
    // SECOND STEP of the bound (round 6).  T_lo is set by the weakest wave: 1.5-3 k elements pass it, and the rank step below costs n^2.  The k-th largest of the
    // 1024 thread MAXIMA is a bound too (k threads hold an element at or above it) and lets ~1.1 k through; it is found to within one bin: the maxima >= T_lo
    // (a few hundred) are counted into 512-1024 equal bins of key space between T_lo and the largest maximum (one LDS atomic per counted thread, one barrier);
    // then wave 0 scans all the bins from the top — lane l sums the 16 bins of block 63 - l, a DPP prefix scan over the lanes finds the block that holds the k-th
    // largest maximum, the lane's own 16 counts the bin — and a second barrier hands the lower edge of that bin, the new bound, to the others.
    {
        unsigned gmax = wave_max[0];
#pragma unroll
        for (int w = 1; w < TK_THREADS / 64; ++w) gmax = gmax > wave_max[w] ? gmax : wave_max[w];
        const unsigned range = gmax - Tlo;
        if (range != 0u) {                                   // (block-uniform)
            const int bsh = range >= 1024u ? 22 - __clz(range) : 0;              // (range >> bsh) < 1024
            if (tmax >= Tlo) atomicAdd(&hist[(tmax - Tlo) >> bsh], 1u);
            __syncthreads();
            if (wave == 0) {                                 // one wave scans (every instruction of a 16-wave block is paid 4 x: four waves share a SIMD); the others wait
                const int blk = 63 - lane;
                const uint4* hp = reinterpret_cast<const uint4*>(hist + 16 * blk);
                const uint4 h0 = hp[0], h1 = hp[1], h2 = hp[2], h3 = hp[3];
                const unsigned hb[16] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w, h2.x, h2.y, h2.z, h2.w, h3.x, h3.y, h3.z, h3.w};
                unsigned own = 0u;
#pragma unroll
                for (int j = 0; j < 16; ++j) own += hb[j];
                const unsigned P = wave_incl_scan(own);                              // maxima in this lane's block and the blocks above it
                const unsigned long long bal = __ballot(P >= (unsigned)a.k);          // never empty: >= k maxima pass T_lo
                const int L = __builtin_ctzll(bal);                                  // the lane whose block holds the k-th largest maximum
                unsigned acc = P - own;                                              // maxima in the blocks above this lane's
                int bin = 0;
                bool found = false;
#pragma unroll
                for (int j = 15; j >= 0; --j) {
                    acc += hb[j];
                    if (!found && acc >= (unsigned)a.k) { found = true; bin = j; }
                }
                const unsigned binL = (unsigned)__shfl(16 * blk + bin, L);
                if (lane == 0) sh_prefix = Tlo + (binL << bsh);
            }
            __syncthreads();
            Tlo = sh_prefix;
        }
    }
    bool fast = false;
    {
        // compaction of the elements >= T_lo into cand[] — in ANY order: the winners are placed by rank of the (key, ~index) pair below.
        // Thread t owns the contiguous index range [t*KCH, (t+1)*KCH); one LDS atomic per wave and round reserves the wave's slots.
        const int j0 = tid * KCH, j1 = min(j0 + KCH, a.HW);
        const bool strided = !in_regs && KCH <= 64;          // up to 64 x 1024 pixels: thread t takes indices t + 1024 c again (coalesced; the
                                                             // scores are L2-resident by now) — the compaction needs no index order
        if (in_regs || strided) {
            unsigned long long qual = 0ull;                  // bit c: the thread's c-th element passes the bound
            if (in_regs) {
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (tid + j * TK_THREADS < a.HW && kreg[j] >= Tlo) qual |= 1ull << j;
            }
            if (reg48) {
#pragma unroll
                for (int b = 0; b < 48; ++b) qual |= (unsigned long long)((b >> 2) < nquads && kq[b] >= Tlo) << b;
            }
            for (int c0 = 0; strided && !reg48 && c0 < KCH; c0 += 8) {             // eight independent loads at a time
                if (klds) {          // the keys are in LDS (this thread wrote exactly these words): no second pass over memory (round 6: the 41 K-pixel maps of 608 x 1088 frames)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (tid + (c0 + j) * TK_THREADS < a.HW && lds_keys[tid + (c0 + j) * TK_THREADS] >= Tlo) qual |= 1ull << (c0 + j);
                    continue;
                }
                if (k16) {           // upper halves only: a superset (exact test below, on the re-read score)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (tid + (c0 + j) * TK_THREADS < a.HW && lds_k16[tid + (c0 + j) * TK_THREADS] >= (unsigned short)(Tlo >> 16)) qual |= 1ull << (c0 + j);
                    continue;
                }
                float vv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) vv[j] = tid + (c0 + j) * TK_THREADS < a.HW ? sc[tid + (c0 + j) * TK_THREADS] : 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (tid + (c0 + j) * TK_THREADS < a.HW && score_key(vv[j]) >= Tlo) qual |= 1ull << (c0 + j);
            }
            TK_STAMP(12);
            if (in_regs || reg48) {
                // The keys are in registers: a prefix sum of the lanes' candidate counts and ONE LDS atomic per wave reserve the slots; every lane then walks its own
                // bits and picks the key out of its registers with a binary tree of selects on the bits of the position (15 / 47 v_cndmask).  (The rounds below — a
                // ballot, an atomic with its return and, for the 48-register maps, a re-read of the score per round — took 2.1-2.3 K cycles in the first wave and left
                // it 1.3-4.7 K at the barrier behind the waves with a four-candidate lane; a statically unrolled walk over the 48 positions, 6-8 K.)
                const unsigned cnt = (unsigned)__builtin_popcountll(qual);
                const unsigned incl = wave_incl_scan(cnt);
                const unsigned wtot = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
                if (wtot != 0u) {                                                // (wave-uniform)
                    unsigned base = 0u;
                    if (lane == 63) base = atomicAdd(&sh_count, wtot);
                    base = (unsigned)__builtin_amdgcn_readlane((int)base, 63);
                    unsigned off = base + incl - cnt;
                    while (qual != 0ull) {
                        const int c = __builtin_ctzll(qual);
                        qual &= qual - 1ull;
                        unsigned key, idx;
                        // (bit masks, v_bfi_b32, not ?: — the compiler turns a tree of selects back into a dynamically indexed array, and that into scratch memory)
                        const unsigned m1 = 0u - (unsigned)(c & 1), m2 = 0u - (unsigned)((c >> 1) & 1), m4 = 0u - (unsigned)((c >> 2) & 1), m8 = 0u - (unsigned)((c >> 3) & 1);
#define TK_SEL(m_, hi_, lo_) (((hi_) & (m_)) | ((lo_) & ~(m_)))
                        if (in_regs) {
                            unsigned t8[8], t4[4];
#pragma unroll
                            for (int i = 0; i < 8; ++i) t8[i] = TK_SEL(m1, kreg[2 * i + 1], kreg[2 * i]);
#pragma unroll
                            for (int i = 0; i < 4; ++i) t4[i] = TK_SEL(m2, t8[2 * i + 1], t8[2 * i]);
                            const unsigned u0 = TK_SEL(m4, t4[1], t4[0]), u1 = TK_SEL(m4, t4[3], t4[2]);
                            key = TK_SEL(m8, u1, u0);
                            idx = (unsigned)(tid + c * TK_THREADS);
                        } else {
                            unsigned t24[24], t12[12], t6[6], t3[3];
#pragma unroll
                            for (int i = 0; i < 24; ++i) t24[i] = TK_SEL(m1, kq[2 * i + 1], kq[2 * i]);
#pragma unroll
                            for (int i = 0; i < 12; ++i) t12[i] = TK_SEL(m2, t24[2 * i + 1], t24[2 * i]);
#pragma unroll
                            for (int i = 0; i < 6; ++i) t6[i] = TK_SEL(m4, t12[2 * i + 1], t12[2 * i]);
#pragma unroll
                            for (int i = 0; i < 3; ++i) t3[i] = TK_SEL(m8, t6[2 * i + 1], t6[2 * i]);
                            const unsigned m16 = 0u - (unsigned)((c >> 4) & 1), m32 = 0u - (unsigned)((c >> 5) & 1);
                            key = TK_SEL(m32, t3[2], TK_SEL(m16, t3[1], t3[0]));
                            idx = (unsigned)(4 * (tid + (c >> 2) * TK_THREADS) + (c & 3));
                        }
#undef TK_SEL
                        if (off < (unsigned)FAST_CAP) cand[off] = ((unsigned long long)key << 32) | (0xFFFFFFFFu - idx);
                        ++off;
                    }
                }
                qual = 0ull;
            }
            TK_STAMP(13);
            for (;;) {                                       // most threads own no candidate at all: 1-3 rounds per wave
                if (__ballot(qual != 0ull) == 0ull) break;
                bool has = qual != 0ull;
                unsigned key = 0u, idx = 0u;
                if (has) {
                    const int c = __builtin_ctzll(qual);
                    qual &= qual - 1;
                    idx = reg48 ? (unsigned)(4 * (tid + (c >> 2) * TK_THREADS) + (c & 3)) : (unsigned)(tid + c * TK_THREADS);
                    if (in_regs) {
                        key = kreg[0];
#pragma unroll
                        for (int j = 1; j < 16; ++j) key = c == j ? kreg[j] : key;
                    } else {
                        key = klds ? lds_keys[idx] : score_key(sc[idx]);
                        has = key >= Tlo;                    // (k16: the upper halves let a few elements below the bound through)
                    }
                }
                const unsigned long long bal = __ballot(has);
                if (bal == 0ull) continue;
                const int leader = __builtin_ctzll(bal);
                unsigned base = 0u;
                if (lane == leader) base = atomicAdd(&sh_count, (unsigned)__builtin_popcountll(bal));
                base = (unsigned)__shfl((int)base, leader);
                if (has) {
                    const unsigned pos = base + (unsigned)__builtin_popcountll(bal & ((1ull << lane) - 1ull));
                    if (pos < (unsigned)FAST_CAP) cand[pos] = ((unsigned long long)key << 32) | (0xFFFFFFFFu - idx);
                }
            }
        } else {                                             // maps beyond the LDS budget: plain per-element reservation
            for (int i = j0; i < j1; ++i) {
                const unsigned key = klds ? lds_keys[tid * KCH + (i - j0)] : score_key(sc[i]);
                if (key >= Tlo) {
                    const unsigned pos = atomicAdd(&sh_count, 1u);
                    if (pos < (unsigned)FAST_CAP) cand[pos] = ((unsigned long long)key << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)i);
                }
            }
        }
        TK_STAMP(14);
        __syncthreads();
        const unsigned total = sh_count;
        TK_STAMP(2);
#ifdef TK_TIMING
        if (tid == 0 && blockIdx.x < 64) tk_stamps[blockIdx.x * 16 + 8] = total;
#endif
        if (total <= (unsigned)FAST_CAP) {                   // else: ties / plateaus en masse -> radix select below
            fast = true;
            // rank of a candidate = how many candidates are greater ((key, ~index) pairs are distinct): the k winners land in canonical
            // order (score desc, index asc) without a sort.  All lanes read the same 16 bytes (two candidates): LDS broadcasts.  With ONE candidate per
            // lane the LDS port bounds the step — a wave's 16-byte read occupies it for ~8 cycles whatever the addresses, n^2 / 64 pairs in all: 30-32 K
            // cycles at n = 569-623 (tools/topk_trace.py; round 6 first tried a 1024-wide bitonic sort above 512 candidates, 26 K, then a radix select
            // of the k-th key before the rank, 19 K).  Each lane therefore ranks FOUR candidates against every pair it reads — a quarter of the LDS
            // traffic for the same compares, which the VALU now bounds (v_cmp_gt_u64 is half rate: 12 cycles per compare-and-count) — and 1 .. 8
            // neighbouring lanes share a quad of candidates and scan interleaved slices of the array; lanes (and whole waves) without a quad skip the scan.
            const unsigned nq = (total + 3u) >> 2;
            int psh = 0;
            while (psh < 3 && (nq << (psh + 1)) <= (unsigned)TK_THREADS) ++psh;
            const unsigned cq = (unsigned)tid >> psh;
            const int part = tid & ((1 << psh) - 1);
            if (cq < nq) {
                unsigned long long mine[4];
                unsigned rank[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int m = 0; m < 4; ++m) mine[m] = cand[4u * cq + m];       // zero beyond total (the array is padded): never written below
                for (unsigned j = (unsigned)part * 8u; j < total; j += 8u << psh) {
                    unsigned long long o[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) o[q] = cand[j + q];
#pragma unroll
                    for (int q = 0; q < 8; ++q)
#pragma unroll
                        for (int m = 0; m < 4; ++m) rank[m] += o[q] > mine[m] ? 1u : 0u;
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    if (psh > 0) rank[m] += (unsigned)__shfl_xor((int)rank[m], 1);
                    if (psh > 1) rank[m] += (unsigned)__shfl_xor((int)rank[m], 2);
                    if (psh > 2) rank[m] += (unsigned)__shfl_xor((int)rank[m], 4);
                    if (part == 0 && 4u * cq + m < total && rank[m] < (unsigned)a.k) win[rank[m]] = mine[m];
                }
            }
            __syncthreads();
            TK_STAMP(3);
            TK_STAMP(4);
        }
    }
    if (!fast) {
    // --- radix select: key T of the k-th largest element, digits of 12 / 10 / 10 bits from the top.  (A 12-bit first digit
    // spreads sigmoid scores, which share 1-2 exponents, over 16x more bins than an 8-bit one: far less LDS-atomic contention.) ---
    unsigned prefix = 0, mask = 0, need = (unsigned)a.k;
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
        const int bits = pass == 0 ? 12 : 10;
        const int shift = pass == 0 ? 20 : (pass == 1 ? 10 : 0);
        const int nbins = 1 << bits;
        const int per = nbins / TK_THREADS;               // bins per thread in the scan: 4 or 1
        for (int i = tid; i < nbins; i += TK_THREADS) hist[i] = 0;
        __syncthreads();
        if (klds) {
            // keys live in LDS in index order (lds_keys[i] = key of pixel i): this pass needs no order and strides over them conflict-free; the ordered
            // compaction below reads thread t's contiguous range t*KCH .. (2-way and worse bank conflicts when KCH is even: the fallback path only)
            for (int i = tid; i < a.HW; i += TK_THREADS) {
                const unsigned key = lds_keys[i];
                if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & (unsigned)(nbins - 1)], 1u);
            }
        } else {
            for (int i = tid; i < a.HW; i += TK_THREADS) {          // maps too large for LDS: every pass streams the scores (L2-resident)
                const unsigned key = score_key(sc[i]);
                if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & (unsigned)(nbins - 1)], 1u);
            }
        }
        __syncthreads();
        // block-wide suffix sums over the bins from the top: thread t owns bins per*t .. per*t+per-1
        unsigned h[4] = {0u, 0u, 0u, 0u};
        unsigned own = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < per) { h[j] = hist[per * tid + j]; own += h[j]; }
        unsigned suf = own;                                // -> sum over lanes >= lane within the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned t = __shfl_down(suf, off);
            if (lane + off < 64) suf += t;
        }
        if (lane == 0) wave_tot[wave] = suf;
        __syncthreads();
        unsigned higher = 0;                               // elements in bins owned by higher waves
        for (int w = wave + 1; w < TK_THREADS / 64; ++w) higher += wave_tot[w];
        const unsigned incl = suf + higher;                // elements in bins >= this thread's lowest bin
        const unsigned above = incl - own;                 // elements in bins above this thread's bins
        if (above < need && need <= incl) {                // the k-th largest falls in one of this thread's bins (exactly one thread)
            unsigned rem = need - above;
            int d = 0;
#pragma unroll
            for (int j = 3; j >= 0; --j) {
                if (j < per) {
                    if (h[j] >= rem) { d = j; break; }
                    rem -= h[j];
                }
            }
            sh_prefix = prefix | ((unsigned)(per * tid + d) << shift);
            sh_need = rem;
        }
        __syncthreads();
        prefix = sh_prefix;
        need = sh_need;                         // elements == T (within the digits seen so far) still to take
        mask |= (unsigned)(nbins - 1) << shift;
    }
    const unsigned T = prefix;                   // exact key of the k-th largest
    // `need` = number of elements with key == T to take (lowest indices first); the rest have key > T.

    // --- ordered compaction: thread t owns the contiguous index range [t*CH, (t+1)*CH); counts packed (gt << 16 | eq) ---
    const int CH = KCH;
    const int i0 = tid * CH, i1 = min(i0 + CH, a.HW);
    unsigned cnt = 0;
    for (int i = i0; i < i1; ++i) {
        const unsigned key = klds ? lds_keys[tid * KCH + (i - i0)] : score_key(sc[i]);
        cnt += key > T ? 0x10000u : 0u;
        cnt += key == T ? 1u : 0u;
    }
    // CH <= 2^24 / 1024 elements per thread may overflow 16 bits in general; totals are bounded by HW <= 2^24, so scan the
    // two counters separately when HW > 65535, packed otherwise (the common case)
    unsigned pos_gt, pos_eq, total_gt;
    if (a.HW <= 65535) {
        const unsigned inc = wave_incl_scan(cnt, lane);
        if (lane == 63) wave_tot[wave] = inc;
        __syncthreads();
        unsigned base = 0, tot = 0;
        for (int w = 0; w < TK_THREADS / 64; ++w) {
            const unsigned t = wave_tot[w];
            if (w < wave) base += t;
            tot += t;
        }
        const unsigned excl = base + inc - cnt;
        pos_gt = excl >> 16;
        pos_eq = excl & 0xFFFFu;
        total_gt = tot >> 16;
    } else {
        const unsigned g = cnt >> 16, e = cnt & 0xFFFFu;
        const unsigned ig = wave_incl_scan(g, lane), ie = wave_incl_scan(e, lane);
        __shared__ unsigned wave_tot2[TK_THREADS / 64];
        if (lane == 63) { wave_tot[wave] = ig; wave_tot2[wave] = ie; }
        __syncthreads();
        unsigned bg = 0, be = 0, tg = 0;
        for (int w = 0; w < TK_THREADS / 64; ++w) {
            if (w < wave) { bg += wave_tot[w]; be += wave_tot2[w]; }
            tg += wave_tot[w];
        }
        pos_gt = bg + ig - g;
        pos_eq = be + ie - e;
        total_gt = tg;
    }
    for (int i = tid; i < a.KP; i += TK_THREADS) cand[i] = 0ull;   // padding sorts last
    __syncthreads();
    for (int i = i0; i < i1; ++i) {
        const unsigned key = klds ? lds_keys[tid * KCH + (i - i0)] : score_key(sc[i]);
        const unsigned long long comp = ((unsigned long long)key << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)i);
        if (key > T) {
            cand[pos_gt++] = comp;
        } else if (key == T) {
            if (pos_eq < need) cand[total_gt + pos_eq] = comp;
            ++pos_eq;
        }
    }
    __syncthreads();

    block_sort_desc(cand, a.KP, tid);                        // descending by (key, ~index)
    }   // !fast

    TK_STAMP(5);
    // --- gathers + box decode for the k winners ---
    const unsigned long long* winners = fast ? win : cand;
    // the winners' offsets into the embedding map, once per detection (the gather below had two integer divisions per ELEMENT: 13 K cycles at k = 300, E = 64);
    // `hist` is free on both paths by now — but `win` aliases its first k * 8 bytes: the offsets live behind them
    long* ebase = reinterpret_cast<long*>(hist) + 1024;
    if (tid < a.k) {
        const unsigned long long comp = winners[tid];
        const int idx = (int)(0xFFFFFFFFu - (unsigned)(comp & 0xFFFFFFFFull));
        const long o = (long)n * a.k + tid;
        a.scores[o] = sc[idx];
        a.indices[o] = idx;
        a.labels[o] = a.ws_label[(long)n * a.HW + idx];
        const int yi = idx / a.W, xi = idx - yi * a.W;
        decode_box(a.box + (long)n * a.bsn + (long)yi * a.bsh + (long)xi * a.bsw, a.bsc, xi, yi, a.W, a.H, a.normalize,
                   a.box_log, a.mult, a.stride, a.boxes + o * 4);
        if (a.reid && a.emb) ebase[tid] = (long)n * a.rsn + (long)yi * a.rsh + (long)xi * a.rsw;
    }
    if (a.reid && a.emb) {
        // embeddings: k*E elements, E-contiguous per detection (one coalesced row when reid is NHWC)
        __syncthreads();
        const bool e_pow2 = (a.E & (a.E - 1)) == 0;
        const int e_sh = 31 - __builtin_clz((unsigned)a.E);
        const int total = a.k * a.E;
        // channel-minor embeddings (what the reid head emits) with E % 4 == 0: 16 bytes per thread and request
        const bool e_vec4 = a.rsc == 1 && (a.E & 3) == 0 && ((a.rsn | a.rsh | a.rsw) & 3) == 0 && ((uintptr_t)a.reid & 15) == 0 && ((uintptr_t)a.emb & 15) == 0;
        for (int t0 = tid * 4; e_vec4 && t0 < total; t0 += 16 * TK_THREADS) {
            f32x4 g[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = t0 + j * 4 * TK_THREADS;
                if (t < total) {
                    const int d = e_pow2 ? (t >> e_sh) : t / a.E, e = t - d * a.E;
                    g[j] = *reinterpret_cast<const f32x4*>(a.reid + ebase[d] + e);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = t0 + j * 4 * TK_THREADS;
                if (t < total) *reinterpret_cast<f32x4*>(a.emb + (long)n * total + t) = g[j];
            }
        }
        for (int t0 = tid; !e_vec4 && t0 < total; t0 += 16 * TK_THREADS) {     // sixteen independent gathers in flight per thread, UNCONDITIONAL (the last round re-reads element total - 1: a load under a per-lane
            float g[16];                                                        // branch is waited for at the join); one at a time the loop was a chain of LDS read -> load -> store: 10-12 K cycles at k = 300, E = 64
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int t = min(t0 + j * TK_THREADS, total - 1);
                const int d = e_pow2 ? (t >> e_sh) : t / a.E, e = t - d * a.E;
                g[j] = a.reid[ebase[d] + (long)e * a.rsc];
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int t = t0 + j * TK_THREADS;
                if (t < total) a.emb[(long)n * total + t] = g[j];
            }
        }
    }
    TK_STAMP(6);
}

// ---- standalone gathers at caller-supplied indices (heads[name].gather_at_indices) ----
__global__ __launch_bounds__(256) void gather_boxes_kernel(const float* __restrict__ box, long sn, long sc, long sh, long sw,
                                                           const long long* __restrict__ indices, float* __restrict__ boxes,
                                                           int N, int H, int W, int k, int normalize, int box_log, float mult,
                                                           float stride) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)N * k) return;
    const int n = (int)(t / k);
    const long long idx = indices[t];
    const int yi = (int)(idx / W), xi = (int)(idx - (long long)yi * W);
    decode_box(box + (long)n * sn + (long)yi * sh + (long)xi * sw, sc, xi, yi, W, H, normalize, box_log, mult, stride,
               boxes + t * 4);
}

__global__ __launch_bounds__(256) void gather_emb_kernel(const float* __restrict__ reid, long sn, long sc, long sh, long sw,
                                                         const long long* __restrict__ indices, float* __restrict__ emb, int N,
                                                         int E, int W, int k) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)N * k * E) return;
    const long d = t / E;
    const int e = (int)(t - d * E);
    const int n = (int)(d / k);
    const long long idx = indices[d];
    const int yi = (int)(idx / W), xi = (int)(idx - (long long)yi * W);
    emb[t] = reid[(long)n * sn + (long)e * sc + (long)yi * sh + (long)xi * sw];
}

template <int VEC>
int launch_cminor(const PeakArgs& a, int P, size_t lds, unsigned blocks, hipStream_t s) {
    switch (P) {
        case 0: hipLaunchKernelGGL((peaks_cminor_kernel<VEC, 0>), dim3(blocks), dim3(256), lds, s, a); break;
        case 1: hipLaunchKernelGGL((peaks_cminor_kernel<VEC, 1>), dim3(blocks), dim3(256), lds, s, a); break;
        case 2: hipLaunchKernelGGL((peaks_cminor_kernel<VEC, 2>), dim3(blocks), dim3(256), lds, s, a); break;
        default: hipLaunchKernelGGL((peaks_cminor_kernel<VEC, 3>), dim3(blocks), dim3(256), lds, s, a); break;
    }
    return cnl::check_launch("peaks_cminor_kernel");
}

}  // namespace cnl_decode
using namespace cnl_decode;

#ifdef TK_TIMING
extern "C" __attribute__((visibility("default"))) int cnl_debug_topk_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(cnl_decode::tk_stamps), sizeof(unsigned long long) * 64 * 16) == hipSuccess ? 0 : 1;
}
#endif

extern "C" size_t cnl_decode_workspace_bytes(int32_t N, int32_t H, int32_t W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)N * H * W * 8 + 256;
}

extern "C" int cnl_decode_f32(const cnl_decode_params* p, void* stream) {
    CNL_REQUIRE(p, CNL_E_BAD_ARG, "cnl_decode_f32: null params");
    CNL_REQUIRE(p->heat && p->box && p->scores && p->indices && p->labels && p->boxes, CNL_E_BAD_ARG,
                "cnl_decode_f32: null tensor pointer");
    CNL_REQUIRE(p->N > 0 && p->C > 0 && p->H > 0 && p->W > 0, CNL_E_BAD_ARG, "cnl_decode_f32: non-positive dimension");
    const long long HWll = (long long)p->H * p->W;
    CNL_REQUIRE(HWll <= (1ll << 24), CNL_E_UNSUPPORTED, "cnl_decode_f32: H*W = %lld exceeds 2^24", HWll);
    const int HW = (int)HWll;
    CNL_REQUIRE(p->k >= 1 && p->k <= 1024 && p->k <= HW, CNL_E_UNSUPPORTED,
                "cnl_decode_f32: num_detections k=%d outside [1, min(1024, H*W=%d)]", p->k, HW);
    CNL_REQUIRE(p->nms_kernel >= 1 && p->nms_kernel <= 7 && (p->nms_kernel & 1), CNL_E_UNSUPPORTED,
                "cnl_decode_f32: nms_kernel=%d must be odd and <= 7", p->nms_kernel);
    CNL_REQUIRE(!p->reid || (p->E > 0 && p->emb), CNL_E_BAD_ARG, "cnl_decode_f32: reid given without E / emb output");
    const size_t need_ws = cnl_decode_workspace_bytes(p->N, p->H, p->W);
    CNL_REQUIRE(p->workspace && p->workspace_bytes >= need_ws, CNL_E_WORKSPACE,
                "cnl_decode_f32: workspace %zu bytes < required %zu", p->workspace_bytes, need_ws);
    CNL_REQUIRE(((uintptr_t)p->workspace & 15) == 0, CNL_E_BAD_ARG, "cnl_decode_f32: workspace must be 16-byte aligned");

    hipStream_t s = (hipStream_t)stream;
    const int P = (p->nms_kernel - 1) / 2;
    PeakArgs a;
    a.heat = p->heat; a.sn = p->heat_sn; a.sc = p->heat_sc; a.sh = p->heat_sh; a.sw = p->heat_sw;
    a.N = p->N; a.C = p->C; a.H = p->H; a.W = p->W;
    a.ws_score = (float*)p->workspace;
    a.ws_label = (int*)((char*)p->workspace + (size_t)p->N * HW * 4);

    int rc;
    const bool cminor = p->heat_sc == 1;
    int vec = 1;
    if (cminor) {
        auto ok = [&](int v) {
            return p->C % v == 0 && p->heat_sn % v == 0 && p->heat_sh % v == 0 && p->heat_sw % v == 0 &&
                   ((uintptr_t)p->heat % (4 * v)) == 0;
        };
        vec = ok(4) ? 4 : (ok(2) ? 2 : 1);
    }
    if (cminor && vec == 4 && p->C % 8 == 0 && p->C / 8 <= PK8_THREADS && P <= 3 && p->W >= 8) {
        Peak8Args q;
        q.CG8 = p->C / 8;
        q.RUNS = PK8_THREADS / q.CG8;
        if (q.RUNS > 32) q.RUNS = 32;                      // blocks at most 128 pixels wide: R x TW x 8 bytes of LDS stays small for any C
        int R8 = 16;
        // pixels per run: 2 for the 3 x 3 pool (P = 1; r6q: 111 registers, four 160-thread workgroups per CU instead of two at 195, 32-pixel blocks: 30.4 -> 28.5 us at C1 =
        // 5.9 TB/s, what a plain read-once stream of the same bytes gets), 4 otherwise
        const int rp = P >= 1 ? 2 : 4;      // (5 x 5: 176 registers at two-pixel runs; four-pixel runs spilled 424)
        const int runs_w = (p->W + rp - 1) / rp;
        if (q.RUNS > runs_w) q.RUNS = runs_w;
        q.TW = q.RUNS * rp;
        a.tiles_x = (p->W + q.TW - 1) / q.TW;
        if ((long long)p->N * a.tiles_x * ((p->H + 15) / 16) < 256) R8 = 4;      // few images: more, shorter strips (results are identical)
        a.strips = (p->H + R8 - 1) / R8;
        a.CG = q.CG8; a.PXB = q.TW; a.R = R8;
        q.p = a;
        const long long blocks = (long long)p->N * a.tiles_x * a.strips;
        CNL_REQUIRE(blocks < (1ll << 31), CNL_E_UNSUPPORTED, "cnl_decode_f32: grid too large");
        const size_t lds = (size_t)R8 * q.TW * 8;
#define PK8_LAUNCH(P_, R_) hipLaunchKernelGGL((peaks_c8_kernel<P_, R_>), dim3((unsigned)blocks), dim3(PK8_THREADS), lds, s, q)
        if (R8 == 16) {
            if (P == 0) PK8_LAUNCH(0, 16); else if (P == 1 && rp == 2) hipLaunchKernelGGL((peaks_c8_kernel<1, 16, 2>), dim3((unsigned)blocks), dim3(PK8_THREADS), lds, s, q); else if (P == 1) PK8_LAUNCH(1, 16); else if (P == 2) hipLaunchKernelGGL((peaks_c8_kernel<2, 16, 2>), dim3((unsigned)blocks), dim3(PK8_THREADS), lds, s, q); else hipLaunchKernelGGL((peaks_c8_kernel<3, 16, 2>), dim3((unsigned)blocks), dim3(PK8_THREADS), lds, s, q);
        } else {
            if (P == 0) PK8_LAUNCH(0, 4); else if (P == 1 && rp == 2) hipLaunchKernelGGL((peaks_c8_kernel<1, 4, 2>), dim3((unsigned)blocks), dim3(PK8_THREADS), lds, s, q); else if (P == 1) PK8_LAUNCH(1, 4); else if (P == 2) hipLaunchKernelGGL((peaks_c8_kernel<2, 4, 2>), dim3((unsigned)blocks), dim3(PK8_THREADS), lds, s, q); else hipLaunchKernelGGL((peaks_c8_kernel<3, 4, 2>), dim3((unsigned)blocks), dim3(PK8_THREADS), lds, s, q);
        }
#undef PK8_LAUNCH
        rc = cnl::check_launch("peaks_c8_kernel");
    } else if (cminor && p->C / vec <= 256) {
        a.CG = p->C / vec;
        a.PXB = 256 / a.CG;
        if (a.PXB > p->W) a.PXB = p->W;
        a.R = 8;                                          // measured best of {4, 8, 16, 32} rows per strip at C1; = CM_R in peaks_cminor_kernel (its batched-load form unrolls over it)
        a.tiles_x = (p->W + a.PXB - 1) / a.PXB;
        a.strips = (p->H + a.R - 1) / a.R;
        const long long blocks = (long long)p->N * a.tiles_x * a.strips;
        CNL_REQUIRE(blocks < (1ll << 31), CNL_E_UNSUPPORTED, "cnl_decode_f32: grid too large");
        const size_t lds = (size_t)a.R * a.PXB * a.CG * 8;
        if (vec == 4) rc = launch_cminor<4>(a, P, lds, (unsigned)blocks, s);
        else if (vec == 2) rc = launch_cminor<2>(a, P, lds, (unsigned)blocks, s);
        else rc = launch_cminor<1>(a, P, lds, (unsigned)blocks, s);
    } else if (P <= 3 && p->C >= PLANES_MIN_C && p->heat_sw == 1 && (p->W & 3) == 0 && ((p->heat_sn | p->heat_sc | p->heat_sh) & 3) == 0 && ((uintptr_t)p->heat & 15) == 0) {
        a.CG = p->C >= 16 ? 16 : (p->C >= 8 ? 8 : 4);      // contiguous rows (NCHW): class planes; class groups x runs = 256 threads
        a.PXB = 256 / a.CG * 4;
        const int R = P <= 1 ? 8 : 4;
        a.R = R;
        a.tiles_x = (p->W + a.PXB - 1) / a.PXB;
        a.strips = (p->H + R - 1) / R;
        const long long blocks = (long long)p->N * a.tiles_x * a.strips;
        CNL_REQUIRE(blocks < (1ll << 31), CNL_E_UNSUPPORTED, "cnl_decode_f32: grid too large");
        const size_t lds = (size_t)R * a.PXB * 8;
        switch (P) {
            case 0: hipLaunchKernelGGL((peaks_planes_kernel<0, 8>), dim3((unsigned)blocks), dim3(256), lds, s, a); break;
            case 1: hipLaunchKernelGGL((peaks_planes_kernel<1, 8>), dim3((unsigned)blocks), dim3(256), lds, s, a); break;
            case 2: hipLaunchKernelGGL((peaks_planes_kernel<2, 4>), dim3((unsigned)blocks), dim3(256), lds, s, a); break;
            default: hipLaunchKernelGGL((peaks_planes_kernel<3, 4>), dim3((unsigned)blocks), dim3(256), lds, s, a); break;
        }
        rc = cnl::check_launch("peaks_planes_kernel");
    } else {
        constexpr int R = 8;
        a.CG = 1; a.PXB = 64; a.R = R;
        a.tiles_x = (p->W + 63) / 64;
        a.strips = (p->H + 4 * R - 1) / (4 * R);
        const long long blocks = (long long)p->N * a.tiles_x * a.strips;
        CNL_REQUIRE(blocks < (1ll << 31), CNL_E_UNSUPPORTED, "cnl_decode_f32: grid too large");
        switch (P) {
            case 0: hipLaunchKernelGGL((peaks_generic_kernel<0, R>), dim3((unsigned)blocks), dim3(256), 0, s, a); break;
            case 1: hipLaunchKernelGGL((peaks_generic_kernel<1, R>), dim3((unsigned)blocks), dim3(256), 0, s, a); break;
            case 2: hipLaunchKernelGGL((peaks_generic_kernel<2, R>), dim3((unsigned)blocks), dim3(256), 0, s, a); break;
            default: hipLaunchKernelGGL((peaks_generic_kernel<3, R>), dim3((unsigned)blocks), dim3(256), 0, s, a); break;
        }
        rc = cnl::check_launch("peaks_generic_kernel");
    }
    if (rc != CNL_OK) return rc;

    TopkArgs t;
    t.ws_score = a.ws_score; t.ws_label = a.ws_label;
    t.box = p->box; t.bsn = p->box_sn; t.bsc = p->box_sc; t.bsh = p->box_sh; t.bsw = p->box_sw;
    t.reid = p->reid; t.rsn = p->reid_sn; t.rsc = p->reid_sc; t.rsh = p->reid_sh; t.rsw = p->reid_sw;
    t.HW = HW; t.W = p->W; t.H = p->H; t.E = p->reid ? p->E : 0; t.k = p->k;
    int kp = 2;
    while (kp < p->k) kp <<= 1;
    t.KP = kp;
    t.normalize = p->normalize_boxes; t.box_log = p->box_log; t.mult = p->box_multiplier; t.stride = p->stride;
    t.scores = p->scores; t.indices = (long long*)p->indices; t.labels = (long long*)p->labels; t.boxes = p->boxes;
    t.emb = p->emb;
    // 40 KB of static LDS + the keys: one workgroup per CU either way (1024 threads)
    const size_t key_words = (size_t)((HW + TK_THREADS - 1) / TK_THREADS) * TK_THREADS;
    t.keys_in_lds = key_words * 4 <= 96 * 1024;
    t.keys16_in_lds = !t.keys_in_lds && key_words * 2 <= 96 * 1024;
    const size_t key_bytes = t.keys_in_lds ? key_words * 4 : (t.keys16_in_lds ? key_words * 2 : 0);
    const int kch = (HW + TK_THREADS - 1) / TK_THREADS;
    const bool r48 = (HW & 3) == 0 && !t.keys_in_lds && kch > 16 && kch <= 48;
    static cnl::DeviceOnce once, once48;
    rc = r48 ? cnl::kernel_setup(once48, reinterpret_cast<const void*>(&topk_kernel<true>), 96 * 1024) : cnl::kernel_setup(once, reinterpret_cast<const void*>(&topk_kernel<false>), 96 * 1024);
    if (rc != CNL_OK) return rc;
    if (r48) hipLaunchKernelGGL(topk_kernel<true>, dim3(p->N), dim3(TK_THREADS), 0, s, t);
    else hipLaunchKernelGGL(topk_kernel<false>, dim3(p->N), dim3(TK_THREADS), key_bytes, s, t);
    return cnl::check_launch("topk_kernel");
}

extern "C" int cnl_gather_boxes_f32(const float* box, int64_t sn, int64_t sc, int64_t sh, int64_t sw, const int64_t* indices,
                                    float* boxes, int32_t N, int32_t H, int32_t W, int32_t k, int32_t normalize_boxes,
                                    int32_t box_log, float box_multiplier, float stride, void* stream) {
    CNL_REQUIRE(box && indices && boxes, CNL_E_BAD_ARG, "cnl_gather_boxes_f32: null tensor pointer");
    CNL_REQUIRE(N > 0 && H > 0 && W > 0 && k > 0, CNL_E_BAD_ARG, "cnl_gather_boxes_f32: non-positive dimension");
    const long total = (long)N * k;
    hipLaunchKernelGGL(gather_boxes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, box, (long)sn,
                       (long)sc, (long)sh, (long)sw, (const long long*)indices, boxes, N, H, W, k, normalize_boxes, box_log,
                       box_multiplier, stride);
    return cnl::check_launch("gather_boxes_kernel");
}

extern "C" int cnl_gather_embeddings_f32(const float* reid, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                                         const int64_t* indices, float* emb, int32_t N, int32_t E, int32_t H, int32_t W,
                                         int32_t k, void* stream) {
    CNL_REQUIRE(reid && indices && emb, CNL_E_BAD_ARG, "cnl_gather_embeddings_f32: null tensor pointer");
    CNL_REQUIRE(N > 0 && E > 0 && H > 0 && W > 0 && k > 0, CNL_E_BAD_ARG, "cnl_gather_embeddings_f32: non-positive dimension");
    const long total = (long)N * k * E;
    hipLaunchKernelGGL(gather_emb_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, reid, (long)sn,
                       (long)sc, (long)sh, (long)sw, (const long long*)indices, emb, N, E, W, k);
    return cnl::check_launch("gather_emb_kernel");
}
