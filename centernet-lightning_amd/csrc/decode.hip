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
// Stage 2 (topk_kernel): one workgroup per image.  Radix select (4 x 8-bit digits, LDS histograms) of the
//   k-th largest score key; ordered compaction (keys above the threshold + the lowest-index ties);
//   bitonic sort of <= 1024 (key, ~index) pairs in LDS -> (score desc, index asc), the canonical
//   order the oracle defines where torch.topk leaves ties unspecified; then the label / ltrb box /
//   embedding gathers and the box decode for the k winners only (the reference transforms the whole
//   4xHxW map first, centernet.py:282-286).
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

constexpr int TK_THREADS = 1024;

// wave-level helpers (64-wide wavefront)
__device__ __forceinline__ unsigned wave_incl_scan(unsigned v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    return v;
}

__global__ __launch_bounds__(TK_THREADS) void topk_kernel(const TopkArgs a) {
    extern __shared__ unsigned lds_keys[];          // [HW] when a.keys_in_lds
    __shared__ unsigned hist[4096];
    __shared__ unsigned wave_tot[TK_THREADS / 64];
    __shared__ unsigned long long cand[1024];
    __shared__ unsigned sh_prefix, sh_need;

    const int n = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const float* sc = a.ws_score + (long)n * a.HW;
    const int KCH = (a.HW + TK_THREADS - 1) / TK_THREADS, KST = KCH | 1;      // indices per thread, LDS row pitch (odd)

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
        if (a.keys_in_lds && pass > 0) {
            // keys live in LDS as [thread][KST] (thread t owns indices t*KCH .. t*KCH+KCH-1; the odd row pitch KST keeps both this
            // loop and the ordered compaction below free of bank conflicts)
            for (int j = 0; j < KCH; ++j) {
                if (tid * KCH + j >= a.HW) break;
                const unsigned key = lds_keys[tid * KST + j];
                if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & (unsigned)(nbins - 1)], 1u);
            }
        } else {
            for (int i = tid; i < a.HW; i += TK_THREADS) {          // coalesced: the only pass over memory when the keys fit in LDS
                const unsigned key = score_key(sc[i]);
                if (a.keys_in_lds) lds_keys[(i / KCH) * KST + (i % KCH)] = key;
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
        const unsigned key = a.keys_in_lds ? lds_keys[tid * KST + (i - i0)] : score_key(sc[i]);
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
        const unsigned key = a.keys_in_lds ? lds_keys[tid * KST + (i - i0)] : score_key(sc[i]);
        const unsigned long long comp = ((unsigned long long)key << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)i);
        if (key > T) {
            cand[pos_gt++] = comp;
        } else if (key == T) {
            if (pos_eq < need) cand[total_gt + pos_eq] = comp;
            ++pos_eq;
        }
    }
    __syncthreads();

    // --- sort descending by (key, ~index) ---
    if (a.KP <= 128) {
        // one wave, two elements per lane (indices lane and lane + 64), exchanges by shuffle: no barriers
        if (wave == 0) {
            unsigned long long e0 = lane < a.KP ? cand[lane] : 0ull;
            unsigned long long e1 = lane + 64 < a.KP ? cand[lane + 64] : 0ull;
            for (int size = 2; size <= 128; size <<= 1) {
                for (int st = size >> 1; st > 0; st >>= 1) {
                    if (st == 64) {
                        // partner of index lane is lane + 64: same lane; direction of the final merge is descending
                        if (e0 < e1) { const unsigned long long t = e0; e0 = e1; e1 = t; }
                    } else {
                        const unsigned long long p0 = __shfl_xor(e0, st), p1 = __shfl_xor(e1, st);
                        const bool lower = (lane & st) == 0;                   // this lane holds the lower index of the pair
                        const bool desc0 = (lane & size) == 0;                 // block direction for index lane
                        const bool desc1 = ((lane + 64) & size) == 0;          // ... and for index lane + 64
                        const bool take_max0 = lower == desc0, take_max1 = lower == desc1;
                        e0 = take_max0 ? (e0 > p0 ? e0 : p0) : (e0 < p0 ? e0 : p0);
                        e1 = take_max1 ? (e1 > p1 ? e1 : p1) : (e1 < p1 ? e1 : p1);
                    }
                }
            }
            cand[lane] = e0;
            cand[lane + 64] = e1;
        }
        __syncthreads();
    } else {
        for (int size = 2; size <= a.KP; size <<= 1) {
            for (int st = size >> 1; st > 0; st >>= 1) {
                if (tid < a.KP) {
                    const int j = tid ^ st;
                    if (j > tid) {
                        const unsigned long long x = cand[tid], y = cand[j];
                        const bool desc = (tid & size) == 0;
                        if (desc ? (x < y) : (x > y)) { cand[tid] = y; cand[j] = x; }
                    }
                }
                __syncthreads();
            }
        }
    }

    // --- gathers + box decode for the k winners ---
    if (tid < a.k) {
        const unsigned long long comp = cand[tid];
        const int idx = (int)(0xFFFFFFFFu - (unsigned)(comp & 0xFFFFFFFFull));
        const long o = (long)n * a.k + tid;
        a.scores[o] = sc[idx];
        a.indices[o] = idx;
        a.labels[o] = a.ws_label[(long)n * a.HW + idx];
        const int yi = idx / a.W, xi = idx - yi * a.W;
        decode_box(a.box + (long)n * a.bsn + (long)yi * a.bsh + (long)xi * a.bsw, a.bsc, xi, yi, a.W, a.H, a.normalize,
                   a.box_log, a.mult, a.stride, a.boxes + o * 4);
    }
    if (a.reid && a.emb) {
        // embeddings: k*E elements, E-contiguous per detection (one coalesced row when reid is NHWC)
        for (int t = tid; t < a.k * a.E; t += TK_THREADS) {
            const int d = t / a.E, e = t - d * a.E;
            const unsigned long long comp = cand[d];
            const int idx = (int)(0xFFFFFFFFu - (unsigned)(comp & 0xFFFFFFFFull));
            const int yi = idx / a.W, xi = idx - yi * a.W;
            a.emb[((long)n * a.k + d) * a.E + e] =
                a.reid[(long)n * a.rsn + (long)e * a.rsc + (long)yi * a.rsh + (long)xi * a.rsw];
        }
    }
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
    if (cminor && p->C / vec <= 256) {
        a.CG = p->C / vec;
        a.PXB = 256 / a.CG;
        if (a.PXB > p->W) a.PXB = p->W;
        a.R = 8;                                          // measured best of {4, 8, 16, 32} rows per strip at C1
        a.tiles_x = (p->W + a.PXB - 1) / a.PXB;
        a.strips = (p->H + a.R - 1) / a.R;
        const long long blocks = (long long)p->N * a.tiles_x * a.strips;
        CNL_REQUIRE(blocks < (1ll << 31), CNL_E_UNSUPPORTED, "cnl_decode_f32: grid too large");
        const size_t lds = (size_t)a.R * a.PXB * a.CG * 8;
        if (vec == 4) rc = launch_cminor<4>(a, P, lds, (unsigned)blocks, s);
        else if (vec == 2) rc = launch_cminor<2>(a, P, lds, (unsigned)blocks, s);
        else rc = launch_cminor<1>(a, P, lds, (unsigned)blocks, s);
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
    const size_t kst = (size_t)(((HW + TK_THREADS - 1) / TK_THREADS) | 1);
    t.keys_in_lds = kst * TK_THREADS * 4 <= 96 * 1024;
    const size_t key_bytes = t.keys_in_lds ? kst * TK_THREADS * 4 : 0;
    static cnl::DeviceOnce once;
    rc = cnl::kernel_setup(once, reinterpret_cast<const void*>(&topk_kernel), 96 * 1024);
    if (rc != CNL_OK) return rc;
    hipLaunchKernelGGL(topk_kernel, dim3(p->N), dim3(TK_THREADS), key_bytes, s, t);
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
