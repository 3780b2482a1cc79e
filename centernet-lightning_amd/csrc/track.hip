// track.hip — the tracker's association costs and track-table update, so that the k x 64
// embeddings of every frame stay in HBM: only the small n x T cost matrices travel to the host for the Hungarian step
// (scipy, as in the reference) and the match list travels back.
//
// Replaces, per frame, of centernet_lightning/models/tracker.py:
//   :133-137  detection-threshold mask + boolean-mask compaction of bboxes / labels / embeddings
//   :150      reid_cost = scipy cdist(det_embeddings, track_embeddings, "cosine")                 -> float64 [n, T]
//   :162      box_cost  = box_iou_distance_matrix | box_giou_distance_matrix (utils/box.py:49-92) -> float32 [n, T]
//   :228,320-321  Track.embedding = e/|e| ; (1-s)*old + s*e/|e|  and  Track.bbox = bbox (use_kalman=False)
//
// Arithmetic mirrors the CPU libraries operation by operation (no fma contraction) so that the integer decisions taken
// from the costs (thresholds, Hungarian assignment) are the reference's: the cosine distance is evaluated in float64 with the
// sequential dot product, the |cos| <= 1 clamp and 1 - cos of scipy's cdist; the box distances in float32 in numpy's order.
#include "cnl_common.h"

#pragma clang fp contract(off)   // one rounding per operation, like numpy / scipy: the costs feed thresholds and the Hungarian step

namespace cnl_track {

constexpr int MAXK = 1024;     // detections per frame (decode's k limit)

// numpy maximum/minimum propagate NaN (fmaxf/fminf do not)
__device__ __forceinline__ float np_max(float a, float b) { return (a != a) ? a : (b != b) ? b : (a > b ? a : b); }
__device__ __forceinline__ float np_min(float a, float b) { return (a != a) ? a : (b != b) ? b : (a < b ? a : b); }

// u.u, v.v and u.v in float64, each accumulated sequentially over e = 0..E-1 (scipy's dot order); the three chains are
// independent, so interleaving them hides the fp64 add latency without changing any result.
__device__ __forceinline__ void seq_dots(const float* __restrict__ u, const float* __restrict__ v, int E, double& uu, double& vv,
                                         double& uv) {
    uu = vv = uv = 0.0;
    int e = 0;
    if ((E & 3) == 0) {      // rows are 16-byte aligned when E % 4 == 0
        for (; e < E; e += 4) {
            const float4 a = *reinterpret_cast<const float4*>(u + e), b = *reinterpret_cast<const float4*>(v + e);
            const double a0 = a.x, a1 = a.y, a2 = a.z, a3 = a.w, b0 = b.x, b1 = b.y, b2 = b.z, b3 = b.w;
            uu = uu + a0 * a0; vv = vv + b0 * b0; uv = uv + a0 * b0;
            uu = uu + a1 * a1; vv = vv + b1 * b1; uv = uv + a1 * b1;
            uu = uu + a2 * a2; vv = vv + b2 * b2; uv = uv + a2 * b2;
            uu = uu + a3 * a3; vv = vv + b3 * b3; uv = uv + a3 * b3;
        }
    }
    for (; e < E; ++e) {
        const double a = u[e], b = v[e];
        uu = uu + a * a; vv = vv + b * b; uv = uv + a * b;
    }
}

// The (tiny) stable compaction {i : score[i] >= thr} into LDS, rebuilt by every workgroup — cheaper than a second launch.  Returns n.
__device__ __forceinline__ int compact_scores(const float* __restrict__ det_score, const int k, const float thr, int* sel, int* wave_sum) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // thread t owns scores 4t..4t+3 (k <= 1024)
    int flag[4], cnt = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid * 4 + i;
        flag[i] = idx < k && det_score[idx] >= thr;         // NaN compares false, as in numpy
        cnt += flag[i];
    }
    int incl = cnt;                                          // inclusive scan of cnt over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    int base = 0, n = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < wave) base += wave_sum[w];
        n += wave_sum[w];
    }
    int pos = base + incl - cnt;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (flag[i]) sel[pos++] = tid * 4 + i;
    __syncthreads();
    return n;
}

// Row mean in float64 in numpy's order (np.add.reduce over a contiguous last axis = pairwise summation: eight strided partial sums per block of <= 128 elements,
// combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), the tail added sequentially; longer rows split in halves rounded down to a multiple of 8) — scipy's cdist
// "correlation" centres both operands with XA.mean(axis=1) before its cosine kernel.
__device__ double np_pairwise_sum(const float* a, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res = res + (double)a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (double)a[j];
        int i = 8;
        for (; i < n - (n % 8); i += 8)
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = r[j] + (double)a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res = res + (double)a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

__device__ __forceinline__ void pair_costs(const int* sel, const int n, const float* __restrict__ det_emb, const float* __restrict__ det_box, const int E,
                                           const float* __restrict__ trk_emb, const float* __restrict__ trk_box, const int T, const int box_mode,
                                           const int reid_metric, double* __restrict__ reid_cost, float* __restrict__ box_cost) {
    const int tid = threadIdx.x;
    const long p = (long)blockIdx.x * 256 + tid;
    if (T <= 0 || p >= (long)n * T) return;
    const int r = (int)(p / T), t = (int)(p - (long)r * T);
    const int d = sel[r];
    if (reid_metric == 0) {   // scipy cdist "cosine": 1 - clamp(u.v / (|u| |v|))
        const float* u = det_emb + (long)d * E;
        const float* v = trk_emb + (long)t * E;
        double uu, vv, uv;
        seq_dots(u, v, E, uu, vv, uv);
        double c = uv / (sqrt(uu) * sqrt(vv));
        if (fabs(c) > 1.0) c = copysign(1.0, c);
        reid_cost[p] = (1.0 - c);
    } else if (reid_metric == 7) {   // scipy cdist "correlation": both rows centred by their float64 mean (numpy's pairwise order), then the cosine kernel
        const float* u = det_emb + (long)d * E;
        const float* v = trk_emb + (long)t * E;
        const double mu = np_pairwise_sum(u, E) / (double)E, mv = np_pairwise_sum(v, E) / (double)E;
        double uu = 0.0, vv = 0.0, uv = 0.0;
        for (int e = 0; e < E; ++e) {
            const double x = (double)u[e] - mu, y = (double)v[e] - mv;
            uu = uu + x * x; vv = vv + y * y; uv = uv + x * y;
        }
        double c = uv / (sqrt(uu) * sqrt(vv));
        if (fabs(c) > 1.0) c = copysign(1.0, c);
        reid_cost[p] = (1.0 - c);
    } else if (reid_metric <= 2) {   // scipy cdist "euclidean" (1) / "sqeuclidean" (2): s += (u - v)^2 sequentially in float64, then sqrt
        const float* u = det_emb + (long)d * E;
        const float* v = trk_emb + (long)t * E;
        double acc = 0.0;
        for (int e = 0; e < E; ++e) {
            const double df = (double)u[e] - (double)v[e];
            acc = acc + df * df;
        }
        reid_cost[p] = reid_metric == 1 ? sqrt(acc) : acc;
    } else {                  // "cityblock" (3), "chebyshev" (4), "canberra" (5), "braycurtis" (6): scipy's element order, float64
        const float* u = det_emb + (long)d * E;
        const float* v = trk_emb + (long)t * E;
        double acc = 0.0, den = 0.0;
        for (int e = 0; e < E; ++e) {
            const double x = (double)u[e], y = (double)v[e];
            const double ad = fabs(x - y);
            if (reid_metric == 3) acc = acc + ad;
            else if (reid_metric == 4) acc = ad > acc ? ad : acc;
            else if (reid_metric == 5) {
                const double q = fabs(x) + fabs(y);
                acc = acc + ad / (q + (q == 0.0 ? 1.0 : 0.0));           // 0 / 0 counts as 0
            } else {
                acc = acc + ad;
                den = den + fabs(x + y);
            }
        }
        reid_cost[p] = reid_metric == 6 ? acc / den : acc;
    }
    if (box_mode) {   // utils/box.py:49-92 in float32, numpy's operation order
        const float4 a = *reinterpret_cast<const float4*>(det_box + (long)d * 4);
        const float4 b = *reinterpret_cast<const float4*>(trk_box + (long)t * 4);
        const float area1 = ((a.z - a.x) * (a.w - a.y));
        const float area2 = ((b.z - b.x) * (b.w - b.y));
        const float w = np_max((np_min(a.z, b.z) - np_max(a.x, b.x)), 0.f);
        const float h = np_max((np_min(a.w, b.w) - np_max(a.y, b.y)), 0.f);
        const float inter = (w * h);
        const float uni = ((area1 + area2) - inter);
        const float iou = (inter / uni);
        float score = iou;
        if (box_mode == 2) {
            const float wi = np_max((np_max(a.z, b.z) - np_min(a.x, b.x)), 0.f);
            const float hi = np_max((np_max(a.w, b.w) - np_min(a.y, b.y)), 0.f);
            const float hull = (wi * hi);
            score = (iou - ((hull - uni) / hull));
        }
        box_cost[p] = (1.f - score);
    }
}

// Every workgroup handles 256 (detection, track) pairs; workgroup 0 also publishes n_det / det_index.
__global__ __launch_bounds__(256) void costs_kernel(const float* __restrict__ det_emb, const float* __restrict__ det_box,
                                                    const float* __restrict__ det_score, int k, int E, float thr,
                                                    const float* __restrict__ trk_emb, const float* __restrict__ trk_box, int T,
                                                    int box_mode, int reid_metric, int* __restrict__ n_det, int* __restrict__ det_index,
                                                    double* __restrict__ reid_cost, float* __restrict__ box_cost) {
    __shared__ int sel[MAXK];
    __shared__ int wave_sum[4];
    const int n = compact_scores(det_score, k, thr, sel, wave_sum);
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) *n_det = n;
        for (int i = threadIdx.x; i < n; i += 256) det_index[i] = sel[i];
    }
    pair_costs(sel, n, det_emb, det_box, E, trk_emb, trk_box, T, box_mode, reid_metric, reid_cost, box_cost);
}

// The same, writing ONE self-describing record (include/centernet_gfx950.h: cnl_track_frame_f32) whose cost matrices are packed by the
// n the kernel itself found — so the host needs no copy of the scores before the launch, and the record may live in page-locked host
// memory mapped into the device's address space: the frame's only device -> host traffic is these stores, no copy engine involved.
__global__ __launch_bounds__(256) void frame_kernel(const float* __restrict__ det_emb, const float* __restrict__ det_box,
                                                    const float* __restrict__ det_score, const void* __restrict__ det_label, int label_kind,
                                                    int k, int E, float thr, const float* __restrict__ trk_emb,
                                                    const float* __restrict__ trk_box, int T, int box_mode, int reid_metric, int with_dets,
                                                    char* __restrict__ record) {
    __shared__ int sel[MAXK];
    __shared__ int wave_sum[4];
    const int n = compact_scores(det_score, k, thr, sel, wave_sum);
    const int off_index = 32, off_dets = (off_index + 4 * k + 7) & ~7, off_reid = (off_dets + (with_dets ? 24 * k : 0) + 7) & ~7;
    const long off_box = off_reid + 8l * n * T;
    if (blockIdx.x == 0) {
        const int tid = threadIdx.x;
        int* hdr = reinterpret_cast<int*>(record);
        if (tid == 0) { hdr[0] = n; hdr[1] = k; hdr[2] = T; hdr[3] = with_dets; hdr[4] = off_index; hdr[5] = off_dets; hdr[6] = off_reid; hdr[7] = (int)off_box; }
        int* det_index = reinterpret_cast<int*>(record + off_index);
        for (int i = tid; i < n; i += 256) det_index[i] = sel[i];
        if (with_dets) {      // the frame's boxes / scores / labels, which the host-side life cycle reads (tracker.py:171-186)
            float* o = reinterpret_cast<float*>(record + off_dets);
            for (int i = tid; i < 4 * k; i += 256) o[i] = det_box[i];
            for (int i = tid; i < k; i += 256) o[4 * k + i] = det_score[i];
            int* ol = reinterpret_cast<int*>(o + 5 * k);
            for (int i = tid; i < k; i += 256)
                ol[i] = label_kind == 1 ? (int)reinterpret_cast<const long long*>(det_label)[i]
                      : label_kind == 2 ? reinterpret_cast<const int*>(det_label)[i]
                      : label_kind == 3 ? (int)reinterpret_cast<const float*>(det_label)[i] : 0;
        }
    }
    pair_costs(sel, n, det_emb, det_box, E, trk_emb, trk_box, T, box_mode, reid_metric, reinterpret_cast<double*>(record + off_reid),
               reinterpret_cast<float*>(record + off_box));
}

// One wave per row of the new track table.
__global__ __launch_bounds__(64) void apply_kernel(const float* __restrict__ trk_emb, const float* __restrict__ trk_box,
                                                   const float* __restrict__ det_emb, const float* __restrict__ det_box,
                                                   const int* __restrict__ src_trk, const int* __restrict__ src_det, int E,
                                                   float keep, float blend, float* __restrict__ new_emb,
                                                   float* __restrict__ new_box) {
    const int r = blockIdx.x, lane = threadIdx.x;
    const int t = src_trk[r], d = src_det[r];
    float inv_den = 0.f;
    if (d >= 0) {      // |e| = sqrt(sum e^2) in float32 (np.linalg.norm of a float32 vector)
        float s = 0.f;
        for (int e = lane; e < E; e += 64) {
            const float x = det_emb[(long)d * E + e];
            s += x * x;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        inv_den = sqrtf(s);
    }
    for (int e = lane; e < E; e += 64) {
        float v;
        if (d < 0) v = trk_emb[(long)t * E + e];
        else {
            const float unit = (det_emb[(long)d * E + e] / inv_den);
            v = t < 0 ? unit : ((keep * trk_emb[(long)t * E + e]) + (blend * unit));
        }
        new_emb[(long)r * E + e] = v;
    }
    if (lane < 4) new_box[(long)r * 4 + lane] = d >= 0 ? det_box[(long)d * 4 + lane] : trk_box[(long)t * 4 + lane];
}

}  // namespace cnl_track
using namespace cnl_track;

extern "C" int cnl_track_costs_metric_f32(const float* det_emb, const float* det_box, const float* det_score, int32_t k, int32_t E,
                                          float detection_threshold, const float* trk_emb, const float* trk_box, int32_t T,
                                          int32_t box_cost, int32_t reid_metric, int32_t* n_det, int32_t* det_index, double* reid_cost,
                                          float* box_cost_out, void* stream) {
    CNL_REQUIRE(det_emb && det_box && det_score && n_det && det_index, CNL_E_BAD_ARG, "cnl_track_costs_f32: null pointer");
    CNL_REQUIRE(k > 0 && E > 0 && T >= 0, CNL_E_BAD_ARG, "cnl_track_costs_f32: bad k/E/T");
    CNL_REQUIRE(k <= MAXK, CNL_E_UNSUPPORTED, "cnl_track_costs_f32: k = %d > %d detections per frame", k, MAXK);
    CNL_REQUIRE(box_cost >= 0 && box_cost <= 2, CNL_E_BAD_ARG, "cnl_track_costs_f32: box_cost must be 0 (none), 1 (iou), 2 (giou)");
    CNL_REQUIRE(reid_metric >= 0 && reid_metric <= 7, CNL_E_BAD_ARG, "cnl_track_costs_metric_f32: reid_metric must be 0 (cosine), 1 (euclidean), 2 (sqeuclidean), 3 (cityblock), 4 (chebyshev), 5 (canberra), 6 (braycurtis) or 7 (correlation)");
    CNL_REQUIRE(T == 0 || (trk_emb && reid_cost), CNL_E_BAD_ARG, "cnl_track_costs_f32: T > 0 without track table / reid_cost");
    CNL_REQUIRE(T == 0 || box_cost == 0 || (trk_box && box_cost_out), CNL_E_BAD_ARG,
                "cnl_track_costs_f32: box cost requested without track boxes / output");
    const long pairs = (long)k * T;
    const unsigned grid = (unsigned)(pairs > 0 ? (pairs + 255) / 256 : 1);
    hipLaunchKernelGGL(costs_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, det_emb, det_box, det_score, k, E,
                       detection_threshold, trk_emb, trk_box, T, box_cost, reid_metric, n_det, det_index, reid_cost, box_cost_out);
    return cnl::check_launch("track costs_kernel");
}

extern "C" int64_t cnl_track_frame_bytes(int32_t k, int32_t T, int32_t with_detections) {
    if (k <= 0 || T < 0) return 0;
    const int64_t off_dets = (32 + 4l * k + 7) & ~7l, off_reid = (off_dets + (with_detections ? 24l * k : 0) + 7) & ~7l;
    return off_reid + 12l * k * T;                  // worst case n = k
}

extern "C" int cnl_track_frame_f32(const float* det_emb, const float* det_box, const float* det_score, const void* det_label, int32_t label_kind,
                                   int32_t k, int32_t E, float detection_threshold, const float* trk_emb, const float* trk_box, int32_t T,
                                   int32_t box_cost, int32_t reid_metric, int32_t with_detections, void* record, int64_t record_bytes,
                                   void* stream) {
    CNL_REQUIRE(det_emb && det_box && det_score && record, CNL_E_BAD_ARG, "cnl_track_frame_f32: null pointer");
    CNL_REQUIRE(k > 0 && E > 0 && T >= 0, CNL_E_BAD_ARG, "cnl_track_frame_f32: bad k/E/T");
    CNL_REQUIRE(k <= MAXK, CNL_E_UNSUPPORTED, "cnl_track_frame_f32: k = %d > %d detections per frame", k, MAXK);
    CNL_REQUIRE(box_cost >= 0 && box_cost <= 2, CNL_E_BAD_ARG, "cnl_track_frame_f32: box_cost must be 0 (none), 1 (iou), 2 (giou)");
    CNL_REQUIRE(reid_metric >= 0 && reid_metric <= 7, CNL_E_BAD_ARG, "cnl_track_frame_f32: reid_metric must be 0 (cosine) .. 7 (correlation)");
    CNL_REQUIRE(label_kind >= 0 && label_kind <= 3 && (label_kind == 0 || det_label), CNL_E_BAD_ARG,
                "cnl_track_frame_f32: label_kind must be 0 (none), 1 (int64), 2 (int32), 3 (float32) with det_label set");
    CNL_REQUIRE(T == 0 || trk_emb, CNL_E_BAD_ARG, "cnl_track_frame_f32: T > 0 without track table");
    CNL_REQUIRE(T == 0 || box_cost == 0 || trk_box, CNL_E_BAD_ARG, "cnl_track_frame_f32: box cost requested without track boxes");
    CNL_REQUIRE(((uintptr_t)record & 7) == 0, CNL_E_BAD_ARG, "cnl_track_frame_f32: record must be 8-byte aligned");
    const int64_t need = cnl_track_frame_bytes(k, T, with_detections);
    CNL_REQUIRE(record_bytes >= need && need < (1l << 31), record_bytes < need ? CNL_E_BAD_ARG : CNL_E_UNSUPPORTED,
                "cnl_track_frame_f32: record holds %ld bytes, k = %d, T = %d needs %ld (cnl_track_frame_bytes; below 2 GiB)", (long)record_bytes, k, T, (long)need);
    const long pairs = (long)k * T;
    const unsigned grid = (unsigned)(pairs > 0 ? (pairs + 255) / 256 : 1);
    hipLaunchKernelGGL(frame_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, det_emb, det_box, det_score, det_label, label_kind, k, E,
                       detection_threshold, trk_emb, trk_box, T, box_cost == 0 ? 0 : box_cost, reid_metric, with_detections ? 1 : 0, (char*)record);
    return cnl::check_launch("track frame_kernel");
}

extern "C" int cnl_track_costs_f32(const float* det_emb, const float* det_box, const float* det_score, int32_t k, int32_t E,
                                   float detection_threshold, const float* trk_emb, const float* trk_box, int32_t T,
                                   int32_t box_cost, int32_t* n_det, int32_t* det_index, double* reid_cost,
                                   float* box_cost_out, void* stream) {
    return cnl_track_costs_metric_f32(det_emb, det_box, det_score, k, E, detection_threshold, trk_emb, trk_box, T, box_cost, 0, n_det, det_index,
                                      reid_cost, box_cost_out, stream);
}

extern "C" int cnl_track_apply_f32(const float* trk_emb, const float* trk_box, const float* det_emb, const float* det_box,
                                   const int32_t* src_trk, const int32_t* src_det, int32_t T_new, int32_t E, double smoothing,
                                   float* new_emb, float* new_box, void* stream) {
    CNL_REQUIRE(T_new >= 0 && E > 0, CNL_E_BAD_ARG, "cnl_track_apply_f32: bad T_new/E");
    if (T_new == 0) return CNL_OK;
    CNL_REQUIRE(det_emb && det_box && src_trk && src_det && new_emb && new_box, CNL_E_BAD_ARG, "cnl_track_apply_f32: null pointer");
    CNL_REQUIRE(new_emb != trk_emb && new_box != trk_box, CNL_E_BAD_ARG, "cnl_track_apply_f32: the new table must not alias the old one");
    // tracker.py:321: (1 - s) and s are Python floats that numpy applies to float32 arrays as float32 scalars
    const float keep = (float)(1.0 - smoothing), blend = (float)smoothing;
    hipLaunchKernelGGL(apply_kernel, dim3((unsigned)T_new), dim3(64), 0, (hipStream_t)stream, trk_emb, trk_box, det_emb, det_box,
                       src_trk, src_det, E, keep, blend, new_emb, new_box);
    return cnl::check_launch("track apply_kernel");
}
