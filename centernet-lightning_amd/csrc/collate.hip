// collate.hip — pack / unpack of the fixed-size detection record that travels through the RCCL
// all-gather (one float32 row per detection: x1 y1 x2 y2 score label-bits [embedding...]).
//
// Replaces the pickled `dist.all_gather_object` of the reference (eval/coco.py:10-18): equal counts on
// every rank (fixed k) make a plain ncclAllGather of this buffer sufficient.
#include "cnl_common.h"

namespace cnl_collate {

__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                   const long long* __restrict__ labels, const float* __restrict__ emb,
                                                   float* __restrict__ rec, long D, int E) {
    const int R = 6 + E;
    const long total = D * R;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const long d = t / R;
        const int f = (int)(t - d * R);
        float v;
        if (f < 4) v = boxes[d * 4 + f];
        else if (f == 4) v = scores[d];
        else if (f == 5) v = __int_as_float((int)labels[d]);
        else v = emb[d * E + (f - 6)];
        rec[t] = v;
    }
}

__global__ __launch_bounds__(256) void unpack_kernel(const float* __restrict__ rec, float* __restrict__ boxes,
                                                     float* __restrict__ scores, long long* __restrict__ labels,
                                                     float* __restrict__ emb, long D, int E) {
    const int R = 6 + E;
    const long total = D * R;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const long d = t / R;
        const int f = (int)(t - d * R);
        const float v = rec[t];
        if (f < 4) boxes[d * 4 + f] = v;
        else if (f == 4) scores[d] = v;
        else if (f == 5) labels[d] = (long long)__float_as_int(v);
        else emb[d * E + (f - 6)] = v;
    }
}

unsigned grid_for(long total) {
    long b = (total + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace cnl_collate
using namespace cnl_collate;

extern "C" int cnl_pack_detections_f32(const float* boxes, const float* scores, const int64_t* labels, const float* emb,
                                       float* rec, int32_t N, int32_t k, int32_t E, void* stream) {
    CNL_REQUIRE(boxes && scores && labels && rec, CNL_E_BAD_ARG, "cnl_pack_detections_f32: null tensor pointer");
    CNL_REQUIRE(N > 0 && k > 0 && E >= 0, CNL_E_BAD_ARG, "cnl_pack_detections_f32: bad N/k/E");
    CNL_REQUIRE(E == 0 || emb, CNL_E_BAD_ARG, "cnl_pack_detections_f32: E > 0 without embeddings");
    const long D = (long)N * k;
    hipLaunchKernelGGL(pack_kernel, dim3(grid_for(D * (6 + E))), dim3(256), 0, (hipStream_t)stream, boxes, scores,
                       (const long long*)labels, emb, rec, D, E);
    return cnl::check_launch("pack_kernel");
}

extern "C" int cnl_unpack_detections_f32(const float* rec, float* boxes, float* scores, int64_t* labels, float* emb,
                                         int32_t N, int32_t k, int32_t E, void* stream) {
    CNL_REQUIRE(boxes && scores && labels && rec, CNL_E_BAD_ARG, "cnl_unpack_detections_f32: null tensor pointer");
    CNL_REQUIRE(N > 0 && k > 0 && E >= 0, CNL_E_BAD_ARG, "cnl_unpack_detections_f32: bad N/k/E");
    CNL_REQUIRE(E == 0 || emb, CNL_E_BAD_ARG, "cnl_unpack_detections_f32: E > 0 without embeddings");
    const long D = (long)N * k;
    hipLaunchKernelGGL(unpack_kernel, dim3(grid_for(D * (6 + E))), dim3(256), 0, (hipStream_t)stream, rec, boxes, scores,
                       (long long*)labels, emb, D, E);
    return cnl::check_launch("unpack_kernel");
}

// torchvision.ops.box_convert(boxes, "xyxy", "xywh") as called at models/centernet.py:207: (x1, y1, x2 - x1, y2 - y1).
namespace cnl_collate {
__global__ __launch_bounds__(256) void xyxy_to_xywh_kernel(const float4* __restrict__ in, float4* __restrict__ out, long n) {
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long)gridDim.x * 256) {
        const float4 b = in[t];
        out[t] = make_float4(b.x, b.y, b.z - b.x, b.w - b.y);
    }
}
}  // namespace cnl_collate

extern "C" int cnl_boxes_xyxy_to_xywh_f32(const float* boxes, float* out, int64_t n, void* stream) {
    CNL_REQUIRE(n >= 0, CNL_E_BAD_ARG, "cnl_boxes_xyxy_to_xywh_f32: negative count");
    if (n == 0) return CNL_OK;
    CNL_REQUIRE(boxes && out, CNL_E_BAD_ARG, "cnl_boxes_xyxy_to_xywh_f32: null tensor pointer");
    CNL_REQUIRE((((uintptr_t)boxes | (uintptr_t)out) & 15) == 0, CNL_E_BAD_ARG, "cnl_boxes_xyxy_to_xywh_f32: 16-byte alignment");
    hipLaunchKernelGGL(cnl_collate::xyxy_to_xywh_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(boxes), reinterpret_cast<float4*>(out), (long)n);
    return cnl::check_launch("xyxy_to_xywh_kernel");
}
