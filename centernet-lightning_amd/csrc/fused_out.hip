// fused_out.hip — the two small halves of a 1x1 convolution folded into the 3x3 launch before it (cnl_conv_params.fuse_w / fuse_part):
// the weight packing and the fixed-order reduction of the per-block partial sums.
//
// Call site: GenericHead's out_conv behind the last ConvBnAct block (reference models/meta.py:24-30) where out_conv has at most 4 output
// channels — the box-size head (4) and the heatmap of one- or two-class models.  As a launch of its own such a conv only streams the 256-channel
// feature map back in (C1: 537 MB, 110 us at 4.9 TB/s: HBM-bound, 1.4 % of the step) to produce 8 MB.  Folded, the row-Winograd epilogue —
// which holds every output value in registers anyway — multiplies it with the 1x1 weights and leaves [8 blocks of 32 couts][pixel][4] partial
// sums (67 MB); this file's kernel adds the blocks IN ORDER (deterministic, batch-invariant), the bias and the activation.
#include "cnl_common.h"

namespace cnl_fused {

// w_ohwi [C2][Cout] -> fw [CoutP][4], zero-padded
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ w, float* __restrict__ fw, int Cout, int C2, int CoutP) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= CoutP * 4) return;
    const int co = t >> 2, c = t & 3;
    fw[t] = (co < Cout && c < C2) ? w[(long)c * Cout + co] : 0.f;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// thread = pixel: y[pixel][0 .. C2) = act(bias + part[0][pixel] + part[1][pixel] + ...)   (16-byte loads, NB independent ones in flight)
template <int NB>
__global__ __launch_bounds__(256) void reduce_kernel(const f32x4* __restrict__ part, long M, int nblocks, int C2, const float* __restrict__ bias,
                                                     float* __restrict__ y, int ldy, unsigned flags) {
    const long m = (long)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int b0 = 0; b0 < nblocks; b0 += NB) {
        f32x4 q[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) q[j] = b0 + j < nblocks ? __builtin_nontemporal_load(part + (long)(b0 + j) * M + m) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NB; ++j) acc += q[j];
    }
    float o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float v = acc[c] + (c < C2 ? bias[c] : 0.f);
        if (flags & CNL_RELU) v = fmaxf(v, 0.f);
        if (flags & CNL_SIGMOID) v = cnl::fast_sigmoid(v);
        o[c] = v;
    }
    float* yp = y + m * ldy;
    if (C2 == 4 && (ldy & 3) == 0 && ((uintptr_t)y & 15) == 0) {
        *reinterpret_cast<f32x4*>(yp) = f32x4{o[0], o[1], o[2], o[3]};
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < C2) yp[c] = o[c];
    }
}

}  // namespace cnl_fused

extern "C" int cnl_fused_out_pack_weights_f32(const float* w_ohwi, float* fuse_w, int32_t Cout, int32_t C2, void* stream) {
    CNL_REQUIRE(w_ohwi && fuse_w, CNL_E_BAD_ARG, "cnl_fused_out_pack_weights_f32: null pointer");
    CNL_REQUIRE(Cout > 0 && C2 >= 1 && C2 <= 4, CNL_E_UNSUPPORTED, "cnl_fused_out_pack_weights_f32: 1 <= C2 <= 4 output channels (got %d)", C2);
    const int CoutP = (Cout + 63) / 64 * 64;
    hipLaunchKernelGGL(cnl_fused::pack_kernel, dim3((unsigned)((CoutP * 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_ohwi, fuse_w, Cout, C2, CoutP);
    return cnl::check_launch("fused_out pack_kernel");
}

extern "C" int cnl_fused_out_reduce_f32(const float* part, int32_t nblocks, int64_t M, int32_t C2, const float* bias, float* y, int32_t ldy,
                                        uint32_t flags, void* stream) {
    CNL_REQUIRE(part && bias && y, CNL_E_BAD_ARG, "cnl_fused_out_reduce_f32: null pointer");
    CNL_REQUIRE(nblocks > 0 && M > 0 && C2 >= 1 && C2 <= 4 && ldy >= C2, CNL_E_BAD_ARG, "cnl_fused_out_reduce_f32: nblocks, M > 0, 1 <= C2 <= 4, ldy >= C2");
    CNL_REQUIRE(((uintptr_t)part & 15) == 0, CNL_E_BAD_ARG, "cnl_fused_out_reduce_f32: part must be 16-byte aligned");
    CNL_REQUIRE(!(flags & ~(CNL_RELU | CNL_SIGMOID)), CNL_E_UNSUPPORTED, "cnl_fused_out_reduce_f32: flags other than CNL_RELU | CNL_SIGMOID");
    hipLaunchKernelGGL(cnl_fused::reduce_kernel<8>, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const cnl_fused::f32x4*>(part), (long)M, nblocks, C2, bias, y, ldy, flags);
    return cnl::check_launch("fused_out reduce_kernel");
}
