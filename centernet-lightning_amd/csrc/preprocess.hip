// preprocess.hip — step BEFORE the hot path (SURVEY.md §8f "next" #2): uint8 HWC camera/cv2 frames -> normalised fp32 NHWC.
//
// Replaces albumentations `A.Normalize(mean, std)` + `ToTensorV2()` of the reference's inference pre-processing
// (README.md:79-87, datasets/utils.py:9-21, datasets/inference.py:26-39) on the GPU: uploading uint8 is 4x less PCIe
// traffic than fp32, and the result is consumed zero-copy by cnl_stem_conv7x7_f32 through its explicit strides
// (a logical-NCHW view of NHWC storage).  Arithmetic follows albumentations' normalize (third-party, absent from the
// reference tree; algorithm restated):  out = (float(x) - mean*255) * (1 / (std*255)),  all in float32, one rounding
// per operation — bit-exact against the numpy oracle.  HBM-bound: 3 B read + 12 B written per pixel.
#include "cnl_common.h"

#pragma clang fp contract(off)

namespace cnl_pre {

__global__ __launch_bounds__(256) void normalize_u8_kernel(const unsigned char* __restrict__ x, float* __restrict__ y, long n_px,
                                                           float m0, float m1, float m2, float r0, float r1, float r2) {
    // thread -> 4 pixels = 12 bytes in (three 32-bit loads), 12 floats out (three 16-byte stores)
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q * 4 < n_px; q += (long)gridDim.x * 256) {
        const long p0 = q * 4;
        if (p0 + 4 <= n_px) {
            const unsigned* src = reinterpret_cast<const unsigned*>(x + p0 * 3);
            const unsigned a = src[0], b = src[1], c = src[2];
            float v[12];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = (float)((a >> (8 * i)) & 255u);
                v[4 + i] = (float)((b >> (8 * i)) & 255u);
                v[8 + i] = (float)((c >> (8 * i)) & 255u);
            }
            const float m[3] = {m0, m1, m2}, r[3] = {r0, r1, r2};
#pragma unroll
            for (int i = 0; i < 12; ++i) v[i] = (v[i] - m[i % 3]) * r[i % 3];
            float4* dst = reinterpret_cast<float4*>(y + p0 * 3);
            dst[0] = make_float4(v[0], v[1], v[2], v[3]);
            dst[1] = make_float4(v[4], v[5], v[6], v[7]);
            dst[2] = make_float4(v[8], v[9], v[10], v[11]);
        } else {
            for (long p = p0; p < n_px; ++p) {
                y[p * 3 + 0] = ((float)x[p * 3 + 0] - m0) * r0;
                y[p * 3 + 1] = ((float)x[p * 3 + 1] - m1) * r1;
                y[p * 3 + 2] = ((float)x[p * 3 + 2] - m2) * r2;
            }
        }
    }
}

}  // namespace cnl_pre

extern "C" int cnl_normalize_u8_nhwc_f32(const uint8_t* x, float* y, int32_t N, int32_t H, int32_t W, const float* mean255,
                                         const float* inv_std255, void* stream) {
    CNL_REQUIRE(x && y && mean255 && inv_std255, CNL_E_BAD_ARG, "cnl_normalize_u8_nhwc_f32: null pointer");
    CNL_REQUIRE(N > 0 && H > 0 && W > 0, CNL_E_BAD_ARG, "cnl_normalize_u8_nhwc_f32: non-positive dimension");
    CNL_REQUIRE(((uintptr_t)x & 3) == 0 && ((uintptr_t)y & 15) == 0, CNL_E_BAD_ARG, "cnl_normalize_u8_nhwc_f32: x must be 4-byte and y 16-byte aligned");
    const long n_px = (long)N * H * W;
    long blocks = (n_px / 4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(cnl_pre::normalize_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, n_px, mean255[0],
                       mean255[1], mean255[2], inv_std255[0], inv_std255[1], inv_std255[2]);
    return cnl::check_launch("normalize_u8_kernel");
}
