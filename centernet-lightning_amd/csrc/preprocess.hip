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

// cv2.resize(..., interpolation=cv2.INTER_LINEAR) on 8-bit images = albumentations A.Resize (README.md:84; datasets/inference.py).
// OpenCV's 8-bit path is fixed point (third-party, absent here; its published algorithm — imgproc/resize.cpp, HResizeLinear /
// VResizeLinear<uchar,int,short> with INTER_RESIZE_COEF_BITS = 11 — is restated):
//   fx = float((dx + 0.5) * scale_x - 0.5); sx = floor(fx); fx -= sx; (sx < 0: fx = 0, sx = 0;  sx >= W-1: fx = 0, sx = W-1)
//   a = (round_half_even((1 - fx) * 2048), round_half_even(fx * 2048)) as int16;   rows alike with fy (no clamp of fy: the two
//   source rows are clipped to the image instead), then
//   D[row][dx] = S[row][sx] * a0 + S[row][min(sx+1, W-1)] * a1                       (int32)
//   dst = (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2              (uint8)
// Half-pixel centres; bit-exact against the numpy oracle (oracle/decode_ref.resize_bilinear_u8).  HBM-bound, thread = output pixel.
__global__ __launch_bounds__(256) void resize_bilinear_u8_kernel(const unsigned char* __restrict__ x, unsigned char* __restrict__ y,
                                                                 int N, int Hi, int Wi, int Ho, int Wo, int C, double scale_x, double scale_y) {
    const long total = (long)N * Ho * Wo;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int dx = (int)(t % Wo);
        const long r_ = t / Wo;
        const int dy = (int)(r_ % Ho), n = (int)(r_ / Ho);
        float fx = (float)(((double)dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= (float)sx;
        if (sx < 0) { fx = 0.f; sx = 0; }
        if (sx >= Wi - 1) { fx = 0.f; sx = Wi - 1; }
        const int a0 = (short)__float2int_rn((1.f - fx) * 2048.f), a1 = (short)__float2int_rn(fx * 2048.f);
        float fy = (float)(((double)dy + 0.5) * scale_y - 0.5);
        const int sy = (int)floorf(fy);
        fy -= (float)sy;
        const int b0 = (short)__float2int_rn((1.f - fy) * 2048.f), b1 = (short)__float2int_rn(fy * 2048.f);
        const int y0 = min(max(sy, 0), Hi - 1), y1 = min(max(sy + 1, 0), Hi - 1);
        const int x1 = min(sx + 1, Wi - 1);
        const unsigned char* r0 = x + ((long)n * Hi + y0) * Wi * C;
        const unsigned char* r1 = x + ((long)n * Hi + y1) * Wi * C;
        unsigned char* dst = y + t * C;
        for (int c = 0; c < C; ++c) {
            const int d0 = (int)r0[sx * C + c] * a0 + (int)r0[x1 * C + c] * a1;
            const int d1 = (int)r1[sx * C + c] * a0 + (int)r1[x1 * C + c] * a1;
            const int v = (((b0 * (d0 >> 4)) >> 16) + ((b1 * (d1 >> 4)) >> 16) + 2) >> 2;
            dst[c] = (unsigned char)min(max(v, 0), 255);
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

extern "C" int cnl_resize_bilinear_u8(const uint8_t* x, uint8_t* y, int32_t N, int32_t H_in, int32_t W_in, int32_t H_out, int32_t W_out,
                                      int32_t C, void* stream) {
    CNL_REQUIRE(x && y, CNL_E_BAD_ARG, "cnl_resize_bilinear_u8: null pointer");
    CNL_REQUIRE(N > 0 && H_in > 0 && W_in > 0 && H_out > 0 && W_out > 0 && C > 0 && C <= 4, CNL_E_BAD_ARG,
                "cnl_resize_bilinear_u8: non-positive dimension or C > 4");
    // OpenCV: inv_scale = dsize / ssize (double), scale = 1 / inv_scale
    const double scale_x = 1.0 / ((double)W_out / (double)W_in), scale_y = 1.0 / ((double)H_out / (double)H_in);
    const long total = (long)N * H_out * W_out;
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(cnl_pre::resize_bilinear_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, N, H_in, W_in,
                       H_out, W_out, C, scale_x, scale_y);
    return cnl::check_launch("resize_bilinear_u8_kernel");
}
