// resize.hip — the HBM-bound halves of the remaining neck options:
//   upsample2x_kernel   nn.Upsample(scale_factor=2, mode="nearest"|"bilinear") (+ the Fuse sum)   models/layers.py:99, 160-174
//   fuse_sum_kernel     the general Fuse node's sum: up to three inputs, gains, the last one resized up / max-pooled  models/layers.py:160-175
//   depthwise3x3_kernel depthwise 3x3 + folded BN + ReLU6 of conv_type="separable"                 models/layers.py:58-62
// Both stream NHWC float4s: one thread per (output pixel, 4 channels); neighbouring pixels' re-reads hit L2.
#include "cnl_common.h"

#pragma clang fp contract(off)   // ATen's bilinear kernel rounds every multiply and add

namespace cnl_resize {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void upsample2x_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                         float* __restrict__ y, int N, int H, int W, int C4, int ldx, int ldr,
                                                         int ldy, int bilinear) {
    const int Ho = 2 * H, Wo = 2 * W;
    const long total = (long)N * Ho * Wo * C4;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int c = (int)(t % C4) * 4;
        long pix = t / C4;
        const int ox = (int)(pix % Wo);
        pix /= Wo;
        const int oy = (int)(pix % Ho);
        const int n = (int)(pix / Ho);
        const float* xn = x + (long)n * H * W * ldx + c;
        f32x4 v;
        if (!bilinear) {
            v = *reinterpret_cast<const f32x4*>(xn + ((long)(oy >> 1) * W + (ox >> 1)) * ldx);
        } else {
            // align_corners=False, scale 2: src = (dst + 0.5) / 2 - 0.5 clamped at 0; lambda in {0, 0.25, 0.75}
            const float sy = fmaxf((oy + 0.5f) * 0.5f - 0.5f, 0.f), sx = fmaxf((ox + 0.5f) * 0.5f - 0.5f, 0.f);
            const int y0 = (int)sy, x0 = (int)sx;
            const int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
            const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
            const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
            const f32x4 a = *reinterpret_cast<const f32x4*>(xn + ((long)y0 * W + x0) * ldx);
            const f32x4 b = *reinterpret_cast<const f32x4*>(xn + ((long)y0 * W + x1) * ldx);
            const f32x4 cc = *reinterpret_cast<const f32x4*>(xn + ((long)y1 * W + x0) * ldx);
            const f32x4 d = *reinterpret_cast<const f32x4*>(xn + ((long)y1 * W + x1) * ldx);
            v = ly0 * (lx0 * a + lx1 * b) + ly1 * (lx0 * cc + lx1 * d);       // ATen upsample_bilinear2d order
        }
        const long opix = ((long)n * Ho + oy) * Wo + ox;
        if (res) v = *reinterpret_cast<const f32x4*>(res + opix * ldr + c) + v;   // Fuse: in1 + resize(in2)
        *reinterpret_cast<f32x4*>(y + opix * ldy + c) = v;
    }
}

__global__ __launch_bounds__(256) void depthwise3x3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, int N, int H,
                                                           int W, int C4, int ldx, int ldy, float lo, float hi) {
    const long total = (long)N * H * W * C4;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int c = (int)(t % C4) * 4;
        long pix = t / C4;
        const int ox = (int)(pix % W);
        pix /= W;
        const int oy = (int)(pix % H);
        const int n = (int)(pix / H);
        const float* xn = x + (long)n * H * W * ldx + c;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy + ky - 1;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox + kx - 1;
                if ((unsigned)ix >= (unsigned)W) continue;
                const f32x4 xv = *reinterpret_cast<const f32x4*>(xn + ((long)iy * W + ix) * ldx);
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (long)(ky * 3 + kx) * C4 * 4 + c);
                acc = acc + xv * wv;
            }
        }
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + c);
        f32x4 v = acc + bv;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fminf(fmaxf(v[i], lo), hi);
        *reinterpret_cast<f32x4*>(y + (((long)n * H + oy) * W + ox) * ldy + c) = v;
    }
}

// Fuse.forward's sum for the general node (layers.py:160-175): y = (g0 . in0 [+ g1 . in1] + gl . resize(last)) / den, the LAST input
// resized on the fly: 0 nearest x2, 1 bilinear x2 (last is H/2 x W/2), 2 MaxPool2d(2, 2) (last is 2H x 2W), 3 same size.
__global__ __launch_bounds__(256) void fuse_sum_kernel(const float* __restrict__ in0, const float* __restrict__ in1,
                                                       const float* __restrict__ last, float* __restrict__ y, int N, int H, int W, int C4,
                                                       int ld0, int ld1, int ldl, int ldy, float g0, float g1, float gl, float den,
                                                       int mode) {
    const long total = (long)N * H * W * C4;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int c = (int)(t % C4) * 4;
        const long opix = t / C4;
        const int ox = (int)(opix % W);
        const int oy = (int)((opix / W) % H);
        const int n = (int)(opix / ((long)W * H));
        f32x4 v;
        if (mode == 3) {
            v = *reinterpret_cast<const f32x4*>(last + opix * ldl + c);
        } else if (mode == 2) {
            const float* p = last + (((long)n * 2 * H + 2 * oy) * 2 * W + 2 * ox) * ldl + c;
            const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + ldl);
            const f32x4 cc = *reinterpret_cast<const f32x4*>(p + (long)2 * W * ldl), d = *reinterpret_cast<const f32x4*>(p + (long)2 * W * ldl + ldl);
#pragma unroll
            for (int i = 0; i < 4; ++i) {          // ATen max_pool2d propagates NaN
                float m = a[i];
                m = (b[i] > m || b[i] != b[i]) ? b[i] : m;
                m = (cc[i] > m || cc[i] != cc[i]) ? cc[i] : m;
                m = (d[i] > m || d[i] != d[i]) ? d[i] : m;
                v[i] = m;
            }
        } else {
            const int Hs = H >> 1, Ws = W >> 1;
            const float* xn = last + (long)n * Hs * Ws * ldl + c;
            if (mode == 0) {
                v = *reinterpret_cast<const f32x4*>(xn + ((long)(oy >> 1) * Ws + (ox >> 1)) * ldl);
            } else {
                const float sy = fmaxf((oy + 0.5f) * 0.5f - 0.5f, 0.f), sx = fmaxf((ox + 0.5f) * 0.5f - 0.5f, 0.f);
                const int y0 = (int)sy, x0 = (int)sx;
                const int y1 = y0 + (y0 < Hs - 1), x1 = x0 + (x0 < Ws - 1);
                const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
                const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
                const f32x4 a = *reinterpret_cast<const f32x4*>(xn + ((long)y0 * Ws + x0) * ldl);
                const f32x4 b = *reinterpret_cast<const f32x4*>(xn + ((long)y0 * Ws + x1) * ldl);
                const f32x4 cc = *reinterpret_cast<const f32x4*>(xn + ((long)y1 * Ws + x0) * ldl);
                const f32x4 d = *reinterpret_cast<const f32x4*>(xn + ((long)y1 * Ws + x1) * ldl);
                v = ly0 * (lx0 * a + lx1 * b) + ly1 * (lx0 * cc + lx1 * d);
            }
        }
        f32x4 acc = *reinterpret_cast<const f32x4*>(in0 + opix * ld0 + c) * g0;
        if (in1) acc = acc + *reinterpret_cast<const f32x4*>(in1 + opix * ld1 + c) * g1;
        acc = (acc + v * gl) / den;
        *reinterpret_cast<f32x4*>(y + opix * ldy + c) = acc;
    }
}

static unsigned grid_for(long total) {
    long b = (total + 255) / 256;
    if (b > 256 * 32) b = 256 * 32;
    return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace cnl_resize
using namespace cnl_resize;

extern "C" int cnl_upsample2x_nhwc_f32(const float* x, const float* residual, float* y, int32_t N, int32_t H_in, int32_t W_in,
                                       int32_t C, int32_t ldx, int32_t ldr, int32_t ldy, int32_t mode, void* stream) {
    CNL_REQUIRE(x && y, CNL_E_BAD_ARG, "cnl_upsample2x_nhwc_f32: null tensor pointer");
    CNL_REQUIRE(N > 0 && H_in > 0 && W_in > 0 && C > 0, CNL_E_BAD_ARG, "cnl_upsample2x_nhwc_f32: non-positive dimension");
    CNL_REQUIRE(mode == 0 || mode == 1, CNL_E_UNSUPPORTED, "cnl_upsample2x_nhwc_f32: mode %d (0 = nearest, 1 = bilinear)", mode);
    CNL_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= C && ldy >= C && (!residual || (ldr % 4 == 0 && ldr >= C)),
                CNL_E_UNSUPPORTED, "cnl_upsample2x_nhwc_f32: C and the pixel strides must be multiples of 4 (C=%d ldx=%d ldy=%d)", C, ldx, ldy);
    CNL_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) & 15) == 0, CNL_E_BAD_ARG, "cnl_upsample2x_nhwc_f32: 16-byte alignment");
    const long total = (long)N * 4 * H_in * W_in * (C / 4);
    hipLaunchKernelGGL(upsample2x_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, residual, y, N, H_in, W_in, C / 4,
                       ldx, ldr, ldy, mode);
    return cnl::check_launch("upsample2x_kernel");
}

extern "C" int cnl_fuse_sum_nhwc_f32(const float* in0, const float* in1, const float* last, float* y, int32_t N, int32_t H, int32_t W,
                                     int32_t C, int32_t ld0, int32_t ld1, int32_t ldl, int32_t ldy, float g0, float g1, float gl,
                                     float den, int32_t mode, void* stream) {
    CNL_REQUIRE(in0 && last && y, CNL_E_BAD_ARG, "cnl_fuse_sum_nhwc_f32: null tensor pointer");
    CNL_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, CNL_E_BAD_ARG, "cnl_fuse_sum_nhwc_f32: non-positive dimension");
    CNL_REQUIRE(mode >= 0 && mode <= 3, CNL_E_UNSUPPORTED, "cnl_fuse_sum_nhwc_f32: mode %d (0 nearest up, 1 bilinear up, 2 max-pool down, 3 none)", mode);
    CNL_REQUIRE(mode >= 2 || (H % 2 == 0 && W % 2 == 0), CNL_E_BAD_ARG, "cnl_fuse_sum_nhwc_f32: an upsampled input needs even output H, W (%dx%d)", H, W);
    CNL_REQUIRE(C % 4 == 0 && ld0 % 4 == 0 && ldl % 4 == 0 && ldy % 4 == 0 && ld0 >= C && ldl >= C && ldy >= C && (!in1 || (ld1 % 4 == 0 && ld1 >= C)),
                CNL_E_UNSUPPORTED, "cnl_fuse_sum_nhwc_f32: C and the pixel strides must be multiples of 4 (C=%d)", C);
    CNL_REQUIRE((((uintptr_t)in0 | (uintptr_t)in1 | (uintptr_t)last | (uintptr_t)y) & 15) == 0, CNL_E_BAD_ARG, "cnl_fuse_sum_nhwc_f32: 16-byte alignment");
    CNL_REQUIRE(den != 0.f, CNL_E_BAD_ARG, "cnl_fuse_sum_nhwc_f32: den == 0");
    const long total = (long)N * H * W * (C / 4);
    hipLaunchKernelGGL(fuse_sum_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in0, in1, last, y, N, H, W, C / 4, ld0, ld1,
                       ldl, ldy, g0, g1, gl, den, mode);
    return cnl::check_launch("fuse_sum_kernel");
}

extern "C" int cnl_depthwise3x3_nhwc_f32(const float* x, const float* w, const float* bias, float* y, int32_t N, int32_t H, int32_t W,
                                         int32_t C, int32_t ldx, int32_t ldy, uint32_t flags, void* stream) {
    CNL_REQUIRE(x && w && bias && y, CNL_E_BAD_ARG, "cnl_depthwise3x3_nhwc_f32: null tensor pointer");
    CNL_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, CNL_E_BAD_ARG, "cnl_depthwise3x3_nhwc_f32: non-positive dimension");
    CNL_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= C && ldy >= C, CNL_E_UNSUPPORTED,
                "cnl_depthwise3x3_nhwc_f32: C and the pixel strides must be multiples of 4 (C=%d ldx=%d ldy=%d)", C, ldx, ldy);
    CNL_REQUIRE(!(flags & ~(uint32_t)(CNL_RELU | CNL_RELU6)), CNL_E_UNSUPPORTED, "cnl_depthwise3x3_nhwc_f32: flags other than CNL_RELU / CNL_RELU6");
    CNL_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)w | (uintptr_t)bias) & 15) == 0, CNL_E_BAD_ARG, "cnl_depthwise3x3_nhwc_f32: 16-byte alignment");
    const float lo = (flags & (CNL_RELU | CNL_RELU6)) ? 0.f : -__builtin_inff();
    const float hi = (flags & CNL_RELU6) ? 6.f : __builtin_inff();
    const long total = (long)N * H * W * (C / 4);
    hipLaunchKernelGGL(depthwise3x3_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, N, H, W, C / 4, ldx,
                       ldy, lo, hi);
    return cnl::check_launch("depthwise3x3_kernel");
}

// ---- deformable convolution, sampling half (DCNv1 / v2: models/layers.py:9-38 DeformableConv2dBlock over torchvision DeformConv2d) ----
// col[n, y, x, k, :] = m_k . bilinear(x[n], y - p + ky + dy_k, x - p + kx + dx_k)   for the K*K taps k = ky*K + kx, all C channels,
// with torchvision's sampling rule (zero outside (-1, H) x (-1, W); corners outside the image contribute zero); the multiply
// with the [Cout, K*K*C] weight is then an ordinary 1x1 convolution on the matrix cores over this col tensor.
// om[n, y, x, 0 .. 2KK) = offsets (dy, dx interleaved per tap), om[.., 2KK .. 3KK) = mask LOGITS (sigmoid applied here) when has_mask.
namespace cnl_resize {
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void deform_sample_kernel(const float* __restrict__ x, const float* __restrict__ om,
                                                            float* __restrict__ col, int N, int H, int W, int C4, int ldx, int ldo,
                                                            int K, int has_mask) {
    const int KK = K * K, pad = (K - 1) / 2;
    const long total = (long)N * H * W * KK * C4;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int c = (int)(t % C4) * 4;
        long q = t / C4;
        const int k = (int)(q % KK);
        const long pix = q / KK;
        const int ox = (int)(pix % W);
        const int oy = (int)((pix / W) % H);
        const int n = (int)(pix / ((long)W * H));
        const float* o = om + pix * ldo;
        const float py = (float)(oy - pad + k / K) + o[2 * k];
        const float px = (float)(ox - pad + k % K) + o[2 * k + 1];
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (py > -1.f && py < (float)H && px > -1.f && px < (float)W) {
            const float hl = floorf(py), wl = floorf(px);
            const int h0 = (int)hl, w0 = (int)wl, h1 = h0 + 1, w1 = w0 + 1;
            const float lh = py - hl, lw = px - wl, hh = 1.f - lh, hw = 1.f - lw;
            const float* xn = x + (long)n * H * W * ldx + c;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 v1 = (h0 >= 0 && w0 >= 0) ? *reinterpret_cast<const f32x4*>(xn + ((long)h0 * W + w0) * ldx) : z;
            const f32x4 v2 = (h0 >= 0 && w1 <= W - 1) ? *reinterpret_cast<const f32x4*>(xn + ((long)h0 * W + w1) * ldx) : z;
            const f32x4 v3 = (h1 <= H - 1 && w0 >= 0) ? *reinterpret_cast<const f32x4*>(xn + ((long)h1 * W + w0) * ldx) : z;
            const f32x4 v4 = (h1 <= H - 1 && w1 <= W - 1) ? *reinterpret_cast<const f32x4*>(xn + ((long)h1 * W + w1) * ldx) : z;
            v = (hh * hw) * v1 + (hh * lw) * v2 + (lh * hw) * v3 + (lh * lw) * v4;
        }
        if (has_mask) v = v * (1.0f / (1.0f + expf(-o[2 * KK + k])));
        *reinterpret_cast<f32x4*>(col + (pix * KK + k) * (long)(C4 * 4) + c) = v;
    }
}
}  // namespace cnl_resize

extern "C" int cnl_deform_sample_nhwc_f32(const float* x, const float* om, float* col, int32_t N, int32_t H, int32_t W, int32_t C,
                                          int32_t ldx, int32_t ldo, int32_t K, int32_t has_mask, void* stream) {
    CNL_REQUIRE(x && om && col, CNL_E_BAD_ARG, "cnl_deform_sample_nhwc_f32: null tensor pointer");
    CNL_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, CNL_E_BAD_ARG, "cnl_deform_sample_nhwc_f32: non-positive dimension");
    CNL_REQUIRE(K == 1 || K == 3 || K == 5, CNL_E_UNSUPPORTED, "cnl_deform_sample_nhwc_f32: kernel size %d (1, 3 or 5)", K);
    CNL_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldx >= C && ldo >= (has_mask ? 3 : 2) * K * K, CNL_E_UNSUPPORTED,
                "cnl_deform_sample_nhwc_f32: C %% 4 != 0 or pixel strides too small (C=%d ldx=%d ldo=%d)", C, ldx, ldo);
    CNL_REQUIRE((((uintptr_t)x | (uintptr_t)col) & 15) == 0, CNL_E_BAD_ARG, "cnl_deform_sample_nhwc_f32: 16-byte alignment");
    const long total = (long)N * H * W * K * K * (C / 4);
    hipLaunchKernelGGL(cnl_resize::deform_sample_kernel, dim3(cnl_resize::grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, om, col,
                       N, H, W, C / 4, ldx, ldo, K, has_mask);
    return cnl::check_launch("deform_sample_kernel");
}
