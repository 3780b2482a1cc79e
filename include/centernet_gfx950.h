/*
 * centernet_gfx950.h — C ABI of libcenternet_gfx950.so
 *
 * MI355X-native (gfx950 / CDNA4) replacement for the CenterNet inference hot path of
 * gau-nernst/centernet-lightning.  The reference is pure Python dispatching to ATen/cuDNN ops; every
 * entry point below replaces the ATen call sites listed beside it (paths relative to the reference
 * root).  The reference-side binding is a ctypes stub.
 *
 * Conventions
 *   - all tensor pointers are DEVICE pointers (HBM); all tensors are fp32 unless stated;
 *   - activations are NHWC ("channels_last"): element (n,y,x,c) lives at ((n*H+y)*W+x)*ld + c where
 *     ld >= C is the pixel stride in elements (lets several heads share one buffer);
 *   - conv weights are OHWI: w[co][ky][kx][ci], K = KH*KW*Cin contiguous per output channel, with the
 *     eval-mode BatchNorm already folded in (scale into w, shift into bias);
 *   - every call is asynchronous on the hipStream_t passed as `stream` (void* to keep HIP headers out
 *     of the binding), allocates nothing, and is re-entrant;
 *   - return value: 0 = CNL_OK, negative = CNL_E_*; cnl_last_error() returns the message of the last
 *     failure on the calling thread.  No C++ exception crosses this boundary.
 */
#ifndef CENTERNET_GFX950_H
#define CENTERNET_GFX950_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: the entry points declared in this header are its whole dynamic symbol table. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define CNL_ABI_VERSION 13   /* 13: CNL_ALGO_F43 + kernel variant 13 (csrc/winograd13.hip: 1-D Winograd F(4,3) along x on the fp16-split arithmetic; its weight pieces are a new tail of the transformed-weight buffer, so cnl_winograd_weight_floats grows for Cin % 32 == 0); 12: cnl_conv_params.w_up + cnl_winograd_up_weight_floats / cnl_winograd_transform_weights_up_f32 (a 3x3 conv behind a folded nearest-2x upsample: pre-summed row-pair weights, two instead of three kernel rows per output row), cnl_sizeof_params (a binder's struct-layout check); 11: cnl_conv_params.fuse_w / fuse_part + cnl_fused_out_pack_weights_f32 / cnl_fused_out_reduce_f32 (a 1x1 conv of <= 4 channels folded into the 3x3 launch before it); the row-Winograd kernels take maps of any even width in packed rows and tensors of >= 4 GiB in groups of images; CNL_ALGO_FORCE + 32 + v; 10: per-image maxima arrays are strided (cnl_absmax_stride() = 32 floats: one cache line per image); the stem entry points take y_absmax; CNL_ALGO_LATENCY and the half-height row-Winograd kernel (csrc/winograd10.hip, variants 10 / 11); the F(4x4,3x3) split kernel is gone (CNL_ALGO_F4, CNL_WINO_F16X2_F4, cnl_winograd_f4_weight_floats, cnl_winograd_transform_weights_f4_f32 removed: slower than the row-Winograd default at 4x its rounding error); 9: CNL_W_SPLIT + cnl_conv_split_weight_floats / cnl_conv_split_weights_f32 (pre-split weights for the fp16-split direct convs); 8: cnl_track_frame_f32 / cnl_track_frame_bytes (one self-describing record per frame, writable straight into mapped host memory), cnl_host_alloc / cnl_host_free; 7: the F(4x4) weight copy is an optional tail of the transformed-weight buffer (cnl_winograd_f4_weight_floats, cnl_winograd_transform_weights_f4_f32), cnl_conv3x3_winograd_variant, row-Winograd kernel behind CNL_WINO_F16X2; 6: cnl_conv_params.splitk / splitk_scratch (reduction split for small grids), cnl_fuse_sum_nhwc_f32; 5: cnl_conv_params.algo (arithmetic class per launch instead of process-wide environment switches), Winograd F(4x4,3x3) kernel, cnl_stem_conv7x7_f32 takes algo, uint8 stem + resize entry points; 4: cnl_conv_params.w_absmax, cnl_conv2d_kernel, cnl_absmax_per_image_f32 (fp16-split direct conv); 3: cnl_conv_params carries x_absmax / y_absmax (tensor-maximum hand-over between conv launches); 2: stem packed weights are [154][64] (cnl_stem_packed_weight_floats), neck-option / tracker / format entry points */

enum {
    CNL_OK = 0,
    CNL_E_BAD_ARG = -1,      /* null pointer, non-positive dimension, inconsistent shapes            */
    CNL_E_UNSUPPORTED = -2,  /* shape outside what the gfx950 kernels cover (e.g. Cin % 32 != 0)     */
    CNL_E_WORKSPACE = -3,    /* workspace pointer null / too small (see cnl_decode_workspace_bytes)  */
    CNL_E_HIP = -4           /* a HIP runtime call failed; text in cnl_last_error()                  */
};

/*
 * cnl_conv_params.algo / the `algo` argument of cnl_stem_conv7x7_f32: the ARITHMETIC CLASS the caller allows for a launch.  Inside a class
 * the kernel is a function of the layer shape and the hints alone — never of the batch size, never of the environment.
 *   CNL_ALGO_AUTO  the default.  fp32 in / fp32 accumulate / fp32 out everywhere; where it pays, each fp32 product is formed on the fp16
 *                  matrix cores from a scaled two-way fp16 split of both operands (three cross terms): every kernel's error against
 *                  float64 is at or below the fp32 matrix core's (tests/test_gpu_conv.py pins that per kernel).
 *   CNL_ALGO_F2    synonym of CNL_ALGO_AUTO (Winograd tiles no larger than F(2x2,3x3)).
 *   CNL_ALGO_F32   fp32 matrix cores only (v_mfma_f32_32x32x2_f32), no split operands anywhere; hints are ignored.
 *   CNL_ALGO_LATENCY  cnl_conv3x3_winograd_f32 only: AUTO's arithmetic on small work items (csrc/winograd10.hip: 4 rows x 64 pixels x 32 couts, two
 *                  workgroups per CU) wherever the row-Winograd kernels apply — for one-image batches, where the default's 8-row x 64-cout items leave
 *                  most CUs idle (a 256 -> 256 conv on a 32 x 32 map: 46 -> 23 us).  Same bits as AUTO wherever AUTO takes a row-Winograd kernel.
 *   CNL_ALGO_F43   cnl_conv3x3_winograd_f32 only (ABI v13): AUTO's choices, except that a 3x3 / stride-1 layer with Cin >= 128 on a map at least 128 pixels wide that the 4-row x 128-pixel
 *                  items of csrc/winograd13.hip tile well (padding <= 1.35 x; packed rows included) runs as 1-D Winograd F(4,3) along x — 108 instead of
 *                  144 matrix instructions per 16-channel chunk, the same split arithmetic, interpolation points {0, -1, 1, 1/2, -2, inf}.  The larger tile's
 *                  transforms amplify rounding: error against float64 2.4-4.6 x the fp32 matrix core's (tests/test_gpu_conv.py pins <= 6 x), inside the
 *                  path's 1e-4 by two orders of magnitude but above AUTO's promise — hence a class of its own, never what AUTO takes.  Batch-invariant.
 *   CNL_ALGO_FORCE + v  tests / A-B measurements: pin kernel variant v (2, 5, 6, 9, 10, 11, 13; 1, 3, 4, 7 in `make experiments` builds) wherever
 *                  it can run at all.
 */
enum {
    CNL_ALGO_AUTO = 0,
    CNL_ALGO_F2 = 1,
    CNL_ALGO_F32 = 2,                /* (3 was CNL_ALGO_F4 until ABI v9: rejected now) */
    CNL_ALGO_LATENCY = 4,
    CNL_ALGO_F43 = 5,
    CNL_ALGO_FORCE = 100
};

/* epilogue / gather flags of cnl_conv2d_nhwc_f32 */
enum {
    CNL_RELU = 1u << 0,          /* y = max(y, 0)                      nn.ReLU      (layers.py:75)        */
    CNL_SIGMOID = 1u << 1,       /* y = 1/(1+exp(-y))                  .sigmoid()   (centernet.py:205)    */
    CNL_UPSAMPLE_IN = 1u << 2,   /* read x through nn.Upsample(scale_factor=2, mode="nearest")
                                    (layers.py:99): logical input is (2*H_in, 2*W_in)                   */
    CNL_UPSAMPLE_OUT_ADD = 1u << 3, /* write y at 2x resolution and add `residual` there:
                                    y[n,2oy+dy,2ox+dx,:] = conv(x)[n,oy,ox,:] + bias + residual[...]
                                    = Fuse.forward's project -> resize("up") -> sum (layers.py:160-174) */
    CNL_RELU6 = 1u << 4,         /* y = min(max(y, 0), 6)              nn.ReLU6     (layers.py:62,66; separable conv);
                                    cnl_conv2d_nhwc_f32 / cnl_deconv2x_nhwc_f32 / cnl_depthwise3x3_nhwc_f32 only      */
    CNL_W_SPLIT = 1u << 5        /* cnl_conv2d_nhwc_f32: p->w is a cnl_conv_split_weights_f32 buffer (the fp32 OHWI weights followed by
                                    their scaled fp16 split): the fp16-split direct kernel reads the pieces instead of splitting
                                    the weights of every chunk again — same bits out; kernels that do not split ignore the tail   */
};

/*
 * One fused convolution layer: Conv2d (+ folded BatchNorm2d) (+ residual add) (+ ReLU | sigmoid).
 * Replaces, per call site:
 *   - ResNet BasicBlock conv3x3/BN/ReLU and the 1x1 stride-2 downsample (torchvision, reached through
 *     backbone.forward_features, models/meta.py:42);
 *   - make_conv(..., conv_type="normal")            models/layers.py:72-77
 *   - Fuse.project 1x1 conv with bias               models/layers.py:152
 *   - GenericHead block_i (ConvBnAct) and out_conv  models/meta.py:24-30
 * Implementation: fp32 MFMA (v_mfma_f32_32x32x2_f32) implicit GEMM, M = N*Ho*Wo, N = Cout,
 * K = KH*KW*Cin, LDS-staged via buffer_load ... lds.
 */
typedef struct cnl_conv_params {
    const float* x;         /* input  [N, H_in, W_in, ldx]                                           */
    const float* w;         /* weight [Cout, KH, KW, Cin]                                            */
    const float* bias;      /* [Cout] (never null; zeros when the layer has none)                    */
    const float* residual;  /* null, or same geometry as y with pixel stride ldr                      */
    float* y;               /* output [N, H_out, W_out, ldy]                                         */
    int32_t N, H_in, W_in, Cin, Cout;
    int32_t KH, KW, stride, pad;
    int32_t ldx, ldy, ldr;  /* pixel strides in elements                                             */
    uint32_t flags;         /* CNL_RELU | CNL_SIGMOID | CNL_UPSAMPLE_IN | CNL_UPSAMPLE_OUT_ADD        */
    /* Optional hand-over of the per-image maximum magnitude of a tensor between launches (cnl_conv3x3_winograd_f32 only; NULL =
     * unused).  Both point to N * cnl_absmax_stride() floats: image n's value sits at element n * cnl_absmax_stride() (32 floats = one 128-byte line
     * per image since ABI v10, see cnl_absmax_stride() below; an array of N packed floats is OUT OF BOUNDS for n > 0).  The fp16-split Winograd kernel scales each image's input by a power of two
     * derived from max |x| of THAT image (an image's result never depends on its batch neighbours): with x_absmax it reads the
     * maxima from device memory instead of making its own pass over x.  A producer given y_absmax folds max |y| of everything it
     * stores for image n into y_absmax[n * cnl_absmax_stride()] (atomic max on the bit pattern: zero the array on the stream before the producer
     * runs); the fp16-split kernels honour it, and so does cnl_conv2d_nhwc_f32's CNL_UPSAMPLE_OUT_ADD epilogue (fp32 matrix cores; FPN Fuse: the 3x3 output
     * conv behind it consumes the figure); the other fp32-matrix-core launches ignore it.  A maximum over a superset of the consumer's
     * channels is a valid, slightly conservative bound.  GUARANTEED RANGE of the split arithmetic: one scale per image means a value 2^-n below
     * the image's maximum keeps min(22, 38 - n) significant bits — outputs whose inputs lie within 1e4 of the image maximum are at the fp32
     * matrix core's error level, at 1e6 the error is 2-6e-5 of the LOCAL output magnitude (inside the path's 1e-4), beyond that pass
     * CNL_ALGO_F32 (tests/test_gpu_conv.py::test_split_arithmetic_with_an_outlier_inside_one_image).  WITHOUT x_absmax a launch of the fp16-split kernels takes at most 1024 images (4096 up
     * to ABI v9: the private scratch of the own pass is strided like every maxima array now); more return CNL_E_UNSUPPORTED.
     *                                                                        */
    const float* x_absmax;
    float* y_absmax;
    /* cnl_conv2d_nhwc_f32 only: device pointer to ONE float, max |w| of this layer's weights (e.g. computed once when the weights
     * are loaded), or NULL.  With both x_absmax and w_absmax the 1x1 / 3x3 convs without up-sampling flags form each fp32 product
     * on the fp16 matrix cores from scaled two-way splits of both operands (csrc/conv_f16x2.hip: same error against float64 as
     * the fp32 matrix-core kernel, 2-3x faster) and honour y_absmax; without them the fp32 matrix-core kernel runs.             */
    const float* w_absmax;
    uint32_t algo;          /* CNL_ALGO_* (0 = CNL_ALGO_AUTO)                                                                       */
    /* cnl_conv2d_nhwc_f32 only, optional (0 / NULL = off): split the reduction (KH*KW*Cin) over `splitk` workgroups per output tile —
     * for launches whose output is too small to fill the chip (one image, 16x16 .. 32x32 maps): slice s writes its partial sums to
     * splitk_scratch[s][N*Ho*Wo][Cout] (cnl_conv2d_splitk_scratch_bytes()), a second kernel adds the slices IN SLICE ORDER (deterministic)
     * and applies bias / residual / activation / y_absmax.  Honoured where the fp16-split kernel runs (x_absmax and w_absmax given,
     * 1x1 or 3x3, no CNL_UPSAMPLE_*); elsewhere the launch runs unsplit.  The result depends on `splitk` (summation grouping), so a
     * caller that needs shard == full batch bit for bit must choose it from the layer shape alone.                                  */
    int32_t splitk;
    float* splitk_scratch;
    size_t splitk_scratch_bytes;
    /* cnl_conv3x3_winograd_f32 only, optional (NULL = off; ABI v11): a following 1x1 convolution with at most 4 output channels — the box-size
     * head's out_conv behind its last 3x3 block (reference models/meta.py:24-30) — folded into THIS launch: besides y, the epilogue writes
     * fuse_part[b][pixel][0..3] = sum over the 32 channels co of block b (co in [32 b, 32 b + 32)) of y[pixel][co] * fuse_w[co][0..3]
     * (fp32, fixed order), for b < ceil(Cout / 64) * 2; cnl_fused_out_reduce_f32 then adds the blocks in order, adds the bias and applies the
     * activation: the 1x1 conv never re-reads the 3x3 conv's output (C1: a 537 MB read, 110 us).  fuse_w: [ceil(Cout / 64) * 64][4] floats,
     * rows >= Cout and columns >= the 1x1 conv's channel count zero (cnl_fused_out_pack_weights_f32); fuse_part: ceil(Cout / 64) * 2 *
     * N * H * W * 4 floats.  Implemented by ONE kernel: the row-Winograd variant 9 (csrc/winograd9.hip).  The dispatcher keeps a launch that
     * carries fuse_w on variant 9 wherever its AUTO / LATENCY / F43 choice would have been another ROW kernel (10 / 11 / 13), but a shape it routes elsewhere
     * (e.g. N = 32, 16 x 16, 512 -> 512: variant 5) fails with CNL_E_UNSUPPORTED — a caller must check cnl_conv3x3_winograd_variant(p) == 9 WITH fuse_w set
     * before relying on the fold (engine.py does).  Deterministic, batch-invariant. */
    const float* fuse_w;
    float* fuse_part;
    /* cnl_conv3x3_winograd_f32 with CNL_UPSAMPLE_IN only, optional (NULL = off; ABI v12): the ROW-PAIR weights of this layer
     * (cnl_winograd_transform_weights_up_f32).  Behind a nearest-2x upsample (make_upsample + ConvBnAct: reference models/layers.py:99,72-77 — the first
     * block of every head behind the simple neck, models/meta.py:24-26) the image rows 2j and 2j+1 are the same source row, so the three kernel rows of an
     * output row meet two distinct input rows: out[2m] = row[2m-1] g0 + row[2m] (g1 + g2), out[2m+1] = row[2m] (g0 + g1) + row[2m+2] g2.  With the four
     * pre-summed sets the row-Winograd kernel (variant 9) issues 96 instead of 144 matrix instructions per 16-channel chunk.  Results: within fp32
     * rounding of the launch without w_up (another grouping of the same products), deterministic, batch-invariant — a function of the shape and of w_up
     * being given.  Ignored (the general form runs) with a residual, with fuse_w, and by every other kernel.                                        */
    const float* w_up;
} cnl_conv_params;

/* The 1x1 conv folded into a 3x3 launch (cnl_conv_params.fuse_w / fuse_part): w_ohwi [C2][Cout] (C2 <= 4) -> fuse_w [ceil(Cout/64)*64][4];
 * and its second half: y[pixel][c] = act(bias[c] + sum_b part[b][pixel][c]), b in order, for M pixels (pixel stride ldy floats, c < C2;
 * flags: CNL_SIGMOID | CNL_RELU).                                                                                                       */
int cnl_fused_out_pack_weights_f32(const float* w_ohwi, float* fuse_w, int32_t Cout, int32_t C2, void* stream);
int cnl_fused_out_reduce_f32(const float* part, int32_t nblocks, int64_t M, int32_t C2, const float* bias, float* y, int32_t ldy,
                             uint32_t flags, void* stream);
int cnl_conv2d_nhwc_f32(const cnl_conv_params* p, void* stream);
/* The row-pair weights of cnl_conv_params.w_up: w_ohwi [Cout][3][3][Cin] (BatchNorm folded) -> u_up, cnl_winograd_up_weight_floats(Cin, Cout) floats
 * (0: Cin % 32 != 0, no such form): [Cin/16][4 positions][4 sets g0, g0+g1, g1+g2, g2][2 fp16 pieces][ceil(Cout/64)*64][16] + one inverse scale per cout. */
size_t cnl_winograd_up_weight_floats(int32_t Cin, int32_t Cout);
int cnl_winograd_transform_weights_up_f32(const float* w_ohwi, float* u_up, int32_t Cin, int32_t Cout, void* stream);
size_t cnl_conv2d_splitk_scratch_bytes(const cnl_conv_params* p);   /* for p->splitk slices; 0 when splitk <= 1 */

/* Which kernel cnl_conv2d_nhwc_f32 takes for *p (a function of the hints, kernel size and flags only — never of the batch). */
#define CNL_CONV_F32 2     /* fp32 matrix cores (csrc/conv_mfma.hip); ignores y_absmax                                    */
#define CNL_CONV_F16X2 5   /* fp16 matrix cores, scaled two-way split (csrc/conv_f16x2.hip); writes y_absmax when given   */
int cnl_conv2d_kernel(const cnl_conv_params* p);

/*
 * 3x3 / stride 1 / pad 1 conv on the nearest-2x upsampled input (nn.Upsample(scale_factor=2) + ConvBnAct: models/layers.py:99,72-77;
 * the first head block behind the simple neck, models/meta.py:24-26) computed as four 2x2 sub-pixel phase convolutions on the
 * LOW-resolution input (16 instead of 36 multiplies per 2x2 output block; see csrc/conv_mfma.hip).  Same cnl_conv_params as
 * cnl_conv2d_nhwc_f32 with flags = CNL_UPSAMPLE_IN (| CNL_RELU | CNL_RELU6), KH = KW = 3, stride 1, pad 1, no residual; H_in / W_in
 * are the LOW-resolution size, y is [N, 2 H_in, 2 W_in, Cout].  p->w: the buffer cnl_up2_pack_weights_f32 fills from the OHWI
 * [Cout][3][3][Cin] weights (Cin % 32 == 0; cnl_up2_weight_floats sizes it): the phase weights [4][Cout][2][2][Cin] in fp32, the
 * same as scaled two-way fp16 split in the kernel's LDS row layout, and the scale.  With x_absmax the phases run on the fp16-split
 * kernel with the pre-split weights (cnl_conv3x3_up2_kernel reports CNL_CONV_F16X2; w_absmax is not needed) and y_absmax is
 * honoured; otherwise on the fp32 matrix cores.
 */
/*
 * Weights of a direct conv (square 1x1 / 3x3 kernel, Cin % 32 == 0) with their fp16 split appended, for flags |= CNL_W_SPLIT:
 * w_buf = [Cout*KH*KW*Cin fp32 OHWI weights (copied from w_ohwi unless w_buf == w_ohwi)][the same count of (hi, lo) fp16 pairs, scaled by the
 * power of two S_w = 2^(14 - e) of max |w| = m 2^e, in the B-row layout of the kernel][S_w + 3 pad floats];
 * cnl_conv_split_weight_floats sizes it (0: shape not supported).  Replaces nothing in the reference: it is the weight half of the
 * operand split that round 2's kernel redid for every 32-channel chunk of every launch (half of its VALU work on the stride-2 3x3 convs).
 */
size_t cnl_conv_split_weight_floats(int32_t Cin, int32_t Cout, int32_t KH, int32_t KW);
int cnl_conv_split_weights_f32(const float* w_ohwi, float* w_buf, int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, void* stream);
size_t cnl_up2_weight_floats(int32_t Cin, int32_t Cout);
int cnl_up2_pack_weights_f32(const float* w_ohwi, float* w_packed, int32_t Cin, int32_t Cout, void* stream);
int cnl_conv3x3_up2_nhwc_f32(const cnl_conv_params* p, void* stream);
int cnl_conv3x3_up2_kernel(const cnl_conv_params* p);

/* The per-image maxima arrays of this API (cnl_conv_params.x_absmax / y_absmax, the stems' y_absmax, `out` below) hold image n's value at
 * element n * cnl_absmax_stride() — 32 floats apart since ABI v10: ONE 128-byte line per image.  (Every wave of a launch folds its maximum
 * into these with device-scope atomics, which the memory side resolves line by line: packed into one line, as in ABI <= 9, the reports of all
 * images queue behind each other.)  An array therefore has N * cnl_absmax_stride() floats; the elements between the slots are never read. */
int cnl_absmax_stride(void);
/* out[n * cnl_absmax_stride()] = max |x[n, :, 0:C]| over the `pixels` pixels of image n (pixel stride ld floats; C % 4 == 0, ld % 4 == 0, x
 * 16-byte aligned): the x_absmax hint for callers whose producer does not report it.  Zeroes the array first (stream-ordered).           */
int cnl_absmax_per_image_f32(const float* x, int32_t N, int64_t pixels, int32_t C, int32_t ld, float* out, void* stream);

/* Output spatial size of the conv itself (before CNL_UPSAMPLE_OUT_ADD doubles it). */
int cnl_conv2d_out_hw(const cnl_conv_params* p, int32_t* H_out, int32_t* W_out);

/*
 * The same fused layer for 3x3 / stride 1 / pad 1 (ResNet BasicBlock convs, make_conv, GenericHead blocks) computed by
 * Winograd F(2x2,3x3): 2.25x fewer matrix-core multiplies, same result up to fp32 rounding (see csrc/winograd.hip).
 * `p->w` must point to the PRE-TRANSFORMED weights produced by cnl_winograd_transform_weights_f32 from the OHWI
 * (BN-folded) weights; flags: CNL_RELU only (upsample / sigmoid variants stay on cnl_conv2d_nhwc_f32). Cin % 8 == 0.
 *
 * Several multiplier arrays serve this entry point, chosen from the layer SHAPE and p->algo (never the batch size): the fp32 matrix
 * core (csrc/winograd2.hip), or — where the channel loop is long (Cin >= 128 or Cout >= 256, Cin % 16 == 0) — the fp16 matrix core
 * fed with a two-way fp16 split of both fp32 operands under a per-image power-of-two scale (three cross terms, fp32
 * accumulation: csrc/winograd5.hip, winograd6.hip; measured error at or below the fp32 matrix core's, half-precision rate = 16x), as
 * F(2x2,3x3).  Round 3: wherever
 * 8-row x 64-pixel work items pad the map by less than 1.5x (Cin % 32 == 0, Cout % 4 == 0), the same split arithmetic runs as 1-D Winograd
 * F(2,3) along x with the three kernel rows in the reduction (csrc/winograd9.hip: per-output-channel weight scales; 0.5-0.8x the time of the
 * 2-D kernels, rounding error below theirs) — same class CNL_WINO_F16X2.
 * cnl_conv3x3_winograd_kernel reports which class a layer takes.  The fp16-split kernels without the x_absmax hint make their own
 * pass over the input and park the per-image maxima in the layer's weight buffer: such hint-less launches of ONE layer must not
 * run concurrently on two streams (launches that carry x_absmax — everything engine.py issues — have no hidden state).
 */
#define CNL_WINO_F32 2
#define CNL_WINO_BF16X3 3      /* experiment builds only */
#define CNL_WINO_F16X2 5
int cnl_conv3x3_winograd_f32(const cnl_conv_params* p, void* stream);
int cnl_conv3x3_winograd_kernel(const cnl_conv_params* p);            /* CNL_WINO_* for this layer shape, < 0: error code */
int cnl_conv3x3_winograd_variant(const cnl_conv_params* p);           /* the kernel behind the class (reporting only): 2 winograd2, 5 / 6 winograd5 / 6
                                                                         [F(2x2,3x3)], 9 winograd9 [F(2,3) along x, kernel rows
                                                                         in the reduction: 2/3 of the direct conv's multiplies], 10 / 11 winograd10
                                                                         [the same on 4-row x 64- / 32-cout items: bit for bit winograd9's
                                                                         output — the class never depends on N, but a launch of at most
                                                                         128 winograd9 items takes 11], 13 winograd13 [F(4,3) along x: 1/2 of the
                                                                         direct conv's multiplies; CNL_ALGO_F43 / FORCE + 13 only]; < 0: error code */
size_t cnl_winograd_weight_floats(int32_t Cin, int32_t Cout);        /* elements of the transformed weight buffer */
int cnl_winograd_transform_weights_f32(const float* w_ohwi, float* u, int32_t Cin, int32_t Cout, void* stream);
/*
 * Step before the path: uint8 HWC frames -> normalised fp32 NHWC, replacing albumentations
 * A.Normalize + ToTensorV2 of the reference's inference pre-processing (README.md:79-87, datasets/utils.py:9-21):
 * y = (float(x) - mean255[c]) * inv_std255[c], with HOST arrays mean255 = mean*255, inv_std255 = 1/(std*255) (3 floats each).
 * x: [N,H,W,3] u8, y: [N,H,W,3] f32 (feed cnl_stem_conv7x7_f32 with strides sn=H*W*3, sc=1, sh=W*3, sw=3).
 */
int cnl_normalize_u8_nhwc_f32(const uint8_t* x, float* y, int32_t N, int32_t H, int32_t W, const float* mean255,
                              const float* inv_std255, void* stream);

/*
 * albumentations A.Resize(height, width) = cv2.resize(img, (W_out, H_out), interpolation=cv2.INTER_LINEAR) on uint8 HWC frames
 * (README.md:84; datasets/utils.py:24-33 build the same pipeline for training): OpenCV's 8-bit fixed-point bilinear rule (half-pixel
 * centres, 11-bit coefficients, the two-stage rounding of VResizeLinear<uchar>), restated in csrc/preprocess.hip.
 * x: [N, H_in, W_in, C] u8 -> y: [N, H_out, W_out, C] u8, C <= 4.
 */
int cnl_resize_bilinear_u8(const uint8_t* x, uint8_t* y, int32_t N, int32_t H_in, int32_t W_in, int32_t H_out, int32_t W_out,
                           int32_t C, void* stream);

/*
 * ResNet stem: Conv2d(3,64,7,stride=2,padding=3,bias=False)+BN+ReLU (torchvision resnet.conv1/bn1/relu).
 * x is read through explicit element strides (sn,sc,sh,sw) so NCHW-contiguous and channels_last
 * callers are both zero-copy (models/meta.py:97-98 precedent); y is NHWC [N, H/2, W/2, 64].
 * w: the PACKED weights that cnl_stem_pack_weights_f32 makes from the OHWI [64][7][7][3] (BN-folded) weights — LDS images copied
 * verbatim by LDS-DMA: the fp32 image [154][64] (row k = ky*22 + kx*3 + c; the 22nd row of every ky is zero) followed by the
 * scaled two-way fp16 split [piece][21 groups of 8 k][64][8] and its power-of-two scale (csrc/stem_f16x2.hip: the default kernel
 * forms each fp32 product on the fp16 matrix cores, input scaled per workgroup patch; algo = CNL_ALGO_F32 selects the fp32
 * matrix-core kernel); cnl_stem_packed_weight_floats() sizes the buffer; bias: [64].
 * y_absmax (all three stem entry points, ABI v10): NULL, or N * cnl_absmax_stride() floats zeroed by the caller on the stream — the fp16-split kernel folds max |y| of
 * image n into y_absmax[n * cnl_absmax_stride()] (atomic max on the bit pattern; the values are post-ReLU), the hand-over the first Winograd layer takes as
 * cnl_conv_params.x_absmax instead of a pass over the stem's output.  The fp32 kernel (CNL_ALGO_F32 without the fused pool) ignores it.
 */
size_t cnl_stem_packed_weight_floats(void);
int cnl_stem_pack_weights_f32(const float* w_ohwi, float* w_packed, void* stream);
int cnl_stem_conv7x7_f32(const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                         const float* w, const float* bias, float* y, float* y_absmax,
                         int32_t N, int32_t H, int32_t W, uint32_t algo, void* stream);

/* The stem and resnet.maxpool in ONE launch: y = MaxPool2d(3, stride 2, padding 1)(ReLU(BN(conv7x7/2(x)))) as NHWC
 * [N, (Ho-1)/2+1, (Wo-1)/2+1, 64] with Ho = (H-1)/2+1, Wo = (W-1)/2+1; bit-identical to cnl_stem_conv7x7_f32 followed by
 * cnl_maxpool3x3s2_nhwc_f32 (always the fp16-split stem kernel), without the stride-2 feature map ever reaching memory.  The call
 * initialises y itself (stream-ordered: the cells that several workgroups merge are zeroed, every other cell is stored once).                                                                                         */
int cnl_stem_conv7x7_maxpool_f32(const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                                 const float* w, const float* bias, float* y, float* y_absmax,
                                 int32_t N, int32_t H, int32_t W, void* stream);

/* The stem on uint8 frames: x is [N, H, W, 3]-like uint8 addressed through BYTE strides (sn, sc, sh, sw); A.Normalize
 * — (float(x) - mean255[c]) * inv_std255[c], the arithmetic of cnl_normalize_u8_nhwc_f32, bit for bit — is applied to the staged patch
 * inside the kernel, so the fp32 image never exists in HBM (3 B/pixel read instead of 12 written + 12 read).  mean255 / inv_std255: HOST
 * arrays of 3 floats.  fuse_maxpool != 0: y is the pooled map of cnl_stem_conv7x7_maxpool_f32, else the conv output of
 * cnl_stem_conv7x7_f32.  Bit-identical to cnl_normalize_u8_nhwc_f32 followed by the fp32-input entry points (CNL_ALGO_AUTO).            */
int cnl_stem_conv7x7_u8(const uint8_t* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, const float* mean255,
                        const float* inv_std255, const float* w, const float* bias, float* y, float* y_absmax, int32_t N, int32_t H, int32_t W,
                        int32_t fuse_maxpool, void* stream);

/* nn.MaxPool2d(kernel_size=3, stride=2, padding=1) on NHWC (torchvision resnet.maxpool). C % 4 == 0. */
int cnl_maxpool3x3s2_nhwc_f32(const float* x, float* y, int32_t N, int32_t H, int32_t W, int32_t C,
                              void* stream);

/*
 * Fused decode = CenterNet.decode_detections (models/centernet.py:229-304)
 *              + EmbeddingHead.gather_at_indices (models/fairmot.py:63-73) when reid != null:
 *   3x3 (nms_kernel) max-pool pseudo-NMS with equality mask, per-pixel max/argmax over classes,
 *   per-image sorted top-k (ties: score desc, flat index asc), label / ltrb-box / embedding gather,
 *   box decode to x1y1x2y2.
 * heat/box/reid element (n,c,y,x) at n*s_n + c*s_c + y*s_h + x*s_w (elements): any NCHW or NHWC view.
 * Outputs: scores [N,k] f32, indices [N,k] i64 (flat y*W+x), labels [N,k] i64, boxes [N,k,4] f32,
 * emb [N,k,E] f32 (may be null when reid is null).
 */
typedef struct cnl_decode_params {
    const float* heat; int64_t heat_sn, heat_sc, heat_sh, heat_sw;
    const float* box;  int64_t box_sn, box_sc, box_sh, box_sw;
    const float* reid; int64_t reid_sn, reid_sc, reid_sh, reid_sw;   /* reid may be null */
    int32_t N, C, H, W, E;
    int32_t k;               /* num_detections, 1 <= k <= min(1024, H*W)                              */
    int32_t nms_kernel;      /* odd, 1..7 (reference default 3, centernet.py:93)                      */
    int32_t normalize_boxes; /* !=0: divide by (W,H) (centernet.py:299-301) else multiply by stride    */
    int32_t box_log;         /* !=0: exp() the offsets first (centernet.py:283-284)                   */
    float box_multiplier;    /* centernet.py:285                                                      */
    float stride;            /* output stride (centernet.py:303), e.g. 4                              */
    float* scores; int64_t* indices; int64_t* labels; float* boxes; float* emb;
    void* workspace; size_t workspace_bytes;
} cnl_decode_params;

size_t cnl_decode_workspace_bytes(int32_t N, int32_t H, int32_t W);
int cnl_decode_f32(const cnl_decode_params* p, void* stream);

/*
 * Standalone gathers at caller-supplied flat indices [N,k] (i64), for the Gen-A per-head calls
 * heads["box_2d"].gather_at_indices / EmbeddingHead.gather_at_indices (models/fairmot.py:141-143, 63-73;
 * arithmetic of CenterNet.gather_and_decode_boxes, models/centernet.py:263-304).  Strides in elements.
 */
int cnl_gather_boxes_f32(const float* box, int64_t sn, int64_t sc, int64_t sh, int64_t sw, const int64_t* indices,
                         float* boxes, int32_t N, int32_t H, int32_t W, int32_t k, int32_t normalize_boxes,
                         int32_t box_log, float box_multiplier, float stride, void* stream);
int cnl_gather_embeddings_f32(const float* reid, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                              const int64_t* indices, float* emb, int32_t N, int32_t E, int32_t H, int32_t W,
                              int32_t k, void* stream);

/*
 * All-gather record (replaces the pickled all_gather_object of eval/coco.py:10-18):
 * rec[n][j][0:4] = box, [4] = score, [5] = bit pattern of (int32)label, [6:6+E] = embedding.
 */
int cnl_pack_detections_f32(const float* boxes, const float* scores, const int64_t* labels, const float* emb,
                            float* rec, int32_t N, int32_t k, int32_t E, void* stream);
int cnl_unpack_detections_f32(const float* rec, float* boxes, float* scores, int64_t* labels, float* emb,
                              int32_t N, int32_t k, int32_t E, void* stream);

/*
 * SURVEY.md §8f next #3 — the remaining neck options of make_upsample / make_conv (models/layers.py:40-101).
 *
 * cnl_deconv2x_nhwc_f32: upsample_type="conv_transpose" = nn.ConvTranspose2d(C, C', K, stride=2, padding=(K + K%2)/2 - 1,
 * output_padding=K%2, bias=False) + BatchNorm2d + ReLU (layers.py:86-93; K = deconv_kernel in {2,3,4}; output is exactly
 * 2H x 2W).  Computed as FOUR sub-pixel phase convolutions on the MFMA implicit-GEMM kernel (no zero-stuffing): phase
 * (dy,dx) writes y[n, 2i+dy, 2j+dx, :].  y = act(deconv(x) + bias) (+ residual at the same 2x position, added after the
 * activation: Fuse's "in1 + resize(in2)", layers.py:160-174).
 * `w` is the PACKED weight: for phase ph = 2*dy + dx a block [Cout][KHp][KWp][Cin] (OHWI), blocks back to back, where with
 * (taps, pad) = cnl_deconv_phase_geometry(K, d) per axis and p = (K + K%2)/2 - 1:
 *     w_packed[ph][co][jy][jx][ci] = scale[co] * W[ci][co][dy + p + 2*(pad_y - jy)][dx + p + 2*(pad_x - jx)]
 * (W = ConvTranspose2d.weight [Cin][Cout][K][K]; scale = folded BN gamma/sqrt(var+eps)); cnl_deconv_weight_floats() = K*K*Cin*Cout.
 */
typedef struct cnl_deconv_params {
    const float* x;         /* input  [N, H_in, W_in, ldx]                         */
    const float* w;         /* packed phase weights (see above)                    */
    const float* bias;      /* [Cout]                                              */
    const float* residual;  /* null, or [N, 2H_in, 2W_in, ldr]                     */
    float* y;               /* output [N, 2H_in, 2W_in, ldy]                       */
    int32_t N, H_in, W_in, Cin, Cout, K;
    int32_t ldx, ldy, ldr;
    uint32_t flags;         /* CNL_RELU | CNL_RELU6                                */
} cnl_deconv_params;
int cnl_deconv2x_nhwc_f32(const cnl_deconv_params* p, void* stream);
int cnl_deconv_phase_geometry(int32_t K, int32_t d, int32_t* taps, int32_t* pad);
size_t cnl_deconv_weight_floats(int32_t Cin, int32_t Cout, int32_t K);

/*
 * nn.Upsample(scale_factor=2, mode="nearest" | "bilinear") (layers.py:99; bilinear = align_corners=False) on NHWC, optionally
 * adding `residual` at the output resolution (Fuse: in1 + resize(in2)).  mode: 0 = nearest, 1 = bilinear.  C % 4 == 0.
 * Nearest feeding a "normal" conv never needs this (CNL_UPSAMPLE_IN folds it into the conv); it exists for bilinear and for
 * the depthwise path.
 */
int cnl_upsample2x_nhwc_f32(const float* x, const float* residual, float* y, int32_t N, int32_t H_in, int32_t W_in, int32_t C,
                            int32_t ldx, int32_t ldr, int32_t ldy, int32_t mode, void* stream);

/* The sum of a general Fuse node (reference models/layers.py:160-175; nodes with three inputs and/or resize="down" — BiFPN's bottom-up
 * path — and weighted_fusion with any resize):
 *     y = (g0 * in0 [+ g1 * in1] + gl * resize(last)) / den          in1 == NULL for a two-input node
 * in0 / in1 / y are [N, H, W, C] NHWC with pixel strides ld0 / ld1 / ldy; `last` is resized on the fly by `mode`:
 *   0  nn.Upsample(2, "nearest")   last is [N, H/2, W/2, C]       1  nn.Upsample(2, "bilinear", align_corners=False)   same shape
 *   2  nn.MaxPool2d(2, 2)          last is [N, 2H, 2W, C]         3  none (already resized, e.g. by cnl_deconv2x_nhwc_f32)
 * Unweighted fusion: g* = den = 1 (bit-exact plain sum); weighted (layers.py:164-167): g_j = relu(w_j), den = sum_j relu(w_j) + 1e-6.
 * C and the strides multiples of 4, pointers 16-byte aligned.  HBM-bound: reads each input once, writes y once. */
int cnl_fuse_sum_nhwc_f32(const float* in0, const float* in1, const float* last, float* y, int32_t N, int32_t H, int32_t W, int32_t C,
                          int32_t ld0, int32_t ld1, int32_t ldl, int32_t ldy, float g0, float g1, float gl, float den, int32_t mode,
                          void* stream);

/*
 * conv_type="separable", depthwise half (layers.py:58-62): nn.Conv2d(C, C, 3, padding=1, groups=C, bias=False) + BN + ReLU6.
 * w: [3][3][C] (tap-major, BN scale folded), bias [C]; flags: CNL_RELU | CNL_RELU6.  C % 4 == 0.  The pointwise half is
 * cnl_conv2d_nhwc_f32 (1x1) with CNL_RELU6.
 */
int cnl_depthwise3x3_nhwc_f32(const float* x, const float* w, const float* bias, float* y, int32_t N, int32_t H, int32_t W,
                              int32_t C, int32_t ldx, int32_t ldy, uint32_t flags, void* stream);

/*
 * conv_type="deformable" (layers.py:9-38, 47-54): DeformableConv2dBlock = offset_conv (+ sigmoid mask_conv for version 2) feeding
 * torchvision DeformConv2d(K x K, stride 1, padding (K-1)/2, bias=False), then BN + ReLU.  Here: offsets and mask logits come from
 * ONE ordinary K x K conv (cnl_conv2d_nhwc_f32, Cout = 2KK [+ KK], weights concatenated); this entry point does the deformable
 * sampling   col[n,y,x,k,:] = sigmoid(mask_k) . bilinear(x[n], y - p + ky + dy_k, x - p + kx + dx_k)   (torchvision's rule: zero
 * outside (-1,H) x (-1,W), corners outside the image contribute zero; offsets (dy, dx) interleaved per tap k = ky*K + kx);
 * the product with the [Cout][K*K*C] weight (OHWI order, BN folded) is a 1x1 cnl_conv2d_nhwc_f32 over col (pixel stride K*K*C).
 * om: [N,H,W,ldo] with channels [0, 2KK) offsets, [2KK, 3KK) mask logits (has_mask != 0).  C % 4 == 0.
 */
int cnl_deform_sample_nhwc_f32(const float* x, const float* om, float* col, int32_t N, int32_t H, int32_t W, int32_t C,
                               int32_t ldx, int32_t ldo, int32_t K, int32_t has_mask, void* stream);

/*
 * Step after the path for the tracking task: association costs of one frame against the current
 * track table, and the track-table update, so that per-frame embeddings never leave HBM — only the n x T cost matrices go to
 * the host for the Hungarian step (scipy, as in the reference) and the match list comes back.
 *
 * cnl_track_costs_f32 replaces models/tracker.py:133-137 (mask = scores >= detection_threshold; boolean-mask compaction),
 * :150 (scipy cdist "cosine", float64) and :162 (utils/box.py:84-92 box_iou_distance_matrix / box_giou_distance_matrix, float32):
 *   det_emb [k,E], det_box [k,4] x1y1x2y2, det_score [k] (k <= 1024); trk_emb [T,E], trk_box [T,4] (T may be 0);
 *   box_cost: 0 = none, 1 = "iou", 2 = "giou";
 *   n_det [1] <- number of detections with score >= threshold; det_index [k] <- their original indices, ascending (first n_det);
 *   reid_cost [n_det,T] f64 and box_cost_out [n_det,T] f32 <- row r is detection det_index[r] (dense, row stride T).
 * cnl_track_apply_f32 replaces Track.__init__ / Track.update_matched (models/tracker.py:228, 305-321, use_kalman=False) and the
 * list rebuild of :186-196 for the device-resident table: new row r is
 *   src_trk[r] >= 0, src_det[r] <  0 : old row src_trk[r] unchanged
 *   src_trk[r] >= 0, src_det[r] >= 0 : emb = (1-s)*old + s*e/|e|, box = det_box[src_det[r]]   (e = det_emb[src_det[r]])
 *   src_trk[r] <  0, src_det[r] >= 0 : emb = e/|e|,               box = det_box[src_det[r]]   (new track)
 * src_trk / src_det are device int32 arrays of T_new entries; new_emb/new_box must not alias the old table.
 */
int cnl_track_costs_f32(const float* det_emb, const float* det_box, const float* det_score, int32_t k, int32_t E,
                        float detection_threshold, const float* trk_emb, const float* trk_box, int32_t T, int32_t box_cost,
                        int32_t* n_det, int32_t* det_index, double* reid_cost, float* box_cost_out, void* stream);
/* the same with the re-ID metric as an argument (models/tracker.py:51, 150: any scipy cdist metric): 0 "cosine", 1 "euclidean",
 * 2 "sqeuclidean", 3 "cityblock", 4 "chebyshev", 5 "canberra", 6 "braycurtis", 7 "correlation" (round 6: rows centred by their float64 mean in numpy's
 * pairwise order, then the cosine kernel) — each in float64 in scipy's operation order (bit for bit scipy's matrix; cosine / correlation to 1e-12).  Other metrics / callables: host fallback in tracker.py (opt-in). */
int cnl_track_costs_metric_f32(const float* det_emb, const float* det_box, const float* det_score, int32_t k, int32_t E,
                               float detection_threshold, const float* trk_emb, const float* trk_box, int32_t T, int32_t box_cost,
                               int32_t reid_metric, int32_t* n_det, int32_t* det_index, double* reid_cost, float* box_cost_out, void* stream);
/*
 * One frame's association in ONE launch and ONE record, so that the host makes a single stream synchronisation per frame and no
 * copy: the same compaction and costs as cnl_track_costs_metric_f32, written as
 *   int32 header[8] = { n, k, T, with_detections, off_index, off_dets, off_reid, off_box }   (byte offsets into the record)
 *   int32 det_index[k]                      at off_index  (first n valid)
 *   f32 boxes[k][4], f32 scores[k], int32 labels[k]   at off_dets   (only with_detections != 0: the frame's detections as the host-side
 *                                            track life cycle reads them, models/tracker.py:171-186; det_label read as label_kind says:
 *                                            1 int64, 2 int32, 3 float32; 0 = no labels, zeros)
 *   f64 reid_cost[n][T]                     at off_reid
 *   f32 box_cost[n][T]                      at off_box = off_reid + 8 n T   (box_cost != 0)
 * The kernel packs the matrices by the n it finds itself: the host needs no copy of the scores before the launch.  `record` holds
 * at least cnl_track_frame_bytes(k, T, with_detections) bytes (the n = k worst case; only the n x T part is written), is 8-byte
 * aligned and may be device memory or page-locked host memory from cnl_host_alloc — then the record crosses PCIe as the kernel's own
 * stores (52 KB for 58 x 70 pairs) and no copy-engine operation sits between the launch and the host's wait.
 */
int64_t cnl_track_frame_bytes(int32_t k, int32_t T, int32_t with_detections);
int cnl_track_frame_f32(const float* det_emb, const float* det_box, const float* det_score, const void* det_label, int32_t label_kind,
                        int32_t k, int32_t E, float detection_threshold, const float* trk_emb, const float* trk_box, int32_t T,
                        int32_t box_cost, int32_t reid_metric, int32_t with_detections, void* record, int64_t record_bytes, void* stream);
/* src_trk / src_det may also be cnl_host_alloc memory (2 x T_new int32 read over PCIe by the kernel: no host -> device copy) */
int cnl_track_apply_f32(const float* trk_emb, const float* trk_box, const float* det_emb, const float* det_box,
                        const int32_t* src_trk, const int32_t* src_det, int32_t T_new, int32_t E, double smoothing,
                        float* new_emb, float* new_box, void* stream);

/*
 * Wire formats: COCO boxes are xywh — torchvision box_convert(boxes, "xyxy", "xywh") of
 * CenterNet.validation_step (models/centernet.py:207): out[i] = (x1, y1, x2 - x1, y2 - y1); n boxes of 4 floats, may be in place.
 */
int cnl_boxes_xyxy_to_xywh_f32(const float* boxes, float* out, int64_t n, void* stream);

int cnl_version(void);
/* sizeof(cnl_conv_params) / sizeof(cnl_decode_params) / sizeof(cnl_deconv_params) (which = 0 / 1 / 2; else 0) as the library was compiled — the structs grow at the end between ABI
 * versions: a binder (ctypes, cgo, JNI ...) compares its own struct's size before the first call (ABI v12). */
size_t cnl_sizeof_params(int32_t which);
/* Copies the calling thread's last error message (NUL-terminated) into buf; returns its length. */
size_t cnl_last_error(char* buf, size_t n);
/*
 * Page-locked host memory that the device addresses through the SAME pointer (hipHostMalloc, mapped + coherent): for small records a
 * kernel writes for the host (cnl_track_frame_f32) or reads from it (cnl_track_apply_f32's index lists).  Kernel stores are visible to
 * the host after the stream is synchronised.  Not for bulk data: every access crosses PCIe.
 */
int cnl_host_alloc(size_t bytes, void** ptr);
int cnl_host_free(void* ptr);


#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* CENTERNET_GFX950_H */
