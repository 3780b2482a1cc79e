// winograd.hip — 3x3 / stride 1 / pad 1 convolution by Winograd: the entry point, the kernel choice and the weight transform.
//
// Same call sites as conv_mfma.hip (ResNet BasicBlock 3x3 convs, make_conv, GenericHead blocks: reference models/meta.py:24-26,
// models/layers.py:72-77) for the layers that are 3x3 stride-1: 90 % of the conv time.
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A
// is one GEMM per transform position: M_xi[tile][co] = sum_ci V_xi[tile][ci] U_xi[co][ci].  Kernels behind cnl_conv3x3_winograd_f32,
// chosen from the layer SHAPE and the arithmetic class the caller allows (cnl_conv_params.algo) — never from the batch size:
//   2  winograd2.hip  F(2x2,3x3) on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), 8x16-pixel blocks, two workgroups per CU:
//                     short channel loops (Cin < 128), Cin % 16 != 0, and everything under CNL_ALGO_F32;
//   5  winograd5.hip  F(2x2,3x3) with every fp32 product formed on the fp16 matrix cores (scaled two-way fp16 split, three cross
//                     terms, fp32 accumulation: error at or below the fp32 MFMA's), 16x16-pixel x 64-cout work items;
//   6  winograd6.hip  the same on 8x16-pixel x 128-cout work items (short channel loop with many couts, or maps that 16-row blocks pad);
//   9  winograd9.hip  1-D F(2,3) along x with the kernel rows folded into the reduction, same split: 8-row x 64-pixel x 64-cout items.
//  10  winograd10.hip the same arithmetic (and weights) on 4-row x 64-pixel x 64-cout items, two workgroups per CU: 16-pixel-wide maps;
//  11                 ... on 4-row x 64-pixel x 32-cout items: 16-pixel maps with Cout <= 256, and the latency class (CNL_ALGO_LATENCY).
//   (8, the F(4x4,3x3) split kernel of round 2, was removed in ABI v10: slower than 9 at 4x its rounding error.)
// The 16x16-pixel-block fp32 kernel this file used to hold, the exact three-way bf16 split (winograd3/4) and the two-waves-per-SIMD
// form of 5 (winograd7) are measured-and-superseded variants: tools/experiments/ (`make -C csrc experiments`, algo = CNL_ALGO_FORCE + variant).
#include "cnl_common.h"
#ifdef CNL_W9_VGPR_SPILLS_OVERRIDE      /* `make variant`: the spill count of THAT build's winograd9.hip */
#define CNL_W9_VGPR_SPILLS CNL_W9_VGPR_SPILLS_OVERRIDE
#elif __has_include("build/w9_usage.h")
#include "build/w9_usage.h"      // CNL_W9_VGPR_SPILLS: vector registers the compiler spilled in winograd9.hip's kernels (Makefile)
#else
#define CNL_W9_VGPR_SPILLS 0
#endif

namespace cnl_wino {

// U = G g G^T per (co, ci), packed [Cin/8][16][CoutP][8]; rows co >= Cout are zero.
__global__ __launch_bounds__(256) void winograd_weights_kernel(const float* __restrict__ w, float* __restrict__ u, int Cin, int Cout,
                                                               int CoutP) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)CoutP * Cin) return;
    const int ci = (int)(t % Cin), co = (int)(t / Cin);
    float g[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) g[i][j] = co < Cout ? w[((long)co * 9 + i * 3 + j) * Cin + ci] : 0.f;   // OHWI
    float h[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        h[0][j] = g[0][j];
        h[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
        h[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
        h[3][j] = g[2][j];
    }
    const int cc = ci >> 3, c8 = ci & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float uu[4] = {h[i][0], 0.5f * (h[i][0] + h[i][1] + h[i][2]), 0.5f * (h[i][0] - h[i][1] + h[i][2]), h[i][2]};
#pragma unroll
        for (int j = 0; j < 4; ++j) u[((((long)cc * 16 + (i * 4 + j)) * CoutP + co) * 8) + c8] = uu[j];
    }
}

}  // namespace cnl_wino
using namespace cnl_wino;

int cnl_wino2_launch(const cnl_conv_params* p, size_t u_floats, void* stream);     // winograd2.hip
size_t cnl_wino5_weight_bytes(int Cin, int Cout);                                  // winograd5.hip
size_t cnl_wino5_scalar_floats();
int cnl_wino5_transform_weights(const float* w_ohwi, const float* u_f32, size_t u_f32_floats, void* u5, float* scal, int Cin, int Cout, void* stream);
int cnl_wino5_own_absmax(const cnl_conv_params* p, float* scal, void* stream);
int cnl_wino5_launch(const cnl_conv_params* p, const void* u5, float* scal, void* stream);
int cnl_wino6_launch(const cnl_conv_params* p, const void* u5, float* scal, void* stream);        // winograd6.hip
size_t cnl_wino9_weight_bytes(int Cin, int Cout);                                  // winograd9.hip
size_t cnl_wino9_scalar_floats(int Cin, int Cout);
int cnl_wino9_transform_weights(const float* w_ohwi, void* u9, float* isu, int Cin, int Cout, void* stream);
bool cnl_wino9_eligible(const cnl_conv_params* p);
size_t cnl_wino9_up_weight_bytes(int Cin, int Cout);                               // the row-pair weight sets of a conv behind a folded upsample (cnl_conv_params.w_up)
int cnl_wino9_up_transform_weights(const float* w_ohwi, void* u9, float* isu, int Cin, int Cout, void* stream);
int cnl_wino9_launch(const cnl_conv_params* p, const void* u9, const float* isu, const float* xmax, void* stream);
int cnl_wino_packed_stride(const cnl_conv_params* p);                               // packed rows of the row kernels (winograd9.hip)
bool cnl_wino10_eligible(const cnl_conv_params* p);                                // winograd10.hip (reads winograd9.hip's weights)
size_t cnl_wino13_weight_bytes(int Cin, int Cout);                                 // winograd13.hip: F(4,3) along x (weights of its own: six transform positions)
size_t cnl_wino13_scalar_floats(int Cin, int Cout);
int cnl_wino13_transform_weights(const float* w_ohwi, void* u13, float* isu, int Cin, int Cout, void* stream);
bool cnl_wino13_eligible(const cnl_conv_params* p);
int cnl_wino13_launch(const cnl_conv_params* p, const void* u13, const float* isu, const float* xmax, void* stream);
int cnl_wino10_launch(const cnl_conv_params* p, const void* u9, const float* isu, const float* xmax, bool cout32, void* stream);
#ifdef CNL_EXPERIMENTS
bool cnl_wino12_eligible(const cnl_conv_params* p);                                // tools/experiments/winograd12.hip (round 5: Cin = 64, the epilogue rides in the next item's chunks)
int cnl_wino12_launch(const cnl_conv_params* p, const void* u9, const float* isu, const float* xmax, void* stream);
int cnl_wino1_launch(const cnl_conv_params* p, void* stream);                       // tools/experiments/winograd1.hip
size_t cnl_wino3_weight_bytes(int Cin, int Cout);                                  // tools/experiments/winograd3.hip
int cnl_wino3_transform_weights(const float* w_ohwi, void* u3, int Cin, int Cout, void* stream);
int cnl_wino3_launch(const cnl_conv_params* p, const void* u3, void* stream);
int cnl_wino4_launch(const cnl_conv_params* p, const void* u3, void* stream);        // tools/experiments/winograd4.hip
int cnl_wino7_launch(const cnl_conv_params* p, const void* u5, float* scal, void* stream);        // tools/experiments/winograd7.hip
#else
static size_t cnl_wino3_weight_bytes(int, int) { return 0; }
#endif

// Layout of the transformed-weight buffer (floats): [fp32 U = [ci/8][16][CoutP][8]] [experiment builds: bf16 x 3 pieces]
// [fp16 x 2 pieces of F(2x2)] [its scalars] [fp16 x 2 pieces of the row-Winograd kernel] [its per-cout scales] [fp16 x 2 pieces of the F(4,3) row
// kernel] [its per-cout scales]; the split copies exist for Cin % 16 == 0 (rows: Cin % 32 == 0) only.
static size_t wino_f32_floats(int Cin, int Cout) {
    const size_t CoutP = (size_t)((Cout + 63) / 64) * 64;
    return (size_t)(Cin / 8) * 16 * CoutP * 8;
}
struct WeightLayout {
    size_t u3, u5, s5, u9, s9, u13, s13, total;     // float offsets
    WeightLayout(int Cin, int Cout) {
        const bool split = Cin % 16 == 0;
        u3 = wino_f32_floats(Cin, Cout);
        u5 = u3 + cnl_wino3_weight_bytes(Cin, Cout) / 4;
        s5 = u5 + cnl_wino5_weight_bytes(Cin, Cout) / 4;
        u9 = s5 + (split ? cnl_wino5_scalar_floats() : 0);
        s9 = u9 + cnl_wino9_weight_bytes(Cin, Cout) / 4;
        u13 = s9 + cnl_wino9_scalar_floats(Cin, Cout);
        s13 = u13 + cnl_wino13_weight_bytes(Cin, Cout) / 4;
        total = s13 + cnl_wino13_scalar_floats(Cin, Cout);
    }
};

extern "C" size_t cnl_winograd_weight_floats(int32_t Cin, int32_t Cout) {
    if (Cin <= 0 || Cout <= 0 || Cin % 8) return 0;
    return WeightLayout(Cin, Cout).total;
}
extern "C" int cnl_winograd_transform_weights_f32(const float* w_ohwi, float* u, int32_t Cin, int32_t Cout, void* stream) {
    CNL_REQUIRE(w_ohwi && u, CNL_E_BAD_ARG, "cnl_winograd_transform_weights_f32: null pointer");
    CNL_REQUIRE(Cin > 0 && Cout > 0 && Cin % 8 == 0, CNL_E_UNSUPPORTED, "cnl_winograd_transform_weights_f32: Cin %% 8 != 0");
    const int CoutP = (Cout + 63) / 64 * 64;
    const long total = (long)CoutP * Cin;
    hipLaunchKernelGGL(winograd_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_ohwi, u, Cin,
                       Cout, CoutP);
    int rc = cnl::check_launch("winograd_weights_kernel");
    if (rc != CNL_OK || Cin % 16) return rc;
    const WeightLayout L(Cin, Cout);
#ifdef CNL_EXPERIMENTS
    rc = cnl_wino3_transform_weights(w_ohwi, u + L.u3, Cin, Cout, stream);
    if (rc != CNL_OK) return rc;
#endif
    rc = cnl_wino5_transform_weights(w_ohwi, u, L.u3, u + L.u5, u + L.s5, Cin, Cout, stream);
    if (rc != CNL_OK) return rc;
    if (rc != CNL_OK || Cin % 32) return rc;
    rc = cnl_wino9_transform_weights(w_ohwi, u + L.u9, u + L.s9, Cin, Cout, stream);
    if (rc != CNL_OK) return rc;
    return cnl_wino13_transform_weights(w_ohwi, u + L.u13, u + L.s13, Cin, Cout, stream);
}

extern "C" size_t cnl_winograd_up_weight_floats(int32_t Cin, int32_t Cout) {
    const size_t b = cnl_wino9_up_weight_bytes(Cin, Cout);
    return b ? b / 4 + cnl_wino9_scalar_floats(Cin, Cout) : 0;
}
extern "C" int cnl_winograd_transform_weights_up_f32(const float* w_ohwi, float* u_up, int32_t Cin, int32_t Cout, void* stream) {
    CNL_REQUIRE(w_ohwi && u_up, CNL_E_BAD_ARG, "cnl_winograd_transform_weights_up_f32: null pointer");
    CNL_REQUIRE(((uintptr_t)u_up & 15) == 0, CNL_E_BAD_ARG, "cnl_winograd_transform_weights_up_f32: u_up must be 16-byte aligned");
    const size_t b = cnl_wino9_up_weight_bytes(Cin, Cout);
    CNL_REQUIRE(b, CNL_E_UNSUPPORTED, "cnl_winograd_transform_weights_up_f32: Cin %% 32 != 0");
    return cnl_wino9_up_transform_weights(w_ohwi, u_up, u_up + b / 4, Cin, Cout, stream);
}

// half the CUs of the current device (MI355X: 128) — the small-grid threshold of the row kernels; a process without a device (the CPU tests of
// the dispatcher) gets the MI355X value
static long long half_the_cus() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || cnl::cu_count(dev, &n) != CNL_OK || n <= 0) {
        (void)hipGetLastError();
        n = 256;
    }
    return n / 2;
}

// which kernel a layer takes (see the file header); CNL_ALGO_FORCE + v pins variant v wherever it can run at all
static int wino_choice(const cnl_conv_params* p) {
    const int upf = (p->flags & CNL_UPSAMPLE_IN) ? 2 : 1;
    const int H = p->H_in * upf, W = p->W_in * upf, CoutP = (p->Cout + 63) / 64 * 64;
    const int items_per_image = ((H + 15) / 16) * ((W + 15) / 16) * (CoutP / 64);
    if (p->algo >= CNL_ALGO_FORCE) {
        const int v = (int)p->algo - CNL_ALGO_FORCE - (p->algo >= CNL_ALGO_FORCE + 32 ? 32 : 0);     // (+ 32: the row kernels without packed rows)
        if (v <= 2 || p->Cin % 16) return v == 1 ? 1 : 2;
        if (v == 9 && !cnl_wino9_eligible(p)) return 5;
        if ((v == 10 || v == 11) && !cnl_wino10_eligible(p)) return 5;
        if (v == 13 && !cnl_wino13_eligible(p)) return cnl_wino9_eligible(p) ? 9 : 5;
#ifdef CNL_EXPERIMENTS
        if (v == 12 && !cnl_wino12_eligible(p)) return cnl_wino9_eligible(p) ? 9 : 5;
#else
        if (v == 12) return cnl_wino9_eligible(p) ? 9 : 5;      // (variant 12 exists in experiment builds only)
#endif
        if (v == 6 && p->Cout % 128) return 5;
        return v;
    }
    if (p->algo == CNL_ALGO_F32 || p->Cin % 16) return 2;
    const long long area = (long long)H * W;
    // the latency class (one-image batches, BASELINE C0): 4-row x 64-pixel x 32-cout row-Winograd items, two workgroups per CU — four times the
    // work items of winograd9's (a 32 x 32 map of one image: 64 instead of 16).  Measured at N = 1 against the default choice
    // (profiles/r04_winograd_variants.txt): layer1 22 -> 16 us, layer2 29 -> 16, layer3 46 -> 23, layer4 61 -> 38, 512 -> 256 @16x16 69 -> 36;
    // not behind a folded upsample (64 -> 512 first head blocks: 44 us on winograd9, 60 here).  The caller's option takes EVERY eligible layer there
    // (also those whose default is winograd5 / 6 or the fp32 kernel: other arithmetic); the default plan moves only winograd9's own layers (below).
    if (p->algo == CNL_ALGO_LATENCY && upf == 1 && cnl_wino10_eligible(p) && !p->fuse_w) return 11;
    // 16-pixel-wide maps (four images side by side in a block row): the half-height items of winograd10.hip give the chip twice the work
    // items of winograd9's and a second workgroup per CU to overlap with (long channel loops: what was measured) — 512 -> 512 @16x16 x 32: 83 us (winograd5: 90-93, winograd9: 97-107),
    // 512 -> 256: 61-65 us with 32-cout items (fp32 kernel: 89-92)
    if (upf == 1 && W == 16 && p->Cin >= 256 && cnl_wino10_eligible(p) && !p->fuse_w) return p->Cout <= 256 ? 11 : 10;
    // the F(4,3) class (CNL_ALGO_F43, winograd13.hip): long channel loops on maps that 4-row x 128-pixel items (packed rows included) pad by at most 1.35 x —
    // the head blocks of 512 x 512 and 608 x 1088 frames.  A function of the shape alone (the padding a long virtual row has, as below).
    if (p->algo == CNL_ALGO_F43 && p->Cin >= 128 && W >= 128 && cnl_wino13_eligible(p)) {      // (narrower maps, packed: measured slower than the F(2,3) kernels — C1 with layer2 / layer3 on this kernel 7.48 -> 8.52 ms)
        const bool packable13 = W % 2 == 0 && W >= 28 && W % 128 != 0;
        const long long wpad13 = packable13 ? (W / 4 * 4 + 4) : ((W + 127) / 128 * 128);
        if ((long long)((H + 3) / 4 * 4) * wpad13 * 100 <= area * 135) return 13;
    }
    // row-Winograd (winograd9.hip): 8-row x 64-pixel x 64-cout work items.  Measured against kernels 2 / 5 / 6 on every 3x3 shape of the
    // three configurations (profiles/r03_winograd9_variants.txt): 0.5-0.8x their time wherever its blocks pad the map by less than ~1.5x
    // (maps at least ~44 pixels wide), channel loops from 32 up, with or without residual / folded upsample
    if (cnl_wino9_eligible(p)) {
        // 32-pixel-wide maps: two images side by side in a block row (a function of the shape alone).  16-pixel-wide maps could take four
        // (the kernel does it when forced), but there kernels 5 / 6 win: 512 -> 512 @16x16 85 vs 110 us, 512 -> 256 77 vs 98
        const int side = (upf == 1 && W == 32) ? 2 : 1;
        // widths that are not a multiple of 64 (34, 68, 136, 272: the maps of 608 x 1088 frames): packed rows — the images of the launch side by
        // side in one virtual row, W + 2 columns each (cnl_wino_packed_stride, winograd9.hip).  The CLASS is chosen from the shape alone, with the
        // padding a long virtual row has ((W + 2) / W); whether a launch then packs (its N decides) cannot change a bit of the result.
        const bool packable = W % 2 == 0 && W >= 14 && !(upf == 1 && (W == 32 || W == 16)) && W % 64 != 0;
        const int R9 = (H + 7) / 8 * 8, R10 = (H + 3) / 4 * 4;
        const long long wpad = packable ? (W + 2) : ((W * side + 63) / 64 * 64);
        const long long pad9 = (long long)R9 * wpad, pad10 = (long long)R10 * wpad;
        if ((pad9 < pad10 ? pad9 : pad10) * 100 <= area * side * 150) {
            // Small grids (small batches: BASELINE C0 is one image): when winograd9's 8-row x 64-cout items number at most half the chip's
            // CUs, the SAME arithmetic runs on 4-row x 32-cout items (winograd10.hip, variant 11: four times the items, two workgroups
            // per CU) — bit for bit the same output (tests/test_gpu_conv.py), so a shard and the full batch still agree exactly although
            // they may take different work-item shapes.  Measured (profiles/r04_small_batch_variants.txt), items of winograd9 -> us 9 / 11:
            // 8: 48 / 24, 16: 32 / 18, 32: 25 / 18, 64: 33 / 22, 128: 63 / 56 (head block), 37 / 30, 28 / 24; 256: 82 / 92 -> stays.
            if (p->fuse_w) return 9;       // a folded 1x1 conv (cnl_conv_params.fuse_w): winograd9's epilogue has it, whatever the grid size
            const int pk = cnl_wino_packed_stride(p);
            const long long bx = pk ? ((long long)p->N * pk + 63) / 64 : (long long)((p->N + side - 1) / side) * ((W * side + 63) / 64);
            const long long items9 = bx * (R9 / 8) * (CoutP / 64);
            if (items9 <= half_the_cus() && upf == 1 && cnl_wino10_eligible(p)) return 11;
            // maps whose height 8-row items pad by a sixth or more over 4-row items (19 rows: 24 vs 20): the half-height items
            if (upf == 1 && R9 * 100 >= R10 * 115 && cnl_wino10_eligible(p)) return p->Cout <= 256 ? 11 : 10;
            // a winograd9 that the compiler could only build with spilled vector registers (scratch loads inside its chunk loop): its layers go
            // to winograd10 — the same bits from 128 accumulators per lane (no folded upsample there: those launches keep winograd9)
            if (CNL_W9_VGPR_SPILLS > 0 && upf == 1 && cnl_wino10_eligible(p)) return 10;
            return 9;
        }
    }
    if (items_per_image >= 8 && (p->Cin >= 128 || p->Cout >= 256)) {      // (Cin 64 -> 256 / 512 / 768: the first head blocks, per head or fused)
        // the 8x16-pixel x 128-cout work items of winograd6.hip: where the channel loop is short and the couts many, and on maps
        // that 16-row blocks pad more than 8-row blocks (19x34, 38x68, 152x272 of 608x1088 frames: -1 .. -4 %)
        const long long pad16 = (long long)((H + 15) / 16 * 16) * ((W + 15) / 16 * 16), pad8 = (long long)((H + 7) / 8 * 8) * ((W + 15) / 16 * 16);
        return (p->Cout % 128 == 0 && (p->Cin < 128 || pad8 < pad16)) ? 6 : 5;
    }
    return 2;
}

extern "C" int cnl_conv3x3_winograd_kernel(const cnl_conv_params* p) {
    CNL_REQUIRE(p, CNL_E_BAD_ARG, "cnl_conv3x3_winograd_kernel: null params");
    CNL_REQUIRE(p->H_in > 0 && p->W_in > 0 && p->Cin > 0 && p->Cout > 0, CNL_E_BAD_ARG, "cnl_conv3x3_winograd_kernel: non-positive dimension");
    const int c = wino_choice(p);
    return (c == 5 || c == 6 || c == 7 || c == 9 || c == 10 || c == 11 || c == 12 || c == 13) ? CNL_WINO_F16X2 : (c == 3 || c == 4) ? CNL_WINO_BF16X3 : CNL_WINO_F32;
}

extern "C" int cnl_conv3x3_winograd_variant(const cnl_conv_params* p) {
    CNL_REQUIRE(p, CNL_E_BAD_ARG, "cnl_conv3x3_winograd_variant: null params");
    CNL_REQUIRE(p->H_in > 0 && p->W_in > 0 && p->Cin > 0 && p->Cout > 0, CNL_E_BAD_ARG, "cnl_conv3x3_winograd_variant: non-positive dimension");
    return wino_choice(p);
}

extern "C" int cnl_conv3x3_winograd_f32(const cnl_conv_params* p, void* stream) {
    CNL_REQUIRE(p, CNL_E_BAD_ARG, "cnl_conv3x3_winograd_f32: null params");
    CNL_REQUIRE(p->x && p->w && p->bias && p->y, CNL_E_BAD_ARG, "cnl_conv3x3_winograd_f32: null tensor pointer");
    CNL_REQUIRE(p->N > 0 && p->H_in > 0 && p->W_in > 0 && p->Cin > 0 && p->Cout > 0, CNL_E_BAD_ARG,
                "cnl_conv3x3_winograd_f32: non-positive dimension");
    CNL_REQUIRE(p->KH == 3 && p->KW == 3 && p->stride == 1 && p->pad == 1, CNL_E_UNSUPPORTED,
                "cnl_conv3x3_winograd_f32: only 3x3 / stride 1 / pad 1");
    CNL_REQUIRE(!(p->flags & (CNL_UPSAMPLE_OUT_ADD | CNL_SIGMOID | CNL_RELU6)), CNL_E_UNSUPPORTED,
                "cnl_conv3x3_winograd_f32: UPSAMPLE_OUT_ADD / SIGMOID are handled by cnl_conv2d_nhwc_f32");
    CNL_REQUIRE(p->Cin % 8 == 0 && p->ldx % 4 == 0 && p->ldx >= p->Cin && p->ldy >= p->Cout, CNL_E_UNSUPPORTED,
                "cnl_conv3x3_winograd_f32: Cin %% 8 != 0 or bad pixel stride");
    CNL_REQUIRE(((uintptr_t)p->x & 15) == 0 && ((uintptr_t)p->w & 15) == 0, CNL_E_BAD_ARG, "cnl_conv3x3_winograd_f32: unaligned x / u");
    CNL_REQUIRE(!p->residual || p->ldr >= p->Cout, CNL_E_BAD_ARG, "cnl_conv3x3_winograd_f32: ldr < Cout");
    CNL_REQUIRE(p->algo <= CNL_ALGO_F32 || p->algo == CNL_ALGO_LATENCY || p->algo == CNL_ALGO_F43 || (p->algo >= CNL_ALGO_FORCE && p->algo <= CNL_ALGO_FORCE + 13 && p->algo != CNL_ALGO_FORCE + 8) ||
                    (p->algo >= CNL_ALGO_FORCE + 32 + 9 && p->algo <= CNL_ALGO_FORCE + 32 + 13),
                CNL_E_BAD_ARG, "cnl_conv3x3_winograd_f32: unknown algo %u (FORCE + 8, the F(4x4) kernel, was removed in ABI v10)", p->algo);
    const int choice = wino_choice(p);
    CNL_REQUIRE(!p->fuse_w || (choice == 9 && p->fuse_part && !p->residual), CNL_E_UNSUPPORTED,
                "cnl_conv3x3_winograd_f32: fuse_w needs a launch the row-Winograd kernel (variant 9) takes, fuse_part and no residual (this one: variant %d)", choice);
    const WeightLayout L(p->Cin, p->Cout);
    float* u = const_cast<float*>(p->w);
    if (choice == 5 || choice == 6 || choice == 7 || choice >= 9) {
        float* s5 = u + L.s5;
        if (choice >= 9) {
            // the per-image maxima: handed over by the producer, else one pass over the input (stream-ordered, scratch = the F(2x2)
            // scalars of this layer: one launch at a time per layer and stream, see the header)
            const float* xmax = p->x_absmax;
            if (!xmax) {
                const int rc = cnl_wino5_own_absmax(p, s5, stream);
                if (rc != CNL_OK) return rc;
                xmax = s5 + 16;
            }
            if (choice == 9) return cnl_wino9_launch(p, u + L.u9, u + L.s9, xmax, stream);
            if (choice == 13) return cnl_wino13_launch(p, u + L.u13, u + L.s13, xmax, stream);
#ifdef CNL_EXPERIMENTS
            if (choice == 12) return cnl_wino12_launch(p, u + L.u9, u + L.s9, xmax, stream);
#endif
            return cnl_wino10_launch(p, u + L.u9, u + L.s9, xmax, choice == 11, stream);
        }
#ifdef CNL_EXPERIMENTS
        if (choice == 7) return cnl_wino7_launch(p, u + L.u5, s5, stream);
#endif
        return (choice == 6 ? cnl_wino6_launch : cnl_wino5_launch)(p, u + L.u5, s5, stream);
    }
#ifdef CNL_EXPERIMENTS
    if (choice == 4) return cnl_wino4_launch(p, u + L.u3, stream);
    if (choice == 3) return cnl_wino3_launch(p, u + L.u3, stream);
    if (choice == 1) return cnl_wino1_launch(p, stream);
#else
    CNL_REQUIRE(choice == 2, CNL_E_UNSUPPORTED, "cnl_conv3x3_winograd_f32: variant %d exists in experiment builds only (make experiments)", choice);
#endif
    return cnl_wino2_launch(p, L.u3, stream);
}
