// Shared by conv_mfma.hip (fp32 matrix cores) and conv_f16x2.hip (fp16 matrix cores, split operands): launch arguments, tile
// configuration and the buffer / LDS-DMA helpers of the direct implicit-GEMM convolution.
#pragma once
#include "cnl_common.h"

namespace cnl_conv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
    const float* x;
    const float* w;
    const float* bias;
    const float* res;
    float* y;
    int N, Hin, Win, Cin, Cout;
    int KH, KW, stride, pad, pad_x;      // pad = rows, pad_x = columns (equal for every public conv; differ for deconv phases)
    int sub_dy, sub_dx;                  // CNL_I_SUBPIXEL: phase of the 2x output grid this launch writes
    int ldx, ldy, ldr;
    int HL, WL;        // logical input size (2x when CNL_UPSAMPLE_IN)
    int Ho, Wo, M;     // conv output size, M = N*Ho*Wo
    int CC, KT, K;     // Cin/32, KH*KW*CC, KH*KW*Cin
    unsigned x_bytes, w_bytes, y_bytes, r_bytes;
    unsigned flags;
    int tiles_n, tiles;
    unsigned mg_hw, sh_hw, mg_w, sh_w;   // magic division by Ho*Wo and by Wo (exact for n < 2^31)
    long long* trace;                    // CNL_TRACE builds only: per-workgroup phase timestamps
    const float* xmax;                   // fp16-split kernel (conv_f16x2.hip): per-image max |x| (N floats), max |w| (1 float),
    const float* wmax;
    unsigned* ymax;                      // and where max |y| per image is folded into (or null)
    unsigned algo;                       // cnl_conv_params.algo (CNL_ALGO_*)
    const float* wscale;                 // sub-pixel phases (SUB): the power-of-two scale of the PRE-SPLIT weights a.w points to; CNL_W_SPLIT: of wsplit
    const float* wsplit = nullptr;       // CNL_W_SPLIT: the weights as scaled fp16 pieces in the kernel's B-row layout (cnl_conv_split_weights_f32)
    // split over the reduction dimension (conv_f16x2.hip, SPLIT): slice s of `ksplit` multiplies chunks [s*kt_per, (s+1)*kt_per) and stores its
    // scaled-back partial sums to part[s][M][Cout]; splitk_reduce_kernel adds them in slice order and applies the epilogue
    int ksplit = 0, kt_per = 0;
    float* part = nullptr;
    unsigned part_bytes = 0;
};

constexpr unsigned CNL_I_SUBPIXEL = 1u << 16;   // internal: y[n, 2oy+sub_dy, 2ox+sub_dx, :] = act(conv + bias) (+ residual there)
constexpr unsigned OOB = 0xFFFFFFF0u;   // voffset that is always >= num_records -> DMA writes zeros / store dropped

template <int WM, int WN, int TM, int TN>
struct Cfg {
    static constexpr int NW = WM * WN;
    static constexpr int THREADS = NW * 64;
    static constexpr int BM = WM * TM * 32;
    static constexpr int BN = WN * TN * 32;
    static constexpr int A_INSTR = BM / (NW * 8);   // buffer_load..lds instructions per wave for A
    static constexpr int B_INSTR = BN / (NW * 8);
    static constexpr int STAGE_BYTES = (BM + BN) * 128;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
    static_assert(NW == 4, "kernel is declared __launch_bounds__(256, 2)");
    static_assert(BM % (NW * 8) == 0 && BN % (NW * 8) == 0, "tile rows must split evenly over waves");
};

typedef __attribute__((address_space(3))) void lds_void;

// amdgcn builtins are wrapped in NON-template device functions: called with template-dependent arguments
// directly inside the kernel template they make hipcc's host pass silently drop the kernel's host stub.
__device__ __forceinline__ void dma16(const float* base, unsigned bytes, char* lds_dst, unsigned voffset, unsigned soffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_dst, 16, voffset, soffset, 0, 0);
}
__device__ __forceinline__ float buf_load(const float* base, unsigned bytes, unsigned voffset, unsigned soffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voffset, soffset, 0));
}
__device__ __forceinline__ void buf_store(float v, float* base, unsigned bytes, unsigned voffset, unsigned soffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, voffset, soffset, CNL_NT_STORES);
}
__device__ __forceinline__ f32x4 buf_load4(const float* base, unsigned bytes, unsigned voffset, unsigned soffset) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, soffset, 0));
}
__device__ __forceinline__ void buf_store4(f32x4 v, float* base, unsigned bytes, unsigned voffset, unsigned soffset) {
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), rsrc, voffset, soffset, CNL_NT_STORES);
}
__device__ __forceinline__ f32x4 lds_read16(const char* p) { return *reinterpret_cast<const f32x4*>(p); }
// n / d for n < 2^31 with host-computed (magic, shift); shift == 0xFF encodes d == 1
__device__ __forceinline__ unsigned fast_div(unsigned n, unsigned magic, unsigned shift) {
    return shift == 0xFFu ? n : (__umulhi(n, magic) >> shift);
}


// conv_f16x2.hip: the same implicit GEMM with each fp32 product formed on the fp16 matrix cores (needs a.xmax and a.wmax)
bool f16x2_eligible(const ConvArgs& a);
int f16x2_launch(const ConvArgs& a, hipStream_t stream);

}  // namespace cnl_conv
