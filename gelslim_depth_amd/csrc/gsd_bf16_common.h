// gsd_bf16_common.h -- shared pieces of the bf16 mixed-precision path (BASELINE.json configs[4]).
//
// Layout: activations NHWC bf16 ("pixel-major": the C channels of one pixel are contiguous, `pitch` elements from one
// pixel to the next, so a tensor may be a channel slice of a wider buffer -- that is how torch.cat([skip, up]) exists
// without ever being copied).  v_mfma_f32_16x16x32_bf16 wants 8 consecutive k per lane; with k = input channel that is
// one 16-byte piece of a pixel, which LDS-DMA (global_load_lds_dwordx4, per-lane source address) moves without touching
// a register.  Weight gradients need k = pixel instead: those tiles are read back with ds_read_b64_tr_b16.
#pragma once
#include "gsd_common.h"
#include "gsd_bf16.h"

typedef unsigned short u16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// D[row=(lane>>4)*4+reg][col=lane&15] += sum_k A[row i=lane&15][k=8*(lane>>4)+e] * B[k][col j=lane&15]
__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ float bf16_to_f32(u16 v) { return __uint_as_float((unsigned)v << 16); }
// round-to-nearest-even through the hardware convert (v_cvt_pk_bf16_f32 keeps NaN a NaN)
__device__ __forceinline__ u16 f32_to_bf16(float v) {
  const __bf16 h = (__bf16)v;
  return __builtin_bit_cast(u16, h);
}
// two values in ONE v_cvt_pk_bf16_f32 (the vector convert; two scalar converts cost two of them plus a shift and an or -- same bits)
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}

// ReLU of two packed bf16 values: as 16-bit integers a bf16 is negative exactly when its sign bit is set, so a packed signed
// max with 0 clears the negative ones (and turns -0 into +0, as fmaxf(x, 0) does); one instruction for two values
__device__ __forceinline__ unsigned relu_pk_bf16(unsigned v) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  const s16x2 r = __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), s16x2{0, 0});
  return __builtin_bit_cast(unsigned, r);
}

// Whole-line stores of a 16-pixel x 64-channel MFMA result (bf16 NHWC, 128 B a pixel).  The MFMA leaves lane (g = lane>>4,
// j = lane&15) with channels 8g..8g+7 (`lo`) and 32+8g.. (`hi`) of pixel j: stored as they lie, every instruction writes sixteen
// 64-byte HALF lines, and two half-line writes of a 128-byte line reach HBM at about half the rate of whole-line ones (a plain
// fill of 1.12 GB: 6.9 TB/s; gsd_bf16_conv3x3_first storing 560 MB that way: 3.35 TB/s).  One DPP rotate by 8 lanes inside each
// row of 16, written only to half of the row (bank_mask), hands lanes j >= 8 the `hi` piece of pixel j-8 and lanes j < 8 the `lo`
// piece of pixel j+8: instruction A then writes pixels 0..7 of the tile as whole 128-byte lines, instruction B pixels 8..15.
//   lane's pixel inside the tile: A: j & 7, B: 8 + (j & 7); channel offset of its 16 bytes: (j < 8 ? 0 : 32) + 8g  (both: gsd_line_*)
__device__ __forceinline__ void gsd_line_pieces(const unsigned (&lo)[4], const unsigned (&hi)[4], u32x4& a, u32x4& b) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // row_ror:8 = 0x128; bank_mask 0xC: only lanes 8..15 of each row take the rotated value, 0x3: only lanes 0..7
    a[i] = (unsigned)__builtin_amdgcn_update_dpp((int)lo[i], (int)hi[i], 0x128, 0xf, 0xC, false);
    b[i] = (unsigned)__builtin_amdgcn_update_dpp((int)hi[i], (int)lo[i], 0x128, 0xf, 0x3, false);
  }
}
__device__ __forceinline__ int gsd_line_pixel(int j) { return j & 7; }                       // (+ 8 for instruction B)
__device__ __forceinline__ int gsd_line_channel(int g, int j) { return (j < 8 ? 0 : 32) + 8 * g; }

// One LDS-DMA instruction (global_load_lds_dwordx4: lane i's 16 bytes to lds + 16 i) that the COMPILER does not see.  For kernels
// that read LDS through an intrinsic without a memory operand (ds_read_b64_tr_b16) while a fill of the other image is in flight:
// hipcc then assumes the read may alias every outstanding __builtin_amdgcn_global_load_lds and puts s_waitcnt vmcnt(0) in front
// of each read -- every fill instruction stalls for its own round trip.  The caller orders fills and reads itself
// (gsd_dma_barrier: explicit vmcnt(0) + barrier); untracked fills only make the compiler's own vmcnt waits more conservative.
// `lds` must be wave-uniform (M0 holds the LDS base; the compiler re-loads M0 before every use of its own).
__device__ __forceinline__ void gsd_dma16_untracked(const void* g, void* lds) {
  const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds;
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(a), "v"(g));
}

struct NhwcD {
  u16* p;
  long long pitch;
  int N, H, W, C;
};
static inline NhwcD to_nhwc(const gsd_nhwc& t) {
  NhwcD d;
  d.p = (u16*)t.ptr; d.pitch = t.pitch; d.N = t.N; d.H = t.H; d.W = t.W; d.C = t.C;
  return d;
}
static inline int gsd_check_nhwc(const gsd_nhwc* t, const char* what) {
  GSD_REQUIRE(t != nullptr && t->ptr != nullptr, GSD_ERR_BAD_ARG, "%s: null tensor", what);
  GSD_REQUIRE(t->N > 0 && t->H > 0 && t->W > 0 && t->C > 0, GSD_ERR_BAD_ARG, "%s: bad dims N=%d H=%d W=%d C=%d", what, t->N,
              t->H, t->W, t->C);
  GSD_REQUIRE(t->pitch >= t->C, GSD_ERR_BAD_ARG, "%s: pitch %lld < C %d", what, (long long)t->pitch, t->C);
  GSD_REQUIRE(((uintptr_t)t->ptr & 15) == 0 && (t->pitch & 7) == 0, GSD_ERR_UNSUPPORTED,
              "%s: base must be 16-byte aligned and pitch a multiple of 8 elements", what);
  GSD_REQUIRE((long long)t->H * t->W * t->pitch < 2147483647LL, GSD_ERR_UNSUPPORTED, "%s: one image exceeds 2^31 elements",
              what);
  return 0;
}

// gsd_bf16_ctgemm.hip: the large-tile kernel behind gsd_bf16_conv_dense for the transposed convolutions' forward and dX
bool gsd_ctgemm_shape(int N, int H, int W, int K, int M, int ntaps, int stride, int scatter_cs);
bool gsd_ctgemm_operands(const gsd_nhwc* in, const gsd_nhwc* out, const gsd_bf16_bnbwd* bw, int ntaps, const int* ty, const int* tx, int H,
                         int W);
int gsd_ctgemm_partial_rows(int N, int H, int W, int M);
int gsd_ctgemm_launch(const gsd_nhwc* in, const void* wt, const gsd_nhwc* out, int K, int M, int ntaps, const int* ty, const int* tx, int H,
                      int W, int scatter_cs, int oy, int ox, const float* bias, float* partials, const gsd_bf16_bnbwd* bw, void* stream);
