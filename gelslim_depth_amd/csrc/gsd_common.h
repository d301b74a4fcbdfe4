// gsd_common.h -- shared device helpers for libgsd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include "gsd.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- host-side error plumbing (thread-local message, see gsd_api.hip) ----------------------
void gsd_set_error(const char* fmt, ...);
#define GSD_REQUIRE(cond, code, ...)                 \
  do {                                               \
    if (!(cond)) {                                   \
      gsd_set_error(__VA_ARGS__);                    \
      return (code);                                 \
    }                                                \
  } while (0)
#define GSD_LAUNCH_CHECK(what)                                                       \
  do {                                                                               \
    hipError_t e_ = hipGetLastError();                                               \
    if (e_ != hipSuccess) {                                                          \
      gsd_set_error("%s: launch failed: %s", (what), hipGetErrorString(e_));         \
      return GSD_ERR_HIP;                                                            \
    }                                                                                \
  } while (0)

// ---- device-side copies of the ABI descriptors (plain structs, passed by value in kernargs) --
struct SrcD {
  const float* p;
  const float* scale;
  const float* shift;
  int C, H, W, oh, ow, relu;
  int ws;   // row pitch in elements (>= W)
  long long ns, cs;
};
struct DstD {
  float* p;
  int C, H, W, oh, ow;
  int ws;   // row pitch in elements (>= W)
  long long ns, cs;
};

static inline SrcD to_srcd(const gsd_src& s) {
  SrcD d;
  d.p = s.ptr; d.scale = s.scale; d.shift = s.shift;
  d.C = s.C; d.H = s.H; d.W = s.W; d.oh = s.off_h; d.ow = s.off_w; d.relu = s.relu;
  d.ws = s.w_stride;
  d.ns = s.n_stride; d.cs = s.c_stride;
  return d;
}
static inline DstD to_dstd(const gsd_dst& s) {
  DstD d;
  d.p = s.ptr; d.C = s.C; d.H = s.H; d.W = s.W; d.oh = s.off_h; d.ow = s.off_w;
  d.ws = s.w_stride;
  d.ns = s.n_stride; d.cs = s.c_stride;
  return d;
}
static inline SrcD null_srcd() {
  SrcD d;
  d.p = nullptr; d.scale = nullptr; d.shift = nullptr;
  d.C = 0; d.H = 0; d.W = 0; d.oh = 0; d.ow = 0; d.relu = 0; d.ws = 0; d.ns = 0; d.cs = 0;
  return d;
}
static inline DstD null_dstd() {
  DstD d;
  d.p = nullptr; d.C = 0; d.H = 0; d.W = 0; d.oh = 0; d.ow = 0; d.ws = 0; d.ns = 0; d.cs = 0;
  return d;
}

// Deferred BatchNorm + ReLU applied to a loaded raw value (channel c of segment s).
__device__ __forceinline__ float apply_affine(float v, float sc, float sh, int relu) {
  v = fmaf(v, sc, sh);
  return relu ? fmaxf(v, 0.f) : v;
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  // v_mfma_f32_16x16x4_f32: A[i=lane&15][k=lane>>4], B[k=lane>>4][j=lane&15],
  // D[row=(lane>>4)*4+reg][col=lane&15]  (cdna_hip_programming.md sec. 3)
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Barrier that PUBLISHES LDS-DMA fills: global_load_lds completes through vmcnt, and hipcc's own wait in front of
// __syncthreads() / the following ds_read depends on its alias analysis of the DMA destination against the reads -- with
// double-buffered images it can prove "different buffer" inside one loop body and then drops the wait across the
// back-edge (observed: s_barrier with no s_waitcnt vmcnt(0) in gconv_bf16_kernel<1,..>, stale tile rows).  So the wait is
// written out.  s_waitcnt immediate on gfx9: vmcnt = bits[3:0] | bits[15:14] << 4, expcnt = bits[6:4], lgkmcnt = bits[11:8].
__device__ __forceinline__ void gsd_dma_barrier() {
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
  __syncthreads();
}

// Logical block id such that every XCD (blocks b, b+8, b+16, ... share one) owns a contiguous id range.
__device__ __forceinline__ int xcd_swizzle(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7;
  const int xcd = bid & 7, slot = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// sum over the 16 lanes that share lane>>4 (one MFMA column group)
__device__ __forceinline__ float reduce16(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}
// Same sum with DPP row shifts (4 VALU instructions, no LDS traffic; __shfl_xor lowers to ds_bpermute): the total of a
// 16-lane row lands in its LAST lane (lane & 15 == 15); the other lanes hold partial sums.
__device__ __forceinline__ float reduce16_to_lane15(float v) {
  // row_shr:n = 0x110 + n; lanes shifted in from outside the row read 0 (bound_ctrl)
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// pitched_ok: the kernel addresses rows through w_stride; otherwise the operand must be row-contiguous (w_stride == W)
static inline int gsd_check_src(const gsd_src& s, const char* what, bool pitched_ok = false) {
  GSD_REQUIRE(s.ptr != nullptr, GSD_ERR_BAD_ARG, "%s: null ptr", what);
  GSD_REQUIRE(s.C > 0 && s.H > 0 && s.W > 0, GSD_ERR_BAD_ARG, "%s: bad dims C=%d H=%d W=%d", what, s.C, s.H, s.W);
  GSD_REQUIRE((s.scale == nullptr) == (s.shift == nullptr), GSD_ERR_BAD_ARG, "%s: scale/shift must come together", what);
  GSD_REQUIRE(s.w_stride >= s.W, GSD_ERR_BAD_ARG, "%s: w_stride %d < W %d", what, s.w_stride, s.W);
  GSD_REQUIRE(pitched_ok || s.w_stride == s.W, GSD_ERR_UNSUPPORTED, "%s: this kernel needs row-contiguous data (w_stride %d != W %d)",
              what, s.w_stride, s.W);
  GSD_REQUIRE(s.c_stride >= (int64_t)s.H * s.w_stride && s.n_stride >= s.c_stride, GSD_ERR_BAD_ARG, "%s: strides too small",
              what);
  return 0;
}
static inline int gsd_check_dst(const gsd_dst& s, const char* what, bool pitched_ok = false) {
  GSD_REQUIRE(s.ptr != nullptr, GSD_ERR_BAD_ARG, "%s: null ptr", what);
  GSD_REQUIRE(s.C > 0 && s.H > 0 && s.W > 0, GSD_ERR_BAD_ARG, "%s: bad dims", what);
  GSD_REQUIRE(s.w_stride >= s.W, GSD_ERR_BAD_ARG, "%s: w_stride %d < W %d", what, s.w_stride, s.W);
  GSD_REQUIRE(pitched_ok || s.w_stride == s.W, GSD_ERR_UNSUPPORTED, "%s: this kernel needs row-contiguous data (w_stride %d != W %d)",
              what, s.w_stride, s.W);
  GSD_REQUIRE(s.c_stride >= (int64_t)s.H * s.w_stride && s.n_stride >= s.c_stride, GSD_ERR_BAD_ARG, "%s: strides too small",
              what);
  return 0;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a property of (kernel, DEVICE): remember per device (bit = device
// ordinal) that it has been set.  The only process-wide state of the library: an idempotent launch-attribute cache that no
// result depends on (setting the attribute twice is harmless, so a lost race only repeats the call).
struct gsd_attr_once {
  std::atomic<uint64_t> mask{0};
};
static inline hipError_t gsd_allow_big_lds(gsd_attr_once& once, const void* fn, int bytes = 160 * 1024) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = 1ull << (dev & 63);
  if (dev < 64 && (once.mask.load(std::memory_order_relaxed) & bit)) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess && dev < 64) once.mask.fetch_or(bit, std::memory_order_relaxed);
  return e;
}
// Tuning knobs are read from the environment on EVERY call (no cached statics: a test or an A/B harness may change them
// inside one process, and the library keeps no mutable state that a result depends on).
static inline int gsd_env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v != nullptr ? atoi(v) : dflt;
}
static inline bool gsd_env_set(const char* name) { return getenv(name) != nullptr; }

// entry points that read a gsd_src without gsd_check_src: the operand must be row-contiguous unless the kernel says otherwise
static inline int gsd_require_rows_contiguous(const gsd_src& s, const char* what) {
  GSD_REQUIRE(s.w_stride == s.W, GSD_ERR_UNSUPPORTED, "%s: this kernel needs row-contiguous data (w_stride %d != W %d)", what,
              s.w_stride, s.W);
  return 0;
}

// Compute units of the current device (256 on MI355X), remembered per process: an idempotent cache like gsd_attr_once -- every
// thread computes the same value.  The planners' run-time models deal blocks over this many CUs.
static inline int gsd_cu_count() {
  static std::atomic<int> n{0};
  int v = n.load(std::memory_order_relaxed);
  if (v == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    n.store(v, std::memory_order_relaxed);
  }
  return v;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }
