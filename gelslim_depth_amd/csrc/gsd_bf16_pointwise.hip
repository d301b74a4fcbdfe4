// gsd_bf16_pointwise.hip -- the HBM-bound kernels of the bf16 path: weight images, first-layer im2col, BatchNorm apply
// (+ReLU), max-pool, the 1x1 output convolution and BatchNorm/ReLU/max-pool backward.  All tensors NHWC bf16 (gsd_nhwc),
// a thread owns 8 consecutive channels of a pixel (one 16-byte load/store), a wave therefore moves 1 KiB contiguous
// when pitch == C.  All arithmetic in fp32.
#include "gsd_bf16_common.h"

namespace {

__device__ __forceinline__ void unpack8(const uint4 v, float f[8]) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float f[8]) {
  return make_uint4(pack_bf16(f[0], f[1]), pack_bf16(f[2], f[3]), pack_bf16(f[4], f[5]), pack_bf16(f[6], f[7]));
}
__device__ __forceinline__ uint4 ld16(const u16* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void st16(u16* p, uint4 v) { *reinterpret_cast<uint4*>(p) = v; }

// ---- weight images ----------------------------------------------------------------------------------------------
struct WImg {
  int T, M, K, Mp, Kp;
};
WImg wimg_dims(int mode, int Cout, int Cin) {
  WImg d;
  switch (mode) {
    case 0: d.T = 9; d.M = Cout; d.K = Cin; break;           // conv3x3 forward      [t][co][ci]
    case 1: d.T = 9; d.M = Cin; d.K = Cout; break;           // conv3x3 dX           [8-t][ci][co]
    case 2: d.T = 1; d.M = Cout; d.K = Cin * 9; break;       // im2col'd first layer [co][ci*9+t]
    case 3: d.T = 1; d.M = 4 * Cout; d.K = Cin; break;       // convT forward        [(kh,kw,co)][ci]
    default: d.T = 4; d.M = Cin; d.K = Cout; break;          // convT dX             [(kh,kw)][ci][co]
  }
  d.Mp = round_up(d.M, 128);
  d.Kp = round_up(d.K, 32);
  return d;
}

__device__ __forceinline__ void weight_image_elements(int mode, const float* __restrict__ w, int Cout, int Cin, u16* __restrict__ out,
                                                      const WImg& d, long long first, long long stride) {
  const long long total = (long long)d.T * d.Mp * d.Kp;
  for (long long e = first; e < total; e += stride) {
    const int k = (int)(e % d.Kp);
    const int m = (int)((e / d.Kp) % d.Mp);
    const int t = (int)(e / ((long long)d.Kp * d.Mp));
    float v = 0.f;
    if (m < d.M && k < d.K) {
      switch (mode) {
        case 0: v = w[((size_t)m * Cin + k) * 9 + t]; break;
        case 1: v = w[((size_t)k * Cin + m) * 9 + (8 - t)]; break;
        case 2: v = w[(size_t)m * Cin * 9 + k]; break;
        case 3: { const int q = m / Cout, co = m - q * Cout; v = w[((size_t)k * Cout + co) * 4 + q]; break; }
        default: v = w[((size_t)m * Cout + k) * 4 + t]; break;
      }
    }
    out[e] = f32_to_bf16(v);
  }
}
__global__ void weight_image_kernel(int mode, const float* __restrict__ w, int Cout, int Cin, u16* __restrict__ out, WImg d) {
  weight_image_elements(mode, w, Cout, Cin, out, d, (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x);
}
// every image of a step in one launch (blockIdx.y = job): as 43 separate launches between the convolutions they are latency,
// 10 us each
constexpr int WJOBS = 32;
constexpr int WPAIRS = 2048;   // (row, k) pairs per block: a job gets blocks in proportion to its size
struct WJob { const float* w; u16* out; int mode, Cout, Cin, first_block; WImg d; };
struct WJobs { WJob j[WJOBS]; int n; };
// A thread owns one (row m, column k) of an image, k fastest, and walks its taps: the fp32 source of a pair's taps is ONE contiguous
// run (9 floats of a 3x3 kernel, 4 of a 2x2 one) and every tap's store is coalesced over k -- an element-major walk reads
// 4 bytes at a stride of 36 and pays three 64-bit divisions per element.  ConvT forward (mode 3, m = (q, co)): the pair is
// (co, k) and its four q rows, when the image has no padded rows.
static inline __host__ __device__ int wimg_pairs(int mode, int Cout, const WImg& d) {
  return (mode == 3 && d.Mp == 4 * Cout) ? Cout * d.Kp : d.Mp * d.Kp;
}
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // the flat parameter arena aligns tensors to 4 bytes only
// A thread owns EIGHT consecutive k of one row: every tap's store is then 16 bytes (2-byte stores move 128 B per wave-instruction),
// and the fp32 source of a (row, k) pair's taps is still one contiguous run.
__global__ __launch_bounds__(256) void weight_images_kernel(WJobs jobs) {
  int q = 0;
  while (q + 1 < jobs.n && (int)blockIdx.x >= jobs.j[q + 1].first_block) ++q;   // (<= 32 jobs: a linear search)
  const WJob& J = jobs.j[q];
  const WImg d = J.d;
  const bool quad = J.mode == 3 && d.Mp == 4 * J.Cout;
  const int k8n = d.Kp >> 3;                                  // Kp % 32 == 0
  const int units = (quad ? J.Cout : d.Mp) * k8n;             // (row, 8 k) units of the job
  const int base = ((int)blockIdx.x - J.first_block) * (WPAIRS / 8);
  const int end = base + WPAIRS / 8 < units ? base + WPAIRS / 8 : units;
  const size_t plane = (size_t)d.Mp * d.Kp;
  for (int ue = base + threadIdx.x; ue < end; ue += 256) {
    const int m = ue / k8n, k0 = (ue - m * k8n) * 8;
    const size_t o = (size_t)m * d.Kp + k0;
    if (quad) {   // m = co; rows q * Cout + co
      float v[4][8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        f4u t4 = f4u{0.f, 0.f, 0.f, 0.f};
        if (k0 + j < d.K) t4 = *reinterpret_cast<const f4u*>(J.w + ((size_t)(k0 + j) * J.Cout + m) * 4);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) v[qq][j] = t4[qq];
      }
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) st16(J.out + (size_t)(qq * J.Cout + m) * d.Kp + k0, pack8(v[qq]));
      continue;
    }
    switch (J.mode) {
      case 0:
      case 1: {
        float v[9][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool in = m < d.M && k0 + j < d.K;
          const float* src = J.mode == 0 ? J.w + ((size_t)m * J.Cin + k0 + j) * 9 : J.w + ((size_t)(k0 + j) * J.Cin + m) * 9;
#pragma unroll
          for (int t = 0; t < 9; ++t) v[t][j] = in ? src[J.mode == 0 ? t : 8 - t] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) st16(J.out + t * plane + o, pack8(v[t]));
        break;
      }
      case 2: {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (m < d.M && k0 + j < d.K) ? J.w[(size_t)m * J.Cin * 9 + k0 + j] : 0.f;
        st16(J.out + o, pack8(v));
        break;
      }
      case 3: {
        float v[8];
        const int qq = m / J.Cout, co = m - qq * J.Cout;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (m < d.M && k0 + j < d.K) ? J.w[((size_t)(k0 + j) * J.Cout + co) * 4 + qq] : 0.f;
        st16(J.out + o, pack8(v));
        break;
      }
      default: {
        float v[4][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          f4u t4 = f4u{0.f, 0.f, 0.f, 0.f};
          if (m < d.M && k0 + j < d.K) t4 = *reinterpret_cast<const f4u*>(J.w + ((size_t)m * J.Cout + k0 + j) * 4);
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t][j] = t4[t];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) st16(J.out + t * plane + o, pack8(v[t]));
        break;
      }
    }
  }
}

// ---- first-layer im2col: x (N,C,H,W) fp32 -> col (N,H,W,Kp) bf16, k = c*9 + tap ---------------------------------------
__global__ __launch_bounds__(256) void im2col3x3_kernel(const float* __restrict__ x, int C, int H, int W, NhwcD col) {
  const int n = blockIdx.y;
  const int groups = col.C >> 3;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)H * W * groups) return;
  const int gk = (int)(e % groups);
  const int p = (int)(e / groups);
  const int h = p / W, wq = p - h * W;
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = gk * 8 + i;
    float v = 0.f;
    if (k < C * 9) {
      const int c = k / 9, t = k - c * 9;
      const int hi = h + t / 3 - 1, wi = wq + t % 3 - 1;
      if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) v = x[(((size_t)n * C + c) * H + hi) * W + wi];
    }
    f[i] = v;
  }
  st16(col.p + ((long long)n * H * W + p) * col.pitch + gk * 8, pack8(f));
}

// ---- a = relu(y * scale + shift) ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_apply_kernel(NhwcD y, const float* __restrict__ scale, const float* __restrict__ shift,
                                                       NhwcD a, int relu, long long npix) {
  const int groups = y.C >> 3;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= npix * groups) return;
  const int gk = (int)(e % groups);
  const long long p = e / groups;
  float f[8];
  unpack8(ld16(y.p + p * y.pitch + gk * 8), f);
  const f32x4 s0 = *reinterpret_cast<const f32x4*>(scale + gk * 8), s1 = *reinterpret_cast<const f32x4*>(scale + gk * 8 + 4);
  const f32x4 h0 = *reinterpret_cast<const f32x4*>(shift + gk * 8), h1 = *reinterpret_cast<const f32x4*>(shift + gk * 8 + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[i] = fmaf(f[i], s0[i], h0[i]);
    f[4 + i] = fmaf(f[4 + i], s1[i], h1[i]);
  }
  if (relu) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = fmaxf(f[i], 0.f);
  }
  st16(a.p + p * a.pitch + gk * 8, pack8(f));
}

// The same, UNR pixels per thread (C / 8 divides 256: every network shape): a block's 256 threads are (256 / groups) pixel lanes x
// groups channel groups, a thread keeps its group's coefficients and walks UNR pixels with all their loads in flight before the
// first store -- four times the bytes in flight per thread, index arithmetic and coefficient loads paid once instead of per pixel.
template <int UNR>
__global__ __launch_bounds__(256) void bn_apply_multi_kernel(NhwcD y, const float* __restrict__ scale, const float* __restrict__ shift,
                                                             NhwcD a, int relu, long long npix) {
  const int groups = y.C >> 3, ppi = 256 / groups;
  const int pl = threadIdx.x / groups, gk = threadIdx.x - pl * groups;
  const long long p0 = (long long)blockIdx.x * (ppi * UNR) + pl;
  const f32x4 s0 = *reinterpret_cast<const f32x4*>(scale + gk * 8), s1 = *reinterpret_cast<const f32x4*>(scale + gk * 8 + 4);
  const f32x4 h0 = *reinterpret_cast<const f32x4*>(shift + gk * 8), h1 = *reinterpret_cast<const f32x4*>(shift + gk * 8 + 4);
  uint4 raw[UNR];
#pragma unroll
  for (int k = 0; k < UNR; ++k) {
    const long long p = p0 + (long long)k * ppi;
    raw[k] = p < npix ? ld16(y.p + p * y.pitch + gk * 8) : make_uint4(0, 0, 0, 0);
  }
#pragma unroll
  for (int k = 0; k < UNR; ++k) {
    const long long p = p0 + (long long)k * ppi;
    float f[8];
    unpack8(raw[k], f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[i] = fmaf(f[i], s0[i], h0[i]);
      f[4 + i] = fmaf(f[4 + i], s1[i], h1[i]);
    }
    if (relu) {
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] = fmaxf(f[i], 0.f);
    }
    if (p < npix) st16(a.p + p * a.pitch + gk * 8, pack8(f));
  }
}

// ---- a = relu(y * scale + shift) AND pooled = MaxPool2d(2)(a) in one pass (unet.py:15-16 followed by :26) ---------------
// The encoder's second unit feeds a max-pool: the stand-alone pool re-reads the activation the apply pass has just written.
// Thread = one 2x2 window x 8 channels over the ceil(H/2) x ceil(W/2) window grid: it reads the window's four raw values,
// writes their four activations (into the concat buffer's skip slice) and, where the window is whole (floor mode drops an
// odd last row / column), their maximum.  Rounding to bf16 is monotonic, so max of the rounded activations == rounded max:
// bit-identical to gsd_bf16_bn_apply + gsd_bf16_maxpool2.
// idx (or null): one u16 per (window, 8-channel group), [N][H/2][W/2][C/8]: two bits per channel = which of the window's four
// activations the pool took -- the first maximum in (0,0),(0,1),(1,0),(1,1) order of the STORED bf16 values, what the backward
// (bn_bwd_reduce_pool_bf16_kernel) otherwise finds by re-reading the four activations (4 x 16 B per thread instead of 2 B).
__global__ __launch_bounds__(256) void bn_apply_pool_kernel(NhwcD y, const float* __restrict__ scale, const float* __restrict__ shift,
                                                            NhwcD a, NhwcD o, int wh, int ww, u16* __restrict__ idx) {
  const int groups = y.C >> 3;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)y.N * wh * ww * groups) return;
  const int gk = (int)(e % groups);
  const long long p = e / groups;
  const int wp = (int)(p % ww);
  const int hp = (int)((p / ww) % wh);
  const int n = (int)(p / ((long long)ww * wh));
  const f32x4 s0 = *reinterpret_cast<const f32x4*>(scale + gk * 8), s1 = *reinterpret_cast<const f32x4*>(scale + gk * 8 + 4);
  const f32x4 h0 = *reinterpret_cast<const f32x4*>(shift + gk * 8), h1 = *reinterpret_cast<const f32x4*>(shift + gk * 8 + 4);
  const bool col2 = 2 * wp + 1 < y.W, row2 = 2 * hp + 1 < y.H;
  const long long pix = ((long long)n * y.H + 2 * hp) * y.W + 2 * wp;
  uint4 raw[4];
  raw[0] = ld16(y.p + pix * y.pitch + gk * 8);
  raw[1] = col2 ? ld16(y.p + (pix + 1) * y.pitch + gk * 8) : raw[0];
  raw[2] = row2 ? ld16(y.p + (pix + y.W) * y.pitch + gk * 8) : raw[0];
  raw[3] = (row2 && col2) ? ld16(y.p + (pix + y.W + 1) * y.pitch + gk * 8) : raw[0];
  float m[8];
  uint4 pk[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float f[8];
    unpack8(raw[q], f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[i] = fmaxf(fmaf(f[i], s0[i], h0[i]), 0.f);
      f[4 + i] = fmaxf(fmaf(f[4 + i], s1[i], h1[i]), 0.f);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = q == 0 ? f[i] : fmaxf(m[i], f[i]);
    const bool ok = (q == 0) || (q == 1 && col2) || (q == 2 && row2) || (q == 3 && row2 && col2);
    pk[q] = pack8(f);
    if (ok) st16(a.p + (pix + (q >> 1) * y.W + (q & 1)) * a.pitch + gk * 8, pk[q]);
  }
  if (row2 && col2) {
    st16(o.p + (((long long)n * o.H + hp) * o.W + wp) * o.pitch + gk * 8, pack8(m));
    if (idx != nullptr) {
      // activations are >= 0: their bf16 bit patterns order like the values, so the arg-max is taken on the 16-bit integers
      const unsigned w0[4] = {pk[0].x, pk[0].y, pk[0].z, pk[0].w}, w1[4] = {pk[1].x, pk[1].y, pk[1].z, pk[1].w};
      const unsigned w2[4] = {pk[2].x, pk[2].y, pk[2].z, pk[2].w}, w3[4] = {pk[3].x, pk[3].y, pk[3].z, pk[3].w};
      unsigned code = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int sh_ = (i & 1) * 16, wd = i >> 1;
        unsigned best = (w0[wd] >> sh_) & 0xffffu, bi = 0;
        const unsigned v1 = (w1[wd] >> sh_) & 0xffffu, v2 = (w2[wd] >> sh_) & 0xffffu, v3 = (w3[wd] >> sh_) & 0xffffu;
        if (v1 > best) { best = v1; bi = 1; }
        if (v2 > best) { best = v2; bi = 2; }
        if (v3 > best) { best = v3; bi = 3; }
        code |= bi << (2 * i);
      }
      idx[(((long long)n * o.H + hp) * o.W + wp) * groups + gk] = (u16)code;
    }
  }
}

// ---- MaxPool2d(2), floor mode (unet.py:26) --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool2_bf16_kernel(NhwcD a, NhwcD o) {
  const int groups = a.C >> 3;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)o.N * o.H * o.W * groups) return;
  const int gk = (int)(e % groups);
  const long long p = e / groups;
  const int wp = (int)(p % o.W);
  const int hp = (int)((p / o.W) % o.H);
  const int n = (int)(p / ((long long)o.W * o.H));
  const u16* b = a.p + (((long long)n * a.H + 2 * hp) * a.W + 2 * wp) * a.pitch + gk * 8;
  float v0[8], v1[8], v2[8], v3[8];
  unpack8(ld16(b), v0);
  unpack8(ld16(b + a.pitch), v1);
  unpack8(ld16(b + (long long)a.W * a.pitch), v2);
  unpack8(ld16(b + (long long)(a.W + 1) * a.pitch), v3);
#pragma unroll
  for (int i = 0; i < 8; ++i) v0[i] = fmaxf(fmaxf(v0[i], v1[i]), fmaxf(v2[i], v3[i]));
  st16(o.p + p * o.pitch + gk * 8, pack8(v0));
}

// ---- 1x1 output conv (+bias), fp32 NCHW result (unet.py:54) --------------------------------------------------------
constexpr int OUTC_MAXK = 4;
// 8 lanes per pixel, each 16-byte pieces c8, c8+8, ... of the pixel's channels (a wave reads 8 pixels x 128 B contiguous
// per pass for C = 64); the 8 partial dot products are combined with DPP-free xor shuffles inside the 8-lane group.
// BN (template): `a` is the RAW output of the last unit and scale / shift its BatchNorm coefficients -- the activation
// bf16(relu(y * scale + shift)), exactly what gsd_bf16_bn_apply would have stored, is formed in registers and never written.
template <bool BN>
__global__ __launch_bounds__(256) void conv1x1_out_bf16_kernel(NhwcD a, const float* __restrict__ w, const float* __restrict__ b,
                                                               int K, float* __restrict__ out, const float* __restrict__ scale,
                                                               const float* __restrict__ shift) {
  const long long HW = (long long)a.H * a.W;
  const long long p = ((long long)blockIdx.x * 256 + threadIdx.x) >> 3;
  const int sub = threadIdx.x & 7;
  const bool ok = p < (long long)a.N * HW;
  float acc[OUTC_MAXK];
#pragma unroll
  for (int k = 0; k < OUTC_MAXK; ++k) acc[k] = 0.f;
  if (ok) {
    const u16* src = a.p + p * a.pitch;
    for (int c = sub * 8; c < a.C; c += 64) {
      float f[8];
      unpack8(ld16(src + c), f);
      if (BN) {
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = bf16_to_f32(f32_to_bf16(fmaxf(fmaf(f[i], scale[c + i], shift[c + i]), 0.f)));
      }
#pragma unroll
      for (int k = 0; k < OUTC_MAXK; ++k)
        if (k < K) {
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[k] = fmaf(f[i], w[(size_t)k * a.C + c + i], acc[k]);
        }
    }
  }
#pragma unroll
  for (int k = 0; k < OUTC_MAXK; ++k) {
    acc[k] += __shfl_xor(acc[k], 1, 64);
    acc[k] += __shfl_xor(acc[k], 2, 64);
    acc[k] += __shfl_xor(acc[k], 4, 64);
  }
  if (ok && sub == 0) {
    const long long n = p / HW, q = p - n * HW;
#pragma unroll
    for (int k = 0; k < OUTC_MAXK; ++k)
      if (k < K) out[(n * K + k) * HW + q] = acc[k] + (b != nullptr ? b[k] : 0.f);
  }
}

// ---- BatchNorm + ReLU (+ max-pool / output conv) backward, pass 1 ---------------------------------------------------
struct BnBwdB {
  NhwcD y, g, a, dpool, dz;
  const u16* idx;   // pool mode: the forward's arg-max codes instead of `a` (or null)
  const float* scale; const float* shift; const float* mean; const float* invstd;
  const float* dout; const float* wout;
  float* partials;
  int pixb, chunks;
};

// grid (chunks, N); block: 256 threads = (256 / groups) pixels x groups 8-channel groups per pass (groups <= 256)
template <int MODE>
__global__ __launch_bounds__(256) void bn_bwd_reduce_bf16_kernel(const BnBwdB P) {
  extern __shared__ float red[];   // [256][24]
  const int C = P.y.C, groups = C >> 3;
  const int tpp = groups < 256 ? groups : 256;     // threads per pixel
  const int ppi = 256 / tpp;                       // pixels per pass
  const int chunk = blockIdx.x, n = blockIdx.y;
  const int HW = P.y.H * P.y.W;
  const int pl = threadIdx.x / tpp, gl = threadIdx.x - pl * tpp;
  float s1[8], s2[8], s3[8];
  const int p_end = min((chunk + 1) * P.pixb, HW);
  for (int gk = gl; gk < groups; gk += tpp) {      // one trip unless C > 2048
    float sc[8], sh[8], mu[8], is[8], wo[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      sc[i] = P.scale[gk * 8 + i]; sh[i] = P.shift[gk * 8 + i]; mu[i] = P.mean[gk * 8 + i]; is[i] = P.invstd[gk * 8 + i];
      wo[i] = MODE == 2 ? P.wout[gk * 8 + i] : 0.f;
      s1[i] = s2[i] = s3[i] = 0.f;
    }
    // the NEXT pixel's loads go out in front of this pixel's store: dz may alias g, so the compiler keeps loads behind older
    // stores -- without the prefetch a thread has one load in flight and every pixel is a dependent round trip
    const int p_first = chunk * P.pixb + pl;
    const bool any = p_first < p_end && pl < ppi;
    uint4 y_nx = make_uint4(0, 0, 0, 0), g_nx = make_uint4(0, 0, 0, 0);
    float d_nx = 0.f;
    if (any) {
      const long long pix0 = (long long)n * HW + p_first;
      y_nx = ld16(P.y.p + pix0 * P.y.pitch + gk * 8);
      if (MODE == 2) d_nx = P.dout[pix0];
      else g_nx = ld16(P.g.p + pix0 * P.g.pitch + gk * 8);
    }
    for (int p = p_first; p < p_end && pl < ppi; p += ppi) {
      const long long pix = (long long)n * HW + p;
      float yv[8], gv[8];
      const uint4 y_cur = y_nx, g_cur = g_nx;
      const float d = d_nx;
      if (p + ppi < p_end) {
        const long long pixn = pix + ppi;
        y_nx = ld16(P.y.p + pixn * P.y.pitch + gk * 8);
        if (MODE == 2) d_nx = P.dout[pixn];
        else g_nx = ld16(P.g.p + pixn * P.g.pitch + gk * 8);
      }
      unpack8(y_cur, yv);
      if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) gv[i] = d * wo[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float av = bf16_to_f32(f32_to_bf16(fmaxf(fmaf(yv[i], sc[i], sh[i]), 0.f)));   // the activation as stored
          s3[i] = fmaf(d, av, s3[i]);
        }
      } else {
        unpack8(g_cur, gv);
      }
      if (MODE == 1) {
        const int h = p / P.y.W, w = p - h * P.y.W;
        const int hp = h >> 1, wp = w >> 1;
        if (hp < P.dpool.H && wp < P.dpool.W) {
          // arg-max of the stored activations of the 2x2 window; first maximum in (0,0),(0,1),(1,0),(1,1) order wins
          const u16* wb = P.a.p + (((long long)n * P.a.H + 2 * hp) * P.a.W + 2 * wp) * P.a.pitch + gk * 8;
          float a0[8], a1[8], a2[8], a3[8], dp[8];
          unpack8(ld16(wb), a0);
          unpack8(ld16(wb + P.a.pitch), a1);
          unpack8(ld16(wb + (long long)P.a.W * P.a.pitch), a2);
          unpack8(ld16(wb + (long long)(P.a.W + 1) * P.a.pitch), a3);
          unpack8(ld16(P.dpool.p + (((long long)n * P.dpool.H + hp) * P.dpool.W + wp) * P.dpool.pitch + gk * 8), dp);
          const int me = ((h & 1) << 1) | (w & 1);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            float best = a0[i];
            int bi = 0;
            if (a1[i] > best) { best = a1[i]; bi = 1; }
            if (a2[i] > best) { best = a2[i]; bi = 2; }
            if (a3[i] > best) { best = a3[i]; bi = 3; }
            if (bi == me) gv[i] += dp[i];
          }
        }
      }
      float dzv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float yn = fmaf(yv[i], sc[i], sh[i]);
        dzv[i] = yn > 0.f ? gv[i] : 0.f;
      }
      const uint4 packed = pack8(dzv);
      st16(P.dz.p + pix * P.dz.pitch + gk * 8, packed);
      float dq[8];
      unpack8(packed, dq);     // sums of the values as stored: the apply pass reads these back
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        s1[i] += dq[i];
        s2[i] = fmaf(dq[i], (yv[i] - mu[i]) * is[i], s2[i]);
      }
    }
    // reduce over the ppi pixel lanes that share this channel group
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      red[threadIdx.x * 24 + i] = s1[i];
      red[threadIdx.x * 24 + 8 + i] = s2[i];
      red[threadIdx.x * 24 + 16 + i] = s3[i];
    }
    __syncthreads();
    if (pl == 0) {
      float* row = P.partials + (size_t)(n * P.chunks + chunk) * 3 * C;
      for (int q = 0; q < 24; ++q) {
        float s = 0.f;
        for (int r = 0; r < ppi; ++r) s += red[(r * tpp + gl) * 24 + q];
        row[(q >> 3) * C + gk * 8 + (q & 7)] = s;
      }
    }
  }
}

// Pool mode, one 2x2 window per (thread, 8-channel group): the four activations of a window are read ONCE (the per-pixel
// form above reads each window four times), the arg-max is taken once, and the four dz are written from the same
// thread.  With odd H / W the last row / column belongs to no window (floor pooling): the threads of the last window
// row / column also carry those pixels, which only see the direct gradient g.
// grid (chunks over the Hp*Wp windows, N); same partial-row layout and count as bn_bwd_reduce_bf16_kernel.
__global__ __launch_bounds__(256) void bn_bwd_reduce_pool_bf16_kernel(const BnBwdB P) {
  extern __shared__ float red[];   // [256][16]
  const int C = P.y.C, groups = C >> 3;
  const int tpp = groups < 256 ? groups : 256, ppi = 256 / tpp;
  const int chunk = blockIdx.x, n = blockIdx.y;
  const int H = P.y.H, W = P.y.W, Hp = P.dpool.H, Wp = P.dpool.W;
  const int pl = threadIdx.x / tpp, gl = threadIdx.x - pl * tpp;
  const int p_end = min((chunk + 1) * P.pixb, Hp * Wp);
  for (int gk = gl; gk < groups; gk += tpp) {
    float sc[8], sh[8], mu[8], is[8], s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      sc[i] = P.scale[gk * 8 + i]; sh[i] = P.shift[gk * 8 + i]; mu[i] = P.mean[gk * 8 + i]; is[i] = P.invstd[gk * 8 + i];
      s1[i] = s2[i] = 0.f;
    }
    // dz of the pixel at `pix` from its raw output and skip gradient (already loaded); add: pooled gradient routed here, or null
    auto pixel_from = [&](const uint4 yraw, const uint4 graw, long long pix, const float* add) {
      float yv[8], gv[8], dzv[8];
      unpack8(yraw, yv);
      unpack8(graw, gv);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float gsum = add != nullptr ? gv[i] + add[i] : gv[i];
        dzv[i] = fmaf(yv[i], sc[i], sh[i]) > 0.f ? gsum : 0.f;
      }
      const uint4 packed = pack8(dzv);
      st16(P.dz.p + pix * P.dz.pitch + gk * 8, packed);
      float dq[8];
      unpack8(packed, dq);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        s1[i] += dq[i];
        s2[i] = fmaf(dq[i], (yv[i] - mu[i]) * is[i], s2[i]);
      }
    };
    auto one_pixel = [&](int h, int w, const float* add) {
      const long long pix = ((long long)n * H + h) * W + w;
      pixel_from(ld16(P.y.p + pix * P.y.pitch + gk * 8), ld16(P.g.p + pix * P.g.pitch + gk * 8), pix, add);
    };
    for (int p = chunk * P.pixb + pl; p < p_end && pl < ppi; p += ppi) {
      const int hp = p / Wp, wp = p - hp * Wp;
      float dp[8], r0[8], r1[8], r2[8], r3[8];
      unpack8(ld16(P.dpool.p + (((long long)n * Hp + hp) * Wp + wp) * P.dpool.pitch + gk * 8), dp);
      if (P.idx != nullptr) {   // the forward (gsd_bf16_bn_apply_pool_idx) left the arg-max: 2 bytes instead of the window's 64
        const unsigned code = P.idx[(((long long)n * Hp + hp) * Wp + wp) * groups + gk];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const unsigned bi = (code >> (2 * i)) & 3u;
          r0[i] = bi == 0 ? dp[i] : 0.f;
          r1[i] = bi == 1 ? dp[i] : 0.f;
          r2[i] = bi == 2 ? dp[i] : 0.f;
          r3[i] = bi == 3 ? dp[i] : 0.f;
        }
      } else {
        const u16* wb = P.a.p + (((long long)n * P.a.H + 2 * hp) * P.a.W + 2 * wp) * P.a.pitch + gk * 8;
        float a0[8], a1[8], a2[8], a3[8];
        unpack8(ld16(wb), a0);
        unpack8(ld16(wb + P.a.pitch), a1);
        unpack8(ld16(wb + (long long)P.a.W * P.a.pitch), a2);
        unpack8(ld16(wb + (long long)(P.a.W + 1) * P.a.pitch), a3);
#pragma unroll
        for (int i = 0; i < 8; ++i) {   // first maximum in (0,0),(0,1),(1,0),(1,1) order wins
          float best = a0[i];
          int bi = 0;
          if (a1[i] > best) { best = a1[i]; bi = 1; }
          if (a2[i] > best) { best = a2[i]; bi = 2; }
          if (a3[i] > best) { best = a3[i]; bi = 3; }
          r0[i] = bi == 0 ? dp[i] : 0.f;
          r1[i] = bi == 1 ? dp[i] : 0.f;
          r2[i] = bi == 2 ? dp[i] : 0.f;
          r3[i] = bi == 3 ? dp[i] : 0.f;
        }
      }
      {
        // the window's eight loads go out together, IN FRONT of its four stores: dz may alias g, so the compiler keeps every load
        // of pixel k + 1 behind the store of pixel k -- four dependent round trips per thread instead of one
        const long long p00 = ((long long)n * H + 2 * hp) * W + 2 * wp;
        const long long px[4] = {p00, p00 + 1, p00 + W, p00 + W + 1};
        uint4 yr[4], gr[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          yr[k] = ld16(P.y.p + px[k] * P.y.pitch + gk * 8);
          gr[k] = ld16(P.g.p + px[k] * P.g.pitch + gk * 8);
        }
        pixel_from(yr[0], gr[0], px[0], r0);
        pixel_from(yr[1], gr[1], px[1], r1);
        pixel_from(yr[2], gr[2], px[2], r2);
        pixel_from(yr[3], gr[3], px[3], r3);
      }
      const bool xcol = wp == Wp - 1 && (W & 1), xrow = hp == Hp - 1 && (H & 1);
      if (xcol) {
        one_pixel(2 * hp, W - 1, nullptr);
        one_pixel(2 * hp + 1, W - 1, nullptr);
      }
      if (xrow) {
        one_pixel(H - 1, 2 * wp, nullptr);
        one_pixel(H - 1, 2 * wp + 1, nullptr);
        if (xcol) one_pixel(H - 1, W - 1, nullptr);
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      red[threadIdx.x * 16 + i] = s1[i];
      red[threadIdx.x * 16 + 8 + i] = s2[i];
    }
    __syncthreads();
    if (pl == 0) {
      float* row = P.partials + (size_t)(n * P.chunks + chunk) * 3 * C;
      for (int q = 0; q < 16; ++q) {
        float s = 0.f;
        for (int r = 0; r < ppi; ++r) s += red[(r * tpp + gl) * 16 + q];
        row[(q >> 3) * C + gk * 8 + (q & 7)] = s;
      }
      for (int i = 0; i < 8; ++i) row[2 * C + gk * 8 + i] = 0.f;   // third column block: unused in pool mode
    }
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_bf16_kernel(NhwcD dz, NhwcD y, const float* __restrict__ scale,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ c1, const float* __restrict__ c2,
                                                                long long npix) {
  const int groups = y.C >> 3;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= npix * groups) return;
  const int gk = (int)(e % groups);
  const long long p = e / groups;
  float d[8], yv[8];
  unpack8(ld16(dz.p + p * dz.pitch + gk * 8), d);
  unpack8(ld16(y.p + p * y.pitch + gk * 8), yv);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = gk * 8 + i;
    const float xh = (yv[i] - mean[c]) * invstd[c];
    d[i] = scale[c] * (d[i] - c1[c] - xh * c2[c]);
  }
  st16(dz.p + p * dz.pitch + gk * 8, pack8(d));
}

// UNR pixels per thread (C / 8 divides 256), as bn_apply_multi_kernel: the 40 coefficient floats of a channel group are loaded
// once per thread instead of once per pixel, and all of a thread's loads are in flight before its first (aliasing) store.
template <int UNR>
__global__ __launch_bounds__(256) void bn_bwd_apply_bf16_multi_kernel(NhwcD dz, NhwcD y, const float* __restrict__ scale,
                                                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                      const float* __restrict__ c1, const float* __restrict__ c2,
                                                                      long long npix) {
  const int groups = y.C >> 3, ppi = 256 / groups;
  const int pl = threadIdx.x / groups, gk = threadIdx.x - pl * groups;
  const long long p0 = (long long)blockIdx.x * (ppi * UNR) + pl;
  float sc[8], mu[8], is[8], k1[8], k2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = gk * 8 + i;
    sc[i] = scale[c]; mu[i] = mean[c]; is[i] = invstd[c]; k1[i] = c1[c]; k2[i] = c2[c];
  }
  uint4 dr[UNR], yr[UNR];
#pragma unroll
  for (int k = 0; k < UNR; ++k) {
    const long long p = p0 + (long long)k * ppi;
    const bool ok = p < npix;
    dr[k] = ok ? ld16(dz.p + p * dz.pitch + gk * 8) : make_uint4(0, 0, 0, 0);
    yr[k] = ok ? ld16(y.p + p * y.pitch + gk * 8) : make_uint4(0, 0, 0, 0);
  }
#pragma unroll
  for (int k = 0; k < UNR; ++k) {
    const long long p = p0 + (long long)k * ppi;
    float d[8], yv[8];
    unpack8(dr[k], d);
    unpack8(yr[k], yv);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float xh = (yv[i] - mu[i]) * is[i];
      d[i] = sc[i] * (d[i] - k1[i] - xh * k2[i]);
    }
    if (p < npix) st16(dz.p + p * dz.pitch + gk * 8, pack8(d));
  }
}

// ---- per-channel sums over a window of every image (ConvTranspose2d bias gradient) -------------------------------------
// stage 1: grid (chunks, N), same thread layout as the BatchNorm reduction; stage 2: one thread per channel, fp64
__global__ __launch_bounds__(256) void channel_sums_stage1(NhwcD t, int y0, int x0, int hh, int ww, int pixb, int chunks,
                                                           float* __restrict__ ws) {
  extern __shared__ float red[];   // [256][8]
  const int C = t.C, groups = C >> 3;
  const int tpp = groups < 256 ? groups : 256, ppi = 256 / tpp;
  const int chunk = blockIdx.x, n = blockIdx.y;
  const int pl = threadIdx.x / tpp, gl = threadIdx.x - pl * tpp;
  const int p_end = min((chunk + 1) * pixb, hh * ww);
  for (int gk = gl; gk < groups; gk += tpp) {
    float s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = 0.f;
    for (int p = chunk * pixb + pl; p < p_end && pl < ppi; p += ppi) {
      const int r = p / ww, c = p - r * ww;
      float f[8];
      unpack8(ld16(t.p + (((long long)n * t.H + y0 + r) * t.W + x0 + c) * t.pitch + gk * 8), f);
#pragma unroll
      for (int i = 0; i < 8; ++i) s[i] += f[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) red[threadIdx.x * 8 + i] = s[i];
    __syncthreads();
    if (pl == 0) {
      float* row = ws + (size_t)(n * chunks + chunk) * C;
      for (int i = 0; i < 8; ++i) {
        float a = 0.f;
        for (int r = 0; r < ppi; ++r) a += red[(r * tpp + gl) * 8 + i];
        row[gk * 8 + i] = a;
      }
    }
  }
}
// block = 64 channels x 4 row lanes (coalesced rows), fixed summation order
__global__ __launch_bounds__(256) void channel_sums_stage2(const float* __restrict__ ws, int rows, int C, float* __restrict__ out) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  double s = 0.0;
  if (c < C)
    for (int r = rl; r < rows; r += 4) s += (double)ws[(size_t)r * C + c];
  __shared__ double red[4][64];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < C) out[c] = (float)(red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ConvTranspose2d bias gradient from what the decoder's dX launch left: out[c] = sum over the partial rows of column col0 + c
// (per-channel sums of the WHOLE gradient plane, from that launch's statistics epilogue) minus the workspace rows (stage-1 sums
// over the F.pad strips outside the transposed convolution's window).  Block = 64 channels x 16 row lanes, fp64, fixed order.
__global__ __launch_bounds__(1024) void convT_bias_combine_kernel(const float* __restrict__ part, int rows, int ld, int col0,
                                                                  const float* __restrict__ ws, int wrows, int C, float* __restrict__ out) {
  constexpr int RL = 16;   // row lanes: a few hundred rows, one block per 64 channels -- the walk is a chain of dependent loads
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  double s = 0.0;
  if (c < C) {
    for (int r = rl; r < rows; r += RL) s += (double)part[(size_t)r * ld + col0 + c];
    for (int r = rl; r < wrows; r += RL) s -= (double)ws[(size_t)r * C + c];
  }
  __shared__ double red[RL][64];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < C) {
    double t = 0.0;
    for (int i = 0; i < RL; ++i) t += red[i][threadIdx.x];
    out[c] = (float)t;
  }
}

// pixels per block of the stand-alone reduce kernels: target_blocks = 0 -> GSD_BF16_BN_BLOCKS (tuning) or the default
int pick_pixb(int N, int HW, int target_blocks = 0) {
  if (target_blocks <= 0) target_blocks = gsd_env_int("GSD_BF16_BN_BLOCKS", 512);   // (2048 until round 4; 512 / 1024 / 2048 measured 25.9-26.1 / 26.15 / 26.3 ms per step)
  long long pixb = ((long long)N * HW + target_blocks - 1) / target_blocks;
  pixb = (pixb + 31) / 32 * 32;
  return (int)(pixb < 32 ? 32 : pixb);
}

int check_c8(const gsd_nhwc* t, const char* what) {
  if (int e = gsd_check_nhwc(t, what)) return e;
  GSD_REQUIRE(t->C % 8 == 0, GSD_ERR_UNSUPPORTED, "%s: C=%d must be a multiple of 8", what, t->C);
  return 0;
}
bool same_extent(const gsd_nhwc* a, const gsd_nhwc* b) { return a->N == b->N && a->H == b->H && a->W == b->W && a->C == b->C; }
long long npix_of(const gsd_nhwc* t) { return (long long)t->N * t->H * t->W; }

}  // namespace

extern "C" int64_t gsd_bf16_weight_image_size(int mode, int Cout, int Cin) {
  if (mode < 0 || mode > 4 || Cout <= 0 || Cin <= 0) return 0;
  const WImg d = wimg_dims(mode, Cout, Cin);
  return (int64_t)d.T * d.Mp * d.Kp;
}

extern "C" int gsd_bf16_weight_image(int mode, const float* w, int Cout, int Cin, void* out, void* stream) {
  GSD_REQUIRE(w && out && mode >= 0 && mode <= 4 && Cout > 0 && Cin > 0, GSD_ERR_BAD_ARG, "gsd_bf16_weight_image: bad argument");
  const WImg d = wimg_dims(mode, Cout, Cin);
  const long long total = (long long)d.T * d.Mp * d.Kp;
  const int grid = (int)(ceil_div64(total, 256) < 8192 ? ceil_div64(total, 256) : 8192);
  hipLaunchKernelGGL(weight_image_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, mode, w, Cout, Cin, (u16*)out, d);
  GSD_LAUNCH_CHECK("gsd_bf16_weight_image");
  return GSD_OK;
}

extern "C" int gsd_bf16_weight_images(const gsd_bf16_wimg_job* jobs, int n, void* stream) {
  GSD_REQUIRE(jobs && n > 0, GSD_ERR_BAD_ARG, "gsd_bf16_weight_images: bad argument");
  for (int i = 0; i < n; ++i)
    GSD_REQUIRE(jobs[i].w && jobs[i].out && jobs[i].mode >= 0 && jobs[i].mode <= 4 && jobs[i].Cout > 0 && jobs[i].Cin > 0, GSD_ERR_BAD_ARG,
                "gsd_bf16_weight_images: bad job %d", i);
  for (int base = 0; base < n; base += WJOBS) {
    WJobs a;
    a.n = n - base < WJOBS ? n - base : WJOBS;
    long long blocks = 0;
    for (int i = 0; i < a.n; ++i) {
      const gsd_bf16_wimg_job& q = jobs[base + i];
      const WImg d = wimg_dims(q.mode, q.Cout, q.Cin);
      a.j[i] = WJob{q.w, (u16*)q.out, q.mode, q.Cout, q.Cin, (int)blocks, d};
      GSD_REQUIRE((long long)d.Mp * d.Kp < 2147483647LL, GSD_ERR_UNSUPPORTED, "gsd_bf16_weight_images: image %d too large", base + i);
      blocks += ceil_div(wimg_pairs(q.mode, q.Cout, d), WPAIRS);
    }
    GSD_REQUIRE(blocks < 2147483647LL, GSD_ERR_UNSUPPORTED, "gsd_bf16_weight_images: images too large");
    hipLaunchKernelGGL(weight_images_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    GSD_LAUNCH_CHECK("gsd_bf16_weight_images");
  }
  return GSD_OK;
}

extern "C" int gsd_bf16_im2col3x3(const float* x, int N, int C, int H, int W, const gsd_nhwc* col, void* stream) {
  GSD_REQUIRE(x && N > 0 && C > 0 && H > 0 && W > 0, GSD_ERR_BAD_ARG, "gsd_bf16_im2col3x3: bad argument");
  if (int e = check_c8(col, "gsd_bf16_im2col3x3 col")) return e;
  GSD_REQUIRE(col->N == N && col->H == H && col->W == W && col->C == round_up(9 * C, 32), GSD_ERR_BAD_ARG,
              "gsd_bf16_im2col3x3: col must be (N,H,W,round_up(9*C,32))");
  GSD_REQUIRE(N <= 65535, GSD_ERR_UNSUPPORTED, "gsd_bf16_im2col3x3: N must be <= 65535");
  const long long per = (long long)H * W * (col->C / 8);
  hipLaunchKernelGGL(im2col3x3_kernel, dim3((unsigned)ceil_div64(per, 256), N), dim3(256), 0, (hipStream_t)stream, x, C, H, W,
                     to_nhwc(*col));
  GSD_LAUNCH_CHECK("gsd_bf16_im2col3x3");
  return GSD_OK;
}

extern "C" int gsd_bf16_bn_apply(const gsd_nhwc* y, const float* scale, const float* shift, const gsd_nhwc* a, int relu,
                                 void* stream) {
  if (int e = check_c8(y, "gsd_bf16_bn_apply y")) return e;
  if (int e = check_c8(a, "gsd_bf16_bn_apply a")) return e;
  GSD_REQUIRE(scale && shift && same_extent(y, a), GSD_ERR_BAD_ARG, "gsd_bf16_bn_apply: bad argument");
  const long long np = npix_of(y), total = np * (y->C / 8);
  const int groups = y->C / 8;
  if (groups <= 256 && 256 % groups == 0 && total >= 256 * 4 * 64) {
    const long long ppb = (256 / groups) * 4;
    hipLaunchKernelGGL(bn_apply_multi_kernel<4>, dim3((unsigned)ceil_div64(np, ppb)), dim3(256), 0, (hipStream_t)stream, to_nhwc(*y), scale, shift,
                       to_nhwc(*a), relu, np);
  } else {
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream, to_nhwc(*y),
                       scale, shift, to_nhwc(*a), relu, np);
  }
  GSD_LAUNCH_CHECK("gsd_bf16_bn_apply");
  return GSD_OK;
}

extern "C" int gsd_bf16_bn_apply_pool_idx(const gsd_nhwc* y, const float* scale, const float* shift, const gsd_nhwc* a,
                                          const gsd_nhwc* pooled, void* idx, void* stream) {
  if (int e = check_c8(y, "gsd_bf16_bn_apply_pool y")) return e;
  if (int e = check_c8(a, "gsd_bf16_bn_apply_pool a")) return e;
  if (int e = check_c8(pooled, "gsd_bf16_bn_apply_pool pooled")) return e;
  GSD_REQUIRE(scale && shift && same_extent(y, a), GSD_ERR_BAD_ARG, "gsd_bf16_bn_apply_pool: bad argument");
  GSD_REQUIRE(y->H > 1 && y->W > 1 && pooled->N == y->N && pooled->C == y->C && pooled->H == y->H / 2 && pooled->W == y->W / 2,
              GSD_ERR_BAD_ARG, "gsd_bf16_bn_apply_pool: pooled must be (N,H/2,W/2,C)");
  const int wh = (y->H + 1) / 2, ww = (y->W + 1) / 2;
  const long long total = (long long)y->N * wh * ww * (y->C / 8);
  hipLaunchKernelGGL(bn_apply_pool_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream, to_nhwc(*y),
                     scale, shift, to_nhwc(*a), to_nhwc(*pooled), wh, ww, (u16*)idx);
  GSD_LAUNCH_CHECK("gsd_bf16_bn_apply_pool");
  return GSD_OK;
}

extern "C" int gsd_bf16_bn_apply_pool(const gsd_nhwc* y, const float* scale, const float* shift, const gsd_nhwc* a,
                                      const gsd_nhwc* pooled, void* stream) {
  return gsd_bf16_bn_apply_pool_idx(y, scale, shift, a, pooled, nullptr, stream);
}

extern "C" int gsd_bf16_maxpool2(const gsd_nhwc* a, const gsd_nhwc* pooled, void* stream) {
  if (int e = check_c8(a, "gsd_bf16_maxpool2 a")) return e;
  if (int e = check_c8(pooled, "gsd_bf16_maxpool2 pooled")) return e;
  GSD_REQUIRE(a->H > 1 && a->W > 1 && pooled->N == a->N && pooled->C == a->C && pooled->H == a->H / 2 && pooled->W == a->W / 2,
              GSD_ERR_BAD_ARG, "gsd_bf16_maxpool2: pooled must be (N,H/2,W/2,C)");
  const long long total = npix_of(pooled) * (a->C / 8);
  hipLaunchKernelGGL(maxpool2_bf16_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     to_nhwc(*a), to_nhwc(*pooled));
  GSD_LAUNCH_CHECK("gsd_bf16_maxpool2");
  return GSD_OK;
}

extern "C" int gsd_bf16_conv1x1_out(const gsd_nhwc* a, const float* w, const float* bias, int K, float* out, void* stream) {
  if (int e = check_c8(a, "gsd_bf16_conv1x1_out a")) return e;
  GSD_REQUIRE(w && out && K >= 1 && K <= OUTC_MAXK, GSD_ERR_UNSUPPORTED, "gsd_bf16_conv1x1_out: n_classes must be in [1,%d]",
              OUTC_MAXK);
  hipLaunchKernelGGL(conv1x1_out_bf16_kernel<false>, dim3((unsigned)ceil_div64(npix_of(a) * 8, 256)), dim3(256), 0, (hipStream_t)stream,
                     to_nhwc(*a), w, bias, K, out, nullptr, nullptr);
  GSD_LAUNCH_CHECK("gsd_bf16_conv1x1_out");
  return GSD_OK;
}

extern "C" int gsd_bf16_bn_relu_conv1x1_out(const gsd_nhwc* y, const float* scale, const float* shift, const float* w, const float* bias,
                                            int K, float* out, void* stream) {
  if (int e = check_c8(y, "gsd_bf16_bn_relu_conv1x1_out y")) return e;
  GSD_REQUIRE(scale && shift && w && out && K >= 1 && K <= OUTC_MAXK, GSD_ERR_UNSUPPORTED,
              "gsd_bf16_bn_relu_conv1x1_out: null argument or n_classes outside [1,%d]", OUTC_MAXK);
  hipLaunchKernelGGL(conv1x1_out_bf16_kernel<true>, dim3((unsigned)ceil_div64(npix_of(y) * 8, 256)), dim3(256), 0, (hipStream_t)stream,
                     to_nhwc(*y), w, bias, K, out, scale, shift);
  GSD_LAUNCH_CHECK("gsd_bf16_bn_relu_conv1x1_out");
  return GSD_OK;
}

extern "C" int gsd_bf16_bn_bwd_partial_rows(int N, int H, int W) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  return N * ceil_div(H * W, pick_pixb(N, H * W));
}

static int bn_bwd_reduce_impl(int mode, const gsd_nhwc* y, const float* scale, const float* shift, const float* mean,
                              const float* invstd, const gsd_nhwc* g, const gsd_nhwc* a, const void* pool_idx, const gsd_nhwc* dpool,
                              const float* dout, const float* wout, const gsd_nhwc* dz, float* partials, void* stream) {
  if (int e = check_c8(y, "gsd_bf16_bn_bwd_reduce y")) return e;
  if (int e = check_c8(dz, "gsd_bf16_bn_bwd_reduce dz")) return e;
  GSD_REQUIRE(mode >= 0 && mode <= 2 && scale && shift && mean && invstd && partials && same_extent(y, dz), GSD_ERR_BAD_ARG,
              "gsd_bf16_bn_bwd_reduce: bad argument");
  GSD_REQUIRE(y->N <= 65535 && (y->C <= 2048 || y->C % 2048 == 0), GSD_ERR_UNSUPPORTED,
              "gsd_bf16_bn_bwd_reduce: N must be <= 65535 and C <= 2048 or a multiple of 2048");
  BnBwdB P;
  P.y = to_nhwc(*y);
  P.dz = to_nhwc(*dz);
  P.g = P.a = P.dpool = P.y;
  if (mode != 2) {
    if (int e = check_c8(g, "gsd_bf16_bn_bwd_reduce g")) return e;
    GSD_REQUIRE(same_extent(y, g), GSD_ERR_BAD_ARG, "gsd_bf16_bn_bwd_reduce: g extent differs from y");
    P.g = to_nhwc(*g);
  } else {
    GSD_REQUIRE(dout && wout, GSD_ERR_BAD_ARG, "gsd_bf16_bn_bwd_reduce: mode OUTC needs dout, wout (n_classes == 1)");
  }
  P.idx = nullptr;
  if (mode == 1) {
    if (int e = check_c8(dpool, "gsd_bf16_bn_bwd_reduce dpool")) return e;
    GSD_REQUIRE(dpool->N == y->N && dpool->C == y->C && dpool->H == y->H / 2 && dpool->W == y->W / 2, GSD_ERR_BAD_ARG,
                "gsd_bf16_bn_bwd_reduce: mode POOL needs dpool (N,H/2,W/2,C)");
    if (pool_idx != nullptr) {
      P.idx = (const u16*)pool_idx;
    } else {
      if (int e = check_c8(a, "gsd_bf16_bn_bwd_reduce a")) return e;
      GSD_REQUIRE(same_extent(y, a), GSD_ERR_BAD_ARG, "gsd_bf16_bn_bwd_reduce: mode POOL needs a (N,H,W,C)");
      P.a = to_nhwc(*a);
    }
    P.dpool = to_nhwc(*dpool);
  }
  P.scale = scale; P.shift = shift; P.mean = mean; P.invstd = invstd;
  P.dout = dout; P.wout = wout; P.partials = partials;
  P.pixb = pick_pixb(y->N, y->H * y->W);
  P.chunks = ceil_div(y->H * y->W, P.pixb);
  const dim3 grid(P.chunks, y->N);
  const size_t lds = 256 * 24 * sizeof(float);
  if (mode == 0) hipLaunchKernelGGL((bn_bwd_reduce_bf16_kernel<0>), grid, dim3(256), lds, (hipStream_t)stream, P);
  else if (mode == 1) {
    // one thread per 2x2 window; SAME number of partial rows as the per-pixel form (the chunks now split the windows)
    P.pixb = ceil_div(P.dpool.H * P.dpool.W, P.chunks);
    hipLaunchKernelGGL(bn_bwd_reduce_pool_bf16_kernel, grid, dim3(256), 256 * 16 * sizeof(float), (hipStream_t)stream, P);
  } else hipLaunchKernelGGL((bn_bwd_reduce_bf16_kernel<2>), grid, dim3(256), lds, (hipStream_t)stream, P);
  GSD_LAUNCH_CHECK("gsd_bf16_bn_bwd_reduce");
  return GSD_OK;
}

extern "C" int gsd_bf16_bn_bwd_reduce(int mode, const gsd_nhwc* y, const float* scale, const float* shift, const float* mean,
                                      const float* invstd, const gsd_nhwc* g, const gsd_nhwc* a, const gsd_nhwc* dpool,
                                      const float* dout, const float* wout, const gsd_nhwc* dz, float* partials, void* stream) {
  return bn_bwd_reduce_impl(mode, y, scale, shift, mean, invstd, g, a, nullptr, dpool, dout, wout, dz, partials, stream);
}

extern "C" int gsd_bf16_bn_bwd_reduce_pool_idx(const gsd_nhwc* y, const float* scale, const float* shift, const float* mean,
                                               const float* invstd, const gsd_nhwc* g, const void* pool_idx, const gsd_nhwc* dpool,
                                               const gsd_nhwc* dz, float* partials, void* stream) {
  GSD_REQUIRE(pool_idx != nullptr, GSD_ERR_BAD_ARG, "gsd_bf16_bn_bwd_reduce_pool_idx: null index");
  return bn_bwd_reduce_impl(1, y, scale, shift, mean, invstd, g, nullptr, pool_idx, dpool, nullptr, nullptr, dz, partials, stream);
}

extern "C" int gsd_bf16_bn_bwd_apply(const gsd_nhwc* dz, const gsd_nhwc* y, const float* scale, const float* mean,
                                     const float* invstd, const float* c1, const float* c2, void* stream) {
  if (int e = check_c8(dz, "gsd_bf16_bn_bwd_apply dz")) return e;
  if (int e = check_c8(y, "gsd_bf16_bn_bwd_apply y")) return e;
  GSD_REQUIRE(scale && mean && invstd && c1 && c2 && same_extent(dz, y), GSD_ERR_BAD_ARG, "gsd_bf16_bn_bwd_apply: bad argument");
  const long long np = npix_of(y), total = np * (y->C / 8);
  const int groups = y->C / 8;
  if (groups <= 256 && 256 % groups == 0 && total >= 256 * 4 * 64) {
    const long long ppb = (256 / groups) * 4;
    hipLaunchKernelGGL(bn_bwd_apply_bf16_multi_kernel<4>, dim3((unsigned)ceil_div64(np, ppb)), dim3(256), 0, (hipStream_t)stream, to_nhwc(*dz),
                       to_nhwc(*y), scale, mean, invstd, c1, c2, np);
  } else {
    hipLaunchKernelGGL(bn_bwd_apply_bf16_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     to_nhwc(*dz), to_nhwc(*y), scale, mean, invstd, c1, c2, np);
  }
  GSD_LAUNCH_CHECK("gsd_bf16_bn_bwd_apply");
  return GSD_OK;
}

extern "C" int64_t gsd_bf16_channel_sums_workspace(int N, int hh, int ww, int C) {
  if (N <= 0 || hh <= 0 || ww <= 0 || C <= 0) return 0;
  return (int64_t)N * ceil_div(hh * ww, pick_pixb(N, hh * ww, 512)) * C;
}

extern "C" int gsd_bf16_channel_sums(const gsd_nhwc* t, int y0, int x0, int hh, int ww, float* out, float* workspace,
                                     int64_t workspace_elems, void* stream) {
  if (int e = check_c8(t, "gsd_bf16_channel_sums t")) return e;
  GSD_REQUIRE(out && workspace && y0 >= 0 && x0 >= 0 && hh > 0 && ww > 0 && y0 + hh <= t->H && x0 + ww <= t->W, GSD_ERR_BAD_ARG,
              "gsd_bf16_channel_sums: window (%d,%d)+(%d,%d) outside (%d,%d)", y0, x0, hh, ww, t->H, t->W);
  GSD_REQUIRE(t->N <= 65535 && (t->C <= 2048 || t->C % 2048 == 0), GSD_ERR_UNSUPPORTED,
              "gsd_bf16_channel_sums: N must be <= 65535 and C <= 2048 or a multiple of 2048");
  const int pixb = pick_pixb(t->N, hh * ww, 512), chunks = ceil_div(hh * ww, pixb);   // 512 partial rows: stage 2 stays short
  GSD_REQUIRE(workspace_elems >= (int64_t)t->N * chunks * t->C, GSD_ERR_WORKSPACE, "gsd_bf16_channel_sums: workspace too small");
  hipLaunchKernelGGL(channel_sums_stage1, dim3(chunks, t->N), dim3(256), 256 * 8 * sizeof(float), (hipStream_t)stream, to_nhwc(*t),
                     y0, x0, hh, ww, pixb, chunks, workspace);
  GSD_LAUNCH_CHECK("gsd_bf16_channel_sums stage1");
  hipLaunchKernelGGL(channel_sums_stage2, dim3(ceil_div(t->C, 64)), dim3(256), 0, (hipStream_t)stream, workspace, t->N * chunks,
                     t->C, out);
  GSD_LAUNCH_CHECK("gsd_bf16_channel_sums stage2");
  return GSD_OK;
}

namespace {
// the (up to four) rectangles of an (H, W) plane outside the window [oy, oy+hh) x [ox, ox+ww): {y0, x0, rows, cols}
int pad_rects(int H, int W, int oy, int ox, int hh, int ww, int (&r)[4][4]) {
  int n = 0;
  auto add = [&](int y0, int x0, int rh, int rw) {
    if (rh > 0 && rw > 0) { r[n][0] = y0; r[n][1] = x0; r[n][2] = rh; r[n][3] = rw; ++n; }
  };
  add(0, 0, oy, W);
  add(oy + hh, 0, H - oy - hh, W);
  add(oy, 0, hh, ox);
  add(oy, ox + ww, hh, W - ox - ww);
  return n;
}
}  // namespace

extern "C" int64_t gsd_bf16_convT_bias_grad_workspace(int N, int H, int W, int oy, int ox, int hh, int ww, int C) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || oy < 0 || ox < 0 || hh <= 0 || ww <= 0 || oy + hh > H || ox + ww > W) return 0;
  int r[4][4];
  const int n = pad_rects(H, W, oy, ox, hh, ww, r);
  int64_t tot = 0;
  for (int i = 0; i < n; ++i) tot += (int64_t)N * ceil_div(r[i][2] * r[i][3], pick_pixb(N, r[i][2] * r[i][3], 512)) * C;
  return tot > 0 ? tot : 1;
}

extern "C" int gsd_bf16_convT_bias_grad(const float* partials, int rows, int ld, int col0, const gsd_nhwc* g, int oy, int ox, int hh,
                                        int ww, float* out, float* workspace, int64_t workspace_elems, void* stream) {
  if (int e = check_c8(g, "gsd_bf16_convT_bias_grad g")) return e;
  GSD_REQUIRE(partials && out && workspace && rows > 0 && ld > 0 && col0 >= 0 && col0 + g->C <= ld, GSD_ERR_BAD_ARG,
              "gsd_bf16_convT_bias_grad: bad partial-row layout (rows %d, ld %d, col0 %d, C %d)", rows, ld, col0, g->C);
  GSD_REQUIRE(oy >= 0 && ox >= 0 && hh > 0 && ww > 0 && oy + hh <= g->H && ox + ww <= g->W, GSD_ERR_BAD_ARG,
              "gsd_bf16_convT_bias_grad: window (%d,%d)+(%d,%d) outside (%d,%d)", oy, ox, hh, ww, g->H, g->W);
  GSD_REQUIRE(g->N <= 65535 && (g->C <= 2048 || g->C % 2048 == 0), GSD_ERR_UNSUPPORTED,
              "gsd_bf16_convT_bias_grad: N must be <= 65535 and C <= 2048 or a multiple of 2048");
  GSD_REQUIRE(workspace_elems >= gsd_bf16_convT_bias_grad_workspace(g->N, g->H, g->W, oy, ox, hh, ww, g->C), GSD_ERR_WORKSPACE,
              "gsd_bf16_convT_bias_grad: workspace too small");
  int r[4][4];
  const int n = pad_rects(g->H, g->W, oy, ox, hh, ww, r);
  int wrows = 0;
  for (int i = 0; i < n; ++i) {
    const int area = r[i][2] * r[i][3];
    const int pixb = pick_pixb(g->N, area, 512), chunks = ceil_div(area, pixb);
    hipLaunchKernelGGL(channel_sums_stage1, dim3(chunks, g->N), dim3(256), 256 * 8 * sizeof(float), (hipStream_t)stream, to_nhwc(*g),
                       r[i][0], r[i][1], r[i][2], r[i][3], pixb, chunks, workspace + (size_t)wrows * g->C);
    GSD_LAUNCH_CHECK("gsd_bf16_convT_bias_grad strips");
    wrows += g->N * chunks;
  }
  hipLaunchKernelGGL(convT_bias_combine_kernel, dim3(ceil_div(g->C, 64)), dim3(1024), 0, (hipStream_t)stream, partials, rows, ld, col0,
                     workspace, wrows, g->C, out);
  GSD_LAUNCH_CHECK("gsd_bf16_convT_bias_grad");
  return GSD_OK;
}
