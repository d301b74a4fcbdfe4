// gsd_bf16_first.hip -- the first layer of the bf16 path WITHOUT the im2col tensor.
//
// `inc`'s first convolution (unet.py:11, 3 -> 64 @320x427) has K = 27: as an implicit GEMM it is one MFMA k-step
// (v_mfma_f32_16x16x32_bf16, K padded to 32) per output tile -- 15 GFLOP per batch of 32 against 560 MB of output: it is bound
// by the HBM write of its result.  The general path materialises col = im2col(x) (280 MB written and read back, plus a
// gather kernel) and runs the dense-tap kernel on it; here both directions read x itself:
//
//   gsd_bf16_conv3x3_first        y[n,h,w,co] = sum_{c,t} bf16(x[n,c,h+t/3-1,w+t%3-1]) * W[co][c*9+t]     (+ BatchNorm partial
//                                 sums, or relu(bn(.)) with running statistics in eval mode)
//   gsd_bf16_wgrad_first          dW[co][c*9+t] = sum_pixels d_raw[pixel][co] * bf16(x[.. + tap t]) with the BatchNorm backward
//                                 of the layer's output applied on the fly (d_raw = scale (dz - c1 - xhat c2), rounded to bf16
//                                 exactly as gsd_bf16_bn_bwd_apply stores it): the first layer has no dX, so dW is d_raw's
//                                 only reader and the apply pass disappears.
//
// Same products as the im2col path (same bf16 operands in the same k order through the same MFMA): the forward is
// bit-identical to gsd_bf16_im2col3x3 + gsd_bf16_conv_dense, dW agrees to the rounding of another summation order.
#include "gsd_bf16_common.h"

namespace {

constexpr int F_TH = 4, F_TW = 64;            // pixel tile of a block: one image row of 64 pixels per wave
constexpr int F_PITCH = F_TW + 2;             // halo row pitch (bf16 elements)
constexpr int F_PLANE = (F_TH + 2) * F_PITCH;

struct FirstP {
  const float* x;     // (N, C, H, W) fp32
  const u16* wt;      // gsd_bf16_weight_image mode 2: [Mpad][32], k = c*9 + t
  u16* out;           // (N, H, W, pitch) bf16; FM_BWD: the gradient w.r.t. the layer's ACTIVATION, read
  long long out_pitch;
  float* partials;    // [gridDim.x][2 * Mpad] or null
  const float* ep_scale;   // FM_EVAL: running-statistics coefficients; FM_BWD: the batch coefficients (mask) ...
  const float* ep_shift;
  const float* bw_mean;    // ... and the batch mean / invstd (xhat)
  const float* bw_invstd;
  int N, C, H, W, M, Mpad;
  int tiles_y, tiles_x, ntiles;
};

// What the tile loop does with the recomputed output y = conv(x):
//   FM_STORE  store it (+ BatchNorm partial sums of the stored values)            -- the unfused train-mode forward
//   FM_EVAL   store relu(y * scale + shift) (running statistics)                   -- eval mode
//   FM_STATS  partial sums only, NO store: the statistics of a raw output that never exists in HBM (gsd_bf16_inc.hip)
//   FM_BWD    pass 1 of the layer's BatchNorm + ReLU backward WITHOUT the stored y: the gradient da w.r.t. the activation is
//             read where the forward would store, dz = da where y * scale + shift > 0, partials = [sum dz | sum dz * xhat]
enum { FM_STORE = 0, FM_EVAL = 1, FM_STATS = 2, FM_BWD = 3 };

// Forward.  Block = 4 waves, wave w owns image row h0 + w of the 4 x 64 tile: four 16-pixel MFMA tiles x MT m-tiles.
// Output channels are permuted as in gsd_bf16_conv.hip (lane group g holds channels g*8 .. g*8+7 and 32 + g*8 ..): 16-byte
// stores, 64 contiguous bytes per pixel and instruction.  The x halo tile of the NEXT pixel tile is fetched into registers
// while this one is multiplied and stored (a tile is ~1 us of work behind ~2 us of load latency).
template <int MT, int FM>
__global__ __launch_bounds__(256, FM == FM_BWD ? 2 : 3) void conv_first_bf16_kernel(const FirstP P) {
  constexpr bool EP = FM == FM_EVAL;
  __shared__ u16 xs[3 * F_PLANE + 8];   // [c][row][col] bf16, + a zero element for k >= 9 C
  __shared__ float sSt[4][2][64];
  __shared__ __attribute__((aligned(16))) float sBw[FM == FM_BWD ? 4 * 64 : 4];   // scale | shift | mean | invstd
  if (FM == FM_BWD && threadIdx.x < P.M) {
    sBw[threadIdx.x] = P.ep_scale[threadIdx.x];
    sBw[64 + threadIdx.x] = P.ep_shift[threadIdx.x];
    sBw[128 + threadIdx.x] = P.bw_mean[threadIdx.x];
    sBw[192 + threadIdx.x] = P.bw_invstd[threadIdx.x];
  }
  constexpr int ZERO = 3 * F_PLANE;
  constexpr int NXE = (3 * F_PLANE + 255) / 256;   // halo elements per thread
  typedef unsigned u32x4s __attribute__((ext_vector_type(4), aligned(8)));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, j = lane & 15;
  if (tid < 8) xs[ZERO + tid] = 0;

  // A operands: MFMA tile m, row i = j holds channel (m>>1)*32 + (j>>2)*8 + (m&1)*4 + (j&3); this lane's k = 8g .. 8g+7
  u32x4 a[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int ch = (m >> 1) * 32 + (j >> 2) * 8 + (m & 1) * 4 + (j & 3);
    a[m] = *reinterpret_cast<const u32x4*>(P.wt + (size_t)ch * 32 + g * 8);
  }
  // B operand gather: k = 8g + e -> (channel c, tap t): halo offset of the tap relative to the output pixel (ZERO: k >= 9 C)
  int off[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 8 * g + e;
    const int c = k / 9, t = k - c * 9;
    off[e] = k < 9 * P.C ? c * F_PLANE + (t / 3) * F_PITCH + (t % 3) : -1;
  }
  const int ch0 = g * 8;   // run A: channels ch0 .. ch0+7 (m-tiles 0, 1); run B: + 32 (m-tiles 2, 3)
  float s1[MT][4], s2[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int e = 0; e < 4; ++e) s1[m][e] = s2[m][e] = 0.f;

  // this thread's halo elements: (channel, row, column) packed, -1 past the tile
  int xel[NXE];
#pragma unroll
  for (int k = 0; k < NXE; ++k) {
    const int i = tid + k * 256;
    const int c = i / F_PLANE, r = (i - c * F_PLANE) / F_PITCH, col = i - c * F_PLANE - r * F_PITCH;
    xel[k] = i < P.C * F_PLANE ? (c << 16 | r << 8 | col) : -1;
  }
  const int tpi = P.tiles_y * P.tiles_x;
  float xv[NXE];
  auto fetch = [&](int tile) {
    const int n = tile / tpi, rem = tile - n * tpi;
    const int ty = rem / P.tiles_x, h0 = ty * F_TH, w0 = (rem - ty * P.tiles_x) * F_TW;
    const float* xn = P.x + (size_t)n * P.C * P.H * P.W;
#pragma unroll
    for (int k = 0; k < NXE; ++k) {
      float v = 0.f;
      if (xel[k] >= 0) {
        const int c = xel[k] >> 16, gh = h0 - 1 + (xel[k] >> 8 & 255), gw = w0 - 1 + (xel[k] & 255);
        if ((unsigned)gh < (unsigned)P.H && (unsigned)gw < (unsigned)P.W) v = xn[((size_t)c * P.H + gh) * P.W + gw];
      }
      xv[k] = v;
    }
  };
  int tile = blockIdx.x;
  if (tile < P.ntiles) fetch(tile);
  for (; tile < P.ntiles; tile += gridDim.x) {
    const int n = tile / tpi, rem = tile - n * tpi;
    const int ty = rem / P.tiles_x, h0 = ty * F_TH, w0 = (rem - ty * P.tiles_x) * F_TW;
    __syncthreads();   // everyone has left the previous tile
#pragma unroll
    for (int k = 0; k < NXE; ++k)
      if (xel[k] >= 0) xs[tid + k * 256] = f32_to_bf16(xv[k]);
    __syncthreads();
    if (tile + (int)gridDim.x < P.ntiles) fetch(tile + gridDim.x);   // flies during this tile's MFMAs and stores
    const int h = h0 + wave;
    if (h < P.H) {     // wave-uniform
      constexpr int TUNR = FM == FM_BWD ? 1 : 4;   // (the backward form unrolled four-wide spills: 64 coefficient + 32 sum registers)
#pragma unroll TUNR
      for (int t = 0; t < 4; ++t) {
        const int px = t * 16 + j, w = w0 + px;
        const int base = wave * F_PITCH + px;
        unsigned v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = xs[off[e] >= 0 ? base + off[e] : ZERO];
        const u32x4 b = {v[0] | v[1] << 16, v[2] | v[3] << 16, v[4] | v[5] << 16, v[6] | v[7] << 16};
        unsigned pk[2 * MT];
        u32x4 da[2];
        if (FM == FM_BWD) {
          da[0] = da[1] = u32x4{0, 0, 0, 0};
          if (w < P.W) {
            const u16* gp = P.out + (((long long)n * P.H + h) * P.W + w) * P.out_pitch + ch0;
            da[0] = *reinterpret_cast<const u32x4s*>(gp);
            if (MT == 4) da[1] = *reinterpret_cast<const u32x4s*>(gp + 32);
          }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          f32x4 acc = mfma_bf16(a[m], b, f32x4{0.f, 0.f, 0.f, 0.f});
          if (FM == FM_BWD) {
            // y as the forward would have stored it; the mask and xhat of gsd_bf16_conv.hip's fused pass 1; the gradient is
            // already rounded (masking commutes with the rounding): sums of the values a stored dz would hold
            const int c = (m >> 1) * 32 + ch0 + (m & 1) * 4;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(sBw + c), sh = *reinterpret_cast<const f32x4*>(sBw + 64 + c);
            const f32x4 mu = *reinterpret_cast<const f32x4*>(sBw + 128 + c), is = *reinterpret_cast<const f32x4*>(sBw + 192 + c);
            const unsigned d01 = da[m >> 1][(m & 1) * 2], d23 = da[m >> 1][(m & 1) * 2 + 1];
            const float dv[4] = {__uint_as_float(d01 << 16), __uint_as_float(d01 & 0xffff0000u), __uint_as_float(d23 << 16),
                                 __uint_as_float(d23 & 0xffff0000u)};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float yv = bf16_to_f32(f32_to_bf16(acc[e]));
              const float q = fmaf(yv, sc[e], sh[e]) > 0.f ? dv[e] : 0.f;
              s1[m][e] += q;
              s2[m][e] = fmaf(q, (yv - mu[e]) * is[e], s2[m][e]);
            }
            continue;
          }
          if (EP) {      // eval: BatchNorm (running statistics) + ReLU
            const int c = (m >> 1) * 32 + ch0 + (m & 1) * 4;
            const f32x4 esc = *reinterpret_cast<const f32x4*>(P.ep_scale + c), esh = *reinterpret_cast<const f32x4*>(P.ep_shift + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaxf(fmaf(acc[e], esc[e], esh[e]), 0.f);
          }
          const unsigned lo = pack_bf16(acc[0], acc[1]), hi = pack_bf16(acc[2], acc[3]);
          pk[2 * m] = lo;
          pk[2 * m + 1] = hi;
          if (!EP && P.partials != nullptr && w < P.W) {   // statistics of the values as stored
            const float q0 = __uint_as_float(lo << 16), q1 = __uint_as_float(lo & 0xffff0000u);
            const float q2 = __uint_as_float(hi << 16), q3 = __uint_as_float(hi & 0xffff0000u);
            s1[m][0] += q0; s2[m][0] = fmaf(q0, q0, s2[m][0]);
            s1[m][1] += q1; s2[m][1] = fmaf(q1, q1, s2[m][1]);
            s1[m][2] += q2; s2[m][2] = fmaf(q2, q2, s2[m][2]);
            s1[m][3] += q3; s2[m][3] = fmaf(q3, q3, s2[m][3]);
          }
        }
        if (FM == FM_STORE || FM == FM_EVAL) {
          if (MT == 4) {   // whole 128-byte lines per instruction (gsd_line_pieces): pixels 0..7 of the step, then 8..15
            u32x4 la, lb;
            gsd_line_pieces({pk[0], pk[1], pk[2], pk[3]}, {pk[4], pk[5], pk[6], pk[7]}, la, lb);
            const int wa = w0 + t * 16 + gsd_line_pixel(j);
            u16* o = P.out + (((long long)n * P.H + h) * P.W + wa) * P.out_pitch + gsd_line_channel(g, j);
            if (wa < P.W) *reinterpret_cast<u32x4s*>(o) = la;
            if (wa + 8 < P.W) *reinterpret_cast<u32x4s*>(o + 8 * P.out_pitch) = lb;
          } else if (w < P.W) {
            u16* o = P.out + (((long long)n * P.H + h) * P.W + w) * P.out_pitch + ch0;
            *reinterpret_cast<u32x4s*>(o) = u32x4{pk[0], pk[1], pk[2], pk[3]};
          }
        }
      }
    }
  }
  if (!EP && P.partials != nullptr) {   // one partial row per block: 16-lane DPP sums, then the four waves through LDS
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a1 = reduce16_to_lane15(s1[m][e]), a2 = reduce16_to_lane15(s2[m][e]);
        if (j == 15) {
          const int c = (m >> 1) * 32 + ch0 + (m & 1) * 4 + e;
          sSt[wave][0][c] = a1;
          sSt[wave][1][c] = a2;
        }
      }
    __syncthreads();
    if (tid < P.M) {
      float* row = P.partials + (size_t)blockIdx.x * (2 * P.Mpad);
      row[tid] = (sSt[0][0][tid] + sSt[1][0][tid]) + (sSt[2][0][tid] + sSt[3][0][tid]);
      row[P.Mpad + tid] = (sSt[0][1][tid] + sSt[1][1][tid]) + (sSt[2][1][tid] + sSt[3][1][tid]);
    }
  }
}

int first_grid(int ntiles) { return ntiles < 2048 ? ntiles : 2048; }   // 8 small blocks per CU; one partial row each

// ---- dW of the first layer --------------------------------------------------------------------------------------------
// D[co][k] = sum over pixels of d_raw[pixel][co] * patch[pixel][k], k = c*9 + t: the reduction runs over PIXELS, so both MFMA
// operands want 8 consecutive pixels per lane.  d_raw (NHWC: channels contiguous) is formed from dz and y in registers, stored
// pixel-major in LDS and read back transposed with ds_read_b64_tr_b16 (pixels {4g..4g+3, 16+4g..} of a 32-pixel step, as in
// gsd_bf16_wgrad.hip); the patches come from THREE copies of the block's x halo tile, copy dx shifted by dx columns, so that
// the 4 consecutive pixels of any tap are one aligned ds_read_b64.
constexpr int G_RS = 64 * 2 + 32;           // d_raw row stride in LDS (bytes): 32 B x odd
constexpr int G_PX = F_TW + 4;              // x copy row pitch (elements; rows stay 8-byte aligned)
constexpr int G_XPLANE = (F_TH + 2) * G_PX;

struct WgFirstBP {
  const float* x;
  const u16* dz;      // (N,H,W,pitch) bf16: dz (y != null) or d_raw itself (y == null)
  long long dz_pitch;
  const u16* y;
  long long y_pitch;
  const float *scale, *mean, *invstd, *c1, *c2;
  const float* shift; // RECOMP: the mask needs y * scale + shift
  const u16* wt0;     // RECOMP: the layer's forward weight image (mode 2)
  float* slabs;       // [gridDim.x][M][32]
  int N, C, H, W, M;
  int tiles_y, tiles_x, ntiles;
};

__device__ __forceinline__ u32x2 tr_read_b64_first(const unsigned char* p) {
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
  return __builtin_bit_cast(u32x2, v);
}

// RECOMP: the layer's raw output y does not exist in HBM (gsd_bf16_inc.hip): `dz` holds the gradient w.r.t. the layer's
// ACTIVATION, and each wave recomputes y for its image row from the x tile that is in LDS anyway (one MFMA k-step per 16 pixels
// and m-tile, rounded to bf16 as the forward would have stored it), masks the gradient with it and forms d_raw in the forward's
// lane layout (a lane owns channels 8g .. 8g+7 and 32 + 8g .. of its pixel).  Same d_raw values, bit for bit, as the (dz, y) form.
template <int MT, bool RECOMP>   // MT = M / 16
__global__ __launch_bounds__(256) void wgrad_first_bf16_kernel(const WgFirstBP P) {
  __shared__ __attribute__((aligned(16))) unsigned char dl[F_TH * F_TW * G_RS];          // d_raw [pixel][64 ch] bf16
  __shared__ __attribute__((aligned(16))) u16 xs[(3 * 3 + 1) * G_XPLANE];                // [dx][c][row][col], + a zero plane
  __shared__ __attribute__((aligned(16))) float sCf[RECOMP ? 6 * 64 : 4];                // scale | shift | mean | invstd | c1 | c2
  constexpr int ZPLANE = 9 * G_XPLANE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15, q = li >> 2, p4 = li & 3;
  for (int i = tid; i < G_XPLANE; i += 256) xs[ZPLANE + i] = 0;
  u32x4 a0w[RECOMP ? MT : 1];
  int offF[8];
  if (RECOMP) {
    if (tid < P.M) {
      sCf[tid] = P.scale[tid]; sCf[64 + tid] = P.shift[tid]; sCf[128 + tid] = P.mean[tid];
      sCf[192 + tid] = P.invstd[tid]; sCf[256 + tid] = P.c1[tid]; sCf[320 + tid] = P.c2[tid];
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {   // forward A operands: MFMA tile m, row li holds channel (m>>1)*32 + (li>>2)*8 + (m&1)*4 + (li&3)
      const int ch = (m >> 1) * 32 + (li >> 2) * 8 + (m & 1) * 4 + (li & 3);
      a0w[m] = *reinterpret_cast<const u32x4*>(P.wt0 + (size_t)ch * 32 + g * 8);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {    // forward B gather: k = 8g + e = (c, t) -> copy t % 3 of channel c, row offset t / 3
      const int k = 8 * g + e;
      const int c = k / 9, t = k - c * 9;
      offF[e] = (k < 9 * P.C ? ((t % 3) * 3 + c) * G_XPLANE + (t / 3) * G_PX : ZPLANE) + wave * G_PX;
    }
  }

  // this thread's 8 channels of every pixel it converts (tid % 8 is the same for all of them)
  const int gk = tid & 7;
  float sc[8], mu[8], is[8], k1[8], k2[8];
  const bool fused = P.y != nullptr;
  const bool gk_ok = gk * 8 < P.M;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = gk_ok ? gk * 8 + i : 0;
    sc[i] = fused ? P.scale[c] : 1.f;
    mu[i] = fused ? P.mean[c] : 0.f;
    is[i] = fused ? P.invstd[c] : 0.f;
    k1[i] = fused ? P.c1[c] : 0.f;
    k2[i] = fused ? P.c2[c] : 0.f;
  }
  // B operand (patches): this lane's column j = li of n-tile nt is k = nt*16 + li = (c, t): copy dx = t % 3, row offset t / 3
  int boff[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int k = nt * 16 + li;
    const int c = k / 9, t = k - c * 9;
    boff[nt] = (k < 9 * P.C ? ((t % 3) * 3 + c) * G_XPLANE + (wave + t / 3) * G_PX : ZPLANE) + 4 * g;   // + step*32 (+16)
  }
  const int a_rd = (wave * F_TW + 4 * g + q) * G_RS + (4 * p4) * 2;   // + step*32*G_RS (+16*G_RS) + m*32

  f32x4 acc[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m][0] = acc[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int tpi = P.tiles_y * P.tiles_x;
  for (int tile = blockIdx.x; tile < P.ntiles; tile += gridDim.x) {
    const int n = tile / tpi, rem = tile - n * tpi;
    const int ty = rem / P.tiles_x, h0 = ty * F_TH, w0 = (rem - ty * P.tiles_x) * F_TW;
    __syncthreads();   // everyone has left the previous tile
    // x halo tile, three column-shifted bf16 copies
    for (int i = tid; i < P.C * (F_TH + 2) * (F_TW + 2); i += 256) {
      const int c = i / ((F_TH + 2) * (F_TW + 2)), r = (i / (F_TW + 2)) % (F_TH + 2), col = i % (F_TW + 2);
      const int gh = h0 - 1 + r, gw = w0 - 1 + col;
      float v = 0.f;
      if ((unsigned)gh < (unsigned)P.H && (unsigned)gw < (unsigned)P.W) v = P.x[(((size_t)n * P.C + c) * P.H + gh) * P.W + gw];
      const u16 b = f32_to_bf16(v);
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
        if (col - dx >= 0 && col - dx < F_TW) xs[(dx * 3 + c) * G_XPLANE + r * G_PX + col - dx] = b;
    }
    if (RECOMP) {
      __syncthreads();   // the x copies are complete (the recomputation reads them)
      typedef unsigned u32x4s __attribute__((ext_vector_type(4), aligned(8)));
      const int h = h0 + wave;
#pragma unroll 2
      for (int nt = 0; nt < 4; ++nt) {
        const int px = nt * 16 + li, w = w0 + px;
        const bool ok = h < P.H && w < P.W;
        u32x4 da[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};
        if (ok) {
          const u16* gp = P.dz + (((long long)n * P.H + h) * P.W + w) * P.dz_pitch + g * 8;
          da[0] = *reinterpret_cast<const u32x4s*>(gp);
          if (MT == 4) da[1] = *reinterpret_cast<const u32x4s*>(gp + 32);
        }
        unsigned v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = xs[offF[e] + px];
        const u32x4 b = {v[0] | v[1] << 16, v[2] | v[3] << 16, v[4] | v[5] << 16, v[6] | v[7] << 16};
        unsigned pk[2 * MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const f32x4 acc = mfma_bf16(a0w[m], b, f32x4{0.f, 0.f, 0.f, 0.f});
          const int c = (m >> 1) * 32 + g * 8 + (m & 1) * 4;
          const f32x4 sc = *reinterpret_cast<const f32x4*>(sCf + c), sh = *reinterpret_cast<const f32x4*>(sCf + 64 + c);
          const f32x4 mu = *reinterpret_cast<const f32x4*>(sCf + 128 + c), is = *reinterpret_cast<const f32x4*>(sCf + 192 + c);
          const f32x4 k1 = *reinterpret_cast<const f32x4*>(sCf + 256 + c), k2 = *reinterpret_cast<const f32x4*>(sCf + 320 + c);
          const unsigned d01 = da[m >> 1][(m & 1) * 2], d23 = da[m >> 1][(m & 1) * 2 + 1];
          const float dv[4] = {__uint_as_float(d01 << 16), __uint_as_float(d01 & 0xffff0000u), __uint_as_float(d23 << 16),
                               __uint_as_float(d23 & 0xffff0000u)};
          float d[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float yv = bf16_to_f32(f32_to_bf16(acc[e]));                     // y as the forward would have stored it
            const float dzv = fmaf(yv, sc[e], sh[e]) > 0.f ? dv[e] : 0.f;          // pass 1's mask
            const float xh = (yv - mu[e]) * is[e];
            d[e] = ok ? sc[e] * (dzv - k1[e] - xh * k2[e]) : 0.f;                  // bn_bwd_apply_bf16_kernel's expression
          }
          pk[2 * m] = pack_bf16(d[0], d[1]);
          pk[2 * m + 1] = pack_bf16(d[2], d[3]);
        }
        unsigned char* dp = dl + (wave * F_TW + px) * G_RS + g * 16;
        *reinterpret_cast<uint4*>(dp) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        if (MT == 4) *reinterpret_cast<uint4*>(dp + 64) = make_uint4(pk[4], pk[5], pk[6], pk[7]);
      }
    } else {
    // d_raw tile: item = (pixel, 8-channel group), a wave covers 8 consecutive pixels x 128 B
  #pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int pxl = (it * 256 + tid) >> 3;               // 0..255: row pxl / 64, column pxl % 64
        const int r = pxl / F_TW, c = pxl - r * F_TW;
        const int h = h0 + r, w = w0 + c;
        float d[8];
  #pragma unroll
        for (int i = 0; i < 8; ++i) d[i] = 0.f;
        if (h < P.H && w < P.W && gk_ok) {
          const long long pix = ((long long)n * P.H + h) * P.W + w;
          const uint4 dv = *reinterpret_cast<const uint4*>(P.dz + pix * P.dz_pitch + gk * 8);
          d[0] = __uint_as_float(dv.x << 16); d[1] = __uint_as_float(dv.x & 0xffff0000u);
          d[2] = __uint_as_float(dv.y << 16); d[3] = __uint_as_float(dv.y & 0xffff0000u);
          d[4] = __uint_as_float(dv.z << 16); d[5] = __uint_as_float(dv.z & 0xffff0000u);
          d[6] = __uint_as_float(dv.w << 16); d[7] = __uint_as_float(dv.w & 0xffff0000u);
          if (fused) {
            const uint4 yv4 = *reinterpret_cast<const uint4*>(P.y + pix * P.y_pitch + gk * 8);
            const float yv[8] = {__uint_as_float(yv4.x << 16), __uint_as_float(yv4.x & 0xffff0000u), __uint_as_float(yv4.y << 16),
                                 __uint_as_float(yv4.y & 0xffff0000u), __uint_as_float(yv4.z << 16), __uint_as_float(yv4.z & 0xffff0000u),
                                 __uint_as_float(yv4.w << 16), __uint_as_float(yv4.w & 0xffff0000u)};
  #pragma unroll
            for (int i = 0; i < 8; ++i) {      // the expression of bn_bwd_apply_bf16_kernel, rounded to bf16 as it stores it
              const float xh = (yv[i] - mu[i]) * is[i];
              d[i] = sc[i] * (d[i] - k1[i] - xh * k2[i]);
            }
          }
        }
        *reinterpret_cast<uint4*>(dl + pxl * G_RS + gk * 16) =
            make_uint4(pack_bf16(d[0], d[1]), pack_bf16(d[2], d[3]), pack_bf16(d[4], d[5]), pack_bf16(d[6], d[7]));
      }
    }
    __syncthreads();
    // wave w: image row h0 + w = 64 pixels = two 32-pixel k-steps
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      u32x4 b[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const u16* bp = xs + boff[nt] + st * 32;
        const u32x2 lo = *reinterpret_cast<const u32x2*>(bp), hi = *reinterpret_cast<const u32x2*>(bp + 16);
        b[nt] = u32x4{lo[0], lo[1], hi[0], hi[1]};
      }
      const unsigned char* ap = dl + a_rd + st * 32 * G_RS;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const u32x2 lo = tr_read_b64_first(ap + m * 32), hi = tr_read_b64_first(ap + m * 32 + 16 * G_RS);
        const u32x4 a = u32x4{lo[0], lo[1], hi[0], hi[1]};
        acc[m][0] = mfma_bf16(a, b[0], acc[m][0]);
        acc[m][1] = mfma_bf16(a, b[1], acc[m][1]);
      }
    }
  }
  // block total: the four waves through LDS (reusing the d_raw tile), one [M][32] slab per block
  __syncthreads();
  float* red = reinterpret_cast<float*>(dl);   // [4 waves][M][32]
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[(wave * P.M + m * 16 + 4 * g + e) * 32 + nt * 16 + li] = acc[m][nt][e];
  __syncthreads();
  float* slab = P.slabs + (size_t)blockIdx.x * P.M * 32;
  for (int i = tid; i < P.M * 32; i += 256)
    slab[i] = (red[i] + red[P.M * 32 + i]) + (red[2 * P.M * 32 + i] + red[3 * P.M * 32 + i]);
}

// dW[co][k] = sum over the blocks' slabs in a fixed order, k < 9 C
__global__ __launch_bounds__(256) void wgrad_first_bf16_reduce(const float* __restrict__ slabs, int nslabs, int M, int K, float* __restrict__ dw) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= M * K) return;
  const int co = i / K, k = i - co * K;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const float* p = slabs + co * 32 + k;
  int b = 0;
  for (; b + 3 < nslabs; b += 4) {
    s0 += p[(size_t)b * M * 32];
    s1 += p[(size_t)(b + 1) * M * 32];
    s2 += p[(size_t)(b + 2) * M * 32];
    s3 += p[(size_t)(b + 3) * M * 32];
  }
  for (; b < nslabs; ++b) s0 += p[(size_t)b * M * 32];
  dw[i] = (s0 + s1) + (s2 + s3);
}

int wg_first_grid(int ntiles) { return ntiles < 768 ? ntiles : 768; }   // 3 blocks per CU (LDS), one slab each

}  // namespace

extern "C" int gsd_bf16_conv3x3_first_supported(int C, int M) { return (C >= 1 && 9 * C <= 32 && (M == 32 || M == 64)) ? 1 : 0; }

extern "C" int gsd_bf16_conv3x3_first_partial_rows(int N, int H, int W, int M) {
  if (N <= 0 || H <= 0 || W <= 0 || M <= 0) return 0;
  const long nt = (long)N * ceil_div(H, F_TH) * ceil_div(W, F_TW);
  return nt < 2147483647L ? first_grid((int)nt) : 0;
}

static int first_launch(FirstP& P, int fm, void* stream, const char* what) {
  P.tiles_y = ceil_div(P.H, F_TH); P.tiles_x = ceil_div(P.W, F_TW);
  const long nt = (long)P.N * P.tiles_y * P.tiles_x;
  GSD_REQUIRE(nt < 2147483647L, GSD_ERR_UNSUPPORTED, "%s: too many tiles", what);
  P.ntiles = (int)nt;
  const int grid = first_grid(P.ntiles);
  const dim3 gd(grid), bd(256);
  hipStream_t st = (hipStream_t)stream;
#define GSD_FIRST_CASE(MT_, FM_) hipLaunchKernelGGL((conv_first_bf16_kernel<MT_, FM_>), gd, bd, 0, st, P)
  if (P.M == 64) {
    switch (fm) {
      case FM_STORE: GSD_FIRST_CASE(4, FM_STORE); break;
      case FM_EVAL: GSD_FIRST_CASE(4, FM_EVAL); break;
      case FM_STATS: GSD_FIRST_CASE(4, FM_STATS); break;
      default: GSD_FIRST_CASE(4, FM_BWD); break;
    }
  } else {
    switch (fm) {
      case FM_STORE: GSD_FIRST_CASE(2, FM_STORE); break;
      case FM_EVAL: GSD_FIRST_CASE(2, FM_EVAL); break;
      case FM_STATS: GSD_FIRST_CASE(2, FM_STATS); break;
      default: GSD_FIRST_CASE(2, FM_BWD); break;
    }
  }
#undef GSD_FIRST_CASE
  GSD_LAUNCH_CHECK(what);
  return GSD_OK;
}

extern "C" int gsd_bf16_conv3x3_first(const float* x, int N, int C, int H, int W, const void* wt, const gsd_nhwc* out, int M,
                                      float* partials, const float* ep_scale, const float* ep_shift, void* stream) {
  GSD_REQUIRE(x && wt, GSD_ERR_BAD_ARG, "gsd_bf16_conv3x3_first: null argument");
  GSD_REQUIRE(gsd_bf16_conv3x3_first_supported(C, M), GSD_ERR_UNSUPPORTED,
              "gsd_bf16_conv3x3_first: needs 9*C <= 32 and M in {32, 64} (got C=%d M=%d); use gsd_bf16_im2col3x3 + gsd_bf16_conv_dense", C, M);
  GSD_REQUIRE(N > 0 && H > 0 && W > 0, GSD_ERR_BAD_ARG, "gsd_bf16_conv3x3_first: bad sizes");
  GSD_REQUIRE((ep_scale == nullptr) == (ep_shift == nullptr) && !(ep_scale != nullptr && partials != nullptr), GSD_ERR_BAD_ARG,
              "gsd_bf16_conv3x3_first: eval coefficients come together and exclude the statistics");
  GSD_REQUIRE(((uintptr_t)wt & 15) == 0, GSD_ERR_BAD_ARG, "gsd_bf16_conv3x3_first: the weight image must be 16-byte aligned");
  FirstP P;
  P.x = x; P.wt = (const u16*)wt; P.out = nullptr; P.out_pitch = 0;
  P.partials = partials; P.ep_scale = ep_scale; P.ep_shift = ep_shift; P.bw_mean = P.bw_invstd = nullptr;
  P.N = N; P.C = C; P.H = H; P.W = W; P.M = M; P.Mpad = round_up(M, 128);
  if (out == nullptr) {   // statistics only: the raw output is never stored (gsd_bf16_inc_conv rebuilds it tile by tile)
    GSD_REQUIRE(partials != nullptr && ep_scale == nullptr, GSD_ERR_BAD_ARG,
                "gsd_bf16_conv3x3_first: without an output tensor the call must ask for the statistics (partials)");
    return first_launch(P, FM_STATS, stream, "gsd_bf16_conv3x3_first");
  }
  if (int e = gsd_check_nhwc(out, "gsd_bf16_conv3x3_first out")) return e;
  GSD_REQUIRE(out->N == N && out->H == H && out->W == W && out->C == M && (out->pitch & 3) == 0, GSD_ERR_BAD_ARG,
              "gsd_bf16_conv3x3_first: out must be (N,H,W,M)");
  P.out = (u16*)out->ptr; P.out_pitch = out->pitch;
  return first_launch(P, ep_scale != nullptr ? FM_EVAL : FM_STORE, stream, "gsd_bf16_conv3x3_first");
}

extern "C" int gsd_bf16_first_bn_bwd_reduce(const float* x, int N, int C, int H, int W, const void* wt, const gsd_nhwc* da,
                                            const float* scale, const float* shift, const float* mean, const float* invstd,
                                            float* partials, void* stream) {
  GSD_REQUIRE(x && wt && scale && shift && mean && invstd && partials, GSD_ERR_BAD_ARG, "gsd_bf16_first_bn_bwd_reduce: null argument");
  if (int e = gsd_check_nhwc(da, "gsd_bf16_first_bn_bwd_reduce da")) return e;
  const int M = da->C;
  GSD_REQUIRE(gsd_bf16_conv3x3_first_supported(C, M), GSD_ERR_UNSUPPORTED,
              "gsd_bf16_first_bn_bwd_reduce: needs 9*C <= 32 and M in {32, 64} (got C=%d M=%d)", C, M);
  GSD_REQUIRE(da->N == N && da->H == H && da->W == W && (da->pitch & 3) == 0, GSD_ERR_BAD_ARG,
              "gsd_bf16_first_bn_bwd_reduce: da must be (N,H,W,M)");
  GSD_REQUIRE(((uintptr_t)wt & 15) == 0, GSD_ERR_BAD_ARG, "gsd_bf16_first_bn_bwd_reduce: the weight image must be 16-byte aligned");
  FirstP P;
  P.x = x; P.wt = (const u16*)wt; P.out = (u16*)da->ptr; P.out_pitch = da->pitch;
  P.partials = partials; P.ep_scale = scale; P.ep_shift = shift; P.bw_mean = mean; P.bw_invstd = invstd;
  P.N = N; P.C = C; P.H = H; P.W = W; P.M = M; P.Mpad = round_up(M, 128);
  return first_launch(P, FM_BWD, stream, "gsd_bf16_first_bn_bwd_reduce");
}

extern "C" int64_t gsd_bf16_wgrad_first_workspace(int N, int H, int W, int M) {
  if (N <= 0 || H <= 0 || W <= 0 || M <= 0) return 0;
  const long nt = (long)N * ceil_div(H, F_TH) * ceil_div(W, F_TW);
  return (int64_t)wg_first_grid(nt < 2147483647L ? (int)nt : 2147483647) * M * 32;
}

static int wgrad_first_impl(const float* x, int N, int C, int H, int W, const gsd_nhwc* dz, const gsd_nhwc* y, const void* wt0,
                            const float* scale, const float* shift, const float* mean, const float* invstd, const float* c1,
                            const float* c2, float* dw, float* workspace, int64_t workspace_elems, void* stream) {
  const bool recompute = wt0 != nullptr;
  GSD_REQUIRE(x && dw && workspace, GSD_ERR_BAD_ARG, "gsd_bf16_wgrad_first: null argument");
  if (int e = gsd_check_nhwc(dz, "gsd_bf16_wgrad_first dz")) return e;
  const int M = dz->C;
  GSD_REQUIRE(gsd_bf16_conv3x3_first_supported(C, M), GSD_ERR_UNSUPPORTED,
              "gsd_bf16_wgrad_first: needs 9*C <= 32 and M in {32, 64} (got C=%d M=%d); use gsd_bf16_im2col3x3 + gsd_bf16_wgrad", C, M);
  GSD_REQUIRE(dz->N == N && dz->H == H && dz->W == W, GSD_ERR_BAD_ARG, "gsd_bf16_wgrad_first: dz must be (N,H,W,M)");
  if (y != nullptr) {
    if (int e = gsd_check_nhwc(y, "gsd_bf16_wgrad_first y")) return e;
    GSD_REQUIRE(y->N == N && y->H == H && y->W == W && y->C == M && scale && mean && invstd && c1 && c2, GSD_ERR_BAD_ARG,
                "gsd_bf16_wgrad_first: the fused BatchNorm backward needs y (N,H,W,M) and its five coefficient vectors");
  }
  if (recompute) {
    GSD_REQUIRE(y == nullptr && scale && shift && mean && invstd && c1 && c2 && ((uintptr_t)wt0 & 15) == 0 && (dz->pitch & 3) == 0,
                GSD_ERR_BAD_ARG, "gsd_bf16_wgrad_first_recompute: needs the forward weight image (16-byte aligned) and six coefficient vectors");
  }
  WgFirstBP P;
  P.x = x; P.dz = (const u16*)dz->ptr; P.dz_pitch = dz->pitch;
  P.y = y ? (const u16*)y->ptr : nullptr; P.y_pitch = y ? y->pitch : 0;
  P.scale = scale; P.mean = mean; P.invstd = invstd; P.c1 = c1; P.c2 = c2;
  P.shift = shift; P.wt0 = (const u16*)wt0;
  P.slabs = workspace;
  P.N = N; P.C = C; P.H = H; P.W = W; P.M = M;
  P.tiles_y = ceil_div(H, F_TH); P.tiles_x = ceil_div(W, F_TW);
  const long nt = (long)N * P.tiles_y * P.tiles_x;
  GSD_REQUIRE(nt < 2147483647L, GSD_ERR_UNSUPPORTED, "gsd_bf16_wgrad_first: too many tiles");
  P.ntiles = (int)nt;
  const int grid = wg_first_grid(P.ntiles);
  GSD_REQUIRE(workspace_elems >= (int64_t)grid * M * 32, GSD_ERR_WORKSPACE, "gsd_bf16_wgrad_first: workspace too small");
  if (recompute) {
    if (M == 64) hipLaunchKernelGGL((wgrad_first_bf16_kernel<4, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, P);
    else hipLaunchKernelGGL((wgrad_first_bf16_kernel<2, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, P);
  } else {
    if (M == 64) hipLaunchKernelGGL((wgrad_first_bf16_kernel<4, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, P);
    else hipLaunchKernelGGL((wgrad_first_bf16_kernel<2, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, P);
  }
  GSD_LAUNCH_CHECK("gsd_bf16_wgrad_first");
  hipLaunchKernelGGL(wgrad_first_bf16_reduce, dim3(ceil_div(M * 9 * C, 256)), dim3(256), 0, (hipStream_t)stream, workspace, grid, M,
                     9 * C, dw);
  GSD_LAUNCH_CHECK("gsd_bf16_wgrad_first reduce");
  return GSD_OK;
}

extern "C" int gsd_bf16_wgrad_first(const float* x, int N, int C, int H, int W, const gsd_nhwc* dz, const gsd_nhwc* y,
                                    const float* scale, const float* mean, const float* invstd, const float* c1, const float* c2,
                                    float* dw, float* workspace, int64_t workspace_elems, void* stream) {
  return wgrad_first_impl(x, N, C, H, W, dz, y, nullptr, scale, nullptr, mean, invstd, c1, c2, dw, workspace, workspace_elems, stream);
}

extern "C" int gsd_bf16_wgrad_first_recompute(const float* x, int N, int C, int H, int W, const void* wt, const gsd_nhwc* da,
                                              const float* scale, const float* shift, const float* mean, const float* invstd,
                                              const float* c1, const float* c2, float* dw, float* workspace,
                                              int64_t workspace_elems, void* stream) {
  GSD_REQUIRE(wt != nullptr, GSD_ERR_BAD_ARG, "gsd_bf16_wgrad_first_recompute: null weight image");
  return wgrad_first_impl(x, N, C, H, W, da, nullptr, wt, scale, shift, mean, invstd, c1, c2, dw, workspace, workspace_elems, stream);
}
