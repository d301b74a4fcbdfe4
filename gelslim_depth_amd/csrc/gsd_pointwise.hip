// gsd_pointwise.hip -- the HBM-bound kernels of the U-Net train step (gfx950): BatchNorm statistics,
// BatchNorm/ReLU/max-pool backward, max-pool forward, 1x1 output conv, loss, fused Adam+EMA, weight
// re-layouts.  All of them are coalesced streaming passes along W (NCHW rows); reductions are
// two-stage and ordered (bitwise reproducible), never float atomics.
#include "gsd_common.h"

#include <cstdarg>
#include <cstdio>

// ---------------------------------------------------------------------------------------------
// error plumbing / library info
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void gsd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* gsd_last_error(void) { return g_err; }
extern "C" const char* gsd_version(void) { return "libgsd 0.1 (gfx950, fp32 MFMA 16x16x4)"; }

// ---------------------------------------------------------------------------------------------
// MFMA lane-map self test
// ---------------------------------------------------------------------------------------------
__global__ void selftest_mfma_kernel(const float* a, const float* b, float* out) {
  const int lane = threadIdx.x;
  const float av = a[(lane & 15) * 4 + (lane >> 4)];   // A[i][k], row-major 16x4
  const float bv = b[(lane >> 4) * 16 + (lane & 15)];  // B[k][j], row-major 4x16
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = mfma16(av, bv, c);
#pragma unroll
  for (int r = 0; r < 4; ++r) out[((lane >> 4) * 4 + r) * 16 + (lane & 15)] = c[r];
}
extern "C" int gsd_selftest_mfma(const float* a, const float* b, float* out, void* stream) {
  GSD_REQUIRE(a && b && out, GSD_ERR_BAD_ARG, "gsd_selftest_mfma: null argument");
  hipLaunchKernelGGL(selftest_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, out);
  GSD_LAUNCH_CHECK("gsd_selftest_mfma");
  return GSD_OK;
}

// ---------------------------------------------------------------------------------------------
// weight re-layouts
// ---------------------------------------------------------------------------------------------
// modes 0/1 (conv3x3): tiled for the LDS-DMA kernel: [mblock][k row][BM], BM = 64 if M <= 64 else 128, so that
//   one K-chunk of one m-block is ONE contiguous LDS image (36 x BM floats).  Inside every group of 64 columns
//   (one wave's output channels) the order is permuted: storage slot l*4+m holds column m*16+l, so the four
//   MFMA A operands of a lane (its 4 m-tiles) are one aligned float4 in LDS.
// modes 2/3 (convT): plain [k row][Mpad].
// mode 6 (convT forward, LDS-DMA kernel): like modes 0/1 with BM = 128 and k rows padded to 32: [mblock][k row][128],
//   one K-chunk of 32 input channels of one m-block is one contiguous 16 KiB LDS image, columns permuted as above.
// mode 7 (convT dgrad, LDS-DMA kernel): the same image shape with k = co*4+kh*2+kw (8 output channels per chunk), m = ci.
static void layout_dims(int mode, int Co, int Ci, int* rows, int* M, int* BM, int* pitch, int* mblocks) {
  switch (mode) {
    case 0: *rows = round_up(Ci, 4) * 9; *M = Co; break;        // k = ci*9+t        m = co
    case 1: *rows = round_up(Co, 4) * 9; *M = Ci; break;        // k = co*9+t (flip) m = ci
    case 2: *rows = round_up(Ci, 16); *M = Co * 4; break;       // k = ci            m = co*4+khkw
    case 3: *rows = round_up(Co, 4) * 4; *M = Ci; break;        // k = co*4+khkw     m = ci
    case 4: *rows = round_up(Ci, 4) * 18; *M = Co; break;       // k = ci*18+r*6+f   m = co   (Winograd F(4,3) rows)
    case 5: *rows = round_up(Co, 4) * 18; *M = Ci; break;       // k = co*18+r*6+f (flip) m = ci
    case 7: *rows = round_up(Co, 8) * 4; *M = Ci; break;        // k = co*4+khkw     m = ci   (convT dgrad, LDS-DMA kernel)
    case 8: *rows = round_up(Ci, 4) * 24; *M = Co; break;       // k = ci*24+fr*6+fc m = co   (Winograd F(2x4,3x3), gsd_conv3x3_w2d)
    case 9: *rows = round_up(Co, 4) * 24; *M = Ci; break;       // k = co*24+fr*6+fc (flip) m = ci
    default: *rows = round_up(Ci, 32); *M = Co * 4; break;      // k = ci            m = co*4+khkw  (mode 6)
  }
  if (mode == 6 || mode == 7) {
    *BM = 128;
    *pitch = 128;
    *mblocks = ceil_div(*M, 128);
  } else if (mode >= 4) {
    *BM = 64;
    *pitch = 64;
    *mblocks = ceil_div(*M, 64);
  } else if (mode <= 1) {
    *BM = *M <= 64 ? 64 : 128;
    *pitch = *BM;
    *mblocks = ceil_div(*M, *BM);
  } else {
    *BM = round_up(*M, 64);
    *pitch = *BM;
    *mblocks = 1;
  }
}
extern "C" int64_t gsd_weight_layout_size(int mode, int Co, int Ci) {
  if (mode < 0 || mode > 9 || Co <= 0 || Ci <= 0) return 0;
  int rows, M, BM, pitch, mblocks;
  layout_dims(mode, Co, Ci, &rows, &M, &BM, &pitch, &mblocks);
  return (int64_t)mblocks * rows * pitch;
}
__global__ void weight_layout_kernel(int mode, const float* __restrict__ w, int Co, int Ci, float* __restrict__ wt,
                                     int rows, int M, int BM, int pitch, int mblocks) {
  const long long total = (long long)mblocks * rows * pitch;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int col = (int)(e % pitch);
    const long long t = e / pitch;
    const int k = (int)(t % rows);
    const int mb = (int)(t / rows);
    int m = mb * BM + col;
    if (mode <= 1 || mode >= 4) {  // un-permute: slot (l*4 + t) of a 64-column group holds column t*16 + l
      const int slot = col & 63;
      m = mb * BM + (col & ~63) + (slot & 3) * 16 + (slot >> 2);
    }
    float v = 0.f;
    if (col < BM && m < M) {
      if (mode == 0) {
        const int ci = k / 9, tp = k % 9;
        if (ci < Ci) v = w[((size_t)m * Ci + ci) * 9 + tp];
      } else if (mode == 1) {
        const int co = k / 9, tp = k % 9;
        if (co < Co) v = w[((size_t)co * Ci + m) * 9 + (8 - tp)];
      } else if (mode == 6) {
        if (k < Ci) v = w[(size_t)k * M + m];  // (Ci, Co*4) is already [k][m]
      } else if (mode == 7) {
        if ((k >> 2) < Co) v = w[(size_t)m * (Co * 4) + k];   // W[ci][co][kh][kw] -> [k = co*4+kh*2+kw][m = ci]
      } else if (mode >= 4) {
        // U = G g for the 3 taps g of kernel row r (dX: the flipped kernel, channels swapped), G of F(4,3):
        // rows (1/4,0,0) (-1/6,-1/6,-1/6) (-1/6,1/6,-1/6) (1/24,1/12,1/6) (1/24,-1/12,1/6) (0,0,1)
        const int kch = k / 18, rem = k % 18, r = rem / 6, f = rem % 6;
        if (kch < (mode == 4 ? Ci : Co)) {
          const float* g = mode == 4 ? w + ((size_t)m * Ci + kch) * 9 + r * 3 : w + ((size_t)kch * Ci + m) * 9 + (2 - r) * 3;
          const float g0 = mode == 4 ? g[0] : g[2], g1 = g[1], g2 = mode == 4 ? g[2] : g[0];
          switch (f) {
            case 0: v = g0 * 0.25f; break;
            case 1: v = -(g0 + g1 + g2) * (1.f / 6.f); break;
            case 2: v = -(g0 - g1 + g2) * (1.f / 6.f); break;
            case 3: v = g0 * (1.f / 24.f) + g1 * (1.f / 12.f) + g2 * (1.f / 6.f); break;
            case 4: v = g0 * (1.f / 24.f) - g1 * (1.f / 12.f) + g2 * (1.f / 6.f); break;
            default: v = g2; break;
          }
        }
      } else if (mode == 2) {
        if (k < Ci) v = w[(size_t)k * M + m];  // (Ci, Co*4) is already [k][m]
      } else {
        const int co = k >> 2;
        if (co < Co) v = w[(size_t)m * (Co * 4) + k];
      }
    }
    wt[e] = v;
  }
}
// Modes 4 / 5 (Winograd U = G g), one thread per (m-block, k channel, kernel row, column): the three taps are read once and
// the six transformed values written (the generic kernel reads them, and divides its way to them, once per OUTPUT element).
__global__ __launch_bounds__(256) void weight_layout_w43_kernel(int mode, const float* __restrict__ w, int Co, int Ci,
                                                                 float* __restrict__ wt, int kpad, int M, int mblocks) {
  const long long total = (long long)mblocks * kpad * 3 * 64;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int col = (int)(e & 63);
    long long t = e >> 6;
    const int r = (int)(t % 3);
    t /= 3;
    const int kch = (int)(t % kpad);
    const int mb = (int)(t / kpad);
    const int m = mb * 64 + (col & 3) * 16 + (col >> 2);   // slot l*4 + t of a 64-column group holds column t*16 + l
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (m < M && kch < (mode == 4 ? Ci : Co)) {
      const float* g = mode == 4 ? w + ((size_t)m * Ci + kch) * 9 + r * 3 : w + ((size_t)kch * Ci + m) * 9 + (2 - r) * 3;
      g0 = mode == 4 ? g[0] : g[2];
      g1 = g[1];
      g2 = mode == 4 ? g[2] : g[0];
    }
    float* o = wt + (((size_t)mb * kpad + kch) * 18 + r * 6) * 64 + col;
    o[0] = g0 * 0.25f;
    o[64] = -(g0 + g1 + g2) * (1.f / 6.f);
    o[128] = -(g0 - g1 + g2) * (1.f / 6.f);
    o[192] = g0 * (1.f / 24.f) + g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
    o[256] = g0 * (1.f / 24.f) - g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
    o[320] = g2;
  }
}
// Modes 8 / 9 (two-dimensional Winograd U = G2 g G4^T), one thread per (m-block, k channel, output channel of the block): the nine
// taps are read once and the 24 transformed values written.  G2 of F(2,3): rows (1,0,0) (1/2,1/2,1/2) (1/2,-1/2,1/2) (0,0,1); G4 of
// F(4,3) as above.  Image of one (m-block, 4-channel chunk): [ci & 3][frequency pair f >> 1 (12)][channel half (2)][l (16)][f & 1][m-tile
// of the half (2)] with channel = half*32 + m-tile*16 + l: a wave of gsd_conv3x3_w2d owns one channel half, and its 16 lanes l read
// the two frequencies x two m-tiles of a pair as 16 consecutive 16-byte pieces.
__global__ __launch_bounds__(256) void weight_layout_w2d_kernel(int mode, const float* __restrict__ w, int Co, int Ci,
                                                                 float* __restrict__ wt, int kpad, int M, int mblocks) {
  const long long total = (long long)mblocks * kpad * 64;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int cm = (int)(e & 63);           // channel inside the m-block
    const long long t = e >> 6;
    const int kch = (int)(t % kpad);
    const int mb = (int)(t / kpad);
    const int m = mb * 64 + cm;
    float g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) g[r][c] = 0.f;
    if (m < M && kch < (mode == 8 ? Ci : Co)) {
      // forward: g = W[m][kch]; dX: the flipped kernel with the channels swapped, g[r][c] = W[kch][m][2-r][2-c]
      const float* src = mode == 8 ? w + ((size_t)m * Ci + kch) * 9 : w + ((size_t)kch * Ci + m) * 9;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) g[r][c] = mode == 8 ? src[r * 3 + c] : src[(2 - r) * 3 + (2 - c)];
    }
    const int half = cm >> 5, mtl = (cm >> 4) & 1, l = cm & 15;
    float* o = wt + ((size_t)mb * kpad + (kch & ~3)) * (24 * 64) + (size_t)(kch & 3) * (12 * 128) + half * 64 + l * 4 + mtl;
#pragma unroll
    for (int fr = 0; fr < 4; ++fr) {
      float gr[3];   // G2 down the kernel's rows
#pragma unroll
      for (int c = 0; c < 3; ++c)
        gr[c] = fr == 0 ? g[0][c] : fr == 3 ? g[2][c] : fr == 1 ? 0.5f * (g[0][c] + g[1][c] + g[2][c]) : 0.5f * (g[0][c] - g[1][c] + g[2][c]);
      const float g0 = gr[0], g1 = gr[1], g2 = gr[2];
      float u[6];
      u[0] = g0 * 0.25f;
      u[1] = -(g0 + g1 + g2) * (1.f / 6.f);
      u[2] = -(g0 - g1 + g2) * (1.f / 6.f);
      u[3] = g0 * (1.f / 24.f) + g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
      u[4] = g0 * (1.f / 24.f) - g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
      u[5] = g2;
#pragma unroll
      for (int fc = 0; fc < 6; ++fc) {
        const int f = fr * 6 + fc;
        o[(f >> 1) * 128 + (f & 1) * 2] = u[fc];
      }
    }
  }
}
extern "C" int gsd_weight_layout(int mode, const float* w, int Co, int Ci, float* wt, void* stream) {
  GSD_REQUIRE(w && wt && mode >= 0 && mode <= 9 && Co > 0 && Ci > 0, GSD_ERR_BAD_ARG, "gsd_weight_layout: bad argument");
  int rows, M, BM, pitch, mblocks;
  layout_dims(mode, Co, Ci, &rows, &M, &BM, &pitch, &mblocks);
  if (mode == 8 || mode == 9) {
    const int kpad = rows / 24;
    const long long threads = (long long)mblocks * kpad * 64;
    const int grid = (int)(ceil_div64(threads, 256) < 16384 ? ceil_div64(threads, 256) : 16384);
    hipLaunchKernelGGL(weight_layout_w2d_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, mode, w, Co, Ci, wt, kpad, M,
                       mblocks);
    GSD_LAUNCH_CHECK("gsd_weight_layout (w2d)");
    return GSD_OK;
  }
  if (mode == 4 || mode == 5) {
    const int kpad = rows / 18;
    const long long threads = (long long)mblocks * kpad * 3 * 64;
    const int grid = (int)(ceil_div64(threads, 256) < 16384 ? ceil_div64(threads, 256) : 16384);
    hipLaunchKernelGGL(weight_layout_w43_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, mode, w, Co, Ci, wt, kpad, M,
                       mblocks);
    GSD_LAUNCH_CHECK("gsd_weight_layout (w43)");
    return GSD_OK;
  }
  const long long total = (long long)mblocks * rows * pitch;
  const int grid = (int)(ceil_div64(total, 256) < 8192 ? ceil_div64(total, 256) : 8192);
  hipLaunchKernelGGL(weight_layout_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, mode, w, Co, Ci, wt, rows, M, BM,
                     pitch, mblocks);
  GSD_LAUNCH_CHECK("gsd_weight_layout");
  return GSD_OK;
}

// All 2-D Winograd weight images of a pass in ONE launch (the fp32 twin of gsd_bf16_weight_images): the per-image launches of a
// step are 33 kernels of 5-20 us whose work is 0.7 GB of traffic.  Blocks are dealt to the jobs in proportion to their size.
namespace {
constexpr int WLB_MAX = 40;
struct WlBatch {
  const float* w[WLB_MAX];
  float* wt[WLB_MAX];
  int mode[WLB_MAX], Co[WLB_MAX], Ci[WLB_MAX], kpad[WLB_MAX], M[WLB_MAX], mblocks[WLB_MAX], first[WLB_MAX + 1];
  int n;
};
}  // namespace
__global__ __launch_bounds__(256) void weight_layout_w2d_batch_kernel(const WlBatch B) {
  int jb = 0;
  while (jb + 1 < B.n && (int)blockIdx.x >= B.first[jb + 1]) ++jb;
  const int mode = B.mode[jb], Co = B.Co[jb], Ci = B.Ci[jb], kpad = B.kpad[jb], M = B.M[jb];
  const float* __restrict__ w = B.w[jb];
  float* __restrict__ wt = B.wt[jb];
  const long long total = (long long)B.mblocks[jb] * kpad * 64;
  const int nb = B.first[jb + 1] - B.first[jb];
  for (long long e = (long long)(blockIdx.x - B.first[jb]) * blockDim.x + threadIdx.x; e < total; e += (long long)nb * blockDim.x) {
    const int cm = (int)(e & 63);
    const long long t = e >> 6;
    const int kch = (int)(t % kpad);
    const int mb = (int)(t / kpad);
    const int m = mb * 64 + cm;
    float g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) g[r][c] = 0.f;
    if (m < M && kch < (mode == 8 ? Ci : Co)) {
      const float* src = mode == 8 ? w + ((size_t)m * Ci + kch) * 9 : w + ((size_t)kch * Ci + m) * 9;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) g[r][c] = mode == 8 ? src[r * 3 + c] : src[(2 - r) * 3 + (2 - c)];
    }
    const int half = cm >> 5, mtl = (cm >> 4) & 1, l = cm & 15;
    float* o = wt + ((size_t)mb * kpad + (kch & ~3)) * (24 * 64) + (size_t)(kch & 3) * (12 * 128) + half * 64 + l * 4 + mtl;
#pragma unroll
    for (int fr = 0; fr < 4; ++fr) {
      float gr[3];
#pragma unroll
      for (int c = 0; c < 3; ++c)
        gr[c] = fr == 0 ? g[0][c] : fr == 3 ? g[2][c] : fr == 1 ? 0.5f * (g[0][c] + g[1][c] + g[2][c]) : 0.5f * (g[0][c] - g[1][c] + g[2][c]);
      const float g0 = gr[0], g1 = gr[1], g2 = gr[2];
      float u[6];
      u[0] = g0 * 0.25f;
      u[1] = -(g0 + g1 + g2) * (1.f / 6.f);
      u[2] = -(g0 - g1 + g2) * (1.f / 6.f);
      u[3] = g0 * (1.f / 24.f) + g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
      u[4] = g0 * (1.f / 24.f) - g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
      u[5] = g2;
#pragma unroll
      for (int fc = 0; fc < 6; ++fc) {
        const int f = fr * 6 + fc;
        o[(f >> 1) * 128 + (f & 1) * 2] = u[fc];
      }
    }
  }
}
extern "C" int gsd_weight_layout_batch(const gsd_wl_job* jobs, int n, void* stream) {
  GSD_REQUIRE(jobs != nullptr && n > 0, GSD_ERR_BAD_ARG, "gsd_weight_layout_batch: bad argument");
  WlBatch B;
  B.n = 0;
  int blocks = 0;
  auto flush = [&]() -> int {
    if (B.n == 0) return GSD_OK;
    B.first[B.n] = blocks;
    hipLaunchKernelGGL(weight_layout_w2d_batch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, B);
    GSD_LAUNCH_CHECK("gsd_weight_layout_batch");
    B.n = 0;
    blocks = 0;
    return GSD_OK;
  };
  for (int i = 0; i < n; ++i) {
    const gsd_wl_job& J = jobs[i];
    GSD_REQUIRE(J.w && J.wt && J.Co > 0 && J.Ci > 0 && J.mode >= 0 && J.mode <= 9, GSD_ERR_BAD_ARG, "gsd_weight_layout_batch: bad job %d", i);
    if (J.mode != 8 && J.mode != 9) {   // the other layouts keep their own launches
      if (int e = gsd_weight_layout(J.mode, J.w, J.Co, J.Ci, J.wt, stream)) return e;
      continue;
    }
    int rows, M, BM, pitch, mblocks;
    layout_dims(J.mode, J.Co, J.Ci, &rows, &M, &BM, &pitch, &mblocks);
    const int kpad = rows / 24;
    const long long threads = (long long)mblocks * kpad * 64;
    int nb = (int)(ceil_div64(threads, 1024) < 2048 ? ceil_div64(threads, 1024) : 2048);   // four elements per thread
    if (nb < 1) nb = 1;
    if (B.n == WLB_MAX) {
      if (int e = flush()) return e;
    }
    const int k = B.n++;
    B.w[k] = J.w; B.wt[k] = J.wt; B.mode[k] = J.mode; B.Co[k] = J.Co; B.Ci[k] = J.Ci; B.kpad[k] = kpad; B.M[k] = M; B.mblocks[k] = mblocks;
    B.first[k] = blocks;
    blocks += nb;
  }
  return flush();
}

// ---------------------------------------------------------------------------------------------
// column sums of a [rows][ncols] fp32 matrix into fp64 (two ordered stages)
// ---------------------------------------------------------------------------------------------
constexpr int RG = 64;  // row groups of stage 1 (part of the workspace contract: callers allocate (1+RG) x columns doubles)
constexpr int CS_LANES = 16;   // row lanes per block: 64 columns x 16 rows in flight, four independent partial sums each
// blockIdx.z = column range `half` (the sum | sum-of-squares halves of a conv partial row are `half_off` apart); a few
// ten thousand partial rows of 64..1024 columns: the grid is (columns/64, 64, halves) blocks of 1024 threads.
__global__ __launch_bounds__(64 * CS_LANES) void colsum_stage1(const float* __restrict__ part, int rows, int ld, int ncols,
                                                               int half_off, double* __restrict__ tmp) {
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int g = blockIdx.y;
  part += (size_t)blockIdx.z * half_off;
  tmp += (size_t)blockIdx.z * RG * ncols;
  const int per = (rows + RG - 1) / RG;
  const int rb = g * per, re = min(rb + per, rows);
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (col < ncols) {
    int r = rb + rl;
    for (; r + 3 * CS_LANES < re; r += 4 * CS_LANES) {   // four loads in flight per thread
      const float a = part[(size_t)r * ld + col], b = part[(size_t)(r + CS_LANES) * ld + col];
      const float c = part[(size_t)(r + 2 * CS_LANES) * ld + col], d = part[(size_t)(r + 3 * CS_LANES) * ld + col];
      s0 += (double)a; s1 += (double)b; s2 += (double)c; s3 += (double)d;
    }
    for (; r < re; r += CS_LANES) s0 += (double)part[(size_t)r * ld + col];
  }
  __shared__ double red[CS_LANES][64];
  red[rl][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rl == 0 && col < ncols) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < CS_LANES; ++i) t += red[i][threadIdx.x];
    tmp[(size_t)g * ncols + col] = t;
  }
}
// out32 (optional): columns [c_begin, c_begin + c_count) of the FIRST half (blockIdx.y == 0) also leave as fp32
__global__ void colsum_stage2(const double* __restrict__ tmp, int ncols, double* __restrict__ sums, float* __restrict__ out32 = nullptr,
                              int c_begin = 0, int c_count = 0) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  tmp += (size_t)blockIdx.y * RG * ncols;
  sums += (size_t)blockIdx.y * ncols;
  if (col < ncols) {
    double s = 0.0;
    for (int g = 0; g < RG; ++g) s += tmp[(size_t)g * ncols + col];
    sums[col] = s;
    if (out32 != nullptr && blockIdx.y == 0 && col >= c_begin && col < c_begin + c_count) out32[col - c_begin] = (float)s;
  }
}

// sums layout: [0..C) sum, [C..2C) sum of squares.  tmp space lives right behind `sums`
// (caller allocates (1+RG)*2*C doubles for `sums`).
extern "C" int gsd_bn_reduce_partials(const float* partials, int rows, int Mpad, int C, double* sums, void* stream) {
  GSD_REQUIRE(partials && sums && rows > 0 && C > 0 && Mpad >= C, GSD_ERR_BAD_ARG, "gsd_bn_reduce_partials: bad argument");
  double* tmp = sums + 2 * C;
  // the two halves (sum | sumsq) are Mpad apart in a partial row: two column ranges of one launch
  hipLaunchKernelGGL(colsum_stage1, dim3(ceil_div(C, 64), RG, 2), dim3(64 * CS_LANES), 0, (hipStream_t)stream, partials, rows,
                     2 * Mpad, C, Mpad, tmp);
  GSD_LAUNCH_CHECK("gsd_bn_reduce_partials stage1");
  hipLaunchKernelGGL(colsum_stage2, dim3(ceil_div(C, 256), 2), dim3(256), 0, (hipStream_t)stream, tmp, C, sums);
  GSD_LAUNCH_CHECK("gsd_bn_reduce_partials stage2");
  return GSD_OK;
}

// Per-channel sums of what a conv launch stored (the first halves of its partial rows), channels [c_begin, c_begin + c_count),
// as fp32 -- the ConvT bias gradient from the statistics epilogue of the dX launch that writes the up-sampled tensor's gradient.
extern "C" int gsd_partials_channel_sums(const float* partials, int rows, int Mpad, int C, int c_begin, int c_count, float* out,
                                         double* sums, void* stream) {
  GSD_REQUIRE(partials && sums && out && rows > 0 && C > 0 && Mpad >= C && c_begin >= 0 && c_count > 0 && c_begin + c_count <= C,
              GSD_ERR_BAD_ARG, "gsd_partials_channel_sums: bad argument");
  double* tmp = sums + 2 * C;
  hipLaunchKernelGGL(colsum_stage1, dim3(ceil_div(C, 64), RG, 1), dim3(64 * CS_LANES), 0, (hipStream_t)stream, partials, rows,
                     2 * Mpad, C, Mpad, tmp);
  GSD_LAUNCH_CHECK("gsd_partials_channel_sums stage1");
  hipLaunchKernelGGL(colsum_stage2, dim3(ceil_div(C, 256), 1), dim3(256), 0, (hipStream_t)stream, tmp, C, sums, out, c_begin, c_count);
  GSD_LAUNCH_CHECK("gsd_partials_channel_sums stage2");
  return GSD_OK;
}

// BatchNorm2d.num_batches_tracked += 1 for every layer of a train-mode forward: one launch for up to 64 int64 counters
struct counter_ptrs { long long* p[64]; };
__global__ void add_counters_kernel(counter_ptrs c, int n, long long delta) {
  const int i = threadIdx.x;
  if (i < n) *c.p[i] += delta;
}
extern "C" int gsd_add_counters(int64_t* const* counters, int n, int64_t delta, void* stream) {
  GSD_REQUIRE(counters && n > 0, GSD_ERR_BAD_ARG, "gsd_add_counters: bad argument");
  for (int base = 0; base < n; base += 64) {
    counter_ptrs c;
    const int m = n - base < 64 ? n - base : 64;
    for (int i = 0; i < m; ++i) {
      GSD_REQUIRE(counters[base + i] != nullptr, GSD_ERR_BAD_ARG, "gsd_add_counters: null counter");
      c.p[i] = (long long*)counters[base + i];
    }
    hipLaunchKernelGGL(add_counters_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, c, m, (long long)delta);
    GSD_LAUNCH_CHECK("gsd_add_counters");
  }
  return GSD_OK;
}

// Non-finite guard (gsd_guard, include/gsd.h): a batch statistic that is NaN/Inf never reaches the running statistics (a
// diverged or NaN-fed step would otherwise poison eval mode for good: every consumer turns a NaN activation into 0 through
// max(., 0), so the loss can stay finite), and the step is marked so that gsd_adam_ema can skip it.
// Returns whether the running statistics may be updated: not with non-finite values, and not once an EARLIER layer of this
// step has raised the guard (behind a NaN layer the activations are all 0 -- finite, but not statistics worth keeping).
__device__ __forceinline__ bool bn_stats_finite(double mu, double var, int* guard_words, int tick) {
  const bool finite = isfinite(mu) && isfinite(var);
  if (guard_words == nullptr) return finite;
  if (!finite) guard_words[0] = tick;   // benign race: every writer stores the same tick
  return finite && guard_words[0] != tick;
}
__global__ void bn_finalize_kernel(const double* __restrict__ sums, int C, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float momentum, float* running_mean,
                                   float* running_var, float* mean, float* invstd, float* scale, float* shift,
                                   int* guard_words, int tick) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mu = sums[c] / count;
  double var = sums[C + c] / count - mu * mu;  // biased (normalisation) variance
  const bool finite = bn_stats_finite(mu, var, guard_words, tick);
  if (var < 0.0) var = 0.0;
  const double is = 1.0 / sqrt(var + (double)eps);
  const float sc = (float)((double)gamma[c] * is);
  mean[c] = (float)mu;
  invstd[c] = (float)is;
  scale[c] = sc;
  shift[c] = (float)((double)beta[c] - mu * (double)gamma[c] * is);
  if (running_mean != nullptr && finite) {
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mu);
    running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unb);
  }
}
// One-launch forms: a block owns 16 channels; its 64 row lanes (4 per wave x 16 waves) sum the partial rows in fp64 in a
// fixed order (strided rows -> xor-shuffle inside the wave -> wave order in LDS), then 16 threads finalise.
// `sums` still receives the per-channel totals (SyncBN and the tests read them).
constexpr int RF_CH = 16, RF_LANES = 64;
template <int NV>
__device__ __forceinline__ bool rf_block_sums(const float* __restrict__ part, int rows, int ld, const int (&off)[NV], int C,
                                              double (&v)[NV]) {
  const int c = blockIdx.x * RF_CH + (threadIdx.x & (RF_CH - 1)), rl = threadIdx.x / RF_CH;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = 0.0;
  if (c < C) {
    int r = rl;
    for (; r + 3 * RF_LANES < rows; r += 4 * RF_LANES) {   // four rows' loads in flight (thousands of rows from the stand-alone reduce kernels); same order of additions
      float f[4][NV];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < NV; ++i) f[u][i] = off[i] >= 0 ? part[(size_t)(r + u * RF_LANES) * ld + off[i] + c] : 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < NV; ++i)
          if (off[i] >= 0) v[i] += (double)f[u][i];
    }
    for (; r < rows; r += RF_LANES) {
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (off[i] >= 0) v[i] += (double)part[(size_t)r * ld + off[i] + c];
    }
  }
  __shared__ double red[NV][RF_LANES / 4][RF_CH];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] += __shfl_xor(v[i], 16);
    v[i] += __shfl_xor(v[i], 32);
    if ((threadIdx.x & 63) < RF_CH) red[i][threadIdx.x >> 6][threadIdx.x & 63] = v[i];
  }
  __syncthreads();
  if (threadIdx.x >= RF_CH || c >= C) return false;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double t = 0.0;
    for (int w = 0; w < RF_LANES / 4; ++w) t += red[i][w][threadIdx.x];
    v[i] = t;
  }
  return true;
}

__global__ __launch_bounds__(1024) void bn_reduce_finalize_kernel(const float* __restrict__ part, int rows, int ld, int off2, int C,
                                                                 double* __restrict__ sums, double count, const float* gamma,
                                                                 const float* beta, float eps, float momentum, float* running_mean,
                                                                 float* running_var, float* mean, float* invstd, float* scale,
                                                                 float* shift, int* guard_words, int tick) {
  const int off[2] = {0, off2};
  double v[2];
  if (!rf_block_sums<2>(part, rows, ld, off, C, v)) return;
  const int c = blockIdx.x * RF_CH + threadIdx.x;
  const double s = v[0], q = v[1];
  sums[c] = s;
  sums[C + c] = q;
  const double mu = s / count;
  double var = q / count - mu * mu;
  const bool finite = bn_stats_finite(mu, var, guard_words, tick);
  if (var < 0.0) var = 0.0;
  const double is = 1.0 / sqrt(var + (double)eps);
  mean[c] = (float)mu;
  invstd[c] = (float)is;
  scale[c] = (float)((double)gamma[c] * is);
  shift[c] = (float)((double)beta[c] - mu * (double)gamma[c] * is);
  if (running_mean != nullptr && finite) {
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mu);
    running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unb);
  }
}

extern "C" int gsd_bn_reduce_finalize(const float* partials, int rows, int Mpad, int C, double* sums, double count,
                                      const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                      float* running_var, float* mean, float* invstd, float* scale, float* shift,
                                      const gsd_guard* guard, void* stream) {
  GSD_REQUIRE(partials && sums && gamma && beta && mean && invstd && scale && shift && rows > 0 && C > 0 && Mpad >= C && count > 0,
              GSD_ERR_BAD_ARG, "gsd_bn_reduce_finalize: bad argument");
  GSD_REQUIRE((running_mean == nullptr) == (running_var == nullptr), GSD_ERR_BAD_ARG,
              "gsd_bn_reduce_finalize: running stats must come together");
  hipLaunchKernelGGL(bn_reduce_finalize_kernel, dim3(ceil_div(C, RF_CH)), dim3(RF_CH * RF_LANES), 0, (hipStream_t)stream, partials, rows, 2 * Mpad,
                     Mpad, C, sums, count, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift,
                     guard ? guard->words : nullptr, guard ? guard->tick : 0);
  GSD_LAUNCH_CHECK("gsd_bn_reduce_finalize");
  return GSD_OK;
}

__global__ __launch_bounds__(1024) void bn_bwd_reduce_finalize_kernel(const float* __restrict__ part, int rows, int ld, int off2,
                                                                     int off3, int C, double* __restrict__ sums, double count,
                                                                     float* dgamma, float* dbeta, float* dwout, float* c1,
                                                                     float* c2) {
  const int off[3] = {0, off2, off3};
  double v[3];
  if (!rf_block_sums<3>(part, rows, ld, off, C, v)) return;
  const int c = blockIdx.x * RF_CH + threadIdx.x;
  sums[c] = v[0];
  sums[C + c] = v[1];
  sums[2 * C + c] = v[2];
  dbeta[c] = (float)v[0];
  dgamma[c] = (float)v[1];
  if (dwout != nullptr) dwout[c] = (float)v[2];
  c1[c] = (float)(v[0] / count);
  c2[c] = (float)(v[1] / count);
}

extern "C" int gsd_bn_bwd_reduce_finalize(const float* partials, int rows, int layout_mpad, int C, double* sums, double count,
                                          float* dgamma, float* dbeta, float* dwout, float* c1, float* c2, void* stream) {
  GSD_REQUIRE(partials && sums && dgamma && dbeta && c1 && c2 && rows > 0 && C > 0 && count > 0, GSD_ERR_BAD_ARG,
              "gsd_bn_bwd_reduce_finalize: bad argument");
  GSD_REQUIRE(layout_mpad == 0 || (layout_mpad >= C && dwout == nullptr), GSD_ERR_BAD_ARG,
              "gsd_bn_bwd_reduce_finalize: the conv-epilogue layout has no third column block");
  // layout_mpad == 0: rows of [sum dz | sum dz*xhat | third] (3*C) from the stand-alone reduce kernels;
  // layout_mpad  > 0: rows of 2*mpad from a dX epilogue (gsd_conv3x3_dgrad_bnrelu / gsd_bf16_bnbwd)
  const int ld = layout_mpad > 0 ? 2 * layout_mpad : 3 * C, off2 = layout_mpad > 0 ? layout_mpad : C;
  hipLaunchKernelGGL(bn_bwd_reduce_finalize_kernel, dim3(ceil_div(C, RF_CH)), dim3(RF_CH * RF_LANES), 0, (hipStream_t)stream, partials, rows, ld,
                     off2, layout_mpad > 0 ? -1 : 2 * C, C, sums, count, dgamma, dbeta, dwout, c1, c2);
  GSD_LAUNCH_CHECK("gsd_bn_bwd_reduce_finalize");
  return GSD_OK;
}

extern "C" int gsd_bn_finalize(const double* sums, int C, double count, const float* gamma, const float* beta, float eps,
                               float momentum, float* running_mean, float* running_var, float* mean, float* invstd,
                               float* scale, float* shift, const gsd_guard* guard, void* stream) {
  GSD_REQUIRE(sums && gamma && beta && mean && invstd && scale && shift && C > 0 && count > 0, GSD_ERR_BAD_ARG,
              "gsd_bn_finalize: bad argument");
  GSD_REQUIRE((running_mean == nullptr) == (running_var == nullptr), GSD_ERR_BAD_ARG,
              "gsd_bn_finalize: running stats must come together");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, sums, C, count, gamma,
                     beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift,
                     guard ? guard->words : nullptr, guard ? guard->tick : 0);
  GSD_LAUNCH_CHECK("gsd_bn_finalize");
  return GSD_OK;
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                      int C, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] / sqrtf(rv[c] + eps);
  scale[c] = sc;
  shift[c] = beta[c] - rm[c] * sc;
}
extern "C" int gsd_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, int C, float* scale, float* shift, void* stream) {
  GSD_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0, GSD_ERR_BAD_ARG,
              "gsd_bn_eval_coeffs: bad argument");
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                     running_mean, running_var, eps, C, scale, shift);
  GSD_LAUNCH_CHECK("gsd_bn_eval_coeffs");
  return GSD_OK;
}

// ---------------------------------------------------------------------------------------------
// BatchNorm + ReLU (+ max-pool / 1x1 output conv) backward, pass 1
// ---------------------------------------------------------------------------------------------
constexpr int BWD_CHUNK = 8192;  // elements of one (n, c) plane handled by one block

struct BnBwdParams {
  const float* raw;
  const float* scale;
  const float* shift;
  const float* mean;
  const float* invstd;
  SrcD da;
  const float* dpool;
  const float* dout;
  const float* wout;
  int K;
  float* dz;
  float* partials;
  int N, C, H, W, chunks;
};

template <int MODE>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const BnBwdParams P) {
  // grid: (chunks, C, N)
  const int chunk = blockIdx.x, c = blockIdx.y, n = blockIdx.z;
  const int HW = P.H * P.W;
  const size_t plane = ((size_t)n * P.C + c) * HW;
  const float sc = P.scale[c], sh = P.shift[c], mu = P.mean[c], is = P.invstd[c];
  const int Hp = P.H >> 1, Wp = P.W >> 1;
  float s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const int e_end = min((chunk + 1) * BWD_CHUNK, HW);
  for (int e = chunk * BWD_CHUNK + threadIdx.x; e < e_end; e += 256) {
    const float x = P.raw[plane + e];
    const float y = fmaf(x, sc, sh);
    float g;
    if constexpr (MODE == 2) {
      g = 0.f;
      for (int k = 0; k < P.K; ++k) {
        const float d = P.dout[((size_t)n * P.K + k) * HW + e];
        g = fmaf(d, P.wout[(size_t)k * P.C + c], g);
        if (k == 0) s3 = fmaf(d, fmaxf(y, 0.f), s3);  // dW_out[0][c] (all of dW_out when K == 1; K > 1: gsd_conv1x1_out_wgrad)
      }
    } else {
      g = 0.f;
      if (P.da.p != nullptr) {
        const int h = e / P.W, w = e - h * P.W;
        g = P.da.p[(size_t)n * P.da.ns + (size_t)c * P.da.cs + (size_t)h * P.da.W + w];
      }
      if constexpr (MODE == 1) {
        const int h = e / P.W, w = e - h * P.W;
        const int hp = h >> 1, wp = w >> 1;
        if (hp < Hp && wp < Wp) {
          // recompute the 2x2 arg-max of relu(bn(raw)); first maximum in (0,0),(0,1),(1,0),(1,1) order wins
          const float* wbase = P.raw + plane + (size_t)(2 * hp) * P.W + 2 * wp;
          float best = fmaxf(fmaf(wbase[0], sc, sh), 0.f);
          int bi = 0;
          float v = fmaxf(fmaf(wbase[1], sc, sh), 0.f);
          if (v > best) { best = v; bi = 1; }
          v = fmaxf(fmaf(wbase[P.W], sc, sh), 0.f);
          if (v > best) { best = v; bi = 2; }
          v = fmaxf(fmaf(wbase[P.W + 1], sc, sh), 0.f);
          if (v > best) { best = v; bi = 3; }
          if (bi == ((h & 1) << 1 | (w & 1))) g += P.dpool[(((size_t)n * P.C + c) * Hp + hp) * Wp + wp];
        }
      }
    }
    const float dzv = y > 0.f ? g : 0.f;
    P.dz[plane + e] = dzv;
    s1 += dzv;
    s2 = fmaf(dzv, (x - mu) * is, s2);
  }
  __shared__ float red[3][4];
  s1 = wave_sum_f(s1);
  s2 = wave_sum_f(s2);
  s3 = wave_sum_f(s3);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s1;
    red[1][threadIdx.x >> 6] = s2;
    red[2][threadIdx.x >> 6] = s3;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int row = n * P.chunks + chunk;
    P.partials[(size_t)row * 3 * P.C + threadIdx.x * P.C + c] =
        red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
  }
}

// Modes 0 and 2 with 16-byte accesses (planes of a multiple of 4 elements, 16-byte aligned operands): a thread owns four
// consecutive elements.  Same sums as the scalar kernel up to the order of addition.
template <int MODE>
__global__ __launch_bounds__(256) void bn_bwd_reduce_vec_kernel(const BnBwdParams P) {
  const int chunk = blockIdx.x, c = blockIdx.y, n = blockIdx.z;
  const int HW = P.H * P.W;
  const size_t plane = ((size_t)n * P.C + c) * HW;
  const float sc = P.scale[c], sh = P.shift[c], mu = P.mean[c], is = P.invstd[c];
  const float wo = MODE == 2 ? P.wout[c] : 0.f;
  const float* gsrc = MODE == 2 ? P.dout + (size_t)n * HW : P.da.p + (size_t)n * P.da.ns + (size_t)c * P.da.cs;
  float s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const int e_end = min((chunk + 1) * BWD_CHUNK, HW);
  for (int e = chunk * BWD_CHUNK + threadIdx.x * 4; e < e_end; e += 1024) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(P.raw + plane + e);
    const f32x4 d = *reinterpret_cast<const f32x4*>(gsrc + e);
    f32x4 dz;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float y = fmaf(x[i], sc, sh);
      float g = d[i];
      if (MODE == 2) {
        s3 = fmaf(d[i], fmaxf(y, 0.f), s3);   // dW_out[0][c]
        g = d[i] * wo;
      }
      dz[i] = y > 0.f ? g : 0.f;
      s1 += dz[i];
      s2 = fmaf(dz[i], (x[i] - mu) * is, s2);
    }
    *reinterpret_cast<f32x4*>(P.dz + plane + e) = dz;
  }
  __shared__ float red[3][4];
  s1 = wave_sum_f(s1);
  s2 = wave_sum_f(s2);
  s3 = wave_sum_f(s3);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s1;
    red[1][threadIdx.x >> 6] = s2;
    red[2][threadIdx.x >> 6] = s3;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int row = n * P.chunks + chunk;
    P.partials[(size_t)row * 3 * P.C + threadIdx.x * P.C + c] =
        red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
  }
}

// Mode 1 (gradient = da + the max-pool backward of dpool), one thread per 2x2 POOLING WINDOW: its four raw values are read once
// (two 8-byte loads) and serve both the arg-max and the four dz -- the element-per-thread kernel re-read the window for every
// element (3.7 TB/s).  Windows cut by an odd H / W keep the elements that exist and get no pooled gradient (floor mode).
constexpr int BWD_WCHUNK = BWD_CHUNK / 4;   // windows per block
__global__ __launch_bounds__(256) void bn_bwd_reduce_pool_kernel(const BnBwdParams P) {
  typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
  const int chunk = blockIdx.x, c = blockIdx.y, n = blockIdx.z;
  const int HW = P.H * P.W;
  const size_t plane = ((size_t)n * P.C + c) * HW;
  const float sc = P.scale[c], sh = P.shift[c], mu = P.mean[c], is = P.invstd[c];
  const int Hp = P.H >> 1, Wp = P.W >> 1, Hc = (P.H + 1) >> 1, Wc = (P.W + 1) >> 1;
  const float* xr = P.raw + plane;
  const float* ga = P.da.p != nullptr ? P.da.p + (size_t)n * P.da.ns + (size_t)c * P.da.cs : nullptr;
  const float* dp = P.dpool + ((size_t)n * P.C + c) * Hp * Wp;
  float* dzp = P.dz + plane;
  float s1 = 0.f, s2 = 0.f;
  const int q_end = min((chunk + 1) * BWD_WCHUNK, Hc * Wc);
  for (int q = chunk * BWD_WCHUNK + threadIdx.x; q < q_end; q += 256) {
    const int hp = q / Wc, wp = q - hp * Wc;
    const int o0 = 2 * hp * P.W + 2 * wp, o1 = o0 + P.W;
    const bool col1 = 2 * wp + 1 < P.W, row1 = 2 * hp + 1 < P.H;
    float x[4] = {0.f, 0.f, 0.f, 0.f}, g[4] = {0.f, 0.f, 0.f, 0.f};
    if (col1) {
      const f32x2u t = *reinterpret_cast<const f32x2u*>(xr + o0);
      x[0] = t[0], x[1] = t[1];
      if (ga != nullptr) { const f32x2u u = *reinterpret_cast<const f32x2u*>(ga + o0); g[0] = u[0], g[1] = u[1]; }
      if (row1) {
        const f32x2u t1 = *reinterpret_cast<const f32x2u*>(xr + o1);
        x[2] = t1[0], x[3] = t1[1];
        if (ga != nullptr) { const f32x2u u = *reinterpret_cast<const f32x2u*>(ga + o1); g[2] = u[0], g[3] = u[1]; }
      }
    } else {
      x[0] = xr[o0];
      if (ga != nullptr) g[0] = ga[o0];
      if (row1) {
        x[2] = xr[o1];
        if (ga != nullptr) g[2] = ga[o1];
      }
    }
    float y[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = fmaf(x[i], sc, sh);
    if (col1 && row1) {   // a whole window: the first maximum of relu(bn(raw)) in (0,0),(0,1),(1,0),(1,1) order takes dpool
      float best = fmaxf(y[0], 0.f);
      int bi = 0;
#pragma unroll
      for (int i = 1; i < 4; ++i) {
        const float v = fmaxf(y[i], 0.f);
        if (v > best) { best = v; bi = i; }
      }
      const float dpv = dp[hp * Wp + wp];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (bi == i) g[i] += dpv;
    }
    float dz[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ex = (i & 1 ? col1 : true) && (i & 2 ? row1 : true);
      dz[i] = (ex && y[i] > 0.f) ? g[i] : 0.f;
      s1 += dz[i];
      s2 = fmaf(dz[i], (x[i] - mu) * is, s2);
    }
    if (col1) {
      *reinterpret_cast<f32x2u*>(dzp + o0) = f32x2u{dz[0], dz[1]};
      if (row1) *reinterpret_cast<f32x2u*>(dzp + o1) = f32x2u{dz[2], dz[3]};
    } else {
      dzp[o0] = dz[0];
      if (row1) dzp[o1] = dz[2];
    }
  }
  __shared__ float red[2][4];
  s1 = wave_sum_f(s1);
  s2 = wave_sum_f(s2);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s1;
    red[1][threadIdx.x >> 6] = s2;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int row = n * P.chunks + chunk;
    P.partials[(size_t)row * 3 * P.C + threadIdx.x * P.C + c] =
        threadIdx.x < 2 ? red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3] : 0.f;
  }
}

// partial rows of one launch: per image, the blocks of the window-per-thread form (>= those of the element forms, whose
// surplus blocks write zeros)
static int bwd_chunks(int H, int W) { return ceil_div(((H + 1) / 2) * ((W + 1) / 2), BWD_WCHUNK); }
extern "C" int gsd_bn_bwd_partial_rows(int N, int C, int H, int W) {
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
  return N * bwd_chunks(H, W);
}

extern "C" int gsd_bn_bwd_reduce(int mode, const float* raw, const float* scale, const float* shift, const float* mean,
                                 const float* invstd, const gsd_src* da, const float* dpool, const float* dout,
                                 const float* wout, int K, float* dz, float* partials, int N, int C, int H, int W,
                                 void* stream) {
  GSD_REQUIRE(raw && scale && shift && mean && invstd && dz && partials, GSD_ERR_BAD_ARG, "gsd_bn_bwd_reduce: null argument");
  GSD_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && mode >= 0 && mode <= 2, GSD_ERR_BAD_ARG, "gsd_bn_bwd_reduce: bad sizes");
  GSD_REQUIRE(N <= 65535 && C <= 65535, GSD_ERR_UNSUPPORTED, "gsd_bn_bwd_reduce: N, C must be <= 65535");
  BnBwdParams P;
  P.raw = raw; P.scale = scale; P.shift = shift; P.mean = mean; P.invstd = invstd;
  P.da = null_srcd();
  if (mode != 2 && da != nullptr && da->ptr != nullptr) {
    GSD_REQUIRE(da->scale == nullptr && da->relu == 0 && da->off_h == 0 && da->off_w == 0 && da->H == H && da->W == W &&
                    da->C >= C,
                GSD_ERR_BAD_ARG, "gsd_bn_bwd_reduce: da must be a plain (>=C,H,W) tensor");
    if (int e = gsd_require_rows_contiguous(*da, "gsd_bn_bwd_reduce da")) return e;
    P.da = to_srcd(*da);
  }
  if (mode == 0) GSD_REQUIRE(P.da.p != nullptr, GSD_ERR_BAD_ARG, "gsd_bn_bwd_reduce: mode PLAIN needs da");
  if (mode == 1) GSD_REQUIRE(dpool != nullptr, GSD_ERR_BAD_ARG, "gsd_bn_bwd_reduce: mode POOL needs dpool");
  if (mode == 2) {
    GSD_REQUIRE(dout != nullptr && wout != nullptr, GSD_ERR_BAD_ARG, "gsd_bn_bwd_reduce: mode OUTC needs dout, wout");
    GSD_REQUIRE(K >= 1 && K <= 8, GSD_ERR_UNSUPPORTED, "gsd_bn_bwd_reduce: backward of the output conv supports 1 <= n_classes <= 8 (got %d)", K);
  }
  P.dpool = dpool; P.dout = dout; P.wout = wout; P.K = K;
  P.dz = dz; P.partials = partials;
  P.N = N; P.C = C; P.H = H; P.W = W;
  P.chunks = bwd_chunks(H, W);
  dim3 grid(P.chunks, C, N);
  const bool scalar = gsd_env_int("GSD_BN_BWD_SCALAR", 0) != 0;   // the element-per-thread kernels (A/B, tests)
  const float* gsrc = mode == 2 ? dout : P.da.p;
  const bool vec = !scalar && (H * W) % 4 == 0 && (((uintptr_t)raw | (uintptr_t)dz | (uintptr_t)gsrc) & 15) == 0 &&
                   (mode == 2 || (P.da.ns % 4 == 0 && P.da.cs % 4 == 0));
  if (mode == 1 && !scalar) hipLaunchKernelGGL(bn_bwd_reduce_pool_kernel, grid, dim3(256), 0, (hipStream_t)stream, P);
  else if (mode == 0 && vec) hipLaunchKernelGGL((bn_bwd_reduce_vec_kernel<0>), grid, dim3(256), 0, (hipStream_t)stream, P);
  else if (mode == 2 && vec && K == 1) hipLaunchKernelGGL((bn_bwd_reduce_vec_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, P);
  else if (mode == 0) hipLaunchKernelGGL((bn_bwd_reduce_kernel<0>), grid, dim3(256), 0, (hipStream_t)stream, P);
  else if (mode == 1) hipLaunchKernelGGL((bn_bwd_reduce_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, P);
  else hipLaunchKernelGGL((bn_bwd_reduce_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, P);
  GSD_LAUNCH_CHECK("gsd_bn_bwd_reduce");
  return GSD_OK;
}

// sums: 3*C doubles followed by RG*3*C doubles of scratch
extern "C" int gsd_bn_bwd_reduce_partials(const float* partials, int rows, int C, double* sums, void* stream) {
  GSD_REQUIRE(partials && sums && rows > 0 && C > 0, GSD_ERR_BAD_ARG, "gsd_bn_bwd_reduce_partials: bad argument");
  double* tmp = sums + 3 * C;
  hipLaunchKernelGGL(colsum_stage1, dim3(ceil_div(3 * C, 64), RG, 1), dim3(64 * CS_LANES), 0, (hipStream_t)stream, partials, rows,
                     3 * C, 3 * C, 0, tmp);
  GSD_LAUNCH_CHECK("gsd_bn_bwd_reduce_partials stage1");
  hipLaunchKernelGGL(colsum_stage2, dim3(ceil_div(3 * C, 256), 1), dim3(256), 0, (hipStream_t)stream, tmp, 3 * C, sums);
  GSD_LAUNCH_CHECK("gsd_bn_bwd_reduce_partials stage2");
  return GSD_OK;
}

__global__ void bn_bwd_finalize_kernel(const double* sl, const double* sg, int C, double count, float* dgamma,
                                       float* dbeta, float* dwout, float* c1, float* c2) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  dbeta[c] = (float)sl[c];
  dgamma[c] = (float)sl[C + c];
  if (dwout != nullptr) dwout[c] = (float)sl[2 * C + c];
  c1[c] = (float)(sg[c] / count);
  c2[c] = (float)(sg[C + c] / count);
}
extern "C" int gsd_bn_bwd_finalize(const double* sums_local, const double* sums_global, int C, double count,
                                   float* dgamma, float* dbeta, float* dwout, float* c1, float* c2, void* stream) {
  GSD_REQUIRE(sums_local && dgamma && dbeta && c1 && c2 && C > 0 && count > 0, GSD_ERR_BAD_ARG,
              "gsd_bn_bwd_finalize: bad argument");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, sums_local,
                     sums_global != nullptr ? sums_global : sums_local, C, count, dgamma, dbeta, dwout, c1, c2);
  GSD_LAUNCH_CHECK("gsd_bn_bwd_finalize");
  return GSD_OK;
}

template <bool VEC4>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(float* __restrict__ dz, const float* __restrict__ raw,
                                                           const float* scale, const float* mean, const float* invstd,
                                                           const float* c1, const float* c2, int C, int HW, int chunks) {
  const int chunk = blockIdx.x, c = blockIdx.y, n = blockIdx.z;
  const size_t plane = ((size_t)n * C + c) * HW;
  const float sc = scale[c], mu = mean[c], is = invstd[c], k1 = c1[c], k2 = c2[c];
  const int e_end = min((chunk + 1) * BWD_CHUNK, HW);
  if (VEC4) {   // HW % 4 == 0 and 16-byte aligned tensors: one 16-byte load / store per lane
    for (int e = chunk * BWD_CHUNK + threadIdx.x * 4; e < e_end; e += 1024) {
      const f32x4 r = *reinterpret_cast<const f32x4*>(raw + plane + e);
      f32x4 d = *reinterpret_cast<const f32x4*>(dz + plane + e);
#pragma unroll
      for (int i = 0; i < 4; ++i) d[i] = sc * (d[i] - k1 - (r[i] - mu) * is * k2);
      *reinterpret_cast<f32x4*>(dz + plane + e) = d;
    }
  } else {
    for (int e = chunk * BWD_CHUNK + threadIdx.x; e < e_end; e += 256) {
      const float xh = (raw[plane + e] - mu) * is;
      dz[plane + e] = sc * (dz[plane + e] - k1 - xh * k2);
    }
  }
}
// Out-of-place form into a PITCHED buffer (rows of `pitch` floats, pitch % 4 == 0, 16-byte aligned): the result is what the
// dW and dX kernels read next, and rows that start 16-byte aligned let them move it as aligned 16-byte LDS-DMA pieces (a
// quarter of the gather instructions).  Same traffic as the in-place pass.  A thread owns 4 consecutive columns of one row:
// four coalesced dword loads per input (the contiguous W = 427 rows are not 16-byte aligned), one 16-byte store; columns
// W .. pitch-1 are written 0 -- the padding value of a plain gradient operand.
__global__ __launch_bounds__(256) void bn_bwd_apply_pitched_kernel(const float* __restrict__ dz, const float* __restrict__ raw,
                                                                   const float* scale, const float* mean, const float* invstd,
                                                                   const float* c1, const float* c2, float* __restrict__ out,
                                                                   int C, int H, int W, int pitch) {
  const int c = blockIdx.y, n = blockIdx.z;
  const int q4 = pitch >> 2;                       // 16-byte pieces per output row
  const int total = H * q4;
  const size_t plane = ((size_t)n * C + c) * (size_t)H * W;
  float* const o = out + ((size_t)n * C + c) * (size_t)H * pitch;
  const float sc = scale[c], mu = mean[c], is = invstd[c], k1 = c1[c], k2 = c2[c];
  typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int h = e / q4, w = (e - h * q4) * 4;
    const size_t src = plane + (size_t)h * W + w;
    f32x4 d;
    if (w + 4 <= W) {   // one (unaligned) 16-byte load per input: the contiguous rows of W = 427 floats are not 16-byte aligned
      const f32x4 g = *reinterpret_cast<const f32x4u*>(dz + src), r = *reinterpret_cast<const f32x4u*>(raw + src);
#pragma unroll
      for (int i = 0; i < 4; ++i) d[i] = sc * (g[i] - k1 - (r[i] - mu) * is * k2);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = w + i < W;
        const float g = ok ? dz[src + i] : 0.f, r = ok ? raw[src + i] : mu;
        d[i] = ok ? sc * (g - k1 - (r - mu) * is * k2) : 0.f;
      }
    }
    *reinterpret_cast<f32x4*>(o + (size_t)h * pitch + w) = d;
  }
}
extern "C" int gsd_bn_bwd_apply(float* dz, const float* raw, const float* scale, const float* mean, const float* invstd,
                                const float* c1, const float* c2, int N, int C, int H, int W, float* out, int out_w_stride,
                                void* stream) {
  GSD_REQUIRE(dz && raw && scale && mean && invstd && c1 && c2 && N > 0 && C > 0 && H > 0 && W > 0, GSD_ERR_BAD_ARG,
              "gsd_bn_bwd_apply: bad argument");
  GSD_REQUIRE(N <= 65535 && C <= 65535, GSD_ERR_UNSUPPORTED, "gsd_bn_bwd_apply: N, C must be <= 65535");
  if (out != nullptr) {
    GSD_REQUIRE(out_w_stride >= W && out_w_stride % 4 == 0 && ((uintptr_t)out & 15) == 0, GSD_ERR_BAD_ARG,
                "gsd_bn_bwd_apply: the pitched destination needs a 16-byte aligned base and a row pitch %% 4 == 0 (got %d for W %d)",
                out_w_stride, W);
    const int total = H * (out_w_stride / 4);
    const int bx = ceil_div(total, 256) < 64 ? ceil_div(total, 256) : 64;
    hipLaunchKernelGGL(bn_bwd_apply_pitched_kernel, dim3(bx, C, N), dim3(256), 0, (hipStream_t)stream, dz, raw, scale, mean,
                       invstd, c1, c2, out, C, H, W, out_w_stride);
    GSD_LAUNCH_CHECK("gsd_bn_bwd_apply (pitched)");
    return GSD_OK;
  }
  const int chunks = ceil_div(H * W, BWD_CHUNK);
  const bool vec4 = (H * W) % 4 == 0 && (((uintptr_t)dz | (uintptr_t)raw) & 15) == 0;
  if (vec4)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(chunks, C, N), dim3(256), 0, (hipStream_t)stream, dz, raw, scale, mean,
                       invstd, c1, c2, C, H * W, chunks);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(chunks, C, N), dim3(256), 0, (hipStream_t)stream, dz, raw, scale, mean,
                       invstd, c1, c2, C, H * W, chunks);
  GSD_LAUNCH_CHECK("gsd_bn_bwd_apply");
  return GSD_OK;
}

// ---------------------------------------------------------------------------------------------
// MaxPool2d(2), floor mode, of relu(bn(raw))
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool2_kernel(const SrcD S, float* __restrict__ y, int C, int Hp, int Wp) {
  // grid: (ceil(Hp*Wp/256), C, N)
  const int c = blockIdx.y, n = blockIdx.z;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= Hp * Wp) return;
  const int hp = e / Wp, wp = e - hp * Wp;
  const float* b = S.p + (size_t)n * S.ns + (size_t)c * S.cs + (size_t)(2 * hp) * S.W + 2 * wp;
  float sc = 1.f, sh = 0.f;
  if (S.scale != nullptr) { sc = S.scale[c]; sh = S.shift[c]; }
  float v0 = fmaf(b[0], sc, sh), v1 = fmaf(b[1], sc, sh), v2 = fmaf(b[S.W], sc, sh), v3 = fmaf(b[S.W + 1], sc, sh);
  float m = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
  if (S.relu) m = fmaxf(m, 0.f);
  y[(((size_t)n * C + c) * Hp + hp) * Wp + wp] = m;
}
extern "C" int gsd_maxpool2(const gsd_src* src, float* y, int N, int C, int H, int W, void* stream) {
  GSD_REQUIRE(src && src->ptr && y && N > 0 && C > 0 && H > 1 && W > 1, GSD_ERR_BAD_ARG, "gsd_maxpool2: bad argument");
  GSD_REQUIRE(src->C == C && src->H == H && src->W == W && src->off_h == 0 && src->off_w == 0, GSD_ERR_BAD_ARG,
              "gsd_maxpool2: src must be the full (C,H,W) tensor");
  if (int e = gsd_require_rows_contiguous(*src, "gsd_maxpool2 src")) return e;
  GSD_REQUIRE(N <= 65535 && C <= 65535, GSD_ERR_UNSUPPORTED, "gsd_maxpool2: N, C must be <= 65535");
  const int Hp = H / 2, Wp = W / 2;
  hipLaunchKernelGGL(maxpool2_kernel, dim3(ceil_div(Hp * Wp, 256), C, N), dim3(256), 0, (hipStream_t)stream,
                     to_srcd(*src), y, C, Hp, Wp);
  GSD_LAUNCH_CHECK("gsd_maxpool2");
  return GSD_OK;
}

// ---------------------------------------------------------------------------------------------
// 1x1 output conv (+bias) of relu(bn(raw))
// ---------------------------------------------------------------------------------------------
constexpr int OUTC_MAXK = 8;
__global__ __launch_bounds__(256) void conv1x1_out_kernel(const SrcD S, const float* __restrict__ w,
                                                          const float* __restrict__ b, int C, int K,
                                                          float* __restrict__ out, int HW) {
  const int n = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  float acc[OUTC_MAXK];
#pragma unroll
  for (int k = 0; k < OUTC_MAXK; ++k) acc[k] = 0.f;
  const float* base = S.p + (size_t)n * S.ns + p;
  for (int c = 0; c < C; ++c) {
    float v = base[(size_t)c * S.cs];
    if (S.scale != nullptr) v = fmaf(v, S.scale[c], S.shift[c]);
    if (S.relu) v = fmaxf(v, 0.f);
#pragma unroll
    for (int k = 0; k < OUTC_MAXK; ++k)
      if (k < K) acc[k] = fmaf(w[k * C + c], v, acc[k]);
  }
#pragma unroll
  for (int k = 0; k < OUTC_MAXK; ++k)
    if (k < K) out[((size_t)n * K + k) * HW + p] = acc[k] + (b != nullptr ? b[k] : 0.f);
}
// four consecutive pixels per thread (16-byte loads and stores): H * W % 4 == 0 and 16-byte aligned operands
__global__ __launch_bounds__(256) void conv1x1_out_vec_kernel(const SrcD S, const float* __restrict__ w,
                                                              const float* __restrict__ b, int C, int K,
                                                              float* __restrict__ out, int HW) {
  const int n = blockIdx.y;
  const int p = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (p >= HW) return;
  f32x4 acc[OUTC_MAXK];
#pragma unroll
  for (int k = 0; k < OUTC_MAXK; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* base = S.p + (size_t)n * S.ns + p;
  for (int c = 0; c < C; ++c) {
    f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)c * S.cs);
    if (S.scale != nullptr) {
      const float sc = S.scale[c], sh = S.shift[c];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = fmaf(v[i], sc, sh);
    }
    if (S.relu) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
    }
#pragma unroll
    for (int k = 0; k < OUTC_MAXK; ++k)
      if (k < K) {
        const float wk = w[k * C + c];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[k][i] = fmaf(wk, v[i], acc[k][i]);
      }
  }
#pragma unroll
  for (int k = 0; k < OUTC_MAXK; ++k)
    if (k < K) {
      const float bk = b != nullptr ? b[k] : 0.f;
      *reinterpret_cast<f32x4*>(out + ((size_t)n * K + k) * HW + p) = f32x4{acc[k][0] + bk, acc[k][1] + bk, acc[k][2] + bk, acc[k][3] + bk};
    }
}
extern "C" int gsd_conv1x1_out(const gsd_src* src, const float* w, const float* b, int C, int K, float* out, int N, int H,
                               int W, void* stream) {
  GSD_REQUIRE(src && src->ptr && w && out && N > 0 && C > 0 && K > 0 && H > 0 && W > 0, GSD_ERR_BAD_ARG,
              "gsd_conv1x1_out: bad argument");
  GSD_REQUIRE(K <= OUTC_MAXK, GSD_ERR_UNSUPPORTED, "gsd_conv1x1_out: n_classes %d > %d", K, OUTC_MAXK);
  GSD_REQUIRE(src->C == C && src->H == H && src->W == W && src->off_h == 0 && src->off_w == 0 &&
                  src->c_stride == (int64_t)H * W && src->w_stride == W,
              GSD_ERR_BAD_ARG, "gsd_conv1x1_out: src must be the full contiguous (C,H,W) tensor");
  GSD_REQUIRE(N <= 65535, GSD_ERR_UNSUPPORTED, "gsd_conv1x1_out: N must be <= 65535");
  const bool vec = (H * W) % 4 == 0 && (((uintptr_t)src->ptr | (uintptr_t)out) & 15) == 0 && src->n_stride % 4 == 0;
  if (vec)
    hipLaunchKernelGGL(conv1x1_out_vec_kernel, dim3(ceil_div(H * W, 1024), N), dim3(256), 0, (hipStream_t)stream,
                       to_srcd(*src), w, b, C, K, out, H * W);
  else
    hipLaunchKernelGGL(conv1x1_out_kernel, dim3(ceil_div(H * W, 256), N), dim3(256), 0, (hipStream_t)stream, to_srcd(*src),
                       w, b, C, K, out, H * W);
  GSD_LAUNCH_CHECK("gsd_conv1x1_out");
  return GSD_OK;
}

// dW of the output conv for n_classes > 1 (the K == 1 case comes out of gsd_bn_bwd_reduce mode 2 as its third sum):
//   dw[k][c] = sum_{n,p} dout[n,k,p] * max(0, raw[n,c,p]*scale[c] + shift[c])
// One block per (pixel chunk, channel, image) leaves K partial sums; the column sums (fp64, then fp32) finish it.
constexpr int OUTW_CHUNK = 8192;
__global__ __launch_bounds__(256) void conv1x1_out_wgrad_kernel(const float* __restrict__ raw, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const float* __restrict__ dout,
                                                                int K, int C, int HW, int chunks, float* __restrict__ partials) {
  const int chunk = blockIdx.x, c = blockIdx.y, n = blockIdx.z;
  const float sc = scale[c], sh = shift[c];
  const float* x = raw + ((size_t)n * C + c) * HW;
  const float* d = dout + (size_t)n * K * HW;
  float s[OUTC_MAXK];
#pragma unroll
  for (int k = 0; k < OUTC_MAXK; ++k) s[k] = 0.f;
  const int e_end = min((chunk + 1) * OUTW_CHUNK, HW);
  for (int e = chunk * OUTW_CHUNK + threadIdx.x; e < e_end; e += 256) {
    const float a = fmaxf(fmaf(x[e], sc, sh), 0.f);
#pragma unroll
    for (int k = 0; k < OUTC_MAXK; ++k)
      if (k < K) s[k] = fmaf(d[(size_t)k * HW + e], a, s[k]);
  }
  __shared__ float red[OUTC_MAXK][4];
#pragma unroll
  for (int k = 0; k < OUTC_MAXK; ++k) {
    const float v = wave_sum_f(s[k]);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    const int row = n * chunks + chunk;
    partials[(size_t)row * K * C + (size_t)threadIdx.x * C + c] =
        red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
  }
}
extern "C" int gsd_conv1x1_out_wgrad_rows(int N, int H, int W) {
  return (N > 0 && H > 0 && W > 0) ? N * ceil_div(H * W, OUTW_CHUNK) : 0;
}
extern "C" int gsd_conv1x1_out_wgrad(const float* raw, const float* scale, const float* shift, const float* dout, int C, int K,
                                     float* dw, float* partials, double* sums, int N, int H, int W, void* stream) {
  GSD_REQUIRE(raw && scale && shift && dout && dw && partials && sums && N > 0 && C > 0 && K > 0 && H > 0 && W > 0, GSD_ERR_BAD_ARG,
              "gsd_conv1x1_out_wgrad: bad argument");
  GSD_REQUIRE(K <= OUTC_MAXK, GSD_ERR_UNSUPPORTED, "gsd_conv1x1_out_wgrad: n_classes %d > %d", K, OUTC_MAXK);
  GSD_REQUIRE(N <= 65535 && C <= 65535, GSD_ERR_UNSUPPORTED, "gsd_conv1x1_out_wgrad: N, C must be <= 65535");
  const int chunks = ceil_div(H * W, OUTW_CHUNK), rows = N * chunks, cols = K * C;
  hipLaunchKernelGGL(conv1x1_out_wgrad_kernel, dim3(chunks, C, N), dim3(256), 0, (hipStream_t)stream, raw, scale, shift, dout, K, C,
                     H * W, chunks, partials);
  GSD_LAUNCH_CHECK("gsd_conv1x1_out_wgrad");
  double* tmp = sums + cols;
  hipLaunchKernelGGL(colsum_stage1, dim3(ceil_div(cols, 64), RG, 1), dim3(64 * CS_LANES), 0, (hipStream_t)stream, partials, rows, cols,
                     cols, 0, tmp);
  GSD_LAUNCH_CHECK("gsd_conv1x1_out_wgrad stage1");
  hipLaunchKernelGGL(colsum_stage2, dim3(ceil_div(cols, 256), 1), dim3(256), 0, (hipStream_t)stream, tmp, cols, sums, dw, 0, cols);
  GSD_LAUNCH_CHECK("gsd_conv1x1_out_wgrad stage2");
  return GSD_OK;
}

// ---------------------------------------------------------------------------------------------
// loss (MSE / L1) forward + gradient
// ---------------------------------------------------------------------------------------------
constexpr int LOSS_BLOCKS = 1024;
template <int KIND>
__global__ __launch_bounds__(256) void loss_stage1(const float* __restrict__ o, const float* __restrict__ t,
                                                   long long numel, float gscale, float* __restrict__ grad,
                                                   float* __restrict__ ws) {
  double s = 0.0;
  const float inv = 1.0f / (float)numel;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < numel; i += (long long)gridDim.x * 256) {
    const float d = o[i] - t[i];
    if (KIND == 0) {
      s += (double)d * (double)d;
      if (grad != nullptr) grad[i] = 2.f * d * inv * gscale;
    } else {
      s += (double)fabsf(d);
      if (grad != nullptr) grad[i] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * inv * gscale;
    }
  }
  __shared__ double red[4];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) reinterpret_cast<double*>(ws)[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void loss_stage2(const float* ws, int nblocks, long long numel, float* loss_out, int* guard_words, int tick) {
  // single wave
  double s = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 64) s += reinterpret_cast<const double*>(ws)[i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) {
    const float loss = (float)(s / (double)numel);
    loss_out[0] = loss;
    if (guard_words != nullptr && !isfinite(loss)) guard_words[0] = tick;   // the reference's `pred_loss.isnan()` test, on the device
  }
}
extern "C" int gsd_loss_fwd_bwd(int kind, const float* o, const float* t, int64_t numel, float grad_scale, float* loss_out,
                                float* grad, float* workspace, const gsd_guard* guard, void* stream) {
  GSD_REQUIRE(o && t && loss_out && workspace && numel > 0 && (kind == 0 || kind == 1), GSD_ERR_BAD_ARG,
              "gsd_loss_fwd_bwd: bad argument");
  GSD_REQUIRE(((uintptr_t)workspace & 7) == 0, GSD_ERR_BAD_ARG, "gsd_loss_fwd_bwd: workspace must be 8-byte aligned");
  int blocks = (int)(ceil_div64(numel, 256) < LOSS_BLOCKS ? ceil_div64(numel, 256) : LOSS_BLOCKS);
  if (kind == 0)
    hipLaunchKernelGGL((loss_stage1<0>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, o, t, (long long)numel,
                       grad_scale, grad, workspace);
  else
    hipLaunchKernelGGL((loss_stage1<1>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, o, t, (long long)numel,
                       grad_scale, grad, workspace);
  GSD_LAUNCH_CHECK("gsd_loss_fwd_bwd stage1");
  hipLaunchKernelGGL(loss_stage2, dim3(1), dim3(64), 0, (hipStream_t)stream, workspace, blocks, (long long)numel, loss_out,
                     guard ? guard->words : nullptr, guard ? guard->tick : 0);
  GSD_LAUNCH_CHECK("gsd_loss_fwd_bwd stage2");
  return GSD_OK;
}

// ---------------------------------------------------------------------------------------------
// fused Adam (coupled L2) + EMA over a flat arena
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_ema_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v,
                                                       float* __restrict__ ema, long long numel, float lr_over_bc1,
                                                       float sqrt_bc2, float b1, float b2, float eps, float wd,
                                                       float one_minus_d, float gscale, int* guard_words, int tick) {
  if (guard_words != nullptr && guard_words[0] == tick) {   // this step saw a non-finite statistic or loss: skip it, count it
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&guard_words[1], 1);
    return;
  }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < numel; i += (long long)gridDim.x * 256) {
    float pv = p[i];
    const float gv = fmaf(wd, pv, g[i] * gscale);   // g += wd*p            (torch _single_tensor_adam)
    float mv = m[i];
    mv = mv + (gv - mv) * (1.f - b1);               // exp_avg.lerp_(g, 1-b1)
    const float vv = fmaf(1.f - b2, gv * gv, v[i] * b2);   // exp_avg_sq.mul_(b2).addcmul_(g,g,1-b2)
    const float denom = sqrtf(vv) / sqrt_bc2 + eps;    // sqrt(v)/sqrt(bc2) + eps
    pv = pv - lr_over_bc1 * (mv / denom);                  // p.addcdiv_(m, denom, -lr/bc1)
    p[i] = pv;
    m[i] = mv;
    v[i] = vv;
    if (ema != nullptr) {
      const float s = ema[i];
      ema[i] = s - one_minus_d * (s - pv);          // torch_ema: shadow.sub_((1-d)*(shadow-param))
    }
  }
}
// A skipped step leaves no trace in the BatchNorm running statistics: snapshot before the step, conditional restore behind it
__global__ void guard_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n, const int* guard_words,
                                  int tick) {
  if (guard_words != nullptr && guard_words[0] != tick) return;   // restore form: only when this step was marked bad
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = src[i];
}
extern "C" int gsd_guard_snapshot(const float* live, float* snapshot, int64_t n, void* stream) {
  GSD_REQUIRE(live && snapshot && n > 0, GSD_ERR_BAD_ARG, "gsd_guard_snapshot: bad argument");
  const int blocks = (int)(ceil_div64(n, 256) < 1024 ? ceil_div64(n, 256) : 1024);
  hipLaunchKernelGGL(guard_copy_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, live, snapshot, (long long)n,
                     (const int*)nullptr, 0);
  GSD_LAUNCH_CHECK("gsd_guard_snapshot");
  return GSD_OK;
}
extern "C" int gsd_guard_restore(const gsd_guard* guard, float* live, const float* snapshot, int64_t n, void* stream) {
  GSD_REQUIRE(guard && guard->words && guard->tick != 0 && live && snapshot && n > 0, GSD_ERR_BAD_ARG,
              "gsd_guard_restore: bad argument");
  const int blocks = (int)(ceil_div64(n, 256) < 1024 ? ceil_div64(n, 256) : 1024);
  hipLaunchKernelGGL(guard_copy_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, snapshot, live, (long long)n,
                     (const int*)guard->words, guard->tick);
  GSD_LAUNCH_CHECK("gsd_guard_restore");
  return GSD_OK;
}

extern "C" int gsd_adam_ema(float* p, const float* g, float* m, float* v, float* ema, int64_t numel, int step, float lr,
                            float beta1, float beta2, float eps, float weight_decay, float ema_decay, float grad_scale,
                            const gsd_guard* guard, void* stream) {
  GSD_REQUIRE(p && g && m && v && numel > 0 && step >= 1, GSD_ERR_BAD_ARG, "gsd_adam_ema: bad argument");
  GSD_REQUIRE(guard == nullptr || (guard->words != nullptr && guard->tick != 0), GSD_ERR_BAD_ARG,
              "gsd_adam_ema: a guard needs its two device words and a non-zero tick");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float lr_over_bc1 = (float)((double)lr / bc1);
  const float sqrt_bc2 = (float)sqrt(bc2);
  const int blocks = (int)(ceil_div64(numel, 256) < 4096 ? ceil_div64(numel, 256) : 4096);
  hipLaunchKernelGGL(adam_ema_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, ema, (long long)numel,
                     lr_over_bc1, sqrt_bc2, beta1, beta2, eps, weight_decay, 1.0f - ema_decay, grad_scale,
                     guard ? guard->words : nullptr, guard ? guard->tick : 0);
  GSD_LAUNCH_CHECK("gsd_adam_ema");
  return GSD_OK;
}

// ---------------------------------------------------------------------------------------------
// inference pre/post-processing: F.interpolate(mode='area') == adaptive average pooling,
// fused with the difference image ((a - base + 255) / 2, image_utils.py:6-10) and the per-channel affine of
// normalize_tactile_image / denormalize_depth_image (normalization_utils.py:4-35,101-129).
//   out[n,c,oh,ow] = A[c'] * mean_{window(oh,ow)} pre(in[n,c,h,w]) + B[c'],  c' = min(c, nab-1)
//   window rows [floor(oh*H/OH), ceil((oh+1)*H/OH)), same for columns (ATen adaptive_avg_pool2d)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void area_resize_affine_kernel(const float* __restrict__ in, const float* __restrict__ base,
                                                                 int C, int H, int W, float* __restrict__ out, int OH, int OW,
                                                                 const float* __restrict__ A, const float* __restrict__ B, int nab,
                                                                 float pre_add, float pre_mul) {
  const int c = blockIdx.y, n = blockIdx.z;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= OH * OW) return;
  const int oh = e / OW, ow = e - oh * OW;
  const int h0 = (int)(((long long)oh * H) / OH), h1 = (int)(((long long)(oh + 1) * H + OH - 1) / OH);
  const int w0 = (int)(((long long)ow * W) / OW), w1 = (int)(((long long)(ow + 1) * W + OW - 1) / OW);
  const size_t plane = ((size_t)n * C + c) * H * W;
  float s = 0.f;
  for (int h = h0; h < h1; ++h)
    for (int w = w0; w < w1; ++w) {
      float v = in[plane + (size_t)h * W + w];
      if (base != nullptr) v = (v - base[plane + (size_t)h * W + w] + pre_add) * pre_mul;
      s += v;
    }
  s /= (float)((h1 - h0) * (w1 - w0));
  const int cc = c < nab ? c : nab - 1;
  out[((size_t)n * C + c) * OH * OW + e] = fmaf(s, A[cc], B[cc]);
}
extern "C" int gsd_area_resize_affine(const float* in, const float* base, int N, int C, int H, int W, float* out, int OH, int OW,
                                      const float* A, const float* B, int nab, float pre_add, float pre_mul, void* stream) {
  GSD_REQUIRE(in && out && A && B && N > 0 && C > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && nab > 0, GSD_ERR_BAD_ARG,
              "gsd_area_resize_affine: bad argument");
  GSD_REQUIRE(N <= 65535 && C <= 65535, GSD_ERR_UNSUPPORTED, "gsd_area_resize_affine: N, C must be <= 65535");
  hipLaunchKernelGGL(area_resize_affine_kernel, dim3(ceil_div(OH * OW, 256), C, N), dim3(256), 0, (hipStream_t)stream, in, base,
                     C, H, W, out, OH, OW, A, B, nab, pre_add, pre_mul);
  GSD_LAUNCH_CHECK("gsd_area_resize_affine");
  return GSD_OK;
}
