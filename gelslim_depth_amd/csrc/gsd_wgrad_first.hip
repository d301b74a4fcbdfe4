// dW of the network's FIRST conv3x3 (3 input channels, 64 output channels at 320x427), with the BatchNorm backward of its
// output applied on the fly.  Replaces, for that one layer, the pair  gsd_bn_bwd_apply + gsd_conv3x3_wgrad  behind
// aten::native_batch_norm_backward / aten::convolution_backward of unet.py:15-18 (DoubleConv's first Conv2d+BatchNorm2d).
//
// Why its own kernel: the layer is nothing like the others.  27 (ci, tap) columns against 64 output channels over 4.4 M
// pixels is 15 GFLOP -- 0.1 ms of matrix-core time -- under 1.1 GB of gradient to read: HBM-bound, where the general
// direct-tap kernel (LDS-DMA gathers sized for >= 16 input channels) spent 1.37 ms.  And the first layer has no dX, so dW is
// the ONLY reader of d_raw = scale * (dz - c1 - (raw - mean) * invstd * c2): forming it in registers from dz and raw removes
// the apply pass (read 2, write 1 tensors of 350 MB: 0.69 ms) and leaves two streamed reads.
//
// GEMM view: M = co (64 = 4 MFMA row tiles), N = (ci, kh, kw) (27 of 2 x 16 columns), K = pixels.  A wave owns one image
// row at a time and walks it in groups of 16 pixels (lane (l16, j) holds pixels 4j .. 4j+3 of the group): A = d_raw straight
// from global memory (unaligned 16-byte loads: W = 427 rows), B = the input image from an LDS window of the block's 4 + 2 rows
// (zero padded), v_mfma_f32_16x16x4_f32 x 32 per group.  Accumulators stay in registers across all the rows a block visits;
// split-K slabs per block and an ordered reduction: bitwise reproducible.
#include "gsd_common.h"

namespace {

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

struct WgFirstParams {
  const float* x;   // the input image: one plain segment
  long long x_ns, x_cs;
  int x_ws;
  const float* dz;    // gradient w.r.t. the BatchNorm output, ReLU mask applied (N, Cout, H, W contiguous)
  const float* raw;   // the conv output the BatchNorm normalised (same shape); unused when scale == nullptr
  const float *scale, *mean, *invstd, *c1, *c2;   // scale == nullptr: dz IS the gradient of the conv output
  float* slabs;       // [gridDim.x][Cout * Cin * 9]
  int N, H, W, Cin, Cout;
  int K9;       // Cin * 9 <= 32
  int G;        // groups of 16 pixels per row
  int XP;       // pitch of an LDS input row: >= 16 G + 2, XP % 32 in {8, 24} (3-way bank conflicts at worst; searched)
  int qpi;      // row quads per image
  int nquads;
};

constexpr int WF_RS = 29;   // pitch of the epilogue's [wave][co][column] image

__device__ __forceinline__ f32x4 wf_load4(const float* p, int nvalid) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (nvalid >= 4) {
    v = *reinterpret_cast<const f32x4u*>(p);
  } else {   // the last group of a row
    if (nvalid > 0) v[0] = p[0];
    if (nvalid > 1) v[1] = p[1];
    if (nvalid > 2) v[2] = p[2];
  }
  return v;
}

template <bool BN>
__global__ __launch_bounds__(256, 4) void wgrad3x3_first_kernel(const WgFirstParams P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [Cin][6][XP]; in the epilogue [4][64][WF_RS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, j = lane >> 4;
  const int cb = blockIdx.y;

  int boff[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int nn = min(nt * 16 + l16, P.K9 - 1);   // columns >= K9 multiply a valid address; their sums are dropped
    const int ci = nn / 9, t = nn - 9 * ci, dh = t / 3, dw = t - 3 * dh;
    boff[nt] = (ci * 6 + wave + dh) * P.XP + dw + 4 * j;   // LDS column c + 1 holds image column c
  }
  float sc[4], mu[4], is[4], k1[4], k2[4];
  int cc[4];
  bool cv[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int co = cb * 64 + mt * 16 + l16;
    cv[mt] = co < P.Cout;
    cc[mt] = cv[mt] ? co : 0;
    if (BN) {
      sc[mt] = P.scale[cc[mt]], mu[mt] = P.mean[cc[mt]], is[mt] = P.invstd[cc[mt]];
      k1[mt] = P.c1[cc[mt]], k2[mt] = P.c2[cc[mt]];
    }
  }
  f32x4 acc[4][2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int win = P.Cin * 6 * P.XP;
  for (int q = blockIdx.x; q < P.nquads; q += gridDim.x) {
    const int n = q / P.qpi, h0 = (q - n * P.qpi) * 4;
    __syncthreads();   // every wave has left the previous quad's window
    for (int e = tid; e < win; e += 256) {
      const int rowi = e / P.XP, col = e - rowi * P.XP;
      const int ci = rowi / 6, h = h0 - 1 + (rowi - 6 * ci), c = col - 1;
      float v = 0.f;
      if ((unsigned)h < (unsigned)P.H && (unsigned)c < (unsigned)P.W)
        v = P.x[(long long)n * P.x_ns + (long long)ci * P.x_cs + (long long)h * P.x_ws + c];
      smem[e] = v;
    }
    __syncthreads();
    const int row = h0 + wave;
    if (row < P.H) {
      size_t rbase[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) rbase[mt] = (((size_t)n * P.Cout + cc[mt]) * P.H + row) * (size_t)P.W + 4 * j;
      for (int g = 0; g < P.G; ++g) {
        const int nv = P.W - (16 * g + 4 * j);   // valid pixels among this lane's four
        f32x4 a[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int nvm = cv[mt] ? nv : 0;
          f32x4 d = wf_load4(P.dz + rbase[mt] + 16 * g, nvm);
          if (BN) {
            const f32x4 r = wf_load4(P.raw + rbase[mt] + 16 * g, nvm);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float t = sc[mt] * (d[i] - k1[mt] - (r[i] - mu[mt]) * is[mt] * k2[mt]);   // gsd_bn_bwd_apply's expression
              d[i] = i < nvm ? t : 0.f;
            }
          }
          a[mt] = d;
        }
        f32x4 b[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int i = 0; i < 4; ++i) b[nt][i] = smem[boff[nt] + 16 * g + i];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma16(a[mt][i], b[nt][i], acc[mt][nt]);
      }
    }
  }

  // the four waves' sums, added in wave order, become this block's slab (dW layout: [co][ci][kh][kw])
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int col = nt * 16 + l16;
        if (col < P.K9) smem[(wave * 64 + mt * 16 + 4 * j + i) * WF_RS + col] = acc[mt][nt][i];
      }
  __syncthreads();
  float* slab = P.slabs + (size_t)blockIdx.x * P.Cout * P.K9;
  for (int e = tid; e < 64 * P.K9; e += 256) {
    const int col_l = e / P.K9, col = e - col_l * P.K9, co = cb * 64 + col_l;
    if (co < P.Cout) {
      float s = smem[col_l * WF_RS + col];
#pragma unroll
      for (int w = 1; w < 4; ++w) s += smem[(w * 64 + col_l) * WF_RS + col];
      slab[(size_t)co * P.K9 + col] = s;
    }
  }
}

// out[e] = sum over slabs, 16 split lanes per element, both stages in a fixed order
__global__ __launch_bounds__(1024) void wgrad_first_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw,
                                                                  int splits, int per) {
  __shared__ float red[16][64];
  const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;
  float s = 0.f;
  if (e < per)
    for (int k = sl; k < splits; k += 16) s += slabs[(size_t)k * per + e];
  red[sl][el] = s;
  __syncthreads();
  if (sl == 0 && e < per) {
    s = red[0][el];
#pragma unroll
    for (int i = 1; i < 16; ++i) s += red[i][el];
    dw[e] = s;
  }
}

struct WfPlan {
  int G, XP, qpi, nquads, grid;
  size_t lds;
};

WfPlan plan_first(int N, int H, int W, int Cin) {
  WfPlan pl;
  pl.G = ceil_div(W, 16);
  int xp = 16 * pl.G + 2;
  while (xp % 32 != 8 && xp % 32 != 24) ++xp;
  pl.XP = xp;
  pl.qpi = ceil_div(H, 4);
  pl.nquads = N * pl.qpi;
  const int win = Cin * 6 * xp, red = 4 * 64 * WF_RS;
  pl.lds = (size_t)(win > red ? win : red) * sizeof(float);
  // five 32-KB blocks per CU at the U-Net's size; a block keeps its accumulators across the quads it walks
  const int cap = 256 * 5;
  pl.grid = pl.nquads < cap ? pl.nquads : cap;
  return pl;
}

}  // namespace

extern "C" int gsd_conv3x3_wgrad_bn_supported(int N, int H, int W, int Cin, int Cout) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  if (gsd_env_int("GSD_WGRAD_FIRST", 1) == 0) return 0;
  if (Cin * 9 > 32 || ceil_div(Cout, 64) > 65535) return 0;
  return plan_first(N, H, W, Cin).lds <= 64 * 1024 ? 1 : 0;
}

extern "C" int64_t gsd_conv3x3_wgrad_bn_workspace(int N, int H, int W, int Cin, int Cout) {
  if (!gsd_conv3x3_wgrad_bn_supported(N, H, W, Cin, Cout)) return 0;
  return (int64_t)plan_first(N, H, W, Cin).grid * Cout * Cin * 9;
}

extern "C" int gsd_conv3x3_wgrad_bn(const gsd_src* a, const float* dz, const float* raw, const float* scale,
                                    const float* mean, const float* invstd, const float* c1, const float* c2, int Cin,
                                    int Cout, float* dw, float* workspace, int64_t workspace_elems, int N, int H, int W,
                                    void* stream) {
  GSD_REQUIRE(a && dz && dw && workspace, GSD_ERR_BAD_ARG, "gsd_conv3x3_wgrad_bn: null argument");
  GSD_REQUIRE(gsd_conv3x3_wgrad_bn_supported(N, H, W, Cin, Cout), GSD_ERR_UNSUPPORTED,
              "gsd_conv3x3_wgrad_bn: serves Cin * 9 <= 32 only (got Cin %d)", Cin);
  GSD_REQUIRE(a->ptr != nullptr && a->C == Cin && a->H == H && a->W == W && a->off_h == 0 && a->off_w == 0, GSD_ERR_BAD_ARG,
              "gsd_conv3x3_wgrad_bn: the activation must be one full (Cin,H,W) segment");
  GSD_REQUIRE(a->scale == nullptr && a->shift == nullptr && a->relu == 0, GSD_ERR_UNSUPPORTED,
              "gsd_conv3x3_wgrad_bn: the activation must be plain (the network's input image)");
  GSD_REQUIRE((scale == nullptr) == (raw == nullptr), GSD_ERR_BAD_ARG, "gsd_conv3x3_wgrad_bn: raw and scale come together");
  GSD_REQUIRE(scale == nullptr || (mean && invstd && c1 && c2), GSD_ERR_BAD_ARG,
              "gsd_conv3x3_wgrad_bn: mean, invstd, c1, c2 are required with scale");
  const WfPlan pl = plan_first(N, H, W, Cin);
  const int64_t need = (int64_t)pl.grid * Cout * Cin * 9;
  GSD_REQUIRE(workspace_elems >= need, GSD_ERR_WORKSPACE, "gsd_conv3x3_wgrad_bn: workspace %lld < %lld elements",
              (long long)workspace_elems, (long long)need);
  WgFirstParams P;
  P.x = a->ptr;
  P.x_ns = a->n_stride;
  P.x_cs = a->c_stride;
  P.x_ws = a->w_stride > 0 ? a->w_stride : W;
  P.dz = dz; P.raw = raw;
  P.scale = scale; P.mean = mean; P.invstd = invstd; P.c1 = c1; P.c2 = c2;
  P.slabs = workspace;
  P.N = N; P.H = H; P.W = W; P.Cin = Cin; P.Cout = Cout;
  P.K9 = Cin * 9;
  P.G = pl.G; P.XP = pl.XP; P.qpi = pl.qpi; P.nquads = pl.nquads;
  const dim3 grid(pl.grid, ceil_div(Cout, 64));
  const hipStream_t st = (hipStream_t)stream;
  if (scale != nullptr)
    hipLaunchKernelGGL(wgrad3x3_first_kernel<true>, grid, dim3(256), pl.lds, st, P);
  else
    hipLaunchKernelGGL(wgrad3x3_first_kernel<false>, grid, dim3(256), pl.lds, st, P);
  GSD_LAUNCH_CHECK("gsd_conv3x3_wgrad_bn");
  const int per = Cout * Cin * 9;
  hipLaunchKernelGGL(wgrad_first_reduce_kernel, dim3(ceil_div(per, 64)), dim3(1024), 0, st, workspace, dw, pl.grid, per);
  GSD_LAUNCH_CHECK("gsd_conv3x3_wgrad_bn reduce");
  return GSD_OK;
}
