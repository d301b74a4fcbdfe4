// dW of the network's FIRST conv3x3 (3 input channels, 64 output channels at 320x427), with the BatchNorm backward of its
// output applied on the fly.  Replaces, for that one layer, the pair  gsd_bn_bwd_apply + gsd_conv3x3_wgrad  behind
// aten::native_batch_norm_backward / aten::convolution_backward of unet.py:15-18 (DoubleConv's first Conv2d+BatchNorm2d).
//
// Why its own kernel: the layer is nothing like the others.  27 (ci, tap) columns against 64 output channels over 4.4 M
// pixels is 15 GFLOP -- 0.1 ms of matrix-core time -- under 1.12 GB of gradient to read: HBM-bound, where the general
// direct-tap kernel (LDS-DMA gathers sized for >= 16 input channels) spent 1.37 ms.  And the first layer has no dX, so dW is
// the ONLY reader of d_raw = scale * (dz - c1 - (raw - mean) * invstd * c2): forming it in registers from dz and raw removes
// the apply pass (read 2, write 1 tensors of 1.12 GB: 0.69 ms) and leaves two streamed reads (2.24 GB).
//
// GEMM view: M = co (64 = one MFMA row tile per wave), N = (ci, kh, kw) (27 of 2 x 16 columns), K = pixels.  A block walks
// image rows in spans of 64 pixels.  A = d_raw: wave w owns output channels 16w .. 16w+15 and loads their 64-pixel pieces of dz
// and raw as 256 contiguous bytes per row (16 lanes x 16 bytes, unaligned: W = 427), one span ahead in registers, then through a
// wave-private LDS image (row pitch 68: conflict-free 16-byte transposed reads) into the MFMA's lane order.  (The first form
// read the MFMA order straight from global memory -- 64 bytes per row and visit -- and FETCH_SIZE showed every 128-byte line
// fetched 2.4 times.)  B = the input image from an LDS window of the block's 4 + 2 rows, zero padded.  16 k-steps x 2 column
// tiles = 32 v_mfma_f32_16x16x4_f32 per wave and span.  Accumulators stay in registers across all the rows a block visits;
// one slab per block and an ordered reduction: bitwise reproducible.
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
  int S;        // spans of 64 pixels per row
  int XP;       // pitch of an LDS input row: >= 64 S + 2, XP % 32 in {8, 24} (3-way bank conflicts at worst; searched)
  int qpi;      // row quads per image
  int nquads;
};

constexpr int WF_AS = 68;   // row pitch of a wave's [16 co][64 px] gradient image: 16-byte reads down a column hit 8 bank groups twice

__device__ __forceinline__ f32x4 wf_load4(const float* p, int nvalid) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (nvalid >= 4) {
    v = *reinterpret_cast<const f32x4u*>(p);
  } else {   // the last pieces of a row
    if (nvalid > 0) v[0] = p[0];
    if (nvalid > 1) v[1] = p[1];
    if (nvalid > 2) v[2] = p[2];
  }
  return v;
}

template <bool BN>
__global__ __launch_bounds__(256, 2) void wgrad3x3_first_kernel(const WgFirstParams P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [Cin][6][XP] input window, then per wave [2][16][WF_AS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, j = lane >> 4;
  const int cb = blockIdx.y;
  const int win = (P.Cin * 6 * P.XP + 3) & ~3;
  float* const aw = smem + win + wave * (2 * 16 * WF_AS);

  int boff[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int nn = min(nt * 16 + l16, P.K9 - 1);   // columns >= K9 multiply a valid address; their sums are dropped
    const int ci = nn / 9, t = nn - 9 * ci, dh = t / 3, dw = t - 3 * dh;
    boff[nt] = (ci * 6 + dh) * P.XP + dw + 4 * j;   // LDS column c + 1 holds image column c
  }
  // MFMA role: lane (l16, j) holds output channel 16 wave + l16, pixels 4j .. 4j+3 of a 16-pixel group
  const int co_m = cb * 64 + wave * 16 + l16;
  const bool cv_m = co_m < P.Cout;
  float sc = 0.f, mu = 0.f, is = 0.f, k1 = 0.f, k2 = 0.f;
  if (BN && cv_m) sc = P.scale[co_m], mu = P.mean[co_m], is = P.invstd[co_m], k1 = P.c1[co_m], k2 = P.c2[co_m];
  // loader role: lane (r4, piece) moves the 16-byte piece `piece` of rows 4k + r4 (k = 0..3) of the wave's 16 channels
  const int r4 = lane >> 4, piece = lane & 15;
  size_t lbase[4];
  bool lv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int co = cb * 64 + wave * 16 + 4 * k + r4;
    lv[k] = co < P.Cout;
    lbase[k] = (size_t)(lv[k] ? co : 0) * P.H * P.W + 4 * piece;
  }
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};

  for (int q = blockIdx.x; q < P.nquads; q += gridDim.x) {
    const int n = q / P.qpi, h0 = (q - n * P.qpi) * 4;
    __syncthreads();   // every wave has left the previous quad's window
    // window rows go round the waves; a wave moves a row as up to 8 coalesced 64-column pieces, the loads of a row in flight together
    for (int rowi = wave; rowi < P.Cin * 6; rowi += 4) {
      const int ci = rowi / 6, h = h0 - 1 + (rowi - 6 * ci);
      const bool hv = (unsigned)h < (unsigned)P.H;
      const float* xr = P.x + (long long)n * P.x_ns + (long long)ci * P.x_cs + (long long)(hv ? h : 0) * P.x_ws - 1;
      float* lr = smem + rowi * P.XP;
      for (int c0 = 0; c0 < P.XP; c0 += 512) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int col = c0 + 64 * k + lane;
          v[k] = (hv && col >= 1 && col <= P.W) ? xr[col] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int col = c0 + 64 * k + lane;
          if (col < P.XP) lr[col] = v[k];
        }
      }
    }
    __syncthreads();
    const int rows = min(4, P.H - h0);
    const size_t nbase = (size_t)n * P.Cout * P.H * P.W;
    f32x4 dn[4], rn[4];
    auto fetch = [&](int r, int s) {   // span s of row h0 + r: this wave's 16 channels x 64 pixels of dz (and raw)
      const int nv = P.W - (64 * s + 4 * piece);
      const size_t off = nbase + (size_t)(h0 + r) * P.W + 64 * s;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        dn[k] = wf_load4(P.dz + off + lbase[k], lv[k] ? nv : 0);
        if (BN) rn[k] = wf_load4(P.raw + off + lbase[k], lv[k] ? nv : 0);
      }
    };
    fetch(0, 0);
    for (int r = 0; r < rows; ++r)
      for (int s = 0; s < P.S; ++s) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          *reinterpret_cast<f32x4*>(&aw[(4 * k + r4) * WF_AS + 4 * piece]) = dn[k];
          if (BN) *reinterpret_cast<f32x4*>(&aw[(16 + 4 * k + r4) * WF_AS + 4 * piece]) = rn[k];
        }
        // the next span's pieces fly during this span's 32 MFMAs
        if (s + 1 < P.S) fetch(r, s + 1);
        else if (r + 1 < rows) fetch(r + 1, 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const float* xb = smem + r * P.XP + 64 * s;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nv = cv_m ? P.W - (64 * s + 16 * g + 4 * j) : 0;   // valid pixels among this lane's four
          f32x4 a = *reinterpret_cast<const f32x4*>(&aw[l16 * WF_AS + 16 * g + 4 * j]);
          if (BN) {
            const f32x4 rw = *reinterpret_cast<const f32x4*>(&aw[(16 + l16) * WF_AS + 16 * g + 4 * j]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float t = sc * (a[i] - k1 - (rw[i] - mu) * is * k2);   // gsd_bn_bwd_apply's expression
              a[i] = i < nv ? t : 0.f;
            }
          }
          f32x4 b[2];
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) b[nt][i] = xb[boff[nt] + 16 * g + i];
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[nt] = mfma16(a[i], b[nt][i], acc[nt]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // the reads above precede the next span's writes
      }
  }

  // a wave's accumulators are whole sums for its 16 channels: straight into this block's slab (dW layout: [co][ci][kh][kw])
  float* slab = P.slabs + (size_t)blockIdx.x * P.Cout * P.K9;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int co = cb * 64 + wave * 16 + 4 * j + i, col = nt * 16 + l16;
      if (co < P.Cout && col < P.K9) slab[(size_t)co * P.K9 + col] = acc[nt][i];
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
  int S, XP, qpi, nquads, grid;
  size_t lds;
};

WfPlan plan_first(int N, int H, int W, int Cin) {
  WfPlan pl;
  pl.S = ceil_div(W, 64);
  int xp = 64 * pl.S + 2;
  while (xp % 32 != 8 && xp % 32 != 24) ++xp;
  pl.XP = xp;
  pl.qpi = ceil_div(H, 4);
  pl.nquads = N * pl.qpi;
  pl.lds = (size_t)(((Cin * 6 * xp + 3) & ~3) + 4 * 2 * 16 * WF_AS) * sizeof(float);
  // two ~66-KB blocks per CU at the U-Net's size; a block keeps its accumulators across the quads it walks
  const int cap = 256 * 2;
  pl.grid = pl.nquads < cap ? pl.nquads : cap;
  return pl;
}

}  // namespace

extern "C" int gsd_conv3x3_wgrad_bn_supported(int N, int H, int W, int Cin, int Cout) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  if (gsd_env_int("GSD_WGRAD_FIRST", 1) == 0) return 0;
  if (Cin * 9 > 32 || ceil_div(Cout, 64) > 65535) return 0;
  return plan_first(N, H, W, Cin).lds <= 80 * 1024 ? 1 : 0;
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
  P.S = pl.S; P.XP = pl.XP; P.qpi = pl.qpi; P.nquads = pl.nquads;
  const dim3 grid(pl.grid, ceil_div(Cout, 64));
  const hipStream_t st = (hipStream_t)stream;
  // one attribute cache per instantiation: the kernel's address keys it (gsd_common.h)
  static gsd_attr_once once_bn, once_plain;
  const void* fn = scale != nullptr ? reinterpret_cast<const void*>(&wgrad3x3_first_kernel<true>)
                                    : reinterpret_cast<const void*>(&wgrad3x3_first_kernel<false>);
  if (hipError_t e = gsd_allow_big_lds(scale != nullptr ? once_bn : once_plain, fn); e != hipSuccess) {
    gsd_set_error("gsd_conv3x3_wgrad_bn: hipFuncSetAttribute: %s", hipGetErrorString(e));
    return GSD_ERR_HIP;
  }
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
