// gsd_convT.hip -- ConvTranspose2d(k=2, s=2) forward and its data gradient as flat implicit GEMMs on
// v_mfma_f32_16x16x4_f32 (gfx950).  "Flat": the GEMM column is a run of BN consecutive pixels of one
// image plane (no halo), because a 2x2 stride-2 transposed convolution has no spatial overlap.
//
//   D[m][pixel] = sum_k  Wt[k][m] * B[k][pixel]
//
//   FWD    k = ci, m = co*4+kh*2+kw, B = x[ci][pixel]; pixel-shuffle + bias epilogue
//          (replaces aten::conv_transpose2d at /root/reference/gelslim_depth/models/unet.py:36,41).
//   DGRAD  k = co*4+kh*2+kw, m = ci, B = dy[co][2h+kh][2w+kw] (space-to-depth staging)
//          (the dX half of aten::convolution_backward for the same operator).
//
// Pixels sit on the MFMA column (lane&15), so the NCHW store of one accumulator register is 16
// consecutive floats per lane group.  FWD applies the producer's deferred BatchNorm scale/shift +
// ReLU on the way into LDS, so relu(bn(x)) never exists in HBM.
//
// Block = 256 threads = 4 waves, wave tile 64 (m) x 64 (pixels) = 4x4 MFMA tiles (64 accumulator
// VGPRs); block tile 64x256 (WM=1,WN=4) for M<=64, 128x128 (WM=2,WN=2) otherwise.  K is walked in
// chunks of 16 k-rows (4 k-steps); the next chunk's global loads are issued before the current
// chunk's MFMAs (register prefetch) and written to LDS after them.  The 3x3 convolutions, which
// dominate the step, live in gsd_conv3x3.hip / gsd_wgrad.hip; these two operators are ~4% of it.
#include "gsd_common.h"

namespace {

enum : int { CT_FWD = 1, CT_DGRAD = 2 };

struct ConvTParams {
  SrcD src;
  DstD dst;
  const float* wt;    // [K][Mpad] (gsd_weight_layout modes 2/3)
  const float* bias;  // FWD only, may be null
  int K, M, Mpad, nchunks, mblocks;
  int N, H, W;        // the LOW-resolution plane (convT input / dgrad output)
  int tiles_flat;
};

template <int MODE, int WM, int WN>
__global__ __launch_bounds__(256) void convT_kernel(const ConvTParams P) {
  constexpr int MT = 4, NT = 4;
  constexpr int BM = WM * 64, BN = WN * 64;
  constexpr int KSTEPS = 4;
  constexpr int WROWS = KSTEPS * 4;
  constexpr int WS = BM + 16;  // == 16 (mod 32): k-groups of one ds_read_b32 land on disjoint banks
  constexpr int PS = BN + 16;
  constexpr int W4 = WROWS * BM / 4;
  constexpr int NW4 = (W4 + 255) / 256;
  constexpr int NXE = MODE == CT_FWD ? (16 * BN / 256) : (8 * BN / 256);

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wl = smem;
  float* Xl = smem + WROWS * WS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int j = lane >> 4, l16 = lane & 15;

  const int mb = blockIdx.x % P.mblocks;
  const int pt = blockIdx.x / P.mblocks;
  const int m0 = mb * BM;
  const int n = pt / P.tiles_flat;
  const int p0 = (pt - n * P.tiles_flat) * BN;
  const int HW = P.H * P.W;

  // per-lane pixel bookkeeping: LDS offset of the lane's B element for k-step 0, and its pixel (or -1)
  int baddr[NT], opix[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int q = (wn * NT + t) * 16 + l16;
    baddr[t] = j * PS + q;
    opix[t] = (p0 + q) < HW ? q : -1;
  }

  // loader bookkeeping: thread owns pixel fq = tid % BN for NXE k-rows
  const int fq = tid % BN;
  const int frow0 = tid / BN;  // 0 when BN==256, 0/1 when BN==128
  const bool f_ok = (p0 + fq) < HW;
  int fh = 0, fw = 0;
  if constexpr (MODE == CT_DGRAD) {
    fh = (p0 + fq) / P.W;
    fw = (p0 + fq) - fh * P.W;
  }
  int wv_off[NW4], wl_off[NW4];
#pragma unroll
  for (int i = 0; i < NW4; ++i) {
    const int idx = tid + i * 256;
    const int r = idx / (BM / 4);
    const int c4 = idx % (BM / 4);
    wv_off[i] = (idx < W4 && m0 + c4 * 4 < P.Mpad) ? r * P.Mpad + m0 + c4 * 4 : -1;
    wl_off[i] = r * WS + c4 * 4;
  }

  float xr[NXE];
  float xr2[MODE == CT_DGRAD ? NXE : 1];  // DGRAD loads float2 (kw = 0,1)
  unsigned xvalid = 0;
  f32x4 wr[NW4];

  auto prefetch = [&](int chunk) {
    const float* wbase = P.wt + (size_t)chunk * WROWS * P.Mpad;
#pragma unroll
    for (int i = 0; i < NW4; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (wv_off[i] >= 0) v = *reinterpret_cast<const f32x4*>(wbase + wv_off[i]);
      wr[i] = v;
    }
    xvalid = 0;
    if constexpr (MODE == CT_FWD) {
#pragma unroll
      for (int i = 0; i < NXE; ++i) {
        const int c = chunk * 16 + frow0 + i * (256 / BN);
        const bool ok = f_ok && c < P.K;
        float v = 0.f;
        if (ok) v = P.src.p[(long long)n * P.src.ns + (long long)c * P.src.cs + p0 + fq];
        xr[i] = v;
        xvalid |= ok ? (1u << i) : 0u;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NXE; ++i) {
        const int row = frow0 + i * (256 / BN);  // 0..7 : (co_i, kh)
        const int co = chunk * 4 + (row >> 1);
        const int kh = row & 1;
        const bool ok = f_ok && co < P.src.C;
        float2 v = make_float2(0.f, 0.f);
        if (ok)
          v = *reinterpret_cast<const float2*>(P.src.p + (long long)n * P.src.ns + (long long)co * P.src.cs +
                                               (long long)(2 * fh + kh) * P.src.W + 2 * fw);
        xr[i] = v.x;
        xr2[i] = v.y;
      }
    }
  };

  auto stage = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < NW4; ++i) {
      if (tid + i * 256 < W4) *reinterpret_cast<f32x4*>(&Wl[wl_off[i]]) = wr[i];
    }
    if constexpr (MODE == CT_FWD) {
#pragma unroll
      for (int i = 0; i < NXE; ++i) {
        const int ch = frow0 + i * (256 / BN);
        const int c = chunk * 16 + ch;
        float sc = 1.f, sh = 0.f;
        if (P.src.scale != nullptr && c < P.K) {
          sc = P.src.scale[c];
          sh = P.src.shift[c];
        }
        const bool ok = (xvalid >> i) & 1u;
        Xl[ch * PS + fq] = ok ? apply_affine(xr[i], sc, sh, P.src.relu) : 0.f;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NXE; ++i) {
        const int row = frow0 + i * (256 / BN);
        const int k = (row >> 1) * 4 + (row & 1) * 2;
        Xl[k * PS + fq] = xr[i];
        Xl[(k + 1) * PS + fq] = xr2[i];
      }
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int a_lane = wm * 64 + l16;
  prefetch(0);
  for (int chunk = 0; chunk < P.nchunks; ++chunk) {
    __syncthreads();  // everyone finished reading the previous chunk's LDS image
    stage(chunk);
    __syncthreads();
    if (chunk + 1 < P.nchunks) prefetch(chunk + 1);
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      float a[MT], b[NT];
#pragma unroll
      for (int m = 0; m < MT; ++m) a[m] = Wl[(s * 4 + j) * WS + a_lane + m * 16];
#pragma unroll
      for (int t = 0; t < NT; ++t) b[t] = Xl[baddr[t] + s * 4 * PS];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = mfma16(a[m], b[t], acc[m][t]);
    }
  }

  // ---- epilogue -------------------------------------------------------------------------------
  const DstD& D = P.dst;
  if constexpr (MODE == CT_FWD) {
    // one accumulator quad = the 2x2 output patch of (co, pixel): two float2 stores
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int co = (m0 + wm * 64 + m * 16 + j * 4) >> 2;
      if (co < D.C) {
        const float bz = P.bias != nullptr ? P.bias[co] : 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (opix[t] >= 0) {
            const int p = p0 + opix[t];
            const int h = p / P.W, w = p - h * P.W;
            float* o = D.p + (long long)n * D.ns + (long long)co * D.cs + (long long)(2 * h) * D.W + 2 * w;
            *reinterpret_cast<float2*>(o) = make_float2(acc[m][t][0] + bz, acc[m][t][1] + bz);
            *reinterpret_cast<float2*>(o + D.W) = make_float2(acc[m][t][2] + bz, acc[m][t][3] + bz);
          }
        }
      }
    }
  } else {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int ci = m0 + wm * 64 + m * 16 + j * 4 + reg;
        if (ci < D.C) {
#pragma unroll
          for (int t = 0; t < NT; ++t)
            if (opix[t] >= 0) D.p[(long long)n * D.ns + (long long)ci * D.cs + p0 + opix[t]] = acc[m][t][reg];
        }
      }
  }
}

template <int MODE, int WM, int WN>
int launch(const ConvTParams& P, int grid, size_t lds, hipStream_t st, const char* what) {
  hipLaunchKernelGGL((convT_kernel<MODE, WM, WN>), dim3(grid), dim3(256), lds, st, P);
  GSD_LAUNCH_CHECK(what);
  return GSD_OK;
}

// common tail of both entry points: tile choice, grid, launch
template <int MODE>
int run(ConvTParams& P, hipStream_t st, const char* what) {
  const bool wide = P.M <= 64;
  const int BM = wide ? 64 : 128, BN = wide ? 256 : 128;
  P.Mpad = round_up(P.M, 64);
  P.mblocks = ceil_div(P.M, BM);
  P.tiles_flat = ceil_div(P.H * P.W, BN);
  const long grid = (long)P.N * P.tiles_flat * P.mblocks;
  GSD_REQUIRE(grid < 2147483647L, GSD_ERR_UNSUPPORTED, "%s: grid too large", what);
  const size_t lds = (size_t)(16 * (BM + 16) + 16 * (BN + 16)) * sizeof(float);  // <= 24 KB, no attribute needed
  if (wide) return launch<MODE, 1, 4>(P, (int)grid, lds, st, what);
  return launch<MODE, 2, 2>(P, (int)grid, lds, st, what);
}

}  // namespace

extern "C" int gsd_convT2x2(const gsd_src* src, const float* wt, const float* bias, int Cin, int Cout,
                            const gsd_dst* dst, int N, int H, int W, void* stream) {
  GSD_REQUIRE(src && dst && wt, GSD_ERR_BAD_ARG, "gsd_convT2x2: null argument");
  GSD_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, GSD_ERR_BAD_ARG, "gsd_convT2x2: bad sizes");
  if (int e = gsd_check_src(*src, "gsd_convT2x2 src")) return e;
  if (int e = gsd_check_dst(*dst, "gsd_convT2x2 dst")) return e;
  GSD_REQUIRE(src->C == Cin && src->H == H && src->W == W && src->off_h == 0 && src->off_w == 0, GSD_ERR_BAD_ARG,
              "gsd_convT2x2: src must be the full (Cin,H,W) tensor");
  GSD_REQUIRE(dst->C == Cout && dst->H == 2 * H && dst->W == 2 * W && dst->off_h == 0 && dst->off_w == 0,
              GSD_ERR_BAD_ARG, "gsd_convT2x2: dst must be (Cout,2H,2W)");
  GSD_REQUIRE(((uintptr_t)dst->ptr & 7) == 0 && (dst->c_stride & 1) == 0 && (dst->n_stride & 1) == 0,
              GSD_ERR_UNSUPPORTED, "gsd_convT2x2: dst must be 8-byte aligned with even strides");
  ConvTParams P;
  P.src = to_srcd(*src);
  P.dst = to_dstd(*dst);
  P.wt = wt;
  P.bias = bias;
  P.K = Cin;
  P.M = Cout * 4;
  P.nchunks = ceil_div(Cin, 16);
  P.N = N; P.H = H; P.W = W;
  return run<CT_FWD>(P, (hipStream_t)stream, "gsd_convT2x2");
}

extern "C" int gsd_convT2x2_dgrad(const gsd_src* src, const float* wt, int Cin, int Cout, const gsd_dst* dst, int N,
                                  int H, int W, void* stream) {
  GSD_REQUIRE(src && dst && wt, GSD_ERR_BAD_ARG, "gsd_convT2x2_dgrad: null argument");
  GSD_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, GSD_ERR_BAD_ARG, "gsd_convT2x2_dgrad: bad sizes");
  if (int e = gsd_check_src(*src, "gsd_convT2x2_dgrad src")) return e;
  if (int e = gsd_check_dst(*dst, "gsd_convT2x2_dgrad dst")) return e;
  GSD_REQUIRE(src->C == Cout && src->H == 2 * H && src->W == 2 * W && src->scale == nullptr && src->relu == 0 &&
                  src->off_h == 0 && src->off_w == 0,
              GSD_ERR_BAD_ARG, "gsd_convT2x2_dgrad: src must be the plain (Cout,2H,2W) gradient");
  GSD_REQUIRE(((uintptr_t)src->ptr & 7) == 0 && (src->c_stride & 1) == 0 && (src->n_stride & 1) == 0,
              GSD_ERR_UNSUPPORTED, "gsd_convT2x2_dgrad: src must be 8-byte aligned with even strides");
  GSD_REQUIRE(dst->C == Cin && dst->H == H && dst->W == W && dst->off_h == 0 && dst->off_w == 0, GSD_ERR_BAD_ARG,
              "gsd_convT2x2_dgrad: dst must be (Cin,H,W)");
  ConvTParams P;
  P.src = to_srcd(*src);
  P.dst = to_dstd(*dst);
  P.wt = wt;
  P.bias = nullptr;
  P.K = Cout * 4;
  P.M = Cin;
  P.nchunks = ceil_div(Cout, 4);
  P.N = N; P.H = H; P.W = W;
  return run<CT_DGRAD>(P, (hipStream_t)stream, "gsd_convT2x2_dgrad");
}
