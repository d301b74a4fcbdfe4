// gsd_igemm.hip -- forward-type convolutions as implicit GEMM on v_mfma_f32_16x16x4_f32 (gfx950).
//
//   D[m = out channel][n = pixel] = sum_k  Wt[k][m] * B[k][pixel]
//
//   MODE 0  conv3x3 p1 s1   k = ci*9 + tap, B = zero-padded 3x3 neighbourhood (halo tile in LDS)
//           replaces aten::convolution at /root/reference/gelslim_depth/models/unet.py:11,14 and,
//           with the dgrad weight layout, the dX half of aten::convolution_backward.
//   MODE 1  convT 2x2 s2    k = ci, m = co*4+kh*2+kw, B = x[ci][pixel]; pixel-shuffle + bias epilogue
//           (unet.py:36,41).
//   MODE 2  convT dgrad     k = co*4+kh*2+kw, m = ci, B = dy[co][2h+kh][2w+kw] (space-to-depth staging)
//
// Pixels sit on the MFMA column (lane&15), so the NCHW store of one accumulator register is 16
// consecutive floats per lane group.  Inputs are NCHW rows loaded coalesced along W, transformed
// (deferred BatchNorm scale/shift + ReLU, zero padding, channel concat of two segments) on their
// way into LDS, so neither relu(bn(x)) nor F.pad/torch.cat (unet.py:46-48) ever exist in HBM.
//
// Block = 256 threads = 4 waves, wave tile 64 (m) x 64 (pixels) = 4x4 MFMA tiles (64 accumulator
// VGPRs); block tile 64x256 (WM=1,WN=4) for Cout<=64, 128x128 (WM=2,WN=2) otherwise.
// K is walked in chunks of 4 input channels (36 k-rows, 9 k-steps) for MODE 0, 16 k-rows for the
// flat modes; the next chunk's global loads are issued before the current chunk's MFMAs
// (register prefetch) and written to LDS after them.
#include "gsd_common.h"

struct IgemmParams {
  SrcD src0, src1;
  DstD dst0, dst1;
  const float* wt;
  const float* bias;
  float* partials;
  int Cin, Cout, Mpad, nchunks, mblocks;
  int N, H, W;
  int TH, TW, tiles_y, tiles_x, WR, WC, PS;  // MODE 0
  int tiles_flat;                            // MODE 1/2
};

template <int MODE, int WM, int WN>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmParams P) {
  constexpr int MT = 4, NT = 4;
  constexpr int BM = WM * 64, BN = WN * 64;
  constexpr int KSTEPS = MODE == 0 ? 9 : 4;
  constexpr int WROWS = KSTEPS * 4;
  constexpr int WS = BM + 16;  // == 16 (mod 32): k-groups of one ds_read_b32 land on disjoint banks
  constexpr int W4 = WROWS * BM / 4;
  constexpr int NW4 = (W4 + 255) / 256;
  constexpr int NXE = MODE == 0 ? 8 : (MODE == 1 ? (16 * BN / 256) : (8 * BN / 256));

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wl = smem;
  float* Xl = smem + WROWS * WS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int j = lane >> 4, l16 = lane & 15;

  const int mb = blockIdx.x % P.mblocks;
  const int pt = blockIdx.x / P.mblocks;
  const int m0 = mb * BM;
  int n, h0 = 0, w0 = 0, p0 = 0;
  if constexpr (MODE == 0) {
    const int tpi = P.tiles_y * P.tiles_x;
    n = pt / tpi;
    const int r = pt - n * tpi;
    const int ty = r / P.tiles_x;
    h0 = ty * P.TH;
    w0 = (r - ty * P.tiles_x) * P.TW;
  } else {
    n = pt / P.tiles_flat;
    p0 = (pt - n * P.tiles_flat) * BN;
  }
  const int HW = P.H * P.W;
  const int PS = MODE == 0 ? P.PS : (BN + 16);

  // ---- per-lane pixel bookkeeping ------------------------------------------------------------
  int baddr[NT];  // LDS offset of this lane's B element for k-step 0 (includes the lane's k row j)
  int opix[NT];   // MODE 0: (r<<16)|c inside the tile, -1 if the lane's pixel is outside it; flat: q or -1
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int q = (wn * NT + t) * 16 + l16;
    if constexpr (MODE == 0) {
      const bool ok = q < P.TH * P.TW;
      const int r = ok ? q / P.TW : 0;
      const int c = ok ? q - r * P.TW : 0;
      baddr[t] = j * PS + r * P.WC + c;
      opix[t] = (ok && (h0 + r) < P.H && (w0 + c) < P.W) ? ((r << 16) | c) : -1;
    } else {
      baddr[t] = j * PS + q;
      opix[t] = (p0 + q) < HW ? q : -1;
    }
  }

  // ---- loader bookkeeping ---------------------------------------------------------------------
  // MODE 0: thread owns window positions tid and tid+256 (row-major (TH+2)x(TW+2) window) for the
  //         4 channels of a chunk.  flat modes: thread owns pixel q = tid % BN for NXE rows.
  int gh[2] = {0, 0}, gw[2] = {0, 0};
  bool pos_ok[2] = {false, false};
  // chunk-invariant per-thread offsets (MODE 0): element offset of each window position inside a channel
  // plane of segment 0 / 1 (-1: outside the segment => zero padding), and of each weight float4 inside a chunk
  int xo0[2] = {-1, -1}, xo1[2] = {-1, -1};
  int wv_off[NW4], wl_off[NW4];
  int fq = 0, frow0 = 0, fh = 0, fw = 0;
  bool f_ok = false;
  if constexpr (MODE == 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pos = tid + i * 256;
      pos_ok[i] = pos < P.WR * P.WC;
      const int rr = pos / P.WC;
      gh[i] = h0 - 1 + rr;
      gw[i] = w0 - 1 + (pos - rr * P.WC);
      if (pos_ok[i]) {
        int hs = gh[i] - P.src0.oh, ws = gw[i] - P.src0.ow;
        if ((unsigned)hs < (unsigned)P.src0.H && (unsigned)ws < (unsigned)P.src0.W) xo0[i] = hs * P.src0.W + ws;
        hs = gh[i] - P.src1.oh;
        ws = gw[i] - P.src1.ow;
        if ((unsigned)hs < (unsigned)P.src1.H && (unsigned)ws < (unsigned)P.src1.W) xo1[i] = hs * P.src1.W + ws;
      }
    }
  } else {
    fq = tid % BN;
    frow0 = tid / BN;  // 0 when BN==256, 0/1 when BN==128
    f_ok = (p0 + fq) < HW;
    if constexpr (MODE == 2) {
      fh = (p0 + fq) / P.W;
      fw = (p0 + fq) - fh * P.W;
    }
  }

#pragma unroll
  for (int i = 0; i < NW4; ++i) {
    const int idx = tid + i * 256;
    const int r = idx / (BM / 4);
    const int c4 = idx % (BM / 4);
    wv_off[i] = (idx < W4 && m0 + c4 * 4 < P.Mpad) ? r * P.Mpad + m0 + c4 * 4 : -1;
    wl_off[i] = r * WS + c4 * 4;
  }
  float xsc[4] = {1.f, 1.f, 1.f, 1.f}, xsh[4] = {0.f, 0.f, 0.f, 0.f};  // per-chunk channel affine (wave-uniform)

  float xr[NXE];   // raw prefetched values (MODE 2: NXE float2 halves stored as 2*NXE floats below)
  float xr2[MODE == 2 ? NXE : 1];
  unsigned xvalid = 0;
  f32x4 wr[NW4];

  auto prefetch = [&](int chunk) {
    // weights: uniform chunk base + chunk-invariant per-thread offset
    const float* wbase = P.wt + (size_t)chunk * WROWS * P.Mpad;
#pragma unroll
    for (int i = 0; i < NW4; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (wv_off[i] >= 0) v = *reinterpret_cast<const f32x4*>(wbase + wv_off[i]);
      wr[i] = v;
    }
    xvalid = 0;
    if constexpr (MODE == 0) {
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        const int c = chunk * 4 + ch;
        const bool first = c < P.src0.C;
        const SrcD& S = first ? P.src0 : P.src1;
        const int cc = first ? c : c - P.src0.C;
        const bool c_ok = c < P.Cin && cc < S.C;
        const float* base = S.p + (long long)n * S.ns + (long long)cc * S.cs;   // wave-uniform
        float sc = 1.f, sh = 0.f;
        if (c_ok && S.scale != nullptr) {
          sc = S.scale[cc];
          sh = S.shift[cc];
        }
        xsc[ch] = sc;
        xsh[ch] = sh;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int off = first ? xo0[i] : xo1[i];
          const bool ok = c_ok && off >= 0;
          float v = 0.f;
          if (ok) v = base[off];
          xr[i * 4 + ch] = v;
          xvalid |= ok ? (1u << (i * 4 + ch)) : 0u;
        }
      }
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int i = 0; i < NXE; ++i) {
        const int ch = frow0 + i * (256 / BN);
        const int c = chunk * 16 + ch;
        const bool ok = f_ok && c < P.Cin;
        float v = 0.f;
        if (ok) v = P.src0.p[(long long)n * P.src0.ns + (long long)c * P.src0.cs + p0 + fq];
        xr[i] = v;
        xvalid |= ok ? (1u << i) : 0u;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NXE; ++i) {
        const int row = frow0 + i * (256 / BN);  // 0..7 : (co_i, kh)
        const int co = chunk * 4 + (row >> 1);
        const int kh = row & 1;
        const bool ok = f_ok && co < P.src0.C;
        float2 v = make_float2(0.f, 0.f);
        if (ok)
          v = *reinterpret_cast<const float2*>(P.src0.p + (long long)n * P.src0.ns + (long long)co * P.src0.cs +
                                               (long long)(2 * fh + kh) * P.src0.W + 2 * fw);
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
    if constexpr (MODE == 0) {
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        const int c = chunk * 4 + ch;
        const int relu = c < P.src0.C ? P.src0.relu : P.src1.relu;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (pos_ok[i]) {
            const bool ok = (xvalid >> (i * 4 + ch)) & 1u;
            const float v = ok ? apply_affine(xr[i * 4 + ch], xsc[ch], xsh[ch], relu) : 0.f;
            Xl[ch * PS + tid + i * 256] = v;
          }
        }
      }
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int i = 0; i < NXE; ++i) {
        const int ch = frow0 + i * (256 / BN);
        const int c = chunk * 16 + ch;
        float sc = 1.f, sh = 0.f;
        if (P.src0.scale != nullptr && c < P.Cin) {
          sc = P.src0.scale[c];
          sh = P.src0.shift[c];
        }
        const bool ok = (xvalid >> i) & 1u;
        Xl[ch * PS + fq] = ok ? apply_affine(xr[i], sc, sh, P.src0.relu) : 0.f;
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
      const int koff = MODE == 0 ? (s / 3) * P.WC + (s % 3) : s * 4 * PS;
      const int arow = MODE == 0 ? (j * 9 + s) : (s * 4 + j);
      float a[MT], b[NT];
#pragma unroll
      for (int m = 0; m < MT; ++m) a[m] = Wl[arow * WS + a_lane + m * 16];
#pragma unroll
      for (int t = 0; t < NT; ++t) b[t] = Xl[baddr[t] + koff];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = mfma16(a[m], b[t], acc[m][t]);
    }
  }

  // ---- epilogue -------------------------------------------------------------------------------
  if constexpr (MODE == 0) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int co = m0 + wm * 64 + m * 16 + j * 4 + reg;
        const bool first = co < P.dst0.C;
        const DstD& D = first ? P.dst0 : P.dst1;
        const int cd = first ? co : co - P.dst0.C;
        const bool co_ok = co < P.Cout && cd < D.C;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (opix[t] >= 0) {
            const float v = acc[m][t][reg];
            s1 += v;
            s2 = fmaf(v, v, s2);
            if (co_ok) {
              const int hd = h0 + (opix[t] >> 16) - D.oh;
              const int wd = w0 + (opix[t] & 0xffff) - D.ow;
              if ((unsigned)hd < (unsigned)D.H && (unsigned)wd < (unsigned)D.W)
                D.p[(long long)n * D.ns + (long long)cd * D.cs + (long long)hd * D.W + wd] = v;
            }
          }
        }
        if (P.partials != nullptr) {
          s1 = reduce16(s1);
          s2 = reduce16(s2);
          if (l16 == 0 && co < P.Mpad) {
            float* row = P.partials + (size_t)(pt * WN + wn) * (2 * P.Mpad);
            row[co] = s1;
            row[P.Mpad + co] = s2;
          }
        }
      }
    }
  } else if constexpr (MODE == 1) {
    const DstD& D = P.dst0;
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
    const DstD& D = P.dst0;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int co = m0 + wm * 64 + m * 16 + j * 4 + reg;
        if (co < D.C) {
#pragma unroll
          for (int t = 0; t < NT; ++t)
            if (opix[t] >= 0)
              D.p[(long long)n * D.ns + (long long)co * D.cs + p0 + opix[t]] = acc[m][t][reg];
        }
      }
  }
}

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
namespace {

int check_src(const gsd_src& s, const char* what) {
  GSD_REQUIRE(s.ptr != nullptr, GSD_ERR_BAD_ARG, "%s: null ptr", what);
  GSD_REQUIRE(s.C > 0 && s.H > 0 && s.W > 0, GSD_ERR_BAD_ARG, "%s: bad dims C=%d H=%d W=%d", what, s.C, s.H, s.W);
  GSD_REQUIRE((s.scale == nullptr) == (s.shift == nullptr), GSD_ERR_BAD_ARG, "%s: scale/shift must come together", what);
  GSD_REQUIRE(s.c_stride >= (int64_t)s.H * s.W && s.n_stride >= s.c_stride, GSD_ERR_BAD_ARG, "%s: strides too small",
              what);
  return 0;
}
int check_dst(const gsd_dst& s, const char* what) {
  GSD_REQUIRE(s.ptr != nullptr, GSD_ERR_BAD_ARG, "%s: null ptr", what);
  GSD_REQUIRE(s.C > 0 && s.H > 0 && s.W > 0, GSD_ERR_BAD_ARG, "%s: bad dims", what);
  GSD_REQUIRE(s.c_stride >= (int64_t)s.H * s.W && s.n_stride >= s.c_stride, GSD_ERR_BAD_ARG, "%s: strides too small",
              what);
  return 0;
}

template <int MODE, int WM, int WN>
int launch(const IgemmParams& P, int grid, size_t lds, hipStream_t st, const char* what) {
  static bool attr_done = false;  // benign race: setting the same attribute twice is harmless
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<MODE, WM, WN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      gsd_set_error("%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e));
      return GSD_ERR_HIP;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL((igemm_kernel<MODE, WM, WN>), dim3(grid), dim3(256), lds, st, P);
  GSD_LAUNCH_CHECK(what);
  return GSD_OK;
}

}  // namespace

extern "C" int gsd_convT2x2(const gsd_src* src, const float* wt, const float* bias, int Cin, int Cout,
                            const gsd_dst* dst, int N, int H, int W, void* stream) {
  GSD_REQUIRE(src && dst && wt, GSD_ERR_BAD_ARG, "gsd_convT2x2: null argument");
  GSD_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, GSD_ERR_BAD_ARG, "gsd_convT2x2: bad sizes");
  if (int e = check_src(*src, "gsd_convT2x2 src")) return e;
  if (int e = check_dst(*dst, "gsd_convT2x2 dst")) return e;
  GSD_REQUIRE(src->C == Cin && src->H == H && src->W == W && src->off_h == 0 && src->off_w == 0, GSD_ERR_BAD_ARG,
              "gsd_convT2x2: src must be the full (Cin,H,W) tensor");
  GSD_REQUIRE(dst->C == Cout && dst->H == 2 * H && dst->W == 2 * W && dst->off_h == 0 && dst->off_w == 0,
              GSD_ERR_BAD_ARG, "gsd_convT2x2: dst must be (Cout,2H,2W)");
  GSD_REQUIRE(((uintptr_t)dst->ptr & 7) == 0 && (dst->c_stride & 1) == 0 && (dst->n_stride & 1) == 0,
              GSD_ERR_UNSUPPORTED, "gsd_convT2x2: dst must be 8-byte aligned with even strides");
  const int M = Cout * 4;
  IgemmParams P;
  P.src0 = to_srcd(*src);
  P.src1 = null_srcd();
  P.dst0 = to_dstd(*dst);
  P.dst1 = null_dstd();
  P.wt = wt; P.bias = bias; P.partials = nullptr;
  P.Cin = Cin; P.Cout = M; P.Mpad = round_up(M, 64);
  P.nchunks = ceil_div(Cin, 16);
  P.N = N; P.H = H; P.W = W;
  P.TH = P.TW = P.tiles_y = P.tiles_x = P.WR = P.WC = P.PS = 0;
  const bool wide = M <= 64;
  const int BM = wide ? 64 : 128, BN = wide ? 256 : 128;
  P.mblocks = ceil_div(M, BM);
  P.tiles_flat = ceil_div(H * W, BN);
  const long grid = (long)N * P.tiles_flat * P.mblocks;
  GSD_REQUIRE(grid < 2147483647L, GSD_ERR_UNSUPPORTED, "gsd_convT2x2: grid too large");
  const size_t lds = (size_t)(16 * (BM + 16) + 16 * (BN + 16)) * sizeof(float);
  if (wide) return launch<1, 1, 4>(P, (int)grid, lds, (hipStream_t)stream, "gsd_convT2x2");
  return launch<1, 2, 2>(P, (int)grid, lds, (hipStream_t)stream, "gsd_convT2x2");
}

extern "C" int gsd_convT2x2_dgrad(const gsd_src* src, const float* wt, int Cin, int Cout, const gsd_dst* dst, int N,
                                  int H, int W, void* stream) {
  GSD_REQUIRE(src && dst && wt, GSD_ERR_BAD_ARG, "gsd_convT2x2_dgrad: null argument");
  GSD_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, GSD_ERR_BAD_ARG, "gsd_convT2x2_dgrad: bad sizes");
  if (int e = check_src(*src, "gsd_convT2x2_dgrad src")) return e;
  if (int e = check_dst(*dst, "gsd_convT2x2_dgrad dst")) return e;
  GSD_REQUIRE(src->C == Cout && src->H == 2 * H && src->W == 2 * W && src->scale == nullptr && src->relu == 0 &&
                  src->off_h == 0 && src->off_w == 0,
              GSD_ERR_BAD_ARG, "gsd_convT2x2_dgrad: src must be the plain (Cout,2H,2W) gradient");
  GSD_REQUIRE(((uintptr_t)src->ptr & 7) == 0 && (src->c_stride & 1) == 0 && (src->n_stride & 1) == 0,
              GSD_ERR_UNSUPPORTED, "gsd_convT2x2_dgrad: src must be 8-byte aligned with even strides");
  GSD_REQUIRE(dst->C == Cin && dst->H == H && dst->W == W && dst->off_h == 0 && dst->off_w == 0, GSD_ERR_BAD_ARG,
              "gsd_convT2x2_dgrad: dst must be (Cin,H,W)");
  IgemmParams P;
  P.src0 = to_srcd(*src);
  P.src1 = null_srcd();
  P.dst0 = to_dstd(*dst);
  P.dst1 = null_dstd();
  P.wt = wt; P.bias = nullptr; P.partials = nullptr;
  P.Cin = Cout * 4;  // GEMM K
  P.Cout = Cin;      // GEMM M
  P.Mpad = round_up(Cin, 64);
  P.nchunks = ceil_div(Cout, 4);
  P.N = N; P.H = H; P.W = W;
  P.TH = P.TW = P.tiles_y = P.tiles_x = P.WR = P.WC = P.PS = 0;
  const bool wide = Cin <= 64;
  const int BM = wide ? 64 : 128, BN = wide ? 256 : 128;
  P.mblocks = ceil_div(Cin, BM);
  P.tiles_flat = ceil_div(H * W, BN);
  const long grid = (long)N * P.tiles_flat * P.mblocks;
  GSD_REQUIRE(grid < 2147483647L, GSD_ERR_UNSUPPORTED, "gsd_convT2x2_dgrad: grid too large");
  const size_t lds = (size_t)(16 * (BM + 16) + 16 * (BN + 16)) * sizeof(float);
  if (wide) return launch<2, 1, 4>(P, (int)grid, lds, (hipStream_t)stream, "gsd_convT2x2_dgrad");
  return launch<2, 2, 2>(P, (int)grid, lds, (hipStream_t)stream, "gsd_convT2x2_dgrad");
}
