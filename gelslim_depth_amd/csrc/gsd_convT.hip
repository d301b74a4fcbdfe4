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

// -------------------------------------------------------------------------------------------------------------------------
// FWD on the LDS-DMA template of the conv3x3 kernels (the register-staged form above ran at 48 % of the fp32 MFMA peak and
// stays for DGRAD): block tile 128 (m = co*4+kh*2+kw) x 128 pixels, K walked in chunks of 32 input channels, both operand
// tiles straight from L2/HBM into a double-buffered LDS image by global_load_lds -- weights as 16 one-KiB pieces of a
// [chunk][32][128] image whose columns are permuted so that a lane's four A operands are one ds_read_b128
// (gsd_weight_layout mode 6), activations as rows of 128 consecutive pixels of a channel plane (16-byte pieces when the
// planes are 16-byte aligned and H*W % 4 == 0 -- true at every level of the U-Net -- else dwords).  MFMA column l16 of
// n-tile t is pixel 4*l16 + t of the wave's 64: the four B operands of a lane are one ds_read_b128 as well, and the
// producer's deferred BatchNorm+ReLU is applied between the read and the MFMA.  Two blocks per CU.
struct ConvTFwdParams {
  SrcD src;
  DstD dst;
  const float* wt;    // mode 6: [mblock][Kpad][128]
  const float* bias;
  int K, Kpad, M, nchunks, mblocks;
  int N, H, W, tiles_flat;
};

__device__ __attribute__((aligned(16))) const float gsd_zero16_ct[4] = {0.f, 0.f, 0.f, 0.f};

template <bool X4>
__global__ __launch_bounds__(256, 2) void convT_fwd_dma_kernel(const ConvTFwdParams P) {
  constexpr int BM = 128, BN = 128, KC = 32, MT = 4, NT = 4;
  constexpr int WIMG = KC * BM, XIMG = KC * BN, BUF = WIMG + XIMG;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int j = lane >> 4, l16 = lane & 15;

  const int lid = xcd_swizzle(blockIdx.x, gridDim.x);   // the m-blocks of a pixel tile share the activation tile: one XCD's L2
  const int mb = lid % P.mblocks;
  const int pt = lid / P.mblocks;
  const int m0 = mb * BM;
  // pixel tiles run over the flattened pixels of the WHOLE batch (a 20 x 26 image is 4.06 tiles of 128): q = n * H*W + p
  const long long q0 = (long long)pt * BN;
  const int HW = P.H * P.W;
  const long long QT = (long long)P.N * HW;

  // deferred BatchNorm coefficients of every input channel (k), padded with the identity
  float* sAff = smem + 2 * BUF;
  for (int c = tid; c < P.Kpad; c += 256) {
    float sc = 1.f, sh = 0.f;
    if (c < P.K && P.src.scale != nullptr) {
      sc = P.src.scale[c];
      sh = P.src.shift[c];
    }
    sAff[c] = sc;
    sAff[P.Kpad + c] = sh;
  }
  const float lo = P.src.relu ? 0.f : -__builtin_inff();

  // ---- DMA lane geometry ---------------------------------------------------------------------------------------------------
  // weights: the chunk image is 16 KiB = 16 pieces of 1 KiB; wave w moves pieces w, w+4, w+8, w+12
  const float* const wsrc = P.wt + (size_t)mb * P.Kpad * BM + lane * 4;
  // activations: X4: a piece = 2 channel rows (lane>>5) x 32 sixteen-byte pieces (lane&31); dword: one row half (64 px)
  const int xrow = X4 ? (lane >> 5) : 0;
  const int xpx = X4 ? (lane & 31) * 4 : lane;
  // this lane's pixel(s) of the tile: image and offset inside a channel plane (X4: H*W % 4 == 0, a piece never straddles images)
  long long xo[X4 ? 1 : 2];
  bool xok[X4 ? 1 : 2];
#pragma unroll
  for (int i = 0; i < (X4 ? 1 : 2); ++i) {
    const long long q = q0 + xpx + 64 * i;
    xok[i] = q < QT;
    const int nn = xok[i] ? (int)(q / HW) : 0;
    xo[i] = (long long)nn * P.src.ns + (xok[i] ? (q - (long long)nn * HW) : 0);
  }
  auto fill = [&](int chunk, int buf) {
    float* Wb = smem + buf * BUF;
    float* Xb = Wb + WIMG;
    const float* wc = wsrc + (size_t)chunk * WIMG;
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds(wc + (wave + 4 * i) * 256, Wb + (wave + 4 * i) * 256, 16, 0, 0);
    const int k0 = chunk * KC;
    if constexpr (X4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 2 * (wave + 4 * i) + xrow;   // channel row of the chunk
        const float* g = (xok[0] && k0 + r < P.K) ? P.src.p + xo[0] + (long long)(k0 + r) * P.src.cs : &gsd_zero16_ct[0];
        __builtin_amdgcn_global_load_lds(g, Xb + (wave + 4 * i) * 256, 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int u = wave + 4 * i;              // 64 half rows of 64 pixels
        const int r = u >> 1, hf = u & 1;
        const float* g = (xok[hf] && k0 + r < P.K) ? P.src.p + xo[hf] + (long long)(k0 + r) * P.src.cs : &gsd_zero16_ct[0];
        __builtin_amdgcn_global_load_lds(g, Xb + u * 64, 4, 0, 0);
      }
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int a_off = j * BM + wm * 64 + l16 * 4;
  const int b_off = WIMG + j * BN + wn * 64 + l16 * 4;
  fill(0, 0);
  for (int chunk = 0; chunk < P.nchunks; ++chunk) {
    const int cur = chunk & 1;
    gsd_dma_barrier();   // this chunk has landed; everyone has left the other image
    if (chunk + 1 < P.nchunks) fill(chunk + 1, cur ^ 1);
    const float* Sb = smem + cur * BUF;
    const float* aff = sAff + chunk * KC + j;
    f32x4 av[2], bv[2];
    float sc[2], sh[2];
    av[0] = *reinterpret_cast<const f32x4*>(&Sb[a_off]);
    bv[0] = *reinterpret_cast<const f32x4*>(&Sb[b_off]);
    sc[0] = aff[0];
    sh[0] = aff[P.Kpad];
#pragma unroll
    for (int s = 0; s < KC / 4; ++s) {
      const int c = s & 1;
      if (s + 1 < KC / 4) {   // the next k-step's operands fly during this one's MFMAs
        av[c ^ 1] = *reinterpret_cast<const f32x4*>(&Sb[a_off + (s + 1) * 4 * BM]);
        bv[c ^ 1] = *reinterpret_cast<const f32x4*>(&Sb[b_off + (s + 1) * 4 * BN]);
        sc[c ^ 1] = aff[(s + 1) * 4];
        sh[c ^ 1] = aff[P.Kpad + (s + 1) * 4];
      }
      float b[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) b[t] = fmaxf(fmaf(bv[c][t], sc[c], sh[c]), lo);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = mfma16(av[c][m], b[t], acc[m][t]);
    }
  }

  // ---- epilogue: an accumulator quad is the 2x2 output patch of (co, pixel); a lane owns 4 consecutive pixels, i.e. 8
  // consecutive floats of two output rows when they lie in one image row: 16-byte stores then, 8-byte stores otherwise
  typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
  const DstD& D = P.dst;
  const long long qa = q0 + wn * 64 + l16 * 4;
  if (qa < QT) {
    const int na = (int)(qa / HW);
    const int pa = (int)(qa - (long long)na * HW);
    const int ha = pa / P.W, wa = pa - ha * P.W;
    const bool one_row = wa + 3 < P.W;   // (also inside the image: H*W % 4 need not hold here)
    if (one_row && qa + 3 < QT) {
      float* const o = D.p + (long long)na * D.ns + (long long)(2 * ha) * D.ws + 2 * wa;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int co = (m0 + wm * 64 + m * 16 + j * 4) >> 2;
        if (co < D.C) {
          const float bz = P.bias != nullptr ? P.bias[co] : 0.f;
          float* const oc = o + (long long)co * D.cs;
          *reinterpret_cast<f32x4u*>(oc) = f32x4{acc[m][0][0] + bz, acc[m][0][1] + bz, acc[m][1][0] + bz, acc[m][1][1] + bz};
          *reinterpret_cast<f32x4u*>(oc + 4) = f32x4{acc[m][2][0] + bz, acc[m][2][1] + bz, acc[m][3][0] + bz, acc[m][3][1] + bz};
          *reinterpret_cast<f32x4u*>(oc + D.ws) = f32x4{acc[m][0][2] + bz, acc[m][0][3] + bz, acc[m][1][2] + bz, acc[m][1][3] + bz};
          *reinterpret_cast<f32x4u*>(oc + D.ws + 4) = f32x4{acc[m][2][2] + bz, acc[m][2][3] + bz, acc[m][3][2] + bz, acc[m][3][3] + bz};
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const long long q = qa + t;
        if (q >= QT) continue;
        const int nn = (int)(q / HW);
        const int p = (int)(q - (long long)nn * HW);
        const int h = p / P.W, w = p - h * P.W;
        float* const o = D.p + (long long)nn * D.ns + (long long)(2 * h) * D.ws + 2 * w;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int co = (m0 + wm * 64 + m * 16 + j * 4) >> 2;
          if (co < D.C) {
            const float bz = P.bias != nullptr ? P.bias[co] : 0.f;
            float* const oc = o + (long long)co * D.cs;
            *reinterpret_cast<float2*>(oc) = make_float2(acc[m][t][0] + bz, acc[m][t][1] + bz);
            *reinterpret_cast<float2*>(oc + D.ws) = make_float2(acc[m][t][2] + bz, acc[m][t][3] + bz);
          }
        }
      }
    }
  }
}

// -------------------------------------------------------------------------------------------------------------------------
// DGRAD on the same LDS-DMA template:  dx[ci][q] = sum_{co,kh,kw} W[ci][co][kh][kw] * dy[co][2h+kh][2w+kw]  (q = low-res pixel).
// Block tile 128 (m = ci) x 128 pixels, K walked in chunks of 32 k-rows = 8 output channels x (kh, kw); weights as 16 one-KiB
// pieces of a [chunk][32][128] image with permuted columns (gsd_weight_layout mode 7: a lane's four A operands are one
// ds_read_b128).  The B operand is dy read with stride 2 in both directions -- as dword gathers that would be 64 DMA
// instructions per wave and chunk.  Instead an LDS row holds, for one (co, kh), the dy values of the tile's 128 low-res
// pixels with kw = 0 / 1 INTERLEAVED: 256 floats = 64 sixteen-byte pieces, a piece = two low-res pixels = four consecutive
// dy floats of row 2h+kh, one DMA instruction per row (16 per chunk, 4 per wave).  A lane reads the 8 floats of its four
// pixels (two ds_read_b128) once per PAIR of k-steps and feeds the even floats to the kw = 0 step, the odd ones to kw = 1.
// Piece p of a row is stored at slot p ^ ((p >> 4) & 1): lanes l and l + 8 of a read then hit different bank groups.
// Pixels are tiled over the flattened VIRTUAL grid q' = (n*H + h)*W' + w with W' = W rounded up to even, so a pair never
// straddles a row; the virtual last column of an odd-width plane (53, 213) is computed and not stored -- its piece reads two
// floats past the end of a dy row, i.e. the next row's, and past the tensor for the very last row: the caller vouches for
// 2 readable floats behind the tensor (gsd_src.slack) or gets the register-staged kernel above.
struct ConvTDgParams {
  SrcD src;           // dy (Cout, 2H, 2W), plain
  DstD dst;           // dx (Cin, H, W)
  const float* wt;    // mode 7: [mblock][Kpad][128], k = co*4 + kh*2 + kw
  int K, Kpad, M, nchunks, mblocks;
  int N, H, W, Wv, tiles_flat;
  // BW (gsd_convT2x2_dgrad_bnrelu): dst is the gradient buffer of the conv+BN+ReLU unit that produced the ConvT's input, whose raw
  // output bw_raw has dst's strides: dz = relu'(bn(raw)) * dx goes to dst, (sum dz, sum dz * xhat) per channel to `partials`
  // ([pixel tile * 2 + pixel half][2 * Mpad], Mpad = round_up(Cin, 64)) -- the reduce pass of that unit's BatchNorm backward disappears
  const float* bw_raw;
  const float* bw_scale;
  const float* bw_shift;
  const float* bw_mean;
  const float* bw_invstd;
  float* partials;
  int Mpad;
};

template <bool BW>
__global__ __launch_bounds__(256, 2) void convT_dgrad_dma_kernel(const ConvTDgParams P) {
  constexpr int BM = 128, BN = 128, KC = 32, MT = 4, NT = 4;
  constexpr int WIMG = KC * BM, XIMG = (KC / 2) * (2 * BN), BUF = WIMG + XIMG;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int j = lane >> 4, l16 = lane & 15;

  const int lid = xcd_swizzle(blockIdx.x, gridDim.x);   // the m-blocks of a pixel tile share the dy tile: one XCD's L2
  const int mb = lid % P.mblocks;
  const int pt = lid / P.mblocks;
  const int m0 = mb * BM;
  const long long q0 = (long long)pt * BN;              // first virtual pixel of the tile
  const int HWv = P.H * P.Wv;
  const long long QT = (long long)P.N * HWv;

  // ---- DMA lane geometry ---------------------------------------------------------------------------------------------------
  const float* const wsrc = P.wt + (size_t)mb * P.Kpad * BM + lane * 4;
  // dy: lane = physical slot of the row; it holds piece pi = slot ^ ((slot >> 4) & 1) = virtual pixels q0 + 2 pi, + 1
  const int pi = lane ^ ((lane >> 4) & 1);
  long long xo = 0;        // float offset of dy[n][.][2h][2w] inside a channel-0 plane set, for kh = 0
  bool xok;
  {
    const long long q = q0 + 2 * pi;
    xok = q < QT;
    const int nn = xok ? (int)(q / HWv) : 0;
    const int rem = xok ? (int)(q - (long long)nn * HWv) : 0;
    const int h = rem / P.Wv, w = rem - h * P.Wv;       // w even, w < W (w + 1 may be the virtual column)
    xo = (long long)nn * P.src.ns + (long long)(2 * h) * P.src.ws + 2 * w;
  }
  auto fill = [&](int chunk, int buf) {
    float* Wb = smem + buf * BUF;
    float* Xb = Wb + WIMG;
    const float* wc = wsrc + (size_t)chunk * WIMG;
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds(wc + (wave + 4 * i) * 256, Wb + (wave + 4 * i) * 256, 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = wave + 4 * i;                        // LDS row = (co_local = r >> 1, kh = r & 1)
      const int co = chunk * 8 + (r >> 1);
      const float* g = (xok && co < P.src.C) ? P.src.p + xo + (long long)co * P.src.cs + (long long)(r & 1) * P.src.ws
                                              : &gsd_zero16_ct[0];
      __builtin_amdgcn_global_load_lds(g, Xb + r * 256, 16, 0, 0);
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // A: k-row 8 p + 2 j + kw of the chunk image (pair p, sub-step kw); B: row 4 p + j = (co_local 2p + (j >> 1), kh = j & 1),
  // logical pieces 2 (wn*16 + l16) and + 1 of it (pixels wn*64 + 4 l16 .. + 3), stored swizzled
  const int a_off = (2 * j) * BM + wm * 64 + l16 * 4;
  const int lp = 2 * (wn * 16 + l16);
  const int b_off0 = WIMG + j * 256 + 4 * (lp ^ ((lp >> 4) & 1));
  const int b_off1 = WIMG + j * 256 + 4 * ((lp + 1) ^ (((lp + 1) >> 4) & 1));
  fill(0, 0);
  for (int chunk = 0; chunk < P.nchunks; ++chunk) {
    const int cur = chunk & 1;
    gsd_dma_barrier();   // this chunk has landed; everyone has left the other image
    if (chunk + 1 < P.nchunks) fill(chunk + 1, cur ^ 1);
    const float* Sb = smem + cur * BUF;
    f32x4 av[2], x0[2], x1[2];
    av[0] = *reinterpret_cast<const f32x4*>(&Sb[a_off]);
    x0[0] = *reinterpret_cast<const f32x4*>(&Sb[b_off0]);
    x1[0] = *reinterpret_cast<const f32x4*>(&Sb[b_off1]);
#pragma unroll
    for (int s = 0; s < KC / 4; ++s) {          // s = 2 p + kw
      const int p = s >> 1, kw = s & 1, c = s & 1, pc = p & 1;
      if (s + 1 < KC / 4) {   // the next k-step's operands fly during this one's MFMAs
        av[c ^ 1] = *reinterpret_cast<const f32x4*>(&Sb[a_off + ((s + 1) >> 1) * 8 * BM + ((s + 1) & 1) * BM]);
        if (kw == 1) {
          x0[pc ^ 1] = *reinterpret_cast<const f32x4*>(&Sb[b_off0 + (p + 1) * 4 * 256]);
          x1[pc ^ 1] = *reinterpret_cast<const f32x4*>(&Sb[b_off1 + (p + 1) * 4 * 256]);
        }
      }
      const float b[NT] = {x0[pc][kw], x0[pc][2 + kw], x1[pc][kw], x1[pc][2 + kw]};
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = mfma16(av[c][m], b[t], acc[m][t]);
    }
  }

  // ---- epilogue: acc[m][t][reg] = dx[ci = m0 + wm*64 + m*16 + 4j + reg][virtual pixel qa + t]; a lane owns 4 consecutive
  // virtual pixels: one 16-byte store per (m, reg) when they are 4 real pixels of one image row
  typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
  const DstD& D = P.dst;
  const long long qa = q0 + wn * 64 + l16 * 4;
  if constexpr (BW) {
    // the block's 128 channels' coefficients through LDS (the operand images are dead now), then per (m, reg): raw, mask, store, sums
    __syncthreads();
    float* sBw = smem;   // [4][128]: scale, shift, mean, invstd
    if (tid < BM) {
      const int c = m0 + tid < D.C ? m0 + tid : 0;
      sBw[tid] = P.bw_scale[c];
      sBw[BM + tid] = P.bw_shift[c];
      sBw[2 * BM + tid] = P.bw_mean[c];
      sBw[3 * BM + tid] = P.bw_invstd[c];
    }
    __syncthreads();
    // this lane's four virtual pixels: plane offset and validity (a pixel past the batch or in the virtual column is not stored)
    long long off[NT];
    bool ok[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const long long q = qa + t;
      ok[t] = q < QT;
      const int nn = ok[t] ? (int)(q / HWv) : 0;
      const int r = ok[t] ? (int)(q - (long long)nn * HWv) : 0;
      const int h = r / P.Wv, w = r - h * P.Wv;
      ok[t] = ok[t] && w < P.W;
      off[t] = (long long)nn * D.ns + (long long)h * D.ws + w;
    }
    const bool row4 = ok[0] && ok[3] && off[3] == off[0] + 3;   // four real pixels of one image row: 16-byte accesses
    float* const prow = P.partials + (size_t)(pt * 2 + wn) * (2 * P.Mpad);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int cl = wm * 64 + m * 16 + j * 4 + reg, ci = m0 + cl;
        const bool c_ok = ci < D.C;
        const long long cp = (long long)(c_ok ? ci : 0) * D.cs;
        const float bsc = sBw[cl], bsh = sBw[BM + cl], bmu = sBw[2 * BM + cl], bis = sBw[3 * BM + cl];
        float x[NT], dz[NT];
        if (row4) {
          const f32x4 tv = *reinterpret_cast<const f32x4u*>(P.bw_raw + cp + off[0]);
          x[0] = tv[0], x[1] = tv[1], x[2] = tv[2], x[3] = tv[3];
        } else {
#pragma unroll
          for (int t = 0; t < NT; ++t) x[t] = ok[t] ? P.bw_raw[cp + off[t]] : 0.f;
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          dz[t] = (c_ok && ok[t] && fmaf(x[t], bsc, bsh) > 0.f) ? acc[m][t][reg] : 0.f;
          s1 += dz[t];
          s2 = fmaf(dz[t], (x[t] - bmu) * bis, s2);
        }
        if (c_ok) {
          if (row4) {
            *reinterpret_cast<f32x4u*>(D.p + cp + off[0]) = f32x4{dz[0], dz[1], dz[2], dz[3]};
          } else {
#pragma unroll
            for (int t = 0; t < NT; ++t)
              if (ok[t]) D.p[cp + off[t]] = dz[t];
          }
        }
        s1 = reduce16_to_lane15(s1);
        s2 = reduce16_to_lane15(s2);
        if (l16 == 15 && ci < P.Mpad) {
          prow[ci] = s1;
          prow[P.Mpad + ci] = s2;
        }
      }
    return;
  }
  if (qa < QT) {
    const int na = (int)(qa / HWv);
    const int ra = (int)(qa - (long long)na * HWv);
    const int ha = ra / P.Wv, wa = ra - ha * P.Wv;
    if (wa + 3 < P.W) {
      float* const o = D.p + (long long)na * D.ns + (long long)ha * D.ws + wa;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int ci = m0 + wm * 64 + m * 16 + j * 4 + reg;
          if (ci < D.C)
            *reinterpret_cast<f32x4u*>(o + (long long)ci * D.cs) = f32x4{acc[m][0][reg], acc[m][1][reg], acc[m][2][reg], acc[m][3][reg]};
        }
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const long long q = qa + t;
        if (q >= QT) continue;
        const int nn = (int)(q / HWv);
        const int r = (int)(q - (long long)nn * HWv);
        const int h = r / P.Wv, w = r - h * P.Wv;
        if (w >= P.W) continue;                       // the virtual column
        float* const o = D.p + (long long)nn * D.ns + (long long)h * D.ws + w;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int ci = m0 + wm * 64 + m * 16 + j * 4 + reg;
            if (ci < D.C) o[(long long)ci * D.cs] = acc[m][t][reg];
          }
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
  GSD_REQUIRE(((uintptr_t)wt & 15) == 0, GSD_ERR_BAD_ARG, "gsd_convT2x2: the weight image must be 16-byte aligned");
  ConvTFwdParams P;
  P.src = to_srcd(*src);
  P.dst = to_dstd(*dst);
  P.wt = wt;
  P.bias = bias;
  P.K = Cin;
  P.Kpad = round_up(Cin, 32);
  P.M = Cout * 4;
  P.nchunks = P.Kpad / 32;
  P.mblocks = ceil_div(P.M, 128);
  P.N = N; P.H = H; P.W = W;
  P.tiles_flat = (int)ceil_div64((int64_t)N * H * W, 128);   // pixel tiles over the whole batch
  const long grid = (long)P.tiles_flat * P.mblocks;
  GSD_REQUIRE(grid < 2147483647L, GSD_ERR_UNSUPPORTED, "gsd_convT2x2: grid too large");
  const size_t lds = (size_t)(2 * (32 * 128 + 32 * 128) + 2 * P.Kpad) * sizeof(float);   // 64 KiB + coefficients: two blocks per CU
  GSD_REQUIRE(lds <= 80 * 1024, GSD_ERR_UNSUPPORTED, "gsd_convT2x2: Cin %d too large for the coefficient table", Cin);
  // 16-byte activation pieces: channel planes start 16-byte aligned and hold a multiple of 4 pixels
  const bool x4 = gsd_env_int("GSD_CONVT_X4", 1) != 0 && ((uintptr_t)src->ptr & 15) == 0 && src->c_stride % 4 == 0 &&
                  src->n_stride % 4 == 0 && (H * W) % 4 == 0;
  static gsd_attr_once big_lds[2];   // per-device caches of an idempotent launch attribute (gsd_common.h)
  const void* fn = x4 ? reinterpret_cast<const void*>(&convT_fwd_dma_kernel<true>) : reinterpret_cast<const void*>(&convT_fwd_dma_kernel<false>);
  if (hipError_t e = gsd_allow_big_lds(big_lds[x4 ? 1 : 0], fn); e != hipSuccess) {
    gsd_set_error("gsd_convT2x2: hipFuncSetAttribute: %s", hipGetErrorString(e));
    return GSD_ERR_HIP;
  }
  if (x4)
    hipLaunchKernelGGL(convT_fwd_dma_kernel<true>, dim3((int)grid), dim3(256), lds, (hipStream_t)stream, P);
  else
    hipLaunchKernelGGL(convT_fwd_dma_kernel<false>, dim3((int)grid), dim3(256), lds, (hipStream_t)stream, P);
  GSD_LAUNCH_CHECK("gsd_convT2x2");
  return GSD_OK;
}

// Which gsd_weight_layout mode gsd_convT2x2_dgrad expects for these arguments: 7 (the LDS-DMA kernel) unless an odd-width
// plane comes without the two readable floats behind the tensor that its virtual last column reads (gsd_src.slack), or the
// tuning switch GSD_CONVT_DG_DMA=0 asks for the register-staged kernel: then 3.
extern "C" int gsd_convT2x2_dgrad_layout(const gsd_src* src, int Cin, int Cout, int N, int H, int W) {
  if (src == nullptr || Cin <= 0 || Cout <= 0 || N <= 0 || H <= 0 || W <= 0) return 3;
  if (gsd_env_int("GSD_CONVT_DG_DMA", 1) == 0) return 3;
  if ((W & 1) && src->slack < 2) return 3;
  if ((int64_t)N * src->n_stride >= (1LL << 40)) return 3;
  return 7;
}

extern "C" int gsd_convT2x2_dgrad(const gsd_src* src, const float* wt, int Cin, int Cout, const gsd_dst* dst, int N,
                                  int H, int W, void* stream) {
  return gsd_convT2x2_dgrad_as(gsd_convT2x2_dgrad_layout(src, Cin, Cout, N, H, W), src, wt, Cin, Cout, dst, N, H, W, stream);
}

// The same with the layout of `wt` stated by the caller (the mode it passed to gsd_weight_layout: 3 or 7) instead of re-derived
// at launch: a caller that built its weight image once (the engine does, per buffer shape) cannot be handed the other kernel
// by an environment switch or a different `slack` at launch time -- a mode the arguments do not admit is refused.
extern "C" int gsd_convT2x2_dgrad_as(int wt_mode, const gsd_src* src, const float* wt, int Cin, int Cout, const gsd_dst* dst, int N,
                                     int H, int W, void* stream) {
  GSD_REQUIRE(src && dst && wt, GSD_ERR_BAD_ARG, "gsd_convT2x2_dgrad: null argument");
  GSD_REQUIRE(wt_mode == 3 || wt_mode == 7, GSD_ERR_BAD_ARG, "gsd_convT2x2_dgrad: weight layout mode %d is neither 3 nor 7", wt_mode);
  if (wt_mode == 7)
    GSD_REQUIRE(!((W & 1) && src->slack < 2) && (int64_t)N * src->n_stride < (1LL << 40), GSD_ERR_BAD_ARG,
                "gsd_convT2x2_dgrad: a mode-7 weight image needs the LDS-DMA kernel, which these arguments do not admit "
                "(odd W = %d needs src->slack >= 2, got %d); build the image with the mode gsd_convT2x2_dgrad_layout returns",
                W, src->slack);
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
  if (wt_mode == 7) {
    GSD_REQUIRE(((uintptr_t)wt & 15) == 0, GSD_ERR_BAD_ARG, "gsd_convT2x2_dgrad: the weight image must be 16-byte aligned");
    ConvTDgParams Q;
    Q.src = to_srcd(*src);
    Q.dst = to_dstd(*dst);
    Q.wt = wt;
    Q.K = Cout * 4;
    Q.Kpad = round_up(Cout, 8) * 4;
    Q.M = Cin;
    Q.nchunks = Q.Kpad / 32;
    Q.mblocks = ceil_div(Cin, 128);
    Q.N = N; Q.H = H; Q.W = W;
    Q.Wv = round_up(W, 2);
    Q.tiles_flat = (int)ceil_div64((int64_t)N * H * Q.Wv, 128);   // pixel tiles over the whole batch's virtual grid
    const long grid = (long)Q.tiles_flat * Q.mblocks;
    GSD_REQUIRE(grid < 2147483647L, GSD_ERR_UNSUPPORTED, "gsd_convT2x2_dgrad: grid too large");
    const size_t lds = (size_t)(2 * (32 * 128 + 16 * 256)) * sizeof(float);   // 64 KiB: two blocks per CU
    static gsd_attr_once big_lds;   // per-device cache of an idempotent launch attribute (gsd_common.h)
    Q.bw_raw = Q.bw_scale = Q.bw_shift = Q.bw_mean = Q.bw_invstd = nullptr;
    Q.partials = nullptr;
    Q.Mpad = 0;
    if (hipError_t e = gsd_allow_big_lds(big_lds, reinterpret_cast<const void*>(&convT_dgrad_dma_kernel<false>)); e != hipSuccess) {
      gsd_set_error("gsd_convT2x2_dgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return GSD_ERR_HIP;
    }
    hipLaunchKernelGGL(convT_dgrad_dma_kernel<false>, dim3((int)grid), dim3(256), lds, (hipStream_t)stream, Q);
    GSD_LAUNCH_CHECK("gsd_convT2x2_dgrad");
    return GSD_OK;
  }
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

// dX of the transposed convolution fused with the backward of the relu(bn(raw)) that produced its INPUT (the ConvT reads the
// activated output of the conv unit below: unet.py:41 after :12-13 / :15-16): dst receives dz = dx * [raw*scale+shift > 0], the
// partials (sum dz, sum dz*xhat) per channel in rows of 2*Mpad floats, Mpad = round_up(Cin, 64) -- as gsd_conv3x3_dgrad_bnrelu,
// so the separate gsd_bn_bwd_reduce pass over that unit (one read of g and raw, one write of g) disappears.  The LDS-DMA kernel
// only (weight layout mode 7); gsd_convT2x2_dgrad_bnrelu_partial_rows: 0 when the arguments do not admit it.
static bool ct_dma_admits(const gsd_src* src, int Cin, int Cout, int N, int H, int W) {
  if (src == nullptr || Cin <= 0 || Cout <= 0 || N <= 0 || H <= 0 || W <= 0) return false;
  return !((W & 1) && src->slack < 2) && (int64_t)N * src->n_stride < (1LL << 40);
}

extern "C" int gsd_convT2x2_dgrad_bnrelu_partial_rows(const gsd_src* src, int Cin, int Cout, int N, int H, int W) {
  if (!ct_dma_admits(src, Cin, Cout, N, H, W)) return 0;
  return (int)ceil_div64((int64_t)N * H * round_up(W, 2), 128) * 2;
}

extern "C" int gsd_convT2x2_dgrad_bnrelu(const gsd_src* src, const float* wt, int Cin, int Cout, const gsd_dst* dst, const float* raw,
                                         const float* scale, const float* shift, const float* mean, const float* invstd,
                                         float* partials, int N, int H, int W, void* stream) {
  GSD_REQUIRE(src && dst && wt && raw && scale && shift && mean && invstd && partials, GSD_ERR_BAD_ARG,
              "gsd_convT2x2_dgrad_bnrelu: null argument");
  GSD_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, GSD_ERR_BAD_ARG, "gsd_convT2x2_dgrad_bnrelu: bad sizes");
  GSD_REQUIRE(ct_dma_admits(src, Cin, Cout, N, H, W), GSD_ERR_UNSUPPORTED,
              "gsd_convT2x2_dgrad_bnrelu: the arguments do not admit the LDS-DMA kernel (gsd_convT2x2_dgrad_bnrelu_partial_rows == 0)");
  if (int e = gsd_check_src(*src, "gsd_convT2x2_dgrad_bnrelu src")) return e;
  if (int e = gsd_check_dst(*dst, "gsd_convT2x2_dgrad_bnrelu dst")) return e;
  GSD_REQUIRE(src->C == Cout && src->H == 2 * H && src->W == 2 * W && src->scale == nullptr && src->relu == 0 && src->off_h == 0 &&
                  src->off_w == 0,
              GSD_ERR_BAD_ARG, "gsd_convT2x2_dgrad_bnrelu: src must be the plain (Cout,2H,2W) gradient");
  GSD_REQUIRE(((uintptr_t)src->ptr & 7) == 0 && (src->c_stride & 1) == 0 && (src->n_stride & 1) == 0, GSD_ERR_UNSUPPORTED,
              "gsd_convT2x2_dgrad_bnrelu: src must be 8-byte aligned with even strides");
  GSD_REQUIRE(dst->C == Cin && dst->H == H && dst->W == W && dst->off_h == 0 && dst->off_w == 0, GSD_ERR_BAD_ARG,
              "gsd_convT2x2_dgrad_bnrelu: dst must be the full (Cin,H,W) gradient buffer (raw shares its strides)");
  GSD_REQUIRE(((uintptr_t)wt & 15) == 0, GSD_ERR_BAD_ARG, "gsd_convT2x2_dgrad_bnrelu: the weight image must be 16-byte aligned");
  ConvTDgParams Q;
  Q.src = to_srcd(*src);
  Q.dst = to_dstd(*dst);
  Q.wt = wt;
  Q.K = Cout * 4;
  Q.Kpad = round_up(Cout, 8) * 4;
  Q.M = Cin;
  Q.nchunks = Q.Kpad / 32;
  Q.mblocks = ceil_div(Cin, 128);
  Q.N = N; Q.H = H; Q.W = W;
  Q.Wv = round_up(W, 2);
  Q.tiles_flat = (int)ceil_div64((int64_t)N * H * Q.Wv, 128);
  Q.bw_raw = raw; Q.bw_scale = scale; Q.bw_shift = shift; Q.bw_mean = mean; Q.bw_invstd = invstd;
  Q.partials = partials;
  Q.Mpad = round_up(Cin, 64);
  const long grid = (long)Q.tiles_flat * Q.mblocks;
  GSD_REQUIRE(grid < 2147483647L, GSD_ERR_UNSUPPORTED, "gsd_convT2x2_dgrad_bnrelu: grid too large");
  const size_t lds = (size_t)(2 * (32 * 128 + 16 * 256)) * sizeof(float);
  static gsd_attr_once big_lds;
  if (hipError_t e = gsd_allow_big_lds(big_lds, reinterpret_cast<const void*>(&convT_dgrad_dma_kernel<true>)); e != hipSuccess) {
    gsd_set_error("gsd_convT2x2_dgrad_bnrelu: hipFuncSetAttribute: %s", hipGetErrorString(e));
    return GSD_ERR_HIP;
  }
  hipLaunchKernelGGL(convT_dgrad_dma_kernel<true>, dim3((int)grid), dim3(256), lds, (hipStream_t)stream, Q);
  GSD_LAUNCH_CHECK("gsd_convT2x2_dgrad_bnrelu");
  return GSD_OK;
}
