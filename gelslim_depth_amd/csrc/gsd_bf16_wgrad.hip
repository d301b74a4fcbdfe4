// gsd_bf16_wgrad.hip -- weight gradients of the bf16 path on v_mfma_f32_16x16x32_bf16, reduction over PIXELS.
//
//   D[t][m][n] = sum over pixels p of  A[p][m] * B[stride*p + tap_t][n]            (fp32 accumulation, fp32 result)
//
//   conv3x3   A = dy (m = co), B = the layer's input activation (n = ci), 9 taps at stride 1  -> dW[co][ci][t]
//             (the dW half of aten::convolution_backward for unet.py:11,14)
//   first     A = dy, B = the im2col'd input (n = ci*9+t), 1 tap                              -> dW[co][ci*9+t]
//   convT     A = x (m = ci), B = gradient of the upsampled tensor (n = co), 4 taps at stride 2 -> dW[ci][co][kh][kw]
//             (unet.py:36)
//
// Both operands are NHWC (channels contiguous) but the MFMA wants 8 consecutive k = PIXELS per lane, so both tiles are
// staged pixel-major in LDS ([pixel][channels], filled by global_load_lds_dwordx4) and read back transposed with
// ds_read_b64_tr_b16.  A lane's 8 k-values are pixels {4g..4g+3} and {16+4g..16+4g+3} of a 32-pixel step (g = lane>>4):
// one 32-lane half then reads 8 CONSECUTIVE pixel rows per instruction, and with a row stride of (32 B x odd) those 8
// rows fall on disjoint banks for every tap shift -- a linear layout, so every read is base + immediate.  (With the
// natural k = 8g..8g+7 a half reads rows {0-3, 8-11}; no linear stride separates rows q and q+8.)
//
// Wave tile: 64 m x 16 n x T taps (36 accumulator tiles for 3x3); block = 4 waves: 128 m x 32 n, or 64 m x 64 n when
// M <= 64.  One LDS image per block, two blocks per CU (the other block's MFMAs cover this block's DMA), split-K over
// pixel tiles with fp32 slabs summed in a fixed order by gwgrad_reduce_kernel => bitwise reproducible.
#include "gsd_bf16_common.h"

#include <cstdlib>

__device__ const uint4 gsd_zero16w[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};

struct GWgradP {
  const u16* a;
  long long a_pitch;
  const u16* b;
  long long b_pitch;
  int Hb, Wb;      // extent of B's buffer
  float* slabs;
  int N, H, W;     // pixel grid of the reduction == extent of A's buffer
  int M, Ncols;
  int T, stride;
  int ty[9], tx[9];
  int TH, TW, tiles_y, tiles_x, HC, HP, KS, NPIX;
  int stages_total, splits, mblocks, nblocks, xcd;
};

__device__ __forceinline__ u32x2 tr_read_b64(const unsigned char* p) {
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
  return __builtin_bit_cast(u32x2, v);
}

template <int HALO, int TT, int WM, int WN>
__global__ __launch_bounds__(256, 2) void gwgrad_bf16_kernel(const GWgradP P) {
  constexpr int BM = WM * 64, BNC = WN * 16;
  constexpr int RSA = BM * 2 + 32, RSB = BNC * 2 + 32;       // row strides: 32 B x odd
  constexpr int KS = HALO ? 4 : 2;                            // 32-pixel k-steps per stage (NPIX = 128 / 64)
  constexpr int MT = 4;
  constexpr int MAXA = (128 * RSA + 4095) / 4096;             // A DMA instructions per wave (NPIX <= 128)
  constexpr int MAXB = (264 * RSB + 4095) / 4096;             // B DMA instructions per wave (<= 264 rows)

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Al = smem;
  unsigned char* Bl = smem + P.NPIX * RSA;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int g = lane >> 4, li = lane & 15, q = li >> 2, p4 = li & 3;

  const int per_split = P.mblocks * P.nblocks;
  // XCD-aware order: the (m-block, n-block) pairs of one split read the same dy / activation pixel range -- logical ids are
  // contiguous per XCD (the hardware deals blocks round-robin over the 8 XCDs, each with its own L2), so one L2 serves them
  const int lid = P.xcd ? xcd_swizzle(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int split = lid / per_split;
  const int rem = lid - split * per_split;
  const int mb = rem % P.mblocks, nb = rem / P.mblocks;
  const int m0 = mb * BM, n0 = nb * BNC;
  const int s_begin = (int)((long long)split * P.stages_total / P.splits);
  const int s_end = (int)((long long)(split + 1) * P.stages_total / P.splits);
  const int nbrows = HALO ? P.HP : TT * P.NPIX;

  // ---- DMA bookkeeping: (row, piece) of every 16-byte piece this lane moves; stage-invariant ------------------------
  // A: row = tile pixel (r, c); packed (r << 20 | c << 8 | piece), -1 = no transfer
  int apk[MAXA];
#pragma unroll
  for (int k = 0; k < MAXA; ++k) {
    const int o = (k * 4 + wave) * 1024 + lane * 16;
    const int row = o / RSA, piece = (o - row * RSA) >> 4;
    int v = -1;
    if (row < P.NPIX && piece < BM / 8) {
      const int r = row / P.TW, c = row - r * P.TW;
      v = (m0 + piece * 8 < P.M) ? ((r << 20) | (c << 8) | piece) : -2;   // -2: zeros (channel beyond M)
    }
    apk[k] = v;
  }
  // B: HALO -> row = halo position (hy, hx); dense -> row = (tap, tile pixel); packed (t << 28 | y << 20 | x << 8 | piece)
  int bpk[MAXB];
#pragma unroll
  for (int k = 0; k < MAXB; ++k) {
    const int o = (k * 4 + wave) * 1024 + lane * 16;
    const int row = o / RSB, piece = (o - row * RSB) >> 4;
    int v = -1;
    if (row < nbrows && piece < BNC / 8) {
      if (n0 + piece * 8 >= P.Ncols) {
        v = -2;
      } else if (HALO) {
        const int hy = row / P.HC, hx = row - hy * P.HC;
        v = (hy << 20) | (hx << 8) | piece;
      } else {
        const int t = row / P.NPIX, qq = row - t * P.NPIX;
        const int r = qq / P.TW, c = qq - r * P.TW;
        v = (t << 28) | (r << 20) | (c << 8) | piece;
      }
    }
    bpk[k] = v;
  }

  // ---- transposed-read offsets -----------------------------------------------------------------------------------
  const int a_rd = (4 * g + q) * RSA + (wm * 64 + 4 * p4) * 2;   // + (k0 [+16]) * RSA + mt * 32
  const int b_rd = (4 * g + q) * RSB + (wn * 16 + 4 * p4) * 2;   // + (brow [+16]) * RSB

  f32x4 acc[MT][TT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < TT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int tpi = P.tiles_y * P.tiles_x;
  for (int stage = s_begin; stage < s_end; ++stage) {
    const int n = stage / tpi;
    const int trem = stage - n * tpi;
    const int tyi = trem / P.tiles_x;
    const int h0 = tyi * P.TH, w0 = (trem - tyi * P.tiles_x) * P.TW;
    const u16* a_img = P.a + (long long)n * P.H * P.W * P.a_pitch + m0;
    const u16* b_img = P.b + (long long)n * P.Hb * P.Wb * P.b_pitch + n0;
    __syncthreads();   // every wave has finished reading the previous stage's images
#pragma unroll
    for (int k = 0; k < MAXA; ++k) {
      if (apk[k] != -1) {
        const void* s = (const void*)gsd_zero16w;
        if (apk[k] >= 0) {
          const int h = h0 + (apk[k] >> 20), w = w0 + ((apk[k] >> 8) & 0xfff);
          if (h < P.H && w < P.W) s = (const void*)(a_img + (long long)(h * P.W + w) * P.a_pitch + (apk[k] & 0xff) * 8);
        }
        __builtin_amdgcn_global_load_lds(s, Al + (k * 4 + wave) * 1024, 16, 0, 0);
      }
    }
#pragma unroll
    for (int k = 0; k < MAXB; ++k) {
      if (bpk[k] != -1) {
        const void* s = (const void*)gsd_zero16w;
        if (bpk[k] >= 0) {
          int hi, wi;
          bool ok = true;
          if (HALO) {
            hi = h0 - 1 + ((bpk[k] >> 20) & 0xff);
            wi = w0 - 1 + ((bpk[k] >> 8) & 0xfff);
          } else {
            const int t = (bpk[k] >> 28) & 7;
            const int h = h0 + ((bpk[k] >> 20) & 0xff), w = w0 + ((bpk[k] >> 8) & 0xfff);
            ok = h < P.H && w < P.W;      // the pixel itself is outside the reduction grid: A is zero there anyway
            hi = P.stride * h + P.ty[t];
            wi = P.stride * w + P.tx[t];
          }
          if (ok && (unsigned)hi < (unsigned)P.Hb && (unsigned)wi < (unsigned)P.Wb)
            s = (const void*)(b_img + (long long)(hi * P.Wb + wi) * P.b_pitch + (bpk[k] & 0xff) * 8);
        }
        __builtin_amdgcn_global_load_lds(s, Bl + (k * 4 + wave) * 1024, 16, 0, 0);
      }
    }
    gsd_dma_barrier();   // vmcnt(0) + barrier: both images have landed
    // One flat, fully unrolled sequence of KS*TT steps (4 MFMAs each).  The B operand of step s+2 and, two steps before a
    // k-step ends, the A operands of the next k-step are read while step s multiplies: every ds_read_b64_tr_b16 has at
    // least one whole step (64 MFMA cycles, plus whatever the CU's other block interleaves) to land.
    auto read_a = [&](int ks, u32x4* a) {
      const unsigned char* ap = Al + a_rd + ks * 32 * RSA;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const u32x2 lo = tr_read_b64(ap + m * 32), hi = tr_read_b64(ap + m * 32 + 16 * RSA);
        a[m] = u32x4{lo[0], lo[1], hi[0], hi[1]};
      }
    };
    auto read_b = [&](int s) {
      const int ks = s / TT, t = s - ks * TT;
      const int q0 = ks * 32;
      const int r = q0 / P.TW, c = q0 - r * P.TW;
      const int brow = HALO ? (r + t / 3) * P.HC + c + t % 3 : t * P.NPIX + q0;
      const unsigned char* bp = Bl + b_rd + brow * RSB;
      const u32x2 lo = tr_read_b64(bp), hi = tr_read_b64(bp + 16 * RSB);
      return u32x4{lo[0], lo[1], hi[0], hi[1]};
    };
    u32x4 a[2][MT], b[3];
    read_a(0, a[0]);
    b[0] = read_b(0);
    if (KS * TT > 1) b[1] = read_b(1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < KS * TT; ++s) {
      const int ks = s / TT, t = s - ks * TT;
      const bool pre_b = s + 2 < KS * TT;
      const bool pre_a = t == (TT >= 2 ? TT - 2 : 0) && ks + 1 < KS;
      if (pre_b) b[(s + 2) % 3] = read_b(s + 2);
      if (pre_a) read_a(ks + 1, a[(ks + 1) & 1]);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m][t] = mfma_bf16(a[ks & 1][m], b[s % 3], acc[m][t]);
      if (pre_a) {          // 8 (+2) reads beside 4 MFMAs
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        }
      } else if (pre_b) {   // 2 reads beside 4 MFMAs
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- slab store: slab[split][t][m][n] ------------------------------------------------------------------------------
#pragma unroll
  for (int t = 0; t < TT; ++t) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int mr = m0 + wm * 64 + m * 16 + g * 4 + e;
        const int col = n0 + wn * 16 + li;
        if (mr < P.M && col < P.Ncols)
          P.slabs[(((size_t)split * TT + t) * P.M + mr) * P.Ncols + col] = acc[m][t][e];
      }
  }
}

// out[(m * NcOut + n) * T + t] = sum over splits of slab[split][t][m][n]   (n < NcOut <= Ncols)
__global__ void gwgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, int splits, int T, int M, int Ncols,
                                     int NcOut) {
  const long long per = (long long)T * M * Ncols;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < per; e += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(e % Ncols);
    const long long r2 = e / Ncols;
    const int m = (int)(r2 % M);
    const int t = (int)(r2 / M);
    if (n < NcOut) {
      float s = 0.f;
      for (int k = 0; k < splits; ++k) s += slabs[(size_t)k * per + e];
      out[((size_t)m * NcOut + n) * T + t] = s;
    }
  }
}

namespace {

struct WPlan {
  bool wide;
  int BM, BNC, NPIX, TH, TW, tiles_y, tiles_x, HC, HP, mblocks, nblocks, stages_total, splits;
  int64_t slab_elems;
  size_t lds;
};

WPlan make_wplan(bool halo, int T, int N, int H, int W, int M, int Ncols) {
  WPlan p;
  p.wide = M <= 64;
  p.BM = p.wide ? 64 : 128;
  p.BNC = p.wide ? 64 : 32;
  p.NPIX = halo ? 128 : 64;
  long best = -1;
  for (int tw = 32; tw <= 64; tw *= 2) {
    const int th = p.NPIX / tw;
    const long cost = (long)ceil_div(H, th) * ceil_div(W, tw);
    if (best < 0 || cost < best) {
      best = cost;
      p.TW = tw;
      p.TH = th;
    }
  }
  p.tiles_y = ceil_div(H, p.TH);
  p.tiles_x = ceil_div(W, p.TW);
  p.HC = p.TW + 2;
  p.HP = (p.TH + 2) * (p.TW + 2);
  p.mblocks = ceil_div(M, p.BM);
  p.nblocks = ceil_div(Ncols, p.BNC);
  p.stages_total = N * p.tiles_y * p.tiles_x;
  const int target = gsd_env_int("GSD_BF16_WGRAD_BLOCKS", 512);   // tuning knob
  int splits = ceil_div(target, p.mblocks * p.nblocks);   // default: ONE round of 2 resident blocks per CU x 256 CUs (measured: 512 -> 866 TFLOP/s, 1024 -> 766, 256 -> 647 over the layer set; fewer splits also halve the slab traffic)
  if (splits > p.stages_total) splits = p.stages_total;
  if (splits < 1) splits = 1;
  p.splits = splits;
  p.slab_elems = (int64_t)splits * T * M * Ncols;
  const int rsa = p.BM * 2 + 32, rsb = p.BNC * 2 + 32;
  p.lds = (size_t)p.NPIX * rsa + (size_t)(halo ? p.HP : T * p.NPIX) * rsb;
  p.lds = (p.lds + 1023) / 1024 * 1024 + 1024;   // DMA instructions are issued in whole 1 KiB pieces
  return p;
}

template <int HALO, int TT, int WM, int WN>
int launch_w(const GWgradP& P, int grid, size_t lds, hipStream_t st, const char* what) {
  static gsd_attr_once big_lds;   // per-device cache of an idempotent launch attribute (gsd_common.h)
  if (hipError_t e = gsd_allow_big_lds(big_lds, reinterpret_cast<const void*>(&gwgrad_bf16_kernel<HALO, TT, WM, WN>)); e != hipSuccess) {
    gsd_set_error("%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e));
    return GSD_ERR_HIP;
  }
  GSD_REQUIRE(lds <= 160 * 1024, GSD_ERR_UNSUPPORTED, "%s: LDS %zu B too large", what, lds);
  hipLaunchKernelGGL((gwgrad_bf16_kernel<HALO, TT, WM, WN>), dim3(grid), dim3(256), lds, st, P);
  GSD_LAUNCH_CHECK(what);
  return GSD_OK;
}

}  // namespace

extern "C" int64_t gsd_bf16_wgrad_workspace(int ntaps, int N, int H, int W, int M, int Ncols) {
  if (ntaps < 1 || ntaps > 9 || N <= 0 || H <= 0 || W <= 0 || M <= 0 || Ncols <= 0) return 0;
  return make_wplan(ntaps == 9, ntaps, N, H, W, M, Ncols).slab_elems;
}

extern "C" int gsd_bf16_wgrad(const gsd_nhwc* a, const gsd_nhwc* b, int ntaps, int stride, const int* ty, const int* tx,
                              float* dw, int ncols_out, float* workspace, int64_t workspace_elems, void* stream) {
  if (int e = gsd_check_nhwc(a, "gsd_bf16_wgrad a")) return e;
  if (int e = gsd_check_nhwc(b, "gsd_bf16_wgrad b")) return e;
  GSD_REQUIRE(dw && workspace && ty && tx, GSD_ERR_BAD_ARG, "gsd_bf16_wgrad: null argument");
  GSD_REQUIRE(a->N == b->N && a->C % 8 == 0 && b->C % 8 == 0, GSD_ERR_BAD_ARG, "gsd_bf16_wgrad: batch sizes differ or C %% 8 != 0");
  GSD_REQUIRE((ntaps == 9 && stride == 1) || ((ntaps == 1 || ntaps == 4) && (stride == 1 || stride == 2)), GSD_ERR_UNSUPPORTED,
              "gsd_bf16_wgrad: 9 taps at stride 1 (3x3 halo) or 1 / 4 dense taps at stride 1 / 2");
  GSD_REQUIRE(ncols_out > 0 && ncols_out <= b->C, GSD_ERR_BAD_ARG, "gsd_bf16_wgrad: ncols_out out of range");
  GSD_REQUIRE(a->H < 256 * 128 && a->W < 4096, GSD_ERR_UNSUPPORTED, "gsd_bf16_wgrad: extent too large");
  const bool halo = ntaps == 9;
  if (halo) {
    GSD_REQUIRE(a->H == b->H && a->W == b->W, GSD_ERR_BAD_ARG, "gsd_bf16_wgrad: 3x3 needs equal extents");
    for (int t = 0; t < 9; ++t)
      GSD_REQUIRE(ty[t] == t / 3 - 1 && tx[t] == t % 3 - 1, GSD_ERR_UNSUPPORTED, "gsd_bf16_wgrad: 9 taps must be the 3x3 p1 stencil");
  }
  const int M = a->C, Ncols = b->C;
  const WPlan pl = make_wplan(halo, ntaps, a->N, a->H, a->W, M, Ncols);
  GSD_REQUIRE(workspace_elems >= pl.slab_elems, GSD_ERR_WORKSPACE, "gsd_bf16_wgrad: workspace %lld < %lld elements",
              (long long)workspace_elems, (long long)pl.slab_elems);
  GWgradP P;
  P.a = (const u16*)a->ptr; P.a_pitch = a->pitch;
  P.b = (const u16*)b->ptr; P.b_pitch = b->pitch; P.Hb = b->H; P.Wb = b->W;
  P.slabs = workspace;
  P.N = a->N; P.H = a->H; P.W = a->W;
  P.M = M; P.Ncols = Ncols;
  P.T = ntaps; P.stride = stride;
  for (int t = 0; t < 9; ++t) { P.ty[t] = t < ntaps ? ty[t] : 0; P.tx[t] = t < ntaps ? tx[t] : 0; }
  P.TH = pl.TH; P.TW = pl.TW; P.tiles_y = pl.tiles_y; P.tiles_x = pl.tiles_x; P.HC = pl.HC; P.HP = pl.HP;
  P.NPIX = pl.NPIX; P.KS = pl.NPIX / 32;
  P.stages_total = pl.stages_total; P.splits = pl.splits; P.mblocks = pl.mblocks; P.nblocks = pl.nblocks;
  P.xcd = gsd_env_int("GSD_BF16_XCD", 1) != 0 ? 1 : 0;
  const int grid = pl.splits * pl.mblocks * pl.nblocks;
  int rc;
  hipStream_t st = (hipStream_t)stream;
  if (halo) rc = pl.wide ? launch_w<1, 9, 1, 4>(P, grid, pl.lds, st, "gsd_bf16_wgrad") : launch_w<1, 9, 2, 2>(P, grid, pl.lds, st, "gsd_bf16_wgrad");
  else if (ntaps == 4) rc = pl.wide ? launch_w<0, 4, 1, 4>(P, grid, pl.lds, st, "gsd_bf16_wgrad") : launch_w<0, 4, 2, 2>(P, grid, pl.lds, st, "gsd_bf16_wgrad");
  else rc = pl.wide ? launch_w<0, 1, 1, 4>(P, grid, pl.lds, st, "gsd_bf16_wgrad") : launch_w<0, 1, 2, 2>(P, grid, pl.lds, st, "gsd_bf16_wgrad");
  if (rc) return rc;
  const long long per = (long long)ntaps * M * Ncols;
  const int rgrid = (int)(ceil_div64(per, 256) < 4096 ? ceil_div64(per, 256) : 4096);
  hipLaunchKernelGGL(gwgrad_reduce_kernel, dim3(rgrid), dim3(256), 0, (hipStream_t)stream, workspace, dw, pl.splits, ntaps, M, Ncols,
                     ncols_out);
  GSD_LAUNCH_CHECK("gsd_bf16_wgrad reduce");
  return GSD_OK;
}
