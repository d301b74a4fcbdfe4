// gsd_bf16_c64.hip -- conv3x3 64 -> 64 channels at full resolution with the weights RESIDENT in LDS (bf16 path).
//
// The 64-channel 3x3 convolutions of the first level (unet.py:14 for `inc` and `up.3.conv`: K = M = 64 at 320 x 427, forward and
// dX: four of a step's 34 conv3x3 launches) are where the DMA-filled kernel (gsd_bf16_conv.hip) is weakest -- 0.285 of the matrix
// pipes busy against 0.446 on the 128-row tile (profiles/r04_f_bf16_pmc_sq_summary.txt): with K = 64 an item is only six
// barrier-separated iterations of 96 MFMAs, each carrying the fills of the next, under an epilogue as long as the K loop.  Their
// weights are 9 x 64 x 64 bf16 = 72 KiB: they fit LDS whole.  So, as in gsd_bf16_inc.hip (whose K loop runs at 18.8 cycles per
// MFMA and SIMD against the DMA-filled kernel's 26):
//
//   * one persistent block of EIGHT waves per CU (two per SIMD: the epilogue is vector work, and a lone wave issues a vector
//     instruction every ~4 cycles) loads the weights once;
//   * per 8 x 64 pixel tile the 10 x 66 halo of the input (64 channels = 128 B a pixel, 83 KiB) comes by LDS-DMA in ONE fill
//     -- issued right behind the previous tile's K loop, so it flies under that tile's epilogue -- and the 18 k-steps (2 channel
//     chunks x 9 taps) run with no fill and no barrier inside;
//   * epilogues: raw output (+ BatchNorm partial sums: forward; without: a plain dX); fused pass 1 of the BatchNorm + ReLU
//     backward of the unit below with its raw output read from HBM (dX, as gsd_bf16_conv3x3's `bw`) -- those loads are issued in
//     FRONT of the next tile's fill: vector-memory operations return in issue order.
// Measured at batch 32 (profiles/r04_c64_vs_dma_kernel.txt): forward 0.415 ms against the DMA kernel's 0.542, dX with the fused
// epilogue 0.609 against 0.677 (it moves 1.84 GB: 0.46 ms at the 4 TB/s a mixed stream reaches).  A third epilogue that recomputed
// the raw output of inc's first convolution from x (to fold gsd_bf16_first_bn_bwd_reduce into inc's second dX) was built,
// bit-identical, and measured 0.80 ms against 0.40 (plain dX here) + 0.20 (the separate pass): removed.
//
// Same products in the same order through the same MFMA as gsd_bf16_conv3x3: outputs bit-identical.
// LDS images as in gsd_bf16_inc.hip: activations [chunk][halo pixel q = r*66 + c][64 B], piece p at slot p ^ (2*((c>>2)&1))
// (bank-conflict free for every tap shift); weights [tap][64 rows][128 B], piece p of row i at slot p ^ (i & 6).
#include "gsd_bf16_common.h"

#include <type_traits>


namespace {

constexpr int C_TW = 64, C_M = 64;
constexpr int C_HC = C_TW + 2;                                         // halo row pitch (pixels)
constexpr int C_W_BYTES = 9 * C_M * 128;                               // 72 KiB
constexpr int C_COEF_BYTES = 4 * C_M * 4;                              // scale | shift | mean | invstd of the unit below

// TH = 8 rows a tile (10 x 66 halo = 82.5 KiB)
template <int TH>
struct C64Geo {
  static constexpr int TR = TH / 2;
  static constexpr int HR = TH + 2, NPH = HR * C_HC;
  static constexpr int ACT_PLANE = NPH * 64;                           // one 32-channel chunk
  static constexpr int SLOTS = 2 * NPH * 4;                            // 16-byte slots of the activation image
  static constexpr int ACT_BYTES = SLOTS * 16;
  static constexpr int NFILL = (SLOTS + 63) / 64;                      // DMA instructions of one fill (the last may be half a wave)
  static constexpr int NF_W = (NFILL + 7) / 8;                         // per wave (the last may not exist)
  static constexpr int LDS_PLAIN = C_W_BYTES + ACT_BYTES + C_COEF_BYTES;
  static_assert(TH % 2 == 0 && ACT_BYTES % 16 == 0, "tile geometry");
  static_assert(ACT_BYTES >= 8 * 2 * 64 * 4, "the block's statistics reuse the activation image");
};
static_assert(C64Geo<8>::LDS_PLAIN <= 160 * 1024, "LDS image too large");

enum { EP_STATS = 0, EP_BNBWD = 1 };

struct C64P {
  const u16* in;      // (N,H,W,pitch) bf16, 64 channels
  long long in_pitch;
  const u16* wt;      // gsd_bf16_weight_image mode 0 (forward) or 1 (dX): [9][Mpad][64]
  u16* out;
  long long out_pitch;
  float* partials;    // [gridDim.x][2 * Mpad] or null
  // EP_BNBWD: raw output of the unit below (out's geometry) and its BatchNorm coefficients
  const u16* bw_y;
  long long bw_pitch;
  const float *bw_scale, *bw_shift, *bw_mean, *bw_invstd;
  int N, H, W, Mpad;
  int tiles_y, tiles_x, ntiles;
  unsigned img_bytes;   // one input image (buffer descriptor of the halo fills; below 2 GiB)
};

typedef unsigned u32x4s __attribute__((ext_vector_type(4), aligned(8)));

template <int EP, int TH>
__global__ __launch_bounds__(512) void conv64_bf16_kernel(const C64P P) {
  using G = C64Geo<TH>;
  constexpr int MT = 4, NW = 8, TR = G::TR;
  constexpr int C_TH = TH, C_NPH = G::NPH, C_ACT_PLANE = G::ACT_PLANE, C_SLOTS = G::SLOTS, C_NFILL = G::NFILL, C_NF_W = G::NF_W;
  constexpr int C_ACT_BYTES = G::ACT_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Wl = smem;
  unsigned char* Al = smem + C_W_BYTES;
  float* sBw = reinterpret_cast<float*>(smem + C_W_BYTES + C_ACT_BYTES);   // scale | shift | mean | invstd
  float* sSt = reinterpret_cast<float*>(Al);                                // [8 waves][2][64], after the last tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wq = wave & 3, wr = wave >> 2;   // the wave's 16-pixel column block and 4-row half of the tile
  const int g = lane >> 4, j = lane & 15;

  // ---- once per block: the weights become resident (72 DMA pieces of 1 KiB), coefficients -----------------------------------
#pragma unroll
  for (int k = 0; k < 72 / NW; ++k) {
    const int pc = k * NW + wave;
    const int slot = pc * 64 + lane;              // 16-byte slot of the image: row * 8 + piece'
    const int row = slot >> 3, pp = slot & 7;
    const int tap = row >> 6, r = row & 63;
    const int piece = pp ^ (r & 6);
    // LDS row r = (m-tile mm, tile row ii) receives the weights of channel (mm>>1)*32 + (ii>>2)*8 + (mm&1)*4 + (ii&3)
    const int srow = (((r >> 5) & 1) << 5) | (((r & 15) >> 2) << 3) | (((r >> 4) & 1) << 2) | (r & 3);
    const u16* src = P.wt + ((long long)(tap * P.Mpad + srow) * 64 + piece * 8);
    __builtin_amdgcn_global_load_lds((const void*)src, Wl + pc * 1024, 16, 0, 0);
  }
  if (EP != EP_STATS && tid < C_M) {
    sBw[tid] = P.bw_scale[tid];
    sBw[C_M + tid] = P.bw_shift[tid];
    sBw[2 * C_M + tid] = P.bw_mean[tid];
    sBw[3 * C_M + tid] = P.bw_invstd[tid];
  }

  // ---- the activation fill: this lane's 16-byte slots (halo pixel, chunk, piece) of its wave's DMA instructions ------------
  // slot s = instruction * 64 + lane = (chunk * 660 + q) * 4 + piece'; the piece it receives is piece' ^ swizzle(column)
  int xpk[C_NF_W];   // (halo row << 20 | halo column << 8 | element offset of the piece inside the pixel), -1: padding slot
#pragma unroll
  for (int k = 0; k < C_NF_W; ++k) {
    const int s = (k * NW + wave) * 64 + lane;
    int v = -1;
    if (s < C_SLOTS) {
      const int ch = s / (C_NPH * 4), rem = s - ch * (C_NPH * 4);
      const int q = rem >> 2, pp = rem & 3;
      const int r = q / C_HC, c = q - r * C_HC;
      const int piece = pp ^ ((c >> 1) & 2);
      v = (r << 20) | (c << 8) | (ch * 32 + piece * 8);
    }
    xpk[k] = v;
  }
  const int tpi = P.tiles_y * P.tiles_x;
  auto decode = [&](int tile, int& n, int& h0, int& w0) {
    n = tile / tpi;
    const int rem = tile - n * tpi;
    const int ty = rem / P.tiles_x;
    h0 = ty * C_TH;
    w0 = (rem - ty * P.tiles_x) * C_TW;
  };
  auto fill = [&](int tile) {
    int n, h0, w0;
    decode(tile, n, h0, w0);
    // through a buffer descriptor of image n: a 32-bit byte offset per piece, zeros for everything outside the image from the range
    // check (an offset beyond num_records) -- no zero line, no 64-bit address, no branch (profiles/ubench/buffer_lds_oob.hip)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(P.in + (long long)n * P.H * P.W * P.in_pitch), 0, P.img_bytes, 0x00020000);
#pragma unroll
    for (int k = 0; k < C_NF_W; ++k) {
      if (k * NW + wave < C_NFILL) {   // wave-uniform
        unsigned v = 0x80000000u;
        const int hi = h0 - 1 + (xpk[k] >> 20), wi = w0 - 1 + ((xpk[k] >> 8) & 0xfff);
        if (xpk[k] >= 0 && (unsigned)hi < (unsigned)P.H && (unsigned)wi < (unsigned)P.W)
          v = (unsigned)(((hi * P.W + wi) * (int)P.in_pitch + (xpk[k] & 0xff)) * 2);
        // (the image ends in the middle of the last instruction: its surplus lanes are switched off, nothing lies behind it)
        if ((k * NW + wave + 1) * 64 <= C_SLOTS || xpk[k] >= 0)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(Al + (k * NW + wave) * 1024), 16, v, 0, 0, 0);
      }
    }
  };

  float s1[MT][4], s2[MT][4];   // this lane's running partial sums over all the block's tiles
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int e = 0; e < 4; ++e) s1[m][e] = s2[m][e] = 0.f;

  int tile = blockIdx.x;
  if (tile < P.ntiles) fill(tile);
  bool counted = false;   // the previous epilogue was the branch-free one: exactly 2 TR stores lie behind this tile's fill
  for (; tile < P.ntiles; tile += gridDim.x) {
    int n, h0, w0;
    decode(tile, n, h0, w0);
    const int next = tile + (int)gridDim.x;
    // This tile's fill (and, the first time, the weights) must have landed.  vmcnt counts loads, stores and LDS-DMA in issue
    // order: behind an interior tile's epilogue its 2 TR stores are the only younger operations, so the wait leaves exactly
    // them in flight instead of draining the stores to HBM (vmcnt(0)) once per tile.
    if (counted) __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * TR));
    else __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();   // ... and everyone has left the previous epilogue

    // ---- 18 k-steps (channel chunk, kernel row, kernel column) x 4 m-tiles x TR pixel rows -------------------------------
    f32x4 acc[MT][TR];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < TR; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      int abase[2], bbase[3];
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) abase[ch] = j * 128 + (((ch * 4 + g) ^ (j & 6)) << 4);
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int c = 16 * wq + j + dx;
        bbase[dx] = (wr * TR * C_HC + c) * 64 + ((g ^ ((c >> 1) & 2)) << 4);
      }
      u32x4 a[2][MT], b[2][TR];
      auto rdA = [&](int s, int m) {
        const int ch = s / 9, tap = s - ch * 9;
        return *reinterpret_cast<const u32x4*>(Wl + tap * (C_M * 128) + m * 2048 + abase[ch]);
      };
      auto rdB = [&](int s, int t) {
        const int ch = s / 9, tap = s - ch * 9, dy = tap / 3, dx = tap - dy * 3;
        return *reinterpret_cast<const u32x4*>(Al + ch * C_ACT_PLANE + (t + dy) * (C_HC * 64) + bbase[dx]);
      };
#pragma unroll
      for (int m = 0; m < MT; ++m) a[0][m] = rdA(0, m);
#pragma unroll
      for (int t = 0; t < TR; ++t) b[0][t] = rdB(0, t);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 18; ++s) {
        // 2 TR micro-steps of {two MFMAs, operand reads for the next k-step}, pinned in this order
#pragma unroll
        for (int i = 0; i < 2 * TR; ++i) {
          const int t = i >> 1, mp = (i & 1) * 2;
          acc[mp][t] = mfma_bf16(a[s & 1][mp], b[s & 1][t], acc[mp][t]);
          acc[mp + 1][t] = mfma_bf16(a[s & 1][mp + 1], b[s & 1][t], acc[mp + 1][t]);
          if (s + 1 < 18) {
            if (i < MT) a[(s + 1) & 1][i] = rdA(s + 1, i);
            if (i < TR) b[(s + 1) & 1][i] = rdB(s + 1, i);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    __syncthreads();   // every wave has left the activation image
    // EP_BNBWD: the raw output of the unit below for this lane's pixels is requested BEFORE the next tile's fill: vector-memory
    // operations return in issue order, and behind 83 KiB of fill these 2 TR loads would come back last (the epilogue would wait
    // for the whole fill)
    u32x4 yr[EP == EP_BNBWD ? TR : 1][2];
    if (EP == EP_BNBWD) {
#pragma unroll
      for (int t = 0; t < TR; ++t) {
        const int h = min(h0 + wr * TR + t, P.H - 1), w = min(w0 + 16 * wq + j, P.W - 1);
        const u16* yp = P.bw_y + ((long long)(n * P.H + h) * P.W + w) * P.bw_pitch + g * 8;
        yr[t][0] = *reinterpret_cast<const u32x4s*>(yp);
        yr[t][1] = *reinterpret_cast<const u32x4s*>(yp + 32);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (next < P.ntiles) fill(next);   // flies under this tile's epilogue
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue ------------------------------------------------------------------------------------------------------
    const bool interior = h0 + C_TH <= P.H && w0 + C_TW <= P.W;
    const long long o_row = (long long)P.W * P.out_pitch;
    u16* const o0 = P.out + ((long long)(n * P.H + h0 + wr * TR) * P.W + (w0 + 16 * wq + j)) * P.out_pitch + g * 8;
    auto epilogue = [&](auto guard_c) {
      constexpr bool GUARD = decltype(guard_c)::value;
#pragma unroll
      for (int t = 0; t < TR; ++t) {
        const int h = h0 + wr * TR + t, w = w0 + 16 * wq + j;
        const bool ok = !GUARD || (h < P.H && w < P.W);
        unsigned pk[2 * MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          f32x4 v = acc[m][t];
          if (EP != EP_STATS) {
            const int cl = (m >> 1) * 32 + g * 8 + (m & 1) * 4;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(sBw + cl), sh = *reinterpret_cast<const f32x4*>(sBw + C_M + cl);
            const f32x4 mu = *reinterpret_cast<const f32x4*>(sBw + 2 * C_M + cl), is = *reinterpret_cast<const f32x4*>(sBw + 3 * C_M + cl);
            const unsigned y01 = yr[t][m >> 1][(m & 1) * 2], y23 = yr[t][m >> 1][(m & 1) * 2 + 1];
            const float yv[4] = {__uint_as_float(y01 << 16), __uint_as_float(y01 & 0xffff0000u), __uint_as_float(y23 << 16),
                                 __uint_as_float(y23 & 0xffff0000u)};
            float dz[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) dz[e] = fmaf(yv[e], sc[e], sh[e]) > 0.f ? v[e] : 0.f;
            const unsigned lo = pack_bf16(dz[0], dz[1]), hi = pack_bf16(dz[2], dz[3]);
            pk[2 * m] = lo;
            pk[2 * m + 1] = hi;
            if (ok) {
              const float q[4] = {__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u), __uint_as_float(hi << 16),
                                  __uint_as_float(hi & 0xffff0000u)};   // sums of the values as stored
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                s1[m][e] += q[e];
                s2[m][e] = fmaf(q[e], (yv[e] - mu[e]) * is[e], s2[m][e]);
              }
            }
          } else {
            const unsigned lo = pack_bf16(v[0], v[1]), hi = pack_bf16(v[2], v[3]);
            pk[2 * m] = lo;
            pk[2 * m + 1] = hi;
            if (ok && P.partials != nullptr) {   // statistics of the values as stored
              const float q0 = __uint_as_float(lo << 16), q1 = __uint_as_float(lo & 0xffff0000u);
              const float q2 = __uint_as_float(hi << 16), q3 = __uint_as_float(hi & 0xffff0000u);
              s1[m][0] += q0; s2[m][0] = fmaf(q0, q0, s2[m][0]);
              s1[m][1] += q1; s2[m][1] = fmaf(q1, q1, s2[m][1]);
              s1[m][2] += q2; s2[m][2] = fmaf(q2, q2, s2[m][2]);
              s1[m][3] += q3; s2[m][3] = fmaf(q3, q3, s2[m][3]);
            }
          }
        }
        if (ok) {
          u16* o = o0 + t * o_row;
          *reinterpret_cast<u32x4s*>(o) = u32x4{pk[0], pk[1], pk[2], pk[3]};
          *reinterpret_cast<u32x4s*>(o + 32) = u32x4{pk[4], pk[5], pk[6], pk[7]};
        }
      }
    };
    if (interior) epilogue(std::integral_constant<bool, false>{});
    else epilogue(std::integral_constant<bool, true>{});
    counted = interior;
  }
  // ---- one partial row per block: 16-lane DPP sums, then the eight waves through LDS (the activation image's space) -------
  gsd_dma_barrier();   // (a last, unread fill may still be in flight)
  if (P.partials != nullptr) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t1 = reduce16_to_lane15(s1[m][e]), t2 = reduce16_to_lane15(s2[m][e]);
        if (j == 15) {
          const int c = (m >> 1) * 32 + g * 8 + (m & 1) * 4 + e;
          sSt[(wave * 2 + 0) * 64 + c] = t1;
          sSt[(wave * 2 + 1) * 64 + c] = t2;
        }
      }
    __syncthreads();
    if (tid < C_M) {
      float* row = P.partials + (size_t)blockIdx.x * (2 * P.Mpad);
      float r1 = 0.f, r2 = 0.f;
#pragma unroll
      for (int wv = 0; wv < NW; ++wv) {
        r1 += sSt[(wv * 2 + 0) * 64 + tid];
        r2 += sSt[(wv * 2 + 1) * 64 + tid];
      }
      row[tid] = r1;
      row[P.Mpad + tid] = r2;
    }
  }
}

int c64_cu_count() {
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
    v = 256;
  return v;
}

long c64_tiles(int N, int H, int W, int th) { return (long)N * ceil_div(H, th) * ceil_div(W, C_TW); }

template <int EP, int TH>
int c64_launch(C64P& P, size_t lds, void* stream, const char* what) {
  const long nt = c64_tiles(P.N, P.H, P.W, TH);
  GSD_REQUIRE(nt < 2147483647L, GSD_ERR_UNSUPPORTED, "%s: too many tiles", what);
  P.tiles_y = ceil_div(P.H, TH); P.tiles_x = ceil_div(P.W, C_TW);
  P.ntiles = (int)nt;
  const int cus = c64_cu_count();
  const int grid = (int)(nt < cus ? nt : cus);
  static gsd_attr_once big_lds;   // per-device cache of an idempotent launch attribute (gsd_common.h)
  if (hipError_t e = gsd_allow_big_lds(big_lds, reinterpret_cast<const void*>(&conv64_bf16_kernel<EP, TH>)); e != hipSuccess) {
    gsd_set_error("%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e));
    return GSD_ERR_HIP;
  }
  hipLaunchKernelGGL((conv64_bf16_kernel<EP, TH>), dim3(grid), dim3(512), lds, (hipStream_t)stream, P);
  GSD_LAUNCH_CHECK(what);
  return GSD_OK;
}

int c64_common(C64P& P, const gsd_nhwc* in, const void* wt, const gsd_nhwc* out, float* partials, const char* what) {
  if (int e = gsd_check_nhwc(in, what)) return e;
  if (int e = gsd_check_nhwc(out, what)) return e;
  GSD_REQUIRE(wt != nullptr && ((uintptr_t)wt & 15) == 0, GSD_ERR_BAD_ARG, "%s: the weight image must be non-null and 16-byte aligned", what);
  GSD_REQUIRE(in->C == C_M && out->C == C_M, GSD_ERR_UNSUPPORTED, "%s: 64 -> 64 channels only (got %d -> %d); use gsd_bf16_conv3x3", what,
              in->C, out->C);
  GSD_REQUIRE(in->N == out->N && in->H == out->H && in->W == out->W, GSD_ERR_BAD_ARG, "%s: in/out extents differ", what);
  GSD_REQUIRE((out->pitch & 3) == 0, GSD_ERR_UNSUPPORTED, "%s: out pitch must be a multiple of 4", what);
  P.in = (const u16*)in->ptr; P.in_pitch = in->pitch; P.wt = (const u16*)wt;
  GSD_REQUIRE((long long)in->H * in->W * in->pitch < (1LL << 30), GSD_ERR_UNSUPPORTED, "gsd_bf16_conv3x3_c64: one input image exceeds 2 GiB");
  P.img_bytes = (unsigned)((long long)in->H * in->W * in->pitch * 2);
  P.out = (u16*)out->ptr; P.out_pitch = out->pitch; P.partials = partials;
  P.bw_y = nullptr; P.bw_pitch = 0; P.bw_scale = P.bw_shift = P.bw_mean = P.bw_invstd = nullptr;
  P.N = in->N; P.H = in->H; P.W = in->W; P.Mpad = round_up(C_M, 128);
  return GSD_OK;
}

}  // namespace

extern "C" int gsd_bf16_conv3x3_c64_supported(int K, int M) { return (K == C_M && M == C_M) ? 1 : 0; }

extern "C" int gsd_bf16_conv3x3_c64_partial_rows(int N, int H, int W) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  const long nt = c64_tiles(N, H, W, 8);
  const int cus = c64_cu_count();
  return (int)(nt < cus ? nt : cus);
}

extern "C" int gsd_bf16_conv3x3_c64(const gsd_nhwc* in, const void* wt, const gsd_nhwc* out, float* partials,
                                    const gsd_bf16_bnbwd* bw, void* stream) {
  C64P P;
  if (int e = c64_common(P, in, wt, out, partials, "gsd_bf16_conv3x3_c64")) return e;
  if (bw == nullptr) return c64_launch<EP_STATS, 8>(P, C64Geo<8>::LDS_PLAIN, stream, "gsd_bf16_conv3x3_c64");
  if (int e = gsd_check_nhwc(bw->y, "gsd_bf16_conv3x3_c64 bw.y")) return e;
  GSD_REQUIRE(bw->scale && bw->shift && bw->mean && bw->invstd && partials, GSD_ERR_BAD_ARG,
              "gsd_bf16_conv3x3_c64: fused BatchNorm backward needs coefficients and partials");
  GSD_REQUIRE(bw->y->N == out->N && bw->y->H == out->H && bw->y->W == out->W && bw->y->C == C_M && (bw->y->pitch & 3) == 0,
              GSD_ERR_BAD_ARG, "gsd_bf16_conv3x3_c64: bw.y must have out's geometry");
  P.bw_y = (const u16*)bw->y->ptr; P.bw_pitch = bw->y->pitch;
  P.bw_scale = bw->scale; P.bw_shift = bw->shift; P.bw_mean = bw->mean; P.bw_invstd = bw->invstd;
  return c64_launch<EP_BNBWD, 8>(P, C64Geo<8>::LDS_PLAIN, stream, "gsd_bf16_conv3x3_c64");
}

