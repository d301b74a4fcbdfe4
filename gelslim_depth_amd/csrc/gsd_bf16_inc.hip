// gsd_bf16_inc.hip -- the north star's named block: the `inc` double convolution (unet.py:7-20 for `inc`, :67; 3 -> 64 -> 64 at
// full resolution) in TRAIN mode without the raw output of its first convolution ever existing in HBM.
//
// Unfused (gsd_bf16_first.hip + gsd_bf16_conv.hip + two BatchNorm-apply passes) the block moves 2.55x its algorithmic bytes:
// the first convolution writes its raw output y0 (one MFMA k-step of arithmetic under 560 MB at batch 32), a pass reads it
// back and writes a0 = relu(bn(y0)), and the 64 -> 64 convolution re-reads a0 with 1.57x halo overlap.  Here:
//
//   statistics of y0      gsd_bf16_conv3x3_first(out = NULL): the same kernel, same partial rows, no store (gsd_bf16_first.hip)
//   gsd_bf16_inc_conv     one persistent kernel per CU.  Per 8 x 64 pixel tile it rebuilds a0 over the tile's 10 x 66 halo from
//                         the 12 x 68 x-tile -- one MFMA k-step (K = 27 -> 32) per 16 pixels and m-tile, then round to bf16 as
//                         the store would, BatchNorm + ReLU, round again: 7 % extra MFMAs -- stages it in LDS as the B operand
//                         of the 64 -> 64 convolution, writes the tile's own a0 pixels to HBM ONCE (the 64 -> 64 layer's dW
//                         reads it in backward) and runs the 18 k-steps (2 channel chunks x 9 taps) against weights that are
//                         RESIDENT in LDS (9 x 64 x 64 bf16 = 72 KiB, loaded once per block): no operand fill, no barrier
//                         inside the K loop.  Writes y1 (raw) + its BatchNorm partial sums.
//   backward              the first layer's BatchNorm backward recomputes y0 from x the same way (gsd_bf16_first_bn_bwd_reduce,
//                         gsd_bf16_wgrad_first_recompute in gsd_bf16_first.hip).
//
// Same products in the same order through the same MFMA as the unfused kernels: a0 and y1 are bit-identical to
// gsd_bf16_conv3x3_first + gsd_bf16_bn_apply + gsd_bf16_conv3x3 given the same statistics.
//
// What bounds it (in-kernel stamps and ablation builds at batch 32, profiles/r04_inc_*.txt): the K loop is 0.20 ms of the
// kernel's 0.46-0.48 (18.8 cycles per MFMA and SIMD against the instruction's 16); the rest is VECTOR work -- rebuild ~150
// instructions per 16 halo pixels, epilogue ~250 per wave tile -- and the stores (0.09 ms).  Forms measured on the way:
//   * 4 waves, one per SIMD, as the DMA-filled kernel runs: 0.54 ms.  A lone wave issues a vector instruction every ~4 cycles
//     (MI355X_MICROARCH.md 'vector-instruction ISSUE cost'): rebuild 12.0k + K loop 10.9k + epilogue 5.4k cycles per tile.
//   * 8 waves in the same phase (this file): rebuild 8.5k + K loop 10.8k + epilogue 4.6k.
//   * 8 waves as two groups half a tile out of phase, so that every SIMD hosts a matrix wave beside a vector wave
//     (profiles/ubench/gsd_bf16_inc_paired.hip, bit-identical): 0.62-0.71 ms.  The vector phase is the longer one, and beside
//     a matrix wave a vector wave runs at a third of its rate (an MFMA holds the SIMD's issue port 8 cycles of 16; its own 4
//     MFMAs per 16 pixels queue behind the partner's): 11-12k cycles per half-step against 5.7-7.2k for the K loop.  Along the
//     way: a lane constant set up at kernel entry and spilled around the loop costs an HBM round trip per tile -- the reload's
//     `s_waitcnt vmcnt(0)` in front of the K loop waits for the x prefetch issued just before it (K loop 10.5k -> 5.7k cycles
//     once the constants were recomputed in place behind an opaque asm).
//
// LDS image of the activation tile: [chunk of 32 channels][halo pixel q = r*66 + c][64 B], 16-byte piece p of pixel (r, c) at
// slot p ^ (2 * ((c >> 2) & 1)): with that swizzle the ds_read_b128 of a B operand (16 consecutive pixels x one piece, lanes
// grouped as the hardware groups them) is bank-conflict free for EVERY tap shift without the 32 B of padding per pixel the
// DMA-filled kernel spends (the swizzle depends on the column only, so kernel rows and channel chunks are address immediates).
// Weights: [tap][64 rows][128 B], piece p of row i at slot p ^ (i & 6); rows permuted as in gsd_bf16_conv.hip so that a lane
// ends up with channels g*8 .. g*8+7 and 32 + g*8 .. of its pixel (16-byte stores).
#include "gsd_bf16_common.h"

#include <type_traits>

namespace {

constexpr int I_TW = 64;          // tile width: wave w owns columns 16w .. 16w+15 of every tile row
constexpr int I_M = 64;           // channels of both convolutions' outputs (and of the second one's input)
constexpr int I_HC = I_TW + 2;    // halo row pitch (pixels)
constexpr int I_XC = I_TW + 4;    // x-tile row pitch (elements)

template <int TH>
struct IncGeo {
  static constexpr int HR = TH + 2, NPH = HR * I_HC, NTH = (NPH + 15) / 16;
  static constexpr int XR = TH + 4, XPLANE = XR * I_XC;
  static constexpr int W_BYTES = 9 * I_M * 128;
  static constexpr int ACT_PLANE = NPH * 64;
  static constexpr int ACT_BYTES = 2 * ACT_PLANE;
  static constexpr int NXE = (3 * XPLANE + 8 + 511) / 512;   // x-tile elements per thread: three channel planes + 8 zero elements
  static constexpr int XS_BYTES = NXE * 512 * 2;             // every thread stores all its NXE slots, unconditionally
  static constexpr int COEF_BYTES = 2 * I_M * 4;
  static constexpr int LDS = W_BYTES + ACT_BYTES + XS_BYTES + COEF_BYTES;
  static_assert(XS_BYTES >= 8 * 2 * 64 * 4, "the block's statistics reuse the x tile");
  static_assert(LDS <= 160 * 1024, "LDS image too large");
};

struct IncP {
  const float* x;     // (N, C, H, W) fp32
  const u16* wt0;     // gsd_bf16_weight_image mode 2: [Mpad][32], k = c*9 + t
  const u16* wt1;     // gsd_bf16_weight_image mode 0: [9][Mpad][64]
  const float* scale0;
  const float* shift0;
  u16* a0;
  long long a0_pitch;
  u16* y1;
  long long y1_pitch;
  float* partials;    // [gridDim.x][2 * Mpad]
  int N, C, H, W, Mpad;
  int tiles_y, tiles_x, ntiles;
};

typedef unsigned u32x4s __attribute__((ext_vector_type(4), aligned(8)));

#ifndef INC_NT      // 1: a0 / y1 leave as non-temporal stores (diagnostic switch while measuring)
#define INC_NT 0
#endif
__device__ __forceinline__ void inc_store16(u16* p, u32x4 v) {
#if INC_NT
  __builtin_nontemporal_store(v, reinterpret_cast<u32x4s*>(p));
#else
  *reinterpret_cast<u32x4s*>(p) = v;
#endif
}

#ifndef INC_ABL     // diagnostic builds only (results are then garbage): 1 no a0 stores, 2 no y1 stores, 4 no statistics,
#define INC_ABL 0   // 8 no K loop, 16 the rebuild's activation is not staged in LDS
#endif
#ifndef INC_STAMP   // diagnostic builds only (profiles/build_diag_one.sh, profiles/bench_inc_block.py): wave 0 of every block
#define INC_STAMP 0 // leaves the shader cycles it spent per phase (rebuild, K loop, epilogue, waiting at the barriers) in a buffer
#endif
#if INC_STAMP
__device__ unsigned long long inc_stamp_buf[512 * 8];
#define INC_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define INC_T(v)
#endif

// Block = 8 waves, two per SIMD: with ONE wave per SIMD a vector instruction issues every ~4 cycles instead of 2
// (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost'), and this kernel's rebuild and epilogue are vector-bound -- in-kernel
// stamps of a 4-wave form: rebuild 12.0k, K loop 10.9k, epilogue 5.4k cycles per item.  Wave w = (column block wq = w & 3,
// row half wr = w >> 2): 64 channels x (TH/2 rows x 16 pixels) = 4 x TH/2 MFMA tiles.
template <int TH>
__global__ __launch_bounds__(512) void inc_fused_bf16_kernel(const IncP P) {
  using G = IncGeo<TH>;
  constexpr int MT = 4, NW = 8, TR = TH / 2;
  static_assert(TH % 2 == 0, "two row halves");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Wl = smem;
  unsigned char* Al = smem + G::W_BYTES;
  u16* Xs = reinterpret_cast<u16*>(smem + G::W_BYTES + G::ACT_BYTES);
  float* sCo = reinterpret_cast<float*>(smem + G::W_BYTES + G::ACT_BYTES + G::XS_BYTES);   // scale0[64] | shift0[64]
  float* sSt = reinterpret_cast<float*>(Xs);   // [8 waves][2][64], after the last item
  constexpr int ZERO = 3 * G::XPLANE;            // (the slots past the channel planes are written with zeros by put_x)
  constexpr int NXE = G::NXE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wq = wave & 3, wr = wave >> 2;
  const int g = lane >> 4, j = lane & 15;

  // ---- once per block: the 64 -> 64 weights become resident (72 DMA pieces of 1 KiB) ---------------------------------------
#pragma unroll
  for (int k = 0; k < 72 / NW; ++k) {
    const int pc = k * NW + wave;
    const int slot = pc * 64 + lane;              // 16-byte slot of the image: row * 8 + piece'
    const int row = slot >> 3, pp = slot & 7;
    const int tap = row >> 6, r = row & 63;
    const int piece = pp ^ (r & 6);
    // LDS row r = (m-tile mm, tile row ii) receives the weights of channel (mm>>1)*32 + (ii>>2)*8 + (mm&1)*4 + (ii&3)
    const int srow = (((r >> 5) & 1) << 5) | (((r & 15) >> 2) << 3) | (((r >> 4) & 1) << 2) | (r & 3);
    const u16* src = P.wt1 + ((long long)(tap * P.Mpad + srow) * 64 + piece * 8);
    __builtin_amdgcn_global_load_lds((const void*)src, Wl + pc * 1024, 16, 0, 0);
  }
  // BatchNorm coefficients of the first unit: LDS (two waves per SIMD cover the read; 32 registers a lane they would cost in the
  // K loop, which lives at the 256-register limit of two waves per SIMD)
  if (tid < I_M) {
    sCo[tid] = P.scale0[tid];
    sCo[I_M + tid] = P.shift0[tid];
  }
  // first convolution: A operands in registers (MFMA tile m, row j holds channel (m>>1)*32 + (j>>2)*8 + (m&1)*4 + (j&3))
  u32x4 a0w[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int ch = (m >> 1) * 32 + (j >> 2) * 8 + (m & 1) * 4 + (j & 3);
    a0w[m] = *reinterpret_cast<const u32x4*>(P.wt0 + (size_t)ch * 32 + g * 8);
  }
  // its B operand gather: k = 8g + e -> (channel c, tap t): x-tile offset of the tap relative to the halo pixel (k >= 9 C: the
  // zero element, through an offset that does not depend on the pixel)
  int off[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 8 * g + e;
    const int c = k / 9, t = k - c * 9;
    off[e] = k < 9 * P.C ? c * G::XPLANE + (t / 3) * I_XC + (t % 3) : -1;
  }
  // this thread's x-tile elements i = tid + 512 k: (channel, row, column) packed, -1 past the planes (those slots receive zeros)
  int xel[NXE];
#pragma unroll
  for (int k = 0; k < NXE; ++k) {
    const int i = tid + k * 512;
    const int c = i / G::XPLANE, r = (i - c * G::XPLANE) / I_XC, col = i - c * G::XPLANE - r * I_XC;
    xel[k] = i < P.C * G::XPLANE ? (c << 16 | r << 8 | col) : -1;
  }
  const int tpi = P.tiles_y * P.tiles_x;
  auto decode = [&](int tile, int& n, int& h0, int& w0) {
    n = tile / tpi;
    const int rem = tile - n * tpi;
    const int ty = rem / P.tiles_x;
    h0 = ty * TH;
    w0 = (rem - ty * P.tiles_x) * I_TW;
  };
  // branch-free: every load is issued (from a clamped, always legal address); the select waits for the data, so it happens in put_x
  float xv[NXE];
  unsigned xok = 0;     // bit k: element k lies inside the image
  auto fetch = [&](int tile) {
    int n, h0, w0;
    decode(tile, n, h0, w0);
    const float* xn = P.x + (size_t)n * P.C * P.H * P.W;
    xok = 0;
#pragma unroll
    for (int k = 0; k < NXE; ++k) {
      const int e = xel[k] >= 0 ? xel[k] : 0;
      const int c = e >> 16, gh = h0 - 2 + (e >> 8 & 255), gw = w0 - 2 + (e & 255);
      const bool ok = xel[k] >= 0 && (unsigned)gh < (unsigned)P.H && (unsigned)gw < (unsigned)P.W;
      const int ghc = min(max(gh, 0), P.H - 1), gwc = min(max(gw, 0), P.W - 1);
      xv[k] = xn[(c * P.H + ghc) * P.W + gwc];
      xok |= ok ? 1u << k : 0u;
    }
  };
  auto put_x = [&]() {
#pragma unroll
    for (int k = 0; k < NXE; ++k) Xs[tid + k * 512] = f32_to_bf16((xok >> k & 1) ? xv[k] : 0.f);
  };

  // second convolution: operand read offsets (see the file header for the two swizzles)
  int abase[2], bbase[3];
#pragma unroll
  for (int ch = 0; ch < 2; ++ch) abase[ch] = j * 128 + (((ch * 4 + g) ^ (j & 6)) << 4);
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int c = 16 * wq + j + dx;
    bbase[dx] = (wr * TR * I_HC + c) * 64 + ((g ^ ((c >> 1) & 2)) << 4);
  }

  float s1[MT][4], s2[MT][4];   // this lane's running BatchNorm sums of y1 over all the block's items
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int e = 0; e < 4; ++e) s1[m][e] = s2[m][e] = 0.f;

  int tile = blockIdx.x;
  if (tile < P.ntiles) {
    fetch(tile);
    put_x();
  }
  gsd_dma_barrier();   // the weights have landed (vmcnt(0)); the first x tile is visible
#if INC_STAMP
  unsigned long long st_b = 0, st_c = 0, st_d = 0, st_w = 0, st_w2 = 0, st_w3 = 0, st_items = 0;
  const unsigned long long st_begin = __builtin_amdgcn_s_memtime(), st_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (; tile < P.ntiles; tile += gridDim.x) {
    int n, h0, w0;
    decode(tile, n, h0, w0);
    INC_T(t0);
    // ---- rebuild a0 = relu(bn(conv(x))) over the halo: one MFMA k-step per 16 halo pixels and m-tile -----------------
    // Two tiles (tt, tt + 8) per iteration, their gathers issued an iteration ahead.
    auto gather = [&](int tt, unsigned (&v)[8]) {
      const int q = tt * 16 + j;
      const int qc = q < G::NPH ? q : G::NPH - 1;
      const int r = qc / I_HC, c = qc - r * I_HC;
      const int base = r * I_XC + c;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = Xs[off[e] >= 0 ? base + off[e] : ZERO];
    };
    // the first unit's coefficients for this lane's 16 channels: read ONCE per item into registers that are dead in the K loop
    // (inside the tile loop hipcc cannot hoist the reads itself -- the activation tile is written through the same LDS array --
    // and each one would be an exposed round trip: 4 x lgkmcnt(0) per tile in the first 8-wave build)
    f32x4 csc[MT], csh[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int cl = (m >> 1) * 32 + g * 8 + (m & 1) * 4;
      csc[m] = *reinterpret_cast<const f32x4*>(sCo + cl);
      csh[m] = *reinterpret_cast<const f32x4*>(sCo + I_M + cl);
    }
    auto finish = [&](int tt, const f32x4 (&acc)[MT]) {
      const int q = tt * 16 + j;
      const int qc = q < G::NPH ? q : G::NPH - 1;
      const int r = qc / I_HC, c = qc - r * I_HC;
      const int h = h0 - 1 + r, w = w0 - 1 + c;
      const bool inimg = (unsigned)h < (unsigned)P.H && (unsigned)w < (unsigned)P.W;   // outside: the zero padding of conv 2
      const unsigned keep = inimg ? 0xffffffffu : 0u;
      unsigned pk[2 * MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const f32x4 csc_m = csc[m], csh_m = csh[m];
        // y0 as the unfused path stores it (two values per conversion), then gsd_bf16_bn_apply's expression; the ReLU on the
        // packed pair: as 16-bit integers a bf16 is negative exactly when its sign bit is set (max with 0 also turns -0 into +0)
        const unsigned y01 = pack_bf16(acc[m][0], acc[m][1]), y23 = pack_bf16(acc[m][2], acc[m][3]);
        const float a0 = fmaf(__uint_as_float(y01 << 16), csc_m[0], csh_m[0]);
        const float a1 = fmaf(__uint_as_float(y01 & 0xffff0000u), csc_m[1], csh_m[1]);
        const float a2 = fmaf(__uint_as_float(y23 << 16), csc_m[2], csh_m[2]);
        const float a3 = fmaf(__uint_as_float(y23 & 0xffff0000u), csc_m[3], csh_m[3]);
        pk[2 * m] = relu_pk_bf16(pack_bf16(a0, a1)) & keep;
        pk[2 * m + 1] = relu_pk_bf16(pack_bf16(a2, a3)) & keep;
      }
      if (q < G::NPH) {
        const int sw = (g ^ ((c >> 1) & 2)) << 4;
#if (INC_ABL) & 16   // diagnostic: the rebuilt activation is not staged (the never-true test keeps pk alive)
        if (pk[0] == 0x12345u)
#endif
        {
          *reinterpret_cast<u32x4*>(Al + q * 64 + sw) = u32x4{pk[0], pk[1], pk[2], pk[3]};
          *reinterpret_cast<u32x4*>(Al + G::ACT_PLANE + q * 64 + sw) = u32x4{pk[4], pk[5], pk[6], pk[7]};
        }
        if (!((INC_ABL) & 1) && inimg && r >= 1 && r <= TH && c >= 1 && c <= I_TW) {   // the tile's own pixels: a0 goes to HBM once
          u16* o = P.a0 + ((long long)(n * P.H + h) * P.W + w) * P.a0_pitch + g * 8;
          inc_store16(o, u32x4{pk[0], pk[1], pk[2], pk[3]});
          inc_store16(o + 32, u32x4{pk[4], pk[5], pk[6], pk[7]});
        }
      }
    };
    {
      unsigned va[8], vb[8];
      gather(wave, va);
      gather(wave + NW, vb);
      for (int tt = wave; tt < G::NTH; tt += 2 * NW) {
        const u32x4 ba = {va[0] | va[1] << 16, va[2] | va[3] << 16, va[4] | va[5] << 16, va[6] | va[7] << 16};
        const u32x4 bb = {vb[0] | vb[1] << 16, vb[2] | vb[3] << 16, vb[4] | vb[5] << 16, vb[6] | vb[7] << 16};
        f32x4 acca[MT], accb[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acca[m] = mfma_bf16(a0w[m], ba, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int m = 0; m < MT; ++m) accb[m] = mfma_bf16(a0w[m], bb, f32x4{0.f, 0.f, 0.f, 0.f});
        gather(tt + 2 * NW, va);        // (past the last tile: clamped reads that nobody uses)
        gather(tt + 3 * NW, vb);
        __builtin_amdgcn_sched_barrier(0);
        finish(tt, acca);
        finish(tt + NW, accb);
      }
    }
    INC_T(t1);
    __syncthreads();   // the activation tile is complete
    INC_T(t2);
    const int next = tile + (int)gridDim.x;
    if (next < P.ntiles) fetch(next);   // flies during the K loop

    // ---- 64 -> 64 convolution: 18 k-steps (channel chunk, kernel row, kernel column) x 4 m-tiles x TR pixel rows ------
    f32x4 acc[MT][TR];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < TR; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 a[2][MT], b[2][TR];
    auto rdA = [&](int s, int m) {
      const int ch = s / 9, tap = s - ch * 9;
      return *reinterpret_cast<const u32x4*>(Wl + tap * (I_M * 128) + m * 2048 + abase[ch]);
    };
    auto rdB = [&](int s, int t) {
      const int ch = s / 9, tap = s - ch * 9, dy = tap / 3, dx = tap - dy * 3;
      return *reinterpret_cast<const u32x4*>(Al + ch * G::ACT_PLANE + (t + dy) * (I_HC * 64) + bbase[dx]);
    };
#pragma unroll
    for (int m = 0; m < MT; ++m) a[0][m] = rdA(0, m);
#pragma unroll
    for (int t = 0; t < TR; ++t) b[0][t] = rdB(0, t);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < (((INC_ABL) & 8) ? 1 : 18); ++s) {
      // 2 TR micro-steps of {two MFMAs, one operand read for the next k-step}, pinned in this order (the LDS instructions keep
      // their program order anyway, so the interleaving has to be written out)
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
    INC_T(t3);
    __syncthreads();   // every wave has left the activation tile
    INC_T(t4);
    if (next < P.ntiles) put_x();   // (the x tile's readers finished before the barrier in front of the K loop)

    // ---- epilogue: y1 (raw, bf16) + the BatchNorm partial sums of the values as stored --------------------------------
    const bool interior = h0 + TH <= P.H && w0 + I_TW <= P.W;
    const long long y_row = (long long)P.W * P.y1_pitch;
    u16* const y_o0 = P.y1 + ((long long)(n * P.H + h0 + wr * TR) * P.W + (w0 + 16 * wq + j)) * P.y1_pitch + g * 8;
    auto epilogue = [&](auto guard_c) {
      constexpr bool GUARD = decltype(guard_c)::value;
#pragma unroll
      for (int t = 0; t < TR; ++t) {
        const int h = h0 + wr * TR + t, w = w0 + 16 * wq + j;
        const bool ok = !GUARD || (h < P.H && w < P.W);
        unsigned pk[2 * MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const f32x4 v = acc[m][t];
          const unsigned lo = pack_bf16(v[0], v[1]), hi = pack_bf16(v[2], v[3]);
          pk[2 * m] = lo;
          pk[2 * m + 1] = hi;
          if (ok && !((INC_ABL) & 4)) {
            const float q0 = __uint_as_float(lo << 16), q1 = __uint_as_float(lo & 0xffff0000u);
            const float q2 = __uint_as_float(hi << 16), q3 = __uint_as_float(hi & 0xffff0000u);
            s1[m][0] += q0; s2[m][0] = fmaf(q0, q0, s2[m][0]);
            s1[m][1] += q1; s2[m][1] = fmaf(q1, q1, s2[m][1]);
            s1[m][2] += q2; s2[m][2] = fmaf(q2, q2, s2[m][2]);
            s1[m][3] += q3; s2[m][3] = fmaf(q3, q3, s2[m][3]);
          }
        }
        if (ok && (!((INC_ABL) & 2) || pk[0] == 0x12345u)) {
          u16* o = y_o0 + t * y_row;
          inc_store16(o, u32x4{pk[0], pk[1], pk[2], pk[3]});
          inc_store16(o + 32, u32x4{pk[4], pk[5], pk[6], pk[7]});
        }
      }
    };
    if (interior) epilogue(std::integral_constant<bool, false>{});
    else epilogue(std::integral_constant<bool, true>{});
    INC_T(t5);
    __syncthreads();   // the next x tile is visible
#if INC_STAMP
    const unsigned long long t6 = __builtin_amdgcn_s_memtime();
    st_b += t1 - t0; st_c += t3 - t2; st_d += t5 - t4; st_w += t2 - t1; st_w2 += t4 - t3; st_w3 += t6 - t5; ++st_items;
#endif
  }
#if INC_STAMP
  if (tid == 0 && blockIdx.x < 512) {
    unsigned long long* o = inc_stamp_buf + 8 * blockIdx.x;
    o[0] = st_b; o[1] = st_c; o[2] = st_d; o[3] = st_w; o[4] = st_items;
    o[5] = __builtin_amdgcn_s_memtime() - st_begin; o[6] = __builtin_amdgcn_s_memrealtime() - st_rt0; o[7] = st_w2 | (st_w3 << 32);
  }
#endif
  // ---- one partial row per block: 16-lane DPP sums, then the eight waves through LDS (the x tile's space) ---------------
  __syncthreads();
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
  if (tid < I_M) {
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

constexpr int INC_TH = 8;

int inc_cu_count() {
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
    v = 256;
  return v;
}

long inc_tiles(int N, int H, int W) { return (long)N * ceil_div(H, INC_TH) * ceil_div(W, I_TW); }

}  // namespace

#if INC_STAMP
extern "C" int gsd_diag_inc_stamps(unsigned long long* host, int nblocks) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(inc_stamp_buf), sizeof(unsigned long long) * 8 * (nblocks < 512 ? nblocks : 512)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int gsd_bf16_inc_supported(int C, int M) { return (C >= 1 && 9 * C <= 32 && M == I_M) ? 1 : 0; }

extern "C" int gsd_bf16_inc_conv_partial_rows(int N, int H, int W) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  const long nt = inc_tiles(N, H, W);
  const int cus = inc_cu_count();
  return (int)(nt < cus ? nt : cus);
}

extern "C" int gsd_bf16_inc_conv(const float* x, int N, int C, int H, int W, const void* wt0, const float* scale0, const float* shift0,
                                 const void* wt1, const gsd_nhwc* a0, const gsd_nhwc* y1, float* partials, void* stream) {
  GSD_REQUIRE(x && wt0 && wt1 && scale0 && shift0 && partials, GSD_ERR_BAD_ARG, "gsd_bf16_inc_conv: null argument");
  if (int e = gsd_check_nhwc(a0, "gsd_bf16_inc_conv a0")) return e;
  if (int e = gsd_check_nhwc(y1, "gsd_bf16_inc_conv y1")) return e;
  GSD_REQUIRE(gsd_bf16_inc_supported(C, a0->C), GSD_ERR_UNSUPPORTED,
              "gsd_bf16_inc_conv: needs 9*C <= 32 and 64 channels (got C=%d M=%d); use gsd_bf16_conv3x3_first + gsd_bf16_bn_apply + "
              "gsd_bf16_conv3x3", C, a0->C);
  GSD_REQUIRE(a0->N == N && a0->H == H && a0->W == W && y1->N == N && y1->H == H && y1->W == W && y1->C == I_M &&
                  (a0->pitch & 3) == 0 && (y1->pitch & 3) == 0,
              GSD_ERR_BAD_ARG, "gsd_bf16_inc_conv: a0 and y1 must be (N,H,W,64)");
  GSD_REQUIRE(((uintptr_t)wt0 & 15) == 0 && ((uintptr_t)wt1 & 15) == 0, GSD_ERR_BAD_ARG,
              "gsd_bf16_inc_conv: the weight images must be 16-byte aligned");
  IncP P;
  P.x = x; P.wt0 = (const u16*)wt0; P.wt1 = (const u16*)wt1; P.scale0 = scale0; P.shift0 = shift0;
  P.a0 = (u16*)a0->ptr; P.a0_pitch = a0->pitch; P.y1 = (u16*)y1->ptr; P.y1_pitch = y1->pitch;
  P.partials = partials;
  P.N = N; P.C = C; P.H = H; P.W = W; P.Mpad = round_up(I_M, 128);
  P.tiles_y = ceil_div(H, INC_TH); P.tiles_x = ceil_div(W, I_TW);
  const long nt = inc_tiles(N, H, W);
  GSD_REQUIRE(nt < 2147483647L, GSD_ERR_UNSUPPORTED, "gsd_bf16_inc_conv: too many tiles");
  P.ntiles = (int)nt;
  const int grid = gsd_bf16_inc_conv_partial_rows(N, H, W);
  static gsd_attr_once big_lds;   // per-device cache of an idempotent launch attribute (gsd_common.h)
  if (hipError_t e = gsd_allow_big_lds(big_lds, reinterpret_cast<const void*>(&inc_fused_bf16_kernel<INC_TH>)); e != hipSuccess) {
    gsd_set_error("gsd_bf16_inc_conv: hipFuncSetAttribute: %s", hipGetErrorString(e));
    return GSD_ERR_HIP;
  }
  hipLaunchKernelGGL(inc_fused_bf16_kernel<INC_TH>, dim3(grid), dim3(512), IncGeo<INC_TH>::LDS, (hipStream_t)stream, P);
  GSD_LAUNCH_CHECK("gsd_bf16_inc_conv");
  return GSD_OK;
}
