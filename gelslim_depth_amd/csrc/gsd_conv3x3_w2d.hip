// gsd_conv3x3_w2d.hip -- conv3x3 (pad 1, stride 1, no bias) forward and dX with the TWO-dimensional Winograd minimal-filtering
// identity F(2 x 4, 3 x 3) on v_mfma_f32_16x16x4_f32 (gfx950): F(4,3) along the image rows (as gsd_conv3x3_w43.hip) combined
// with F(2,3) down the columns.
//
// Same operator as gsd_conv3x3.hip / gsd_conv3x3_w43.hip (aten::convolution at /root/reference/gelslim_depth/models/unet.py:11,14
// and the dX half of aten::convolution_backward), same fp32 storage and fp32 accumulation.  A tile of 2 x 4 outputs of one
// channel needs a 4 x 6 input window and, per input channel, 4 x 6 = 24 products instead of 2 * 4 * 9 = 72:
//
//   Y (2x4) = A2^T [ (G2 g G4^T) .* (B2^T d B4) ] A4,     d = the 4 x 6 window, g = the 3 x 3 kernel
//
// a third of the direct form's multiplications and two thirds of the row-only form's (36 per 2 x 4 outputs).  The contraction
// over the input channels stays on the MFMA: 24 GEMMs M_f[m][tile] = sum_ci U_f[ci][m] * V_f[ci][tile], f = (fr, fc).
// F(2,3) is the mildest Winograd transform there is (constants 1 and 1/2): the op-level tests bound the combined form at the same
// 1e-5 relative L1 against the fp64 oracle as the row-only form (north-star tolerance: 1e-3).
//
// Block = 4 waves on a 64-channel x 256-pixel tile, as the row-only kernel, but the waves split it 2 (channel halves) x 2 (pixel
// halves): a wave owns 32 output channels x 16 tiles (128 pixels) = 2 MFMA m-tiles x 24 frequencies = 192 accumulator registers
// (64 channels x 16 tiles would need 384), two blocks per CU.  Per 4-channel chunk a wave issues 24 k-steps x 2 MFMAs = 48 MFMAs for
// 128 pixels where the row-only kernel issues 72 for 64.  The weight image of a chunk is 96 x 64 floats = 24 KiB (U = G2 g G4^T,
// laid out once per optimiser step by gsd_weight_layout modes 8 / 9); the (TH + 2) x (TW + 2) halo window, the LDS-DMA fills, the
// deferred BatchNorm + ReLU of the sources (NaN-sentinel padding), the two source segments (concat), the two cropped destinations,
// the BatchNorm partial sums and the fused BatchNorm-backward dX epilogue are those of gsd_conv3x3_w43.hip (its straight-fill
// form: every 4-channel chunk lies inside one source segment -- always true in the U-Net).
//
// What sets this kernel's rate is its vector-to-MFMA instruction ratio (an fp32 MFMA stream hides LDS reads and scalar instructions
// but not vector instructions: profiles/r05_mfma_f32_issue_ubench.txt), hence: the operand transform, the deferred-BatchNorm affine
// and the output transform on PAIRS of floats (packed fp32 instructions, bit-identical to the scalar form); the chunk loop unrolled
// by two so that the LDS image offsets are instruction immediates; 16-byte halo pieces wherever the source admits them (HM);
// plane offsets of the epilogue as scalar arithmetic.  K slabs (SPLIT) for the launches that would leave the chip idle.
#include "gsd_common.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

__device__ const float gsd_pad_w2d[2] = {0.f, __builtin_nanf("")};

typedef float f32x4v __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

struct W2DParams {
  SrcD src0, src1;
  DstD dst0, dst1;
  const float* wt;   // [mblocks][nchunks][4 ci][12 frequency pairs][2 channel halves][16][2 (f & 1)][2 m-tiles]: gsd_weight_layout modes 8 / 9
  float* partials;   // [pixel tiles * NWP][2 * Mpad]: row (pixel tile, pixel group of the block)
  const float* bw_raw;
  const float* bw_scale;
  const float* bw_shift;
  const float* bw_mean;
  const float* bw_invstd;
  int Cin, Cout, Mpad, nchunks, mblocks;
  int N, H, W;
  int TH, TW, TWq, tiles_y, tiles_x, WR, WC, WCp, PS, NPV;
  int NP, NI;   // X4: 16-byte pieces per window row (TW / 4 + 2), DMA instructions per channel plane
  int mgrp, pgrp, ptiles;   // block order of the deep levels: passes of mgrp m-blocks over groups of pgrp of the ptiles pixel tiles
  int nslab;    // K slabs (SPLIT): block (tile, slab k, m-block) runs chunks [k n / S, (k + 1) n / S) and stores its un-reduced
  float* slabs; // 2 x 4 outputs per channel and Winograd tile to [tile][slab][m-block][64 channels][16 NWP tiles][8] floats
};

#ifndef W2D_PIPE   // 1: pin the interleave of a frequency row's MFMAs with the next row's transform (sched_group_barrier)
#define W2D_PIPE 1
#endif
#ifndef W2D_PK   // 1: the operand transform on pairs of floats (v_pk_add_f32 / v_pk_fma_f32): same operations in the same order on
#define W2D_PK 1  // every element -- bit-identical -- in about half the vector instructions
#endif
#ifndef W2D_ABL   // diagnostic builds only (profiles/build_diag_one.sh; results are then garbage): 1 no weight fills after a block's first,
#define W2D_ABL 0 // 2 no halo fills after the first, 4 barrier without the wait for the fills, 8 no MFMAs, 16 no operand transform,
                  // 32 no barrier (own fills only)
#endif
namespace {
constexpr int W2D_BM = 64;
constexpr int W2D_WTILE = 96 * W2D_BM;   // floats per weight chunk (24 KiB)
}  // namespace

// ---- epilogue (shared by the conv kernel and the K-slab reducer): a lane holds, per (m-tile, register) = channel, the 2 x 4 outputs
// of its Winograd tile (get_y); NCHW stores (two destination segments with crop), BatchNorm partial sums, or the fused
// BatchNorm-backward form.  sBw: the block's [4][64] coefficients of that form in LDS.
template <int NWP, class GetY>
__device__ __forceinline__ void w2d_epilogue(const W2DParams& P, const float* sBw, const int n, const int h0, const int w0, const int tr2,
                                             const int tq, const int vmask, const int m0, const int mh, const int ph, const int j,
                                             const int l16, const int pt, GetY get_y) {
  constexpr int BM = 64;
  // ---- epilogue: Y = A2^T M A4, NCHW stores (two destination segments with crop), BatchNorm partial sums -------------------------
  // per destination and tile row: element offset of the row's first pixel inside a plane, and the mask of its pixels that are stored
  // (scalars, not arrays: `first ? off0(a) : off1(a)` on arrays makes hipcc select between two ADDRESSES and keep the arrays in scratch)
  int off0_0 = 0, off0_1 = 0, off1_0 = 0, off1_1 = 0, sm0_0 = 0, sm0_1 = 0, sm1_0 = 0, sm1_1 = 0;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int h = h0 + 2 * tr2 + a, w = w0 + 4 * tq;
    const int vm = (vmask >> (4 * a)) & 15;
    int hd = h - P.dst0.oh, wd = w - P.dst0.ow;
    if ((unsigned)hd < (unsigned)P.dst0.H) {
      (a == 0 ? off0_0 : off0_1) = hd * P.dst0.ws + wd;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if ((vm >> i & 1) && (unsigned)(wd + i) < (unsigned)P.dst0.W) (a == 0 ? sm0_0 : sm0_1) |= 1 << i;
    }
    hd = h - P.dst1.oh;
    wd = w - P.dst1.ow;
    if ((unsigned)hd < (unsigned)P.dst1.H) {
      (a == 0 ? off1_0 : off1_1) = hd * P.dst1.ws + wd;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if ((vm >> i & 1) && (unsigned)(wd + i) < (unsigned)P.dst1.W) (a == 0 ? sm1_0 : sm1_1) |= 1 << i;
    }
  }
  auto OFF0 = [&](int a) { return a == 0 ? off0_0 : off0_1; };   // (a is a constant of the unrolled loops)
  auto OFF1 = [&](int a) { return a == 0 ? off1_0 : off1_1; };
  auto SM0 = [&](int a) { return a == 0 ? sm0_0 : sm0_1; };
  auto SM1 = [&](int a) { return a == 0 ? sm1_0 : sm1_1; };
  const long long lane0 = (long long)(j * 4) * P.dst0.cs, lane1 = (long long)(j * 4) * P.dst1.cs;
  float* const d0 = P.dst0.p + (long long)n * P.dst0.ns;
  float* const d1 = P.dst1.p + (long long)n * P.dst1.ns;
  float* const prow = P.partials != nullptr ? P.partials + (size_t)(pt * NWP + ph) * (2 * P.Mpad) : nullptr;

  if (P.bw_raw == nullptr) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        // channel = wave-uniform part cu + the lane's 4 j: the plane offset of cu is scalar arithmetic, the lane's share (lane0 / lane1)
        // is multiplied once -- a 64-bit vector multiply per channel otherwise
        const int cu = m0 + mh * 32 + m * 16 + reg, co = cu + j * 4;
        const bool first = co < P.dst0.C;
        const int cd = first ? co : co - P.dst0.C;
        const bool co_ok = co < P.Cout && (first || cd < P.dst1.C);
        float* const plane = first ? d0 + (long long)cu * P.dst0.cs + lane0 : d1 + (long long)(cu - P.dst0.C) * P.dst1.cs + lane1;
        float y[2][4];
        get_y(m, reg, y);
        // statistics over the pixels that are STORED (for a cropped second destination -- the backward of F.pad -- the sums are
        // those of the crop, e.g. the ConvT bias gradient)
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int sm = co_ok ? (first ? SM0(a) : SM1(a)) : 0;
          float* const px = plane + (first ? OFF0(a) : OFF1(a));
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (sm >> i & 1) {
              s1 += y[a][i];
              s2 = fmaf(y[a][i], y[a][i], s2);
            }
          }
          if (sm == 15) {
            *reinterpret_cast<f32x4v*>(px) = f32x4{y[a][0], y[a][1], y[a][2], y[a][3]};
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (sm >> i & 1) px[i] = y[a][i];
          }
        }
        if (prow != nullptr) {
          s1 = reduce16_to_lane15(s1);
          s2 = reduce16_to_lane15(s2);
          if (l16 == 15 && co < P.Mpad) {
            prow[co] = s1;
            prow[P.Mpad + co] = s2;
          }
        }
      }
    }
  } else {
    // dst0 is the gradient buffer of a conv+BN+ReLU unit whose raw output has the same geometry: dz = relu'(bn(raw)) * dX
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int cu = m0 + mh * 32 + m * 16 + reg, co = cu + j * 4;
        const long long cplane = (long long)n * P.dst0.ns + (co < P.Cout ? (long long)cu * P.dst0.cs + lane0 : 0);
        float xr[2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const float* const rp = P.bw_raw + cplane + OFF0(a);
          if (SM0(a) == 15) {
            const f32x4 t = *reinterpret_cast<const f32x4v*>(rp);
            xr[a][0] = t[0], xr[a][1] = t[1], xr[a][2] = t[2], xr[a][3] = t[3];
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) xr[a][i] = (SM0(a) >> i & 1) ? rp[i] : 0.f;
          }
        }
        const int cl = mh * 32 + m * 16 + j * 4 + reg;
        const float bsc = sBw[cl], bsh = sBw[BM + cl], bmu = sBw[2 * BM + cl], bis = sBw[3 * BM + cl];
        float y[2][4];
        get_y(m, reg, y);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int sm = co < P.Cout ? SM0(a) : 0;
          float* const px = d0 + ((long long)cu * P.dst0.cs + lane0) + OFF0(a);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float x = xr[a][i];
            const float dz = ((sm >> i & 1) && fmaf(x, bsc, bsh) > 0.f) ? y[a][i] : 0.f;
            y[a][i] = dz;
            s1 += dz;
            s2 = fmaf(dz, (x - bmu) * bis, s2);
          }
          if (sm == 15) {
            *reinterpret_cast<f32x4v*>(px) = f32x4{y[a][0], y[a][1], y[a][2], y[a][3]};
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (sm >> i & 1) px[i] = y[a][i];
          }
        }
        if (prow != nullptr) {
          s1 = reduce16_to_lane15(s1);
          s2 = reduce16_to_lane15(s2);
          if (l16 == 15 && co < P.Mpad) {
            prow[co] = s1;
            prow[P.Mpad + co] = s2;
          }
        }
      }
    }
  }
}

// PLAIN: no source segment carries a deferred BatchNorm or ReLU (every dX launch; the pooled / up-sampled sources of the forward)
// NWP: pixel groups of 16 tiles (128 pixels) per block.  2: four waves, 64 channels x 256 pixels, two blocks per CU.  4: eight
// waves, 64 x 512 pixels, one block per CU -- the 24-KiB weight chunk then feeds twice the MFMAs: with a third of the direct
// form's multiplications the L2 -> LDS fills (30 KiB per 192 MFMAs in the four-wave form) are what the kernel waits for.
//
// X4: the halo windows move as ALIGNED 16-byte pieces (global_load_lds_dwordx4) instead of dword gathers: a quarter of the halo's
// DMA instructions (2 instead of 6 per channel plane of a 10 x 34 window), and the LDS-DMA issue -- 60-180 cycles an instruction --
// is what this kernel waits for besides its MFMAs.  Possible when every source row starts 16-byte aligned and a piece lies wholly
// inside or wholly outside a row: ONE plain source segment with a row pitch that is a multiple of 4 floats whose pad columns hold
// zeros -- the row-pitched d_raw buffer every dX launch reads (gsd_bn_bwd_apply's out-of-place form).  A window row is the NP =
// TW/4 + 2 pieces that cover image columns w0-4 .. w0+TW+3; the planes are shifted by ONE float in LDS so that image column w0-1
// lands 16-byte aligned and the consumer reads stay one b128 + one b64 per window row.
//
// HM = 2 ("U4"): the same 16-byte pieces on the same w0-4 piece grid, straight from UNALIGNED rows -- any source (a
// global_load_lds_dwordx4 takes any 4-byte aligned global address at full rate).  On that grid a piece never straddles the LEFT
// image edge of a segment that starts at column 0; one that straddles a segment's right edge (W % 4 != 0: every level of the
// U-Net) is loaded as it lies in memory -- the caller vouches for 4 readable floats around the tensor, gsd_src.slack -- and the
// lane that moved it overwrites its outside floats with the padding value once its own fills have landed, in front of the chunk's
// barrier (only lanes of blocks at that edge do anything).
//
// SPLIT (K slabs, as gsd_conv3x3_w43.hip): a launch whose tile grid leaves most of the chip's 512 block slots empty (the 40 x 53 and
// 20 x 26 levels at small batches) is cut along the input channels; the output transform is linear, so each slab stores its own
// Y = A2^T M A4 and w2d_slab_reduce_kernel adds the slabs in slab order and runs the epilogue.  A slab that starts inside the
// second (concat) segment starts its fills there.
template <bool PLAIN, int NWP, int HM = 0, bool SPLIT = false>
__global__ __launch_bounds__(128 * NWP, 2) void conv3x3_w2d_kernel(const W2DParams P) {
  constexpr bool X4 = HM == 1, U4 = HM == 2, PC = HM != 0;   // PC: the halo lies in LDS as 16-byte pieces
  constexpr int W2D_NONE = -2147483647 - 1, W2D_PAD = -2147483647;   // lane offsets: no position / a padding position (prefilled)
  static_assert(!X4 || PLAIN, "aligned 16-byte halo pieces: a plain, row-pitched source");
  constexpr int BM = W2D_BM, WTILE = W2D_WTILE, NT = 128 * NWP, NW = 2 * NWP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int PS = P.PS;
  const int BUF = WTILE + 4 * PS;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ph = wave8 % NWP, mh = wave8 / NWP;   // the wave's pixel group (16 of the block's 16 NWP tiles) and 32-channel half
  const int j = lane >> 4, l16 = lane & 15;

  // the m-blocks of one pixel tile read the same halo: every XCD gets a contiguous range of logical ids (pixel tile major)
  const int lid = xcd_swizzle(blockIdx.x, gridDim.x);
  int mbb = lid % P.mblocks;
  const int slab = SPLIT ? (lid / P.mblocks) % P.nslab : 0;
  int pt = SPLIT ? lid / P.mblocks / P.nslab : lid / P.mblocks;
  if (!SPLIT && P.mgrp < P.mblocks) {
    // Deep levels: an m-block's weight image (24 KiB per chunk) is megabytes, and the blocks in flight on an XCD are at arbitrary
    // phases of their chunk loops once the first round is over -- with every m-block of a pixel tile in flight at once, the eight or
    // sixteen weight streams evict each other from the 4-MiB L2 and nearly every weight fill misses it (measured: 7.0 GB of L2 misses
    // per 40 x 53 512 -> 512 launch, the weight fills are 7.9 GB).  So the ids walk groups of `pgrp` pixel tiles, and inside a group
    // the m-blocks in passes of `mgrp` (as many as fit the L2): the blocks in flight share mgrp weight images; the tile windows
    // (a quarter of the weights' bytes) are what a later pass reads again.
    const int gsz = P.pgrp * P.mblocks;
    const int g = lid / gsz;
    int r = lid - g * gsz;
    const int p0 = g * P.pgrp;
    const int pn = min(P.pgrp, P.ptiles - p0);          // (the last pixel group may be short)
    const int per = pn * P.mgrp;
    const int mg = r / per;
    r -= mg * per;
    const int gn = min(P.mgrp, P.mblocks - mg * P.mgrp);   // (and the last pass)
    const int pl = r / gn;
    pt = p0 + pl;
    mbb = mg * P.mgrp + (r - pl * gn);
  }
  const int c_lo = SPLIT ? (int)((long)slab * P.nchunks / P.nslab) : 0;
  const int c_hi = SPLIT ? (int)((long)(slab + 1) * P.nchunks / P.nslab) : P.nchunks;
  const int m0 = mbb * BM;
  const int tpi = P.tiles_y * P.tiles_x;
  const int n = pt / tpi;
  const int rt = pt - n * tpi;
  const int ty = rt / P.tiles_x;
  const int h0 = ty * P.TH, w0 = (rt - ty * P.tiles_x) * P.TW;

  // ---- this lane's Winograd tile: 2 x 4 pixels (rows 2*tr2, 2*tr2+1; columns 4*tq .. 4*tq+3) of the block's TH x TW tile ------
  const int q = ph * 16 + l16;
  const bool q_ok = q < (P.TH >> 1) * P.TWq && q < 16 * NWP;
  const int tr2 = q_ok ? q / P.TWq : 0;
  const int tq = q_ok ? q - tr2 * P.TWq : 0;
  const int baddr = WTILE + j * PS + (2 * tr2) * P.WCp + 4 * tq + (PC ? 4 : 0);   // halo columns 4*tq .. 4*tq+5 of halo rows 2*tr2 .. 2*tr2+3
  int vmask = 0;   // bits 0..3: pixels of the tile's first row that exist in the image, bits 4..7: of its second row
  if (q_ok) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
      if (h0 + 2 * tr2 + a < P.H) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (w0 + 4 * tq + i < P.W) vmask |= 1 << (4 * a + i);
      }
  }

  // ---- halo DMA lane geometry: the block's waves cover the (up to) 128 NW window positions once, dword gathers --------------------
  int xo0[2], xo1[2];
  bool p_on[2];
  int pmask = 0;   // U4: floats of this lane's pieces that lie outside their row (bits 4 pp .. 4 pp + 3: first segment, + 8: second)
#pragma unroll
  for (int pp = 0; pp < 2; ++pp) {
    xo0[pp] = xo1[pp] = W2D_NONE;
    if constexpr (PC) {
      // unit u = (channel plane u / NI, instruction u % NI) of the chunk: its 64 lanes are 64 consecutive pieces of the plane
      const int u = wave8 + NW * pp;
      p_on[pp] = u < 4 * P.NI;
      const int piece = (u % P.NI) * 64 + lane;
      const int rr = piece / P.NP, pc = piece - rr * P.NP;
      if (rr < P.WR) {
        const int gh = h0 - 1 + rr, gw = w0 - 4 + 4 * pc;
        if constexpr (X4) {
          xo0[pp] = ((unsigned)gh < (unsigned)P.src0.H && gw >= 0 && gw + 4 <= P.src0.ws) ? gh * P.src0.ws + gw : W2D_PAD;
        } else {
          int hs = gh - P.src0.oh, c0 = gw - P.src0.ow;
          xo0[pp] = W2D_PAD;
          if ((unsigned)hs < (unsigned)P.src0.H && c0 + 3 >= 0 && c0 < P.src0.W) {
            xo0[pp] = hs * P.src0.ws + c0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (c0 + e < 0 || c0 + e >= P.src0.W) pmask |= 1 << (4 * pp + e);
          }
          hs = gh - P.src1.oh;
          c0 = gw - P.src1.ow;
          xo1[pp] = W2D_PAD;
          if (P.src1.C > 0 && (unsigned)hs < (unsigned)P.src1.H && c0 + 3 >= 0 && c0 < P.src1.W) {
            xo1[pp] = hs * P.src1.ws + c0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (c0 + e < 0 || c0 + e >= P.src1.W) pmask |= 1 << (8 + 4 * pp + e);
          }
        }
      }
      continue;
    }
    p_on[pp] = wave8 + NW * pp < P.NPV;
    const int pos = (wave8 + NW * pp) * 64 + lane;
    const int rr = pos / P.WCp, cc = pos - rr * P.WCp;
    if (rr < P.WR && cc < P.WC) {
      const int gh = h0 - 1 + rr, gw = w0 - 1 + cc;
      int hs = gh - P.src0.oh, ws = gw - P.src0.ow;
      xo0[pp] = ((unsigned)hs < (unsigned)P.src0.H && (unsigned)ws < (unsigned)P.src0.W) ? hs * P.src0.ws + ws : W2D_PAD;
      hs = gh - P.src1.oh;
      ws = gw - P.src1.ow;
      xo1[pp] = ((unsigned)hs < (unsigned)P.src1.H && (unsigned)ws < (unsigned)P.src1.W) ? hs * P.src1.ws + ws : W2D_PAD;
    }
  }
  const int f_sw = P.src1.C > 0 ? P.src0.C / 4 : -1;           // first chunk of the second (concat) segment
  const bool start1 = SPLIT && f_sw >= 0 && c_lo >= f_sw;      // this slab's chunks all lie in the second segment
  {
    // padding positions of the block's first segment, once, in all 2 x 4 channel planes (own positions only); visible after the first barrier
    const float pad0 = (start1 ? P.src1.relu : P.src0.relu) ? __builtin_nanf("") : 0.f;
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
      if (p_on[pp] && (start1 ? xo1[pp] : xo0[pp]) == W2D_PAD) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          if constexpr (PC) {   // the unit's own plane
            const int u = wave8 + NW * pp;
#pragma unroll
            for (int e = 0; e < 4; ++e) smem[b * BUF + WTILE + (u / P.NI) * PS + 1 + (u % P.NI) * 256 + lane * 4 + e] = pad0;
          } else {
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) smem[b * BUF + WTILE + ch * PS + (wave8 + NW * pp) * 64 + lane] = pad0;
          }
        }
      }
  }
  long long d_cs = start1 ? P.src1.cs : P.src0.cs;
  // channel plane of the next halo slot: the slab's first channel inside its segment
  const float* d_base = (start1 ? P.src1.p + (long long)n * P.src1.ns : P.src0.p + (long long)n * P.src0.ns) +
                        (long long)(c_lo - (start1 ? f_sw : 0)) * 4 * d_cs;
  long long f_xl[2];   // the current segment's lane offsets as 64-bit values (the address add is then a single instruction)
#pragma unroll
  for (int pp = 0; pp < 2; ++pp) f_xl[pp] = start1 ? xo1[pp] : xo0[pp];
  // the switch to the second segment happens once per block, between two chunks: new plane pointer and lane offsets, and that
  // segment's padding positions are written into each LDS image the first time it is filled from it
  auto begin_fill = [&](int chunk, int buf) {
    if (f_sw < 0 || start1 || (chunk != f_sw && chunk != f_sw + 1)) return;
    if (chunk == f_sw) {
      d_base = P.src1.p + (long long)n * P.src1.ns;
      d_cs = P.src1.cs;
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) f_xl[pp] = xo1[pp];
    }
    const float pad1 = P.src1.relu ? __builtin_nanf("") : 0.f;
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
      if (p_on[pp] && f_xl[pp] == W2D_PAD) {
        if constexpr (PC) {
          const int u = wave8 + NW * pp;
#pragma unroll
          for (int e = 0; e < 4; ++e) smem[buf * BUF + WTILE + (u / P.NI) * PS + 1 + (u % P.NI) * 256 + lane * 4 + e] = pad1;
        } else {
#pragma unroll
          for (int ch = 0; ch < 4; ++ch) smem[buf * BUF + WTILE + ch * PS + (wave8 + NW * pp) * 64 + lane] = pad1;
        }
      }
  };
  auto halo_slot = [&](int ch, float* Xb) {   // input channel ch of the chunk: the lanes that have a pixel move it
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
      if (p_on[pp] && f_xl[pp] > W2D_PAD) __builtin_amdgcn_global_load_lds(d_base + f_xl[pp], Xb + ch * PS + (wave8 + NW * pp) * 64, 4, 0, 0);
    d_base += d_cs;
  };
  // X4: unit pp of this wave (one instruction of one of the chunk's four planes); d_base stays at the chunk's first plane
  auto halo_unit = [&](int pp, float* Xb) {
    const int u = wave8 + NW * pp;
    if (p_on[pp] && f_xl[pp] > W2D_PAD)
      __builtin_amdgcn_global_load_lds(d_base + (u / P.NI) * d_cs + f_xl[pp], Xb + (u / P.NI) * PS + 1 + (u % P.NI) * 256, 16, 0, 0);
    if (pp == 1) d_base += 4 * d_cs;
  };
  constexpr int WPW = 24 / NW;   // 1-KiB weight pieces per wave and chunk (6 or 3)
  const float* const wsrc0 = P.wt + (size_t)mbb * P.nchunks * WTILE + wave8 * (WPW * 256) + lane * 4;
  // the wave's pieces of the 24 are adjacent: they share LDS bases (M0) and differ in the instruction's immediate offset, which
  // moves the global and the LDS address alike
  auto weight_fill = [&](int chunk, float* Wn) {
    const float* wg = wsrc0 + (size_t)chunk * WTILE;
    float* wl = Wn + wave8 * (WPW * 256);
    __builtin_amdgcn_global_load_lds(wg, wl, 16, 0, 0);
    __builtin_amdgcn_global_load_lds(wg, wl, 16, 1024, 0);
    __builtin_amdgcn_global_load_lds(wg, wl, 16, 2048, 0);
    if constexpr (WPW == 6) {
      __builtin_amdgcn_global_load_lds(wg, wl, 16, 3072, 0);
      __builtin_amdgcn_global_load_lds(wg + 1024, wl + 1024, 16, 0, 0);
      __builtin_amdgcn_global_load_lds(wg + 1024, wl + 1024, 16, 1024, 0);
    }
  };

  const int Kpad = P.nchunks * 4;
  float* sAff = smem + 2 * BUF;
  for (int c = tid; c < Kpad; c += NT) {
    const bool first = c < P.src0.C;
    const SrcD& S = first ? P.src0 : P.src1;
    const int cc = first ? c : c - P.src0.C;
    float sc = 1.f, sh = 0.f;
    if (c < P.Cin && cc < S.C && S.scale != nullptr) {
      sc = S.scale[cc];
      sh = S.shift[cc];
    }
    sAff[c] = sc;
    sAff[Kpad + c] = sh;
  }
  float* sBw = sAff + 2 * Kpad;   // [4][64]: scale, shift, mean, invstd of the fused BatchNorm-backward epilogue
  if (P.bw_raw != nullptr) {
    for (int c = tid; c < BM; c += NT) {
      const int co = m0 + c < P.Cout ? m0 + c : 0;
      sBw[c] = P.bw_scale[co];
      sBw[BM + c] = P.bw_shift[co];
      sBw[2 * BM + c] = P.bw_mean[co];
      sBw[3 * BM + c] = P.bw_invstd[co];
    }
  }
  const float lo0 = P.src0.relu ? 0.f : -__builtin_inff(), lo1 = P.src1.relu ? 0.f : -__builtin_inff();

  f32x4 acc[2][24];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int f = 0; f < 24; ++f) acc[m][f] = f32x4{0.f, 0.f, 0.f, 0.f};

  // V row = B4^T t of one frequency row: the 6 values of a (column-transformed) window row -> the 6 row frequencies
  auto row_transform = [&](const float (&d)[6], float (&v)[6]) {
    const float a = fmaf(-4.f, d[2], d[4]), b = fmaf(-4.f, d[1], d[3]);
    const float c = d[4] - d[2], e = 2.f * (d[3] - d[1]);
    v[0] = fmaf(4.f, d[0], fmaf(-5.f, d[2], d[4]));
    v[1] = a + b;
    v[2] = a - b;
    v[3] = c + e;
    v[4] = c - e;
    v[5] = fmaf(4.f, d[1], fmaf(-5.f, d[3], d[5]));
  };

  const int a_lane = mh * 64 + l16 * 4;   // this wave's (f, f+1) x two m-tiles of a frequency pair: 16 lanes read 256 contiguous bytes
  begin_fill(c_lo, 0);
  weight_fill(c_lo, smem);
  if constexpr (PC) {
    halo_unit(0, smem + WTILE);
    halo_unit(1, smem + WTILE);
  } else {
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) halo_slot(ch, smem + WTILE);
  }

  // the chunk loop, unrolled by two: the LDS image a chunk reads (`cur`) is then a constant of each copy, and the image offsets
  // fold into the instructions' immediate fields instead of costing an address addition per base register and chunk
  auto run_chunk = [&](const int chunk, auto cur_c) {
    constexpr int cur = decltype(cur_c)::value;
    if constexpr (U4) {
      __builtin_amdgcn_s_waitcnt(0x0F70);   // this wave's fills of the chunk have landed
      // the outside floats of the straddling pieces this lane moved (the lane state still is the one the chunk was filled with)
      const bool seg1 = f_sw >= 0 && chunk >= f_sw;
      const int pm = seg1 ? pmask >> 8 : pmask & 0xff;
      if (pm != 0) {
        const float padv = (seg1 ? P.src1.relu : P.src0.relu) ? __builtin_nanf("") : 0.f;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
          const int u = wave8 + NW * pp;
          float* pq = smem + cur * BUF + WTILE + (u / P.NI) * PS + 1 + (u % P.NI) * 256 + lane * 4;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (pm >> (4 * pp + e) & 1) pq[e] = padv;
        }
      }
      __syncthreads();
    } else if ((W2D_ABL) & 32) {
      __builtin_amdgcn_s_waitcnt(0x0F70);   // diagnostic: own fills only, no barrier at all (racy: what the barrier itself costs)
    } else if ((W2D_ABL) & 4) {
      __syncthreads();
    } else {
      gsd_dma_barrier();   // the chunk's fills have landed; everyone has left the other image
    }
    const int kc = chunk * 4 + j;
    float sc = 1.f, sh = 0.f, lo = 0.f;
    if constexpr (!PLAIN) {
      sc = sAff[kc], sh = sAff[Kpad + kc];
      lo = kc < P.src0.C ? lo0 : (kc < P.Cin ? lo1 : -__builtin_inff());
    }
    const bool more = chunk + 1 < c_hi;
    const float* Wc = smem + cur * BUF;
    float d[4][6];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 ra = *reinterpret_cast<const f32x4*>(&Wc[baddr + i * P.WCp]);
      const f32x2v rb = *reinterpret_cast<const f32x2v*>(&Wc[baddr + i * P.WCp + 4]);
      d[i][0] = ra[0], d[i][1] = ra[1], d[i][2] = ra[2], d[i][3] = ra[3], d[i][4] = rb[0], d[i][5] = rb[1];
    }
    // A operands: one ds_read_b128 = this wave's two m-tiles of TWO consecutive frequencies (the weight image pairs them), read one
    // pair (four MFMAs) ahead
    f32x4 av[2];
    av[0] = *reinterpret_cast<const f32x4*>(&Wc[(j * 12) * 128 + a_lane]);
    if constexpr (!PLAIN) {
#if W2D_PK
      const f32x2v sc2 = {sc, sc}, sh2 = {sh, sh};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 6; c += 2) {   // (the fused multiply-add on pairs; there is no packed fp32 max)
          const f32x2v y = __builtin_elementwise_fma(f32x2v{d[i][c], d[i][c + 1]}, sc2, sh2);
          d[i][c] = fmaxf(y[0], lo);
          d[i][c + 1] = fmaxf(y[1], lo);
        }
#else
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 6; ++c) d[i][c] = fmaxf(fmaf(d[i][c], sc, sh), lo);
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
    // frequency rows in the order that retires window rows early: t0 = d0 - d2, t3 = d1 - d3, t1 = d1 + d2, t2 = d2 - d1.
    // Software pipeline over the rows: the operand transform of row fi + 1 (about 19 vector instructions) is issued between the
    // 12 MFMAs of row fi -- an MFMA holds the SIMD's vector issue for 8 of its 32 cycles, three vector instructions fit its shadow.
    constexpr int FR[4] = {0, 3, 1, 2};
    auto freq_row = [&](int fr, float (&v)[6]) {
#if (W2D_PK) && !((W2D_ABL) & 16)
      // pairs (t0,t1), (t2,t3), (t4,t5) of the column-transformed row, then
      //   (a, c) = t4 + (-4,-1) t2      (b, e) = t3 + (-4,-1) t1      (v1, v2) = a + (1,-1) b      (v3, v4) = c + (2,-2) e
      //   (v0, v5) = 4 (t0,t1) + ((t4,t5) - 5 (t2,t3))
      // -- element for element the fused multiply-adds of row_transform (a multiplication by 1, 2 or -1 is exact)
      f32x2v tp[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const f32x2v r0 = {d[0][2 * k], d[0][2 * k + 1]}, r1 = {d[1][2 * k], d[1][2 * k + 1]};
        const f32x2v r2 = {d[2][2 * k], d[2][2 * k + 1]}, r3 = {d[3][2 * k], d[3][2 * k + 1]};
        tp[k] = fr == 0 ? r0 - r2 : fr == 3 ? r1 - r3 : fr == 1 ? r1 + r2 : r2 - r1;
      }
      const f32x2v m41 = {-4.f, -1.f}, p1m1 = {1.f, -1.f}, p2m2 = {2.f, -2.f}, m5 = {-5.f, -5.f}, p4 = {4.f, 4.f};
      const f32x2v ac = __builtin_elementwise_fma(f32x2v{tp[1][0], tp[1][0]}, m41, f32x2v{tp[2][0], tp[2][0]});
      const f32x2v be = __builtin_elementwise_fma(f32x2v{tp[0][1], tp[0][1]}, m41, f32x2v{tp[1][1], tp[1][1]});
      const f32x2v v12 = __builtin_elementwise_fma(f32x2v{be[0], be[0]}, p1m1, f32x2v{ac[0], ac[0]});
      const f32x2v v34 = __builtin_elementwise_fma(f32x2v{be[1], be[1]}, p2m2, f32x2v{ac[1], ac[1]});
      const f32x2v v05 = __builtin_elementwise_fma(tp[0], p4, __builtin_elementwise_fma(tp[1], m5, tp[2]));
      v[0] = v05[0], v[1] = v12[0], v[2] = v12[1], v[3] = v34[0], v[4] = v34[1], v[5] = v05[1];
      return;
#endif
      float t[6];
#pragma unroll
      for (int c = 0; c < 6; ++c)
        t[c] = fr == 0 ? d[0][c] - d[2][c] : fr == 3 ? d[1][c] - d[3][c] : fr == 1 ? d[1][c] + d[2][c] : d[2][c] - d[1][c];
#if (W2D_ABL) & 16
#pragma unroll
      for (int c = 0; c < 6; ++c) v[c] = t[c];
#else
      row_transform(t, v);
#endif
    };
    float v[2][6];
    freq_row(FR[0], v[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int fi = 0; fi < 4; ++fi) {
      const int fr = FR[fi];
      if (fi + 1 < 4) freq_row(FR[fi + 1], v[(fi + 1) & 1]);
#pragma unroll
      for (int fc = 0; fc < 6; ++fc) {
        const int s = fi * 6 + fc, f = fr * 6 + fc;
        if ((s & 1) == 0 && s + 2 < 24) {
          const int fn = FR[(s + 2) / 6] * 6 + (s + 2) % 6;
          av[((s >> 1) + 1) & 1] = *reinterpret_cast<const f32x4*>(&Wc[(j * 12 + (fn >> 1)) * 128 + a_lane]);
        }
        const f32x4& ap = av[(s >> 1) & 1];
#if (W2D_ABL) & 8
        acc[0][f][0] += ap[(s & 1) * 2] * v[fi & 1][fc];
        acc[1][f][0] += ap[(s & 1) * 2 + 1] * v[fi & 1][fc];
#else
        acc[0][f] = mfma16(ap[(s & 1) * 2], v[fi & 1][fc], acc[0][f]);
        acc[1][f] = mfma16(ap[(s & 1) * 2 + 1], v[fi & 1][fc], acc[1][f]);
#endif
        // the next chunk's fills ride in the first k-steps: the weights in one k-step (shared LDS bases), then the halo
        if (more && s < 3) {
          float* Wn = smem + (cur ^ 1) * BUF;
          if (s == 0) {
            begin_fill(chunk + 1, cur ^ 1);
            if (!((W2D_ABL) & 1)) weight_fill(chunk + 1, Wn);
          } else if (!((W2D_ABL) & 2)) {
            if constexpr (PC) {
              halo_unit(s - 1, Wn + WTILE);
            } else {
              halo_slot(2 * s - 2, Wn + WTILE);
              halo_slot(2 * s - 1, Wn + WTILE);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#if W2D_PIPE
      if (fi > 0 || !more) {   // (row 0 carries the fills: its k-steps are pinned above)
#pragma unroll
        for (int g = 0; g < 6; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // two MFMAs
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // at most one LDS read
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // four vector instructions of the next row's transform
        }
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  for (int chunk = c_lo; chunk < c_hi; chunk += 2) {
    run_chunk(chunk, std::integral_constant<int, 0>{});
    if (chunk + 1 < c_hi) run_chunk(chunk + 1, std::integral_constant<int, 1>{});
  }

  // Y = A2^T M A4 of one channel's tile: down the columns first (24 -> 12 values), then along the rows (12 -> 2 x 4 outputs)
  auto out_transform = [&](int m, int reg, float (&y)[2][4]) __attribute__((always_inline)) {
#if W2D_PK
    // two channels at once: accumulator registers (2 rp, 2 rp + 1) of a quad are an aligned pair, so the same additions and fused
    // multiply-adds run as v_pk_add_f32 / v_pk_fma_f32; an odd reg takes the second halves of what its even neighbour computed
    // (the compiler merges the two calls' identical packed instructions)
    const int r0 = reg & ~1;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      f32x2v R[6];
#pragma unroll
      for (int fc = 0; fc < 6; ++fc) {
        const f32x2v M0 = {acc[m][fc][r0], acc[m][fc][r0 + 1]}, M1 = {acc[m][6 + fc][r0], acc[m][6 + fc][r0 + 1]};
        const f32x2v M2 = {acc[m][12 + fc][r0], acc[m][12 + fc][r0 + 1]}, M3 = {acc[m][18 + fc][r0], acc[m][18 + fc][r0 + 1]};
        R[fc] = a == 0 ? M0 + M1 + M2 : M1 - M2 - M3;
      }
      const f32x2v p12 = R[1] + R[2], m12 = R[1] - R[2], p34 = R[3] + R[4], m34 = R[3] - R[4];
      const f32x2v c2 = {2.f, 2.f}, c4 = {4.f, 4.f}, c8 = {8.f, 8.f};
      const f32x2v y0 = R[0] + p12 + p34, y1 = __builtin_elementwise_fma(c2, m34, m12), y2 = __builtin_elementwise_fma(c4, p34, p12);
      const f32x2v y3 = __builtin_elementwise_fma(c8, m34, m12) + R[5];
      y[a][0] = y0[reg & 1], y[a][1] = y1[reg & 1], y[a][2] = y2[reg & 1], y[a][3] = y3[reg & 1];
    }
    return;
#endif
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      float R[6];
#pragma unroll
      for (int fc = 0; fc < 6; ++fc) {
        const float M1 = acc[m][6 + fc][reg], M2 = acc[m][12 + fc][reg];
        R[fc] = a == 0 ? acc[m][fc][reg] + M1 + M2 : M1 - M2 - acc[m][18 + fc][reg];
      }
      const float p12 = R[1] + R[2], m12 = R[1] - R[2], p34 = R[3] + R[4], m34 = R[3] - R[4];
      y[a][0] = R[0] + p12 + p34;
      y[a][1] = fmaf(2.f, m34, m12);
      y[a][2] = fmaf(4.f, p34, p12);
      y[a][3] = fmaf(8.f, m34, m12) + R[5];
    }
  };

  if constexpr (SPLIT) {
    // the un-reduced outputs of this slab: 32 bytes per lane and channel, 512-byte runs per 16 lanes
    float* const sl = P.slabs + ((size_t)((size_t)pt * P.nslab + slab) * P.mblocks + mbb) * (size_t)(BM * 128 * NWP);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        float y[2][4];
        out_transform(m, reg, y);
        float* const o = sl + ((size_t)(mh * 32 + m * 16 + j * 4 + reg) * (16 * NWP) + q) * 8;
        *reinterpret_cast<f32x4*>(o) = f32x4{y[0][0], y[0][1], y[0][2], y[0][3]};
        *reinterpret_cast<f32x4*>(o + 4) = f32x4{y[1][0], y[1][1], y[1][2], y[1][3]};
      }
    return;
  }
  w2d_epilogue<NWP>(P, sBw, n, h0, w0, tr2, tq, vmask, m0, mh, ph, j, l16, pt, out_transform);
}

// The second half of a K-slab launch: one block per (pixel tile, m-block) with the conv kernel's thread -> (tile, channel) map adds
// the slabs IN SLAB ORDER (run-to-run bitwise) and runs the conv kernel's epilogue on the sums.
template <int NWP>
__global__ __launch_bounds__(128 * NWP) void w2d_slab_reduce_kernel(const W2DParams P) {
  constexpr int BM = W2D_BM;
  __shared__ float sBw[4 * BM];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ph = wave8 % NWP, mh = wave8 / NWP;
  const int j = lane >> 4, l16 = lane & 15;
  const int lid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int mbb = lid % P.mblocks;
  const int pt = lid / P.mblocks;
  const int m0 = mbb * BM;
  const int tpi = P.tiles_y * P.tiles_x;
  const int n = pt / tpi;
  const int rt = pt - n * tpi;
  const int ty = rt / P.tiles_x;
  const int h0 = ty * P.TH, w0 = (rt - ty * P.tiles_x) * P.TW;
  const int q = ph * 16 + l16;
  const bool q_ok = q < (P.TH >> 1) * P.TWq && q < 16 * NWP;
  const int tr2 = q_ok ? q / P.TWq : 0;
  const int tq = q_ok ? q - tr2 * P.TWq : 0;
  int vmask = 0;
  if (q_ok) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
      if (h0 + 2 * tr2 + a < P.H) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (w0 + 4 * tq + i < P.W) vmask |= 1 << (4 * a + i);
      }
  }
  if (P.bw_raw != nullptr) {
    for (int c = tid; c < BM; c += 128 * NWP) {
      const int co = m0 + c < P.Cout ? m0 + c : 0;
      sBw[c] = P.bw_scale[co];
      sBw[BM + c] = P.bw_shift[co];
      sBw[2 * BM + c] = P.bw_mean[co];
      sBw[3 * BM + c] = P.bw_invstd[co];
    }
    __syncthreads();
  }
  const size_t tile_elems = (size_t)(BM * 128 * NWP);
  const float* const s0 = P.slabs + ((size_t)pt * P.nslab * P.mblocks + mbb) * tile_elems;
  auto get_y = [&](int m, int reg, float (&y)[2][4]) __attribute__((always_inline)) {
    const float* o = s0 + ((size_t)(mh * 32 + m * 16 + j * 4 + reg) * (16 * NWP) + q) * 8;
    f32x4 a = *reinterpret_cast<const f32x4*>(o), b = *reinterpret_cast<const f32x4*>(o + 4);
    for (int s = 1; s < P.nslab; ++s) {
      o += (size_t)P.mblocks * tile_elems;
      a += *reinterpret_cast<const f32x4*>(o);
      b += *reinterpret_cast<const f32x4*>(o + 4);
    }
    y[0][0] = a[0], y[0][1] = a[1], y[0][2] = a[2], y[0][3] = a[3];
    y[1][0] = b[0], y[1][1] = b[1], y[1][2] = b[2], y[1][3] = b[3];
  };
  w2d_epilogue<NWP>(P, sBw, n, h0, w0, tr2, tq, vmask, m0, mh, ph, j, l16, pt, get_y);
}

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
namespace {

struct W2DPlan {
  int TH, TW, TWq, tiles_y, tiles_x, mblocks, WR, WC, WCp, PS, nwp;
};

// LDS bank cost of the consumers' halo reads (one ds_read_b128 + one ds_read_b64 per window row; a lane's tile rows are 2 apart):
// sum over the two pixel halves of the LDS cycles per read pair.
int w2d_read_cycles(int TWq, int LP, int PS, int nwp) {
  static const int g128[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                  {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
  int total = 0;
  for (int ph = 0; ph < nwp; ++ph) {
    int addr[64];
    for (int lane = 0; lane < 64; ++lane) {
      const int q = ph * 16 + (lane & 15);
      addr[lane] = (lane >> 4) * PS + 2 * (q / TWq) * LP + 4 * (q % TWq);
    }
    for (int half = 0; half < 2; ++half) {
      for (int g = 0; g < 2; ++g) {   // ds_read_b128: 16-lane groups, 16 slots of 16 B
        int worst = 0;
        for (int slot = 0; slot < 16; ++slot) {
          int distinct = 0, seen[16];
          for (int i = 0; i < 16; ++i) {
            const int a = addr[g128[g][i] + 32 * half];
            if ((a / 4) % 16 != slot) continue;
            bool dup = false;
            for (int k = 0; k < distinct; ++k) dup = dup || seen[k] == a;
            if (!dup) seen[distinct++] = a;
          }
          worst = distinct > worst ? distinct : worst;
        }
        total += worst;
      }
      int worst = 0;   // ds_read_b64 at +4 floats: 32-lane halves, 32 slots of 8 B
      for (int slot = 0; slot < 32; ++slot) {
        int distinct = 0, seen[32];
        for (int i = 0; i < 32; ++i) {
          const int a = addr[i + 32 * half] + 4;
          if ((a / 2) % 32 != slot) continue;
          bool dup = false;
          for (int k = 0; k < distinct; ++k) dup = dup || seen[k] == a;
          if (!dup) seen[distinct++] = a;
        }
        worst = distinct > worst ? distinct : worst;
      }
      total += worst;
    }
  }
  return total;
}

// TH x TW output tile of 16 nwp two-row Winograd tiles (256 or 512 pixels) whose padded halo window fits the 128 NW DMA positions:
// fewest blocks; among equals 32-wide rows, then the widest.  The LDS row pitch and plane stride are the ones with the fewest
// bank conflicts.  GSD_W2D_WAVES = 4 | 8 (tuning): the four-wave (two blocks per CU) or the eight-wave block.
bool plan_w2d(int N, int H, int W, int M, W2DPlan* best) {
  long best_cost = -1;
  const int force_tw = gsd_env_int("GSD_W2D_TW", 0);   // tuning
  const int nwp = gsd_env_int("GSD_W2D_WAVES", 4) == 8 ? 4 : 2;   // measured (profiles/r05_w2d_vs_w43.txt): four waves win
  best->nwp = nwp;
  const int maxpos = 256 * nwp;
  static const int tws[4] = {32, 64, 16, 8};
  for (int k = 0; k < 4; ++k) {
    const int tw = tws[k];
    if (force_tw && tw != force_tw) continue;
    const int twq = tw / 4;
    int th = 2 * (16 * nwp / twq);
    const int wcp0 = round_up(tw + 2, 4);
    if ((th + 2) * wcp0 > maxpos) continue;
    const int ty = ceil_div(H, th);
    th = round_up(ceil_div(H, ty), 2);
    const long blocks = (long)ty * ceil_div(W, tw) * N;
    // 8-pixel rows (32 x 8 tiles) save a few blocks on 213-pixel rows (135 against 140 per image) but their halo is 34 rows of 40
    // bytes -- and too many pieces for the 16-byte fills: they have to save GSD_W2D_TW8_PCT percent (default 8) to be taken
    // (measured: step 97.35 -> 96.66 ms; 8 x 32 instead of 16 x 16 tiles at 320 x 427, 560 against 540 per image: +0.2 ms, not taken)
    const long cost = (blocks * 8 + (tw == 32 ? 0 : tw == 64 ? 1 : tw == 16 ? 2 : 3)) * (tw == 8 ? 100 + gsd_env_int("GSD_W2D_TW8_PCT", 8) : 100);
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      best->TH = th; best->TW = tw; best->TWq = twq;
      best->tiles_y = ty; best->tiles_x = ceil_div(W, tw);
      best->WR = th + 2; best->WC = tw + 2;
    }
  }
  best->mblocks = ceil_div(M, W2D_BM);
  if (best_cost < 0) return false;
  // LDS row pitch and plane stride of the chosen tile: a search over 36 candidates of ~10^5 operations each, i.e. a fraction of a
  // millisecond of HOST time -- per (tile, block form) it is done once and remembered (an idempotent cache like cu_count(): every
  // thread computes the same value, the key is published last)
  // (key, WCp, PS) travel in ONE 64-bit atomic: a reader never sees the key of one entry with the payload of another (with the key
  //  and the payload in separate words two writers of colliding keys could hand a reader a torn pair, and the LDS size would then
  //  be computed from another PS than the kernel's)
  static std::atomic<uint64_t> memo[16];
  const int th = best->TH, tw = best->TW, key = (th << 16) | (tw << 4) | nwp;
  std::atomic<uint64_t>& mm = memo[(th * 7 + tw + nwp) & 15];
  {
    const uint64_t v = mm.load(std::memory_order_acquire);
    if (v != 0 && (int)(v >> 40) == key) {
      best->WCp = (int)(v >> 20) & 0xFFFFF;
      best->PS = (int)v & 0xFFFFF;
      return true;
    }
  }
  const int wcp0 = round_up(tw + 2, 4);
  int bc = -1;
  for (int c = wcp0; c <= wcp0 + 12 && (th + 2) * c <= maxpos; c += 4)
    for (int ps = round_up((th + 2) * c, 4) + 4; ps < round_up((th + 2) * c, 4) + 4 + 36; ps += 4) {
      const int cyc = w2d_read_cycles(best->TWq, c, ps, nwp);
      if (bc < 0 || cyc < bc) {
        bc = cyc;
        best->WCp = c;
        best->PS = ps;
      }
    }
  if (key < (1 << 24) && best->WCp < (1 << 20) && best->PS < (1 << 20))
    mm.store(((uint64_t)key << 40) | ((uint64_t)best->WCp << 20) | (uint64_t)best->PS, std::memory_order_release);
  return true;
}

// X4: plane stride of the shifted planes (row pitch 4 NP floats) with the fewest bank conflicts of the consumers' reads
int w2d_x4_plane_stride(int TWq, int WCp, int WR) {
  static std::atomic<uint64_t> memo[8];   // (key, PS) in one 64-bit atomic, as in plan_w2d
  const int key = (TWq << 20) | (WCp << 8) | WR;
  std::atomic<uint64_t>& mm = memo[(TWq + WR) & 7];
  {
    const uint64_t v = mm.load(std::memory_order_acquire);
    if (v != 0 && (int)(v >> 32) == key) return (int)(v & 0xFFFFFFFFu);
  }
  int best = -1, ps_best = WR * WCp + 4;
  for (int ps = WR * WCp + 4; ps < WR * WCp + 4 + 68; ps += 4) {
    const int c = w2d_read_cycles(TWq, WCp, ps, 2);
    if (best < 0 || c < best) {
      best = c;
      ps_best = ps;
    }
  }
  mm.store(((uint64_t)(unsigned)key << 32) | (uint64_t)(unsigned)ps_best, std::memory_order_release);
  return ps_best;
}

template <bool PLAIN, int NWP, int HM = 0, bool SPLIT = false>
int launch_w2d(const W2DParams& P, int grid, size_t lds, hipStream_t st) {
  static gsd_attr_once big_lds;   // per-device cache of an idempotent launch attribute (gsd_common.h)
  const void* fn = reinterpret_cast<const void*>(&conv3x3_w2d_kernel<PLAIN, NWP, HM, SPLIT>);
  if (hipError_t e = gsd_allow_big_lds(big_lds, fn); e != hipSuccess) {
    gsd_set_error("gsd_conv3x3_w2d: hipFuncSetAttribute: %s", hipGetErrorString(e));
    return GSD_ERR_HIP;
  }
  GSD_REQUIRE(lds <= 160 * 1024, GSD_ERR_UNSUPPORTED, "gsd_conv3x3_w2d: LDS image %zu B too large", lds);
  hipLaunchKernelGGL((conv3x3_w2d_kernel<PLAIN, NWP, HM, SPLIT>), dim3(grid), dim3(128 * NWP), lds, st, P);
  GSD_LAUNCH_CHECK("gsd_conv3x3_w2d");
  if constexpr (SPLIT) {
    hipLaunchKernelGGL((w2d_slab_reduce_kernel<NWP>), dim3(grid / P.nslab), dim3(128 * NWP), 0, st, P);
    GSD_LAUNCH_CHECK("gsd_conv3x3_w2d (slab sums)");
  }
  return GSD_OK;
}

// Modelled run time in microseconds of a launch of `base` (tile, m-block) blocks of `nchunks` chunks cut into S slabs: a CU with
// k = ceil(blocks / 256) blocks runs pairs at 2.07 us per chunk and block (+ 5 us per block) and an odd last block at 0.66 of that;
// the slab sums cost 12 us + the slabs' bytes at 6 TB/s (the constants of gsd_conv3x3_w43.hip's model, the kernel's own rate).
double w2d_time_us(long base, int nchunks, int S, bool bw) {
  const long cus = gsd_cu_count();
  const long k = (base * S + cus - 1) / cus;
  const double cu = (double)(k / 2) + (k & 1 ? 0.66 : 0.0);
  double t = cu * (2.07 * nchunks / S + 5.0);
  if (S > 1) t += 12.0 + (double)(S + 1 + (bw ? 1 : 0)) * base * 65536.0 / 6.0e6;
  return t;
}

// GSD_W2D_SPLIT: 0 / 1 never, S >= 2 that many slabs (tuning); default: what the model picks (a split has to buy 3 %)
int w2d_pick_slabs(long base, int nchunks, bool bw) {
  const int forced = gsd_env_int("GSD_W2D_SPLIT", -1);
  if (forced == 0 || forced == 1) return 1;
  int best = 1;
  double tb = w2d_time_us(base, nchunks, 1, bw) * (forced > 1 ? 1e9 : 0.97);
  for (int S = 2; S <= 8 && nchunks / S >= 8; ++S) {
    if (forced > 1 && S != forced) continue;
    const double t = w2d_time_us(base, nchunks, S, bw);
    if (t < tb) {
      tb = t;
      best = S;
    }
  }
  return best;
}

}  // namespace

// 1: the shape and its operands fit the two-dimensional form (every 4-channel chunk inside one source segment)
extern "C" int gsd_conv3x3_w2d_supported(int Cin, int C0) {
  return (Cin > 0 && Cin % 4 == 0 && C0 > 0 && C0 <= Cin && C0 % 4 == 0) ? 1 : 0;
}

extern "C" int gsd_conv3x3_w2d_partial_rows(int N, int H, int W, int Cout) {
  if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
  W2DPlan p;
  if (!plan_w2d(N, H, W, Cout, &p)) return 0;
  return N * p.tiles_y * p.tiles_x * p.nwp;
}

// MFMA instructions of one launch (all blocks, padding included)
extern "C" int64_t gsd_conv3x3_w2d_mfma_count(int N, int H, int W, int Cin, int Cout) {
  W2DPlan p;
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || !plan_w2d(N, H, W, Cout, &p)) return 0;
  return (int64_t)N * p.tiles_y * p.tiles_x * p.mblocks * ceil_div(Cin, 4) * (2 * p.nwp * 48);
}

// Modelled run time of the launch in microseconds, as gsd_conv3x3_w43_estimate_us: a CU with k = ceil(blocks / 256) blocks runs
// pairs at 2.17 us per chunk and block (+ 5 us per block) and an odd last block at 0.66 of that (fitted to
// profiles/r05_w2d_vs_w43.txt: the 20 x 26 and 40 x 53 layers at batch 8, where k is 1-3).
extern "C" double gsd_conv3x3_w2d_estimate_us(int N, int H, int W, int Cin, int Cout) {
  W2DPlan p;
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || !plan_w2d(N, H, W, Cout, &p)) return 0.0;
  const long blocks = (long)N * p.tiles_y * p.tiles_x * p.mblocks;
  if (p.nwp != 2) return (double)((blocks + gsd_cu_count() - 1) / gsd_cu_count()) * (4.6 * ceil_div(Cin, 4) + 5.0);   // (the eight-wave block: one per CU, twice the pixels)
  return w2d_time_us(blocks, ceil_div(Cin, 4), 1, false);
}

// ... with the K-slab form where it pays (train mode with a workspace: what the engine's launches run)
extern "C" double gsd_conv3x3_w2d_estimate_slabs_us(int N, int H, int W, int Cin, int Cout) {
  W2DPlan p;
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || !plan_w2d(N, H, W, Cout, &p)) return 0.0;
  if (p.nwp != 2) return gsd_conv3x3_w2d_estimate_us(N, H, W, Cin, Cout);
  const long blocks = (long)N * p.tiles_y * p.tiles_x * p.mblocks;
  return w2d_time_us(blocks, ceil_div(Cin, 4), w2d_pick_slabs(blocks, ceil_div(Cin, 4), false), false);
}

// Floats of K-slab scratch a train-mode launch of this shape wants (0: it runs unsplit); the launcher takes the capacity and
// shrinks S to what fits.
extern "C" int64_t gsd_conv3x3_w2d_workspace(int N, int H, int W, int Cin, int Cout) {
  W2DPlan p;
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % 4 != 0 || !plan_w2d(N, H, W, Cout, &p) || p.nwp != 2) return 0;
  const long blocks = (long)N * p.tiles_y * p.tiles_x * p.mblocks;
  const int S = std::max(w2d_pick_slabs(blocks, Cin / 4, false), w2d_pick_slabs(blocks, Cin / 4, true));
  return S > 1 ? (int64_t)blocks * S * (W2D_BM * 256) : 0;
}

extern "C" double gsd_conv3x3_w43_estimate_us(int N, int H, int W, int Cin, int Cout, int slabs);

// 1: a caller that has both forms' weight layouts at hand should run this launch through gsd_conv3x3_w2d instead of
// gsd_conv3x3_w43.  train == 0 (eval-mode inference): always -- the two-dimensional form neither folds rows across images nor
// cuts K slabs, so image i of a batch gets the bits the image alone gets, whatever the batch.  train != 0: the modelled times
// decide (the deep levels at small batches stay with the row form's K slabs).  GSD_CONV_W2D = 0 never, 1 always.
extern "C" int gsd_conv3x3_prefers_w2d(int N, int H, int W, int Cin, int Cout, int train) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin < 16 || Cin % 4 != 0 || Cout <= 0) return 0;
  const int forced = gsd_env_int("GSD_CONV_W2D", -1);
  if (forced == 0 || forced == 1) return forced;
  if (!train) return 1;
  const double a = gsd_conv3x3_w2d_estimate_slabs_us(N, H, W, Cin, Cout), b = gsd_conv3x3_w43_estimate_us(N, H, W, Cin, Cout, 1);
  return a > 0.0 && b > 0.0 && a < b ? 1 : 0;
}

static int w2d_impl(const gsd_src* src, int nsrc, const float* wt, int Cin, int Cout, const gsd_dst* dst, int ndst, float* partials,
                    const float* bw_raw, const float* bw_scale, const float* bw_shift, const float* bw_mean, const float* bw_invstd,
                    int N, int H, int W, void* stream, float* ws = nullptr, int64_t ws_elems = 0) {
  GSD_REQUIRE(src && dst && wt, GSD_ERR_BAD_ARG, "gsd_conv3x3_w2d: null argument");
  GSD_REQUIRE(nsrc >= 1 && nsrc <= 2 && ndst >= 1 && ndst <= 2, GSD_ERR_BAD_ARG, "gsd_conv3x3_w2d: nsrc/ndst must be 1 or 2");
  GSD_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, GSD_ERR_BAD_ARG, "gsd_conv3x3_w2d: bad sizes");
  GSD_REQUIRE(H < 32768 && W < 32768, GSD_ERR_UNSUPPORTED, "gsd_conv3x3_w2d: H, W must be < 32768");
  GSD_REQUIRE(((uintptr_t)wt & 15) == 0, GSD_ERR_BAD_ARG, "gsd_conv3x3_w2d: weight layout must be 16-byte aligned");
  GSD_REQUIRE(gsd_conv3x3_w2d_supported(Cin, src[0].C), GSD_ERR_UNSUPPORTED,
              "gsd_conv3x3_w2d: Cin=%d and the first segment's %d channels must be multiples of 4 (use gsd_conv3x3_w43)", Cin, src[0].C);
  int csum = 0;
  for (int i = 0; i < nsrc; ++i) {
    if (int e = gsd_check_src(src[i], "gsd_conv3x3_w2d src", true)) return e;
    GSD_REQUIRE(src[i].scale == nullptr || src[i].relu != 0, GSD_ERR_UNSUPPORTED,
                "gsd_conv3x3_w2d: an affine source segment must also have relu (zero padding uses a NaN sentinel)");
    GSD_REQUIRE((int64_t)src[i].H * src[i].w_stride < (1LL << 31), GSD_ERR_UNSUPPORTED, "gsd_conv3x3_w2d: plane too large");
    csum += src[i].C;
  }
  GSD_REQUIRE(csum == Cin, GSD_ERR_BAD_ARG, "gsd_conv3x3_w2d: source segments hold %d channels, Cin=%d", csum, Cin);
  csum = 0;
  for (int i = 0; i < ndst; ++i) {
    if (int e = gsd_check_dst(dst[i], "gsd_conv3x3_w2d dst", true)) return e;
    GSD_REQUIRE((int64_t)dst[i].H * dst[i].w_stride < (1LL << 31), GSD_ERR_UNSUPPORTED, "gsd_conv3x3_w2d: plane too large");
    csum += dst[i].C;
  }
  GSD_REQUIRE(csum == Cout, GSD_ERR_BAD_ARG, "gsd_conv3x3_w2d: destination segments hold %d channels, Cout=%d", csum, Cout);

  W2DPlan pl;
  GSD_REQUIRE(plan_w2d(N, H, W, Cout, &pl), GSD_ERR_UNSUPPORTED, "gsd_conv3x3_w2d: no tile shape");
  W2DParams P;
  P.src0 = to_srcd(src[0]);
  P.src1 = nsrc > 1 ? to_srcd(src[1]) : null_srcd();
  P.dst0 = to_dstd(dst[0]);
  P.dst1 = ndst > 1 ? to_dstd(dst[1]) : null_dstd();
  P.wt = wt;
  P.partials = partials;
  P.bw_raw = bw_raw; P.bw_scale = bw_scale; P.bw_shift = bw_shift; P.bw_mean = bw_mean; P.bw_invstd = bw_invstd;
  P.Cin = Cin; P.Cout = Cout;
  P.Mpad = round_up(Cout, 64);
  P.nchunks = Cin / 4;
  P.mblocks = pl.mblocks;
  P.N = N; P.H = H; P.W = W;
  P.TH = pl.TH; P.TW = pl.TW; P.TWq = pl.TWq; P.tiles_y = pl.tiles_y; P.tiles_x = pl.tiles_x;
  P.WR = pl.WR; P.WC = pl.WC; P.WCp = pl.WCp; P.PS = pl.PS;
  P.NPV = ceil_div(P.WR * P.WCp, 64);
  GSD_REQUIRE(P.NPV <= 4 * pl.nwp, GSD_ERR_UNSUPPORTED, "gsd_conv3x3_w2d: halo window too large");
  // block order: as many m-blocks per pass as have their weight images (24 KiB per chunk) in 3 MiB of the XCD's 4-MiB L2, over
  // groups of as many pixel tiles as make 64 blocks (what an XCD of the four-wave form holds).  GSD_W2D_MGROUP: 0 all m-blocks of a
  // pixel tile together (the order of the shallow levels), n that many; GSD_W2D_PGROUP the pixel tiles per group.  (tuning / A-B runs)
  {
    int g = (gsd_env_int("GSD_W2D_L2KB", 3072) << 10) / (P.nchunks * W2D_WTILE * 4);
    const int forced = gsd_env_int("GSD_W2D_MGROUP", -1);
    if (forced == 0) g = P.mblocks;
    if (forced > 0) g = forced;
    P.mgrp = std::min(std::max(g, 1), P.mblocks);
    P.pgrp = std::max(gsd_env_int("GSD_W2D_PGROUP", 64 / P.mgrp), 1);
    P.ptiles = N * pl.tiles_y * pl.tiles_x;
  }
  const long base = (long)N * pl.tiles_y * pl.tiles_x * P.mblocks;
  // K slabs: only with a workspace (the engine lends one in train mode), only in the four-wave form, and never more than fit
  int S = (ws != nullptr && pl.nwp == 2) ? w2d_pick_slabs(base, P.nchunks, bw_raw != nullptr) : 1;
  while (S > 1 && (int64_t)base * S * (W2D_BM * 256) > ws_elems) --S;
  if (S > 1 && P.nchunks / S < 2) S = 1;
  P.nslab = S;
  P.slabs = S > 1 ? ws : nullptr;
  if (S > 1) GSD_REQUIRE(((uintptr_t)ws & 15) == 0, GSD_ERR_BAD_ARG, "gsd_conv3x3_w2d: the K-slab workspace must be 16-byte aligned");
  const long grid = base * S;
  GSD_REQUIRE(grid < 2147483647L, GSD_ERR_UNSUPPORTED, "gsd_conv3x3_w2d: grid too large");
  bool plain = true;
  for (int i = 0; i < nsrc; ++i) plain = plain && src[i].scale == nullptr && src[i].relu == 0;
  // 16-byte halo pieces: one plain source whose rows start 16-byte aligned (pitch, plane and image strides multiples of 4 floats);
  // its pad columns must hold zeros -- the engine's row-pitched d_raw buffer does (GSD_W2D_X4=0: dword gathers, A/B runs)
  P.NP = pl.TW / 4 + 2;
  P.NI = ceil_div(P.WR * P.NP, 64);
  const bool x4 = plain && nsrc == 1 && pl.nwp == 2 && 4 * P.NI <= 8 && gsd_env_int("GSD_W2D_X4", 1) != 0 &&
                  ((uintptr_t)src[0].ptr & 15) == 0 && src[0].w_stride % 4 == 0 && src[0].c_stride % 4 == 0 && src[0].n_stride % 4 == 0 &&
                  src[0].off_h == 0 && src[0].off_w == 0 && src[0].w_stride >= round_up(src[0].W, 4);
  // unaligned 16-byte pieces for every other source: each segment vouches for 4 readable floats around its tensor (slack), lane
  // offsets stay 32-bit.  GSD_W2D_U4=0 keeps the dword gathers.  Default 1 since the loop's other vector work was halved (packed
  // transforms, constant image offsets): forward layer set 21.9 -> 21.2 ms, step -0.4 ms, bit-identical (when first built, against
  // the scalar transforms, it measured neutral: 97.7-97.9 ms either way)
  bool u4 = !x4 && pl.nwp == 2 && 4 * P.NI <= 8 && gsd_env_int("GSD_W2D_U4", 1) != 0;
  for (int i = 0; i < nsrc && u4; ++i) u4 = src[i].slack >= 4;
  if (x4 || u4) {
    P.WCp = 4 * P.NP;
    P.PS = w2d_x4_plane_stride(pl.TWq, P.WCp, P.WR);
  }
  const size_t lds = (size_t)(2 * (W2D_WTILE + 4 * P.PS) + 2 * 4 * P.nchunks + 4 * W2D_BM) * sizeof(float);
  if (gsd_env_set("GSD_W2D_TRACE"))
    fprintf(stderr, "w2d M%d K%d %dx%d N%d nsrc %d ndst %d plain %d x4 %d u4 %d | ptr&15 %d ws %d cs%%4 %d ns%%4 %d NI %d tile %dx%d slabs %d\n", Cout, Cin, H, W, N,
            nsrc, ndst, (int)plain, (int)x4, (int)u4, (int)((uintptr_t)src[0].ptr & 15), src[0].w_stride, (int)(src[0].c_stride % 4),
            (int)(src[0].n_stride % 4), P.NI, pl.TH, pl.TW, S);
  if (S > 1) {
    if (x4) return launch_w2d<true, 2, 1, true>(P, (int)grid, lds, (hipStream_t)stream);
    if (u4) return plain ? launch_w2d<true, 2, 2, true>(P, (int)grid, lds, (hipStream_t)stream) : launch_w2d<false, 2, 2, true>(P, (int)grid, lds, (hipStream_t)stream);
    return plain ? launch_w2d<true, 2, 0, true>(P, (int)grid, lds, (hipStream_t)stream) : launch_w2d<false, 2, 0, true>(P, (int)grid, lds, (hipStream_t)stream);
  }
  if (x4) return launch_w2d<true, 2, 1>(P, (int)grid, lds, (hipStream_t)stream);
  if (u4) return plain ? launch_w2d<true, 2, 2>(P, (int)grid, lds, (hipStream_t)stream) : launch_w2d<false, 2, 2>(P, (int)grid, lds, (hipStream_t)stream);
  if (pl.nwp == 2)
    return plain ? launch_w2d<true, 2>(P, (int)grid, lds, (hipStream_t)stream) : launch_w2d<false, 2>(P, (int)grid, lds, (hipStream_t)stream);
  return plain ? launch_w2d<true, 4>(P, (int)grid, lds, (hipStream_t)stream) : launch_w2d<false, 4>(P, (int)grid, lds, (hipStream_t)stream);
}

extern "C" int gsd_conv3x3_w2d(const gsd_src* src, int nsrc, const float* wt, int Cin, int Cout, const gsd_dst* dst, int ndst,
                               float* partials, int N, int H, int W, void* stream) {
  return w2d_impl(src, nsrc, wt, Cin, Cout, dst, ndst, partials, nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, stream);
}

extern "C" int gsd_conv3x3_w2d_dgrad_bnrelu(const gsd_src* src, const float* wt, int Cin, int Cout, const gsd_dst* dst,
                                            const float* raw, const float* scale, const float* shift, const float* mean,
                                            const float* invstd, float* partials, int N, int H, int W, void* stream) {
  GSD_REQUIRE(dst && raw && scale && shift && mean && invstd && partials, GSD_ERR_BAD_ARG,
              "gsd_conv3x3_w2d_dgrad_bnrelu: null argument");
  GSD_REQUIRE(dst->C == Cout && dst->H == H && dst->W == W && dst->off_h == 0 && dst->off_w == 0, GSD_ERR_BAD_ARG,
              "gsd_conv3x3_w2d_dgrad_bnrelu: dst must be the full (Cout,H,W) gradient buffer (raw shares its strides)");
  return w2d_impl(src, 1, wt, Cin, Cout, dst, 1, partials, raw, scale, shift, mean, invstd, N, H, W, stream);
}

// The same two with K-slab scratch lent by the caller (gsd_conv3x3_w2d_workspace floats; any capacity is safe: the launcher shrinks
// the slab count to what fits, 0 or a null pointer runs unsplit): what a train-mode schedule calls.  The sum over the input
// channels is then taken slab by slab in a fixed order -- run-to-run bitwise, not bit-equal to the unsplit launch.
extern "C" int gsd_conv3x3_w2d_ws(const gsd_src* src, int nsrc, const float* wt, int Cin, int Cout, const gsd_dst* dst, int ndst,
                                  float* partials, float* workspace, int64_t workspace_elems, int N, int H, int W, void* stream) {
  return w2d_impl(src, nsrc, wt, Cin, Cout, dst, ndst, partials, nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, stream, workspace,
                  workspace_elems);
}

extern "C" int gsd_conv3x3_w2d_dgrad_bnrelu_ws(const gsd_src* src, const float* wt, int Cin, int Cout, const gsd_dst* dst,
                                               const float* raw, const float* scale, const float* shift, const float* mean,
                                               const float* invstd, float* partials, float* workspace, int64_t workspace_elems, int N,
                                               int H, int W, void* stream) {
  GSD_REQUIRE(dst && raw && scale && shift && mean && invstd && partials, GSD_ERR_BAD_ARG,
              "gsd_conv3x3_w2d_dgrad_bnrelu: null argument");
  GSD_REQUIRE(dst->C == Cout && dst->H == H && dst->W == W && dst->off_h == 0 && dst->off_w == 0, GSD_ERR_BAD_ARG,
              "gsd_conv3x3_w2d_dgrad_bnrelu: dst must be the full (Cout,H,W) gradient buffer (raw shares its strides)");
  return w2d_impl(src, 1, wt, Cin, Cout, dst, 1, partials, raw, scale, shift, mean, invstd, N, H, W, stream, workspace, workspace_elems);
}
