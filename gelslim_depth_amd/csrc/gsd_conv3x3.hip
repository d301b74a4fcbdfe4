// gsd_conv3x3.hip -- conv3x3 (pad 1, stride 1, no bias) forward and dX as implicit GEMM on
// v_mfma_f32_16x16x4_f32, LDS-DMA form (gfx950).
//
//   D[m = out channel][n = pixel] = sum_{ci,tap} Wt[(ci,tap)][m] * B[(ci,tap)][pixel]
//
// Replaces aten::convolution at /root/reference/gelslim_depth/models/unet.py:11,14 (forward) and, with the
// dgrad weight layout (taps flipped, channels swapped), the dX half of aten::convolution_backward.
//
// Data movement: both operand tiles go HBM/L2 -> LDS with global_load_lds (weights 16 B/lane from a
// pre-tiled, pre-padded layout; the zero-padded (TH+2)x(TW+2) input halo 4 B/lane along NCHW rows), double
// buffered, one barrier per K-chunk of 4 input channels (36 k-rows, 9 MFMA k-steps): the DMA of chunk c+1 is
// in flight while chunk c is multiplied, no VGPR staging, no ds_write.  Deferred BatchNorm+ReLU of the
// producer is applied after the ds_read (b = max(fma(raw, scale[ci], shift[ci]), lo)), i.e. inside the MFMA
// loop where VALU work is free; zero padding / channel padding / F.pad offsets of the second (concat)
// segment are DMA'd from a sentinel (quiet NaN under a ReLU: max(NaN, 0) = 0; else 0).
//
// Pixels sit on the MFMA column (lane&15): one accumulator register = 16 consecutive floats of an NCHW row.
// Wave tile 64 (m) x 64 (pixels) = 4x4 MFMA tiles; block = 4 waves: 64x256 for M <= 64, 128x128 otherwise.
#include "gsd_common.h"

#include <cstdlib>


__device__ const float gsd_pad_c3[2] = {0.f, __builtin_nanf("")};

struct Conv3Params {
  SrcD src0, src1;
  DstD dst0, dst1;
  const float* wt;   // [mblocks][nchunks*36][BM], columns permuted so a lane's 4 m-tiles are one float4
  float* partials;
  // fused backward of relu(bn(raw)) on the destination (dgrad only): dst receives dz = d * [raw*scale+shift > 0] and
  // the partial sums become (sum dz, sum dz*xhat); all null for a plain convolution
  const float* bw_raw;
  const float* bw_scale;
  const float* bw_shift;
  const float* bw_mean;
  const float* bw_invstd;
  int Cin, Cout, Mpad, nchunks, mblocks;
  int N, H, W;
  int TH, TW, tiles_y, tiles_x, WR, WC, PS, NPV;
};

template <int WM, int WN, int NT>
__global__ __launch_bounds__(256, NT == 4 ? 3 : 2) void conv3x3_dma_kernel(const Conv3Params P) {
  constexpr int MT = 4;
  constexpr int BM = WM * 64;
  constexpr int WS = BM;             // unpadded rows: conflict-free for the ds_read_b128 A fetch (9*WS == 0 mod 64 banks)
  constexpr int WTILE = 36 * WS;     // floats per chunk
  constexpr int W4 = WTILE / 4;      // float4s per chunk
  constexpr int NWI = (W4 + 255) / 256;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int PS = P.PS;
  const int BUF = WTILE + 4 * PS;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int j = lane >> 4, l16 = lane & 15;

  const int mb = blockIdx.x % P.mblocks;
  const int pt = blockIdx.x / P.mblocks;
  const int m0 = mb * BM;
  const int tpi = P.tiles_y * P.tiles_x;
  const int n = pt / tpi;
  const int rt = pt - n * tpi;
  const int ty = rt / P.tiles_x;
  const int h0 = ty * P.TH, w0 = (rt - ty * P.tiles_x) * P.TW;

  // ---- per-lane pixel bookkeeping for the MFMA B operand / epilogue ------------------------------
  int baddr[NT], opix[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int q = (wn * NT + t) * 16 + l16;
    const bool ok = q < P.TH * P.TW;
    const int r = ok ? q / P.TW : 0;
    const int c = ok ? q - r * P.TW : 0;
    baddr[t] = WTILE + j * PS + r * P.WC + c;
    opix[t] = (ok && (h0 + r) < P.H && (w0 + c) < P.W) ? ((r << 16) | c) : -1;
  }

  // ---- DMA lane geometry: this wave owns window position chunks p = wave and wave+4 ------------------
  // offset of the position inside a channel plane of segment 0 / 1; -1: zero padding; -2: beyond the window
  int xo0[2], xo1[2];
#pragma unroll
  for (int pp = 0; pp < 2; ++pp) {
    const int pos = (wave + 4 * pp) * 64 + lane;
    xo0[pp] = xo1[pp] = -2;
    if (pos < P.WR * P.WC) {
      const int rr = pos / P.WC;
      const int gh = h0 - 1 + rr, gw = w0 - 1 + (pos - rr * P.WC);
      int hs = gh - P.src0.oh, ws = gw - P.src0.ow;
      xo0[pp] = ((unsigned)hs < (unsigned)P.src0.H && (unsigned)ws < (unsigned)P.src0.W) ? hs * P.src0.W + ws : -1;
      hs = gh - P.src1.oh;
      ws = gw - P.src1.ow;
      xo1[pp] = ((unsigned)hs < (unsigned)P.src1.H && (unsigned)ws < (unsigned)P.src1.W) ? hs * P.src1.W + ws : -1;
    }
  }
  const float* wsrc0 = P.wt + (size_t)mb * P.nchunks * WTILE;
  const bool p_on[2] = {wave < P.NPV, wave + 4 < P.NPV};

  // DMA-side running state over the channel sequence (wave-uniform except d_xo): which segment the next
  // channel comes from, its plane pointer, the padding sentinel, how many channels the segment has left.
  int d_seg = 0, d_left = P.src0.C;
  const float* d_base = P.src0.p + (long long)n * P.src0.ns;
  long long d_cs = P.src0.cs;
  const float* d_sent = P.src0.relu ? &gsd_pad_c3[1] : &gsd_pad_c3[0];
  int d_xo[2] = {xo0[0], xo0[1]};

  // One DMA "slot" = one wave instruction group: slots 0..NWI-1 move the weight chunk (16 B per lane), slot NWI+ch
  // moves input channel ch of the chunk (this wave's <= 2 position chunks).  The 9 k-steps of a chunk issue one
  // slot each for the NEXT chunk, after their MFMAs, so DMA issue is spread between the MFMA bursts instead of
  // sitting in front of the first one (sandbox: profiles/ubench/conv_loop2.hip, +3 %).
  static_assert(NWI + 4 <= 10, "two slots behind each of the first five k-steps");
  auto dma_slot = [&](int slot, int chunk, int buf) {
    float* Wb = smem + buf * BUF;
    if (slot < NWI) {
      const float* wsrc = wsrc0 + (size_t)chunk * WTILE + tid * 4;
      if (tid + slot * 256 < W4)
        __builtin_amdgcn_global_load_lds(wsrc + slot * 1024, Wb + (slot * 256 + wave * 64) * 4, 16, 0, 0);
    } else if (slot < NWI + 4) {
      const int ch = slot - NWI;
      float* Xb = Wb + WTILE;
      if (d_left == 0 && d_seg == 0) {  // first segment exhausted: continue in the concatenated second one
        d_seg = 1;
        d_left = P.src1.C;
        d_base = P.src1.p + (long long)n * P.src1.ns;
        d_cs = P.src1.cs;
        d_sent = P.src1.relu ? &gsd_pad_c3[1] : &gsd_pad_c3[0];
        d_xo[0] = xo1[0];
        d_xo[1] = xo1[1];
      }
      const bool c_ok = d_left > 0;
      const float* sentinel = c_ok ? d_sent : &gsd_pad_c3[0];
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        if (p_on[pp] && d_xo[pp] != -2) {
          const float* g = (c_ok && d_xo[pp] >= 0) ? d_base + d_xo[pp] : sentinel;
          __builtin_amdgcn_global_load_lds(g, Xb + ch * PS + (wave + 4 * pp) * 64, 4, 0, 0);
        }
      }
      if (c_ok) {
        d_base += d_cs;
        --d_left;
      }
    }
  };

  // Per-channel deferred-BN coefficients of the whole K range, once, into LDS (behind the two tile images):
  // sAff[c] = scale, sAff[Kpad + c] = shift.  A lane then fetches its k-row's pair per chunk with two ds_reads
  // instead of two dependent global loads.
  const int Kpad = P.nchunks * 4;
  float* sAff = smem + 2 * BUF;
  for (int c = tid; c < Kpad; c += 256) {
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
  // fused BatchNorm-backward epilogue: the block's output-channel coefficients, also through LDS (read back with
  // ds_read in the epilogue, so that no global load sits between its stores)
  float* sBw = sAff + 2 * Kpad;   // [4][BM]: scale, shift, mean, invstd
  if (P.bw_raw != nullptr) {
    for (int c = tid; c < BM; c += 256) {
      const int co = m0 + c < P.Cout ? m0 + c : 0;
      sBw[c] = P.bw_scale[co];
      sBw[BM + c] = P.bw_shift[co];
      sBw[2 * BM + c] = P.bw_mean[co];
      sBw[3 * BM + c] = P.bw_invstd[co];
    }
  }
  const float lo0 = P.src0.relu ? 0.f : -__builtin_inff(), lo1 = P.src1.relu ? 0.f : -__builtin_inff();

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int a_lane = wm * 64 + l16 * 4;
#pragma unroll
  for (int slot = 0; slot < NWI + 4; ++slot) dma_slot(slot, 0, 0);
  for (int chunk = 0; chunk < P.nchunks; ++chunk) {
    const int cur = chunk & 1;
    gsd_dma_barrier();  // this chunk's DMA has landed (vmcnt(0) + barrier); everyone has left the other buffer
    const int kc = chunk * 4 + j;   // this lane's k row (input channel) in this chunk
    const float sc = sAff[kc], sh = sAff[Kpad + kc];
    const float lo = kc < P.src0.C ? lo0 : (kc < P.Cin ? lo1 : -__builtin_inff());
    const bool more = chunk + 1 < P.nchunks;
    const float* Wc = smem + cur * BUF;
    // Software pipeline over the 9 k-steps of the chunk: the operands of k-step s+1 are read while k-step s multiplies,
    // and the order is PINNED (sched_barrier): left alone, hipcc sinks every ds_read to just before its first use and puts
    // s_waitcnt lgkmcnt(0) in front of each k-step's MFMAs, i.e. every wave eats the LDS latency once per 16 MFMAs.
    f32x4 av[2];
    float br[2][NT];
    av[0] = *reinterpret_cast<const f32x4*>(&Wc[(j * 9) * WS + a_lane]);   // out channels m*16+l16, m = 0..3
#pragma unroll
    for (int t = 0; t < NT; ++t) br[0][t] = Wc[baddr[t]];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      const int cs = s & 1, ns = cs ^ 1;
      const int koff_n = ((s + 1) / 3) * P.WC + ((s + 1) % 3);
      float b[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) b[t] = fmaxf(fmaf(br[cs][t], sc, sh), lo);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = mfma16(av[cs][m], b[t], acc[m][t]);
        if (s + 1 < 9) {
          if (m == 0) av[ns] = *reinterpret_cast<const f32x4*>(&Wc[(j * 9 + s + 1) * WS + a_lane]);
#pragma unroll
          for (int t = 0; t < NT; ++t)
            if (m >= 1 && (t * 3) / NT == m - 1) br[ns][t] = Wc[baddr[t] + koff_n];
        }
        if (m == MT - 1 && more && s < 5) {   // two DMA slots behind each of the first five k-steps: the last four cover the flight
          dma_slot(2 * s, chunk + 1, cur ^ 1);
          dma_slot(2 * s + 1, chunk + 1, cur ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---- epilogue: NCHW store (two destination segments with crop) + BatchNorm partial sums ---------
  int ooff0[NT], ooff1[NT];  // element offset of the lane's pixel inside a plane of dst0 / dst1, -1: cropped
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    ooff0[t] = ooff1[t] = -1;
    if (opix[t] >= 0) {
      const int h = h0 + (opix[t] >> 16), w = w0 + (opix[t] & 0xffff);
      int hd = h - P.dst0.oh, wd = w - P.dst0.ow;
      if ((unsigned)hd < (unsigned)P.dst0.H && (unsigned)wd < (unsigned)P.dst0.W) ooff0[t] = hd * P.dst0.ws + wd;
      hd = h - P.dst1.oh;
      wd = w - P.dst1.ow;
      if ((unsigned)hd < (unsigned)P.dst1.H && (unsigned)wd < (unsigned)P.dst1.W) ooff1[t] = hd * P.dst1.ws + wd;
    }
  }
  float* const d0 = P.dst0.p + (long long)n * P.dst0.ns;
  float* const d1 = P.dst1.p + (long long)n * P.dst1.ns;
  if (P.bw_raw == nullptr) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int co = m0 + wm * 64 + m * 16 + j * 4 + reg;
        const bool first = co < P.dst0.C;
        const int cd = first ? co : co - P.dst0.C;
        const bool co_ok = co < P.Cout && (first || cd < P.dst1.C);
        float* const plane = first ? d0 + (long long)cd * P.dst0.cs : d1 + (long long)cd * P.dst1.cs;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (opix[t] >= 0) {
            const float v = acc[m][t][reg];
            s1 += v;
            s2 = fmaf(v, v, s2);
            const int off = first ? ooff0[t] : ooff1[t];
            if (co_ok && off >= 0) plane[off] = v;
          }
        }
        if (P.partials != nullptr) {
          s1 = reduce16_to_lane15(s1);
          s2 = reduce16_to_lane15(s2);
          if (l16 == 15 && co < P.Mpad) {
            float* row = P.partials + (size_t)(pt * WN + wn) * (2 * P.Mpad);
            row[co] = s1;
            row[P.Mpad + co] = s2;
          }
        }
      }
    }
  } else {
    // dst0 is the gradient buffer of a conv+BN+ReLU unit whose raw output has the same geometry: dz = relu'(bn(raw)) * dX.
    // Loads and stores share vmcnt on gfx950: a load between two stores makes the second wait for the first.  So the
    // coefficients come from LDS, and the raw values are loaded 16 at a time (one m-tile) in front of their 16 stores.
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float xr[4][NT];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int co = m0 + wm * 64 + m * 16 + j * 4 + reg;
        const float* const rplane = P.bw_raw + (long long)n * P.dst0.ns + (long long)(co < P.Cout ? co : 0) * P.dst0.cs;
#pragma unroll
        for (int t = 0; t < NT; ++t) xr[reg][t] = rplane[ooff0[t] >= 0 ? ooff0[t] : 0];
      }
      __builtin_amdgcn_sched_barrier(0);   // keep this m-tile's 16 loads together, in front of its 16 stores
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int co = m0 + wm * 64 + m * 16 + j * 4 + reg;
        float* const plane = d0 + (long long)co * P.dst0.cs;
        const int cl = wm * 64 + m * 16 + j * 4 + reg;
        const float bsc = sBw[cl], bsh = sBw[BM + cl], bmu = sBw[2 * BM + cl], bis = sBw[3 * BM + cl];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (co < P.Cout && opix[t] >= 0 && ooff0[t] >= 0) {
            const float x = xr[reg][t];
            const float dz = fmaf(x, bsc, bsh) > 0.f ? acc[m][t][reg] : 0.f;
            plane[ooff0[t]] = dz;
            s1 += dz;
            s2 = fmaf(dz, (x - bmu) * bis, s2);
          }
        }
        if (P.partials != nullptr) {
          s1 = reduce16_to_lane15(s1);
          s2 = reduce16_to_lane15(s2);
          if (l16 == 15 && co < P.Mpad) {
            float* row = P.partials + (size_t)(pt * WN + wn) * (2 * P.Mpad);
            row[co] = s1;
            row[P.Mpad + co] = s2;
          }
        }
      }
    }
  }
}

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
namespace {

// Pick a TH x TW output tile with TH*TW <= BN and (TH+2)*(TW+2) <= 512.  First the minimum number of
// tiles (= wasted MFMA columns) over all shapes; then, among shapes within 5 % of that minimum, the one
// that is best for memory: rows that are multiples of 16 pixels (one accumulator register = one 64-B row
// segment), then the widest rows (long coalesced halo rows, few row wraps per wave).
void choose_tile(int H, int W, int BN, int* TH, int* TW) {
  long min_tiles = -1;
  for (int pass = 0; pass < 2; ++pass) {
    long best_score = -1;
    for (int tw = 1; tw <= W + 15 && tw <= BN; ++tw) {
      int th = BN / tw;
      if (th > H) th = H;
      while (th > 1 && (th + 2) * (tw + 2) > 512) --th;
      if ((th + 2) * (tw + 2) > 512) continue;
      const int ty = ceil_div(H, th);
      th = ceil_div(H, ty);  // smallest th giving the same number of row tiles
      const long tiles = (long)ty * ceil_div(W, tw);
      if (pass == 0) {
        if (min_tiles < 0 || tiles < min_tiles) min_tiles = tiles;
      } else if (tiles * 100 <= min_tiles * 105) {
        const long score = (tw % 16 == 0 ? 1000000L : 0L) + (long)(tw > W ? W : tw) * 1000 - tiles;
        if (score > best_score) {
          best_score = score;
          *TH = th;
          *TW = tw;
        }
      }
    }
  }
}

int plane_stride_16mod32(int n) {  // smallest PS >= n with PS % 32 == 16
  int ps = (n / 32) * 32 + 16;
  if (ps < n) ps += 32;
  return ps;
}

struct ConvPlan {
  bool wide;  // true: 64x256 block tile (M<=64), false: 128x128 or 128x256
  bool big;   // 128x256 block tile (wave tile 64 x 128, 2 blocks per CU): tuning option, measured no faster than 128x128
  int BM, BN, WN, TH, TW, tiles_y, tiles_x, mblocks;
};

ConvPlan plan_conv3x3(int H, int W, int M) {
  const int big_min = gsd_env_int("GSD_CONV_BIG", 0);   // tuning: min H*W for 128x256
  ConvPlan p;
  p.wide = M <= 64;
  p.big = !p.wide && big_min > 0 && H * W >= big_min;
  p.BM = p.wide ? 64 : 128;
  p.BN = (p.wide || p.big) ? 256 : 128;
  p.WN = p.wide ? 4 : 2;
  choose_tile(H, W, p.BN, &p.TH, &p.TW);
  p.tiles_y = ceil_div(H, p.TH);
  p.tiles_x = ceil_div(W, p.TW);
  p.mblocks = ceil_div(M, p.BM);
  return p;
}

template <int WM, int WN, int NT>
int launch(const Conv3Params& P, int grid, size_t lds, hipStream_t st) {
  static gsd_attr_once big_lds;   // per-device cache of an idempotent launch attribute (gsd_common.h)
  if (hipError_t e = gsd_allow_big_lds(big_lds, reinterpret_cast<const void*>(&conv3x3_dma_kernel<WM, WN, NT>)); e != hipSuccess) {
    gsd_set_error("gsd_conv3x3: hipFuncSetAttribute: %s", hipGetErrorString(e));
    return GSD_ERR_HIP;
  }
  hipLaunchKernelGGL((conv3x3_dma_kernel<WM, WN, NT>), dim3(grid), dim3(256), lds, st, P);
  GSD_LAUNCH_CHECK("gsd_conv3x3");
  return GSD_OK;
}

}  // namespace

extern "C" int gsd_conv3x3_partial_rows(int N, int H, int W, int Cout) {
  if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
  ConvPlan p = plan_conv3x3(H, W, Cout);
  return N * p.tiles_y * p.tiles_x * p.WN;
}

// Which conv3x3 kernel serves this shape: 0 = direct taps (this file), 1 = Winograd F(4,3) along rows (gsd_conv3x3_w43.hip).
// Both count MFMA instructions per launch including tile padding; the Winograd form wins when it needs clearly fewer
// (ideal: half; measured x1.3-1.7 on the U-Net's layers, profiles/bench_conv_forms.py).  GSD_CONV_ALGO=0|1 forces one.
extern "C" int64_t gsd_conv3x3_w43_mfma_count(int N, int H, int W, int Cin, int Cout);
extern "C" int gsd_conv3x3_algo(int N, int H, int W, int Cin, int Cout) {
  const char* env = getenv("GSD_CONV_ALGO");   // read per call: the tests switch forms inside one process
  const int forced = env ? atoi(env) : -1;
  if (forced == 0 || forced == 1) return forced;
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  if (Cin < 16) return 0;   // a K loop of 1-3 chunks is all prologue + epilogue, and the Winograd epilogue is the longer one
  const ConvPlan p = plan_conv3x3(H, W, Cout);
  const int64_t direct = (int64_t)N * p.tiles_y * p.tiles_x * p.mblocks * ceil_div(Cin, 4) * (p.big ? 1152 : 576);
  const int64_t wino = gsd_conv3x3_w43_mfma_count(N, H, W, Cin, Cout);
  return wino > 0 && wino * 10 <= direct * 8 ? 1 : 0;
}

static int conv3x3_impl(const gsd_src* src, int nsrc, const float* wt, int Cin, int Cout, const gsd_dst* dst, int ndst,
                        float* partials, const float* bw_raw, const float* bw_scale, const float* bw_shift,
                        const float* bw_mean, const float* bw_invstd, int N, int H, int W, void* stream) {
  GSD_REQUIRE(src && dst && wt, GSD_ERR_BAD_ARG, "gsd_conv3x3: null argument");
  GSD_REQUIRE(nsrc >= 1 && nsrc <= 2 && ndst >= 1 && ndst <= 2, GSD_ERR_BAD_ARG, "gsd_conv3x3: nsrc/ndst must be 1 or 2");
  GSD_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, GSD_ERR_BAD_ARG, "gsd_conv3x3: bad sizes");
  GSD_REQUIRE(H < 32768 && W < 32768, GSD_ERR_UNSUPPORTED, "gsd_conv3x3: H, W must be < 32768");
  GSD_REQUIRE(((uintptr_t)wt & 15) == 0, GSD_ERR_BAD_ARG, "gsd_conv3x3: weight layout must be 16-byte aligned");
  int csum = 0;
  for (int i = 0; i < nsrc; ++i) {
    if (int e = gsd_check_src(src[i], "gsd_conv3x3 src")) return e;
    GSD_REQUIRE(src[i].scale == nullptr || src[i].relu != 0, GSD_ERR_UNSUPPORTED,
                "gsd_conv3x3: an affine source segment must also have relu (zero padding uses a NaN sentinel)");
    GSD_REQUIRE((int64_t)src[i].H * src[i].W < (1LL << 31), GSD_ERR_UNSUPPORTED, "gsd_conv3x3: plane too large");
    csum += src[i].C;
  }
  GSD_REQUIRE(csum == Cin, GSD_ERR_BAD_ARG, "gsd_conv3x3: source segments hold %d channels, Cin=%d", csum, Cin);
  csum = 0;
  for (int i = 0; i < ndst; ++i) {
    if (int e = gsd_check_dst(dst[i], "gsd_conv3x3 dst", true)) return e;   // the epilogue addresses rows through w_stride
    csum += dst[i].C;
  }
  GSD_REQUIRE(csum == Cout, GSD_ERR_BAD_ARG, "gsd_conv3x3: destination segments hold %d channels, Cout=%d", csum, Cout);

  ConvPlan pl = plan_conv3x3(H, W, Cout);
  Conv3Params P;
  P.src0 = to_srcd(src[0]);
  P.src1 = nsrc > 1 ? to_srcd(src[1]) : null_srcd();
  P.dst0 = to_dstd(dst[0]);
  P.dst1 = ndst > 1 ? to_dstd(dst[1]) : null_dstd();
  P.wt = wt;
  P.partials = partials;
  P.bw_raw = bw_raw; P.bw_scale = bw_scale; P.bw_shift = bw_shift; P.bw_mean = bw_mean; P.bw_invstd = bw_invstd;
  P.Cin = Cin;
  P.Cout = Cout;
  P.Mpad = round_up(Cout, 64);
  P.nchunks = ceil_div(Cin, 4);
  P.mblocks = pl.mblocks;
  P.N = N; P.H = H; P.W = W;
  P.TH = pl.TH; P.TW = pl.TW; P.tiles_y = pl.tiles_y; P.tiles_x = pl.tiles_x;
  P.WR = pl.TH + 2; P.WC = pl.TW + 2;
  P.PS = plane_stride_16mod32(P.WR * P.WC);
  P.NPV = ceil_div(P.WR * P.WC, 64);
  GSD_REQUIRE(P.NPV <= 8, GSD_ERR_UNSUPPORTED, "gsd_conv3x3: halo window too large");
  const long grid = (long)N * pl.tiles_y * pl.tiles_x * pl.mblocks;
  GSD_REQUIRE(grid < 2147483647L, GSD_ERR_UNSUPPORTED, "gsd_conv3x3: grid too large");
  size_t lds = (size_t)(2 * (36 * pl.BM + 4 * P.PS) + 2 * 4 * P.nchunks + 4 * pl.BM) * sizeof(float);   // 2 tile images + BN coefficients (input side, output side)
  const int lds_min = gsd_env_int("GSD_CONV_LDS_MIN", 0);   // tuning: cap blocks/CU
  if ((size_t)lds_min > lds) lds = lds_min;
  if (pl.wide) return launch<1, 4, 4>(P, (int)grid, lds, (hipStream_t)stream);
  if (pl.big) return launch<2, 2, 8>(P, (int)grid, lds, (hipStream_t)stream);
  return launch<2, 2, 4>(P, (int)grid, lds, (hipStream_t)stream);
}

extern "C" int gsd_conv3x3(const gsd_src* src, int nsrc, const float* wt, int Cin, int Cout, const gsd_dst* dst,
                           int ndst, float* partials, int N, int H, int W, void* stream) {
  return conv3x3_impl(src, nsrc, wt, Cin, Cout, dst, ndst, partials, nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W,
                      stream);
}

extern "C" int gsd_conv3x3_dgrad_bnrelu(const gsd_src* src, const float* wt, int Cin, int Cout, const gsd_dst* dst,
                                        const float* raw, const float* scale, const float* shift, const float* mean,
                                        const float* invstd, float* partials, int N, int H, int W, void* stream) {
  GSD_REQUIRE(dst && raw && scale && shift && mean && invstd && partials, GSD_ERR_BAD_ARG,
              "gsd_conv3x3_dgrad_bnrelu: null argument");
  GSD_REQUIRE(dst->C == Cout && dst->H == H && dst->W == W && dst->off_h == 0 && dst->off_w == 0, GSD_ERR_BAD_ARG,
              "gsd_conv3x3_dgrad_bnrelu: dst must be the full (Cout,H,W) gradient buffer (raw shares its strides)");
  return conv3x3_impl(src, 1, wt, Cin, Cout, dst, 1, partials, raw, scale, shift, mean, invstd, N, H, W, stream);
}
