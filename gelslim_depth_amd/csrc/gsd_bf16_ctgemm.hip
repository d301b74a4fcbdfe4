// gsd_bf16_ctgemm.hip -- the two GEMM-shaped halves of ConvTranspose2d(k=2, s=2) (unet.py:36,41) on a large-tile kernel.
//
//   forward   out[n, 2h+kh+oy, 2w+kw+ox, co] = bias[co] + sum_ci x[n,h,w,ci] * Wt[(kh,kw,co)][ci]      (K = Cin,   M = 4 Cout)
//   dX        dx[n,h,w,ci] = sum_{kh,kw,co} g[n, 2h+kh+oy, 2w+kw+ox, co] * Wd[(kh,kw)][ci][co]          (K = 4 Cout, M = Cin)
//             (+ the fused pass 1 of the BatchNorm+ReLU backward of the unit below, as in gsd_bf16_conv.hip)
//
// Both are plain GEMMs over the INPUT-resolution pixels with no spatial reuse: every operand byte is used once per tile, so
// what the DMA-filled conv kernel's 128 x 256 tile with one wave per SIMD leaves exposed is everything (fills per FLOP, the
// epilogue of a short K loop, the barrier bubbles): 0.2-0.36 PFLOP/s.  Here: block = 8 waves (two per SIMD, 256 registers
// each), tile 256 m x 256 pixels (or 128 x 256 with 64-pixel wave tiles when M = 128), 64 k per barrier, LDS rows of 128 B
// (64 k) with the 16-byte piece p of row r at slot p ^ (r & 6): conflict-free ds_read_b128 for the MFMA operand pattern
// (16 consecutive rows x one piece), filled by global_load_lds_dwordx4 in pieces of 8 whole rows; the pixel tile is a run
// of 256 consecutive pixels of the flattened (n,h,w) grid (no 2-D tile padding; the scatter / gather decodes (n,h,w) with
// one reciprocal multiply per item).  Persistent blocks, XCD-aware order, next item's first fill under the epilogue.
// Same accumulation order as gconv_bf16_kernel<1,..> (taps, then 32-channel chunks): outputs are bit-identical to it.
#include "gsd_bf16_common.h"

#include <type_traits>

namespace {

struct CtP {
  const u16* a;            // pixel operand: x (forward) or the gradient slice g (dX)
  int a_pitch, Ha, Wa;     // its pixel pitch and buffer extent
  const u16* wt;           // [taps][Mpad][Kt]
  int Kt, Mpad, taps;      // k per tap (a multiple of 64)
  u16* out;
  long long out_pitch;
  int Hob, Wob;
  int N, H, W, P;          // pixel grid of the GEMM, P = N*H*W
  float rW, rHW;           // 1/W, 1/(H*W)
  int M, mblocks, ntile, xcd;
  int Cs, oy, ox;          // forward: scatter (m = q*Cs + co); dX: the taps read (2h + (q>>1) + oy, 2w + (q&1) + ox)
  const float* bias;
  float* partials;
  const u16* bw_y;
  long long bw_pitch;
  unsigned out_bytes, y_bytes;   // dX: extent of the buffer descriptors of out and bw_y (the last valid byte + 1)
  const float* bw_scale; const float* bw_shift; const float* bw_mean; const float* bw_invstd;
};

// p / d for 0 <= p < 2^24 (exact float), rd = 1/d
__device__ __forceinline__ int fdiv(int p, int d, float rd) {
  int q = (int)((float)p * rd);
  const int r = p - q * d;
  if (r < 0) --q;
  else if (r >= d) ++q;
  return q;
}

template <int DX, int WM, int WN, int NT>
__global__ __launch_bounds__(512) void ctgemm_bf16_kernel(const CtP P) {
  static_assert(WM * WN == 8, "eight waves");
  constexpr int BM = WM * 64, NPX = WN * NT * 16, ROWS = BM + NPX;
  constexpr int MT = 4;
  constexpr int STAGE = ROWS * 128;              // bytes of one 64-k stage
  constexpr int NWI = BM / 64, NXI = NPX / 64;   // DMA instructions per wave and stage (8 waves x 8 rows per instruction)
  constexpr int NS = NWI + NXI;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sBw = reinterpret_cast<float*>(smem + 2 * STAGE);   // [4][BM] coefficients of the fused BatchNorm backward
  float* sSt = sBw + 4 * BM;                                 // [8 waves][2][64] statistics of the block

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int g = lane >> 4, j = lane & 15;

  // persistent block: one m-block, pixel tiles pt, pt + pt_step, ... < pt_end (XCD-aware order as in gsd_bf16_conv.hip)
  int mb, pt, pt_step, pt_end, prow;
  if (P.xcd) {
    const int x = blockIdx.x & 7, l = blockIdx.x >> 3, nq = (int)(gridDim.x >> 3) / P.mblocks;
    mb = l % P.mblocks;
    pt = (int)((long long)P.ntile * x / 8) + l / P.mblocks;
    pt_step = nq;
    pt_end = (int)((long long)P.ntile * (x + 1) / 8);
    prow = x * nq + l / P.mblocks;
  } else {
    mb = blockIdx.x % P.mblocks;
    pt = blockIdx.x / P.mblocks;
    pt_step = gridDim.x / P.mblocks;
    pt_end = P.ntile;
    prow = blockIdx.x / P.mblocks;
  }
  const int m0 = mb * BM;
  const bool fused = DX && P.bw_y != nullptr;
  if (fused) {
    for (int c = tid; c < BM; c += 512) {
      sBw[c] = P.bw_scale[m0 + c];
      sBw[BM + c] = P.bw_shift[m0 + c];
      const float is = P.bw_invstd[m0 + c];
      sBw[2 * BM + c] = -P.bw_mean[m0 + c] * is;   // xhat = fma(y, invstd, -mean * invstd)
      sBw[3 * BM + c] = is;
    }
    for (int c = tid; c < 8 * 128; c += 512) sSt[c] = 0.f;
  }

  // ---- DMA bookkeeping: instruction i of this wave fills LDS rows i*64 + r0 (+ BM for the pixels), r0 = wave*8 + lane/8, with
  // source piece (lane & 7) ^ (r0 & 6) of that row (the XOR swizzle lives in the source address)
  const int r0 = wave * 8 + (lane >> 3);
  const int piece = (lane & 7) ^ (r0 & 6);
  // weight rows: LDS row (64-block, m-tile mm, tile row ii) holds channel (mm>>1)*32 + (ii>>2)*8 + (mm&1)*4 + (ii&3), so that a lane's
  // accumulators are two runs of 8 consecutive channels (16-byte stores); r0 < 64 carries (mm, ii)
  const int srow = (((r0 >> 5) & 1) << 5) | (((r0 & 15) >> 2) << 3) | (((r0 >> 4) & 1) << 2) | (r0 & 3);
  const int woff0 = (m0 + srow) * P.Kt + piece * 8;   // + i * 64 * Kt
  int xoff[NXI];
  auto prep = [&](int p0) {
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const int p = min(p0 + i * 64 + r0, P.P - 1);   // rows beyond the grid repeat the last pixel; their results are not stored
      if (DX) {
        const int n = fdiv(p, P.H * P.W, P.rHW), rem = p - n * P.H * P.W;
        const int h = fdiv(rem, P.W, P.rW), w = rem - h * P.W;
        xoff[i] = ((n * P.Ha + 2 * h + P.oy) * P.Wa + 2 * w + P.ox) * P.a_pitch + piece * 8;
      } else {
        xoff[i] = p * P.a_pitch + piece * 8;
      }
    }
  };
  const int S = P.taps * (P.Kt >> 6);   // stages per item
  // one DMA instruction of the fill of the stage at (tap f_tap, channel f_c0) -- wave-uniform -- into buffer `buf`
  auto dma_slot = [&](int slot, const u16* wsrc, const u16* asrc, int buf) {
    unsigned char* base = smem + buf * STAGE;
    if (slot < NWI) {
      __builtin_amdgcn_global_load_lds((const void*)(wsrc + woff0 + slot * 64 * P.Kt), base + (slot * 8 + wave) * 1024, 16, 0, 0);
    } else {
      const int i = slot - NWI;
      __builtin_amdgcn_global_load_lds((const void*)(asrc + xoff[i]), base + BM * 128 + (i * 8 + wave) * 1024, 16, 0, 0);
    }
  };
  auto w_src = [&](int tap, int c0) { return P.wt + (size_t)tap * P.Mpad * P.Kt + c0; };
  auto a_src = [&](int tap, int c0) { return P.a + c0 + (DX ? ((tap >> 1) * P.Wa + (tap & 1)) * P.a_pitch : 0); };

  // ---- operand read offsets: row (.. + j), piece ks*4 + g at slot (ks*4 + g) ^ (j & 6); the second k-step flips bit 2 of the slot
  const int sl0 = (g ^ (j & 6)) << 4;
  const int aoff[2] = {(wm * 64 + j) * 128 + sl0, (wm * 64 + j) * 128 + (sl0 ^ 64)};                       // + m * 2048
  const int boff[2] = {(BM + wn * NT * 16 + j) * 128 + sl0, (BM + wn * NT * 16 + j) * 128 + (sl0 ^ 64)};   // + t * 2048

  if (pt >= pt_end) {   // (an XCD's range can be shorter than its blocks): this block's partial rows are zeros
    if (fused && lane < 64) {
      float* row = P.partials + (size_t)(prow * WN + wn) * (2 * P.Mpad);
      row[m0 + wm * 64 + lane] = row[P.Mpad + m0 + wm * 64 + lane] = 0.f;
    }
    return;
  }

  // epilogue constants: a lane's 16 channels are two runs of 8: ch0 .. and ch0 + 32 ..
  const int ch0 = m0 + wm * 64 + g * 8;
  long long ooff[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    int co = ch0 + 32 * k, q = 0;
    if (!DX) {
      q = co / P.Cs;
      co -= q * P.Cs;
    }
    ooff[k] = DX ? (long long)co : ((long long)((q >> 1) + P.oy) * P.Wob + (q & 1) + P.ox) * P.out_pitch + co;
  }
  typedef unsigned u32x4s __attribute__((ext_vector_type(4), aligned(8)));
  const int cob = DX ? 0 : ch0 % P.Cs;   // forward: channel of the lane's first accumulator inside its quadrant

  int p0 = pt * NPX, gs = 0;
  prep(p0);
#pragma unroll
  for (int sl = 0; sl < NS; ++sl) dma_slot(sl, w_src(0, 0), a_src(0, 0), 0);

  while (true) {
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      f32x4 init = f32x4{0.f, 0.f, 0.f, 0.f};
      if (!DX && P.bias != nullptr) {   // Cs % 64 == 0: the wave's 64 channels share their quadrant
        const int co = cob + (m >> 1) * 32 + (m & 1) * 4;
        init = f32x4{P.bias[co], P.bias[co + 1], P.bias[co + 2], P.bias[co + 3]};
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[m][t] = init;
    }
    const int next = pt + pt_step;
    int f_tap = 0, f_c0 = 0;   // (tap, first channel) of the stage being FILLED
    for (int s = 0; s < S; ++s, ++gs) {
      gsd_dma_barrier();   // this stage has landed (vmcnt(0)) and every wave has left the other buffer
      f_c0 += 64;
      if (f_c0 == P.Kt) {
        f_c0 = 0;
        ++f_tap;
      }
      if (s + 1 == S) {    // the fill belongs to the next item (after the last item it repeats a fill nobody reads: no branch)
        f_tap = f_c0 = 0;
        if (next < pt_end) prep(next * NPX);
      }
      const u16* wsrc = w_src(f_tap, f_c0);
      const u16* asrc = a_src(f_tap, f_c0);
      const unsigned char* Sc = smem + (gs & 1) * STAGE;
      const int fbuf = (gs + 1) & 1;
      u32x4 a[2][MT], b[NT];
#pragma unroll
      for (int m = 0; m < MT; ++m) a[0][m] = *reinterpret_cast<const u32x4*>(Sc + aoff[0] + m * 2048);
#pragma unroll
      for (int t = 0; t < NT; ++t) b[t] = *reinterpret_cast<const u32x4*>(Sc + boff[0] + t * 2048);
      __builtin_amdgcn_sched_barrier(0);
      constexpr int STEPS = MT * NT / 2;                  // micro-steps of two MFMAs per k-step
      constexpr int SPK = (NS + 1) / 2;                   // DMA slots per k-step
      constexpr int SGAP = STEPS / SPK > 0 ? STEPS / SPK : 1;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
          const int m = i % MT, t = 2 * (i / MT);
          acc[m][t] = mfma_bf16(a[ks][m], b[t], acc[m][t]);
          acc[m][t + 1] = mfma_bf16(a[ks][m], b[t + 1], acc[m][t + 1]);
          if (ks == 0) {   // operands of the second k-step: the A tiles first, then the B pairs as they die
            if (i < MT) a[1][i] = *reinterpret_cast<const u32x4*>(Sc + aoff[1] + i * 2048);
            if (i >= MT && (i % MT) < 2) {
              const int bt = 2 * (i / MT - 1) + (i % MT);
              b[bt] = *reinterpret_cast<const u32x4*>(Sc + boff[1] + bt * 2048);
            }
          } else if (i < 2) {
            b[NT - 2 + i] = *reinterpret_cast<const u32x4*>(Sc + boff[1] + (NT - 2 + i) * 2048);   // the last pair
          }
          if (i % SGAP == SGAP - 1 && i / SGAP < SPK && ks * SPK + i / SGAP < NS) dma_slot(ks * SPK + i / SGAP, wsrc, asrc, fbuf);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }

    // ---- epilogue ---------------------------------------------------------------------------------------------------
    const int pw = p0 + wn * NT * 16 + j;   // this lane's pixel of n-tile 0; n-tile t: + 16 t
    const bool interior = p0 + NPX <= P.P;
    // Lane constants of the epilogue are re-derived here, per item, from a value the compiler cannot see through: hoisted to the
    // kernel's entry they are spilled around the K loop, and a scratch reload in the epilogue waits (vmcnt) for the next item's
    // fill that was issued just before it.
    int ch0v = ch0;
    asm volatile("" : "+v"(ch0v));
    if (DX) {
      // Buffer accesses: one 32-bit byte offset per access from a uniform descriptor (64-bit per-lane addresses for the tile's
      // pixels do not fit beside 128 accumulators), and the descriptor's range check replaces the guarded variant of the
      // epilogue: beyond the grid's last pixel (the last tile only) loads return zeros and stores are dropped.
      const unsigned opitch = (unsigned)P.out_pitch * 2u, ypitch = (unsigned)P.bw_pitch * 2u;
      const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)P.out, 0, P.out_bytes, 0x00020000);
      if (fused) {
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)P.bw_y, 0, P.y_bytes, 0x00020000);
        // Fused pass 1 of BatchNorm+ReLU backward, one run of 8 channels (m-tiles 2k, 2k+1) at a time: its coefficients and its
        // 16 sums stay in registers over the tile's pixels, the raw outputs are loaded half a run at a time IN FRONT of that
        // half's stores (loads and stores share vmcnt), and the run's sums leave for the wave's LDS cells by DPP.
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          f32x4 sc[2], sh[2];
          float s1[2][4], s2[2][4];
          const float* cf = sBw + (ch0v - m0) + k * 32;   // this run's coefficients: + mm * 4; shift + BM, -mean*invstd + 2 BM, invstd + 3 BM
#pragma unroll
          for (int mm = 0; mm < 2; ++mm) {
            sc[mm] = *reinterpret_cast<const f32x4*>(cf + mm * 4);
            sh[mm] = *reinterpret_cast<const f32x4*>(cf + BM + mm * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) s1[mm][e] = s2[mm][e] = 0.f;
          }
          const unsigned chb = (unsigned)(ch0v + 32 * k) * 2u;
          constexpr int QB = NT >= 8 ? 2 : NT / 2;   // pixels per batch of loads
#pragma unroll
          for (int t0 = 0; t0 < NT; t0 += QB) {
            u32x4 yr[QB];
#pragma unroll
            for (int tt = 0; tt < QB; ++tt) yr[tt] = __builtin_amdgcn_raw_buffer_load_b128(ry, (unsigned)(pw + (t0 + tt) * 16) * ypitch + chb, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tt = 0; tt < QB; ++tt) {
              const int t = t0 + tt, p = pw + t * 16;
              const bool pix_ok = interior || p < P.P;   // a pixel beyond the grid adds nothing to the sums (its store is dropped)
              unsigned pk[4];
#pragma unroll
              for (int mm = 0; mm < 2; ++mm) {
                const int m = 2 * k + mm;
                const f32x4 nm = *reinterpret_cast<const f32x4*>(cf + 2 * BM + mm * 4), is = *reinterpret_cast<const f32x4*>(cf + 3 * BM + mm * 4);
                const unsigned y01 = yr[tt][mm * 2], y23 = yr[tt][mm * 2 + 1];
                float yv[4] = {__uint_as_float(y01 << 16), __uint_as_float(y01 & 0xffff0000u), __uint_as_float(y23 << 16),
                               __uint_as_float(y23 & 0xffff0000u)};
                float dz[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) dz[e] = (fmaf(yv[e], sc[mm][e], sh[mm][e]) > 0.f && pix_ok) ? acc[m][t][e] : 0.f;
                const unsigned lo = pack_bf16(dz[0], dz[1]), hi = pack_bf16(dz[2], dz[3]);
                pk[2 * mm] = lo;
                pk[2 * mm + 1] = hi;
                const float q[4] = {__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u), __uint_as_float(hi << 16),
                                    __uint_as_float(hi & 0xffff0000u)};   // sums of the values as stored
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  s1[mm][e] += q[e];
                  s2[mm][e] = fmaf(q[e], fmaf(yv[e], is[e], nm[e]), s2[mm][e]);
                }
              }
              __builtin_amdgcn_raw_buffer_store_b128(u32x4{pk[0], pk[1], pk[2], pk[3]}, ro, (unsigned)p * opitch + chb, 0, 0);
              __builtin_amdgcn_sched_barrier(0);   // pixel by pixel: interleaved, the temporaries of several pixels spill
            }
          }
          float* cell = sSt + wave * 128 + k * 32 + (ch0v - m0 - wm * 64);   // each cell belongs to one lane (j == 15 of its row): no atomics
#pragma unroll
          for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float a1 = reduce16_to_lane15(s1[mm][e]), a2 = reduce16_to_lane15(s2[mm][e]);
              if (j == 15) {
                cell[mm * 4 + e] += a1;
                cell[64 + mm * 4 + e] += a2;
              }
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const unsigned o = (unsigned)(pw + t * 16) * opitch + (unsigned)ch0v * 2u;
          unsigned pk[2 * MT];
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            pk[2 * m] = pack_bf16(acc[m][t][0], acc[m][t][1]);
            pk[2 * m + 1] = pack_bf16(acc[m][t][2], acc[m][t][3]);
          }
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{pk[0], pk[1], pk[2], pk[3]}, ro, o, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{pk[4], pk[5], pk[6], pk[7]}, ro, o + 64u, 0, 0);
        }
      }
    } else {
      // scatter: (n, h, w) of n-tile 0 by one reciprocal multiply, the other tiles by carrying 16 pixels on (W >= 16)
      int n = fdiv(pw, P.H * P.W, P.rHW);
      int rem = pw - n * P.H * P.W;
      int h = fdiv(rem, P.W, P.rW), w = rem - h * P.W;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const bool pix_ok = interior || pw + t * 16 < P.P;
        u16* ot = P.out + ((long long)(n * P.Hob + 2 * h) * P.Wob + 2 * w) * P.out_pitch;
        unsigned pk[2 * MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          pk[2 * m] = pack_bf16(acc[m][t][0], acc[m][t][1]);
          pk[2 * m + 1] = pack_bf16(acc[m][t][2], acc[m][t][3]);
        }
        if (pix_ok) {
          *reinterpret_cast<u32x4s*>(ot + ooff[0]) = u32x4{pk[0], pk[1], pk[2], pk[3]};
          *reinterpret_cast<u32x4s*>(ot + ooff[1]) = u32x4{pk[4], pk[5], pk[6], pk[7]};
        }
        w += 16;
        if (w >= P.W) {
          w -= P.W;
          if (++h == P.H) {
            h = 0;
            ++n;
          }
        }
      }
    }
    if (next >= pt_end) break;
    pt = next;
    p0 = pt * NPX;
  }
  gsd_dma_barrier();   // the last (unread) fill must have landed before the block gives its LDS back
  if (fused) {
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's cells are written (the row below is read by the same wave)
    float* row = P.partials + (size_t)(prow * WN + wn) * (2 * P.Mpad);
    const int mrow = m0 + wm * 64 + lane;
    row[mrow] = sSt[wave * 128 + lane];
    row[P.Mpad + mrow] = sSt[wave * 128 + 64 + lane];
  }
}

int cu_count_ct() {
  static int n = 0;   // benign race: every thread computes the same value
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    n = v;
  }
  return n;
}

struct CtPlan {
  bool ok;
  int BM, NPX, WN, mblocks, ntile, grid;
};

// the tile of a (pixels, M) problem, or ok = false: the shape stays on gconv_bf16_kernel<1,..>
CtPlan ct_plan(long long Ppix, int M, bool dx) {
  CtPlan p;
  p.ok = false;
  if (Ppix <= 0 || Ppix >= (1 << 24)) return p;
  const int force = gsd_env_int("GSD_BF16_CT_BM", 0);   // tuning: 128 / 256
  if (M % 256 == 0 && force != 128 && !(dx && force != 256)) { p.BM = 256; p.WN = 2; }
  else if (M % 128 == 0 && dx) { p.BM = 128; p.WN = 4; }
  else return p;
  p.NPX = 256;
  p.mblocks = M / p.BM;
  p.ntile = (int)((Ppix + p.NPX - 1) / p.NPX);
  long grid = cu_count_ct() / p.mblocks * p.mblocks;
  if (grid < p.mblocks) return p;
  const long items = (long)p.ntile * p.mblocks;
  if (grid > items) grid = items;
  p.grid = (int)grid;
  p.ok = true;
  return p;
}

template <int DX, int WM, int WN, int NT>
int launch_ct(const CtP& P, int grid, hipStream_t st) {
  constexpr int BM = WM * 64, NPX = WN * NT * 16;
  const size_t lds = (size_t)2 * (BM + NPX) * 128 + (size_t)(4 * BM + 8 * 128) * sizeof(float);
  static gsd_attr_once big_lds;
  if (hipError_t e = gsd_allow_big_lds(big_lds, reinterpret_cast<const void*>(&ctgemm_bf16_kernel<DX, WM, WN, NT>)); e != hipSuccess) {
    gsd_set_error("gsd_bf16_conv_dense (large tile): hipFuncSetAttribute: %s", hipGetErrorString(e));
    return GSD_ERR_HIP;
  }
  GSD_REQUIRE(lds <= 160 * 1024, GSD_ERR_UNSUPPORTED, "gsd_bf16_conv_dense (large tile): LDS %zu B too large", lds);
  hipLaunchKernelGGL((ctgemm_bf16_kernel<DX, WM, WN, NT>), dim3((unsigned)grid), dim3(512), lds, st, P);
  GSD_LAUNCH_CHECK("gsd_bf16_conv_dense (large tile)");
  return GSD_OK;
}

}  // namespace

// Which shapes the large-tile kernel takes (everything else stays on gconv_bf16_kernel<1,..>): forward = 1 tap at stride 1 with
// the scatter epilogue, dX = 4 taps at stride 2 without it.  A function of the SHAPE only, so that the number of BatchNorm
// partial rows a launch will write can be asked in advance (gsd_bf16_conv_dense_partial_rows).
bool gsd_ctgemm_shape(int N, int H, int W, int K, int M, int ntaps, int stride, int scatter_cs) {
  if (gsd_env_int("GSD_BF16_CTGEMM", 1) == 0) return false;
  if (K <= 0 || K % 64 != 0 || W < 16 || N <= 0 || H <= 0) return false;
  const bool fwd = ntaps == 1 && stride == 1 && scatter_cs > 0 && scatter_cs % 64 == 0, dx = ntaps == 4 && stride == 2 && scatter_cs == 0;
  return (fwd || dx) && ct_plan((long long)N * H * W, M, dx).ok;
}

int gsd_ctgemm_partial_rows(int N, int H, int W, int M) {
  const CtPlan pl = ct_plan((long long)N * H * W, M, true);
  return pl.ok ? pl.grid / pl.mblocks * pl.WN : 0;
}

// the operand conditions of a shape gsd_ctgemm_shape accepted: 32-bit element offsets, the four taps are the 2 x 2 block at (oy, ox)
// and stay inside the buffer (the kernel has no zero line)
bool gsd_ctgemm_operands(const gsd_nhwc* in, const gsd_nhwc* out, const gsd_bf16_bnbwd* bw, int ntaps, const int* ty, const int* tx, int H,
                         int W) {
  // 32-bit BYTE offsets per pixel, also for the (up to 512) pixel slots past the last item's end: (pixels + 512) * pitch * 2 < 2^32
  auto fits = [](const gsd_nhwc* t) { return ((long long)t->N * t->H * t->W + 512) * t->pitch < 2147483647LL; };
  if (!fits(in) || !fits(out)) return false;
  if (bw != nullptr && !fits(bw->y)) return false;
  if (ntaps == 1) return ty[0] == 0 && tx[0] == 0 && in->H == H && in->W == W;
  const int oy = ty[0], ox = tx[0];
  if (oy < 0 || ox < 0 || ty[1] != oy || tx[1] != ox + 1 || ty[2] != oy + 1 || tx[2] != ox || ty[3] != oy + 1 || tx[3] != ox + 1) return false;
  return 2 * (H - 1) + oy + 1 < in->H && 2 * (W - 1) + ox + 1 < in->W;
}

int gsd_ctgemm_launch(const gsd_nhwc* in, const void* wt, const gsd_nhwc* out, int K, int M, int ntaps, const int* ty, const int* tx, int H,
                      int W, int scatter_cs, int oy, int ox, const float* bias, float* partials, const gsd_bf16_bnbwd* bw, void* stream) {
  const CtPlan pl = ct_plan((long long)in->N * H * W, M, ntaps == 4);
  CtP P;
  P.a = (const u16*)in->ptr; P.a_pitch = (int)in->pitch; P.Ha = in->H; P.Wa = in->W;
  P.wt = (const u16*)wt; P.Kt = K; P.Mpad = gsd_bf16_conv_mpad(M); P.taps = ntaps;
  P.out = (u16*)out->ptr; P.out_pitch = out->pitch; P.Hob = out->H; P.Wob = out->W;
  P.N = in->N; P.H = H; P.W = W; P.P = in->N * H * W;
  P.rW = 1.0f / (float)W; P.rHW = 1.0f / (float)(H * W);
  P.M = M; P.mblocks = pl.mblocks; P.ntile = pl.ntile;
  P.xcd = (gsd_env_int("GSD_BF16_XCD", 1) != 0 && pl.grid % 8 == 0 && (pl.grid / 8) % pl.mblocks == 0) ? 1 : 0;
  P.Cs = scatter_cs; P.oy = ntaps == 4 ? ty[0] : oy; P.ox = ntaps == 4 ? tx[0] : ox;
  P.bias = bias; P.partials = partials;
  P.bw_y = nullptr; P.bw_pitch = 0; P.bw_scale = P.bw_shift = P.bw_mean = P.bw_invstd = nullptr;
  P.out_bytes = (unsigned)(((long long)(P.P - 1) * out->pitch + M) * 2);
  P.y_bytes = 0;
  if (bw != nullptr) {
    P.bw_y = (const u16*)bw->y->ptr; P.bw_pitch = bw->y->pitch;
    P.y_bytes = (unsigned)(((long long)(P.P - 1) * bw->y->pitch + M) * 2);
    P.bw_scale = bw->scale; P.bw_shift = bw->shift; P.bw_mean = bw->mean; P.bw_invstd = bw->invstd;
  }
  hipStream_t st = (hipStream_t)stream;
  if (ntaps == 1) return launch_ct<0, 4, 2, 8>(P, pl.grid, st);
  if (pl.BM == 256) return launch_ct<1, 4, 2, 8>(P, pl.grid, st);
  return launch_ct<1, 2, 4, 4>(P, pl.grid, st);
}
