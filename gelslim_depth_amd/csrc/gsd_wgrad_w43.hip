// gsd_wgrad_w43.hip -- dW of conv3x3 with the transposed Winograd F(4,3) identity along image rows (gfx950).
//
//   dW[co][ci][r][s] = sum_{n,h,w} dy[n,co,h,w] * a[n,ci,h+r-1,w+s-1]
//
// (the dW half of aten::convolution_backward for /root/reference/gelslim_depth/models/unet.py:11,14).  The forward
// identity y = A^T[(G g) .* (B^T d)] is linear in g, so for every tile of 4 horizontally adjacent outputs
//
//   dg = G^T [ (A dy) .* (B^T d) ]
//
// with the SAME input transform B^T d as the forward kernel (gsd_conv3x3_w43.hip), dy transformed by A (4 -> 6 values)
// and G^T applied once at the very end.  Per (co, ci, kernel row r) and tile that is 6 products instead of 12:
//
//   D_{f,r}[co][ci] = sum_tiles U_f[co][tile] * V_{f,r}[tile][ci]          (18 accumulators per (co,ci) instead of 9)
//   dW[co][ci][r][s] = sum_f G[f][s] * D_{f,r}[co][ci]                      (wgrad_w43_reduce_kernel, after the split sum)
//
// GEMM view: M = co, N = ci, K = tiles (4 per v_mfma_f32_16x16x4_f32).  Block = 4 or 8 waves (see the kernel); wave tile
// 32 co x 16 ci x 18 = 144 accumulator registers.  A stage is 16 tiles (64 pixels, TH rows x TW columns): dy rows
// [co][64 px, tile-major] and the activation halo windows [ci][(TH+2) x (TW+2)] reach LDS by global_load_lds (dword
// gathers, zero padding / out-of-segment positions from a sentinel: NaN under a ReLU, else 0), double buffered, two
// waves per SIMD.  The deferred BatchNorm+ReLU of the activation is applied after the ds_read (a lane's input channel is
// fixed), then B^T; A dy needs 9 VALU operations per 4 values.  Split-K over stages with ordered slab reduction as in
// gsd_wgrad.hip: bitwise reproducible.
#include "gsd_common.h"
#include <type_traits>

#include <cstdio>
#include <cstdlib>

__device__ const float gsd_pad_wg43[2] = {0.f, __builtin_nanf("")};
__device__ __attribute__((aligned(16))) const float gsd_zero16_wg43[4] = {0.f, 0.f, 0.f, 0.f};
__device__ __attribute__((aligned(16))) const float gsd_nan16_wg43[4] = {__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""),
                                                                         __builtin_nanf("")};

typedef float f32x2w __attribute__((ext_vector_type(2)));

// Diagnostic build only (-DGSD_WG43_STAMPS; never in the product library): s_memtime stamps around the segments of a stage.
#ifdef GSD_WG43_STAMPS
static unsigned long long* g_wg43_stamp_buf = nullptr;
extern "C" void gsd_wg43_set_stamp_buffer(void* p) { g_wg43_stamp_buf = (unsigned long long*)p; }
#define WG43_STAMP(i)                                                                            \
  {                                                                                              \
    unsigned long long t_;                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                   \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    st_acc[i] += t_ - st_prev;                                                                   \
    st_prev = t_;                                                                                \
  }
#else
#define WG43_STAMP(i) {}
#endif

struct WgW43Params {
  SrcD a0, a1;  // activation (B operand), up to two concatenated segments
  SrcD dy;      // gradient w.r.t. the raw conv output (plain)
  float* slabs; // [split][9 = r*3+s][M][Ncols]: G^T applied per split
  int M, Ncols;
  int N, H, W;
  int TH, TW, TWq, tiles_y, tiles_x, WR, WC, WCp, XS;
  int stages_total, splits, mblocks, nblocks;
  unsigned long long* stamps;   // diagnostic builds only
};

namespace {
constexpr int WG_DS = 68;      // dy row stride in LDS, dword gathers: 64 pixels + 4 (bank spread)
constexpr int WG_DS_X4 = 64;   // AX4: rows are contiguous 256-byte runs (one DMA instruction fills four), XOR-swizzled instead
}

// s_waitcnt vmcnt(n) for a run-time n (gfx9 encoding: vmcnt = bits[3:0] | bits[15:14] << 4; expcnt / lgkmcnt untouched)
__device__ __forceinline__ void wg43_wait_vmcnt(int n) {
#define WG43_W(k) case k: __builtin_amdgcn_s_waitcnt(0x0F70 | ((k) & 15) | (((k) >> 4) << 14)); break;
  switch (n) {
    WG43_W(1) WG43_W(2) WG43_W(3) WG43_W(4) WG43_W(5) WG43_W(6) WG43_W(7) WG43_W(8) WG43_W(9) WG43_W(10) WG43_W(11) WG43_W(12)
    WG43_W(13) WG43_W(14) WG43_W(15) WG43_W(16) WG43_W(17) WG43_W(18) WG43_W(19) WG43_W(20) WG43_W(21) WG43_W(22) WG43_W(23)
    WG43_W(24) WG43_W(25) WG43_W(26) WG43_W(27) WG43_W(28) WG43_W(29) WG43_W(30) WG43_W(31) WG43_W(32) WG43_W(33) WG43_W(34)
    WG43_W(35) WG43_W(36) WG43_W(37) WG43_W(38) WG43_W(39) WG43_W(40)
    default: __builtin_amdgcn_s_waitcnt(0x0F70); break;   // vmcnt(0)
  }
#undef WG43_W
}

// NWM x NWN waves of 32 co x 16 ci: (2,2) block 64 co x 32 ci, 4 waves, two blocks per CU; (4,2) 128 co x 32 ci and (2,4)
// 64 co x 64 ci, 8 waves, one block per CU: one operand's tile is then amortised over twice the MFMAs (24 instead of 32
// DMA instructions per wave and stage; measured 8 % faster than two 4-wave blocks).
//
// AX4: dy comes from a PITCHED buffer (rows 16-byte aligned: gsd_bn_bwd_apply's out-of-place form) and moves as aligned
// 16-byte pieces -- a piece is one Winograd tile (4 pixels), an instruction fills four 64-pixel rows: 32 instead of 128
// DMA instructions per stage for dy, i.e. 12 instead of 24 per wave (stamps, profiles/stamp_wgrad.py: the waves spend ~40 %
// of their time issuing the fills, ~250 cycles per instruction).  The LDS rows are then contiguous (no padding between
// them), so the 16 rows of a ds_read_b128 would collide on 4 banks: tile t of row r is stored at slot t ^ (r & 15)
// (swizzle on the SOURCE address of the DMA and on the read; cdna_hip_programming.md rule 21).
//
// R3 (8-wave blocks): a ring of THREE LDS images and a half-stage stagger between the two waves of a SIMD.  Stamps put a wave
// at ~30 % of its time issuing fills, ~55 % multiplying, ~15 % at the barrier -- and with two images both partners issue at
// about the same time, so the matrix pipe idles.  Now waves 0-3 issue their share of the fills for stage s+2 and THEN multiply
// stage s, waves 4-7 multiply stage s and THEN issue: one partner's ~4,600 cycles of fill issue lie beside the other's 144
// MFMAs (4,608 cycles).  The fills go two stages ahead, so the late half's land in time; the wait before the barrier is a
// counted vmcnt that leaves exactly this wave's youngest fills in flight.
//
// PLAIN: no activation segment carries a deferred BatchNorm / ReLU (the pooled sources of the encoder's first convs): the
// transform drops its 36 fma/max per k-step.  Measured with the transform forced plain: -3.8 % kernel time.  (Choosing per
// BLOCK inside one kernel -- the upsampled half of a concat is plain too -- doubles the stage code and sends the
// accumulators to scratch.)
//
// BX4: the activation windows move as 16-byte pieces too -- straight from the UNALIGNED rows (W = 427, 213, ...): a
// global_load_lds_dwordx4 takes any 4-byte aligned global address at full rate (profiles/ubench/dma_global_align.hip).  The
// block's BN window planes lie back to back in LDS (plane = WR rows x WCp/4 pieces + dummy pieces up to an ODD piece count, so
// the 16 channels of a ds_read_b128 spread over all 8 bank groups) and an instruction's 64 lanes are 64 consecutive pieces of
// that image, whichever channels and rows they fall into: 16 instead of 64 instructions per stage for 32 channels of a 6 x 20
// window.  Pieces wholly outside the image come from a 16-byte sentinel; a piece that STRADDLES the left or right image edge
// is loaded as it lies in memory (its outside part is the neighbouring row's data: the caller guarantees 4 readable floats
// before and after the tensor, gsd_src.slack) and the lane that moved it overwrites that part with the padding value once its
// own fills have landed (vmcnt(0)), in front of the stage's barrier.
//
// RR (the 4 x 16 stage, TWq == 4: k-step ks is tile row ks, lane group j tile column j): window row w feeds tile rows w-2, w-1, w
// -- as kernel rows 2, 1, 0 -- with the SAME six values for a lane, so its B^T relu(bn(.)) is computed once and kept across
// the k-steps (a ring of three rows): 6 row transforms and window reads per stage instead of 12; same products in the same
// order, bit-identical.
// RR == 2: the same for the 8 x 8 stage (TWq == 2: k-step ks holds tile rows 2ks and 2ks+1, lane groups j < 2 / j >= 2; a lane's
// window rows are 2ks + (j>>1) + {0,1,2}, the last of which is the first of its next k-step): 9 instead of 12 row transforms.
#ifndef WG43_LATE_KS   // the k-step after which the late half of an 8-wave block issues its fills
#define WG43_LATE_KS 0
#endif
template <int NWM, int NWN, bool AX4, bool R3, bool PLAIN, bool BX4, int RR>
__global__ __launch_bounds__(64 * NWM * NWN, NWM * NWN == 4 ? 2 : 1) void wgrad3x3_w43_kernel(const WgW43Params P) {
  static_assert(!R3 || NWM * NWN == 8, "the three-image ring is the 8-wave form");
  static_assert(!BX4 || (AX4 && !R3), "16-byte window pieces come with 16-byte dy pieces and two LDS images");
  static_assert(RR == 0 || BX4, "row reuse is instantiated for the 16-byte-piece form only");
  constexpr int BM = 32 * NWM, BN = 16 * NWN, NW = NWM * NWN, DS = AX4 ? WG_DS_X4 : WG_DS, MT = 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int XS = P.XS;
  // BX4: the window image is whole 64-piece instructions long (the last one's surplus lanes write dummies behind the planes)
  const int BUF = BX4 ? BM * DS + (((BN * (XS >> 2)) + 63) >> 6) * 256 : BM * DS + BN * XS;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef GSD_WG43_STAMPS
  unsigned long long st_acc[4] = {0, 0, 0, 0}, st_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif
  const int wm = wave / NWN, wn = wave % NWN;
  const int j = lane >> 4, l16 = lane & 15;

  const int lid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int per_split = P.mblocks * P.nblocks;
  const int split = lid / per_split;
  const int rem = lid - split * per_split;
  const int mb = rem % P.mblocks, nb = rem / P.mblocks;
  const int m0 = mb * BM, n0 = nb * BN;
  const int s_begin = (int)((long long)split * P.stages_total / P.splits);
  const int s_end = (int)((long long)(split + 1) * P.stages_total / P.splits);

  // ---- DMA lane geometry -------------------------------------------------------------------------------------------------
  // A: lane = pixel (tile t = lane>>2, element lane&3) of the stage in tile-major order
  // AX4: lane = (row lr = lane>>4 of the instruction's four rows, slot q = lane&15), slot q of row r holds tile q ^ (r & 15);
  // the wave's instructions are i' = wave + NW*k (rows 4*i' .. 4*i'+3), so (row & 15) = (4*wave + lr) & 15 for all of them
  const int a_lr = lane >> 4;
  const int a_t = AX4 ? ((lane & 15) ^ ((4 * wave + a_lr) & 15)) : (lane >> 2);
  const int a_r = a_t / P.TWq;
  const int a_c = (a_t - a_r * P.TWq) * 4 + (AX4 ? 0 : (lane & 3));
  // B: window positions p*64 + lane -> (row, column) of the padded window and the float offset from the window origin in
  // a plane of segment A / B (their widths may differ).  Every instruction moves all 64 lanes (no exec masks on the
  // issue path): a lane past the window re-reads the window origin into the slack behind it (XS >= npv*64).
  const int npv = (P.WR * P.WCp + 63) >> 6;
  int b_rr[4], b_cc[4], oA[4], oB[4];
  if constexpr (!BX4) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int pos = p * 64 + lane;
      b_rr[p] = pos / P.WCp;
      b_cc[p] = pos - b_rr[p] * P.WCp;
      const bool in_win = b_rr[p] < P.WR && b_cc[p] < P.WC;
      if (!in_win) b_rr[p] = b_cc[p] = 0;
      oA[p] = b_rr[p] * P.a0.ws + b_cc[p];
      oB[p] = b_rr[p] * P.a1.ws + b_cc[p];
    }
  }

  // deferred BatchNorm+ReLU of this lane's input channel
  float sc = 1.f, sh = 0.f, lo = -__builtin_inff();
  {
    const int c = n0 + wn * 16 + l16;
    const bool first = c < P.a0.C;
    const SrcD& S = first ? P.a0 : P.a1;
    const int cc = first ? c : c - P.a0.C;
    if (c < P.Ncols && cc < S.C) {
      if (S.scale != nullptr) {
        sc = S.scale[cc];
        sh = S.shift[cc];
      }
      if (S.relu) lo = 0.f;
    }
  }

  // The activation channels [n0, n0 + BN) of a block almost always lie in ONE source segment (always in the U-Net: the concat
  // boundary is a multiple of 64).  Then everything about the window fills that depends on the segment is a block constant:
  // copy the segment's descriptor into scalars once and keep one set of lane offsets.
  const bool b_seg1 = P.a1.C > 0 && n0 >= P.a0.C;
  const bool b_one = P.a1.C == 0 || b_seg1 || n0 + BN <= P.a0.C;
  const float* const S_p = b_seg1 ? P.a1.p : P.a0.p;
  const long long S_ns = b_seg1 ? P.a1.ns : P.a0.ns, S_cs = b_seg1 ? P.a1.cs : P.a0.cs;
  const int S_H = b_seg1 ? P.a1.H : P.a0.H, S_W = b_seg1 ? P.a1.W : P.a0.W, S_ws = b_seg1 ? P.a1.ws : P.a0.ws;
  const int S_oh = b_seg1 ? P.a1.oh : P.a0.oh, S_ow = b_seg1 ? P.a1.ow : P.a0.ow;
  const int S_c0 = n0 - (b_seg1 ? P.a0.C : 0) + wave;                         // this wave's first channel inside the segment
  const int S_left = (b_seg1 ? P.a1.C : P.a0.C) - S_c0;                        // channel i of the wave exists iff NW * i < S_left
  const float* const S_sent = (b_seg1 ? P.a1.relu : P.a0.relu) ? &gsd_pad_wg43[1] : &gsd_pad_wg43[0];
  int oS[4];
  if constexpr (!BX4) {
#pragma unroll
    for (int p = 0; p < 4; ++p) oS[p] = b_rr[p] * S_ws + b_cc[p];
  }
  // BX4: instruction i = wave + NW*k of the stage moves pieces 64 i .. 64 i + 63 of the block's [BN][XS/4] piece image; this
  // lane's piece of instruction k: float offset from the first channel's window origin, and (row, first column, channel)
  // packed for the edge stages (bit 31: a dummy piece or one past the image)
  constexpr int KB = 4;   // instructions per wave at most (the host checks)
  int x_off[KB], x_meta[KB];
  const int NPr = P.WCp >> 2, NPc = XS >> 2, NI = (BN * NPc + 63) >> 6;
  const int S_chan = (b_seg1 ? P.a1.C : P.a0.C) - (n0 - (b_seg1 ? P.a0.C : 0));   // channels of the segment from n0 on
  const float padv = (b_seg1 ? P.a1.relu : P.a0.relu) ? __builtin_nanf("") : 0.f;
  const float* const S_sent16 = (b_seg1 ? P.a1.relu : P.a0.relu) ? &gsd_nan16_wg43[0] : &gsd_zero16_wg43[0];
  if constexpr (BX4) {
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int i = wave + NW * k;
      const int pid = 64 * i + lane;
      const int ch = pid / NPc, pp = pid - ch * NPc;
      const int row = pp / NPr, pc = pp - row * NPr;
      const bool dummy = i >= NI || ch >= BN || row >= P.WR;
      x_off[k] = dummy ? 0 : (int)(ch * S_cs) + row * S_ws + 4 * pc;
      x_meta[k] = dummy ? (int)0x80000000u : (row | (4 * pc) << 8 | ch << 16);
    }
  }
  int fix0 = 0, fix1 = 0;   // per LDS image: the window columns [gl0, gl1) and [gr0, gr1) to overwrite with the padding value

  // Address arithmetic is kept out of the per-instruction path (the kernel is VALU-bound next to 144 MFMAs per stage): per
  // stage one scalar window origin per segment, per lane the constants above; a stage whose window lies inside the
  // segment ("interior", the common case) needs no per-lane validity at all, a border stage computes it once per segment.
  // coordinates of the next stage to issue, carried incrementally (two integer divisions per stage cost ~40 scalar
  // instructions, and with two waves per SIMD every instruction of a wave costs an issue slot)
  int st_n, st_ty, st_tx;
  {
    const int tpi = P.tiles_y * P.tiles_x;
    st_n = s_begin / tpi;
    const int rs = s_begin - st_n * tpi;
    st_ty = rs / P.tiles_x;
    st_tx = rs - st_ty * P.tiles_x;
  }
  auto issue_dma = [&](int stage, int buf) __attribute__((always_inline)) {
    const int n = st_n, h0 = st_ty * P.TH, w0 = st_tx * P.TW;
    if (++st_tx == P.tiles_x) {
      st_tx = 0;
      if (++st_ty == P.tiles_y) {
        st_ty = 0;
        ++st_n;
      }
    }
    float* Ab = smem + buf * BUF;
    float* Bb = Ab + BM * DS;
#ifdef WG43_ABL
    const bool skipA = ((WG43_ABL) & 1) && stage > s_begin + 1, skipB = ((WG43_ABL) & 2) && stage > s_begin + 1;
#else
    const bool skipA = false, skipB = false;
#endif
    // ---- A: dy rows ----
    if constexpr (AX4) {
      const bool inside = h0 + P.TH <= P.H && w0 + P.TW <= P.dy.ws && m0 + BM <= P.M;
      const bool pix_ok = (h0 + a_r) < P.H && (w0 + a_c) < P.W;   // the piece's first pixel exists; columns W.. of the pitch hold 0
      const float* rbase = P.dy.p + (long long)n * P.dy.ns + (long long)(m0 + 4 * wave + a_lr) * P.dy.cs +
                           ((long long)(h0 + a_r) * P.dy.ws + (w0 + a_c));
      const long long rstep = (long long)(4 * NW) * P.dy.cs;
      const int ninstr = skipA ? 0 : BM / 4 / NW;
      if (inside) {
#pragma unroll 4
        for (int i = 0; i < ninstr; ++i) {
          __builtin_amdgcn_global_load_lds(rbase, Ab + (wave + NW * i) * (4 * DS), 16, 0, 0);
          rbase += rstep;
        }
      } else {
#pragma unroll 1
        for (int i = 0; i < ninstr; ++i) {
          const float* g = (pix_ok && m0 + 4 * (wave + NW * i) + a_lr < P.M) ? rbase : &gsd_zero16_wg43[0];
          __builtin_amdgcn_global_load_lds(g, Ab + (wave + NW * i) * (4 * DS), 16, 0, 0);
          rbase += rstep;
        }
      }
    } else {
      const bool inside = h0 + P.TH <= P.H && w0 + P.TW <= P.W;
      const bool pix_ok = (h0 + a_r) < P.H && (w0 + a_c) < P.W;
      const int aoff = (h0 + a_r) * P.dy.ws + (w0 + a_c);
      const float* rbase = P.dy.p + (long long)n * P.dy.ns + (long long)(m0 + wave) * P.dy.cs;
      const long long rstep = (long long)NW * P.dy.cs;
      const int nrows = skipA ? 0 : BM / NW;
      if (inside && m0 + BM <= P.M) {
#pragma unroll 4
        for (int i = 0; i < nrows; ++i) {
          __builtin_amdgcn_global_load_lds(rbase + aoff, Ab + (wave + NW * i) * DS, 4, 0, 0);
          rbase += rstep;
        }
      } else {
#pragma unroll 4
        for (int i = 0; i < nrows; ++i) {
          const float* g = (pix_ok && m0 + wave + NW * i < P.M) ? rbase + aoff : &gsd_pad_wg43[0];
          __builtin_amdgcn_global_load_lds(g, Ab + (wave + NW * i) * DS, 4, 0, 0);
          rbase += rstep;
        }
      }
    }
    // ---- B: activation windows ----
    if constexpr (BX4) {
      const int hs = h0 - 1 - S_oh, ws = w0 - 1 - S_ow;
      const float* cb = S_p + (long long)n * S_ns + (long long)(n0 - (b_seg1 ? P.a0.C : 0)) * S_cs + ((long long)hs * S_ws + ws);
      const bool in = hs >= 0 && hs + P.WR <= S_H && ws >= 0 && ws + P.WCp <= S_W && S_chan >= BN;
      int fx = 0;
      if (in) {
#pragma unroll
        for (int k = 0; k < KB; ++k)
          if (wave + NW * k < NI) {
            const float* gp = cb + x_off[k];
            float* dstp = Bb + (wave + NW * k) * 256;
            if (!skipB) __builtin_amdgcn_global_load_lds(gp, dstp, 16, 0, 0);
          }
      } else {
#pragma unroll 1   // (edge stages: one piece at a time keeps the address temporaries of four out of the register budget)
        for (int k = 0; k < KB; ++k)
          if (wave + NW * k < NI) {
            const int m = x_meta[k];
            const int r = hs + (m & 255), c0 = ws + (m >> 8 & 255), ch = m >> 16 & 255;
            const bool ok = m >= 0 && ch < S_chan && (unsigned)r < (unsigned)S_H && c0 + 3 >= 0 && c0 < S_W;
            const float* gp = ok ? cb + x_off[k] : S_sent16;
            float* dstp = Bb + (wave + NW * k) * 256;
            if (!skipB) __builtin_amdgcn_global_load_lds(gp, dstp, 16, 0, 0);
          }
        // the outside part of a straddling piece: window columns [4 (cl / 4), cl) on the left, [cr, 4 ceil(cr / 4)) on the right
        const int cl = ws < 0 ? -ws : 0, cr = S_W - ws;
        if (cl & 3) fx |= (cl & ~3) | cl << 8;
        if (cr > 0 && cr < P.WCp && (cr & 3)) fx |= cr << 16 | ((cr + 3) & ~3) << 24;
      }
      if (buf) fix1 = fx; else fix0 = fx;
      return;
    }
    if (b_one && !skipB) {
      // one segment for the whole block: a scalar plane pointer that advances by NW channels + the lanes' fixed window offsets;
      // interior stages need nothing else, border stages one validity bit per window position
      const int hs = h0 - 1 - S_oh, ws = w0 - 1 - S_ow;
      const bool in = hs >= 0 && hs + P.WR <= S_H && ws >= 0 && ws + P.WC <= S_W;
      const float* cb = S_p + (long long)n * S_ns + (long long)S_c0 * S_cs + ((long long)hs * S_ws + ws);
      const long long cstep = (long long)NW * S_cs;
      float* Xd = Bb + wave * XS;
      if (in && NW * (BN / NW - 1) < S_left) {
#pragma unroll
        for (int i = 0; i < BN / NW; ++i) {
#pragma unroll
          for (int p = 0; p < 4; ++p)
            if (p < npv) {
              const float* gp = cb + oS[p];
              float* dstp = Xd + p * 64;
              __builtin_amdgcn_global_load_lds(gp, dstp, 4, 0, 0);
            }
          cb += cstep;
          Xd += NW * XS;
        }
      } else {
        int vm = 0;   // bit p = window position p*64+lane exists in the segment
#pragma unroll
        for (int p = 0; p < 4; ++p)
          if ((unsigned)(hs + b_rr[p]) < (unsigned)S_H && (unsigned)(ws + b_cc[p]) < (unsigned)S_W) vm |= 1 << p;
#pragma unroll
        for (int i = 0; i < BN / NW; ++i) {
          const bool c_ok = NW * i < S_left;
          const float* const sent = c_ok ? S_sent : &gsd_pad_wg43[0];
#pragma unroll
          for (int p = 0; p < 4; ++p)
            if (p < npv) {
              const float* gp = (c_ok && (vm >> p & 1)) ? cb + oS[p] : sent;
              float* dstp = Xd + p * 64;
              __builtin_amdgcn_global_load_lds(gp, dstp, 4, 0, 0);
            }
          cb += cstep;
          Xd += NW * XS;
        }
      }
      return;
    }
    // (a block whose channels straddle the two segments: the general form)
    // (named scalars, not arrays: an array indexed by the segment lands in scratch memory, with a vmcnt(0) per access)
    const int hsA = h0 - 1 - P.a0.oh, wsA = w0 - 1 - P.a0.ow, hsB = h0 - 1 - P.a1.oh, wsB = w0 - 1 - P.a1.ow;
    const int woA = hsA * P.a0.ws + wsA, woB = hsB * P.a1.ws + wsB;
    const bool inA = hsA >= 0 && hsA + P.WR <= P.a0.H && wsA >= 0 && wsA + P.WC <= P.a0.W;
    const bool inB = hsB >= 0 && hsB + P.WR <= P.a1.H && wsB >= 0 && wsB + P.WC <= P.a1.W;
    int vmA = 0, vmB = 0;   // border stages: bit p = position p*64+lane exists in the segment
    if (!inA) {
#pragma unroll
      for (int p = 0; p < 4; ++p)
        if ((unsigned)(hsA + b_rr[p]) < (unsigned)P.a0.H && (unsigned)(wsA + b_cc[p]) < (unsigned)P.a0.W) vmA |= 1 << p;
    }
    if (!inB && P.a1.C > 0) {
#pragma unroll
      for (int p = 0; p < 4; ++p)
        if ((unsigned)(hsB + b_rr[p]) < (unsigned)P.a1.H && (unsigned)(wsB + b_cc[p]) < (unsigned)P.a1.W) vmB |= 1 << p;
    }
    const int nch = skipB ? 0 : BN / NW;
#pragma unroll 2
    for (int i = 0; i < nch; ++i) {
      const int ch = wave + NW * i;
      const int c = n0 + ch;
      const bool first = c < P.a0.C;
      const SrcD& S = first ? P.a0 : P.a1;
      const int cc = first ? c : c - P.a0.C;
      const bool c_ok = c < P.Ncols && cc < S.C;
      float* Xd = Bb + ch * XS;
      const float* cbase = S.p + (long long)n * S.ns + (long long)(c_ok ? cc : 0) * S.cs + (first ? woA : woB);
      // per-position offsets of this channel's segment, selected by VALUE (a select between the two arrays themselves
      // would put them in scratch memory)
      int o[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) o[p] = first ? oA[p] : oB[p];
      if (c_ok && (first ? inA : inB)) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
          if (p < npv) __builtin_amdgcn_global_load_lds(cbase + o[p], Xd + p * 64, 4, 0, 0);
      } else {
        const float* sentinel = (c_ok && S.relu) ? &gsd_pad_wg43[1] : &gsd_pad_wg43[0];
        const int m = c_ok ? (first ? vmA : vmB) : 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          if (p < npv) {
            const unsigned long long pa = reinterpret_cast<unsigned long long>(cbase + o[p]);
            const unsigned long long ps = reinterpret_cast<unsigned long long>(sentinel);
            const bool ok = (m >> p & 1) != 0;
            const unsigned lo32 = ok ? (unsigned)pa : (unsigned)ps, hi32 = ok ? (unsigned)(pa >> 32) : (unsigned)(ps >> 32);
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(((unsigned long long)hi32 << 32) | lo32), Xd + p * 64, 4, 0, 0);
          }
        }
      }
    }
  };

  f32x4 acc[MT][18];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < 18; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // operand addresses of the 4 k-steps of a stage: k-step ks multiplies tiles 4*ks + j
  int a_off[4], b_off[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int t = 4 * ks + j;
    const int trow = t / P.TWq, tq = t - trow * P.TWq;
    a_off[ks] = (wm * 32 + l16) * DS + 4 * (AX4 ? (t ^ l16) : t);   // AX4: rows wm*32 + l16 (+16): (row & 15) == l16
    b_off[ks] = BM * DS + (wn * 16 + l16) * XS + trow * P.WCp + 4 * tq;
  }

  // `late`: issue the next stage's DMA after the first k-step instead of in front of the stage (see the stage loop)
  auto row_transform = [&](const f32x4& xa, const f32x2w& xb, float (&v)[6]) {   // B^T relu(bn(raw)) of one window row
    float d0 = xa[0], d1 = xa[1], d2 = xa[2], d3 = xa[3], d4 = xb[0], d5 = xb[1];
    if constexpr (!PLAIN) {
      d0 = fmaxf(fmaf(d0, sc, sh), lo), d1 = fmaxf(fmaf(d1, sc, sh), lo), d2 = fmaxf(fmaf(d2, sc, sh), lo);
      d3 = fmaxf(fmaf(d3, sc, sh), lo), d4 = fmaxf(fmaf(d4, sc, sh), lo), d5 = fmaxf(fmaf(d5, sc, sh), lo);
    }
    const float a = fmaf(-4.f, d2, d4), b = fmaf(-4.f, d1, d3);
    const float c = d4 - d2, e = 2.f * (d3 - d1);
    v[0] = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
    v[1] = a + b;
    v[2] = a - b;
    v[3] = c + e;
    v[4] = c - e;
    v[5] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
  };
  auto compute = [&](int cur, bool late, int next_stage, bool more) __attribute__((always_inline)) {
    const float* Sb = smem + cur * BUF;
    if constexpr (RR == 1) {
      // b_off[0] = this lane's channel plane + 4 j: window row w of tile column j is at + w * WCp
      f32x4 ya[2][MT];
      f32x4 xa;
      f32x2w xb;
      float V[3][6];   // V[w % 3]: window row w
#pragma unroll
      for (int m = 0; m < MT; ++m) ya[0][m] = *reinterpret_cast<const f32x4*>(&Sb[a_off[0] + m * 16 * DS]);
#pragma unroll
      for (int w = 0; w < 3; ++w) {
        xa = *reinterpret_cast<const f32x4*>(&Sb[b_off[0] + w * P.WCp]);
        xb = *reinterpret_cast<const f32x2w*>(&Sb[b_off[0] + w * P.WCp + 4]);
        row_transform(xa, xb, V[w]);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int cb = ks & 1, nb2 = cb ^ 1;
        if (ks + 1 < 4) {   // the next k-step's dy tile and the one new window row fly during this k-step's MFMAs
#pragma unroll
          for (int m = 0; m < MT; ++m) ya[nb2][m] = *reinterpret_cast<const f32x4*>(&Sb[a_off[ks + 1] + m * 16 * DS]);
          xa = *reinterpret_cast<const f32x4*>(&Sb[b_off[0] + (ks + 3) * P.WCp]);
          xb = *reinterpret_cast<const f32x2w*>(&Sb[b_off[0] + (ks + 3) * P.WCp + 4]);
        }
        float U[MT][6];
#pragma unroll
        for (int m = 0; m < MT; ++m) {   // U = A dy
          const float y0 = ya[cb][m][0], y1 = ya[cb][m][1], y2 = ya[cb][m][2], y3 = ya[cb][m][3];
          const float p = y0 + y2, q = y1 + y3;
          const float a = fmaf(4.f, y2, y0), b = 2.f * fmaf(4.f, y3, y1);
          U[m][0] = y0;
          U[m][1] = p + q;
          U[m][2] = p - q;
          U[m][3] = a + b;
          U[m][4] = a - b;
          U[m][5] = y3;
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int f = 0; f < 6; ++f) acc[m][r * 6 + f] = mfma16(U[m][f], V[(ks + r) % 3][f], acc[m][r * 6 + f]);
#ifdef WG43_LATE_HALF   // diagnostic: the late half issues after the first 18 MFMAs of the stage
          if (ks == 0 && m == 0 && late && more) issue_dma(next_stage, cur ^ 1);
#endif
        }
#if defined(WG43_ABL) && ((WG43_ABL) & 8)   // diagnostic: no row transform after the first three
        if (ks + 1 < 4) { V[ks % 3][0] = xa[0]; V[ks % 3][1] = xa[1]; V[ks % 3][2] = xa[2]; V[ks % 3][3] = xa[3]; V[ks % 3][4] = xb[0]; V[ks % 3][5] = xb[1]; }
#else
        if (ks + 1 < 4) row_transform(xa, xb, V[ks % 3]);   // window row ks + 3 takes the place of row ks
#endif
#ifndef WG43_LATE_HALF
        if (ks == WG43_LATE_KS && late && more) {
          WG43_STAMP(2)
          issue_dma(next_stage, cur ^ 1);
          WG43_STAMP(1)
        }
#endif
      }
      WG43_STAMP(2)
      return;
    }
    if constexpr (RR == 2) {
      // b_off[0] = plane + (j >> 1) * WCp + 4 * (j & 1): window row w (relative to the lane's first) is at + w * WCp
      f32x4 ya[2][MT];
      f32x4 xa[2];
      f32x2w xb[2];
      float V[3][6];   // V[w % 3]: relative window row w
#pragma unroll
      for (int m = 0; m < MT; ++m) ya[0][m] = *reinterpret_cast<const f32x4*>(&Sb[a_off[0] + m * 16 * DS]);
#pragma unroll
      for (int w = 0; w < 3; ++w) {
        xa[0] = *reinterpret_cast<const f32x4*>(&Sb[b_off[0] + w * P.WCp]);
        xb[0] = *reinterpret_cast<const f32x2w*>(&Sb[b_off[0] + w * P.WCp + 4]);
        row_transform(xa[0], xb[0], V[w]);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int cb = ks & 1, nb2 = cb ^ 1;
        if (ks + 1 < 4) {   // the next k-step's dy tile and its two new window rows fly during this k-step's MFMAs
#pragma unroll
          for (int m = 0; m < MT; ++m) ya[nb2][m] = *reinterpret_cast<const f32x4*>(&Sb[a_off[ks + 1] + m * 16 * DS]);
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            xa[i] = *reinterpret_cast<const f32x4*>(&Sb[b_off[0] + (2 * ks + 3 + i) * P.WCp]);
            xb[i] = *reinterpret_cast<const f32x2w*>(&Sb[b_off[0] + (2 * ks + 3 + i) * P.WCp + 4]);
          }
        }
        float U[MT][6];
#pragma unroll
        for (int m = 0; m < MT; ++m) {   // U = A dy
          const float y0 = ya[cb][m][0], y1 = ya[cb][m][1], y2 = ya[cb][m][2], y3 = ya[cb][m][3];
          const float p = y0 + y2, q = y1 + y3;
          const float a = fmaf(4.f, y2, y0), b = 2.f * fmaf(4.f, y3, y1);
          U[m][0] = y0;
          U[m][1] = p + q;
          U[m][2] = p - q;
          U[m][3] = a + b;
          U[m][4] = a - b;
          U[m][5] = y3;
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int f = 0; f < 6; ++f) acc[m][r * 6 + f] = mfma16(U[m][f], V[(2 * ks + r) % 3][f], acc[m][r * 6 + f]);
        if (ks + 1 < 4) {   // relative rows 2ks+3, 2ks+4 take the places of rows 2ks, 2ks+1
          row_transform(xa[0], xb[0], V[(2 * ks + 3) % 3]);
          row_transform(xa[1], xb[1], V[(2 * ks + 4) % 3]);
        }
        if (ks == WG43_LATE_KS && late && more) {
          WG43_STAMP(2)
          issue_dma(next_stage, cur ^ 1);
          WG43_STAMP(1)
        }
      }
      WG43_STAMP(2)
      return;
    }
    f32x4 ya[2][MT];
    f32x4 ra[2][3];
    f32x2w rb[2][3];
#pragma unroll
    for (int m = 0; m < MT; ++m) ya[0][m] = *reinterpret_cast<const f32x4*>(&Sb[a_off[0] + m * 16 * DS]);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      ra[0][r] = *reinterpret_cast<const f32x4*>(&Sb[b_off[0] + r * P.WCp]);
      rb[0][r] = *reinterpret_cast<const f32x2w*>(&Sb[b_off[0] + r * P.WCp + 4]);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int cb = ks & 1, nb2 = cb ^ 1;
      if (ks + 1 < 4) {   // the next k-step's raw operands fly during this one's transforms and MFMAs
#pragma unroll
        for (int m = 0; m < MT; ++m) ya[nb2][m] = *reinterpret_cast<const f32x4*>(&Sb[a_off[ks + 1] + m * 16 * DS]);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          ra[nb2][r] = *reinterpret_cast<const f32x4*>(&Sb[b_off[ks + 1] + r * P.WCp]);
          rb[nb2][r] = *reinterpret_cast<const f32x2w*>(&Sb[b_off[ks + 1] + r * P.WCp + 4]);
        }
      }
      float U[MT][6], V[3][6];
#pragma unroll
      for (int m = 0; m < MT; ++m) {   // U = A dy
        const float y0 = ya[cb][m][0], y1 = ya[cb][m][1], y2 = ya[cb][m][2], y3 = ya[cb][m][3];
        const float p = y0 + y2, q = y1 + y3;
        const float a = fmaf(4.f, y2, y0), b = 2.f * fmaf(4.f, y3, y1);
        U[m][0] = y0;
        U[m][1] = p + q;
        U[m][2] = p - q;
        U[m][3] = a + b;
        U[m][4] = a - b;
        U[m][5] = y3;
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {    // V = B^T relu(bn(raw))
        float d0 = ra[cb][r][0], d1 = ra[cb][r][1], d2 = ra[cb][r][2], d3 = ra[cb][r][3], d4 = rb[cb][r][0], d5 = rb[cb][r][1];
        if constexpr (!PLAIN) {
          d0 = fmaxf(fmaf(d0, sc, sh), lo), d1 = fmaxf(fmaf(d1, sc, sh), lo), d2 = fmaxf(fmaf(d2, sc, sh), lo);
          d3 = fmaxf(fmaf(d3, sc, sh), lo), d4 = fmaxf(fmaf(d4, sc, sh), lo), d5 = fmaxf(fmaf(d5, sc, sh), lo);
        }
        const float a = fmaf(-4.f, d2, d4), b = fmaf(-4.f, d1, d3);
        const float c = d4 - d2, e = 2.f * (d3 - d1);
        V[r][0] = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
        V[r][1] = a + b;
        V[r][2] = a - b;
        V[r][3] = c + e;
        V[r][4] = c - e;
        V[r][5] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int f = 0; f < 6; ++f) acc[m][r * 6 + f] = mfma16(U[m][f], V[r][f], acc[m][r * 6 + f]);
      if (ks == WG43_LATE_KS && late && more) {
        WG43_STAMP(2)
        issue_dma(next_stage, cur ^ 1);
        WG43_STAMP(1)
      }
    }
    WG43_STAMP(2)   // reads + transforms + 144 MFMAs
  };

  const int nst = s_end - s_begin;
  if constexpr (R3) {
    // DMA instructions this wave issues per stage (every one is issued whatever its lanes' validity): the counted wait below
    const int per_stage = (AX4 ? BM / 4 / NW : BM / NW) + (BN / NW) * npv;
    const bool late = wave >= 4;
    if (nst > 0) issue_dma(s_begin, 0);
    if (nst > 1) issue_dma(s_begin + 1, 1);
    WG43_STAMP(3)   // prologue
    int cur = 0, nxt = 2;   // image of stage `it`, image that stage it+2 is filled into
    for (int it = 0; it < nst; ++it) {
      // stage `it` has landed once every wave has seen all but its youngest fills (those of stage it+1, when there is one) land
      wg43_wait_vmcnt(it + 1 < nst ? per_stage : 0);
      __syncthreads();
      WG43_STAMP(0)
      const bool more = it + 2 < nst;
      if (!late && more) issue_dma(s_begin + it + 2, nxt);
      WG43_STAMP(1)
      compute(cur, false, 0, false);
      if (late && more) {
        issue_dma(s_begin + it + 2, nxt);
        WG43_STAMP(1)
      }
      cur = cur == 2 ? 0 : cur + 1;
      nxt = nxt == 2 ? 0 : nxt + 1;
    }
  } else {
    if (nst > 0) issue_dma(s_begin, 0);
    WG43_STAMP(3)   // prologue
    // the stage loop, unrolled by two: which of the two LDS images a stage reads is a constant of each copy, so the image offsets
    // fold into the instructions' immediate fields (as in gsd_conv3x3_w2d.hip)
    auto run_stage = [&](const int it, auto cur_c) {
      constexpr int cur = decltype(cur_c)::value;
      if constexpr (BX4) {
        __builtin_amdgcn_s_waitcnt(0x0F70);   // this wave's fills of the stage have landed
        const int fx = cur ? fix1 : fix0;
        if (fx) {   // a stage at the left / right image edge: the outside floats of the straddling pieces THIS lane moved get the
                    // padding value (behind the lane's own fills, in front of the barrier: no second barrier)
          float* Bb = smem + cur * BUF + BM * DS;
          const int gl0 = fx & 255, gl1 = fx >> 8 & 255, gr0 = fx >> 16 & 255, gr1 = fx >> 24 & 255;
#pragma unroll
          for (int k = 0; k < KB; ++k) {
            const int m = x_meta[k];
            if (wave + NW * k < NI && m >= 0) {
              const int c0 = m >> 8 & 255;
              float* pp = Bb + (wave + NW * k) * 256 + lane * 4;
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if ((c0 + e >= gl0 && c0 + e < gl1) || (c0 + e >= gr0 && c0 + e < gr1)) pp[e] = padv;
            }
          }
        }
#if defined(WG43_ABL) && ((WG43_ABL) & 4)   // diagnostic: no barrier per stage (results are then garbage)
#else
        __syncthreads();   // everyone's fills (and patches) are in; everyone has left the other image
#endif
      } else {
        gsd_dma_barrier();   // this stage's DMA has landed; everyone has left the other image
      }
      WG43_STAMP(0)
      // The barrier puts the two waves of a SIMD in phase, and a wave that issues its ~24 gathers (plus their address
      // work) keeps the matrix pipe idle: the SIMD's second wave (waves 4..7 of an 8-wave block) therefore multiplies its
      // first k-step BEFORE it issues its share of the next stage's DMA.
#ifdef WG43_NO_LATE   // diagnostic: every wave issues its fills in front of the stage
      const bool late = false;
#else
      const bool late = NW == 8 && wave >= 4;
#endif
      if (!late && it + 1 < nst) issue_dma(s_begin + it + 1, cur ^ 1);
      WG43_STAMP(1)   // this wave's share of the next stage's DMA
      compute(cur, late, s_begin + it + 1, it + 1 < nst);
    };
    for (int it = 0; it < nst; it += 2) {
      run_stage(it, std::integral_constant<int, 0>{});
      if (it + 1 < nst) run_stage(it + 1, std::integral_constant<int, 1>{});
    }
  }

#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int mr = m0 + wm * 32 + m * 16 + j * 4 + reg;
      const int col = n0 + wn * 16 + l16;
      if (mr < P.M && col < P.Ncols) {
        // G^T (6 frequencies -> 3 taps of kernel row r) HERE, per split: it is linear, so the slab reduction only adds -- and
        // the slabs are 9 planes instead of 18 (half the stores of this epilogue, half the reducer's reads)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const float D0 = acc[m][r * 6 + 0][reg], D1 = acc[m][r * 6 + 1][reg], D2 = acc[m][r * 6 + 2][reg];
          const float D3 = acc[m][r * 6 + 3][reg], D4 = acc[m][r * 6 + 4][reg], D5 = acc[m][r * 6 + 5][reg];
          float* const o = P.slabs + (((size_t)split * 9 + r * 3) * P.M + mr) * P.Ncols + col;
          const size_t pl = (size_t)P.M * P.Ncols;
          o[0] = 0.25f * D0 - (1.f / 6.f) * (D1 + D2) + (1.f / 24.f) * (D3 + D4);
          o[pl] = (1.f / 6.f) * (D2 - D1) + (1.f / 12.f) * (D3 - D4);
          o[2 * pl] = (1.f / 6.f) * (D3 + D4 - D1 - D2) + D5;
        }
      }
    }
#ifdef GSD_WG43_STAMPS
  WG43_STAMP(3)   // epilogue (slab stores)
  if (P.stamps != nullptr && lane == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) P.stamps[((size_t)blockIdx.x * NW + wave) * 4 + i] = st_acc[i];
  }
#endif
}

// slab[split][r*3+s][co][ci] -> dW[co][ci][r][s] = sum over splits, in a fixed order (the blocks applied G^T themselves).
// Block = 64 elements x 16 split lanes: the layers with few (co, ci) pairs are the ones with hundreds of splits, and one
// thread per element would walk them serially (95 us per launch on average before, most of it latency).
template <int WR_LANES>
__global__ __launch_bounds__(WR_LANES == 1 ? 256 : 64 * WR_LANES) void wgrad_w43_reduce_kernel(const float* __restrict__ slabs,
                                                                                              float* __restrict__ dw, int splits,
                                                                                              int M, int Ncols) {
  constexpr int EL = WR_LANES == 1 ? 256 : 64;   // elements per block
  const long long plane = (long long)M * Ncols;
  const long long total = 3 * plane;
  const int el = threadIdx.x % EL, sl = threadIdx.x / EL;
  __shared__ float red[3][WR_LANES][EL];
  for (long long e0 = (long long)blockIdx.x * EL; e0 < total; e0 += (long long)gridDim.x * EL) {
    const long long e = e0 + el;
    const bool ok = e < total;
    const int r = ok ? (int)(e / plane) : 0;
    const long long mc = ok ? e - r * plane : 0;
#pragma unroll
    for (int f = 0; f < 3; ++f) {
      float s = 0.f;
      if (ok)
        for (int k = sl; k < splits; k += WR_LANES) s += slabs[((size_t)k * 9 + r * 3 + f) * plane + mc];
      red[f][sl][el] = s;
    }
    __syncthreads();
    if (sl == 0 && ok) {
      float* o = dw + (size_t)mc * 9 + r * 3;
#pragma unroll
      for (int f = 0; f < 3; ++f) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < WR_LANES; ++i) s += red[f][i][el];
        o[f] = s;
      }
    }
    __syncthreads();
  }
}

namespace {

struct WgW43Plan {
  int TH, TW, TWq, tiles_y, tiles_x, WR, WC, WCp, XS, BM, BN, mblocks, nblocks, stages_total, splits;
  int64_t slab_elems;
};

WgW43Plan plan_wg43(int N, int H, int W, int M, int Ncols) {
  WgW43Plan p;
  long best = -1;
  const int force_tw = gsd_env_int("GSD_WG43_TW", 0);   // tuning
  for (int tw = 4; tw <= 64; tw *= 2) {   // 16 tiles = TH rows x TW/4 tiles
    if (force_tw && tw != force_tw) continue;
    const int th = 64 / tw;
    const long tiles = (long)ceil_div(H, th) * ceil_div(W, tw);
    // fewest stages, weighted by what a stage of that shape costs (measured, profiles/bench_wgrad_forms.py with
    // GSD_WG43_TW: 4 x 16 is the cheapest, 8 x 8 and 2 x 32 are ~7 % dearer, 1 x 64 and 16 x 4 much dearer)
    const long cost = tiles * (tw == 16 ? 100 : (tw == 8 || tw == 32) ? 107 : 160);
    if (best < 0 || cost < best) {
      best = cost;
      p.TH = th; p.TW = tw;
    }
  }
  p.TWq = p.TW / 4;
  p.tiles_y = ceil_div(H, p.TH);
  p.tiles_x = ceil_div(W, p.TW);
  p.WR = p.TH + 2; p.WC = p.TW + 2; p.WCp = round_up(p.WC, 4);
  // channel stride: whole 64-lane DMA instructions land inside the channel's slot, and = 4 mod 8 floats so that the 16
  // channels of a ds_read_b128 hit 16 different bank groups
  p.XS = round_up(((p.WR * p.WCp + 63) / 64) * 64, 8) + 4;
  const bool small = gsd_env_set("GSD_WG43_SMALL");   // tuning: 4-wave blocks only
  p.BM = (M >= 128 && !small) ? 128 : 64;
  p.BN = (p.BM == 64 && Ncols >= 64 && !small) ? 64 : 32;
  p.mblocks = ceil_div(M, p.BM);
  p.nblocks = ceil_div(Ncols, p.BN);
  p.stages_total = N * p.tiles_y * p.tiles_x;
  const int target = gsd_env_int("GSD_WGRAD_BLOCKS", 512);
  int splits = ceil_div(p.BM * p.BN > 64 * 32 ? target / 2 : target, p.mblocks * p.nblocks);   // one round of resident blocks
  if (splits > p.stages_total) splits = p.stages_total;
  if (splits > 2048) splits = 2048;
  if (splits < 1) splits = 1;
  p.splits = splits;
  p.slab_elems = (int64_t)splits * 9 * M * Ncols;
  return p;
}

}  // namespace

// 1: gsd_conv3x3_wgrad serves this shape with the Winograd form.  GSD_WGRAD_ALGO=0|1 forces one (tuning, A/B runs).
int gsd_wgrad_w43_use(int N, int H, int W, int Cin, int Cout) {
  const char* env = getenv("GSD_WGRAD_ALGO");   // read per call: the tests switch forms inside one process
  const int forced = env ? atoi(env) : -1;
  if (forced == 0) return 0;
  if (forced == 1) return 1;
  return Cin >= 16 && Cout >= 16;   // the 3-channel first layer keeps the pixel-split direct kernel
}

// 1 when gsd_conv3x3_wgrad serves this shape with a kernel that takes a pitched dy (w_stride > W): the engine then lets
// gsd_bn_bwd_apply write d_raw into a pitched buffer.
extern "C" int gsd_conv3x3_wgrad_takes_pitched_dy(int N, int H, int W, int Cin, int Cout) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  return gsd_wgrad_w43_use(N, H, W, Cin, Cout);
}

// MFMA instructions of one launch: stages x 18 frequencies-rows x 16 tiles / 4 per (16 co x 16 ci) pair of the padded blocks
int64_t gsd_wgrad_w43_mfma_count(int N, int H, int W, int Cin, int Cout) {
  const WgW43Plan p = plan_wg43(N, H, W, Cout, Cin);
  return (int64_t)p.stages_total * 72 * (p.mblocks * p.BM / 16) * (p.nblocks * p.BN / 16);
}

int64_t gsd_wgrad_w43_workspace(int N, int H, int W, int Cin, int Cout) { return plan_wg43(N, H, W, Cout, Cin).slab_elems; }

// slab[split][9][Cout][Cin] -> dW (Cout,Cin,3,3): the ordered split sum, shared with the 2-D form (gsd_wgrad_w2d.hip)
int gsd_wgrad_w43_reduce_run(const float* workspace, float* dw, int splits, int Cout, int Cin, void* stream) {
  const long long per = 3LL * Cout * Cin;
  if (splits >= 64) {
    const int rgrid = (int)(ceil_div64(per, 64) < 8192 ? ceil_div64(per, 64) : 8192);
    hipLaunchKernelGGL(wgrad_w43_reduce_kernel<16>, dim3(rgrid), dim3(1024), 0, (hipStream_t)stream, workspace, dw, splits, Cout, Cin);
  } else if (splits >= 8) {
    const int rgrid = (int)(ceil_div64(per, 64) < 8192 ? ceil_div64(per, 64) : 8192);
    hipLaunchKernelGGL(wgrad_w43_reduce_kernel<4>, dim3(rgrid), dim3(256), 0, (hipStream_t)stream, workspace, dw, splits, Cout, Cin);
  } else {
    const int rgrid = (int)(ceil_div64(per, 256) < 8192 ? ceil_div64(per, 256) : 8192);
    hipLaunchKernelGGL(wgrad_w43_reduce_kernel<1>, dim3(rgrid), dim3(256), 0, (hipStream_t)stream, workspace, dw, splits, Cout, Cin);
  }
  GSD_LAUNCH_CHECK("gsd_conv3x3_wgrad (w43) reduce");
  return GSD_OK;
}

// arguments already validated by gsd_conv3x3_wgrad
int gsd_wgrad_w43_run(const gsd_src* a, int nsrc, const gsd_src* dy, int Cin, int Cout, float* dw, float* workspace,
                      int64_t workspace_elems, int N, int H, int W, void* stream) {
  const WgW43Plan pl = plan_wg43(N, H, W, Cout, Cin);
  GSD_REQUIRE(workspace_elems >= pl.slab_elems, GSD_ERR_WORKSPACE, "gsd_conv3x3_wgrad: workspace %lld < %lld elements",
              (long long)workspace_elems, (long long)pl.slab_elems);
  WgW43Params P;
  P.a0 = to_srcd(a[0]);
  P.a1 = nsrc > 1 ? to_srcd(a[1]) : null_srcd();
  P.dy = to_srcd(*dy);
  P.slabs = workspace;
  P.stamps = nullptr;
#ifdef GSD_WG43_STAMPS
  P.stamps = g_wg43_stamp_buf;
#endif
  P.M = Cout; P.Ncols = Cin;
  P.N = N; P.H = H; P.W = W;
  P.TH = pl.TH; P.TW = pl.TW; P.TWq = pl.TWq; P.tiles_y = pl.tiles_y; P.tiles_x = pl.tiles_x;
  P.WR = pl.WR; P.WC = pl.WC; P.WCp = pl.WCp; P.XS = pl.XS;
  P.stages_total = pl.stages_total; P.splits = pl.splits; P.mblocks = pl.mblocks; P.nblocks = pl.nblocks;
  GSD_REQUIRE(pl.WR * pl.WCp <= 256, GSD_ERR_UNSUPPORTED, "gsd_conv3x3_wgrad: halo window too large");
  const long grid = (long)pl.splits * pl.mblocks * pl.nblocks;
  // dy as aligned 16-byte pieces: rows, planes and images of the gradient buffer start 16-byte aligned
  const bool ax4 = gsd_env_int("GSD_WG43_AX4", 1) != 0 && dy->w_stride % 4 == 0 && ((uintptr_t)dy->ptr & 15) == 0 &&
                   dy->c_stride % 4 == 0 && dy->n_stride % 4 == 0 && pl.TW % 4 == 0;
  // three LDS images + half-stage stagger for the 8-wave forms (GSD_WG43_R3=1; measured 2 % SLOWER: a wave alone needs ~66
  // cycles per MFMA in its multiply phase -- hipcc sinks the operand reads to their uses at 229 of 256 registers -- so the
  // partner's fills do not lie beside idle pipe time)
  const bool eight = pl.BM == 128 || pl.BN == 64;
  // activation windows as 16-byte pieces from the unaligned rows: every block's channels lie in one segment, the caller
  // vouches for 4 readable floats around each segment tensor (straddling pieces), at most 4 instructions per wave
  bool bx4 = gsd_env_int("GSD_WG43_BX4", 1) != 0 && ax4 && gsd_env_int("GSD_WG43_R3", 0) == 0 &&
             (nsrc == 1 || a[0].C % pl.BN == 0);
  for (int i = 0; i < nsrc; ++i) bx4 = bx4 && a[i].slack >= 4;
  if (bx4) {
    int xs = pl.WR * pl.WCp;          // WCp % 4 == 0
    if ((xs / 4) % 2 == 0) xs += 4;   // an odd number of pieces per plane
    const int nw = (pl.BM == 128 || pl.BN == 64) ? 8 : 4;
    const int ni = (pl.BN * (xs / 4) + 63) / 64;
    bool fits = (ni + nw - 1) / nw <= 4 && pl.WR < 256 && pl.WCp < 256;
    for (int i = 0; i < nsrc; ++i) fits = fits && (int64_t)pl.BN * a[i].c_stride < (1LL << 31);
    if (fits) P.XS = xs;
    else bx4 = false;
  }
  // row reuse: the 4 x 16 stage (k-step = tile row)
  const int rr = (bx4 && gsd_env_int("GSD_WG43_RR", 1) != 0) ? (pl.TW == 16 && pl.TH == 4 ? 1 : (pl.TW == 8 && pl.TH == 8 ? 2 : 0)) : 0;
  const size_t img = (size_t)(pl.BM * (ax4 ? WG_DS_X4 : WG_DS) + (bx4 ? (pl.BN * (P.XS / 4) + 63) / 64 * 256 : pl.BN * P.XS)) * sizeof(float);
  const bool r3 = gsd_env_int("GSD_WG43_R3", 0) != 0 && eight && 3 * img <= 160 * 1024;
  const size_t lds = (r3 ? 3 : 2) * img;
  bool plain = gsd_env_int("GSD_WG43_PLAIN", 1) != 0;   // no deferred BatchNorm / ReLU on any activation segment
  for (int i = 0; i < nsrc; ++i) plain = plain && a[i].scale == nullptr && a[i].relu == 0;
  if (gsd_env_set("GSD_WG43_TRACE"))   // tuning: one line per launch
    fprintf(stderr, "wg43 M%d N%d %dx%d B%d tile %dx%d stages %d splits %d blocks %ld BM %d BN %d ax4 %d bx4 %d rr %d plain %d lds %zu\n", Cout,
            Cin, H, W, N, pl.TH, pl.TW, pl.stages_total, pl.splits, grid, pl.BM, pl.BN, (int)ax4, (int)bx4, rr, (int)plain, lds);
  const dim3 g((int)grid);
  const hipStream_t st = (hipStream_t)stream;
  // one launcher per instantiation: the kernel's address keys the per-device cache of the launch attribute (gsd_common.h)
#define WG43_LAUNCH(NWM_, NWN_, AX4_, R3_, PL_, BX_, RR_)                                                                          \
  do {                                                                                                                  \
    static gsd_attr_once once;                                                                                          \
    const void* fn = reinterpret_cast<const void*>(&wgrad3x3_w43_kernel<NWM_, NWN_, AX4_, R3_, PL_, BX_, RR_>);                          \
    if (hipError_t e = gsd_allow_big_lds(once, fn); e != hipSuccess) {                                                  \
      gsd_set_error("gsd_conv3x3_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));                                \
      return GSD_ERR_HIP;                                                                                               \
    }                                                                                                                   \
    hipLaunchKernelGGL((wgrad3x3_w43_kernel<NWM_, NWN_, AX4_, R3_, PL_, BX_, RR_>), g, dim3(64 * NWM_ * NWN_), lds, st, P);             \
  } while (0)
  if (pl.BM == 128) {
    if (ax4 && r3) WG43_LAUNCH(4, 2, true, true, false, false, 0);
    else if (bx4 && rr == 1 && plain) WG43_LAUNCH(4, 2, true, false, true, true, 1);
    else if (bx4 && rr == 2 && plain) WG43_LAUNCH(4, 2, true, false, true, true, 2);
    else if (bx4 && rr == 1) WG43_LAUNCH(4, 2, true, false, false, true, 1);
    else if (bx4 && rr == 2) WG43_LAUNCH(4, 2, true, false, false, true, 2);
    else if (bx4 && plain) WG43_LAUNCH(4, 2, true, false, true, true, 0);
    else if (bx4) WG43_LAUNCH(4, 2, true, false, false, true, 0);
    else if (ax4 && plain) WG43_LAUNCH(4, 2, true, false, true, false, 0);
    else if (ax4) WG43_LAUNCH(4, 2, true, false, false, false, 0);
    else if (r3) WG43_LAUNCH(4, 2, false, true, false, false, 0);
    else WG43_LAUNCH(4, 2, false, false, false, false, 0);
  } else if (pl.BN == 64) {
    if (ax4 && r3) WG43_LAUNCH(2, 4, true, true, false, false, 0);
    else if (bx4 && rr == 1 && plain) WG43_LAUNCH(2, 4, true, false, true, true, 1);
    else if (bx4 && rr == 2 && plain) WG43_LAUNCH(2, 4, true, false, true, true, 2);
    else if (bx4 && rr == 1) WG43_LAUNCH(2, 4, true, false, false, true, 1);
    else if (bx4 && rr == 2) WG43_LAUNCH(2, 4, true, false, false, true, 2);
    else if (bx4 && plain) WG43_LAUNCH(2, 4, true, false, true, true, 0);
    else if (bx4) WG43_LAUNCH(2, 4, true, false, false, true, 0);
    else if (ax4 && plain) WG43_LAUNCH(2, 4, true, false, true, false, 0);
    else if (ax4) WG43_LAUNCH(2, 4, true, false, false, false, 0);
    else if (r3) WG43_LAUNCH(2, 4, false, true, false, false, 0);
    else WG43_LAUNCH(2, 4, false, false, false, false, 0);
  } else {
    if (bx4 && rr == 1 && plain) WG43_LAUNCH(2, 2, true, false, true, true, 1);
    else if (bx4 && rr == 2 && plain) WG43_LAUNCH(2, 2, true, false, true, true, 2);
    else if (bx4 && rr == 1) WG43_LAUNCH(2, 2, true, false, false, true, 1);
    else if (bx4 && rr == 2) WG43_LAUNCH(2, 2, true, false, false, true, 2);
    else if (bx4 && plain) WG43_LAUNCH(2, 2, true, false, true, true, 0);
    else if (bx4) WG43_LAUNCH(2, 2, true, false, false, true, 0);
    else if (ax4 && plain) WG43_LAUNCH(2, 2, true, false, true, false, 0);
    else if (ax4) WG43_LAUNCH(2, 2, true, false, false, false, 0);
    else WG43_LAUNCH(2, 2, false, false, false, false, 0);
  }
#undef WG43_LAUNCH
  GSD_LAUNCH_CHECK("gsd_conv3x3_wgrad (w43)");
  return gsd_wgrad_w43_reduce_run(workspace, dw, pl.splits, Cout, Cin, stream);
}
