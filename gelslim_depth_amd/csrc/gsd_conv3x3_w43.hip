// gsd_conv3x3_w43.hip -- conv3x3 (pad 1, stride 1, no bias) forward and dX with the Winograd minimal-filtering
// identity F(4,3) applied ALONG THE IMAGE ROWS, on v_mfma_f32_16x16x4_f32 (gfx950).
//
// Same operator as gsd_conv3x3.hip (aten::convolution at /root/reference/gelslim_depth/models/unet.py:11,14 and the dX
// half of aten::convolution_backward), same fp32 storage and fp32 accumulation, half the multiplications: four
// horizontally adjacent outputs of one kernel row need 6 products instead of 12,
//
//   y[0..3] = A^T [ (G g) .* (B^T d) ],   d = 6 consecutive inputs of the row, g = the 3 taps of one kernel row,
//
// and the three kernel rows and the input channels are the contraction that stays on the MFMA:
//
//   M_f[m][tile] = sum_{ci, r} U_f[(ci,r)][m] * V_f[(ci,r)][tile],   f = 0..5
//
// i.e. six GEMMs with K = 3*Cin instead of one with K = 9*Cin, over N = pixels/4 "tiles".  Accumulators: 6 per 4
// pixels (1.5x the direct form).  U = G g is computed once per optimiser step by gsd_weight_layout (modes 4/5), V = B^T d
// by the consumer lanes between the ds_read and the MFMA (13 VALU operations per 6 values, next to the deferred
// BatchNorm+ReLU that is already applied there), y = A^T M in the epilogue.  F(4,3) in one dimension is numerically
// benign in fp32 (largest transform constants 8 and 1/24): the op-level tests bound it at 1e-5 relative L1 against
// the fp64 oracle, two orders below the 1e-3 north-star tolerance.
//
// Block = 4 waves, all on the same 64 output channels; each wave owns 16 tiles = 64 pixels: block tile 64 (m) x 256
// pixels.  Per 4-channel chunk a wave issues 18 k-steps (3 kernel rows x 6 frequencies) x 4 MFMAs; the weight image
// of a chunk is 72 x 64 floats = 18 KiB, exactly the direct kernel's (36 x 128), and it is amortised over 256 pixels
// so the L2 -> LDS rate per MFMA-second equals the direct 128 x 128 kernel's although the MFMA time halves.
// Data movement, deferred BatchNorm, NaN-sentinel padding, concat sources, two cropped destinations, BatchNorm
// partial sums and the fused BatchNorm-backward dX epilogue are those of gsd_conv3x3.hip.
#include "gsd_common.h"

#include <cstdlib>

__device__ const float gsd_pad_w43[2] = {0.f, __builtin_nanf("")};
__device__ __attribute__((aligned(16))) const float gsd_pad16_w43[8] = {0.f, 0.f, 0.f, 0.f, __builtin_nanf(""), __builtin_nanf(""),
                                                                          __builtin_nanf(""), __builtin_nanf("")};

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Diagnostic build only (-DGSD_W43_STAMPS; never in the product library): s_memtime stamps around the segments of a K-chunk,
// summed per wave in scalar registers and written to a buffer of their own (cdna_hip_programming.md sec. 7, In-kernel stamps).
#ifdef GSD_W43_STAMPS
static unsigned long long* g_w43_stamp_buf = nullptr;
extern "C" void gsd_w43_set_stamp_buffer(void* p) { g_w43_stamp_buf = (unsigned long long*)p; }
#define W43_STAMP(i)                                                                             \
  {                                                                                              \
    unsigned long long t_;                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                   \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    st_acc[i] += t_ - st_prev;                                                                   \
    st_prev = t_;                                                                                \
  }
#else
#define W43_STAMP(i) {}
#endif

struct W43Params {
  SrcD src0, src1;
  DstD dst0, dst1;
  const float* wt;   // [mblocks][nchunks*72][64]: row (ci_local*18 + r*6 + f), columns permuted (slot l*4+t = column t*16+l)
  float* partials;
  const float* bw_raw;
  const float* bw_scale;
  const float* bw_shift;
  const float* bw_mean;
  const float* bw_invstd;
  int Cin, Cout, Mpad, nchunks, mblocks;
  int N, H, W;
  int TH, TW, TWq, tiles_y, tiles_x, WR, WC, WCp, PS, NPV;
  int NP, RPI, NI, RO;   // 16-byte halo pieces (X4): pieces per window row, rows per DMA instruction, instructions per plane, read offset
  int fold;              // tile rows run over the padded flat rows of the whole batch (see w43_row)
  int nslab;             // K-slab form (SPLIT): the block's channel chunks are slab k's share, its y goes to `slabs`
  float* slabs;          // [nslab][tile blocks][64 channels][64 Winograd tiles][4 pixels]: un-reduced outputs of the K slabs
  unsigned long long* stamps;   // diagnostic builds only
};

namespace {
constexpr int W43_BM = 64;
constexpr int W43_WTILE = 72 * W43_BM;      // floats per weight chunk
constexpr int W43_W4 = W43_WTILE / 4;       // float4s
constexpr int W43_NWI = (W43_W4 + 255) / 256;
}  // namespace

// Row folding (small images): a 20 x 26 image fills a 64-tile block to 68 %.  With P.fold the tile grid runs over the
// "padded flat rows" of the WHOLE batch, rho = n * (H + 1) + h + 1: row n * (H + 1) is a zero row that serves as the bottom
// padding of image n-1 and as the top padding of image n, so a block's halo window is still TH + 2 consecutive rows and
// nothing else in the kernel changes -- a block simply covers the tail of one image and the head of the next.  Returns the
// image and the row inside it (-1: a zero row or a row past the batch).
__device__ __forceinline__ void w43_row(const W43Params& P, int n, int rho, int& nn, int& hh) {
  if (!P.fold) {
    nn = n;
    hh = rho;
    return;
  }
  nn = rho >= 0 ? rho / (P.H + 1) : 0;
  hh = (rho >= 0 && nn < P.N) ? rho - nn * (P.H + 1) - 1 : -1;
  if (nn >= P.N) nn = 0;
}

// Loader wave of the producer / consumer form of the kernel below (NL > 0): it issues EVERY LDS-DMA of the block -- the 18
// one-KiB pieces of the weight chunk and the halo windows of the chunk's four input channels -- one chunk ahead of the four
// MFMA waves, which then run a loop of ds_read + transform + MFMA only.  Measured with in-kernel stamps (profiles/
// stamp_conv.py): a wave that issues its share of the fills itself spends ~250 cycles per global_load_lds (plus the
// segment / padding bookkeeping around it) and the five k-steps that carried them took as long as the other thirteen
// together; a wave that does nothing else issues them back to back (cdna_hip_programming.md, LDS-DMA loader rings).
template <bool X4, int NL>
__device__ __forceinline__ void w43_loader(const W43Params& P, float* smem, int BUF, int n, int h0, int w0, int mb, int lane,
                                           int lw) {
  constexpr int WTILE = W43_WTILE;
  constexpr int NIL = ((X4 ? 2 : 8) + NL - 1) / NL;   // halo instructions per loader wave and channel plane
  const int PS = P.PS;
  int xo0[NIL], xo1[NIL], ldo[NIL];
#pragma unroll
  for (int k = 0; k < NIL; ++k) {
    const int idx = lw + NL * k;
    xo0[k] = xo1[k] = -2;
    if constexpr (X4) {
      ldo[k] = idx * P.RPI * P.WCp + 1;            // + 1 float: image column w0-1 then sits 16-byte aligned
      const int rl = lane / P.NP, pc = lane - rl * P.NP;
      const int rr = idx * P.RPI + rl;
      if (idx < P.NI && rl < P.RPI && rr < P.WR) {
        const int gh = h0 - 1 + rr, gw = w0 - 4 + 4 * pc;
        int hs = gh - P.src0.oh, ws = gw - P.src0.ow;
        xo0[k] = ((unsigned)hs < (unsigned)P.src0.H && ws >= 0 && ws < P.src0.W && ws + 4 <= P.src0.ws) ? hs * P.src0.ws + ws : -1;
        hs = gh - P.src1.oh;
        ws = gw - P.src1.ow;
        xo1[k] = ((unsigned)hs < (unsigned)P.src1.H && ws >= 0 && ws < P.src1.W && ws + 4 <= P.src1.ws) ? hs * P.src1.ws + ws : -1;
      }
    } else {
      ldo[k] = idx * 64;
      const int pos = idx * 64 + lane;
      const int rr = pos / P.WCp, cc = pos - rr * P.WCp;
      if (idx < P.NPV && rr < P.WR && cc < P.WC) {
        const int gh = h0 - 1 + rr, gw = w0 - 1 + cc;
        int hs = gh - P.src0.oh, ws = gw - P.src0.ow;
        xo0[k] = ((unsigned)hs < (unsigned)P.src0.H && (unsigned)ws < (unsigned)P.src0.W) ? hs * P.src0.ws + ws : -1;
        hs = gh - P.src1.oh;
        ws = gw - P.src1.ow;
        xo1[k] = ((unsigned)hs < (unsigned)P.src1.H && (unsigned)ws < (unsigned)P.src1.W) ? hs * P.src1.ws + ws : -1;
      }
    }
  }
  {
    // padding positions of the first segment: written once, in both images and all four channel planes; a fill of a
    // first-segment channel then only moves the lanes that have a pixel
    const float pad0 = P.src0.relu ? __builtin_nanf("") : 0.f;
#pragma unroll
    for (int k = 0; k < NIL; ++k)
      if (xo0[k] == -1) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int ch = 0; ch < 4; ++ch) {
            float* d = smem + b * BUF + WTILE + ch * PS + ldo[k];
            if constexpr (X4) {
#pragma unroll
              for (int e = 0; e < 4; ++e) d[lane * 4 + e] = pad0;
            } else {
              d[lane] = pad0;
            }
          }
      }
  }
  int d_seg = 0, d_left = P.src0.C;
  const float* d_base = P.src0.p + (long long)n * P.src0.ns;
  long long d_cs = P.src0.cs;
  const float* const zeros = X4 ? &gsd_pad16_w43[0] : &gsd_pad_w43[0];
  const float* d_sent = X4 ? (P.src0.relu ? &gsd_pad16_w43[4] : &gsd_pad16_w43[0]) : (P.src0.relu ? &gsd_pad_w43[1] : &gsd_pad_w43[0]);
  const float* const wsrc = P.wt + (size_t)mb * P.nchunks * WTILE + lane * 4;

  auto fill = [&](int chunk, int buf) {
    float* Wb = smem + buf * BUF;
    const float* wc = wsrc + (size_t)chunk * WTILE;
#pragma unroll
    for (int i = 0; i < 18; ++i)
      if (i % NL == lw) __builtin_amdgcn_global_load_lds(wc + i * 256, Wb + i * 256, 16, 0, 0);
    float* Xb = Wb + WTILE;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      if (d_left == 0 && d_seg == 0) {
        d_seg = 1;
        d_left = P.src1.C;
        d_base = P.src1.p + (long long)n * P.src1.ns;
        d_cs = P.src1.cs;
        d_sent = X4 ? (P.src1.relu ? &gsd_pad16_w43[4] : &gsd_pad16_w43[0]) : (P.src1.relu ? &gsd_pad_w43[1] : &gsd_pad_w43[0]);
      }
      const bool c_ok = d_left > 0;
      if (c_ok && d_seg == 0) {
#pragma unroll
        for (int k = 0; k < NIL; ++k)
          if (xo0[k] >= 0) {
            const float* gp = d_base + xo0[k];
            __builtin_amdgcn_global_load_lds(gp, Xb + ch * PS + ldo[k], X4 ? 16 : 4, 0, 0);
          }
      } else {
        // second (concat) segment and K padding: every window position is written, padding from the sentinel
#pragma unroll
        for (int k = 0; k < NIL; ++k) {
          const int xo = d_seg == 0 ? xo0[k] : xo1[k];
          if (xo != -2) {
            const float* gp = (c_ok && xo >= 0) ? d_base + xo : (c_ok ? d_sent : zeros);
            __builtin_amdgcn_global_load_lds(gp, Xb + ch * PS + ldo[k], X4 ? 16 : 4, 0, 0);
          }
        }
      }
      if (c_ok) {
        d_base += d_cs;
        --d_left;
      }
    }
  };

  fill(0, 0);
  for (int chunk = 0; chunk < P.nchunks; ++chunk) {
    gsd_dma_barrier();   // chunk `chunk` has landed (this wave's vmcnt); the MFMA waves have left the other image
    if (chunk + 1 < P.nchunks) fill(chunk + 1, (chunk + 1) & 1);
  }
}

// Epilogue shared by the convolution kernel (y = A^T M of its accumulators) and the K-slab reducer (y = sum of the slabs' tiles): NCHW
// stores into two destination segments with crop, BatchNorm partial sums, or the fused BatchNorm-backward form.  The lane owns the
// four pixels (h_t, w0 + 4 tq ..) of image n_t (fold) / n for the 16 output channels m0 + m * 16 + j * 4 + reg.
template <int WM, class GetY>
__device__ __forceinline__ void w43_epilogue(const W43Params& P, int n, int pt, int wave, int wm, int mbb, int m0, int j, int l16,
                                             int n_t, int h_t, int w0, int tq, int vmask, const float* sBw, GetY&& get_y) {
  constexpr int MT = 4, BM = W43_BM, BMB = WM * BM;
  // ---- epilogue: y = A^T M, NCHW stores (two destination segments with crop), BatchNorm partial sums ----------------------
  // per destination: element offset of the tile's first pixel inside a plane, and the mask of its pixels that are stored
  int off0 = 0, off1 = 0, sm0 = 0, sm1 = 0;
  {
    const int h = h_t, w = w0 + 4 * tq;
    const int fo0 = P.fold ? n_t * (int)P.dst0.ns : 0, fo1 = P.fold ? n_t * (int)P.dst1.ns : 0;
    int hd = h - P.dst0.oh, wd = w - P.dst0.ow;
    if (h >= 0 && (unsigned)hd < (unsigned)P.dst0.H) {
      off0 = fo0 + hd * P.dst0.ws + wd;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if ((vmask >> i & 1) && (unsigned)(wd + i) < (unsigned)P.dst0.W) sm0 |= 1 << i;
    }
    hd = h - P.dst1.oh;
    wd = w - P.dst1.ow;
    if (h >= 0 && (unsigned)hd < (unsigned)P.dst1.H) {
      off1 = fo1 + hd * P.dst1.ws + wd;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if ((vmask >> i & 1) && (unsigned)(wd + i) < (unsigned)P.dst1.W) sm1 |= 1 << i;
    }
  }
  float* const d0 = P.dst0.p + (long long)n * P.dst0.ns;
  float* const d1 = P.dst1.p + (long long)n * P.dst1.ns;
  float* const prow = P.partials != nullptr ? P.partials + (size_t)(pt * 4 + wave) * (2 * P.Mpad) : nullptr;

  if (P.bw_raw == nullptr) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int co = m0 + m * 16 + j * 4 + reg;
        const bool first = co < P.dst0.C;
        const int cd = first ? co : co - P.dst0.C;
        const bool co_ok = co < P.Cout && (first || cd < P.dst1.C);
        float* const px = (first ? d0 + (long long)cd * P.dst0.cs : d1 + (long long)cd * P.dst1.cs) + (first ? off0 : off1);
        const int sm = co_ok ? (first ? sm0 : sm1) : 0;
        float y[4];
        get_y(m, reg, y);
        // statistics over the pixels that are STORED (the destination's window: for a cropped second destination -- the
        // backward of F.pad -- the sums are those of the crop, e.g. the ConvT bias gradient; the same pixels as `vmask` for a
        // full-size destination)
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (sm >> i & 1) {
            s1 += y[i];
            s2 = fmaf(y[i], y[i], s2);
          }
        }
        if (sm == 15) {
          *reinterpret_cast<f32x4u*>(px) = f32x4{y[0], y[1], y[2], y[3]};
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (sm >> i & 1) px[i] = y[i];
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
    // dst0 is the gradient buffer of a conv+BN+ReLU unit whose raw output has the same geometry: dz = relu'(bn(raw)) * dX.
    // Loads and stores share vmcnt: the raw values of one m-tile (4 rows x 4 pixels) are loaded together in front of its
    // stores, the coefficients come from LDS.
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float xr[4][4];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int co = m0 + m * 16 + j * 4 + reg;
        const float* const rp = P.bw_raw + (long long)n * P.dst0.ns + (long long)(co < P.Cout ? co : 0) * P.dst0.cs + off0;
        if (sm0 == 15) {
          const f32x4 t = *reinterpret_cast<const f32x4u*>(rp);
          xr[reg][0] = t[0]; xr[reg][1] = t[1]; xr[reg][2] = t[2]; xr[reg][3] = t[3];
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) xr[reg][i] = (sm0 >> i & 1) ? rp[i] : 0.f;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int co = m0 + m * 16 + j * 4 + reg;
        float* const px = d0 + (long long)co * P.dst0.cs + off0;
        const int cl = wm * BM + m * 16 + j * 4 + reg;
        const float bsc = sBw[cl], bsh = sBw[BMB + cl], bmu = sBw[2 * BMB + cl], bis = sBw[3 * BMB + cl];
        const int sm = co < P.Cout ? sm0 : 0;
        float y[4];
        get_y(m, reg, y);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float x = xr[reg][i];
          const float dz = ((sm >> i & 1) && fmaf(x, bsc, bsh) > 0.f) ? y[i] : 0.f;
          y[i] = dz;
          s1 += dz;
          s2 = fmaf(dz, (x - bmu) * bis, s2);
        }
        if (sm == 15) {
          *reinterpret_cast<f32x4u*>(px) = f32x4{y[0], y[1], y[2], y[3]};
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (sm >> i & 1) px[i] = y[i];
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

// WM = groups of 4 waves per block: 1 -> 64 m x 256 px, two blocks per CU; 2 -> 128 m x 256 px (two 64-channel weight
// images side by side), 8 waves, one block per CU: the halo DMA -- the expensive part of the data movement -- is then shared
// by twice the MFMAs (9 instead of 13 DMA instructions per wave and chunk).  Tuning option (GSD_W43_BIG), not the default.
//
// X4: the halo windows move as ALIGNED 16-byte pieces (global_load_lds_dwordx4 with a per-lane source address) instead of
// dword gathers: possible when every source row starts 16-byte aligned (row pitch, plane and image strides multiples of 4
// floats, pad offset a multiple of 4) -- the engine allocates its activations that way.  A window row is then the NP =
// TW/4 + 2 pieces that cover image columns w0-4 .. w0+TW+3, and the wave-uniform LDS base is shifted by ONE float so that
// image column w0-1 lands 16-byte aligned and the consumer reads stay one b128 + one b64 per kernel row.  A quarter of the
// gather instructions (2 per channel plane instead of 6), each a coalesced run of 16-byte lanes.  Columns >= W inside the
// pitch come from memory: producers keep the padding value there (NaN under a ReLU'd BatchNorm, else 0).
//
// NL > 0: producer / consumer form -- NL extra waves per block issue all the DMA (w43_loader above); WM must be 1.
//
// FAST: straight halo fills, no per-slot bookkeeping (which segment, how many channels it has left, which sentinel: a
// chain of scalar branches per slot in the general form): possible when a chunk of 4 channels never straddles the two
// source segments and Cin % 4 == 0 (always true in the U-Net).  Stamps (profiles/stamp_conv.py) put the per-slot form at
// ~1200 of the cycles a wave spends per chunk; measured -7 % kernel time over the U-Net's layer set.
//
// PLAIN: no source segment carries a deferred BatchNorm or ReLU (every dX launch: dy is plain; the pooled and upsampled
// sources of the forward): the operand transform drops its 12 fma/max per kernel row and the per-chunk scale / shift
// reads.  Measured with the transform forced plain over the layer set: -2.5 % kernel time.
//
// X4M = 2 ("U4"): the halo moves as 16-byte pieces straight from UNALIGNED rows (a global_load_lds_dwordx4 takes any 4-byte
// aligned global address at full rate, profiles/ubench/dma_global_align.hip).  The four channel planes of a chunk lie back to
// back in LDS (plane = PS / 4 pieces: WR rows x WCp / 4 pieces + the bank-spread dummies) and an instruction's 64 lanes are 64
// consecutive pieces of that image: ceil(PS / 64) (7 for a 6 x 68 window) instead of 32 exec-masked dword instructions per
// chunk and block.  Pieces wholly outside the image come from a 16-byte sentinel (no prefilled padding, nothing masked);
// a piece that STRADDLES the left or right image edge is loaded as it lies in memory (the caller vouches for 4 readable floats
// around the tensor: gsd_src.slack) and the lane that moved it overwrites its outside floats with the padding value once its
// own fills have landed (vmcnt(0)), in front of the chunk's barrier.  Ablation (fills removed, profiles/build_diag.sh
// -DW43_ABL): the dword halo fills cost 9.5 % of the kernel's time, the weight fills 3 %, the barrier 2 %.
//
// SPLIT ("K slabs"): a launch whose tile grid covers a fraction of the chip's 512 block slots (the 40 x 53 and 20 x 26 levels at
// small per-GPU batches: 152-600 blocks of 128-256 chunks each) is cut along the input channels instead: block (tile, slab k)
// runs chunks [k nchunks / S, (k+1) nchunks / S) and stores its UN-reduced y = A^T M tile to P.slabs; w43_slab_reduce_kernel adds
// the S slabs in slab order and runs this kernel's epilogue (crop, statistics, fused BatchNorm-backward) on the sums.
template <int WM, int X4M, int NL, bool FAST, bool PLAIN, bool SPLIT = false>
#ifndef W43_ABL   // diagnostic builds: 1 no weight fills, 2 no halo fills, 4 no barrier per chunk, 8 no wait for the fills, 16 / 32 halo fills from hot addresses (results are then garbage)
#define W43_ABL 0
#endif
#ifndef W43_MIN_WAVES   // diagnostic builds: waves per SIMD the register allocation must admit for the 4-wave form
#define W43_MIN_WAVES 2
#endif
__global__ __launch_bounds__(256 * WM + 64 * NL, NL > 0 ? 3 : (WM == 1 ? W43_MIN_WAVES : 1)) void conv3x3_w43_kernel(const W43Params P) {
  static_assert(NL == 0 || WM == 1, "loader waves serve one 64-channel weight image");
  constexpr bool X4 = X4M == 1, U4 = X4M == 2;
  static_assert(!U4 || (FAST && WM == 1 && NL == 0), "unaligned 16-byte halo pieces: the straight-fill 4-wave form");
  static_assert(!SPLIT || (FAST && WM == 1 && NL == 0 && !U4), "K slabs: the straight-fill 4-wave form");
  constexpr int MT = 4, BM = W43_BM, WS = BM, WTILE = W43_WTILE, W4 = W43_W4 * WM, NT = 256 * WM + 64 * NL, NWAVE = 4 * WM;
  constexpr int NWI = (W4 + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int PS = P.PS;
  const int BUF = U4 ? WM * WTILE + P.NI * 256 : WM * WTILE + 4 * PS;   // U4: whole 64-piece instructions

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave8 >> 2, wave = wave8 & 3;   // wave: pixel group of the wave, wm: its 64-channel group
#ifdef GSD_W43_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif
  const int j = lane >> 4, l16 = lane & 15;

  // The m-blocks of one pixel tile read the same halo: hardware deals blocks round-robin over the 8 XCDs, so give every XCD
  // a contiguous range of logical ids (pixel tile major, m-block minor) and its L2 serves the halo once.
  const int lid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int mbb = lid % P.mblocks;              // P.mblocks counts blocks (WM 64-channel groups each)
  const int slab = SPLIT ? (lid / P.mblocks) % P.nslab : 0;   // SPLIT: pixel tile major, then slab, m-block minor
  const int pt = SPLIT ? lid / (P.mblocks * P.nslab) : lid / P.mblocks;
  const int c_lo = SPLIT ? (int)((long long)slab * P.nchunks / P.nslab) : 0;            // this block's channel chunks
  const int c_hi = SPLIT ? (int)((long long)(slab + 1) * P.nchunks / P.nslab) : P.nchunks;
  const int mb = mbb * WM + wm;
  const int m0 = mb * BM;
  const int tpi = P.tiles_y * P.tiles_x;
  const int n = P.fold ? 0 : pt / tpi;            // fold: the tile rows cover the whole batch, the image is a per-lane value
  const int rt = pt - n * tpi;
  const int ty = rt / P.tiles_x;
  const int h0 = ty * P.TH, w0 = (rt - ty * P.tiles_x) * P.TW;   // fold: h0 is a padded flat row (w43_row)

  // ---- this lane's Winograd tile: 4 pixels (tr, 4*tq .. 4*tq+3) of the block's TH x TW output tile ----------------------
  const int q = wave * 16 + l16;
  const bool q_ok = q < P.TH * P.TWq;
  const int tr = q_ok ? q / P.TWq : 0;
  const int tq = q_ok ? q - tr * P.TWq : 0;
  const int baddr = WM * WTILE + j * PS + tr * P.WCp + 4 * tq + (X4 ? 4 : 0);   // halo columns 4*tq .. 4*tq+5 of halo rows tr .. tr+2
  int vmask = 0;                                            // pixels of the tile that exist in the image
  int n_t, h_t;                                             // this lane's image and row
  w43_row(P, n, h0 + tr, n_t, h_t);
  if (q_ok && h_t >= 0 && h_t < P.H) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (w0 + 4 * tq + i < P.W) vmask |= 1 << i;
  }

  // ---- DMA lane geometry (as gsd_conv3x3.hip, window rows padded to WCp floats) ------------------------------------------
  constexpr int NPP = 2 / WM;   // dword form: position chunks of 64 per wave (the block's waves cover the 512 window positions
                                // once); X4: DMA units (channel plane, instruction) per wave and chunk
  int xo0[NPP], xo1[NPP];
  int u_ch[NPP], u_lds[NPP];    // X4: the unit's channel of the chunk and its float offset inside the channel plane
  bool p_on[NPP];
#pragma unroll
  for (int pp = 0; pp < NPP; ++pp) {
    xo0[pp] = xo1[pp] = -2;
    u_ch[pp] = u_lds[pp] = 0;
    if constexpr (X4) {
      const int u = wave8 + NWAVE * pp;            // units 0 .. 4*NI-1: channel u / NI, instruction u % NI
      p_on[pp] = u < 4 * P.NI;
      u_ch[pp] = u / P.NI;
      const int ii = u - u_ch[pp] * P.NI;
      u_lds[pp] = ii * P.RPI * P.WCp + 1;          // + 1 float: image column w0-1 then sits 16-byte aligned
      const int rl = lane / P.NP, pc = lane - rl * P.NP;
      const int rr = ii * P.RPI + rl;
      if (p_on[pp] && rl < P.RPI && rr < P.WR) {
        int nn, gh;
        w43_row(P, n, h0 - 1 + rr, nn, gh);
        const int gw = w0 - 4 + 4 * pc;
        const int fo0 = P.fold ? nn * (int)P.src0.ns : 0, fo1 = P.fold ? nn * (int)P.src1.ns : 0;
        int hs = gh - P.src0.oh, ws = gw - P.src0.ow;
        xo0[pp] = ((unsigned)hs < (unsigned)P.src0.H && ws >= 0 && ws < P.src0.W && ws + 4 <= P.src0.ws) ? fo0 + hs * P.src0.ws + ws : -1;
        hs = gh - P.src1.oh;
        ws = gw - P.src1.ow;
        xo1[pp] = ((unsigned)hs < (unsigned)P.src1.H && ws >= 0 && ws < P.src1.W && ws + 4 <= P.src1.ws) ? fo1 + hs * P.src1.ws + ws : -1;
      }
    } else {
      p_on[pp] = wave8 + NWAVE * pp < P.NPV;
      const int pos = (wave8 + NWAVE * pp) * 64 + lane;
      const int rr = pos / P.WCp, cc = pos - rr * P.WCp;
      if (rr < P.WR && cc < P.WC) {
        int nn, gh;
        w43_row(P, n, h0 - 1 + rr, nn, gh);
        const int gw = w0 - 1 + cc;
        const int fo0 = P.fold ? nn * (int)P.src0.ns : 0, fo1 = P.fold ? nn * (int)P.src1.ns : 0;
        int hs = gh - P.src0.oh, ws = gw - P.src0.ow;
        xo0[pp] = ((unsigned)hs < (unsigned)P.src0.H && (unsigned)ws < (unsigned)P.src0.W) ? fo0 + hs * P.src0.ws + ws : -1;
        hs = gh - P.src1.oh;
        ws = gw - P.src1.ow;
        xo1[pp] = ((unsigned)hs < (unsigned)P.src1.H && (unsigned)ws < (unsigned)P.src1.W) ? fo1 + hs * P.src1.ws + ws : -1;
      }
    }
  }
  // U4: instruction i = wave + 4 k of a chunk moves pieces 64 i .. 64 i + 63 of the chunk's [4 planes][PS / 4 pieces] image.
  // Per source segment: the lane's float offset from the chunk's first channel plane (W43_SENT: sentinel piece) and the floats of
  // the piece that lie outside the image (bits 0..3: patched with the padding value after landing).
  constexpr int KH = 2, W43_SENT = -2147483647 - 1;
  int h_off0[KH], h_off1[KH], h_pm0[KH], h_pm1[KH];
  bool u_patch = false;   // block constant: some window row has a piece that straddles the left or right edge of a segment
  if constexpr (U4) {
    const int o0 = w0 - 1 - P.src0.ow, o1 = w0 - 1 - P.src1.ow;
    u_patch = (o0 < 0 && (-o0 & 3)) || (P.src0.W > o0 && P.src0.W < o0 + P.WCp && ((P.src0.W - o0) & 3));
    if (P.src1.C > 0) u_patch = u_patch || (o1 < 0 && (-o1 & 3)) || (P.src1.W > o1 && P.src1.W < o1 + P.WCp && ((P.src1.W - o1) & 3));
    const int NPr = P.WCp >> 2, NPc = PS >> 2;
#pragma unroll
    for (int k = 0; k < KH; ++k) {
      const int i = wave8 + 4 * k;
      const int pid = 64 * i + lane;
      const int ch = pid / NPc, pq = pid - ch * NPc;
      const int row = pq / NPr, pc = pq - row * NPr;
      h_off0[k] = h_off1[k] = W43_SENT;   // (an offset of -1 is a real one: the piece in front of the tensor's first row)
      h_pm0[k] = h_pm1[k] = 0;
      if (i < P.NI && ch < 4 && row < P.WR) {
        int nn, gh;
        w43_row(P, n, h0 - 1 + row, nn, gh);
        const int gw = w0 - 1 + 4 * pc;
        const int fo0 = P.fold ? nn * (int)P.src0.ns : 0, fo1 = P.fold ? nn * (int)P.src1.ns : 0;
        int hs = gh - P.src0.oh, c0 = gw - P.src0.ow;
        if ((unsigned)hs < (unsigned)P.src0.H && c0 + 3 >= 0 && c0 < P.src0.W) {
          h_off0[k] = ch * (int)P.src0.cs + fo0 + hs * P.src0.ws + c0;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c0 + e < 0 || c0 + e >= P.src0.W) h_pm0[k] |= 1 << e;
        }
        hs = gh - P.src1.oh;
        c0 = gw - P.src1.ow;
        if (P.src1.C > 0 && (unsigned)hs < (unsigned)P.src1.H && c0 + 3 >= 0 && c0 < P.src1.W) {
          h_off1[k] = ch * (int)P.src1.cs + fo1 + hs * P.src1.ws + c0;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c0 + e < 0 || c0 + e >= P.src1.W) h_pm1[k] |= 1 << e;
        }
      }
    }
  }
  const float* u_base = P.src0.p + (long long)n * P.src0.ns;   // U4: first channel plane of the next chunk to fill
  const float* u_sent = P.src0.relu ? &gsd_pad16_w43[4] : &gsd_pad16_w43[0];
  long long u_cs = P.src0.cs;
  const float* wsrc0 = P.wt + (size_t)(mbb * WM) * P.nchunks * WTILE;   // the block's WM weight images follow each other
  const long long wlane = tid * 4;   // this lane's float offset inside a 1 KiB weight piece group
  long long xl0[NPP];   // first segment's offsets as 64-bit lane values (the address add is then a single instruction)
#pragma unroll
  for (int pp = 0; pp < NPP; ++pp) xl0[pp] = xo0[pp];
  // SPLIT: a slab that lies wholly in the second (concat) segment starts there -- its padding positions, plane pointer and lane offsets
  const bool s1_start = SPLIT && P.src1.C > 0 && c_lo * 4 > P.src0.C;
  if constexpr (NL == 0 && !U4)
  {
    // padding positions of the first segment, once, in all 2 x 4 channel planes (own positions only: the lanes that
    // would otherwise DMA the sentinel there on every fill); visible to the consumers after the first barrier
    const float pad0 = (s1_start ? P.src1.relu : P.src0.relu) ? __builtin_nanf("") : 0.f;
#pragma unroll
    for (int pp = 0; pp < NPP; ++pp)
      if (p_on[pp] && (s1_start ? xo1[pp] : xo0[pp]) == -1) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          if constexpr (X4) {   // the unit's own plane: the other planes' pieces belong to other units
#pragma unroll
            for (int e = 0; e < 4; ++e) smem[b * BUF + WM * WTILE + u_ch[pp] * PS + u_lds[pp] + lane * 4 + e] = pad0;
          } else {
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) smem[b * BUF + WM * WTILE + ch * PS + (wave8 + NWAVE * pp) * 64 + lane] = pad0;
          }
        }
      }
  }
  int d_seg = 0, d_left = P.src0.C;
  const float* d_base = P.src0.p + (long long)n * P.src0.ns;
  long long d_cs = P.src0.cs;
  if constexpr (SPLIT) {   // first channel plane of the slab
    if (s1_start) {
      d_base = P.src1.p + (long long)n * P.src1.ns + (long long)(c_lo * 4 - P.src0.C) * P.src1.cs;
      d_cs = P.src1.cs;
    } else {
      d_base += (long long)(c_lo * 4) * P.src0.cs;
    }
  }
  const float* d_sent = X4 ? (P.src0.relu ? &gsd_pad16_w43[4] : &gsd_pad16_w43[0]) : (P.src0.relu ? &gsd_pad_w43[1] : &gsd_pad_w43[0]);
  int d_xo[NPP];
#pragma unroll
  for (int pp = 0; pp < NPP; ++pp) d_xo[pp] = xo0[pp];

  // FAST ("straight fill"): no per-slot bookkeeping at all.  Every chunk lies inside one source segment (src0.C % 4 == 0,
  // Cin % 4 == 0), so a halo slot is: the lanes that have a pixel move it, the plane pointer advances by one channel.  The
  // switch to the second (concat) segment happens once per block, between two chunks: new plane pointer and lane offsets,
  // and the padding positions of that segment are written into each LDS image the first time it is filled from it.
  const int f_sw = (P.src1.C > 0 && !s1_start) ? P.src0.C / 4 : -1;   // first chunk of the second segment
  int f_xo[NPP];
  long long f_xl[NPP];
#pragma unroll
  for (int pp = 0; pp < NPP; ++pp) {
    f_xo[pp] = s1_start ? xo1[pp] : xo0[pp];
    f_xl[pp] = f_xo[pp];
  }
  auto begin_fill = [&](int chunk, int buf) {
    if constexpr (U4) {
      if (chunk == f_sw) {   // the second (concat) segment from here on
        u_base = P.src1.p + (long long)n * P.src1.ns;
        u_sent = P.src1.relu ? &gsd_pad16_w43[4] : &gsd_pad16_w43[0];
        u_cs = P.src1.cs;
#pragma unroll
        for (int k = 0; k < KH; ++k) h_off0[k] = h_off1[k];
      }
      return;
    }
    if (f_sw < 0 || (chunk != f_sw && chunk != f_sw + 1)) return;
    if (chunk == f_sw) {
      d_base = P.src1.p + (long long)n * P.src1.ns;
      d_cs = P.src1.cs;
#pragma unroll
      for (int pp = 0; pp < NPP; ++pp) {
        f_xo[pp] = xo1[pp];
        f_xl[pp] = xo1[pp];
      }
    }
    const float pad1 = P.src1.relu ? __builtin_nanf("") : 0.f;
#pragma unroll
    for (int pp = 0; pp < NPP; ++pp)
      if (p_on[pp] && f_xo[pp] == -1) {
        if constexpr (X4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) smem[buf * BUF + WM * WTILE + u_ch[pp] * PS + u_lds[pp] + lane * 4 + e] = pad1;
        } else {
#pragma unroll
          for (int ch = 0; ch < 4; ++ch) smem[buf * BUF + WM * WTILE + ch * PS + (wave8 + NWAVE * pp) * 64 + lane] = pad1;
        }
      }
  };
  auto fast_halo = [&](int ch, float* Xb) {
    if constexpr (U4) {
      // the wave's (at most) two instructions of the chunk ride in halo slots 0 and 2; slot 3 moves on to the next chunk
      if (ch == 0 || ch == 2) {
        const int k = ch >> 1;
        if (wave8 + 4 * k < P.NI) {
          const float* gp = h_off0[k] != W43_SENT ? u_base + h_off0[k] : u_sent;
          float* dstp = Xb + (wave8 + 4 * k) * 256;
          __builtin_amdgcn_global_load_lds(gp, dstp, 16, 0, 0);
        }
      }
      if (ch == 3) u_base += 4 * u_cs;
      return;
    }
#pragma unroll
    for (int pp = 0; pp < NPP; ++pp) {
      if constexpr (X4) {
        if (p_on[pp] && u_ch[pp] == ch && f_xo[pp] >= 0) {
          const float* gp = d_base + f_xl[pp];
          float* dstp = Xb + ch * PS + u_lds[pp];
          __builtin_amdgcn_global_load_lds(gp, dstp, 16, 0, 0);
        }
      } else {
        if (p_on[pp] && f_xo[pp] >= 0) {
#if (W43_ABL) & 16   // diagnostic: every halo lane reads chunk 0's address (same instructions, bytes from the L1 / L2 instead of HBM)
          const float* gp = P.src0.p + f_xl[pp];
#elif (W43_ABL) & 32   // diagnostic: every halo lane reads ONE address
          const float* gp = P.src0.p;
#else
          const float* gp = d_base + f_xl[pp];
#endif
          float* dstp = Xb + ch * PS + (wave8 + NWAVE * pp) * 64;
          __builtin_amdgcn_global_load_lds(gp, dstp, 4, 0, 0);
        }
      }
    }
    d_base += d_cs;
  };

  // slots 0..NWI-1: the weight chunk (16 B per lane); slot NWI+ch: input channel ch of the chunk
  auto dma_slot = [&](int slot, int chunk, int buf) {
    float* Wb = smem + buf * BUF;
    if constexpr (FAST) {
      if (slot >= NWI && slot < NWI + 4) {
        fast_halo(slot - NWI, Wb + WM * WTILE);
        return;
      }
    }
    if (slot < NWI) {
      const int e = tid + slot * NT;                 // 16-byte piece of the block's WM weight images (LDS: linear in e)
      if constexpr (WM == 1) {
        // one image: scalar chunk base + the lane's fixed offset.  The last slot is half full (1152 = 4.5 x 256 pieces):
        // it goes to waves 2 and 3, because waves 0 and 1 already move two position chunks of every halo plane where
        // waves 2 and 3 move one -- the barrier waits for the busiest wave.
        if (slot < NWI - 1) {
          __builtin_amdgcn_global_load_lds(wsrc0 + (size_t)chunk * WTILE + slot * (NT * 4) + wlane, Wb + (slot * NT + wave8 * 64) * 4, 16, 0, 0);
        } else if (wave8 >= 2) {
          __builtin_amdgcn_global_load_lds(wsrc0 + (size_t)chunk * WTILE + (slot * NT - 128) * 4 + wlane, Wb + (slot * NT - 128 + wave8 * 64) * 4, 16, 0, 0);
        }
      } else {
        const int img = e >= W43_W4 ? 1 : 0;
        const float* wsrc = wsrc0 + ((size_t)img * P.nchunks + chunk) * WTILE + (e - img * W43_W4) * 4;
        if (e < W4) __builtin_amdgcn_global_load_lds(wsrc, Wb + (slot * NT + wave8 * 64) * 4, 16, 0, 0);
      }
    } else if (slot < NWI + 4) {
      const int ch = slot - NWI;
      float* Xb = Wb + WM * WTILE;
      if (d_left == 0 && d_seg == 0) {
        d_seg = 1;
        d_left = P.src1.C;
        d_base = P.src1.p + (long long)n * P.src1.ns;
        d_cs = P.src1.cs;
        d_sent = X4 ? (P.src1.relu ? &gsd_pad16_w43[4] : &gsd_pad16_w43[0]) : (P.src1.relu ? &gsd_pad_w43[1] : &gsd_pad_w43[0]);
#pragma unroll
        for (int pp = 0; pp < NPP; ++pp) d_xo[pp] = xo1[pp];
      }
      const bool c_ok = d_left > 0;
      if constexpr (X4) {
        // 16-byte pieces: the unit (channel plane ch, instruction ii) of the wave that owns it; the other waves only keep the
        // channel bookkeeping below in step
#pragma unroll
        for (int pp = 0; pp < NPP; ++pp) {
          if (p_on[pp] && u_ch[pp] == ch) {
            float* dstp = Xb + ch * PS + u_lds[pp];
            if (c_ok && d_seg == 0) {
              if (xo0[pp] >= 0) {
                const float* gp = d_base + xl0[pp];
                __builtin_amdgcn_global_load_lds(gp, dstp, 16, 0, 0);
              }   // padding pieces: written once, below
            } else if (d_xo[pp] != -2) {
              const float* g = (c_ok && d_xo[pp] >= 0) ? d_base + d_xo[pp] : (c_ok ? d_sent : &gsd_pad16_w43[0]);
              __builtin_amdgcn_global_load_lds(g, dstp, 16, 0, 0);
            }
          }
        }
      } else if (c_ok && d_seg == 0) {
        // fast path, first segment: the padding positions of every plane were written once at kernel start (below), so a
        // fill only moves the lanes that have a pixel -- no select, no sentinel pointer, one v_lshl_add_u64 per instruction
#pragma unroll
        for (int pp = 0; pp < NPP; ++pp)
          if (p_on[pp] && xo0[pp] >= 0)
            __builtin_amdgcn_global_load_lds(d_base + xl0[pp], Xb + ch * PS + (wave8 + NWAVE * pp) * 64, 4, 0, 0);
      } else {
        // second (concat) segment and K padding: every window position is written, padding from the sentinel
        const float* sentinel = c_ok ? d_sent : &gsd_pad_w43[0];
#pragma unroll
        for (int pp = 0; pp < NPP; ++pp) {
          if (p_on[pp] && d_xo[pp] != -2) {
            const float* g = (c_ok && d_xo[pp] >= 0) ? d_base + d_xo[pp] : sentinel;
            __builtin_amdgcn_global_load_lds(g, Xb + ch * PS + (wave8 + NWAVE * pp) * 64, 4, 0, 0);
          }
        }
      }
      if (c_ok) {
        d_base += d_cs;
        --d_left;
      }
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
  constexpr int BMB = WM * BM;    // output channels of the block
  float* sBw = sAff + 2 * Kpad;   // [4][BMB]: scale, shift, mean, invstd of the fused BatchNorm-backward epilogue
  if (P.bw_raw != nullptr) {
    for (int c = tid; c < BMB; c += NT) {
      const int co = mbb * BMB + c < P.Cout ? mbb * BMB + c : 0;
      sBw[c] = P.bw_scale[co];
      sBw[BMB + c] = P.bw_shift[co];
      sBw[2 * BMB + c] = P.bw_mean[co];
      sBw[3 * BMB + c] = P.bw_invstd[co];
    }
  }
  if constexpr (NL > 0) {
    if (wave8 >= NWAVE) {   // wave-uniform: the loader waves never reach the MFMA loop or the epilogue
      w43_loader<X4, NL>(P, smem, BUF, n, h0, w0, mbb, lane, wave8 - NWAVE);
      return;
    }
  }
  const float lo0 = P.src0.relu ? 0.f : -__builtin_inff(), lo1 = P.src1.relu ? 0.f : -__builtin_inff();

  f32x4 acc[MT][6];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int f = 0; f < 6; ++f) acc[m][f] = f32x4{0.f, 0.f, 0.f, 0.f};

  // V = B^T d of one kernel row: raw halo values -> deferred BatchNorm+ReLU -> the 6 frequency operands
  auto transform = [&](const f32x4& ra, const f32x2& rb, float sc, float sh, float lo, float (&v)[6]) {
    float d0 = ra[0], d1 = ra[1], d2 = ra[2], d3 = ra[3], d4 = rb[0], d5 = rb[1];
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

  const int a_lane = wm * WTILE + l16 * 4;
  if constexpr (NL == 0) {
    if constexpr (FAST) begin_fill(c_lo, 0);
#pragma unroll
    for (int slot = 0; slot < NWI + 4; ++slot) dma_slot(slot, c_lo, 0);
  }
  W43_STAMP(5)   // prologue
  for (int chunk = c_lo; chunk < c_hi; ++chunk) {
    const int cur = (chunk - c_lo) & 1;
    if constexpr (U4) {
      __builtin_amdgcn_s_waitcnt(0x0F70);   // this wave's fills of the chunk have landed
      if (u_patch) {   // a block at the left / right image edge: the outside floats of the straddling pieces this lane moved
        const bool seg1 = f_sw >= 0 && chunk >= f_sw;
        const float padv = (seg1 ? P.src1.relu : P.src0.relu) ? __builtin_nanf("") : 0.f;
        float* Xh = smem + cur * BUF + WM * WTILE;
#pragma unroll
        for (int k = 0; k < KH; ++k) {
          const int pm = seg1 ? h_pm1[k] : h_pm0[k];
          if (pm) {
            float* pp = Xh + (wave8 + 4 * k) * 256 + lane * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (pm >> e & 1) pp[e] = padv;
          }
        }
      }
      if (!((W43_ABL) & 4)) __syncthreads();
    } else if ((W43_ABL) & 4) {
      __builtin_amdgcn_s_waitcnt(0x0F70);
    } else if ((W43_ABL) & 8) {   // barrier without waiting for the fills: what their latency costs
      __syncthreads();
    } else {
      gsd_dma_barrier();
    }
    W43_STAMP(0)   // wait for the chunk's DMA + barrier
    const int kc = chunk * 4 + j;
    float sc = 1.f, sh = 0.f, lo = 0.f;
    if constexpr (!PLAIN) {
      sc = sAff[kc], sh = sAff[Kpad + kc];
      lo = kc < P.src0.C ? lo0 : (kc < P.Cin ? lo1 : -__builtin_inff());
    }
    const bool more = chunk + 1 < c_hi;
    const float* Wc = smem + cur * BUF;
#ifndef W43_PF   // k-steps the weight operand is read ahead of its MFMAs
#define W43_PF 1
#endif
    f32x4 av[W43_PF + 1];
    f32x4 ra[2];
    f32x2 rb[2];
    float v[6];
    ra[0] = *reinterpret_cast<const f32x4*>(&Wc[baddr]);
    rb[0] = *reinterpret_cast<const f32x2*>(&Wc[baddr + 4]);
#pragma unroll
    for (int s = 0; s < W43_PF; ++s) av[s] = *reinterpret_cast<const f32x4*>(&Wc[(j * 18 + s) * WS + a_lane]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      transform(ra[r & 1], rb[r & 1], sc, sh, lo, v);
      if (r < 2) {   // the next kernel row's raw values fly during this row's 24 MFMAs
        ra[(r + 1) & 1] = *reinterpret_cast<const f32x4*>(&Wc[baddr + (r + 1) * P.WCp]);
        rb[(r + 1) & 1] = *reinterpret_cast<const f32x2*>(&Wc[baddr + (r + 1) * P.WCp + 4]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (r == 0) W43_STAMP(1)   // first operand reads + first transform: no MFMA of this wave yet
#pragma unroll
      for (int f = 0; f < 6; ++f) {
        const int s = r * 6 + f, cs = s % (W43_PF + 1);
        if (s + W43_PF < 18)
          av[(s + W43_PF) % (W43_PF + 1)] = *reinterpret_cast<const f32x4*>(&Wc[(j * 18 + s + W43_PF) * WS + a_lane]);
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m][f] = mfma16(av[cs][m], v[f], acc[m][f]);
        if constexpr (NL == 0) {
          if constexpr (FAST && WM == 1) {
            // weights first, in ONE k-step: a wave's four pieces are adjacent (4 KiB), so they share one LDS base (M0) and differ
            // in the instruction's immediate offset, which moves the global and the LDS address alike -- a changed M0 between
            // two LDS-DMA instructions costs the wave ~45 cycles at two blocks per CU (profiles/ubench/dma_m0.hip); then the halo
            if (more && s < 3) {
              if (s == 0) {
                begin_fill(chunk + 1, cur ^ 1);
                if (!((W43_ABL) & 1)) {
                float* Wn = smem + (cur ^ 1) * BUF;
                const float* wg = wsrc0 + (size_t)(chunk + 1) * WTILE + wave8 * 1024 + lane * 4;
                float* wl = Wn + wave8 * 1024;
                __builtin_amdgcn_global_load_lds(wg, wl, 16, 0, 0);
                __builtin_amdgcn_global_load_lds(wg, wl, 16, 1024, 0);
                __builtin_amdgcn_global_load_lds(wg, wl, 16, 2048, 0);
                __builtin_amdgcn_global_load_lds(wg, wl, 16, 3072, 0);
                if (wave8 >= 2) {   // pieces 16, 17 of the 18
                  const float* wg2 = wsrc0 + (size_t)(chunk + 1) * WTILE + (14 + wave8) * 256 + lane * 4;
                  float* wl2 = Wn + (14 + wave8) * 256;
                  __builtin_amdgcn_global_load_lds(wg2, wl2, 16, 0, 0);
                }
                }
              } else if (!((W43_ABL) & 2)) {
                dma_slot(NWI + 2 * s - 2, chunk + 1, cur ^ 1);
                dma_slot(NWI + 2 * s - 1, chunk + 1, cur ^ 1);
              }
            }
          } else if constexpr (FAST) {
            if (more && s < 5) {
              if (s == 0) begin_fill(chunk + 1, cur ^ 1);
              dma_slot(2 * s, chunk + 1, cur ^ 1);
              dma_slot(2 * s + 1, chunk + 1, cur ^ 1);
            }
          } else if (more && s < 5) {
#ifdef W43_STAMP_DMA
            W43_STAMP(2)
#endif
            dma_slot(2 * s, chunk + 1, cur ^ 1);
            dma_slot(2 * s + 1, chunk + 1, cur ^ 1);
#ifdef W43_STAMP_DMA
            W43_STAMP(6)   // the DMA slots alone
#endif
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s == ((FAST && WM == 1) ? 2 : 4)) W43_STAMP(2)   // the k-steps that carry the next chunk's DMA issue
      }
    }
    W43_STAMP(3)     // the other thirteen k-steps (52 MFMAs)
  }

  // ---- epilogue: y = A^T M, NCHW stores (two destination segments with crop), BatchNorm partial sums ----------------------
  auto out_transform = [&](int m, int reg, float (&y)[4]) {
    const float M0 = acc[m][0][reg], M1 = acc[m][1][reg], M2 = acc[m][2][reg];
    const float M3 = acc[m][3][reg], M4 = acc[m][4][reg], M5 = acc[m][5][reg];
    const float p12 = M1 + M2, m12 = M1 - M2, p34 = M3 + M4, m34 = M3 - M4;
    y[0] = M0 + p12 + p34;
    y[1] = fmaf(2.f, m34, m12);
    y[2] = fmaf(4.f, p34, p12);
    y[3] = fmaf(8.f, m34, m12) + M5;
  };

  if constexpr (SPLIT) {
    // K slab: the un-reduced tile, every lane its 16 B (padding tiles and channels included; the reducer masks)
    float* const tile = P.slabs + (((size_t)slab * (gridDim.x / P.nslab) + (size_t)pt * P.mblocks + mbb) * BM) * 256 + (wave * 16 + l16) * 4;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        float y[4];
        out_transform(m, reg, y);
        *reinterpret_cast<f32x4*>(tile + (m * 16 + j * 4 + reg) * 256) = f32x4{y[0], y[1], y[2], y[3]};
      }
  } else {
    w43_epilogue<WM>(P, n, pt, wave, wm, mbb, m0, j, l16, n_t, h_t, w0, tq, vmask, sBw, out_transform);
  }
#ifdef GSD_W43_STAMPS
  W43_STAMP(4)   // epilogue
  if (P.stamps != nullptr && lane == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) P.stamps[((size_t)blockIdx.x * NWAVE + wave8) * 8 + i] = st_acc[i];
  }
#endif
}

// K-slab reducer: block (pixel tile, m-block) adds the S un-reduced 64 x 256 tiles the SPLIT blocks left in P.slabs, in slab order
// (run-to-run bitwise), and runs the convolution's epilogue on the sums -- same lane -> (channel, pixel) map, same destinations,
// same partial rows, same fused BatchNorm-backward form.  HBM-bound: (S + 1) x 64 KiB per block.
__global__ __launch_bounds__(256) void w43_slab_reduce_kernel(const W43Params P) {
  __shared__ float sBw[4 * W43_BM];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane >> 4, l16 = lane & 15;
  const int lid = blockIdx.x;
  const int mbb = lid % P.mblocks, pt = lid / P.mblocks;
  const int tpi = P.tiles_y * P.tiles_x;
  const int n = P.fold ? 0 : pt / tpi;
  const int rt = pt - n * tpi;
  const int ty = rt / P.tiles_x;
  const int h0 = ty * P.TH, w0 = (rt - ty * P.tiles_x) * P.TW;
  const int q = wave * 16 + l16;
  const bool q_ok = q < P.TH * P.TWq;
  const int tr = q_ok ? q / P.TWq : 0;
  const int tq = q_ok ? q - tr * P.TWq : 0;
  int vmask = 0, n_t, h_t;
  w43_row(P, n, h0 + tr, n_t, h_t);
  if (q_ok && h_t >= 0 && h_t < P.H) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (w0 + 4 * tq + i < P.W) vmask |= 1 << i;
  }
  if (P.bw_raw != nullptr) {
    if (tid < W43_BM) {
      const int co = mbb * W43_BM + tid < P.Cout ? mbb * W43_BM + tid : 0;
      sBw[tid] = P.bw_scale[co];
      sBw[W43_BM + tid] = P.bw_shift[co];
      sBw[2 * W43_BM + tid] = P.bw_mean[co];
      sBw[3 * W43_BM + tid] = P.bw_invstd[co];
    }
    __syncthreads();
  }
  const size_t sstride = (size_t)gridDim.x * W43_BM * 256;
  const float* const tile = P.slabs + (size_t)lid * W43_BM * 256 + q * 4;
  f32x4 ys[4][4];   // all 16 channel rows of slab 0 in flight, then one slab after the other on top
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) ys[m][reg] = *reinterpret_cast<const f32x4*>(tile + (m * 16 + j * 4 + reg) * 256);
  for (int k = 1; k < P.nslab; ++k) {
    const float* const tk = tile + (size_t)k * sstride;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) ys[m][reg] += *reinterpret_cast<const f32x4*>(tk + (m * 16 + j * 4 + reg) * 256);
  }
  auto get_y = [&](int m, int reg, float (&y)[4]) {
    y[0] = ys[m][reg][0], y[1] = ys[m][reg][1], y[2] = ys[m][reg][2], y[3] = ys[m][reg][3];
  };
  w43_epilogue<1>(P, n, pt, wave, 0, mbb, mbb * W43_BM, j, l16, n_t, h_t, w0, tq, vmask, sBw, get_y);
}

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
namespace {

struct W43Plan {
  int TH, TW, TWq, tiles_y, tiles_x, mblocks, WR, WC, WCp, fold;
};

int halo_read_cycles(int TWq, int LP, int PS, int off);

// TH x TW output tile of up to 64 Winograd tiles whose padded halo window fits the 512 DMA positions.  Every block does the
// work of 64 tiles whatever it covers, so: fewest blocks; among equals 32-wide rows, then the widest.
// Small images (levels 3-4 of the U-Net: 40 x 53, 20 x 26) fill such a tile badly (83 % / 68 %): for them the tile rows run
// over the padded flat rows of the whole batch (`fold`, see w43_row) and the tile may be 7 or 14 Winograd tiles wide
// (28 / 56 pixels: 53 -> 2 x 28, 26 -> 1 x 28), with the LDS row pitch chosen for the fewest bank conflicts of the halo reads.
bool plan_w43(int N, int H, int W, int M, W43Plan* best) {
  long best_cost = -1;
  const int force_tw = gsd_env_int("GSD_W43_TW", 0);   // tuning
  const int fold_mode = gsd_env_int("GSD_W43_FOLD", -1);   // tuning: 0 never, 1 whenever possible, -1 when it saves > 4 % of the blocks
  best->fold = 0;
  long plain_blocks = -1;
  for (int fold = 0; fold <= 1; ++fold) {
    if (fold && (fold_mode == 0 || N <= 1 || (long)H * W > 8192)) continue;
    static const int tws[9] = {32, 64, 16, 8, 4, 28, 56, 24, 48};
    for (int k = 0; k < (fold ? 9 : 5); ++k) {   // (7- and 14-tile rows without folding, folding at 80 x 106: measured, no gain)
      const int tw = tws[k];
      if (force_tw && tw != force_tw) continue;
      const int twq = tw / 4;
      int th = 64 / twq;
      int wcp = round_up(tw + 2, 4);
      while (th > 1 && (th + 2) * wcp > 512) --th;
      if ((th + 2) * wcp > 512) continue;
      const int rows = fold ? N * (H + 1) : H;
      if (th > rows) th = rows;
      const int ty = ceil_div(rows, th);
      if (!fold) th = ceil_div(H, ty);
      const long blocks = (long)ty * ceil_div(W, tw) * (fold ? 1 : N);
      if (!fold && (plain_blocks < 0 || blocks < plain_blocks)) plain_blocks = blocks;
      if (fold && fold_mode < 0 && blocks * 104 > plain_blocks * 100) continue;
      if ((twq & (twq - 1)) != 0 || fold) {   // odd widths: pick the LDS row pitch with the fewest bank conflicts
        int bc = -1;
        for (int c = round_up(tw + 2, 4); c <= round_up(tw + 2, 4) + 12 && (th + 2) * c <= 512; c += 4) {
          const int cyc = halo_read_cycles(twq, c, round_up((th + 2) * c, 4) + 4, 0);
          if (bc < 0 || cyc < bc) {
            bc = cyc;
            wcp = c;
          }
        }
      }
      const long cost = blocks * 8 + (tw == 32 ? 0 : tw == 64 ? 1 : tw == 16 ? 2 : 3);
      if (best_cost < 0 || cost < best_cost) {
        best_cost = cost;
        best->fold = fold;
        best->TH = th; best->TW = tw; best->TWq = twq;
        best->tiles_y = ty; best->tiles_x = ceil_div(W, tw);
        best->WR = th + 2; best->WC = tw + 2; best->WCp = wcp;
      }
    }
  }
  best->mblocks = ceil_div(M, W43_BM);
  return best_cost >= 0;
}

// LDS bank cost of the consumers' halo reads (one ds_read_b128 + one ds_read_b64 per kernel row; lane -> tile as in the
// kernel) for a row pitch LP and plane stride PS: sum over the four pixel groups of the LDS cycles per read pair.
int halo_read_cycles(int TWq, int LP, int PS, int off) {
  static const int g128[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                  {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
  int total = 0;
  for (int wave = 0; wave < 4; ++wave) {
    int addr[64];
    for (int lane = 0; lane < 64; ++lane) {
      const int q = wave * 16 + (lane & 15);
      addr[lane] = (lane >> 4) * PS + (q / TWq) * LP + 4 * (q % TWq) + off;
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
      int worst = 0;                  // ds_read_b64 at +4 floats: 32-lane halves, 32 slots of 8 B
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

// one launcher per kernel instantiation (the address of the kernel keys the per-device launch-attribute cache)
template <int WM, int X4, int NL, bool FAST, bool PLAIN = false, bool SPLIT = false>
int launch_one(const W43Params& P, int grid, size_t lds, hipStream_t st) {
  static gsd_attr_once big_lds;   // per-device cache of an idempotent launch attribute (gsd_common.h)
  const void* fn = reinterpret_cast<const void*>(&conv3x3_w43_kernel<WM, X4, NL, FAST, PLAIN, SPLIT>);
  if (hipError_t e = gsd_allow_big_lds(big_lds, fn); e != hipSuccess) {
    gsd_set_error("gsd_conv3x3_w43: hipFuncSetAttribute: %s", hipGetErrorString(e));
    return GSD_ERR_HIP;
  }
  GSD_REQUIRE(lds <= 160 * 1024, GSD_ERR_UNSUPPORTED, "gsd_conv3x3_w43: LDS image %zu B too large", lds);
  hipLaunchKernelGGL((conv3x3_w43_kernel<WM, X4, NL, FAST, PLAIN, SPLIT>), dim3(grid), dim3(256 * WM + 64 * NL), lds, st, P);
  GSD_LAUNCH_CHECK("gsd_conv3x3_w43");
  return GSD_OK;
}

// K-slab form: the SPLIT blocks (grid = tile blocks x slabs), then the reducer over the tile blocks
int launch_split(const W43Params& P, int base_grid, size_t lds, hipStream_t st, bool x4, bool plain) {
  const int grid = base_grid * P.nslab;
  int e = x4 ? (plain ? launch_one<1, 1, 0, true, true, true>(P, grid, lds, st) : launch_one<1, 1, 0, true, false, true>(P, grid, lds, st))
             : (plain ? launch_one<1, 0, 0, true, true, true>(P, grid, lds, st) : launch_one<1, 0, 0, true, false, true>(P, grid, lds, st));
  if (e != GSD_OK) return e;
  hipLaunchKernelGGL(w43_slab_reduce_kernel, dim3(base_grid), dim3(256), 0, st, P);
  GSD_LAUNCH_CHECK("gsd_conv3x3_w43 (slab reduce)");
  return GSD_OK;
}

int launch_w43(const W43Params& P, int grid, size_t lds, hipStream_t st, int wm, bool x4, int nl, bool fast, bool plain, bool u4) {
  if (u4) return plain ? launch_one<1, 2, 0, true, true>(P, grid, lds, st) : launch_one<1, 2, 0, true, false>(P, grid, lds, st);
  if (nl == 1) return x4 ? launch_one<1, true, 1, false>(P, grid, lds, st) : launch_one<1, false, 1, false>(P, grid, lds, st);
  if (nl == 2) return x4 ? launch_one<1, true, 2, false>(P, grid, lds, st) : launch_one<1, false, 2, false>(P, grid, lds, st);
  if (wm == 2) return x4 ? launch_one<2, true, 0, false>(P, grid, lds, st) : launch_one<2, false, 0, false>(P, grid, lds, st);
  if (fast && plain) return x4 ? launch_one<1, true, 0, true, true>(P, grid, lds, st) : launch_one<1, false, 0, true, true>(P, grid, lds, st);
  if (fast) return x4 ? launch_one<1, true, 0, true>(P, grid, lds, st) : launch_one<1, false, 0, true>(P, grid, lds, st);
  return x4 ? launch_one<1, true, 0, false>(P, grid, lds, st) : launch_one<1, false, 0, false>(P, grid, lds, st);
}

// K slabs of a launch of `base` tile blocks of `nchunks` 4-channel chunks (1: the launch runs as it is).  Two blocks are resident
// per CU and the dispatcher deals blocks over the 256 CUs, so a CU ends up with k = ceil(blocks / 256) of them and the launch takes
// as long as that CU: pairs of blocks at the shared rate and, for an odd k, one block that has the CU to itself and runs 1.8x
// faster.  The 20 x 26 level at batch 8 is 304 (152) blocks of 128-256 chunks: k = 2 (1) where 1.19 (0.59) would do.  Cutting the
// chunks into S slabs multiplies the blocks and divides their length; it costs the fixed part of a block S times over and the
// reducer's launch and pass over (S + 1) x 64 KiB per tile block.  Constants fitted to profiles/r05_kslabs_b{8,16,32}.txt (117
// timings of 22 launch shapes, S = 1..8: rms error 4 %; the fitted model picks the fastest measured S on 20 of the 22 shapes
// and is within 0.5 % on the other two): 2.6 us per chunk of a block that shares its CU, 4 us per block, a lone block at 0.55 of a
// pair's time, the reducer at 12 us + 6 TB/s.  GSD_W43_SPLIT: 0 / 1 never, S >= 2 that many wherever the shape admits it.
double w43_time_us(long base, int nchunks, int S, bool bw) {
  const long cus = gsd_cu_count();
  const long k = (base * S + cus - 1) / cus;
  const double cu = (double)(k / 2) + (k & 1 ? 0.55 : 0.0);
  double t = cu * (2.6 * nchunks / S + 4.0);
  if (S > 1) t += 12.0 + (double)(S + 1 + (bw ? 1 : 0)) * base * 65536.0 / 6.0e6;
  return t;
}

int w43_pick_slabs(long base, int nchunks, bool bw) {
  const int forced = gsd_env_int("GSD_W43_SPLIT", -1);
  if (forced == 0 || forced == 1) return 1;
  auto t_us = [&](int S) { return w43_time_us(base, nchunks, S, bw); };
  int best = 1;
  double tb = t_us(1) * (forced > 1 ? 1e9 : 0.97);   // a split has to buy 3 %
  for (int S = 2; S <= 8 && nchunks / S >= 8; ++S) {
    if (forced > 1 && S != forced) continue;
    const double t = t_us(S);
    if (t < tb) {
      tb = t;
      best = S;
    }
  }
  return best;
}

}  // namespace

// Floats of K-slab scratch gsd_conv3x3_w43_ws wants for this shape (0: it runs unsplit).
extern "C" int64_t gsd_conv3x3_w43_workspace(int N, int H, int W, int Cin, int Cout) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % 4 != 0) return 0;
  W43Plan p;
  if (!plan_w43(N, H, W, Cout, &p)) return 0;
  const long base = (long)(p.fold ? 1 : N) * p.tiles_y * p.tiles_x * p.mblocks;
  // (the launch picks its own S with bw as it is called, and never more than fits: size for the larger of the two)
  const int Sa = w43_pick_slabs(base, Cin / 4, true), Sb = w43_pick_slabs(base, Cin / 4, false);
  const int S = Sa > Sb ? Sa : Sb;
  return S > 1 ? (int64_t)S * base * W43_BM * 256 : 0;
}

// Modelled run time of the launch in microseconds (w43_pick_slabs' model; slabs != 0: with the K-slab form where it pays):
// what gsd_conv3x3_prefers_w2d compares the two-dimensional form against.
extern "C" double gsd_conv3x3_w43_estimate_us(int N, int H, int W, int Cin, int Cout, int slabs) {
  W43Plan p;
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || !plan_w43(N, H, W, Cout, &p)) return 0.0;
  const long base = (long)(p.fold ? 1 : N) * p.tiles_y * p.tiles_x * p.mblocks;
  const int nchunks = ceil_div(Cin, 4);
  const int S = (slabs && Cin % 4 == 0) ? w43_pick_slabs(base, nchunks, false) : 1;
  return w43_time_us(base, nchunks, S, false);
}

extern "C" int gsd_conv3x3_w43_partial_rows(int N, int H, int W, int Cout) {
  if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
  W43Plan p;
  if (!plan_w43(N, H, W, Cout, &p)) return 0;
  return (p.fold ? 1 : N) * p.tiles_y * p.tiles_x * 4;
}

// MFMA instructions of one launch (all blocks, padding included), for gsd_conv3x3_algo
extern "C" int64_t gsd_conv3x3_w43_mfma_count(int N, int H, int W, int Cin, int Cout) {
  W43Plan p;
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || !plan_w43(N, H, W, Cout, &p)) return 0;
  return (int64_t)(p.fold ? 1 : N) * p.tiles_y * p.tiles_x * p.mblocks * ceil_div(Cin, 4) * (4 * 72);
}

static int w43_impl(const gsd_src* src, int nsrc, const float* wt, int Cin, int Cout, const gsd_dst* dst, int ndst,
                    float* partials, const float* bw_raw, const float* bw_scale, const float* bw_shift, const float* bw_mean,
                    const float* bw_invstd, int N, int H, int W, void* stream, float* ws = nullptr, int64_t ws_elems = 0) {
  GSD_REQUIRE(src && dst && wt, GSD_ERR_BAD_ARG, "gsd_conv3x3_w43: null argument");
  GSD_REQUIRE(nsrc >= 1 && nsrc <= 2 && ndst >= 1 && ndst <= 2, GSD_ERR_BAD_ARG, "gsd_conv3x3_w43: nsrc/ndst must be 1 or 2");
  GSD_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, GSD_ERR_BAD_ARG, "gsd_conv3x3_w43: bad sizes");
  GSD_REQUIRE(H < 32768 && W < 32768, GSD_ERR_UNSUPPORTED, "gsd_conv3x3_w43: H, W must be < 32768");
  GSD_REQUIRE(((uintptr_t)wt & 15) == 0, GSD_ERR_BAD_ARG, "gsd_conv3x3_w43: weight layout must be 16-byte aligned");
  int csum = 0;
  for (int i = 0; i < nsrc; ++i) {
    if (int e = gsd_check_src(src[i], "gsd_conv3x3_w43 src", true)) return e;
    GSD_REQUIRE(src[i].scale == nullptr || src[i].relu != 0, GSD_ERR_UNSUPPORTED,
                "gsd_conv3x3_w43: an affine source segment must also have relu (zero padding uses a NaN sentinel)");
    GSD_REQUIRE((int64_t)src[i].H * src[i].w_stride < (1LL << 31), GSD_ERR_UNSUPPORTED, "gsd_conv3x3_w43: plane too large");
    csum += src[i].C;
  }
  GSD_REQUIRE(csum == Cin, GSD_ERR_BAD_ARG, "gsd_conv3x3_w43: source segments hold %d channels, Cin=%d", csum, Cin);
  csum = 0;
  for (int i = 0; i < ndst; ++i) {
    if (int e = gsd_check_dst(dst[i], "gsd_conv3x3_w43 dst", true)) return e;
    csum += dst[i].C;
  }
  GSD_REQUIRE(csum == Cout, GSD_ERR_BAD_ARG, "gsd_conv3x3_w43: destination segments hold %d channels, Cout=%d", csum, Cout);

  W43Plan pl;
  GSD_REQUIRE(plan_w43(N, H, W, Cout, &pl), GSD_ERR_UNSUPPORTED, "gsd_conv3x3_w43: no tile shape");
  if (pl.fold) {   // folded rows carry the image offset in 32-bit lane offsets
    for (int i = 0; i < nsrc; ++i)
      GSD_REQUIRE((int64_t)N * src[i].n_stride < (1LL << 31), GSD_ERR_UNSUPPORTED, "gsd_conv3x3_w43: batch too large for row folding");
    for (int i = 0; i < ndst; ++i)
      GSD_REQUIRE((int64_t)N * dst[i].n_stride < (1LL << 31), GSD_ERR_UNSUPPORTED, "gsd_conv3x3_w43: batch too large for row folding");
  }
  W43Params P;
  P.src0 = to_srcd(src[0]);
  P.src1 = nsrc > 1 ? to_srcd(src[1]) : null_srcd();
  P.dst0 = to_dstd(dst[0]);
  P.dst1 = ndst > 1 ? to_dstd(dst[1]) : null_dstd();
  P.wt = wt;
  P.stamps = nullptr;
#ifdef GSD_W43_STAMPS
  P.stamps = g_w43_stamp_buf;
#endif
  P.partials = partials;
  P.bw_raw = bw_raw; P.bw_scale = bw_scale; P.bw_shift = bw_shift; P.bw_mean = bw_mean; P.bw_invstd = bw_invstd;
  P.Cin = Cin;
  P.Cout = Cout;
  P.Mpad = round_up(Cout, 64);
  P.nchunks = ceil_div(Cin, 4);
  // measured (profiles/bench_conv_forms.py): two independent 4-wave blocks per CU hide each other's barriers better than one
  // 8-wave block shares its halo (+6 % for the 8-wave form at Cin <= 512, +1.5 % at Cin = 1024); GSD_W43_BIG=1 selects it
  const bool big = gsd_env_set("GSD_W43_BIG");
  // producer / consumer form (default): GSD_W43_NL loader waves per block (0: every wave issues its share of the DMA itself)
  const int nl = big ? 0 : gsd_env_int("GSD_W43_NL", 0);   // measured: a loader wave cannot keep up without a deeper ring
  // straight fills (no per-slot bookkeeping): every 4-channel chunk lies inside one source segment
  const bool fast = gsd_env_int("GSD_W43_FAST", 1) != 0 && Cin % 4 == 0 && (nsrc == 1 || src[0].C % 4 == 0);
  GSD_REQUIRE(nl >= 0 && nl <= 2, GSD_ERR_BAD_ARG, "gsd_conv3x3_w43: GSD_W43_NL must be 0, 1 or 2");
  const int WM = (pl.mblocks % 2 == 0 && big) ? 2 : 1;
  P.mblocks = pl.mblocks / WM;
  P.N = N; P.H = H; P.W = W;
  P.TH = pl.TH; P.TW = pl.TW; P.TWq = pl.TWq; P.tiles_y = pl.tiles_y; P.tiles_x = pl.tiles_x;
  P.WR = pl.WR; P.WC = pl.WC; P.WCp = pl.WCp;
  P.PS = round_up(P.WR * P.WCp, 4) + 4;   // + one bank group: the four channel planes of a k-step start 16 B apart (mod 4)
  P.NPV = ceil_div(P.WR * P.WCp, 64);
  GSD_REQUIRE(P.NPV <= 8, GSD_ERR_UNSUPPORTED, "gsd_conv3x3_w43: halo window too large");
  // 16-byte halo pieces: every source row must start 16-byte aligned in the consumer's column grid
  P.NP = pl.TW / 4 + 2;
  P.RPI = 64 / P.NP;
  P.NI = ceil_div(P.WR, P.RPI);
  P.RO = 0;
  bool x4 = gsd_env_int("GSD_W43_X4", 1) != 0 && P.NI <= 2;
  for (int i = 0; i < nsrc && x4; ++i)
    x4 = ((uintptr_t)src[i].ptr & 15) == 0 && src[i].off_w % 4 == 0 && src[i].n_stride % 4 == 0 && src[i].c_stride % 4 == 0 &&
         (i == 0 ? P.src0.ws : P.src1.ws) % 4 == 0;
  if (x4) {
    P.WCp = P.NP * 4;      // LDS row pitch: the NP pieces of a window row, contiguous (one instruction spans RPI rows)
    P.RO = 4;
    int best = -1;
    for (int ps = P.WR * P.WCp + 4; ps < P.WR * P.WCp + 4 + 68; ps += 4) {
      const int c = halo_read_cycles(pl.TWq, P.WCp, ps, P.RO);
      if (best < 0 || c < best) {
        best = c;
        P.PS = ps;
      }
    }
  }
  // unaligned 16-byte halo pieces (U4): straight-fill 4-wave form, every source vouches for 4 readable floats around its tensor
  // measured (profiles/bench_conv_ab.py, U4 against the dword form): +2..3 % where most blocks lie inside the image (7 and 4 tile
  // columns: levels 0, 1), -4..-10 % where every block touches both edges (levels 3, 4); over the whole train step the auto
  // rule is within noise (109.6 against 109.6 ms), so the form stays an option.  The fills removed altogether (-DW43_ABL) are
  // worth 9.5 %: it is the halo's memory traffic and latency the kernel waits for, not the instruction count -- unlike the dW
  // kernel, where the same change bought 11 %.  GSD_W43_U4 = 0 off (default), 1 where tiles_x >= 4, 2 always
  const int u4_env = gsd_env_int("GSD_W43_U4", 0);
  bool u4 = (u4_env == 2 || (u4_env == 1 && pl.tiles_x >= 4)) && fast && WM == 1 && nl == 0;
  for (int i = 0; i < nsrc; ++i)
    u4 = u4 && src[i].slack >= 4 && 4 * src[i].c_stride + (int64_t)(pl.fold ? N : 1) * src[i].n_stride < (1LL << 31);
  if (u4) {
    const int wcp = round_up(pl.WC, 4), ps = round_up(pl.WR * wcp, 4) + 4;
    if (ceil_div(ps, 64) <= 8) {
      x4 = false;
      P.WCp = wcp; P.PS = ps; P.RO = 0;
      P.NP = wcp / 4;
      P.NI = ceil_div(ps, 64);   // 4 planes x PS / 4 pieces, 64 pieces per instruction
    } else {
      u4 = false;
    }
  }
  P.fold = pl.fold;
  P.nslab = 1;
  P.slabs = nullptr;
  const long grid = (long)(pl.fold ? 1 : N) * pl.tiles_y * pl.tiles_x * P.mblocks;
  GSD_REQUIRE(grid < 2147483647L, GSD_ERR_UNSUPPORTED, "gsd_conv3x3_w43: grid too large");
  GSD_REQUIRE(!pl.fold || nl == 0, GSD_ERR_UNSUPPORTED, "gsd_conv3x3_w43: the loader-wave form does not fold rows (GSD_W43_FOLD=0)");
  const size_t lds = (size_t)(2 * (WM * W43_WTILE + (u4 ? P.NI * 256 : 4 * P.PS)) + 2 * 4 * P.nchunks + 4 * WM * W43_BM) * sizeof(float);
  bool plain = gsd_env_int("GSD_W43_PLAIN", 1) != 0;   // no deferred BatchNorm / ReLU on any source segment
  for (int i = 0; i < nsrc; ++i) plain = plain && src[i].scale == nullptr && src[i].relu == 0;
  // K slabs (the caller lends scratch): straight-fill 4-wave form only; the slab count shrinks to what the scratch holds
  if (ws != nullptr && fast && WM == 1 && nl == 0 && !u4) {
    GSD_REQUIRE(((uintptr_t)ws & 15) == 0, GSD_ERR_BAD_ARG, "gsd_conv3x3_w43: workspace must be 16-byte aligned");
    int S = w43_pick_slabs(grid, P.nchunks, bw_raw != nullptr);
    while (S > 1 && (int64_t)S * grid * W43_BM * 256 > ws_elems) --S;
    if (S > 1 && grid * S < 2147483647L) {
      P.nslab = S;
      P.slabs = ws;
      return launch_split(P, (int)grid, lds, (hipStream_t)stream, x4, plain);
    }
  }
  return launch_w43(P, (int)grid, lds, (hipStream_t)stream, WM, x4, nl, fast, plain, u4);
}

extern "C" int gsd_conv3x3_w43(const gsd_src* src, int nsrc, const float* wt, int Cin, int Cout, const gsd_dst* dst, int ndst,
                               float* partials, int N, int H, int W, void* stream) {
  return w43_impl(src, nsrc, wt, Cin, Cout, dst, ndst, partials, nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, stream);
}

extern "C" int gsd_conv3x3_w43_ws(const gsd_src* src, int nsrc, const float* wt, int Cin, int Cout, const gsd_dst* dst, int ndst,
                                  float* partials, float* ws, int64_t ws_elems, int N, int H, int W, void* stream) {
  return w43_impl(src, nsrc, wt, Cin, Cout, dst, ndst, partials, nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, stream, ws,
                  ws_elems);
}

extern "C" int gsd_conv3x3_w43_dgrad_bnrelu(const gsd_src* src, const float* wt, int Cin, int Cout, const gsd_dst* dst,
                                            const float* raw, const float* scale, const float* shift, const float* mean,
                                            const float* invstd, float* partials, int N, int H, int W, void* stream) {
  GSD_REQUIRE(dst && raw && scale && shift && mean && invstd && partials, GSD_ERR_BAD_ARG,
              "gsd_conv3x3_w43_dgrad_bnrelu: null argument");
  GSD_REQUIRE(dst->C == Cout && dst->H == H && dst->W == W && dst->off_h == 0 && dst->off_w == 0, GSD_ERR_BAD_ARG,
              "gsd_conv3x3_w43_dgrad_bnrelu: dst must be the full (Cout,H,W) gradient buffer (raw shares its strides)");
  return w43_impl(src, 1, wt, Cin, Cout, dst, 1, partials, raw, scale, shift, mean, invstd, N, H, W, stream);
}

extern "C" int gsd_conv3x3_w43_dgrad_bnrelu_ws(const gsd_src* src, const float* wt, int Cin, int Cout, const gsd_dst* dst,
                                               const float* raw, const float* scale, const float* shift, const float* mean,
                                               const float* invstd, float* partials, float* ws, int64_t ws_elems, int N, int H,
                                               int W, void* stream) {
  GSD_REQUIRE(dst && raw && scale && shift && mean && invstd && partials, GSD_ERR_BAD_ARG,
              "gsd_conv3x3_w43_dgrad_bnrelu: null argument");
  GSD_REQUIRE(dst->C == Cout && dst->H == H && dst->W == W && dst->off_h == 0 && dst->off_w == 0, GSD_ERR_BAD_ARG,
              "gsd_conv3x3_w43_dgrad_bnrelu: dst must be the full (Cout,H,W) gradient buffer (raw shares its strides)");
  return w43_impl(src, 1, wt, Cin, Cout, dst, 1, partials, raw, scale, shift, mean, invstd, N, H, W, stream, ws, ws_elems);
}
