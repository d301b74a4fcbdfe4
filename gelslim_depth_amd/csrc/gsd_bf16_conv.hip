// gsd_bf16_conv.hip -- forward-type convolutions of the bf16 path as implicit GEMM on v_mfma_f32_16x16x32_bf16.
//
//   out[pixel][m] = sum_taps sum_k  in[pixel moved by the tap][k] * Wt[tap][m][k]          (fp32 accumulation)
//
//   MODE 0  conv3x3 p1 s1 (unet.py:11,14 forward, and with transposed/flipped weights the dX half of its backward):
//           one (TH+2)x(TW+2) halo tile of 32 input channels in LDS serves all 9 taps; K walks (channel chunk, kernel
//           row), three taps per barrier.
//   MODE 1  "dense" taps without spatial reuse: 1x1 (the im2col'd first layer, K = 27 -> 32), ConvTranspose2d k2 s2
//           forward (1 tap, M = (kh,kw,co), scatter epilogue + bias; unet.py:36,41) and its dX (4 taps at stride 2).
//
// Operands reach LDS by global_load_lds_dwordx4 with per-lane source addresses (the halo gather, zero padding from a
// zero line, and the XOR swizzle of the weight tile all happen in the address computation; no register staging):
//   weights     [tap][BM rows][64 B], 16-byte piece g of row r stored at slot g ^ {0,2,3,1}[(r>>2)&3]
//   activations [pixel][96 B] (64 B of data + 32 B never written): with 6 pieces per pixel the ds_read_b128 of the B
//               operand (16 consecutive pixels x 4 pieces) is bank-conflict free for EVERY tap shift; 64-byte pixels
//               are 2-way conflicted whenever the shift is not a multiple of 4 pixels.
// Output-channel order inside a wave's 64: LDS weight row m*16 + i of MFMA tile m holds channel
// (m>>1)*32 + (i>>2)*8 + (m&1)*4 + (i&3) (a permutation of the DMA's SOURCE rows only), so lane group g ends up with channels
// g*8 .. g*8+7 (m-tiles 0, 1) and 32 + g*8 .. (m-tiles 2, 3) of its pixel: the epilogue stores (and the fused
// BatchNorm-backward loads) are 16-byte accesses, two per pixel and lane instead of four 8-byte ones, and the four lane
// groups of a pixel write 64 contiguous bytes per instruction -- the epilogue is bound by the store ISSUE rate, not by
// bandwidth.
// Block = 4 waves; wave tile 64 (m) x 128 (pixels) = 4 x 8 MFMA tiles = 128 accumulator registers; block tile
// 128 x 256 (WM=2, WN=2) or 64 x 512 (WM=1, WN=4): L2->LDS traffic per FLOP falls with the PIXEL extent of the tile
// (weights are re-read per pixel tile), which is why the tile is wide in pixels.
#include "gsd_bf16_common.h"

#include <type_traits>

__device__ const uint4 gsd_zero16[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};

struct GConvP {
  const u16* in;
  long long in_pitch;
  int Hin, Win;
  const u16* wt;  // [ntaps][Mpad][K]
  u16* out;
  long long out_pitch;
  int Hob, Wob;   // output BUFFER extent
  int N, H, W;    // GEMM pixel grid
  int K, M, Mpad, mblocks, nitems, xcd;
  int ntaps, stride;
  int ty[9], tx[9];
  int TH, TW, tiles_y, tiles_x, HC, HP;
  int buf;         // MODE 0: fills through buffer descriptors (image and weight image below 2 GiB)
  unsigned img_bytes, wt_bytes;
  int Cs, oy, ox;  // scatter (Cs > 0): m = q*Cs + co -> out pixel (2h + (q>>1) + oy, 2w + (q&1) + ox), channel co
  const float* bias;
  float* partials;
  // fused BatchNorm+ReLU backward pass 1 (dX launches): out receives dz = dX where relu(bn(bw_y)) > 0, partials the sums
  // of dz and dz*xhat.  bw_y has out's geometry (pitch bw_pitch).
  // eval-mode fusion: out = relu(acc * ep_scale[m] + ep_shift[m]) (BatchNorm with running statistics folded into the
  // coefficients + ReLU in the epilogue: the raw convolution output is never stored)
  const float* ep_scale; const float* ep_shift;
  const u16* bw_y;
  long long bw_pitch;
  const float* bw_scale; const float* bw_shift; const float* bw_mean; const float* bw_invstd;
};

#ifndef GCONV_ABL   // diagnostic builds (profiles/build_diag_one.sh; results are then garbage): 1 the iteration barrier does not wait
#define GCONV_ABL 0 // for the fills, 2 no fills after an item's first, 4 no epilogue stores, 8 no barrier (wait only), 16 no operand
                    // reads after an iteration's first tap, 32 no MFMAs
#endif
#ifndef GCONV_CONTIG   // 0 (default): instruction k of wave w fills piece 4k + w.  1 (diagnostic builds): a wave's DMA instructions fill
#define GCONV_CONTIG 0 // ONE contiguous LDS range, up to four sharing an LDS base (M0 set once, the 1-KiB steps in the instruction's
#endif                 // immediate offset) -- measured 17 % SLOWER over the layer set (5.93 vs 5.08 ms): the interleaved order stays
// LDS-DMA of one 1-KiB piece into piece index `pc` of the buffer at `buf`: with GCONV_CONTIG the LDS base is that of the
// piece's group of four and the remainder is the instruction's immediate, which moves the global address alike (undone here)
__device__ __forceinline__ void gconv_dma_piece(const void* g, unsigned char* buf, int pc) {
#if GCONV_CONTIG
  unsigned char* base = buf + (pc & ~3) * 1024;
  const char* gp = (const char*)g;
  switch (pc & 3) {
    case 0: __builtin_amdgcn_global_load_lds((const void*)gp, base, 16, 0, 0); break;
    case 1: __builtin_amdgcn_global_load_lds((const void*)(gp - 1024), base, 16, 1024, 0); break;
    case 2: __builtin_amdgcn_global_load_lds((const void*)(gp - 2048), base, 16, 2048, 0); break;
    default: __builtin_amdgcn_global_load_lds((const void*)(gp - 3072), base, 16, 3072, 0); break;
  }
#else
  __builtin_amdgcn_global_load_lds(g, buf + pc * 1024, 16, 0, 0);
#endif
}
// BUF (template, MODE 0): fills go through buffer descriptors (buffer_load_dwordx4 ... lds): one 32-bit offset per piece, the image /
// weight base and the channel chunk in the scalar offset, and an out-of-image or pad piece is an offset beyond num_records -- the
// hardware writes ZEROS for it (profiles/ubench/buffer_lds_oob.hip).  BUF = 0: per-lane 64-bit addresses, a zero line and a branch
// per piece (the form until round 4; still used when an image or the weight image exceeds 2 GiB, and by GSD_BF16_CONV_BUF=0).
__device__ __forceinline__ void gconv_dma_piece_buf(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned char* buf, int pc) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(buf + pc * 1024), 16, voff, soff, 0, 0);
}
#ifndef GCONV_STAMP   // diagnostic builds only: every block leaves (shader cycles, 100-MHz ticks) of its K-loop life in a buffer of
#define GCONV_STAMP 0 // its own -> the clock the chip holds under this kernel (profiles/bench_bf16_conv.py, GSD_DIAG_STAMPS=1)
#endif
#if GCONV_STAMP
__device__ unsigned long long gconv_stamp_buf[2 * 4096];
extern "C" int gsd_diag_gconv_stamps(unsigned long long* host, int nblocks) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(gconv_stamp_buf), sizeof(unsigned long long) * 2 * (nblocks < 4096 ? nblocks : 4096)) == hipSuccess ? 0 : 1;
}
#endif
template <int MODE, int WM, int WN, int BUF = 0>
__global__ __launch_bounds__(256) void gconv_bf16_kernel(const GConvP P) {
  static_assert(!BUF || MODE == 0, "buffer-addressed fills: the 3x3 form");
  constexpr int BM = WM * 64, NPX = WN * 128;
  constexpr int MT = 4, NT = 8;
  // k-steps per barrier: MODE 0 the 3 taps of a kernel row; MODE 1 NSUB consecutive 32-channel sub-chunks of one tap
  // (two for the 128 x 256 tile: half the barriers and twice the lead time of every fill; the 64 x 512 tile has no LDS
  // for a second activation plane and is only used by the first layer, K = 32)
  constexpr int NSUB = (MODE == 1 && WN == 2) ? 2 : 1;
  constexpr int NTAPI = MODE == 0 ? 3 : NSUB;
  constexpr int WBUF = NTAPI * BM * 64;             // bytes of one weight image
  constexpr int NWI = NTAPI * BM / 16 / 4;          // weight DMA instructions per wave per iteration
  constexpr int XPP = NPX * 3 / 32 / 4;             // MODE 1: activation DMA instructions per wave and plane
  constexpr int MAXX = MODE == 0 ? (WN == 2 ? 10 : 16) : NSUB * XPP;   // activation DMA instructions per wave

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int XBUF = MAXX * 4096;   // every DMA instruction moves a whole 1 KiB piece (pad and slack lanes carry zeros)
  unsigned char* Wl = smem;
  unsigned char* Xl = smem + 2 * WBUF;
  float* sBw = reinterpret_cast<float*>(smem + 2 * WBUF + 2 * XBUF);   // [4][BM] output-side BatchNorm coefficients (fused dX)
  float* sSt = sBw + 4 * BM;                                           // [4 waves][2][64] running partial sums of the block

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int g = lane >> 4, j = lane & 15;
  // piece (1 KiB) index inside a buffer that instruction k of this wave fills (NI = instructions per wave for that buffer)
  auto piece = [&](int k, int ni) { return GCONV_CONTIG ? wave * ni + k : k * 4 + wave; };

  // Persistent block: it walks pixel tiles pt, pt + pt_step, ... < pt_end for ONE m-block (gridDim.x is a multiple of mblocks),
  // so every weight address and coefficient is fixed per block.
  // XCD-aware order (P.xcd): the hardware deals blocks round-robin over the 8 XCDs, each with its own L2.  XCD x takes the
  // contiguous pixel-tile range [ntile x/8, ntile (x+1)/8) and its gridDim/8 blocks sweep it together -- the m-blocks of a pixel
  // tile (same activation tile) and the neighbouring tiles (shared halo rows) then sit behind ONE L2 at about the same time,
  // instead of the activation tile being fetched once per XCD that hosts one of its m-blocks.
  const int ntile = P.nitems / P.mblocks;
  int mb, pt, pt_step, pt_end;
  int prow;   // this block's row group of the BatchNorm partials: unique per (pixel-tile lane of the grid), shared by its m-blocks
  if (P.xcd) {
    const int x = blockIdx.x & 7, l = blockIdx.x >> 3, nq = (int)(gridDim.x >> 3) / P.mblocks;
    mb = l % P.mblocks;
    pt = (int)((long long)ntile * x / 8) + l / P.mblocks;
    pt_step = nq;
    pt_end = (int)((long long)ntile * (x + 1) / 8);
    prow = x * nq + l / P.mblocks;
  } else {
    mb = blockIdx.x % P.mblocks;
    pt = blockIdx.x / P.mblocks;
    pt_step = gridDim.x / P.mblocks;
    pt_end = ntile;
    prow = blockIdx.x / P.mblocks;
  }
  const int m0 = mb * BM;
  const int tpi = P.tiles_y * P.tiles_x;
  const int cbs = P.TW >> 4;  // 16-pixel column blocks per tile row
  auto decode = [&](int pt_, int& n, int& h0, int& w0) {
    const int pt = pt_;
    n = pt / tpi;
    const int trem = pt - n * tpi;
    const int tyi = trem / P.tiles_x;
    h0 = tyi * P.TH;
    w0 = (trem - tyi * P.tiles_x) * P.TW;
  };
  if (P.partials != nullptr) {   // each (wave, row) cell is only ever touched by one lane: no synchronisation needed
    for (int c = tid; c < 4 * 2 * 64; c += 256) sSt[c] = 0.f;
  }
  if (P.ep_scale != nullptr) {   // shares the coefficient cells with the fused dX mode (they exclude each other)
    for (int c = tid; c < BM; c += 256) {
      const int co = m0 + c < P.M ? m0 + c : 0;
      sBw[c] = P.ep_scale[co];
      sBw[BM + c] = P.ep_shift[co];
    }
  }
  if (P.bw_y != nullptr) {   // visible to everyone after the first barrier of the K loop
    for (int c = tid; c < BM; c += 256) {
      const int co = m0 + c < P.M ? m0 + c : 0;
      sBw[c] = P.bw_scale[co];
      sBw[BM + c] = P.bw_shift[co];
      sBw[2 * BM + c] = P.bw_mean[co];
      sBw[3 * BM + c] = P.bw_invstd[co];
    }
  }

  // ---- DMA bookkeeping (chunk-invariant part of every source address) ---------------------------------------------
  int woff[NWI];
#pragma unroll
  for (int k = 0; k < NWI; ++k) {
    const int idx = piece(k, NWI) * 64 + lane;
    const int tapk = idx / (BM * 4);
    const int row = (idx >> 2) % BM;
    const int gg = (idx & 3) ^ ((0x1320 >> (((row >> 2) & 3) * 4)) & 3);   // {0,2,3,1}
    // LDS row `row` = (64-block, m-tile mm, tile row ii) receives the weights of channel (mm>>1)*32 + (ii>>2)*8 + (mm&1)*4 + (ii&3)
    const int srow = (row & ~63) | (((row >> 5) & 1) << 5) | (((row & 15) >> 2) << 3) | (((row >> 4) & 1) << 2) | (row & 3);
    woff[k] = MODE == 0 ? (tapk * P.Mpad + m0 + srow) * P.K + gg * 8 : (m0 + srow) * P.K + tapk * 32 + gg * 8;
  }
  // activations, tile-invariant part: (y << 16 | x << 4 | slot) of this lane's 16-byte piece -- halo position (MODE 0) or
  // tile pixel (MODE 1) -- or -2: a pad / slack piece.  Those are filled from the zero line like out-of-image pixels:
  // with no lane skipped the whole fill is straight-line code that can be interleaved with the MFMAs.
  int xpk[MAXX];
#pragma unroll
  for (int k = 0; k < MAXX; ++k) {
    const int o = piece(MODE == 0 ? k : k % XPP, MODE == 0 ? MAXX : XPP) * 1024 + lane * 16;
    const int px = o / 96, slot = (o - px * 96) >> 4;
    const int rowlen = MODE == 0 ? P.HC : P.TW;
    int v = -2;
    if (px < (MODE == 0 ? P.HP : NPX) && slot < 4) {
      const int y = px / rowlen, x = px - y * rowlen;
      v = (y << 16) | (x << 4) | slot;
    }
    xpk[k] = v;
  }
  // per item: MODE 0 -> element offset inside the image of the piece, negative: zeros;
  //           MODE 1 -> xpk when the tile pixel exists, negative: zeros
  int xoff[MAXX];
  auto prep = [&](int h0, int w0) {
#pragma unroll
    for (int k = 0; k < MAXX; ++k) {
      int v = -2;
      if (xpk[k] != -2) {
        const int y = xpk[k] >> 16, x = (xpk[k] >> 4) & 0xfff, slot = xpk[k] & 15;
        if (MODE == 0) {
          const int hi = h0 - 1 + y, wi = w0 - 1 + x;
          v = ((unsigned)hi < (unsigned)P.Hin && (unsigned)wi < (unsigned)P.Win) ? (int)((hi * P.Win + wi) * P.in_pitch) + slot * 8 : -1;
          if (BUF) v = v >= 0 ? v * 2 : (int)0x80000000u;   // byte offset inside the image, or beyond it: zeros
        } else {
          v = (h0 + y < P.H && w0 + x < P.W) ? xpk[k] : -1;
        }
      }
      if (BUF && v == -2) v = (int)0x80000000u;   // pad / slack pieces: zeros as well
      xoff[k] = v;
    }
  };

  // DMA of iteration `it` of the item at (n, h0, w0) -- whose offsets are in xoff -- into the buffers of GLOBAL
  // iteration git (the parity keeps alternating across items, so an item's first fill never hits a buffer in use).
  // Slots [0, NWI) move the weights, slots [NWI, NWI + MAXX) the activations; every slot is one branch-free instruction.
  auto dma_slot = [&](int slot, int it, int git, int n, int h0, int w0) {
    int chunk, tap0;
    if (MODE == 0) {
      chunk = it / 3;
      tap0 = (it - chunk * 3) * 3;
    } else {
      const int nch = (P.K >> 5) / NSUB;      // iterations per tap
      tap0 = it / nch;
      chunk = (it - tap0 * nch) * NSUB;       // first 32-channel chunk of the iteration
    }
    if (BUF) {
      // descriptors are rebuilt here from uniform values (scalar ALU): the weight image, and image n of the activations
      if (slot < NWI) {
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)P.wt, 0, P.wt_bytes, 0x00020000);
        gconv_dma_piece_buf(rw, (unsigned)woff[slot] * 2u, (unsigned)((tap0 * P.Mpad * P.K + chunk * 32) * 2), Wl + (git & 1) * WBUF, piece(slot, NWI));
      } else {
        const int k = slot - NWI;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(P.in + (long long)n * P.Hin * P.Win * P.in_pitch), 0, P.img_bytes, 0x00020000);
        gconv_dma_piece_buf(rx, (unsigned)xoff[k], (unsigned)(chunk * 64), Xl + ((git / 3) & 1) * XBUF, piece(k, MAXX));
      }
      return;
    }
    if (slot < NWI) {
      const u16* wsrc = P.wt + (long long)tap0 * P.Mpad * P.K + chunk * 32;
      gconv_dma_piece((const void*)(wsrc + woff[slot]), Wl + (git & 1) * WBUF, piece(slot, NWI));
      return;
    }
    const int k = slot - NWI;
    const u16* src = P.in + (long long)n * P.Hin * P.Win * P.in_pitch + chunk * 32;
    const void* sp = (const void*)gsd_zero16;
    if (MODE == 0) {
      if (xoff[k] >= 0) sp = (const void*)(src + xoff[k]);
      gconv_dma_piece(sp, Xl + ((git / 3) & 1) * XBUF, piece(k, MAXX));
    } else {
      if (xoff[k] >= 0) {
        const int hi = P.stride * (h0 + (xoff[k] >> 16)) + P.ty[tap0], wi = P.stride * (w0 + ((xoff[k] >> 4) & 0xfff)) + P.tx[tap0];
        if ((unsigned)hi < (unsigned)P.Hin && (unsigned)wi < (unsigned)P.Win)
          sp = (const void*)(src + (long long)(hi * P.Win + wi) * P.in_pitch + (k / XPP) * 32 + (xoff[k] & 15) * 8);
      }
      gconv_dma_piece(sp, Xl + (git & 1) * XBUF + (k / XPP) * (XPP * 4096), piece(k % XPP, XPP));   // plane k / XPP
    }
  };
  // slots of one iteration: the weights always; the activations with every iteration (MODE 1) or with a chunk's first
  // kernel row (MODE 0)
  auto nslots = [&](int it) { return (MODE == 1 || it % 3 == 0) ? NWI + MAXX : NWI; };
  auto issue = [&](int it, int git, int n, int h0, int w0) {
#pragma unroll
    for (int sl = 0; sl < NWI + MAXX; ++sl)
      if (sl < nslots(it)) dma_slot(sl, it, git, n, h0, w0);
  };

  // ---- operand read offsets ------------------------------------------------------------------------------------
  const int aoff = (wm * 64 + j) * 64 + ((g ^ ((0x1320 >> (((j >> 2) & 3) * 4)) & 3)) << 4);
  int boff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int nt = wn * NT + t;
    if (MODE == 0) {
      const int r = nt / cbs, cb = nt - r * cbs;
      boff[t] = (r * P.HC + cb * 16 + j) * 96 + g * 16;
    } else {
      boff[t] = (nt * 16 + j) * 96 + g * 16;
    }
  }

  const int iters = MODE == 0 ? (P.K >> 5) * 3 : P.ntaps * ((P.K >> 5) / NSUB);
  if (pt >= pt_end) {   // (an XCD's range can be shorter than its blocks): this block's partial rows are zeros
    if (P.partials != nullptr && lane < 64) {
      float* row = P.partials + (size_t)(prow * WN + wn) * (2 * P.Mpad);
      const int mrow = m0 + wm * 64 + lane;
      if (mrow < P.Mpad) row[mrow] = row[P.Mpad + mrow] = 0.f;
    }
    return;
  }
  int n, h0, w0, git = 0;
  decode(pt, n, h0, w0);
  prep(h0, w0);
  issue(0, 0, n, h0, w0);
#if GCONV_STAMP
  const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  float s1[MT][4], s2[MT][4];   // this lane's running BatchNorm sums over all the block's items
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) s1[m][r] = s2[m][r] = 0.f;
  while (true) {
  // The accumulators start at the bias (MODE 1 only): no load is then left for the epilogue, where it would sit between
  // the stores -- loads and stores share vmcnt on gfx950, and hipcc answers a load of unknown age inside divergent
  // control flow with s_waitcnt vmcnt(0) in front of EVERY store (measured: 13 us per block).
  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    f32x4 init = f32x4{0.f, 0.f, 0.f, 0.f};
    if (MODE == 1 && P.bias != nullptr) {
      const int mrow = m0 + wm * 64 + (m >> 1) * 32 + g * 8 + (m & 1) * 4;
      if (mrow < P.M) {
        const int co = P.Cs > 0 ? mrow % P.Cs : mrow;
        init = f32x4{P.bias[co], P.bias[co + 1], P.bias[co + 2], P.bias[co + 3]};
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[m][t] = init;
  }

  const int next = pt + pt_step;
  int nn = n, nh0 = h0, nw0 = w0;
  int it = 0;
  // One iteration = one barrier = NTAPI taps.  KH (MODE 0: the kernel row, = it % 3) is a compile-time constant so that the
  // iteration body is straight-line code: the fill of the NEXT iteration (weights; plus the next chunk's halo tile when
  // KH == 2) is issued slot by slot BETWEEN the MFMAs, in the 8 of an MFMA's 16 cycles that leave the issue port free,
  // instead of in front of them.
  auto body = [&](auto khc) {
    constexpr int KH = decltype(khc)::value;
    constexpr int NS = (MODE == 1 || KH == 2) ? NWI + MAXX : NWI;     // DMA slots of this iteration
    constexpr int SPT = (NS + NTAPI - 1) / NTAPI;                       // per tap
#if (GCONV_ABL) & 1
    __syncthreads();
#elif (GCONV_ABL) & 8
    __builtin_amdgcn_s_waitcnt(0x0F70);
#else
    gsd_dma_barrier();   // iteration git's DMA has landed (vmcnt(0) + barrier) and every wave has left the other buffers
#endif
    int f_it = it + 1, f_n = n, f_h0 = h0, f_w0 = w0;
    if (it + 1 == iters) {   // the fill belongs to the next item: it flies during this item's last iteration and epilogue
      f_it = 0;              // (after the last item it repeats this item's first fill, which nobody reads: no branch)
      if (next < pt_end) decode(next, nn, nh0, nw0);
      prep(nh0, nw0);
      f_n = nn; f_h0 = nh0; f_w0 = nw0;
    }
    const unsigned char* Wc = Wl + (git & 1) * WBUF + aoff;
    const unsigned char* Xc = MODE == 0 ? Xl + ((git / 3) & 1) * XBUF + KH * P.HC * 96 : Xl + (git & 1) * XBUF;
    u32x4 a[2][MT], b[NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) a[0][m] = *reinterpret_cast<const u32x4*>(Wc + m * 1024);
#pragma unroll
    for (int t = 0; t < NT; ++t) b[t] = *reinterpret_cast<const u32x4*>(Xc + boff[t]);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int BSTEP = MODE == 0 ? 96 : NPX * 96;   // B operand of the next tap: one pixel on (MODE 0), the next plane (MODE 1)
#pragma unroll
    for (int kw = 0; kw < NTAPI; ++kw) {
      // 16 micro-steps of {two MFMAs, one or two operand reads for the next tap, one DMA slot}, pinned in this order: the
      // LDS-touching instructions (ds_read, global_load_lds) keep their program order anyway, so the interleaving has to
      // be written out -- a sched_group_barrier pattern over the whole tap leaves the reads bunched in front.
      // Order: pixel-tile PAIR major (micro-step i: tiles 2(i/4), 2(i/4)+1 x m-tile i%4): a pair's B operands are dead after
      // four micro-steps and are re-read IN PLACE for the next tap (one set of B registers instead of two); the last pair's
      // at the start of the next tap.  The A operands (all four live through the tap) stay double-buffered.
#pragma unroll
      for (int i = 0; i < MT * NT / 2; ++i) {
        const int m = i % MT, t = 2 * (i / MT);
#if (GCONV_ABL) & 32   // no MFMAs (one VALU op keeps the operand reads alive)
        acc[m][t][0] += __uint_as_float(a[kw & 1][m][0] ^ b[t][0]);
        acc[m][t + 1][0] += __uint_as_float(a[kw & 1][m][1] ^ b[t + 1][1]);
#else
        acc[m][t] = mfma_bf16(a[kw & 1][m], b[t], acc[m][t]);
        acc[m][t + 1] = mfma_bf16(a[kw & 1][m], b[t + 1], acc[m][t + 1]);
#endif
#if !((GCONV_ABL) & 16)   // 16: no operand reads after an iteration's first tap
        if (kw + 1 < NTAPI) {
          if (i < MT) a[(kw + 1) & 1][i] = *reinterpret_cast<const u32x4*>(Wc + (kw + 1) * BM * 64 + i * 1024);
          if (i >= MT && (i % MT) < 2) {   // i = 4, 5 -> b[0], b[1]; 8, 9 -> b[2], b[3]; 12, 13 -> b[4], b[5]
            const int bt = 2 * (i / MT - 1) + (i % MT);
            b[bt] = *reinterpret_cast<const u32x4*>(Xc + boff[bt] + (kw + 1) * BSTEP);
          }
        }
        if (kw > 0 && i < 2) b[NT - 2 + i] = *reinterpret_cast<const u32x4*>(Xc + boff[NT - 2 + i] + kw * BSTEP);   // the last pair
#endif
        if (i < SPT && kw * SPT + i < NS && !(((GCONV_ABL) & 2) && git > 0)) dma_slot(kw * SPT + i, f_it, git + 1, f_n, f_h0, f_w0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    ++it;
    ++git;
  };
  while (it < iters) {
    if (MODE == 0) {
      body(std::integral_constant<int, 0>{});
      body(std::integral_constant<int, 1>{});
      body(std::integral_constant<int, 2>{});
    } else {
      body(std::integral_constant<int, 0>{});
    }
  }

  // ---- epilogue: bf16 store (16 channels per lane and pixel), optional bias / scatter / BatchNorm partial sums -----------

  // A lane's 16 output channels are two runs of 8 (see the file header): resolve them / their scatter quadrants once
  const int ch0 = m0 + wm * 64 + g * 8;    // first channel of the lane's run A (m-tiles 0, 1); run B (m-tiles 2, 3) starts 32 later
  long long ooff[2];                        // element offset of a run relative to the (scaled) pixel: channel + scatter shift
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    int co = ch0 + 32 * k, q = 0;
    if (P.Cs > 0) {                         // Cs % 16 == 0: the 8 channels of a run share their quadrant
      q = co / P.Cs;
      co -= q * P.Cs;
    }
    ooff[k] = ((long long)((q >> 1) + P.oy) * P.Wob + (q & 1) + P.ox) * P.out_pitch + co;
  }
  const int sm = P.Cs > 0 ? 2 : 1;
  // Interior tiles (all 256 / 512 pixels inside the image) take a store path WITHOUT per-store branches: `guard` is a
  // compile-time constant in each instantiation of the lambda.  M % 16 == 0: the 8 channels of a run exist together.
  const bool ch_ok[2] = {ch0 < P.M, ch0 + 32 < P.M};
  const bool interior = h0 + P.TH <= P.H && w0 + P.TW <= P.W;
  typedef unsigned u32x4s __attribute__((ext_vector_type(4), aligned(8)));   // pitches are multiples of 4 elements: 8-byte aligned
  auto epilogue = [&](auto guard_c) {
    constexpr bool GUARD = decltype(guard_c)::value;
    if (P.bw_y != nullptr) {
      // Fused pass 1 of BatchNorm+ReLU backward.  The raw outputs are loaded half a wave tile at a time IN FRONT of that
      // half's stores (loads and stores share vmcnt: a load between two stores serialises them); coefficients from LDS.
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        u32x4 yr[NT / 2][2];
#pragma unroll
        for (int tt = 0; tt < NT / 2; ++tt) {
          const int t = half * (NT / 2) + tt, nt = wn * NT + t;
          const int r = nt / cbs, cb = nt - r * cbs;
          const int h = GUARD ? min(h0 + r, P.H - 1) : h0 + r, w = GUARD ? min(w0 + cb * 16 + j, P.W - 1) : w0 + cb * 16 + j;
          const u16* yp = P.bw_y + ((long long)(n * P.H + h) * P.W + w) * P.bw_pitch;
          yr[tt][0] = *reinterpret_cast<const u32x4s*>(yp + (ch_ok[0] ? ch0 : 0));
          yr[tt][1] = *reinterpret_cast<const u32x4s*>(yp + (ch_ok[1] ? ch0 + 32 : 0));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tt = 0; tt < NT / 2; ++tt) {
          const int t = half * (NT / 2) + tt, nt = wn * NT + t;
          const int r = nt / cbs, cb = nt - r * cbs;
          const int h = h0 + r, w = w0 + cb * 16 + j;
          const bool pix_ok = !GUARD || (h < P.H && w < P.W);
          u16* ot = P.out + ((long long)(n * P.Hob + h) * P.Wob + w) * P.out_pitch + ch0;
          unsigned pk[2 * MT];
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const int cl = wm * 64 + (m >> 1) * 32 + g * 8 + (m & 1) * 4;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(sBw + cl), sh = *reinterpret_cast<const f32x4*>(sBw + BM + cl);
            const f32x4 mu = *reinterpret_cast<const f32x4*>(sBw + 2 * BM + cl), is = *reinterpret_cast<const f32x4*>(sBw + 3 * BM + cl);
            const unsigned y01 = yr[tt][m >> 1][(m & 1) * 2], y23 = yr[tt][m >> 1][(m & 1) * 2 + 1];
            float yv[4] = {__uint_as_float(y01 << 16), __uint_as_float(y01 & 0xffff0000u), __uint_as_float(y23 << 16),
                           __uint_as_float(y23 & 0xffff0000u)};
            float dz[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) dz[e] = fmaf(yv[e], sc[e], sh[e]) > 0.f ? acc[m][t][e] : 0.f;
            const unsigned lo = pack_bf16(dz[0], dz[1]), hi = pack_bf16(dz[2], dz[3]);
            pk[2 * m] = lo;
            pk[2 * m + 1] = hi;
            if (ch_ok[m >> 1] && pix_ok) {
              const float q[4] = {__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u), __uint_as_float(hi << 16),
                                  __uint_as_float(hi & 0xffff0000u)};   // sums of the values as stored
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                s1[m][e] += q[e];
                s2[m][e] = fmaf(q[e], (yv[e] - mu[e]) * is[e], s2[m][e]);
              }
            }
          }
          if (ch_ok[0] && pix_ok && !((GCONV_ABL) & 4)) *reinterpret_cast<u32x4s*>(ot) = u32x4{pk[0], pk[1], pk[2], pk[3]};
          if (ch_ok[1] && pix_ok && !((GCONV_ABL) & 4)) *reinterpret_cast<u32x4s*>(ot + 32) = u32x4{pk[4], pk[5], pk[6], pk[7]};
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int nt = wn * NT + t;
        const int r = nt / cbs, cb = nt - r * cbs;
        const int h = h0 + r, w = w0 + cb * 16 + j;
        const bool pix_ok = !GUARD || (h < P.H && w < P.W);
        u16* ot = P.out + ((long long)(n * P.Hob + sm * h) * P.Wob + sm * w) * P.out_pitch;
        unsigned pk[2 * MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          f32x4 v = acc[m][t];
          if (P.ep_scale != nullptr) {
            const int cl = wm * 64 + (m >> 1) * 32 + g * 8 + (m & 1) * 4;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(sBw + cl), sh = *reinterpret_cast<const f32x4*>(sBw + BM + cl);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], sc[e], sh[e]), 0.f);
          }
          const unsigned lo = pack_bf16(v[0], v[1]);
          const unsigned hi = pack_bf16(v[2], v[3]);
          pk[2 * m] = lo;
          pk[2 * m + 1] = hi;
          if (ch_ok[m >> 1] && pix_ok && P.partials != nullptr) {   // statistics of the values as stored (what the BatchNorm kernel will read back)
            const float q0 = __uint_as_float(lo << 16), q1 = __uint_as_float(lo & 0xffff0000u);
            const float q2 = __uint_as_float(hi << 16), q3 = __uint_as_float(hi & 0xffff0000u);
            s1[m][0] += q0; s2[m][0] = fmaf(q0, q0, s2[m][0]);
            s1[m][1] += q1; s2[m][1] = fmaf(q1, q1, s2[m][1]);
            s1[m][2] += q2; s2[m][2] = fmaf(q2, q2, s2[m][2]);
            s1[m][3] += q3; s2[m][3] = fmaf(q3, q3, s2[m][3]);
          }
        }
        if (ch_ok[0] && pix_ok && !((GCONV_ABL) & 4)) *reinterpret_cast<u32x4s*>(ot + ooff[0]) = u32x4{pk[0], pk[1], pk[2], pk[3]};
        if (ch_ok[1] && pix_ok && !((GCONV_ABL) & 4)) *reinterpret_cast<u32x4s*>(ot + ooff[1]) = u32x4{pk[4], pk[5], pk[6], pk[7]};
      }
    }
  };
  if (interior) epilogue(std::integral_constant<bool, false>{});
  else epilogue(std::integral_constant<bool, true>{});
  if (next >= pt_end) break;
  pt = next;
  n = nn;
  h0 = nh0;
  w0 = nw0;
  }
  gsd_dma_barrier();   // vmcnt(0): the last (unread) fill must have landed before the block gives its LDS back
#if GCONV_STAMP
  if (tid == 0 && blockIdx.x < 4096) {
    gconv_stamp_buf[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st_c0;
    gconv_stamp_buf[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - st_r0;
  }
#endif
  if (P.partials != nullptr) {
    // The block is persistent, so its statistics are too: every lane has summed its pixels of ALL the block's items in
    // registers; the 16-lane rows are summed with DPP once, here, and ONE partial row per (block, wave) leaves for HBM (a few
    // hundred rows per launch instead of one per pixel tile, so the column reduction behind it is nearly free).
    float* cell = sSt + wave * 128 + g * 8;    // cell index = channel offset inside the wave's 64: (m>>1)*32 + g*8 + (m&1)*4 + r
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a1 = reduce16_to_lane15(s1[m][r]), a2 = reduce16_to_lane15(s2[m][r]);
        if (j == 15) {
          cell[(m >> 1) * 32 + (m & 1) * 4 + r] = a1;
          cell[64 + (m >> 1) * 32 + (m & 1) * 4 + r] = a2;
        }
      }
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's cells are written (the row below is read by the same wave)
  }
  if (P.partials != nullptr && lane < 64) {
    float* row = P.partials + (size_t)(prow * WN + wn) * (2 * P.Mpad);
    const int mrow = m0 + wm * 64 + lane;
    if (mrow < P.Mpad) {
      row[mrow] = sSt[wave * 128 + lane];
      row[P.Mpad + mrow] = sSt[wave * 128 + 64 + lane];
    }
  }
}

// ---- host side ----------------------------------------------------------------------------------------------------
namespace {

struct Plan {
  bool wide;   // M <= 64: 64 x 512 tile, else 128 x 256
  int BM, NPX, TH, TW, tiles_y, tiles_x, mblocks, Mpad, HC, HP;
};

Plan make_plan(int H, int W, int M) {
  Plan p;
  p.wide = M <= 64;
  p.BM = p.wide ? 64 : 128;
  p.NPX = p.wide ? 512 : 256;
  long best = -1;
  const int force_tw = gsd_env_int("GSD_BF16_TW", 0);   // tuning: force the tile width (16, 32 or 64)
  for (int tw = 16; tw <= 64; tw *= 2) {
    if (force_tw && tw != force_tw) continue;
    const int th = p.NPX / tw;
    const long cost = (long)ceil_div(H, th) * ceil_div(W, tw);   // tiles; ties -> wider rows (longer DMA row segments)
    if (best < 0 || cost <= best) {
      best = cost;
      p.TW = tw;
      p.TH = th;
    }
  }
  p.tiles_y = ceil_div(H, p.TH);
  p.tiles_x = ceil_div(W, p.TW);
  p.Mpad = round_up(M, 128);
  p.mblocks = ceil_div(M, p.BM);
  p.HC = p.TW + 2;
  p.HP = (p.TH + 2) * (p.TW + 2);
  return p;
}

int cu_count() {
  static int n = 0;   // benign race: every thread computes the same value
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    n = v;
  }
  return n;
}

// items = (pixel tile, m-block) pairs; one persistent block per CU (the LDS image allows one), a multiple of mblocks
long launch_grid(long items, int mblocks) {
  long grid = cu_count() / mblocks * mblocks;
  if (grid < mblocks) grid = mblocks;
  return grid > items ? items : grid;
}

template <int MODE, int WM, int WN, int BUF = 0>
int launch(GConvP& P, long items, size_t lds, hipStream_t st, const char* what) {
  GSD_REQUIRE(items > 0 && items < 2147483647L, GSD_ERR_UNSUPPORTED, "%s: %ld work items out of range", what, items);
  P.nitems = (int)items;
  const long grid = launch_grid(items, P.mblocks);
  P.xcd = (gsd_env_int("GSD_BF16_XCD", 1) != 0 && grid % 8 == 0 && (grid / 8) % P.mblocks == 0) ? 1 : 0;
  static gsd_attr_once big_lds;   // per-device cache of an idempotent launch attribute (gsd_common.h)
  if (hipError_t e = gsd_allow_big_lds(big_lds, reinterpret_cast<const void*>(&gconv_bf16_kernel<MODE, WM, WN, BUF>)); e != hipSuccess) {
    gsd_set_error("%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e));
    return GSD_ERR_HIP;
  }
  GSD_REQUIRE(grid > 0 && grid < 2147483647L, GSD_ERR_UNSUPPORTED, "%s: grid %ld out of range", what, grid);
  GSD_REQUIRE(lds <= 160 * 1024, GSD_ERR_UNSUPPORTED, "%s: LDS %zu B too large", what, lds);
  hipLaunchKernelGGL((gconv_bf16_kernel<MODE, WM, WN, BUF>), dim3((unsigned)grid), dim3(256), lds, st, P);
  GSD_LAUNCH_CHECK(what);
  return GSD_OK;
}

}  // namespace

extern "C" int gsd_bf16_conv_mpad(int M) { return M > 0 ? round_up(M, 128) : 0; }

extern "C" int gsd_bf16_conv_partial_rows(int N, int H, int W, int M) {
  if (N <= 0 || H <= 0 || W <= 0 || M <= 0) return 0;
  const Plan p = make_plan(H, W, M);
  return (int)(launch_grid((long)N * p.tiles_y * p.tiles_x * p.mblocks, p.mblocks) / p.mblocks) * (p.wide ? 4 : 2);
}

static int conv3x3_impl(const gsd_nhwc* in, const void* wt, const gsd_nhwc* out, int K, int M, float* partials,
                        const gsd_bf16_bnbwd* bw, const float* ep_scale, const float* ep_shift, void* stream) {
  if (int e = gsd_check_nhwc(in, "gsd_bf16_conv3x3 in")) return e;
  if (int e = gsd_check_nhwc(out, "gsd_bf16_conv3x3 out")) return e;
  GSD_REQUIRE(wt != nullptr, GSD_ERR_BAD_ARG, "gsd_bf16_conv3x3: null weights");
  GSD_REQUIRE(K > 0 && K % 32 == 0 && in->C == K, GSD_ERR_UNSUPPORTED, "gsd_bf16_conv3x3: K=%d must be a multiple of 32 and in->C",
              K);
  GSD_REQUIRE(M > 0 && M % 16 == 0 && out->C == M, GSD_ERR_UNSUPPORTED, "gsd_bf16_conv3x3: M=%d must be a multiple of 16 and out->C",
              M);
  GSD_REQUIRE(in->N == out->N && in->H == out->H && in->W == out->W, GSD_ERR_BAD_ARG, "gsd_bf16_conv3x3: in/out extents differ");
  GSD_REQUIRE((out->pitch & 3) == 0, GSD_ERR_UNSUPPORTED, "gsd_bf16_conv3x3: out pitch must be a multiple of 4");
  const Plan pl = make_plan(in->H, in->W, M);
  GConvP P;
  P.in = (const u16*)in->ptr; P.in_pitch = in->pitch; P.Hin = in->H; P.Win = in->W;
  P.wt = (const u16*)wt;
  P.out = (u16*)out->ptr; P.out_pitch = out->pitch; P.Hob = out->H; P.Wob = out->W;
  P.N = in->N; P.H = in->H; P.W = in->W;
  P.K = K; P.M = M; P.Mpad = pl.Mpad; P.mblocks = pl.mblocks;
  P.ntaps = 9; P.stride = 1;
  for (int t = 0; t < 9; ++t) { P.ty[t] = t / 3 - 1; P.tx[t] = t % 3 - 1; }
  P.TH = pl.TH; P.TW = pl.TW; P.tiles_y = pl.tiles_y; P.tiles_x = pl.tiles_x; P.HC = pl.HC; P.HP = pl.HP;
  P.Cs = 0; P.oy = P.ox = 0;
  {
    const long long ib = (long long)in->H * in->W * in->pitch * 2, wb = (long long)9 * pl.Mpad * K * 2;
    P.buf = (ib < (1LL << 31) && wb < (1LL << 31) && gsd_env_int("GSD_BF16_CONV_BUF", 1) != 0) ? 1 : 0;
    P.img_bytes = (unsigned)ib;
    P.wt_bytes = (unsigned)wb;
  }
  P.bias = nullptr;
  P.partials = partials;
  P.ep_scale = ep_scale; P.ep_shift = ep_shift;
  P.bw_y = nullptr; P.bw_pitch = 0; P.bw_scale = P.bw_shift = P.bw_mean = P.bw_invstd = nullptr;
  if (bw != nullptr) {
    if (int e = gsd_check_nhwc(bw->y, "gsd_bf16_conv3x3 bw.y")) return e;
    GSD_REQUIRE(bw->scale && bw->shift && bw->mean && bw->invstd && partials, GSD_ERR_BAD_ARG,
                "gsd_bf16_conv3x3: fused BatchNorm backward needs coefficients and partials");
    GSD_REQUIRE(bw->y->N == out->N && bw->y->H == out->H && bw->y->W == out->W && bw->y->C == M && (bw->y->pitch & 3) == 0,
                GSD_ERR_BAD_ARG, "gsd_bf16_conv3x3: bw.y must have out's geometry");
    P.bw_y = (const u16*)bw->y->ptr; P.bw_pitch = bw->y->pitch;
    P.bw_scale = bw->scale; P.bw_shift = bw->shift; P.bw_mean = bw->mean; P.bw_invstd = bw->invstd;
  }
  const long grid = (long)P.N * pl.tiles_y * pl.tiles_x * pl.mblocks;
  const size_t lds = (size_t)2 * 3 * pl.BM * 64 + (size_t)2 * (pl.wide ? 16 : 10) * 4096 + (size_t)(4 * pl.BM + 512) * sizeof(float);
  if (P.buf) {
    if (pl.wide) return launch<0, 1, 4, 1>(P, grid, lds, (hipStream_t)stream, "gsd_bf16_conv3x3");
    return launch<0, 2, 2, 1>(P, grid, lds, (hipStream_t)stream, "gsd_bf16_conv3x3");
  }
  if (pl.wide) return launch<0, 1, 4>(P, grid, lds, (hipStream_t)stream, "gsd_bf16_conv3x3");
  return launch<0, 2, 2>(P, grid, lds, (hipStream_t)stream, "gsd_bf16_conv3x3");
}

static int conv_dense_impl(const gsd_nhwc* in, const void* wt, const gsd_nhwc* out, int K, int M, int ntaps, int stride,
                           const int* ty, const int* tx, int H, int W, int scatter_cs, int oy, int ox, const float* bias,
                           float* partials, const gsd_bf16_bnbwd* bw, const float* ep_scale, const float* ep_shift, void* stream) {
  if (int e = gsd_check_nhwc(in, "gsd_bf16_conv_dense in")) return e;
  if (int e = gsd_check_nhwc(out, "gsd_bf16_conv_dense out")) return e;
  GSD_REQUIRE(wt != nullptr && ty != nullptr && tx != nullptr, GSD_ERR_BAD_ARG, "gsd_bf16_conv_dense: null argument");
  GSD_REQUIRE(K > 0 && K % 32 == 0 && in->C == K, GSD_ERR_UNSUPPORTED, "gsd_bf16_conv_dense: K=%d must be a multiple of 32 and in->C",
              K);
  GSD_REQUIRE(M > 0 && M % 16 == 0, GSD_ERR_UNSUPPORTED, "gsd_bf16_conv_dense: M=%d must be a multiple of 16", M);
  GSD_REQUIRE(ntaps >= 1 && ntaps <= 9 && (stride == 1 || stride == 2), GSD_ERR_BAD_ARG, "gsd_bf16_conv_dense: bad taps/stride");
  GSD_REQUIRE(H > 0 && W > 0 && H < 65536 && W < 4096 && in->N == out->N, GSD_ERR_BAD_ARG, "gsd_bf16_conv_dense: bad grid");
  GSD_REQUIRE((out->pitch & 3) == 0, GSD_ERR_UNSUPPORTED, "gsd_bf16_conv_dense: out pitch must be a multiple of 4");
  if (scatter_cs > 0) {
    GSD_REQUIRE(M == 4 * scatter_cs && scatter_cs % 16 == 0 && out->C == scatter_cs, GSD_ERR_BAD_ARG,
                "gsd_bf16_conv_dense: scatter needs M == 4*Cs, Cs %% 16 == 0, out->C == Cs");
    GSD_REQUIRE(oy >= 0 && ox >= 0 && 2 * H + oy <= out->H && 2 * W + ox <= out->W, GSD_ERR_BAD_ARG,
                "gsd_bf16_conv_dense: scattered block (%d,%d)+(%d,%d) leaves the output buffer (%d,%d)", 2 * H, 2 * W, oy, ox,
                out->H, out->W);
    GSD_REQUIRE(partials == nullptr, GSD_ERR_UNSUPPORTED, "gsd_bf16_conv_dense: no statistics in scatter mode");
  } else {
    GSD_REQUIRE(out->C == M && H <= out->H && W <= out->W, GSD_ERR_BAD_ARG, "gsd_bf16_conv_dense: out must hold (H,W,M)");
  }
  if (bw != nullptr) {
    if (int e = gsd_check_nhwc(bw->y, "gsd_bf16_conv_dense bw.y")) return e;
    GSD_REQUIRE(scatter_cs == 0 && bias == nullptr && bw->scale && bw->shift && bw->mean && bw->invstd && partials, GSD_ERR_BAD_ARG,
                "gsd_bf16_conv_dense: fused BatchNorm backward needs plain output, coefficients and partials");
    GSD_REQUIRE(bw->y->N == out->N && bw->y->H == out->H && bw->y->W == out->W && out->H == H && out->W == W && bw->y->C == M &&
                    (bw->y->pitch & 3) == 0,
                GSD_ERR_BAD_ARG, "gsd_bf16_conv_dense: bw.y must have out's geometry (and out the GEMM's pixel grid)");
  }
  if (ep_scale == nullptr && (partials == nullptr) == (bw == nullptr) && gsd_ctgemm_shape(in->N, H, W, K, M, ntaps, stride, scatter_cs)) {
    // the transposed convolutions' forward and dX: the large-tile kernel (gsd_bf16_ctgemm.hip)
    if (gsd_ctgemm_operands(in, out, bw, ntaps, ty, tx, H, W))
      return gsd_ctgemm_launch(in, wt, out, K, M, ntaps, ty, tx, H, W, scatter_cs, oy, ox, bias, partials, bw, stream);
    GSD_REQUIRE(partials == nullptr, GSD_ERR_UNSUPPORTED,
                "gsd_bf16_conv_dense: taps that leave the buffer with fused statistics (gsd_bf16_conv_dense_partial_rows counted the large-tile kernel's rows)");
  }
  const Plan pl = make_plan(H, W, M);
  GConvP P;
  P.in = (const u16*)in->ptr; P.in_pitch = in->pitch; P.Hin = in->H; P.Win = in->W;
  P.wt = (const u16*)wt;
  P.out = (u16*)out->ptr; P.out_pitch = out->pitch; P.Hob = out->H; P.Wob = out->W;
  P.N = in->N; P.H = H; P.W = W;
  P.K = K; P.M = M; P.Mpad = pl.Mpad; P.mblocks = pl.mblocks;
  P.ntaps = ntaps; P.stride = stride;
  for (int t = 0; t < 9; ++t) { P.ty[t] = t < ntaps ? ty[t] : 0; P.tx[t] = t < ntaps ? tx[t] : 0; }
  P.TH = pl.TH; P.TW = pl.TW; P.tiles_y = pl.tiles_y; P.tiles_x = pl.tiles_x; P.HC = 0; P.HP = 0;
  P.Cs = scatter_cs; P.oy = oy; P.ox = ox;
  P.buf = 0; P.img_bytes = P.wt_bytes = 0;
  P.bias = bias;
  P.partials = partials;
  P.ep_scale = ep_scale; P.ep_shift = ep_shift;
  P.bw_y = nullptr; P.bw_pitch = 0; P.bw_scale = P.bw_shift = P.bw_mean = P.bw_invstd = nullptr;
  if (bw != nullptr) {
    P.bw_y = (const u16*)bw->y->ptr; P.bw_pitch = bw->y->pitch;
    P.bw_scale = bw->scale; P.bw_shift = bw->shift; P.bw_mean = bw->mean; P.bw_invstd = bw->invstd;
  }
  const long grid = (long)P.N * pl.tiles_y * pl.tiles_x * pl.mblocks;
  const int nsub = pl.wide ? 1 : 2;   // k-steps per barrier (kernel: NSUB)
  GSD_REQUIRE(K % (32 * nsub) == 0, GSD_ERR_UNSUPPORTED, "gsd_bf16_conv_dense: K=%d must be a multiple of 64 when M > 64", K);
  const size_t lds = (size_t)2 * nsub * pl.BM * 64 + (size_t)2 * nsub * pl.NPX * 96 + (size_t)(4 * pl.BM + 512) * sizeof(float);
  if (pl.wide) return launch<1, 1, 4>(P, grid, lds, (hipStream_t)stream, "gsd_bf16_conv_dense");
  return launch<1, 2, 2>(P, grid, lds, (hipStream_t)stream, "gsd_bf16_conv_dense");
}

extern "C" int gsd_bf16_conv_dense_partial_rows(int N, int H, int W, int K, int M, int ntaps, int stride) {
  if (gsd_ctgemm_shape(N, H, W, K, M, ntaps, stride, 0)) return gsd_ctgemm_partial_rows(N, H, W, M);
  return gsd_bf16_conv_partial_rows(N, H, W, M);
}

extern "C" int gsd_bf16_conv3x3(const gsd_nhwc* in, const void* wt, const gsd_nhwc* out, int K, int M, float* partials,
                                const gsd_bf16_bnbwd* bw, void* stream) {
  return conv3x3_impl(in, wt, out, K, M, partials, bw, nullptr, nullptr, stream);
}

extern "C" int gsd_bf16_conv_dense(const gsd_nhwc* in, const void* wt, const gsd_nhwc* out, int K, int M, int ntaps, int stride,
                                   const int* ty, const int* tx, int H, int W, int scatter_cs, int oy, int ox, const float* bias,
                                   float* partials, const gsd_bf16_bnbwd* bw, void* stream) {
  return conv_dense_impl(in, wt, out, K, M, ntaps, stride, ty, tx, H, W, scatter_cs, oy, ox, bias, partials, bw, nullptr, nullptr, stream);
}

extern "C" int gsd_bf16_conv3x3_bnrelu(const gsd_nhwc* in, const void* wt, const gsd_nhwc* out, int K, int M, const float* scale,
                                       const float* shift, void* stream) {
  GSD_REQUIRE(scale && shift, GSD_ERR_BAD_ARG, "gsd_bf16_conv3x3_bnrelu: null coefficients");
  return conv3x3_impl(in, wt, out, K, M, nullptr, nullptr, scale, shift, stream);
}

extern "C" int gsd_bf16_conv1x1_bnrelu(const gsd_nhwc* in, const void* wt, const gsd_nhwc* out, int K, int M, const float* scale,
                                       const float* shift, void* stream) {
  GSD_REQUIRE(scale && shift && in && out, GSD_ERR_BAD_ARG, "gsd_bf16_conv1x1_bnrelu: null argument");
  const int z = 0;
  return conv_dense_impl(in, wt, out, K, M, 1, 1, &z, &z, out->H, out->W, 0, 0, 0, nullptr, nullptr, nullptr, scale, shift, stream);
}
