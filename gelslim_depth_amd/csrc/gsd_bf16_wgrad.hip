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
  unsigned a_img_bytes, b_img_bytes;   // one image of each operand (buffer descriptors; below 2 GiB)
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
    // Fills through buffer descriptors (buffer_load_dwordx4 ... lds): one 32-bit byte offset per piece from the image's base, and a
    // piece with nothing to fetch (outside the image, a channel beyond M / Ncols) is an offset beyond num_records --
    // the hardware writes zeros for it (profiles/ubench/buffer_lds_oob.hip): no zero line, no 64-bit address, no branch per piece.
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(P.a + (long long)n * P.H * P.W * P.a_pitch), 0, P.a_img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(P.b + (long long)n * P.Hb * P.Wb * P.b_pitch), 0, P.b_img_bytes, 0x00020000);
    __syncthreads();   // every wave has finished reading the previous stage's images
#pragma unroll
    for (int k = 0; k < MAXA; ++k) {
      unsigned v = 0x80000000u;
      if (apk[k] >= 0) {
        const int h = h0 + (apk[k] >> 20), w = w0 + ((apk[k] >> 8) & 0xfff);
        if (h < P.H && w < P.W) v = (unsigned)(((h * P.W + w) * (int)P.a_pitch + m0 + (apk[k] & 0xff) * 8) * 2);
      }
      if (apk[k] != -1)   // (-1: this lane's 16 bytes are row padding or lie behind the image -- inside the B image: not touched)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(Al + (k * 4 + wave) * 1024), 16, v, 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < MAXB; ++k) {
      unsigned v = 0x80000000u;
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
          v = (unsigned)(((hi * P.Wb + wi) * (int)P.b_pitch + n0 + (bpk[k] & 0xff) * 8) * 2);
      }
      if (bpk[k] != -1)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(Bl + (k * 4 + wave) * 1024), 16, v, 0, 0, 0);
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

// ---- dense 4-tap form on a large tile (the transposed convolutions' dW) -------------------------------------------------------
// With no halo to share, the 128 x (32 x 4 taps) tile above moves 32 KB into LDS for 32 MFMAs per wave (its fills cost as much
// as its arithmetic: 0.16-0.25 PFLOP/s).  Here: eight waves, tile 256 m x (64 n x 4 taps) (or 128 m when M == 128), stages of 64
// LINEAR pixels (no 2-D tile padding; the stride-2 gather decodes (n,h,w) per fill row), two LDS images: the next stage's
// fill flies during this stage's 64 MFMAs per wave.  Same transposed-read layout (rows of 32 B x odd), same slab / reduce.
struct GWBigP {
  const u16* a;
  int a_pitch;
  const u16* b;
  int b_pitch, Hb, Wb;
  float* slabs;
  int P, H, W;          // pixels of the reduction (N*H*W) and the grid they come from
  float rW, rHW;
  int M, Ncols;
  int ty[4], tx[4];
  int stages_total, splits, mblocks, nblocks, xcd;
};

template <int WM, int WN>
__global__ __launch_bounds__(512) void gwgrad_big_bf16_kernel(const GWBigP P) {
  static_assert(WM * WN == 8, "eight waves");
  constexpr int BM = WM * 64, NTW = 4 / WN;   // n-tiles per wave: 64 n / WN / 16
  constexpr int BNC = 64, TT = 4, MT = 4, NPX = 64, KS = 2;
  constexpr int RSA = BM * 2 + 32, RSB = BNC * 2 + 32;     // 32 B x odd
  constexpr int ABYTES = NPX * RSA, BBYTES = TT * NPX * RSB;
  constexpr int NA = (ABYTES + 8191) / 8192, NB = (BBYTES + 8191) / 8192;   // DMA instructions per wave (8 waves x 1 KiB)
  constexpr int IMG = ((ABYTES + 1023) / 1024 + (BBYTES + 1023) / 1024) * 1024;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int g = lane >> 4, li = lane & 15, q = li >> 2, p4 = li & 3;

  const int per_split = P.mblocks * P.nblocks;
  const int lid = P.xcd ? xcd_swizzle(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int split = lid / per_split;
  const int rem = lid - split * per_split;
  const int mb = rem % P.mblocks, nb = rem / P.mblocks;
  const int m0 = mb * BM, n0 = nb * BNC;
  const int s_begin = (int)((long long)split * P.stages_total / P.splits);
  const int s_end = (int)((long long)(split + 1) * P.stages_total / P.splits);

  // ---- DMA bookkeeping: instruction k of this wave fills the 1 KiB piece k*8 + wave of an image; -1: nothing to move (row pad) --
  int arow[NA], aoffc[NA];
#pragma unroll
  for (int k = 0; k < NA; ++k) {
    const int o = (k * 8 + wave) * 1024 + lane * 16;
    const int row = o / RSA, piece = (o - row * RSA) >> 4;
    const bool ok = row < NPX && piece < BM / 8;
    arow[k] = ok ? row : -1;
    aoffc[k] = row * P.a_pitch + m0 + piece * 8;
  }
  // B rows are a stride-2 gather: element offset of pixel p's 2 x 2 block = ((n*Hb + 2h)*Wb + 2w)*pitch.  Decoding (n,h,w) per fill
  // instruction costs more vector issue than the stage's MFMAs leave free, so one wave decodes a stage's 64 pixels ONCE into an
  // LDS table (two stages ahead of its use, two slots) and a fill instruction adds its lane-constant part: tap, columns, piece.
  int* sTab = reinterpret_cast<int*>(smem + 2 * IMG);   // [2][64]
  int bpx[NB], bofc[NB];   // table index of the row's pixel (-1: nothing to move), and (ty*Wb + tx)*pitch + n0 + piece*8
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    const int o = (k * 8 + wave) * 1024 + lane * 16;
    const int row = o / RSB, piece = (o - row * RSB) >> 4;
    const bool ok = row < TT * NPX && piece < BNC / 8;
    const int t = ok ? row / NPX : 0;
    bpx[k] = ok ? row % NPX : -1;
    bofc[k] = (P.ty[t] * P.Wb + P.tx[t]) * P.b_pitch + n0 + piece * 8;
  }
  auto decode_stage = [&](int stage) {   // one wave: lane = pixel of the stage
    const int p = min(stage * NPX + lane, P.P - 1);   // beyond the grid A is zero: any valid B row does
    const int HW = P.H * P.W;
    int n = (int)((float)p * P.rHW);
    int r = p - n * HW;
    if (r < 0) { --n; r += HW; } else if (r >= HW) { ++n; r -= HW; }
    int h = (int)((float)r * P.rW);
    int w = r - h * P.W;
    if (w < 0) { --h; w += P.W; } else if (w >= P.W) { ++h; w -= P.W; }
    sTab[(stage & 1) * 64 + lane] = ((n * P.Hb + 2 * h) * P.Wb + 2 * w) * P.b_pitch;
  };
  constexpr int AOFF = 0, BOFF = ((ABYTES + 1023) / 1024) * 1024;
  auto dma_a = [&](int k, int stage, int buf) {
    if (arow[k] < 0) return;
    const int p0 = stage * NPX;
    const void* s = (const void*)gsd_zero16w;
    if (p0 + arow[k] < P.P) s = (const void*)(P.a + (long long)p0 * P.a_pitch + aoffc[k]);
    gsd_dma16_untracked(s, smem + buf * IMG + AOFF + (k * 8 + wave) * 1024);   // (untracked: see gsd_bf16_common.h)
  };
  auto dma_b = [&](int k, int toff, int buf) {   // toff: the table entry of the row's pixel
    if (bpx[k] < 0) return;
    gsd_dma16_untracked((const void*)(P.b + toff + bofc[k]), smem + buf * IMG + BOFF + (k * 8 + wave) * 1024);
  };
  auto table = [&](int k, int stage) { return bpx[k] >= 0 ? sTab[(stage & 1) * 64 + bpx[k]] : 0; };

  const int a_rd = (4 * g + q) * RSA + (wm * 64 + 4 * p4) * 2;            // + ks*32*RSA + m*32 (+ 16*RSA)
  const int b_rd = BOFF + (4 * g + q) * RSB + (wn * NTW * 16 + 4 * p4) * 2;   // + (t*NPX + ks*32)*RSB + nt*32 (+ 16*RSB)

  f32x4 acc[MT][TT * NTW];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < TT * NTW; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (s_begin < s_end) {
    if (wave == 0) decode_stage(s_begin);
    if (wave == 1 && s_begin + 1 < s_end) decode_stage(s_begin + 1);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NA; ++k) dma_a(k, s_begin, 0);
#pragma unroll
    for (int k = 0; k < NB; ++k) dma_b(k, table(k, s_begin), 0);
  }
  for (int stage = s_begin; stage < s_end; ++stage) {
    const int buf = (stage - s_begin) & 1;
    gsd_dma_barrier();   // this stage has landed, every wave has left the other image, the table of stage + 1 is published
    const unsigned char* Im = smem + buf * IMG;
    const int fst = stage + 1 < s_end ? stage + 1 : stage;   // after the last stage: a fill nobody reads (no branch)
    int toff[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) toff[k] = table(k, fst);
    if (wave == 0 && stage + 2 < s_end) decode_stage(stage + 2);   // into the slot whose readers (the fills of `stage`) are done
    auto read_a = [&](int ks, u32x4* a) {
      const unsigned char* ap = Im + a_rd + ks * 32 * RSA;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const u32x2 lo = tr_read_b64(ap + m * 32), hi = tr_read_b64(ap + m * 32 + 16 * RSA);
        a[m] = u32x4{lo[0], lo[1], hi[0], hi[1]};
      }
    };
    auto read_b = [&](int st) {   // st = (ks, t, nt)
      const int ks = st / (TT * NTW), tn = st - ks * (TT * NTW), t = tn / NTW, nt = tn - t * NTW;
      const unsigned char* bp = Im + b_rd + (t * NPX + ks * 32) * RSB + nt * 32;
      const u32x2 lo = tr_read_b64(bp), hi = tr_read_b64(bp + 16 * RSB);
      return u32x4{lo[0], lo[1], hi[0], hi[1]};
    };
    constexpr int NST = KS * TT * NTW;          // steps of 4 MFMAs
    constexpr int NSL = NA + NB;                // DMA slots of the next stage's fill
    static_assert(NST >= NSL, "a fill slot per step at most");
    constexpr int GAP = NST / NSL;
    u32x4 a[2][MT], b[3];
    read_a(0, a[0]);
    b[0] = read_b(0);
    b[1] = read_b(1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      const int ks = st / (TT * NTW), tn = st - ks * (TT * NTW);
      if (st + 2 < NST) b[(st + 2) % 3] = read_b(st + 2);
      if (tn == TT * NTW - 2 && ks + 1 < KS) read_a(ks + 1, a[(ks + 1) & 1]);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m][tn] = mfma_bf16(a[ks & 1][m], b[st % 3], acc[m][tn]);
      if (st % GAP == GAP - 1) {   // one slot of the next stage's fill
        const int sl = st / GAP;
        if (sl < NA) dma_a(sl, fst, buf ^ 1);
        else if (sl < NSL) dma_b(sl - NA, toff[sl - NA], buf ^ 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  gsd_dma_barrier();   // the last (unread) fill must have landed before the block gives its LDS back

  // ---- slab store: slab[split][t][m][n] ------------------------------------------------------------------------------
#pragma unroll
  for (int tn = 0; tn < TT * NTW; ++tn) {
    const int t = tn / NTW, nt = tn - t * NTW;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int mr = m0 + wm * 64 + m * 16 + g * 4 + e;
        const int col = n0 + wn * NTW * 16 + nt * 16 + li;
        if (mr < P.M && col < P.Ncols) P.slabs[(((size_t)split * TT + t) * P.M + mr) * P.Ncols + col] = acc[m][tn][e];
      }
  }
}

// out[(m * NcOut + n) * T + t] = sum over splits of slab[split][t][m][n]   (n < NcOut <= Ncols)
// SG > 1: the splits of an element are summed by SG threads (split k goes to thread k % SG, each adds its own in ascending order,
// the SG partial sums are added in ascending order): a fixed order, so still bitwise reproducible.  With hundreds of splits of
// a small dW (64 x 64 x 9: 36,864 elements, 512 splits) one thread per element is 144 blocks each walking 512 dependent loads.
template <int SG>
__global__ __launch_bounds__(256) void gwgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, int splits, int T, int M,
                                                            int Ncols, int NcOut) {
  constexpr int EPB = 256 / SG;   // elements per block
  __shared__ float part[SG][EPB];
  const long long per = (long long)T * M * Ncols;
  const int el = threadIdx.x % EPB, sg = threadIdx.x / EPB;
  for (long long e0 = (long long)blockIdx.x * EPB; e0 < per; e0 += (long long)gridDim.x * EPB) {
    const long long e = e0 + el;
    float s = 0.f;
    if (e < per) {
      float s4[4] = {0.f, 0.f, 0.f, 0.f};   // four loads in flight; combined in a fixed order
      int k = sg;
      for (; k + 3 * SG < splits; k += 4 * SG) {
#pragma unroll
        for (int u = 0; u < 4; ++u) s4[u] += slabs[(size_t)(k + u * SG) * per + e];
      }
      for (; k < splits; k += SG) s4[0] += slabs[(size_t)k * per + e];
      s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    }
    if (SG > 1) {
      part[sg][el] = s;
      __syncthreads();
      if (sg == 0) {
#pragma unroll
        for (int g = 1; g < SG; ++g) s += part[g][el];
      }
    }
    if (sg == 0 && e < per) {
      const int n = (int)(e % Ncols);
      const long long r2 = e / Ncols;
      const int m = (int)(r2 % M);
      const int t = (int)(r2 / M);
      if (n < NcOut) out[((size_t)m * NcOut + n) * T + t] = s;
    }
    if (SG > 1) __syncthreads();
  }
}

// Large dW (>= 64 k (m, n) pairs): a thread owns one (m, n) and all T taps -- its T results are consecutive in the output
// ([m][n][t]: a wave writes one contiguous run instead of 4-byte pieces T floats apart), the slab reads stay coalesced over n.
// Same order of additions per element as gwgrad_reduce_kernel<1>.
template <int T>
__global__ __launch_bounds__(256) void gwgrad_reduce_mn_kernel(const float* __restrict__ slabs, float* __restrict__ out, int splits, int M,
                                                               int Ncols, int NcOut) {
  const long long mn = (long long)M * Ncols, per = (long long)T * mn;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < mn; e += (long long)gridDim.x * 256) {
    const int n = (int)(e % Ncols), m = (int)(e / Ncols);
    float r[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const float* sp = slabs + (size_t)t * mn + e;
      float s4[4] = {0.f, 0.f, 0.f, 0.f};
      int k = 0;
      for (; k + 3 < splits; k += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) s4[u] += sp[(size_t)(k + u) * per];
      }
      for (; k < splits; ++k) s4[0] += sp[(size_t)k * per];
      r[t] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    }
    if (n < NcOut) {
      float* o = out + ((size_t)m * NcOut + n) * T;
#pragma unroll
      for (int t = 0; t < T; ++t) o[t] = r[t];
    }
  }
}

// the reduce launch: many splits of a small dW -> 8 threads per element
static void launch_reduce(const float* slabs, float* out, int splits, int T, int M, int Ncols, int NcOut, hipStream_t st) {
  const long long per = (long long)T * M * Ncols;
  if (splits >= 32 && per <= (1 << 20)) {
    const int grid = (int)(ceil_div64(per, 32) < 8192 ? ceil_div64(per, 32) : 8192);
    hipLaunchKernelGGL(gwgrad_reduce_kernel<8>, dim3(grid), dim3(256), 0, st, slabs, out, splits, T, M, Ncols, NcOut);
  } else if ((long long)M * Ncols >= 65536 && (T == 9 || T == 4)) {
    const int grid = (int)(ceil_div64((long long)M * Ncols, 256) < 4096 ? ceil_div64((long long)M * Ncols, 256) : 4096);
    if (T == 9) hipLaunchKernelGGL(gwgrad_reduce_mn_kernel<9>, dim3(grid), dim3(256), 0, st, slabs, out, splits, M, Ncols, NcOut);
    else hipLaunchKernelGGL(gwgrad_reduce_mn_kernel<4>, dim3(grid), dim3(256), 0, st, slabs, out, splits, M, Ncols, NcOut);
  } else {
    const int grid = (int)(ceil_div64(per, 256) < 4096 ? ceil_div64(per, 256) : 4096);
    hipLaunchKernelGGL(gwgrad_reduce_kernel<1>, dim3(grid), dim3(256), 0, st, slabs, out, splits, T, M, Ncols, NcOut);
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

namespace {

struct BigPlan {
  bool ok;
  int BM, mblocks, nblocks, stages_total, splits;
  int64_t slab_elems;
};

// the large-tile form takes 4 dense taps with M % 128 == 0 and Ncols % 64 == 0 (GSD_BF16_WGRAD_BIG=0: never)
BigPlan make_bigplan(int ntaps, int N, int H, int W, int M, int Ncols) {
  BigPlan p;
  p.ok = false;
  p.slab_elems = 0;
  const long long P = (long long)N * H * W;
  if (ntaps != 4 || M % 128 != 0 || Ncols % 64 != 0 || P >= (1 << 24) || gsd_env_int("GSD_BF16_WGRAD_BIG", 1) == 0) return p;
  p.BM = M % 256 == 0 ? 256 : 128;
  p.mblocks = M / p.BM;
  p.nblocks = Ncols / 64;
  p.stages_total = (int)((P + 63) / 64);
  int cus = 256, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  int splits = cus / (p.mblocks * p.nblocks);   // one block per CU (its two LDS images fill the CU)
  if (splits < 1) splits = 1;
  if (splits > p.stages_total) splits = p.stages_total;
  p.splits = splits;
  p.slab_elems = (int64_t)splits * 4 * M * Ncols;
  p.ok = true;
  return p;
}

template <int WM, int WN>
int launch_big(const GWBigP& P, int grid, hipStream_t st) {
  constexpr int BM = WM * 64;
  constexpr size_t img = ((size_t)(64 * (BM * 2 + 32) + 1023) / 1024 + (size_t)(4 * 64 * (64 * 2 + 32) + 1023) / 1024) * 1024;
  static gsd_attr_once big_lds;
  if (hipError_t e = gsd_allow_big_lds(big_lds, reinterpret_cast<const void*>(&gwgrad_big_bf16_kernel<WM, WN>)); e != hipSuccess) {
    gsd_set_error("gsd_bf16_wgrad (large tile): hipFuncSetAttribute: %s", hipGetErrorString(e));
    return GSD_ERR_HIP;
  }
  GSD_REQUIRE(2 * img + 512 <= 160 * 1024, GSD_ERR_UNSUPPORTED, "gsd_bf16_wgrad (large tile): LDS %zu B too large", 2 * img + 512);
  hipLaunchKernelGGL((gwgrad_big_bf16_kernel<WM, WN>), dim3(grid), dim3(512), 2 * img + 512, st, P);
  GSD_LAUNCH_CHECK("gsd_bf16_wgrad (large tile)");
  return GSD_OK;
}

}  // namespace

extern "C" int64_t gsd_bf16_wgrad_workspace(int ntaps, int N, int H, int W, int M, int Ncols) {
  if (ntaps < 1 || ntaps > 9 || N <= 0 || H <= 0 || W <= 0 || M <= 0 || Ncols <= 0) return 0;
  const int64_t a = make_wplan(ntaps == 9, ntaps, N, H, W, M, Ncols).slab_elems, b = make_bigplan(ntaps, N, H, W, M, Ncols).slab_elems;
  return a > b ? a : b;   // (the large-tile form also needs stride 2 and in-buffer taps: whichever form runs, this is enough)
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
  const BigPlan bp = make_bigplan(ntaps, a->N, a->H, a->W, M, Ncols);
  if (bp.ok && stride == 2 && (long long)a->N * a->H * a->W * a->pitch < 2147483647LL && (long long)b->N * b->H * b->W * b->pitch < 2147483647LL) {
    bool inside = true;   // the large-tile form gathers without a zero line: every tap of every pixel must be in b's buffer
    for (int t = 0; t < 4; ++t) inside = inside && ty[t] >= 0 && tx[t] >= 0 && 2 * (a->H - 1) + ty[t] < b->H && 2 * (a->W - 1) + tx[t] < b->W;
    if (inside) {
      GSD_REQUIRE(workspace_elems >= bp.slab_elems, GSD_ERR_WORKSPACE, "gsd_bf16_wgrad: workspace %lld < %lld elements",
                  (long long)workspace_elems, (long long)bp.slab_elems);
      GWBigP Q;
      Q.a = (const u16*)a->ptr; Q.a_pitch = (int)a->pitch;
      Q.b = (const u16*)b->ptr; Q.b_pitch = (int)b->pitch; Q.Hb = b->H; Q.Wb = b->W;
      Q.slabs = workspace;
      Q.P = a->N * a->H * a->W; Q.H = a->H; Q.W = a->W;
      Q.rW = 1.0f / (float)a->W; Q.rHW = 1.0f / (float)(a->H * a->W);
      Q.M = M; Q.Ncols = Ncols;
      for (int t = 0; t < 4; ++t) { Q.ty[t] = ty[t]; Q.tx[t] = tx[t]; }
      Q.stages_total = bp.stages_total; Q.splits = bp.splits; Q.mblocks = bp.mblocks; Q.nblocks = bp.nblocks;
      Q.xcd = gsd_env_int("GSD_BF16_XCD", 1) != 0 ? 1 : 0;
      const int grid = bp.splits * bp.mblocks * bp.nblocks;
      const int rc = bp.BM == 256 ? launch_big<4, 2>(Q, grid, (hipStream_t)stream) : launch_big<2, 4>(Q, grid, (hipStream_t)stream);
      if (rc) return rc;
      launch_reduce(workspace, dw, bp.splits, 4, M, Ncols, ncols_out, (hipStream_t)stream);
      GSD_LAUNCH_CHECK("gsd_bf16_wgrad reduce");
      return GSD_OK;
    }
  }
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
  GSD_REQUIRE((long long)a->H * a->W * a->pitch < (1LL << 30) && (long long)b->H * b->W * b->pitch < (1LL << 30), GSD_ERR_UNSUPPORTED,
              "gsd_bf16_wgrad: one image of an operand exceeds 2 GiB");
  P.a_img_bytes = (unsigned)((long long)a->H * a->W * a->pitch * 2);
  P.b_img_bytes = (unsigned)((long long)b->H * b->W * b->pitch * 2);
  const int grid = pl.splits * pl.mblocks * pl.nblocks;
  int rc;
  hipStream_t st = (hipStream_t)stream;
  if (halo) rc = pl.wide ? launch_w<1, 9, 1, 4>(P, grid, pl.lds, st, "gsd_bf16_wgrad") : launch_w<1, 9, 2, 2>(P, grid, pl.lds, st, "gsd_bf16_wgrad");
  else if (ntaps == 4) rc = pl.wide ? launch_w<0, 4, 1, 4>(P, grid, pl.lds, st, "gsd_bf16_wgrad") : launch_w<0, 4, 2, 2>(P, grid, pl.lds, st, "gsd_bf16_wgrad");
  else rc = pl.wide ? launch_w<0, 1, 1, 4>(P, grid, pl.lds, st, "gsd_bf16_wgrad") : launch_w<0, 1, 2, 2>(P, grid, pl.lds, st, "gsd_bf16_wgrad");
  if (rc) return rc;
  launch_reduce(workspace, dw, pl.splits, ntaps, M, Ncols, ncols_out, (hipStream_t)stream);
  GSD_LAUNCH_CHECK("gsd_bf16_wgrad reduce");
  return GSD_OK;
}
