// gsd_wgrad_w2d.hip -- dW of conv3x3 with the transposed TWO-dimensional Winograd identity F(2x4, 3x3) (gfx950).
//
//   dW[co][ci][r][s] = sum_{n,h,w} dy[n,co,h,w] * a[n,ci,h+r-1,w+s-1]
//
// (the dW half of aten::convolution_backward for /root/reference/gelslim_depth/models/unet.py:11,14).  The forward identity of
// gsd_conv3x3_w2d.hip, Y = A2^T[(G2 g G4^T) .* (B2^T d B4)]A4, is linear in g, so for every tile of 2 x 4 outputs
//
//   dg = G2^T [ (A2 dY A4^T) .* (B2^T d B4) ] G4
//
// with the forward's input transform of the 4 x 6 window d, dY (2 x 4) transformed to 4 x 6, and G2^T . G4 applied once at the
// very end: 24 products per (co, ci) and 8 pixels instead of 36 in the row form (gsd_wgrad_w43.hip) and 72 in the direct one.
//
//   D_f[co][ci] = sum_tiles U_f[co][tile] * V_f[tile][ci]      f = (fr, fc), 24 frequencies
//
// GEMM view: M = co, N = ci, K = tiles (4 per v_mfma_f32_16x16x4_f32).  What made this form lose when every wave transformed its
// own operands (7 vector instructions per MFMA at a 16 x 16 wave tile, 264 registers at 32 x 16) is the
// transform work, so here it is done ONCE PER BLOCK and shared through LDS:
//
//   * a block of 8 waves owns BM x BN = 128 co x 32 ci (or 64 x 64 for the 64-channel layers); a k-step is 4 tiles (32 pixels:
//     1 x 4, 2 x 2 or 4 x 1 tiles, chosen per shape); one block per CU, split-K over k-steps;
//   * per k-step every thread takes one or two small transform TASKS -- "U" (one (co, tile): two aligned 16-byte pieces of the
//     row-pitched dy -> A4 along the rows, A2 down the columns -> 24 values) and "V row" (one window row of a (ci, tile): 6 floats ->
//     deferred BatchNorm+ReLU -> B4^T; the B2^T column transform takes the partner row from the lane's quad by DPP) -- in packed
//     fp32 math, and stores the results as [tile][channel][24 frequencies] images;
//   * the raw values of the NEXT k-step wait in LDS, moved by LDS-DMA: dy pieces in per-thread private slots, the activation windows
//     as one image per block and k-step (the halo shared by its tiles, filled as runs of consecutive 16-byte pieces); nothing raw is
//     held in registers across MFMAs (192 accumulators + one piece's values is all that fits);
//   * a wave's 48 MFMAs of k-step i (tile 32 co x 16 ci x 24 frequencies = 192 accumulator registers; operands frequency-major: one
//     ds_read_b128 is four frequencies of (channel l16, tile j), 18 reads per 48 MFMAs, no vector work) run in six groups with the
//     transform of k-step i+1 in PIECES between them -- a software pipeline inside the wave: with separate phases all eight waves of
//     the CU sat in the same phase and the matrix pipes were 0.53 busy (docs/LOG_r06.md);
//   * two transform images and two window images (k-step parity), one barrier per k-step.
//
// ~1.9 vector instructions per MFMA including addressing (row form: 2.7-3.3), on two thirds of the row form's MFMAs; 23.6 ms for
// the network's 17 layers at batch 32 against 29.5.  The kernel lives at the register limit and must compile to ZERO scratch
// (tests/test_abi.py): with spills hipcc stored two slots on some paths of a branchy prologue only and reloaded them on all.
// Split-K over k-steps with ordered slab reduction (the row form's reducer and slab layout): bitwise reproducible.
//
// Conventions: U row 3 is +dY row 1 and V row 3 is d3 - d1 (both signs of the textbook F(2,3) flipped: same products); the six
// frequencies of a row are stored in the order [1, 2, 3, 4, 0, 5] (what the packed transforms produce as register pairs).
#include "gsd_common.h"
#include <type_traits>

#include <cstdio>
#include <cstdlib>

typedef float f32x2d __attribute__((ext_vector_type(2)));

// Diagnostic builds only (-DWG2D_ABL=mask via profiles/build_diag_one.sh; never in the product library; results are then garbage):
// 1 no MFMAs, 2 no LDS-DMA fills after the first two k-steps, 4 no transform after the first two, 8 no operand reads, 16 no barrier,
// 32 no dy fills, 64 no window fills, 128 every fill reads offset 0 (issue cost without the memory system's).  Any non-zero mask
// also runs the phase-separated loop (transform, fills, MFMAs one after the other) instead of the pipelined one.
#ifndef WG2D_ABL
#define WG2D_ABL 0
#endif

struct WgW2dParams {
  SrcD a0, a1;   // activation (B operand), up to two concatenated segments
  SrcD dy;       // gradient w.r.t. the raw conv output (plain, row pitch % 4 == 0, 16-byte aligned)
  float* slabs;  // [split][9 = r*3+s][M][Ncols]: G2^T . G4 applied per split
  int M, Ncols;
  int N, H, W;
  int KY, KX, kx_log2;   // tiles of a k-step: KY x KX == 4
  int tiles_y, tiles_x, sy_n, sx_n;
  int WR, NP, NI;        // window of a k-step: 2 KY + 2 rows x KX + 1 pieces of 4 floats per channel; NI 64-piece fills per block
  int ksteps_total, splits, mblocks, nblocks;
};

__device__ __forceinline__ float w2d_dpp(float v, const int ctrl_is_pair) {
  // quad_perm [2,2,1,1] (0x5A): V column transform partners; quad_perm [1,0,3,2] (0xB1): the other row of a dy pair
  return ctrl_is_pair ? __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true))
                      : __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x5A, 0xf, 0xf, true));
}

// NWM x NWN waves of 32 co x 16 ci.  (4,2): 128 co x 32 ci -- U tasks: one (co, tile) per thread, V row tasks: one per thread.
// (2,4): 64 co x 64 ci -- U tasks split by dy row (the A2 column transform takes the other row by DPP), two V row tasks per thread.
// PLAIN: no activation segment carries a deferred BatchNorm / ReLU.
template <int NWM, int NWN, bool PLAIN>
__global__ __launch_bounds__(512, 1) void wgrad3x3_w2d_kernel(const WgW2dParams P) {
  static_assert(NWM * NWN == 8, "8-wave blocks");
  constexpr int BM = 32 * NWM, BN = 16 * NWN;
  constexpr bool UROW = BM == 64;
  constexpr int NV = BN / 32;
  constexpr int TSU = BM * 24 + 4, TSV = BN * 24 + 4;   // tile strides: an odd number of 16-byte slots (conflict-free b128 reads)
  constexpr int BUF = 4 * TSU + 4 * TSV;
  constexpr int NPD = UROW ? 1 : 2;               // dy pieces per thread and k-step (private slots: [piece][wave][64 lanes x 4 floats])
  constexpr int WINI = BN == 32 ? 10 : 20;        // window image of the block's BN channels: at most BN * 20 pieces = WINI fills of 1 KiB
  constexpr int KB = BN == 32 ? 2 : 3;            // window fills per wave at most
  constexpr int WIN = NPD * 8 * 256;              // first float of the window image
  constexpr int IMG = WIN + 2 * WINI * 256;       // first float of image 0 (two window images: k-step parity, like the images)
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int j = lane >> 4, l16 = lane & 15;

  const int lid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int per_split = P.mblocks * P.nblocks;
  const int split = lid / per_split;
  const int rem = lid - split * per_split;
  const int mb = rem % P.mblocks, nb = rem / P.mblocks;
  const int m0 = mb * BM, n0 = nb * BN;
  const int s_begin = (int)((long long)split * P.ksteps_total / P.splits);
  const int s_end = (int)((long long)(split + 1) * P.ksteps_total / P.splits);
  const int nst = s_end - s_begin;

  // ---- the block's activation segment (the host guarantees that its BN channels lie in one) ---------------------------------
  const bool seg1 = P.a1.C > 0 && n0 >= P.a0.C;
  const float* const S_p = seg1 ? P.a1.p : P.a0.p;
  const long long S_ns = seg1 ? P.a1.ns : P.a0.ns, S_cs = seg1 ? P.a1.cs : P.a0.cs;
  const int S_H = seg1 ? P.a1.H : P.a0.H, S_W = seg1 ? P.a1.W : P.a0.W, S_ws = seg1 ? P.a1.ws : P.a0.ws;
  const int S_oh = seg1 ? P.a1.oh : P.a0.oh, S_ow = seg1 ? P.a1.ow : P.a0.ow;
  const int S_c0 = n0 - (seg1 ? P.a0.C : 0);   // first channel of the block inside the segment

  // ---- transform tasks of this thread -----------------------------------------------------------------------------------------
  // (the tile coordinates inside the k-step are needed again in border k-steps only: packed into one register there)
  const int kxm = P.KX - 1;
  // U: (co, tile[, dy row])
  const int u_t = UROW ? (tid >> 1) & 3 : tid & 3;
  const int u_r = UROW ? tid & 1 : 0;
  const int u_co = UROW ? tid >> 3 : tid >> 2;
  const int u_tyl = u_t >> P.kx_log2, u_txl = u_t & kxm;
  const unsigned u_voff = (unsigned)((long long)u_co * P.dy.cs + (long long)(2 * u_tyl + u_r) * P.dy.ws + 4 * u_txl) * 4u;
  const unsigned u_wr = (unsigned)(IMG + u_t * TSU + u_co * 24) * 4u;                // byte offsets of the thread's results in image 0
  // V rows: (ci, tile, window row kr); lane quad = the four rows of one (ci, tile)
  const int v_kr = tid & 3, v_t = (tid >> 2) & 3, v_ci = tid >> 4;
  const int v_tyl = v_t >> P.kx_log2, v_txl = v_t & kxm;
  const unsigned v_rd = (unsigned)(WIN + ((v_ci * P.WR + 2 * v_tyl + v_kr) * P.NP + v_txl) * 4) * 4u;   // its 6 floats in the window image
  const unsigned v_wr = (unsigned)(IMG + 4 * TSU + v_t * TSV + v_ci * 24 + v_kr * 6) * 4u;
  int geo_packed = u_tyl | u_txl << 3 | v_tyl << 6 | v_txl << 9 | v_kr << 12 | u_r << 15;
  asm volatile("" : "+v"(geo_packed));   // opaque: otherwise hipcc keeps the six fields in six registers for the whole kernel
  const float v_sgn = (tid & 3) == 1 ? 1.f : -1.f;   // kr0: r0 - r2, kr1: r1 + r2, kr2: r2 - r1, kr3: r3 - r1
  const float u_sgn = (UROW && (tid & 1)) ? -1.f : 1.f;    // UROW: row 0 forms r0 + r1, row 1 forms r0 - r1
  // deferred BatchNorm of the block's BN channels: (scale, shift) pairs in LDS behind the images, read per task and k-step (one
  // ds_read_b64 instead of two registers per task held for the whole kernel; the kernel lives at the register limit)
  constexpr int SCS = IMG + 2 * BUF;
  float lo = -__builtin_inff();
  if constexpr (!PLAIN) {
    const SrcD& S = seg1 ? P.a1 : P.a0;
    if (tid < BN) {
      f32x2d v = f32x2d{1.f, 0.f};
      if (S.scale != nullptr) v = f32x2d{S.scale[S_c0 + tid], S.shift[S_c0 + tid]};
      *reinterpret_cast<f32x2d*>(smem + SCS + 2 * tid) = v;
    }
    if (S.relu) lo = 0.f;
    // (published by the first barrier of the pipeline below)
  }

  // ---- coordinates of the next k-step to LOAD, carried incrementally ---------------------------------------------------------
  int st_n, st_sy, st_sx;
  {
    const int per = P.sy_n * P.sx_n;
    st_n = s_begin / per;
    const int rs = s_begin - st_n * per;
    st_sy = rs / P.sx_n;
    st_sx = rs - st_sy * P.sx_n;
  }

  // Raw operands of one k-step wait in LDS (by LDS-DMA), not in registers (14 of them held across the MFMA phase did not fit 256):
  //   * dy: every thread moves ITS OWN pieces (one or two rows of 16 bytes) into a private slot -- piece p of wave w occupies 1 KiB
  //     at (p * 8 + w) * 256 floats, lane l its bytes [16 l, 16 l + 16) -- and reads them back itself a k-step later: the only
  //     ordering needed is the thread's own vmcnt(0) before the read, and program order read -> next fill;
  //   * activation windows: ONE image per block and k-step, [channel][window row][piece of 4 floats], the halo shared by the tiles
  //     of the k-step (80 instead of 128 bytes per row at 1 x 4 tiles) and moved as runs of consecutive pieces by consecutive lanes
  //     (fill i = pieces 64 i .. 64 i + 63 of the image, wave w issues fills w, w + 8, ...).  Another wave's pieces are read, so
  //     there are two window images (k-step parity) and the fills are published by the k-step's barrier (vmcnt(0) in front of it).
  int r_mask = 0;            // edge k-steps: bit 0/1 dy row ok, bits 2..7: window columns of the V tasks ok
  bool r_edge = false;       // wave-uniform: the masks apply
  float* const raw_w = smem + wave * 256;                       // this wave's slot of piece 0 (wave-uniform: the DMA's LDS base)
  const float* const raw_r = smem + wave * 256 + lane * 4;  // this lane's 16 bytes of piece 0
  // window fills of this wave: piece 64 (wave + 8 k) + lane = (channel, window row, piece) -> byte offset from the k-step's window
  // origin in the first channel's plane; row and piece packed for border k-steps (-1: a dummy lane behind the image)
  unsigned x_off[KB];
  int x_meta = 0;   // 8 bits per fill: window row | piece << 4 | dummy << 7
  {
    const int per_ch = P.WR * P.NP;
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int pid = 64 * (wave + 8 * k) + lane;
      const int ch = pid / per_ch, rem = pid - ch * per_ch;
      const int wr = rem / P.NP, pp = rem - wr * P.NP;
      const bool dummy = ch >= BN;
      x_off[k] = dummy ? 0u : (unsigned)((long long)ch * S_cs + (long long)wr * S_ws + 4 * pp) * 4u;
      x_meta |= (dummy ? 128 : (wr | pp << 4)) << (8 * k);
    }
    asm volatile("" : "+v"(x_meta));
  }

  // Addresses: a wave-uniform 64-bit base that depends on the IMAGE only (the block's first channel plane) plus an unsigned 32-bit
  // byte offset per lane = (k-step origin, scalar) + (task constant).  A lane whose piece must not be read where it lies takes
  // offset 0 instead.  (The first form of this code selected between 64-bit pointers per lane and held 14 raw registers across
  // the MFMA phase; at 256 registers hipcc then spilled the zero-extended offsets on some paths of the branchy prologue only and
  // reloaded them on all -- wild addresses, a memory fault at the 160x213 level.  No 64-bit vector address selects here, the raw
  // values wait in LDS, and tests/test_abi.py checks that these kernels use no scratch.)
  auto load = [&](auto wb_c) __attribute__((always_inline)) {
    constexpr int wb = decltype(wb_c)::value;   // window image of the k-step (its parity inside the block's range)
    const int n = st_n, ty0 = st_sy * P.KY, tx0 = st_sx * P.KX;
    if (++st_sx == P.sx_n) {
      st_sx = 0;
      if (++st_sy == P.sy_n) {
        st_sy = 0;
        ++st_n;
      }
    }
    const char* const dblk = reinterpret_cast<const char*>(P.dy.p + (long long)n * P.dy.ns + (long long)m0 * P.dy.cs);
    // (the activation base is 4 floats IN FRONT of the plane, inside the slack the caller vouches for: the piece that starts at
    //  column -1 of row 0 of the block's first channel has offset -4 bytes from the plane, and the offsets are UNSIGNED 32-bit --
    //  a zero-extended -4 is 4 GiB away: the memory fault of this kernel's first LDS-DMA builds)
    const char* vblk = reinterpret_cast<const char*>(S_p + (long long)n * S_ns + (long long)S_c0 * S_cs) - 16;
    asm volatile("" : "+s"(vblk));   // (a scalar: otherwise hipcc adds the -16 per lane and fill, a 64-bit vector addition each)
    const int hs = 2 * ty0 - 1 - S_oh, wsx = 4 * tx0 - 1 - S_ow;
    const unsigned d_org = (unsigned)(2 * ty0 * P.dy.ws + 4 * tx0) * 4u;   // k-step origin inside a dy plane, bytes
    const int v_org = (hs * S_ws + wsx) * 4 + 16;                          // ... from vblk (< 0 only where every lane is masked)
    const bool tiles_in = ty0 + P.KY <= P.tiles_y && tx0 + P.KX <= P.tiles_x;
    const bool inside = tiles_in && 2 * (ty0 + P.KY) <= P.H && hs >= 0 && hs + 2 * P.KY + 2 <= S_H && wsx >= 0 &&
                        wsx + 4 * P.KX + 2 <= S_W;
    r_edge = !inside;
    unsigned o_y0 = d_org + u_voff, o_y1 = o_y0 + (unsigned)P.dy.ws * 4u;
    if (!inside) {
      int m = 0;
      int geo = geo_packed;
      asm volatile("" : "+v"(geo));   // (as x_meta below)
      {
        const int ty = ty0 + (geo & 7), tx = tx0 + (geo >> 3 & 7);
        const bool t_ok = ty < P.tiles_y && tx < P.tiles_x;
        const int h = 2 * ty + (geo >> 15 & 1);
        const bool ok0 = t_ok && h < P.H, ok1 = t_ok && h + 1 < P.H;
        o_y0 = ok0 ? o_y0 : 0u;
        o_y1 = ok1 ? o_y1 : 0u;
        m = (ok0 ? 1 : 0) | (ok1 ? 2 : 0);
      }
      {
        const int vy = geo >> 6 & 7, vx = geo >> 9 & 7;
        const int ty = ty0 + vy, tx = tx0 + vx;
        const bool t_ok = ty < P.tiles_y && tx < P.tiles_x;
        const int row = hs + 2 * vy + (geo >> 12 & 3), c0 = wsx + 4 * vx;
        const bool r_ok = t_ok && (unsigned)row < (unsigned)S_H;
        // the valid columns of the 6-float row piece are the range [max(0, -c0), min(6, S_W - c0)): a mask from two shifts
        const int lo_ = c0 < 0 ? -c0 : 0, hi_ = S_W - c0;
        const int lo = lo_ < 6 ? lo_ : 6, hi = hi_ < 0 ? 0 : (hi_ < 6 ? hi_ : 6);   // shift counts in [0, 6]
        const int cm = (r_ok && hi > lo) ? ((1 << hi) - 1) & ~((1 << lo) - 1) : 0;
        m |= cm << 2;
      }
      r_mask = m;
    }
    auto fill = [&](const char* base, unsigned off, float* dst) __attribute__((always_inline)) {
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(base + off), dst, 16, 0, 0);
    };
    if constexpr (WG2D_ABL & 128) o_y0 = o_y1 = 0u;
    // window pieces: one with a column inside the segment lies within 3 floats of its row's ends (slack >= 4) and is read where it
    // lies, partly outside or not (the transform masks by column); one without is not read where it lies (offset 0)
    if constexpr (!(WG2D_ABL & 64)) {
#pragma unroll
      for (int k = 0; k < KB; ++k)
        if (wave + 8 * k < P.NI) {
          unsigned off = (unsigned)v_org + x_off[k];
          if (!inside) {
            int mt = x_meta;
            asm volatile("" : "+v"(mt));   // (unpacked HERE, in border k-steps only: hipcc would hoist the fields out of the loop into registers)
            mt >>= 8 * k;
            const int row = hs + (mt & 15), c0 = wsx + 4 * (mt >> 4 & 7);
            const bool ok = !(mt & 128) && (unsigned)row < (unsigned)S_H && c0 + 3 >= 0 && c0 < S_W;
            off = ok ? off : 0u;
          }
          if constexpr (WG2D_ABL & 128) off = 0u;
          fill(vblk, off, smem + WIN + wb * (WINI * 256) + (wave + 8 * k) * 256);
        }
    }
    // dy pieces go into the slots this thread reads its raw dy rows from: those reads have to have RETURNED before a fill can land
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
    if constexpr (!(WG2D_ABL & 32)) {
      fill(dblk, o_y0, raw_w);
      if constexpr (!UROW) fill(dblk, o_y1, raw_w + 8 * 256);
    }
  };

  // transform this thread's raw pieces into LDS image `buf` (compile-time constant)
  auto transform = [&](auto buf_c) __attribute__((always_inline)) {
    constexpr int buf = decltype(buf_c)::value;
    // Packed fp32 math (v_pk_add_f32 / v_pk_fma_f32: two floats for the issue slot of one; an fp32 MFMA stream does not hide vector
    // instructions, profiles/r05_mfma_f32_issue_ubench.txt).  The six frequencies of a row are stored in the order
    // [1, 2, 3, 4, 0, 5]: the transforms produce (1,2), (3,4) and (0,5) as register pairs, and U, V and the epilogue only have to agree.
    const f32x2d p1m1 = {1.f, -1.f}, p2m2 = {2.f, -2.f}, c4 = {4.f, 4.f};
    // A4 of one dy row (y0..y3) -> (U1, U2), (U3, U4); U0 = y0, U5 = y3
    auto a4_row = [&](const f32x4& y, f32x2d& u12, f32x2d& u34) __attribute__((always_inline)) {
      const f32x2d lo2 = {y[0], y[1]}, hi2 = {y[2], y[3]};
      const f32x2d pq = lo2 + hi2;                                   // (y0 + y2, y1 + y3)
      const f32x2d ab = __builtin_elementwise_fma(c4, hi2, lo2);     // (y0 + 4 y2, y1 + 4 y3)
      u12 = __builtin_elementwise_fma(f32x2d{pq[1], pq[1]}, p1m1, f32x2d{pq[0], pq[0]});
      u34 = __builtin_elementwise_fma(f32x2d{ab[1], ab[1]}, p2m2, f32x2d{ab[0], ab[0]});
    };
    // ---- U ----
    {
      f32x4 y0 = *reinterpret_cast<const f32x4*>(raw_r), y1 = y0;
      if constexpr (!UROW) y1 = *reinterpret_cast<const f32x4*>(raw_r + 8 * 256);
      if (r_edge) {
        if (!(r_mask & 1)) y0 = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (!UROW)
          if (!(r_mask & 2)) y1 = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      f32x2d a12, a34;
      a4_row(y0, a12, a34);
      if constexpr (UROW) {
        // this lane holds A4 of ONE dy row; rows fr of U = [r0, r0 + r1, r0 - r1, r1]: lane r = 0 stores (own, own + x) as rows 0, 1,
        // lane r = 1 stores (x - own, own) as rows 2, 3
        const f32x2d sg = {u_sgn, u_sgn};
        const f32x2d a05 = {y0[0], y0[3]};
        const f32x2d x12 = {w2d_dpp(a12[0], 1), w2d_dpp(a12[1], 1)}, x34 = {w2d_dpp(a34[0], 1), w2d_dpp(a34[1], 1)};
        const f32x2d x05 = {w2d_dpp(a05[0], 1), w2d_dpp(a05[1], 1)};
        const f32x2d t12 = __builtin_elementwise_fma(a12, sg, x12), t34 = __builtin_elementwise_fma(a34, sg, x34);
        const f32x2d t05 = __builtin_elementwise_fma(a05, sg, x05);
        float* const o_own = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + u_wr) + buf * BUF + ((tid & 1) ? 18 : 0);
        float* const o_t1 = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + u_wr) + buf * BUF + ((tid & 1) ? 12 : 6);
        *reinterpret_cast<f32x2d*>(o_own) = a12;
        *reinterpret_cast<f32x2d*>(o_own + 2) = a34;
        *reinterpret_cast<f32x2d*>(o_own + 4) = a05;
        *reinterpret_cast<f32x2d*>(o_t1) = t12;
        *reinterpret_cast<f32x2d*>(o_t1 + 2) = t34;
        *reinterpret_cast<f32x2d*>(o_t1 + 4) = t05;
      } else {
        f32x2d b12, b34;
        a4_row(y1, b12, b34);
        const f32x2d s12 = a12 + b12, s34 = a34 + b34, d12 = a12 - b12, d34 = a34 - b34;
        const float s0 = y0[0] + y1[0], s5 = y0[3] + y1[3], d0 = y0[0] - y1[0], d5 = y0[3] - y1[3];
        // rows fr of U = [r0, r0 + r1, r0 - r1, r1], 24 consecutive floats as six 16-byte groups
        float* const op = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + u_wr) + buf * BUF;
        *reinterpret_cast<f32x4*>(op) = f32x4{a12[0], a12[1], a34[0], a34[1]};
        *reinterpret_cast<f32x4*>(op + 4) = f32x4{y0[0], y0[3], s12[0], s12[1]};
        *reinterpret_cast<f32x4*>(op + 8) = f32x4{s34[0], s34[1], s0, s5};
        *reinterpret_cast<f32x4*>(op + 12) = f32x4{d12[0], d12[1], d34[0], d34[1]};
        *reinterpret_cast<f32x4*>(op + 16) = f32x4{d0, d5, b12[0], b12[1]};
        *reinterpret_cast<f32x4*>(op + 20) = f32x4{b34[0], b34[1], y1[0], y1[3]};
      }
    }
    // ---- V rows ----
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float* const wp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(smem) + v_rd) + buf * (WINI * 256) + i * 32 * (P.WR * P.NP * 4);
      const f32x4 ra = *reinterpret_cast<const f32x4*>(wp);
      const f32x2d rb = *reinterpret_cast<const f32x2d*>(wp + 4);
      f32x2d t0 = {ra[0], ra[1]}, t1 = {ra[2], ra[3]}, t2 = rb;
      if constexpr (!PLAIN) {
        const f32x2d ss = *reinterpret_cast<const f32x2d*>(smem + SCS + 2 * ((tid >> 4) + 32 * i));
        const f32x2d sc2 = {ss[0], ss[0]}, sh2 = {ss[1], ss[1]}, lo2 = {lo, lo};
        t0 = __builtin_elementwise_max(__builtin_elementwise_fma(t0, sc2, sh2), lo2);
        t1 = __builtin_elementwise_max(__builtin_elementwise_fma(t1, sc2, sh2), lo2);
        t2 = __builtin_elementwise_max(__builtin_elementwise_fma(t2, sc2, sh2), lo2);
      }
      if (r_edge) {
        const int m = r_mask;
        if (!(m & 4)) t0[0] = 0.f;
        if (!(m & 8)) t0[1] = 0.f;
        if (!(m & 16)) t1[0] = 0.f;
        if (!(m & 32)) t1[1] = 0.f;
        if (!(m & 64)) t2[0] = 0.f;
        if (!(m & 128)) t2[1] = 0.f;
      }
      // B4^T of the window row d0..d5 = (t0, t1, t2), as gsd_conv3x3_w2d.hip forms it
      const f32x2d m41 = {-4.f, -1.f}, m5 = {-5.f, -5.f};
      const f32x2d ac = __builtin_elementwise_fma(f32x2d{t1[0], t1[0]}, m41, f32x2d{t2[0], t2[0]});   // (d4 - 4 d2, d4 - d2)
      const f32x2d be = __builtin_elementwise_fma(f32x2d{t0[1], t0[1]}, m41, f32x2d{t1[1], t1[1]});   // (d3 - 4 d1, d3 - d1)
      const f32x2d v12 = __builtin_elementwise_fma(f32x2d{be[0], be[0]}, p1m1, f32x2d{ac[0], ac[0]});
      const f32x2d v34 = __builtin_elementwise_fma(f32x2d{be[1], be[1]}, p2m2, f32x2d{ac[1], ac[1]});
      const f32x2d v05 = __builtin_elementwise_fma(t0, c4, __builtin_elementwise_fma(t1, m5, t2));
      // B2^T down the window column: the partner row from the lane's quad
      const f32x2d sg = {v_sgn, v_sgn};
      const f32x2d x12 = {w2d_dpp(v12[0], 0), w2d_dpp(v12[1], 0)}, x34 = {w2d_dpp(v34[0], 0), w2d_dpp(v34[1], 0)};
      const f32x2d x05 = {w2d_dpp(v05[0], 0), w2d_dpp(v05[1], 0)};
      float* const op = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + v_wr) + buf * BUF + i * (32 * 24);
      *reinterpret_cast<f32x2d*>(op) = __builtin_elementwise_fma(sg, x12, v12);
      *reinterpret_cast<f32x2d*>(op + 2) = __builtin_elementwise_fma(sg, x34, v34);
      *reinterpret_cast<f32x2d*>(op + 4) = __builtin_elementwise_fma(sg, x05, v05);
    }
  };

  f32x4 acc[2][24];   // (zeroed behind the pipeline's prologue: 192 registers that the prologue's address work does not have to avoid)

  const int a_rd = IMG + j * TSU + (wm * 32 + l16) * 24;
  const int b_rd = IMG + 4 * TSU + j * TSV + (wn * 16 + l16) * 24;

  // One operand group = four frequencies of (channel l16, tile j) per ds_read_b128: 3 reads feed 8 MFMAs.  The groups are NOT
  // double-buffered in the source: 12 operand registers instead of 24 keep the kernel inside 256 registers without scratch, and the
  // SIMD's other wave multiplies while this one waits for its reads.
  auto multiply = [&](auto buf_c) __attribute__((always_inline)) {
    constexpr int buf = decltype(buf_c)::value;
    const float* const Sb = smem + buf * BUF;
#pragma unroll
    for (int g = 0; g < 6; ++g) {
      f32x4 a0, a1, b;
      if constexpr (WG2D_ABL & 8) {
        a0 = a1 = b = f32x4{1.f, 2.f, 3.f, (float)g};
      } else {
        a0 = *reinterpret_cast<const f32x4*>(Sb + a_rd + 4 * g);
        a1 = *reinterpret_cast<const f32x4*>(Sb + a_rd + 16 * 24 + 4 * g);
        b = *reinterpret_cast<const f32x4*>(Sb + b_rd + 4 * g);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (WG2D_ABL & 1) {
          acc[0][4 * g + e][e] += a0[e] * b[e];
          acc[1][4 * g + e][e] += a1[e] * b[e];
        } else {
          acc[0][4 * g + e] = mfma16(a0[e], b[e], acc[0][4 * g + e]);
          acc[1][4 * g + e] = mfma16(a1[e], b[e], acc[1][4 * g + e]);
        }
      }
    }
  };

  // ---- pipeline: image (it & 1) holds the transforms of k-step `it`; the raw registers hold k-step it + 1 ---------------------
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  if (nst > 0) {
    load(I0{});
    gsd_dma_barrier();   // vmcnt(0) + barrier: everyone's fills of the first k-step are in
    transform(I0{});
    if (nst > 1) load(I1{});
    gsd_dma_barrier();
  }
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int f = 0; f < 24; ++f) acc[m][f] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (nst > 0) {
    auto step = [&](const int it, auto cur_c) __attribute__((always_inline)) {
      constexpr int cur = decltype(cur_c)::value;
      if (it + 1 < nst && !((WG2D_ABL & 4) && it > 1)) transform(std::integral_constant<int, cur ^ 1>{});
      __builtin_amdgcn_sched_barrier(0);   // (phases in program order: the register budget is 192 accumulators + one phase's values)
      if (it + 2 < nst && !((WG2D_ABL & 2) && it > 1)) load(cur_c);   // k-step it + 2 has this one's parity
      __builtin_amdgcn_sched_barrier(0);
      multiply(cur_c);
      // the fills of k-step it + 2 have the MFMA phase to land; published to the other waves (window pieces) by the barrier
      if constexpr (!(WG2D_ABL & 16)) gsd_dma_barrier();
    };
#ifndef WG2D_PIPE   // 1: the transform of k-step it + 1 in pieces BETWEEN the MFMA groups of k-step it (software pipeline inside the wave)
#define WG2D_PIPE 1
#endif
    // With separate phases a k-step was a chain of exposed latencies: raw reads -> transform -> stores | fills | operand reads ->
    // MFMAs, every wave of the CU in the same phase (barrier): stamps showed a wave in its MFMA phase for 36 % of a k-step and
    // removing a third of the vector instructions changed nothing.  Here the vector work rides between the MFMA groups: a piece
    // of ~10-16 instructions works on values that were read a group earlier, clustered (the first vector instruction in an MFMA gap
    // costs 12.6 cycles, each further one 4: profiles/r05_mfma_f32_issue_ubench.txt), and its LDS latencies lie behind MFMAs.
    auto step_pipe = [&](const int it, auto cur_c) __attribute__((always_inline)) {
      constexpr int cur = decltype(cur_c)::value, nxt = cur ^ 1;
      const bool tr = it + 1 < nst, ld = it + 2 < nst;
      const float* const Sb = smem + cur * BUF;
      const f32x2d p1m1 = {1.f, -1.f}, p2m2 = {2.f, -2.f}, c4 = {4.f, 4.f};
      f32x4 a0, a1, b;
      auto rd_ops = [&](const int g) __attribute__((always_inline)) {
        a0 = *reinterpret_cast<const f32x4*>(Sb + a_rd + 4 * g);
        a1 = *reinterpret_cast<const f32x4*>(Sb + a_rd + 16 * 24 + 4 * g);
        b = *reinterpret_cast<const f32x4*>(Sb + b_rd + 4 * g);
      };
      auto mm = [&](const int g) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[0][4 * g + e] = mfma16(a0[e], b[e], acc[0][4 * g + e]);
          acc[1][4 * g + e] = mfma16(a1[e], b[e], acc[1][4 * g + e]);
        }
      };
      auto a4_row = [&](const f32x4& y, f32x2d& u12, f32x2d& u34) __attribute__((always_inline)) {
        const f32x2d lo2 = {y[0], y[1]}, hi2 = {y[2], y[3]};
        const f32x2d pq = lo2 + hi2;
        const f32x2d ab = __builtin_elementwise_fma(c4, hi2, lo2);
        u12 = __builtin_elementwise_fma(f32x2d{pq[1], pq[1]}, p1m1, f32x2d{pq[0], pq[0]});
        u34 = __builtin_elementwise_fma(f32x2d{ab[1], ab[1]}, p2m2, f32x2d{ab[0], ab[0]});
      };
      // masks of the k-step being transformed (load() below replaces r_mask / r_edge by the next one's)
      const int t_mask = r_mask;
      const bool t_edge = r_edge;
      f32x4 y0, y1;
      f32x2d a12, a34, b12, b34;
      f32x4 ra;
      f32x2d rb, v12, v34, v05;
      float* const u_out = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + u_wr) + nxt * BUF;
      auto rd_y = [&]() __attribute__((always_inline)) {
        y0 = *reinterpret_cast<const f32x4*>(raw_r);
        y1 = y0;
        if constexpr (!UROW) y1 = *reinterpret_cast<const f32x4*>(raw_r + 8 * 256);
      };
      auto u_rows = [&]() __attribute__((always_inline)) {      // A4 along the dy rows
        if (t_edge) {
          if (!(t_mask & 1)) y0 = f32x4{0.f, 0.f, 0.f, 0.f};
          if constexpr (!UROW)
            if (!(t_mask & 2)) y1 = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        a4_row(y0, a12, a34);
        if constexpr (!UROW) a4_row(y1, b12, b34);
      };
      auto u_cols = [&]() __attribute__((always_inline)) {      // A2 down the columns, stores
        if constexpr (UROW) {
          const f32x2d sg = {u_sgn, u_sgn};
          const f32x2d a05 = {y0[0], y0[3]};
          const f32x2d x12 = {w2d_dpp(a12[0], 1), w2d_dpp(a12[1], 1)}, x34 = {w2d_dpp(a34[0], 1), w2d_dpp(a34[1], 1)};
          const f32x2d x05 = {w2d_dpp(a05[0], 1), w2d_dpp(a05[1], 1)};
          float* const o_own = u_out + ((tid & 1) ? 18 : 0);
          float* const o_t1 = u_out + ((tid & 1) ? 12 : 6);
          *reinterpret_cast<f32x2d*>(o_own) = a12;
          *reinterpret_cast<f32x2d*>(o_own + 2) = a34;
          *reinterpret_cast<f32x2d*>(o_own + 4) = a05;
          *reinterpret_cast<f32x2d*>(o_t1) = __builtin_elementwise_fma(a12, sg, x12);
          *reinterpret_cast<f32x2d*>(o_t1 + 2) = __builtin_elementwise_fma(a34, sg, x34);
          *reinterpret_cast<f32x2d*>(o_t1 + 4) = __builtin_elementwise_fma(a05, sg, x05);
        } else {
          const f32x2d s12 = a12 + b12, s34 = a34 + b34, d12 = a12 - b12, d34 = a34 - b34;
          const float s0 = y0[0] + y1[0], s5 = y0[3] + y1[3], d0 = y0[0] - y1[0], d5 = y0[3] - y1[3];
          *reinterpret_cast<f32x4*>(u_out) = f32x4{a12[0], a12[1], a34[0], a34[1]};
          *reinterpret_cast<f32x4*>(u_out + 4) = f32x4{y0[0], y0[3], s12[0], s12[1]};
          *reinterpret_cast<f32x4*>(u_out + 8) = f32x4{s34[0], s34[1], s0, s5};
          *reinterpret_cast<f32x4*>(u_out + 12) = f32x4{d12[0], d12[1], d34[0], d34[1]};
          *reinterpret_cast<f32x4*>(u_out + 16) = f32x4{d0, d5, b12[0], b12[1]};
          *reinterpret_cast<f32x4*>(u_out + 20) = f32x4{b34[0], b34[1], y1[0], y1[3]};
        }
      };
      auto rd_v = [&](const int i) __attribute__((always_inline)) {
        const float* const wp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(smem) + v_rd) + nxt * (WINI * 256) + i * 32 * (P.WR * P.NP * 4);
        ra = *reinterpret_cast<const f32x4*>(wp);
        rb = *reinterpret_cast<const f32x2d*>(wp + 4);
      };
      auto v_rows = [&](const int i) __attribute__((always_inline)) {   // deferred BatchNorm + ReLU, masks, B4^T along the window row
        f32x2d t0 = {ra[0], ra[1]}, t1 = {ra[2], ra[3]}, t2 = rb;
        if constexpr (!PLAIN) {
          const f32x2d ss = *reinterpret_cast<const f32x2d*>(smem + SCS + 2 * ((tid >> 4) + 32 * i));
          const f32x2d sc2 = {ss[0], ss[0]}, sh2 = {ss[1], ss[1]}, lo2 = {lo, lo};
          t0 = __builtin_elementwise_max(__builtin_elementwise_fma(t0, sc2, sh2), lo2);
          t1 = __builtin_elementwise_max(__builtin_elementwise_fma(t1, sc2, sh2), lo2);
          t2 = __builtin_elementwise_max(__builtin_elementwise_fma(t2, sc2, sh2), lo2);
        }
        if (t_edge) {
          const int m = t_mask;
          if (!(m & 4)) t0[0] = 0.f;
          if (!(m & 8)) t0[1] = 0.f;
          if (!(m & 16)) t1[0] = 0.f;
          if (!(m & 32)) t1[1] = 0.f;
          if (!(m & 64)) t2[0] = 0.f;
          if (!(m & 128)) t2[1] = 0.f;
        }
        const f32x2d m41 = {-4.f, -1.f}, m5 = {-5.f, -5.f};
        const f32x2d ac = __builtin_elementwise_fma(f32x2d{t1[0], t1[0]}, m41, f32x2d{t2[0], t2[0]});
        const f32x2d be = __builtin_elementwise_fma(f32x2d{t0[1], t0[1]}, m41, f32x2d{t1[1], t1[1]});
        v12 = __builtin_elementwise_fma(f32x2d{be[0], be[0]}, p1m1, f32x2d{ac[0], ac[0]});
        v34 = __builtin_elementwise_fma(f32x2d{be[1], be[1]}, p2m2, f32x2d{ac[1], ac[1]});
        v05 = __builtin_elementwise_fma(t0, c4, __builtin_elementwise_fma(t1, m5, t2));
      };
      auto v_cols = [&](const int i) __attribute__((always_inline)) {   // B2^T down the window column (quad partners by DPP), stores
        int tq = threadIdx.x;
        asm volatile("" : "+v"(tq));   // (the sign is re-derived here, three instructions, instead of living in a register: as geo_packed)
        const float sgn = (tq & 3) == 1 ? 1.f : -1.f;
        const f32x2d sg = {sgn, sgn};
        const f32x2d x12 = {w2d_dpp(v12[0], 0), w2d_dpp(v12[1], 0)}, x34 = {w2d_dpp(v34[0], 0), w2d_dpp(v34[1], 0)};
        const f32x2d x05 = {w2d_dpp(v05[0], 0), w2d_dpp(v05[1], 0)};
        float* const op = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + v_wr) + nxt * BUF + i * (32 * 24);
        *reinterpret_cast<f32x2d*>(op) = __builtin_elementwise_fma(sg, x12, v12);
        *reinterpret_cast<f32x2d*>(op + 2) = __builtin_elementwise_fma(sg, x34, v34);
        *reinterpret_cast<f32x2d*>(op + 4) = __builtin_elementwise_fma(sg, x05, v05);
      };
#define WG2D_SB __builtin_amdgcn_sched_barrier(0)
#ifndef WG2D_VPRIO     // diagnostic: wave priority during the vector pieces
#define WG2D_VPRIO 0
#endif
#ifndef WG2D_SPLITMM   // diagnostic: a piece between the two halves of an MFMA group instead of behind it
#define WG2D_SPLITMM 0
#endif
      auto mm_half = [&](const int g, const int h) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 2 * h; e < 2 * h + 2; ++e) {
          acc[0][4 * g + e] = mfma16(a0[e], b[e], acc[0][4 * g + e]);
          acc[1][4 * g + e] = mfma16(a1[e], b[e], acc[1][4 * g + e]);
        }
      };
      // one MFMA group, the next group's operand reads behind it, and a vector piece
      auto group = [&](const int g, auto piece) __attribute__((always_inline)) {
        if constexpr (WG2D_SPLITMM) {
          mm_half(g, 0); WG2D_SB;
          if constexpr (WG2D_VPRIO) __builtin_amdgcn_s_setprio(WG2D_VPRIO);
          piece();
          if constexpr (WG2D_VPRIO) __builtin_amdgcn_s_setprio(0);
          WG2D_SB;
          mm_half(g, 1); WG2D_SB;
          if (g + 1 < 6) rd_ops(g + 1);
          WG2D_SB;
        } else {
          mm(g); WG2D_SB;
          if (g + 1 < 6) rd_ops(g + 1);
          if constexpr (WG2D_VPRIO) __builtin_amdgcn_s_setprio(WG2D_VPRIO);
          piece();
          if constexpr (WG2D_VPRIO) __builtin_amdgcn_s_setprio(0);
          WG2D_SB;
        }
      };
      rd_ops(0);
      if (tr) rd_y();
      WG2D_SB;
      group(0, [&]() __attribute__((always_inline)) { if (ld) load(cur_c); });   // fills of k-step it + 2 (this one's parity); waits for rd_y
      group(1, [&]() __attribute__((always_inline)) { if (tr) { u_rows(); if constexpr (UROW) u_cols(); } });
      group(2, [&]() __attribute__((always_inline)) { if (tr) { rd_v(0); if constexpr (!UROW) u_cols(); } });
      group(3, [&]() __attribute__((always_inline)) { if (tr) v_rows(0); });
      group(4, [&]() __attribute__((always_inline)) { if (tr) { v_cols(0); if constexpr (NV > 1) rd_v(1); } });
      group(5, [&]() __attribute__((always_inline)) {
        if constexpr (NV > 1) {
          if (tr) { v_rows(1); v_cols(1); }
        }
      });
#undef WG2D_SB
      gsd_dma_barrier();
    };
    for (int it = 0; it < nst; it += 2) {
      if constexpr (WG2D_PIPE != 0 && WG2D_ABL == 0) {
        step_pipe(it, I0{});
        if (it + 1 < nst) step_pipe(it + 1, I1{});
      } else {
        step(it, I0{});
        if (it + 1 < nst) step(it + 1, I1{});
      }
    }
  }

  // ---- epilogue: G4 along fc, G2^T along fr, per split (linear: the slab reduction only adds) ---------------------------------
  const size_t pl = (size_t)P.M * P.Ncols;
  // (lane coordinates re-derived behind the loop from a laundered thread id: nothing of the epilogue's addressing is held in a
  //  register -- or spilled -- across the loop)
  int tid_e = threadIdx.x;
  asm volatile("" : "+v"(tid_e));
  const int j_e = (tid_e & 63) >> 4, l16_e = tid_e & 15;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int mr = m0 + wm * 32 + m * 16 + j_e * 4 + reg;
      const int col = n0 + wn * 16 + l16_e;
      float E[4][3];
#pragma unroll
      for (int fr = 0; fr < 4; ++fr) {
        // (stored order of a row's frequencies: [1, 2, 3, 4, 0, 5])
        const float D1 = acc[m][fr * 6 + 0][reg], D2 = acc[m][fr * 6 + 1][reg], D3 = acc[m][fr * 6 + 2][reg];
        const float D4 = acc[m][fr * 6 + 3][reg], D0 = acc[m][fr * 6 + 4][reg], D5 = acc[m][fr * 6 + 5][reg];
        E[fr][0] = 0.25f * D0 - (1.f / 6.f) * (D1 + D2) + (1.f / 24.f) * (D3 + D4);
        E[fr][1] = (1.f / 6.f) * (D2 - D1) + (1.f / 12.f) * (D3 - D4);
        E[fr][2] = (1.f / 6.f) * (D3 + D4 - D1 - D2) + D5;
      }
      float* const o = P.slabs + ((size_t)split * 9 * P.M + mr) * P.Ncols + col;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const float h = 0.5f * (E[1][s] + E[2][s]);
        o[(0 * 3 + s) * pl] = E[0][s] + h;
        o[(1 * 3 + s) * pl] = 0.5f * (E[1][s] - E[2][s]);
        o[(2 * 3 + s) * pl] = h + E[3][s];
      }
    }
}

namespace {

struct WgW2dPlan {
  int KY, KX, kx_log2, tiles_y, tiles_x, sy_n, sx_n, BM, BN, mblocks, nblocks, ksteps_total, splits;
  int64_t slab_elems;
  bool ok;
};

WgW2dPlan plan_wg2d(int N, int H, int W, int M, int Ncols) {
  WgW2dPlan p;
  p.tiles_y = ceil_div(H, 2);
  p.tiles_x = ceil_div(W, 4);
  long best = -1;
  const int force_kx = gsd_env_int("GSD_WG2D_KX", 0);   // tuning
  for (int kx = 4; kx >= 1; kx /= 2) {
    if (force_kx && kx != force_kx) continue;
    const int ky = 4 / kx;
    // fewest k-steps, weighted by what a k-step of that shape costs (profiles/r06_wg2d_kstep_shapes.txt, batch 32): 2 x 2 tiles cost
    // what 1 x 4 tiles cost within 2 % either way (a 6 x 12 window instead of 4 x 20) and get the ties, 4 x 1 tiles 7-16 % more
    const long steps = (long)ceil_div(p.tiles_y, ky) * ceil_div(p.tiles_x, kx) * (kx == 4 ? 100 : kx == 2 ? 99 : 112);
    if (best < 0 || steps < best) {
      best = steps;
      p.KY = ky; p.KX = kx;
    }
  }
  p.kx_log2 = p.KX == 4 ? 2 : p.KX == 2 ? 1 : 0;
  p.sy_n = ceil_div(p.tiles_y, p.KY);
  p.sx_n = ceil_div(p.tiles_x, p.KX);
  p.BM = M >= 128 ? 128 : 64;
  p.BN = p.BM == 128 ? 32 : 64;
  p.ok = M % p.BM == 0 && Ncols % p.BN == 0;
  p.mblocks = ceil_div(M, p.BM);
  p.nblocks = ceil_div(Ncols, p.BN);
  p.ksteps_total = N * p.sy_n * p.sx_n;
  const int target = gsd_env_int("GSD_WG2D_BLOCKS", 256);   // one block per CU
  int splits = ceil_div(target, p.mblocks * p.nblocks);
  if (splits > p.ksteps_total) splits = p.ksteps_total;
  if (splits > 2048) splits = 2048;
  if (splits < 1) splits = 1;
  p.splits = splits;
  p.slab_elems = (int64_t)splits * 9 * M * Ncols;
  return p;
}

}  // namespace

// floats of slab scratch the 2-D form wants for a shape (0: the shape is not served)
int64_t gsd_wgrad_w2d_workspace(int N, int H, int W, int Cin, int Cout) {
  const WgW2dPlan p = plan_wg2d(N, H, W, Cout, Cin);
  return p.ok ? p.slab_elems : 0;
}

// MFMA instructions of one launch: k-steps x 24 frequencies per (16 co x 16 ci) pair
int64_t gsd_wgrad_w2d_mfma_count(int N, int H, int W, int Cin, int Cout) {
  const WgW2dPlan p = plan_wg2d(N, H, W, Cout, Cin);
  return p.ok ? (int64_t)p.ksteps_total * 24 * (Cout / 16) * (Cin / 16) : 0;
}

// 1: the arguments admit the 2-D form (shape, segment geometry, alignment, slack); GSD_WGRAD_W2D=0 switches it off
int gsd_wgrad_w2d_use(const gsd_src* a, int nsrc, const gsd_src* dy, int Cin, int Cout, int N, int H, int W) {
  if (gsd_env_int("GSD_WGRAD_W2D", 1) == 0) return 0;
  const WgW2dPlan p = plan_wg2d(N, H, W, Cout, Cin);
  if (!p.ok) return 0;
  if (dy->w_stride % 4 != 0 || ((uintptr_t)dy->ptr & 15) != 0 || dy->c_stride % 4 != 0 || dy->n_stride % 4 != 0) return 0;
  if (dy->w_stride < 4 * p.tiles_x) return 0;
  if ((int64_t)p.BM * dy->c_stride * 4 >= (1LL << 31)) return 0;   // lane offsets are 32-bit byte offsets inside a block's planes
  if (nsrc == 2 && a[0].C % p.BN != 0) return 0;
  for (int i = 0; i < nsrc; ++i) {
    if (a[i].slack < 4) return 0;
    if ((int64_t)p.BN * a[i].c_stride * 4 >= (1LL << 31)) return 0;
    if ((int64_t)(a[i].H + 4) * a[i].w_stride * 4 >= (1LL << 31)) return 0;   // (k-step origins inside a plane are 32-bit too)
  }
  return 1;
}

// arguments already validated by gsd_conv3x3_wgrad and gsd_wgrad_w2d_use
int gsd_wgrad_w2d_run(const gsd_src* a, int nsrc, const gsd_src* dy, int Cin, int Cout, float* workspace, int64_t workspace_elems,
                      int N, int H, int W, int* splits_out, void* stream) {
  const WgW2dPlan pl = plan_wg2d(N, H, W, Cout, Cin);
  GSD_REQUIRE(pl.ok, GSD_ERR_UNSUPPORTED, "gsd_conv3x3_wgrad (w2d): shape not served");
  GSD_REQUIRE(workspace_elems >= pl.slab_elems, GSD_ERR_WORKSPACE, "gsd_conv3x3_wgrad: workspace %lld < %lld elements",
              (long long)workspace_elems, (long long)pl.slab_elems);
  WgW2dParams P;
  P.a0 = to_srcd(a[0]);
  P.a1 = nsrc > 1 ? to_srcd(a[1]) : null_srcd();
  P.dy = to_srcd(*dy);
  P.slabs = workspace;
  P.M = Cout; P.Ncols = Cin;
  P.N = N; P.H = H; P.W = W;
  P.KY = pl.KY; P.KX = pl.KX; P.kx_log2 = pl.kx_log2;
  P.tiles_y = pl.tiles_y; P.tiles_x = pl.tiles_x; P.sy_n = pl.sy_n; P.sx_n = pl.sx_n;
  P.WR = 2 * pl.KY + 2; P.NP = pl.KX + 1;
  P.NI = (pl.BN * P.WR * P.NP + 63) / 64;
  P.ksteps_total = pl.ksteps_total; P.splits = pl.splits; P.mblocks = pl.mblocks; P.nblocks = pl.nblocks;
  bool plain = true;
  for (int i = 0; i < nsrc; ++i) plain = plain && a[i].scale == nullptr && a[i].relu == 0;
  const long grid = (long)pl.splits * pl.mblocks * pl.nblocks;
  // LDS (floats): dy slots [1 or 2 pieces][8 waves][256] | two window images of 10 / 20 KiB | two images of [4 tiles][BM | BN][24] (+4) | (scale, shift)[BN]
  const size_t lds = ((size_t)(pl.BM == 64 ? 1 : 2) * 8 * 256 + (size_t)2 * (pl.BN == 32 ? 10 : 20) * 256 +
                      (size_t)2 * (4 * (pl.BM * 24 + 4) + 4 * (pl.BN * 24 + 4)) + 2 * pl.BN) * sizeof(float);
  GSD_REQUIRE(P.NI <= 8 * (pl.BN == 32 ? 2 : 3) && P.NI <= (pl.BN == 32 ? 10 : 20), GSD_ERR_UNSUPPORTED, "gsd_conv3x3_wgrad (w2d): window image too large");
  if (gsd_env_set("GSD_WG43_TRACE"))   // tuning: one line per launch
    fprintf(stderr, "wg2d M%d N%d %dx%d B%d kstep %dx%d ksteps %d splits %d blocks %ld BM %d BN %d plain %d lds %zu\n", Cout, Cin, H, W,
            N, pl.KY, pl.KX, pl.ksteps_total, pl.splits, grid, pl.BM, pl.BN, (int)plain, lds);
  const dim3 g((int)grid);
  const hipStream_t st = (hipStream_t)stream;
#define WG2D_LAUNCH(NWM_, NWN_, PL_)                                                                              \
  do {                                                                                                            \
    static gsd_attr_once once;                                                                                    \
    const void* fn = reinterpret_cast<const void*>(&wgrad3x3_w2d_kernel<NWM_, NWN_, PL_>);                        \
    if (hipError_t e = gsd_allow_big_lds(once, fn); e != hipSuccess) {                                            \
      gsd_set_error("gsd_conv3x3_wgrad (w2d): hipFuncSetAttribute: %s", hipGetErrorString(e));                    \
      return GSD_ERR_HIP;                                                                                         \
    }                                                                                                             \
    hipLaunchKernelGGL((wgrad3x3_w2d_kernel<NWM_, NWN_, PL_>), g, dim3(512), lds, st, P);                         \
  } while (0)
  if (pl.BM == 128) {
    if (plain) WG2D_LAUNCH(4, 2, true);
    else WG2D_LAUNCH(4, 2, false);
  } else {
    if (plain) WG2D_LAUNCH(2, 4, true);
    else WG2D_LAUNCH(2, 4, false);
  }
#undef WG2D_LAUNCH
  GSD_LAUNCH_CHECK("gsd_conv3x3_wgrad (w2d)");
  *splits_out = pl.splits;
  return GSD_OK;
}
