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
//   * a block of 4 waves owns 64 co x 32 ci; a k-step is 4 tiles (32 pixels); TWO blocks per CU;
//   * per k-step every thread takes three small transform TASKS: "U" (one (co, tile): two aligned 16-byte pieces of the row-pitched
//     dy -> 24 values) and two "V rows" (one window row of a (ci, tile): 6 floats -> deferred BatchNorm+ReLU -> B4^T row transform;
//     the B2^T column transform takes the other rows from the lane's quad by DPP), all in packed fp32 math, and stores the results
//     as a [tile][channel][24 frequencies] image;
//   * barrier; the MFMA phase of a wave (tile 32 co x 16 ci x 24 frequencies = 192 accumulator registers) reads its operands
//     frequency-major: one ds_read_b128 is four frequencies of (channel l16, tile j), 18 reads per 48 MFMAs, no vector work; barrier.
//
// Why 4-wave blocks, two per CU, with ONE image and two barriers per k-step: a vector phase and an MFMA phase of the SAME wave do
// not overlap, and the first form of this kernel (8 waves, two images, one barrier) put all eight waves of a CU in the same phase:
// stamps showed a wave in its MFMA phase for 36 % of a k-step, 16 % at the barrier, the matrix pipes 0.53 busy, and removing a
// third of the vector instructions changed nothing.  Two independent blocks per CU fall into opposite phases by themselves (each
// SIMD hosts one wave of each): one block's MFMAs run beside the other's transforms, with no code duplicated for a stagger.
// (A stagger inside one 8-wave block -- the two arms in one loop or in two loops -- made hipcc spill 70-800 bytes per lane.)
//
// The raw operands of the NEXT k-step wait in LDS, moved by LDS-DMA while this k-step is transformed and multiplied: registers
// hold 192 accumulators + one phase's values, nothing is held across the MFMA phase (tests/test_abi.py: no scratch).
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
// 1 no MFMAs, 2 no LDS-DMA fills after the first two k-steps, 4 no transform after the first two, 8 no operand reads
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

__device__ __forceinline__ float w2d_dpp(float v) {   // quad_perm [2,2,1,1]: the partner row of the B2^T column transform
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x5A, 0xf, 0xf, true));
}

namespace {
constexpr int WG2D_BM = 64, WG2D_BN = 32, WG2D_NT = 256;
constexpr int WG2D_TSU = WG2D_BM * 24 + 4, WG2D_TSV = WG2D_BN * 24 + 4;   // tile strides: an odd number of 16-byte slots (conflict-free b128 reads)
constexpr int WG2D_WINI = 10;                                              // window image: at most BN * 20 pieces = 10 fills of 1 KiB
constexpr int WG2D_DY = 0, WG2D_WIN = 2 * 4 * 256;                         // floats: dy slots [2 pieces][4 waves][256] | two window images |
constexpr int WG2D_IMG = WG2D_WIN + 2 * WG2D_WINI * 256;                   // ... the image [4 tiles][BM][24] + [4 tiles][BN][24] |
constexpr int WG2D_SCS = WG2D_IMG + 4 * WG2D_TSU + 4 * WG2D_TSV;           // ... (scale, shift)[BN]
constexpr int WG2D_LDS = WG2D_SCS + 2 * WG2D_BN;
}

// PLAIN: no activation segment carries a deferred BatchNorm / ReLU.
template <bool PLAIN>
__global__ __launch_bounds__(256, 2) void wgrad3x3_w2d_kernel(const WgW2dParams P) {
  constexpr int BM = WG2D_BM, BN = WG2D_BN, TSU = WG2D_TSU, TSV = WG2D_TSV, WINI = WG2D_WINI;
  constexpr int WIN = WG2D_WIN, IMG = WG2D_IMG, SCS = WG2D_SCS;
  constexpr int KB = 3;   // window fills per wave at most
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
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
  const int kxm = P.KX - 1;
  // U: (co, tile): tile fastest, so that a quad of lanes reads 64 contiguous bytes of a dy row
  const int u_t = tid & 3, u_co = tid >> 2;
  const int u_tyl = u_t >> P.kx_log2, u_txl = u_t & kxm;
  const unsigned u_voff = (unsigned)((long long)u_co * P.dy.cs + (long long)(2 * u_tyl) * P.dy.ws + 4 * u_txl) * 4u;
  const unsigned u_wr = (unsigned)(IMG + u_t * TSU + u_co * 24) * 4u;     // byte offset of the thread's 24 results in the image
  // V rows: (ci, tile, window row kr) and the same with ci + 16; lane quad = the four rows of one (ci, tile)
  const int v_kr = tid & 3, v_t = (tid >> 2) & 3, v_ci = tid >> 4;
  const int v_tyl = v_t >> P.kx_log2, v_txl = v_t & kxm;
  const unsigned v_rd = (unsigned)(WIN + ((v_ci * P.WR + 2 * v_tyl + v_kr) * P.NP + v_txl) * 4) * 4u;   // its 6 floats in window image 0
  const unsigned v_wr = (unsigned)(IMG + 4 * TSU + v_t * TSV + v_ci * 24 + v_kr * 6) * 4u;
  // (the tile coordinates inside the k-step are needed again in border k-steps only: one packed, opaque register -- hipcc would keep
  //  the fields in separate registers for the whole kernel, and the kernel lives at the register limit)
  int geo_packed = u_tyl | u_txl << 3 | v_tyl << 6 | v_txl << 9 | v_kr << 12;
  asm volatile("" : "+v"(geo_packed));
  const float v_sgn = (tid & 3) == 1 ? 1.f : -1.f;   // kr0: r0 - r2, kr1: r1 + r2, kr2: r2 - r1, kr3: r3 - r1
  // deferred BatchNorm of the block's BN channels: (scale, shift) pairs in LDS, read per task and k-step (one ds_read_b64 instead of
  // two registers per task held for the whole kernel)
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

  // Raw operands of a k-step wait in LDS (by LDS-DMA), not in registers (14 of them held across the MFMA phase did not fit 256):
  //   * dy: every thread moves ITS OWN two row pieces of 16 bytes into a private slot -- piece p of wave w occupies 1 KiB at
  //     (p * 4 + w) * 256 floats, lane l its bytes [16 l, 16 l + 16) -- and reads them back itself a k-step later: the only
  //     ordering needed is the thread's own vmcnt(0) before the read, and its reads having returned before the next fill;
  //   * activation windows: ONE image per block and k-step, [channel][window row][piece of 4 floats], the halo shared by the tiles
  //     of the k-step (80 instead of 128 bytes per row at 1 x 4 tiles) and moved as runs of consecutive pieces by consecutive lanes
  //     (fill i = pieces 64 i .. 64 i + 63 of the image, wave w issues fills w, w + 4, w + 8).  Another wave's pieces are read, so
  //     there are two window images (k-step parity) and the fills are published by a barrier (vmcnt(0) in front of it).
  int r_mask = 0;            // border k-steps: bit 0/1 dy row ok, bits 2..7: window columns of the V tasks ok
  bool r_edge = false;       // wave-uniform: the masks apply
  float* const raw_w = smem + WG2D_DY + wave * 256;                  // this wave's slot of piece 0 (wave-uniform: the DMA's LDS base)
  const float* const raw_r = smem + WG2D_DY + wave * 256 + lane * 4;   // this lane's 16 bytes of piece 0
  // window fills of this wave: piece 64 (wave + 4 k) + lane = (channel, window row, piece) -> byte offset from the k-step's window
  // origin in the first channel's plane; row and piece packed (8 bits per fill: row | piece << 4 | dummy << 7) for border k-steps
  unsigned x_off[KB];
  int x_meta = 0;
  {
    const int per_ch = P.WR * P.NP;
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int pid = 64 * (wave + 4 * k) + lane;
      const int ch = pid / per_ch, rm = pid - ch * per_ch;
      const int wr = rm / P.NP, pp = rm - wr * P.NP;
      const bool dummy = ch >= BN;
      x_off[k] = dummy ? 0u : (unsigned)((long long)ch * S_cs + (long long)wr * S_ws + 4 * pp) * 4u;
      x_meta |= (dummy ? 128 : (wr | pp << 4)) << (8 * k);
    }
    asm volatile("" : "+v"(x_meta));
  }

  // Addresses: a wave-uniform 64-bit base that depends on the IMAGE only (the block's first channel plane) plus an unsigned 32-bit
  // byte offset per lane = (k-step origin, scalar) + (task constant).  A lane whose piece must not be read where it lies takes
  // offset 0 instead.  (The first form of this code selected between 64-bit pointers per lane and held the raw values in registers;
  // at 256 registers hipcc then spilled the zero-extended offsets on some paths of the branchy prologue only and reloaded them on
  // all -- wild addresses, a memory fault at the 160x213 level.  No 64-bit vector address selects here.)
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
    const char* const vblk = reinterpret_cast<const char*>(S_p + (long long)n * S_ns + (long long)S_c0 * S_cs) - 16;
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
      asm volatile("" : "+v"(geo));   // (unpacked HERE, in border k-steps only: hipcc would hoist the fields out of the loop)
      {
        const int ty = ty0 + (geo & 7), tx = tx0 + (geo >> 3 & 7);
        const bool t_ok = ty < P.tiles_y && tx < P.tiles_x;
        const int h = 2 * ty;
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
        int cm = 0;
#pragma unroll
        for (int c = 0; c < 6; ++c) cm |= (r_ok && (unsigned)(c0 + c) < (unsigned)S_W) ? 1 << c : 0;
        m |= cm << 2;
      }
      r_mask = m;
    }
    auto fill = [&](const char* base, unsigned off, float* dst) __attribute__((always_inline)) {
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(base + off), dst, 16, 0, 0);
    };
    // window pieces: one with a column inside the segment lies within 3 floats of its row's ends (slack >= 4) and is read where it
    // lies, partly outside or not (the transform masks by column); one without is not read where it lies (offset 0)
#pragma unroll
    for (int k = 0; k < KB; ++k)
      if (wave + 4 * k < P.NI) {
        unsigned off = (unsigned)v_org + x_off[k];
        if (!inside) {
          int mt = x_meta;
          asm volatile("" : "+v"(mt));
          mt >>= 8 * k;
          const int row = hs + (mt & 15), c0 = wsx + 4 * (mt >> 4 & 7);
          const bool ok = !(mt & 128) && (unsigned)row < (unsigned)S_H && c0 + 3 >= 0 && c0 < S_W;
          off = ok ? off : 0u;
        }
        fill(vblk, off, smem + WIN + wb * (WINI * 256) + (wave + 4 * k) * 256);
      }
    // dy pieces go into the slots this thread has just read (read_raw): its reads have to have RETURNED before a fill can land
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
    fill(dblk, o_y0, raw_w);
    fill(dblk, o_y1, raw_w + 4 * 256);
  };

  // A k-step's iteration: (1) the thread's raw values of k-step `it` come from LDS into registers, (2) the fills of k-step it + 1
  // are issued -- their destinations are free: the dy slots were just read, the other window image was last read an iteration ago --
  // (3) the transforms are computed and stored, barrier, (4) the k-step is multiplied, (5) vmcnt(0) + barrier.  The fills have the
  // transform AND the MFMA phase to land.
  f32x4 t_y0, t_y1, t_ra[2];
  f32x2d t_rb[2];
  int t_mask = 0;
  bool t_edge = false;
  auto read_raw = [&](auto wb_c) __attribute__((always_inline)) {
    constexpr int wb = decltype(wb_c)::value;
    t_y0 = *reinterpret_cast<const f32x4*>(raw_r);
    t_y1 = *reinterpret_cast<const f32x4*>(raw_r + 4 * 256);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float* const wp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(smem) + v_rd) + wb * (WINI * 256) + i * 16 * (P.WR * P.NP * 4);
      t_ra[i] = *reinterpret_cast<const f32x4*>(wp);
      t_rb[i] = *reinterpret_cast<const f32x2d*>(wp + 4);
    }
    t_mask = r_mask;   // (of the k-step whose raw values these are: the next load() overwrites r_mask / r_edge)
    t_edge = r_edge;
  };

  // Packed fp32 math (v_pk_add_f32 / v_pk_fma_f32: two floats for the issue slot of one; an fp32 MFMA stream does not hide vector
  // instructions, profiles/r05_mfma_f32_issue_ubench.txt).
  auto transform = [&]() __attribute__((always_inline)) {
    const f32x2d p1m1 = {1.f, -1.f}, p2m2 = {2.f, -2.f}, c4 = {4.f, 4.f};
    // A4 of one dy row (y0..y3) -> (U1, U2), (U3, U4); U0 = y0, U5 = y3
    auto a4_row = [&](const f32x4& y, f32x2d& u12, f32x2d& u34) __attribute__((always_inline)) {
      const f32x2d lo2 = {y[0], y[1]}, hi2 = {y[2], y[3]};
      const f32x2d pq = lo2 + hi2;                                   // (y0 + y2, y1 + y3)
      const f32x2d ab = __builtin_elementwise_fma(c4, hi2, lo2);     // (y0 + 4 y2, y1 + 4 y3)
      u12 = __builtin_elementwise_fma(f32x2d{pq[1], pq[1]}, p1m1, f32x2d{pq[0], pq[0]});
      u34 = __builtin_elementwise_fma(f32x2d{ab[1], ab[1]}, p2m2, f32x2d{ab[0], ab[0]});
    };
    // ---- U: rows fr = [r0, r0 + r1, r0 - r1, r1] of A4-transformed dy rows, 24 consecutive floats as six 16-byte groups ----
    {
      f32x4 y0 = t_y0, y1 = t_y1;
      if (t_edge) {
        if (!(t_mask & 1)) y0 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!(t_mask & 2)) y1 = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      f32x2d a12, a34, b12, b34;
      a4_row(y0, a12, a34);
      a4_row(y1, b12, b34);
      const f32x2d s12 = a12 + b12, s34 = a34 + b34, d12 = a12 - b12, d34 = a34 - b34;
      const float s0 = y0[0] + y1[0], s5 = y0[3] + y1[3], d0 = y0[0] - y1[0], d5 = y0[3] - y1[3];
      float* const op = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + u_wr);
      *reinterpret_cast<f32x4*>(op) = f32x4{a12[0], a12[1], a34[0], a34[1]};
      *reinterpret_cast<f32x4*>(op + 4) = f32x4{y0[0], y0[3], s12[0], s12[1]};
      *reinterpret_cast<f32x4*>(op + 8) = f32x4{s34[0], s34[1], s0, s5};
      *reinterpret_cast<f32x4*>(op + 12) = f32x4{d12[0], d12[1], d34[0], d34[1]};
      *reinterpret_cast<f32x4*>(op + 16) = f32x4{d0, d5, b12[0], b12[1]};
      *reinterpret_cast<f32x4*>(op + 20) = f32x4{b34[0], b34[1], y1[0], y1[3]};
    }
    // ---- V rows ----
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const f32x4 ra = t_ra[i];
      f32x2d t0 = {ra[0], ra[1]}, t1 = {ra[2], ra[3]}, t2 = t_rb[i];
      if constexpr (!PLAIN) {
        const f32x2d ss = *reinterpret_cast<const f32x2d*>(smem + SCS + 2 * ((tid >> 4) + 16 * i));
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
      // B4^T of the window row d0..d5 = (t0, t1, t2), as gsd_conv3x3_w2d.hip forms it
      const f32x2d m41 = {-4.f, -1.f}, m5 = {-5.f, -5.f};
      const f32x2d ac = __builtin_elementwise_fma(f32x2d{t1[0], t1[0]}, m41, f32x2d{t2[0], t2[0]});   // (d4 - 4 d2, d4 - d2)
      const f32x2d be = __builtin_elementwise_fma(f32x2d{t0[1], t0[1]}, m41, f32x2d{t1[1], t1[1]});   // (d3 - 4 d1, d3 - d1)
      const f32x2d v12 = __builtin_elementwise_fma(f32x2d{be[0], be[0]}, p1m1, f32x2d{ac[0], ac[0]});
      const f32x2d v34 = __builtin_elementwise_fma(f32x2d{be[1], be[1]}, p2m2, f32x2d{ac[1], ac[1]});
      const f32x2d v05 = __builtin_elementwise_fma(t0, c4, __builtin_elementwise_fma(t1, m5, t2));
      // B2^T down the window column: the partner row from the lane's quad
      const f32x2d sg = {v_sgn, v_sgn};
      const f32x2d x12 = {w2d_dpp(v12[0]), w2d_dpp(v12[1])}, x34 = {w2d_dpp(v34[0]), w2d_dpp(v34[1])};
      const f32x2d x05 = {w2d_dpp(v05[0]), w2d_dpp(v05[1])};
      float* const op = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + v_wr) + i * (16 * 24);
      *reinterpret_cast<f32x2d*>(op) = __builtin_elementwise_fma(sg, x12, v12);
      *reinterpret_cast<f32x2d*>(op + 2) = __builtin_elementwise_fma(sg, x34, v34);
      *reinterpret_cast<f32x2d*>(op + 4) = __builtin_elementwise_fma(sg, x05, v05);
    }
  };

  f32x4 acc[2][24];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int f = 0; f < 24; ++f) acc[m][f] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int a_rd = IMG + j * TSU + (wm * 32 + l16) * 24;
  const int b_rd = IMG + 4 * TSU + j * TSV + (wn * 16 + l16) * 24;

  // One operand group = four frequencies of (channel l16, tile j) per ds_read_b128: 3 reads feed 8 MFMAs.  One register set: the
  // SIMD's other wave (of the CU's other block) multiplies or transforms while this one waits for its reads.
  auto multiply = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < 6; ++g) {
      f32x4 a0, a1, b;
      if constexpr (WG2D_ABL & 8) {
        a0 = a1 = b = f32x4{1.f, 2.f, 3.f, (float)g};
      } else {
        a0 = *reinterpret_cast<const f32x4*>(smem + a_rd + 4 * g);
        a1 = *reinterpret_cast<const f32x4*>(smem + a_rd + 16 * 24 + 4 * g);
        b = *reinterpret_cast<const f32x4*>(smem + b_rd + 4 * g);
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

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  if (nst > 0) {
    load(I0{});
    gsd_dma_barrier();   // vmcnt(0) + barrier: everyone's fills of the first k-step (and the BatchNorm pairs) are in
    auto step = [&](const int it, auto cur_c) __attribute__((always_inline)) {
      constexpr int cur = decltype(cur_c)::value;
      using NXT = std::integral_constant<int, cur ^ 1>;
      const bool tr = !((WG2D_ABL & 4) && it > 1);
      // The vector phase runs at raised priority: beside another block's wave that issues MFMAs back to back, vector instructions
      // of a wave of equal priority get one issue slot per MFMA (~40 cycles each, profiles/r05_mfma_f32_issue_ubench.txt)
      __builtin_amdgcn_s_setprio(3);
      if (tr) read_raw(cur_c);
      __builtin_amdgcn_sched_barrier(0);   // (phases in program order: the register budget is 192 accumulators + one phase's values)
      if (it + 1 < nst && !((WG2D_ABL & 2) && it > 1)) load(NXT{});
      __builtin_amdgcn_sched_barrier(0);
      if (tr) transform();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(0);
      // the image is complete: a RAW barrier behind lgkmcnt(0) -- __syncthreads() would also wait for the fills in flight (vmcnt(0)) ...
      __builtin_amdgcn_s_waitcnt(0xC07F);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      multiply();
      gsd_dma_barrier();      // ... which have until here to land: vmcnt(0) + barrier; everyone has left the image
    };
    for (int it = 0; it < nst; it += 2) {
      step(it, I0{});
      if (it + 1 < nst) step(it + 1, I1{});
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
    // fewest k-steps, weighted by what a k-step of that shape costs: 2 x 2 and 4 x 1 tiles fetch shorter runs of more rows
    // (measured per k-step against 1 x 4, batch 32: +2 ... +13 % and +13 ... +55 %, profiles/r06_wg2d_kstep_shapes.txt)
    const long steps = (long)ceil_div(p.tiles_y, ky) * ceil_div(p.tiles_x, kx) * (kx == 4 ? 100 : kx == 2 ? 106 : 130);
    if (best < 0 || steps < best) {
      best = steps;
      p.KY = ky; p.KX = kx;
    }
  }
  p.kx_log2 = p.KX == 4 ? 2 : p.KX == 2 ? 1 : 0;
  p.sy_n = ceil_div(p.tiles_y, p.KY);
  p.sx_n = ceil_div(p.tiles_x, p.KX);
  p.BM = WG2D_BM;
  p.BN = WG2D_BN;
  p.ok = M % p.BM == 0 && Ncols % p.BN == 0;
  p.mblocks = ceil_div(M, p.BM);
  p.nblocks = ceil_div(Ncols, p.BN);
  p.ksteps_total = N * p.sy_n * p.sx_n;
  const int target = gsd_env_int("GSD_WG2D_BLOCKS", 512);   // two blocks per CU
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
  if ((int64_t)p.BM * dy->c_stride * 4 >= (1LL << 31)) return 0;
  if (nsrc == 2 && a[0].C % p.BN != 0) return 0;
  for (int i = 0; i < nsrc; ++i) {
    if (a[i].slack < 4) return 0;
    if ((int64_t)p.BN * a[i].c_stride * 4 >= (1LL << 31)) return 0;
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
  const size_t lds = (size_t)WG2D_LDS * sizeof(float);   // 2 x 65.3 KiB per CU
  GSD_REQUIRE(P.NI <= 4 * 3 && P.NI <= WG2D_WINI, GSD_ERR_UNSUPPORTED, "gsd_conv3x3_wgrad (w2d): window image too large");
  if (gsd_env_set("GSD_WG43_TRACE"))   // tuning: one line per launch
    fprintf(stderr, "wg2d M%d N%d %dx%d B%d kstep %dx%d ksteps %d splits %d blocks %ld BM %d BN %d plain %d lds %zu\n", Cout, Cin, H, W,
            N, pl.KY, pl.KX, pl.ksteps_total, pl.splits, grid, pl.BM, pl.BN, (int)plain, lds);
  const dim3 g((int)grid);
  const hipStream_t st = (hipStream_t)stream;
#define WG2D_LAUNCH(PL_)                                                                                          \
  do {                                                                                                            \
    static gsd_attr_once once;                                                                                    \
    const void* fn = reinterpret_cast<const void*>(&wgrad3x3_w2d_kernel<PL_>);                                    \
    if (hipError_t e = gsd_allow_big_lds(once, fn); e != hipSuccess) {                                            \
      gsd_set_error("gsd_conv3x3_wgrad (w2d): hipFuncSetAttribute: %s", hipGetErrorString(e));                    \
      return GSD_ERR_HIP;                                                                                         \
    }                                                                                                             \
    hipLaunchKernelGGL((wgrad3x3_w2d_kernel<PL_>), g, dim3(WG2D_NT), lds, st, P);                                 \
  } while (0)
  if (plain) WG2D_LAUNCH(true);
  else WG2D_LAUNCH(false);
#undef WG2D_LAUNCH
  GSD_LAUNCH_CHECK("gsd_conv3x3_wgrad (w2d)");
  *splits_out = pl.splits;
  return GSD_OK;
}
