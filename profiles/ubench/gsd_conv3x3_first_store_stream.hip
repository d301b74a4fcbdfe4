// DIAGNOSTIC SOURCE, not built into libgsd.so (round 4): a store-stream forward kernel for the fp32 first layer.
// Correct (6 op-test shapes vs the oracle at 2e-5) and exactly as fast as the general direct-tap kernel: 0.539 vs 0.538 ms at batch 32
// (profiles/bench_first_fp32.py; 0.634 ms before the four 64-byte segments of a channel row were made consecutive instructions).
// What bounds both is the write pattern, not the arithmetic: 64-byte runs into 64 channel planes whose rows (427 floats) are never
// 128-byte aligned, i.e. partial-line writes -- a plain fill of the same 1.12 GB takes 0.16 ms (profiles/bench_write_bw.py).

// gsd_conv3x3_first.hip -- forward of the network's FIRST conv3x3 in fp32 (unet.py:11 for `inc`: 3 -> 64 channels at 320 x 427).
//
// 27 (ci, tap) products per output against 64 output channels over 4.4 M pixels is 15 GFLOP -- 0.1 ms of matrix-core time --
// under 1.12 GB of raw output to write: a store-stream kernel.  The general direct-tap kernel (gsd_conv3x3.hip: LDS-DMA gathers
// and a weight image sized for >= 16 input channels, the deferred-BatchNorm machinery at every operand read) spends 0.45 ms on
// it -- 12.6 vector instructions per MFMA, matrix pipes 0.31 busy (profiles/r04_f_pmc_sq_summary.txt) -- where a plain fill of
// the same 1.12 GB takes 0.16 ms (profiles/bench_write_bw.py).
//
// GEMM view on v_mfma_f32_16x16x4_f32, with the PIXELS on the MFMA's row dimension:
//   D[m = pixel][n = channel] = sum_k  patch[pixel][k] * W[channel][k],   k = ci*9 + tap, padded 27 -> 28 = 7 k-steps of 4
// so that an accumulator register holds 4 CONSECUTIVE pixels of one channel: one 16-byte store per lane, 64 contiguous bytes
// per channel and 16-pixel tile, instead of the four dword stores a channels-on-rows tile needs.  The weights (B operand,
// k = lane>>4, channel = lane&15) live in registers for the whole kernel, read from the module's (Cout, Cin, 3, 3) tensor as it
// is -- no weight-layout launch.  The patches (A operand) are gathered from an LDS window of the block's 4 + 2 image rows; the
// next tile's window is fetched into registers while this one is multiplied and stored.  BatchNorm partial sums of the raw
// output are kept per lane (its channel, its four pixels) and leave as one row per block.
#include "gsd_common.h"

namespace {

constexpr int CF_TH = 4, CF_TW = 64;            // pixel tile of a block: one image row of 64 pixels per wave
constexpr int CF_PITCH = CF_TW + 2;             // window row pitch (floats)
constexpr int CF_PLANE = (CF_TH + 2) * CF_PITCH;
constexpr int CF_KS = 7;                        // k-steps of 4: 9 * Cin <= 28

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

struct CFirstP {
  const float* x;     // (N, C, H, W) contiguous
  const float* w;     // (M, C, 3, 3) contiguous: the module's own weights
  float* out;         // (N, M, H, W) contiguous
  float* partials;    // [gridDim.x][2 * Mpad] or null
  int N, C, H, W, M, Mpad;
  int tiles_y, tiles_x, ntiles;
};

template <int NT>   // NT = M / 16 channel tiles
__global__ __launch_bounds__(256, 3) void conv3x3_first_kernel(const CFirstP P) {
  __shared__ float xs[3 * CF_PLANE + 4];   // [c][row][col], + a zero element for k >= 9 C
  __shared__ float sSt[4][2][16 * NT];
  constexpr int ZERO = 3 * CF_PLANE;
  constexpr int NXE = (3 * CF_PLANE + 255) / 256;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, i = lane & 15;   // A operand: row (pixel) i, k = g; B operand: k = g, column (channel) i; D: pixels 4g..4g+3 of channel i
  if (tid < 4) xs[ZERO + tid] = 0.f;

  // B operands: W[channel nt*16 + i][k = 4s + g], zero for k >= 9 C
  float bw[CF_KS][NT];
#pragma unroll
  for (int s = 0; s < CF_KS; ++s)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int k = 4 * s + g;
      bw[s][nt] = k < 9 * P.C ? P.w[(size_t)(nt * 16 + i) * (9 * P.C) + k] : 0.f;
    }
  // A operand gather: k = 4s + g -> (channel c, tap t): window offset of the tap relative to the output pixel
  int off[CF_KS];
#pragma unroll
  for (int s = 0; s < CF_KS; ++s) {
    const int k = 4 * s + g;
    const int c = k / 9, t = k - c * 9;
    off[s] = k < 9 * P.C ? c * CF_PLANE + (t / 3) * CF_PITCH + (t % 3) : -1;
  }
  float s1[NT], s2[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) s1[nt] = s2[nt] = 0.f;

  // this thread's window elements: (channel, row, column) packed, -1 past the window
  int xel[NXE];
#pragma unroll
  for (int k = 0; k < NXE; ++k) {
    const int e = tid + k * 256;
    const int c = e / CF_PLANE, r = (e - c * CF_PLANE) / CF_PITCH, col = e - c * CF_PLANE - r * CF_PITCH;
    xel[k] = e < P.C * CF_PLANE ? (c << 16 | r << 8 | col) : -1;
  }
  const int tpi = P.tiles_y * P.tiles_x;
  float xv[NXE];
  auto fetch = [&](int tile) {
    const int n = tile / tpi, rem = tile - n * tpi;
    const int ty = rem / P.tiles_x, h0 = ty * CF_TH, w0 = (rem - ty * P.tiles_x) * CF_TW;
    const float* xn = P.x + (size_t)n * P.C * P.H * P.W;
#pragma unroll
    for (int k = 0; k < NXE; ++k) {
      float v = 0.f;
      if (xel[k] >= 0) {
        const int c = xel[k] >> 16, gh = h0 - 1 + (xel[k] >> 8 & 255), gw = w0 - 1 + (xel[k] & 255);
        if ((unsigned)gh < (unsigned)P.H && (unsigned)gw < (unsigned)P.W) v = xn[((size_t)c * P.H + gh) * P.W + gw];
      }
      xv[k] = v;
    }
  };
  int tile = blockIdx.x;
  if (tile < P.ntiles) fetch(tile);
  for (; tile < P.ntiles; tile += gridDim.x) {
    const int n = tile / tpi, rem = tile - n * tpi;
    const int ty = rem / P.tiles_x, h0 = ty * CF_TH, w0 = (rem - ty * P.tiles_x) * CF_TW;
    __syncthreads();   // everyone has left the previous window
#pragma unroll
    for (int k = 0; k < NXE; ++k)
      if (xel[k] >= 0) xs[tid + k * 256] = xv[k];
    __syncthreads();
    if (tile + (int)gridDim.x < P.ntiles) fetch(tile + gridDim.x);   // flies during this tile's MFMAs and stores
    const int h = h0 + wave;
    if (h < P.H) {     // wave-uniform
      float* orow = P.out + ((size_t)n * P.M * P.H + h) * P.W;     // + channel * H * W + w
      // all 4 x NT accumulator tiles of the row first, the stores afterwards channel tile by channel tile: the four 64-byte
      // segments a lane group writes for one channel are then consecutive instructions (256 contiguous bytes per channel row
      // arrive together instead of ~1000 cycles apart)
      f32x4 acc[4][NT];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {     // 16 pixels of the wave's row
        const int base = wave * CF_PITCH + mt * 16 + i;
        float a[CF_KS];
#pragma unroll
        for (int s = 0; s < CF_KS; ++s) a[s] = xs[off[s] >= 0 ? base + off[s] : ZERO];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < CF_KS; ++s)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma16(a[s], bw[s][nt], acc[mt][nt]);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float* och = orow + (size_t)(nt * 16 + i) * P.H * P.W;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int w = w0 + mt * 16 + 4 * g;     // this lane's four pixels: w .. w+3 of channel nt*16 + i
          const int nv = P.W - w;                  // how many of them exist
          const f32x4 v = acc[mt][nt];
          if (nv >= 4) {
            *reinterpret_cast<f32x4u*>(och + w) = v;
            s1[nt] += (v[0] + v[1]) + (v[2] + v[3]);
            s2[nt] = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], fmaf(v[3], v[3], s2[nt]))));
          } else if (nv > 0) {
#pragma unroll
            for (int e = 0; e < 3; ++e)
              if (e < nv) {
                och[w + e] = v[e];
                s1[nt] += v[e];
                s2[nt] = fmaf(v[e], v[e], s2[nt]);
              }
          }
        }
      }
    }
  }
  if (P.partials != nullptr) {   // one partial row per block: the four lane groups of a channel, then the four waves, through LDS
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float a1 = s1[nt], a2 = s2[nt];
      a1 += __shfl_xor(a1, 16, 64); a1 += __shfl_xor(a1, 32, 64);
      a2 += __shfl_xor(a2, 16, 64); a2 += __shfl_xor(a2, 32, 64);
      if (g == 0) {
        sSt[wave][0][nt * 16 + i] = a1;
        sSt[wave][1][nt * 16 + i] = a2;
      }
    }
    __syncthreads();
    if (tid < P.M) {
      float* row = P.partials + (size_t)blockIdx.x * (2 * P.Mpad);
      row[tid] = (sSt[0][0][tid] + sSt[1][0][tid]) + (sSt[2][0][tid] + sSt[3][0][tid]);
      row[P.Mpad + tid] = (sSt[0][1][tid] + sSt[1][1][tid]) + (sSt[2][1][tid] + sSt[3][1][tid]);
    }
  }
}

int cfirst_grid(long ntiles) { return (int)(ntiles < 2048 ? ntiles : 2048); }   // 8 small blocks per CU; one partial row each

}  // namespace

extern "C" int gsd_conv3x3_first_supported(int N, int H, int W, int Cin, int Cout) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  if (gsd_env_int("GSD_CONV_FIRST", 1) == 0) return 0;
  if (9 * Cin > 4 * CF_KS || (Cout != 16 && Cout != 32 && Cout != 48 && Cout != 64)) return 0;
  return (long)N * ceil_div(H, CF_TH) * ceil_div(W, CF_TW) < 2147483647L ? 1 : 0;
}

extern "C" int gsd_conv3x3_first_partial_rows(int N, int H, int W) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  return cfirst_grid((long)N * ceil_div(H, CF_TH) * ceil_div(W, CF_TW));
}

extern "C" int gsd_conv3x3_first(const float* x, const float* w, int Cin, int Cout, float* out, float* partials, int N, int H,
                                 int W, void* stream) {
  GSD_REQUIRE(x && w && out, GSD_ERR_BAD_ARG, "gsd_conv3x3_first: null argument");
  GSD_REQUIRE(gsd_conv3x3_first_supported(N, H, W, Cin, Cout), GSD_ERR_UNSUPPORTED,
              "gsd_conv3x3_first: serves 9*Cin <= 28 and Cout in {16, 32, 48, 64} (got %d -> %d); use gsd_conv3x3", Cin, Cout);
  CFirstP P;
  P.x = x; P.w = w; P.out = out; P.partials = partials;
  P.N = N; P.C = Cin; P.H = H; P.W = W; P.M = Cout; P.Mpad = round_up(Cout, 64);
  P.tiles_y = ceil_div(H, CF_TH); P.tiles_x = ceil_div(W, CF_TW);
  P.ntiles = N * P.tiles_y * P.tiles_x;
  const int grid = cfirst_grid(P.ntiles);
  hipStream_t st = (hipStream_t)stream;
  switch (Cout / 16) {
    case 1: hipLaunchKernelGGL(conv3x3_first_kernel<1>, dim3(grid), dim3(256), 0, st, P); break;
    case 2: hipLaunchKernelGGL(conv3x3_first_kernel<2>, dim3(grid), dim3(256), 0, st, P); break;
    case 3: hipLaunchKernelGGL(conv3x3_first_kernel<3>, dim3(grid), dim3(256), 0, st, P); break;
    default: hipLaunchKernelGGL(conv3x3_first_kernel<4>, dim3(grid), dim3(256), 0, st, P); break;
  }
  GSD_LAUNCH_CHECK("gsd_conv3x3_first");
  return GSD_OK;
}
