// Micro-benchmark (tuning aid, not product): the conv3x3 K-chunk loop of gsd_conv3x3.hip rebuilt feature by feature,
// to see which ingredient costs MFMA throughput on gfx950.  One block = 4 waves, wave tile 64x64 (16 MFMA 16x16x4 per
// k-step, 9 k-steps per chunk).
//   F&1  LDS operand reads (1 b128 A + 4 b32 B per k-step)      F&2  deferred-BN transform on B (fma+max)
//   F&4  one __syncthreads per chunk                             F&8  LDS-DMA of the next chunk (5 dwordx4 + 8 dword per wave)
//   F&16 double-buffer address switching (cur^1)
// hipcc --offload-arch=gfx950 -O3 conv_loop.hip -o conv_loop && ./conv_loop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int F>
__global__ __launch_bounds__(256) void k(float* out, const float* __restrict__ g, int chunks, float sc, float sh) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int WT = 36 * 128, PS = 208, BUF = WT + 4 * PS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane >> 4, l16 = lane & 15, wm = wave >> 1, wn = wave & 1;
  for (int i = tid; i < 2 * BUF; i += 256) smem[i] = (float)(i & 7) * 0.25f;
  __syncthreads();
  float a[4] = {1.f, 2.f, 3.f, 4.f}, b[4] = {1.f, 0.5f, 0.25f, 2.f};
  f32x4 acc[4][4];
  for (int m = 0; m < 4; ++m) for (int t = 0; t < 4; ++t) acc[m][t] = f32x4{0, 0, 0, 0};
  int baddr[4];
  for (int t = 0; t < 4; ++t) baddr[t] = WT + j * PS + ((wn * 4 + t) * 16 + l16) / 32 * 34 + ((wn * 4 + t) * 16 + l16) % 32;
  const float* gw = g + (size_t)blockIdx.x % 64 * 4608 + tid * 4;
  const float* gx = g + 1000000 + (size_t)(blockIdx.x % 1024) * 4096 + lane;
  for (int c = 0; c < chunks; ++c) {
    const int cur = (F & 16) ? (c & 1) : 0;
    if (F & 4) __syncthreads();
    if (F & 8) {
      float* Wb = smem + (cur ^ 1) * BUF;
#pragma unroll
      for (int i = 0; i < 5; ++i)
        if (tid + i * 256 < 1152) __builtin_amdgcn_global_load_lds(gw + (size_t)c * 4608 % 262144 + i * 1024, Wb + (i * 256 + wave * 64) * 4, 16, 0, 0);
#pragma unroll
      for (int ch = 0; ch < 4; ++ch)
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
          if (wave + 4 * pp < 4) __builtin_amdgcn_global_load_lds(gx + ch * 640 + (wave + 4 * pp) * 64 + (c & 63) * 4096, Wb + WT + ch * PS + (wave + 4 * pp) * 64, 4, 0, 0);
    }
    const float* Wc = smem + cur * BUF;
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      if (F & 1) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(&Wc[(j * 9 + s) * 128 + wm * 64 + l16 * 4]);
#pragma unroll
        for (int m = 0; m < 4; ++m) a[m] = av[m];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float v = Wc[baddr[t] + (s / 3) * 34 + (s % 3)];
          if (F & 2) v = fmaxf(fmaf(v, sc, sh), 0.f);
          b[t] = v;
        }
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[t], acc[m][t], 0, 0, 0);
    }
  }
  float s = 0;
  for (int m = 0; m < 4; ++m) for (int t = 0; t < 4; ++t) s += acc[m][t][0] + acc[m][t][3];
  out[blockIdx.x * 256 + tid] = s;
}

template <int F>
void run(const char* name, float* out, const float* g) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<F>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  printf("%-44s", name);
  for (int bpc = 1; bpc <= 3; ++bpc) {
    const size_t lds = bpc == 1 ? 100 * 1024 : (bpc == 2 ? 60 * 1024 : 44 * 1024);
    const int blocks = 256 * bpc * 4, chunks = 400;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<F>), dim3(blocks), dim3(256), lds, 0, out, g, 10, 1.01f, 0.1f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<F>), dim3(blocks), dim3(256), lds, 0, out, g, chunks, 1.01f, 0.1f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * chunks * 9 * 16 * 2048.0;
    printf("  %d/CU %6.1f", bpc, flop / (ms * 1e-3) / 1e12);
  }
  printf("  TFLOP/s\n");
}
int main() {
  float *out, *g;
  (void)hipMalloc(&out, 256 * 3 * 4 * 256 * 4);
  (void)hipMalloc(&g, 64 << 20);
  (void)hipMemset(g, 0, 64 << 20);
  run<0>("mfma only", out, g);
  run<1>("+ LDS reads", out, g);
  run<3>("+ LDS reads + transform", out, g);
  run<7>("+ LDS reads + transform + barrier", out, g);
  run<15>("+ LDS + transform + barrier + DMA", out, g);
  run<31>("+ LDS + transform + barrier + DMA + dbuf", out, g);
  run<8 | 4>("barrier + DMA only", out, g);
  run<1 | 4>("LDS reads + barrier", out, g);
  return 0;
}
