// Sandbox for the conv3x3 K-chunk loop structure (tuning aid, not product).  4-wave blocks, wave tile 64x64.
// Parameters: KS k-steps per chunk (9 = 4 channels, 18 = 8 channels), DMA issue position, blocks per CU.
// hipcc --offload-arch=gfx950 -O3 conv_loop2.hip -o conv_loop2 && ./conv_loop2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// POS: 0 = all DMA right after the barrier; 1 = DMA spread: one slot after each k-step's MFMAs; 2 = no DMA
template <int KS, int POS, int NWV>
__global__ __launch_bounds__(NWV * 64) void k(float* out, const float* __restrict__ g, int chunks, float sc, float sh) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BMc = NWV == 4 ? 128 : 64, NT_ = NWV * 64;
  constexpr int WT = KS * 4 * BMc, PS = 208, NCH = KS * 4 / 9, BUF = WT + NCH * PS;
  constexpr int NWI = (WT / 4 + NT_ - 1) / NT_;           // weight DMA x4 per thread
  constexpr int NXI = NCH * 4 / NWV;                      // X DMA per wave (4 chunks of 64 positions per channel / 4 waves)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane >> 4, l16 = lane & 15, wm = NWV == 4 ? wave >> 1 : 0, wn = wave & 1;
  for (int i = tid; i < 2 * BUF; i += NT_) smem[i] = (float)(i & 7) * 0.25f;
  __syncthreads();
  f32x4 acc[4][4];
  for (int m = 0; m < 4; ++m) for (int t = 0; t < 4; ++t) acc[m][t] = f32x4{0, 0, 0, 0};
  int baddr[4];
  for (int t = 0; t < 4; ++t) baddr[t] = WT + j * PS + ((wn * 4 + t) * 16 + l16) / 32 * 34 + ((wn * 4 + t) * 16 + l16) % 32;
  const float* gw = g + (size_t)(blockIdx.x % 64) * 4608 + tid * 4;
  const float* gx = g + 1000000 + (size_t)(blockIdx.x % 1024) * 4096 + lane;
  auto dma_slot = [&](int c, int slot, float* Wb) {
    if (slot < NWI) {
      if (tid + slot * NT_ < WT / 4) __builtin_amdgcn_global_load_lds(gw + ((size_t)c * 4608) % 262144 + slot * NT_ * 4, Wb + (slot * NT_ + wave * 64) * 4, 16, 0, 0);
    } else if (slot < NWI + NXI) {
      const int q = slot - NWI;
      __builtin_amdgcn_global_load_lds(gx + (q / (4 / NWV)) * 640 + ((q % (4 / NWV)) * NWV + wave) * 64 + (c & 63) * 4096, Wb + WT + (q / (4 / NWV)) * PS + ((q % (4 / NWV)) * NWV + wave) * 64, 4, 0, 0);
    }
  };
  for (int c = 0; c < chunks; ++c) {
    const int cur = c & 1;
    __syncthreads();
    float* Wb = smem + (cur ^ 1) * BUF;
    if (POS == 0) {
#pragma unroll
      for (int s = 0; s < NWI + NXI; ++s) dma_slot(c, s, Wb);
    }
    const float* Wc = smem + cur * BUF;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      float a[4], b[4];
      const f32x4 av = *reinterpret_cast<const f32x4*>(&Wc[(j * KS + s) * BMc + wm * 64 + l16 * 4]);
#pragma unroll
      for (int m = 0; m < 4; ++m) a[m] = av[m];
#pragma unroll
      for (int t = 0; t < 4; ++t) b[t] = fmaxf(fmaf(Wc[baddr[t] + ((s % 9) / 3) * 34 + (s % 3) + (s / 9) * 4 * PS], sc, sh), 0.f);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[t], acc[m][t], 0, 0, 0);
      if (POS == 1) {
        constexpr int PER = (NWI + NXI + KS - 1) / KS;
#pragma unroll
        for (int q = 0; q < PER; ++q) dma_slot(c, s * PER + q, Wb);
      }
    }
  }
  float s = 0;
  for (int m = 0; m < 4; ++m) for (int t = 0; t < 4; ++t) s += acc[m][t][0] + acc[m][t][3];
  out[blockIdx.x * NT_ + tid] = s;
}

template <int KS, int POS, int NWV>
void run(const char* name, float* out, const float* g) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<KS, POS, NWV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  constexpr int BUF = KS * 4 * (NWV == 4 ? 128 : 64) + (KS * 4 / 9) * 208;
  printf("%-44s", name);
  for (int wps = 1; wps <= 3; ++wps) {            // target waves per SIMD
    const int bpc = wps * 4 / NWV;                 // blocks per CU
    size_t lds = (size_t)2 * BUF * 4;
    const size_t cap = 150 * 1024 / bpc;
    if (lds > cap) { printf("  %dw/SIMD    n/a", wps); continue; }
    if (wps < 3) lds = cap - 8 * 1024 > lds ? cap - 8 * 1024 : lds;   // pad LDS so that only bpc blocks fit
    const int blocks = 256 * bpc * 4, chunks = 3600 / KS;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KS, POS, NWV>), dim3(blocks), dim3(NWV * 64), lds, 0, out, g, 10, 1.01f, 0.1f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KS, POS, NWV>), dim3(blocks), dim3(NWV * 64), lds, 0, out, g, chunks, 1.01f, 0.1f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * NWV * chunks * KS * 16 * 2048.0;
    printf("  %dw/SIMD %6.1f", wps, flop / (ms * 1e-3) / 1e12);
  }
  printf("  TFLOP/s\n");
}
int main() {
  float *out, *g;
  (void)hipMalloc(&out, 256 * 6 * 4 * 256 * 4);
  (void)hipMalloc(&g, 64 << 20);
  (void)hipMemset(g, 0, 64 << 20);
  run<9, 2, 4>("4-wave blocks, no DMA", out, g);
  run<9, 0, 4>("4-wave blocks, DMA after barrier", out, g);
  run<9, 2, 2>("2-wave blocks, no DMA", out, g);
  run<9, 0, 2>("2-wave blocks, DMA after barrier", out, g);
  run<9, 1, 2>("2-wave blocks, DMA spread", out, g);
  return 0;
}
