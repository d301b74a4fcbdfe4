// Micro-benchmark (tuning aid, not product): does a wave that issues LDS-DMA slow down the MFMA wave it shares a SIMD with?
// 8 waves per block (two per SIMD), no barrier in the loop.  Waves 0-3 run 72 MFMAs per iteration (+ a few LDS reads);
// waves 4-7 either idle (exit), or issue LDS-DMA continuously (throttled by a counted vmcnt), or run VALU-only work.
//   hipcc --offload-arch=gfx950 -O3 dma_partner.hip -o dma_partner && ./dma_partner
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PARTNER, int WIDTH>   // PARTNER 0 idle, 1 DMA issue, 2 VALU loop
__global__ __launch_bounds__(512, 1) void k(const float* __restrict__ src, unsigned piece_mask, float* out, unsigned long long* cyc, int iters,
                                           volatile int* stop) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 8192; i += 512) smem[i] = 0.001f * (i & 15);
  __syncthreads();
  if (wave >= 4) {
    if (PARTNER == 0) return;
    if (PARTNER == 1) {
      // about as long as the MFMA waves run: iters * 72 * 32 cycles / ~100 cycles per DMA
      const int n = iters * 20;
      for (int d = 0; d < n; ++d) {
        const unsigned piece = ((unsigned)blockIdx.x * 64u + (unsigned)(d * 4 + wave)) & piece_mask;
        if (WIDTH == 16)
          __builtin_amdgcn_global_load_lds(src + (size_t)piece * 256 + lane * 4, smem + 8192 + ((d & 7) * 4 + (wave & 3)) * 256, 16, 0, 0);
        else
          __builtin_amdgcn_global_load_lds(src + (size_t)piece * 256 + lane * 3, smem + 8192 + ((d & 7) * 4 + (wave & 3)) * 64, 4, 0, 0);
        __builtin_amdgcn_s_waitcnt(0x0F70 | 8);   // vmcnt(8): at most 8 in flight
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);
      return;
    }
    float v = lane * 0.5f;
    for (int d = 0; d < iters * 600; ++d) v = fmaf(v, 1.0001f, 0.5f);
    out[blockIdx.x * 512 + tid] = v;
    return;
  }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = 1.f + lane * 0.001f, b = 0.5f;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 72; ++m) {
      if ((m & 7) == 0) b = smem[(lane + m + it) & 4095];
      acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 7], 0, 0, 0);
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 512 + tid] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int PARTNER, int WIDTH>
void run(const char* name, const float* src, size_t nf, float* out, unsigned long long* cyc) {
  const int iters = 200, blocks = 256;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<PARTNER, WIDTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<PARTNER, WIDTH>), dim3(blocks), dim3(512), 98304, 0, src, (unsigned)(nf / 256) - 1u, out, cyc, iters, nullptr);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h(blocks * 4);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : h) s += (double)v;
  printf("%-52s %7.0f cycles per 72 MFMA (floor 2304)\n", name, s / h.size() / iters);
}

int main() {
  float *small, *out;
  unsigned long long* cyc;
  const size_t ns = 1u << 22;
  hipMalloc(&small, ns * 4); hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 1 << 16);
  hipMemset(small, 0, ns * 4);
  run<0, 16>("partner idle", small, ns, out, cyc);
  run<1, 16>("partner issues 1-KiB LDS-DMA pieces continuously", small, ns, out, cyc);
  run<1, 4>("partner issues dword LDS-DMA continuously", small, ns, out, cyc);
  run<2, 16>("partner runs a dependent VALU chain", small, ns, out, cyc);
  return 0;
}
