// Micro-benchmark (tuning aid, not product): does changing M0 (the LDS base of an LDS-DMA) between two global_load_lds cost
// the issuing wave anything?  Every wave issues 8 one-KiB pieces per iteration, then 72 MFMAs, then waits + barrier:
//   A: eight different LDS bases (eight M0 writes)          B: two bases, four pieces each through the instruction's
//   immediate offset (which moves the global AND the LDS address) -- same bytes, same addresses.
//   hipcc --offload-arch=gfx950 -O3 dma_m0.hip -o dma_m0 && ./dma_m0
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ src, unsigned piece_mask, float* out, unsigned long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 12288; i += 256) smem[i] = 0.001f * (i & 15);
  __syncthreads();
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = 1.f + lane * 0.001f, b = 0.5f;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
    const unsigned piece = ((unsigned)blockIdx.x * 64u + (unsigned)(it * 32 + wave * 8)) & piece_mask;
    const float* g = src + (size_t)piece * 256 + lane * 4;      // 8 KiB contiguous per wave
    float* l = smem + 4096 + wave * 2048;                        // 8 KiB contiguous per wave
    if (MODE == 0) {
#pragma unroll
      for (int d = 0; d < 8; ++d) __builtin_amdgcn_global_load_lds(g + d * 256, l + d * 256, 16, 0, 0);
    } else if (MODE == 1) {
#pragma unroll
      for (int grp = 0; grp < 2; ++grp) {
        __builtin_amdgcn_global_load_lds(g + grp * 1024, l + grp * 1024, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(g + grp * 1024, l + grp * 1024, 16, 1024, 0);
        __builtin_amdgcn_global_load_lds(g + grp * 1024, l + grp * 1024, 16, 2048, 0);
        __builtin_amdgcn_global_load_lds(g + grp * 1024, l + grp * 1024, 16, 3072, 0);
      }
    }
#pragma unroll
    for (int m = 0; m < 72; ++m) {
      if ((m & 7) == 0) b = smem[(lane + m + it) & 4095];
      acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 7], 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  // checksum of the landed image so that wrong LDS placement shows
  float chk = 0;
  for (int i = lane; i < 2048; i += 64) chk += smem[4096 + wave * 2048 + i];
  out[blockIdx.x * 256 + tid] = s + chk;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int MODE>
double run(const char* name, const float* src, size_t nf, float* out, unsigned long long* cyc, int blocks) {
  const int iters = 200;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 49152 + 4096 * 4, 0, src, (unsigned)(nf / 256) - 1u, out, cyc, iters);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h(blocks * 4);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<float> o(blocks * 256);
  hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost);
  double s = 0, chk = 0;
  for (auto v : h) s += (double)v;
  for (auto v : o) chk += v;
  printf("%-44s %d blocks: %7.0f cycles/iter/wave   checksum %.6e\n", name, blocks, s / h.size() / iters, chk);
  return s;
}

int main() {
  float *small, *out;
  unsigned long long* cyc;
  const size_t ns = 1u << 22;   // 16 MiB
  hipMalloc(&small, ns * 4); hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 1 << 16);
  std::vector<float> h(ns);
  for (size_t i = 0; i < ns; ++i) h[i] = (float)((i * 2654435761u >> 20) & 1023) * 1e-3f;
  hipMemcpy(small, h.data(), ns * 4, hipMemcpyHostToDevice);
  for (int blocks : {256, 512}) {
    run<2>("no DMA", small, ns, out, cyc, blocks);
    run<0>("8 pieces, 8 LDS bases (M0 per piece)", small, ns, out, cyc, blocks);
    run<1>("8 pieces, 2 LDS bases + immediate offsets", small, ns, out, cyc, blocks);
  }
  return 0;
}
