// Micro-test (tuning aid, not product): does global_load_lds_dwordx4 accept a GLOBAL address that is only 4-byte aligned
// (rows of 427 floats start at any multiple of 4 bytes), and what does it cost against aligned pieces and dword gathers?
//   correctness: one wave moves 64 x 16 bytes from src + shift floats; the image is read back.
//   timing: every block streams `iters` x 8 DMA instructions over its own region (x4 aligned / x4 shifted / dword), s_memtime.
//   hipcc --offload-arch=gfx950 -O3 dma_global_align.hip -o dma_global_align && ./dma_global_align
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k(const float* __restrict__ src, float* __restrict__ out, int shift_floats) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 512; i += 64) smem[i] = -1.f;
  __syncthreads();
  const float* g = src + shift_floats + lane * 4;
  float* l = smem + 16;
  __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  for (int i = lane; i < 512; i += 64) out[i] = smem[i];
}

// MODE 0: x4 aligned, 1: x4 shifted by `shift` floats, 2: dword (4x the instructions for the same bytes)
template <int MODE>
__global__ __launch_bounds__(256) void t(const float* __restrict__ src, long long* __restrict__ cyc, float* __restrict__ sink, int iters,
                                         int shift) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* base = src + ((size_t)blockIdx.x * 4 + wave) * (size_t)iters * 2048 + (MODE == 1 ? shift : 0);
  float* l = smem + wave * 2048;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const float* g = base + (size_t)it * 2048;
    if (MODE < 2) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float* gp = g + k * 256 + lane * 4;
        float* lp = l + k * 256;
        __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        const float* gp = g + k * 64 + lane;
        float* lp = l + k * 64;
        __builtin_amdgcn_global_load_lds(gp, lp, 4, 0, 0);
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
  }
  const long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
  __syncthreads();
  if (sink != nullptr) sink[threadIdx.x] = smem[threadIdx.x];
}

int main() {
  float *src, *out;
  hipMalloc(&src, 8192);
  hipMalloc(&out, 4096);
  std::vector<float> h(1024), o(512);
  for (int i = 0; i < 1024; ++i) h[i] = (float)i;
  hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
  for (int s = 0; s < 4; ++s) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, src, out, s);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(o.data(), out, 2048, hipMemcpyDeviceToHost);
    int bad = 0, first = -1;
    for (int i = 0; i < 256; ++i)
      if (o[16 + i] != (float)(i + s)) { ++bad; if (first < 0) first = i; }
    printf("global address +%d floats: %s, %d/256 wrong (first %d: got %g)\n", s, hipGetErrorString(e), bad, first,
           first >= 0 ? o[16 + first] : 0.f);
  }
  const int blocks = 512, iters = 64;
  const size_t n = (size_t)blocks * 4 * iters * 2048 + 64;
  float* big;
  long long* cyc;
  hipMalloc(&big, n * 4);
  hipMemset(big, 0, n * 4);
  hipMalloc(&cyc, blocks * 4 * 8);
  std::vector<long long> hc(blocks * 4);
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 3; ++mode) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(t<0>, dim3(blocks), dim3(256), 32768, 0, big, cyc, nullptr, iters, 1);
      if (mode == 1) hipLaunchKernelGGL(t<1>, dim3(blocks), dim3(256), 32768, 0, big, cyc, nullptr, iters, 1);
      if (mode == 2) hipLaunchKernelGGL(t<2>, dim3(blocks), dim3(256), 32768, 0, big, cyc, nullptr, iters, 1);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(hc.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
      double s = 0;
      for (auto c : hc) s += (double)c;
      const double gb = (double)blocks * 4 * iters * 8192 / 1e9;
      printf("mode %d (%s): %.3f ms, %.2f TB/s, %.0f cycles per wave per 8 KiB\n", mode,
             mode == 0 ? "x4 aligned" : (mode == 1 ? "x4 +1 float" : "dword"), ms, gb / ms, s / hc.size() / iters);
    }
  return 0;
}
