// Micro-test (tuning aid, not product): does global_load_lds_dwordx4 accept an LDS destination whose wave-uniform base is
// only 4-byte aligned (base + 4, + 8, + 12 bytes)?  Each lane moves 16 bytes; the image is then read back with ds_read_b32.
//   hipcc --offload-arch=gfx950 -O3 dma_lds_align.hip -o dma_lds_align && ./dma_lds_align
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k(const float* __restrict__ src, float* __restrict__ out, int shift_floats) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 512; i += 64) smem[i] = -1.f;
  __syncthreads();
  __builtin_amdgcn_global_load_lds(src + lane * 4, smem + 16 + shift_floats, 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  for (int i = lane; i < 512; i += 64) out[i] = smem[i];
}

int main() {
  float *src, *out;
  hipMalloc(&src, 4096);
  hipMalloc(&out, 4096);
  std::vector<float> h(256), o(512);
  for (int i = 0; i < 256; ++i) h[i] = (float)i;
  hipMemcpy(src, h.data(), 1024, hipMemcpyHostToDevice);
  for (int s = 0; s < 4; ++s) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, src, out, s);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(o.data(), out, 2048, hipMemcpyDeviceToHost);
    int bad = 0, first = -1;
    for (int i = 0; i < 256; ++i)
      if (o[16 + s + i] != (float)i) { ++bad; if (first < 0) first = i; }
    printf("lds base +%d floats: %s, %d/256 wrong (first %d: got %g), before %g after %g\n", s, hipGetErrorString(e), bad, first,
           first >= 0 ? o[16 + s + first] : 0.f, o[15 + s], o[16 + s + 256]);
  }
  return 0;
}
