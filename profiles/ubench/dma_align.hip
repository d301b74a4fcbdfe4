// Does global_load_lds_dwordx4 accept (a) a 4-byte-aligned (not 16-byte-aligned) per-lane global source and
// (b) a 4-byte-aligned LDS destination base?  (gfx950 tuning probe, not product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* g, float* out, int src_off, int dst_off) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 512; i += 64) smem[i] = -1.f;
  __syncthreads();
  __builtin_amdgcn_global_load_lds(g + src_off + lane * 4, smem + dst_off, 16, 0, 0);
  __syncthreads();
  for (int i = lane; i < 512; i += 64) out[i] = smem[i];
}
int main() {
  std::vector<float> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = (float)i;
  float *g, *o;
  (void)hipMalloc(&g, 4096 * 4); (void)hipMalloc(&o, 512 * 4);
  (void)hipMemcpy(g, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  for (int so = 0; so < 4; ++so)
    for (int d = 0; d < 4; ++d) {
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, g, o, so, d);
      std::vector<float> r(512);
      (void)hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
      int bad = 0;
      for (int i = 0; i < 256; ++i) if (r[d + i] != (float)(so + i)) ++bad;
      printf("src_off %d dst_off %d : %s (bad %d)  first: %g %g %g %g %g\n", so, d, bad ? "MISMATCH" : "ok", bad, r[0], r[1], r[2], r[3], r[4]);
    }
  return 0;
}
