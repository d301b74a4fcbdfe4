// Does an out-of-range lane of `buffer_load_dwordx4 ... lds` (LDS-DMA through a buffer descriptor) write ZEROS to its 16 bytes of LDS,
// or leave them alone?  (If zeros: the zero padding of a halo tile needs no zero line, no select and no branch -- an out-of-image
// piece is just an offset beyond num_records.)   hipcc --offload-arch=gfx950 -O3 buffer_lds_oob.hip -o buffer_lds_oob && ./buffer_lds_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, float* o, unsigned nbytes) {
  extern __shared__ unsigned char sm[];
  float* f = (float*)sm;
  for (int i = threadIdx.x; i < 1024; i += 256) f[i] = -7.f;   // garbage the DMA must overwrite
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned voff = (threadIdx.x & 1) ? 0xFFFFFFF0u : threadIdx.x * 16;   // odd lanes: far beyond num_records
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(sm + w * 1024), 16, voff, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 256) o[i] = f[i];
}
int main() {
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 1.f + i;
  float *x, *o;
  hipMalloc(&x, 4096); hipMalloc(&o, 4096);
  hipMemcpy(x, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, x, o, 4096u);
  std::vector<float> r(1024);
  hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
  int zeros = 0, kept = 0, good = 0, other = 0;
  for (int t = 0; t < 256; ++t)
    for (int e = 0; e < 4; ++e) {
      const float v = r[t * 4 + e];
      if (t & 1) { if (v == 0.f) ++zeros; else if (v == -7.f) ++kept; else ++other; }
      else { if (v == 1.f + t * 4 + e) ++good; else ++other; }
    }
  printf("in-range dwords correct %d / 512; out-of-range dwords: zero %d, untouched %d; other %d\n", good, zeros, kept, other);
  return 0;
}
