// Micro-benchmark (tuning aid, not product): what does it cost a wave to ISSUE global_load_lds instructions on gfx950?
// One block per CU (or two), W waves; mode selects who issues:  every wave issues NDMA pieces per iteration between MFMA
// groups, or only wave 0 ("loader") does.  Pieces: 1 KiB contiguous (dwordx4) or 256 B dword gathers.  The source buffer is
// small (L2 resident) or large (HBM).  Reports cycles per iteration per wave (s_memtime), i.e. how far the DMA issue
// stretches a loop of 72 MFMAs.
//   hipcc --offload-arch=gfx950 -O3 dma_issue.hip -o dma_issue && ./dma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NDMA, int WIDTH, int LOADER, int MFMAS>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ src, size_t src_floats, float* out, unsigned long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 16384; i += 256) smem[i] = 0.001f * (i & 15);
  __syncthreads();
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = 1.f + lane * 0.001f, b = 0.5f;
  const unsigned piece_mask = (unsigned)(src_floats / 256) - 1u;
  unsigned long long t0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
    const bool issue = LOADER ? wave == 0 : true;
    if (issue) {
      const int n = LOADER ? NDMA * 4 : NDMA;
#pragma unroll 1
      for (int d = 0; d < n; ++d) {
        // cheap address: 32-bit piece index masked to the (power-of-two) buffer size
        const unsigned piece = ((unsigned)blockIdx.x * 256u + (unsigned)(it * 64 + d * 4 + wave)) & piece_mask;
        const float* g = src + (size_t)piece * 256 + lane * (WIDTH / 4);
        if (WIDTH == 16)
          __builtin_amdgcn_global_load_lds(g, smem + 4096 + ((d * 4 + wave) & 15) * 256, 16, 0, 0);
        else
          __builtin_amdgcn_global_load_lds(g, smem + 4096 + ((d * 4 + wave) & 15) * 64, 4, 0, 0);
      }
    }
#pragma unroll
    for (int m = 0; m < MFMAS; ++m) {
      if ((m & 7) == 0) b = smem[(lane + m + it) & 4095];
      acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 7], 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
  }
  unsigned long long t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + tid] = s + smem[4096 + tid];
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int NDMA, int WIDTH, int LOADER, int MFMAS>
void run(const char* name, const float* src, size_t nf, float* out, unsigned long long* cyc, int blocks) {
  const int iters = 200;
  hipLaunchKernelGGL((k<NDMA, WIDTH, LOADER, MFMAS>), dim3(blocks), dim3(256), 65536 + 16384, 0, src, nf, out, cyc, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NDMA, WIDTH, LOADER, MFMAS>), dim3(blocks), dim3(256), 65536 + 16384, 0, src, nf, out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 4);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : h) s += (double)v;
  const double per_iter = s / h.size() / iters;
  printf("%-44s %s src, %d blocks: %8.0f cycles/iter/wave  (MFMA floor %d)  %.3f ms\n", name, nf > (64u << 20) ? "HBM" : "L2 ", blocks,
         per_iter, MFMAS * 32 * (blocks > 256 ? 2 : 1), ms);
}

int main() {
  float *small, *big, *out;
  unsigned long long* cyc;
  const size_t ns = 1u << 20, nb = 1u << 28;   // 4 MiB, 1 GiB
  hipMalloc(&small, ns * 4); hipMalloc(&big, nb * 4); hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 1 << 16);
  hipMemset(small, 0, ns * 4); hipMemset(big, 0, nb * 4);
  for (int blocks : {256, 512}) {
    run<0, 16, 0, 72>("no DMA, 72 MFMA", small, ns, out, cyc, blocks);
    run<10, 16, 0, 72>("every wave: 10 x 1KiB pieces + 72 MFMA", small, ns, out, cyc, blocks);
    run<10, 16, 0, 72>("every wave: 10 x 1KiB pieces + 72 MFMA", big, nb, out, cyc, blocks);
    run<10, 4, 0, 72>("every wave: 10 x dword pieces + 72 MFMA", small, ns, out, cyc, blocks);
    run<10, 4, 0, 72>("every wave: 10 x dword pieces + 72 MFMA", big, nb, out, cyc, blocks);
    run<10, 16, 1, 72>("wave 0 only: 40 x 1KiB pieces + 72 MFMA", small, ns, out, cyc, blocks);
    run<10, 16, 1, 72>("wave 0 only: 40 x 1KiB pieces + 72 MFMA", big, nb, out, cyc, blocks);
    run<10, 16, 0, 0>("every wave: 10 x 1KiB pieces, no MFMA", small, ns, out, cyc, blocks);
    run<10, 16, 0, 0>("every wave: 10 x 1KiB pieces, no MFMA", big, nb, out, cyc, blocks);
    run<4, 16, 0, 72>("every wave: 4 x 1KiB pieces + 72 MFMA", small, ns, out, cyc, blocks);
    run<4, 16, 0, 72>("every wave: 4 x 1KiB pieces + 72 MFMA", big, nb, out, cyc, blocks);
  }
  return 0;
}
