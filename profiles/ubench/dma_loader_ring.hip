// Micro-benchmark (tuning aid, not product): producer / consumer block on gfx950 -- 4 MFMA waves + NL loader waves that issue
// ALL the LDS-DMA of the block (NX4 1-KiB pieces + NDW dword pieces per iteration) into a RING-deep ring of LDS images, one
// barrier per iteration.  The loader fills image (it + RING - 1) while the MFMA waves work on image it; before the barrier it
// waits with a COUNTED vmcnt that leaves the younger fills in flight.  Prints cycles per iteration per MFMA wave against the
// MFMA floor (72 MFMA x 32 cycles x blocks per CU).
//   hipcc --offload-arch=gfx950 -O3 dma_loader_ring.hip -o dma_loader_ring && ./dma_loader_ring
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void wait_vmcnt() {   // gfx9 encoding: vmcnt = bits[3:0] | bits[15:14] << 4; expcnt/lgkmcnt untouched
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}

template <int NL, int RING, int NX4, int NDW>
__global__ __launch_bounds__(256 + 64 * NL, 3) void k(const float* __restrict__ src, unsigned piece_mask, float* out, unsigned long long* cyc,
                                                     int iters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int IMG = 6144;   // floats per ring image (24 KiB)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < RING * IMG; i += 256 + 64 * NL) smem[i] = 0.001f * (i & 15);
  __syncthreads();
  unsigned long long t0, t1;
  if (wave >= 4) {   // ---- loader ----
    const int lw = wave - 4;
    constexpr int PER = (NX4 + NDW + NL - 1) / NL;   // DMA instructions per loader wave and iteration (upper bound)
    auto fill = [&](int it) {
      float* img = smem + (it % RING) * IMG;
      const unsigned base = ((unsigned)blockIdx.x * 977u + (unsigned)it * 64u) & piece_mask;
#pragma unroll
      for (int d = 0; d < NX4; ++d)
        if (d % NL == lw) __builtin_amdgcn_global_load_lds(src + (size_t)((base + d) & piece_mask) * 256 + lane * 4, img + (d % 20) * 256, 16, 0, 0);
#pragma unroll
      for (int d = 0; d < NDW; ++d)
        if ((d + NX4) % NL == lw)
          __builtin_amdgcn_global_load_lds(src + (size_t)((base + 32 + d) & piece_mask) * 256 + lane * 3, img + 5120 + (d % 16) * 64, 4, 0, 0);
    };
#pragma unroll
    for (int r = 0; r < RING - 1; ++r) fill(r);
    for (int it = 0; it < iters; ++it) {
      // image `it` must have landed: leave the RING-2 younger fills in flight
      if (RING >= 3) wait_vmcnt<(RING - 2) * PER>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      fill(it + RING - 1);
    }
    wait_vmcnt<0>();
    return;
  }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = 1.f + lane * 0.001f;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    const float* img = smem + (it % RING) * IMG;
    f32x4 b4 = *reinterpret_cast<const f32x4*>(&img[(lane * 4) & 4095]);
#pragma unroll
    for (int m = 0; m < 72; ++m) {
      if ((m & 3) == 3) b4 = *reinterpret_cast<const f32x4*>(&img[(lane * 4 + m * 64) & 4095]);
      acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b4[m & 3], acc[m & 7], 0, 0, 0);
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + tid] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int NL, int RING, int NX4, int NDW>
void run(const char* name, const float* src, size_t nf, float* out, unsigned long long* cyc, int blocks) {
  const int iters = 200;
  const size_t lds = (size_t)RING * 6144 * 4;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<NL, RING, NX4, NDW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<NL, RING, NX4, NDW>), dim3(blocks), dim3(256 + 64 * NL), lds, 0, src, (unsigned)(nf / 256) - 1u, out, cyc, iters);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%s: %s\n", name, hipGetErrorString(e)); return; }
  }
  std::vector<unsigned long long> h(blocks * 4);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : h) s += (double)v;
  printf("%-34s NL %d ring %d  %2d x4 + %2d dword  %s src, %d blocks: %7.0f cycles/iter  (MFMA floor %d)\n", name, NL, RING, NX4, NDW,
         nf > (64u << 20) ? "HBM" : "L2 ", blocks, s / h.size() / iters, 2304 * (blocks > 256 ? 2 : 1));
}

int main() {
  float *small, *big, *out;
  unsigned long long* cyc;
  const size_t ns = 1u << 20, nb = 1u << 28;   // 4 MiB, 1 GiB
  hipMalloc(&small, ns * 4); hipMalloc(&big, nb * 4); hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 1 << 16);
  hipMemset(small, 0, ns * 4); hipMemset(big, 0, nb * 4);
  for (int blocks : {256, 512}) {
    run<1, 2, 18, 24>("conv-like fill", small, ns, out, cyc, blocks);
    run<1, 3, 18, 24>("conv-like fill", small, ns, out, cyc, blocks);
    run<1, 3, 18, 24>("conv-like fill", big, nb, out, cyc, blocks);
    run<2, 3, 18, 24>("conv-like fill", big, nb, out, cyc, blocks);
    run<1, 3, 18, 8>("conv-like fill, x4 halo", small, ns, out, cyc, blocks);
    run<1, 3, 18, 8>("conv-like fill, x4 halo", big, nb, out, cyc, blocks);
    run<1, 2, 18, 8>("conv-like fill, x4 halo", big, nb, out, cyc, blocks);
    run<1, 3, 8, 0>("light fill", big, nb, out, cyc, blocks);
  }
  return 0;
}
