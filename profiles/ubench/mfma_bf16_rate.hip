// Micro-benchmark (tuning aid, not product): cycles per v_mfma_f32_16x16x32_bf16 for ONE wave per SIMD with the conv kernel's
// register pattern -- 32 accumulator tiles (4 m x 8 pixel tiles), A/B operands in registers -- and, optionally, the conv
// kernel's operand traffic (12 ds_read_b128 per 32 MFMAs) and its barrier per 96 MFMAs.  Random operands (the clock depends on
// the data).  Reports shader cycles per MFMA (s_memtime) and the clock the chip held (s_memtime / s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 mfma_bf16_rate.hip -o mfma_bf16_rate && ./mfma_bf16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// READS: 0 none (operands stay in registers), 1 the conv kernel's 12 ds_read_b128 per tap, interleaved as there, every tap's
//        operands read during the tap before it (also across the barrier); 2 the same but the FIRST tap's operands are read
//        after the iteration's barrier, as the conv kernel has to (the data lands with the barrier).
// RDOFF: byte offset of the operand image inside the 150-KiB LDS allocation (0, or 81920: reads above 64 KiB, fills below).
//        3 = the single-buffered two-blocks-per-CU model: {barrier, NDMA fills in a burst, vmcnt(0), barrier, first tap's reads,
//        96 MFMAs with the other taps' reads interleaved}; launch 2 x 256 blocks with <= 75 KiB of LDS each.
// NDMA:  LDS-DMA pieces (1 KiB, L2-resident source) per wave and iteration, one per micro-step from the iteration's start.
// BARRIER: 1 = s_barrier per 3 taps.  WAVES: waves per block (4 = one per SIMD, 8 = two).
template <int READS, int BARRIER, int WAVES, int NDMA, int DSTART, int DSTEP, int PW, int RDOFF, int OCC>
__global__ __launch_bounds__(WAVES * 64, OCC) void k(const unsigned* __restrict__ src, float* out, unsigned long long* cyc, int iters, size_t srcmask) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 49152 / 4; i += WAVES * 64) reinterpret_cast<unsigned*>(smem + RDOFF)[i] = src[(blockIdx.x * 97 + i) & 0xffff];
  __syncthreads();
  f32x4 acc[4][8];
  for (int m = 0; m < 4; ++m)
    for (int t = 0; t < 8; ++t) acc[m][t] = f32x4{0, 0, 0, 0};
  u32x4 a[2][4], b[8];
  const unsigned char* A0 = smem + RDOFF + (wave & 1) * 4096 + lane * 16;
  const unsigned char* B0 = smem + RDOFF + 16384 + (wave >> 1 & 1) * 12288 + (lane & 15) * 96 + (lane >> 4) * 16;
  for (int m = 0; m < 4; ++m) a[0][m] = a[1][m] = *reinterpret_cast<const u32x4*>(A0 + m * 1024);
  for (int t = 0; t < 8; ++t) b[t] = *reinterpret_cast<const u32x4*>(B0 + t * 1536);
  unsigned long long t0, r0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
  if (PW && wave >= 4) {   // producer wave: after each barrier, NDMA pieces (DSTART: s_sleep units before the first, DSTEP: between pieces)
    for (int it = 0; it < iters; ++it) {
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
      if (DSTART) __builtin_amdgcn_s_sleep(DSTART);
#pragma unroll
      for (int d = 0; d < NDMA; ++d) {
        __builtin_amdgcn_global_load_lds(src + (((size_t)blockIdx.x * 8 + wave + it * 16 + d) & srcmask) * 256 + lane * 4, smem + 49152 + (d % 6) * 4096 + (wave & 3) * 1024, 16, 0, 0);
        if (DSTEP > 1) __builtin_amdgcn_s_sleep(DSTEP);
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (lane == 0) cyc[(blockIdx.x * WAVES + wave) * 2] = 0, cyc[(blockIdx.x * WAVES + wave) * 2 + 1] = 1;
    return;
  }
  for (int it = 0; it < iters; ++it) {
    if (BARRIER) { if (NDMA && !PW) __builtin_amdgcn_s_waitcnt(0x0F70); __syncthreads(); }
    if (READS == 3) {
#pragma unroll
      for (int d = 0; d < NDMA; ++d)
        __builtin_amdgcn_global_load_lds(src + (((size_t)blockIdx.x * 8 + wave + it * 16 + d) & srcmask) * 256 + lane * 4, smem + 49152 + (d % 6) * 4096 + wave * 1024, 16, 0, 0);
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
    }
    if (READS == 2 || READS == 3) {
#pragma unroll
      for (int m = 0; m < 4; ++m) a[0][m] = *reinterpret_cast<const u32x4*>(A0 + (it & 3) * 2048 + m * 1024);
#pragma unroll
      for (int t = 0; t < 8; ++t) b[t] = *reinterpret_cast<const u32x4*>(B0 + t * 1536 + (it & 1) * 96);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = i % 4, t = 2 * (i / 4);
        acc[m][t] = mfma(a[kw & 1][m], b[t], acc[m][t]);
        acc[m][t + 1] = mfma(a[kw & 1][m], b[t + 1], acc[m][t + 1]);
        if (READS && !((READS == 2 || READS == 3) && kw == 2)) {
          if (i < 4) a[(kw + 1) & 1][i] = *reinterpret_cast<const u32x4*>(A0 + ((kw + 1 + it) & 3) * 8192 / 4 + i * 1024);
          if (i >= 4 && (i % 4) < 2) {
            const int bt = 2 * (i / 4 - 1) + (i % 4);
            b[bt] = *reinterpret_cast<const u32x4*>(B0 + bt * 1536 + ((kw + 1) & 1) * 96);
          }
          if (i < 2) b[6 + i] = *reinterpret_cast<const u32x4*>(B0 + (6 + i) * 1536 + (kw & 1) * 96);
        }
        {
          constexpr int ms = 0;  // placeholder
          const int step = kw * 16 + i - DSTART;
          if (!PW && READS != 3 && step >= 0 && step % DSTEP == 0 && step / DSTEP < NDMA)
            __builtin_amdgcn_global_load_lds(src + (((size_t)blockIdx.x * 8 + wave + it * 16 + step / DSTEP) & srcmask) * 256 + lane * 4, smem + (RDOFF ? 0 : 49152) + (step / DSTEP % 6) * 4096 + wave * 1024, 16, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  unsigned long long t1, r1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
  float s = 0;
  for (int m = 0; m < 4; ++m)
    for (int t = 0; t < 8; ++t) s += acc[m][t][0] + acc[m][t][3];
  out[blockIdx.x * WAVES * 64 + tid] = s;
  if (lane == 0) {
    cyc[(blockIdx.x * WAVES + wave) * 2] = t1 - t0;
    cyc[(blockIdx.x * WAVES + wave) * 2 + 1] = r1 - r0;
  }
}

template <int READS, int BARRIER, int WAVES, int NDMA, int DSTART = 0, int DSTEP = 1, int PW = 0, int RDOFF = 0, int OCC = 1>
void run(const char* name, const unsigned* src, float* out, unsigned long long* cyc, int blocks, size_t lds, size_t srcmask = 63) {
  const int iters = 4000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<READS, BARRIER, WAVES, NDMA, DSTART, DSTEP, PW, RDOFF, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<READS, BARRIER, WAVES, NDMA, DSTART, DSTEP, PW, RDOFF, OCC>), dim3(blocks), dim3(WAVES * 64), lds, 0, src, out, cyc, iters, srcmask);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<READS, BARRIER, WAVES, NDMA, DSTART, DSTEP, PW, RDOFF, OCC>), dim3(blocks), dim3(WAVES * 64), lds, 0, src, out, cyc, iters, srcmask);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((size_t)blocks * WAVES * 2);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> c, clk;
  for (int i = 0; i < blocks * WAVES; ++i) if (h[2 * i] > 0) { c.push_back((double)h[2 * i]); clk.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1); }
  std::sort(c.begin(), c.end()); std::sort(clk.begin(), clk.end());
  const double mf = 96.0 * iters;
  const double tf = 2.0 * 16 * 16 * 32 * mf * blocks * (PW ? 4 : WAVES) / (ms * 1e-3) / 1e12;
  printf("%-44s %6.2f cyc/MFMA/wave  clock %.2f GHz  %.3f ms  %7.1f TFLOP/s\n", name, c[c.size() / 2] / mf, clk[clk.size() / 2], ms, tf);
}

int main() {
  unsigned* src; float* out; unsigned long long* cyc;
  const size_t SRCW = (size_t)64 << 20;   // 256 MiB of source words: fills that miss L2
  hipMalloc(&src, SRCW * 4); hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&cyc, 1024 * 8 * 2 * 8);
  std::vector<unsigned> h(65536);
  srand(1);
  for (auto& v : h) {   // two random bf16 in [-1, 1)
    auto bf = [](float f) { unsigned u; memcpy(&u, &f, 4); return u >> 16; };
    v = bf(rand() / (float)RAND_MAX * 2 - 1) | (bf(rand() / (float)RAND_MAX * 2 - 1) << 16);
  }
  for (size_t o = 0; o < SRCW; o += h.size()) hipMemcpy(src + o, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const size_t big = 150 * 1024;   // one block per CU, as in the conv kernel
  const size_t hbm = (SRCW / 256) - 1;   // piece mask over the whole 256-MiB source
  run<1, 1, 4, 0>("reads ahead + barrier", src, out, cyc, 256, big);
  run<2, 1, 4, 0>("first tap read after the barrier", src, out, cyc, 256, big);
  run<2, 1, 4, 9, 2, 4>("product model: first tap after barrier + 9 DMA (L2 source)", src, out, cyc, 256, big);
  run<2, 1, 4, 9, 2, 4>("product model, fills from HBM", src, out, cyc, 256, big, hbm);
  run<1, 1, 4, 9, 2, 4>("reads ahead + 9 DMA, fills from HBM", src, out, cyc, 256, big, hbm);
  run<3, 1, 4, 6>("single-buffered, ONE block/CU, 6 DMA burst (L2)", src, out, cyc, 256, big);
  run<3, 1, 4, 6, 0, 1, 0, 0, 2>("single-buffered, TWO blocks/CU, 6 DMA burst (L2)", src, out, cyc, 512, 75 * 1024);
  run<3, 1, 4, 6, 0, 1, 0, 0, 2>("single-buffered, TWO blocks/CU, 6 DMA burst (HBM)", src, out, cyc, 512, 75 * 1024, hbm);
  run<3, 1, 4, 11, 0, 1, 0, 0, 2>("single-buffered, TWO blocks/CU, 11 DMA burst (HBM)", src, out, cyc, 512, 75 * 1024, hbm);
  run<2, 1, 4, 9, 2, 4, 0, 0, 2>("double-buffered product model, TWO blocks/CU (L2)", src, out, cyc, 512, 75 * 1024);
  return 0;
}
