// Micro-benchmark (tuning aid, not product): the conv kernel's K-loop model of mfma_bf16_rate.hip -- a 64 x 128 wave tile (128
// accumulator registers), 12 ds_read_b128 per 32-channel tap, a barrier per 3 taps, NDMA LDS-DMA pieces per iteration -- with the
// tap's products issued as 16 v_mfma_f32_32x32x16_bf16 (32 cycles each) instead of 32 v_mfma_f32_16x16x32_bf16 (16 cycles each).
// Same FLOPs, same operand bytes, HALF the MFMA instructions: an MFMA holds the SIMD's vector issue for 8 cycles of its 16 or of
// its 32, so the reads and fills that share one wave's issue port have three times the room.  SHAPE 16 / 32 selects the instruction.
//   hipcc --offload-arch=gfx950 -O3 mfma_bf16_rate32.hip -o mfma_bf16_rate32.bin && ./mfma_bf16_rate32.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// READS 1: every tap's operands are read during the tap before it (also across the barrier); 2: the first tap's after the barrier
template <int SHAPE, int READS, int NDMA, int DSTART, int DSTEP>
__global__ __launch_bounds__(256) void k(const unsigned* __restrict__ src, float* out, unsigned long long* cyc, int iters, size_t srcmask) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 49152 / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = src[(blockIdx.x * 97 + i) & 0xffff];
  __syncthreads();
  // 12 operand registers sets per tap in both shapes: 4 "A" reads + 8 "B" reads of 16 bytes per lane
  u32x4 a[2][4], b[8];
  const unsigned char* A0 = smem + (wave & 1) * 4096 + lane * 16;
  const unsigned char* B0 = smem + 16384 + (wave >> 1 & 1) * 12288 + (lane & 15) * 96 + (lane >> 4) * 16;
  for (int m = 0; m < 4; ++m) a[0][m] = a[1][m] = *reinterpret_cast<const u32x4*>(A0 + m * 1024);
  for (int t = 0; t < 8; ++t) b[t] = *reinterpret_cast<const u32x4*>(B0 + t * 1536);
  f32x4 acc4[4][8];
  f32x16 acc16[2][4];
  for (int m = 0; m < 4; ++m)
    for (int t = 0; t < 8; ++t) acc4[m][t] = f32x4{0, 0, 0, 0};
  for (int m = 0; m < 2; ++m)
    for (int t = 0; t < 4; ++t)
      for (int e = 0; e < 16; ++e) acc16[m][t][e] = 0.f;
  unsigned long long t0, r0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
  for (int it = 0; it < iters; ++it) {
    if (NDMA) __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (READS == 2) {
#pragma unroll
      for (int m = 0; m < 4; ++m) a[0][m] = *reinterpret_cast<const u32x4*>(A0 + (it & 3) * 2048 + m * 1024);
#pragma unroll
      for (int t = 0; t < 8; ++t) b[t] = *reinterpret_cast<const u32x4*>(B0 + t * 1536 + (it & 1) * 96);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      // 16 micro-steps per tap; micro-step i carries one or two operand reads for the next tap and (maybe) one fill
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (SHAPE == 16) {
          const int m = i % 4, t = 2 * (i / 4);
          acc4[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[kw & 1][m]), __builtin_bit_cast(bf16x8, b[t]), acc4[m][t], 0, 0, 0);
          acc4[m][t + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[kw & 1][m]), __builtin_bit_cast(bf16x8, b[t + 1]), acc4[m][t + 1], 0, 0, 0);
        } else {
          // ONE 32x32x16 per micro-step: (k-half kh, m-tile mt, pixel tile nt): operand registers a[.][2 kh + mt], b[2 nt + kh] (same 12 registers per tap)
          const int kh = i / 8, mt = (i / 4) & 1, nt = i % 4;
          acc16[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[kw & 1][2 * kh + mt]), __builtin_bit_cast(bf16x8, b[2 * nt + kh]), acc16[mt][nt], 0, 0, 0);
        }
        if (!(READS == 2 && kw == 2)) {
          if (i < 4) a[(kw + 1) & 1][i] = *reinterpret_cast<const u32x4*>(A0 + ((kw + 1 + it) & 3) * 8192 / 4 + i * 1024);
          if (SHAPE == 16) {
            if (i >= 4 && (i % 4) < 2) {
              const int bt = 2 * (i / 4 - 1) + (i % 4);
              b[bt] = *reinterpret_cast<const u32x4*>(B0 + bt * 1536 + ((kw + 1) & 1) * 96);
            }
            if (i < 2) b[6 + i] = *reinterpret_cast<const u32x4*>(B0 + (6 + i) * 1536 + (kw & 1) * 96);
          } else {
            // the k-half 0 B registers (b[0,2,4,6]) die after micro-step 7, the k-half 1 ones with the tap: re-read in place
            if (i >= 8 && i < 12) b[2 * (i - 8)] = *reinterpret_cast<const u32x4*>(B0 + 2 * (i - 8) * 1536 + ((kw + 1) & 1) * 96);
            if (i < 4) b[2 * i + 1] = *reinterpret_cast<const u32x4*>(B0 + (2 * i + 1) * 1536 + (kw & 1) * 96);
          }
        }
        const int step = kw * 16 + i - DSTART;
        if (NDMA && step >= 0 && step % DSTEP == 0 && step / DSTEP < NDMA)
          __builtin_amdgcn_global_load_lds(src + (((size_t)blockIdx.x * 8 + wave + it * 16 + step / DSTEP) & srcmask) * 256 + lane * 4, smem + 49152 + (step / DSTEP % 6) * 4096 + wave * 1024, 16, 0, 0);
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
    for (int t = 0; t < 8; ++t) s += acc4[m][t][0] + acc4[m][t][3];
  for (int m = 0; m < 2; ++m)
    for (int t = 0; t < 4; ++t) s += acc16[m][t][0] + acc16[m][t][15];
  out[blockIdx.x * 256 + tid] = s;
  if (lane == 0) {
    cyc[(blockIdx.x * 4 + wave) * 2] = t1 - t0;
    cyc[(blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0;
  }
}

template <int SHAPE, int READS, int NDMA, int DSTART = 0, int DSTEP = 1>
void run(const char* name, const unsigned* src, float* out, unsigned long long* cyc, size_t srcmask = 63) {
  const int iters = 4000, blocks = 256;
  const size_t lds = 150 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<SHAPE, READS, NDMA, DSTART, DSTEP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<SHAPE, READS, NDMA, DSTART, DSTEP>), dim3(blocks), dim3(256), lds, 0, src, out, cyc, iters, srcmask);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<SHAPE, READS, NDMA, DSTART, DSTEP>), dim3(blocks), dim3(256), lds, 0, src, out, cyc, iters, srcmask);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((size_t)blocks * 4 * 2);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> c, clk;
  for (int i = 0; i < blocks * 4; ++i) if (h[2 * i] > 0) { c.push_back((double)h[2 * i]); clk.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1); }
  std::sort(c.begin(), c.end()); std::sort(clk.begin(), clk.end());
  const double taps = 3.0 * iters;                 // a tap = 32 16x16x32 or 16 32x32x16: 512 MFMA cycles
  const double tf = 2.0 * 64 * 128 * 32 * taps * blocks * 4 / (ms * 1e-3) / 1e12;
  printf("%-64s %6.1f cycles per tap (512 = the matrix pipe's)  clock %.2f GHz  %.3f ms  %7.1f TFLOP/s\n", name, c[c.size() / 2] / taps, clk[clk.size() / 2], ms, tf);
}

int main() {
  unsigned* src; float* out; unsigned long long* cyc;
  const size_t SRCW = (size_t)64 << 20;
  hipMalloc(&src, SRCW * 4); hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&cyc, 1024 * 8 * 2 * 8);
  std::vector<unsigned> h(65536);
  srand(1);
  for (auto& v : h) {
    auto bf = [](float f) { unsigned u; memcpy(&u, &f, 4); return u >> 16; };
    v = bf(rand() / (float)RAND_MAX * 2 - 1) | (bf(rand() / (float)RAND_MAX * 2 - 1) << 16);
  }
  for (size_t o = 0; o < SRCW; o += h.size()) hipMemcpy(src + o, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const size_t hbm = (SRCW / 256) - 1;
  run<16, 1, 0>("16x16x32: reads ahead + barrier", src, out, cyc);
  run<32, 1, 0>("32x32x16: reads ahead + barrier", src, out, cyc);
  run<16, 2, 0>("16x16x32: first tap read after the barrier", src, out, cyc);
  run<32, 2, 0>("32x32x16: first tap read after the barrier", src, out, cyc);
  run<16, 2, 9, 2, 4>("16x16x32: product model (first tap after barrier + 9 DMA, HBM)", src, out, cyc, hbm);
  run<32, 2, 9, 2, 4>("32x32x16: product model (first tap after barrier + 9 DMA, HBM)", src, out, cyc, hbm);
  run<16, 2, 9, 2, 1>("16x16x32: product model, the 9 DMA back to back", src, out, cyc, hbm);
  run<32, 2, 9, 2, 1>("32x32x16: product model, the 9 DMA back to back", src, out, cyc, hbm);
  return 0;
}
