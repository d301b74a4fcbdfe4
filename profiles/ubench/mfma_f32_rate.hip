// v_mfma_f32_16x16x4_f32 issue rate: one or two waves per SIMD, accumulators in AGPRs or VGPRs, with or without a vector
// instruction in every gap.  usage: ./mfma_f32_rate   (prints TFLOP/s and cycles per MFMA and SIMD at the measured clock)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NACC>   // MODE bit 0: accumulators "+v" instead of "+a"; bit 1: a v_fma between MFMAs; bit 2: builtin instead of asm
__global__ __launch_bounds__(512) void rate_kernel(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x, b = b0, x = a0, y = b0;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 xx = {a0, b0}, bb = {b0, a0};
  int sc = 7;
  f32x4 ld = {0.f, 0.f, 0.f, 0.f};
  __shared__ float lds[4096];
  lds[threadIdx.x] = a0;
  const int laddr = (threadIdx.x & 63) * 16;
  const bool valu_only = (MODE & 256) && (threadIdx.x >> 6) >= 4;   // 8-wave blocks: waves 4-7 only do vector work
  if (valu_only) {
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(b));
    }
    out[blockIdx.x * 512 + threadIdx.x] = x + (float)(__builtin_readcyclecounter() - t0);
    return;
  }
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if constexpr (MODE & 4) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      } else if constexpr (MODE & 1) {
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      } else {
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
      }
      if constexpr (MODE & 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(b));
      if constexpr (MODE & 8) asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %2, %2, %1, %1" : "+v"(x), "+v"(y) : "v"(b));
      if constexpr (MODE & 16) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(xx) : "v"(bb));
      if constexpr (MODE & 32) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc) : : "scc");
      if constexpr (MODE & 512) asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1" : "+s"(sc) : : "scc");
      if constexpr (MODE & 1024) asm volatile("s_nop 0");
      if constexpr (MODE & 2048) asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %2, %2, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %2, %2, %1, %1" : "+v"(x), "+v"(y) : "v"(b));
      if constexpr (MODE & 64) asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"(laddr));
      if constexpr (MODE & 128) asm volatile("v_mov_b32 %0, %1" : "=v"(y) : "v"(b));
    }
  }
  long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float s = x + y + xx[0] + xx[1] + (float)sc + ld[0];
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (float)(t1 - t0);
}

template <int MODE, int NACC>
void run(const char* name, int blocks, int threads = 256) {
  float* out;
  hipMalloc(&out, ((1 << 20) + 16) * sizeof(float));
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  rate_kernel<MODE, NACC><<<blocks, threads>>>(out, 100, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  rate_kernel<MODE, NACC><<<blocks, threads>>>(out, iters, 1.f, 2.f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  float cyc; hipMemcpy(&cyc, out + (1 << 20), 4, hipMemcpyDeviceToHost);
  const double mf = (double)blocks * 4 * iters * NACC;       // MFMAs (4 waves per block)
  const double tf = mf * 2048 / (ms * 1e-3) / 1e12;
  const double waves_per_simd = blocks / 256.0;
  printf("%-58s blocks %4d  %.3f ms  %7.1f TFLOP/s  %.1f shader cycles per MFMA of a wave (%.1f per SIMD)\n", name, blocks, ms, tf,
         cyc / ((double)iters * NACC), cyc / ((double)iters * NACC) / waves_per_simd);
  hipFree(out);
}

int main() {
  run<0, 32>("asm, AGPR acc, 32 chains, 1 wave/SIMD", 256);
  run<0, 32>("asm, AGPR acc, 32 chains, 2 waves/SIMD", 512);
  run<1, 32>("asm, VGPR acc, 32 chains, 1 wave/SIMD", 256);
  run<1, 32>("asm, VGPR acc, 32 chains, 2 waves/SIMD", 512);
  run<2, 32>("asm, AGPR acc + v_fma per gap, 1 wave/SIMD", 256);
  run<2, 32>("asm, AGPR acc + v_fma per gap, 2 waves/SIMD", 512);
  run<3, 32>("asm, VGPR acc + v_fma per gap, 1 wave/SIMD", 256);
  run<4, 32>("builtin, 32 chains, 1 wave/SIMD", 256);
  run<4, 32>("builtin, 32 chains, 2 waves/SIMD", 512);
  run<8, 32>("AGPR acc + 2 v_fma per gap, 1 wave/SIMD", 256);
  run<16, 32>("AGPR acc + v_pk_fma per gap, 1 wave/SIMD", 256);
  run<32, 32>("AGPR acc + s_add per gap, 1 wave/SIMD", 256);
  run<512, 32>("AGPR acc + 4 s_add per gap, 1 wave/SIMD", 256);
  run<512, 32>("AGPR acc + 4 s_add per gap, 2 waves/SIMD", 512);
  run<1024, 32>("AGPR acc + s_nop 0 per gap, 1 wave/SIMD", 256);
  run<2048, 32>("AGPR acc + 4 v_fma per gap, 1 wave/SIMD", 256);
  run<2048, 32>("AGPR acc + 4 v_fma per gap, 2 waves/SIMD", 512);
  run<2, 32>("AGPR acc + v_fma per gap, 2 waves/SIMD (again)", 512);
  run<64, 32>("AGPR acc + ds_read_b128 per gap, 1 wave/SIMD", 256);
  run<128, 32>("AGPR acc + v_mov per gap, 1 wave/SIMD", 256);
  run<256, 32>("MFMA waves 0-3 + v_fma-only waves 4-7 (8-wave blocks)", 256, 512);
  return 0;
}
