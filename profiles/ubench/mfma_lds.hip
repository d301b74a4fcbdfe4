// Micro-benchmark (tuning aid, not product): what a 64x64-per-wave fp32 MFMA tile loop sustains on gfx950 when every
// k-step's operands come from LDS, for 16x16x4 (16 MFMA + 8 ds_read_b32 per K=4) and 32x32x2 (8 MFMA + 8 ds_read_b32),
// at 1/2/3 blocks (of 4 waves) per CU.   hipcc --offload-arch=gfx950 -O3 mfma_lds.hip -o mfma_lds && ./mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND, bool READ>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  extern __shared__ float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += 256) smem[i] = (float)(i & 7) * 0.25f;
  __syncthreads();
  float a[4] = {1.f, 2.f, 3.f, 4.f}, b[4] = {1.f, 0.5f, 0.25f, 2.f};
  if constexpr (KIND == 0) {
    f32x4 acc[4][4];
    for (int m = 0; m < 4; ++m) for (int t = 0; t < 4; ++t) acc[m][t] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 9; ++s) {
        if (READ) {
#pragma unroll
          for (int m = 0; m < 4; ++m) a[m] = smem[(s * 4 + (lane >> 4)) * 144 + (lane & 15) + m * 16 + (it & 3) * 4];
#pragma unroll
          for (int t = 0; t < 4; ++t) b[t] = smem[5184 + (lane >> 4) * 208 + t * 18 + (lane & 15) + s + (it & 3) * 2];
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[t], acc[m][t], 0, 0, 0);
      }
    }
    float s = 0;
    for (int m = 0; m < 4; ++m) for (int t = 0; t < 4; ++t) s += acc[m][t][0] + acc[m][t][3];
    out[blockIdx.x * 256 + tid] = s;
  } else {
    f32x16 acc[2][2];
    for (int m = 0; m < 2; ++m) for (int t = 0; t < 2; ++t) for (int r = 0; r < 16; ++r) acc[m][t][r] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 18; ++s) {   // 18 K=2 steps == 9 K=4 steps
        if (READ) {
#pragma unroll
          for (int m = 0; m < 2; ++m) a[m] = smem[(s * 2 + (lane >> 5)) * 144 + (lane & 31) + m * 32 + (it & 3) * 4];
#pragma unroll
          for (int t = 0; t < 2; ++t) b[t] = smem[5184 + (lane >> 5) * 208 + t * 34 + (lane & 31) + s + (it & 3) * 2];
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[t], acc[m][t], 0, 0, 0);
      }
    }
    float s = 0;
    for (int m = 0; m < 2; ++m) for (int t = 0; t < 2; ++t) s += acc[m][t][0] + acc[m][t][15];
    out[blockIdx.x * 256 + tid] = s;
  }
}

template <int KIND, bool READ>
void run(const char* name, float* out) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<KIND, READ>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int bpc = 1; bpc <= 3; ++bpc) {
    const size_t lds = bpc == 1 ? 100 * 1024 : (bpc == 2 ? 60 * 1024 : 40 * 1024);
    const int blocks = 256 * bpc * 4, iters = 400;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, READ>), dim3(blocks), dim3(256), lds, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, READ>), dim3(blocks), dim3(256), lds, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 9 * 16 * 2048.0;
    printf("%-28s blocks/CU %d : %7.1f TFLOP/s\n", name, bpc, flop / (ms * 1e-3) / 1e12);
  }
}
int main() {
  float* out; hipMalloc(&out, 256 * 3 * 4 * 256 * 4);
  run<0, false>("16x16x4 no LDS reads", out);
  run<0, true>("16x16x4 8 ds_read/16 mfma", out);
  run<1, false>("32x32x2 no LDS reads", out);
  run<1, true>("32x32x2 8 ds_read/8 mfma", out);
  return 0;
}
