// gsd_dataset.hip -- the reference's dataset path as HBM-bound HIP kernels, so the train step is fed from a
// device-resident dataset instead of host __getitem__ loops + pinned H2D copies.
//
//   gsd_ingest_images    finger split (strided channel view) + difference image + F.interpolate(mode='area')
//                        /root/reference/gelslim_depth/datasets/general_dataset.py:61-83, image_utils.py:6-15
//   gsd_channel_stats    per-channel min / max / mean / unbiased std over the whole dataset
//                        general_dataset.py:199-220
//   gsd_gather_affine    batch assembly: rows picked by a (shuffled) index vector, normalised on the way
//                        general_dataset.py:222-236 (__getitem__ + normalize_sample), normalization_utils.py:4-35,67-99
//
//   gsd_gaussian_blur    torchvision.transforms.functional.gaussian_blur on the resized depth planes (reflect padding,
//                        depthwise 2-D correlation with the outer-product kernel): image_utils.py:17-19, called from
//                        general_dataset.py:74-76,84-86 when depth_image_blur_kernel > 1
//
// All are one pass over their input at HBM rate: no LDS, coalesced along the pixel index.
#include "gsd_common.h"

namespace {

constexpr int SG = 64;  // row groups of the stats reduction

template <typename T>
__global__ __launch_bounds__(256) void ingest_kernel(const T* __restrict__ in, const T* __restrict__ base, int C, int H, int W,
                                                     long long in_ns, long long in_cs, long long base_ns, long long base_cs,
                                                     float* __restrict__ out, int OH, int OW, float pre_add, float pre_mul) {
  const int c = blockIdx.y, n = blockIdx.z;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= OH * OW) return;
  const int oh = e / OW, ow = e - oh * OW;
  // adaptive-average-pool window == F.interpolate(mode='area'): [floor(o*I/O), ceil((o+1)*I/O))
  const int h0 = (int)(((long long)oh * H) / OH), h1 = (int)(((long long)(oh + 1) * H + OH - 1) / OH);
  const int w0 = (int)(((long long)ow * W) / OW), w1 = (int)(((long long)(ow + 1) * W + OW - 1) / OW);
  const T* ip = in + (long long)n * in_ns + (long long)c * in_cs;
  const T* bp = base != nullptr ? base + (long long)n * base_ns + (long long)c * base_cs : nullptr;
  float s = 0.f;
  for (int h = h0; h < h1; ++h)
    for (int w = w0; w < w1; ++w) {
      float v = (float)ip[(size_t)h * W + w];
      if (bp != nullptr) v = (v - (float)bp[(size_t)h * W + w] + pre_add) * pre_mul;
      s += v;
    }
  out[((size_t)n * C + c) * OH * OW + e] = s / (float)((h1 - h0) * (w1 - w0));
}

// stage 1: grid (C, SG); block reduces its share of the N*HW elements of channel c
__global__ __launch_bounds__(256) void stats_stage1(const float* __restrict__ x, long long N, int C, long long HW,
                                                    double* __restrict__ ws) {
  const int c = blockIdx.x, g = blockIdx.y;
  const long long total = N * HW;
  const long long per = (total + SG - 1) / SG;
  const long long b = (long long)g * per, e = b + per < total ? b + per : total;
  double s = 0.0, s2 = 0.0;
  float mn = __builtin_inff(), mx = -__builtin_inff();
  for (long long i = b + threadIdx.x; i < e; i += 256) {
    const long long n = i / HW, p = i - n * HW;
    const float v = x[((size_t)n * C + c) * HW + p];
    s += (double)v;
    s2 += (double)v * (double)v;
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
  }
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    s2 += __shfl_xor(s2, o, 64);
    mn = fminf(mn, __shfl_xor(mn, o, 64));
    mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  }
  __shared__ double red[4][4];
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6][0] = s;
    red[threadIdx.x >> 6][1] = s2;
    red[threadIdx.x >> 6][2] = (double)mn;
    red[threadIdx.x >> 6][3] = (double)mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* o = ws + ((size_t)c * SG + g) * 4;
    o[0] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    o[1] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    o[2] = fmin(fmin(red[0][2], red[1][2]), fmin(red[2][2], red[3][2]));
    o[3] = fmax(fmax(red[0][3], red[1][3]), fmax(red[2][3], red[3][3]));
  }
}
// stage 2: one thread per channel; out[c] = {min, max, mean, std (unbiased, torch.std default)}
__global__ void stats_stage2(const double* __restrict__ ws, int C, double count, double* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, s2 = 0.0, mn = __builtin_inf(), mx = -__builtin_inf();
  for (int g = 0; g < SG; ++g) {
    const double* p = ws + ((size_t)c * SG + g) * 4;
    s += p[0];
    s2 += p[1];
    mn = fmin(mn, p[2]);
    mx = fmax(mx, p[3]);
  }
  const double mean = s / count;
  double var = count > 1.0 ? (s2 - s * mean) / (count - 1.0) : __builtin_nan("");
  if (var < 0.0) var = 0.0;
  out[4 * c + 0] = mn;
  out[4 * c + 1] = mx;
  out[4 * c + 2] = mean;
  out[4 * c + 3] = sqrt(var);
}

// grid (ceil(HW/1024), C, B): float4 path when HW % 4 == 0 (rows then stay 16-byte aligned)
template <bool VEC4>
__global__ __launch_bounds__(256) void gather_affine_kernel(const float* __restrict__ src, const long long* __restrict__ idx,
                                                            long long M, int C, long long HW, const float* __restrict__ A,
                                                            const float* __restrict__ Bc, int nab, float* __restrict__ out) {
  const int c = blockIdx.y, b = blockIdx.z;
  const long long row = idx[b];
  const int cc = c < nab ? c : nab - 1;
  const float a = A[cc], bb = Bc[cc];
  const bool ok = row >= 0 && row < M;   // an out-of-range index is the caller's bug: make it loud (NaN), never read OOB
  const float* s = src + ((size_t)(ok ? row : 0) * C + c) * HW;
  float* o = out + ((size_t)b * C + c) * HW;
  const float bad = __builtin_nanf("");
  if constexpr (VEC4) {
    const long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= HW) return;
    f32x4 v = *reinterpret_cast<const f32x4*>(s + e);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = ok ? fmaf(v[k], a, bb) : bad;
    *reinterpret_cast<f32x4*>(o + e) = v;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long e = ((long long)blockIdx.x * 4 + k) * 256 + threadIdx.x;
      if (e < HW) o[e] = ok ? fmaf(s[e], a, bb) : bad;
    }
  }
}

// out[p][h][w] = sum_{i,j} k2[i][j] * in[p][reflect(h+i-r)][reflect(w+j-r)],  r = K/2, reflect without repeating the edge
// (torch's F.pad(mode="reflect")); taps accumulated in row-major (i, j) order like a direct depthwise conv2d
__global__ __launch_bounds__(256) void gaussian_blur_kernel(const float* __restrict__ in, int H, int W, const float* __restrict__ k2,
                                                            int K, float* __restrict__ out) {
  const long long p = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= H * W) return;
  const int h = e / W, w = e - h * W, r = K / 2;
  const float* ip = in + (size_t)p * H * W;
  float s = 0.f;
  for (int i = 0; i < K; ++i) {
    int hh = h + i - r;
    hh = hh < 0 ? -hh : (hh >= H ? 2 * (H - 1) - hh : hh);
    for (int j = 0; j < K; ++j) {
      int ww = w + j - r;
      ww = ww < 0 ? -ww : (ww >= W ? 2 * (W - 1) - ww : ww);
      s = fmaf(k2[i * K + j], ip[(size_t)hh * W + ww], s);
    }
  }
  out[(size_t)p * H * W + e] = s;
}

}  // namespace

extern "C" int gsd_gaussian_blur(const float* in, int64_t planes, int H, int W, const float* kernel2d, int K, float* out,
                                 void* stream) {
  GSD_REQUIRE(in && out && kernel2d && in != out && planes > 0 && H > 0 && W > 0, GSD_ERR_BAD_ARG, "gsd_gaussian_blur: bad argument");
  GSD_REQUIRE(K >= 1 && (K & 1) == 1, GSD_ERR_BAD_ARG, "gsd_gaussian_blur: the kernel size must be odd and positive");
  GSD_REQUIRE(K / 2 < H && K / 2 < W, GSD_ERR_UNSUPPORTED,
              "gsd_gaussian_blur: reflect padding needs kernel_size/2 < H and W (as torch's F.pad does)");
  for (int64_t p0 = 0; p0 < planes; p0 += 65535) {
    const int np = (int)(planes - p0 < 65535 ? planes - p0 : 65535);
    hipLaunchKernelGGL(gaussian_blur_kernel, dim3(ceil_div(H * W, 256), np), dim3(256), 0, (hipStream_t)stream,
                       in + (size_t)p0 * H * W, H, W, kernel2d, K, out + (size_t)p0 * H * W);
    GSD_LAUNCH_CHECK("gsd_gaussian_blur");
  }
  return GSD_OK;
}

extern "C" int gsd_ingest_images(const void* in, const void* base, int dtype, int N, int C, int H, int W, int64_t in_n_stride,
                                 int64_t in_c_stride, int64_t base_n_stride, int64_t base_c_stride, float* out, int OH,
                                 int OW, float pre_add, float pre_mul, void* stream) {
  GSD_REQUIRE(in && out && N > 0 && C > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, GSD_ERR_BAD_ARG,
              "gsd_ingest_images: bad argument");
  GSD_REQUIRE(dtype == 0 || dtype == 1, GSD_ERR_UNSUPPORTED, "gsd_ingest_images: dtype must be 0 (f32) or 1 (u8)");
  GSD_REQUIRE(in_c_stride >= (int64_t)H * W && (base == nullptr || base_c_stride >= (int64_t)H * W), GSD_ERR_BAD_ARG,
              "gsd_ingest_images: channel stride smaller than a plane");
  GSD_REQUIRE(N <= 65535 && C <= 65535, GSD_ERR_UNSUPPORTED, "gsd_ingest_images: N, C must be <= 65535 per call");
  const dim3 grid(ceil_div(OH * OW, 256), C, N);
  if (dtype == 0)
    hipLaunchKernelGGL(ingest_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)in, (const float*)base, C,
                       H, W, (long long)in_n_stride, (long long)in_c_stride, (long long)base_n_stride,
                       (long long)base_c_stride, out, OH, OW, pre_add, pre_mul);
  else
    hipLaunchKernelGGL(ingest_kernel<unsigned char>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned char*)in,
                       (const unsigned char*)base, C, H, W, (long long)in_n_stride, (long long)in_c_stride,
                       (long long)base_n_stride, (long long)base_c_stride, out, OH, OW, pre_add, pre_mul);
  GSD_LAUNCH_CHECK("gsd_ingest_images");
  return GSD_OK;
}

extern "C" int64_t gsd_channel_stats_workspace(int C) { return C > 0 ? (int64_t)C * SG * 4 : 0; }

extern "C" int gsd_channel_stats(const float* x, int64_t N, int C, int64_t HW, double* out, double* workspace, void* stream) {
  GSD_REQUIRE(x && out && workspace && N > 0 && C > 0 && HW > 0, GSD_ERR_BAD_ARG, "gsd_channel_stats: bad argument");
  GSD_REQUIRE(C <= 65535, GSD_ERR_UNSUPPORTED, "gsd_channel_stats: C must be <= 65535");
  hipLaunchKernelGGL(stats_stage1, dim3(C, SG), dim3(256), 0, (hipStream_t)stream, x, (long long)N, C, (long long)HW, workspace);
  GSD_LAUNCH_CHECK("gsd_channel_stats stage1");
  hipLaunchKernelGGL(stats_stage2, dim3(ceil_div(C, 64)), dim3(64), 0, (hipStream_t)stream, workspace, C,
                     (double)N * (double)HW, out);
  GSD_LAUNCH_CHECK("gsd_channel_stats stage2");
  return GSD_OK;
}

extern "C" int gsd_gather_affine(const float* src, const int64_t* idx, int64_t M, int B, int C, int64_t HW, const float* A,
                                 const float* Bc, int nab, float* out, void* stream) {
  GSD_REQUIRE(src && idx && A && Bc && out && M > 0 && B > 0 && C > 0 && HW > 0 && nab > 0, GSD_ERR_BAD_ARG,
              "gsd_gather_affine: bad argument");
  GSD_REQUIRE(B <= 65535 && C <= 65535, GSD_ERR_UNSUPPORTED, "gsd_gather_affine: B, C must be <= 65535");
  const dim3 grid((unsigned)ceil_div64(HW, 1024), C, B);
  const bool vec = (HW % 4 == 0) && (((uintptr_t)src | (uintptr_t)out) & 15) == 0;
  if (vec)
    hipLaunchKernelGGL(gather_affine_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, src, (const long long*)idx,
                       (long long)M, C, (long long)HW, A, Bc, nab, out);
  else
    hipLaunchKernelGGL(gather_affine_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, src, (const long long*)idx,
                       (long long)M, C, (long long)HW, A, Bc, nab, out);
  GSD_LAUNCH_CHECK("gsd_gather_affine");
  return GSD_OK;
}
