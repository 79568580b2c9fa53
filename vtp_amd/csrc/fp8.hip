// fp8 (e4m3, OCP -- the gfx950 format) forward path of BASELINE config 5: per-tensor scaled quantisation of GEMM operands.
//   q = e4m3(clamp(x * scale, -448, 448)),  scale = 448 / amax(x)  (the caller keeps 1 / scale for the GEMM's alpha)
// The GEMM itself is vtp_gemm_nt_fp8 (gemm.hip / gemm8p.hip, v_mfma_scale_f32_32x32x64_f8f6f4).  HBM-bound row kernels.
#include "common.h"
#include "vtp_hip.h"

namespace vtp {

constexpr float E4M3_MAX = 448.f;

__device__ __forceinline__ uint32_t pack4_e4m3(float a, float b, float c, float d) {
  a = fminf(fmaxf(a, -E4M3_MAX), E4M3_MAX);
  b = fminf(fmaxf(b, -E4M3_MAX), E4M3_MAX);
  c = fminf(fmaxf(c, -E4M3_MAX), E4M3_MAX);
  d = fminf(fmaxf(d, -E4M3_MAX), E4M3_MAX);
  uint32_t w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0u, false);
  return __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
}

// 8 elements per thread: 16 B (bf16) or 32 B (f32) in, 8 B out
template <bool F32IN>
__global__ __launch_bounds__(256) void quantize_e4m3_kernel(const void* __restrict__ src, uint8_t* __restrict__ dst, long n8,
                                                           const float* __restrict__ scale_dev, float scale) {
  if (scale_dev) scale = *scale_dev;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256L) {
    float v[8];
    if constexpr (F32IN) {
      const f32x4 a = *(const f32x4*)((const float*)src + 8 * i), b = *(const f32x4*)((const float*)src + 8 * i + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = a[e] * scale;
        v[4 + e] = b[e] * scale;
      }
    } else {
      const bf16x8 a = *(const bf16x8*)((const bf16*)src + 8 * i);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = bf2f(a[e]) * scale;
    }
    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
    *(u32x2*)(dst + 8 * i) = u32x2{pack4_e4m3(v[0], v[1], v[2], v[3]), pack4_e4m3(v[4], v[5], v[6], v[7])};
  }
}

// amax[0] = max(amax[0], max |x|)   (non-negative floats order like their bit patterns: integer atomicMax)
template <bool F32IN>
__global__ __launch_bounds__(256) void amax_kernel(const void* __restrict__ src, long n8, float* __restrict__ amax) {
  __shared__ float red[4];
  float m = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256L) {
    if constexpr (F32IN) {
      const f32x4 a = *(const f32x4*)((const float*)src + 8 * i), b = *(const f32x4*)((const float*)src + 8 * i + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) m = fmaxf(m, fmaxf(fabsf(a[e]), fabsf(b[e])));
    } else {
      const bf16x8 a = *(const bf16x8*)((const bf16*)src + 8 * i);
#pragma unroll
      for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(bf2f(a[e])));
    }
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    atomicMax((unsigned int*)amax, __float_as_uint(m));
  }
}

// e4m3 -> f32 (test / debug aid: what the MFMA sees)
__global__ __launch_bounds__(256) void dequantize_e4m3_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, long n4, float inv_scale) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256L) {
    const uint32_t w = *(const uint32_t*)(src + 4 * i);
    f32x4 o;
    o[0] = __builtin_amdgcn_cvt_f32_fp8(w, 0);
    o[1] = __builtin_amdgcn_cvt_f32_fp8(w, 1);
    o[2] = __builtin_amdgcn_cvt_f32_fp8(w, 2);
    o[3] = __builtin_amdgcn_cvt_f32_fp8(w, 3);
    *(f32x4*)(dst + 4 * i) = o * inv_scale;
  }
}

}  // namespace vtp
using namespace vtp;

static inline int fp8_grid(long items) {
  long b = (items + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

extern "C" int vtp_quantize_e4m3(const void* src, int src_is_f32, void* dst, long n, const float* scale_dev, float scale, void* stream) {
  VTP_REQUIRE(src && dst && n > 0 && n % 8 == 0, "vtp_quantize_e4m3: bad argument (n %% 8 == 0)");
  if (src_is_f32)
    hipLaunchKernelGGL(quantize_e4m3_kernel<true>, dim3(fp8_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, src, (uint8_t*)dst, n / 8,
                       scale_dev, scale);
  else
    hipLaunchKernelGGL(quantize_e4m3_kernel<false>, dim3(fp8_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, src, (uint8_t*)dst, n / 8,
                       scale_dev, scale);
  return check_launch("quantize_e4m3");
}

extern "C" int vtp_amax(const void* src, int src_is_f32, long n, float* amax, void* stream) {
  VTP_REQUIRE(src && amax && n > 0 && n % 8 == 0, "vtp_amax: bad argument (n %% 8 == 0)");
  if (src_is_f32)
    hipLaunchKernelGGL(amax_kernel<true>, dim3(fp8_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, src, n / 8, amax);
  else
    hipLaunchKernelGGL(amax_kernel<false>, dim3(fp8_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, src, n / 8, amax);
  return check_launch("amax");
}

extern "C" int vtp_dequantize_e4m3(const void* src, float* dst, long n, float inv_scale, void* stream) {
  VTP_REQUIRE(src && dst && n > 0 && n % 4 == 0, "vtp_dequantize_e4m3: bad argument (n %% 4 == 0)");
  hipLaunchKernelGGL(dequantize_e4m3_kernel, dim3(fp8_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src, dst, n / 4,
                     inv_scale);
  return check_launch("dequantize_e4m3");
}
