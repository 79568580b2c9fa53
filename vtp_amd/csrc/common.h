// Common device/host helpers for the VTP gfx950 (CDNA4) kernels.  MI355X only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;   // MFMA bf16 operand type
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define VTP_OK 0
#define VTP_ERR_ARG -1

namespace vtp {

void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define VTP_REQUIRE(cond, ...)                 \
  do {                                         \
    if (!(cond)) {                             \
      vtp::set_error(__VA_ARGS__);             \
      return VTP_ERR_ARG;                      \
    }                                          \
  } while (0)

__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for blocks of NW waves; `red` is NW floats of LDS.  All threads get the result.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  if constexpr (NW == 1) return v;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) t += red[i];
  return t;
}

// 2^x as ONE v_exp_f32 (exp2f() adds a denormal-range fix-up: 6 VALU instead of 1); results below 2^-126 flush to 0, which is
// what a softmax numerator wants
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// sigmoid / SiLU with ONE v_rcp_f32 instead of an IEEE division (v_div_scale x2, v_rcp, 3 fma, v_div_fmas, v_div_fixup: 9
// instructions per element -- a third of the SwiGLU epilogue's VALU work).  v_rcp_f32 is accurate to 1 ulp; every caller rounds
// the result to bf16 next, so the value changes only within ~1.5 fp32 ulp of a bf16 rounding boundary (about 2e-5 of the elements,
// by one bf16 ulp -- inside the per-kernel parity bar, tests/test_kernels_gpu.py).
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

// SwiGLU backward of 8 elements (ffn.py:78-81 under bf16 autocast): x1 | x2 pre-activations, dh the gradient of silu(x1) x2
//   -> o1 = d/dx1, o2 = d/dx2 with the roundings of eager bf16 autograd (gs = bf16(dh x2), silu(x1) rounded to bf16 before the
//   product with dh).  Written on fp32 PAIRS: v_pk_mul / v_pk_add / v_pk_fma carry two elements per instruction -- the fused
//   w3-dgrad epilogue is bound by exactly this VALU work (2 781 instructions per wave and 256x256 tile with the scalar form).
//   The standalone kernels and the GEMM epilogue share this function, so they stay bit-identical (tests/test_kernels_gpu.py).
__device__ __forceinline__ void swiglu_bwd8(const bf16x8& x1, const bf16x8& x2, const bf16x8& dh, bf16x8& o1, bf16x8& o2) {
  const f32x2 one = {1.f, 1.f}, nl2e = {-1.4426950408889634f, -1.4426950408889634f};
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const f32x2 a = {bf2f(x1[e]), bf2f(x1[e + 1])}, b = {bf2f(x2[e]), bf2f(x2[e + 1])}, gd = {bf2f(dh[e]), bf2f(dh[e + 1])};
    const f32x2 t = a * nl2e;
    const f32x2 den = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + one;
    const f32x2 sg = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};  // sigmoid(x1)
    const bf16x2 gsb = __builtin_convertvector(gd * b, bf16x2);                          // grad wrt silu(x1), bf16 like autograd
    const f32x2 gs = {bf2f(gsb[0]), bf2f(gsb[1])};
    const f32x2 dsilu = sg * __builtin_elementwise_fma(a, one - sg, one);                // silu'(x1) = sg (1 + x1 (1 - sg))
    const bf16x2 r1 = __builtin_convertvector(gs * dsilu, bf16x2);
    const bf16x2 sb = __builtin_convertvector(a * sg, bf16x2);                           // silu(x1) rounded to bf16
    const bf16x2 r2 = __builtin_convertvector(gd * f32x2{bf2f(sb[0]), bf2f(sb[1])}, bf16x2);
    o1[e] = r1[0];
    o1[e + 1] = r1[1];
    o2[e] = r2[0];
    o2[e + 1] = r2[1];
  }
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// QuickGELU (layers/activation.py:5-12): x * sigmoid(1.702 x), and its derivative s (1 + 1.702 x (1 - s))
__device__ __forceinline__ float quick_gelu_f(float x) { return x * sigmoid_f(1.702f * x); }
__device__ __forceinline__ float quick_gelu_grad(float x) {
  const float s = sigmoid_f(1.702f * x);
  return s * (1.f + 1.702f * x * (1.f - s));
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }


}  // namespace vtp
