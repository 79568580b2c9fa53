// Self-supervised (DINO / iBOT) head kernels for gfx950: token-row gather/scatter, weight-normalised prototype layer
// (dino_head.py:47-49), teacher softmax-centering and the student cross-entropy over K prototypes.  All of them are
// HBM-bound row kernels (K up to 65536 columns per row): one workgroup per row, 16-B accesses, two passes per row.
//
// Reference: DINOHead.forward (vtp/models/heads/dino_head.py:65-89), the teacher/student token buffers of
// VTP.get_teacher_forward_outputs / get_student_ssl_outputs (vtp/models/vtp.py:410-484).  The losses are NOT in the
// reference (DINOv2 conventions, parity unpinned).
#include "common.h"
#include "vtp_hip.h"

namespace vtp {

// dst[t, :] = idx[t] >= 0 ? src[idx[t], :] : 0        (bf16 rows of width D, D % 8 == 0)
__global__ __launch_bounds__(256) void gather_token_rows_kernel(const bf16* __restrict__ src, const int* __restrict__ idx,
                                                                bf16* __restrict__ dst, int T, int D) {
  const int d8 = D / 8;
  const long total = (long)T * d8;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const int t = (int)(i / d8), c = (int)(i % d8);
    const int r = idx[t];
    bf16x8 v;
    if (r >= 0) {
      v = *(const bf16x8*)(src + (long)r * D + 8 * c);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
    }
    *(bf16x8*)(dst + (long)t * D + 8 * c) = v;
  }
}

// d_src[idx[t], :] = d_dst[t, :] for idx[t] >= 0  (indices are unique: every token row feeds the head at most once)
__global__ __launch_bounds__(256) void scatter_token_rows_kernel(const bf16* __restrict__ d_dst, const int* __restrict__ idx,
                                                                 bf16* __restrict__ d_src, int T, int D) {
  const int d8 = D / 8;
  const long total = (long)T * d8;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const int t = (int)(i / d8), c = (int)(i % d8);
    const int r = idx[t];
    if (r >= 0) *(bf16x8*)(d_src + (long)r * D + 8 * c) = *(const bf16x8*)(d_dst + (long)t * D + 8 * c);
  }
}

// weight_norm(Linear(C -> K)): W_eff[k, :] = g[k] * v[k, :] / ||v[k, :]||.  One wave per prototype row (C <= 512).
__global__ __launch_bounds__(256) void weight_norm_prep_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                               bf16* __restrict__ weff, bf16* __restrict__ weffT,
                                                               float* __restrict__ inv_norm, int K, int C) {
  const int lane = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= K) return;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float x = v[(long)k * C + c];
    s += x * x;
  }
  s = wave_sum(s);
  const float inv = rsqrtf(s);
  const float sc = g[k] * inv;
  for (int c = lane; c < C; c += 64) {
    const bf16 w = f2bf(v[(long)k * C + c] * sc);
    weff[(long)k * C + c] = w;
    if (weffT) weffT[(long)c * K + k] = w;
  }
  if (lane == 0) inv_norm[k] = inv;
}

// The same for C <= 256 with the transposed copy written in full lines (round 5): a workgroup takes 64 prototype rows, keeps their
// bf16 values in an LDS tile and writes W_eff^T as [c][64 consecutive k] row pieces (128 B per wave-instruction) -- the kernel above
// scatters 2-byte elements 2 K bytes apart (152 us for the 65536 x 256 head of the benchmarked step against ~30 us of HBM time).
// Same arithmetic per element (wave-sum of squares in the same lane order, rsqrt, one rounding): bit-identical outputs.
__global__ __launch_bounds__(256) void weight_norm_prep_tiled_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                                     bf16* __restrict__ weff, bf16* __restrict__ weffT,
                                                                     float* __restrict__ inv_norm, int K, int C) {
  __shared__ bf16 tile[64][256 + 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k0 = blockIdx.x * 64;
  for (int r = wave; r < 64; r += 4) {  // wave = one prototype row at a time, 16 rows per wave
    const int k = k0 + r;
    if (k >= K) break;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float x = v[(long)k * C + c];
      s += x * x;
    }
    s = wave_sum(s);
    const float inv = rsqrtf(s);
    const float sc = g[k] * inv;
    for (int c = lane; c < C; c += 64) {
      const bf16 w = f2bf(v[(long)k * C + c] * sc);
      weff[(long)k * C + c] = w;
      tile[r][c] = w;
    }
    if (lane == 0) inv_norm[k] = inv;
  }
  __syncthreads();
  const int k = k0 + lane;
  if (k < K)
    for (int c = wave; c < C; c += 4) weffT[(long)c * K + k] = tile[lane][c];
}

// dv += (g/||v||) * (dW - <dW, vhat> vhat) ; dg += <dW, vhat>        (vhat = v/||v||)
__global__ __launch_bounds__(256) void weight_norm_bwd_kernel(const float* __restrict__ dW, const float* __restrict__ v,
                                                              const float* __restrict__ g, const float* __restrict__ inv_norm,
                                                              float* __restrict__ dv, float* __restrict__ dg, int K, int C) {
  const int lane = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= K) return;
  const float inv = inv_norm[k];
  float dot = 0.f;
  for (int c = lane; c < C; c += 64) dot += dW[(long)k * C + c] * v[(long)k * C + c] * inv;
  dot = wave_sum(dot);
  const float sc = g[k] * inv;
  for (int c = lane; c < C; c += 64) {
    const float vh = v[(long)k * C + c] * inv;
    dv[(long)k * C + c] += sc * (dW[(long)k * C + c] - dot * vh);
  }
  if (lane == 0) dg[k] += dot;
}

// block-wide (max, sum-exp) of a row of K bf16 logits scaled by `sc` after subtracting an optional f32 center
__device__ __forceinline__ void row_lse(const bf16* __restrict__ row, const float* __restrict__ center, float sc, int K,
                                        float* red, float& mx_out, float& se_out) {
  float mx = -INFINITY, se = 0.f;
  for (int k = threadIdx.x * 8; k < K; k += 256 * 8) {
    bf16x8 v = *(const bf16x8*)(row + k);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = (bf2f(v[e]) - (center ? center[k + e] : 0.f)) * sc;
      if (x > mx) {
        se = se * __expf(mx - x) + 1.f;
        mx = x;
      } else {
        se += __expf(x - mx);
      }
    }
  }
  // combine lanes: (mx, se) pairs
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(mx, o, 64), s2 = __shfl_xor(se, o, 64);
    const float m = fmaxf(mx, m2);
    se = (mx == -INFINITY ? 0.f : se * __expf(mx - m)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - m));
    mx = m;
  }
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    red[2 * w] = mx;
    red[2 * w + 1] = se;
  }
  __syncthreads();
  float m = fmaxf(fmaxf(red[0], red[2]), fmaxf(red[4], red[6]));
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += (red[2 * i] == -INFINITY) ? 0.f : red[2 * i + 1] * __expf(red[2 * i] - m);
  mx_out = m;
  se_out = s;
}

// teacher targets: probs[r, :] = softmax((logits[r, :] - center) * inv_temp)   (bf16 out)
__global__ __launch_bounds__(256) void softmax_center_kernel(const bf16* __restrict__ logits, const float* __restrict__ center,
                                                             float inv_temp, bf16* __restrict__ probs, int K,
                                                             const float* __restrict__ inv_temp_dev) {
  __shared__ float red[8];
  if (inv_temp_dev) inv_temp = *inv_temp_dev;  // device-resident temperature: a captured hipGraph follows a schedule
  const bf16* row = logits + (long)blockIdx.x * K;
  float mx, se;
  row_lse(row, center, inv_temp, K, red, mx, se);
  const float inv = 1.f / se;
  bf16* out = probs + (long)blockIdx.x * K;
  for (int k = threadIdx.x * 8; k < K; k += 256 * 8) {
    bf16x8 v = *(const bf16x8*)(row + k), o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(__expf((bf2f(v[e]) - (center ? center[k + e] : 0.f)) * inv_temp - mx) * inv);
    *(bf16x8*)(out + k) = o;
  }
}

// register-resident variant for K <= 65536: 1024 threads hold the whole centred, scaled row (<= 64 values per lane), so logits
// and the f32 centre are each read ONCE per row (the three-pass kernel above re-reads 128 KB + 256 KB per pass at K = 65536)
__global__ __launch_bounds__(1024) void softmax_center_reg_kernel(const bf16* __restrict__ logits, const float* __restrict__ center,
                                                                  float inv_temp, bf16* __restrict__ probs, int K,
                                                                  const float* __restrict__ inv_temp_dev) {
  __shared__ float red[32];
  if (inv_temp_dev) inv_temp = *inv_temp_dev;
  const bf16* row = logits + (long)blockIdx.x * K;
  float z[8][8];
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int k = (c * 1024 + threadIdx.x) * 8;
    if (k < K) {
      const bf16x8 v = *(const bf16x8*)(row + k);
      f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
      if (center) {
        c0 = *(const f32x4*)(center + k);
        c1 = *(const f32x4*)(center + k + 4);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        z[c][e] = (bf2f(v[e]) - (e < 4 ? c0[e] : c1[e - 4])) * inv_temp;
        mx = fmaxf(mx, z[c][e]);
      }
    }
  }
  mx = wave_max(mx);
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) red[w] = mx;
  __syncthreads();
  mx = red[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) mx = fmaxf(mx, red[i]);
  float se = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int k = (c * 1024 + threadIdx.x) * 8;
    if (k < K) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        z[c][e] = __expf(z[c][e] - mx);
        se += z[c][e];
      }
    }
  }
  se = wave_sum(se);
  if (lane == 0) red[16 + w] = se;
  __syncthreads();
  se = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) se += red[16 + i];
  const float inv = 1.f / se;
  bf16* out = probs + (long)blockIdx.x * K;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int k = (c * 1024 + threadIdx.x) * 8;
    if (k < K) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(z[c][e] * inv);
      *(bf16x8*)(out + k) = o;
    }
  }
}

// student cross-entropy against the sum of up to two teacher target rows:
//   loss_sum += w * sum_targets( - sum_k p_t[k] * log_softmax(s * inv_temp)[k] )
//   d_logits  = w * inv_temp * (n_targets * softmax(s * inv_temp) - sum_targets p_t)
__global__ __launch_bounds__(256) void dino_ce_kernel(const bf16* __restrict__ S, const bf16* __restrict__ P,
                                                      const int* __restrict__ t0, const int* __restrict__ t1,
                                                      const float* __restrict__ w, float inv_temp,
                                                      float* __restrict__ loss_sum, bf16* __restrict__ dS, int K) {
  __shared__ float red[8];
  const int r = blockIdx.x;
  const bf16* row = S + (long)r * K;
  bf16* drow = dS + (long)r * K;
  const float wr = w[r];
  const int i0 = t0[r], i1 = t1[r];
  if (wr == 0.f || i0 < 0) {  // padding row
    for (int k = threadIdx.x * 8; k < K; k += 256 * 8) {
      bf16x8 z;
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = (bf16)0.f;
      *(bf16x8*)(drow + k) = z;
    }
    return;
  }
  float mx, se;
  row_lse(row, nullptr, inv_temp, K, red, mx, se);
  const float lse = mx + __logf(se);
  const float nt = i1 >= 0 ? 2.f : 1.f;
  const bf16* p0 = P + (long)i0 * K;
  const bf16* p1 = i1 >= 0 ? P + (long)i1 * K : nullptr;
  float loss = 0.f;
  for (int k = threadIdx.x * 8; k < K; k += 256 * 8) {
    bf16x8 s = *(const bf16x8*)(row + k), a = *(const bf16x8*)(p0 + k), b, o;
    if (p1) b = *(const bf16x8*)(p1 + k);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float lsm = bf2f(s[e]) * inv_temp - lse;
      const float q = bf2f(a[e]) + (p1 ? bf2f(b[e]) : 0.f);
      loss -= q * lsm;
      o[e] = f2bf(wr * inv_temp * (nt * __expf(lsm) - q));
    }
    *(bf16x8*)(drow + k) = o;
  }
  __shared__ float red2[4];
  loss = block_sum<4>(loss, red2);
  if (threadIdx.x == 0) unsafeAtomicAdd(loss_sum, wr * loss);
}

// center <- momentum * center + (1 - momentum) * (col_sum * inv_count)
// count_ptr != NULL: the (all-reduced) row count lives on the device (ranks contribute different numbers of masked tokens)
__global__ __launch_bounds__(256) void center_ema_kernel(float* __restrict__ center, const float* __restrict__ col_sum,
                                                         float inv_count, const float* __restrict__ count_ptr, float momentum,
                                                         int K) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (count_ptr) inv_count = 1.f / fmaxf(count_ptr[0], 1.f);
  if (k < K) center[k] = momentum * center[k] + (1.f - momentum) * col_sum[k] * inv_count;
}

}  // namespace vtp
using namespace vtp;

static inline int ssl_grid(long items) {
  long b = (items + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

extern "C" int vtp_gather_token_rows(const void* src, const int* idx, void* dst, int T, int D, void* stream) {
  VTP_REQUIRE(src && idx && dst && T > 0 && D > 0 && D % 8 == 0, "vtp_gather_token_rows: bad argument (D %% 8 == 0)");
  hipLaunchKernelGGL(gather_token_rows_kernel, dim3(ssl_grid((long)T * D / 8)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)src, idx, (bf16*)dst, T, D);
  return check_launch("gather_token_rows");
}

extern "C" int vtp_scatter_token_rows(const void* d_dst, const int* idx, void* d_src, int T, int D, void* stream) {
  VTP_REQUIRE(d_dst && idx && d_src && T > 0 && D > 0 && D % 8 == 0, "vtp_scatter_token_rows: bad argument (D %% 8 == 0)");
  hipLaunchKernelGGL(scatter_token_rows_kernel, dim3(ssl_grid((long)T * D / 8)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)d_dst, idx, (bf16*)d_src, T, D);
  return check_launch("scatter_token_rows");
}

extern "C" int vtp_weight_norm_prep(const float* v, const float* g, void* weff, void* weffT, float* inv_norm, int K, int C,
                                    void* stream) {
  VTP_REQUIRE(v && g && weff && inv_norm && K > 0 && C > 0, "vtp_weight_norm_prep: bad argument");
  if (weffT && C <= 256)
    hipLaunchKernelGGL(weight_norm_prep_tiled_kernel, dim3(cdiv(K, 64)), dim3(256), 0, (hipStream_t)stream, v, g, (bf16*)weff,
                       (bf16*)weffT, inv_norm, K, C);
  else
    hipLaunchKernelGGL(weight_norm_prep_kernel, dim3(cdiv(K, 4)), dim3(256), 0, (hipStream_t)stream, v, g, (bf16*)weff,
                       (bf16*)weffT, inv_norm, K, C);
  return check_launch("weight_norm_prep");
}

extern "C" int vtp_weight_norm_bwd(const float* dW, const float* v, const float* g, const float* inv_norm, float* dv, float* dg,
                                   int K, int C, void* stream) {
  VTP_REQUIRE(dW && v && g && inv_norm && dv && dg && K > 0 && C > 0, "vtp_weight_norm_bwd: bad argument");
  hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3(cdiv(K, 4)), dim3(256), 0, (hipStream_t)stream, dW, v, g, inv_norm, dv, dg, K, C);
  return check_launch("weight_norm_bwd");
}

static int launch_softmax_center(const void* logits, const float* center, float inv_temp, const float* inv_temp_dev, void* probs, int T,
                                 int K, void* stream) {
  if (K <= 65536 && K >= 8192)
    hipLaunchKernelGGL(softmax_center_reg_kernel, dim3(T), dim3(1024), 0, (hipStream_t)stream, (const bf16*)logits, center,
                       inv_temp, (bf16*)probs, K, inv_temp_dev);
  else
    hipLaunchKernelGGL(softmax_center_kernel, dim3(T), dim3(256), 0, (hipStream_t)stream, (const bf16*)logits, center, inv_temp,
                       (bf16*)probs, K, inv_temp_dev);
  return check_launch("softmax_center");
}

extern "C" int vtp_softmax_center(const void* logits, const float* center, float inv_temp, void* probs, int T, int K,
                                  void* stream) {
  VTP_REQUIRE(logits && probs && T > 0 && K > 0 && K % 8 == 0, "vtp_softmax_center: bad argument (K %% 8 == 0)");
  return launch_softmax_center(logits, center, inv_temp, nullptr, probs, T, K, stream);
}

extern "C" int vtp_softmax_center_dev(const void* logits, const float* center, const float* inv_temp, void* probs, int T, int K,
                                      void* stream) {
  VTP_REQUIRE(logits && probs && inv_temp && T > 0 && K > 0 && K % 8 == 0, "vtp_softmax_center_dev: bad argument (K %% 8 == 0)");
  return launch_softmax_center(logits, center, 0.f, inv_temp, probs, T, K, stream);
}

extern "C" int vtp_dino_ce(const void* student_logits, const void* teacher_probs, const int* t_idx0, const int* t_idx1,
                           const float* row_weight, float inv_temp, float* loss_sum, void* d_student_logits, int T, int K,
                           void* stream) {
  VTP_REQUIRE(student_logits && teacher_probs && t_idx0 && t_idx1 && row_weight && loss_sum && d_student_logits,
              "vtp_dino_ce: null pointer");
  VTP_REQUIRE(T > 0 && K > 0 && K % 8 == 0, "vtp_dino_ce: bad shape (K %% 8 == 0)");
  hipLaunchKernelGGL(dino_ce_kernel, dim3(T), dim3(256), 0, (hipStream_t)stream, (const bf16*)student_logits,
                     (const bf16*)teacher_probs, t_idx0, t_idx1, row_weight, inv_temp, loss_sum, (bf16*)d_student_logits, K);
  return check_launch("dino_ce");
}

extern "C" int vtp_center_ema(float* center, const float* col_sum, float inv_count, const float* count_ptr, float momentum,
                              int K, void* stream) {
  VTP_REQUIRE(center && col_sum && K > 0, "vtp_center_ema: bad argument");
  hipLaunchKernelGGL(center_ema_kernel, dim3(cdiv(K, 256)), dim3(256), 0, (hipStream_t)stream, center, col_sum, inv_count,
                     count_ptr, momentum, K);
  return check_launch("center_ema");
}
