// Residual-wiring extras of SelfAttentionBlock (vtp/models/layers/block.py:20-118,207-289; misc.py:7-26) that the default
// configurations switch off: stochastic depth ("sample drop": the residual branch runs on a random subset of the images and
// is added back scaled by batch / kept) and LayerScale (per-channel gamma on the branch output) -- as HBM-bound row kernels.
//
// Token buffers are row-concatenated segments of B images x N tokens (engine.py); an image is N consecutive rows.
#include "common.h"
#include "vtp_hip.h"

namespace vtp {

// dst[i*N + t, :] = src[idx[i]*N + t, :]   (f32 copy, optional)   and / or   dst_b = bf16(scale * src rows) (optional)
__global__ __launch_bounds__(256) void gather_image_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx,
                                                                float* __restrict__ dst, bf16* __restrict__ dst_b, int n_img,
                                                                long N, int D, float scale) {
  const int d4 = D / 4;
  const long total = (long)n_img * N * d4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const int c = (int)(i % d4);
    const long row = i / d4;
    const int img = (int)(row / N);
    const long t = row - (long)img * N;
    const f32x4 v = *(const f32x4*)(src + ((long)idx[img] * N + t) * D + 4 * c);
    if (dst) *(f32x4*)(dst + row * D + 4 * c) = v;
    if (dst_b) *(bf16x4*)(dst_b + row * D + 4 * c) = __builtin_convertvector(v * scale, bf16x4);
  }
}

// dst[idx[i]*N + t, :] = (accumulate ? dst : 0) + alpha * src[i*N + t, :]     (idx unique: torch.index_add over distinct images)
__global__ __launch_bounds__(256) void scatter_image_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx,
                                                                 float* __restrict__ dst, int n_img, long N, int D, float alpha,
                                                                 int accumulate) {
  const int d4 = D / 4;
  const long total = (long)n_img * N * d4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const int c = (int)(i % d4);
    const long row = i / d4;
    const int img = (int)(row / N);
    const long t = row - (long)img * N;
    f32x4 v = *(const f32x4*)(src + row * D + 4 * c) * alpha;
    float* p = dst + ((long)idx[img] * N + t) * D + 4 * c;
    if (accumulate) v += *(const f32x4*)p;
    *(f32x4*)p = v;
  }
}

// LayerScale backward through y = gamma * (x W^T + b) given the UNSCALED weight-gradient G = dy^T x and cs = colsum(dy):
//   dW[n,:] += gamma[n] G[n,:] ;  db[n] += gamma[n] cs[n] ;  dgamma[n] += sum_k W[n,k] G[n,k] + b[n] cs[n]
// (sum_m dy[m,n] f[m,n] with f = x W^T + b, rewritten so that f is never stored).  One wave per output row n.
__global__ __launch_bounds__(256) void layerscale_wgrad_kernel(const float* __restrict__ G, const float* __restrict__ W,
                                                               const float* __restrict__ bias, const float* __restrict__ cs,
                                                               const float* __restrict__ gamma, float* __restrict__ dW,
                                                               float* __restrict__ db, float* __restrict__ dgamma, int N, int K) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const float g = gamma[n];
  float dot = 0.f;
  for (int k = lane * 4; k < K; k += 256) {
    const f32x4 gv = *(const f32x4*)(G + (size_t)n * K + k), wv = *(const f32x4*)(W + (size_t)n * K + k);
    f32x4 o = *(const f32x4*)(dW + (size_t)n * K + k);
    o += gv * g;
    *(f32x4*)(dW + (size_t)n * K + k) = o;
    dot += gv[0] * wv[0] + gv[1] * wv[1] + gv[2] * wv[2] + gv[3] * wv[3];
  }
  dot = wave_sum(dot);
  if (lane == 0) {
    const float c = cs[n];
    dgamma[n] += dot + (bias ? bias[n] * c : 0.f);
    if (db) db[n] += g * c;
  }
}

// dstT[k, n] = bf16(W[n, k] * gamma[n])   (the dgrad operand of a LayerScale'd linear: dx = (dy * gamma) W)
__global__ __launch_bounds__(256) void scaled_transpose_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                                               bf16* __restrict__ dstT, int N, int K) {
  __shared__ float tile[64][65];
  const int n0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int n = n0 + r, k = k0 + tx;
    tile[r][tx] = (n < N && k < K) ? W[(size_t)n * K + k] * gamma[n] : 0.f;
  }
  __syncthreads();
  for (int c = ty; c < 64; c += 4) {
    const int k = k0 + c, n = n0 + tx;
    if (k < K && n < N) dstT[(size_t)k * N + n] = f2bf(tile[tx][c]);
  }
}

// QK normalisation (attention.py:67-68,119-120: q = RMSNorm(head_dim)(q), k likewise, before RoPE; weights [64] shared by the
// heads).  qkv bf16 [M, 3D] as the projection packs it; a (row, head) of q or k is 64 elements = eight 16-B chunks = eight
// consecutive lanes.  out = bf16(bf16(x * rsqrt(mean x^2 + eps)) * w) for the q / k parts (the reference's `.type_as(x) * weight`
// followed by the bf16 cast of apply_rope / SDPA), v is copied; inv [M, 2 heads] keeps rsqrt(..) for the backward.
__global__ __launch_bounds__(256) void qk_norm_fwd_kernel(const bf16* __restrict__ qkv, const float* __restrict__ wq,
                                                          const float* __restrict__ wk, bf16* __restrict__ out, float* __restrict__ inv,
                                                          long M, int D, float eps) {
  const int cpr = 3 * D / 8, cpp = D / 8;  // 16-B chunks per row / per part
  const long total = M * cpr;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < (total + 255) / 256 * 256; i += (long)gridDim.x * 256L) {
    const bool live = i < total;
    const long row = live ? i / cpr : 0;
    const int c = live ? (int)(i - row * cpr) : 0, part = c / cpp, d0 = (c % 8) * 8;
    bf16x8 v = live ? *(const bf16x8*)(qkv + row * 3 * D + (long)c * 8) : bf16x8{};
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ss += bf2f(v[e]) * bf2f(v[e]);
    ss += __shfl_xor(ss, 1, 64);
    ss += __shfl_xor(ss, 2, 64);
    ss += __shfl_xor(ss, 4, 64);
    if (!live) continue;
    if (part < 2) {
      const float r = rsqrtf(ss * (1.f / 64.f) + eps);
      const float* w = part == 0 ? wq : wk;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = f2bf(bf2f(f2bf(bf2f(v[e]) * r)) * w[d0 + e]);
      if ((c & 7) == 0) inv[row * (2 * D / 64) + (c / 8)] = r;  // chunk / 8 = part * heads + head
    }
    *(bf16x8*)(out + row * 3 * D + (long)c * 8) = v;
  }
}

// in place on the q / k parts of dqkv (gradient w.r.t. the normalised, un-rotated q, k): with xh = x * inv, g = dy (.) w,
//   dx = inv * (g - xh * mean(g (.) xh)),   dw[d] += sum over rows and heads of dy[d] * bf16(xh[d])
__global__ __launch_bounds__(256) void qk_norm_bwd_kernel(bf16* __restrict__ dqkv, const bf16* __restrict__ qkv, const float* __restrict__ inv,
                                                          const float* __restrict__ wq, const float* __restrict__ wk, float* __restrict__ dwq,
                                                          float* __restrict__ dwk, long M, int D) {
  __shared__ float red[2][64];
  const int cpr = 2 * D / 8, cpp = D / 8;  // only the q / k chunks
  const long total = M * cpr;
  if (threadIdx.x < 128) red[threadIdx.x >> 6][threadIdx.x & 63] = 0.f;
  __syncthreads();
  float dw[2][8] = {};
  for (long i = blockIdx.x * 256L + threadIdx.x; i < (total + 255) / 256 * 256; i += (long)gridDim.x * 256L) {
    const bool live = i < total;
    const long row = live ? i / cpr : 0;
    const int c = live ? (int)(i - row * cpr) : 0, part = c / cpp, d0 = (c % 8) * 8;
    const long off = row * 3 * D + (long)c * 8;
    const bf16x8 x = live ? *(const bf16x8*)(qkv + off) : bf16x8{};
    const bf16x8 dy = live ? *(const bf16x8*)(dqkv + off) : bf16x8{};
    const float r = live ? inv[row * (2 * D / 64) + (c / 8)] : 0.f;
    const float* w = part == 0 ? wq : wk;
    float xh[8], g[8], dot = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      xh[e] = bf2f(x[e]) * r;
      g[e] = bf2f(dy[e]) * w[d0 + e];
      dot += g[e] * xh[e];
    }
    dot += __shfl_xor(dot, 1, 64);
    dot += __shfl_xor(dot, 2, 64);
    dot += __shfl_xor(dot, 4, 64);
    if (!live) continue;
    dot *= 1.f / 64.f;
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      o[e] = f2bf(r * (g[e] - xh[e] * dot));
      dw[part][e] += bf2f(dy[e]) * bf2f(f2bf(xh[e]));
    }
    *(bf16x8*)(dqkv + off) = o;
  }
  // a thread always sees the same d0 (256 and the chunk counts are multiples of 8): LDS reduce, then one atomic per d and block
  const int d0 = (threadIdx.x % 8) * 8;
#pragma unroll
  for (int pt = 0; pt < 2; ++pt)
#pragma unroll
    for (int e = 0; e < 8; ++e) atomicAdd(&red[pt][d0 + e], dw[pt][e]);
  __syncthreads();
  if (threadIdx.x < 128) {
    const float v = red[threadIdx.x >> 6][threadIdx.x & 63];
    if (v != 0.f) atomicAdd((threadIdx.x >> 6 ? dwk : dwq) + (threadIdx.x & 63), v);
  }
}

}  // namespace vtp
using namespace vtp;

static inline int extras_grid(long items) {
  long b = (items + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

extern "C" int vtp_gather_image_rows(const float* src, const int* img_idx, float* dst, void* dst_bf16, int n_img, long N, int D,
                                     float scale, void* stream) {
  VTP_REQUIRE(src && img_idx && (dst || dst_bf16) && n_img > 0 && N > 0 && D > 0 && D % 4 == 0, "vtp_gather_image_rows: bad argument");
  hipLaunchKernelGGL(gather_image_rows_kernel, dim3(extras_grid((long)n_img * N * D / 4)), dim3(256), 0, (hipStream_t)stream, src,
                     img_idx, dst, (bf16*)dst_bf16, n_img, N, D, scale);
  return check_launch("gather_image_rows");
}

extern "C" int vtp_scatter_image_rows(const float* src, const int* img_idx, float* dst, int n_img, long N, int D, float alpha,
                                      int accumulate, void* stream) {
  VTP_REQUIRE(src && img_idx && dst && n_img > 0 && N > 0 && D > 0 && D % 4 == 0, "vtp_scatter_image_rows: bad argument");
  hipLaunchKernelGGL(scatter_image_rows_kernel, dim3(extras_grid((long)n_img * N * D / 4)), dim3(256), 0, (hipStream_t)stream, src,
                     img_idx, dst, n_img, N, D, alpha, accumulate);
  return check_launch("scatter_image_rows");
}

extern "C" int vtp_layerscale_wgrad(const float* G, const float* W, const float* bias, const float* colsum, const float* gamma,
                                    float* dW, float* db, float* dgamma, int N, int K, void* stream) {
  VTP_REQUIRE(G && W && colsum && gamma && dW && dgamma && N > 0 && K > 0 && K % 4 == 0, "vtp_layerscale_wgrad: bad argument");
  hipLaunchKernelGGL(layerscale_wgrad_kernel, dim3(cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, G, W, bias, colsum, gamma, dW, db,
                     dgamma, N, K);
  return check_launch("layerscale_wgrad");
}

extern "C" int vtp_scaled_transpose(const float* W, const float* gamma, void* dstT, int N, int K, void* stream) {
  VTP_REQUIRE(W && gamma && dstT && N > 0 && K > 0, "vtp_scaled_transpose: bad argument");
  hipLaunchKernelGGL(scaled_transpose_kernel, dim3(cdiv(K, 64), cdiv(N, 64)), dim3(256), 0, (hipStream_t)stream, W, gamma,
                     (bf16*)dstT, N, K);
  return check_launch("scaled_transpose");
}

extern "C" int vtp_qk_norm_fwd(const void* qkv, const float* wq, const float* wk, void* out, float* inv, long M, int D, float eps,
                               void* stream) {
  VTP_REQUIRE(qkv && wq && wk && out && inv && M > 0 && D > 0 && D % 64 == 0, "vtp_qk_norm_fwd: bad argument (head_dim 64: D %% 64 == 0)");
  hipLaunchKernelGGL(qk_norm_fwd_kernel, dim3(extras_grid(M * (3 * D / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16*)qkv, wq, wk,
                     (bf16*)out, inv, M, D, eps);
  return check_launch("qk_norm_fwd");
}

extern "C" int vtp_qk_norm_bwd(void* dqkv, const void* qkv, const float* inv, const float* wq, const float* wk, float* dwq, float* dwk,
                               long M, int D, void* stream) {
  VTP_REQUIRE(dqkv && qkv && inv && wq && wk && dwq && dwk && M > 0 && D > 0 && D % 64 == 0, "vtp_qk_norm_bwd: bad argument");
  long blocks = (M * (2 * D / 8) + 255) / 256;
  hipLaunchKernelGGL(qk_norm_bwd_kernel, dim3((int)(blocks > 1024 ? 1024 : blocks)), dim3(256), 0, (hipStream_t)stream, (bf16*)dqkv,
                     (const bf16*)qkv, inv, wq, wk, dwq, dwk, M, D);
  return check_launch("qk_norm_bwd");
}
