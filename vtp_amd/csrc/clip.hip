// CLIP head + text-tower glue kernels (gfx950): token embedding, EOT-row gather/scatter, L2 normalisation and the
// contrastive (InfoNCE) loss.  These operate on a few hundred rows (B captions / images x D_t features): they are
// latency-bound, kept in fp32 (logits are scaled by up to 100, so bf16 features would cost ~0.4 in logit error).
//
// Reference call sites: token_embedding + positional_embedding (modeling_vtp.py:296-297), text_global_pool 'argmax'
// (text_transformer.py:213-228), F.normalize (modeling_vtp.py:276,310), logits = exp(logit_scale) * I @ T^T
// (modeling_vtp.py:329).  The loss itself is not in the reference (OpenCLIP ClipLoss convention, parity unpinned).
#include "common.h"
#include "vtp_hip.h"

namespace vtp {

// x[b*T + t, :] = table[ids[b,t], :] + pos[t, :]   ; eot[b] = argmax_t ids[b, t] (first maximum)
__global__ __launch_bounds__(256) void embed_tokens_kernel(const long* __restrict__ ids, const float* __restrict__ table,
                                                           const float* __restrict__ pos, float* __restrict__ x,
                                                           int* __restrict__ eot, int B, int T, int D) {
  const int b = blockIdx.x;
  const int d4 = D / 4;
  for (int i = threadIdx.x; i < T * d4; i += 256) {
    const int t = i / d4, c = i % d4;
    const long id = ids[(long)b * T + t];
    f32x4 e = *(const f32x4*)(table + id * D + 4 * c);
    f32x4 p = *(const f32x4*)(pos + (long)t * D + 4 * c);
    *(f32x4*)(x + ((long)b * T + t) * D + 4 * c) = e + p;
  }
  if (threadIdx.x == 0 && eot) {
    long best = ids[(long)b * T];
    int bi = 0;
    for (int t = 1; t < T; ++t) {
      const long v = ids[(long)b * T + t];
      if (v > best) { best = v; bi = t; }
    }
    eot[b] = bi;
  }
}

// d_table[ids[b,t], :] += dx[b*T+t, :]  (atomics: tokens repeat) ;  d_pos[t, :] += sum_b dx[b*T+t, :]
__global__ __launch_bounds__(256) void embed_tokens_bwd_kernel(const long* __restrict__ ids, const float* __restrict__ dx,
                                                               float* __restrict__ d_table, float* __restrict__ d_pos,
                                                               int B, int T, int D) {
  const int t = blockIdx.x;
  for (int d = threadIdx.x; d < D; d += 256) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
      const float g = dx[((long)b * T + t) * D + d];
      s += g;
      unsafeAtomicAdd(d_table + ids[(long)b * T + t] * D + d, g);
    }
    d_pos[(long)t * D + d] += s;
  }
}

// out[b, :] = x[b*T + idx[b], :]  (f32)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x, const int* __restrict__ idx,
                                                          float* __restrict__ out, int B, int T, int D) {
  const int b = blockIdx.x;
  const float* src = x + ((long)b * T + idx[b]) * D;
  for (int d = threadIdx.x; d < D; d += 256) out[(long)b * D + d] = src[d];
}

// dx[B*T, D] (f32) and dxb (bf16) = 0 except row b*T + idx[b] = dy[b, :]
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ dy, const int* __restrict__ idx,
                                                           float* __restrict__ dx, bf16* __restrict__ dxb, int B, int T, int D) {
  const int row = blockIdx.x;  // over B*T
  const int b = row / T, t = row % T;
  const bool hit = (t == idx[b]);
  for (int d = threadIdx.x; d < D; d += 256) {
    const float v = hit ? dy[(long)b * D + d] : 0.f;
    dx[(long)row * D + d] = v;
    if (dxb) dxb[(long)row * D + d] = f2bf(v);
  }
}

// y = x / max(||x||, eps) ; inv[b] = 1 / max(||x||, eps)          (F.normalize, eps 1e-12)
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         float* __restrict__ inv, int B, int D, float eps) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  float s = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) {
    const float v = x[(long)b * D + d];
    s += v * v;
  }
  s = block_sum<4>(s, red);
  const float r = 1.f / fmaxf(sqrtf(s), eps);
  for (int d = threadIdx.x; d < D; d += 256) y[(long)b * D + d] = x[(long)b * D + d] * r;
  if (threadIdx.x == 0) inv[b] = r;
}

// dx = inv * (dy - y * <y, dy>)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                         const float* __restrict__ inv, float* __restrict__ dx, int B, int D) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  float s = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) s += dy[(long)b * D + d] * y[(long)b * D + d];
  s = block_sum<4>(s, red);
  const float r = inv[b];
  for (int d = threadIdx.x; d < D; d += 256) dx[(long)b * D + d] = r * (dy[(long)b * D + d] - y[(long)b * D + d] * s);
}

// logits[m, n] = scale * <A[m,:], B[n,:]>   (block per m; A row staged in LDS)
__global__ __launch_bounds__(256) void clip_logits_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                          const float* __restrict__ logit_scale, float* __restrict__ logits,
                                                          int M, int N, int D) {
  extern __shared__ float arow[];
  const int m = blockIdx.x;
  for (int d = threadIdx.x; d < D; d += 256) arow[d] = A[(long)m * D + d];
  __syncthreads();
  const float s = __expf(logit_scale[0]);
  for (int n = threadIdx.x; n < N; n += 256) {
    const float* br = Bm + (long)n * D;
    float acc = 0.f;
    for (int d = 0; d < D; d += 4) {
      f32x4 bv = *(const f32x4*)(br + d);
      acc += arow[d] * bv[0] + arow[d + 1] * bv[1] + arow[d + 2] * bv[2] + arow[d + 3] * bv[3];
    }
    logits[(long)m * N + n] = s * acc;
  }
}

// row-wise cross entropy with label = label0 + m:  loss_sum += w * (lse - logit[label]);  logits <- G = w*(softmax - onehot)
// dls_sum += sum_n G[m,n] * logits[m,n]   (gradient w.r.t. the logit_scale parameter, since d logits / d ls = logits)
__global__ __launch_bounds__(256) void clip_ce_kernel(float* __restrict__ logits, int M, int N, int label0, float w,
                                                      float* __restrict__ loss_sum, float* __restrict__ dls_sum) {
  __shared__ float red[4];
  const int m = blockIdx.x;
  float* row = logits + (long)m * N;
  const int label = label0 + m;
  const float l_label = row[label];  // read before any thread overwrites the row with G (barriers below order it)
  float mx = -INFINITY;
  for (int n = threadIdx.x; n < N; n += 256) mx = fmaxf(mx, row[n]);
  mx = wave_max(mx);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float se = 0.f;
  for (int n = threadIdx.x; n < N; n += 256) se += __expf(row[n] - mx);
  se = block_sum<4>(se, red);
  const float lse = mx + __logf(se);
  float dls = 0.f;
  for (int n = threadIdx.x; n < N; n += 256) {
    const float l = row[n];
    const float g = w * (__expf(l - lse) - (n == label ? 1.f : 0.f));
    dls += g * l;
    row[n] = g;
  }
  dls = block_sum<4>(dls, red);
  if (threadIdx.x == 0) {
    unsafeAtomicAdd(loss_sum, w * (lse - l_label));
    unsafeAtomicAdd(dls_sum, dls);
  }
}

// out[m, :] (+)= scale * sum_n G[m,n] * B[n, :]        (block per m)
__global__ __launch_bounds__(256) void clip_gb_kernel(const float* __restrict__ G, const float* __restrict__ Bm,
                                                      const float* __restrict__ logit_scale, float* __restrict__ out,
                                                      int M, int N, int D, int accumulate) {
  extern __shared__ float grow[];
  const int m = blockIdx.x;
  for (int n = threadIdx.x; n < N; n += 256) grow[n] = G[(long)m * N + n];
  __syncthreads();
  const float s = __expf(logit_scale[0]);
  for (int d = threadIdx.x; d < D; d += 256) {
    float acc = 0.f;
    for (int n = 0; n < N; ++n) acc += grow[n] * Bm[(long)n * D + d];
    const long o = (long)m * D + d;
    out[o] = (accumulate ? out[o] : 0.f) + s * acc;
  }
}

// out[n, :] (+)= scale * sum_m G[m,n] * A[m, :]        (block per n)
__global__ __launch_bounds__(256) void clip_gta_kernel(const float* __restrict__ G, const float* __restrict__ A,
                                                       const float* __restrict__ logit_scale, float* __restrict__ out,
                                                       int M, int N, int D, int accumulate) {
  extern __shared__ float gcol[];
  const int n = blockIdx.x;
  for (int m = threadIdx.x; m < M; m += 256) gcol[m] = G[(long)m * N + n];
  __syncthreads();
  const float s = __expf(logit_scale[0]);
  for (int d = threadIdx.x; d < D; d += 256) {
    float acc = 0.f;
    for (int m = 0; m < M; ++m) acc += gcol[m] * A[(long)m * D + d];
    const long o = (long)n * D + d;
    out[o] = (accumulate ? out[o] : 0.f) + s * acc;
  }
}

}  // namespace vtp
using namespace vtp;

extern "C" int vtp_embed_tokens(const long* ids, const float* table, const float* pos, float* x, int* eot, int B, int T, int D,
                                void* stream) {
  VTP_REQUIRE(ids && table && pos && x && B > 0 && T > 0 && D > 0 && D % 4 == 0, "vtp_embed_tokens: bad argument (D %% 4 == 0)");
  hipLaunchKernelGGL(embed_tokens_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, ids, table, pos, x, eot, B, T, D);
  return check_launch("embed_tokens");
}

extern "C" int vtp_embed_tokens_bwd(const long* ids, const float* dx, float* d_table, float* d_pos, int B, int T, int D,
                                    void* stream) {
  VTP_REQUIRE(ids && dx && d_table && d_pos && B > 0 && T > 0 && D > 0, "vtp_embed_tokens_bwd: bad argument");
  hipLaunchKernelGGL(embed_tokens_bwd_kernel, dim3(T), dim3(256), 0, (hipStream_t)stream, ids, dx, d_table, d_pos, B, T, D);
  return check_launch("embed_tokens_bwd");
}

extern "C" int vtp_gather_rows(const float* x, const int* idx, float* out, int B, int T, int D, void* stream) {
  VTP_REQUIRE(x && idx && out && B > 0 && T > 0 && D > 0, "vtp_gather_rows: bad argument");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, idx, out, B, T, D);
  return check_launch("gather_rows");
}

extern "C" int vtp_scatter_rows(const float* dy, const int* idx, float* dx, void* dx_bf16, int B, int T, int D, void* stream) {
  VTP_REQUIRE(dy && idx && dx && B > 0 && T > 0 && D > 0, "vtp_scatter_rows: bad argument");
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(B * T), dim3(256), 0, (hipStream_t)stream, dy, idx, dx, (bf16*)dx_bf16, B, T, D);
  return check_launch("scatter_rows");
}

extern "C" int vtp_l2norm_fwd(const float* x, float* y, float* inv_norm, int B, int D, float eps, void* stream) {
  VTP_REQUIRE(x && y && inv_norm && B > 0 && D > 0, "vtp_l2norm_fwd: bad argument");
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, y, inv_norm, B, D, eps);
  return check_launch("l2norm_fwd");
}

extern "C" int vtp_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, float* dx, int B, int D, void* stream) {
  VTP_REQUIRE(dy && y && inv_norm && dx && B > 0 && D > 0, "vtp_l2norm_bwd: bad argument");
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dy, y, inv_norm, dx, B, D);
  return check_launch("l2norm_bwd");
}

// building blocks of the pairwise contrastive losses, exported for the SigLIP variant (vtp_siglip_pairs in losses.hip):
//   logits[m,n] = exp(logit_scale) <A[m,:], B[n,:]> ;  out[m,:] (+)= exp(ls) sum_n G[m,n] B[n,:] ;  out[n,:] (+)= exp(ls) sum_m G[m,n] A[m,:]
extern "C" int vtp_clip_logits(const float* A, const float* Bm, const float* logit_scale, float* logits, int M, int N, int D,
                               void* stream) {
  VTP_REQUIRE(A && Bm && logit_scale && logits && M > 0 && N > 0 && D > 0 && D % 4 == 0 && (size_t)D * 4 <= 65536, "vtp_clip_logits: bad argument");
  hipLaunchKernelGGL(clip_logits_kernel, dim3(M), dim3(256), D * 4, (hipStream_t)stream, A, Bm, logit_scale, logits, M, N, D);
  return check_launch("clip_logits");
}

extern "C" int vtp_clip_grad_rows(const float* G, const float* Bm, const float* logit_scale, float* out, int M, int N, int D,
                                  int accumulate, void* stream) {
  VTP_REQUIRE(G && Bm && logit_scale && out && M > 0 && N > 0 && D > 0 && (size_t)N * 4 <= 65536, "vtp_clip_grad_rows: bad argument");
  hipLaunchKernelGGL(clip_gb_kernel, dim3(M), dim3(256), N * 4, (hipStream_t)stream, G, Bm, logit_scale, out, M, N, D, accumulate);
  return check_launch("clip_grad_rows");
}

extern "C" int vtp_clip_grad_cols(const float* G, const float* A, const float* logit_scale, float* out, int M, int N, int D,
                                  int accumulate, void* stream) {
  VTP_REQUIRE(G && A && logit_scale && out && M > 0 && N > 0 && D > 0 && (size_t)M * 4 <= 65536, "vtp_clip_grad_cols: bad argument");
  hipLaunchKernelGGL(clip_gta_kernel, dim3(N), dim3(256), M * 4, (hipStream_t)stream, G, A, logit_scale, out, M, N, D, accumulate);
  return check_launch("clip_grad_cols");
}

extern "C" int vtp_clip_loss(const float* img_local, const float* txt_local, const float* img_all, const float* txt_all,
                             const float* logit_scale, int B_local, int B_all, int D, int label_offset, float* loss_sum,
                             float* d_img_local, float* d_txt_local, float* d_img_all, float* d_txt_all,
                             float* d_logit_scale, float* scratch /* 2 * B_local * B_all floats */, void* stream) {
  VTP_REQUIRE(img_local && txt_local && img_all && txt_all && logit_scale && loss_sum && d_img_local && d_txt_local &&
                  d_img_all && d_txt_all && d_logit_scale && scratch,
              "vtp_clip_loss: null pointer");
  VTP_REQUIRE(B_local > 0 && B_all >= B_local && D > 0 && D % 4 == 0 && label_offset >= 0 && label_offset + B_local <= B_all,
              "vtp_clip_loss: bad shape");
  VTP_REQUIRE((size_t)D * 4 <= 65536 && (size_t)B_all * 4 <= 65536, "vtp_clip_loss: D and B_all must fit an LDS row (<= 16384)");
  hipStream_t s = (hipStream_t)stream;
  float* L1 = scratch;                         // image -> text logits [B_local, B_all]
  float* L2 = scratch + (size_t)B_local * B_all;  // text -> image logits
  const float w = 0.5f / (float)B_local;
  hipLaunchKernelGGL(clip_logits_kernel, dim3(B_local), dim3(256), D * 4, s, img_local, txt_all, logit_scale, L1, B_local, B_all, D);
  hipLaunchKernelGGL(clip_logits_kernel, dim3(B_local), dim3(256), D * 4, s, txt_local, img_all, logit_scale, L2, B_local, B_all, D);
  hipLaunchKernelGGL(clip_ce_kernel, dim3(B_local), dim3(256), 0, s, L1, B_local, B_all, label_offset, w, loss_sum, d_logit_scale);
  hipLaunchKernelGGL(clip_ce_kernel, dim3(B_local), dim3(256), 0, s, L2, B_local, B_all, label_offset, w, loss_sum, d_logit_scale);
  // local-feature gradients: dI_l = s * G1 * T_all ; dT_l = s * G2 * I_all
  hipLaunchKernelGGL(clip_gb_kernel, dim3(B_local), dim3(256), B_all * 4, s, L1, txt_all, logit_scale, d_img_local, B_local, B_all, D, 0);
  hipLaunchKernelGGL(clip_gb_kernel, dim3(B_local), dim3(256), B_all * 4, s, L2, img_all, logit_scale, d_txt_local, B_local, B_all, D, 0);
  // gathered-feature gradients (to be reduce-scattered across ranks): dT_all = s * G1^T * I_l ; dI_all = s * G2^T * T_l
  hipLaunchKernelGGL(clip_gta_kernel, dim3(B_all), dim3(256), B_local * 4, s, L1, img_local, logit_scale, d_txt_all, B_local, B_all, D, 0);
  hipLaunchKernelGGL(clip_gta_kernel, dim3(B_all), dim3(256), B_local * 4, s, L2, txt_local, logit_scale, d_img_all, B_local, B_all, D, 0);
  return check_launch("clip_loss");
}
