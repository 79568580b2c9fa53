// Loss-head variants the reference names but does not ship (parity UNPINNED -- each kernel
// is tested against an fp64 restatement of the upstream definition in oracle/loss_oracle.py):
//   * SigLIP pairwise sigmoid loss  (init_logit_bias set: vtp/models/vtp.py:180,185-188 creates `logit_bias`; OpenCLIP SigLipLoss)
//   * KoLeo nearest-neighbour entropy regulariser on the student cls tokens  (DINOv2 KoLeoLoss)
//   * Sinkhorn-Knopp centring of the teacher targets                         (DINOv2 sinkhorn_knopp_teacher, 3 iterations)
// All HBM / latency bound; the [B, B] similarity products of SigLIP reuse clip.hip's kernels (vtp_clip_logits / vtp_clip_grad_*).
#include "common.h"
#include "vtp_hip.h"

namespace vtp {

// ------------------------------------------------------------------------------------------------ SigLIP
// logits z[m,n] = L[m,n] + bias (L = scale * <I_m, T_n> from clip_logits_kernel), y = +1 on the diagonal (n == label0 + m) else -1
//   loss += w * sum_n softplus(-y z) ;  L <- G = dloss/dz = -w y sigmoid(-y z) ;  dls += sum G L (d z / d log_scale = L) ; db += sum G
__global__ __launch_bounds__(256) void siglip_pair_kernel(float* __restrict__ L, const float* __restrict__ bias, int N, int label0,
                                                          float w, float* __restrict__ loss_sum, float* __restrict__ dls_sum,
                                                          float* __restrict__ dbias_sum) {
  __shared__ float red[4];
  const int m = blockIdx.x;
  float* row = L + (long)m * N;
  const int label = label0 + m;
  const float b = bias[0];
  float loss = 0.f, dls = 0.f, db = 0.f;
  for (int n = threadIdx.x; n < N; n += 256) {
    const float l = row[n], z = l + b;
    const float y = n == label ? 1.f : -1.f;
    const float t = -y * z;
    loss += fmaxf(t, 0.f) + log1pf(__expf(-fabsf(t)));   // softplus(t), overflow-safe
    const float g = -w * y / (1.f + __expf(-t));           // -w y sigmoid(t)
    dls += g * l;
    db += g;
    row[n] = g;
  }
  loss = block_sum<4>(loss, red);
  dls = block_sum<4>(dls, red);
  db = block_sum<4>(db, red);
  if (threadIdx.x == 0) {
    unsafeAtomicAdd(loss_sum, w * loss);
    unsafeAtomicAdd(dls_sum, dls);
    unsafeAtomicAdd(dbias_sum, db);
  }
}

// ------------------------------------------------------------------------------------------------ KoLeo
// x: [B, D] f32 (already L2-normalised, eps 1e-8).  Block per row i: nearest neighbour j != i by largest dot product.
__global__ __launch_bounds__(256) void koleo_nn_kernel(const float* __restrict__ xn, int* __restrict__ nn, int B, int D) {
  extern __shared__ float xi[];
  __shared__ float bv[256];
  __shared__ int bi[256];
  const int i = blockIdx.x;
  for (int d = threadIdx.x; d < D; d += 256) xi[d] = xn[(long)i * D + d];
  __syncthreads();
  float best = -INFINITY;
  int arg = -1;
  for (int j = threadIdx.x; j < B; j += 256) {
    if (j == i) continue;
    const float* xj = xn + (long)j * D;
    float acc = 0.f;
    for (int d = 0; d < D; ++d) acc += xi[d] * xj[d];
    if (acc > best) {
      best = acc;
      arg = j;
    }
  }
  bv[threadIdx.x] = best;
  bi[threadIdx.x] = arg;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      const float o = bv[threadIdx.x + s];
      const int oj = bi[threadIdx.x + s];
      // ties -> the smaller index (torch.max returns the first maximal index)
      if (oj >= 0 && (o > bv[threadIdx.x] || (o == bv[threadIdx.x] && (bi[threadIdx.x] < 0 || oj < bi[threadIdx.x])))) {
        bv[threadIdx.x] = o;
        bi[threadIdx.x] = oj;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) nn[i] = bi[0];
}

// loss += -w * log(|x_i - x_nn + pd_eps| + eps) ;  d_xn[i] += g_i, d_xn[nn_i] -= g_i  with  g_i = -w (x_i - x_nn + pd_eps) / (dist (dist + eps))
__global__ __launch_bounds__(256) void koleo_loss_kernel(const float* __restrict__ xn, const int* __restrict__ nn, float* __restrict__ d_xn,
                                                         float* __restrict__ loss_sum, int B, int D, float w, float eps, float pd_eps) {
  __shared__ float red[4];
  const int i = blockIdx.x, j = nn[i];
  float s = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) {
    const float df = xn[(long)i * D + d] - xn[(long)j * D + d] + pd_eps;
    s += df * df;
  }
  s = block_sum<4>(s, red);
  const float dist = sqrtf(s);
  const float c = -w / (fmaxf(dist, 1e-30f) * (dist + eps));
  for (int d = threadIdx.x; d < D; d += 256) {
    const float g = c * (xn[(long)i * D + d] - xn[(long)j * D + d] + pd_eps);
    unsafeAtomicAdd(d_xn + (long)i * D + d, g);
    unsafeAtomicAdd(d_xn + (long)j * D + d, -g);
  }
  if (threadIdx.x == 0) unsafeAtomicAdd(loss_sum, -w * __logf(dist + eps));
}

// ------------------------------------------------------------------------------------------------ Sinkhorn-Knopp
// Q[t,k] = E[t,k] u[t] v[k],  E = exp(logits * inv_temp - shift).  The iterations only ever need two kinds of sums:
//   colsum[k] = sum_t E[t,k] u[t]        (over the samples, per prototype)   -> v[k] = 1 / (K colsum[k])
//   rowsum[t] = sum_k E[t,k] v[k]        (over the prototypes, per sample)   -> u[t] = 1 / (B_total rowsum[t])
// so Q is never materialised until the final pass.  Rows t >= n_rows (device count, optional) are padding.
__global__ __launch_bounds__(256) void sk_max_kernel(const bf16* __restrict__ logits, float inv_temp, float* __restrict__ mx_out,
                                                     int T, int K, const int* __restrict__ rows_dev) {
  __shared__ float red[4];
  if (rows_dev) T = min(T, rows_dev[0]);
  float mx = -INFINITY;
  const long total = (long)T * (K / 8);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const bf16x8 v = *(const bf16x8*)(logits + i * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) mx = fmaxf(mx, bf2f(v[e]) * inv_temp);
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    // float atomic max through the monotone int mapping (values may be negative)
    int* p = (int*)mx_out;
    int old = *p, assumed;
    do {
      assumed = old;
      if (__int_as_float(assumed) >= mx) break;
      old = atomicCAS(p, assumed, __float_as_int(mx));
    } while (assumed != old);
  }
}

// colsum[k] += sum_{t in this block's row range} E[t,k] u[t]     grid = (K / 2048, row blocks); thread owns 8 consecutive k
__global__ __launch_bounds__(256) void sk_colsum_kernel(const bf16* __restrict__ logits, const float* __restrict__ u,
                                                        const float* __restrict__ shift, float inv_temp, float* __restrict__ colsum,
                                                        int T, int K, int rows_per_block, const int* __restrict__ rows_dev) {
  if (rows_dev) T = min(T, rows_dev[0]);
  const int k = (blockIdx.x * 256 + threadIdx.x) * 8;
  if (k >= K) return;
  const int t0 = blockIdx.y * rows_per_block, t1 = min(T, t0 + rows_per_block);
  const float sh = shift[0];
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int t = t0; t < t1; ++t) {
    const bf16x8 v = *(const bf16x8*)(logits + (long)t * K + k);
    const float ut = u ? u[t] : 1.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += __expf(bf2f(v[e]) * inv_temp - sh) * ut;
  }
  if (t1 > t0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) unsafeAtomicAdd(colsum + k + e, acc[e]);
  }
}

// rowsum[t] = sum_k E[t,k] v[k]   (block per row)
__global__ __launch_bounds__(256) void sk_rowsum_kernel(const bf16* __restrict__ logits, const float* __restrict__ v,
                                                        const float* __restrict__ shift, float inv_temp, float* __restrict__ rowsum,
                                                        int K) {
  __shared__ float red[4];
  const bf16* row = logits + (long)blockIdx.x * K;
  const float sh = shift[0];
  float s = 0.f;
  for (int k = threadIdx.x * 8; k < K; k += 2048) {
    const bf16x8 x = *(const bf16x8*)(row + k);
    const f32x4 v0 = *(const f32x4*)(v + k), v1 = *(const f32x4*)(v + k + 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += __expf(bf2f(x[e]) * inv_temp - sh) * (e < 4 ? v0[e] : v1[e - 4]);
  }
  s = block_sum<4>(s, red);
  if (threadIdx.x == 0) rowsum[blockIdx.x] = s;
}

// v[k] = 1 / (K colsum[k])   |   u[t] = 1 / (count rowsum[t])   (count from device memory when given)
__global__ __launch_bounds__(256) void sk_recip_kernel(const float* __restrict__ sums, float* __restrict__ out, int n, float count,
                                                       const float* __restrict__ count_dev) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (count_dev) count = fmaxf(count_dev[0], 1.f);
  if (i < n) out[i] = 1.f / (count * fmaxf(sums[i], 1e-37f));
}

// probs[t,k] = bf16(E[t,k] u[t] v[k] * B_total)   (the targets handed to the cross entropy; rows sum to 1)
__global__ __launch_bounds__(256) void sk_final_kernel(const bf16* __restrict__ logits, const float* __restrict__ u,
                                                       const float* __restrict__ v, const float* __restrict__ shift, float inv_temp,
                                                       bf16* __restrict__ probs, int K, float count, const float* __restrict__ count_dev) {
  if (count_dev) count = fmaxf(count_dev[0], 1.f);
  const long t = blockIdx.x;
  const float sh = shift[0], ut = u[t] * count;
  for (int k = threadIdx.x * 8; k < K; k += 2048) {
    const bf16x8 x = *(const bf16x8*)(logits + t * K + k);
    const f32x4 v0 = *(const f32x4*)(v + k), v1 = *(const f32x4*)(v + k + 4);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(__expf(bf2f(x[e]) * inv_temp - sh) * ut * (e < 4 ? v0[e] : v1[e - 4]));
    *(bf16x8*)(probs + t * K + k) = o;
  }
}

__global__ void sk_set_kernel(float* p, float v) { p[0] = v; }

}  // namespace vtp
using namespace vtp;

extern "C" int vtp_siglip_pairs(float* logits, const float* logit_bias, int B_local, int B_all, int label_offset, float weight,
                                float* loss_sum, float* d_logit_scale, float* d_logit_bias, void* stream) {
  VTP_REQUIRE(logits && logit_bias && loss_sum && d_logit_scale && d_logit_bias, "vtp_siglip_pairs: null pointer");
  VTP_REQUIRE(B_local > 0 && B_all >= B_local && label_offset >= 0 && label_offset + B_local <= B_all, "vtp_siglip_pairs: bad shape");
  hipLaunchKernelGGL(siglip_pair_kernel, dim3(B_local), dim3(256), 0, (hipStream_t)stream, logits, logit_bias, B_all, label_offset, weight,
                     loss_sum, d_logit_scale, d_logit_bias);
  return check_launch("siglip_pairs");
}

extern "C" int vtp_koleo(const float* xn, int* nn_scratch, float* d_xn, float* loss_sum, int B, int D, float weight, float eps,
                         void* stream) {
  VTP_REQUIRE(xn && nn_scratch && d_xn && loss_sum && B > 1 && D > 0 && (size_t)D * 4 <= 65536, "vtp_koleo: bad argument (B > 1)");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(koleo_nn_kernel, dim3(B), dim3(256), D * 4, s, xn, nn_scratch, B, D);
  hipLaunchKernelGGL(koleo_loss_kernel, dim3(B), dim3(256), 0, s, xn, nn_scratch, d_xn, loss_sum, B, D, weight, eps, 1e-8f);
  return check_launch("koleo");
}

extern "C" int vtp_sinkhorn_knopp(const void* logits, float inv_temp, void* probs, float* u, float* v, float* scratch, int T, int K,
                                  float count, const float* count_dev, const int* n_rows_dev, int n_iters, int phase, void* stream) {
  // phase -1: the whole single-process algorithm; phases 0..: building blocks for the data-parallel trainer, which must
  // all-reduce the per-prototype sums (and B_total) between them -- see vtp_amd/train.py
  //   phase 0: scratch[0] = global max (to be max-all-reduced), u = 1          phase 1: scratch[1..K] = colsum (to be sum-all-reduced)
  //   phase 2: v = 1 / (K colsum); rowsum; u = 1 / (count rowsum)              phase 3: probs
  VTP_REQUIRE(logits && probs && u && v && scratch && T > 0 && K > 0 && K % 8 == 0, "vtp_sinkhorn_knopp: bad argument (K %% 8 == 0)");
  hipStream_t s = (hipStream_t)stream;
  const bf16* lg = (const bf16*)logits;
  float* shift = scratch;
  float* colsum = scratch + 8;
  float* rowsum = scratch + 8 + K;
  const int rpb = 64;
  const dim3 cgrid(cdiv(K, 2048), cdiv(T, rpb));
  auto do_max = [&]() {
    hipLaunchKernelGGL(sk_set_kernel, dim3(1), dim3(1), 0, s, shift, -INFINITY);
    long items = (long)T * (K / 8);
    int g = (int)((items + 255) / 256);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(sk_max_kernel, dim3(g), dim3(256), 0, s, lg, inv_temp, shift, T, K, n_rows_dev);
  };
  auto do_colsum = [&](const float* uu) {
    hipMemsetAsync(colsum, 0, (size_t)K * 4, s);
    hipLaunchKernelGGL(sk_colsum_kernel, cgrid, dim3(256), 0, s, lg, uu, shift, inv_temp, colsum, T, K, rpb, n_rows_dev);
  };
  auto do_v_u = [&]() {
    hipLaunchKernelGGL(sk_recip_kernel, dim3(cdiv(K, 256)), dim3(256), 0, s, colsum, v, K, (float)K, (const float*)nullptr);
    hipLaunchKernelGGL(sk_rowsum_kernel, dim3(T), dim3(256), 0, s, lg, v, shift, inv_temp, rowsum, K);
    hipLaunchKernelGGL(sk_recip_kernel, dim3(cdiv(T, 256)), dim3(256), 0, s, rowsum, u, T, count, count_dev);
  };
  auto do_final = [&]() {
    hipLaunchKernelGGL(sk_final_kernel, dim3(T), dim3(256), 0, s, lg, u, v, shift, inv_temp, (bf16*)probs, K, count, count_dev);
  };
  if (phase < 0) {
    do_max();
    for (int it = 0; it < n_iters; ++it) {
      do_colsum(it == 0 ? nullptr : u);
      do_v_u();
    }
    do_final();
  } else if (phase == 0) {
    do_max();
  } else if (phase == 1) {
    do_colsum(n_iters == 0 ? nullptr : u);  // n_iters doubles as "iteration index" in the phased form
  } else if (phase == 2) {
    do_v_u();
  } else {
    do_final();
  }
  return check_launch("sinkhorn_knopp");
}
