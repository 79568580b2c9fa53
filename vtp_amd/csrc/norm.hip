// RMSNorm / LayerNorm forward + backward for the fp32 residual stream (HBM-bound, one wave per row).
//   fwd:  y(bf16) = norm(x f32) * w (+ b);  stats[m] = (mean, rstd)    [normalization.py:17-22, nn.LayerNorm]
//   bwd:  dx f32 = dres + d(norm)/dx ;  dw/db accumulated with one atomicAdd per column per workgroup.
// Rows are read as float4 per lane (16 B), row-resident in registers (D <= 2048).
#include "common.h"
#include <cstdlib>
#include "vtp_hip.h"

namespace vtp {

constexpr int NORM_MAXC = 8;  // float4 chunks per lane -> D <= 8*64*4 = 2048 (template NC = ceil(D/256))

template <int KIND, int NC>  // KIND 0 rms, 1 layernorm; NC float4 chunks per lane
__global__ __launch_bounds__(256) void norm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ b, bf16* __restrict__ y,
                                                       float* __restrict__ stats, int M, int D, float eps,
                                                       uint8_t* __restrict__ y8 = nullptr, const float* __restrict__ q_scale = nullptr) {
  // y8 / q_scale: the fp8 forward path -- the bf16-rounded output leaves as e4m3(y * q_scale[0]) (1 byte per element) instead
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (size_t)row * D;
  f32x4 v[NC];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = (c * 64 + lane) * 4;
    if (col < D) {
      v[c] = *(const f32x4*)(xr + col);
      s += (KIND == 0) ? (v[c][0] * v[c][0] + v[c][1] * v[c][1] + v[c][2] * v[c][2] + v[c][3] * v[c][3])
                       : (v[c][0] + v[c][1] + v[c][2] + v[c][3]);
    }
  }
  s = wave_sum(s);
  float mean = 0.f, rstd;
  if (KIND == 0) {
    rstd = rsqrtf(s / D + eps);
  } else {
    mean = s / D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = (c * 64 + lane) * 4;
      if (col < D) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[c][e] - mean;
          q += d * d;
        }
      }
    }
    q = wave_sum(q);
    rstd = rsqrtf(q / D + eps);
  }
  bf16* yr = y + (size_t)row * D;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = (c * 64 + lane) * 4;
    if (col < D) {
      f32x4 wv = *(const f32x4*)(w + col);
      f32x4 o;
      if (KIND == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = v[c][e] * rstd * wv[e];
      } else {
        f32x4 bv = *(const f32x4*)(b + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[c][e] - mean) * rstd * wv[e] + bv[e];
      }
      const bf16x4 ob = __builtin_convertvector(o, bf16x4);
      if (y8) {
        const float qs = *q_scale;
        float q[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] = fminf(fmaxf(bf2f(ob[e]) * qs, -448.f), 448.f);
        uint32_t wq = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], 0u, false);
        wq = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], wq, true);
        *(uint32_t*)(y8 + (size_t)row * D + col) = wq;
      } else {
        *(bf16x4*)(yr + col) = ob;
      }
    }
  }
  if (lane == 0 && stats) {
    stats[2 * row] = mean;
    stats[2 * row + 1] = rstd;
  }
}

template <int KIND, int NC>
__global__ __launch_bounds__(256, 2) void norm_bwd_kernel(const bf16* __restrict__ dy, const float* __restrict__ x,
                                                       const float* __restrict__ w, const float* __restrict__ stats,
                                                       const float* __restrict__ dres, float* __restrict__ dx,
                                                       bf16* __restrict__ dxb, float* __restrict__ dw, float* __restrict__ db,
                                                       float* __restrict__ dxsum, int M, int D) {
  constexpr int R = NC <= 3 ? 2 : 1;  // rows in flight per wave: the x / dy loads of all R rows are issued before the first reduction
  const int lane = threadIdx.x & 63;
  const int wv_id = threadIdx.x >> 6;
  // dbacc: LayerNorm bias gradient (column sums of dy) -- or, for RMSNorm (no bias), the column sums of the bf16 OUTPUT
  // (dxsum): the bias gradient of the linear layer whose dy this output is, so that layer needs no separate colsum pass
  f32x4 wreg[NC], dwacc[NC], dbacc[NC], dsacc[KIND == 1 ? NC : 1];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = (c * 64 + lane) * 4;
    if (col < D) wreg[c] = *(const f32x4*)(w + col);
    dwacc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dbacc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (KIND == 1) dsacc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int row0 = (blockIdx.x * 4 + wv_id) * R; row0 < M; row0 += gridDim.x * 4 * R) {
    // narrow rows (NC <= 3): the residual-gradient row is fetched together with x and dy, ahead of the two wave
    // reductions (2 waves/SIMD leave 256 VGPRs per lane); wide rows load it after them to stay inside the budget
    constexpr bool PRE = NC <= 3;
    f32x4 xv[R][NC], dr[PRE ? R : 1][PRE ? NC : 1];
    bf16x4 gy[R][NC];
    float mean[R], rstd[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = min(row0 + r, M - 1);
      mean[r] = stats[2 * row];
      rstd[r] = stats[2 * row + 1];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int col = (c * 64 + lane) * 4;
        if (col < D) {
          xv[r][c] = *(const f32x4*)(x + (size_t)row * D + col);
          gy[r][c] = *(const bf16x4*)(dy + (size_t)row * D + col);
          if constexpr (PRE) {
            if (dres) dr[r][c] = *(const f32x4*)(dres + (size_t)row * D + col);
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = row0 + r;
      if (row >= M) break;
      f32x4 xh[NC], g[NC];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int col = (c * 64 + lane) * 4;
        if (col < D) {
          f32x4 gyf = __builtin_convertvector(gy[r][c], f32x4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            xh[c][e] = (xv[r][c][e] - mean[r]) * rstd[r];
            g[c][e] = gyf[e] * wreg[c][e];
            s1 += g[c][e];
            s2 += g[c][e] * xh[c][e];
            dwacc[c][e] += gyf[e] * xh[c][e];
            if (KIND == 1) dbacc[c][e] += gyf[e];
          }
        }
      }
      s2 = wave_sum(s2) / D;
      if (KIND == 1) s1 = wave_sum(s1) / D; else s1 = 0.f;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int col = (c * 64 + lane) * 4;
        if (col < D) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = rstd[r] * (g[c][e] - s1 - xh[c][e] * s2);
          if (dres) {
            if constexpr (PRE) o += dr[r][c];
            else o += *(const f32x4*)(dres + (size_t)row * D + col);
          }
          *(f32x4*)(dx + (size_t)row * D + col) = o;
          if (dxb) {
            const bf16x4 ob = __builtin_convertvector(o, bf16x4);
            *(bf16x4*)(dxb + (size_t)row * D + col) = ob;
            if (dxsum) {
              const f32x4 of = __builtin_convertvector(ob, f32x4);
              if (KIND == 1) dsacc[c] += of; else dbacc[c] += of;
            }
          }
        }
      }
    }
  }
  // reduce dw/db over the 4 waves of the block through LDS, then one atomic per column
  __shared__ float red[4][NC * 256];
  for (int pass = 0; pass < 3; ++pass) {
    // pass 0: dw | pass 1: db (LayerNorm) | pass 2: dxsum (RMSNorm keeps it in dbacc, LayerNorm in dsacc)
    if (pass == 0 && dw == nullptr) continue;
    if (pass == 1 && (KIND == 0 || db == nullptr)) continue;
    if (pass == 2 && (dxsum == nullptr || dxb == nullptr)) continue;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        red[wv_id][(c * 64 + lane) * 4 + e] =
            pass == 0 ? dwacc[c][e] : (pass == 1 ? dbacc[c][e] : (KIND == 1 ? dsacc[KIND == 1 ? c : 0][e] : dbacc[c][e]));
    __syncthreads();
    float* out = pass == 0 ? dw : (pass == 1 ? db : dxsum);
    for (int col = threadIdx.x; col < D; col += 256)
      unsafeAtomicAdd(out + col, red[0][col] + red[1][col] + red[2][col] + red[3][col]);
  }
}

}  // namespace vtp
using namespace vtp;

#define NORM_DISPATCH(KERNEL, KIND, ...)                                                            \
  do {                                                                                              \
    const int nc = cdiv(D, 256);                                                                    \
    if (nc <= 1) hipLaunchKernelGGL((KERNEL<KIND, 1>), grid, block, 0, (hipStream_t)stream, __VA_ARGS__);      \
    else if (nc == 2) hipLaunchKernelGGL((KERNEL<KIND, 2>), grid, block, 0, (hipStream_t)stream, __VA_ARGS__); \
    else if (nc == 3) hipLaunchKernelGGL((KERNEL<KIND, 3>), grid, block, 0, (hipStream_t)stream, __VA_ARGS__); \
    else if (nc == 4) hipLaunchKernelGGL((KERNEL<KIND, 4>), grid, block, 0, (hipStream_t)stream, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<KIND, 8>), grid, block, 0, (hipStream_t)stream, __VA_ARGS__);              \
  } while (0)

extern "C" int vtp_norm_fwd(const float* x, const float* w, const float* b, void* y, float* stats, int M, int D,
                            float eps, int kind, void* stream) {
  VTP_REQUIRE(x && w && y, "vtp_norm_fwd: null pointer");
  VTP_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= NORM_MAXC * 256, "vtp_norm_fwd: need 0 < D <= 2048, D %% 4 == 0 (D=%d)", D);
  VTP_REQUIRE(kind == 0 || (kind == 1 && b), "vtp_norm_fwd: kind must be 0 (rms) or 1 (layernorm, needs bias)");
  dim3 grid(cdiv(M, 4)), block(256);
  if (kind == 0)
    NORM_DISPATCH(norm_fwd_kernel, 0, x, w, b, (bf16*)y, stats, M, D, eps);
  else
    NORM_DISPATCH(norm_fwd_kernel, 1, x, w, b, (bf16*)y, stats, M, D, eps);
  return check_launch("norm_fwd");
}

extern "C" int vtp_norm_fwd_e4m3(const float* x, const float* w, const float* b, void* y8, const float* q_scale, float* stats, int M,
                                 int D, float eps, int kind, void* stream) {
  VTP_REQUIRE(x && w && y8 && q_scale, "vtp_norm_fwd_e4m3: null pointer");
  VTP_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= NORM_MAXC * 256, "vtp_norm_fwd_e4m3: need 0 < D <= 2048, D %% 4 == 0 (D=%d)", D);
  VTP_REQUIRE(kind == 0 || (kind == 1 && b), "vtp_norm_fwd_e4m3: kind must be 0 (rms) or 1 (layernorm, needs bias)");
  dim3 grid(cdiv(M, 4)), block(256);
  if (kind == 0)
    NORM_DISPATCH(norm_fwd_kernel, 0, x, w, b, (bf16*)nullptr, stats, M, D, eps, (uint8_t*)y8, q_scale);
  else
    NORM_DISPATCH(norm_fwd_kernel, 1, x, w, b, (bf16*)nullptr, stats, M, D, eps, (uint8_t*)y8, q_scale);
  return check_launch("norm_fwd_e4m3");
}

extern "C" int vtp_norm_bwd(const void* dy, const float* x, const float* w, const float* stats, const float* dres,
                            float* dx, void* dx_bf16, float* dw, float* db, float* dx_colsum, int M, int D, int kind,
                            void* stream) {
  VTP_REQUIRE(dy && x && w && stats && dx, "vtp_norm_bwd: null pointer");
  VTP_REQUIRE(!dx_colsum || dx_bf16, "vtp_norm_bwd: dx_colsum sums the bf16 output and needs dx_bf16");
  VTP_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= NORM_MAXC * 256, "vtp_norm_bwd: need 0 < D <= 2048, D %% 4 == 0 (D=%d)", D);
  VTP_REQUIRE(kind == 0 || kind == 1, "vtp_norm_bwd: kind must be 0 or 1");
  int blocks = cdiv(M, D <= 768 ? 8 : 4);
  // every block ends with one atomic per column and target (dw, db, the column sums of dx): with 1024 blocks that is 1024 adds queued on
  // each of ~2 300 addresses, ~11 us of serialised atomics -- a third of the launch at 8 192 rows.  Fewer, longer-running blocks
  // (tools/norm_bench.py, round 6): 8 192 rows 39.0 -> 25.0 us (256 blocks), 16 448 rows 45.1 -> 39.1, 34 144 rows 87.2 -> 83.7 (512;
  // 256 blocks no longer keep enough loads in flight there), 2 464 rows 15.4 -> 13.3 (128).  VTP_NORM_BWD_BLOCKS=n overrides.
  static int cap_env = -1;
  if (cap_env < 0) {
    const char* e = getenv("VTP_NORM_BWD_BLOCKS");
    cap_env = e && atoi(e) > 0 ? atoi(e) : 0;
  }
  const int cap = cap_env ? cap_env : (M >= 24576 ? 512 : (M >= 4096 ? 256 : 128));
  if (blocks > cap) blocks = cap;
  dim3 grid(blocks), block(256);
  if (kind == 0)
    NORM_DISPATCH(norm_bwd_kernel, 0, (const bf16*)dy, x, w, stats, dres, dx, (bf16*)dx_bf16, dw, db, dx_colsum, M, D);
  else
    NORM_DISPATCH(norm_bwd_kernel, 1, (const bf16*)dy, x, w, stats, dres, dx, (bf16*)dx_bf16, dw, db, dx_colsum, M, D);
  return check_launch("norm_bwd");
}
