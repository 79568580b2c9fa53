// Attention for short sequences (N <= 320: the 257-token global crops and 37-token local crops of the trunk, 256-token
// decoder): one workgroup per (image, head) keeps the WHOLE K/V (forward, dQ) or Q/dO (dK/dV) of that head in LDS, so the
// key loop has no barriers and no global loads.  Same math and MFMA operand mapping as the tiled kernels of attention.hip
// (swapped product S^T = K Q^T, one lane = one query; plain product for dK/dV, one lane = one key).
//
// LDS image of a [rows][64] bf16 matrix: 128-B rows, 16-B chunk c of row r stored at chunk c ^ key(r),
//   key(r) = (((r>>1)&1) << 2) | ((r>>2)&3)
// -- a bit-permutation of (r>>1)&7, so ds_read_b128 of 32 different rows (A-operand row fragments) is conflict-free exactly as
// in the GEMM, and bit 2 alternates between rows r and r+2, so the four rows x 64 B of a ds_read_b64_tr_b16 half-wave land on
// disjoint bank quarters.  The image is filled by LDS-DMA (global_load_lds_dwordx4), the swizzle applied on the source side.
//
// Forward is two-pass over the resident keys (row maximum first, then exp / sum / PV): no running rescale of the accumulator.
#include "common.h"
#include <cstdlib>
#include "vtp_hip.h"

namespace vtp {

constexpr float LOG2E_R = 1.4426950408889634f;
constexpr float LN2_R = 0.6931471805599453f;
constexpr int RES_MAXN = 320;

struct AttnResArgs {
  const bf16 *q, *k, *v, *o, *d_o;
  bf16 *out, *dq, *dk, *dv;
  float* lse;
  float* delta;           // written by the dQ kernel (rowsum(dO * O)), read by the dK/dV kernel that follows it
  const bf16 *rope_sin, *rope_cos;  // [hw, 64] tables or null: dq / dk leave inverse-rotated (gradient w.r.t. the un-rotated q, k)
  int rope_prefix;
  int B, N, heads, npad;  // npad = N rounded up to 32
  long sb, sn, sbo, sno;
  float scale;
};

__device__ __forceinline__ int swz_key(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }

// all waves: rows [0, npad) of `base` (row stride sn elements, rows clamped to N-1) -> swizzled LDS image
__device__ __forceinline__ void stage_resident(const bf16* __restrict__ base, long sn, int N, int npad, char* lds, int wave,
                                               int nwaves, int lane) {
  const int rr = lane >> 3, cp = lane & 7;
  for (int piece = wave; piece < (npad >> 3); piece += nwaves) {
    const int row = piece * 8 + rr;
    const int c = cp ^ swz_key(row);
    const bf16* src = base + (long)min(row, N - 1) * sn + c * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds + piece * 1024), 16, 0, 0);
  }
}

// A-operand fragment, row-major: lane row = row0 + (lane&31), k-step ks (16 d): 16 B at d = 16 ks + 8 hi
__device__ __forceinline__ bf16x8 frag_row(const char* lds, int row0, int ks, int lane) {
  const int r = row0 + (lane & 31);
  return *(const bf16x8*)(lds + r * 128 + (((2 * ks + (lane >> 5)) ^ swz_key(r)) << 4));
}
// A-operand fragment of the TRANSPOSED matrix (lane row = d = dblk*32 + (lane&31); k-slots = sequence positions
// row0 + 16 ks + 4 hi + {0..3} and + 8 + {0..3}, the accumulator register order) via ds_read_b64_tr_b16
__device__ __forceinline__ bf16x8 frag_tr(const char* lds, int dblk, int row0, int ks, int lane) {
  const int i = lane & 15, g = (lane >> 4) & 1, hi = lane >> 5;
  const int r = row0 + ks * 16 + hi * 4 + (i >> 2);
  const int col = dblk * 32 + g * 16 + (i & 3) * 4;
  const char* p0 = lds + r * 128 + (((col >> 3) ^ swz_key(r)) << 4) + ((col & 7) << 1);
  const char* p1 = lds + (r + 8) * 128 + (((col >> 3) ^ swz_key(r + 8)) << 4) + ((col & 7) << 1);
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p0);
  bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p1);
  return __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ bf16x8 pack8r(const f32x16& a, int base) {
  f32x4 x = {a[base], a[base + 1], a[base + 2], a[base + 3]}, y = {a[base + 4], a[base + 5], a[base + 6], a[base + 7]};
  bf16x4 xb = __builtin_convertvector(x, bf16x4), yb = __builtin_convertvector(y, bf16x4);
  return __builtin_shufflevector(xb, yb, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ void zero16r(f32x16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
}
__device__ __forceinline__ void wait_all_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// wait until at most n (wave-uniform, small) vector-memory operations of this wave are outstanding
__device__ __forceinline__ void wait_vmcnt_upto(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;  // conservative
  }
}

// inverse RoPE of one gradient row held as two f32x4 pieces (d = d0..d0+3 and d0+32..d0+35), bit-identical to
// rope_qk_kernel<true> applied to the bf16-rounded values: dx1 = g1*cos1 + g2*sin2 ; dx2 = g2*cos2 - g1*sin1
__device__ __forceinline__ void store_grad_pair(bf16* row, int d0, f32x4 lo, f32x4 hi, const bf16* sin_t, const bf16* cos_t, long t) {
  bf16x4 a4 = __builtin_convertvector(lo, bf16x4), b4 = __builtin_convertvector(hi, bf16x4);
  if (sin_t) {
    const bf16x4 c1 = *(const bf16x4*)(cos_t + t + d0), c2 = *(const bf16x4*)(cos_t + t + d0 + 32);
    const bf16x4 s1 = *(const bf16x4*)(sin_t + t + d0), s2 = *(const bf16x4*)(sin_t + t + d0 + 32);
    bf16x4 o1, o2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = bf2f(a4[e]), b = bf2f(b4[e]);
      o1[e] = f2bf(bf2f(f2bf(a * bf2f(c1[e]))) + bf2f(f2bf(b * bf2f(s2[e]))));
      o2[e] = f2bf(bf2f(f2bf(b * bf2f(c2[e]))) + bf2f(f2bf(-a * bf2f(s1[e]))));
    }
    a4 = o1;
    b4 = o2;
  }
  *(bf16x4*)(row + d0) = a4;
  *(bf16x4*)(row + d0 + 32) = b4;
}

// ------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(640) void attn_fwd_res_kernel(const AttnResArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + p.npad * 128;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6, hi = lane >> 5;
  const int h = blockIdx.x, b = blockIdx.y;
  const bf16* qb = p.q + (long)b * p.sb + h * 64;
  const int q0 = (blockIdx.z * nwaves + wave) * 32, qi = q0 + (lane & 31);  // a head's query blocks are split over gridDim.z
  const bool active = q0 < p.npad;                                          // workgroups (each stages all of K / V)
  bf16x8 qf[4];
  {  // issued first: vmcnt retires in order, and pass 1 needs q and K but not V
    const bf16* qr = qb + (long)min(qi, p.N - 1) * p.sn + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(qr + ks * 16);
  }
  stage_resident(p.k + (long)b * p.sb + h * 64, p.sn, p.N, p.npad, Ks, wave, nwaves, lane);
  stage_resident(p.v + (long)b * p.sb + h * 64, p.sn, p.N, p.npad, Vs, wave, nwaves, lane);
  // the youngest operations of this wave are its V pieces, which stream in behind pass 1 (vmcnt retires in order)
  wait_vmcnt_upto(((p.npad >> 3) - wave + nwaves - 1) / nwaves);
  __builtin_amdgcn_s_barrier();  // raw barrier: __syncthreads() would drain vmcnt (the V pieces) first
  asm volatile("" ::: "memory");
  const float sc2 = p.scale * LOG2E_R;
  const int nkb = p.npad >> 5, last = nkb - 1;
  // pass 1: row maximum of the raw scores (scale > 0 commutes with max)
  float mx = -INFINITY;
  for (int kb = 0; kb < (active ? nkb : 0); ++kb) {
    f32x16 s;
    zero16r(s);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(Ks, kb * 32, ks, lane), qf[ks], s, 0, 0, 0);
    if (kb == last) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= p.N) s[r] = -INFINITY;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[r]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float m2 = mx * sc2;
  wait_all_dma();   // V landed
  __syncthreads();
  if (!active) return;
  // pass 2: p = exp2(s*sc2 - m2), l = sum p, O^T += V^T P^T
  f32x16 oacc[2];
  zero16r(oacc[0]);
  zero16r(oacc[1]);
  float l = 0.f;
  for (int kb = 0; kb < nkb; ++kb) {
    f32x16 s;
    zero16r(s);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(Ks, kb * 32, ks, lane), qf[ks], s, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = fast_exp2(s[r] * sc2 - m2);
    if (kb == last) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= p.N) s[r] = 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) l += s[r];
    const bf16x8 pf0 = pack8r(s, 0), pf1 = pack8r(s, 8);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Vs, db, kb * 32, 0, lane), pf0, oacc[db], 0, 0, 0);
      oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Vs, db, kb * 32, 1, lane), pf1, oacc[db], 0, 0, 0);
    }
  }
  l += __shfl_xor(l, 32, 64);
  if (qi >= p.N) return;
  const float inv = 1.f / l;
  bf16* orow = p.out + (long)b * p.sbo + (long)qi * p.sno + h * 64;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = oacc[db][4 * g + e] * inv;
      *(bf16x4*)(orow + db * 32 + 8 * g + 4 * hi) = __builtin_convertvector(v, bf16x4);
    }
  if (hi == 0 && p.lse) p.lse[((long)b * p.heads + h) * p.N + qi] = (m2 + log2f(l)) * LN2_R;
}

// ------------------------------------------------------------------------------------------------ backward: dQ
__global__ __launch_bounds__(640) void attn_bwd_dq_res_kernel(const AttnResArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + p.npad * 128;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6, hi = lane >> 5;
  const int h = blockIdx.x, b = blockIdx.y;
  stage_resident(p.k + (long)b * p.sb + h * 64, p.sn, p.N, p.npad, Ks, wave, nwaves, lane);
  stage_resident(p.v + (long)b * p.sb + h * 64, p.sn, p.N, p.npad, Vs, wave, nwaves, lane);
  const int q0 = (blockIdx.z * nwaves + wave) * 32, qi = q0 + (lane & 31), qc = min(qi, p.N - 1);
  bf16x8 qf[4], dof[4];
  {
    const bf16* qr = p.q + (long)b * p.sb + h * 64 + (long)qc * p.sn + hi * 8;
    const bf16* gr = p.d_o + (long)b * p.sbo + h * 64 + (long)qc * p.sno + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = *(const bf16x8*)(qr + ks * 16);
      dof[ks] = *(const bf16x8*)(gr + ks * 16);
    }
  }
  const long srow = ((long)b * p.heads + h) * p.N + qc;
  const float lse2 = p.lse[srow] * LOG2E_R;
  float dlt = 0.f;  // delta = rowsum(dO * O): this lane's 32 elements + the other half-wave's
  {
    const bf16* orow = p.o + (long)b * p.sbo + h * 64 + (long)qc * p.sno + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 ov = *(const bf16x8*)(orow + ks * 16);
#pragma unroll
      for (int e = 0; e < 8; ++e) dlt += bf2f(ov[e]) * bf2f(dof[ks][e]);
    }
    dlt += __shfl_xor(dlt, 32, 64);
    if (hi == 0 && qi < p.N) p.delta[srow] = dlt;
  }
  wait_all_dma();
  __syncthreads();
  if (q0 >= p.npad) return;
  const float sc2 = p.scale * LOG2E_R;
  const int nkb = p.npad >> 5, last = nkb - 1;
  f32x16 dq[2];
  zero16r(dq[0]);
  zero16r(dq[1]);
  for (int kb = 0; kb < nkb; ++kb) {
    f32x16 s, dp;
    zero16r(s);
    zero16r(dp);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(Ks, kb * 32, ks, lane), qf[ks], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(Vs, kb * 32, ks, lane), dof[ks], dp, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = fast_exp2(s[r] * sc2 - lse2) * (dp[r] - dlt) * p.scale;
    if (kb == last) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= p.N) s[r] = 0.f;
    }
    const bf16x8 d0 = pack8r(s, 0), d1 = pack8r(s, 8);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Ks, db, kb * 32, 0, lane), d0, dq[db], 0, 0, 0);
      dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Ks, db, kb * 32, 1, lane), d1, dq[db], 0, 0, 0);
    }
  }
  if (qi >= p.N) return;
  bf16* drow = p.dq + (long)b * p.sb + (long)qi * p.sn + h * 64;
  const bool rot = p.rope_sin && qi >= p.rope_prefix;
  const long t = (long)(qi - p.rope_prefix) * 64;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 lo, hv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      lo[e] = dq[0][4 * g + e];
      hv[e] = dq[1][4 * g + e];
    }
    store_grad_pair(drow, 8 * g + 4 * hi, lo, hv, rot ? p.rope_sin : nullptr, p.rope_cos, t);
  }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
__global__ __launch_bounds__(640) void attn_bwd_dkv_res_kernel(const AttnResArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qs = smem;
  char* Gs = smem + p.npad * 128;
  float* lse_s = (float*)(smem + 2 * p.npad * 128);
  float* dlt_s = lse_s + p.npad;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6, hi = lane >> 5;
  const int h = blockIdx.x, b = blockIdx.y;
  stage_resident(p.q + (long)b * p.sb + h * 64, p.sn, p.N, p.npad, Qs, wave, nwaves, lane);
  stage_resident(p.d_o + (long)b * p.sbo + h * 64, p.sno, p.N, p.npad, Gs, wave, nwaves, lane);
  const long srow0 = ((long)b * p.heads + h) * p.N;
  for (int i = threadIdx.x; i < p.npad; i += blockDim.x) {
    const int qn = min(i, p.N - 1);
    lse_s[i] = p.lse[srow0 + qn] * LOG2E_R;
    dlt_s[i] = p.delta[srow0 + qn];
  }
  const int k0 = (blockIdx.z * nwaves + wave) * 32, ki = k0 + (lane & 31), kc = min(ki, p.N - 1);
  bf16x8 kf[4], vf[4];
  {
    const bf16* kr = p.k + (long)b * p.sb + h * 64 + (long)kc * p.sn + hi * 8;
    const bf16* vr = p.v + (long)b * p.sb + h * 64 + (long)kc * p.sn + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = *(const bf16x8*)(kr + ks * 16);
      vf[ks] = *(const bf16x8*)(vr + ks * 16);
    }
  }
  wait_all_dma();
  __syncthreads();
  if (k0 >= p.npad) return;
  const float sc2 = p.scale * LOG2E_R;
  const int nqb = p.npad >> 5, last = nqb - 1;
  f32x16 dk[2], dv[2];
  zero16r(dk[0]);
  zero16r(dk[1]);
  zero16r(dv[0]);
  zero16r(dv[1]);
  for (int qblk = 0; qblk < nqb; ++qblk) {
    f32x16 s, dp;
    zero16r(s);
    zero16r(dp);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(Qs, qblk * 32, ks, lane), kf[ks], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(Gs, qblk * 32, ks, lane), vf[ks], dp, 0, 0, 0);
    }
    f32x16 pr;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 l4 = *(const f32x4*)(lse_s + qblk * 32 + 8 * g + 4 * hi);
      const f32x4 d4 = *(const f32x4*)(dlt_s + qblk * 32 + 8 * g + 4 * hi);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        float pv = fast_exp2(s[r] * sc2 - l4[e]);
        if (qblk == last && qblk * 32 + 8 * g + 4 * hi + e >= p.N) pv = 0.f;
        pr[r] = pv;
        s[r] = pv * (dp[r] - d4[e]) * p.scale;
      }
    }
    const bf16x8 p0 = pack8r(pr, 0), p1 = pack8r(pr, 8), d0 = pack8r(s, 0), d1 = pack8r(s, 8);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Gs, db, qblk * 32, 0, lane), p0, dv[db], 0, 0, 0);
      dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Gs, db, qblk * 32, 1, lane), p1, dv[db], 0, 0, 0);
      dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Qs, db, qblk * 32, 0, lane), d0, dk[db], 0, 0, 0);
      dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Qs, db, qblk * 32, 1, lane), d1, dk[db], 0, 0, 0);
    }
  }
  if (ki >= p.N) return;
  bf16* krow = p.dk + (long)b * p.sb + (long)ki * p.sn + h * 64;
  bf16* vrow = p.dv + (long)b * p.sb + (long)ki * p.sn + h * 64;
  const bool rot = p.rope_sin && ki >= p.rope_prefix;
  const long t = (long)(ki - p.rope_prefix) * 64;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 lo, hv, c0, c1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      lo[e] = dk[0][4 * g + e];
      hv[e] = dk[1][4 * g + e];
      c0[e] = dv[0][4 * g + e];
      c1[e] = dv[1][4 * g + e];
    }
    store_grad_pair(krow, 8 * g + 4 * hi, lo, hv, rot ? p.rope_sin : nullptr, p.rope_cos, t);
    store_grad_pair(vrow, 8 * g + 4 * hi, c0, c1, nullptr, nullptr, 0);
  }
}

// a head with >= 4 row blocks is split over two workgroups (each stages the whole K / V or Q / dO image, 2 x 37 KB at N = 257):
// two of them fit a CU, so one's staging overlaps the other's key loop
// -- worth it only while heads x images leaves CUs short of work (<= 512 workgroups: +1.2 % on the 32-image rec step, -0.4 %
// when the 64-image passes already run three full rounds and the second copy of K / V is pure extra traffic)
static int res_waves_per_block(int nw, int groups) {
  static const int mode = getenv("VTP_ATTN_SPLIT") ? atoi(getenv("VTP_ATTN_SPLIT")) : -1;  // 0 never, 1 always, -1 by size
  const bool split = mode == 1 || (mode != 0 && groups <= 512);
  return (split && nw >= 4) ? (nw + 1) / 2 : nw;
}

template <typename K>
static void set_lds(K kern, int bytes) {
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// host side, called by vtp_attn_fwd / vtp_attn_bwd (attention.hip) for non-causal N <= RES_MAXN
int attn_resident_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int N, int heads, long sb,
                      long sn, long sbo, long sno, float scale, hipStream_t s) {
  AttnResArgs a = {};
  a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.out = (bf16*)o; a.lse = lse;
  a.B = B; a.N = N; a.heads = heads; a.npad = (N + 31) / 32 * 32; a.sb = sb; a.sn = sn; a.sbo = sbo; a.sno = sno; a.scale = scale;
  static bool attr = false;
  if (!attr) {
    set_lds(attn_fwd_res_kernel, 2 * RES_MAXN * 128);
    attr = true;
  }
  const int nw = a.npad / 32, wpb = res_waves_per_block(nw, heads * B);
  hipLaunchKernelGGL(attn_fwd_res_kernel, dim3(heads, B, (nw + wpb - 1) / wpb), dim3(64 * wpb), 2 * a.npad * 128, s, a);
  return check_launch("attn_fwd_resident");
}

int attn_resident_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                      float* delta, void* dq, void* dk, void* dv, const void* rope_sin, const void* rope_cos, int rope_prefix,
                      int B, int N, int heads, long sb, long sn, long sbo, long sno, float scale, hipStream_t s) {
  AttnResArgs a = {};
  a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.o = (const bf16*)o; a.d_o = (const bf16*)d_o;
  a.lse = (float*)lse; a.delta = delta; a.dq = (bf16*)dq; a.dk = (bf16*)dk; a.dv = (bf16*)dv;
  a.rope_sin = (const bf16*)rope_sin; a.rope_cos = (const bf16*)rope_cos; a.rope_prefix = rope_prefix;
  a.B = B; a.N = N; a.heads = heads; a.npad = (N + 31) / 32 * 32; a.sb = sb; a.sn = sn; a.sbo = sbo; a.sno = sno; a.scale = scale;
  static bool attr = false;
  if (!attr) {
    set_lds(attn_bwd_dq_res_kernel, 2 * RES_MAXN * 128);
    set_lds(attn_bwd_dkv_res_kernel, 2 * RES_MAXN * 128 + 8 * RES_MAXN);
    attr = true;
  }
  const int nw = a.npad / 32, wpb = res_waves_per_block(nw, heads * B);
  const dim3 grid(heads, B, (nw + wpb - 1) / wpb), block(64 * wpb);
  hipLaunchKernelGGL(attn_bwd_dq_res_kernel, grid, block, 2 * a.npad * 128, s, a);
  hipLaunchKernelGGL(attn_bwd_dkv_res_kernel, grid, block, 2 * a.npad * 128 + 8 * a.npad, s, a);
  return check_launch("attn_bwd_resident");
}

}  // namespace vtp
