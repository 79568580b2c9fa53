// Flash-style fused attention for gfx950, head_dim = 64, bf16 in / fp32 accumulate  (replaces
// F.scaled_dot_product_attention, attention.py:124, and nn.MultiheadAttention's core in the text tower).
// This file: the TILED kernels (any N, causal or not: 1025-token sequences at 512^2, the causal 77-token text tower) and the
// C entry points; short non-causal sequences dispatch to the LDS-resident kernels of attention_resident.hip.
//
// MFMA: v_mfma_f32_32x32x16_bf16.  D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
//
// Forward and the dQ pass use the *swapped* product S^T = K Q^T so that one lane owns one query row (column of
// S^T): the online softmax is lane-local (16 registers + one cross-half exchange), and P^T in the accumulator
// registers is directly the B operand of O^T = V^T P^T (register r = 8*ks + j is k-slot j of k-step ks for BOTH
// operands, so the hardware's slot->k mapping cancels).  The A operand V^T (and K^T for dQ) must be contiguous
// along keys: the [keys][64] tile is staged row-major and read with the LDS transpose read ds_read_b64_tr_b16.
// The dK/dV pass uses the plain product S = Q K^T (one lane owns one key) so P and dS are B operands of
// dV^T = dO^T P and dK^T = Q^T dS; Q^T / dO^T come from transpose reads of the row-major tiles as well.
//
// LDS images: tiles read as rows (16-B fragments of 32 different rows) use a 144-B row stride (16 distinct slots);
// tiles read transposed use a 192-B row stride (the 4 rows x 64 B of a half-wave land on disjoint bank quarters).
#include "common.h"
#include <cstdlib>
#include "vtp_hip.h"

namespace vtp {

constexpr int KT = 64;        // keys (or queries) per staged tile
constexpr int RS = 72;        // row stride (elements) of row-major [64][64] tiles
constexpr int RT = 96;        // row stride (elements) of row-major tiles that are read with ds_read_b64_tr_b16 (4 rows x 64 B
                              // of one half-wave land on 4 disjoint 16-dword bank ranges: 0, 48, 32, 16)
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

struct AttnArgs {
  const bf16 *q, *k, *v, *o, *d_o;
  bf16 *out, *dq, *dk, *dv;
  float* lse;
  float* delta;
  int B, N, heads;
  long sb, sn;    // q/k/v (and dq/dk/dv) batch / token strides in elements
  long sbo, sno;  // o / d_o strides
  float scale;
};

__device__ __forceinline__ bf16x8 cat4(bf16x4 a, bf16x4 b) {
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// stage a [64 rows][64 d] tile row-major into up to two LDS images with different row strides (16-B writes only)
template <int S1, int S2>
__device__ __forceinline__ void stage_tile2(const bf16* __restrict__ base, long sn, int row0, int N, bf16* d1, bf16* d2) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = threadIdx.x + i * 256;
    const int r = c >> 3, dc = (c & 7) * 8;
    const int n = min(row0 + r, N - 1);
    bf16x8 v = *(const bf16x8*)(base + (long)n * sn + dc);
    if (S1 > 0) *(bf16x8*)(d1 + r * S1 + dc) = v;
    if (S2 > 0) *(bf16x8*)(d2 + r * S2 + dc) = v;
  }
}

// A-operand fragment of the TRANSPOSED tile, formed with the gfx950 LDS transpose read from a row-major [seq][64 d]
// image (row stride RT): lane row d = dblk*32 + (lane&31); k-slots are seq = blk*32 + 16*ks + 4*hi + {0..3} and + 8 +
// {0..3} -- the accumulator register order.  In each 16-lane group lane i supplies the address of row (i>>2), columns
// 4*(i&3).., and receives column i of the 4 x 16 block.
__device__ __forceinline__ bf16x8 frag_trr(const bf16* tile, int dblk, int blk, int ks, int lane) {
  const int i = lane & 15, g = (lane >> 4) & 1, hi = lane >> 5;
  const bf16* p = tile + (blk * 32 + ks * 16 + hi * 4 + (i >> 2)) * RT + dblk * 32 + g * 16 + (i & 3) * 4;
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p);
  bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p + 8 * RT));
  return cat4(lo, hi4);
}

// A-operand fragment from a row-major tile: lane row (lane&31) of 32-row block `blk`, k-step ks (16 d), 16 B
__device__ __forceinline__ bf16x8 frag_rm(const bf16* rm, int blk, int ks, int lane) {
  return *(const bf16x8*)(rm + (blk * 32 + (lane & 31)) * RS + ks * 16 + (lane >> 5) * 8);
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int base) {
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = f2bf(a[base + e]);
  return r;
}

__device__ __forceinline__ void zero16(f32x16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// ------------------------------------------------------------------------------------------------ forward
template <bool CAUSAL>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs p) {
  __shared__ __attribute__((aligned(16))) bf16 Ks[KT * RS];
  __shared__ __attribute__((aligned(16))) bf16 Vr[KT * RT];   // V row-major, read transposed (tr16)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int qw0 = blockIdx.x * 128;
  const int q0 = qw0 + wave * 32;
  const int qi = q0 + (lane & 31);
  const bool wave_active = q0 < p.N;
  const bf16* qb = p.q + (long)b * p.sb + h * 64;
  const bf16* kb_ = p.k + (long)b * p.sb + h * 64;
  const bf16* vb = p.v + (long)b * p.sb + h * 64;
  const float sc2 = p.scale * LOG2E;

  bf16x8 qf[4];
  {
    const bf16* qr = qb + (long)min(qi, p.N - 1) * p.sn + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(qr + ks * 16);
  }
  f32x16 oacc[2];
  zero16(oacc[0]);
  zero16(oacc[1]);
  float m_i = -1e30f, l_i = 0.f;

  const int kend = CAUSAL ? min(p.N, qw0 + 128) : p.N;
  const int ntiles = (kend + KT - 1) / KT;
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();
    stage_tile2<RS, 0>(kb_, p.sn, t * KT, p.N, Ks, nullptr);
    stage_tile2<RT, 0>(vb, p.sn, t * KT, p.N, Vr, nullptr);
    __syncthreads();
    if (!wave_active) continue;
    const int nkb = min(2, (kend - t * KT + 31) / 32);
    for (int kb = 0; kb < nkb; ++kb) {
      const int key0 = t * KT + kb * 32;
      if (CAUSAL && key0 > q0 + 31) break;
      f32x16 s;
      zero16(s);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm(Ks, kb, ks, lane), qf[ks], s, 0, 0, 0);
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        float x = s[r] * sc2;
        if (key >= p.N || (CAUSAL && key > qi)) x = -INFINITY;
        s[r] = x;
        mx = fmaxf(mx, x);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_i, mx);
      const float alpha = fast_exp2(m_i - m_new);
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = fast_exp2(s[r] - m_new);
        rs += s[r];
      }
      rs += __shfl_xor(rs, 32, 64);
      l_i = l_i * alpha + rs;
      m_i = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        oacc[0][r] *= alpha;
        oacc[1][r] *= alpha;
      }
      const bf16x8 pf0 = pack8(s, 0), pf1 = pack8(s, 8);
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_trr(Vr, db, kb, 0, lane), pf0, oacc[db], 0, 0, 0);
        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_trr(Vr, db, kb, 1, lane), pf1, oacc[db], 0, 0, 0);
      }
    }
  }
  if (!wave_active || qi >= p.N) return;
  const float inv = 1.f / l_i;
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
  if (hi == 0 && p.lse) p.lse[((long)b * p.heads + h) * p.N + qi] = (m_i + log2f(l_i)) * LN2;
}

// ------------------------------------------------------------------------------------------------ delta = rowsum(dO * O)
__global__ __launch_bounds__(256) void attn_delta_kernel(const AttnArgs p) {
  const long total = (long)p.B * p.heads * p.N * 8;  // 8 lanes per (b,h,n) row
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
    const int part = (int)(idx & 7);
    const long row = idx >> 3;
    const int n = (int)(row % p.N);
    const int h = (int)((row / p.N) % p.heads);
    const long b = row / ((long)p.N * p.heads);
    const long off = b * p.sbo + (long)n * p.sno + h * 64 + part * 8;
    bf16x8 a = *(const bf16x8*)(p.o + off), g = *(const bf16x8*)(p.d_o + off);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += bf2f(a[e]) * bf2f(g[e]);
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (part == 0) p.delta[row] = s;
  }
}

// ------------------------------------------------------------------------------------------------ backward: dQ
template <bool CAUSAL>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnArgs p) {
  __shared__ __attribute__((aligned(16))) bf16 Ks[KT * RS];
  __shared__ __attribute__((aligned(16))) bf16 Vs[KT * RS];
  __shared__ __attribute__((aligned(16))) bf16 Kr[KT * RT];   // second row-major K image, read transposed (tr16)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int qw0 = blockIdx.x * 128;
  const int q0 = qw0 + wave * 32;
  const int qi = q0 + (lane & 31);
  const int qc = min(qi, p.N - 1);
  const bool wave_active = q0 < p.N;
  const bf16* kb_ = p.k + (long)b * p.sb + h * 64;
  const bf16* vb = p.v + (long)b * p.sb + h * 64;
  const float sc2 = p.scale * LOG2E;

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
  const float lse2 = p.lse[srow] * LOG2E;
  const float dlt = p.delta[srow];
  f32x16 dq[2];
  zero16(dq[0]);
  zero16(dq[1]);

  const int kend = CAUSAL ? min(p.N, qw0 + 128) : p.N;
  const int ntiles = (kend + KT - 1) / KT;
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();
    stage_tile2<RS, RT>(kb_, p.sn, t * KT, p.N, Ks, Kr);
    stage_tile2<RS, 0>(vb, p.sn, t * KT, p.N, Vs, nullptr);
    __syncthreads();
    if (!wave_active) continue;
    const int nkb = min(2, (kend - t * KT + 31) / 32);
    for (int kb = 0; kb < nkb; ++kb) {
      const int key0 = t * KT + kb * 32;
      if (CAUSAL && key0 > q0 + 31) break;
      f32x16 s, dp;
      zero16(s);
      zero16(dp);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm(Ks, kb, ks, lane), qf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm(Vs, kb, ks, lane), dof[ks], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        float pr = fast_exp2(s[r] * sc2 - lse2);
        if (key >= p.N || (CAUSAL && key > qi)) pr = 0.f;
        s[r] = pr * (dp[r] - dlt) * p.scale;
      }
      const bf16x8 d0 = pack8(s, 0), d1 = pack8(s, 8);
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_trr(Kr, db, kb, 0, lane), d0, dq[db], 0, 0, 0);
        dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_trr(Kr, db, kb, 1, lane), d1, dq[db], 0, 0, 0);
      }
    }
  }
  if (!wave_active || qi >= p.N) return;
  bf16* drow = p.dq + (long)b * p.sb + (long)qi * p.sn + h * 64;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = dq[db][4 * g + e];
      *(bf16x4*)(drow + db * 32 + 8 * g + 4 * hi) = __builtin_convertvector(v, bf16x4);
    }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
template <bool CAUSAL>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const AttnArgs p) {
  __shared__ __attribute__((aligned(16))) bf16 Qs[KT * RS];
  __shared__ __attribute__((aligned(16))) bf16 Gs[KT * RS];   // dO row-major
  __shared__ __attribute__((aligned(16))) bf16 Qr[KT * RT];   // row-major images read transposed (tr16)
  __shared__ __attribute__((aligned(16))) bf16 Gr[KT * RT];
  __shared__ __attribute__((aligned(16))) float lse_s[KT];
  __shared__ __attribute__((aligned(16))) float dlt_s[KT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int kw0 = blockIdx.x * 128;
  const int k0 = kw0 + wave * 32;
  const int ki = k0 + (lane & 31);
  const int kc = min(ki, p.N - 1);
  const bool wave_active = k0 < p.N;
  const bf16* qb = p.q + (long)b * p.sb + h * 64;
  const bf16* gb = p.d_o + (long)b * p.sbo + h * 64;
  const float sc2 = p.scale * LOG2E;

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
  f32x16 dk[2], dv[2];
  zero16(dk[0]);
  zero16(dk[1]);
  zero16(dv[0]);
  zero16(dv[1]);

  const int ntiles = (p.N + KT - 1) / KT;
  const int t0 = CAUSAL ? (kw0 / KT) : 0;
  const long srow0 = ((long)b * p.heads + h) * p.N;
  for (int t = t0; t < ntiles; ++t) {
    __syncthreads();
    stage_tile2<RS, RT>(qb, p.sn, t * KT, p.N, Qs, Qr);
    stage_tile2<RS, RT>(gb, p.sno, t * KT, p.N, Gs, Gr);
    if (threadIdx.x < KT) {
      const int qn = min(t * KT + (int)threadIdx.x, p.N - 1);
      lse_s[threadIdx.x] = p.lse[srow0 + qn] * LOG2E;
      dlt_s[threadIdx.x] = p.delta[srow0 + qn];
    }
    __syncthreads();
    if (!wave_active) continue;
    const int nqb = min(2, (p.N - t * KT + 31) / 32);
    for (int qblk = 0; qblk < nqb; ++qblk) {
      const int qs0 = t * KT + qblk * 32;
      if (CAUSAL && qs0 + 31 < k0) continue;
      f32x16 s, dp;
      zero16(s);
      zero16(dp);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm(Qs, qblk, ks, lane), kf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm(Gs, qblk, ks, lane), vf[ks], dp, 0, 0, 0);
      }
      f32x16 pr;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 l4 = *(const f32x4*)(lse_s + qblk * 32 + 8 * g + 4 * hi);
        const f32x4 d4 = *(const f32x4*)(dlt_s + qblk * 32 + 8 * g + 4 * hi);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const int qn = qs0 + 8 * g + 4 * hi + e;
          float pv = fast_exp2(s[r] * sc2 - l4[e]);
          if (qn >= p.N || ki >= p.N || (CAUSAL && ki > qn)) pv = 0.f;
          pr[r] = pv;
          s[r] = pv * (dp[r] - d4[e]) * p.scale;
        }
      }
      const bf16x8 p0 = pack8(pr, 0), p1 = pack8(pr, 8), d0 = pack8(s, 0), d1 = pack8(s, 8);
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_trr(Gr, db, qblk, 0, lane), p0, dv[db], 0, 0, 0);
        dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_trr(Gr, db, qblk, 1, lane), p1, dv[db], 0, 0, 0);
        dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_trr(Qr, db, qblk, 0, lane), d0, dk[db], 0, 0, 0);
        dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_trr(Qr, db, qblk, 1, lane), d1, dk[db], 0, 0, 0);
      }
    }
  }
  if (!wave_active || ki >= p.N) return;
  bf16* krow = p.dk + (long)b * p.sb + (long)ki * p.sn + h * 64;
  bf16* vrow = p.dv + (long)b * p.sb + (long)ki * p.sn + h * 64;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 a, c;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[e] = dk[db][4 * g + e];
        c[e] = dv[db][4 * g + e];
      }
      *(bf16x4*)(krow + db * 32 + 8 * g + 4 * hi) = __builtin_convertvector(a, bf16x4);
      *(bf16x4*)(vrow + db * 32 + 8 * g + 4 * hi) = __builtin_convertvector(c, bf16x4);
    }
}

// short non-causal sequences: K/V (Q/dO) resident in LDS, one workgroup per (image, head) -- attention_resident.hip
int attn_resident_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int N, int heads, long sb,
                      long sn, long sbo, long sno, float scale, hipStream_t s);
int attn_resident_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                      float* delta, void* dq, void* dk, void* dv, const void* rope_sin, const void* rope_cos, int rope_prefix,
                      int B, int N, int heads, long sb, long sn, long sbo, long sno, float scale, hipStream_t s);
static bool use_resident(int N, int causal) {
  return !causal && N <= 320;
}

}  // namespace vtp
using namespace vtp;

static int check_attn(const char* who, int B, int N, int heads, long sb, long sn, long sbo, long sno) {
  VTP_REQUIRE(B > 0 && N > 0 && heads > 0, "%s: bad shape B=%d N=%d heads=%d", who, B, N, heads);
  VTP_REQUIRE(sn % 8 == 0 && sb % 8 == 0 && sno % 8 == 0 && sbo % 8 == 0, "%s: strides must be multiples of 8 elements", who);
  VTP_REQUIRE(B <= 65535 && heads <= 65535, "%s: B and heads must be <= 65535", who);
  return VTP_OK;
}

extern "C" int vtp_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int N, int heads,
                            long sb_qkv, long sn_qkv, long sb_o, long sn_o, float scale, int causal, void* stream) {
  VTP_REQUIRE(q && k && v && o, "vtp_attn_fwd: null pointer");
  if (int e = check_attn("vtp_attn_fwd", B, N, heads, sb_qkv, sn_qkv, sb_o, sn_o)) return e;
  AttnArgs a = {};
  a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.out = (bf16*)o; a.lse = lse;
  a.B = B; a.N = N; a.heads = heads; a.sb = sb_qkv; a.sn = sn_qkv; a.sbo = sb_o; a.sno = sn_o; a.scale = scale;
  if (use_resident(N, causal))
    return attn_resident_fwd(q, k, v, o, lse, B, N, heads, sb_qkv, sn_qkv, sb_o, sn_o, scale, (hipStream_t)stream);
  dim3 grid(cdiv(N, 128), heads, B);
  if (causal) hipLaunchKernelGGL(attn_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(attn_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("attn_fwd");
}

extern "C" int vtp_rope_qk(void* qkv, const void* sin, const void* cos, int B, int N, int heads, int prefix, int inverse,
                           void* stream);

extern "C" int vtp_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                            float* delta, void* dq, void* dk, void* dv, const void* rope_sin, const void* rope_cos,
                            int rope_prefix, int B, int N, int heads, long sb_qkv, long sn_qkv, long sb_o, long sn_o,
                            float scale, int causal, void* stream) {
  VTP_REQUIRE(q && k && v && o && d_o && lse && delta && dq && dk && dv, "vtp_attn_bwd: null pointer");
  VTP_REQUIRE((rope_sin == nullptr) == (rope_cos == nullptr), "vtp_attn_bwd: rope_sin and rope_cos go together");
  VTP_REQUIRE(!rope_sin || (rope_prefix >= 0 && rope_prefix <= N), "vtp_attn_bwd: rope_prefix must be in [0, N]");
  if (int e = check_attn("vtp_attn_bwd", B, N, heads, sb_qkv, sn_qkv, sb_o, sn_o)) return e;
  hipStream_t s = (hipStream_t)stream;
  if (use_resident(N, causal))  // delta and the inverse RoPE of dq / dk are fused into the resident kernels
    return attn_resident_bwd(q, k, v, o, d_o, lse, delta, dq, dk, dv, rope_sin, rope_cos, rope_prefix, B, N, heads, sb_qkv,
                             sn_qkv, sb_o, sn_o, scale, s);
  AttnArgs a = {};
  a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.o = (const bf16*)o; a.d_o = (const bf16*)d_o;
  a.lse = (float*)lse; a.delta = delta; a.dq = (bf16*)dq; a.dk = (bf16*)dk; a.dv = (bf16*)dv;
  a.B = B; a.N = N; a.heads = heads; a.sb = sb_qkv; a.sn = sn_qkv; a.sbo = sb_o; a.sno = sn_o; a.scale = scale;
  const long rows8 = (long)B * heads * N * 8;
  int dblocks = (int)((rows8 + 255) / 256);
  if (dblocks > 4096) dblocks = 4096;
  hipLaunchKernelGGL(attn_delta_kernel, dim3(dblocks), dim3(256), 0, s, a);
  dim3 grid(cdiv(N, 128), heads, B);
  if (causal) {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<true>, grid, dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<false>, grid, dim3(256), 0, s, a);
  }
  if (int e = check_launch("attn_bwd")) return e;
  if (rope_sin) {  // tiled path: the standalone inverse-RoPE kernel on the packed [*, 3*heads*64] gradient buffer
    VTP_REQUIRE((const bf16*)dk == (const bf16*)dq + heads * 64 && sn_qkv == 3L * heads * 64 && sb_qkv == (long)N * sn_qkv,
                "vtp_attn_bwd: fused inverse RoPE on the tiled path needs the packed qkv gradient layout");
    return vtp_rope_qk(dq, rope_sin, rope_cos, B, N, heads, rope_prefix, 1, stream);
  }
  return VTP_OK;
}
