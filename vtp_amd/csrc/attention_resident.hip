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
// Forward: attn_fwd_res2_kernel (default: two query tiles per wave, single-pass online softmax, K / V streamed behind the key
// loop, odd last tile split over the waves); attn_fwd_res_kernel is the round-1 two-pass variant (shapes the
// v2 work split does not cover).  Backward: one fused kernel at 33 .. 66 and 225 .. 258 tokens (attn_bwd_fused_kernel: the shapes of the
// step), dQ and dK/dV kernels otherwise; delta and the inverse RoPE are fused in both.
#include "common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include "vtp_hip.h"

namespace vtp {

constexpr float LOG2E_R = 1.4426950408889634f;
constexpr float LN2_R = 0.6931471805599453f;
constexpr int RES_MAXN = 320;
constexpr int RES_SPLIT_COLS = 4;  // valid queries of an odd last tile up to which it is split over the waves (fwd v2)

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
  int stagger;            // experiment: start every second workgroup of the first round this many 10-ns ticks late
  long long* timing;      // diagnostics (vtp_attn_debug): [workgroup][16 waves][4] s_memrealtime stamps of the backward kernels, or null
};

// phase stamps of the backward kernels: 0 start, 1 operands staged, 2 loop done, 3 gradients stored
#define BWD_STAMP(i)                                                                                                     \
  do {                                                                                                                   \
    if (__builtin_expect(p.timing != nullptr, 0) && lane == 0)                                                           \
      p.timing[((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + wave) * 4 + (i)] = wall_clock64(); \
  } while (0)

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

// The same two fragment reads with the address split into a per-lane part (loop-invariant VGPR) and a wave-uniform part (SGPR, the
// row block of the current step), added by an instruction the compiler cannot look through: left to itself hipcc strength-reduces
// every fragment address into its own VGPR incremented per step -- 18 address registers in the dK/dV loop, which pushed it over
// the 168-VGPR budget of three waves per SIMD and put a scratch reload (with a full vmcnt(0) drain) into every step.
//   row fragment:  lane part row_part(lane) ^ (ks << 5)           (the chunk swizzle is an XOR, ks selects bits 5..6)
//   tr fragment:   lane part tr_part(lane, half) ^ (dblk << 6)    (half = rows +0 / +8), + ks * 2048 as immediate
__device__ __forceinline__ int row_part(int lane) {
  const int r = lane & 31;
  return r * 128 + (((lane >> 5) ^ swz_key(r)) << 4);
}
__device__ __forceinline__ int tr_part(int lane, int half) {
  const int i = lane & 15, g = (lane >> 4) & 1, hi = lane >> 5;
  const int r = hi * 4 + (i >> 2) + 8 * half;
  const int col = g * 16 + (i & 3) * 4;
  return r * 128 + (((col >> 3) ^ swz_key(r)) << 4) + ((col & 7) << 1);
}
__device__ __forceinline__ int lds_addr(int lane_part, int uniform_part) {
  int a;
  asm("v_add_u32 %0, %1, %2" : "=v"(a) : "s"(uniform_part), "v"(lane_part));
  return a;
}
__device__ __forceinline__ bf16x8 frag_row_at(int addr) { return *(const __attribute__((address_space(3))) bf16x8*)(size_t)addr; }
__device__ __forceinline__ bf16x8 frag_tr_at(int addr0, int addr1, int ks) {
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(size_t)(addr0 + ks * 2048));
  bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(size_t)(addr1 + ks * 2048));
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
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
    case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
    case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
    case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
    case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break;
    case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 25: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break;
    case 26: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
    case 27: asm volatile("s_waitcnt vmcnt(27)" ::: "memory"); break;
    case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
    case 29: asm volatile("s_waitcnt vmcnt(29)" ::: "memory"); break;
    case 30: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
    case 31: asm volatile("s_waitcnt vmcnt(31)" ::: "memory"); break;
    case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
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
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6, hi = lane >> 5;  // wave: an SGPR (uniform branches, scalar wait counts)
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

// ------------------------------------------------------------------------------------------------ forward, v2
// One wave owns T = 2 query tiles (64 queries): every K / V fragment read from LDS feeds both tiles, and the two tiles are
// independent instruction streams.  Single pass with a running maximum (online softmax): the accumulator is rescaled only when a
// maximum in the wave moved.  The softmax row of a query is split over lane (keys 4hi + ..) and lane ^ 32: v_permlane32_swap
// exchanges the halves.
//
// Work split of one (image, head): waves = floor(tiles / 2), each with two full tiles against ALL key blocks.  An odd last tile
// (N = 257 = 8 x 32 + 1: one valid query) would cost a fifth wave a whole tile of work on one SIMD -- instead every wave takes
// that tile against a quarter of the key blocks, and the partial (max, sum, O) are merged through a few hundred bytes of LDS
// (split-K over the keys, only the valid columns travel).  4 waves x <= 256 VGPRs and 74 KB of LDS: two workgroups share a CU,
// one's staging / stores under the other's key loop.
__device__ __forceinline__ float both_halves_sum(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// v_max3_f32 without the NaN-quieting v_max x, x the compiler puts in front of fmaxf operands (scores are finite or -inf)
__device__ __forceinline__ float max3f(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float max2f(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// online-softmax key loop over blocks kb0, kb0 + step, ... < nkb for T query tiles; m in the log2 domain (scores * sc2).
// The VALU work per score is what bounds this kernel (16 v_exp_f32 per lane and tile at quarter rate = the time of the tile's
// 8 MFMAs), so everything around the exponentials is kept to packed / 3-operand instructions: 8 v_max3, 8 v_pk_fma, 8 v_pk_add,
// 8 v_cvt_pk_bf16 per tile and key block.
// stream_ops > 0: K / V arrive by LDS-DMA behind the loop, key block by key block, `stream_ops` vector-memory operations of this
// wave per block (issued in block order; blocks 0 and 1 have landed before the loop): block kb is ready once at most
// stream_ops * (nkb - 1 - kb) operations of every wave are outstanding.
constexpr float LAZY_LOG2 = 6.f;

template <int T>
__device__ __forceinline__ void attn_key_loop(const char* Ks, const char* Vs, const bf16x8 (&qf)[T][4], int N, int nkb, int kb0, int step,
                                              float sc2, int lane, f32x16 (&oacc)[T][2], float (&m)[T], float (&l)[T],
                                              int stream_ops = 0) {
  const int hi = lane >> 5, last = nkb - 1;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const f32x2 sc2v = {sc2, sc2};
#pragma unroll
  for (int t = 0; t < T; ++t) {
    oacc[t][0] = zero;
    oacc[t][1] = zero;
    m[t] = -INFINITY;
    l[t] = 0.f;
  }
  for (int kb = kb0; kb < nkb; kb += step) {
    // the stream arrives faster than the loop consumes it: two sync points (after blocks 0-1: blocks 2-4, then the rest) instead
    // of a counted wait + barrier in front of every block
    if (stream_ops > 0 && (kb == 2 || kb == 5)) {
      wait_vmcnt_upto(__builtin_amdgcn_readfirstlane(kb == 2 ? stream_ops * max(nkb - 5, 0) : 0));
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    f32x16 s[T];
    {
      bf16x8 kf[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) kf[ks] = frag_row(Ks, kb * 32, ks, lane);
#pragma unroll
      for (int t = 0; t < T; ++t) {
        s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[t][0], zero, 0, 0, 0);
#pragma unroll
        for (int ks = 1; ks < 4; ++ks) s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[t][ks], s[t], 0, 0, 0);
      }
    }
    // V^T fragments of this block: issued now, landed by the time the probabilities are packed
    bf16x8 vf[2][2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) vf[db][ks] = frag_tr(Vs, db, kb * 32, ks, lane);
    if (kb == last && (N & 31)) {
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= N) s[t][r] = -INFINITY;
    }
    float mn[T];
    bool moved = false;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      float bm = max3f(s[t][0], s[t][1], s[t][2]);
#pragma unroll
      for (int r = 3; r < 15; r += 2) bm = max3f(bm, s[t][r], s[t][r + 1]);
      bm = max2f(bm, s[t][15]);
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(bm), __float_as_uint(bm), false, false);
      bm = max2f(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      // lazy running maximum: it follows the block maximum only when that exceeds it by more than 2^LAZY -- until then the
      // probabilities are merely scaled by up to 2^LAZY (exact in fp32 / bf16: a power-of-two factor that cancels in O / l)
      const float cand = bm * sc2;
      mn[t] = cand > m[t] + LAZY_LOG2 ? cand : m[t];
      moved |= mn[t] > m[t];
    }
    if (__any(moved)) {  // wave-uniform: rare after the first key blocks
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const float alpha = fast_exp2(m[t] - mn[t]);
        l[t] *= alpha;
        oacc[t][0] *= alpha;
        oacc[t][1] *= alpha;
        m[t] = mn[t];
      }
    }
    bf16x8 pf[T][2];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const f32x2 mv = {-m[t], -m[t]};
      f32x2 acc = {0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        f32x2 v = {s[t][2 * i], s[t][2 * i + 1]};
        v = __builtin_elementwise_fma(v, sc2v, mv);
        v[0] = fast_exp2(v[0]);
        v[1] = fast_exp2(v[1]);
        acc += v;
        s[t][2 * i] = v[0];
        s[t][2 * i + 1] = v[1];
      }
      l[t] += acc[0] + acc[1];
      pf[t][0] = pack8r(s[t], 0);
      pf[t][1] = pack8r(s[t], 8);
    }
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int t = 0; t < T; ++t) oacc[t][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[db][ks], pf[t][ks], oacc[t][db], 0, 0, 0);
      }
  }
}

// one query row of O^T (this lane: d = 32 db + 8 g + 4 hi + e) -> bf16 out row.  The two half-lanes of a query trade 8-byte
// pieces first so that each writes 16 contiguous bytes: 4 stores of 16 B per lane instead of 8 of 8 B.
__device__ __forceinline__ void store_o_row(bf16* orow, const f32x16 (&o)[2], float inv, int hi, bool valid) {
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      uint32_t x[2], y[2];  // x: piece g = 2j, y: piece g = 2j + 1 of this lane
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const f32x2 vx = {o[db][8 * j + 2 * w] * inv, o[db][8 * j + 2 * w + 1] * inv};
        const f32x2 vy = {o[db][8 * j + 4 + 2 * w] * inv, o[db][8 * j + 4 + 2 * w + 1] * inv};
        const bf16x2 bx = __builtin_convertvector(vx, bf16x2), by = __builtin_convertvector(vy, bf16x2);
        x[w] = __builtin_bit_cast(uint32_t, bx);
        y[w] = __builtin_bit_cast(uint32_t, by);
      }
      // permlane32_swap(x, y): lanes < 32 end with (own x, partner's x), lanes >= 32 with (partner's y, own y)
      uint32_t out[4];
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const auto r = __builtin_amdgcn_permlane32_swap(x[w], y[w], false, false);
        out[w] = r[0];
        out[2 + w] = r[1];
      }
      if (valid) {
        typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
        *(u32x4*)(orow + db * 32 + 16 * j + 8 * hi) = u32x4{out[0], out[1], out[2], out[3]};
      }
    }
}

// LDS-DMA of one 8-row piece from inline asm: the compiler does not see it, so it puts no s_waitcnt vmcnt(0) in front of LDS reads
// it cannot prove disjoint (that would serialise the stream and the key loop); completion is counted by hand.  The dynamic segment
// is the kernel's only LDS, so LDS addresses are offsets from smem.
__device__ __forceinline__ void stage_piece_asm(const bf16* base, long sn, int N, int piece, unsigned img_off, int lane) {
  const int row = piece * 8 + (lane >> 3);
  const int c = (lane & 7) ^ swz_key(row);
  const unsigned voff = (unsigned)(min(row, N - 1) * (int)sn + c * 8) * 2u;
  const unsigned dst = __builtin_amdgcn_readfirstlane(img_off + piece * 1024);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(base), "s"(dst)
               : "memory");
}
// this wave's pieces of key block kb of both images (K at LDS offset 0, V at img): pieces 4 kb + {wave, wave + nwaves, ..} < 4 kb + 4
__device__ __forceinline__ void stage_block_asm(const bf16* kbase, const bf16* vbase, long sn, int N, int kb, unsigned img, int wave,
                                                int nwaves, int lane) {
  for (int j = wave; j < 4; j += nwaves) {
    stage_piece_asm(kbase, sn, N, kb * 4 + j, 0, lane);
    stage_piece_asm(vbase, sn, N, kb * 4 + j, img, lane);
  }
}

// phase stamps of one workgroup (VTP_ATTN_TIMING=1: the launcher passes a buffer in p.delta and prints the cycle counts)
#define ATTN_TSTAMP(i)                                                                            \
  do {                                                                                            \
    if (p.delta && blockIdx.x == 3 && blockIdx.y == 5 && lane == 0)                               \
      ((long long*)p.delta)[wave * 8 + (i)] = __builtin_readcyclecounter();                       \
  } while (0)

__global__ __launch_bounds__(256, 2) void attn_fwd_res2_kernel(const AttnResArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + p.npad * 128;
  float* part = (float*)(smem + 2 * p.npad * 128);  // [wave][column][half][34]: partial (m, l, O) of the split last tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6, hi = lane >> 5;
  const int h = blockIdx.x, b = blockIdx.y;
  const bf16* qb = p.q + (long)b * p.sb + h * 64;
  const int ntiles = p.npad >> 5, nkb = ntiles, q0 = wave * 64;
  ATTN_TSTAMP(0);
  const bool split_last = (ntiles & 1) && ntiles > 1;  // the launcher guarantees N - 32 (ntiles - 1) <= RES_SPLIT_COLS then
  const int qL = (ntiles - 1) * 32;
  bf16x8 qf[2][4], qfl[1][4];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const bf16* qr = qb + (long)min(q0 + 32 * t + (lane & 31), p.N - 1) * p.sn + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[t][ks] = *(const bf16x8*)(qr + ks * 16);
  }
  if (split_last) {
    const bf16* qr = qb + (long)min(qL + (lane & 31), p.N - 1) * p.sn + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qfl[0][ks] = *(const bf16x8*)(qr + ks * 16);
  }
  // K / V stream in key-block order: the query fragments and blocks 0, 1 are waited for here (a wait the compiler models, so it
  // adds none of its own later), the rest is issued behind them and lands while the key loop runs
  const bf16* kbase = p.k + (long)b * p.sb + h * 64;
  const bf16* vbase = p.v + (long)b * p.sb + h * 64;
  const unsigned img = p.npad * 128;
  stage_block_asm(kbase, vbase, p.sn, p.N, 0, img, wave, nwaves, lane);
  if (nkb > 1) stage_block_asm(kbase, vbase, p.sn, p.N, 1, img, wave, nwaves, lane);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  ATTN_TSTAMP(1);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  ATTN_TSTAMP(2);
  for (int kb = 2; kb < nkb; ++kb) stage_block_asm(kbase, vbase, p.sn, p.N, kb, img, wave, nwaves, lane);
  const int stream_ops = 2 * ((4 - wave + nwaves - 1) / nwaves);  // this wave's DMA operations per key block
  const float sc2 = p.scale * LOG2E_R;
  const long srow0 = ((long)b * p.heads + h) * p.N;
  {
    f32x16 oacc[2][2];
    float m[2], l[2];
    const bool two = q0 + 32 < p.npad && !(split_last && q0 + 32 >= qL);
    if (two) attn_key_loop<2>(Ks, Vs, qf, p.N, nkb, 0, 1, sc2, lane, oacc, m, l, stream_ops);
    else {
      f32x16 (&o1)[1][2] = *(f32x16(*)[1][2])&oacc;
      attn_key_loop<1>(Ks, Vs, *(const bf16x8(*)[1][4])&qf, p.N, nkb, 0, 1, sc2, lane, o1, *(float(*)[1])&m, *(float(*)[1])&l,
                       stream_ops);
    }
    ATTN_TSTAMP(3);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t == 1 && !two) break;
      const int qi = q0 + 32 * t + (lane & 31);
      const float lt = both_halves_sum(l[t]);
      const bool valid = qi < p.N;
      store_o_row(p.out + (long)b * p.sbo + (long)min(qi, p.N - 1) * p.sno + h * 64, oacc[t], 1.f / lt, hi, valid);
      if (valid && hi == 0 && p.lse) p.lse[srow0 + qi] = (m[t] + log2f(lt)) * LN2_R;
    }
  }
  ATTN_TSTAMP(4);
  if (!split_last) return;
  // ---- the odd last tile: this wave's share of the key blocks, then the merge
  const int cols = p.N - qL;  // valid queries of the tile
  {
    f32x16 oacc[1][2];
    float m[1], l[1];
    attn_key_loop<1>(Ks, Vs, qfl, p.N, nkb, wave, nwaves, sc2, lane, oacc, m, l);
    const float lt = both_halves_sum(l[0]);
    if ((lane & 31) < cols) {
      float* dst = part + ((wave * cols + (lane & 31)) * 2 + hi) * 34;
      dst[0] = m[0];
      dst[1] = lt;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        dst[2 + r] = oacc[0][0][r];
        dst[18 + r] = oacc[0][1][r];
      }
    }
  }
  ATTN_TSTAMP(5);
  __syncthreads();
  ATTN_TSTAMP(6);
  if (wave != 0) return;
  {
    const int c = min(lane & 31, cols - 1);
    float M = -INFINITY;
    for (int w = 0; w < nwaves; ++w) M = fmaxf(M, part[((w * cols + c) * 2 + hi) * 34]);
    f32x16 o[2];
    zero16r(o[0]);
    zero16r(o[1]);
    float L = 0.f;
    for (int w = 0; w < nwaves; ++w) {
      const float* src = part + ((w * cols + c) * 2 + hi) * 34;
      const float sc = fast_exp2(src[0] - M);
      L += src[1] * sc;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o[0][r] += src[2 + r] * sc;
        o[1][r] += src[18 + r] * sc;
      }
    }
    const int qi = qL + (lane & 31);
    const bool valid = (lane & 31) < cols;
    store_o_row(p.out + (long)b * p.sbo + (long)min(qi, p.N - 1) * p.sno + h * 64, o, 1.f / L, hi, valid);
    if (valid && hi == 0 && p.lse) p.lse[srow0 + qi] = (M + log2f(L)) * LN2_R;
  }
}

// ------------------------------------------------------------------------------------------------ backward: dQ
__global__ __launch_bounds__(640) void attn_bwd_dq_res_kernel(const AttnResArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + p.npad * 128;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6, hi = lane >> 5;
  const int h = blockIdx.x, b = blockIdx.y;
  BWD_STAMP(0);
  if (__builtin_expect(p.stagger > 0, 0)) {
    const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (wg < 256 && ((wg >> 3) & 1)) {
      const unsigned long long t0 = wall_clock64();
      while (wall_clock64() - t0 < (unsigned long long)p.stagger) __builtin_amdgcn_s_sleep(16);
    }
  }
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
  BWD_STAMP(1);
  if (q0 >= p.npad) return;
  const float sc2 = p.scale * LOG2E_R;
  const f32x2 sc2v = {sc2, sc2}, nlse = {-lse2, -lse2}, scv = {p.scale, p.scale}, ndlt = {-dlt * p.scale, -dlt * p.scale};
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
    // dS = P (.) (dP - delta) * scale with P = exp2(S sc2 - lse2): packed fp32 (two scores per v_pk_fma / v_pk_mul) around the
    // two quarter-rate exponentials -- the VALU work, not the 12 MFMAs, is what bounds this loop
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      f32x2 e = {s[2 * i], s[2 * i + 1]}, g = {dp[2 * i], dp[2 * i + 1]};
      e = __builtin_elementwise_fma(e, sc2v, nlse);
      g = __builtin_elementwise_fma(g, scv, ndlt);
      e[0] = fast_exp2(e[0]);
      e[1] = fast_exp2(e[1]);
      e *= g;
      s[2 * i] = e[0];
      s[2 * i + 1] = e[1];
    }
    if (kb == last && (p.N & 31)) {
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
  BWD_STAMP(2);
  if (qi < p.N) {
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
  BWD_STAMP(3);
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
  BWD_STAMP(0);
  if (__builtin_expect(p.stagger > 0, 0)) {
    const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (wg < 256 && ((wg >> 3) & 1)) {
      const unsigned long long t0 = wall_clock64();
      while (wall_clock64() - t0 < (unsigned long long)p.stagger) __builtin_amdgcn_s_sleep(16);
    }
  }
  stage_resident(p.q + (long)b * p.sb + h * 64, p.sn, p.N, p.npad, Qs, wave, nwaves, lane);
  stage_resident(p.d_o + (long)b * p.sbo + h * 64, p.sno, p.N, p.npad, Gs, wave, nwaves, lane);
  const long srow0 = ((long)b * p.heads + h) * p.N;
  for (int i = threadIdx.x; i < p.npad; i += blockDim.x) {
    const int qn = min(i, p.N - 1);
    lse_s[i] = -p.lse[srow0 + qn] * LOG2E_R;  // pre-negated / pre-scaled: they enter the key loop as fma addends
    dlt_s[i] = -p.delta[srow0 + qn] * p.scale;
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
  BWD_STAMP(1);
  if (k0 >= p.npad) return;
  const float sc2 = p.scale * LOG2E_R;
  const f32x2 sc2v = {sc2, sc2}, scv = {p.scale, p.scale};
  const int nqb = p.npad >> 5, last = nqb - 1;
  f32x16 dk[2], dv[2];
  zero16r(dk[0]);
  zero16r(dk[1]);
  zero16r(dv[0]);
  zero16r(dv[1]);
  // per-lane parts of the fragment addresses (8 VGPRs for the whole loop; LDS addresses are offsets: the dynamic segment is the
  // kernel's only LDS)
  int rpart[4], tpart[2][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) rpart[ks] = row_part(lane) ^ (ks << 5);
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) tpart[db][hf] = tr_part(lane, hf) ^ (db << 6);
  const int g_off = p.npad * 128;
  for (int qblk = 0; qblk < nqb; ++qblk) {
    f32x16 s, dp;
    zero16r(s);
    zero16r(dp);
    const int q_off = __builtin_amdgcn_readfirstlane(qblk * 4096);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row_at(lds_addr(rpart[ks], q_off)), kf[ks], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row_at(lds_addr(rpart[ks], q_off + g_off)), vf[ks], dp, 0, 0, 0);
    }
    f32x16 pr;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 l4 = *(const f32x4*)(lse_s + qblk * 32 + 8 * g + 4 * hi);  // -lse * log2(e) of the four query rows
      const f32x4 d4 = *(const f32x4*)(dlt_s + qblk * 32 + 8 * g + 4 * hi);  // -delta * scale
#pragma unroll
      for (int e2 = 0; e2 < 2; ++e2) {
        const int r = 4 * g + 2 * e2;
        f32x2 e = {s[r], s[r + 1]}, gq = {dp[r], dp[r + 1]};
        e = __builtin_elementwise_fma(e, sc2v, f32x2{l4[2 * e2], l4[2 * e2 + 1]});
        gq = __builtin_elementwise_fma(gq, scv, f32x2{d4[2 * e2], d4[2 * e2 + 1]});
        e[0] = fast_exp2(e[0]);
        e[1] = fast_exp2(e[1]);
        if (qblk == last && (p.N & 31)) {
          if (qblk * 32 + 8 * g + 4 * hi + 2 * e2 >= p.N) e[0] = 0.f;
          if (qblk * 32 + 8 * g + 4 * hi + 2 * e2 + 1 >= p.N) e[1] = 0.f;
        }
        pr[r] = e[0];
        pr[r + 1] = e[1];
        e *= gq;
        s[r] = e[0];
        s[r + 1] = e[1];
      }
    }
    const bf16x8 p0 = pack8r(pr, 0), p1 = pack8r(pr, 8), d0 = pack8r(s, 0), d1 = pack8r(s, 8);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int tq0 = lds_addr(tpart[db][0], q_off), tq1 = lds_addr(tpart[db][1], q_off);
      const int tg0 = lds_addr(tpart[db][0], q_off + g_off), tg1 = lds_addr(tpart[db][1], q_off + g_off);
      dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_at(tg0, tg1, 0), p0, dv[db], 0, 0, 0);
      dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_at(tg0, tg1, 1), p1, dv[db], 0, 0, 0);
      dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_at(tq0, tq1, 0), d0, dk[db], 0, 0, 0);
      dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_at(tq0, tq1, 1), d1, dk[db], 0, 0, 0);
    }
  }
  BWD_STAMP(2);
  if (ki < p.N) {
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
  BWD_STAMP(3);
}

// ------------------------------------------------------------------------------------------------ backward, fused
// N = 225 .. 258 (the 257-token passes of the trunk and the 256-token decoder: 8 row blocks, or 8 + a 9th with one or two rows).
// What the two per-head kernels above lose at these sizes (tools/attn_bwd_timeline.py, profiles/r03_attn_bwd_timeline.log):
//   * 9 equal waves on 4 SIMDs: the SIMD with three of them sets the workgroup's time (loop 9.0 us vs 3.9 us for its fastest wave);
//   * more than half of a workgroup's 18 .. 23 us is not the loop: every lane fetches its own q / dO / O (or k / v) row as 16-B
//     pieces of 32 different 4.6-KB-strided rows per instruction (6 .. 7 us until the operands have arrived, alone on the chip
//     still 3 .. 4 us), and each of q, k, v, dO crosses HBM twice (once per kernel, once as LDS image, once as rows).
// Here ONE workgroup of 8 waves (two per SIMD, up to 256 VGPRs) per head stages the four images Q, K, V, dO (4 x 36 KB) by LDS-DMA
// -- whole 128-B lines, every tensor read once -- and runs both products from them: phase 1 the lane is a query (wave w = query
// block w: dQ), phase 2 the lane is a key (wave w = key block w: dK, dV); the row operands of either phase are fragments of the
// images, delta and lse pass from phase 1 to phase 2 through LDS.  The rows of the odd 9th block are spread: wave w computes the
// 9th block's contribution against block w (wave 0 also against block 8) after its own loop, the partial sums travel through LDS
// slots [block][row][64] and are added in block order (deterministic) by the threads that store them.
constexpr int PB_WAVES = 8;
constexpr int PB_XROWS = 2;  // rows of the odd 9th block (N - 256 <= 2)

__device__ __forceinline__ int stage_image_asm(const bf16* base, long sn, int N, int npad, unsigned img_off, int wave, int nwaves,
                                               int lane) {
  int n = 0;
  for (int piece = wave; piece < (npad >> 3); piece += nwaves, ++n) stage_piece_asm(base, sn, N, piece, img_off, lane);
  return n;
}

// one (query block, key block) step of the dQ computation, the lane being a query: s / dp from the K / V row fragments at LDS
// offset `koff` (V image `img` bytes behind), dS, and dq += dS K through the transposed fragments
__device__ __forceinline__ void dq_step(const bf16x8 (&qf)[4], const bf16x8 (&dof)[4], f32x16 (&dq)[2], const int (&rpart)[4],
                                        const int (&tpart)[2][2], int koff, int img, f32x2 sc2v, f32x2 nlse, f32x2 scv, f32x2 ndlt,
                                        int key0, int N, bool mask, int hi) {
  f32x16 s, dp;
  zero16r(s);
  zero16r(dp);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row_at(lds_addr(rpart[ks], koff)), qf[ks], s, 0, 0, 0);
    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row_at(lds_addr(rpart[ks], koff + img)), dof[ks], dp, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    f32x2 e = {s[2 * i], s[2 * i + 1]}, g = {dp[2 * i], dp[2 * i + 1]};
    e = __builtin_elementwise_fma(e, sc2v, nlse);
    g = __builtin_elementwise_fma(g, scv, ndlt);
    e[0] = fast_exp2(e[0]);
    e[1] = fast_exp2(e[1]);
    e *= g;
    s[2 * i] = e[0];
    s[2 * i + 1] = e[1];
  }
  if (mask) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (key0 + (r & 3) + 8 * (r >> 2) + 4 * hi >= N) s[r] = 0.f;
  }
  const bf16x8 d0 = pack8r(s, 0), d1 = pack8r(s, 8);
#pragma unroll
  for (int db = 0; db < 2; ++db) {
    const int t0 = lds_addr(tpart[db][0], koff), t1 = lds_addr(tpart[db][1], koff);
    dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_at(t0, t1, 0), d0, dq[db], 0, 0, 0);
    dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_at(t0, t1, 1), d1, dq[db], 0, 0, 0);
  }
}

// delta = rowsum(dO * O) of this lane's query (summed over the two half-waves)
__device__ __forceinline__ float row_delta(const bf16x8 (&of)[4], const bf16x8 (&dof)[4]) {
  float dlt = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int e = 0; e < 8; ++e) dlt += bf2f(of[ks][e]) * bf2f(dof[ks][e]);
  return dlt + __shfl_xor(dlt, 32, 64);
}

#define PB_STAMP(i)                                                                                                   \
  do {                                                                                                                \
    if (__builtin_expect(p.timing != nullptr, 0) && lane == 0) p.timing[((size_t)hid * 16 + wave) * 4 + (i)] = wall_clock64(); \
  } while (0)

// one (key block, query block) step of the dK / dV computation, the lane being a key: s / dp from the Q / dO row fragments at LDS
// offset `qoff` (dO image `img` bytes behind), P and dS with the per-query -lse / -delta from LDS, dv += dO^T P, dk += Q^T dS
__device__ __forceinline__ void dkv_step(const bf16x8 (&kf)[4], const bf16x8 (&vf)[4], f32x16 (&dk)[2], f32x16 (&dv)[2],
                                         const int (&rpart)[4], const int (&tpart)[2][2], int qoff, int img, const float* lse_s,
                                         const float* dlt_s, f32x2 sc2v, f32x2 scv, int q0, int N, bool mask, int hi) {
  f32x16 s, dp;
  zero16r(s);
  zero16r(dp);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row_at(lds_addr(rpart[ks], qoff)), kf[ks], s, 0, 0, 0);
    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row_at(lds_addr(rpart[ks], qoff + img)), vf[ks], dp, 0, 0, 0);
  }
  f32x16 pr;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 l4 = *(const f32x4*)(lse_s + q0 + 8 * g + 4 * hi);  // -lse * log2(e) of the four query rows
    const f32x4 d4 = *(const f32x4*)(dlt_s + q0 + 8 * g + 4 * hi);  // -delta * scale
#pragma unroll
    for (int e2 = 0; e2 < 2; ++e2) {
      const int r = 4 * g + 2 * e2;
      f32x2 e = {s[r], s[r + 1]}, gq = {dp[r], dp[r + 1]};
      e = __builtin_elementwise_fma(e, sc2v, f32x2{l4[2 * e2], l4[2 * e2 + 1]});
      gq = __builtin_elementwise_fma(gq, scv, f32x2{d4[2 * e2], d4[2 * e2 + 1]});
      e[0] = fast_exp2(e[0]);
      e[1] = fast_exp2(e[1]);
      if (mask) {
        if (q0 + 8 * g + 4 * hi + 2 * e2 >= N) e[0] = 0.f;
        if (q0 + 8 * g + 4 * hi + 2 * e2 + 1 >= N) e[1] = 0.f;
      }
      pr[r] = e[0];
      pr[r + 1] = e[1];
      e *= gq;
      s[r] = e[0];
      s[r + 1] = e[1];
    }
  }
  const bf16x8 p0 = pack8r(pr, 0), p1 = pack8r(pr, 8), d0 = pack8r(s, 0), d1 = pack8r(s, 8);
#pragma unroll
  for (int db = 0; db < 2; ++db) {
    const int tq0 = lds_addr(tpart[db][0], qoff), tq1 = lds_addr(tpart[db][1], qoff);
    const int tg0 = lds_addr(tpart[db][0], qoff + img), tg1 = lds_addr(tpart[db][1], qoff + img);
    dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_at(tg0, tg1, 0), p0, dv[db], 0, 0, 0);
    dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_at(tg0, tg1, 1), p1, dv[db], 0, 0, 0);
    dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_at(tq0, tq1, 0), d0, dk[db], 0, 0, 0);
    dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_at(tq0, tq1, 1), d1, dk[db], 0, 0, 0);
  }
}

// the 9th block's rows of one gradient: partial sums of the row blocks -> slots; the storing threads add them in block order
__device__ __forceinline__ void put_slot(float* slot, const f32x16 (&acc)[2], int hi) {
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[db][4 * g + e];
      *(f32x4*)(slot + 32 * db + 8 * g + 4 * hi) = v;
    }
}
__device__ __forceinline__ void store_acc_rows(bf16* row, const f32x16 (&acc)[2], int hi, const bf16* sin_t, const bf16* cos_t, long t) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 lo, hv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      lo[e] = acc[0][4 * g + e];
      hv[e] = acc[1][4 * g + e];
    }
    store_grad_pair(row, 8 * g + 4 * hi, lo, hv, sin_t, cos_t, t);
  }
}

// W = waves = full row blocks per head: 8 (N = 225 .. 258) or 2 (N = 33 .. 66: the 37-token local crops, four workgroups per CU)
template <int W>
__global__ __launch_bounds__(64 * W) void attn_bwd_fused_kernel(const AttnResArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5;
  const int img = p.npad * 128, nb = p.npad >> 5, last = nb - 1;
  const bool odd = nb > W;
  const int nx = odd ? p.N - 32 * W : 0;  // valid rows of the 9th block
  // LDS: Q | K | V | dO images, then -lse log2(e) and -delta scale per query, then the slots of the 9th block's dq, dk, dv rows
  const int QO = 0, KO = img, VO = 2 * img, GO = 3 * img;
  float* lse_s = (float*)(smem + 4 * img);
  float* dlt_s = lse_s + p.npad;
  float* xq = dlt_s + p.npad;           // [W + 1][PB_XROWS][64]
  float* xk = xq + (W + 1) * PB_XROWS * 64;
  float* xv = xk + (W + 1) * PB_XROWS * 64;
  const int hid = blockIdx.x, b = hid / p.heads, h = hid - b * p.heads;
  PB_STAMP(0);
  stage_image_asm(p.q + (long)b * p.sb + h * 64, p.sn, p.N, p.npad, QO, wave, W, lane);
  stage_image_asm(p.d_o + (long)b * p.sbo + h * 64, p.sno, p.N, p.npad, GO, wave, W, lane);
  stage_image_asm(p.k + (long)b * p.sb + h * 64, p.sn, p.N, p.npad, KO, wave, W, lane);
  stage_image_asm(p.v + (long)b * p.sb + h * 64, p.sn, p.N, p.npad, VO, wave, W, lane);
  const float sc2 = p.scale * LOG2E_R;
  const f32x2 sc2v = {sc2, sc2}, scv = {p.scale, p.scale};
  int rpart[4], tpart[2][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) rpart[ks] = row_part(lane) ^ (ks << 5);
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) tpart[db][hf] = tr_part(lane, hf) ^ (db << 6);
  const long srow0 = ((long)b * p.heads + h) * p.N;
  const int ri = wave * 32 + (lane & 31);                     // this lane's row of block `wave` (query in phase 1, key in phase 2)
  const int rx = 32 * W + (lane & 31);                 // ... and of the 9th block
  // O rows and lse of this lane's queries (the only per-lane global loads left)
  bf16x8 of[4], of2[4];
  float lse_raw, lse_raw2 = 0.f;
  {
    const int qc = min(ri, p.N - 1);
    const bf16* orow = p.o + (long)b * p.sbo + h * 64 + (long)qc * p.sno + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) of[ks] = *(const bf16x8*)(orow + ks * 16);
    lse_raw = p.lse[srow0 + qc];
    if (odd) {
      const int qc2 = min(rx, p.N - 1);
      const bf16* orow2 = p.o + (long)b * p.sbo + h * 64 + (long)qc2 * p.sno + hi * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) of2[ks] = *(const bf16x8*)(orow2 + ks * 16);
      lse_raw2 = p.lse[srow0 + qc2];
    }
  }
  wait_all_dma();
  __syncthreads();
  PB_STAMP(1);
  // ------------------------------------------------------------------------------------------ phase 1: dQ (the lane is a query)
  {
    bf16x8 qf[4], dof[4];
    const int own = __builtin_amdgcn_readfirstlane(wave * 4096);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = frag_row_at(lds_addr(rpart[ks], QO + own));
      dof[ks] = frag_row_at(lds_addr(rpart[ks], GO + own));
    }
    const float dlt = row_delta(of, dof);
    if (hi == 0) {
      lse_s[ri] = -lse_raw * LOG2E_R;
      dlt_s[ri] = -dlt * p.scale;
      if (ri < p.N) p.delta[srow0 + ri] = dlt;
    }
    {
      const float lse2 = lse_raw * LOG2E_R;
      const f32x2 nlse = {-lse2, -lse2}, ndlt = {-dlt * p.scale, -dlt * p.scale};
      f32x16 dq[2];
      zero16r(dq[0]);
      zero16r(dq[1]);
      for (int kb = 0; kb < nb; ++kb)
        dq_step(qf, dof, dq, rpart, tpart, __builtin_amdgcn_readfirstlane(KO + kb * 4096), img, sc2v, nlse, scv, ndlt, kb * 32, p.N,
                kb == last && (p.N & 31), hi);
      if (ri < p.N) {
        const bool rot = p.rope_sin && ri >= p.rope_prefix;
        store_acc_rows(p.dq + (long)b * p.sb + (long)ri * p.sn + h * 64, dq, hi, rot ? p.rope_sin : nullptr, p.rope_cos,
                       (long)(ri - p.rope_prefix) * 64);
      }
    }
    if (odd) {  // the 9th block's queries against key block `wave` (wave 0: and against the 9th key block)
      const int ox = __builtin_amdgcn_readfirstlane(W * 4096);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        qf[ks] = frag_row_at(lds_addr(rpart[ks], QO + ox));
        dof[ks] = frag_row_at(lds_addr(rpart[ks], GO + ox));
      }
      const float dlt2 = row_delta(of2, dof);
      if (wave == 0 && hi == 0) {
        lse_s[rx] = -lse_raw2 * LOG2E_R;
        dlt_s[rx] = -dlt2 * p.scale;
        if (rx < p.N) p.delta[srow0 + rx] = dlt2;
      }
      const float lse2 = lse_raw2 * LOG2E_R;
      const f32x2 nlse = {-lse2, -lse2}, ndlt = {-dlt2 * p.scale, -dlt2 * p.scale};
      for (int kb = wave; kb < nb; kb += W) {
        f32x16 dqx[2];
        zero16r(dqx[0]);
        zero16r(dqx[1]);
        dq_step(qf, dof, dqx, rpart, tpart, __builtin_amdgcn_readfirstlane(KO + kb * 4096), img, sc2v, nlse, scv, ndlt, kb * 32, p.N,
                kb == last && (p.N & 31), hi);
        if ((lane & 31) < nx) put_slot(xq + (kb * PB_XROWS + (lane & 31)) * 64, dqx, hi);
      }
    }
  }
  __syncthreads();  // lse_s / dlt_s and the 9th block's dq slots are complete
  PB_STAMP(2);
  if (odd && (int)threadIdx.x < nx * 8) {
    const int row = threadIdx.x >> 3, d0 = 4 * (threadIdx.x & 7), qrow = 32 * W + row;
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hv = {0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < nb; ++kb) {
      lo += *(const f32x4*)(xq + (kb * PB_XROWS + row) * 64 + d0);
      hv += *(const f32x4*)(xq + (kb * PB_XROWS + row) * 64 + 32 + d0);
    }
    const bool rot = p.rope_sin && qrow >= p.rope_prefix;
    store_grad_pair(p.dq + (long)b * p.sb + (long)qrow * p.sn + h * 64, d0, lo, hv, rot ? p.rope_sin : nullptr, p.rope_cos,
                    (long)(qrow - p.rope_prefix) * 64);
  }
  // ------------------------------------------------------------------------------------------ phase 2: dK, dV (the lane is a key)
  {
    bf16x8 kf[4], vf[4];
    const int own = __builtin_amdgcn_readfirstlane(wave * 4096);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = frag_row_at(lds_addr(rpart[ks], KO + own));
      vf[ks] = frag_row_at(lds_addr(rpart[ks], VO + own));
    }
    {
      f32x16 dk[2], dv[2];
      zero16r(dk[0]);
      zero16r(dk[1]);
      zero16r(dv[0]);
      zero16r(dv[1]);
      for (int qb = 0; qb < nb; ++qb)
        dkv_step(kf, vf, dk, dv, rpart, tpart, __builtin_amdgcn_readfirstlane(QO + qb * 4096), GO - QO, lse_s, dlt_s, sc2v, scv, qb * 32,
                 p.N, qb == last && (p.N & 31), hi);
      if (ri < p.N) {
        const bool rot = p.rope_sin && ri >= p.rope_prefix;
        store_acc_rows(p.dk + (long)b * p.sb + (long)ri * p.sn + h * 64, dk, hi, rot ? p.rope_sin : nullptr, p.rope_cos,
                       (long)(ri - p.rope_prefix) * 64);
        store_acc_rows(p.dv + (long)b * p.sb + (long)ri * p.sn + h * 64, dv, hi, nullptr, nullptr, 0);
      }
    }
    if (odd) {  // the 9th block's keys against query block `wave` (wave 0: and against the 9th query block)
      const int ox = __builtin_amdgcn_readfirstlane(W * 4096);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        kf[ks] = frag_row_at(lds_addr(rpart[ks], KO + ox));
        vf[ks] = frag_row_at(lds_addr(rpart[ks], VO + ox));
      }
      for (int qb = wave; qb < nb; qb += W) {
        f32x16 dkx[2], dvx[2];
        zero16r(dkx[0]);
        zero16r(dkx[1]);
        zero16r(dvx[0]);
        zero16r(dvx[1]);
        dkv_step(kf, vf, dkx, dvx, rpart, tpart, __builtin_amdgcn_readfirstlane(QO + qb * 4096), GO - QO, lse_s, dlt_s, sc2v, scv,
                 qb * 32, p.N, qb == last && (p.N & 31), hi);
        if ((lane & 31) < nx) {
          put_slot(xk + (qb * PB_XROWS + (lane & 31)) * 64, dkx, hi);
          put_slot(xv + (qb * PB_XROWS + (lane & 31)) * 64, dvx, hi);
        }
      }
    }
  }
  if (odd) {
    __syncthreads();
    if ((int)threadIdx.x < nx * 8) {
      const int row = threadIdx.x >> 3, d0 = 4 * (threadIdx.x & 7), krow_i = 32 * W + row;
      f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hv = lo, c0 = lo, c1 = lo;
      for (int qb = 0; qb < nb; ++qb) {
        lo += *(const f32x4*)(xk + (qb * PB_XROWS + row) * 64 + d0);
        hv += *(const f32x4*)(xk + (qb * PB_XROWS + row) * 64 + 32 + d0);
        c0 += *(const f32x4*)(xv + (qb * PB_XROWS + row) * 64 + d0);
        c1 += *(const f32x4*)(xv + (qb * PB_XROWS + row) * 64 + 32 + d0);
      }
      const bool rot = p.rope_sin && krow_i >= p.rope_prefix;
      store_grad_pair(p.dk + (long)b * p.sb + (long)krow_i * p.sn + h * 64, d0, lo, hv, rot ? p.rope_sin : nullptr, p.rope_cos,
                      (long)(krow_i - p.rope_prefix) * 64);
      store_grad_pair(p.dv + (long)b * p.sb + (long)krow_i * p.sn + h * 64, d0, c0, c1, nullptr, nullptr, 0);
    }
  }
  PB_STAMP(3);
}

// a head with >= 4 row blocks is split over two workgroups (each stages the whole K / V or Q / dO image, 2 x 37 KB at N = 257):
// two of them fit a CU, so one's staging overlaps the other's key loop
// -- worth it only while heads x images leaves CUs short of work (<= 512 workgroups: +1.2 % on the 32-image rec step, -0.4 %
// when the 64-image passes already run three full rounds and the second copy of K / V is pure extra traffic)
static int res_waves_per_block(int nw, int groups) {
  return (groups <= 512 && nw >= 4) ? (nw + 1) / 2 : nw;
}

template <typename K>
static void set_lds(K kern, int bytes) {
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

static long long* g_attn_timing = nullptr;
static int g_attn_wpb = 0;      // diagnostics: > 0 per-head kernels with that many waves per workgroup, -1 per-head kernels, 0 default
static int g_attn_lds_pad = 0;
static int g_attn_stagger = 0;  // diagnostics: extra dynamic LDS bytes per backward workgroup (occupancy experiments)

// host side, called by vtp_attn_fwd / vtp_attn_bwd (attention.hip) for non-causal N <= RES_MAXN
int attn_resident_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int N, int heads, long sb,
                      long sn, long sbo, long sno, float scale, hipStream_t s) {
  AttnResArgs a = {};
  a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.out = (bf16*)o; a.lse = lse;
  a.B = B; a.N = N; a.heads = heads; a.npad = (N + 31) / 32 * 32; a.sb = sb; a.sn = sn; a.sbo = sbo; a.sno = sno; a.scale = scale;
  static bool attr = false;
  if (!attr) {
    set_lds(attn_fwd_res_kernel, 2 * RES_MAXN * 128);
    set_lds(attn_fwd_res2_kernel, 2 * RES_MAXN * 128 + 5 * RES_SPLIT_COLS * 2 * 34 * 4);
    attr = true;
  }
  {
    const int ntiles = a.npad / 32, cols = N - 32 * (ntiles - 1);
    const bool odd = (ntiles & 1) && ntiles > 1;
    if (!odd || cols <= RES_SPLIT_COLS) {  // two query tiles per wave, single pass, odd last tile split over the key blocks
      const int waves = std::max(1, ntiles / 2);
      const int lds = 2 * a.npad * 128 + (odd ? waves * cols * 2 * 34 * 4 : 0);
      static long long* tbuf = nullptr;
      static const bool timing = getenv("VTP_ATTN_TIMING") != nullptr;
      if (timing && !tbuf && hipMalloc((void**)&tbuf, 64 * 8) != hipSuccess) tbuf = nullptr;
      if (timing) a.delta = (float*)tbuf;
      hipLaunchKernelGGL(attn_fwd_res2_kernel, dim3(heads, B), dim3(64 * waves), lds, s, a);
      if (timing && tbuf) {  // diagnosis only: synchronous
        int occ = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, attn_fwd_res2_kernel, 64 * waves, lds);
        fprintf(stderr, "[attn timing] workgroups per CU by the runtime's count: %d (lds %d B, %d waves)\n", occ, lds, waves);
        long long hbuf[64];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(hbuf, tbuf, sizeof hbuf, hipMemcpyDeviceToHost);
        for (int w = 0; w < waves; ++w) {
          fprintf(stderr, "[attn timing] wave %d:", w);
          for (int i = 1; i < 7; ++i) fprintf(stderr, " t%d=%lld", i, hbuf[w * 8 + i] - hbuf[0]);
          fprintf(stderr, "\n");
        }
      }
      return check_launch("attn_fwd_resident2");
    }
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
  a.timing = g_attn_timing;
  a.stagger = g_attn_stagger;
  static bool attr = false;
  if (!attr) {
    set_lds(attn_bwd_dq_res_kernel, 2 * RES_MAXN * 128 + 32768);  // (+ room for the diagnostic pad)
    set_lds(attn_bwd_dkv_res_kernel, 2 * RES_MAXN * 128 + 8 * RES_MAXN + 32768);
    attr = true;
  }
  const int nw = a.npad / 32;
  auto fits = [&](int W) { return nw == W || (nw == W + 1 && N - 32 * W <= PB_XROWS); };
  // (the LDS-DMA of the fused kernel addresses a head's rows with 32-bit byte offsets from a 64-bit base)
  const bool off32 = (long)N * sn * 2 < (1L << 31) && (long)N * sno * 2 < (1L << 31);
  if (g_attn_wpb == 0 && off32 && (fits(PB_WAVES) || fits(2))) {
    // fused kernel: one workgroup of W waves per head, the four operand images in LDS
    auto lds_of = [](int npad, int W) { return 4 * npad * 128 + 8 * npad + 3 * (W + 1) * PB_XROWS * 256; };
    static bool attr_p = false;
    if (!attr_p) {
      set_lds(attn_bwd_fused_kernel<PB_WAVES>, lds_of(32 * (PB_WAVES + 1), PB_WAVES));
      set_lds(attn_bwd_fused_kernel<2>, lds_of(32 * 3, 2));
      attr_p = true;
    }
    if (fits(PB_WAVES))
      hipLaunchKernelGGL(attn_bwd_fused_kernel<PB_WAVES>, dim3(B * heads), dim3(64 * PB_WAVES), lds_of(a.npad, PB_WAVES), s, a);
    else
      hipLaunchKernelGGL(attn_bwd_fused_kernel<2>, dim3(B * heads), dim3(128), lds_of(a.npad, 2), s, a);
    return check_launch("attn_bwd_fused");
  }
  const int wpb = g_attn_wpb > 0 ? std::min(g_attn_wpb, nw) : res_waves_per_block(nw, heads * B);
  const dim3 grid(heads, B, (nw + wpb - 1) / wpb), block(64 * wpb);
  if (a.timing) {  // diagnosis only
    int o1 = -1, o2 = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&o1, attn_bwd_dq_res_kernel, block.x, 2 * a.npad * 128);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&o2, attn_bwd_dkv_res_kernel, block.x, 2 * a.npad * 128 + 8 * a.npad);
    fprintf(stderr, "[attn timing] workgroups per CU by the runtime's count: dQ %d, dK/dV %d (%d waves, lds %d / %d B)\n", o1, o2,
            (int)block.x / 64, 2 * a.npad * 128, 2 * a.npad * 128 + 8 * a.npad);
  }
  hipLaunchKernelGGL(attn_bwd_dq_res_kernel, grid, block, 2 * a.npad * 128 + g_attn_lds_pad, s, a);
  if (a.timing) a.timing += (size_t)grid.x * grid.y * grid.z * 64;  // second half of the buffer: the dK / dV kernel
  hipLaunchKernelGGL(attn_bwd_dkv_res_kernel, grid, block, 2 * a.npad * 128 + 8 * a.npad + g_attn_lds_pad, s, a);
  return check_launch("attn_bwd_resident");
}

}  // namespace vtp

// diagnostics (tools/attn_bwd_timeline.py): device buffer of 2 x [workgroups][16 waves][4] 64-bit s_memrealtime stamps written by
// the resident backward kernels (dQ kernel first, dK/dV kernel behind it), or null = off.  Process-global.
extern "C" int vtp_attn_debug(void* timing, int lds_pad, int waves_per_wg, int stagger_ticks) {
  vtp::g_attn_stagger = stagger_ticks;
  vtp::g_attn_wpb = waves_per_wg;
  vtp::g_attn_timing = (long long*)timing;
  vtp::g_attn_lds_pad = lds_pad;
  return 0;
}
