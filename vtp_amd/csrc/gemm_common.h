// Shared pieces of the bf16 MFMA GEMM kernels (gemm.hip: ring / 2-barrier main loops; gemm8p.hip: 256x256 8-phase main loop):
// the argument block, the row remaps and the fused epilogues.  Device code only.
#pragma once
#include <cstdlib>
#include "common.h"
#include "vtp_hip.h"

namespace vtp {

// zero source block for K / row tails of the LDS-DMA staging (one copy per translation unit: device code is not linked across TUs)
static __device__ __attribute__((aligned(16))) unsigned int g_zero_block[16] = {0};

struct GemmArgs {
  const bf16* A;   // [M, lda]
  const bf16* B;   // [N, ldb]
  void* C;         // bf16 or f32 [*, ldc]
  void* C2;        // secondary output (SwiGLU: x12 pre-activations; GELU: pre-activation), may be null
  const float* bias;   // [N] or null   (SwiGLU: interleaved [2H])
  const float* gamma;  // [N] LayerScale or null
  const float* resid;  // f32 [*, ldc] residual (EPI_F32 only) or null
  int M, N, K;
  int lda, ldb, ldc, ldc2;
  int a_grp, a_pre;  // A row remap: row(m) = m + (m / a_grp + 1) * a_pre   (a_grp == 0: identity)
  int c_grp, c_pre;  // C row remap (same formula; c_grp < 0: SwiGLU de-interleave)
  int b_grp, b_pre;  // TRANS mode only: token-row remap of the B operand
  int k_split;       // K elements per blockIdx.z slice (multiple of 64)
  int xcd_swizzle;
  float alpha;
  // 3x3 convolution as an implicit GEMM over a zero-bordered NHWC image stack (vtp_conv3x3): A = [NB*(H+2)*(W+2), conv_cin]
  // pixel rows, k-tile kt reads tap (kt*64)/conv_cin at row offset (ky-1)*conv_w2 + (kx-1).  conv_cin == 0: plain GEMM.
  int conv_cin, conv_w2, conv_p, conv_h2;
  // fused apply_rope (attention.py:70-89) in the bf16 epilogue of the qkv projection: output row m is rotated with row
  // rope_pos[m] of the bf16 tables [*, 64] (rope_pos[m] < 0: cls / prefix rows, not rotated); only columns < rope_cols (the
  // q and k thirds; a multiple of 128) are rotated.  rope_pos == null: plain epilogue.
  const int* rope_pos;
  const bf16* rope_sin;
  const bf16* rope_cos;
  int rope_cols;
  // TRANS (weight-gradient) mode: if non-null, the column sums of A over the reduction rows (= the bias gradient of the linear
  // layer, db[m] = sum_t dy[t, m]) are ACCUMULATED into colsum[remap_row(m, c_grp, c_pre)] by the kernels that support it
  float* colsum;
  // fused SwiGLU backward in the bf16 epilogue of the w3 dgrad GEMM (ffn.py:78-81 backward): the GEMM result is dh [M, N = H]; with
  // swiglu_pre = the saved pre-activations x12 bf16 [M, 2N] (16-column groups = 8 of x1 | 8 of x2, row stride swiglu_ld) the
  // epilogue writes dx12 = d(silu(x1) x2)/d(x12) (.) dh into C as [M, 2N] (row stride ldc) and dh itself never reaches HBM.
  // null: plain epilogue.
  const bf16* swiglu_pre;
  int swiglu_ld;
  int act_quick;  // EPI_GELU only: 1 = QuickGELU x sigmoid(1.702 x) (text_quick_gelu, layers/activation.py:5-12) instead of erf-GELU
  // in-launch split-K combine of the 256 x 256 kernel (few-tile shapes with a long K): fp32 partial sums [tiles][splits] x 256 KiB and
  // one arrival ticket per tile; null: off.  Set by the launcher (per-stream scratch), never by callers.
  float* part;
  int* ticket;
  // dynamic tile assignment of the persistent 256 x 256 kernels (round 6): per-stream queue words in device memory -- heads of the 8
  // per-XCD tile queues at tq[16 x], the exit counter at tq[128] (all zero between launches); null: tiles are dealt statically.
  // Set by the launcher, never by callers.
  int* tq;
  // diagnostics (vtp_gemm_debug): per workgroup and tile, s_memrealtime stamps {tile start, k loop done, epilogue issued}; null = off
  unsigned long long* timing;
  int dbg_delay;  // diagnostics: > 0: every second workgroup (per XCD) starts this many 10-ns ticks late (lock-step experiments)
};

// VTP_GEMM_CUS=n (read once): the persistent kernels launch n workgroup slots instead of one per CU (rounded down to a multiple of 8)
// -- leaves CUs to a kernel that holds them for the whole backward (RCCL channels at N > 1; INTEGRATION.md "Running beside RCCL")
static inline int gemm_cu_cap(int cus) {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("VTP_GEMM_CUS");
    v = e ? atoi(e) : 0;
    if (v < 0) v = 0;
  }
  if (v >= 8 && v < cus) return v - v % 8;
  return cus;
}

#ifndef VTP_EPI_F32_PF
#define VTP_EPI_F32_PF 1  // residual rows requested this many LDS passes ahead in the staged fp32 epilogue (A/B builds: -DVTP_EPI_F32_PF=1)
#endif
enum { EPI_BF16 = 0, EPI_F32 = 1, EPI_SWIGLU = 2, EPI_GELU = 3, EPI_F32_ATOMIC = 4, EPI_F32_SLAB = 5, EPI_CONV_RELU = 6,
       EPI_CONV_MASK = 7 };

__device__ __forceinline__ int remap_row(int m, int grp, int pre) {
  if (grp > 0) return m + (m / grp + 1) * pre;
  if (grp < 0) return ((m >> 4) << 3) + (m & 7) + ((m & 8) ? pre : 0);  // SwiGLU de-interleave
  return m;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// does the epilogue of this (EPI, shape) stage its output through the wave's private LDS region?  (wave-uniform; the caller
// must make sure the region is free -- e.g. a barrier after the last fragment reads when the region aliases the operand ring)
template <int EPI, bool TRANS, int WTN, int REGION>
__device__ __forceinline__ bool gemm_epilogue_uses_lds(const GemmArgs& p) {
  if constexpr (TRANS || WTN < 64) return false;
  if constexpr (EPI == EPI_BF16) return (p.N & 7) == 0 && (p.xcd_swizzle & 2);
  if constexpr (EPI == EPI_SWIGLU) return (p.xcd_swizzle & 2) && (p.N & 15) == 0;
  if constexpr (EPI == EPI_F32) return REGION / 4096 == 1 && (p.xcd_swizzle & 2) && (p.N & 3) == 0;
  return false;
}

// Epilogue of one wave: acc[i][j] = 32x32 fp32 tile (i over the wave's WTN / 32 column blocks, j over its WTM / 32 row blocks)
// of the output tile at (m0, n0); the wave's sub-tile starts at row wm * WTM, column wn * WTN.  MFMA operands were swapped
// (weight rows = A operand), so a lane holds, for output row m = .. + (lane & 31), columns nb + 8*q + 4*hi + (0..3), q = 0..3.
// reg: this wave's private LDS staging region of REGION bytes (or null: direct stores); reg2: a second private region of
// REGION / 2 bytes (or null) -- with it the SwiGLU epilogue stages a row block's pre-activations and hidden values side by side
// (one LDS write -> read -> store chain per block instead of two back to back).
// XMODE (bf16 epilogue only): which fused extra the LDS-staged store path carries -- -1: decided at run time from the argument block
// (the ring kernels) | 0: none | 1: apply_rope (rope_pos) | 2: SwiGLU backward (swiglu_pre).  The 256 x 256 kernel instantiates
// 0 / 1 / 2 separately: the plain variant then carries neither the extras' registers (the pre-activation prefetch alone is 32)
// nor conditionally waited loads that make hipcc drain `vmcnt(0)` at the head of the k loop.
template <int EPI, bool TRANS, int WTM, int WTN, int REGION, int XMODE = -1>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x16 (&acc)[WTN / 32][WTM / 32], char* reg, int m0, int n0,
                                              int wm, int wn, int lane_in, int zslice = blockIdx.z, char* reg2 = nullptr) {
  int lane = lane_in;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  // Everything the epilogue derives from the lane id (rows, swizzled LDS offsets, column offsets: ~35 registers) is invariant over
  // the persistent tile loop, so hipcc hoists it in front of that loop, cannot keep it in registers through the k loop, and spills
  // it -- and every scratch reload inside the store loops below is followed by `s_waitcnt vmcnt(0)`, i.e. drains all the global
  // stores issued so far AND the LDS-DMA prefetch of the next tile (measured: 4.5 us of a bf16 epilogue, 12.8 us of the SwiGLU
  // one).  An opaque copy of the lane id per tile keeps those values tile-local: recomputed with a few VALU, never spilled.
  // (not for the fp32-residual epilogue and the SwiGLU-backward variant: their own live sets -- residual / pre-activation
  // prefetch -- leave no room to recompute; measured slower with it)
  if constexpr (EPI != EPI_F32) asm volatile("" : "+v"(lane));
  const int hi = lane >> 5;
  const bool x_rope = XMODE < 0 ? p.rope_pos != nullptr : XMODE == 1;
  const bool x_swiglu = XMODE < 0 ? p.swiglu_pre != nullptr : XMODE == 2;
  // ---- epilogue: lane holds, for output row m, columns nb + 8*q + 4*hi + (0..3), q = 0..3 ----
  // bf16 outputs go through LDS so that the global stores are full 128/256-B row segments (16 B per lane, consecutive lanes
  // = consecutive addresses) instead of 8-B pieces of 32 different rows per instruction: each wave transposes 32-row
  // blocks of its sub-tile in a private region of the ring slot the last k-tile occupied (XOR-swizzled 16-B chunks).
  bool staged_out = false;
  if constexpr (EPI == EPI_BF16 && WTN >= 64 && !TRANS) {
    if (reg && (p.N & 7) == 0 && (p.xcd_swizzle & 2)) {
      staged_out = true;
      constexpr int RB = WTN * 2, CPR = RB / 16;
      static_assert(32 * RB <= REGION, "wave region too small for a 32-row block");
      const int lane_t = lane;
      constexpr int TPB = 32 * CPR / 64;  // 16-B items per lane and 32-row block
      // the lane's bias values do not depend on the row block: ONE round of loads per tile (in front of the first block's
      // conversions) instead of one L2 round trip inside every block
      // (XMODE 1, apply_rope, keeps the bias values like the plain variant: a pre-pass that moved the bias into the accumulators to free
      // registers for the rotation tables was tried in round 5 and measured no faster; removed)
      constexpr bool HAS_BIAS = XMODE != 2;  // (the SwiGLU-backward instantiation is a dgrad: no bias, no registers for one)
      f32x4 bias_v[HAS_BIAS ? TN : 1][4];
      if constexpr (HAS_BIAS) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int n = n0 + wn * WTN + i * 32 + 8 * q + 4 * hi;
            bias_v[i][q] = (p.bias && n < p.N) ? *(const f32x4*)(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
          }
      }
      // fused SwiGLU backward: the x1 | x2 pre-activations of row block j + 1 are requested before block j is transposed, converted
      // and stored (two register sets): the epilogue is bound by the bytes a CU keeps in flight, not by HBM -- with one block's loads
      // issued behind the previous block's stores every block paid a full load round trip (25 us per tile, 11 B/clk per CU)
      bf16x8 sx1[2][TPB], sx2[2][TPB];
      auto fetch_pre = [&](int j, bf16x8 (&d1)[TPB], bf16x8 (&d2)[TPB]) {
#pragma unroll
        for (int t = 0; t < TPB; ++t) {
          const int idx = t * 64 + lane, rr = idx / CPR, c = idx % CPR;
          const int m = min(m0 + wm * WTM + j * 32 + rr, p.M - 1), n = min(n0 + wn * WTN + c * 8, p.N - 8);
          const bf16* xr = p.swiglu_pre + (size_t)m * p.swiglu_ld + 2 * n;
          d1[t] = *(const bf16x8*)xr;
          d2[t] = *(const bf16x8*)(xr + 8);
        }
      };
      if (x_swiglu) fetch_pre(0, sx1[0], sx2[0]);
      // fused apply_rope, XMODE 1: the table rows of a block are two DEPENDENT loads (position of the row, then its cos / sin chunks), and
      // issued behind the previous block's stores they also wait for those (one in-order vmcnt): three exposed round trips per 32-row
      // block, 12 us of a 256 x 256 tile that computes for 20.  Here the positions of the wave tile's rows are loaded once (row k*64 + lane
      // in lane `lane`, handed out by ds_bpermute), and the tables of block j + 1 are requested after block j's rotation has consumed
      // its own, item by item, each BEFORE that item's store: older than it in the queue, in flight under the next block's conversions.
      const bool rope_wave = XMODE == 1 && n0 + wn * WTN < p.rope_cols;  // wave-uniform: this wave's columns are q / k heads
      int rpos[XMODE == 1 ? WTM / 64 : 1];
      bf16x8 rcs[XMODE == 1 ? TPB : 1], rsn[XMODE == 1 ? TPB : 1];
      int rps[XMODE == 1 ? TPB : 1];
      auto fetch_tab = [&](int j, int t) {
        {
          const int row = j * 32 + (t * 64 + lane_t) / CPR;  // row of the wave tile this lane handles as item t of block j
          const int pos = __shfl(rpos[(row >> 6) % (WTM / 64)], row & 63, 64);
          rps[t] = pos;
          const size_t off = (size_t)max(pos, 0) * 64 + (lane_t % CPR & 7) * 8;
          rcs[t] = *(const bf16x8*)(p.rope_cos + off);
          rsn[t] = *(const bf16x8*)(p.rope_sin + off);
        }
      };
      if constexpr (XMODE == 1) {
        static_assert(WTM % 64 == 0 && (64 % CPR) == 0, "rope position hand-out assumes 64-row groups");
        if (rope_wave) {
#pragma unroll
          for (int k = 0; k < WTM / 64; ++k) {
            const int m = m0 + wm * WTM + k * 64 + lane_t;
            rpos[k] = m < p.M ? p.rope_pos[m] : -1;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        if (x_swiglu && j + 1 < TM) fetch_pre(j + 1, sx1[(j + 1) & 1], sx2[(j + 1) & 1]);
        // (per row block again an opaque lane id: the LDS / global addressing of the four unrolled blocks is otherwise shared, lives
        // across the whole epilogue and spills -- with a drain of every load in flight behind each reload)
        int lane = lane_t, hi = lane_t >> 5, r = lane_t & 31;
        if constexpr (XMODE == 2 || XMODE == 1) {
          asm volatile("" : "+v"(lane));
          hi = lane >> 5;
          r = lane & 31;
        }
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int nl = i * 32 + 8 * q + 4 * hi;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e] * p.alpha;
            if constexpr (HAS_BIAS) v += bias_v[i][q];
            *(bf16x4*)(reg + r * RB + (((nl >> 3) ^ (r & (CPR - 1))) << 4) + hi * 8) = __builtin_convertvector(v, bf16x4);
          }
        if constexpr (XMODE == 1) {
          if (rope_wave) {  // per item: read back, rotate, request the item's tables of the NEXT block, store
            if (j == 0) {  // (behind block 0's conversions: the tables take the registers its accumulators leave)
#pragma unroll
              for (int t = 0; t < TPB; ++t) fetch_tab(0, t);
            }
#pragma unroll
            for (int t = 0; t < TPB; ++t) {
              const int idx = t * 64 + lane, rr = idx / CPR, c = idx % CPR;
              bf16x8 val = *(const bf16x8*)(reg + rr * RB + ((c ^ (rr & (CPR - 1))) << 4));
              typedef __attribute__((ext_vector_type(4))) int i32x4;
              const i32x4 vi = (i32x4)val;
              i32x4 pi;
#pragma unroll
              for (int w = 0; w < 4; ++w) pi[w] = __shfl_xor(vi[w], 4, 64);
              const bf16x8 prt = (bf16x8)pi;
              const float sg = (lane & 4) ? 1.f : -1.f;
              if (rps[t] >= 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                  val[e] = f2bf(bf2f(f2bf(bf2f(val[e]) * bf2f(rcs[t][e]))) + bf2f(f2bf(sg * bf2f(prt[e]) * bf2f(rsn[t][e]))));
              }
              if (j + 1 < TM) fetch_tab(j + 1, t);
              __builtin_amdgcn_sched_barrier(0);
              const int m = m0 + wm * WTM + j * 32 + rr, n = n0 + wn * WTN + c * 8;
              if (m < p.M && n < p.N) *(bf16x8*)((bf16*)p.C + (size_t)remap_row(m, p.c_grp, p.c_pre) * p.ldc + n) = val;
            }
            continue;
          }
        }
#pragma unroll
        for (int t = 0; t < TPB; ++t) {
          const int idx = t * 64 + lane, rr = idx / CPR, c = idx % CPR;
          bf16x8 val = *(const bf16x8*)(reg + rr * RB + ((c ^ (rr & (CPR - 1))) << 4));
          const int m = m0 + wm * WTM + j * 32 + rr, n = n0 + wn * WTN + c * 8;
          if (XMODE < 0 && x_rope && n0 + wn * WTN < p.rope_cols) {  // wave-uniform: this wave's columns are q / k heads (run-time variant)
            // a head is 8 consecutive 16-B chunks of the row image: the partner element d +- 32 sits in chunk c ^ 4 = lane ^ 4.
            // out = x*cos + rot_half(x)*sin with the three eager-bf16 roundings of the reference (rope_qk_kernel, bit-identical)
            typedef __attribute__((ext_vector_type(4))) int i32x4;
            const i32x4 vi = (i32x4)val;
            i32x4 pi;
#pragma unroll
            for (int w = 0; w < 4; ++w) pi[w] = __shfl_xor(vi[w], 4, 64);
            const bf16x8 prt = (bf16x8)pi;
            const int pos = m < p.M ? p.rope_pos[m] : -1;
            if (pos >= 0) {
              const bf16x8 cs = *(const bf16x8*)(p.rope_cos + (size_t)pos * 64 + (c & 7) * 8);
              const bf16x8 sn = *(const bf16x8*)(p.rope_sin + (size_t)pos * 64 + (c & 7) * 8);
              const float sg = (c & 4) ? 1.f : -1.f;
#pragma unroll
              for (int e = 0; e < 8; ++e)
                val[e] = f2bf(bf2f(f2bf(bf2f(val[e]) * bf2f(cs[e]))) + bf2f(f2bf(sg * bf2f(prt[e]) * bf2f(sn[e]))));
            }
          }
          if (x_swiglu) {  // wave-uniform: val = dh of one 8-column group -> dx1 | dx2 (same roundings as swiglu_bwd_kernel)
            if (m < p.M && n < p.N) {
              bf16x8 o1, o2;
              swiglu_bwd8(sx1[j & 1][t], sx2[j & 1][t], val, o1, o2);
              bf16* orow = (bf16*)p.C + (size_t)m * p.ldc + 2 * n;
              *(bf16x8*)orow = o1;
              *(bf16x8*)(orow + 8) = o2;
            }
            continue;
          }
          if (m < p.M && n < p.N) *(bf16x8*)((bf16*)p.C + (size_t)remap_row(m, p.c_grp, p.c_pre) * p.ldc + n) = val;
        }
      }
    }
  }
  if constexpr (EPI == EPI_SWIGLU && WTN >= 64 && !TRANS) {
    // SwiGLU: the pre-activations x12 (interleaved gemm columns, bf16) and the hidden activations (WTN/2 columns) both leave
    // through LDS; the activation itself is computed lane-locally first (quads (q, q+1) = (w1, w2) columns)
    if (reg && (p.xcd_swizzle & 2) && (p.N & 15) == 0) {
      staged_out = true;
      constexpr int RB = WTN * 2, CPR = RB / 16, RBH = WTN, CPRH = RBH / 16;
      static_assert(32 * RB <= REGION, "wave region too small for a 32-row block");
      const int r = lane & 31;
      f32x4 bias_v[TN][4];  // one round of bias loads per tile (see the bf16 path)
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wn * WTN + i * 32 + 8 * q + 4 * hi;
          bias_v[i][q] = n < p.N ? *(const f32x4*)(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        bf16x4 hs[TN][2];
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int q = 0; q < 4; q += 2) {
            const int nl = i * 32 + 8 * q + 4 * hi;
            f32x4 x1, x2, hsw;
            const f32x4 b1 = bias_v[i][q], b2 = bias_v[i][q + 1];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              x1[e] = acc[i][j][4 * q + e] * p.alpha + b1[e];
              x2[e] = acc[i][j][4 * q + 4 + e] * p.alpha + b2[e];
            }
            const bf16x4 x1b = __builtin_convertvector(x1, bf16x4), x2b = __builtin_convertvector(x2, bf16x4);
#pragma unroll
            for (int e = 0; e < 4; ++e) hsw[e] = bf2f(f2bf(silu_f(bf2f(x1b[e])))) * bf2f(x2b[e]);
            hs[i][q >> 1] = __builtin_convertvector(hsw, bf16x4);
            if (p.C2) {
              *(bf16x4*)(reg + r * RB + (((nl >> 3) ^ (r & (CPR - 1))) << 4) + hi * 8) = x1b;
              *(bf16x4*)(reg + r * RB + ((((nl >> 3) + 1) ^ (r & (CPR - 1))) << 4) + hi * 8) = x2b;
            }
          }
        char* regh = reg2 ? reg2 : reg;  // hidden block: its own region when there is one (written before x12 is read back)
        auto write_hidden = [&]() {      // local column (2i + q/2)*8 + 4hi of a [32][WTN/2] block
#pragma unroll
          for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
              *(bf16x4*)(regh + r * RBH + ((((2 * i + h2)) ^ (r & (CPRH - 1))) << 4) + hi * 8) = hs[i][h2];
        };
        if (reg2) write_hidden();
        if (p.C2) {
#pragma unroll
          for (int t = 0; t < 32 * CPR / 64; ++t) {
            const int idx = t * 64 + lane, rr = idx / CPR, c = idx % CPR;
            const bf16x8 val = *(const bf16x8*)(reg + rr * RB + ((c ^ (rr & (CPR - 1))) << 4));
            const int m = m0 + wm * WTM + j * 32 + rr, n = n0 + wn * WTN + c * 8;
            if (m < p.M && n < p.N) *(bf16x8*)((bf16*)p.C2 + (size_t)remap_row(m, p.c_grp, p.c_pre) * p.ldc2 + n) = val;
          }
        }
        if (!reg2) write_hidden();
#pragma unroll
        for (int t = 0; t < 32 * CPRH / 64; ++t) {
          const int idx = t * 64 + lane, rr = idx / CPRH, c = idx % CPRH;
          const bf16x8 val = *(const bf16x8*)(regh + rr * RBH + ((c ^ (rr & (CPRH - 1))) << 4));
          const int m = m0 + wm * WTM + j * 32 + rr, jh = ((n0 + wn * WTN) >> 1) + c * 8;
          if (m < p.M && 2 * jh < p.N) *(bf16x8*)((bf16*)p.C + (size_t)remap_row(m, p.c_grp, p.c_pre) * p.ldc + jh) = val;
        }
      }
    }
  }
  if constexpr (EPI == EPI_F32 && WTN >= 64 && !TRANS && REGION / 4096 == 1) {  // measured: pays for the 8-wave
    // 128x128 / 256x128 kernels (128-B staging rows); the 256-B-row variants of the 4-wave / 256x256 kernels lost 5-7 %
    // fp32 residual epilogue: raw accumulators through LDS in column groups that fit the wave's region; bias / LayerScale /
    // residual are applied after the read-back, where each lane owns 4 consecutive columns of one row (coalesced resid loads)
    if (reg && (p.xcd_swizzle & 2) && (p.N & 3) == 0) {
      staged_out = true;
      constexpr int IG = (REGION / 4096 >= TN) ? TN : (REGION / 4096);   // 32-column blocks per pass (32 rows x 128 B each)
      static_assert(IG >= 1 && TN % IG == 0, "bad column grouping");
      constexpr int RB = IG * 128, CPR = RB / 16;
      constexpr int T = 32 * CPR / 64, NPASS = TM * (TN / IG);
      const int r = lane & 31;
      // the residual rows of pass k + 1 are requested before pass k goes through LDS (they are then OLDER than pass k's stores in the
      // in-order VMEM queue: waiting for them does not wait for those stores): one HBM round trip per tile is exposed instead of one
      // per pass (the dominant cost of this epilogue at K = 768: 8 dependent round trips per 256x256 tile)
      // PF passes ahead (round 6: 1 -> 2; with one pass ahead a wave had 4 KiB of residual rows in flight and the tile's NPASS
      // dependent round trips -- not bandwidth -- set the length of this epilogue: tools/wgrad_timeline.py, 38 us per 256 x 256 tile)
      constexpr int PF = VTP_EPI_F32_PF;
      f32x4 rv[PF + 1][T];
      auto coords = [&](int pass, int t, int& m, int& n) {
        const int j = pass / (TN / IG), ig = pass % (TN / IG);
        const int idx = t * 64 + lane, rr = idx / CPR, c = idx % CPR;
        m = m0 + wm * WTM + j * 32 + rr;
        n = n0 + wn * WTN + ig * IG * 32 + c * 4;
      };
      auto fetch = [&](int pass, f32x4 (&dst)[T]) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
          int m, n;
          coords(pass, t, m, n);
          dst[t] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (p.resid && m < p.M && n < p.N) dst[t] = *(const f32x4*)(p.resid + (size_t)remap_row(m, p.c_grp, p.c_pre) * p.ldc + n);
        }
      };
#pragma unroll
      for (int f = 0; f < PF; ++f)
        if (f < NPASS) fetch(f, rv[f]);
      // bias / LayerScale values of the lane's columns: the column of item t is the same for every t and row block (64 % CPR == 0),
      // so they are loaded once per column group instead of inside every pass
      static_assert(64 % CPR == 0, "a lane's column must not depend on t");
      f32x4 bs[TN / IG], gm[TN / IG];
#pragma unroll
      for (int ig = 0; ig < TN / IG; ++ig) {
        const int n = n0 + wn * WTN + ig * IG * 32 + (lane % CPR) * 4;
        bs[ig] = (p.bias && n < p.N) ? *(const f32x4*)(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        gm[ig] = (p.gamma && n < p.N) ? *(const f32x4*)(p.gamma + n) : f32x4{1.f, 1.f, 1.f, 1.f};
      }
#pragma unroll
      for (int pass = 0; pass < NPASS; ++pass) {
        const int j = pass / (TN / IG), ig = pass % (TN / IG);
        if (pass + PF < NPASS) fetch(pass + PF, rv[(pass + PF) % (PF + 1)]);
#pragma unroll
        for (int ii = 0; ii < IG; ++ii)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[ig * IG + ii][j][4 * q + e] * p.alpha;
            *(f32x4*)(reg + r * RB + (((ii * 8 + 2 * q + hi) ^ (r & (CPR - 1))) << 4)) = v;
          }
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const int idx = t * 64 + lane, rr = idx / CPR, c = idx % CPR;
          f32x4 v = *(const f32x4*)(reg + rr * RB + ((c ^ (rr & (CPR - 1))) << 4));
          int m, n;
          coords(pass, t, m, n);
          if (m < p.M && n < p.N) {
            const size_t off = (size_t)remap_row(m, p.c_grp, p.c_pre) * p.ldc + n;
            v = (v + bs[ig]) * gm[ig] + rv[pass % (PF + 1)][t];
            *(f32x4*)((float*)p.C + off) = v;
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    if (staged_out) break;
    const int m = m0 + wm * WTM + j * 32 + (lane & 31);
    if (m >= p.M) continue;
    const int mc = remap_row(m, p.c_grp, p.c_pre);
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int nb = n0 + wn * WTN + i * 32 + 4 * hi;
      if constexpr (EPI == EPI_SWIGLU) {
        // interleaved weight rows: 16-row groups = [8 rows of w1 | 8 rows of w2]; quads (0,1) and (2,3) pair up.
#pragma unroll
        for (int q = 0; q < 4; q += 2) {
          const int n1 = nb + 8 * q;  // gemm column of the w1 quad; w2 quad is n1 + 8
          if (n1 >= p.N) continue;
          f32x4 b1 = *(const f32x4*)(p.bias + n1);
          f32x4 b2 = *(const f32x4*)(p.bias + n1 + 8);
          f32x4 x1, x2, hsw;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            x1[e] = acc[i][j][4 * q + e] * p.alpha + b1[e];
            x2[e] = acc[i][j][4 * q + 4 + e] * p.alpha + b2[e];
          }
          bf16x4 x1b = __builtin_convertvector(x1, bf16x4), x2b = __builtin_convertvector(x2, bf16x4);
          if (p.C2) {
            bf16* c2 = (bf16*)p.C2 + (size_t)mc * p.ldc2;
            *(bf16x4*)(c2 + n1) = x1b;
            *(bf16x4*)(c2 + n1 + 8) = x2b;
          }
          // match the eager bf16 rounding points of the reference: silu(bf16 x1) -> bf16, * bf16 x2 -> bf16
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float s = bf2f(f2bf(silu_f(bf2f(x1b[e]))));
            hsw[e] = s * bf2f(x2b[e]);
          }
          const int jh = (n1 >> 4) * 8 + (n1 & 7);  // hidden column
          *(bf16x4*)((bf16*)p.C + (size_t)mc * p.ldc + jh) = __builtin_convertvector(hsw, bf16x4);
        }
      } else if constexpr (EPI == EPI_CONV_RELU || EPI == EPI_CONV_MASK) {
        // pixel row m of the zero-bordered stack: border rows are written as zeros so the next layer's taps read padding
        const int r = m % p.conv_p, yy = r / p.conv_w2, xx = r - yy * p.conv_w2;
        const bool border = yy == 0 || yy == p.conv_h2 - 1 || xx == 0 || xx == p.conv_w2 - 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = nb + 8 * q;
          if (n >= p.N) continue;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
          if constexpr (EPI == EPI_CONV_RELU) {
            f32x4 b = *(const f32x4*)(p.bias + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = border ? 0.f : fmaxf(v[e] + b[e], 0.f);
          } else {  // input gradient: ReLU mask of the layer input (C2 = that activation; null = no ReLU in front)
            if (p.C2) {
              bf16x4 a = *(const bf16x4*)((const bf16*)p.C2 + (size_t)mc * p.ldc2 + n);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = bf2f(a[e]) > 0.f ? v[e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = border ? 0.f : v[e];
          }
          *(bf16x4*)((bf16*)p.C + (size_t)mc * p.ldc + n) = __builtin_convertvector(v, bf16x4);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = nb + 8 * q;
          if (n >= p.N) continue;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e] * p.alpha;
          if constexpr (EPI == EPI_F32_ATOMIC) {
            float* c = (float*)p.C + (size_t)mc * p.ldc + n;
#pragma unroll
            for (int e = 0; e < 4; ++e) unsafeAtomicAdd(c + e, v[e]);
          } else if constexpr (EPI == EPI_F32_SLAB) {
            // split-K partial: slice z writes its own [rows, ldc] slab with plain 16-B stores (no atomics);
            // vtp_reduce_slabs sums the slabs afterwards.  slab stride (in float4 units) travels in ldc2.
            *(f32x4*)((float*)p.C + (size_t)zslice * (size_t)p.ldc2 * 4 + (size_t)mc * p.ldc + n) = v;
          } else {
            if (p.bias) {
              f32x4 b = *(const f32x4*)(p.bias + n);
              v += b;
            }
            if constexpr (EPI == EPI_BF16) {
              *(bf16x4*)((bf16*)p.C + (size_t)mc * p.ldc + n) = __builtin_convertvector(v, bf16x4);
            } else if constexpr (EPI == EPI_GELU) {
              bf16x4 pre = __builtin_convertvector(v, bf16x4);
              if (p.C2) *(bf16x4*)((bf16*)p.C2 + (size_t)mc * p.ldc2 + n) = pre;
              f32x4 g;
#pragma unroll
              for (int e = 0; e < 4; ++e) g[e] = p.act_quick ? quick_gelu_f(bf2f(pre[e])) : gelu_erf(bf2f(pre[e]));
              *(bf16x4*)((bf16*)p.C + (size_t)mc * p.ldc + n) = __builtin_convertvector(g, bf16x4);
            } else {  // EPI_F32: out = resid + gamma * (acc + bias)
              if (p.gamma) {
                f32x4 g = *(const f32x4*)(p.gamma + n);
                v *= g;
              }
              if (p.resid) {
                f32x4 r = *(const f32x4*)(p.resid + (size_t)mc * p.ldc + n);
                v += r;
              }
              *(f32x4*)((float*)p.C + (size_t)mc * p.ldc + n) = v;
            }
          }
        }
      }
    }
  }
}

}  // namespace vtp
