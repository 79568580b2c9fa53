// 256 x 256 x 64 bf16 MFMA GEMM for gfx950 (CDNA4) with ONE wave per SIMD and a software-pipelined wave (tile configuration 10).
// NT only (forward / dgrad: C = A B^T), same LDS images, swizzles and epilogues (gemm_common.h) as gemm8p.hip; what differs is
// how the matrix pipe is kept busy.
//
// gemm8p.hip hides LDS / staging latency by ALTERNATING two wave groups per SIMD across barriers (8 barrier intervals per k-tile);
// measured there (profiles/r04_pmc_sq_cal.json): the matrix pipe is busy 70 % of the k loop, halving the number of intervals
// buys 2-3 %, LDS is 27 % busy -- the loss is the hand-over between the groups, not a resource.  Here a wave never hands over:
//   * 4 waves = 2 (M) x 2 (N), one per SIMD, each with a 128 x 128 wave tile = 4 x 4 accumulators of v_mfma_f32_32x32x16_bf16
//     (256 accumulator registers; gfx950 gives a lone wave 512: __launch_bounds__(256, 1)) -- 0.5 ds_read_b128 per MFMA instead of 0.75;
//   * a k-tile (64 deep) = 2 STEPS of 32 MFMAs (k = 32); the fragments of step s + 1 (16 ds_read_b128) are fetched into the second
//     register set WHILE step s multiplies, one read behind every second MFMA -- the wave issues MFMAs back to back and everything
//     else in their shadow (groups of { 2 MFMAs, 1 fragment read, <= 2 LDS-DMA pieces } pinned with sched_barrier);
//   * LDS: ring of TWO k-tiles (2 x 64 KiB; a k-tile = four 16-KiB NT images: A rows 0..127 | A rows 128..255 | B columns 0..127 |
//     B columns 128..255, wave w stages image w: 16 LDS-DMA pieces of 1 KiB) + 4 x 8 KiB epilogue staging = 160 KiB;
//   * ONE barrier per k-tile, in front of the odd step: by then every wave has read all fragments of k-tile T (the odd step's were
//     fetched during the even step) and k-tile T + 1 has landed (`s_waitcnt vmcnt(0)`: its pieces were issued a k-tile earlier);
//     behind it k-tile T + 2 is issued into the slot k-tile T leaves -- in the first half of the odd step, so the youngest piece has
//     ~1.5 steps to land -- and the odd step's prefetch reads the first fragments of k-tile T + 1.  The stream runs ACROSS the
//     output tiles of the persistent workgroup: the epilogue of tile i runs with k-tiles 0 and 1 of tile i + 1 in the ring.
//   * epilogue: the shared wave epilogue (gemm_common.h) twice, on the two 128 x 64 halves of the wave tile -- the instruction
//     sequence of gemm8p.hip per half, so results are BIT-IDENTICAL to tile configuration 8 (same k order per element).
// Limits (launcher): N % 256 == 0 (no column clamp in the B staging), no A row remap, no split-K / conv / timing modes.
#include "gemm_common.h"
#include "gemm4w_ktile.inc"
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace vtp {

namespace {
constexpr int W4_IMG = 16384;
constexpr int W4_KT = 4 * W4_IMG;  // one staged k-tile
constexpr int W4_RING = 2 * W4_KT;
constexpr int W4_REGION = 8192;                  // per wave: 4 KiB staging + 2 KiB second region (SwiGLU) + spare
constexpr int W4_LDS = W4_RING + 4 * W4_REGION;  // 163840 B = the CU's 160 KiB
}  // namespace

// LDS-DMA of 16 B per lane: scalar 64-bit base + 32-bit lane offset, M0 = LDS destination of the wave's 1-KiB piece (inline asm:
// hipcc must not see these loads -- their completion is counted by hand, see gemm8p.hip p8_glds16)
__device__ __forceinline__ void w4_glds16(const char* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ void w4_mfma_drain() {  // the last MFMA's 16 passes before a VALU reads its accumulator
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory");
}

// the k loop of one output tile (nk k-tiles, nk even), hand-scheduled: gemm4w_ktile.inc / tools/gen_gemm4w_ktile.py.  State carried
// from tile to tile: acc (zeroed by the caller), fl / fp = the A-lo and B fragments of the tile's first k-tile (the previous tile's last
// k-tile fetched them), cur = the staging cursor (lane offsets of the even / odd pieces of the wave's A and B shares, cc = k-tiles left
// in the cursor's output tile).
struct W4Cursor {
  unsigned pea, poa, peb, pob, cc;
};
struct W4Stage {  // wave-uniform staging constants
  const char *mata, *matb;
  unsigned stepa, stepb, kadva, kadvb, da[2], db[2];
};
__device__ __forceinline__ void w4_tile_kloop(f32x16 (&acc)[4][4], bf16x8 (&fl)[8], bf16x8 (&fp)[16], const unsigned (&adX)[2][4],
                                              const unsigned (&adY)[2][4], W4Cursor& cur, const W4Stage& st, unsigned vmaxa, unsigned vmaxb,
                                              unsigned tadva, unsigned tadvb, unsigned nk) {
  bf16x8 fh[8], fq[16];
  unsigned t0, t1, sm, sadva, sadvb, cnt = nk >> 1;
  asm volatile(W4_TILE_ASM
               : W4_TILE_OUTS, [pea] "+v"(cur.pea), [poa] "+v"(cur.poa), [peb] "+v"(cur.peb), [pob] "+v"(cur.pob), [vt0] "=&v"(t0),
                 [vt1] "=&v"(t1), [cc] "+s"(cur.cc), [cnt] "+s"(cnt), [sm] "=&s"(sm), [sadva] "=&s"(sadva), [sadvb] "=&s"(sadvb)
               : W4_TILE_ADDRS, [vmaxa] "v"(vmaxa), [vmaxb] "v"(vmaxb), [mata] "s"(st.mata), [matb] "s"(st.matb), [stepa] "s"(st.stepa),
                 [stepb] "s"(st.stepb), [kadva] "s"(st.kadva), [kadvb] "s"(st.kadvb), [tadva] "s"(tadva), [tadvb] "s"(tadvb), [nkr] "s"(nk),
                 [da0] "s"(st.da[0]), [da1] "s"(st.da[1]), [db0] "s"(st.db[0]), [db1] "s"(st.db[1])
               : "memory", "scc");
}

template <int... Is, class F>
__device__ __forceinline__ void w4_for(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}

template <int EPI, int XMODE>
__global__ __launch_bounds__(256, 1) void gemm4w_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // (LDS offset 0: the kernel has no static LDS -- the slot flip XORs bit 16)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;  // the wave's 128 x 128 sub-tile: rows wr * 128, columns wc * 128
  const int hi = lane >> 5;

  const int tiles_m = (p.M + 255) >> 8;
  const int tiles_n = (p.N + 255) >> 8;
  const int ntiles = tiles_m * tiles_n;
  const int G = gridDim.x, bx = blockIdx.x;
  const int n_my = (ntiles - bx + G - 1) / G;
  auto tile_origin = [&](int i, int& m0, int& n0) {
    int wg = bx + i * G;
    if (p.xcd_swizzle & 1) {  // bijective on [0, ntiles): XCD x owns a contiguous chunk of the tile list
      const int q = ntiles >> 3, r = ntiles & 7, x = wg & 7;
      wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (wg >> 3);
    }
    n0 = (wg % tiles_n) << 8;
    m0 = (wg / tiles_n) << 8;
  };
  const int nk = p.K >> 6;  // (K % 128 == 0: launcher)

  // ---------------------------------------------------------------- staging (LDS-DMA) side
  // a staged k-tile = four 16-KiB NT images (A rows 0..127 | A rows 128..255 | B columns 0..127 | B columns 128..255 of the tile; 128-B
  // rows, 16-B chunk index XOR ((row >> 1) & 7)).  Wave w stages rows (w & 1) * 64 + [0, 64) of A image w >> 1 AND of B image w >> 1: 8
  // LDS-DMA pieces of 1 KiB each; piece i = image rows (w & 1) * 64 + 8 i + prow, the lane's chunk lane & 7 holds source chunk
  // (lane & 7) ^ (prow >> 1) ^ 4 (i & 1).  Source = matrix base (scalar) + 32-bit lane offset (gemm8p_fits: the operand spans < 4 GiB)
  // = tile row offset + k-tile * 128 B + i * step8 + [prow * ld * 2 + chunk * 16], clamped to the last 16 bytes of the operand's last
  // row (rows beyond the matrix read valid memory; they only feed outputs the epilogue masks).
  const int prow = lane >> 3;
  const int share = (wave >> 1) * 128 + (wave & 1) * 64;  // first tile row / column of the wave's shares
  const unsigned step8a = (unsigned)(8 * p.lda * 2), step8b = (unsigned)(8 * p.ldb * 2);
  const unsigned offa_e = (unsigned)((prow * p.lda + (((lane & 7) ^ (prow >> 1)) << 3)) * 2);
  const unsigned offa_o = (unsigned)((prow * p.lda + (((lane & 7) ^ (prow >> 1) ^ 4) << 3)) * 2) + step8a;
  const unsigned offb_e = (unsigned)((prow * p.ldb + (((lane & 7) ^ (prow >> 1)) << 3)) * 2);
  const unsigned offb_o = (unsigned)((prow * p.ldb + (((lane & 7) ^ (prow >> 1) ^ 4) << 3)) * 2) + step8b;
  const unsigned vmaxa = (unsigned)(((size_t)(p.M - 1) * p.lda + p.K - 8) * 2), vmaxb = (unsigned)(((size_t)(p.N - 1) * p.ldb + p.K - 8) * 2);
  auto tile_off = [&](int i, unsigned& oa, unsigned& ob) {  // byte offsets of the first rows of the wave's shares in my tile i (past my
    int m0s, n0s;                                           // last tile: stay on it)
    tile_origin(i < n_my ? i : n_my - 1, m0s, n0s);
    oa = (unsigned)(m0s + share) * (unsigned)(p.lda * 2);
    ob = (unsigned)(n0s + share) * (unsigned)(p.ldb * 2);
  };
  W4Stage st;
  st.mata = (const char*)p.A;
  st.matb = (const char*)p.B;
  st.stepa = 2 * step8a;
  st.stepb = 2 * step8b;
  st.kadva = 128u - 8u * step8a;
  st.kadvb = 128u - 8u * step8b;
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    st.da[sl] = (unsigned)(size_t)smem + sl * W4_KT + (wave >> 1) * W4_IMG + (wave & 1) * 8192;
    st.db[sl] = st.da[sl] + 2 * W4_IMG;
  }
  // the cursor: my-tile index s_i, k-tile s_c inside it; after the stream's last k-tile it wraps to k-tile 0 of the last tile (the
  // surplus pieces land in a slot nobody reads any more)
  int s_i = 0, s_c = 0;
  auto issue_ktile = [&](int slot) {  // prologue: the wave's 16 pieces of the cursor's k-tile at once, then the cursor moves on
    unsigned oa, ob;
    tile_off(s_i, oa, ob);
    w4_for(std::make_integer_sequence<int, 8>{}, [&](auto it) {
      constexpr int I = decltype(it)::value;
      w4_glds16(st.mata, min(oa + s_c * 128 + (I & ~1) * step8a + ((I & 1) ? offa_o : offa_e), vmaxa), st.da[slot] + I * 1024);
      w4_glds16(st.matb, min(ob + s_c * 128 + (I & ~1) * step8b + ((I & 1) ? offb_o : offb_e), vmaxb), st.db[slot] + I * 1024);
    });
    if (++s_c == nk) {
      s_c = 0;
      ++s_i;
    }
  };

  // ---------------------------------------------------------------- fragment (LDS read) side
  bf16x8 fl[8], fp[16];  // A-lo fragments [ks * 2 + row block 0 | 1], B fragments [ks * 4 + column block]: see tools/gen_gemm4w_ktile.py
  f32x16 acc[4][4];      // [column block][row block]
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  zero_acc();

  // ---------------------------------------------------------------- prologue: k-tiles 0 and 1 of the stream staged, the first fragments read
  issue_ktile(0);
  issue_ktile(1);
  W4Cursor cur;
  {
    unsigned oa, ob;
    tile_off(s_i, oa, ob);
    cur.pea = oa + s_c * 128 + offa_e;
    cur.poa = oa + s_c * 128 + offa_o;
    cur.peb = ob + s_c * 128 + offb_e;
    cur.pob = ob + s_c * 128 + offb_o;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  for (int ti = 0; ti < n_my; ++ti) {
    int m0, n0;
    tile_origin(ti, m0, n0);
    // A-lo and B fragments of the tile's first k-tile (ring slot 0: nk is even).  The previous tile's last k-tile fetched them too, but
    // carrying 96 registers through the epilogue costs spills (and a drain of every store in flight at each reload): read them again.
    // (everything derived from the lane id is recomputed per tile from an opaque copy: hoisted out of the tile loop it would live
    // through the epilogue, spill, and every reload drains the stores in flight -- see gemm_common.h gemm_epilogue)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int sw = (ln >> 1) & 7, rowoff = (ln & 31) * 128, hi2 = ln >> 5;
    // LDS byte addresses of the lane's 16-B chunk of k-step ks in row block 0 of the wave's A (X) / B (Y) image, per ring slot
    unsigned adX[2][4], adY[2][4];
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const unsigned lo = rowoff + (((2 * ks + hi2) ^ sw) << 4);
        adX[sl][ks] = (unsigned)(size_t)smem + sl * W4_KT + wr * W4_IMG + lo;
        adY[sl][ks] = (unsigned)(size_t)smem + sl * W4_KT + (2 + wc) * W4_IMG + lo;
      }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int lo = rowoff + (((2 * ks + hi2) ^ sw) << 4);
#pragma unroll
      for (int b = 0; b < 2; ++b) fl[ks * 2 + b] = *(const bf16x8*)(smem + wr * W4_IMG + b * 4096 + lo);
#pragma unroll
      for (int i = 0; i < 4; ++i) fp[ks * 4 + i] = *(const bf16x8*)(smem + (2 + wc) * W4_IMG + i * 4096 + lo);
    }
    // the cursor leaves its tile exactly once during the nk k-tiles of this loop: the jump to the first k-tile of the next one
    cur.cc = (unsigned)(nk - s_c);
    unsigned oa0, ob0, oa1, ob1;
    tile_off(s_i, oa0, ob0);
    tile_off(s_i + 1, oa1, ob1);
    const unsigned back = (unsigned)(nk - 1) * 128u;
    w4_tile_kloop(acc, fl, fp, adX, adY, cur, st, vmaxa, vmaxb, oa1 - oa0 - back - 8u * step8a, ob1 - ob0 - back - 8u * step8b, (unsigned)nk);
    if (s_i < n_my) ++s_i;
    w4_mfma_drain();
    char* reg = gemm_epilogue_uses_lds<EPI, false, 64, 4096>(p) ? smem + W4_RING + wave * W4_REGION : nullptr;
    char* reg2 = (EPI == EPI_SWIGLU && reg) ? reg + 4096 : nullptr;
#if !defined(W4_DIAG) || W4_DIAG < 4  // (4: timing experiment without the epilogues)
    if constexpr (EPI == EPI_BF16 && XMODE != 2) {
      // the wave's whole 128 x 128 sub-tile in one pass of 256-B row images (8 KiB region): half as many LDS round trips on the single
      // wave a SIMD has here as two 128 x 64 passes (element-wise the same arithmetic: still bit-identical to configuration 8)
      char* reg8 = gemm_epilogue_uses_lds<EPI, false, 128, W4_REGION>(p) ? smem + W4_RING + wave * W4_REGION : nullptr;
      gemm_epilogue<EPI, false, 128, 128, W4_REGION, XMODE>(p, acc, reg8, m0, n0, wr, wc, lane, 0, nullptr);
    } else {
      gemm_epilogue<EPI, false, 128, 64, 4096, XMODE>(p, *(f32x16(*)[2][4]) & acc[0], reg, m0, n0, wr, wc * 2, lane, 0, reg2);
      gemm_epilogue<EPI, false, 128, 64, 4096, XMODE>(p, *(f32x16(*)[2][4]) & acc[2], reg, m0, n0, wr, wc * 2 + 1, lane, 0, reg2);
    }
#else
    if (p.M < 0) gemm_epilogue<EPI, false, 128, 64, 4096, XMODE>(p, *(f32x16(*)[2][4]) & acc[0], reg, m0, n0, wr, wc * 2, lane, 0, reg2);
#endif
    zero_acc();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // surplus pieces of the stream's tail
}

template <int EPI, int XMODE = 0>
static int launch4w(const GemmArgs& a, hipStream_t s) {
  auto kern = gemm4w_kernel<EPI, XMODE>;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS);
    attr_set = true;
  }
  static int slots = 0;
  if (!slots) {
    int dev = 0;
    hipDeviceProp_t prop;
    (void)hipGetDevice(&dev);
    (void)hipGetDeviceProperties(&prop, dev);
    slots = prop.multiProcessorCount - prop.multiProcessorCount % 8;
    if (slots < 8) slots = 8;
  }
  const int ntiles = cdiv(a.M, 256) * cdiv(a.N, 256);
  hipLaunchKernelGGL(kern, dim3(ntiles > slots ? slots : ntiles), dim3(256), W4_LDS, s, a);
  return check_launch("gemm4w_nt");
}

// entry point used by the dispatcher of gemm.hip (tile configuration 10); the caller has checked N % 256 == 0 (the B staging has no
// column clamp), a_grp == 0, splits == 1 and no conv / timing mode
int launch_gemm4w_nt(const GemmArgs& a, int epi, hipStream_t s) {
  switch (epi) {
    case EPI_BF16:
      if (a.rope_pos) return launch4w<EPI_BF16, 1>(a, s);
      if (a.swiglu_pre) return launch4w<EPI_BF16, 2>(a, s);
      return launch4w<EPI_BF16, 0>(a, s);
    case EPI_F32: return launch4w<EPI_F32>(a, s);
    case EPI_SWIGLU: return launch4w<EPI_SWIGLU>(a, s);
    case EPI_GELU: return launch4w<EPI_GELU>(a, s);
    default: set_error("gemm4w: unsupported epilogue %d", epi); return VTP_ERR_ARG;
  }
}

}  // namespace vtp
