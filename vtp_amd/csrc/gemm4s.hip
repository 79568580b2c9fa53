// EXPERIMENT (round 5; tile configuration 11, reached only through vtp_set_gemm_tuning(11, ..) -- tools/gemm4s_probe.py): the K = 768
// NT GEMM with a SPLIT ACCUMULATOR SET, one wave per SIMD, so that the epilogue of output tile i runs inside the k loop of tile i + 1.
//
// Why: the K = 768 shapes of the step (qkv, SwiGLU forward, fp32-residual projections, w3 dgrad: ~19 of 35 GEMM-ms) spend as long in
// their epilogue as in their 12-k-tile main loop, with the matrix pipe idle (26-30 % busy).  The 256 x 256
// kernels fill the register file with ONE tile's accumulators; here a wave owns a 128 x 64 sub-tile (128 accumulator registers) twice:
// set A accumulates tile i + 1 while set B -- tile i, finished -- is converted and stored, one 32 x 32 block per k-tile, between the
// MFMAs.  Written with builtin MFMAs and explicit program order (sched_barrier after every MFMA pair) rather than generated assembly,
// to find out what the compiler-scheduled form of this structure reaches before an assembly generator is written for it.
//
//   * workgroup = 4 waves = 2 (M) x 2 (N), tile 256 x 128, one workgroup per CU (__launch_bounds__(256, 1): 512 registers per wave);
//   * a staged k-tile (64 deep) = three 16-KiB images (A rows 0..127 | A rows 128..255 | B columns 0..127; 128-B rows, 16-B chunk index
//     XOR ((row >> 1) & 7): the layout of gemm4w.hip); ring of THREE k-tiles = 144 KiB, staged two k-tiles ahead by LDS-DMA, the
//     staging cursor running across the tiles of the persistent workgroup; one wait + one barrier per k-tile;
//   * K = 768 only (12 k-tiles, fully unrolled: ring slot, `vmcnt` immediates and the epilogue block of every k-tile are compile-time);
//     N % 128 == 0; plain bf16 epilogue (+ bias), direct 8-byte stores through a buffer resource (rows beyond M are dropped by the
//     range check: no branches in the unrolled body);
//   * element-wise the arithmetic and k order of the other kernels: results are bit-identical to tile configuration 8.
#include "gemm_common.h"
#include <type_traits>
#include <utility>

namespace vtp {

namespace {
constexpr int S4_IMG = 16384;
constexpr int S4_KT = 3 * S4_IMG;   // one staged k-tile
constexpr int S4_NSLOT = 3;
constexpr int S4_STAGE = 4096;               // per wave: one 32-row x 128-B slab of the finished tile
constexpr int S4_LDS = S4_NSLOT * S4_KT + 4 * S4_STAGE;  // 163840 B = the CU's 160 KiB
constexpr int S4_NK = 12;           // K = 768
}  // namespace

__device__ __forceinline__ void s4_glds16(const char* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}
template <int N>
__device__ __forceinline__ void s4_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int... Is, class F>
__device__ __forceinline__ void s4_for(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}

// stores the epilogue of a finished tile issues inside k-tile KT of the next one: one 32 x 32 block (4 stores per lane) in k-tiles 1 .. 8
__host__ __device__ constexpr int s4_stores(int kt) { return (kt >= 1 && kt <= 8) ? 4 : 0; }

// The wait in front of k-tile KT: `vmcnt(12)`.  The pieces of k-tile KT were issued during k-tile KT - 2; younger than its last piece are
// the 12 pieces of k-tile KT + 1 (issued during KT - 1) and that k-tile's epilogue stores.  gfx950 counts loads and stores in ONE vmcnt;
// waiting for "at most 12 outstanding" is correct whether the stores retire in order with the loads or ahead of them (the 12 younger
// LOADS retire in order behind the piece we need), and at worst also waits for the first pieces of k-tile KT + 1.
constexpr int S4_WAIT = 12;

__global__ __launch_bounds__(256, 1) void gemm4s_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;  // the wave's 128 x 64 sub-tile: rows wr * 128, columns wc * 64
  const int hi = lane >> 5, r32 = lane & 31;

  const int tiles_n = p.N >> 7;
  const int ntiles = ((p.M + 255) >> 8) * tiles_n;
  const int G = gridDim.x, bx = blockIdx.x;
  const int n_my = (ntiles - bx + G - 1) / G;
  auto tile_origin = [&](int i, int& m0, int& n0) {
    int wg = bx + i * G;
    if (p.xcd_swizzle & 1) {
      const int q = ntiles >> 3, r = ntiles & 7, x = wg & 7;
      wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (wg >> 3);
    }
    n0 = (wg % tiles_n) << 7;
    m0 = (wg / tiles_n) << 8;
  };

  // ---------------------------------------------------------------- staging: 48 pieces of 1 KiB per k-tile, 12 per wave
  // A pieces: image rows 8 pa + prow of the 256 tile rows, pa = wave * 8 + (0..7); B pieces: rows 8 pb + prow of the 128 tile columns,
  // pb = wave * 4 + (0..3).  The lane's 16-B slot (lane & 7) holds source chunk (lane & 7) ^ ((row >> 1) & 7), row = 8 piece + prow.
  const int prow = lane >> 3;
  const unsigned vmaxa = (unsigned)(((size_t)(p.M - 1) * p.lda + p.K - 8) * 2), vmaxb = (unsigned)(((size_t)(p.N - 1) * p.ldb + p.K - 8) * 2);
  const char* mata = (const char*)p.A;
  const char* matb = (const char*)p.B;
  unsigned offa[8], offb[4], dsta[8], dstb[4];  // lane offsets of the source (VGPR) | wave-uniform LDS destinations (SGPR: they go to M0)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int pa = wave * 8 + i, row = pa * 8 + prow;  // piece 0 .. 31, tile row 0 .. 255
    offa[i] = (unsigned)((row * p.lda + (((lane & 7) ^ ((row >> 1) & 7)) << 3)) * 2);
    dsta[i] = (unsigned)((pa >> 4) * S4_IMG + (pa & 15) * 1024);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pb = wave * 4 + i, row = pb * 8 + prow;  // piece 0 .. 15, tile column 0 .. 127
    offb[i] = (unsigned)((row * p.ldb + (((lane & 7) ^ ((row >> 1) & 7)) << 3)) * 2);
    dstb[i] = (unsigned)(2 * S4_IMG + pb * 1024);
  }
  const unsigned lds0 = (unsigned)(size_t)smem;
  // the staging stream: k-tile s_c of my tile s_i (past my last tile: stay on it -- the surplus pieces land in slots nobody reads)
  int s_i = 0, s_c = 0;
  unsigned s_oa = 0, s_ob = 0;  // byte offsets of the cursor's tile origin rows in A / B
  auto cursor_tile = [&]() {
    int m0s, n0s;
    tile_origin(s_i < n_my ? s_i : n_my - 1, m0s, n0s);
    s_oa = (unsigned)m0s * (unsigned)(p.lda * 2);
    s_ob = (unsigned)n0s * (unsigned)(p.ldb * 2);
  };
  cursor_tile();
  auto advance = [&]() {
    if (++s_c == S4_NK) {
      s_c = 0;
      ++s_i;
      cursor_tile();
    }
  };
  auto issue_piece_a = [&](int i, int slot) { s4_glds16(mata, min(s_oa + s_c * 128 + offa[i], vmaxa), lds0 + slot * S4_KT + dsta[i]); };
  auto issue_piece_b = [&](int i, int slot) { s4_glds16(matb, min(s_ob + s_c * 128 + offb[i], vmaxb), lds0 + slot * S4_KT + dstb[i]); };

  // ---------------------------------------------------------------- fragments
  bf16x8 fa[2][4], fb[2][2];  // [k-step parity][block]; they live across k-tiles: with S4_TWOBAR the last k-step of a k-tile fetches the next one's first
  const unsigned frow = (unsigned)(r32 * 128), fsw = (unsigned)((lane >> 1) & 7);
  auto frag_off = [&](int ks) { return frow + ((((unsigned)(2 * ks + hi)) ^ fsw) << 4); };
  // single fragment reads (S4_INTERLEAVE places them one by one between the MFMAs)
  auto read_a = [&](int slot, int ks, int par, int j) {
    fa[par][j] = *(const bf16x8*)(smem + slot * S4_KT + wr * S4_IMG + j * 4096 + frag_off(ks));
  };
  auto read_b = [&](int slot, int ks, int par, int i) {
    fb[par][i] = *(const bf16x8*)(smem + slot * S4_KT + 2 * S4_IMG + wc * 8192 + i * 4096 + frag_off(ks));
  };
  f32x16 accA[2][4], accB[2][4];  // [column block][row block]
  auto zero = [&](f32x16 (&a)[2][4]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) a[i][j][e] = 0.f;
  };
  zero(accA);  // (the loop overwrites both sets before it reads them: these only give the arrays a defined value)
  zero(accB);

  // output: rows beyond M fall outside the buffer resource's range and are dropped
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)((size_t)p.M * p.ldc * 2), 0x00020000);

  // quarter q of the epilogue of block (bi, bj) of the finished tile whose lane base offset is `obase` (bytes: row m0 + wr 128 + r32,
  // column n0 + wc 64 + 4 hi): four accumulator registers -> bf16 -> one 8-byte store.  The accumulator reads are `asm volatile`:
  // left to itself hipcc converts the whole finished tile right behind its last MFMA, cannot hold the 64 results next to the k loop's
  // fragments, spills them and reloads them inside the loop -- with `s_waitcnt vmcnt(0)`, i.e. a drain of the LDS-DMA in flight, in
  // front of every store (first build of this file: 142 spilled registers, 16 scratch reloads per k-tile).  The row / column deltas of
  // the block go into the store's scalar and immediate offsets, so one address VGPR serves the whole tile.  (No bias in this experiment:
  // a global load inside the overlapped epilogue would be waited for behind the LDS-DMA pieces in flight.)
  auto epi_quarter = [&](const f32x16& a, int bi, int bj, int q, unsigned obase, unsigned rowstep32) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[e]) : "a"(a[4 * q + e]));
    const bf16x4 o = __builtin_convertvector(v, bf16x4);
#if defined(S4_DIAG) && S4_DIAG == 2
    if (p.M < 0)
#endif
    __builtin_amdgcn_raw_buffer_store_b64(*(const u32x2_t*)&o, rsrc, obase + (unsigned)((bi * 32 + 8 * q) * 2), bj * rowstep32, 0);
  };
  auto lane_base = [&](int m0, int n0) { return (unsigned)(((size_t)(m0 + wr * 128 + r32) * p.ldc + n0 + wc * 64 + 4 * hi) * 2); };
  const unsigned rowstep32 = (unsigned)(32 * p.ldc * 2);
  // LDS-staged variant (S4_STAGED): a 32-row x 64-column slab (row block bj, both column blocks) goes through the wave's private 4-KiB
  // region -- 8 ds_write_b64 per lane in one k-tile, 4 ds_read_b128 + 4 full-line 16-byte stores in the next -- instead of 8-byte
  // pieces of 32 rows per store instruction.  16-B chunk index XOR (row & 7): conflict-free writes and reads.
  char* stage = smem + S4_NSLOT * S4_KT + wave * S4_STAGE;
  auto stage_write = [&](const f32x16& a, int bi, int q) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[e]) : "a"(a[4 * q + e]));
    *(bf16x4*)(stage + r32 * 128 + (((bi * 4 + q) ^ (r32 & 7)) << 4) + hi * 8) = __builtin_convertvector(v, bf16x4);
  };
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
  const int srow = lane >> 3, schunk = lane & 7;  // read-back: item t = rows 8 t + srow, 16-B chunk schunk
  auto stage_read = [&](int t) { return *(const u32x4_t*)(stage + (t * 8 + srow) * 128 + ((schunk ^ ((t * 8 + srow) & 7)) << 4)); };
  auto lane_base16 = [&](int m0, int n0) { return (unsigned)(((size_t)(m0 + wr * 128 + srow) * p.ldc + n0 + wc * 64 + schunk * 8) * 2); };
  const unsigned rowstep8 = (unsigned)(8 * p.ldc * 2);
  auto stage_store = [&](u32x4_t v, int bj, int t, unsigned obase16) {
#if defined(S4_DIAG) && S4_DIAG == 2
    if (p.M < 0)
#endif
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, obase16, (bj * 4 + t) * rowstep8, 0);
  };

  auto read_frags = [&](int slot, int ks, int par) {
    const char* abase = smem + slot * S4_KT + wr * S4_IMG;
    const char* bbase = smem + slot * S4_KT + 2 * S4_IMG + wc * 8192;
    const unsigned o = frag_off(ks);
#pragma unroll
    for (int j = 0; j < 4; ++j) fa[par][j] = *(const bf16x8*)(abase + j * 4096 + o);
#pragma unroll
    for (int i = 0; i < 2; ++i) fb[par][i] = *(const bf16x8*)(bbase + i * 4096 + o);
  };

  // one output tile's k loop into `cur`, with the epilogue of the previous tile (`prev`, origin pm0 / pn0) inside it
  auto tile_loop = [&](auto have_prev_c, f32x16 (&cur)[2][4], f32x16 (&prev)[2][4], int pm0, int pn0) {
    constexpr bool HAVE_PREV = decltype(have_prev_c)::value;
    const unsigned obase = lane_base(pm0, pn0);
    const unsigned obase16 = lane_base16(pm0, pn0);
    (void)obase; (void)obase16;
    s4_for(std::make_integer_sequence<int, S4_NK>{}, [&](auto ktc) {
      constexpr int KT = decltype(ktc)::value;
      constexpr int SLOT = KT % S4_NSLOT, NEXT = (KT + 2) % S4_NSLOT;
#ifndef S4_TWOBAR
      // k-tile KT has landed (this wave's pieces), then everybody's; the slot of k-tile KT - 1 is free for k-tile KT + 2
      s4_wait_vm<S4_WAIT>();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      read_frags(SLOT, 0, 0);
#else
      // TWO barriers per k-tile (S4_TWOBAR).  B1, here: every wave has finished reading k-tile KT - 1, so its slot may be refilled with
      // k-tile KT + 2 -- no wait: that k-tile KT has landed was certified by B2 of the previous k-tile, whose last k-step already
      // fetched this one's first fragments (no LDS latency exposed behind the barrier).  B2, in front of k-step 2: k-tile KT + 1 has
      // landed everywhere -- it was issued during k-tile KT - 1; younger than its last piece are the 6 pieces of k-tile KT + 2 issued in
      // k-steps 0 and 1 (the staged epilogue stores sit in k-step 2: none in between) -> vmcnt(6).
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#endif
#ifdef S4_STAGED
      u32x4_t sv[4];
#endif
      s4_for(std::make_integer_sequence<int, 4>{}, [&](auto ksc) {
        constexpr int KS = decltype(ksc)::value;
        constexpr int PAR = KS & 1;
#ifdef S4_TWOBAR
        if constexpr (KS == 2) {
          s4_wait_vm<6>();
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
        }
        constexpr bool HAS_NEXT = true;
        constexpr int NSLOT_R = KS < 3 ? SLOT : (KT + 1) % S4_NSLOT, NKS = KS < 3 ? KS + 1 : 0;  // ks 3 fetches the next k-tile's first fragments
#else
        constexpr bool HAS_NEXT = KS < 3;
        constexpr int NSLOT_R = SLOT, NKS = KS + 1;
#endif
#ifndef S4_INTERLEAVE
        if constexpr (HAS_NEXT) read_frags(NSLOT_R, NKS, PAR ^ 1);
        // the k-tile's 12 LDS-DMA pieces: three per k-step (A pieces 2 KS, 2 KS + 1, B piece KS)
#if !defined(S4_DIAG) || S4_DIAG != 1  // (S4_DIAG: timing experiments, WRONG results -- 1 = no staging inside the loop, 2 = no epilogue stores)
        issue_piece_a(2 * KS, NEXT);
        issue_piece_a(2 * KS + 1, NEXT);
        issue_piece_b(KS, NEXT);
#endif
#endif
        __builtin_amdgcn_sched_barrier(0);
        // the tile's first k-step accumulates onto the constant zero (no zeroing pass), its last one writes the finished sums INTO THE
        // OTHER SET (`prev`, whose epilogue ended in k-tile 8): the sets never change roles, so there is one loop body and no register
        // hand-over -- a ping-pong of two bodies made hipcc park one set in scratch (96 spilled registers, reloaded one by one in the loop)
        constexpr bool FIRST = KT == 0 && KS == 0, LAST = KT == S4_NK - 1 && KS == 3;
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const f32x16 r = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[PAR][i], fa[PAR][j], FIRST ? zero16 : cur[i][j], 0, 0, 0);
            if constexpr (LAST) prev[i][j] = r;
            else cur[i][j] = r;
#ifdef S4_INTERLEAVE
            // S4_INTERLEAVE: the next k-step's fragment reads and this k-step's three LDS-DMA pieces go BETWEEN the MFMAs (the wave
            // issues in order: instructions placed in front of a block of eight MFMAs run with the matrix pipe idle -- the first build
            // spent ~150 of every ~400 cycles of a k-step that way)
            if (i == 0) {
              if constexpr (HAS_NEXT) read_a(NSLOT_R, NKS, PAR ^ 1, j);
            } else {
              if constexpr (HAS_NEXT) {
                if (j == 2) read_b(NSLOT_R, NKS, PAR ^ 1, 0);
                if (j == 3) read_b(NSLOT_R, NKS, PAR ^ 1, 1);
              }
#if !defined(S4_DIAG) || S4_DIAG != 1
              if (j == 0) issue_piece_a(2 * KS, NEXT);
              if (j == 1) issue_piece_a(2 * KS + 1, NEXT);
              if (j == 2) issue_piece_b(KS, NEXT);
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
#endif
          }
#ifndef S4_STAGED
          if constexpr (HAVE_PREV && s4_stores(KT) > 0) {
            // one quarter of the previous tile's block KT - 1 in the shadow of each MFMA pair of k-step 1 (its 4 stores = s4_stores)
            if constexpr (KS == 1) {
              constexpr int B = KT - 1, BI = B >> 2, BJ = B & 3;
              epi_quarter(prev[BI][BJ], BI, BJ, j, obase, rowstep32);
            }
          }
#else
          if constexpr (HAVE_PREV && KT >= 1 && KT <= 8) {
            constexpr int BJ = (KT - 1) >> 1;
            if constexpr (((KT - 1) & 1) == 0) {  // k-tile 2 BJ + 1: the slab's 8 quarter-blocks into LDS, one per MFMA pair of k-steps 1 and 2
              if constexpr (KS == 1) stage_write(prev[0][BJ], 0, j);
              if constexpr (KS == 2) stage_write(prev[1][BJ], 1, j);
            } else {                              // k-tile 2 BJ + 2: read back in k-step 1, full-line stores in k-step 2
              if constexpr (KS == 1) sv[j] = stage_read(j);
              if constexpr (KS == 2) stage_store(sv[j], BJ, j, obase16);
            }
          }
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      advance();
    });
  };

  // ---------------------------------------------------------------- prologue: k-tiles 0 and 1 of the stream
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_piece_a(i, kt);
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_piece_b(i, kt);
    advance();
  }
  __builtin_amdgcn_sched_barrier(0);
#ifdef S4_TWOBAR
  s4_wait_vm<12>();  // k-tile 0 of the stream has landed (k-tile 1's 12 pieces may still be in flight)
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  read_frags(0, 0, 0);
#endif

  int pm0 = 0, pn0 = 0;
  tile_loop(std::false_type{}, accA, accB, 0, 0);  // (n_my >= 1: the grid never exceeds the tile count)
  tile_origin(0, pm0, pn0);
  for (int ti = 1; ti < n_my; ++ti) {
    tile_loop(std::true_type{}, accA, accB, pm0, pn0);  // tile ti accumulates in A; tile ti - 1 leaves from B inside the loop
    tile_origin(ti, pm0, pn0);
  }
  // the last tile's epilogue has no k loop to hide in
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory");
  {
    const unsigned obase = lane_base(pm0, pn0);
    const unsigned obase16 = lane_base16(pm0, pn0);
    (void)obase; (void)obase16;
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        epi_quarter(accB[b >> 2][b & 3], b >> 2, b & 3, q, obase, rowstep32);
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// entry point used by the dispatcher of gemm.hip (tile configuration 11: experiment).  Returns VTP_ERR_ARG for shapes it does not take.
int launch_gemm4s_nt(const GemmArgs& a, int epi, hipStream_t s) {
  if (epi != EPI_BF16 || a.bias || a.alpha != 1.f || a.rope_pos || a.swiglu_pre || a.a_grp || a.c_grp || a.conv_cin || a.K != 64 * S4_NK || a.N % 128 != 0 ||
      (size_t)a.M * a.ldc * 2 >= (1ull << 31) || (size_t)a.M * a.lda * 2 >= (1ull << 32)) {
    set_error("gemm4s (experiment): plain bf16 epilogue without bias, K = 768, N %% 128 == 0 only");
    return VTP_ERR_ARG;
  }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm4s_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S4_LDS);
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
  const int ntiles = cdiv(a.M, 256) * (a.N / 128);
  hipLaunchKernelGGL(gemm4s_kernel, dim3(ntiles > slots ? slots : ntiles), dim3(256), S4_LDS, s, a);
  return check_launch("gemm4s_nt");
}

}  // namespace vtp
