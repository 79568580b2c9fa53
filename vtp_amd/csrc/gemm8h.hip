// 128 x 256 x 64 "half-size" variant of the 8-phase bf16 MFMA GEMM main loop for gfx950 (CDNA4): TWO co-resident workgroups per CU
// (tile configuration 9).  NT only (forward / dgrad: C = A B^T), same operand layouts, swizzles and epilogues (gemm_common.h) as
// gemm8p.hip; what differs is who overlaps with whom.
//
// Why: in the 256 x 256 kernel (gemm8p.hip) one workgroup owns the CU (160 KiB LDS, 2 x 256 VGPRs per SIMD), so a tile's epilogue
// -- 3 .. 25 us of VALU / LDS / store work at K = 768 against a ~19 us k loop -- runs with the matrix pipe idle, and the k loop runs
// with the HBM store path idle.  Here a workgroup is 4 waves (one per SIMD) with the SAME 128 x 64 wave tile (128 accumulator
// VGPRs, <= 256 in all) and 80 KiB of LDS, so two workgroups share a CU and are scheduled independently: one's epilogue (and
// its tile-boundary bubbles) overlaps the other's k loop, and a launch has twice as many, half as large tiles (finer quantisation
// on 256 CUs).  Price: an A panel is shared by 4 instead of 8 waves -- 48 KiB staged per 128 x 256 x 64 k-tile (85 FLOP/B from L2
// instead of 128) and 12 instead of 8 LDS-DMA pieces per wave and k-tile.
//
//   * 4 waves = 1 (M) x 4 (N); wave tile 128 x 64 = 2 x 4 accumulators of v_mfma_f32_32x32x16_bf16.
//   * a k-tile (64 deep) is staged as THREE 16-KiB slots in the order they are consumed:
//       j = 0  B-first   columns wc*64 + [0,32)  of every wave column wc      image [128 rows][64 k]
//       j = 1  A         rows [0,128) of the tile (rows 0..63 are read in phase 0, rows 64..127 in phase 2)
//       j = 2  B-second  columns wc*64 + [32,64)
//     (128-B rows, 16-B chunk index XOR ((row >> 1) & 7): the NT image of gemm8p.hip).  Each wave issues 4 LDS-DMA pieces
//     (global_load_lds_dwordx4, 1 KiB) per slot.
//   * ring of FOUR slots (64 KiB); slot roles: B-second always lives in slot 2; B-first / A / spare rotate over slots {0, 1, 3}
//     with period 3 k-tiles (T % 3 = 0: B1 -> 0, A -> 1 | 1: B1 -> 3, A -> 0 | 2: B1 -> 1, A -> 3).  The staging cursor issues
//       phase 1 of k-tile T:  A(T+1)   into the slot B1(T) left in phase 0
//       phase 2            :  B2(T+1)  into slot 2 (B2(T) was read in phase 1)
//       phase 3            :  B1(T+2)  into the slot A(T) left in phase 2
//     so every slot has three phases (~0.75 k-tile) between issue and first read, ACROSS output tiles (persistent workgroups).
//   * one k-tile = 4 phases; phase = { fragment reads | LDS-DMA issue | 8 MFMAs (one 64 x 32 quadrant x k = 64) | counted vmcnt |
//     s_barrier }.  ONE barrier per phase: a slot is overwritten in the phase after its last read, behind the barrier that ends
//     the reading phase; a slot is read in the phase after the `s_waitcnt vmcnt(4 | 8)` + barrier that retires its pieces.
//     There is no second wave group to alternate with inside the workgroup: the partner on each SIMD is the OTHER workgroup's
//     wave, unsynchronised (MFMA segments run at raised priority).
//   * 16 KiB of dedicated epilogue staging (4 KiB per wave): 80 KiB per workgroup, two per CU.
//   Measured (profiles/r04_gemm8h_bench.log): two resident workgroups per CU (occupancy query: 2) run the 34144 x 2304 x 768 bf16 GEMM
//   in 138 us, one per CU in 188 us, the 256 x 256 kernel in 129 us: the barrier-enforced alternation of gemm8p.hip's two wave
//   groups uses the matrix pipe ~10 % better than two free-running workgroups, so this kernel is dispatched only where its
//   epilogue overlap / finer tail pays (gemm.hip use_8h_nt).  A variant with the MFMAs of a quadrant one phase BEHIND its
//   fragment reads (the reads complete under the previous quadrant's MFMAs, B-first double-buffered in registers) measured
//   2-3 % SLOWER on every shape (fp32-residual epilogue: spills) and was dropped: LDS read latency is not what a lone workgroup waits for.
#include "gemm_common.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace vtp {

namespace {
constexpr int H8_SLOT = 16384;
constexpr int H8_RING = 4 * H8_SLOT;
constexpr int H8_REGION = 4096;
constexpr int H8_LDS = H8_RING + 4 * H8_REGION;  // 81920 B: two workgroups fill a CU's 160 KiB
}  // namespace

// LDS-DMA of 16 B per lane: scalar 64-bit base + 32-bit lane offset, M0 = LDS destination of the wave's 1-KiB piece (inline asm: see
// gemm8p.hip p8_glds16 -- hipcc must not see these loads, their completion is counted by hand)
// the wave's FOUR pieces of a slot in one statement: M0 is saved / restored once and stepped by 1 KiB between the pieces
__device__ __forceinline__ void h8_glds16x4(const char* sbase, unsigned v0, unsigned v1, unsigned v2, unsigned v3, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(sbase), "s"(lds_dst)
      : "memory", "scc");
}
// B slots (N % 256 == 0: no column clamp): the four pieces differ in a wave-uniform row offset (-> four scalar bases) and in the XOR
// swizzle of the k chunk only by the parity of the piece (-> two lane offsets)
__device__ __forceinline__ void h8_glds16x4b(const char* b0, const char* b1, const char* b2, const char* b3, unsigned v_even,
                                             unsigned v_odd, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %6\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(v_even), "v"(v_odd), "s"(b0), "s"(b1), "s"(b2), "s"(b3), "s"(lds_dst)
      : "memory", "scc");
}
__device__ __forceinline__ void h8_glds16_v(const char* vaddr, unsigned lds_dst) {  // per-lane 64-bit source (K tails)
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(vaddr), "s"(lds_dst)
               : "memory");
}

struct H8Frags {
  bf16x8 f[4];
};

int* gemm_tile_queue(hipStream_t s);  // gemm8p.hip: per-stream queue words of the dynamic tile assignment
bool gemm_dyn_enabled();

// DYN: tiles drawn from the per-XCD queues of p.tq instead of the static list bx, bx + G, ... -- the protocol of gemm8p_body<.., DYN>
// (gemm8p.hip): two tiles from one blocking fetch_add(2), tile i + 2 drawn at the start of tile i's epilogue, LDS mailbox, read at
// k-tile 1 of tile i + 1 (the cursor crosses into the next tile in k-tile nk - 2: nk >= 3)
template <int EPI, int XMODE, bool DYN = false>
__global__ __launch_bounds__(256, 2) void gemm8h_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // = the wave's column block wc (0 .. 3)
  const int hi = lane >> 5;

  const int tiles_m = (p.M + 127) >> 7;
  const int tiles_n = (p.N + 255) >> 8;
  const int ntiles = tiles_m * tiles_n;
  const int G = gridDim.x, bx = blockIdx.x;
  const int n_my = DYN ? 0x10000 : (ntiles - bx + G - 1) / G;
  int q_start = 0, q_n = 0, d_cur = -1, d_next = -1;  // DYN: my queue's chunk of the tile list; tile computed / tile the cursor moves to next
  bool d_more = DYN, d_mail = false, s_stop = false;
  int* const d_head = DYN ? p.tq + (bx & 7) * 16 : nullptr;
  int* const mbox = (int*)(smem + H8_RING);  // (the epilogue staging area is idle during the k loops)
  auto pos_origin = [&](int wg, int& m0, int& n0) {
    n0 = (wg % tiles_n) << 8;
    m0 = (wg / tiles_n) << 7;
  };
  auto tile_origin = [&](int i, int& m0, int& n0) {
    int wg = bx + i * G;
    if (p.xcd_swizzle & 1) {  // bijective on [0, ntiles): XCD x owns a contiguous chunk of the tile list
      const int q = ntiles >> 3, r = ntiles & 7, x = wg & 7;
      wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (wg >> 3);
    }
    pos_origin(wg, m0, n0);
  };
  auto tq_exit = [&]() {
    if (tid == 0) {
      const int t = __hip_atomic_fetch_add(p.tq + 128, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == G - 1) {
#pragma unroll
        for (int x = 0; x < 8; ++x) __hip_atomic_store(p.tq + 16 * x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p.tq + 128, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };
  if constexpr (DYN) {
    const int q = ntiles >> 3, r = ntiles & 7, x = bx & 7;
    q_n = q + (x < r ? 1 : 0);
    q_start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    if (tid == 0) *mbox = __hip_atomic_fetch_add(d_head, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int j = __builtin_amdgcn_readfirstlane(*(volatile int*)mbox);
    __syncthreads();
    if (j >= q_n) {
      tq_exit();
      return;
    }
    d_cur = q_start + j;
    d_more = j + 1 < q_n;
    d_next = d_more ? d_cur + 1 : -1;
  }
  const int nk = (p.K + 63) >> 6;
  const int H = n_my * nk * 3;  // slots this workgroup streams (DYN: unknown -- the cursor stops when a draw comes back empty)

  // ---------------------------------------------------------------- staging (LDS-DMA) side
  const char* zsrc = (const char*)g_zero_block;
  // A slot: per-lane byte offsets (k-tile 0) from the matrix base for the wave's 4 pieces; rows beyond the matrix are clamped to valid
  // ones (they only feed output rows that the epilogue masks).  Piece i of wave w = image rows (4 w + i) * 8 + (lane >> 3), the lane's
  // 16-B chunk lane & 7 holds source chunk (lane & 7) ^ ((row >> 1) & 7).
  // B slots (N % 256 == 0, checked by the launcher): image row r = 32 w + 8 i + prow is weight row n0 + 64 w + 8 i + prow (+ 32 for
  // B-second) -- the tile, the wave and the piece go into the scalar base, the lane keeps prow and its chunk (two swizzle parities)
  unsigned offA[4];
  const int prow = lane >> 3;
  const unsigned offB_even = (unsigned)((prow * p.ldb + (((lane & 7) ^ (prow >> 1)) << 3)) * 2);
  const unsigned offB_odd = (unsigned)((prow * p.ldb + (((lane & 7) ^ (prow >> 1) ^ 4) << 3)) * 2);
  const size_t stepB = (size_t)8 * p.ldb * 2;  // bytes between the weight rows of consecutive pieces
  const char* s_pbt = nullptr;                 // B base of the cursor's tile and wave: row n0 + 64 w, k-tile 0
  auto set_src = [&](int m0, int n0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (wave * 4 + i) * 8 + prow;  // row of the 128-row slot image
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      const int m1 = remap_row(min(m0 + r, p.M - 1), p.a_grp, p.a_pre);
      offA[i] = (unsigned)(((size_t)m1 * p.lda + c * 8) * 2);
    }
    s_pbt = (const char*)p.B + (size_t)(n0 + wave * 64) * p.ldb * 2;
  };
  // LDS byte address of slot role j of k-tile number T (see the header): B2 -> slot 2; rot = T % 3 picks B1 / A among {0, 1, 3}
  auto slot_of = [](int j, int rot) -> int {
    if (j == 2) return 2 * H8_SLOT;
    if (j == 0) return (rot == 0 ? 0 : rot == 1 ? 3 : 1) * H8_SLOT;
    return (rot == 0 ? 1 : rot == 1 ? 0 : 3) * H8_SLOT;
  };

  // staging cursor -- wave-uniform (SGPRs): slot counter, my-tile index, k-tile inside that tile, k-tile number mod 3, operand bases
  // advanced to the cursor's k-tile, "no K tail in this k-tile"
  int s_h = 0, s_i = 0, s_kt = 0, s_rot = 0;
  const unsigned lds0 = (unsigned)(size_t)smem + wave * 4096;  // this wave's 4 pieces inside a slot
  const char* s_pa = (const char*)p.A;
  const char* s_pb = nullptr;  // = s_pbt advanced to the cursor's k-tile (set with the first tile below)
  bool s_fast = p.K >= 64;
  auto issue = [&](auto jt) {  // the slot under the cursor has type J == s_h % 3 (the call sites keep this invariant)
    constexpr int J = decltype(jt)::value;
    const unsigned dst = lds0 + slot_of(J, s_rot);
    const char* bb = s_pb + (J == 2 ? 4 * stepB : 0);  // (B slots) piece 0 of this wave: + 32 weight rows for B-second
    if (__builtin_expect(s_fast, 1)) {
      if constexpr (J == 1) h8_glds16x4(s_pa, offA[0], offA[1], offA[2], offA[3], dst);
      else h8_glds16x4b(bb, bb + stepB, bb + 2 * stepB, bb + 3 * stepB, offB_even, offB_odd, dst);
    } else {  // K tail (the last k-tile of a tile only): zero-fill per lane
      const int krem = p.K - s_kt * 64;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + prow;
        const int kq = ((lane & 7) ^ ((r >> 1) & 7)) * 8;  // this lane's k offset (elements) inside the k-tile
        const char* src = (J == 1) ? s_pa + offA[i] : bb + i * stepB + ((i & 1) ? offB_odd : offB_even);
        h8_glds16_v((kq < krem) ? src : zsrc, dst + i * 1024);
      }
    }
    ++s_h;
    if constexpr (J == 2) {  // the cursor moves on to the next k-tile
      s_pa += 128;
      s_pb += 128;
      s_rot = s_rot == 2 ? 0 : s_rot + 1;
      if (++s_kt == nk) {
        s_kt = 0;
        s_pa = (const char*)p.A;
        if constexpr (DYN) {
          if (d_next >= 0) {
            int m0s, n0s;
            pos_origin(d_next, m0s, n0s);
            set_src(m0s, n0s);
          } else {
            s_stop = true;
          }
        } else if (++s_i < n_my) {
          int m0s, n0s;
          tile_origin(s_i, m0s, n0s);
          set_src(m0s, n0s);
        }
        s_pb = s_pbt;
      }
      s_fast = p.K - s_kt * 64 >= 64;
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  // counted waits (this wave's pieces; everything issued later may stay in flight).  While the cursor is still issuing (s_h < H) the
  // stream is in its steady state: in front of phase 1 only B1 of the next k-tile (4 pieces) is younger than the B2 needed, in front
  // of the next k-tile's phase 0 B2 and B1 (8 pieces) are younger than the A needed.  Once the cursor has run dry (the last k-tile or
  // two of the workgroup): drain.
  auto wait_b2 = [&]() {
    if (DYN ? !s_stop : s_h < H) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto wait_next_ktile = [&]() {
    if (DYN ? !s_stop : s_h < H) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  // ---------------------------------------------------------------- fragment (LDS read) side
  const int sw = (lane >> 1) & 7;
  const int rowoff = (lane & 31) * 128;
  auto load_rows = [&](const char* rows, H8Frags& fr) {  // 32 rows of 128 B at `rows`, swizzled 16-B chunks 2 ks + hi
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fr.f[ks] = *(const bf16x8*)(rows + rowoff + (((2 * ks + hi) ^ sw) << 4));
  };
  f32x16 acc[2][4];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  zero_acc();
  auto mma2 = [&](f32x16& c0, f32x16& c1, const H8Frags& w, const H8Frags& x0, const H8Frags& x1) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.f[ks], x0.f[ks], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.f[ks], x1.f[ks], c1, 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto phase_end = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---------------------------------------------------------------- prologue: B1(0), A(0), B2(0), B1(1) in flight, the first two landed
  {
    int m0s, n0s;
    if constexpr (DYN) pos_origin(d_cur, m0s, n0s);
    else tile_origin(0, m0s, n0s);
    set_src(m0s, n0s);
    s_pb = s_pbt;
  }
  if (DYN ? !s_stop : s_h < H) issue(I0{});
  if (DYN ? !s_stop : s_h < H) issue(I1{});
  if (DYN ? !s_stop : s_h < H) issue(I2{});
  if (DYN ? !s_stop : s_h < H) issue(I0{});
  if (s_h >= 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // B1(0), A(0) landed; B2(0), B1(1) in flight
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // a one-k-tile stream (H = 3)
  phase_end();

  int rot = 0;  // k-tiles computed so far (across my tiles) mod 3
  for (int ti = 0; ti < n_my; ++ti) {
    int m0, n0;
    if constexpr (DYN) pos_origin(d_cur, m0, n0);
    else tile_origin(ti, m0, n0);
    for (int kt = 0; kt < nk; ++kt) {
      if constexpr (DYN) {
        if (kt == 1 && d_mail) {  // left by thread 0 at the end of the previous epilogue: every wave is >= 4 barriers past that
          d_next = __builtin_amdgcn_readfirstlane(*(volatile int*)mbox);
          d_more = d_next >= 0;
        }
      }
      const char* sb1 = smem + slot_of(0, rot);
      const char* sa = smem + slot_of(1, rot);
      const char* sb2 = smem + 2 * H8_SLOT;
      H8Frags b1, b2, a1[2], a2[2];
      // ---- phase 0: B-first + A rows 0..63 -> quadrant (cols 0..31, rows 0..63)
      load_rows(sb1 + wave * 4096, b1);
      load_rows(sa, a1[0]);
      load_rows(sa + 4096, a1[1]);
      __builtin_amdgcn_sched_barrier(0);
      mma2(acc[0][0], acc[0][1], b1, a1[0], a1[1]);
      __builtin_amdgcn_sched_barrier(0);
      wait_b2();  // B2 of this k-tile (read in phase 1)
      phase_end();
      // ---- phase 1: B-second -> quadrant (cols 32..63, rows 0..63); A of the next k-tile goes where B-first was
      load_rows(sb2 + wave * 4096, b2);
      __builtin_amdgcn_sched_barrier(0);
      if (DYN ? !s_stop : s_h < H) issue(I1{});
      __builtin_amdgcn_sched_barrier(0);
      mma2(acc[1][0], acc[1][1], b2, a1[0], a1[1]);
      phase_end();
      // ---- phase 2: A rows 64..127 -> quadrant (cols 32..63, rows 64..127); B-second of the next k-tile into slot 2
      load_rows(sa + 8192, a2[0]);
      load_rows(sa + 12288, a2[1]);
      __builtin_amdgcn_sched_barrier(0);
      if (DYN ? !s_stop : s_h < H) issue(I2{});
      __builtin_amdgcn_sched_barrier(0);
      mma2(acc[1][2], acc[1][3], b2, a2[0], a2[1]);
      phase_end();
      // ---- phase 3: quadrant (cols 0..31, rows 64..127); B-first of the k-tile after next goes where A was
      if (DYN ? !s_stop : s_h < H) issue(I0{});
      __builtin_amdgcn_sched_barrier(0);
      mma2(acc[0][2], acc[0][3], b1, a2[0], a2[1]);
      __builtin_amdgcn_sched_barrier(0);
      wait_next_ktile();  // B1 and A of the next k-tile (read in its phase 0)
      phase_end();
      rot = rot == 2 ? 0 : rot + 1;
    }
    char* reg = gemm_epilogue_uses_lds<EPI, false, 64, H8_REGION>(p) ? smem + H8_RING + wave * H8_REGION : nullptr;
    // (all four ring slots hold operands of the next k-tiles here: no second staging region for the SwiGLU epilogue)
    int d_t = 0;
    if constexpr (DYN) {  // the tile after next
      if (d_more && tid == 0) d_t = __hip_atomic_fetch_add(d_head, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    gemm_epilogue<EPI, false, 128, 64, H8_REGION, XMODE>(p, acc, reg, m0, n0, 0, wave, lane, 0, nullptr);
    zero_acc();
    if constexpr (DYN) {
      d_mail = d_more;
      if (d_more && tid == 0) *(volatile int*)mbox = d_t < q_n ? q_start + d_t : -1;
      d_cur = d_next;
      d_next = -1;
      if (d_cur < 0) break;
    }
  }
  if constexpr (DYN) tq_exit();
}

template <int EPI, int XMODE = 0>
static int launch8h(const GemmArgs& a, hipStream_t s) {
  auto kern = gemm8h_kernel<EPI, XMODE>;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, H8_LDS);
    attr_set = true;
  }
  static int slots = 0;
  if (!slots) {
    int dev = 0;
    hipDeviceProp_t prop;
    (void)hipGetDevice(&dev);
    (void)hipGetDeviceProperties(&prop, dev);
    int cus = prop.multiProcessorCount - prop.multiProcessorCount % 8;
    if (cus < 8) cus = 8;
    cus = gemm_cu_cap(cus);
    slots = 2 * cus;  // two resident workgroups per CU
    if (const char* e = getenv("VTP_GEMM8H_WG_PER_CU")) slots = atoi(e) > 0 ? atoi(e) * cus : slots;  // diagnostics
    if (getenv("VTP_GEMM8H_DEBUG")) {
      int nb = -1;
      hipError_t rc = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, 256, H8_LDS);
      fprintf(stderr, "[gemm8h] occupancy query: %d workgroups / CU (rc %d), LDS %d B, grid slots %d\n", nb, (int)rc, H8_LDS, slots);
    }
  }
  const int ntiles = cdiv(a.M, 128) * cdiv(a.N, 256);
  if (ntiles > slots && a.K >= 192 && (a.xcd_swizzle & 1) && gemm_dyn_enabled()) {  // persistent launch: tiles drawn from the queues
    if (int* tq = gemm_tile_queue(s)) {
      auto dk = gemm8h_kernel<EPI, XMODE, true>;
      static bool dattr = false;
      if (!dattr) {
        (void)hipFuncSetAttribute((const void*)dk, hipFuncAttributeMaxDynamicSharedMemorySize, H8_LDS);
        dattr = true;
      }
      GemmArgs b = a;
      b.tq = tq;
      hipLaunchKernelGGL(dk, dim3(slots), dim3(256), H8_LDS, s, b);
      return check_launch("gemm8h_nt_dyn");
    }
  }
  hipLaunchKernelGGL(kern, dim3(ntiles > slots ? slots : ntiles), dim3(256), H8_LDS, s, a);
  return check_launch("gemm8h_nt");
}

// entry point used by the dispatcher of gemm.hip (tile configuration 9); the caller has checked gemm8p_fits (32-bit staging offsets)
// and N % 256 == 0 (the B staging has no column clamp)
int launch_gemm8h_nt(const GemmArgs& a, int epi, hipStream_t s) {
  switch (epi) {
    case EPI_BF16:
      if (a.rope_pos) return launch8h<EPI_BF16, 1>(a, s);
      if (a.swiglu_pre) return launch8h<EPI_BF16, 2>(a, s);
      return launch8h<EPI_BF16, 0>(a, s);
    case EPI_F32: return launch8h<EPI_F32>(a, s);
    case EPI_SWIGLU: return launch8h<EPI_SWIGLU>(a, s);
    case EPI_GELU: return launch8h<EPI_GELU>(a, s);
    default: set_error("gemm8h: unsupported epilogue %d", epi); return VTP_ERR_ARG;
  }
}

}  // namespace vtp
