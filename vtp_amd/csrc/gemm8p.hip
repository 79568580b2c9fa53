// 256 x 256 x 64 "8-phase" bf16 MFMA GEMM main loop for gfx950 (CDNA4): the large-M GEMMs of the VTP train step
// (row-concatenated trunk passes, M = 16k .. 34k tokens; weight gradients with K = tokens).  Same contract, operand layouts,
// swizzles and epilogues (gemm_common.h) as gemm.hip; what differs is the schedule of the k loop.
//
//   * 8 waves = 2 (M) x 4 (N), wave tile 128 x 64 = 4 x 2 accumulators of v_mfma_f32_32x32x16_bf16 (128 VGPRs); waves w and
//     w + 4 share a SIMD and belong to different wave groups (group = w >> 2 = the wave's row half).
//   * a k-tile (64 deep) is staged as FOUR 16-KiB half-tiles, in the order they are consumed:
//       j = 0  B-first   columns  wc*64 + [0,32)   of every wave column wc      (4 ds_read_b128 per wave)
//       j = 1  A-first   rows     wr*128 + [0,64)  of every wave row wr         (8 reads)
//       j = 2  B-second  columns  wc*64 + [32,64)
//       j = 3  A-second  rows     wr*128 + [64,128)
//     NT image of a half-tile: [128 rows][64 k] (128-B rows, 16-B chunk index XOR ((row >> 1) & 7));
//     TN image (weight gradients, operands [k = tokens][columns]): [64 k][128 cols] (256-B rows, chunk XOR 4 * (k & 3)), read
//     with ds_read_b64_tr_b16.  Each wave issues 2 LDS-DMA pieces (global_load_lds_dwordx4, 1 KiB) per half-tile.
//   * ring of 8 half-tile slots (2 k-tiles); the staging cursor runs 7 half-tiles ahead of the compute cursor, ACROSS output
//     tiles (persistent workgroups: the next tile's operands stream in under the current tile's last k-tiles and epilogue).
//   * one k-tile = 2 phases (round 4; -DVTP_P8_FOUR_PHASE builds the earlier four-phase schedule the file name comes from: 8 barrier
//     intervals per k-tile); phase A = { reads of B-first, A-first, B-second | DMA issue | s_barrier | 16 MFMAs (rows 0..63 of the wave
//     tile x k = 64) | s_barrier }, phase B = { reads of A-second | 3 DMA issues | s_barrier | 16 MFMAs | s_barrier }
//     [four-phase: phase p = { fragment reads of p | LDS-DMA issue of half-tile g + 7 | s_barrier | 8 MFMAs (one 64 x 32 quadrant of
//     the wave tile x k = 64) | s_barrier }].  The two wave groups run ONE barrier apart (group 1 executes
//     one extra barrier in front of a tile's k loop, group 0 one behind it), so on every SIMD one wave is in its MFMA segment
//     while its partner reads LDS / issues DMA: the matrix pipe is never shared and never idle for longer than a barrier.
//   * `s_waitcnt vmcnt(6)` once per k-tile (phase 3): every half-tile of the next k-tile has landed, 3 stay in flight.
//   Hazards (why the order j = B-first, A-first, B-second, A-second): the slot overwritten in phase p held the half-tile
//   consumed >= 2 phases earlier, except B-first (consumed in phase 0, overwritten in phase 1): its four reads are issued
//   first and retired by `lgkmcnt(8)` in front of phase 0's barrier.
//   * dedicated 32 KiB epilogue staging (4 KiB per wave) beside the 128 KiB ring: 160 KiB = all of a CU's LDS, 1 workgroup / CU.
#include "gemm_common.h"
#include "gemm_group.h"
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <unordered_map>

namespace vtp {

namespace {
constexpr int P8_SLOT = 16384;
constexpr int P8_RING = 8 * P8_SLOT;
constexpr int P8_REGION = 4096;
constexpr int P8_LDS = P8_RING + 8 * P8_REGION;
}  // namespace

__device__ __forceinline__ void p8_wait_vm_halftiles(int n) {  // at most n (0..4) half-tiles = 2 n DMA pieces outstanding
  if (n >= 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (n == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (n == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (n == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// LDS-DMA of 16 B per lane with a scalar 64-bit base + a 32-bit lane offset (no vector address arithmetic), M0 = LDS
// destination of the wave's 1-KiB piece.  Inline asm because hipcc turns the "tail or not" choice of the source pointer into a
// per-lane 64-bit select in front of every piece otherwise; its completion is counted by hand (the vmcnt waits below).
__device__ __forceinline__ void p8_glds16(const char* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}

// the same with a per-lane 64-bit source address (K tails: lanes past the end read the zero block)
__device__ __forceinline__ void p8_glds16_v(const char* vaddr, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(vaddr), "s"(lds_dst)
               : "memory");
}

typedef __attribute__((ext_vector_type(4))) int p8_i32x4;
typedef __attribute__((ext_vector_type(8))) int p8_i32x8;
// the MFMA operands of one 32-row (or 32-column) block over a k-tile of 128 bytes per row: four 16-B fragments for the bf16
// 32x32x16 MFMA, or two 32-B register tuples for the fp8 32x32x64 one (each filled by two 16-B LDS reads -- same addresses)
template <bool F8>
struct P8Frags {
  bf16x8 f[4];
};
template <>
struct P8Frags<true> {
  p8_i32x8 g[2];
};

// one (tile, K slice) of a grouped weight-gradient launch: the split-K combine of its tile happens in the launch (see GroupArgs)
struct GroupTile {
  float* part;   // fp32 partial sums, fragment-major [tiles][splits][8 waves][32][64 lanes] float4; null: no combine (one slice)
  int* ticket;   // one arrival counter per tile (zero between launches: the last arriver resets it)
  int tile, splits;
};

// DYN (persistent NT launches, one K slice, >= 3 k-tiles per tile): the workgroup does not own tiles bx, bx + G, ... but DRAWS them
// from per-XCD queues in device memory (p.tq: heads of 8 queues that hold the same contiguous chunks of the tile list as the static
// XCD-aware order; the workgroup draws from queue bx & 7).  A launch that cannot get all of its CUs at once -- an RCCL kernel holds
// 16 .. 32 of them for the whole backward, or another stream's GEMM is still running -- then ends when the TILES are done, not when
// the statically-assigned tile list of a late-starting workgroup is (tools/cu_thief.py).  The staging cursor runs into the next tile
// 1.75 k-tiles ahead of the MFMAs, so a workgroup holds TWO tiles beyond the one it computes: tiles 0 and 1 come from one blocking
// fetch_add(2) at the start; tile i + 2 is drawn by thread 0 at the START of tile i's epilogue (an ordinary compiler-visible atomic:
// the epilogue's own loads drain the vector-memory queue anyway), goes into an LDS mailbox at its END and is taken into an SGPR by
// every wave at k-tile 1 of tile i + 1 -- before the cursor needs it (k-tile nk - 2).  (A returning atomic from inline asm inside the
// k loop was tried first: its destination register is written when the result arrives, long after the statement, and the compiler
// -- which cannot know -- had meanwhile given the register to something else: memory faults.)  The last workgroup to leave zeroes the
// queue words.
template <int EPI, bool TRANS, int VAR = 0, int XMODE = 0, bool DYN = false>
__device__ __forceinline__ void gemm8p_body(const GemmArgs& p, int bx, int zslice, const int G, const bool flat, const GroupTile gt) {
  // VAR bit 3: the operands are fp8 (e4m3, OCP) -- same bytes, same staging, same fragment reads (the launcher passes K and the
  // leading dimensions in 2-byte units); only the MFMA changes: v_mfma_f32_32x32x64_f8f6f4 takes 32 B per lane, i.e. two of
  // the 16-B fragments, and A and B use the same (fragment, byte) -> k-slot map, so the dot products pair up the right elements
  constexpr bool FP8 = (VAR & 8) != 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int hi = lane >> 5;

  const int tiles_m = (p.M + 255) >> 8;
  const int tiles_n = (p.N + 255) >> 8;
  const int ntiles = tiles_m * tiles_n;
  // (bx, zslice, G): this workgroup computes tiles bx, bx + G, ... of K slice zslice (the kernels below derive them from the grid)
  const int n_my = DYN ? 0x10000 : (ntiles - bx + G - 1) / G;
  // DYN: queue bx & 7 = positions [q_start, q_start + q_n) of the tile list; d_cur = position being computed, d_next = the one the
  // staging cursor moves to next (-1: none), all wave-uniform
  int q_start = 0, q_n = 0, d_cur = -1, d_next = -1;
  bool d_more = DYN;      // the last draw returned a tile: keep drawing
  bool d_mail = false;    // the previous tile's epilogue left a draw in the mailbox
  int* const d_head = DYN ? p.tq + (bx & 7) * 16 : nullptr;
  if constexpr (DYN) {
    const int q = ntiles >> 3, r = ntiles & 7, x = bx & 7;
    q_n = q + (x < r ? 1 : 0);
    q_start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  }
  auto pos_origin = [&](int wg, int& m0, int& n0) {
    n0 = (wg % tiles_n) << 8;
    m0 = (wg / tiles_n) << 8;
  };
  auto tile_origin = [&](int i, int& m0, int& n0) {
    int wg = bx + i * G;
    if ((p.xcd_swizzle & 1) && !flat) {  // bijective on [0, ntiles): XCD x owns a contiguous chunk of the tile list
      const int q = ntiles >> 3, r = ntiles & 7, x = wg & 7;
      wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (wg >> 3);
    }
    pos_origin(wg, m0, n0);
  };
  auto tq_exit = [&]() {  // once per workgroup, behind its last draw: the last one to leave resets the queue words for the next launch
    if (tid == 0) {
      const int t = __hip_atomic_fetch_add(p.tq + 128, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == G - 1) {
#pragma unroll
        for (int x = 0; x < 8; ++x) __hip_atomic_store(p.tq + 16 * x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p.tq + 128, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };
  int* const mbox = (int*)(smem + P8_RING);  // DYN: mailbox word (the epilogue staging area is idle during the k loops)
  if constexpr (DYN) {  // the first two tiles: one blocking draw of two consecutive queue positions
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
  const int kbeg = zslice * p.k_split;
  const int kend = min(p.K, kbeg + p.k_split);
  const int nk = (kend - kbeg + 63) >> 6;
  const int H = n_my * nk * 4;  // half-tiles this workgroup streams (DYN: unknown -- the cursor stops when a draw comes back empty)

  // ---------------------------------------------------------------- staging (LDS-DMA) side
  const char* zsrc = (const char*)g_zero_block;
  // per-lane byte offsets (k-tile 0) from the matrix base -- B for half-tile types 0 / 2, A for 1 / 3; rows (NT) and columns
  // (TN) beyond the matrix are clamped to valid ones: they only feed output rows / columns that the epilogue masks
  unsigned off[4][2];
  int kq[2];  // NT: this lane's k offset (elements) inside the k-tile | TN: its k row inside the k-tile
  auto set_src = [&](int m0, int n0) {
    if constexpr (!TRANS) {
      const int prow = lane >> 3, slot = lane & 7;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 8 + prow;  // row of the 128-row half-tile image
        const int c = slot ^ ((r >> 1) & 7);
        kq[i] = c * 8;
        const int nb = n0 + (r >> 5) * 64 + (r & 31);
        const int mb = m0 + (r >> 6) * 128 + (r & 63);
        const int n1 = min(nb, p.N - 1), n2 = min(nb + 32, p.N - 1);
        const int m1 = remap_row(min(mb, p.M - 1), p.a_grp, p.a_pre), m2 = remap_row(min(mb + 64, p.M - 1), p.a_grp, p.a_pre);
        off[0][i] = (unsigned)(((size_t)n1 * p.ldb + kbeg + c * 8) * 2);
        off[2][i] = (unsigned)(((size_t)n2 * p.ldb + kbeg + c * 8) * 2);
        off[1][i] = (unsigned)(((size_t)m1 * p.lda + kbeg + c * 8) * 2);
        off[3][i] = (unsigned)(((size_t)m2 * p.lda + kbeg + c * 8) * 2);
      }
    } else {
      // piece q of a [64 k][128 cols] image: byte q*1024 + lane*16 -> k row r = q*4 + (lane >> 4), stored slot sp = lane & 15,
      // source chunk s = sp ^ 4*(r & 3)
      const int sp = lane & 15;
      const int s = sp ^ (4 * ((lane >> 4) & 3));
      const int cc = s * 8;  // column inside the 128-column image
      const int nb = n0 + (cc >> 5) * 64 + (cc & 31);
      const int mb = m0 + (cc >> 6) * 128 + (cc & 63);
      const int n1 = min(nb, p.N - 8), n2 = min(nb + 32, p.N - 8), m1 = min(mb, p.M - 8), m2 = min(mb + 64, p.M - 8);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 4 + (lane >> 4);
        kq[i] = r;
        const size_t ta = (size_t)(kbeg + r) * p.lda, tb = (size_t)(kbeg + r) * p.ldb;
        off[0][i] = (unsigned)((tb + n1) * 2);
        off[2][i] = (unsigned)((tb + n2) * 2);
        off[1][i] = (unsigned)((ta + m1) * 2);
        off[3][i] = (unsigned)((ta + m2) * 2);
      }
    }
  };

  // staging cursor -- all wave-uniform (SGPRs), updated incrementally so that the load segment of a phase carries a handful
  // of scalar instructions: half-tile counter, my-tile index, k-tile inside that tile, LDS destination of this wave's piece 0
  // of the half-tile under the cursor, the operand bases advanced to the cursor's k-tile, "no K tail in this k-tile"
  int s_h = 0, s_i = 0, s_kt = 0;
  bool s_stop = false;  // DYN: the tile under the cursor was the last one
  unsigned s_dst = (unsigned)(size_t)smem + wave * 2048;
  const size_t step_a = TRANS ? (size_t)128 * p.lda : 128, step_b = TRANS ? (size_t)128 * p.ldb : 128;
  const char* s_pa = (const char*)p.A;
  const char* s_pb = (const char*)p.B;
  bool s_fast = kend - kbeg >= 64;
  // one 1-KiB LDS-DMA piece (i = 0, 1) of the half-tile under the cursor; its type J == s_h & 3 (the call sites keep this invariant)
  auto issue_piece = [&](auto jt, int i) {
    constexpr int J = decltype(jt)::value;
    const char* base = (J & 1) ? s_pa : s_pb;
    if (s_fast) {
      p8_glds16(base, off[J][i], s_dst + i * 1024);
    } else {  // K tail (the last k-tile of a tile only): zero-fill per lane
      const int krem = kend - kbeg - s_kt * 64;
      // inline asm like the fast path: ONE compiler-visible LDS-DMA in this loop makes hipcc drain `vmcnt(0)` in front of the
      // fragment reads of every k-tile (it cannot see the asm pieces, so its own count of what is in flight is always "this one")
      const char* z = (kq[i] < krem) ? base + off[J][i] : zsrc;
      p8_glds16_v(z, s_dst + i * 1024);
    }
  };
  auto advance = [&](auto jt) {
    constexpr int J = decltype(jt)::value;
    ++s_h;
    s_dst = (s_dst + P8_SLOT) & (P8_RING - 1);  // the ring starts at LDS address 0 (the dynamic segment is the only LDS)
    if constexpr (J == 3) {
      s_pa += step_a;
      s_pb += step_b;
      if (++s_kt == nk) {
        s_kt = 0;
        s_pa = (const char*)p.A;
        s_pb = (const char*)p.B;
        if constexpr (DYN) {
          if (d_next >= 0) {
            int m0s, n0s;
            pos_origin(d_next, m0s, n0s);
            set_src(m0s, n0s);
          } else {
            s_stop = true;  // no further tile: the cursor stops (s_h stays the true count of half-tiles issued: the vmcnt waits use it)
          }
        } else if (++s_i < n_my) {
          int m0s, n0s;
          tile_origin(s_i, m0s, n0s);
          set_src(m0s, n0s);
        }
      }
      s_fast = kend - kbeg - s_kt * 64 >= 64;
    }
  };
  auto issue = [&](auto jt) {
    constexpr int J = decltype(jt)::value;
    if (__builtin_expect(s_fast, 1)) {  // straight-line: 2 x { m0 = LDS destination ; global_load_lds scalar-base + lane offset }
      const char* base = (J & 1) ? s_pa : s_pb;
      p8_glds16(base, off[J][0], s_dst);
      p8_glds16(base, off[J][1], s_dst + 1024);
    } else {
      issue_piece(jt, 0);
      issue_piece(jt, 1);
    }
    advance(jt);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;

  // ---------------------------------------------------------------- fragment (LDS read) side
  const int sw = (lane >> 1) & 7;
  const int rowoff = (lane & 31) * 128;
  const int t_li = lane & 15, t_rq = t_li >> 2, t_cin = ((lane >> 4) & 1) * 16 + (t_li & 3) * 4;
  auto tr_frag = [&](const char* base, int colbase, int ks) -> bf16x8 {
    const int col = colbase + t_cin;
    const char* b0 = base + (ks * 16 + hi * 8 + t_rq) * 256 + (((col >> 3) ^ (4 * t_rq)) << 4) + ((col & 7) << 1);
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)b0);
    bf16x4 h4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(b0 + 4 * 256));
    return __builtin_shufflevector(lo, h4, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  using Frags = P8Frags<FP8>;
  auto load_rows = [&](const char* rows, Frags& fr) {  // NT: 32 rows of 128 B at `rows`, swizzled 16-B chunks 2 ks + hi
    if constexpr (FP8) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const p8_i32x4 c0 = *(const p8_i32x4*)(rows + rowoff + (((4 * kk + hi) ^ sw) << 4));
        const p8_i32x4 c1 = *(const p8_i32x4*)(rows + rowoff + (((4 * kk + 2 + hi) ^ sw) << 4));
        fr.g[kk] = __builtin_shufflevector(c0, c1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fr.f[ks] = *(const bf16x8*)(rows + rowoff + (((2 * ks + hi) ^ sw) << 4));
    }
  };
  auto load_b = [&](const char* slot, Frags& fr) {  // this wave's 32 columns of a B half-tile, k = 0..63
    if constexpr (!TRANS) load_rows(slot + wc * 4096, fr);
    else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fr.f[ks] = tr_frag(slot, wc * 32, ks);
    }
  };
  auto load_a = [&](const char* slot, Frags (&fr)[2]) {  // this wave's 64 rows of an A half-tile
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      if constexpr (!TRANS) load_rows(slot + wr * 8192 + jj * 4096, fr[jj]);
      else {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fr[jj].f[ks] = tr_frag(slot, wr * 64 + jj * 32, ks);
      }
    }
  };

  // TN: column sums of A (bias gradient) ride along -- the wc == 0 waves of the first tile column add up the A fragments they
  // load anyway (v_dot2_f32_bf16 against ones: 4 VALU per fragment, in the shadow of the MFMAs), one atomic per row per tile.
  // (Spreading the sums over the four waves of a row group was measured slower: 556 vs 510 us for a trunk block's grouped launch.)
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
  bool do_csum = false;
  auto add_csum = [&](const Frags (&fr)[2], int jbase) {
    if constexpr (!FP8) {
      typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
      const bf16x2_t ones = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            const bf16x2_t v = {fr[jj].f[ks][2 * w], fr[jj].f[ks][2 * w + 1]};
            csum[jbase + jj] = __builtin_amdgcn_fdot2_f32_bf16(v, ones, csum[jbase + jj], false);
          }
    }
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
  // one MFMA segment: a 64 x 32 quadrant x k = 64, at raised priority (the partner wave on this SIMD is in its load segment)
  auto mma2 = [&](f32x16& c0, f32x16& c1, const Frags& w, const Frags& x0, const Frags& x1) {
    __builtin_amdgcn_s_setprio(1);
    if constexpr (FP8) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w.g[kk], x0.g[kk], c0, 0, 0, 0, 0, 0, 0);  // e4m3 x e4m3; zero scale operands select the unscaled v_mfma_f32_32x32x64_f8f6f4
        c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w.g[kk], x1.g[kk], c1, 0, 0, 0, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.f[ks], x0.f[ks], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.f[ks], x1.f[ks], c1, 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto seg_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  if (__builtin_expect(p.dbg_delay > 0, 0) && ((blockIdx.x >> 3) & 1)) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)p.dbg_delay) __builtin_amdgcn_s_sleep(16);
  }
  // ---------------------------------------------------------------- prologue: 7 half-tiles in flight, the first 4 landed
  {
    int m0s, n0s;
    if constexpr (DYN) pos_origin(d_cur, m0s, n0s);
    else tile_origin(0, m0s, n0s);
    set_src(m0s, n0s);
  }
  if (DYN ? !s_stop : s_h < H) issue(I0{});
  if (DYN ? !s_stop : s_h < H) issue(I1{});
  if (DYN ? !s_stop : s_h < H) issue(I2{});
  if (DYN ? !s_stop : s_h < H) issue(I3{});
  if (DYN ? !s_stop : s_h < H) issue(I0{});
  if (DYN ? !s_stop : s_h < H) issue(I1{});
  if (DYN ? !s_stop : s_h < H) issue(I2{});
  p8_wait_vm_halftiles(s_h - 4);
  seg_barrier();

  int ktg = 0;  // k-tiles computed so far (across my tiles): ring half = ktg & 1
  unsigned long long cyc0 = 0;
  auto stamp = [&](int ti, int which) {
    if (__builtin_expect(p.timing != nullptr, 0) && tid == 0 && ti < 16) {
      p.timing[((size_t)blockIdx.x * 16 + ti) * 4 + which] = wall_clock64();
      // slot 3: shader cycles (s_memtime) the k loop took -- with the 100 MHz stamps this gives the clock the CU ran at
      if (which == 0) cyc0 = clock64();
      if (which == 1) p.timing[((size_t)blockIdx.x * 16 + ti) * 4 + 3] = clock64() - cyc0;
    }
  };
  for (int ti = 0; ti < n_my; ++ti) {
    int m0, n0;
    if constexpr (DYN) pos_origin(d_cur, m0, n0);
    else tile_origin(ti, m0, n0);
    stamp(ti, 0);
    if constexpr (TRANS) do_csum = p.colsum != nullptr && wc == 0 && n0 == 0;
    if (wr == 1) seg_barrier();  // group 1 runs one barrier behind group 0 through this tile's k loop
    for (int kt = 0; kt < nk; ++kt, ++ktg) {
      if (__builtin_expect(p.timing != nullptr, 0) && tid == 0 && ti == 1 && kt < 16)  // diagnostics: start of every k-tile of tile 1
        p.timing[(size_t)(gridDim.x + blockIdx.x) * 64 + kt] = wall_clock64();
      const char* kb = smem + (ktg & 1) * (4 * P8_SLOT);
      Frags b1, b2, a1[2], a2[2];
      if constexpr (DYN) {
        if (kt == 1 && d_mail) {  // written by thread 0 at the end of the previous epilogue: every wave is >= 4 barriers past that
          d_next = __builtin_amdgcn_readfirstlane(*(volatile int*)mbox);
          d_more = d_next >= 0;
        }
      }
#ifndef VTP_P8_FOUR_PHASE
      // Two phases of 16 MFMAs per k-tile (the schedule in use since round 4: half as many barrier intervals per k-tile as the four
      // phases of 8 MFMAs below; measured 1-3 % per launch, +0.3 % on the step in three of three same-box repetitions,
      // profiles/r04_2ph_step_ab.log).  The two wave groups still run one barrier apart.
      // ---- phase A: B-first, A-first, B-second -> quadrants (cols 0..63, rows 0..63); issues A-second of the next k-tile
      load_b(kb, b1);
      load_a(kb + P8_SLOT, a1);
      load_b(kb + 2 * P8_SLOT, b2);
      __builtin_amdgcn_sched_barrier(0);
      if (DYN ? !s_stop : s_h < H) issue(I3{});
      __builtin_amdgcn_sched_barrier(0);
      p8_wait_vm_halftiles(s_h - 4 * ktg - 4);  // A-second of THIS k-tile (read in phase B)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // phase B overwrites the three slots read here
      seg_barrier();
      mma2(acc[0][0], acc[0][1], b1, a1[0], a1[1]);
      mma2(acc[1][0], acc[1][1], b2, a1[0], a1[1]);
      if constexpr (TRANS) {
        if (do_csum) add_csum(a1, 0);
      }
      seg_barrier();
      // ---- phase B: A-second -> quadrants (cols 0..63, rows 64..127); issues B-first, A-first, B-second of the k-tile after next
      load_a(kb + 3 * P8_SLOT, a2);
      __builtin_amdgcn_sched_barrier(0);
      if (DYN ? !s_stop : s_h < H) issue(I0{});
      if (DYN ? !s_stop : s_h < H) issue(I1{});
      if (DYN ? !s_stop : s_h < H) issue(I2{});
      __builtin_amdgcn_sched_barrier(0);
      p8_wait_vm_halftiles(s_h - 4 * ktg - 7);  // B-first, A-first, B-second of the NEXT k-tile (read in its phase A)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the next phase A overwrites the slot read here
      seg_barrier();
      mma2(acc[1][2], acc[1][3], b2, a2[0], a2[1]);
      mma2(acc[0][2], acc[0][3], b1, a2[0], a2[1]);
      if constexpr (TRANS) {
        if (do_csum) add_csum(a2, 2);
      }
      seg_barrier();
    }
#else
      static_assert(!DYN, "the tile queue is wired into the two-phase schedule only");
      // ---- phase 0: B-first + A-first -> quadrant (cols 0..31, rows 0..63)
      load_b(kb, b1);
      __builtin_amdgcn_sched_barrier(0);
      load_a(kb + P8_SLOT, a1);
      __builtin_amdgcn_sched_barrier(0);
      if (DYN ? !s_stop : s_h < H) issue(I3{});
      __builtin_amdgcn_sched_barrier(0);
      // the B-first reads (issued first) are done: phase 1 overwrites that slot (TN: 8 + 16 tr reads, the counter holds 15)
      if constexpr (!TRANS) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
      seg_barrier();
      mma2(acc[0][0], acc[0][1], b1, a1[0], a1[1]);
      if constexpr (TRANS) {
        if (do_csum) add_csum(a1, 0);
      }
      seg_barrier();
      // ---- phase 1: B-second -> quadrant (cols 32..63, rows 0..63)
      load_b(kb + 2 * P8_SLOT, b2);
      __builtin_amdgcn_sched_barrier(0);
      if (DYN ? !s_stop : s_h < H) issue(I0{});
      seg_barrier();
      mma2(acc[1][0], acc[1][1], b2, a1[0], a1[1]);
      seg_barrier();
      // ---- phase 2: A-second -> quadrant (cols 32..63, rows 64..127)
      load_a(kb + 3 * P8_SLOT, a2);
      __builtin_amdgcn_sched_barrier(0);
      if (DYN ? !s_stop : s_h < H) issue(I1{});
      seg_barrier();
      mma2(acc[1][2], acc[1][3], b2, a2[0], a2[1]);
      if constexpr (TRANS) {
        if (do_csum) add_csum(a2, 2);
      }
      seg_barrier();
      // ---- phase 3: quadrant (cols 0..31, rows 64..127); the next k-tile's four half-tiles must have landed
      if (DYN ? !s_stop : s_h < H) issue(I2{});
      p8_wait_vm_halftiles(s_h - 4 * (ktg + 2));
      seg_barrier();
      mma2(acc[0][2], acc[0][3], b1, a2[0], a2[1]);
      seg_barrier();
    }
#endif
    if (wr == 0) seg_barrier();  // re-align the groups: both run the epilogue together
    stamp(ti, 1);
    if constexpr (TRANS) {
      if (do_csum) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = csum[j] + __shfl_xor(csum[j], 32, 64);  // the two k halves of the fragment layout
          const int m = m0 + wr * 128 + j * 32 + (lane & 31);
          if (hi == 0 && m < p.M) unsafeAtomicAdd(p.colsum + remap_row(m, p.c_grp, p.c_pre), v);
          csum[j] = 0.f;
        }
      }
    }
    if constexpr (EPI != EPI_F32_SLAB && EPI != EPI_F32_ATOMIC) {
      if (gt.part != nullptr) {
        // split-K combine inside the launch (cdna_hip_programming.md, "In-launch split-K reduction"): every slice publishes its
        // accumulators (fragment-major: a lane re-reads exactly its own positions) -> every wave drains its stores -> barrier ->
        // lane 0 takes a ticket.  The slice that draws the last ticket acquires, adds the other slices' partials to its registers
        // in slice order and runs the epilogue; nobody ever waits for another workgroup.
        // write-through (sc1) 16-B stores through a wave-uniform buffer descriptor: the partial sums leave the XCD's L2 as they
        // are written, so no agent-scope release (an L2 write-back of everything dirty) is needed before the ticket
        // (cdna_hip_programming.md, "publish-large": 3.0 vs 8.2 us per 64 KB workgroup; here 256 KB)
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
        const f32x4* mine = (const f32x4*)gt.part + (((size_t)gt.tile * gt.splits + zslice) * 8 + wave) * 2048;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)mine, 0, 2048 * 16, 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              f32x4 v;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
              __builtin_amdgcn_raw_buffer_store_b128((u32x4_t)v, rsrc, lane * 16 + ((i * 4 + j) * 4 + q) * 1024, 0, /*sc1*/ 16);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flag = (int*)(smem + P8_RING);  // the epilogue staging area is idle here (the epilogue follows the combine)
        if (tid == 0) {
          const int t = __hip_atomic_fetch_add(gt.ticket + gt.tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int last = t == gt.splits - 1;
          if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(gt.ticket + gt.tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
          }
          *flag = last;
        }
        __syncthreads();
        const bool last = *flag != 0;
        __syncthreads();  // the flag word is part of the ring's tail region: nobody rewrites it before everyone has read it
        if (!last) {
          stamp(ti, 2);
          zero_acc();
          continue;
        }
        for (int z = 0; z < gt.splits; ++z) {
          if (z == zslice) continue;
          const f32x4* other = (const f32x4*)gt.part + (((size_t)gt.tile * gt.splits + z) * 8 + wave) * 2048 + lane;
          // NL loads in flight per lane (the fragment registers are dead here); the NT epilogues keep more values live across the
          // combine (bias, residual prefetch), so they take the partials in smaller bites
          constexpr int NL = TRANS ? 16 : 8;
#pragma unroll
          for (int g = 0; g < 32 / NL; ++g) {
            f32x4 v[NL];
#pragma unroll
            for (int t = 0; t < NL; ++t) v[t] = other[(g * NL + t) * 64];
#pragma unroll
            for (int t = 0; t < NL; ++t) {
              const int u = g * NL + t;
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[u >> 4][(u >> 2) & 3][4 * (u & 3) + e] += v[t][e];
            }
            asm volatile("" ::: "memory");
          }
        }
      }
    }
    char* reg = gemm_epilogue_uses_lds<EPI, TRANS, 64, P8_REGION>(p) ? smem + P8_RING + wave * P8_REGION : nullptr;
    // the ring slot under the staging cursor is free (its half-tile was consumed >= 2 phases ago, the next LDS-DMA into it is issued
    // after this epilogue): the wave's own 2 KiB of it are a second staging region
    char* reg2 = (EPI == EPI_SWIGLU && reg) ? smem + s_dst : nullptr;
    int d_t = 0;
    if constexpr (DYN) {  // the tile after next
      if (d_more && tid == 0) d_t = __hip_atomic_fetch_add(d_head, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    gemm_epilogue<EPI, TRANS, 128, 64, P8_REGION, XMODE>(p, acc, reg, m0, n0, wr, wc, lane, zslice, reg2);
    stamp(ti, 2);
    zero_acc();
    if constexpr (DYN) {
      d_mail = d_more;
      if (d_more && tid == 0) *(volatile int*)mbox = d_t < q_n ? q_start + d_t : -1;
      d_cur = d_next;
      d_next = -1;  // (known at k-tile 1 of the next tile, if a draw is in the mailbox)
      if (d_cur < 0) break;
    }
  }
  if constexpr (DYN) tq_exit();
}

// the persistent NT launch with dynamic tile assignment (gemm8p_body<.., DYN = true>): grid = the CU count, one K slice
template <int EPI, int VAR = 0, int XMODE = 0>
__global__ __launch_bounds__(512) void gemm8p_dyn_kernel(const GemmArgs p) {
  gemm8p_body<EPI, false, VAR, XMODE, true>(p, blockIdx.x, 0, (int)gridDim.x, false, GroupTile{nullptr, nullptr, 0, 1});
}

template <int EPI, bool TRANS, int VAR = 0, int XMODE = 0>
__global__ __launch_bounds__(512) void gemm8p_kernel(const GemmArgs p) {
  // split-K launches arrive as ONE flat grid of ntiles x splits workgroups (bit 2 of xcd_swizzle): workgroups are dealt to the 8
  // XCDs round-robin, and XCD x takes a contiguous chunk of the (split-major) list -- so the tiles that stream the same K slice of
  // A / B sit behind ONE L2 and that slice is fetched from HBM once, not once per XCD
  const int ntiles = ((p.M + 255) >> 8) * ((p.N + 255) >> 8);
  const bool flat = (p.xcd_swizzle & 4) != 0;
  int bx = blockIdx.x, zslice = blockIdx.z;
  if (flat) {
    const int W = gridDim.x, q = W >> 3, r = W & 7, x = bx & 7;
    const int c = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bx >> 3);
    zslice = c / ntiles;
    bx = c - zslice * ntiles;
  }
  GroupTile gt{nullptr, nullptr, 0, 1};
  if (flat && p.part != nullptr) gt = GroupTile{p.part, p.ticket, bx, (int)gridDim.x / ntiles};
  gemm8p_body<EPI, TRANS, VAR, XMODE>(p, bx, zslice, flat ? ntiles : (int)gridDim.x, flat, gt);
}

// Grouped weight gradients: up to 8 problems C_g[M_g, N_g] (+)= A_g[K, M_g]^T B_g[K, N_g] over the SAME K token rows (the four linear
// maps of a transformer block) as ONE launch of (sum of their 256 x 256 tiles) x splits workgroups -- one (tile, K slice) each,
// split-major over the XCDs like the flat split-K launches.  The slices of a tile are combined in the launch by the last arriver
// (gemm8p_body), which also applies the epilogue (C = C + sum or C = sum, SwiGLU row de-interleave, fused bias-gradient column sums):
// no slab buffers, no reduce launches, no separate column-sum launches.
__global__ __launch_bounds__(512) void gemm8p_grouped_tn_kernel(const GroupArgs ga) {
  const int W = gridDim.x, q = W >> 3, r = W & 7, x = blockIdx.x & 7;
  const int c = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + ((int)blockIdx.x >> 3);
  const int zslice = c / ga.ntiles, tile = c - zslice * ga.ntiles;
  int g = 0;
  for (int i = 1; i < ga.nprob; ++i)
    if ((int)ga.probs[i].tile0 <= tile) g = i;
  const GroupProblem& pr = ga.probs[g];
  GemmArgs p{};
  p.A = pr.A; p.B = pr.B; p.C = pr.C; p.resid = pr.accumulate ? pr.C : nullptr; p.colsum = pr.colsum;
  p.M = (int)pr.M; p.N = (int)pr.N; p.K = ga.K; p.lda = (int)pr.lda; p.ldb = (int)pr.ldb; p.ldc = (int)pr.ldc;
  p.c_grp = (int)pr.c_grp; p.c_pre = (int)pr.c_pre; p.k_split = ga.k_split; p.alpha = 1.f; p.xcd_swizzle = 0;
  p.timing = ga.timing;
  const int nt = ((p.M + 255) >> 8) * ((p.N + 255) >> 8);
  gemm8p_body<EPI_F32, true, 0>(p, tile - (int)pr.tile0, zslice, nt, true,
                                GroupTile{ga.splits > 1 ? ga.part : nullptr, ga.ticket, tile, ga.splits});
}

// diagnostics (vtp_gemm_debug): stamp buffer and a cap on the persistent grid (0 = every CU)
static unsigned long long* g_p8_timing = nullptr;
static int g_p8_grid = 0, g_p8_delay = 0;

// scratch of the in-launch split-K combine: launches on one stream are serialised and may share a buffer, concurrent streams (the
// text tower beside the decoder) must not.  A pool of 8 slots (256 tile-slices x 256 KiB + tickets each) is allocated at the FIRST
// combine launch of the process -- an eager one: the trainers warm up eagerly before any stream capture, and allocating under
// capture is not possible -- and every stream handle is bound to a slot the first time it is seen (no allocation then, so a
// capture stream that shows up later is fine).  No pool (allocation failed, first use under capture) or no free slot (more than 8
// streams): the dispatcher (vtp_gemm_nt) launches the shape without the K split.
struct CombineScratch {
  float* part = nullptr;
  int* ticket = nullptr;
};
static CombineScratch* combine_scratch(hipStream_t s) {
  constexpr int SLOTS = 8;
  static std::mutex mu;
  static CombineScratch pool[SLOTS];
  static std::unordered_map<hipStream_t, int> slot_of;
  static bool ready = false, failed = false;
  std::lock_guard<std::mutex> lock(mu);
  if (!ready) {
    if (failed) return nullptr;  // one attempt per process: callers fall back to the unsplit launch
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;  // no allocation under capture
    float* part = nullptr;
    int* ticket = nullptr;
    if (hipMalloc((void**)&part, (size_t)SLOTS * 256 * 65536 * sizeof(float)) != hipSuccess) {
      failed = true;
      return nullptr;
    }
    if (hipMalloc((void**)&ticket, (size_t)SLOTS * 1024 * sizeof(int)) != hipSuccess ||
        hipMemset(ticket, 0, (size_t)SLOTS * 1024 * sizeof(int)) != hipSuccess) {
      (void)hipFree(part);
      if (ticket) (void)hipFree(ticket);
      failed = true;
      return nullptr;
    }
    for (int i = 0; i < SLOTS; ++i) pool[i] = CombineScratch{part + (size_t)i * 256 * 65536, ticket + i * 1024};
    ready = true;
  }
  auto it = slot_of.find(s);
  if (it == slot_of.end()) {
    // a 9th distinct stream would have to share partial sums and tickets with a stream that may run concurrently: refuse
    if ((int)slot_of.size() >= SLOTS) return nullptr;
    it = slot_of.emplace(s, (int)slot_of.size()).first;
  }
  return &pool[it->second];
}

// queue words of the dynamic tile assignment (GemmArgs::tq): 256 ints per stream (heads of the 8 per-XCD queues 64 B apart, the exit
// counter), zero between launches.  Launches on one stream are serialised and share a slot; every stream handle is bound to its own
// slot the first time it is seen.  The pool (64 slots) is allocated at the first persistent launch outside stream capture (the trainers
// warm up eagerly); no pool / no free slot: the launch keeps the static tile lists.
int* gemm_tile_queue(hipStream_t s) {  // (shared with gemm8h.hip: launches on one stream are serialised, whatever the kernel)
  constexpr int SLOTS = 64;
  static std::mutex mu;
  static int* pool = nullptr;
  static std::unordered_map<hipStream_t, int> slot_of;
  static bool failed = false;
  std::lock_guard<std::mutex> lock(mu);
  if (!pool) {
    if (failed) return nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
    if (hipMalloc((void**)&pool, (size_t)SLOTS * 256 * sizeof(int)) != hipSuccess || hipMemset(pool, 0, (size_t)SLOTS * 256 * sizeof(int)) != hipSuccess) {
      pool = nullptr;
      failed = true;
      return nullptr;
    }
  }
  auto it = slot_of.find(s);
  if (it == slot_of.end()) {
    if ((int)slot_of.size() >= SLOTS) return nullptr;
    it = slot_of.emplace(s, (int)slot_of.size()).first;
  }
  return pool + it->second * 256;
}

// dynamic tile assignment of the persistent launches: VTP_GEMM_DYN=0 / 1 in the environment wins; otherwise what vtp_set_gemm_dynamic
// asked for (the trainers turn it on when they run beside collectives); default off -- on a chip the step has to itself the static
// lists are 0.4 % faster (no draw at the start of a workgroup, no exit count: 702 vs 699 images/s, profiles/r06_dyn_tiles.log)
static int g_dyn_mode = 0;
bool gemm_dyn_enabled() {
  static int env = -2;
  if (env == -2) {
    const char* e = getenv("VTP_GEMM_DYN");
    env = e ? (e[0] == '0' ? 0 : 1) : -1;
  }
  return env >= 0 ? env == 1 : g_dyn_mode == 1;
}
// may this stream use the in-launch split-K combine?  (allocates the scratch pool at the first call outside stream capture; false:
// the dispatcher launches the shape unsplit)
bool gemm8p_combine_ready(hipStream_t s) { return combine_scratch(s) != nullptr; }

template <int EPI, bool TRANS, int VAR = 0, int XMODE = 0>
static int launch8p(const GemmArgs& a0, int splits, hipStream_t s) {
  GemmArgs a = a0;
  a.timing = g_p8_timing;
  a.dbg_delay = g_p8_delay;
  if constexpr (EPI != EPI_F32_SLAB && EPI != EPI_F32_ATOMIC) {
    if (splits > 1) {  // in-launch combine: tiles x splits <= 256 (the caller's policy)
      CombineScratch* c = combine_scratch(s);
      if (!c || cdiv(a.M, 256) * cdiv(a.N, 256) * splits > 256) {
        set_error("gemm8p: split-K combine scratch unavailable or too many tile slices");
        return VTP_ERR_ARG;
      }
      a.part = c->part;
      a.ticket = c->ticket;
    }
  }
  auto kern = gemm8p_kernel<EPI, TRANS, VAR, XMODE>;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS);
    attr_set = true;
  }
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    (void)hipGetDevice(&dev);
    (void)hipGetDeviceProperties(&prop, dev);
    cus = prop.multiProcessorCount - prop.multiProcessorCount % 8;
    if (cus < 8) cus = 8;
  }
  const int ntiles = cdiv(a.M, 256) * cdiv(a.N, 256);
  if (splits > 1) {
    GemmArgs f = a;
    f.xcd_swizzle |= 4;
    hipLaunchKernelGGL(kern, dim3(ntiles * splits), dim3(512), P8_LDS, s, f);
    return check_launch(TRANS ? "gemm8p_tn" : "gemm8p_nt");
  }
  const int cap = g_p8_grid > 0 ? g_p8_grid : gemm_cu_cap(cus);
  if constexpr (!TRANS && EPI != EPI_F32_SLAB && EPI != EPI_F32_ATOMIC) {
    // persistent launch: tiles drawn from per-XCD queues (gemm8p_body DYN) when the shape allows the look-ahead (>= 4 k-tiles per tile)
    if (splits == 1 && ntiles > cap && a.K >= 256 && (a.xcd_swizzle & 1) && !a.timing && gemm_dyn_enabled()) {
      int* tq = gemm_tile_queue(s);
      if (tq) {
        auto dk = gemm8p_dyn_kernel<EPI, VAR, XMODE>;
        static bool dattr = false;
        if (!dattr) {
          hipFuncSetAttribute((const void*)dk, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS);
          dattr = true;
        }
        a.tq = tq;
        hipLaunchKernelGGL(dk, dim3(cap), dim3(512), P8_LDS, s, a);
        return check_launch("gemm8p_nt_dyn");
      }
    }
  }
  dim3 grid(splits == 1 && ntiles > cap ? cap : ntiles, 1, splits);
  hipLaunchKernelGGL(kern, grid, dim3(512), P8_LDS, s, a);
  return check_launch(TRANS ? "gemm8p_tn" : "gemm8p_nt");
}

// the staging addresses are 32-bit byte offsets from the operand bases
bool gemm8p_fits(const GemmArgs& a, bool trans) {
  const size_t lim = (size_t)1 << 32;
  if (trans) return (size_t)a.K * a.lda * 2 < lim && (size_t)a.K * a.ldb * 2 < lim;
  size_t rows_a = a.M;
  if (a.a_grp > 0) rows_a += ((size_t)a.M / a.a_grp + 1) * a.a_pre;
  return rows_a * a.lda * 2 < lim && (size_t)a.N * a.ldb * 2 < lim;
}

// entry points used by the dispatchers of gemm.hip
int launch_gemm8p_nt(const GemmArgs& a, int epi, int splits, hipStream_t s) {
  switch (epi) {
    case EPI_BF16:  // the fused extras of the LDS-staged store path are separate instantiations (gemm_epilogue XMODE)
      if (a.rope_pos) return launch8p<EPI_BF16, false, 0, 1>(a, splits, s);
      if (a.swiglu_pre) return launch8p<EPI_BF16, false, 0, 2>(a, splits, s);
      return launch8p<EPI_BF16, false, 0, 0>(a, splits, s);
    case EPI_F32: return launch8p<EPI_F32, false>(a, splits, s);
    case EPI_SWIGLU: return launch8p<EPI_SWIGLU, false>(a, splits, s);
    case EPI_GELU: return launch8p<EPI_GELU, false>(a, splits, s);
    case EPI_F32_ATOMIC: return launch8p<EPI_F32_ATOMIC, false>(a, splits, s);
    case EPI_F32_SLAB: return launch8p<EPI_F32_SLAB, false>(a, splits, s);
    default: set_error("gemm8p: unsupported epilogue %d", epi); return VTP_ERR_ARG;
  }
}

// fp8 (e4m3) operands, NT, forward epilogues only; `a` carries K, lda, ldb in 2-byte units (see the kernel's FP8 note)
int launch_gemm8p_nt_fp8(const GemmArgs& a, int epi, hipStream_t s) {
  switch (epi) {
    case EPI_BF16: return a.rope_pos ? launch8p<EPI_BF16, false, 8, 1>(a, 1, s) : launch8p<EPI_BF16, false, 8, 0>(a, 1, s);
    case EPI_F32: return launch8p<EPI_F32, false, 8>(a, 1, s);
    case EPI_SWIGLU: return launch8p<EPI_SWIGLU, false, 8>(a, 1, s);
    default: set_error("gemm8p fp8: unsupported epilogue %d", epi); return VTP_ERR_ARG;
  }
}

int launch_gemm8p_tn(const GemmArgs& a, int epi, int splits, hipStream_t s) {
  if (epi == EPI_F32) return launch8p<EPI_F32, true>(a, 1, s);
  if (epi == EPI_F32_ATOMIC) return launch8p<EPI_F32_ATOMIC, true>(a, splits, s);
  return launch8p<EPI_F32_SLAB, true>(a, splits, s);
}

}  // namespace vtp

// One launch for the weight gradients of a transformer block (see gemm8p_grouped_tn_kernel).  probs: device array of `nprob`
// GroupProblem records (16 x int64 each); ntiles = sum of their 256 x 256 tile counts; part / ticket: device scratch of
// ntiles * splits * 256 KiB and ntiles ints (ticket zero-initialised by the caller once; the kernel leaves it zero).
namespace vtp {
int launch_gemm4w_grouped_tn(const GroupArgs& ga, hipStream_t s);  // gemm4w_tn.hip
int launch_gemm4w_grouped_tn_items(const GroupArgs& ga, int nitems, hipStream_t s);
}
// kernel: 0 = the 8-phase kernel | 1 = the one-wave-per-SIMD kernel (gemm4w_tn.hip; the caller guarantees K % 8 == 0 and, as for
// every grouped launch, M_g, N_g, lda, ldb multiples of 8 and operands within 32-bit byte offsets)
extern "C" int vtp_gemm_tn_grouped_k(const void* probs, int nprob, int ntiles, int K, int splits, void* part, void* ticket, int kernel,
                                     void* stream) {
  using namespace vtp;
  VTP_REQUIRE(probs && nprob >= 1 && nprob <= 8, "vtp_gemm_tn_grouped: 1..8 problems");
  VTP_REQUIRE(ntiles >= 1 && K >= 1 && splits >= 1, "vtp_gemm_tn_grouped: bad shape (ntiles %d, K %d, splits %d)", ntiles, K, splits);
  VTP_REQUIRE(splits == 1 || (part && ticket), "vtp_gemm_tn_grouped: split-K needs the partial-sum and ticket buffers");
  VTP_REQUIRE(kernel == 0 || (kernel == 1 && K % 8 == 0), "vtp_gemm_tn_grouped: kernel %d not available for K = %d", kernel, K);
  GroupArgs ga{};
  ga.probs = (const GroupProblem*)probs; ga.part = (float*)part; ga.ticket = (int*)ticket;
  ga.nprob = nprob; ga.ntiles = ntiles; ga.K = K;
  ga.k_split = ((K + splits - 1) / splits + 63) / 64 * 64;
  ga.splits = (K + ga.k_split - 1) / ga.k_split;
  ga.timing = g_p8_timing;
  if (kernel == 1) return launch_gemm4w_grouped_tn(ga, (hipStream_t)stream);  // (stamps [workgroup][8] under vtp_gemm_debug)
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm8p_grouped_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS);
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm8p_grouped_tn_kernel, dim3(ntiles * ga.splits), dim3(512), P8_LDS, (hipStream_t)stream, ga);
  return check_launch("gemm8p_grouped_tn");
}
// The same launch from an explicit work-item list (one-wave-per-SIMD kernel only): items = device array of nitems records of 8 int32
// {tile, kbeg, kcount, nparts, part, 0, 0, 0} -- one workgroup each; the items of a tile partition [0, K) into nparts ranges (kbeg
// multiples of 64), part = 0 .. nparts - 1.  slots = the largest nparts: part holds ntiles * slots * 65536 floats.
extern "C" int vtp_gemm_tn_grouped_items(const void* probs, int nprob, int ntiles, int K, const void* items, int nitems, int slots,
                                         void* part, void* ticket, void* stream) {
  using namespace vtp;
  VTP_REQUIRE(probs && nprob >= 1 && nprob <= 8, "vtp_gemm_tn_grouped_items: 1..8 problems");
  VTP_REQUIRE(items && nitems >= ntiles && ntiles >= 1 && K >= 8 && K % 8 == 0 && slots >= 1,
              "vtp_gemm_tn_grouped_items: bad shape (ntiles %d, nitems %d, K %d, slots %d)", ntiles, nitems, K, slots);
  VTP_REQUIRE(slots == 1 || (part && ticket), "vtp_gemm_tn_grouped_items: split tiles need the partial-sum and ticket buffers");
  GroupArgs ga{};
  ga.probs = (const GroupProblem*)probs; ga.part = (float*)part; ga.ticket = (int*)ticket;
  ga.nprob = nprob; ga.ntiles = ntiles; ga.K = K; ga.k_split = K; ga.splits = slots;
  ga.timing = g_p8_timing;
  ga.items = (const GroupItem*)items;
  return launch_gemm4w_grouped_tn_items(ga, nitems, (hipStream_t)stream);
}
extern "C" int vtp_gemm_tn_grouped(const void* probs, int nprob, int ntiles, int K, int splits, void* part, void* ticket,
                                   void* stream) {
  return vtp_gemm_tn_grouped_k(probs, nprob, ntiles, K, splits, part, ticket, 0, stream);
}

extern "C" int vtp_set_gemm_dynamic(int on) {
  vtp::g_dyn_mode = on ? 1 : 0;
  return VTP_OK;
}

// diagnostics for tools/gemm8p_timeline.py: `timing` = device buffer of [workgroups][16 tiles][4] u64 s_memrealtime stamps (100 MHz)
// written by the 8-phase kernel (null: off); grid_limit caps its persistent grid (0: every CU); delay_ticks > 0 starts every second
// workgroup of an XCD that many 10-ns ticks late (is the chip's lock-step what serialises k loops and epilogue stores?)
extern "C" int vtp_gemm_debug(void* timing, int grid_limit, int delay_ticks) {
  // process-global hooks of PRODUCTION kernels: anything but "all off" is accepted only in a process that asked for
  // diagnostics (VTP_DIAG=1 in its environment: the tools/ scripts set it); the "no stores" mode of round 3 is gone.  The stamp
  // buffer must hold (workgroups of the launch + 256) x 64 u64 values -- the kernels index it by workgroup and do not check.
  const bool off = timing == nullptr && grid_limit == 0 && delay_ticks == 0;
  const char* diag = getenv("VTP_DIAG");
  if (!off && !(diag && diag[0] == '1')) {
    vtp::set_error("vtp_gemm_debug: diagnostics hooks need VTP_DIAG=1 in the environment");
    return VTP_ERR_ARG;
  }
  if (delay_ticks < 0 || grid_limit < 0) {
    vtp::set_error("vtp_gemm_debug: negative grid limit / delay");
    return VTP_ERR_ARG;
  }
  vtp::g_p8_timing = (unsigned long long*)timing;
  vtp::g_p8_grid = grid_limit;
  vtp::g_p8_delay = delay_ticks;
  return VTP_OK;
}
