// bf16 MFMA GEMM for gfx950:  C[M,N] = A[M,K] * B[N,K]^T  (+ fused epilogue), fp32 accumulate.
//
// Both operands are K-contiguous ("NT"): activations [tokens, features] and nn.Linear weights
// [out, in] are used as stored.  dgrad / wgrad reach this kernel through pre-transposed operands
// (weights: cached W^T; activations: transpose kernels in elementwise.hip).
//
// Structure (CDNA4): WAVES_M x WAVES_N waves, workgroup tile BM x BN x 64, a STAGES-deep LDS ring filled by
// `global_load_lds_dwordx4` (LDS-DMA, 16 B/lane, no VGPR round trip) with *counted* `s_waitcnt vmcnt(N)` and a raw
// `s_barrier`, so STAGES-2 further k-tiles stay in flight across the barrier while one is being multiplied.
// The LDS image of a tile is [row][64 k] bf16 (128-B rows) with the 16-byte chunk index XOR-swizzled by
// ((row>>1)&7) so that every ds_read_b128 lane group hits 16 distinct 16-B slots (conflict-free); LDS-DMA writes
// lane-linear, so the swizzle is applied to the per-lane *global source* address and to the read address.
// MFMA: v_mfma_f32_32x32x16_bf16 with the operands swapped (weight rows feed the A operand, token rows the B operand)
// so that each lane ends up holding 4 consecutive output columns of one output row -> 8/16-byte epilogue stores and
// lane-local SwiGLU / bias.  Workgroup ids are remapped so that each XCD (private 4 MiB L2) owns a contiguous band
// of output rows and sweeps the weight panel (blocks b, b+8, ... run on the same XCD).
#include "common.h"
#include "gemm_common.h"
#include <cstdlib>

namespace vtp {


// TRANS = false: A [M, K], B [N, K] (both K-contiguous).
// TRANS = true : A [K, M], B [K, N] row-major (K = reduction = tokens): C = A^T B without materialising the transposes
//                (weight gradients dW = dY^T X straight from the activation layouts).  The LDS image of a tile is then
//                [64 k][BM or BN cols] as in global memory (slot index XOR 4*(k&3)), and MFMA fragments are formed with
//                ds_read_b64_tr_b16: 16 lanes fetch a 4(k) x 16(col) block and each lane receives one column.
// PIPE = true: software-pipelined k-tile body -- the LDS-DMA pieces of the next tile are issued in four slices between
// the MFMA groups (their issue cost hides behind the matrix pipe) and the fragments of k-step ks+1 are read while the
// MFMAs of k-step ks execute.
template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, int EPI, bool TRANS, bool PIPE>
__global__ __launch_bounds__(64 * WAVES_M* WAVES_N) void gemm_nt_kernel(const GemmArgs p) {
  constexpr int BK = 64;
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;  // wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW;  // 1-KiB LDS-DMA pieces per wave per k-tile
  constexpr int P = PA + PB;
  static_assert(PA >= 1 && PB >= 1 && TM >= 1 && TN >= 1, "bad tile config");
  static_assert(STAGES >= 2 && STAGES <= 4, "2..4 stages");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WAVES_M, wn = wave / WAVES_M;
  const int hi = lane >> 5;

  const int tiles_m = (p.M + BM - 1) / BM;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int ntiles = tiles_m * tiles_n;
  const int G = gridDim.x;  // persistent: G <= ntiles workgroups, workgroup b owns tiles b, b + G, b + 2G, ...
  const int n_my = (ntiles - (int)blockIdx.x + G - 1) / G;
  // i-th tile of this workgroup -> output origin.  XCD x (= workgroup % 8; G is a multiple of 8 whenever G < ntiles)
  // walks a contiguous chunk of the tile list, n fastest: a chunk = a band of output rows x all weight panels.
  auto tile_origin = [&](int i, int& m0, int& n0) {
    int wg = blockIdx.x + i * G;
    if (p.xcd_swizzle & 1) {  // bijective on [0, ntiles)
      const int q = ntiles >> 3, r = ntiles & 7, x = wg & 7;
      wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (wg >> 3);
    }
    n0 = (wg % tiles_n) * BN;
    m0 = (wg / tiles_n) * BM;
  };
  const int kbeg = blockIdx.z * p.k_split;
  const int kend = min(p.K, kbeg + p.k_split);
  const int nk = (kend - kbeg + BK - 1) / BK;

  // ---- LDS-DMA source pointers (per lane, per piece) ----
  const int prow = lane >> 3;  // row inside an 8-row piece
  const int slot = lane & 7;   // 16-B slot inside the 128-B row
  const char* a_src[PA];
  const char* b_src[PB];
  int a_kc[PA], b_kc[PB];  // NT: this lane's source k offset (elements) inside the k-tile | TRANS: tile row (k) of the piece
  const char* zsrc = (const char*)g_zero_block;
  auto set_src = [&](int m0, int n0) {
    if constexpr (!TRANS) {
  #pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int row = (wave * PA + i) * 8 + prow;
        const int c = slot ^ ((row >> 1) & 7);
        int m = min(m0 + row, p.M - 1);
        m = remap_row(m, p.a_grp, p.a_pre);
        a_src[i] = (const char*)(p.A + (size_t)m * p.lda + kbeg + c * 8);
        a_kc[i] = c * 8;
      }
  #pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int row = (wave * PB + i) * 8 + prow;
        const int c = slot ^ ((row >> 1) & 7);
        const int n = min(n0 + row, p.N - 1);
        b_src[i] = (const char*)(p.B + (size_t)n * p.ldb + kbeg + c * 8);
        b_kc[i] = c * 8;
      }
    } else {
      // piece q of a [64 k][W cols] tile: byte q*1024 + lane*16 -> tile row r, stored slot s' ; source chunk s = s' ^ 4*(r&3)
  #pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int byte = (wave * PA + i) * 1024 + lane * 16;
        const int r = byte / (BM * 2);
        const int sp = (byte % (BM * 2)) >> 4;
        const int col = m0 + ((sp ^ (4 * (r & 3))) << 3);
        a_kc[i] = r;
        a_src[i] = col < p.M ? (const char*)(p.A + col) : nullptr;  // column base; the token row is added per stage
      }
  #pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int byte = (wave * PB + i) * 1024 + lane * 16;
        const int r = byte / (BN * 2);
        const int sp = (byte % (BN * 2)) >> 4;
        const int col = n0 + ((sp ^ (4 * (r & 3))) << 3);
        b_kc[i] = r;
        b_src[i] = col < p.N ? (const char*)(p.B + col) : nullptr;
      }
    }
  };

  // q < 0: issue every piece of the tile; q in 0..3: only the pieces j with j % 4 == q (A pieces are j = 0..PA-1)
  auto stage_part = [&](int buf, int kt, int q) {
    char* abase = smem + buf * STAGE_BYTES;
    char* bbase = abase + A_BYTES;
    const int krem = kend - kbeg - kt * BK;  // valid k elements left in this tile (>0)
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      if (q >= 0 && (i & 3) != q) continue;
      const char* s;
      if constexpr (!TRANS) {
        if constexpr (EPI == EPI_CONV_RELU || EPI == EPI_CONV_MASK) {
          long off = (long)kt * (BK * 2);
          if (p.conv_cin) {  // tap-shifted pixel rows (guard rows above / below the stack absorb the +-(W+3) reach)
            const int k0 = kt * BK, tap = k0 / p.conv_cin;
            const int ky = tap / 3, kx = tap - 3 * ky;
            off = ((long)((ky - 1) * p.conv_w2 + (kx - 1)) * p.lda + (k0 - tap * p.conv_cin)) * 2;
          }
          s = (a_kc[i] < krem) ? a_src[i] + off : zsrc;
        } else {
          s = (a_kc[i] < krem) ? a_src[i] + (size_t)kt * (BK * 2) : zsrc;
        }
      } else {
        const int t = kbeg + kt * BK + a_kc[i];
        s = (a_kc[i] < krem && a_src[i]) ? a_src[i] + (size_t)remap_row(t, p.a_grp, p.a_pre) * p.lda * 2 : zsrc;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                       (__attribute__((address_space(3))) void*)(abase + (wave * PA + i) * 1024),
                                       16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      if (q >= 0 && ((i + PA) & 3) != q) continue;
      const char* s;
      if constexpr (!TRANS) {
        s = (b_kc[i] < krem) ? b_src[i] + (size_t)kt * (BK * 2) : zsrc;
      } else {
        const int t = kbeg + kt * BK + b_kc[i];
        s = (b_kc[i] < krem && b_src[i]) ? b_src[i] + (size_t)remap_row(t, p.b_grp, p.b_pre) * p.ldb * 2 : zsrc;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                       (__attribute__((address_space(3))) void*)(bbase + (wave * PB + i) * 1024),
                                       16, 0, 0);
    }
  };
  auto stage = [&](int buf, int kt) { stage_part(buf, kt, -1); };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // per-lane LDS read offsets: row (lane&31), swizzle ((row>>1)&7) is lane-constant (tile offsets are multiples of 32)
  const int sw = (lane >> 1) & 7;
  const int rowoff = (lane & 31) * 128;

  // staging cursor: the (tile, k-tile) whose LDS-DMA is issued next.  It runs STAGES-1 k-tiles ahead of the compute cursor
  // ACROSS tile boundaries, so a tile's first k-tiles stream in while the previous tile finishes and stores its output.
  int s_i = 0, s_kt = 0;
  {
    int m0s, n0s;
    tile_origin(0, m0s, n0s);
    set_src(m0s, n0s);
  }
  auto advance = [&]() {
    if (++s_kt == nk) {
      s_kt = 0;
      if (++s_i < n_my) {
        int m0s, n0s;
        tile_origin(s_i, m0s, n0s);
        set_src(m0s, n0s);
      }
    }
  };
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s_i < n_my) {
      stage(s, s_kt);
      advance();
    }

  int buf = 0;     // ring slot of the k-tile being computed
  int landed = 0;  // k-tiles already known to be in LDS (all DMA drained before the previous tile's output stores)
  for (int ti = 0; ti < n_my; ++ti) {
  int m0, n0;
  tile_origin(ti, m0, n0);
  for (int kt = 0; kt < nk; ++kt) {
    // k-tile (ti, kt) has landed once at most (k-tiles issued after it) * P of this wave's DMA pieces are outstanding
    if (landed > 0) {
      --landed;
    } else {
      const int rem = (n_my - 1 - ti) * nk + (nk - 1 - kt);
      const int later = min(STAGES - 2, rem);
      if (later >= 2) wait_vmcnt<2 * P>();
      else if (later == 1) wait_vmcnt<P>();
      else wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();  // every wave's pieces of this k-tile landed; every wave finished reading the previous one
    asm volatile("" ::: "memory");
    const bool prefetch = s_i < n_my;
    int nb = buf + STAGES - 1;
    if (nb >= STAGES) nb -= STAGES;  // the slot the previous k-tile occupied
    if (!PIPE && prefetch) {
      stage(nb, s_kt);
      advance();
    }
    const char* abase = smem + buf * STAGE_BYTES;
    const char* bbase = abase + A_BYTES;

    auto load_frags = [&](int ks, bf16x8* wf, bf16x8* xf) {
      if constexpr (!TRANS) {
        const int coff = ((2 * ks + hi) ^ sw) << 4;
#pragma unroll
        for (int i = 0; i < TN; ++i)
          wf[i] = *(const bf16x8*)(bbase + (wn * WTN + i * 32) * 128 + rowoff + coff);
#pragma unroll
        for (int j = 0; j < TM; ++j)
          xf[j] = *(const bf16x8*)(abase + (wm * WTM + j * 32) * 128 + rowoff + coff);
      } else {
        // lane (i = lane&15, g = (lane>>4)&1, hi): column c = base + 16 g + i of the block is delivered to this lane;
        // it supplies the address of row (i>>2), 4 columns starting at 4*(i&3).  Two reads = k 8hi..8hi+3 and 8hi+4..8hi+7.
        const int li = lane & 15, lg = (lane >> 4) & 1;
        const int rq = li >> 2;                       // row inside the 4-row block == (row & 3) -> swizzle key
        const int cin = lg * 16 + (li & 3) * 4;       // column offset inside the 32-column fragment block
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          const int col = wn * WTN + i * 32 + cin;
          const int soff = (((col >> 3) ^ (4 * rq)) << 4) + ((col & 7) << 1);
          const char* b0 = bbase + (ks * 16 + hi * 8 + rq) * (BN * 2) + soff;
          bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)b0);
          bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(b0 + 4 * (BN * 2)));
          wf[i] = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const int col = wm * WTM + j * 32 + cin;
          const int soff = (((col >> 3) ^ (4 * rq)) << 4) + ((col & 7) << 1);
          const char* a0 = abase + (ks * 16 + hi * 8 + rq) * (BM * 2) + soff;
          bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)a0);
          bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0 + 4 * (BM * 2)));
          xf[j] = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
        }
      }
    };

    if constexpr (PIPE) {
      bf16x8 wf[2][TN], xf[2][TM];
      load_frags(0, wf[0], xf[0]);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks < 3) load_frags(ks + 1, wf[(ks + 1) & 1], xf[(ks + 1) & 1]);
        if (prefetch) stage_part(nb, s_kt, ks);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks & 1][i], xf[ks & 1][j], acc[i][j], 0, 0, 0);
      }
      if (prefetch) advance();
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 wf[TN], xf[TM];
        load_frags(ks, wf, xf);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
      }
    }
    if (++buf == STAGES) buf = 0;
  }
  if (ti + 1 < n_my) {
    // drain the DMA of the next tile's first k-tiles (issued up to a k-tile ago: all but landed) BEFORE the output stores,
    // so that the stores -- which share vmcnt -- never sit in front of a load the next iterations wait for
    wait_vmcnt<0>();
    landed = min(STAGES - 1, (n_my - 1 - ti) * nk);
  }

  // ---- epilogue (gemm_common.h): bf16 / SwiGLU / fp32-residual outputs go through LDS so that the global stores are full
  // 128/256-B row segments: each wave transposes 32-row blocks of its sub-tile in a private region of the ring slot the last
  // k-tile occupied
  {
    constexpr int REGION = STAGE_BYTES / NW;
    char* reg = nullptr;
    if (gemm_epilogue_uses_lds<EPI, TRANS, WTN, REGION>(p)) {
      const int lb = (buf == 0 ? STAGES : buf) - 1;
      __builtin_amdgcn_s_barrier();  // every wave finished reading the last k-tile
      asm volatile("" ::: "memory");
      reg = smem + lb * STAGE_BYTES + wave * REGION;
    }
    gemm_epilogue<EPI, TRANS, WTM, WTN, REGION>(p, acc, reg, m0, n0, wm, wn, lane);
  }
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }  // tile loop
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, int EPI, bool TRANS, bool PIPE = false>
static int launch_cfg(const GemmArgs& a, int splits, hipStream_t s) {
  constexpr int LDS = STAGES * (BM + BN) * 64 * 2;
  static bool attr_set = false;
  auto kern = gemm_nt_kernel<BM, BN, WAVES_M, WAVES_N, STAGES, EPI, TRANS, PIPE>;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  static int slots = 0;  // resident workgroups the device holds of this kernel (CUs x occupancy)
  if (!slots) {
    int occ = 1, dev = 0;
    hipDeviceProp_t prop;
    (void)hipGetDevice(&dev);
    (void)hipGetDeviceProperties(&prop, dev);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kern, 64 * WAVES_M * WAVES_N, LDS);
    slots = prop.multiProcessorCount * (occ < 1 ? 1 : occ);
    slots -= slots % 8;
  }
  const int ntiles = cdiv(a.M, BM) * cdiv(a.N, BN);
  // persistent when the tile list exceeds one resident wave of workgroups and the split-K factor is 1
  dim3 grid(splits == 1 && ntiles > slots ? slots : ntiles, 1, splits);
  hipLaunchKernelGGL(kern, grid, dim3(64 * WAVES_M * WAVES_N), LDS, s, a);
  return check_launch(TRANS ? "gemm_tn" : "gemm_nt");
}

// tile configurations (cfg id): 0 = 128x128 4 waves 2 stages | 1 = 128x128 4w 3 stages | 2 = 256x128 8w 2 stages |
// 3 = 256x128 8w 3 stages | 4 = 256x256 8w 2 stages | 5 = 128x128 8w 2 stages | 6 = 128x128 4w 4 stages
static int g_force_cfg = -1;
static int g_xcd_swizzle = 3;  // bit 0: XCD-aware tile order | bit 1: LDS-staged full-line stores for bf16 outputs
static int swz_flags() { return g_xcd_swizzle; }


// gemm8p.hip: 256x256 8-phase main loop (cfg 8)
int launch_gemm8p_nt(const GemmArgs& a, int epi, int splits, hipStream_t s);
int launch_gemm8p_tn(const GemmArgs& a, int epi, int splits, hipStream_t s);
bool gemm8p_fits(const GemmArgs& a, bool trans);
bool gemm8p_combine_ready(hipStream_t s);
// gemm8h.hip: 128x256 half-size variant, two workgroups per CU (cfg 9; NT, one K slice)
int launch_gemm8h_nt(const GemmArgs& a, int epi, hipStream_t s);
// gemm4w.hip: 256x256, one software-pipelined wave per SIMD (cfg 10; NT, one K slice, no A row remap)
int launch_gemm4w_nt(const GemmArgs& a, int epi, hipStream_t s);

template <int EPI, bool TRANS>
static int launch_gemm(const GemmArgs& a, int splits, int cfg, hipStream_t s) {
  if (cfg == 10) {
    if constexpr (!TRANS && (EPI == EPI_BF16 || EPI == EPI_F32 || EPI == EPI_SWIGLU || EPI == EPI_GELU)) {
      if (splits == 1 && a.N % 256 == 0 && a.K % 128 == 0 && a.a_grp == 0 && a.conv_cin == 0 && !a.timing &&
          gemm8p_fits(a, false))
        return launch_gemm4w_nt(a, EPI, s);
    }
    cfg = TRANS ? 5 : 8;
  }
  if (cfg == 9) {
    if constexpr (!TRANS && (EPI == EPI_BF16 || EPI == EPI_F32 || EPI == EPI_SWIGLU || EPI == EPI_GELU)) {
      if (splits == 1 && a.N % 256 == 0 && gemm8p_fits(a, false)) return launch_gemm8h_nt(a, EPI, s);
    }
    cfg = TRANS ? 5 : 8;
  }
  if (cfg == 8) {
    if constexpr (TRANS) {
      if (a.a_grp == 0 && a.b_grp == 0 && gemm8p_fits(a, true)) return launch_gemm8p_tn(a, EPI, splits, s);
      cfg = 5;  // token-row remaps (patch-embed / bottleneck wgrads): the ring kernel
    } else if constexpr (EPI != EPI_CONV_RELU && EPI != EPI_CONV_MASK) {
      if (gemm8p_fits(a, false)) return launch_gemm8p_nt(a, EPI, splits, s);
      cfg = 5;
    }
  }
  if constexpr (TRANS) {  // weight-gradient shapes only: keep the instantiation count small
    switch (cfg) {
      case 2: return launch_cfg<256, 128, 4, 2, 2, EPI, true>(a, splits, s);
      case 3: return launch_cfg<256, 128, 4, 2, 3, EPI, true>(a, splits, s);
      case 5: return launch_cfg<128, 128, 4, 2, 2, EPI, true>(a, splits, s);
      case 21: return launch_cfg<128, 128, 4, 2, 2, EPI, true, true>(a, splits, s);
      case 16: return launch_cfg<128, 128, 2, 2, 2, EPI, true, true>(a, splits, s);
      default: return launch_cfg<128, 128, 2, 2, 2, EPI, true>(a, splits, s);
    }
  } else {
    switch (cfg) {
      case 1: return launch_cfg<128, 128, 2, 2, 3, EPI, false>(a, splits, s);
      case 2: return launch_cfg<256, 128, 4, 2, 2, EPI, false>(a, splits, s);
      case 3: return launch_cfg<256, 128, 4, 2, 3, EPI, false>(a, splits, s);
      case 4:
        // the bf16 instantiation of the 256 x 256 ring tile spilled 462 VGPRs (the run-time RoPE / SwiGLU-backward extras on top of 256
        // accumulator registers) and is only a fallback since the 8-phase kernel took these shapes: it is the 256 x 128 tile now
        if constexpr (EPI == EPI_BF16) return launch_cfg<256, 128, 4, 2, 2, EPI, false>(a, splits, s);
        else return launch_cfg<256, 256, 4, 2, 2, EPI, false>(a, splits, s);
      case 5: return launch_cfg<128, 128, 4, 2, 2, EPI, false>(a, splits, s);
      case 6: return launch_cfg<128, 128, 2, 2, 4, EPI, false>(a, splits, s);
      case 7: return launch_cfg<128, 64, 4, 1, 3, EPI, false>(a, splits, s);   // few-tile shapes: twice the workgroups
      case 16: return launch_cfg<128, 128, 2, 2, 2, EPI, false, true>(a, splits, s);
      case 18: return launch_cfg<256, 128, 4, 2, 2, EPI, false, true>(a, splits, s);
      case 21: return launch_cfg<128, 128, 4, 2, 2, EPI, false, true>(a, splits, s);
      default: return launch_cfg<128, 128, 2, 2, 2, EPI, false>(a, splits, s);
    }
  }
}

// the 256x256 8-phase kernel (cfg 8, gemm8p.hip; one workgroup per CU) against the ring kernels: measured on MI355X
// (tools/gemm8p_bench.py, profiles/r03_gemm8p_bench.log) at M = 34144 / 16448 / 8224 rows.  With the round-3 epilogues (bias /
// LayerScale / residual prefetch per tile, tile-local addressing) it wins or ties from ~96 tiles on -- also where its tile list
// fills only 1.6 or 2.03 rounds of the CUs (fp32-residual K = 768 projection: x1.6 .. 1.7; w3 dgrad at 16448 rows: x1.4) -- except
// the short-K bf16 shapes with ~100 tiles (x0.96)
static bool use_8p_nt(int M, int N, int K, int epilogue) {
  if (N < 256 || K < 256) return false;
  const int tiles = cdiv(M, 256) * cdiv(N, 256);
  // K = 256 with thousands of tiles (the DINO head's prototype logits: 2816 x 65536, four k-tiles per tile): all epilogue -- the
  // LDS-staged full-line stores of the 8-phase kernel run it at 157 us, the ring kernels' 256 x 256 configuration at 525 us
  if (K < 512) return epilogue == VTP_EPI_BF16 && tiles >= 1024;
  switch (epilogue) {
    case VTP_EPI_F32: return tiles >= 128;  // (96 tiles = the pixel decoder's 8192 x 768: the 128 x 64 ring tiles win by 12-18 %)
    case VTP_EPI_SWIGLU: return tiles >= 96;
    case VTP_EPI_BF16:
    case VTP_EPI_GELU: return tiles >= (K < 1024 ? 192 : 96);
    default: return false;
  }
}

// the 128 x 256 half-size kernel with two workgroups per CU (cfg 9, gemm8h.hip) against the dispatch above: measured per shape in
// tools/gemm8h_bench.py (profiles/r04_gemm8h_bench.log).  Its free-running workgroup pairs use the matrix pipe ~10 % worse than the
// 256 x 256 kernel's barrier-coupled wave groups (long-K dgrads x0.88 .. 0.92), so it is taken only where one workgroup's epilogue
// hiding under the other's k loop, or its finer tail, outweighs that:
//   * w3 dgrad with the SwiGLU backward in its epilogue (25 us of epilogue per 256 x 256 tile) above one round of tiles: x1.16 .. 1.19;
//   * qkv + RoPE and the SwiGLU forward with at most ~4 rounds of 256 x 256 tiles (M = 16448 / 8192): x1.10 .. 1.13;
//   * plain bf16 epilogues with K <= 2304 where 256 x 256 tiles fill at most half the chip but 128 x 256 tiles fill >= 3/4 of it
//     (pixel decoder dgrads M = 8192, N = 768: x1.04 .. 1.15; text-tower c_proj dgrad 2464 x 3072 x 768: x1.24).
// fp32-residual and GELU epilogues never won (x0.83 .. 1.03).
static bool use_8h_nt(const GemmArgs& a, int epilogue) {
  if (a.N % 256 != 0 || a.N < 256 || a.K < 512) return false;
  const int t256 = cdiv(a.M, 256) * (a.N / 256), t8h = cdiv(a.M, 128) * (a.N / 256);
  if (epilogue == VTP_EPI_BF16) {
    if (a.swiglu_pre) return t256 > 256;
    if (a.rope_pos) return t256 >= 192 && t256 <= 768;
    return t256 <= 128 && t8h >= 192 && a.K <= 2304;
  }
  if (epilogue == VTP_EPI_SWIGLU) return t256 > 768 && t256 <= 1280;
  return false;
}

// weight gradients (TN, K = tokens): 8-phase kernel when tiles x splits is one round of the CUs and every K slice keeps >= 16
// k-tiles (vtp_gemm_tn_splits picks the split factor accordingly)
static bool use_8p_tn(int M, int N, int K, int splits, const GemmArgs& a) {
  if (a.a_grp || a.b_grp) return false;
  const int wgs = cdiv(M, 256) * cdiv(N, 256) * splits;
  return wgs >= 160 && wgs <= 256 && K / splits >= 1024;
}

// split factor of the in-launch K combine (1 = off).  Measured per shape in the step (gpurun_out/gemm_table_c*.txt): the publish +
// combine costs ~30 us per launch, so it pays only where it halves a LONG k loop on a half-empty chip -- the pixel decoder's w12
// dgrad (M = 8192, N = 768, K = 4096: 118.7 -> 81.7 us); with K <= 3072 or more than two slices every shape got slower
static int combine_splits(int M, int N, int K) {
  if (N < 256 || M < 1024 || K < 4096) return 1;
  const int tiles = cdiv(M, 256) * cdiv(N, 256);
  return (tiles >= 48 && tiles <= 128) ? 2 : 1;
}

// measured on MI355X at the VTP-B train-step shapes (tools/gemm_bench.py, profiles/gemm_bench_r01.log)
static int pick_cfg(int M, int N, int K, int epilogue, int splits) {
  if (g_force_cfg >= 0) return g_force_cfg;
  if (M < 128 || N < 128) return 0;
  if (splits > 1) return 3;                     // split-K wgrad: long K, few tiles -> 256x128, 3 stages
  if (use_8p_nt(M, N, K, epilogue)) return 8;
  // big-M GEMMs (the row-concatenated list forward, M = 34k): 256x256 tiles halve the LDS / L2 traffic per flop; they
  // need >= 1.5 resident waves of tiles to beat the 128x128 kernels' finer quantisation (tools/gemm_bench.py 34144)
  if (cdiv(M, 256) * cdiv(N, 256) >= 384 && (N >= 2304 || K >= (epilogue == VTP_EPI_F32 ? 4096 : 2048))) return 4;
  // few tiles (text tower M = 2464, DINO head M = 2816: 120 .. 360 tiles of 128 x 128 on 256 CUs): 128 x 64 tiles, 4 x 1 waves, three
  // stages -- twice the workgroups, three of them per CU (tools/text_gemm_ab.py: N = 768, K = 3072 fp32-residual 50.3 -> 30.0 us,
  // bf16 36.1 -> 28.1 us; not the GELU epilogue: 35.0 -> 38.7 us)
  if ((epilogue == VTP_EPI_BF16 || epilogue == VTP_EPI_F32) && K >= 512 && cdiv(M, 128) * cdiv(N, 128) < 400) return 7;
  if (K >= 4096) return 21;               // long K: pipelined 8-wave 128x128 (DMA issue + fragment prefetch between MFMAs)
  if (epilogue == VTP_EPI_SWIGLU) return 0;     // N = 2H wide: plenty of tiles, 4-wave 128x128
  return 5;                                     // short K (768..2304): 8-wave 128x128 hides the DMA latency best
}

// the one-wave-per-SIMD kernel with the hand-scheduled k loop (cfg 10, gemm4w.hip): its k loop keeps the matrix pipe 85 % busy against 67 %
// (profiles/r04_pmc_sq_cal_16384x4096x8192_cfg{8,10}.json; wall-time gain 11 .. 14 %: the chip answers the higher utilisation with a
// ~10 % lower clock), but a tile's epilogue runs on ONE wave per SIMD and nothing overlaps it.  Measured per shape
// (tools/gemm8h_bench.py with ALT_CFG=10, profiles/r04_gemm4w_bench.log): plain bf16 epilogue with K >= 2048 and at least ~3/4 of a
// round of 256 x 256 tiles x1.05 .. 1.12 (the w12 / qkv dgrads at M = 34144 and 16448); K = 768 x0.97 .. 0.99; the fused epilogues
// (RoPE x0.86, SwiGLU x0.94, SwiGLU backward x0.95, fp32 residual x0.88 .. 0.92) and few-tile shapes lose.
static bool use_4w_nt(const GemmArgs& a, int epilogue) {
  if (epilogue != VTP_EPI_BF16 || a.rope_pos || a.swiglu_pre || a.a_grp || a.conv_cin || a.timing) return false;
  if (a.N % 256 != 0 || a.K % 128 != 0 || a.K < 2048) return false;
  return cdiv(a.M, 256) * (a.N / 256) >= 192;
}

// the dispatch of the NT entry points: the measured table above, then the half-size / one-wave-per-SIMD kernels where they win
static int pick_cfg_nt(const GemmArgs& a, int epilogue, int splits) {
  const int cfg = pick_cfg(a.M, a.N, a.K, epilogue, splits);
  static const bool use_4w = [] { const char* e = getenv("VTP_GEMM4W"); return !(e && e[0] == '0'); }();  // VTP_GEMM4W=0: same-box A/B
  if (use_4w && g_force_cfg < 0 && splits == 1 && use_4w_nt(a, epilogue) && gemm8p_fits(a, false)) return 10;
  static const bool use_8h = [] { const char* e = getenv("VTP_GEMM8H"); return !(e && e[0] == '0'); }();  // VTP_GEMM8H=0: same-box A/B of the step
  if (use_8h && g_force_cfg < 0 && splits == 1 && a.a_grp == 0 && use_8h_nt(a, epilogue) && gemm8p_fits(a, false)) return 9;
  return cfg;
}

}  // namespace vtp

using namespace vtp;

template <int EPI>
static int launch_conv(const GemmArgs& a, hipStream_t s) {
  // M is millions of pixel rows, K = 9*Cin is long: the widest tile the channel count fills
  if (a.N <= 64) return launch_cfg<256, 64, 4, 2, 2, EPI, false>(a, 1, s);
  if (a.N <= 128) return launch_cfg<256, 128, 4, 2, 2, EPI, false>(a, 1, s);
  return launch_cfg<256, 256, 4, 2, 2, EPI, false>(a, 1, s);
}

extern "C" int vtp_conv3x3(const void* x, const void* w, const float* bias, void* y, const void* relu_mask, int NB, int H,
                           int W, int Cin, int Cout, int taps, int mode, void* stream) {
  VTP_REQUIRE(x && w && y, "vtp_conv3x3: null operand");
  VTP_REQUIRE(NB > 0 && H > 0 && W > 0, "vtp_conv3x3: bad shape");
  VTP_REQUIRE(taps == 9 || taps == 1, "vtp_conv3x3: taps must be 9 (3x3) or 1 (pre-unfolded rows)");
  VTP_REQUIRE(Cin % 8 == 0 && Cout % 4 == 0 && (taps == 1 || Cin % 64 == 0),
              "vtp_conv3x3: Cin must be a multiple of 64 (8 for pre-unfolded rows), Cout of 4");
  VTP_REQUIRE(mode == 0 ? bias != nullptr : true, "vtp_conv3x3: forward mode needs a bias");
  GemmArgs a{};
  a.A = (const bf16*)x; a.B = (const bf16*)w; a.C = y; a.C2 = (void*)relu_mask; a.bias = bias;
  a.M = NB * (H + 2) * (W + 2); a.N = Cout; a.K = taps * Cin;
  a.lda = Cin; a.ldb = taps * Cin; a.ldc = Cout; a.ldc2 = Cout; a.alpha = 1.f;
  a.k_split = (a.K + 63) / 64 * 64;
  a.xcd_swizzle = swz_flags();
  a.conv_cin = taps == 9 ? Cin : 0; a.conv_w2 = W + 2; a.conv_h2 = H + 2; a.conv_p = (H + 2) * (W + 2);
  hipStream_t s = (hipStream_t)stream;
  return mode == 0 ? launch_conv<EPI_CONV_RELU>(a, s) : launch_conv<EPI_CONV_MASK>(a, s);
}

extern "C" int vtp_gemm_splits(int K, int splits) {
  if (splits < 1) splits = 1;
  const int ks = ((K + splits - 1) / splits + 63) / 64 * 64;
  return (K + ks - 1) / ks;
}

// split-K factor for a weight-gradient GEMM C[M,N] = A[K,M]^T B[K,N] (K = tokens): prefers the 8-phase kernel's operating point
// (256x256 tiles x splits = one round of the CUs, >= 16 k-tiles per slice), else the ring kernel's (128x128 tiles x splits just
// under one resident wave of 512 workgroups, >= 8 k-tiles per slice).  Returns the effective number of slices.
extern "C" int vtp_gemm_tn_splits(int M, int N, int K) {
  const int t256 = cdiv(M, 256) * cdiv(N, 256);
  int s8 = 256 / t256;
  if (s8 > K / 1024) s8 = K / 1024;
  if (g_force_cfg < 0 && s8 >= 1) {
    const int eff = vtp_gemm_splits(K, s8);
    if (t256 * eff >= 160 && t256 * eff <= 256 && K / eff >= 1024) return eff;
  }
  const int t128 = cdiv(M, 128) * cdiv(N, 128);
  int s = 512 / t128;
  if (s > K / 512) s = K / 512;
  if (s > 16) s = 16;
  if (s < 1) s = 1;
  return vtp_gemm_splits(K, s);
}

extern "C" int vtp_set_gemm_tuning(int force_cfg, int xcd_swizzle) {
  g_force_cfg = force_cfg;
  g_xcd_swizzle = xcd_swizzle;
  return VTP_OK;
}

// host-only: which kernel configuration vtp_gemm_nt picks for a shape (no launch): 8 = 256x256 8-phase kernel, 7 = 128x64 ring
// tiles, 5 / 0 / 21 / 4 / ... = the other ring configurations; bits 8.. = in-launch split-K slices when > 1.  Lets the dispatch
// table be pinned by a CPU test (tests/test_host_logic.py) -- the policy is measured per shape and easy to break by an edit.
extern "C" int vtp_gemm_nt_config(int M, int N, int K, int epilogue) {
  if (M <= 0 || N <= 0 || K <= 0) return -1;
  GemmArgs a{};  // plain epilogue of the given kind (no fused RoPE / SwiGLU backward: those have their own entry points)
  a.M = M; a.N = N; a.K = K; a.lda = K; a.ldb = K;
  int cfg = pick_cfg_nt(a, epilogue, 1);
  int cs = 1;
  if (g_force_cfg < 0 && epilogue <= VTP_EPI_GELU && gemm8p_fits(a, false)) cs = combine_splits(M, N, K);  // as vtp_gemm_nt decides
  if (cs > 1) cfg = 8;
  return cfg | (cs > 1 ? cs << 8 : 0);
}

extern "C" int vtp_gemm_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, void* C2, int ldc2,
                           const float* bias, const float* gamma, const float* resid, int M, int N, int K, int epilogue,
                           int a_grp, int a_pre, int c_grp, int c_pre, int splits, float alpha, void* stream) {
  VTP_REQUIRE(A && B && C, "vtp_gemm_nt: null operand");
  VTP_REQUIRE(M > 0 && N > 0 && K > 0, "vtp_gemm_nt: bad shape M=%d N=%d K=%d", M, N, K);
  const int act_quick = epilogue == VTP_EPI_QUICK_GELU;  // the GELU epilogue with the other activation: same kernels, same dispatch
  if (act_quick) epilogue = VTP_EPI_GELU;
  VTP_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "vtp_gemm_nt: K, lda, ldb must be multiples of 8 (16-B rows)");
  VTP_REQUIRE(N % 4 == 0 && ldc % 4 == 0, "vtp_gemm_nt: N and ldc must be multiples of 4");
  VTP_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 16 == 0), "vtp_gemm_nt: operands must be 16-B aligned");
  VTP_REQUIRE(splits >= 1, "vtp_gemm_nt: splits must be >= 1");
  VTP_REQUIRE(splits == 1 || epilogue == VTP_EPI_F32_ATOMIC || epilogue == VTP_EPI_F32_SLAB,
              "vtp_gemm_nt: split-K needs the atomic or slab epilogue");
  VTP_REQUIRE(epilogue != VTP_EPI_F32_SLAB || (ldc2 > 0), "vtp_gemm_nt: slab epilogue needs ldc2 = slab stride / 4 (in float4 units)");
  GemmArgs a{};
  a.A = (const bf16*)A; a.B = (const bf16*)B; a.C = C; a.C2 = C2; a.bias = bias; a.gamma = gamma; a.resid = resid;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldc2 = ldc2;
  a.a_grp = a_grp; a.a_pre = a_pre; a.c_grp = c_grp; a.c_pre = c_pre; a.alpha = alpha;
  a.b_grp = 0; a.b_pre = 0;
  a.act_quick = act_quick;
  a.xcd_swizzle = swz_flags();
  int ks = ((K + splits - 1) / splits + 63) / 64 * 64;
  a.k_split = ks;
  splits = (K + ks - 1) / ks;
  hipStream_t s = (hipStream_t)stream;
  int cfg = pick_cfg_nt(a, epilogue, splits);
  // few output tiles, long K (the pixel decoder's and the text tower's dgrads, their K = 2048 .. 4096 projections): the 256 x 256
  // kernel with the K range cut into slices that are combined INSIDE the launch by the last-arriving slice (which then runs the
  // normal epilogue) -- tiles x slices fills the CUs that 30 .. 100 tiles alone leave idle
  int cs = 1;
  if (splits == 1 && g_force_cfg < 0 && epilogue <= VTP_EPI_GELU && gemm8p_fits(a, false)) cs = combine_splits(M, N, K);
  if (cs > 1 && !gemm8p_combine_ready(s)) cs = 1;  // no scratch for this stream: the unsplit launch (slower, never wrong)
  if (cs > 1) {
    a.k_split = ((K + cs - 1) / cs + 63) / 64 * 64;
    cs = (K + a.k_split - 1) / a.k_split;
    cfg = 8;
  }
  switch (epilogue) {
    case VTP_EPI_BF16: return launch_gemm<EPI_BF16, false>(a, cs, cfg, s);
    case VTP_EPI_F32: return launch_gemm<EPI_F32, false>(a, cs, cfg, s);
    case VTP_EPI_SWIGLU:
      VTP_REQUIRE(N % 16 == 0 && bias, "vtp_gemm_nt: SwiGLU epilogue needs interleaved N %% 16 == 0 and a bias");
      return launch_gemm<EPI_SWIGLU, false>(a, cs, cfg, s);
    case VTP_EPI_GELU: return launch_gemm<EPI_GELU, false>(a, cs, cfg, s);
    case VTP_EPI_F32_ATOMIC: return launch_gemm<EPI_F32_ATOMIC, false>(a, splits, cfg, s);
    case VTP_EPI_F32_SLAB: return launch_gemm<EPI_F32_SLAB, false>(a, splits, cfg, s);
    default: VTP_REQUIRE(false, "vtp_gemm_nt: unknown epilogue %d", epilogue);
  }
  return VTP_OK;
}

// w3 dgrad with the SwiGLU backward in its epilogue (ffn.py:78-81 backward): dh = dy W3 is formed per tile and leaves as
// dx12[M, 2H] = d(silu(x1) x2)/d(x1 | x2) (.) dh with x12 the saved pre-activations (interleaved 8 | 8 like the forward's w12
// output) -- one launch and no dh round trip instead of vtp_gemm_nt + vtp_swiglu_bwd.  A = dy bf16 [M, K = D], WT = W3^T bf16 [H, D].
extern "C" int vtp_gemm_dgrad_swiglu(const void* A, int lda, const void* WT, int ldb, const void* x12, int ldx, void* dx12, int ldc,
                                     int M, int H, int K, void* stream) {
  VTP_REQUIRE(A && WT && x12 && dx12, "vtp_gemm_dgrad_swiglu: null operand");
  VTP_REQUIRE(M > 0 && H > 0 && K > 0 && H % 8 == 0 && K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldx % 8 == 0 && ldc % 8 == 0 &&
              ldx >= 2 * H && ldc >= 2 * H, "vtp_gemm_dgrad_swiglu: H, K and the leading dimensions must be multiples of 8, ldx / ldc >= 2 H");
  VTP_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)WT % 16 == 0) && ((uintptr_t)x12 % 16 == 0) && ((uintptr_t)dx12 % 16 == 0),
              "vtp_gemm_dgrad_swiglu: operands must be 16-B aligned");
  GemmArgs a{};
  a.A = (const bf16*)A; a.B = (const bf16*)WT; a.C = dx12;
  a.M = M; a.N = H; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.alpha = 1.f;
  a.xcd_swizzle = swz_flags() | 2;  // the fused backward lives in the LDS-staged store path
  a.k_split = (K + 63) / 64 * 64;
  a.swiglu_pre = (const bf16*)x12; a.swiglu_ld = ldx;
  return launch_gemm<EPI_BF16, false>(a, 1, pick_cfg_nt(a, VTP_EPI_BF16, 1), (hipStream_t)stream);
}

// fp8 (e4m3, OCP) forward GEMM (BASELINE config 5): C = alpha * (A8 B8^T) (+ bias, + residual / SwiGLU), A8 [M, K] and B8 [N, K]
// row-major bytes, alpha = 1 / (scale_a * scale_b) of the per-tensor quantisation.  Runs the 256 x 256 8-phase kernel with the
// 32x32x64 f8f6f4 MFMA; K and the leading dimensions are in fp8 elements and multiples of 16 here.
namespace vtp {
int launch_gemm8p_nt_fp8(const GemmArgs& a, int epi, hipStream_t s);
}
extern "C" int vtp_gemm_nt_fp8(const void* A8, int lda, const void* B8, int ldb, void* C, int ldc, void* C2, int ldc2,
                               const float* bias, const float* resid, int M, int N, int K, int epilogue, float alpha,
                               const int* rope_pos, const void* rope_sin, const void* rope_cos, int rope_cols, void* stream) {
  VTP_REQUIRE(A8 && B8 && C, "vtp_gemm_nt_fp8: null operand");
  VTP_REQUIRE(!rope_pos || (epilogue == VTP_EPI_BF16 && rope_sin && rope_cos && rope_cols % 128 == 0 && rope_cols <= N && N % 8 == 0 &&
                            ldc % 8 == 0), "vtp_gemm_nt_fp8: fused RoPE needs the bf16 epilogue, both tables and rope_cols %% 128 == 0");
  VTP_REQUIRE(M > 0 && N > 0 && K > 0, "vtp_gemm_nt_fp8: bad shape M=%d N=%d K=%d", M, N, K);
  VTP_REQUIRE(K % 16 == 0 && lda % 16 == 0 && ldb % 16 == 0, "vtp_gemm_nt_fp8: K, lda, ldb must be multiples of 16 (16-B rows)");
  VTP_REQUIRE(N % 4 == 0 && ldc % 4 == 0, "vtp_gemm_nt_fp8: N and ldc must be multiples of 4");
  VTP_REQUIRE(((uintptr_t)A8 % 16 == 0) && ((uintptr_t)B8 % 16 == 0) && ((uintptr_t)C % 16 == 0), "vtp_gemm_nt_fp8: operands must be 16-B aligned");
  VTP_REQUIRE(epilogue == VTP_EPI_BF16 || epilogue == VTP_EPI_F32 || epilogue == VTP_EPI_SWIGLU, "vtp_gemm_nt_fp8: forward epilogues only");
  VTP_REQUIRE(epilogue != VTP_EPI_SWIGLU || (N % 16 == 0 && bias), "vtp_gemm_nt_fp8: SwiGLU epilogue needs interleaved N %% 16 == 0 and a bias");
  GemmArgs a{};
  a.A = (const bf16*)A8; a.B = (const bf16*)B8; a.C = C; a.C2 = C2; a.bias = bias; a.resid = resid;
  a.M = M; a.N = N; a.K = K / 2; a.lda = lda / 2; a.ldb = ldb / 2; a.ldc = ldc; a.ldc2 = ldc2;  // 2-byte units for the staging side
  a.alpha = alpha;
  a.xcd_swizzle = swz_flags();
  a.k_split = (a.K + 63) / 64 * 64;
  if (rope_pos) {  // apply_rope in the epilogue, as in vtp_gemm_qkv_rope
    a.xcd_swizzle |= 2;
    a.rope_pos = rope_pos; a.rope_sin = (const bf16*)rope_sin; a.rope_cos = (const bf16*)rope_cos; a.rope_cols = rope_cols;
  }
  return launch_gemm8p_nt_fp8(a, epilogue, (hipStream_t)stream);
}

// qkv projection + apply_rope in one launch (attention.py:115 + :70-89): C bf16 [M, N] = A W^T + bias, then the q and k thirds
// (columns < rope_cols) of every row m with rope_pos[m] >= 0 are rotated with row rope_pos[m] of the bf16 sin / cos tables.
extern "C" int vtp_gemm_qkv_rope(const void* A, int lda, const void* W, int ldb, const float* bias, void* C, int ldc, int M, int N,
                                 int K, const int* rope_pos, const void* rope_sin, const void* rope_cos, int rope_cols,
                                 void* stream) {
  VTP_REQUIRE(A && W && C && rope_pos && rope_sin && rope_cos, "vtp_gemm_qkv_rope: null operand");
  VTP_REQUIRE(M > 0 && N > 0 && K > 0, "vtp_gemm_qkv_rope: bad shape M=%d N=%d K=%d", M, N, K);
  VTP_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && N % 8 == 0 && ldc % 8 == 0, "vtp_gemm_qkv_rope: K, N, lda, ldb, ldc must be multiples of 8");
  VTP_REQUIRE(rope_cols % 128 == 0 && rope_cols <= N, "vtp_gemm_qkv_rope: rope_cols must be a multiple of 128 (head_dim 64, two heads per 128-column wave tile)");
  VTP_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)W % 16 == 0) && ((uintptr_t)C % 16 == 0) && ((uintptr_t)rope_sin % 16 == 0) &&
              ((uintptr_t)rope_cos % 16 == 0), "vtp_gemm_qkv_rope: operands must be 16-B aligned");
  GemmArgs a{};
  a.A = (const bf16*)A; a.B = (const bf16*)W; a.C = C; a.bias = bias;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.alpha = 1.f;
  a.xcd_swizzle = swz_flags() | 2;  // the rotation lives in the LDS-staged store path
  a.k_split = (K + 63) / 64 * 64;
  a.rope_pos = rope_pos; a.rope_sin = (const bf16*)rope_sin; a.rope_cos = (const bf16*)rope_cos; a.rope_cols = rope_cols;
  return launch_gemm<EPI_BF16, false>(a, 1, pick_cfg_nt(a, VTP_EPI_BF16, 1), (hipStream_t)stream);
}

// C[M,N] (f32) = A[K,M]^T * B[K,N]  (A, B bf16 row-major with the reduction dimension K = tokens as rows):
// the weight-gradient GEMM dW = dY^T X straight from the activation layouts (no transposed copies).
extern "C" int vtp_colsum_bf16(const void* in, int ld, float* out, int colsum_swiglu_h, int in_grp, int in_pre, int R, int C,
                               void* stream);

extern "C" int vtp_gemm_tn(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int ldc2, const float* resid,
                           int M, int N, int K, int epilogue, int a_grp, int a_pre, int b_grp, int b_pre, int c_grp, int c_pre,
                           int splits, float* a_colsum, void* stream) {
  VTP_REQUIRE(A && B && C, "vtp_gemm_tn: null operand");
  VTP_REQUIRE(M > 0 && N > 0 && K > 0, "vtp_gemm_tn: bad shape M=%d N=%d K=%d", M, N, K);
  VTP_REQUIRE(M % 8 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, "vtp_gemm_tn: M, N, lda, ldb must be multiples of 8");
  VTP_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 16 == 0), "vtp_gemm_tn: operands must be 16-B aligned");
  VTP_REQUIRE(epilogue == VTP_EPI_F32 || epilogue == VTP_EPI_F32_SLAB || epilogue == VTP_EPI_F32_ATOMIC,
              "vtp_gemm_tn: epilogue must be F32 (accumulate via resid), F32_SLAB or F32_ATOMIC");
  VTP_REQUIRE(splits >= 1 && (splits == 1 || epilogue != VTP_EPI_F32), "vtp_gemm_tn: split-K needs the slab or the atomic epilogue");
  VTP_REQUIRE(epilogue != VTP_EPI_F32_SLAB || ldc2 > 0, "vtp_gemm_tn: slab epilogue needs ldc2 = slab stride / 4");
  GemmArgs a{};
  a.A = (const bf16*)A; a.B = (const bf16*)B; a.C = C; a.C2 = nullptr; a.bias = nullptr; a.gamma = nullptr; a.resid = resid;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldc2 = ldc2;
  a.a_grp = a_grp; a.a_pre = a_pre; a.b_grp = b_grp; a.b_pre = b_pre; a.c_grp = c_grp; a.c_pre = c_pre; a.alpha = 1.f;
  a.xcd_swizzle = swz_flags();
  int ks = ((K + splits - 1) / splits + 63) / 64 * 64;
  a.k_split = ks;
  splits = (K + ks - 1) / ks;
  // tools/gemm_tn_bench.py on MI355X: the 8-wave 128x128 tile wins on every wgrad shape (transpose reads want more waves)
  int cfg = g_force_cfg >= 0 ? g_force_cfg : (use_8p_tn(M, N, K, splits, a) ? 8 : 5);
  if (cfg != 2 && cfg != 3 && cfg != 5 && cfg != 8 && cfg != 16 && cfg != 21) cfg = 0;
  hipStream_t s = (hipStream_t)stream;
  if (a_colsum) {  // bias gradient db[m] += sum_t A[t, m]: fused into the 8-phase kernel, a separate pass otherwise
    VTP_REQUIRE(c_grp <= 0, "vtp_gemm_tn: a_colsum supports the identity and the SwiGLU (c_grp = -1) row maps only");
    if (cfg == 8 && a_grp == 0 && b_grp == 0 && gemm8p_fits(a, true)) {
      a.colsum = a_colsum;
    } else {
      const int rc = vtp_colsum_bf16(A, lda, a_colsum, c_grp < 0 ? c_pre : 0, a_grp, a_pre, K, M, stream);
      if (rc != VTP_OK) return rc;
    }
  }
  if (epilogue == VTP_EPI_F32) return launch_gemm<EPI_F32, true>(a, 1, cfg, s);
  if (epilogue == VTP_EPI_F32_ATOMIC) return launch_gemm<EPI_F32_ATOMIC, true>(a, splits, cfg, s);
  return launch_gemm<EPI_F32_SLAB, true>(a, splits, cfg, s);
}
