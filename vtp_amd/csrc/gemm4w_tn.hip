// Grouped weight gradients C_g[M_g, N_g] (+)= A_g[K, M_g]^T B_g[K, N_g] on the one-wave-per-SIMD 256 x 256 kernel: the TN counterpart of
// gemm4w.hip (same X / Y schedule, hand-scheduled k loop as generated inline asm: gemm4w_tn_ktile.inc / tools/gen_gemm4w_ktile.py),
// launched like gemm8p_grouped_tn_kernel -- one workgroup per (256 x 256 tile, K slice), slices combined in the launch by the last
// arriver, bias-gradient column sums riding along -- and bit-identical to it per K slice (same k order per output element, same MFMA).
//
//   * 4 waves = 2 (M) x 2 (N), wave tile 128 x 128, 256 accumulator AGPRs, fragments in PHYSICAL VGPRs v64..v255 (the asm owns them);
//   * a staged k-tile (64 k rows) = eight 8-KiB sub-images [64 k][64 columns] with 128-B rows -- A columns wr * 128 + {0..63 | 64..127},
//     B columns wc * 128 + {0..63 | 64..127} -- whose 16-B chunk index is XORed with 4 ((k >> 1) & 1): the eight k rows one
//     ds_read_b64_tr_b16 touches (k0 + {0..3, 8..11}, 64 B each) then cover the four 64-B positions of the 256-B bank window (two k
//     rows of 128 B) twice each.  (XOR 4 (k & 1) left half the banks idle: 49 % of the LDS cycles were conflict cycles.)
//   * wave w stages A sub-image w (in the Y step) and B sub-image w (in the X step): 8 LDS-DMA pieces of 8 k rows each; pieces whose k
//     rows lie beyond the slice read a zero block (so a slice needs no K-tail path, and an odd k-tile count is padded with a zero one);
//   * ring of two k-tiles (128 KiB) + 32 KiB (4 KiB per wave of fp32 epilogue staging; the arrival flag of the combine in its last
//     word): 160 KiB, one workgroup per CU.
// Limits (launcher; otherwise the 8-phase kernel takes the launch): M, N, K multiples of 8, 32-bit operand offsets.
#include "gemm_common.h"
#include "gemm_group.h"
#include "gemm4w_tn_ktile.inc"

namespace vtp {

namespace {
constexpr int W4T_KT = 65536;  // one staged k-tile: sub-images A0lo A0hi A1lo A1hi B0lo B0hi B1lo B1hi
constexpr int W4T_RING = 2 * W4T_KT;
constexpr int W4T_LDS = W4T_RING + 32768;
}  // namespace

__device__ __forceinline__ void w4t_glds16(const char* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}

// wave-uniform values of the problem record: pinned into SGPRs, as the asm operands need them (inside a loop over segments -- the
// stream-K experiment of round 4, commit 88a5214 -- hipcc reads the record with vector loads)
__device__ __forceinline__ int w4t_uni(long v) { return __builtin_amdgcn_readfirstlane((int)v); }
template <class T>
__device__ __forceinline__ T* w4t_uni(T* v) {
  const unsigned long long u = (unsigned long long)v;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return (T*)(((unsigned long long)hi << 32) | lo);
}

struct W4TStage {  // wave-uniform staging constants
  const char *mata, *matb, *zb;
  unsigned stepa, stepb, da[2], db[2];
};

// the k loop of a (tile, K range): ONE asm statement holding both generated loops (with / without the bias-gradient column sums) behind
// a scalar branch -- see tools/gen_gemm4w_ktile.py
__device__ __forceinline__ void w4t_kloop(bool csum, f32x16 (&acc)[4][4], float (&cs)[4], const unsigned (&aE)[2], const unsigned (&aO)[2],
                                          const unsigned (&bE)[2], const unsigned (&bO)[2], unsigned& pea, unsigned& peb, unsigned zoff,
                                          const W4TStage& st, int& kra, int& krb, unsigned nk2) {
  unsigned vt, sm, cnt = nk2;
  unsigned long long sb;
  const unsigned ones = 0x3F803F80u;
  const unsigned docs = __builtin_amdgcn_readfirstlane(csum ? 1u : 0u);
  asm volatile(W4T_TILE_ASM_BOTH
               : W4T_TILE_ACC, [cs0] "+v"(cs[0]), [cs1] "+v"(cs[1]), [cs2] "+v"(cs[2]), [cs3] "+v"(cs[3]), [pea] "+v"(pea), [peb] "+v"(peb),
                 [vt] "=&v"(vt), [kra] "+s"(kra), [krb] "+s"(krb), [cnt] "+s"(cnt), [sm] "=&s"(sm), [sb] "=&s"(sb)
               : [aE0] "v"(aE[0]), [aO0] "v"(aO[0]), [bE0] "v"(bE[0]), [bO0] "v"(bO[0]), [aE1] "v"(aE[1]), [aO1] "v"(aO[1]), [bE1] "v"(bE[1]),
                 [bO1] "v"(bO[1]), [zoff] "v"(zoff), [mata] "s"(st.mata), [matb] "s"(st.matb), [zb] "s"(st.zb), [stepa] "s"(st.stepa),
                 [stepb] "s"(st.stepb), [ones] "s"(ones), [docs] "s"(docs), [da0] "s"(st.da[0]), [da1] "s"(st.da[1]), [db0] "s"(st.db[0]),
                 [db1] "s"(st.db[1])
               : "memory", "scc", "vcc", W4T_TILE_CLOBBERS);
}

// one (tile, K range) of a grouped launch: k rows [kbeg, kbeg + kcount) of the tile's reduction; `nparts` workgroups share the tile and
// this one is number `part` of them (slot of its partial sums; nparts == 1: no combine).  Returns after the tile's epilogue if this
// workgroup arrived last, after the publish otherwise.
__device__ __forceinline__ void w4t_segment(const GroupArgs& ga, int tile, int kbeg, int kcount, int nparts, int part) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  // diagnostics (vtp_gemm_debug, tools/wgrad_timeline.py): 100-MHz stamps per workgroup -- [0] start, [1] first operands landed,
  // [2] k loop done, [3] published + ticket drawn, [4] done, [5] tile, [6] K slice, [7] XCC id
  auto stamp = [&](int which) {
    if (__builtin_expect(ga.timing != nullptr, 0) && tid == 0) ga.timing[(size_t)blockIdx.x * 8 + which] = wall_clock64();
  };
  if (__builtin_expect(ga.timing != nullptr, 0) && tid == 0) {
    ga.timing[(size_t)blockIdx.x * 8 + 5] = (unsigned long long)tile;
    ga.timing[(size_t)blockIdx.x * 8 + 6] = (unsigned long long)part;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    ga.timing[(size_t)blockIdx.x * 8 + 7] = (unsigned long long)(xcc & 15);
  }
  stamp(0);
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int hi = lane >> 5;

  int g = 0;
  for (int i = 1; i < ga.nprob; ++i)
    if ((int)ga.probs[i].tile0 <= tile) g = i;
  const GroupProblem& pr = ga.probs[g];
  GemmArgs p{};
  p.A = w4t_uni(pr.A); p.B = w4t_uni(pr.B); p.C = w4t_uni(pr.C); p.resid = w4t_uni(pr.accumulate) ? (const float*)p.C : nullptr;
  p.colsum = w4t_uni(pr.colsum);
  p.M = w4t_uni(pr.M); p.N = w4t_uni(pr.N); p.K = ga.K; p.lda = w4t_uni(pr.lda); p.ldb = w4t_uni(pr.ldb); p.ldc = w4t_uni(pr.ldc);
  p.c_grp = w4t_uni(pr.c_grp); p.c_pre = w4t_uni(pr.c_pre); p.k_split = ga.k_split; p.alpha = 1.f;
  p.xcd_swizzle = 2;  // (bit 1: the LDS-staged store path of the shared epilogue)
  const int lt = tile - w4t_uni(pr.tile0), tiles_n = (p.N + 255) >> 8;
  const int n0 = (lt % tiles_n) << 8, m0 = (lt / tiles_n) << 8;
  const unsigned nk2 = (unsigned)((kcount + 127) >> 7);  // pairs of k-tiles (an odd count is padded with an all-zero k-tile)

  // ---------------------------------------------------------------- staging: wave w = A sub-image w and B sub-image w of every k-tile
  // piece i (0..7) = k rows 8 i + (lane >> 3) of the k-tile; the lane's 16-B chunk slot lane & 7 holds source chunk
  // (lane & 7) ^ 4 ((lane >> 4) & 1); columns beyond the matrix are clamped (they only feed outputs the epilogue masks)
  const int krow = lane >> 3, chunk = (lane & 7) ^ (4 * ((krow >> 1) & 1));
  const int cola = min(m0 + wave * 64 + chunk * 8, p.M - 8), colb = min(n0 + wave * 64 + chunk * 8, p.N - 8);
  unsigned pea = (unsigned)(((size_t)(kbeg + krow) * p.lda + cola) * 2), peb = (unsigned)(((size_t)(kbeg + krow) * p.ldb + colb) * 2);
  const unsigned zoff = (lane & 3) * 16;
  W4TStage st;
  st.mata = (const char*)p.A;
  st.matb = (const char*)p.B;
  st.zb = (const char*)g_zero_block;
  st.stepa = (unsigned)(8 * p.lda * 2);
  st.stepb = (unsigned)(8 * p.ldb * 2);
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    st.da[sl] = (unsigned)(size_t)smem + sl * W4T_KT + wave * 8192;
    st.db[sl] = st.da[sl] + 32768;
  }
  // prologue: k-tiles 0 and 1 of the slice
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool valid = kt * 64 + 8 * i < kcount;  // wave-uniform
      w4t_glds16(valid ? st.mata : st.zb, valid ? pea : zoff, st.da[kt] + i * 1024);
      w4t_glds16(valid ? st.matb : st.zb, valid ? peb : zoff, st.db[kt] + i * 1024);
      pea += st.stepa;
      peb += st.stepb;
    }
  int kra = kcount - 128, krb = kcount - 128;  // k rows of the slice the cursor has not staged yet

  // ---------------------------------------------------------------- fragment reads (ds_read_b64_tr_b16: see tools/gen_gemm4w_ktile.py)
  // the lane addresses 4 consecutive columns (8 B) of k row hi * 8 + t_rq of a 16-row k-step; 32-column block b of a sub-image
  // = chunks 4 b + [0, 4), swizzled with bit 1 of the k row
  unsigned aE[2], aO[2], bE[2], bO[2];
  {
    const int t_li = lane & 15, t_rq = t_li >> 2;
    const int cq = ((lane >> 4) & 1) * 2 + ((t_li & 3) >> 1), sz = 4 * ((t_rq >> 1) & 1);
    const unsigned rowb = (unsigned)((hi * 8 + t_rq) * 128 + (t_li & 1) * 8);
    const unsigned lb0 = rowb + (unsigned)(((0 + cq) ^ sz) << 4), lb1 = rowb + (unsigned)(((4 + cq) ^ sz) << 4);
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      const unsigned ba = (unsigned)(size_t)smem + sl * W4T_KT + (wr * 2) * 8192, bb = (unsigned)(size_t)smem + sl * W4T_KT + (4 + wc * 2) * 8192;
      aE[sl] = ba + lb0;
      aO[sl] = ba + lb1;
      bE[sl] = bb + lb0;
      bO[sl] = bb + lb1;
    }
  }
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  const bool do_csum = p.colsum != nullptr && n0 == 0;  // (workgroup-uniform: the waves of the first tile column; wc == 0 publishes)

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  stamp(1);
  w4t_kloop(do_csum, acc, cs, aE, aO, bE, bO, pea, peb, zoff, st, kra, krb, nk2);
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7\n\ts_waitcnt vmcnt(0)" ::: "memory");  // last MFMA passes; surplus zero pieces
  stamp(2);

  if (do_csum && wc == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v = cs[j] + __shfl_xor(cs[j], 32, 64);  // the two k halves of the fragment layout
      const int m = m0 + wr * 128 + j * 32 + (lane & 31);
      if (hi == 0 && m < p.M) unsafeAtomicAdd(p.colsum + remap_row(m, p.c_grp, p.c_pre), v);
    }
  }
  if (nparts > 1) {
    // combine inside the launch (the protocol of gemm8p_body): publish the accumulators fragment-major with write-through
    // stores -> drain -> barrier -> ticket; the last arriver adds the other slices' partials in slice order and runs the epilogue
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
    const f32x4* mine = (const f32x4*)ga.part + (((size_t)tile * ga.splits + part) * 4 + wave) * 4096;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)mine, 0, 4096 * 16, 0x00020000);
#pragma unroll
    for (int u = 0; u < 64; ++u) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[u >> 4][(u >> 2) & 3][4 * (u & 3) + e];
      __builtin_amdgcn_raw_buffer_store_b128((u32x4_t)v, rsrc, lane * 16 + u * 1024, 0, /*sc1*/ 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = (int*)(smem + W4T_LDS - 16);  // (outside the 4 KiB the epilogue stages through per wave)
    if (tid == 0) {
      const int t = __hip_atomic_fetch_add(ga.ticket + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = t == nparts - 1;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(ga.ticket + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
      }
      *flag = last;
    }
    __syncthreads();
    stamp(3);
    if (*flag == 0) return;
    for (int z = 0; z < nparts; ++z) {
      if (z == part) continue;
      const f32x4* other = (const f32x4*)ga.part + (((size_t)tile * ga.splits + z) * 4 + wave) * 4096 + lane;
      // NLD 16-B loads in flight per lane (the fragment registers v64..v255 are dead here): a partial tile is 64 loads per lane, and
      // with 8 in flight its 8 dependent round trips were most of the last arriver's combine (tools/wgrad_timeline.py)
      constexpr int NLD = 16;
#pragma unroll
      for (int gq = 0; gq < 64 / NLD; ++gq) {
        f32x4 v[NLD];
#pragma unroll
        for (int t = 0; t < NLD; ++t) v[t] = other[(gq * NLD + t) * 64];
#pragma unroll
        for (int t = 0; t < NLD; ++t) {
          const int u = gq * NLD + t;
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[u >> 4][(u >> 2) & 3][4 * (u & 3) + e] += v[t][e];
        }
        asm volatile("" ::: "memory");
      }
    }
  }
  // C (+)= acc through the LDS-staged fp32 store path of the shared epilogue (full 128-B row segments, the rows of C read ahead; the
  // fragment roles are those of the NT kernels, so the NT instantiation applies; element-wise the arithmetic of the direct path)
  char* reg = smem + W4T_RING + wave * 8192;
  gemm_epilogue<EPI_F32, false, 128, 64, 4096, 0>(p, *(f32x16(*)[2][4]) & acc[0], reg, m0, n0, wr, wc * 2, lane, 0, nullptr);
  gemm_epilogue<EPI_F32, false, 128, 64, 4096, 0>(p, *(f32x16(*)[2][4]) & acc[2], reg, m0, n0, wr, wc * 2 + 1, lane, 0, nullptr);
  stamp(4);
}


// one workgroup per (tile, K slice): the launch geometry of gemm8p_grouped_tn_kernel (tiles x splits workgroups, split-major over the XCDs)
__global__ __launch_bounds__(256, 1) void gemm4w_grouped_tn_kernel(const GroupArgs ga) {
  const int W = gridDim.x, q = W >> 3, r = W & 7, x = blockIdx.x & 7;
  const int c = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + ((int)blockIdx.x >> 3);
  const int zslice = c / ga.ntiles, tile = c - zslice * ga.ntiles;
  const int kbeg = zslice * ga.k_split;
  w4t_segment(ga, tile, kbeg, min(ga.K, kbeg + ga.k_split) - kbeg, ga.splits, zslice);
}

// item-list launch (vtp_gemm_tn_grouped_items): every workgroup reads its (tile, K range, slot) from a host-built table, so the tiles of
// a launch need not be cut alike -- the tiles that also sum the columns of A (bias gradients: 64 v_dot2 per k-tile beside the 64 MFMAs,
// measured 27 % slower per k-tile, tools/wgrad_timeline.py) get one slice more than the others and the launch ends level
__global__ __launch_bounds__(256, 1) void gemm4w_grouped_tn_items_kernel(const GroupArgs ga) {
  const int W = gridDim.x, q = W >> 3, r = W & 7, x = blockIdx.x & 7;
  const int c = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + ((int)blockIdx.x >> 3);
  const GroupItem& it = ga.items[c];
  w4t_segment(ga, w4t_uni(it.tile), w4t_uni(it.kbeg), w4t_uni(it.kcount), w4t_uni(it.nparts), w4t_uni(it.part));
}

int launch_gemm4w_grouped_tn_items(const GroupArgs& ga, int nitems, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm4w_grouped_tn_items_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, W4T_LDS);
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm4w_grouped_tn_items_kernel, dim3(nitems), dim3(256), W4T_LDS, s, ga);
  return check_launch("gemm4w_grouped_tn_items");
}

// launcher used by vtp_gemm_tn_grouped_k (gemm8p.hip, kernel = 1).  The problem records live in DEVICE memory, so the C entry can only
// check what it is passed by value (token count % 8); the per-problem limits of this kernel -- M_g, N_g and both leading dimensions
// multiples of 8 (16-B staging pieces, `min(.., M - 8)` column clamps), operand spans below 4 GiB (32-bit lane offsets) -- are enforced
// where the records are BUILT: ops.WgradGroup.add / finalize raise ValueError (vtp_amd/ops.py), and a C caller must do the same.
int launch_gemm4w_grouped_tn(const GroupArgs& ga, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm4w_grouped_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, W4T_LDS);
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm4w_grouped_tn_kernel, dim3(ga.ntiles * ga.splits), dim3(256), W4T_LDS, s, ga);
  return check_launch("gemm4w_grouped_tn");
}

}  // namespace vtp
