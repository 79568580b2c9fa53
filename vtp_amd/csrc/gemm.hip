// bf16 MFMA GEMM for gfx950:  C[M,N] = A[M,K] * B[N,K]^T  (+ fused epilogue), fp32 accumulate.
//
// Both operands are K-contiguous ("NT"): activations [tokens, features] and nn.Linear weights
// [out, in] are used as stored.  dgrad / wgrad reach this kernel through pre-transposed operands
// (weights: cached W^T; activations: transpose kernels in elementwise.hip).
//
// Structure (CDNA4): 256 threads = 4 waves (2 x 2), workgroup tile BM x BN x 64, double-buffered LDS
// filled by `global_load_lds_dwordx4` (LDS-DMA, 16 B/lane, no VGPR round trip).  The LDS image of a
// tile is [row][64 k] bf16 (128 B rows) with the 16-byte chunk index XOR-swizzled by ((row>>1)&7) so
// that every ds_read_b128 lane group hits 16 distinct 16-B slots (conflict-free); because LDS-DMA
// writes lane-linear, the swizzle is applied to the per-lane *global source* address and to the read
// address (cdna guide rule 21).  MFMA: v_mfma_f32_32x32x16_bf16 with the operands swapped (weight
// rows feed the A operand, token rows the B operand) so that each lane ends up holding 4 consecutive
// output columns of one output row -> 8/16-byte epilogue stores and lane-local SwiGLU / bias.
#include "common.h"
#include "vtp_hip.h"

namespace vtp {

__device__ __attribute__((aligned(16))) unsigned int g_zero_block[16] = {0};

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
  int c_grp, c_pre;  // C row remap (same formula)
  int k_split;       // K elements per blockIdx.z slice (multiple of 64)
  float alpha;
};

enum { EPI_BF16 = 0, EPI_F32 = 1, EPI_SWIGLU = 2, EPI_GELU = 3, EPI_F32_ATOMIC = 4 };

__device__ __forceinline__ int remap_row(int m, int grp, int pre) {
  if (grp > 0) return m + (m / grp + 1) * pre;
  if (grp < 0) return ((m >> 4) << 3) + (m & 7) + ((m & 8) ? pre : 0);  // SwiGLU de-interleave
  return m;
}

template <int BM, int BN, int EPI>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const GemmArgs p) {
  constexpr int BK = 64;
  constexpr int TM = BM / 64;  // 32-row m tiles per wave
  constexpr int TN = BN / 64;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int PA = BM / 32, PB = BN / 32;  // 1-KiB LDS-DMA pieces per wave per k-tile
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int hi = lane >> 5;

  const int tiles_m = (p.M + BM - 1) / BM;
  const int tile_m = blockIdx.x % tiles_m;
  const int tile_n = blockIdx.x / tiles_m;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kbeg = blockIdx.z * p.k_split;
  const int kend = min(p.K, kbeg + p.k_split);
  const int nk = (kend - kbeg + BK - 1) / BK;

  // ---- LDS-DMA source pointers (per lane, per piece) ----
  const int prow = lane >> 3;  // row inside an 8-row piece
  const int slot = lane & 7;   // 16-B slot inside the 128-B row
  const char* a_src[PA];
  const char* b_src[PB];
  int a_kc[PA], b_kc[PB];  // this lane's source k offset (elements) inside the k-tile
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
  const char* zsrc = (const char*)g_zero_block;

  auto stage = [&](int buf, int kt) {
    char* abase = smem + buf * (A_BYTES + B_BYTES);
    char* bbase = abase + A_BYTES;
    const int krem = kend - kbeg - kt * BK;  // valid k elements left in this tile (>0)
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const char* s = (a_kc[i] < krem) ? a_src[i] + (size_t)kt * (BK * 2) : zsrc;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                       (__attribute__((address_space(3))) void*)(abase + (wave * PA + i) * 1024),
                                       16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const char* s = (b_kc[i] < krem) ? b_src[i] + (size_t)kt * (BK * 2) : zsrc;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                       (__attribute__((address_space(3))) void*)(bbase + (wave * PB + i) * 1024),
                                       16, 0, 0);
    }
  };

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

  if (nk > 0) stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
    const char* abase = smem + (kt & 1) * (A_BYTES + B_BYTES);
    const char* bbase = abase + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int coff = ((2 * ks + hi) ^ sw) << 4;
      bf16x8 wf[TN], xf[TM];
#pragma unroll
      for (int i = 0; i < TN; ++i)
        wf[i] = *(const bf16x8*)(bbase + (wn * (BN / 2) + i * 32) * 128 + rowoff + coff);
#pragma unroll
      for (int j = 0; j < TM; ++j)
        xf[j] = *(const bf16x8*)(abase + (wm * (BM / 2) + j * 32) * 128 + rowoff + coff);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue: lane holds, for output row m, columns nb + 8*q + 4*hi + (0..3), q = 0..3 ----
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = m0 + wm * (BM / 2) + j * 32 + (lane & 31);
    if (m >= p.M) continue;
    const int mc = remap_row(m, p.c_grp, p.c_pre);
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int nb = n0 + wn * (BN / 2) + i * 32 + 4 * hi;
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
            x1[e] = acc[i][j][4 * q + e] + b1[e];
            x2[e] = acc[i][j][4 * q + 4 + e] + b2[e];
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
              for (int e = 0; e < 4; ++e) g[e] = gelu_erf(bf2f(pre[e]));
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

template <int BM, int BN, int EPI>
static int launch_gemm(const GemmArgs& a, int splits, hipStream_t s) {
  constexpr int LDS = 2 * (BM + BN) * 64 * 2;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_nt_kernel<BM, BN, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  dim3 grid(cdiv(a.M, BM) * cdiv(a.N, BN), 1, splits);
  hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, EPI>), grid, dim3(256), LDS, s, a);
  return check_launch("gemm_nt");
}

}  // namespace vtp

using namespace vtp;

extern "C" int vtp_gemm_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, void* C2, int ldc2,
                           const float* bias, const float* gamma, const float* resid, int M, int N, int K, int epilogue,
                           int a_grp, int a_pre, int c_grp, int c_pre, int splits, float alpha, void* stream) {
  VTP_REQUIRE(A && B && C, "vtp_gemm_nt: null operand");
  VTP_REQUIRE(M > 0 && N > 0 && K > 0, "vtp_gemm_nt: bad shape M=%d N=%d K=%d", M, N, K);
  VTP_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "vtp_gemm_nt: K, lda, ldb must be multiples of 8 (16-B rows)");
  VTP_REQUIRE(N % 4 == 0 && ldc % 4 == 0, "vtp_gemm_nt: N and ldc must be multiples of 4");
  VTP_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 16 == 0), "vtp_gemm_nt: operands must be 16-B aligned");
  VTP_REQUIRE(splits >= 1, "vtp_gemm_nt: splits must be >= 1");
  VTP_REQUIRE(splits == 1 || epilogue == VTP_EPI_F32_ATOMIC, "vtp_gemm_nt: split-K needs the atomic epilogue");
  GemmArgs a;
  a.A = (const bf16*)A; a.B = (const bf16*)B; a.C = C; a.C2 = C2; a.bias = bias; a.gamma = gamma; a.resid = resid;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldc2 = ldc2;
  a.a_grp = a_grp; a.a_pre = a_pre; a.c_grp = c_grp; a.c_pre = c_pre; a.alpha = alpha;
  int ks = ((K + splits - 1) / splits + 63) / 64 * 64;
  a.k_split = ks;
  splits = (K + ks - 1) / ks;
  hipStream_t s = (hipStream_t)stream;
  switch (epilogue) {
    case VTP_EPI_BF16: return launch_gemm<128, 128, EPI_BF16>(a, 1, s);
    case VTP_EPI_F32: return launch_gemm<128, 128, EPI_F32>(a, 1, s);
    case VTP_EPI_SWIGLU:
      VTP_REQUIRE(N % 16 == 0 && bias, "vtp_gemm_nt: SwiGLU epilogue needs interleaved N %% 16 == 0 and a bias");
      return launch_gemm<128, 128, EPI_SWIGLU>(a, 1, s);
    case VTP_EPI_GELU: return launch_gemm<128, 128, EPI_GELU>(a, 1, s);
    case VTP_EPI_F32_ATOMIC: return launch_gemm<128, 128, EPI_F32_ATOMIC>(a, splits, s);
    default: VTP_REQUIRE(false, "vtp_gemm_nt: unknown epilogue %d", epilogue);
  }
  return VTP_OK;
}
