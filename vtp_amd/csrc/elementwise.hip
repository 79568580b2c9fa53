// HBM-bound data-movement / elementwise kernels of the VTP hot path (gfx950).  All accesses are 8-16 B per lane.
#include "common.h"
#include "vtp_hip.h"

namespace vtp {

// ---------------------------------------------------------------- RoPE (attention.py:12-23,70-89)
// work item = (row, part{q,k}, head, 8-pair group): 16 B from the x[0:32) half + 16 B from the x[32:64) half.
template <bool INVERSE>
__global__ __launch_bounds__(256) void rope_qk_kernel(bf16* __restrict__ qkv, const bf16* __restrict__ sin_t,
                                                      const bf16* __restrict__ cos_t, int B, int N, int heads,
                                                      int prefix) {
  const int D = heads * 64;
  const long per_row = 2L * heads * 4;
  const long total = (long)B * N * per_row;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
    const long row = idx / per_row;
    const int rem = (int)(idx - row * per_row);
    const int n = (int)(row % N);
    if (n < prefix) continue;
    const int part = rem / (heads * 4);
    const int h = (rem / 4) % heads;
    const int g = rem & 3;
    bf16* p = qkv + row * (3L * D) + (long)part * D + h * 64 + g * 8;
    const long t = (long)(n - prefix) * 64 + g * 8;
    bf16x8 x1 = *(const bf16x8*)p, x2 = *(const bf16x8*)(p + 32);
    bf16x8 c1 = *(const bf16x8*)(cos_t + t), c2 = *(const bf16x8*)(cos_t + t + 32);
    bf16x8 s1 = *(const bf16x8*)(sin_t + t), s2 = *(const bf16x8*)(sin_t + t + 32);
    bf16x8 o1, o2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float a = bf2f(x1[e]), b = bf2f(x2[e]);
      if (!INVERSE) {
        // out = x*cos + rot_half(x)*sin ; rot_half(x) = [-x2, x1]  -- three bf16 roundings like eager
        o1[e] = f2bf(bf2f(f2bf(a * bf2f(c1[e]))) + bf2f(f2bf(-b * bf2f(s1[e]))));
        o2[e] = f2bf(bf2f(f2bf(b * bf2f(c2[e]))) + bf2f(f2bf(a * bf2f(s2[e]))));
      } else {
        // transpose: dx1 = g1*cos1 + g2*sin2 ; dx2 = g2*cos2 - g1*sin1
        o1[e] = f2bf(bf2f(f2bf(a * bf2f(c1[e]))) + bf2f(f2bf(b * bf2f(s2[e]))));
        o2[e] = f2bf(bf2f(f2bf(b * bf2f(c2[e]))) + bf2f(f2bf(-a * bf2f(s1[e]))));
      }
    }
    *(bf16x8*)p = o1;
    *(bf16x8*)(p + 32) = o2;
  }
}

// ---------------------------------------------------------------- im2col for the k=s=16 patch-embed conv
// work item = (token, c, ky): 16 contiguous pixels -> 16 bf16 at K offset c*256 + ky*16.
// pre: patch p of image b goes to row b * (h w + pre) + pre + p -- pre = 1 is the token layout of the trunk (row 0 of an image = cls)
__global__ __launch_bounds__(256) void im2col16_kernel(const float* __restrict__ img, bf16* __restrict__ patches,
                                                       int B, int H, int W, int pre) {
  const int h = H / 16, w = W / 16;
  const long total = (long)B * h * w * 48;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
    const int ck = (int)(idx % 48);
    const long tok = idx / 48;
    const int c = ck / 16, ky = ck % 16;
    const int x = (int)(tok % w);
    const int y = (int)((tok / w) % h);
    const long b = tok / ((long)w * h);
    const float* src = img + ((b * 3 + c) * H + (y * 16 + ky)) * (long)W + x * 16;
    bf16* dst = patches + (tok + (b + 1) * pre) * 768 + c * 256 + ky * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 v = *(const f32x4*)(src + 4 * i);
      *(bf16x4*)(dst + 4 * i) = __builtin_convertvector(v, bf16x4);
    }
  }
}

// inverse of im2col16 for the gradient w.r.t. the input image (k = s = 16: patches do not overlap, so the fold is a permutation):
// d_img[b, c, 16y + ky, 16x + 0..15] = d_patches[token (b, y, x), c*256 + ky*16 + 0..15]
__global__ __launch_bounds__(256) void col2im16_kernel(const float* __restrict__ dpatches, float* __restrict__ dimg, int B, int H, int W) {
  const int h = H / 16, w = W / 16;
  const long total = (long)B * h * w * 48;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
    const int ck = (int)(idx % 48);
    const long tok = idx / 48;
    const int c = ck / 16, ky = ck % 16;
    const int x = (int)(tok % w);
    const int y = (int)((tok / w) % h);
    const long b = tok / ((long)w * h);
    const float* src = dpatches + tok * 768 + c * 256 + ky * 16;
    float* dst = dimg + ((b * 3 + c) * H + (y * 16 + ky)) * (long)W + x * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) *(f32x4*)(dst + 4 * i) = *(const f32x4*)(src + 4 * i);
  }
}

// PixelShuffle(16) of the token-major decoder output (+ optional fused L1 loss / gradient).
template <int MODE>  // 0: write f32 image; 1: L1 loss + dt; 2: dt = un-shuffled image gradient (`img` holds d_img)
__global__ __launch_bounds__(256) void shuffle16_kernel(const bf16* __restrict__ t, float* __restrict__ img,
                                                        const float* __restrict__ target, bf16* __restrict__ dt,
                                                        float* __restrict__ loss_sum, int B, int h, int w, float gscale) {
  const int H = h * 16, W = w * 16;
  const long total = (long)B * h * w * 48;
  float local = 0.f;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
    const int ci = (int)(idx % 48);
    const long tok = idx / 48;
    const int c = ci / 16, i = ci % 16;
    const int x = (int)(tok % w);
    const int y = (int)((tok / w) % h);
    const long b = tok / ((long)w * h);
    const long toff = tok * 768 + c * 256 + i * 16;
    const long ioff = ((b * 3 + c) * H + (y * 16 + i)) * (long)W + x * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (MODE == 2) {  // backward of PixelShuffle: gather the f32 image gradient into the token-major bf16 layout
        *(bf16x4*)(dt + toff + 4 * q) = __builtin_convertvector(*(const f32x4*)(img + ioff + 4 * q), bf16x4);
        continue;
      }
      f32x4 v = __builtin_convertvector(*(const bf16x4*)(t + toff + 4 * q), f32x4);
      if (MODE == 0) {
        *(f32x4*)(img + ioff + 4 * q) = v;
      } else {
        f32x4 tg = *(const f32x4*)(target + ioff + 4 * q);
        f32x4 g;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[e] - tg[e];
          local += fabsf(d);
          g[e] = d > 0.f ? gscale : (d < 0.f ? -gscale : 0.f);
        }
        *(bf16x4*)(dt + toff + 4 * q) = __builtin_convertvector(g, bf16x4);
      }
    }
  }
  if (MODE == 1) {
    __shared__ float red[4];
    const float s = block_sum<4>(local, red);
    if (threadIdx.x == 0) unsafeAtomicAdd(loss_sum, s);
  }
}

// ---------------------------------------------------------------- token assembly (cls row, mask-token substitution)
__global__ __launch_bounds__(256) void assemble_tokens_kernel(float* __restrict__ x, const float* __restrict__ cls,
                                                              const float* __restrict__ mask_token,
                                                              const unsigned char* __restrict__ masks, int B, int N, int D) {
  const int d4 = D / 4;
  const long total = (long)B * N * d4;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
    const int c = (int)(idx % d4);
    const long row = idx / d4;
    const int n = (int)(row % N);
    const long b = row / N;
    if (n == 0) {
      *(f32x4*)(x + row * D + 4 * c) = *(const f32x4*)(cls + 4 * c);
    } else if (masks && masks[b * (N - 1) + (n - 1)]) {
      *(f32x4*)(x + row * D + 4 * c) = *(const f32x4*)(mask_token + 4 * c);
    }
  }
}

// ---------------------------------------------------------------- bf16 transpose (+ column sums)
// 64x64 tiles through LDS: 16-B global reads, 2-B scattered LDS writes into the transposed image, 16-B LDS reads
// and 16-B global writes.
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16* __restrict__ in, int ld_in,
                                                             bf16* __restrict__ out, int ld_out,
                                                             float* __restrict__ colsum, int R, int C, int csum_H, int in_grp, int in_pre) {
  __shared__ bf16 tile[64][72];  // [c][r], row stride 144 B
  __shared__ float csum[64];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
  if (tid < 64) csum[tid] = 0.f;
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int chunk = tid + it * 256;
    const int r = chunk >> 3, cc = (chunk & 7) * 8;
    bf16x8 v;
    if (r0 + r < R && c0 + cc < C) {
      const int rr_ = r0 + r;
      const int rin = in_grp > 0 ? rr_ + (rr_ / in_grp + 1) * in_pre : rr_;
      v = *(const bf16x8*)(in + (size_t)rin * ld_in + c0 + cc);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[cc + e][r] = v[e];
    if (colsum) {
      // reduce over the 8 row-groups that share this column chunk inside the wave (lanes differ in bits 3..5)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float s = bf2f(v[e]);
        s += __shfl_xor(s, 8, 64);
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if ((tid & 63) < 8) atomicAdd(&csum[cc + e], s);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int chunk = tid + it * 256;
    const int c = chunk >> 3, rr = (chunk & 7) * 8;
    if (c0 + c < C && r0 + rr < ld_out) {
      *(bf16x8*)(out + (size_t)(c0 + c) * ld_out + r0 + rr) = *(const bf16x8*)&tile[c][rr];
    }
  }
  if (colsum && tid < 64 && c0 + tid < C) {
    const int c = c0 + tid;
    const int dst = csum_H > 0 ? ((c >> 4) << 3) + (c & 7) + ((c & 8) ? csum_H : 0) : c;
    unsafeAtomicAdd(colsum + dst, csum[tid]);
  }
}

// out[c'] += sum_r in[r, c]  (bias gradients); block = 64 columns x 256 rows
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16* __restrict__ in, int ld, float* __restrict__ out,
                                                          int R, int C, int csum_H, int in_grp, int in_pre,
                                                          const int* __restrict__ rows_dev) {
  __shared__ float part[32][65];
  if (rows_dev) R = min(R, rows_dev[0]);  // row count from device memory: a captured graph serves any count <= the launch's R
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 256;
  const int tid = threadIdx.x;
  const int cc = (tid & 7) * 8, rr = tid >> 3;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 + cc < C) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int r = r0 + rr + 32 * k;
      if (r < R) {
        const int rin = in_grp > 0 ? r + (r / in_grp + 1) * in_pre : r;
        bf16x8 v = *(const bf16x8*)(in + (size_t)rin * ld + c0 + cc);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[e]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) part[rr][cc + e] = acc[e];
  __syncthreads();
  if (tid < 64 && c0 + tid < C) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) s += part[k][tid];
    const int c = c0 + tid;
    const int dst = csum_H > 0 ? ((c >> 4) << 3) + (c & 7) + ((c & 8) ? csum_H : 0) : c;
    unsafeAtomicAdd(out + dst, s);
  }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ in, bf16* __restrict__ out, long n) {
  const long n4 = n / 4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256L)
    *(bf16x4*)(out + 4 * i) = __builtin_convertvector(*(const f32x4*)(in + 4 * i), bf16x4);
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) out[n4 * 4 + threadIdx.x] = f2bf(in[n4 * 4 + threadIdx.x]);
}

// f32 [R,C] -> bf16 [C,R]
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ in, bf16* __restrict__ out, int R, int C) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4)
    tile[r][tx] = (r0 + r < R && c0 + tx < C) ? in[(size_t)(r0 + r) * C + c0 + tx] : 0.f;
  __syncthreads();
  for (int c = ty; c < 64; c += 4)
    if (c0 + c < C && r0 + tx < R) out[(size_t)(c0 + c) * R + r0 + tx] = f2bf(tile[tx][c]);
}


// ---------------------------------------------------------------- batched compute-weight refresh
// One launch converts every fp32 master weight of a model into its bf16 compute copies: W (bf16 [R,C]) and
// W^T (bf16 [C,R]); mode 1 builds the SwiGLU-interleaved [8 rows w1 | 8 rows w2] matrix from (w1, w2); mode 2
// interleaves the two fp32 bias vectors.  Descriptors live in device memory (8 x int64 each).
struct PrepDesc {
  const float* src;
  const float* src2;
  void* dst;
  void* dstT;
  long R, C;
  long mode;
  long tile_start;
};

__global__ __launch_bounds__(256) void prep_weights_kernel(const PrepDesc* __restrict__ descs, int n, int tile_base) {
  __shared__ float tile[64][65];
  // tile_base: a launch over a RUN of the descriptor table (vtp_prep_weights_range: the layers of one gradient bucket) passes the
  // run's first descriptor and the absolute index of its first tile
  const long tile_id = (long)blockIdx.x + tile_base;
  int lo = 0, hi = n - 1;
  while (lo < hi) {  // last descriptor with tile_start <= tile_id
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].tile_start <= tile_id) lo = mid; else hi = mid - 1;
  }
  const PrepDesc d = descs[lo];
  const int t = (int)(tile_id - d.tile_start);
  const int R = (int)d.R, C = (int)d.C;
  if (d.mode == 2) {  // bias interleave, 1-D: 256 elements per tile
    const int g = t * 256 + threadIdx.x;
    if (g < R) {
      const int j = (g >> 4) * 8 + (g & 7);
      ((float*)d.dst)[g] = (g & 8) ? d.src2[j] : d.src[j];
    }
    return;
  }
  const int tiles_c = (C + 63) / 64;
  const int r0 = (t / tiles_c) * 64, c0 = (t % tiles_c) * 64;
  bf16* dst = (bf16*)d.dst;
  bf16* dstT = (bf16*)d.dstT;
  const bool aligned = ((((uintptr_t)d.src | (uintptr_t)d.src2) & 15) | (((uintptr_t)dst | (uintptr_t)dstT) & 7)) == 0;
  if ((C & 3) == 0 && (R & 3) == 0 && aligned) {
    // 16-B reads of the fp32 master (16 lanes x float4 = one 256-B row segment), 8-B writes of both bf16 copies: the scalar version
    // below moved 2 bytes per lane and store instruction (0.9 ms per step for the ~270 M weights of the VTP-B step)
    const int q = threadIdx.x & 15, rr = threadIdx.x >> 4;  // column quad, row of a 16-row pass
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int r = pass * 16 + rr, g = r0 + r, c = c0 + 4 * q;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (g < R && c < C) {
        const float* src = d.src;
        int j = g;
        if (d.mode == 1) {
          j = (g >> 4) * 8 + (g & 7);
          src = (g & 8) ? d.src2 : d.src;
        }
        v = *(const f32x4*)(src + (size_t)j * C + c);
        if (dst) *(bf16x4*)(dst + (size_t)g * C + c) = __builtin_convertvector(v, bf16x4);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) tile[r][4 * q + e] = v[e];
    }
    if (!dstT) return;
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int c = pass * 16 + rr, r = 4 * q;  // output row c0 + c of the transpose, its columns r0 + r .. + 3
      if (c0 + c < C && r0 + r < R) {
        const f32x4 v = {tile[r][c], tile[r + 1][c], tile[r + 2][c], tile[r + 3][c]};
        *(bf16x4*)(dstT + (size_t)(c0 + c) * R + r0 + r) = __builtin_convertvector(v, bf16x4);
      }
    }
    return;
  }
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    float v = 0.f;
    const int g = r0 + r;
    if (g < R && c0 + tx < C) {
      if (d.mode == 1) {
        const int j = (g >> 4) * 8 + (g & 7);
        v = ((g & 8) ? d.src2 : d.src)[(size_t)j * C + c0 + tx];
      } else {
        v = d.src[(size_t)g * C + c0 + tx];
      }
      if (dst) dst[(size_t)g * C + c0 + tx] = f2bf(v);
    }
    tile[r][tx] = v;
  }
  if (!dstT) return;
  __syncthreads();
  for (int c = ty; c < 64; c += 4)
    if (c0 + c < C && r0 + tx < R) dstT[(size_t)(c0 + c) * R + r0 + tx] = f2bf(tile[tx][c]);
}

// ---------------------------------------------------------------- SwiGLU / GELU backward
// dx12 = d(hidden)/d(x12) * dh in the interleaved [M, 2H] layout: plain grid-stride elementwise kernel (the default path)
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16* __restrict__ dh, const bf16* __restrict__ x12,
                                                         bf16* __restrict__ dx12, int M, int H) {
  const int g8 = H / 8;
  const long total = (long)M * g8;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
    const int g = (int)(idx % g8);
    const long m = idx / g8;
    bf16x8 d = *(const bf16x8*)(dh + m * H + 8 * g);
    bf16x8 x1 = *(const bf16x8*)(x12 + m * 2L * H + 16 * g);
    bf16x8 x2 = *(const bf16x8*)(x12 + m * 2L * H + 16 * g + 8);
    bf16x8 o1, o2;
    swiglu_bwd8(x1, x2, d, o1, o2);
    *(bf16x8*)(dx12 + m * 2L * H + 16 * g) = o1;
    *(bf16x8*)(dx12 + m * 2L * H + 16 * g + 8) = o2;
  }
}

// same, with the w1 / w2 bias gradients fused (used when db12 != null).  A wave covers 64 consecutive 8-column groups of one row
// (2 KiB contiguous of x12); the 4 waves of a workgroup take different rows (2 in flight each) and walk the row dimension, so
// the bias gradients of w1 / w2 -- column sums of the bf16 dx12 -- accumulate in registers, are reduced across the 4 waves
// through LDS and leave as one atomic per column per workgroup (db12 f32 [2H] = [b1 | b2], may be null).
__global__ __launch_bounds__(256) void swiglu_bwd_rows_kernel(const bf16* __restrict__ dh, const bf16* __restrict__ x12,
                                                              bf16* __restrict__ dx12, float* __restrict__ db12, int M, int H) {
  __shared__ float red[4][64 * 16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int g = blockIdx.y * 64 + lane;
  const bool live = g < H / 8;
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
  const int step = gridDim.x * 4;
  if (live) {
    for (int m0 = blockIdx.x * 4 + wv; m0 < M; m0 += 2 * step) {
      bf16x8 d[2], x1[2], x2[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const long m = min(m0 + r * step, M - 1);
        d[r] = *(const bf16x8*)(dh + m * H + 8 * g);
        x1[r] = *(const bf16x8*)(x12 + m * 2L * H + 16 * g);
        x2[r] = *(const bf16x8*)(x12 + m * 2L * H + 16 * g + 8);
      }
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const long m = m0 + r * step;
        if (m >= M) break;
        bf16x8 o1, o2;
        swiglu_bwd8(x1[r], x2[r], d[r], o1, o2);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s1[e] += bf2f(o1[e]);
          s2[e] += bf2f(o2[e]);
        }
        *(bf16x8*)(dx12 + m * 2L * H + 16 * g) = o1;
        *(bf16x8*)(dx12 + m * 2L * H + 16 * g + 8) = o2;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[wv][lane * 16 + e] = s1[e];
    red[wv][lane * 16 + 8 + e] = s2[e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 16; i += 256) {
    const int l = i >> 4, e = i & 15, gg = blockIdx.y * 64 + l;
    if (gg < H / 8) {
      const float t = red[0][i] + red[1][i] + red[2][i] + red[3][i];
      unsafeAtomicAdd(db12 + (e < 8 ? 8 * gg + e : H + 8 * gg + e - 8), t);
    }
  }
}

template <bool QUICK>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ pre,
                                                       bf16* __restrict__ dx, long n) {
  const long n8 = n / 8;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256L) {
    bf16x8 g = *(const bf16x8*)(dy + 8 * i), p = *(const bf16x8*)(pre + 8 * i), o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(g[e]) * (QUICK ? quick_gelu_grad(bf2f(p[e])) : gelu_erf_grad(bf2f(p[e]))));
    *(bf16x8*)(dx + 8 * i) = o;
  }
}

// ---------------------------------------------------------------- optimizer / EMA (flat buffers)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    bf16* __restrict__ pb, long n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2_sqrt, float gs,
                                                    const float* __restrict__ hyper, const uint8_t* __restrict__ nodecay4 = nullptr) {
  if (hyper) {  // device-resident hyper-parameters: a captured hipGraph replays with fresh values every step
    lr = hyper[0]; b1 = hyper[1]; b2 = hyper[2]; eps = hyper[3]; wd = hyper[4]; bc1 = hyper[5]; bc2_sqrt = hyper[6]; gs = hyper[7];
  }
  const long n4 = n / 4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256L) {
    f32x4 pv = *(const f32x4*)(p + 4 * i), gv = *(const f32x4*)(g + 4 * i);
    f32x4 mv = *(const f32x4*)(m + 4 * i), vv = *(const f32x4*)(v + 4 * i);
    const float keep = (nodecay4 && nodecay4[i]) ? 1.f : 1.f - lr * wd;  // parameters are padded to 4 elements: one flag per float4
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gg = gv[e] * gs;
      pv[e] *= keep;
      mv[e] = b1 * mv[e] + (1.f - b1) * gg;
      vv[e] = b2 * vv[e] + (1.f - b2) * gg * gg;
      const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
      pv[e] -= (lr / bc1) * (mv[e] / denom);
    }
    *(f32x4*)(p + 4 * i) = pv;
    *(f32x4*)(m + 4 * i) = mv;
    *(f32x4*)(v + 4 * i) = vv;
    if (pb) *(bf16x4*)(pb + 4 * i) = __builtin_convertvector(pv, bf16x4);
  }
}

// dst[i] (+)= sum_s slabs[s*stride + i]   (split-K wgrad partials -> flat gradient buffer)
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs, long stride, int S,
                                                           float* __restrict__ dst, long n, int accumulate) {
  const long n4 = n / 4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256L) {
    f32x4 a = accumulate ? *(const f32x4*)(dst + 4 * i) : (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s) a += *(const f32x4*)(slabs + (long)s * stride + 4 * i);
    *(f32x4*)(dst + 4 * i) = a;
  }
}

__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ t, const float* __restrict__ s, long n, float mom,
                                                  const float* __restrict__ mom_ptr) {
  if (mom_ptr) mom = mom_ptr[0];  // device-resident momentum: graph replays follow the teacher-momentum schedule
  const long n4 = n / 4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256L) {
    f32x4 tv = *(const f32x4*)(t + 4 * i), sv = *(const f32x4*)(s + 4 * i);
    *(f32x4*)(t + 4 * i) = tv * mom + sv * (1.f - mom);
  }
}

// AdamW of one gradient bucket with the EMA-teacher update of the same elements fused in (vtp.py:388-401: t = mom t + (1 - mom) s on
// the freshly updated student), for the optimizer lane of the training step (vtp_amd/train.py): launched on a side stream as soon as a
// bucket's gradients are final, it runs beside the rest of the backward -- so a block is SHORT-LIVED (4096 elements, no grid-stride
// loop: it takes a free CU slot between the persistent GEMM workgroups and gives it back after one round of loads and stores) and
// requests all of its 16-20 float4 loads before the first use.  hyper[9] = teacher momentum.  Same arithmetic, element by element,
// as adamw_kernel followed by ema_kernel.
__global__ __launch_bounds__(256) void adamw_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, float* __restrict__ t, long n4,
                                                        const float* __restrict__ hyper, const uint8_t* __restrict__ nodecay4) {
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], bc1 = hyper[5], bc2_sqrt = hyper[6], gs = hyper[7];
  const float mom = hyper[9];
  const long base = blockIdx.x * 1024L + threadIdx.x;
  f32x4 pv[4], gv[4], mv[4], vv[4], tv[4];
  bool nd[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long i = base + u * 256;
    if (i < n4) {
      pv[u] = *(const f32x4*)(p + 4 * i);
      gv[u] = *(const f32x4*)(g + 4 * i);
      mv[u] = *(const f32x4*)(m + 4 * i);
      vv[u] = *(const f32x4*)(v + 4 * i);
      if (t) tv[u] = *(const f32x4*)(t + 4 * i);
      nd[u] = nodecay4 && nodecay4[i];
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long i = base + u * 256;
    if (i >= n4) continue;
    const float keep = nd[u] ? 1.f : 1.f - lr * wd;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gg = gv[u][e] * gs;
      pv[u][e] *= keep;
      mv[u][e] = b1 * mv[u][e] + (1.f - b1) * gg;
      vv[u][e] = b2 * vv[u][e] + (1.f - b2) * gg * gg;
      const float denom = sqrtf(vv[u][e]) / bc2_sqrt + eps;
      pv[u][e] -= (lr / bc1) * (mv[u][e] / denom);
    }
    *(f32x4*)(p + 4 * i) = pv[u];
    *(f32x4*)(m + 4 * i) = mv[u];
    *(f32x4*)(v + 4 * i) = vv[u];
    if (t) *(f32x4*)(t + 4 * i) = tv[u] * mom + pv[u] * (1.f - mom);
  }
}

// out[d] += sum_b in[b*stride + d]   (cls-token gradient: rows b*N of the [B,N,D] stream)
// 64 columns per workgroup, the batch rows split over its four waves (B sequentially dependent loads per thread cost 45 us at B = 256;
// the four partial sums are added in wave order: deterministic)
__global__ __launch_bounds__(256) void strided_rowsum_kernel(const float* __restrict__ in, long stride, float* __restrict__ out, int B, int D) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (d < D)
    for (int b = wave; b < B; b += 4) s += in[(long)b * stride + d];
  part[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && d < D) out[d] += (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

// backward of the token assembly (vision_transformer.py:189-219): rows whose patch embedding was replaced by mask_token send their
// gradient to mask_token and nothing to the patch embedding; with d_cls, row 0 of every image sends its gradient to cls_token (and is
// zeroed in the bf16 copy as well: what remains there is exactly the gradient of the patch-embed OUTPUT in the token-row layout).
// One workgroup per `rpb` <= 256 consecutive token rows, one wave per quarter of them: lane l reads the kind of the wave's l-th row, two
// ballots give the masked / cls rows, and only those are read -- 64 lanes x 16 column chunks in flight per row; the four waves' sums meet
// in LDS, one atomic per column, target and workgroup.  (Until round 6: one thread per column scanning 16 rows, 1 028 workgroups x 768
// atomics on the same 768 addresses -- 95 us for the 64 x 257 global crops of the benchmark step -- and a separate row-0 sum.)
__global__ __launch_bounds__(256) void token_rows_bwd_kernel(const float* __restrict__ dx, bf16* __restrict__ dxb,
                                                             const unsigned char* __restrict__ masks, float* __restrict__ d_mask,
                                                             float* __restrict__ d_cls, int B, int N, int D, int rpb) {
  extern __shared__ float part[];  // [2][4][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long rows = (long)B * N;
  const int rpw = (rpb + 3) >> 2;  // <= 64
  const long w0 = (long)blockIdx.x * rpb + (long)wave * rpw;
  long w1 = w0 + rpw;
  const long b1 = (long)(blockIdx.x + 1) * rpb;
  if (w1 > b1) w1 = b1;
  if (w1 > rows) w1 = rows;
  bool f_mask = false, f_cls = false;
  {
    const long r = w0 + lane;
    if (r < w1) {
      const int n = (int)(r % N);
      f_cls = d_cls != nullptr && n == 0;
      f_mask = masks != nullptr && n != 0 && masks[(r / N) * (N - 1) + (n - 1)] != 0;
    }
  }
  const unsigned long long hit[2] = {__ballot(f_mask), __ballot(f_cls)};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    float* out = part + (size_t)k * 4 * D + (size_t)wave * D;
    for (int d0 = 0; d0 < D; d0 += 1024) {
      float acc[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) acc[c] = 0.f;
      unsigned long long m = hit[k];
      while (m) {
        const int i = __builtin_ctzll(m);
        m &= m - 1;
        const long o = (w0 + i) * D + d0 + lane;
        float v[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = (d0 + c * 64 + lane < D) ? dx[o + c * 64] : 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          acc[c] += v[c];
          if (d0 + c * 64 + lane < D) dxb[o + c * 64] = (bf16)0.f;
        }
      }
#pragma unroll
      for (int c = 0; c < 16; ++c)
        if (d0 + c * 64 + lane < D) out[d0 + c * 64 + lane] = acc[c];
    }
  }
  __syncthreads();
  const bool any_mask = __syncthreads_or(hit[0] != 0ull), any_cls = __syncthreads_or(hit[1] != 0ull);
  if (any_mask)
    for (int d = threadIdx.x; d < D; d += 256) unsafeAtomicAdd(d_mask + d, (part[d] + part[D + d]) + (part[2 * D + d] + part[3 * D + d]));
  if (any_cls) {
    const float* pc = part + (size_t)4 * D;
    for (int d = threadIdx.x; d < D; d += 256) unsafeAtomicAdd(d_cls + d, (pc[d] + pc[D + d]) + (pc[2 * D + d] + pc[3 * D + d]));
  }
}

static inline int grid_for(long items, int cap = 4096) {
  long b = (items + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace vtp
using namespace vtp;

extern "C" int vtp_rope_qk(void* qkv, const void* sin, const void* cos, int B, int N, int heads, int prefix,
                           int inverse, void* stream) {
  VTP_REQUIRE(qkv && sin && cos, "vtp_rope_qk: null pointer");
  VTP_REQUIRE(B > 0 && N > 0 && heads > 0 && prefix >= 0 && prefix <= N, "vtp_rope_qk: bad shape (prefix must be in [0,N])");
  const long items = (long)B * N * 2 * heads * 4;
  if (inverse)
    hipLaunchKernelGGL(rope_qk_kernel<true>, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, (bf16*)qkv,
                       (const bf16*)sin, (const bf16*)cos, B, N, heads, prefix);
  else
    hipLaunchKernelGGL(rope_qk_kernel<false>, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, (bf16*)qkv,
                       (const bf16*)sin, (const bf16*)cos, B, N, heads, prefix);
  return check_launch("rope_qk");
}

extern "C" int vtp_im2col16(const float* img, void* patches, int B, int H, int W, void* stream) {
  VTP_REQUIRE(img && patches, "vtp_im2col16: null pointer");
  VTP_REQUIRE(B > 0 && H > 0 && W > 0 && H % 16 == 0 && W % 16 == 0, "vtp_im2col16: H and W must be positive multiples of 16");
  const long items = (long)B * (H / 16) * (W / 16) * 48;
  hipLaunchKernelGGL(im2col16_kernel, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, img, (bf16*)patches, B, H, W, 0);
  return check_launch("im2col16");
}

extern "C" int vtp_im2col16_rows(const float* img, void* rows, int B, int H, int W, int prefix, void* stream) {
  VTP_REQUIRE(img && rows, "vtp_im2col16_rows: null pointer");
  VTP_REQUIRE(B > 0 && H > 0 && W > 0 && H % 16 == 0 && W % 16 == 0 && prefix >= 0, "vtp_im2col16_rows: H and W must be positive multiples of 16, prefix >= 0");
  const long items = (long)B * (H / 16) * (W / 16) * 48;
  hipLaunchKernelGGL(im2col16_kernel, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, img, (bf16*)rows, B, H, W, prefix);
  return check_launch("im2col16_rows");
}

extern "C" int vtp_col2im16(const float* dpatches, float* dimg, int B, int H, int W, void* stream) {
  VTP_REQUIRE(dpatches && dimg, "vtp_col2im16: null pointer");
  VTP_REQUIRE(B > 0 && H > 0 && W > 0 && H % 16 == 0 && W % 16 == 0, "vtp_col2im16: H and W must be positive multiples of 16");
  const long items = (long)B * (H / 16) * (W / 16) * 48;
  hipLaunchKernelGGL(col2im16_kernel, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, dpatches, dimg, B, H, W);
  return check_launch("col2im16");
}

extern "C" int vtp_assemble_tokens(float* x, const float* cls, const float* mask_token, const unsigned char* masks,
                                   int B, int N, int D, void* stream) {
  VTP_REQUIRE(x && cls, "vtp_assemble_tokens: null pointer");
  VTP_REQUIRE(!masks || mask_token, "vtp_assemble_tokens: masks given without mask_token");
  VTP_REQUIRE(B > 0 && N > 1 && D % 4 == 0, "vtp_assemble_tokens: bad shape");
  const long items = (long)B * N * (D / 4);
  hipLaunchKernelGGL(assemble_tokens_kernel, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, x, cls, mask_token, masks, B, N, D);
  return check_launch("assemble_tokens");
}

extern "C" int vtp_transpose_bf16(const void* in, int ld_in, void* out, int ld_out, float* colsum, int colsum_swiglu_h,
                                  int in_grp, int in_pre, int R, int C, void* stream) {
  VTP_REQUIRE(in && out, "vtp_transpose_bf16: null pointer");
  VTP_REQUIRE(R > 0 && C > 0 && C % 8 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0 && ld_out >= R && ld_in >= C,
              "vtp_transpose_bf16: need C, ld_in, ld_out multiples of 8, ld_out >= R");
  dim3 grid(cdiv(C, 64), cdiv(ld_out, 64));
  hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)in, ld_in, (bf16*)out, ld_out, colsum, R, C, colsum_swiglu_h, in_grp, in_pre);
  return check_launch("transpose_bf16");
}

extern "C" int vtp_colsum_bf16(const void* in, int ld, float* out, int colsum_swiglu_h, int in_grp, int in_pre, int R, int C,
                               void* stream) {
  VTP_REQUIRE(in && out && R > 0 && C > 0 && C % 8 == 0 && ld % 8 == 0, "vtp_colsum_bf16: bad argument (C, ld %% 8 == 0)");
  hipLaunchKernelGGL(colsum_bf16_kernel, dim3(cdiv(C, 64), cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)in, ld,
                     out, R, C, colsum_swiglu_h, in_grp, in_pre, (const int*)nullptr);
  return check_launch("colsum_bf16");
}

extern "C" int vtp_colsum_bf16_rows(const void* in, int ld, float* out, const int* n_rows_dev, int R_max, int C, void* stream) {
  VTP_REQUIRE(in && out && n_rows_dev && R_max > 0 && C > 0 && C % 8 == 0 && ld % 8 == 0, "vtp_colsum_bf16_rows: bad argument");
  hipLaunchKernelGGL(colsum_bf16_kernel, dim3(cdiv(C, 64), cdiv(R_max, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)in,
                     ld, out, R_max, C, 0, 0, 0, n_rows_dev);
  return check_launch("colsum_bf16_rows");
}

static int launch_token_rows_bwd(const float* dx, void* dx_bf16, const unsigned char* masks, float* d_mask_token, float* d_cls_token,
                                 int B, int N, int D, void* stream, const char* what) {
  const long rows = (long)B * N;
  long rpb = cdiv(rows, 512L);  // <= 512 workgroups x D atomics per target; a wave handles at most 64 rows
  if (rpb < 16) rpb = 16;
  if (rpb > 256) rpb = 256;
  hipLaunchKernelGGL(token_rows_bwd_kernel, dim3(cdiv(rows, rpb)), dim3(256), (size_t)8 * D * sizeof(float), (hipStream_t)stream, dx,
                     (bf16*)dx_bf16, masks, d_mask_token, d_cls_token, B, N, D, (int)rpb);
  return check_launch(what);
}

extern "C" int vtp_mask_rows_bwd(const float* dx, void* dx_bf16, const unsigned char* masks, float* d_mask_token, int B, int N,
                                 int D, void* stream) {
  VTP_REQUIRE(dx && dx_bf16 && masks && d_mask_token && B > 0 && N > 1 && D > 0 && D <= 2048, "vtp_mask_rows_bwd: bad argument (D <= 2048)");
  return launch_token_rows_bwd(dx, dx_bf16, masks, d_mask_token, nullptr, B, N, D, stream, "mask_rows_bwd");
}

extern "C" int vtp_token_rows_bwd(const float* dx, void* dx_bf16, const unsigned char* masks, float* d_mask_token, float* d_cls_token,
                                  int B, int N, int D, void* stream) {
  VTP_REQUIRE(dx && dx_bf16 && d_cls_token && (masks == nullptr) == (d_mask_token == nullptr) && B > 0 && N > 1 && D > 0 && D <= 2048,
              "vtp_token_rows_bwd: bad argument (masks and d_mask_token together or neither; D <= 2048)");
  return launch_token_rows_bwd(dx, dx_bf16, masks, d_mask_token, d_cls_token, B, N, D, stream, "token_rows_bwd");
}

extern "C" int vtp_strided_rowsum(const float* in, long stride, float* out, int B, int D, void* stream) {
  VTP_REQUIRE(in && out && B > 0 && D > 0, "vtp_strided_rowsum: bad argument");
  hipLaunchKernelGGL(strided_rowsum_kernel, dim3(cdiv(D, 64)), dim3(256), 0, (hipStream_t)stream, in, stride, out, B, D);
  return check_launch("strided_rowsum");
}

extern "C" int vtp_cast_f32_bf16(const float* in, void* out, long n, void* stream) {
  VTP_REQUIRE(in && out && n > 0, "vtp_cast_f32_bf16: bad argument");
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, in, (bf16*)out, n);
  return check_launch("cast_f32_bf16");
}

extern "C" int vtp_cast_transpose_f32_bf16(const float* in, void* out, int R, int C, void* stream) {
  VTP_REQUIRE(in && out && R > 0 && C > 0, "vtp_cast_transpose_f32_bf16: bad argument");
  hipLaunchKernelGGL(cast_transpose_kernel, dim3(cdiv(C, 64), cdiv(R, 64)), dim3(256), 0, (hipStream_t)stream, in, (bf16*)out, R, C);
  return check_launch("cast_transpose");
}


extern "C" int vtp_prep_weights(const void* descs, int n, int total_tiles, void* stream) {
  VTP_REQUIRE(descs && n > 0 && total_tiles > 0, "vtp_prep_weights: bad argument");
  hipLaunchKernelGGL(prep_weights_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, (const PrepDesc*)descs, n, 0);
  return check_launch("prep_weights");
}

extern "C" int vtp_prep_weights_range(const void* descs, int n, int tile_base, int n_tiles, void* stream) {
  VTP_REQUIRE(descs && n > 0 && tile_base >= 0 && n_tiles > 0, "vtp_prep_weights_range: bad argument");
  hipLaunchKernelGGL(prep_weights_kernel, dim3(n_tiles), dim3(256), 0, (hipStream_t)stream, (const PrepDesc*)descs, n, tile_base);
  return check_launch("prep_weights_range");
}

extern "C" int vtp_swiglu_bwd(const void* dh, const void* x12, void* dx12, float* db12, int M, int H, void* stream) {
  VTP_REQUIRE(dh && x12 && dx12 && M > 0 && H > 0 && H % 8 == 0, "vtp_swiglu_bwd: bad argument (H %% 8 == 0)");
  const int gy = cdiv(H / 8, 64);
  int gx = 2048 / gy;  // ~8 workgroups per CU in total
  if (gx > (M + 7) / 8) gx = (M + 7) / 8;
  if (gx < 1) gx = 1;
  if (db12)
    hipLaunchKernelGGL(swiglu_bwd_rows_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const bf16*)dh,
                       (const bf16*)x12, (bf16*)dx12, db12, M, H);
  else
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for((long)M * (H / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)dh, (const bf16*)x12, (bf16*)dx12, M, H);
  return check_launch("swiglu_bwd");
}

extern "C" int vtp_gelu_bwd(const void* dy, const void* pre, void* dx, long n, void* stream) {
  VTP_REQUIRE(dy && pre && dx && n > 0 && n % 8 == 0, "vtp_gelu_bwd: bad argument (n %% 8 == 0)");
  hipLaunchKernelGGL(gelu_bwd_kernel<false>, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16*)dy, (const bf16*)pre, (bf16*)dx, n);
  return check_launch("gelu_bwd");
}

extern "C" int vtp_quick_gelu_bwd(const void* dy, const void* pre, void* dx, long n, void* stream) {
  VTP_REQUIRE(dy && pre && dx && n > 0 && n % 8 == 0, "vtp_quick_gelu_bwd: bad argument (n %% 8 == 0)");
  hipLaunchKernelGGL(gelu_bwd_kernel<true>, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16*)dy, (const bf16*)pre, (bf16*)dx, n);
  return check_launch("quick_gelu_bwd");
}

extern "C" int vtp_pixel_shuffle16(const void* t, float* img, int B, int h, int w, void* stream) {
  VTP_REQUIRE(t && img && B > 0 && h > 0 && w > 0, "vtp_pixel_shuffle16: bad argument");
  hipLaunchKernelGGL(shuffle16_kernel<0>, dim3(grid_for((long)B * h * w * 48)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)t, img, nullptr, nullptr, nullptr, B, h, w, 0.f);
  return check_launch("pixel_shuffle16");
}

extern "C" int vtp_pixel_unshuffle16(const float* d_img, void* dt, int B, int h, int w, void* stream) {
  VTP_REQUIRE(d_img && dt && B > 0 && h > 0 && w > 0, "vtp_pixel_unshuffle16: bad argument");
  hipLaunchKernelGGL(shuffle16_kernel<2>, dim3(grid_for((long)B * h * w * 48)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)nullptr, const_cast<float*>(d_img), nullptr, (bf16*)dt, nullptr, B, h, w, 0.f);
  return check_launch("pixel_unshuffle16");
}

extern "C" int vtp_l1_loss_fwd_bwd(const void* t, const float* target, void* dt, float* loss_sum, int B, int h, int w,
                                   float gscale, void* stream) {
  VTP_REQUIRE(t && target && dt && loss_sum && B > 0 && h > 0 && w > 0, "vtp_l1_loss_fwd_bwd: bad argument");
  hipLaunchKernelGGL(shuffle16_kernel<1>, dim3(grid_for((long)B * h * w * 48, 1024)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)t, nullptr, target, (bf16*)dt, loss_sum, B, h, w, gscale);
  return check_launch("l1_loss_fwd_bwd");
}

extern "C" int vtp_adamw(float* p, const float* g, float* m, float* v, void* p_bf16, long n, float lr, float beta1,
                         float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream) {
  VTP_REQUIRE(p && g && m && v && n > 0 && n % 4 == 0 && step >= 1, "vtp_adamw: bad argument (n %% 4 == 0, step >= 1)");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16*)p_bf16, n, lr,
                     beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale, (const float*)nullptr);
  return check_launch("adamw");
}

extern "C" int vtp_adamw_dev(float* p, const float* g, float* m, float* v, void* p_bf16, long n, const float* hyper,
                             void* stream) {
  VTP_REQUIRE(p && g && m && v && hyper && n > 0 && n % 4 == 0, "vtp_adamw_dev: bad argument (n %% 4 == 0)");
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16*)p_bf16, n, 0.f,
                     0.f, 0.f, 0.f, 0.f, 1.f, 1.f, 1.f, hyper);
  return check_launch("adamw_dev");
}

extern "C" int vtp_adamw_dev_masked(float* p, const float* g, float* m, float* v, void* p_bf16, const void* nodecay4, long n,
                                    const float* hyper, void* stream) {
  VTP_REQUIRE(p && g && m && v && hyper && nodecay4 && n > 0 && n % 4 == 0, "vtp_adamw_dev_masked: bad argument (n %% 4 == 0)");
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16*)p_bf16, n, 0.f,
                     0.f, 0.f, 0.f, 0.f, 1.f, 1.f, 1.f, hyper, (const uint8_t*)nodecay4);
  return check_launch("adamw_dev_masked");
}

extern "C" int vtp_reduce_slabs(const float* slabs, long stride, int S, float* dst, long n, int accumulate, void* stream) {
  VTP_REQUIRE(slabs && dst && S >= 1 && n > 0 && n % 4 == 0 && stride % 4 == 0, "vtp_reduce_slabs: bad argument (n, stride %% 4 == 0)");
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3(grid_for(n / 4, 2048)), dim3(256), 0, (hipStream_t)stream, slabs, stride, S, dst, n, accumulate);
  return check_launch("reduce_slabs");
}

extern "C" int vtp_ema(float* t, const float* s, long n, float momentum, void* stream) {
  VTP_REQUIRE(t && s && n > 0 && n % 4 == 0, "vtp_ema: bad argument (n %% 4 == 0)");
  hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, t, s, n, momentum, (const float*)nullptr);
  return check_launch("ema");
}

extern "C" int vtp_ema_dev(float* t, const float* s, long n, const float* momentum, void* stream) {
  VTP_REQUIRE(t && s && momentum && n > 0 && n % 4 == 0, "vtp_ema_dev: bad argument (n %% 4 == 0)");
  hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, t, s, n, 0.f, momentum);
  return check_launch("ema_dev");
}

extern "C" int vtp_adamw_ema_dev(float* p, const float* g, float* m, float* v, float* teacher, const void* nodecay4, long n,
                                 const float* hyper, void* stream) {
  VTP_REQUIRE(p && g && m && v && hyper && n > 0 && n % 4 == 0, "vtp_adamw_ema_dev: bad argument (n %% 4 == 0)");
  VTP_REQUIRE(n / 4096 < 0x7fffffffL, "vtp_adamw_ema_dev: range too long for one launch");
  hipLaunchKernelGGL(adamw_ema_kernel, dim3((unsigned)((n / 4 + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, teacher,
                     n / 4, hyper, (const uint8_t*)nodecay4);
  return check_launch("adamw_ema_dev");
}
