// LPIPS perceptual distance (reference: vtp/utils/lpips.py:61-175) around the implicit-GEMM 3x3 convolutions of gemm.hip.
//
// Activation layout: zero-bordered NHWC image stacks, bf16 [NB, H+2, W+2, C] flattened to pixel rows, with W+3 guard rows
// of zeros in front of and behind the stack (the 3x3 taps of vtp_conv3x3 are plain row offsets of +-(W+3) at most).
// The kernels here are the HBM-bound pieces: first-layer unfold (ScalingLayer + PixelShuffle fused), 2x2 max-pool forward /
// backward (+ ReLU mask + tap gradient), the per-pixel normalise / difference / 1x1 "lin" head with its gradient, and the
// fold of the first layer's input gradient back into the token-major reconstruction gradient.
#include "common.h"
#include "vtp_hip.h"

namespace vtp {

// ---------------------------------------------------------------------------------------------------- first-layer unfold
// out row (b, y, x) of the bordered stack = 27 values img_scaled[c, y-1+ky-1, x-1+kx-1] in (ky, kx, c) order + 5 zeros.
// source: tok != null -> reconstruction tokens bf16 [n*hw, 768] (pre-PixelShuffle, pixel_decoder.py:158-161);
//         else img f32 NCHW [n, 3, H, W].  ScalingLayer (lpips.py:103-114): (v - shift_c) / scale_c.
__global__ __launch_bounds__(256) void lpips_unfold3_kernel(const bf16* __restrict__ tok, const float* __restrict__ img,
                                                            bf16* __restrict__ out, int n, int H, int W, float sh0, float sh1,
                                                            float sh2, float is0, float is1, float is2) {
  const int W2 = W + 2, P = (H + 2) * W2;
  const long row = blockIdx.x * 256L + threadIdx.x;
  if (row >= (long)n * P) return;
  const int b = (int)(row / P), r = (int)(row % P), y = r / W2, x = r % W2;
  bf16 v[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = f2bf(0.f);
  if (y >= 1 && y <= H && x >= 1 && x <= W) {
    const float sh[3] = {sh0, sh1, sh2}, is[3] = {is0, is1, is2};
    const int wt = W >> 4;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int Y = y - 1 + ky - 1, X = x - 1 + kx - 1;
        if (Y < 0 || Y >= H || X < 0 || X >= W) continue;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float p;
          if (tok) p = bf2f(tok[((long)b * (H >> 4) * wt + (Y >> 4) * wt + (X >> 4)) * 768 + c * 256 + (Y & 15) * 16 + (X & 15)]);
          else p = img[(((long)b * 3 + c) * H + Y) * W + X];
          v[(ky * 3 + kx) * 3 + c] = f2bf((p - sh[c]) * is[c]);
        }
      }
  }
  bf16x8* o = (bf16x8*)(out + row * 32);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    bf16x8 t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = v[i * 8 + e];
    o[i] = t;
  }
}

// fold of the first layer's input gradient: dA0 bf16 [n*P, 32] (border rows zero) -> dt bf16 [n*hw, 768] += d_img / scale_c
__global__ __launch_bounds__(256) void lpips_fold3_bwd_kernel(const bf16* __restrict__ dA, bf16* __restrict__ dt, int n, int H,
                                                              int W, float is0, float is1, float is2) {
  const int W2 = W + 2, P = (H + 2) * W2;
  const long idx = blockIdx.x * 256L + threadIdx.x;
  if (idx >= (long)n * H * W) return;
  const int b = (int)(idx / (H * W)), rem = (int)(idx % (H * W)), Y = rem / W, X = rem % W;
  float g[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      // output pixel (y, x) (bordered coords) used image pixel (Y, X) through tap (ky, kx):  Y = y-1+ky-1
      const int y = Y + 2 - ky, x = X + 2 - kx;
      if (y < 1 || y > H || x < 1 || x > W) continue;
      const bf16* p = dA + ((long)b * P + y * W2 + x) * 32 + (ky * 3 + kx) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) g[c] += bf2f(p[c]);
    }
  const float is[3] = {is0, is1, is2};
  const int wt = W >> 4;
  bf16* o = dt + ((long)b * (H >> 4) * wt + (Y >> 4) * wt + (X >> 4)) * 768 + (Y & 15) * 16 + (X & 15);
#pragma unroll
  for (int c = 0; c < 3; ++c) o[c * 256] = f2bf(bf2f(o[c * 256]) + g[c] * is[c]);
}

// ---------------------------------------------------------------------------------------------------- 2x2 max-pool
// in [NB, H+2, W+2, C] -> out [NB, H/2+2, W/2+2, C] (borders written as zeros); one thread = one output pixel x 8 channels
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const bf16* __restrict__ in, bf16* __restrict__ out, int NB, int H,
                                                           int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1, W2 = W + 2, Wo2 = Wo + 2, cg = C >> 3;
  const long total = (long)NB * (Ho + 2) * Wo2 * cg;
  const long idx = blockIdx.x * 256L + threadIdx.x;
  if (idx >= total) return;
  const int c8 = (int)(idx % cg);
  const long pix = idx / cg;
  const int x = (int)(pix % Wo2), y = (int)((pix / Wo2) % (Ho + 2)), b = (int)(pix / ((long)Wo2 * (Ho + 2)));
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = f2bf(0.f);
  if (y >= 1 && y <= Ho && x >= 1 && x <= Wo) {
    const bf16* p = in + (((long)b * (H + 2) + (2 * y - 1)) * W2 + (2 * x - 1)) * C + c8 * 8;
    const bf16x8 a = *(const bf16x8*)p, bq = *(const bf16x8*)(p + C), c = *(const bf16x8*)(p + (long)W2 * C),
                 d = *(const bf16x8*)(p + (long)W2 * C + C);
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = f2bf(fmaxf(fmaxf(bf2f(a[e]), bf2f(bq[e])), fmaxf(bf2f(c[e]), bf2f(d[e]))));
  }
  *(bf16x8*)(out + pix * C + c8 * 8) = r;
}

// backward through [ReLU -> (tap) -> max-pool]: dY = (Y > 0) * (route(dP) + tap), route = first maximum of the window in
// row-major order (ATen's max_pool2d backward).  One thread = one pooled pixel x 8 channels -> four dY vectors.
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const bf16* __restrict__ Y, const bf16* __restrict__ dP,
                                                           const bf16* __restrict__ tap, bf16* __restrict__ dY, int NB, int H,
                                                           int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1, W2 = W + 2, Wo2 = Wo + 2, cg = C >> 3;
  const long total = (long)NB * Ho * Wo * cg;
  const long idx = blockIdx.x * 256L + threadIdx.x;
  if (idx >= total) return;
  const int c8 = (int)(idx % cg);
  const long pix = idx / cg;
  const int x = (int)(pix % Wo) + 1, y = (int)((pix / Wo) % Ho) + 1, b = (int)(pix / ((long)Wo * Ho));
  const long base = (((long)b * (H + 2) + (2 * y - 1)) * W2 + (2 * x - 1)) * C + c8 * 8;
  const long offs[4] = {0, C, (long)W2 * C, (long)W2 * C + C};
  bf16x8 v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = *(const bf16x8*)(Y + base + offs[i]);
  const bf16x8 g = *(const bf16x8*)(dP + (((long)b * (Ho + 2) + y) * Wo2 + x) * C + c8 * 8);
  bf16x8 o[4];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float a0 = bf2f(v[0][e]), a1 = bf2f(v[1][e]), a2 = bf2f(v[2][e]), a3 = bf2f(v[3][e]);
    const float mx = fmaxf(fmaxf(a0, a1), fmaxf(a2, a3));
    const int am = a0 == mx ? 0 : (a1 == mx ? 1 : (a2 == mx ? 2 : 3));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float t = (i == am) ? bf2f(g[e]) : 0.f;
      o[i][e] = f2bf(t);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    bf16x8 r = o[i];
    if (tap) {
      const bf16x8 t = *(const bf16x8*)(tap + base + offs[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = f2bf(bf2f(r[e]) + bf2f(t[e]));
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = bf2f(v[i][e]) > 0.f ? r[e] : f2bf(0.f);
    *(bf16x8*)(dY + base + offs[i]) = r;
  }
}

// ---------------------------------------------------------------------------------------------------- LPIPS head of one tap
// f0 / f1 bf16 [n*P, C] (same bordered layout; f1 = the target's features): per interior pixel
//   n_i = f_i / (|f_i| + 1e-10)   (normalize_tensor, lpips.py:169-171),   d = sum_c w_c (n0_c - n1_c)^2   (NetLinLayer 1x1)
//   val[b] += d / (H*W)           (spatial_average);   df0 = gscale * d(d)/d(f0), masked by the ReLU in front (f0 > 0)
// LPP = lanes per pixel = C / 8.
template <int LPP>
__global__ __launch_bounds__(256) void lpips_tap_kernel(const bf16* __restrict__ f0, const bf16* __restrict__ f1,
                                                        const float* __restrict__ w, float* __restrict__ val,
                                                        bf16* __restrict__ df0, int n, int H, int W, float gscale) {
  constexpr int C = LPP * 8, PPW = 64 / LPP;  // pixels per wave
  __shared__ float red[4];
  const int W2 = W + 2, P = (H + 2) * W2;
  const int lane = threadIdx.x & 63, sub = lane % LPP;
  const int b = blockIdx.y;                     // one image per grid row: a block's partial sum goes to one val[b]
  const f32x4 w0 = *(const f32x4*)(w + sub * 8), w1 = *(const f32x4*)(w + sub * 8 + 4);
  float ww[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) ww[e] = e < 4 ? w0[e] : w1[e - 4];
  float contrib = 0.f;
  for (int r = (blockIdx.x * 4 + (threadIdx.x >> 6)) * PPW + lane / LPP; r < P; r += gridDim.x * 4 * PPW) {
    const int y = r / W2, x = r - y * W2;
    if (y < 1 || y > H || x < 1 || x > W) continue;  // border rows: df0 stays zero (never written)
    const long pix = (long)b * P + r;
    const bf16x8 a = *(const bf16x8*)(f0 + pix * C + sub * 8), t = *(const bf16x8*)(f1 + pix * C + sub * 8);
    float fa[8], ft[8];
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      fa[e] = bf2f(a[e]);
      ft[e] = bf2f(t[e]);
      s0 += fa[e] * fa[e];
      s1 += ft[e] * ft[e];
    }
#pragma unroll
    for (int o = 1; o < LPP; o <<= 1) {
      s0 += __shfl_xor(s0, o, 64);
      s1 += __shfl_xor(s1, o, 64);
    }
    const float r0 = sqrtf(s0), inv0 = 1.f / (r0 + 1e-10f), inv1 = 1.f / (sqrtf(s1) + 1e-10f);
    float d = 0.f, tsum = 0.f, dn[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float df = fa[e] * inv0 - ft[e] * inv1;
      d += ww[e] * df * df;
      dn[e] = 2.f * gscale * ww[e] * df;
      tsum += dn[e] * fa[e];
    }
    contrib += d;  // every lane's share; summed over the block below
#pragma unroll
    for (int o = 1; o < LPP; o <<= 1) tsum += __shfl_xor(tsum, o, 64);
    if (df0) {
      const float k2 = r0 > 0.f ? tsum * inv0 * inv0 / r0 : 0.f;
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(fa[e] > 0.f ? dn[e] * inv0 - fa[e] * k2 : 0.f);
      *(bf16x8*)(df0 + pix * C + sub * 8) = o;
    }
  }
  // note: the shuffles above sit inside a loop whose trip count is uniform within each LPP-lane group (same pixel), and the
  // `continue` is taken by whole groups, so the participating lanes of every shuffle agree
  const float tot = block_sum<4>(contrib, red);
  if (threadIdx.x == 0 && tot != 0.f) atomicAdd(val + b, tot / (float)(H * W));
}

}  // namespace vtp

using namespace vtp;

extern "C" int vtp_lpips_unfold3(const void* tok, const float* img, void* out, int n, int H, int W, const float* shift,
                                 const float* scale, void* stream) {
  VTP_REQUIRE((tok != nullptr) != (img != nullptr), "vtp_lpips_unfold3: exactly one of tok / img");
  VTP_REQUIRE(out && shift && scale && n > 0 && H % 16 == 0 && W % 16 == 0, "vtp_lpips_unfold3: bad argument");
  const long rows = (long)n * (H + 2) * (W + 2);
  hipLaunchKernelGGL(lpips_unfold3_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)tok, img,
                     (bf16*)out, n, H, W, shift[0], shift[1], shift[2], 1.f / scale[0], 1.f / scale[1], 1.f / scale[2]);
  return check_launch("lpips_unfold3");
}

extern "C" int vtp_lpips_fold3_bwd(const void* dA, void* dt, int n, int H, int W, const float* scale, void* stream) {
  VTP_REQUIRE(dA && dt && scale && n > 0 && H % 16 == 0 && W % 16 == 0, "vtp_lpips_fold3_bwd: bad argument");
  hipLaunchKernelGGL(lpips_fold3_bwd_kernel, dim3(cdiv((long)n * H * W, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)dA, (bf16*)dt, n, H, W, 1.f / scale[0], 1.f / scale[1], 1.f / scale[2]);
  return check_launch("lpips_fold3_bwd");
}

extern "C" int vtp_maxpool2_fwd(const void* in, void* out, int NB, int H, int W, int C, void* stream) {
  VTP_REQUIRE(in && out && NB > 0 && H % 2 == 0 && W % 2 == 0 && C % 8 == 0, "vtp_maxpool2_fwd: bad argument");
  const long total = (long)NB * (H / 2 + 2) * (W / 2 + 2) * (C / 8);
  hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)in, (bf16*)out,
                     NB, H, W, C);
  return check_launch("maxpool2_fwd");
}

extern "C" int vtp_maxpool2_bwd(const void* Y, const void* dP, const void* tap, void* dY, int NB, int H, int W, int C,
                                void* stream) {
  VTP_REQUIRE(Y && dP && dY && NB > 0 && H % 2 == 0 && W % 2 == 0 && C % 8 == 0, "vtp_maxpool2_bwd: bad argument");
  const long total = (long)NB * (H / 2) * (W / 2) * (C / 8);
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)Y,
                     (const bf16*)dP, (const bf16*)tap, (bf16*)dY, NB, H, W, C);
  return check_launch("maxpool2_bwd");
}

extern "C" int vtp_lpips_tap(const void* f0, const void* f1, const float* w, float* val, void* df0, int n, int H, int W, int C,
                             float gscale, void* stream) {
  VTP_REQUIRE(f0 && f1 && w && val && n > 0, "vtp_lpips_tap: bad argument");
  VTP_REQUIRE(C == 64 || C == 128 || C == 256 || C == 512, "vtp_lpips_tap: C must be 64, 128, 256 or 512 (VGG16 taps)");
  const long pixels = (long)(H + 2) * (W + 2);  // per image (grid.y = image)
  hipStream_t s = (hipStream_t)stream;
#define VTP_TAP(L)                                                                                                          \
  hipLaunchKernelGGL(lpips_tap_kernel<L>, dim3(min(cdiv(pixels, 4L * (64 / L)), 96), n), dim3(256), 0, s, (const bf16*)f0,     \
                     (const bf16*)f1, w, val, (bf16*)df0, n, H, W, gscale)
  switch (C) {
    case 64: VTP_TAP(8); break;
    case 128: VTP_TAP(16); break;
    case 256: VTP_TAP(32); break;
    default: VTP_TAP(64); break;
  }
#undef VTP_TAP
  return check_launch("lpips_tap");
}
