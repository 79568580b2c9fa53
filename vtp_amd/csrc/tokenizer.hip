// Image <-> latent boundary of the downstream tokenizer flow (generation/tokenizer/vtp_tokenizer.py:74-111,
// generation/tools/extract_features_vtp.py:88-118): byte-image packing on either side of the encode / decode towers and the
// per-channel latent statistics -- HBM-bound byte / float row kernels, bit-exact against the torch ops the reference chains.
#include "common.h"
#include "vtp_hip.h"

namespace vtp {

// ToTensor + Normalize (+ RandomHorizontalFlip(p in {0, 1})), vtp_tokenizer.py:74-81:
//   out[b,c,y,x] = (float(u8[b,y,xs,c]) / 255 - mean[c]) / std[c],  xs = flip ? W-1-x : x      (same op order as torchvision)
// One thread per 4 output pixels of one channel plane: coalesced 16-byte stores; the 3-byte-strided reads hit in L2.
__global__ __launch_bounds__(256) void u8_to_images_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, long B, int H,
                                                           int W, f32x4 mean, f32x4 stdv, int flip) {
  const long total = B * 3 * H * (W / 4);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const int x4 = (int)(i % (W / 4));
    long r = i / (W / 4);
    const int y = (int)(r % H);
    r /= H;
    const int c = (int)(r % 3);
    const long b = r / 3;
    const uint8_t* row = src + ((b * H + y) * W) * 3 + c;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int x = x4 * 4 + j;
      const int xs = flip ? W - 1 - x : x;
      v[j] = ((float)row[(long)xs * 3] / 255.0f - mean[c]) / stdv[c];
    }
    *(f32x4*)(dst + ((b * 3 + c) * H + y) * (long)W + x4 * 4) = v;
  }
}

// decode_to_images tail (vtp_tokenizer.py:105-111): Normalize(inv_mean, inv_std) -> * 255 -> clamp(0, 255) -> uint8 (truncation)
// -> NHWC.   out[b,y,x,c] = u8(clamp(((img[b,c,y,x] - sub[c]) / div[c]) * 255, 0, 255))
// One thread per 4 pixels: three 16-byte plane reads, one 12-byte packed store.
__global__ __launch_bounds__(256) void images_to_u8_kernel(const float* __restrict__ img, uint8_t* __restrict__ out, long B, int H, int W,
                                                           f32x4 sub, f32x4 dv) {
  const long total = B * H * (W / 4);
  const long plane = (long)H * W;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const long b = i / (plane / 4);
    const long p = (i - b * (plane / 4)) * 4;  // pixel offset inside the plane
    uint32_t w[3] = {0, 0, 0};
    f32x4 v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = *(const f32x4*)(img + (b * 3 + c) * plane + p);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float t = ((v[c][j] - sub[c]) / dv[c]) * 255.0f;
        t = fminf(fmaxf(t, 0.0f), 255.0f);
        const uint32_t byte = (uint32_t)t;  // torch .to(uint8): truncation
        const int k = j * 3 + c;
        w[k >> 2] |= byte << ((k & 3) * 8);
      }
    uint32_t* o = (uint32_t*)(out + (b * plane + p) * 3);  // 12 bytes per 4 pixels, 4-byte aligned (p % 4 == 0)
    o[0] = w[0];
    o[1] = w[1];
    o[2] = w[2];
  }
}

// per-channel sum and sum of squares of latents f32 [B, C, hw] in fp64 (one block per (channel, batch slice))
__global__ __launch_bounds__(256) void latent_stats_kernel(const float* __restrict__ lat, double* __restrict__ sums, long B, int C,
                                                           int hw, int slices) {
  __shared__ double red[2][4];
  const int c = blockIdx.x / slices, s = blockIdx.x % slices;
  double a = 0.0, q = 0.0;
  for (long b = s; b < B; b += slices) {
    const float* p = lat + (b * C + c) * (long)hw;
    for (int i = threadIdx.x; i < hw; i += 256) {
      const double v = (double)p[i];
      a += v;
      q += v * v;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o, 64);
    q += __shfl_xor(q, o, 64);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][w] = a;
    red[1][w] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(sums + c, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(sums + C + c, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

}  // namespace vtp
using namespace vtp;

static inline int tok_grid(long items) {
  long b = (items + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

extern "C" int vtp_u8_to_images(const void* u8_nhwc, float* img_nchw, long B, int H, int W, const float* mean3, const float* std3,
                                int flip, void* stream) {
  VTP_REQUIRE(u8_nhwc && img_nchw && mean3 && std3 && B > 0 && H > 0 && W > 0 && W % 4 == 0, "vtp_u8_to_images: bad argument");
  const f32x4 m = {mean3[0], mean3[1], mean3[2], 0.f}, s = {std3[0], std3[1], std3[2], 1.f};
  hipLaunchKernelGGL(u8_to_images_kernel, dim3(tok_grid(B * 3 * H * (W / 4))), dim3(256), 0, (hipStream_t)stream,
                     (const uint8_t*)u8_nhwc, img_nchw, B, H, W, m, s, flip);
  return check_launch("u8_to_images");
}

extern "C" int vtp_images_to_u8(const float* img_nchw, void* u8_nhwc, long B, int H, int W, const float* sub3, const float* div3,
                                void* stream) {
  VTP_REQUIRE(img_nchw && u8_nhwc && sub3 && div3 && B > 0 && H > 0 && W > 0 && W % 4 == 0, "vtp_images_to_u8: bad argument");
  const f32x4 a = {sub3[0], sub3[1], sub3[2], 0.f}, d = {div3[0], div3[1], div3[2], 1.f};
  hipLaunchKernelGGL(images_to_u8_kernel, dim3(tok_grid(B * H * (W / 4))), dim3(256), 0, (hipStream_t)stream, img_nchw,
                     (uint8_t*)u8_nhwc, B, H, W, a, d);
  return check_launch("images_to_u8");
}

extern "C" int vtp_latent_channel_stats(const float* latents, double* sums, long B, int C, int hw, void* stream) {
  VTP_REQUIRE(latents && sums && B > 0 && C > 0 && hw > 0, "vtp_latent_channel_stats: bad argument");
  const int slices = (int)(B < 64 ? B : 64);
  hipLaunchKernelGGL(latent_stats_kernel, dim3(C * slices), dim3(256), 0, (hipStream_t)stream, latents, sums, B, C, hw, slices);
  return check_launch("latent_channel_stats");
}
