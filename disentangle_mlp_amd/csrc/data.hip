// Image I/O kernels either side of the training step, for gfx950 (HBM-bound byte work):
//
//   vg_u8_gather_normalize  the input pipeline of /root/reference/dataloader/dataset.py:37-43
//       (ToTensor + Normalize(0.5, 0.5) of a shuffled batch) on a uint8 image cache that is
//       resident in HBM: out[b][c][h][w] = (cache[index[b]][h][w][c] / 255 - mean) / std.
//   vg_minmax + vg_image_grid_u8  torchvision.utils.save_image(..., normalize=True) as called by
//       /root/reference/utils/utils.py:12,19,21,29,36 (torchvision 0.2.1 make_grid semantics:
//       min-max over the whole batch, +1e-5 in the denominator, padding 2, truncating 8-bit
//       quantisation) -- the uint8 HWC grid is what PIL writes, and what FID's Inception reads.
//
// All arithmetic is IEEE fp32 in the reference's operation order (correctly rounded division),
// so results are bit-identical to the CPU pipeline.
#include "common.hpp"
#include "vaegan_hip.h"

namespace {

constexpr int NT = 256;

// 4 consecutive pixels of one RGB image per thread: 12 contiguous bytes in, one float4 per
// channel plane out (consecutive threads -> consecutive 16-byte pieces of each plane).
__global__ __launch_bounds__(NT) void u8_gather_normalize_rgb_kernel(const uint8_t* __restrict__ cache,
                                                                    const int64_t* __restrict__ index,
                                                                    float* __restrict__ out, int HW, float mean,
                                                                    float stdv) {
  const int b = blockIdx.y;
  const int q = blockIdx.x * NT + threadIdx.x;   // pixel quad
  if (q * 4 >= HW) return;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(cache + (size_t)index[b] * HW * 3) + (size_t)q * 3;
  const uint32_t w0 = src[0], w1 = src[1], w2 = src[2];
  const uint8_t px[12] = {(uint8_t)w0, (uint8_t)(w0 >> 8), (uint8_t)(w0 >> 16), (uint8_t)(w0 >> 24),
                          (uint8_t)w1, (uint8_t)(w1 >> 8), (uint8_t)(w1 >> 16), (uint8_t)(w1 >> 24),
                          (uint8_t)w2, (uint8_t)(w2 >> 8), (uint8_t)(w2 >> 16), (uint8_t)(w2 >> 24)};
  float* ob = out + (size_t)b * 3 * HW + (size_t)q * 4;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    f32x4 v;
#pragma unroll
    for (int p = 0; p < 4; ++p) v[p] = ((float)px[p * 3 + c] / 255.0f - mean) / stdv;
    *reinterpret_cast<f32x4*>(ob + (size_t)c * HW) = v;
  }
}

// any channel count / size: one output element per thread
__global__ __launch_bounds__(NT) void u8_gather_normalize_kernel(const uint8_t* __restrict__ cache,
                                                                const int64_t* __restrict__ index,
                                                                float* __restrict__ out, int C, int HW, float mean,
                                                                float stdv) {
  const int b = blockIdx.y;
  const int e = blockIdx.x * NT + threadIdx.x;   // c * HW + pixel
  if (e >= C * HW) return;
  const int c = e / HW, p = e - c * HW;
  const uint8_t u = cache[((size_t)index[b] * HW + p) * C + c];
  out[(size_t)b * C * HW + e] = ((float)u / 255.0f - mean) / stdv;
}

// ---- min / max of a tensor (exact in any order); stage 1 -> ws[2*blocks], stage 2 -> out[2]
__global__ __launch_bounds__(NT) void minmax_partial_kernel(const float* __restrict__ x, size_t n,
                                                           float* __restrict__ ws) {
  __shared__ float red[2][NT / 64];
  float lo = INFINITY, hi = -INFINITY;
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
    const float v = x[i];
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_down(lo, o, 64));
    hi = fmaxf(hi, __shfl_down(hi, o, 64));
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { red[0][wid] = lo; red[1][wid] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) { lo = fminf(lo, red[0][i]); hi = fmaxf(hi, red[1][i]); }
    ws[2 * blockIdx.x] = lo;
    ws[2 * blockIdx.x + 1] = hi;
  }
}

__global__ __launch_bounds__(64) void minmax_final_kernel(const float* __restrict__ ws, int blocks,
                                                         float* __restrict__ out) {
  float lo = INFINITY, hi = -INFINITY;
  for (int i = threadIdx.x; i < blocks; i += 64) {
    lo = fminf(lo, ws[2 * i]);
    hi = fmaxf(hi, ws[2 * i + 1]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_down(lo, o, 64));
    hi = fmaxf(hi, __shfl_down(hi, o, 64));
  }
  if (threadIdx.x == 0) { out[0] = lo; out[1] = hi; }
}

// One thread per grid pixel; writes the 3 bytes of an HWC pixel.  minmax == nullptr: no
// normalisation (values are only scaled by 255, clamped and truncated).
__global__ __launch_bounds__(NT) void image_grid_u8_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ minmax,
                                                          uint8_t* __restrict__ grid, int B, int C, int H, int W,
                                                          int xmaps, int padding, int GH, int GW, float pad_value) {
  const int e = blockIdx.x * NT + threadIdx.x;
  if (e >= GH * GW) return;
  const int gy = e / GW, gx = e - gy * GW;
  const int ch = H + padding, cw = W + padding;
  int k = -1, py = 0, px = 0;
  if (B == 1 && padding == 0) {
    k = 0; py = gy; px = gx;
  } else {
    const int ty = gy - padding, tx = gx - padding;
    if (ty >= 0 && tx >= 0) {
      const int yy = ty / ch, xx = tx / cw;
      py = ty - yy * ch; px = tx - xx * cw;
      if (py < H && px < W && xx < xmaps && yy * xmaps + xx < B) k = yy * xmaps + xx;
    }
  }
  float lo = 0.f, den = 1.f;
  if (minmax) {
    lo = minmax[0];
    den = (float)((double)minmax[1] - (double)lo + 1e-5);   // python: max - min + 1e-5, then an fp32 divide
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = pad_value;
    if (k >= 0) {
      v = x[(((size_t)k * C + (C == 1 ? 0 : c)) * H + py) * W + px];
      if (minmax) v = (fminf(fmaxf(v, lo), minmax[1]) + (-lo)) / den;
    }
    v = fminf(fmaxf(v * 255.0f, 0.f), 255.0f);
    grid[(size_t)e * 3 + c] = (uint8_t)v;
  }
}

}  // namespace

extern "C" int vg_u8_gather_normalize(const uint8_t* cache, const int64_t* index, float* out, int B, int C, int H,
                                      int W, float mean, float stdv, void* stream) {
  if (!cache || !index || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || !(stdv != 0.f)) return VG_ERR_BAD_ARG;
  if (B > 65535) return VG_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int HW = H * W;
  if (C == 3 && (HW & 3) == 0 && ((uintptr_t)cache & 3) == 0 && ((uintptr_t)out & 15) == 0) {
    hipLaunchKernelGGL(u8_gather_normalize_rgb_kernel, dim3(cdiv(HW / 4, NT), B), dim3(NT), 0, st, cache, index, out,
                       HW, mean, stdv);
  } else {
    hipLaunchKernelGGL(u8_gather_normalize_kernel, dim3(cdiv(C * HW, NT), B), dim3(NT), 0, st, cache, index, out, C,
                       HW, mean, stdv);
  }
  VG_CHECK_LAUNCH();
  return 0;
}

static int minmax_blocks(size_t n) {
  const size_t b = (n + (size_t)NT * 8 - 1) / ((size_t)NT * 8);
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

extern "C" size_t vg_minmax_workspace_bytes(size_t n) { return (size_t)minmax_blocks(n) * 2 * sizeof(float); }

extern "C" int vg_minmax(const float* x, size_t n, float* out2, void* workspace, size_t workspace_bytes,
                         void* stream) {
  if (!x || !out2 || n == 0) return VG_ERR_BAD_ARG;
  const int blocks = minmax_blocks(n);
  if (!workspace || workspace_bytes < (size_t)blocks * 2 * sizeof(float)) return VG_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(minmax_partial_kernel, dim3(blocks), dim3(NT), 0, st, x, n, (float*)workspace);
  VG_CHECK_LAUNCH();
  hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(64), 0, st, (const float*)workspace, blocks, out2);
  VG_CHECK_LAUNCH();
  return 0;
}

extern "C" int vg_image_grid_shape(int B, int H, int W, int nrow, int padding, int* grid_h, int* grid_w) {
  if (B <= 0 || H <= 0 || W <= 0 || nrow <= 0 || padding < 0 || !grid_h || !grid_w) return VG_ERR_BAD_ARG;
  if (B == 1) {            // make_grid returns the single image un-padded
    *grid_h = H; *grid_w = W;
    return 0;
  }
  const int xmaps = nrow < B ? nrow : B, ymaps = (B + xmaps - 1) / xmaps;
  *grid_h = (H + padding) * ymaps + padding;
  *grid_w = (W + padding) * xmaps + padding;
  return 0;
}

extern "C" int vg_image_grid_u8(const float* x, const float* minmax, uint8_t* grid, int B, int C, int H, int W,
                                int nrow, int padding, float pad_value, void* stream) {
  if (!x || !grid || (C != 1 && C != 3)) return VG_ERR_BAD_ARG;
  int GH, GW;
  if (vg_image_grid_shape(B, H, W, nrow, padding, &GH, &GW)) return VG_ERR_BAD_ARG;
  const int xmaps = nrow < B ? nrow : B;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(image_grid_u8_kernel, dim3(cdiv(GH * GW, NT)), dim3(NT), 0, st, x, minmax, grid, B, C, H, W,
                     xmaps, B == 1 ? 0 : padding, GH, GW, pad_value);
  VG_CHECK_LAUNCH();
  return 0;
}
