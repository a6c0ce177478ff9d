// Stride-1 5x5 transposed convolution with <= 4 output channels as a direct fp32 VALU
// kernel for gfx950: the decoder's last layer (deconv4 32 -> 3, model.py:507) and the data
// gradient of the discriminator's first layer (convs.0 3 <- 32, model.py:389).
//
// With 3 output channels an MFMA tile would be 3/32 full; the fp32 vector peak equals the
// fp32 MFMA peak on this chip, so a register-tiled FMA kernel is the right tool:
//   * workgroup = 16 x 64 output pixels of one image, thread = 4 consecutive pixels x COUT
//     accumulators; the input patch [CK][20][68] is staged in LDS with coalesced row loads
//     (zero-filled halo) and read back as aligned 16-byte vectors (8 floats feed 5 taps x
//     4 pixels), the filter taps are wave-uniform and come through scalar loads.
//   y[b][co][oh][ow] = bias[co] + sum_{ci,kh,kw} x[b][ci][oh+2-kh][ow+2-kw] * w[ci][co][kh][kw]
#include "common.hpp"
#include "vaegan_hip.h"

namespace {

constexpr int TH = 16, TW = 64, CK = 8, PH = TH + 4, PWS = TW + 4;  // PWS % 4 == 0: aligned b128 reads
constexpr int NT = 256;
constexpr int WSLOT = 20;   // taps of one (channel, kh): up to 4 output channels x 5 kw

template <int COUT>
__global__ __launch_bounds__(NT) void convT_s1_thin_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           int Cin, int H, int W, int tiles_w, int tiles_hw) {
  __shared__ __attribute__((aligned(16))) float patch[CK * PH * PWS];
  // filter taps of the chunk: [c][kh][co*5+kw], 20 floats per (c, kh) (4 cout x 5 kw) -> five broadcast
  // b128 reads
  __shared__ __attribute__((aligned(16))) float wl[CK * 5 * WSLOT];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / tiles_hw, sp = blockIdx.x % tiles_hw;
  const int oh0 = (sp / tiles_w) * TH, ow0 = (sp % tiles_w) * TW;
  const int r = tid / 16, cg = (tid % 16) * 4;   // this thread: row r, pixels cg..cg+3 of the tile
  const int HW = H * W;
  const float* xb = x + (size_t)b * Cin * HW;

  float acc[COUT][4];
#pragma unroll
  for (int co = 0; co < COUT; ++co)
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[co][p] = 0.f;

  for (int e = tid; e < CK * PH * PWS; e += NT) patch[e] = 0.f;   // halo words stay 0 for the whole kernel
  __syncthreads();

  for (int c0 = 0; c0 < Cin; c0 += CK) {
    // stage patch rows oh0-2 .. oh0+TH+1, cols ow0-2 .. ow0+TW+1 of CK channels with direct
    // global -> LDS loads (no VGPR round trip, all of a wave's rows in flight at once): a
    // wavefront copies one 68-float row per step (64 lanes + a 4-lane tail).  Lanes / rows
    // outside the image are masked off; those LDS words were zeroed once before the loop (the
    // halo pattern of a tile is the same for every chunk) -- only a channel tail is re-zeroed.
    {
      const int lane = tid & 63, wid = tid >> 6;
      for (int rr = wid; rr < CK * PH; rr += NT / 64) {
        const int c = rr / PH, pr = rr - c * PH;
        const int ih = oh0 - 2 + pr;
        const bool cok = (c0 + c) < Cin;
        const bool rok = cok && ih >= 0 && ih < H;
        const float* src = xb + (size_t)min(c0 + c, Cin - 1) * HW + (size_t)min(max(ih, 0), H - 1) * W;
        const int iw0 = ow0 - 2 + lane, iw1 = iw0 + 64;
        if (rok && iw0 >= 0 && iw0 < W)
          __builtin_amdgcn_global_load_lds(src + iw0, patch + rr * PWS, 4, 0, 0);
        if (rok && lane < PWS - 64 && iw1 >= 0 && iw1 < W)
          __builtin_amdgcn_global_load_lds(src + iw1, patch + rr * PWS + 64, 4, 0, 0);
        if (!cok && ih >= 0 && ih < H) {      // channel tail (Cin % CK != 0): stale data -> 0
          patch[rr * PWS + lane] = 0.f;
          if (lane < PWS - 64) patch[rr * PWS + 64 + lane] = 0.f;
        }
      }
      for (int e = tid; e < CK * 5 * WSLOT; e += NT) {
        const int j = e % WSLOT, kh = (e / WSLOT) % 5, c = e / (5 * WSLOT);
        const int co = j / 5, kw = j - co * 5;
        const bool ok = j < COUT * 5 && (c0 + c) < Cin;
        const float v = w[((size_t)min(c0 + c, Cin - 1) * COUT + min(co, COUT - 1)) * 25 + kh * 5 + kw];
        wl[e] = ok ? v : 0.f;
      }
    }
    __syncthreads();
    const int cmax = min(CK, Cin - c0);
    for (int c = 0; c < cmax; ++c) {
#pragma unroll
      for (int kh = 0; kh < 5; ++kh) {
        // output row r needs input row r+2-kh  ->  patch row r + 4 - kh
        const float* row = patch + (c * PH + r + 4 - kh) * PWS + cg;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(row);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(row + 4);
        const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const f32x4* wq = reinterpret_cast<const f32x4*>(wl + (c * 5 + kh) * WSLOT);   // uniform: broadcast
        float wv[WSLOT];
#pragma unroll
        for (int j4 = 0; j4 < (COUT * 5 + 3) / 4; ++j4) {
          const f32x4 t = wq[j4];
          wv[4 * j4] = t[0];
          wv[4 * j4 + 1] = t[1];
          wv[4 * j4 + 2] = t[2];
          wv[4 * j4 + 3] = t[3];
        }
#pragma unroll
        for (int kw = 0; kw < 5; ++kw) {
#pragma unroll
          for (int co = 0; co < COUT; ++co) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
              acc[co][p] = fmaf(v[p + 4 - kw], wv[co * 5 + kw], acc[co][p]);   // iw = ow+2-kw
          }
        }
      }
    }
    __syncthreads();
  }
  const int oh = oh0 + r, ow = ow0 + cg;
  if (oh < H) {
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      const float bv = bias ? bias[co] : 0.f;
      float* yp = y + (((size_t)b * COUT + co) * H + oh) * W + ow;
      if (ow + 3 < W && (W & 3) == 0) {
        *reinterpret_cast<f32x4*>(yp) = f32x4{acc[co][0] + bv, acc[co][1] + bv, acc[co][2] + bv, acc[co][3] + bv};
      } else {
#pragma unroll
        for (int p = 0; p < 4; ++p)
          if (ow + p < W) yp[p] = acc[co][p] + bv;
      }
    }
  }
}

template <int COUT>
int launch_thin(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int H, int W,
                hipStream_t st) {
  const int tiles_w = cdiv(W, TW), tiles_h = cdiv(H, TH);
  const long grid = (long)B * tiles_w * tiles_h;
  if (grid <= 0 || grid > 0x7fffffffL) return VG_ERR_BAD_ARG;
  hipLaunchKernelGGL(convT_s1_thin_kernel<COUT>, dim3((unsigned)grid), dim3(NT), 0, st, x, w, bias, y, Cin, H, W,
                     tiles_w, tiles_w * tiles_h);
  VG_CHECK_LAUNCH();
  return 0;
}

}  // namespace

int vg_internal_convT_s1_thin(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int H,
                              int W, int Cout, hipStream_t st) {
  switch (Cout) {
    case 1: return launch_thin<1>(x, w, bias, y, B, Cin, H, W, st);
    case 2: return launch_thin<2>(x, w, bias, y, B, Cin, H, W, st);
    case 3: return launch_thin<3>(x, w, bias, y, B, Cin, H, W, st);
    case 4: return launch_thin<4>(x, w, bias, y, B, Cin, H, W, st);
    default: return VG_ERR_BAD_ARG;
  }
}
