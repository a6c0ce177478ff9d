// Stride-1 5x5 transposed convolution 32 -> (<= 3) channels in the split-bf16 arithmetic for gfx950: the decoder's last
// layer (deconv4 32 -> 3, /root/reference/models/model.py:507) and the data gradient of the discriminator's first layer
// (convs.0 3 <- 32, model.py:389).  One pass over the 32-channel tensor (67 MB at B = 128, 64 x 64), HBM-bound by design:
//
//   y[b][co][oh][ow] = bias[co] + sum_{ci,kh,kw} x[b][ci][oh+2-kh][ow+2-kw] * w[ci][co][kh][kw]
//
// An MFMA tile with the 3 output channels as one of its dimensions would be 3/32 full.  Instead the filter taps of a
// row join the channels: for an OUTPUT row oh and 16 consecutive INPUT columns
//
//   D[m][n = co*5 + kw] = sum_{kh} sum_{ci} x[ci][oh+2-kh][iw0+m] * w[ci][co][kh][kw]        (16 x 16, K = 5 x 32)
//
// is five v_mfma_f32_16x16x32_bf16 steps (one per kh, K = the 32 input channels, 15 of 16 columns used), and
// y[oh][ow][co] = sum_kw D[ow + 2 - kw - iw0][co*5 + kw] is a 5-term sum along an anti-diagonal, done through LDS.
//
//   * the small operand lives in registers: the filter as B fragments [kh][plane] (60 VGPRs for 3 planes), loaded and
//     split once per wavefront;
//   * the big operand never touches LDS: a wavefront owns a strip of 16 input columns and marches down the rows; the
//     A fragment of an input row (lane = column m, 8 channels: 8 loads of 4 bytes, 64-byte segments) goes global ->
//     registers four rows ahead, is split into planes once (with the producer's BatchNorm + activation applied if
//     given) and then serves the five output rows it contributes to from a 5-row register window;
//   * a workgroup = W / 16 wavefronts side by side covers whole rows, so the +-2 column spill of D stays inside the
//     workgroup's LDS row buffer ([W + 4 columns][16 + 1 floats], double-buffered: one barrier per output row);
//     thread (co, ow) sums its 5 terms, adds the bias and stores 256-byte runs.
// Arithmetic: operands split exactly into NP bf16 planes, the plane products with index sum < NP, fp32 accumulate
// (NP = 3: fp32-equivalent, as conv_ring.hip).
#include "common.hpp"
#include "vaegan_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

constexpr int TCIN = 32, TRB = 16;            // input channels (= one MFMA K), output rows per workgroup
constexpr int TPITCH = 17;                    // floats per column of the LDS row buffer (16 + 1: conflict-free reads)

struct TArgs {
  const float* x;
  const float* w;
  const float* bias;
  const float* in_scale;      // fused BatchNorm of the producer: x read as act(x * in_scale[ci] + in_shift[ci]); or NULL
  const float* in_shift;
  float in_slope;             // activation as max(v, 0) + slope * min(v, 0)
  float* y;
  int B, H, W, bands;
};

template <int NP>
__device__ __forceinline__ void split_frag(float* v, bf16x8* out) {
#pragma unroll
  for (int p = 0; p < NP; ++p) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const __bf16 h = (__bf16)v[j];
      out[p][j] = h;
      v[j] -= (float)h;
    }
  }
}

template <int COUT, int NP>
__global__ __launch_bounds__(512) void convT_s1_thin_mfma_kernel(TArgs A) {
  extern __shared__ float drow[];              // [2 buffers][W + 4 columns][TPITCH]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kblk = lane >> 4;   // fragment row (input column / filter column) and 8-channel block
  const int H = A.H, W = A.W, HW = H * W;
  const int b = blockIdx.x / A.bands, r0 = (blockIdx.x % A.bands) * TRB;
  const int iw0 = wid * 16;
  const int bufsz = (W + 4) * TPITCH;

  // ---- filter fragments: B[k = ci][n = co*5 + kw] of every kh, split once
  bf16x8 bf[5][NP];
  {
    const int co = m / 5, kw = m % 5;
#pragma unroll
    for (int kh = 0; kh < 5; ++kh) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float t = A.w[((kblk * 8 + j) * COUT + min(co, COUT - 1)) * 25 + kh * 5 + kw];   // read, then select
        v[j] = (m < COUT * 5) ? t : 0.f;
      }
      split_frag<NP>(v, bf[kh]);
    }
  }
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = A.in_scale ? A.in_scale[kblk * 8 + j] : 1.f;
    sh[j] = A.in_scale ? A.in_shift[kblk * 8 + j] : 0.f;
  }

  // zero the halo columns (2 on each side) of both row buffers once: no wavefront ever writes them
  for (int e = tid; e < 2 * 4 * TPITCH; e += blockDim.x) {
    const int bufi = e / (4 * TPITCH), r = (e / TPITCH) % 4, c = e % TPITCH;
    drow[bufi * bufsz + ((r < 2) ? r : W + r) * TPITCH + c] = 0.f;
  }

  const float* xp = A.x + (size_t)b * TCIN * HW + (size_t)(kblk * 8) * HW + iw0 + m;   // + ih * W + j * HW
  float raw[5][8];                             // input rows in flight (global -> registers), ring by (step % 5)
  bf16x8 win[5][NP];                           // planes of the last five input rows, ring by (step % 5)
  auto load_row = [&](int slot, int ih) {      // rows outside the image: not loaded (zeros at the split)
    if (ih >= 0 && ih < H) {
#pragma unroll
      for (int j = 0; j < 8; ++j) raw[slot][j] = xp[(size_t)ih * W + (size_t)j * HW];
    }
  };
  auto split_row = [&](int slot, int ih) {
    float v[8];
    const bool ok = ih >= 0 && ih < H;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = fmaf(raw[slot][j], sc[j], sh[j]);                   // the producing layer's BatchNorm ...
      t = act_slope(t, A.in_slope);                                 // ... and activation, on load
      v[j] = ok ? t : 0.f;                                          // padding pads the activated tensor
    }
    split_frag<NP>(v, win[slot]);
  };

  // this thread's outputs in the epilogue: (co, ow), for tid < COUT * W
  const int eco = tid / W, eow = tid % W;
  const bool eok = tid < COUT * W;
  const float ebias = (eok && A.bias) ? A.bias[eco] : 0.f;
  float* yp = A.y + ((size_t)b * COUT + (eok ? eco : 0)) * HW + eow;

  // Step s brings input row r0 - 2 + s into window slot s % 5 and (from s = 4 on) finishes output row r0 + s - 4,
  // which reads the rows of steps s - kh.  Loads run 4 steps ahead.
  const int nsteps = min(TRB, H - r0) + 4;
#pragma unroll
  for (int s = 0; s < 4; ++s) load_row(s, r0 - 2 + s);
  for (int s0 = 0; s0 < nsteps; s0 += 5) {
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int s = s0 + u;
      if (s < nsteps) {                                             // uniform
        load_row((u + 4) % 5, r0 - 2 + s + 4);                      // slot of step s - 1: split a step ago
        split_row(u, r0 - 2 + s);
        if (s >= 4) {
          const int oh = r0 + s - 4;
          // one accumulator per product class (plane index sum), issued so that no MFMA accumulates into the result
          // of the one just before it (that dependency is waited out in compiler-inserted s_nops); added smallest
          // class first
          f32x4 dc[NP];
#pragma unroll
          for (int c = 0; c < NP; ++c) dc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kh = 0; kh < 5; ++kh) {                          // input row oh + 2 - kh = step s - kh
            const bf16x8* a = win[(u + 5 - kh) % 5];
            if constexpr (NP == 3) {                                // classes 2 1 2 0 2 1
              dc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], bf[kh][0], dc[2], 0, 0, 0);
              dc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], bf[kh][0], dc[1], 0, 0, 0);
              dc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], bf[kh][1], dc[2], 0, 0, 0);
              dc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bf[kh][0], dc[0], 0, 0, 0);
              dc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bf[kh][2], dc[2], 0, 0, 0);
              dc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bf[kh][1], dc[1], 0, 0, 0);
            } else {                                                // classes 1 0 1
              dc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], bf[kh][0], dc[1], 0, 0, 0);
              dc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bf[kh][0], dc[0], 0, 0, 0);
              dc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bf[kh][1], dc[1], 0, 0, 0);
            }
          }
          f32x4 d = dc[NP - 1];
#pragma unroll
          for (int c = NP - 2; c >= 0; --c) d += dc[c];
          // D[row 4*kblk + r][column m] -> row buffer [input column + 2][n]
          float* buf = drow + (oh & 1) * bufsz;
#pragma unroll
          for (int r = 0; r < 4; ++r) buf[(iw0 + 4 * kblk + r + 2) * TPITCH + m] = d[r];
          __syncthreads();
          if (eok) {
            float acc = ebias;
#pragma unroll
            for (int kw = 0; kw < 5; ++kw) acc += buf[(eow + 4 - kw) * TPITCH + eco * 5 + kw];   // iw = ow + 2 - kw
            yp[(size_t)oh * W] = acc;
          }
        }
      }
    }
  }
}

template <int COUT, int NP>
int launch_thin_mfma(const TArgs& A, hipStream_t st) {
  const int threads = (A.W / 16) * 64;
  const size_t lds = (size_t)2 * (A.W + 4) * TPITCH * sizeof(float);
  hipLaunchKernelGGL((convT_s1_thin_mfma_kernel<COUT, NP>), dim3((unsigned)(A.B * A.bands)), dim3(threads), lds, st, A);
  VG_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" int vg_convT5x5_s1_thin_bf16split_ok(int Cin, int H, int W, int Cout) {
  return Cin == TCIN && Cout >= 1 && Cout <= 3 && H >= 1 && W >= 16 && W <= 128 && W % 16 == 0 && Cout * W <= (W / 16) * 64;
}

extern "C" int vg_convT5x5_s1_thin_bf16split(const float* x, const float* w, const float* bias, float* y, int B, int Cin,
                                             int H, int W, int Cout, int planes, const float* in_scale,
                                             const float* in_shift, int in_act, void* stream) {
  if (!x || !w || !y || B <= 0 || (planes != 2 && planes != 3)) return VG_ERR_BAD_ARG;
  if (!vg_convT5x5_s1_thin_bf16split_ok(Cin, H, W, Cout)) return VG_ERR_BAD_ARG;
  if ((in_scale == nullptr) != (in_shift == nullptr) || in_act < VG_ACT_NONE || in_act > VG_ACT_LRELU) return VG_ERR_BAD_ARG;
  TArgs A;
  A.x = x; A.w = w; A.bias = bias; A.y = y;
  A.in_scale = in_scale; A.in_shift = in_shift;
  A.in_slope = (!in_scale || in_act == VG_ACT_NONE) ? 1.f : (in_act == VG_ACT_RELU ? 0.f : 0.2f);
  A.B = B; A.H = H; A.W = W; A.bands = cdiv(H, TRB);
  if ((long)B * A.bands > 0x7fffffffL || (size_t)TCIN * H * W > 0x7fffffffUL) return VG_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  switch (Cout * 10 + planes) {
    case 12: return launch_thin_mfma<1, 2>(A, st);
    case 13: return launch_thin_mfma<1, 3>(A, st);
    case 22: return launch_thin_mfma<2, 2>(A, st);
    case 23: return launch_thin_mfma<2, 3>(A, st);
    case 32: return launch_thin_mfma<3, 2>(A, st);
    case 33: return launch_thin_mfma<3, 3>(A, st);
    default: return VG_ERR_BAD_ARG;
  }
}
