// 5x5 / pad 2 / stride {1,2} convolution and transposed convolution (forward and
// data gradient) for gfx950 as an implicit GEMM on the fp32-input MFMA
// (v_mfma_f32_32x32x2_f32: exact fp32, a k-ordered fmaf chain), NCHW in / out.
//
//   vg_conv5x5_fwd   = nn.Conv2d forward   (/root/reference/models/model.py:450,453,456
//                      encoder; :389,392,395,398 discriminator) and the data
//                      gradient of the transposed convolutions.
//   vg_convT5x5_fwd  = nn.ConvTranspose2d forward with the literal output_size of
//                      model.py:558-564 (deconv1..4 at :495-507) and the data
//                      gradient of the convolutions.
//
// Work decomposition (one 256-thread workgroup = 4 wavefronts of 64 lanes):
//   * output tile = TN output channels ("rows" of D) x TM = NB*TH*TW pixels
//     ("columns"): D[cout][pixel], so the 32 lanes of an MFMA column group store
//     32 consecutive pixels of an NCHW row.
//   * the K loop runs over input-channel chunks of CK channels.  Per chunk the
//     workgroup stages into LDS (double buffered, next chunk prefetched into
//     registers under the MFMAs): the NCHW input patch [NB][CK][PH][PW] with
//     coalesced row loads and zero-filled halo, and the filter slab [TN][CK*NTAP].
//   * im2col never exists in memory: lane (j,h) reads patch[c0+2cp+h][...] at a
//     per-lane base + compile-time immediate; one staged input pixel feeds up to
//     25/S^2 MFMA operands.
//   * the stride-2 transposed convolution is decomposed into its 4 output-parity
//     classes (3x3, 3x2, 2x3, 2x2 taps): no zero insertion, no atomics; a block
//     owns one class of one input-space tile, heavy classes are dispatched first.
#include "common.hpp"
#include "vaegan_hip.h"

namespace {

enum { MODE_FWD = 0, MODE_TR = 1 };

template <int MODE_, int S_, int NB_, int TH_, int TW_, int TN_, int WC_, int WP_, int CK_>
struct Cfg {
  static constexpr int MODE = MODE_, S = S_, NB = NB_, TH = TH_, TW = TW_, TN = TN_, WC = WC_, WP = WP_,
                       CK = CK_;
  static constexpr int NT = 64 * WC * WP;
  static constexpr int TM = NB * TH * TW;
  static constexpr int FC = TN / 32 / WC;  // 32-row cout fragments per wave
  static constexpr int FP = TM / 32 / WP;  // 32-col pixel fragments per wave
  static constexpr int NTMAX = (MODE == MODE_FWD) ? 5 : (5 + S - 1) / S;  // taps per dim, largest class
  static constexpr int PH = (MODE == MODE_FWD) ? S * (TH - 1) + 5 : TH + NTMAX - 1;
  static constexpr int PW = (MODE == MODE_FWD) ? S * (TW - 1) + 5 : TW + NTMAX - 1;
  static constexpr int PWP = PW | 1;
  static constexpr int NP = NB * CK * PH * PWP;  // patch floats per stage
  static constexpr int NQP = cdiv(NP, NT);
  static constexpr int RLMAX = CK * NTMAX * NTMAX;
  static constexpr int WSMAX = RLMAX | 1;
  static constexpr int STAGE = NP + TN * WSMAX;
  static constexpr int NCLS = (MODE == MODE_FWD) ? 1 : S * S;
  static_assert(TN % (32 * WC) == 0 && TM % (32 * WP) == 0 && CK % 2 == 0, "tile shape");
  static_assert(RLMAX <= NT, "one filter row per pass at least");
};

struct Args {
  const float* x;
  const float* w;
  const float* bias;
  float* y;
  int B, Cin, XH, XW, Cout, YH, YW;
  int ntiles_n, tiles_w, tiles_hw, blocks_per_cls;
};

// R, SS: output parity class of the transposed convolution (0,0 for MODE_FWD).
template <class C, int R, int SS>
__device__ __forceinline__ void igemm_body(const Args& A, float* smem, int bid) {
  constexpr int MODE = C::MODE, S = C::S, NB = C::NB, TH = C::TH, TW = C::TW, TN = C::TN, CK = C::CK;
  constexpr int NT = C::NT, FC = C::FC, FP = C::FP, PH = C::PH, PW = C::PW, PWP = C::PWP;
  constexpr int NP = C::NP, NQP = C::NQP, NTMAX = C::NTMAX;
  constexpr int NTH = (MODE == MODE_FWD) ? 5 : (5 - R + S - 1) / S;   // taps along h in this class
  constexpr int NTW = (MODE == MODE_FWD) ? 5 : (5 - SS + S - 1) / S;
  constexpr int NTAP = NTH * NTW;
  constexpr int RL = CK * NTAP;    // filter row: one cout, CK channels, NTAP taps
  constexpr int WS = RL | 1;       // odd LDS row stride: conflict-free b32 operand reads
  constexpr int RP = NT / RL;      // filter rows staged per pass
  constexpr int NQW = cdiv(TN, RP);
  constexpr int PSTEP = (MODE == MODE_FWD) ? S : 1;  // patch step per tile pixel

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int h = lane >> 5, l32 = lane & 31;
  const int wc = wid % C::WC, wp = wid / C::WC;
  const int nt = bid % A.ntiles_n;  // blocks sharing an XCD (bid % 8) share a filter slab when ntiles_n | 8
  bid /= A.ntiles_n;
  const int sp = bid % A.tiles_hw, bg = bid / A.tiles_hw;
  const int th0 = (sp / A.tiles_w) * TH, tw0 = (sp % A.tiles_w) * TW;  // tile origin (tile space)
  const int b0 = bg * NB, n0 = nt * TN;
  const int Cin = A.Cin, Cout = A.Cout, XH = A.XH, XW = A.XW;
  const int HW = XH * XW;
  // patch origin in input coordinates
  const int ih0 = (MODE == MODE_FWD) ? th0 * S - 2 : th0 - (NTMAX - 1 - 2 / S);
  const int iw0 = (MODE == MODE_FWD) ? tw0 * S - 2 : tw0 - (NTMAX - 1 - 2 / S);
  const float* xb = A.x + (size_t)b0 * Cin * HW;

  // ---- per-thread patch staging map (invariant over the K loop)
  int pofs[NQP];
#pragma unroll
  for (int q = 0; q < NQP; ++q) {
    const int e = tid + q * NT;
    const int col = e % PWP;
    int t = e / PWP;
    const int r = t % PH;
    t /= PH;
    const int c = t % CK, nb = t / CK;
    const int ih = ih0 + r, iw = iw0 + col;
    const bool ok = (e < NP) && (col < PW) && ih >= 0 && ih < XH && iw >= 0 && iw < XW && (b0 + nb) < A.B;
    pofs[q] = ok ? (nb * Cin + c) * HW + ih * XW + iw : -1;
  }

  // ---- filter staging map: RL consecutive lanes stage one filter row
  const int w_sub = tid / RL, w_r = tid % RL;
  const bool wrow_ok = w_sub < RP;
  const int w_c = w_r / NTAP, w_t = w_r % NTAP;
  const int w_kh = (MODE == MODE_FWD) ? w_t / 5 : R + S * (w_t / NTW);
  const int w_kw = (MODE == MODE_FWD) ? w_t % 5 : SS + S * (w_t % NTW);
  // conv weight [Cout][Cin][25]; transposed-conv weight [Cin][Cout][25]
  const size_t w_row_stride = (MODE == MODE_FWD) ? (size_t)Cin * 25 : 25;
  const size_t w_ch_stride = (MODE == MODE_FWD) ? 25 : (size_t)Cout * 25;
  const float* wsrc = A.w + (size_t)(n0 + w_sub) * w_row_stride + w_c * w_ch_stride + w_kh * 5 + w_kw;

  float preg[NQP], wreg[NQW];
  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int q = 0; q < NQP; ++q) {
      const int e = tid + q * NT;
      const int c = (e / (PWP * PH)) % CK;
      const bool ok = pofs[q] >= 0 && (c0 + c) < Cin;
      preg[q] = ok ? xb[pofs[q] + c0 * HW] : 0.f;
    }
    const float* wp_ = wsrc + c0 * w_ch_stride;
    const bool cok = wrow_ok && (c0 + w_c) < Cin;
#pragma unroll
    for (int q = 0; q < NQW; ++q) {
      const int co_l = w_sub + q * RP;
      const bool ok = cok && co_l < TN && (n0 + co_l) < Cout;
      wreg[q] = ok ? wp_[(size_t)q * RP * w_row_stride] : 0.f;
    }
  };
  auto store_chunk = [&](float* st) {
#pragma unroll
    for (int q = 0; q < NQP; ++q) {
      const int e = tid + q * NT;
      if (e < NP) st[e] = preg[q];
    }
    float* wl = st + NP;
#pragma unroll
    for (int q = 0; q < NQW; ++q) {
      if (wrow_ok && (w_sub + q * RP) < TN) wl[(w_sub + q * RP) * WS + w_r] = wreg[q];
    }
  };

  // ---- per-lane operand bases
  int base_w[FC], base_p[FP];
#pragma unroll
  for (int g = 0; g < FC; ++g) base_w[g] = NP + ((wc * FC + g) * 32 + l32) * WS + h * NTAP;
#pragma unroll
  for (int f = 0; f < FP; ++f) {
    const int m = (wp * FP + f) * 32 + l32;
    const int nb = m / (TH * TW), r = m % (TH * TW);
    base_p[f] = ((nb * CK + h) * PH + PSTEP * (r / TW)) * PWP + PSTEP * (r % TW);
  }

  f32x16 acc[FC][FP];
#pragma unroll
  for (int g = 0; g < FC; ++g)
#pragma unroll
    for (int f = 0; f < FP; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[g][f][r] = 0.f;

  const int nchunks = (Cin + CK - 1) / CK;
  load_chunk(0);
  store_chunk(smem);
  __syncthreads();

  for (int ch = 0; ch < nchunks; ++ch) {
    const float* st = smem + (ch & 1) * C::STAGE;
    const bool more = (ch + 1) < nchunks;
    if (more) load_chunk((ch + 1) * CK);
#pragma unroll
    for (int cp = 0; cp < CK / 2; ++cp) {
#pragma unroll
      for (int t = 0; t < NTAP; ++t) {
        // patch row/col offset of this tap relative to the tile pixel
        const int ro = (MODE == MODE_FWD) ? t / 5 : NTMAX - 1 - t / NTW;
        const int cof = (MODE == MODE_FWD) ? t % 5 : NTMAX - 1 - t % NTW;
        float a[FC], b[FP];
#pragma unroll
        for (int g = 0; g < FC; ++g) a[g] = st[base_w[g] + cp * 2 * NTAP + t];
#pragma unroll
        for (int f = 0; f < FP; ++f) b[f] = st[base_p[f] + (2 * cp * PH + ro) * PWP + cof];
#pragma unroll
        for (int g = 0; g < FC; ++g)
#pragma unroll
          for (int f = 0; f < FP; ++f) acc[g][f] = mfma32(a[g], b[f], acc[g][f]);
      }
    }
    if (more) store_chunk(smem + ((ch + 1) & 1) * C::STAGE);
    __syncthreads();
  }

  // ---- epilogue: + bias, NCHW store
  const int YH = A.YH, YW = A.YW;
#pragma unroll
  for (int f = 0; f < FP; ++f) {
    const int m = (wp * FP + f) * 32 + l32;
    const int nb = m / (TH * TW), r = m % (TH * TW);
    const int th = th0 + r / TW, tw = tw0 + r % TW, b = b0 + nb;
    const int oh = (MODE == MODE_FWD) ? th : S * th + R;
    const int ow = (MODE == MODE_FWD) ? tw : S * tw + SS;
    const bool pok = b < A.B && oh < YH && ow < YW;
    float* yb = A.y + ((size_t)b * Cout * YH + oh) * YW + ow;
#pragma unroll
    for (int g = 0; g < FC; ++g) {
#pragma unroll
      for (int r16 = 0; r16 < 16; ++r16) {
        const int co = n0 + (wc * FC + g) * 32 + acc_row(r16, lane);
        if (pok && co < Cout) {
          const float bv = A.bias ? A.bias[co] : 0.f;
          yb[(size_t)co * YH * YW] = acc[g][f][r16] + bv;
        }
      }
    }
  }
}

template <class C>
__global__ __launch_bounds__(C::NT, 2) void conv5x5_igemm_kernel(Args A) {
  __shared__ float smem[2 * C::STAGE];
  int bid = blockIdx.x;
  if constexpr (C::NCLS == 1) {
    igemm_body<C, 0, 0>(A, smem, bid);
  } else {
    const int cls = bid / A.blocks_per_cls;  // class 0 (3x3 taps) first: longest blocks start earliest
    bid -= cls * A.blocks_per_cls;
    switch (cls) {
      case 0: igemm_body<C, 0, 0>(A, smem, bid); break;
      case 1: igemm_body<C, 0, 1>(A, smem, bid); break;
      case 2: igemm_body<C, 1, 0>(A, smem, bid); break;
      default: igemm_body<C, 1, 1>(A, smem, bid); break;
    }
  }
}

template <class C>
int launch(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int XH, int XW,
           int Cout, hipStream_t st) {
  Args A;
  A.x = x; A.w = w; A.bias = bias; A.y = y;
  A.B = B; A.Cin = Cin; A.XH = XH; A.XW = XW; A.Cout = Cout;
  int tsh, tsw;  // tile-space extent
  if (C::MODE == MODE_FWD) {
    A.YH = (XH - 1) / C::S + 1; A.YW = (XW - 1) / C::S + 1;
    tsh = A.YH; tsw = A.YW;
  } else {
    A.YH = XH * C::S; A.YW = XW * C::S;
    tsh = XH; tsw = XW;
  }
  const int tiles_h = cdiv(tsh, C::TH);
  A.tiles_w = cdiv(tsw, C::TW);
  A.tiles_hw = tiles_h * A.tiles_w;
  A.ntiles_n = cdiv(Cout, C::TN);
  const long per_cls = (long)A.ntiles_n * A.tiles_hw * cdiv(B, C::NB);
  A.blocks_per_cls = (int)per_cls;
  const long grid = per_cls * C::NCLS;
  if (grid <= 0 || grid > 0x7fffffffL) return VG_ERR_BAD_ARG;
  hipLaunchKernelGGL(conv5x5_igemm_kernel<C>, dim3((unsigned)grid), dim3(C::NT), 0, st, A);
  VG_CHECK_LAUNCH();
  return 0;
}

//                       MODE     S NB TH TW  TN WC WP CK
using F2_w32_n128 = Cfg<MODE_FWD, 2, 1, 4, 32, 128, 2, 2, 2>;
using F2_w16_n128 = Cfg<MODE_FWD, 2, 1, 8, 16, 128, 2, 2, 2>;
using F2_w8_n128 = Cfg<MODE_FWD, 2, 2, 8, 8, 128, 2, 2, 2>;
using F2_w32_n64 = Cfg<MODE_FWD, 2, 1, 4, 32, 64, 2, 2, 2>;
using F2_w16_n64 = Cfg<MODE_FWD, 2, 1, 8, 16, 64, 2, 2, 2>;
using F2_w8_n64 = Cfg<MODE_FWD, 2, 2, 8, 8, 64, 2, 2, 2>;
using F2_w32_n32 = Cfg<MODE_FWD, 2, 1, 4, 32, 32, 1, 4, 2>;
using F2_w8_n32 = Cfg<MODE_FWD, 2, 2, 8, 8, 32, 1, 4, 2>;
using F1_w32_n128 = Cfg<MODE_FWD, 1, 1, 4, 32, 128, 2, 2, 2>;
using F1_w32_n32 = Cfg<MODE_FWD, 1, 1, 4, 32, 32, 1, 4, 2>;
using F1_w8_n32 = Cfg<MODE_FWD, 1, 2, 8, 8, 32, 1, 4, 2>;

using T2_w32_n128 = Cfg<MODE_TR, 2, 1, 4, 32, 128, 2, 2, 4>;
using T2_w16_n128 = Cfg<MODE_TR, 2, 1, 8, 16, 128, 2, 2, 4>;
using T2_w8_n128 = Cfg<MODE_TR, 2, 2, 8, 8, 128, 2, 2, 4>;
using T2_w16_n64 = Cfg<MODE_TR, 2, 1, 8, 16, 64, 2, 2, 4>;
using T2_w8_n64 = Cfg<MODE_TR, 2, 2, 8, 8, 64, 2, 2, 4>;
using T2_w32_n32 = Cfg<MODE_TR, 2, 1, 4, 32, 32, 1, 4, 4>;
using T2_w8_n32 = Cfg<MODE_TR, 2, 2, 8, 8, 32, 1, 4, 4>;
using T1_w32_n128 = Cfg<MODE_TR, 1, 1, 4, 32, 128, 2, 2, 2>;
using T1_w32_n32 = Cfg<MODE_TR, 1, 1, 4, 32, 32, 1, 4, 2>;
using T1_w8_n32 = Cfg<MODE_TR, 1, 2, 8, 8, 32, 1, 4, 2>;

}  // namespace

extern "C" int vg_conv5x5_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Cin,
                              int H, int W, int Cout, int stride, void* stream) {
  if (!x || !w || !y || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return VG_ERR_BAD_ARG;
  if (stride != 1 && stride != 2) return VG_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int OW = (W - 1) / stride + 1;
  if (stride == 2) {
    if (Cout > 64) {
      if (OW >= 32) return launch<F2_w32_n128>(x, w, bias, y, B, Cin, H, W, Cout, st);
      if (OW >= 16) return launch<F2_w16_n128>(x, w, bias, y, B, Cin, H, W, Cout, st);
      return launch<F2_w8_n128>(x, w, bias, y, B, Cin, H, W, Cout, st);
    }
    if (Cout > 32) {
      if (OW >= 32) return launch<F2_w32_n64>(x, w, bias, y, B, Cin, H, W, Cout, st);
      if (OW >= 16) return launch<F2_w16_n64>(x, w, bias, y, B, Cin, H, W, Cout, st);
      return launch<F2_w8_n64>(x, w, bias, y, B, Cin, H, W, Cout, st);
    }
    if (OW >= 32) return launch<F2_w32_n32>(x, w, bias, y, B, Cin, H, W, Cout, st);
    return launch<F2_w8_n32>(x, w, bias, y, B, Cin, H, W, Cout, st);
  }
  if (Cout > 32) return launch<F1_w32_n128>(x, w, bias, y, B, Cin, H, W, Cout, st);
  if (OW >= 32) return launch<F1_w32_n32>(x, w, bias, y, B, Cin, H, W, Cout, st);
  return launch<F1_w8_n32>(x, w, bias, y, B, Cin, H, W, Cout, st);
}

extern "C" int vg_convT5x5_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Cin,
                               int H, int W, int Cout, int stride, void* stream) {
  if (!x || !w || !y || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return VG_ERR_BAD_ARG;
  if (stride != 1 && stride != 2) return VG_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (stride == 2) {  // tiles live in input space (H x W)
    if (Cout > 64) {
      if (W >= 32) return launch<T2_w32_n128>(x, w, bias, y, B, Cin, H, W, Cout, st);
      if (W >= 16) return launch<T2_w16_n128>(x, w, bias, y, B, Cin, H, W, Cout, st);
      return launch<T2_w8_n128>(x, w, bias, y, B, Cin, H, W, Cout, st);
    }
    if (Cout > 32) {
      if (W >= 16) return launch<T2_w16_n64>(x, w, bias, y, B, Cin, H, W, Cout, st);
      return launch<T2_w8_n64>(x, w, bias, y, B, Cin, H, W, Cout, st);
    }
    if (W >= 32) return launch<T2_w32_n32>(x, w, bias, y, B, Cin, H, W, Cout, st);
    return launch<T2_w8_n32>(x, w, bias, y, B, Cin, H, W, Cout, st);
  }
  if (Cout > 32) return launch<T1_w32_n128>(x, w, bias, y, B, Cin, H, W, Cout, st);
  if (W >= 32) return launch<T1_w32_n32>(x, w, bias, y, B, Cin, H, W, Cout, st);
  return launch<T1_w8_n32>(x, w, bias, y, B, Cin, H, W, Cout, st);
}
