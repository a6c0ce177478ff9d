// OPT-IN arithmetic mode of the 5x5 forward convolution (vg_conv5x5_fwd_bf16x3) for gfx950:
// every fp32 operand is split into two bf16 values (hi = bf16(x), lo = bf16(x - hi)) and each
// product is evaluated as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation -- 3 bf16 MFMAs per 16 k in 96 cycles where the exact-fp32 MFMA needs 8 x 64.
// Measured error of a 128 -> 256 stride-2 layer against fp64: ~4e-6 relative (fp32 path: 3.5e-7),
// so this is NOT the default: the product path stays exact fp32 (conv_igemm.hip); see DESIGN.md
// section 8 item 1 for the measurements that motivate this kernel.
//
// Same computation as vg_conv5x5_fwd (nn.Conv2d forward, /root/reference/models/model.py:389-398,
// 450-456, and the data gradient of the transposed convolutions :495-507); requires Cin % 16 == 0.
//
// Structure (one workgroup = 4 wavefronts, 128 output channels x 128 pixels, 2 x 2 fragments of
// 32 x 32 per wavefront):
//   * K step = 16 input channels of one tap (MFMA k-block 0 / 1 = channels 0-7 / 8-15).
//   * The input patch of a 16-channel chunk lives in LDS channel-innermost, [plane hi/lo][k-block]
//     [row][column parity][column] x 8 bf16 (16 B): a lane's B operand is one ds_read_b128 at a
//     per-lane base + compile-time tap offset; for stride 2 the even / odd columns are kept apart
//     so that the 32 pixels of a fragment read consecutive 16-byte units.  The fp32 -> hi/lo split
//     happens once per element, when the prefetched registers are written to LDS.
//   * The filter never touches LDS: it comes pre-split and pre-packed (vg_conv5x5_pack_bf16x3)
//     as [chunk*25 + tap][plane][k-block][cout] x 8 bf16, so a lane's A operand is one 16-byte
//     global load (32 consecutive cout = 512 contiguous bytes), prefetched one tap ahead.
#include "common.hpp"
#include "vaegan_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

constexpr int XNT = 256, XTN = 128;

template <int S_, int NB_, int TH_, int TW_>
struct XCfg {
  static constexpr int S = S_, NB = NB_, TH = TH_, TW = TW_, TM = NB * TH * TW;
  static constexpr int PH = S * (TH - 1) + 5, PW = S * (TW - 1) + 5;
  static constexpr int COLS = ((PW + S - 1) / S + 3) & ~3;   // 16-byte units per (row, parity); multiple of 4: rows 2 apart stay bank-disjoint
  static constexpr int ROWU = S * COLS;                       // units per patch row
  static constexpr int IMGU = NB * PH * ROWU;                 // units per (plane, k-block) image
  static constexpr int NUNIT = 2 * NB * PH * PW;              // staged units per chunk (k-block, image, row, real column)
  static constexpr int NQ = cdiv(NUNIT, XNT);
  static_assert(TM == 128, "pixel tile");
};

struct XArgs {
  const float* x;
  const bf16x8* w;     // packed filter
  const float* bias;
  float* y;
  int B, Cin, XH, XW, Cout, CoutP, YH, YW;
  int ntiles_n, tiles_w, tiles_hw, nblocks;
};

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

template <class C>
__global__ __launch_bounds__(XNT, 2) void conv5x5_fwd_bf16x3_kernel(XArgs A) {
  constexpr int S = C::S, NB = C::NB, TH = C::TH, TW = C::TW, PH = C::PH, PW = C::PW;
  constexpr int COLS = C::COLS, ROWU = C::ROWU, IMGU = C::IMGU, NQ = C::NQ;
  __shared__ f32x4 lds[4 * IMGU];     // [plane][k-block][image][row][parity][column]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int kb = lane >> 5, l32 = lane & 31;
  const int wc = wid & 1, wp = wid >> 1;
  // XCD-aware placement, as in conv_igemm.hip: the cout tiles of one pixel tile share an XCD
  int nt, pt;
  {
    const int bid = blockIdx.x, ntn = A.ntiles_n, npatch = A.nblocks / ntn, full = (npatch / 8) * 8 * ntn;
    if (bid < full) {
      const int xcd = bid & 7, j = bid >> 3;
      pt = (j / ntn) * 8 + xcd;
      nt = j % ntn;
    } else {
      const int t = bid - full;
      pt = (npatch / 8) * 8 + t / ntn;
      nt = t % ntn;
    }
  }
  const int sp = pt % A.tiles_hw, bg = pt / A.tiles_hw;
  const int th0 = (sp / A.tiles_w) * TH, tw0 = (sp % A.tiles_w) * TW;
  const int b0 = bg * NB, n0 = nt * XTN;
  const int Cin = A.Cin, Cout = A.Cout, XH = A.XH, XW = A.XW, HW = XH * XW;
  const int ih0 = th0 * S - 2, iw0 = tw0 * S - 2;
  const float* xb = A.x + (size_t)b0 * Cin * HW;

  // ---- staging map: unit e = (k-block, image, row, column); addresses clamped, validity masked
  int pofs[NQ], pdst[NQ];
  unsigned pvalid = 0;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int e = tid + q * XNT;
    const int col = e % PW;
    int t = e / PW;
    const int r = t % PH;
    t /= PH;
    const int nb = t % NB, kbs = min(t / NB, 1);
    const int ih = ih0 + r, iw = iw0 + col;
    const bool ok = (e < C::NUNIT) && ih >= 0 && ih < XH && iw >= 0 && iw < XW && (b0 + nb) < A.B;
    const int nbc = min(nb, A.B - 1 - b0), ihc = min(max(ih, 0), XH - 1), iwc = min(max(iw, 0), XW - 1);
    pofs[q] = (nbc * Cin + kbs * 8) * HW + ihc * XW + iwc;
    pdst[q] = (e < C::NUNIT) ? kbs * IMGU + (nb * PH + r) * ROWU + (S == 2 ? (col & 1) * COLS + (col >> 1) : col) : -1;
    pvalid |= ok ? (1u << q) : 0u;
  }
  static_assert(NQ <= 32, "validity mask");

  float preg[NQ][8];
  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) preg[q][j] = xb[pofs[q] + (c0 + j) * HW];
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const bool ok = (pvalid >> q) & 1u;
      bf16x8 hi, lo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = ok ? preg[q][j] : 0.f;
        const __bf16 h = (__bf16)v;
        hi[j] = h;
        lo[j] = (__bf16)(v - (float)h);
      }
      if (pdst[q] >= 0) {
        lds[pdst[q]] = __builtin_bit_cast(f32x4, hi);
        lds[pdst[q] + 2 * IMGU] = __builtin_bit_cast(f32x4, lo);
      }
    }
  };

  // ---- per-lane operand bases
  int base_b[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int m = (wp * 2 + f) * 32 + l32;
    const int nb = m / (TH * TW), r = m % (TH * TW);
    base_b[f] = kb * IMGU + (nb * PH + S * (r / TW)) * ROWU + (r % TW);   // column S*pw: parity 0, index pw
  }
  const int CoutP = A.CoutP;
  const bf16x8* wa[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) wa[g] = A.w + (size_t)kb * CoutP + n0 + (wc * 2 + g) * 32 + l32;
  const size_t wstep = (size_t)4 * CoutP;    // units per (chunk, tap) step

  f32x16 acc[2][2];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[g][f][r] = 0.f;

  const int nchunks = Cin / 16;
  load_chunk(0);
  store_chunk();
  bf16x8 a_hi[2][2], a_lo[2][2];     // [buffer][fragment]
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    a_hi[0][g] = wa[g][0];
    a_lo[0][g] = wa[g][2 * CoutP];
  }
  __syncthreads();

  for (int ch = 0; ch < nchunks; ++ch) {
    const bool more = (ch + 1) < nchunks;
    if (more) load_chunk((ch + 1) * 16);
    // one filter row (5 taps) per trip of a rolled loop: keeps the filter prefetch one tap deep
    // (fully unrolled, hipcc hoists all 25 taps' loads and spills); 5 is odd, so the prefetched
    // step is moved back into buffer 0 at the end of every row and `cur` stays compile-time.
#pragma unroll 1
    for (int kh = 0; kh < 5; ++kh) {
      const bf16x8* wrow[2] = {wa[0] + ((size_t)ch * 25 + kh * 5) * wstep, wa[1] + ((size_t)ch * 25 + kh * 5) * wstep};
      const int rowoff = kh * ROWU;
#pragma unroll
      for (int kw = 0; kw < 5; ++kw) {
        const int cur = kw & 1, nxt = cur ^ 1;
        // next tap's filter fragments (the pack has one spare step after the last one)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          a_hi[nxt][g] = wrow[g][(size_t)(kw + 1) * wstep];
          a_lo[nxt][g] = wrow[g][(size_t)(kw + 1) * wstep + 2 * CoutP];
        }
        const int imm = (S == 2 ? (kw & 1) * COLS + (kw >> 1) : kw);
        bf16x8 b_hi[2], b_lo[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          b_hi[f] = __builtin_bit_cast(bf16x8, lds[base_b[f] + rowoff + imm]);
          b_lo[f] = __builtin_bit_cast(bf16x8, lds[base_b[f] + rowoff + imm + 2 * IMGU]);
        }
        // small terms first, product-major: four independent accumulators between dependent MFMAs
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int f = 0; f < 2; ++f) acc[g][f] = mfma_bf16(a_lo[cur][g], b_hi[f], acc[g][f]);
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int f = 0; f < 2; ++f) acc[g][f] = mfma_bf16(a_hi[cur][g], b_lo[f], acc[g][f]);
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int f = 0; f < 2; ++f) acc[g][f] = mfma_bf16(a_hi[cur][g], b_hi[f], acc[g][f]);
      }
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        a_hi[0][g] = a_hi[1][g];
        a_lo[0][g] = a_lo[1][g];
      }
    }
    __syncthreads();
    if (more) {
      store_chunk();
      __syncthreads();
    }
  }

  // ---- epilogue: + bias, NCHW store (as conv_igemm.hip)
  const int YH = A.YH, YW = A.YW;
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int m = (wp * 2 + f) * 32 + l32;
    const int nb = m / (TH * TW), r = m % (TH * TW);
    const int oh = th0 + r / TW, ow = tw0 + r % TW, b = b0 + nb;
    const bool pok = b < A.B && oh < YH && ow < YW;
    float* yb = A.y + ((size_t)b * Cout * YH + oh) * YW + ow;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      float bv[16];
#pragma unroll
      for (int r16 = 0; r16 < 16; ++r16) {
        const int co = min(n0 + (wc * 2 + g) * 32 + acc_row(r16, lane), Cout - 1);
        bv[r16] = A.bias ? A.bias[co] : 0.f;
      }
#pragma unroll
      for (int r16 = 0; r16 < 16; ++r16) {
        const int co = n0 + (wc * 2 + g) * 32 + acc_row(r16, lane);
        if (pok && co < Cout) yb[(size_t)co * YH * YW] = acc[g][f][r16] + bv[r16];
      }
    }
  }
}

template <class C>
int launch_x(const float* x, const bf16x8* w, const float* bias, float* y, int B, int Cin, int XH, int XW, int Cout,
             hipStream_t st) {
  XArgs A;
  A.x = x; A.w = w; A.bias = bias; A.y = y;
  A.B = B; A.Cin = Cin; A.XH = XH; A.XW = XW; A.Cout = Cout; A.CoutP = (Cout + 127) & ~127;
  A.YH = (XH - 1) / C::S + 1; A.YW = (XW - 1) / C::S + 1;
  A.tiles_w = cdiv(A.YW, C::TW);
  A.tiles_hw = cdiv(A.YH, C::TH) * A.tiles_w;
  A.ntiles_n = cdiv(Cout, XTN);
  const long grid = (long)A.ntiles_n * A.tiles_hw * cdiv(B, C::NB);
  if (grid <= 0 || grid > 0x7fffffffL) return VG_ERR_BAD_ARG;
  A.nblocks = (int)grid;
  hipLaunchKernelGGL(conv5x5_fwd_bf16x3_kernel<C>, dim3((unsigned)grid), dim3(XNT), 0, st, A);
  VG_CHECK_LAUNCH();
  return 0;
}

template <int S>
int dispatch_x(const float* x, const bf16x8* w, const float* bias, float* y, int B, int Cin, int XH, int XW, int Cout,
               hipStream_t st) {
  const int tsw = (XW - 1) / S + 1;
  if (tsw >= 32) return launch_x<XCfg<S, 1, 4, 32>>(x, w, bias, y, B, Cin, XH, XW, Cout, st);
  if (tsw >= 16) return launch_x<XCfg<S, 1, 8, 16>>(x, w, bias, y, B, Cin, XH, XW, Cout, st);
  return launch_x<XCfg<S, 2, 8, 8>>(x, w, bias, y, B, Cin, XH, XW, Cout, st);
}

// packed[(chunk*25 + tap)][plane][k-block][CoutP] x 8 bf16; one spare (zero) step at the end.
__global__ __launch_bounds__(256) void pack_bf16x3_kernel(const float* __restrict__ w, bf16x8* __restrict__ p,
                                                         int Cout, int Cin, int CoutP, int nsteps) {
  const int co = blockIdx.x * 256 + threadIdx.x;
  const int s = blockIdx.y;                 // chunk*25 + tap, nsteps = spare
  if (co >= CoutP) return;
  const int c16 = s / 25, tap = s % 25;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    bf16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ci = c16 * 16 + kb * 8 + j;
      const float v = (s < nsteps && co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * 25 + tap] : 0.f;
      const __bf16 h = (__bf16)v;
      hi[j] = h;
      lo[j] = (__bf16)(v - (float)h);
    }
    p[((size_t)s * 4 + kb) * CoutP + co] = hi;
    p[((size_t)s * 4 + 2 + kb) * CoutP + co] = lo;
  }
}

}  // namespace

extern "C" size_t vg_conv5x5_packed_bf16x3_bytes(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0 || Cin % 16) return 0;
  return (size_t)(Cin / 16 * 25 + 1) * 4 * ((Cout + 127) & ~127) * 16;
}

extern "C" int vg_conv5x5_pack_bf16x3(const float* w, void* packed, int Cout, int Cin, void* stream) {
  if (!w || !packed || Cout <= 0 || Cin <= 0 || Cin % 16 || ((uintptr_t)packed & 15)) return VG_ERR_BAD_ARG;
  const int CoutP = (Cout + 127) & ~127, nsteps = Cin / 16 * 25;
  hipLaunchKernelGGL(pack_bf16x3_kernel, dim3(cdiv(CoutP, 256), nsteps + 1), dim3(256), 0, (hipStream_t)stream, w,
                     (bf16x8*)packed, Cout, Cin, CoutP, nsteps);
  VG_CHECK_LAUNCH();
  return 0;
}

extern "C" int vg_conv5x5_fwd_bf16x3(const float* x, const void* packed, const float* bias, float* y, int B, int Cin,
                                     int H, int W, int Cout, int stride, void* stream) {
  if (!x || !packed || !y || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return VG_ERR_BAD_ARG;
  if ((stride != 1 && stride != 2) || Cin % 16 || ((uintptr_t)packed & 15)) return VG_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const bf16x8* w = (const bf16x8*)packed;
  if (stride == 2) return dispatch_x<2>(x, w, bias, y, B, Cin, H, W, Cout, st);
  return dispatch_x<1>(x, w, bias, y, B, Cin, H, W, Cout, st);
}
