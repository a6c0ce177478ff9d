// OPT-IN arithmetic mode of the 5x5 convolution / transposed convolution forward kernels
// (vg_conv5x5_fwd_bf16split, vg_convT5x5_fwd_bf16split) for gfx950: every fp32 operand is split into two
// bf16 values (hi = bf16(x), lo = bf16(x - hi)) and each product is evaluated as
// lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- 3 bf16 MFMAs per
// 16 k in 96 cycles where the exact-fp32 MFMA needs 8 x 64.  Measured error against fp64:
// ~4.5e-6 relative (exact-fp32 path: 5e-7 .. 1e-6), so this is NOT the default: the product path
// stays exact fp32 (conv_igemm.hip).  DESIGN.md section 8 has the measurements behind it.
//
// Same computations as vg_conv5x5_fwd / vg_convT5x5_fwd (nn.Conv2d / nn.ConvTranspose2d forward of
// /root/reference/models/model.py:389-398, 450-456, 495-507 and each other's data gradients);
// requires Cin % 16 == 0.
//
// Structure (one workgroup = 4 wavefronts as WC x WP, FC x FP fragments of 32 x 32 per wavefront):
//   * K step = 16 input channels of one tap (MFMA k-block 0 / 1 = channels 0-7 / 8-15).
//   * The input patch of a 16-channel chunk lives in LDS channel-innermost, [plane hi/lo][k-block]
//     [image][row][column] x 8 bf16 (16 B): a lane's B operand is one ds_read_b128 at a per-lane
//     base + tap offset.  For the stride-2 forward convolution the even / odd columns of a row are
//     kept apart so that the 32 pixels of a fragment read consecutive 16-byte units; row strides are
//     chosen so that the rows a fragment spans fall on disjoint banks.  The fp32 -> hi/lo split
//     happens once per element, when the prefetched registers are written to LDS.
//   * The filter never touches LDS: it comes pre-split and pre-packed (vg_conv5x5_pack_bf16split) as
//     [parity class][chunk][tap][plane][k-block][cout] x 8 bf16, so a lane's A operand is one 16-byte
//     global load (32 consecutive cout = 512 contiguous bytes), prefetched one tap ahead.
//   * The stride-2 transposed convolution runs as its 4 output-parity classes (3x3, 3x2, 2x3, 2x2
//     taps), exactly as in conv_igemm.hip.
#include "common.hpp"
#include "vaegan_hip.h"

namespace {

enum { X_FWD = 0, X_TR = 1 };
constexpr int XNT = 256;
constexpr int VG_PACK_SPARE = 6;   // zero steps behind a pack (the ring kernel's DMA runs up to six steps ahead: RING_MAX_SLOTS)

// 16-byte units per patch row (FWD stride 2: per column parity).  Chosen so that the patch rows a
// 32-pixel fragment spans start on disjoint groups of 16 units (= 256 B, one LDS bank row):
// fragments of 2 rows x 16 pixels need the two rows 0 (mod 16) apart, 4 rows x 8 pixels 8 (mod 16).
constexpr int row_units(int mode, int S, int TW, int PW) {
  if (mode == X_FWD && S == 2) {                       // returns COLS (per parity); a row is 2 * COLS units
    const int need = (PW + 1) / 2;
    if (TW == 8) { int c = need; while ((c & 3) != 2) ++c; return c; }   // 2 rows * 2 * COLS = 8 (mod 16)
    return (need + 3) & ~3;                                              // 2 rows * 2 * COLS = 0 (mod 16)
  }
  if (TW == 16) return (PW + 15) & ~15;
  if (TW == 8) { int c = PW; while ((c & 15) != 8) ++c; return c; }
  return (PW + 3) & ~3;
}

// WC x WP wavefronts (WC * WP = 4), FC x FP fragments each: cout tile 32*WC*FC, pixel tile 32*WP*FP
// NP_: operand planes -- 2: hi/lo split, 3 products ("bf16x3", ~4.5e-6); 3: exact hi/mid/lo split of the fp32
// mantissa (8 + 8 + 8 bits), 6 products, every dropped term below 2^-24 ("bf16x6", fp32-equivalent)
template <int MODE_, int S_, int NB_, int TH_, int TW_, int WC_, int FC_, int FP_, int NP_, bool F16_ = false>
struct XCfg {
  static constexpr int MODE = MODE_, S = S_, NB = NB_, TH = TH_, TW = TW_, WC = WC_, WP = 4 / WC_, FC = FC_, FP = FP_;
  static constexpr int NP = NP_;
  static constexpr bool F16 = F16_;       // fp16 planes (NP = 2): common.hpp, "split arithmetics"
  static_assert(!F16_ || NP_ == 2, "fp16 planes: hi + lo");
  static constexpr int TN = 32 * WC * FC, TM = NB * TH * TW;
  static constexpr int NTMAX = (MODE == X_FWD) ? 5 : (5 + S - 1) / S;
  static constexpr int PH = (MODE == X_FWD) ? S * (TH - 1) + 5 : TH + NTMAX - 1;
  static constexpr int PW = (MODE == X_FWD) ? S * (TW - 1) + 5 : TW + NTMAX - 1;
  static constexpr bool SPLIT = (MODE == X_FWD && S == 2);       // even / odd columns kept apart
  static constexpr int COLS = row_units(MODE, S, TW, PW);
  static constexpr int ROWU = SPLIT ? 2 * COLS : COLS;            // units per patch row
  static constexpr int IMGU = NB * PH * ROWU;                     // units per (plane, k-block) image
  static constexpr int NUNIT = 2 * NB * PH * PW;                  // staged units per chunk
  static constexpr int NQ = cdiv(NUNIT, XNT);
  static constexpr int NCLS = (MODE == X_FWD) ? 1 : S * S;
  static_assert(TM == 32 * WP * FP, "pixel tile");
  static_assert(WC == 1 || WC == 2 || WC == 4, "wavefront grid");
};

struct XArgs {
  const float* x;
  const bf16x8* w;     // packed filter
  const float* bias;
  float* y;
  int B, Cin, XH, XW, Cout, CoutP, YH, YW;
  int ntiles_n, tiles_w, tiles_hw, blocks_per_cls;
  int ksplit, cps;       // forward only: grid-level split of the channel chunks (deep-K, small-grid layers)
  size_t ysplit;         // elements per partial output slab (then y points at the slabs)
  const float* in_amax;  // fp16 planes: in_amax[0] >= max |x| (device); w_unscale[0]: the pack's trailer
  const float* w_unscale;
};

// taps of the output-parity classes before (R, SS) (same order as conv_igemm.hip)
__host__ __device__ constexpr int x_taps_before(int S, int R, int SS) {
  int n = 0;
  for (int r = 0; r < S; ++r)
    for (int s = 0; s < S; ++s) {
      if (r == R && s == SS) return n;
      n += ((5 - r + S - 1) / S) * ((5 - s + S - 1) / S);
    }
  return n;
}

template <class C, int R, int SS>
__device__ __forceinline__ void bf16split_body(const XArgs& A, f32x4* lds, int bid, int split) {
  constexpr int MODE = C::MODE, S = C::S, NB = C::NB, TH = C::TH, TW = C::TW, PH = C::PH, PW = C::PW;
  constexpr int COLS = C::COLS, ROWU = C::ROWU, IMGU = C::IMGU, NQ = C::NQ, FC = C::FC, FP = C::FP, NP = C::NP;
  constexpr int NTMAX = C::NTMAX;
  constexpr bool F16 = C::F16;
  constexpr int NTH = (MODE == X_FWD) ? 5 : (5 - R + S - 1) / S;   // taps along h / w in this class
  constexpr int NTW = (MODE == X_FWD) ? 5 : (5 - SS + S - 1) / S;
  constexpr int NTAP = NTH * NTW;
  constexpr int PSTEP = (MODE == X_FWD) ? S : 1;                   // patch rows / columns per tile pixel

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int kb = lane >> 5, l32 = lane & 31;
  const int wc = wid % C::WC, wp = wid / C::WC;
  // XCD-aware placement, as in conv_igemm.hip: the cout tiles of one pixel tile share an XCD
  int nt, pt;
  {
    const int ntn = A.ntiles_n, npatch = A.blocks_per_cls / ntn, full = (npatch / 8) * 8 * ntn;
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
  const int b0 = bg * NB, n0 = nt * C::TN;
  const int Cin = A.Cin, Cout = A.Cout, XH = A.XH, XW = A.XW, HW = XH * XW;
  const int ih0 = (MODE == X_FWD) ? th0 * S - 2 : th0 - (NTMAX - 1 - 2 / S);
  const int iw0 = (MODE == X_FWD) ? tw0 * S - 2 : tw0 - (NTMAX - 1 - 2 / S);
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
    pdst[q] = (e < C::NUNIT) ? kbs * IMGU + (nb * PH + r) * ROWU + (C::SPLIT ? (col & 1) * COLS + (col >> 1) : col) : -1;
    pvalid |= ok ? (1u << q) : 0u;
  }
  static_assert(NQ <= 32, "validity mask");

  float x_scale = 1.f, x_unscale = 1.f;     // fp16 planes: exact power of two from the caller's bound on max |x|
  if constexpr (F16) {
    const float amax = *A.in_amax;
    x_scale = f16_scale_of(amax);
    x_unscale = f16_unscale_of(amax);
  }
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
      f32x4 pl[NP];
      float vv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = ok ? preg[q][j] : 0.f;
        vv[j] = F16 ? v * x_scale : v;
      }
      split_planes16<NP, F16>(vv, pl);
      if (pdst[q] >= 0) {
#pragma unroll
        for (int p = 0; p < NP; ++p) lds[pdst[q] + p * 2 * IMGU] = pl[p];
      }
    }
  };

  // ---- per-lane operand bases
  int base_b[FP];
#pragma unroll
  for (int f = 0; f < FP; ++f) {
    const int m = (wp * FP + f) * 32 + l32;
    const int nb = m / (TH * TW), r = m % (TH * TW);
    // FWD stride 2: column 2*pw has parity 0 and index pw
    base_b[f] = kb * IMGU + (nb * PH + PSTEP * (r / TW)) * ROWU + (C::SPLIT ? (r % TW) : PSTEP * (r % TW));
  }
  const int CoutP = A.CoutP;
  const size_t wstep = (size_t)2 * NP * CoutP;    // units per (chunk, tap) step
  const int nchunks = Cin / 16;
  const bf16x8* wa[FC];                      // this class' first step, this lane's cout and k-block
#pragma unroll
  for (int g = 0; g < FC; ++g)
    wa[g] = A.w + (size_t)x_taps_before(S, R, SS) * nchunks * wstep + (size_t)kb * CoutP + n0 + (wc * FC + g) * 32 + l32;

  f32x16 acc[FC][FP];
#pragma unroll
  for (int g = 0; g < FC; ++g)
#pragma unroll
    for (int f = 0; f < FP; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[g][f][r] = 0.f;

  const int c_begin = split * A.cps, c_end = min(c_begin + A.cps, nchunks);   // this workgroup's channel chunks
  load_chunk(c_begin * 16);
  store_chunk();
  bf16x8 av[2][FC][NP];     // [buffer][fragment][plane]
#pragma unroll
  for (int g = 0; g < FC; ++g)
#pragma unroll
    for (int p = 0; p < NP; ++p) av[0][g][p] = wa[g][(size_t)c_begin * NTAP * wstep + (size_t)p * 2 * CoutP];
  __syncthreads();

  for (int ch = c_begin; ch < c_end; ++ch) {
    const bool more = (ch + 1) < c_end;
    if (more) load_chunk((ch + 1) * 16);
    // one filter row (NTW taps) per trip of a rolled loop: keeps the filter prefetch one tap deep
    // (fully unrolled, hipcc hoists every tap's loads and spills).  When NTW is odd the prefetched
    // step is moved back into buffer 0 at the end of the row, so `cur` stays compile-time.
#pragma unroll 1
    for (int ta = 0; ta < NTH; ++ta) {
      const bf16x8* wrow[FC];
#pragma unroll
      for (int g = 0; g < FC; ++g) wrow[g] = wa[g] + ((size_t)ch * NTAP + ta * NTW) * wstep;
      const int rowoff = (MODE == X_FWD) ? ta * ROWU : (NTMAX - 1 - ta) * ROWU;
      // Pinned software pipeline (hipcc otherwise sinks the prefetch loads next to their first use, one
      // vmcnt wait per MFMA pair): at the top of a tap issue the NEXT tap's filter fragments (global) and
      // pixel fragments (LDS, within the row), then run this tap's MFMAs on registers loaded a tap ago.
      constexpr bool BPF = (NP == 2);      // pixel fragments one tap ahead too (3 planes: no registers left for it)
      constexpr bool PIN = (NP == 2) || C::WC == 4;   // 3 planes, 2 x 2 fragments: pinning makes the allocator spill
      bf16x8 bv[BPF ? 2 : 1][FP][NP];
      auto read_b = [&](int buf, int t) {
        const int imm = (MODE == X_TR) ? (NTMAX - 1 - t) : (C::SPLIT ? (t & 1) * COLS + (t >> 1) : t);
#pragma unroll
        for (int f = 0; f < FP; ++f)
#pragma unroll
          for (int p = 0; p < NP; ++p) bv[buf][f][p] = __builtin_bit_cast(bf16x8, lds[base_b[f] + rowoff + imm + p * 2 * IMGU]);
      };
      if (BPF) read_b(0, 0);
#pragma unroll
      for (int tb = 0; tb < NTW; ++tb) {
        const int cur = tb & 1, nxt = cur ^ 1;
        // next tap's filter fragments (the pack has one spare step after the last one)
#pragma unroll
        for (int g = 0; g < FC; ++g)
#pragma unroll
          for (int p = 0; p < NP; ++p) av[nxt][g][p] = wrow[g][(size_t)(tb + 1) * wstep + (size_t)p * 2 * CoutP];
        if (BPF) {
          if (tb + 1 < NTW) read_b(nxt, tb + 1);
        } else {
          read_b(0, tb);
        }
        if (PIN) __builtin_amdgcn_sched_barrier(0);
        // products with plane index sum <= NP - 1, smallest terms first; product-major so that independent
        // accumulators sit between dependent MFMAs
#pragma unroll
        for (int sum = NP - 1; sum >= 0; --sum)
#pragma unroll
          for (int pa = sum; pa >= 0; --pa)
#pragma unroll
            for (int g = 0; g < FC; ++g)
#pragma unroll
              for (int f = 0; f < FP; ++f) acc[g][f] = mfma_split16<F16>(av[cur][g][pa], bv[BPF ? cur : 0][f][sum - pa], acc[g][f]);
        if (PIN) __builtin_amdgcn_sched_barrier(0);
      }
      if (NTW & 1) {
#pragma unroll
        for (int g = 0; g < FC; ++g)
#pragma unroll
          for (int p = 0; p < NP; ++p) av[0][g][p] = av[1][g][p];
      }
    }
    __syncthreads();
    if (more) {
      store_chunk();
      __syncthreads();
    }
  }

  if constexpr (F16) {   // undo the two operands' power-of-two scales (two exact multiplications)
    const float w_unscale = *A.w_unscale;
#pragma unroll
    for (int g = 0; g < FC; ++g)
#pragma unroll
      for (int f = 0; f < FP; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][f][r] = acc[g][f][r] * x_unscale * w_unscale;
  }
  // ---- epilogue: + bias, NCHW store (as conv_igemm.hip)
  const int YH = A.YH, YW = A.YW;
#pragma unroll
  for (int f = 0; f < FP; ++f) {
    const int m = (wp * FP + f) * 32 + l32;
    const int nb = m / (TH * TW), r = m % (TH * TW);
    const int th = th0 + r / TW, tw = tw0 + r % TW, b = b0 + nb;
    const int oh = (MODE == X_FWD) ? th : S * th + R;
    const int ow = (MODE == X_FWD) ? tw : S * tw + SS;
    const bool pok = b < A.B && oh < YH && ow < YW;
    float* yb = A.y + (size_t)split * A.ysplit + ((size_t)b * Cout * YH + oh) * YW + ow;
#pragma unroll
    for (int g = 0; g < FC; ++g) {
      float bv[16];
#pragma unroll
      for (int r16 = 0; r16 < 16; ++r16) {
        const int co = min(n0 + (wc * FC + g) * 32 + acc_row(r16, lane), Cout - 1);
        bv[r16] = (A.bias && split == 0) ? A.bias[co] : 0.f;     // partial slabs: the bias goes in once
      }
#pragma unroll
      for (int r16 = 0; r16 < 16; ++r16) {
        const int co = n0 + (wc * FC + g) * 32 + acc_row(r16, lane);
        if (pok && co < Cout) yb[(size_t)co * YH * YW] = acc[g][f][r16] + bv[r16];
      }
    }
  }
}

template <class C>
__global__ __launch_bounds__(XNT, 2) void conv5x5_bf16split_kernel(XArgs A) {
  __shared__ f32x4 lds[2 * C::NP * C::IMGU];     // [plane][k-block][image][row][(parity)][column]
  int bid = blockIdx.x;
  if constexpr (C::NCLS == 1) {
    const int split = bid / A.blocks_per_cls;
    bf16split_body<C, 0, 0>(A, lds, bid - split * A.blocks_per_cls, split);
  } else {
    const int cls = bid / A.blocks_per_cls;   // class 0 (3x3 taps) first: longest blocks start earliest
    bid -= cls * A.blocks_per_cls;
    switch (cls) {
      case 0: bf16split_body<C, 0, 0>(A, lds, bid, 0); break;
      case 1: bf16split_body<C, 0, 1>(A, lds, bid, 0); break;
      case 2: bf16split_body<C, 1, 0>(A, lds, bid, 0); break;
      default: bf16split_body<C, 1, 1>(A, lds, bid, 0); break;
    }
  }
}

struct XSplit {
  int k;          // 1: no split
  float* slabs;   // k partial outputs
  const float* in_amax;   // fp16 planes only
};

template <class C>
int launch_x(const float* x, const bf16x8* w, const float* bias, float* y, int B, int Cin, int XH, int XW, int Cout,
             XSplit xs, hipStream_t st) {
  XArgs A;
  A.x = x; A.w = w; A.bias = bias; A.y = y;
  A.B = B; A.Cin = Cin; A.XH = XH; A.XW = XW; A.Cout = Cout; A.CoutP = (Cout + 127) & ~127;
  int tsh, tsw;
  if (C::MODE == X_FWD) {
    A.YH = (XH - 1) / C::S + 1; A.YW = (XW - 1) / C::S + 1;
    tsh = A.YH; tsw = A.YW;
  } else {
    A.YH = XH * C::S; A.YW = XW * C::S;
    tsh = XH; tsw = XW;
  }
  A.tiles_w = cdiv(tsw, C::TW);
  A.tiles_hw = cdiv(tsh, C::TH) * A.tiles_w;
  A.ntiles_n = cdiv(Cout, C::TN);
  const long per_cls = (long)A.ntiles_n * A.tiles_hw * cdiv(B, C::NB);
  const int ksplit = (C::MODE == X_FWD) ? xs.k : 1;
  const long grid = per_cls * C::NCLS * ksplit;
  if (grid <= 0 || grid > 0x7fffffffL) return VG_ERR_BAD_ARG;
  A.blocks_per_cls = (int)per_cls;
  A.ksplit = ksplit;
  A.cps = cdiv(Cin / 16, ksplit);
  A.ysplit = (size_t)B * Cout * A.YH * A.YW;
  if (ksplit > 1) A.y = xs.slabs;
  A.in_amax = xs.in_amax;
  A.w_unscale = (const float*)(w + (size_t)(Cin / 16 * 25 + VG_PACK_SPARE) * 2 * C::NP * A.CoutP);      // the pack's trailer
  hipLaunchKernelGGL(conv5x5_bf16split_kernel<C>, dim3((unsigned)grid), dim3(XNT), 0, st, A);
  VG_CHECK_LAUNCH();
  if (ksplit > 1) {
    if (A.ysplit > 0x7fffffffUL) return VG_ERR_BAD_ARG;
    return vg_internal_wgrad_reduce(xs.slabs, y, (int)A.ysplit, ksplit, st);   // fixed-order sum of the slabs
  }
  return 0;
}

VG_KNOB(int, g_x_tile_override, -1);   // diagnostics: 0 = 128 cout x 128 px, 1 = 64 x 128, 2 = 64 x 64, 3 = 32 cout x 128 px, 4 = 32 x 256 (transposed)

template <int MODE, int S, int WC, int FC, int FP, int NP, bool F16>
int dispatch_geom(const float* x, const bf16x8* w, const float* bias, float* y, int B, int Cin, int XH, int XW,
                  int Cout, XSplit xs, hipStream_t st) {
  const int tsw = (MODE == X_FWD) ? (XW - 1) / S + 1 : XW;
  constexpr int TM = 32 * (4 / WC) * FP;
  if constexpr (TM == 128) {
    if (tsw >= 32) return launch_x<XCfg<MODE, S, 1, 4, 32, WC, FC, FP, NP, F16>>(x, w, bias, y, B, Cin, XH, XW, Cout, xs, st);
    if (tsw >= 16) return launch_x<XCfg<MODE, S, 1, 8, 16, WC, FC, FP, NP, F16>>(x, w, bias, y, B, Cin, XH, XW, Cout, xs, st);
    return launch_x<XCfg<MODE, S, 2, 8, 8, WC, FC, FP, NP, F16>>(x, w, bias, y, B, Cin, XH, XW, Cout, xs, st);
  } else if constexpr (TM == 256) {
    if (tsw >= 32) return launch_x<XCfg<MODE, S, 1, 8, 32, WC, FC, FP, NP, F16>>(x, w, bias, y, B, Cin, XH, XW, Cout, xs, st);
    if (tsw >= 16) return launch_x<XCfg<MODE, S, 1, 16, 16, WC, FC, FP, NP, F16>>(x, w, bias, y, B, Cin, XH, XW, Cout, xs, st);
    return launch_x<XCfg<MODE, S, 4, 8, 8, WC, FC, FP, NP, F16>>(x, w, bias, y, B, Cin, XH, XW, Cout, xs, st);
  } else {
    if (tsw >= 32) return launch_x<XCfg<MODE, S, 1, 2, 32, WC, FC, FP, NP, F16>>(x, w, bias, y, B, Cin, XH, XW, Cout, xs, st);
    if (tsw >= 16) return launch_x<XCfg<MODE, S, 1, 4, 16, WC, FC, FP, NP, F16>>(x, w, bias, y, B, Cin, XH, XW, Cout, xs, st);
    return launch_x<XCfg<MODE, S, 1, 8, 8, WC, FC, FP, NP, F16>>(x, w, bias, y, B, Cin, XH, XW, Cout, xs, st);
  }
}

// Biggest tile that still gives every CU two workgroups (256 CUs): 128 cout x 128 px, else 64 cout x
// 128 px, else 64 x 64; 32 cout x 128 px (4 wavefronts along the pixels) for thin outputs.
template <int MODE, int S, int NP, bool F16 = false>
int dispatch_x(const float* x, const bf16x8* w, const float* bias, float* y, int B, int Cin, int XH, int XW, int Cout,
               XSplit xs, hipStream_t st) {
  const int tsw = (MODE == X_FWD) ? (XW - 1) / S + 1 : XW, tsh = (MODE == X_FWD) ? (XH - 1) / S + 1 : XH;
  const int ncls = (MODE == X_TR) ? S * S : 1;
  const long px128 = (long)cdiv(B * cdiv(tsh, 8) * cdiv(tsw, 8) * 64, 128) * ncls;
  int var = 2;
  if (Cout <= 32) var = (MODE == X_TR && px128 >= 1024) ? 4 : 3;      // transposed, large grid: 256 pixels per workgroup
  else if (MODE == X_FWD && Cout > 64 && px128 * cdiv(Cout, 128) >= 512) var = 0;   // transposed: 64 x 128 measured faster
  else if (px128 * cdiv(Cout, 64) >= 512) var = 1;
  // 3 planes: the 128 x 128 tile with the four wavefronts along cout (each loads only its own filter fragment)
  // measured 0-15 % faster than the 2 x 2 arrangement, never slower
  if (NP == 3 && (var == 0 || var == 1) && Cout >= 128 && px128 * cdiv(Cout, 128) >= 256) var = 5;
  if (MODE == X_FWD && xs.k > 1) var = 0;             // split-K is sized for the 128 x 128 tile
  else if (g_x_tile_override >= 0 && g_x_tile_override <= 5 && !(g_x_tile_override == 4 && MODE == X_FWD))
    var = g_x_tile_override;
  if (var == 0) return dispatch_geom<MODE, S, 2, 2, 2, NP, F16>(x, w, bias, y, B, Cin, XH, XW, Cout, xs, st);
  if (var == 1) return dispatch_geom<MODE, S, 2, 1, 2, NP, F16>(x, w, bias, y, B, Cin, XH, XW, Cout, xs, st);
  if (var == 3) return dispatch_geom<MODE, S, 1, 1, 1, NP, F16>(x, w, bias, y, B, Cin, XH, XW, Cout, xs, st);
  if (var == 5) return dispatch_geom<MODE, S, 4, 1, 4, NP, F16>(x, w, bias, y, B, Cin, XH, XW, Cout, xs, st);   // 4 wavefronts along cout
  if constexpr (MODE == X_TR) {
    if (var == 4) return dispatch_geom<MODE, S, 1, 1, 2, NP, F16>(x, w, bias, y, B, Cin, XH, XW, Cout, xs, st);
  }
  return dispatch_geom<MODE, S, 2, 1, 1, NP, F16>(x, w, bias, y, B, Cin, XH, XW, Cout, xs, st);
}

// packed[class][chunk][step][plane][k-block][CoutP] x 8 bf16 (+ VG_PACK_SPARE zero steps at the end: the ring
// kernel's DMA runs three steps ahead of the MFMAs).
//   transposed = 0, S = 1: w is [Cout][Cin][5][5], one class of 25 taps (kh*5 + kw); a step is one tap, its two
//                   k-blocks are channels 0-7 / 8-15 of the chunk;
//   transposed = 0, S = 2 (conv_ring.hip): a step is 8 channels x 2 consecutive taps -- steps 0-11 taps (2s, 2s+1)
//                   of channels 0-7, step 12 tap 24 of channels 0-7 | 8-15, steps 13-24 taps of channels 8-15;
//   transposed = 1: w is [Cin][Cout][5][5], S*S parity classes, tap (a, b) of class (R, SS) is
//                   (kh, kw) = (R + S*a, SS + S*b).
constexpr int PK_CO = 8, PK_PITCH = 401;         // channels per workgroup (512 workgroups for a 256 x 256 filter: with 32
                                                 // channels and 128 workgroups it ran 21 us); floats per channel in LDS (400 + 1)
// One workgroup = PK_CO output channels x one 16-channel chunk: the PK_CO x 16 x 25 filter values are read with contiguous
// loads into LDS (400 contiguous floats per channel, or 800 per input channel for the transposed layout) and every
// (channel, step, k-block) unit is built from there.  (The first version read each value with its own 4-byte load, a
// lane's 16 values 100 bytes apart and the lanes 12.8 KB apart: 12.7 us for the 6.5 MB filter, 0.29 ms per iteration.)
// planes & VG_PLANES_F16_FLAG: fp16 planes -- the filter is multiplied by f16_scale_of(w_amax[0]) first and the inverse of
// that power of two is left in the pack's trailer (one float behind the spare steps) for the convolution's epilogue.
__device__ __forceinline__ void pack_bf16split_body(float* T, const float* __restrict__ w, bf16x8* __restrict__ p,
                                                    int Cout, int Cin, int CoutP, int nsteps, int transposed, int S,
                                                    int planes, const float* __restrict__ w_amax, int bx, int by) {
  const int tid = threadIdx.x;
  const bool f16 = (planes & VG_PLANES_F16_FLAG) != 0;
  planes &= 0xff;
  const float wscale = f16 ? f16_scale_of(*w_amax) : 1.f;
  if (f16 && bx == 0 && by == 0 && tid == 0)
    *(float*)(p + (size_t)(nsteps + VG_PACK_SPARE) * planes * 2 * CoutP) = f16_unscale_of(*w_amax);
  const int co0 = bx * PK_CO, c16 = by, nchunks = Cin / 16;
  if (!transposed) {
    for (int e0 = tid; e0 < PK_CO * 400; e0 += 4 * 256) {
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = min(e0 + k * 256, PK_CO * 400 - 1), col = e / 400, r = e - col * 400;
        const float t = w[((size_t)min(co0 + col, Cout - 1) * Cin + c16 * 16) * 25 + r];      // read, then select
        v[k] = (co0 + col < Cout) ? t : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = e0 + k * 256, col = e / 400, r = e - col * 400;
        if (e < PK_CO * 400) T[col * PK_PITCH + r] = v[k];
      }
    }
  } else {
    for (int e0 = tid; e0 < 16 * PK_CO * 25; e0 += 4 * 256) {
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = min(e0 + k * 256, 16 * PK_CO * 25 - 1), ci = e / (PK_CO * 25), r = e - ci * (PK_CO * 25);
        const int col = r / 25;
        const float t = w[((size_t)(c16 * 16 + ci) * Cout + min(co0 + col, Cout - 1)) * 25 + (r - col * 25)];
        v[k] = (co0 + col < Cout) ? t : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = e0 + k * 256, ci = e / (PK_CO * 25), r = e - ci * (PK_CO * 25), col = r / 25;
        if (e < 16 * PK_CO * 25) T[col * PK_PITCH + ci * 25 + (r - col * 25)] = v[k];
      }
    }
  }
  __syncthreads();
  // units: (local step q, k-block kb, channel): lanes = consecutive channels (contiguous 16-byte stores)
  const int col = tid & (PK_CO - 1);
  const bool cok = co0 + col < CoutP;
  for (int it = tid / PK_CO; it < 50; it += 256 / PK_CO) {
    const int q = it >> 1, kb = it & 1;
    int half = kb, tap = q, step = c16 * 25 + q;
    if (!transposed && S == 2) {                 // ring layout: half-chunk pairing (two taps per step)
      half = (q < 12) ? 0 : (q == 12 ? kb : 1);
      tap = (q < 12) ? 2 * q + kb : (q == 12 ? 24 : 2 * (q - 13) + kb);
    } else if (transposed) {                     // parity classes, class-major step order
      const int kh = q / 5, kw = q - 5 * kh;
      const int R = kh % S, SS = kw % S;
      int base = 0;
      for (int r2 = 0; r2 < S; ++r2)
        for (int s2 = 0; s2 < S; ++s2)
          if (r2 < R || (r2 == R && s2 < SS)) base += ((5 - r2 + S - 1) / S) * ((5 - s2 + S - 1) / S) * nchunks;
      const int nth = (5 - R + S - 1) / S, ntw = (5 - SS + S - 1) / S;
      step = base + c16 * (nth * ntw) + (kh / S) * ntw + kw / S;
    }
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = T[col * PK_PITCH + (half * 8 + j) * 25 + tap] * wscale;
    if (f16) {
      f32x4 q2[2];
      split_planes16<2, true>(v, q2);
      if (cok) {
        p[((size_t)step * 4 + kb) * CoutP + co0 + col] = __builtin_bit_cast(bf16x8, q2[0]);
        p[((size_t)step * 4 + 2 + kb) * CoutP + co0 + col] = __builtin_bit_cast(bf16x8, q2[1]);
      }
      continue;
    }
    for (int pl = 0; pl < planes; ++pl) {       // hi, (mid,) lo
      bf16x8 qv;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)v[j];
        qv[j] = h;
        v[j] -= (float)h;
      }
      if (cok) p[((size_t)step * planes * 2 + pl * 2 + kb) * CoutP + co0 + col] = qv;
    }
  }
  // the spare steps (the DMA ring's run-ahead reads them): zeros, written by the first chunk's workgroups
  if (c16 == 0 && cok) {
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.f;
    for (int u = tid / PK_CO; u < VG_PACK_SPARE * planes * 2; u += 256 / PK_CO) p[((size_t)nsteps * planes * 2 + u) * CoutP + co0 + col] = z;
  }
}

__global__ __launch_bounds__(256) void pack_bf16split_kernel(const float* __restrict__ w, bf16x8* __restrict__ p,
                                                         int Cout, int Cin, int CoutP, int nsteps, int transposed,
                                                         int S, int planes, const float* __restrict__ w_amax) {
  __shared__ float T[PK_CO * PK_PITCH];          // [channel][ci16 * 25 + tap]
  pack_bf16split_body(T, w, p, Cout, Cin, CoutP, nsteps, transposed, S, planes, w_amax, blockIdx.x, blockIdx.y);
}

// Several filters in ONE launch (every filter an optimizer step has just changed): a pack is latency-bound (8 us for
// 0.2-6.5 MB), so 6-16 of them side by side take about as long as one.
constexpr int PK_MAXE = 24;
struct PackBatch {
  const float* w[PK_MAXE];
  const float* w_amax[PK_MAXE];
  bf16x8* p[PK_MAXE];
  int Cout[PK_MAXE], Cin[PK_MAXE], transposed[PK_MAXE], S[PK_MAXE];
  unsigned first_block[PK_MAXE + 1];
  int count;
};

__global__ __launch_bounds__(256) void pack_bf16split_multi_kernel(PackBatch P, int planes) {
  __shared__ float T[PK_CO * PK_PITCH];
  int t = 0;
  while (t + 1 < P.count && blockIdx.x >= P.first_block[t + 1]) ++t;
  const int CoutP = (P.Cout[t] + 127) & ~127, nbx = CoutP / PK_CO;
  const int b = blockIdx.x - P.first_block[t];
  pack_bf16split_body(T, P.w[t], P.p[t], P.Cout[t], P.Cin[t], CoutP, P.Cin[t] / 16 * 25, P.transposed[t], P.S[t], planes,
                      P.w_amax[t], b % nbx, b / nbx);
}

// Forward, deep K on a small grid (the 8 x 8-pixel layers): the 128 x 128 tile with the channel chunks split
// over k workgroups, partial outputs summed in a fixed order.  1 = no split.
int fwd_ksplit(int B, int Cin, int H, int W, int Cout, int S) {
  if (g_x_tile_override >= 0) return 1;
  const int tsw = (W - 1) / S + 1, tsh = (H - 1) / S + 1, nchunks = Cin / 16;
  const long wgs = (long)cdiv(B * cdiv(tsh, 8) * cdiv(tsw, 8) * 64, 128) * cdiv(Cout, 128);
  if (Cout <= 64 || wgs >= 512 || nchunks < 8) return 1;
  int k = 2;     // every split must own at least one chunk: (k - 1) * ceil(nchunks / k) < nchunks
  while (wgs * k < 512 && k < 8 && nchunks / (2 * k) >= 2 && (2 * k - 1) * cdiv(nchunks, 2 * k) < nchunks) k *= 2;
  return k;
}

// 2 / 3 bf16 planes, or 2 fp16 planes (VG_PLANES_F16 | 2)
int planes_ok(int planes) { return planes == 2 || planes == 3 || planes == (VG_PLANES_F16_FLAG | 2); }

int x_args_ok(const float* x, const void* packed, float* y, int B, int Cin, int H, int W, int Cout, int stride,
              int planes) {
  if (!x || !packed || !y || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
  return (stride == 1 || stride == 2) && Cin % 16 == 0 && ((uintptr_t)packed & 15) == 0 && planes_ok(planes);
}

}  // namespace

#ifdef VG_TUNING
extern "C" int vg_debug_set_conv_bf16split_tile(int variant) {
  g_x_tile_override = variant;
  return 0;
}

extern "C" int vg_debug_set_conv_ring_tile(int variant) {
  vg_internal_ring_set_variant(variant);
  return 0;
}
#endif

extern "C" size_t vg_conv5x5_packed_bf16split_bytes(int Cout, int Cin, int planes) {
  if (Cout <= 0 || Cin <= 0 || Cin % 16 || !planes_ok(planes)) return 0;
  // fp16 planes: + a 16-byte trailer holding the inverse of the filter's power-of-two scale
  return (size_t)(Cin / 16 * 25 + VG_PACK_SPARE) * 2 * (planes & 0xff) * ((Cout + 127) & ~127) * 16 +
         ((planes & VG_PLANES_F16_FLAG) ? 16 : 0);
}

extern "C" int vg_conv5x5_pack_bf16split(const float* w, void* packed, int Cout, int Cin, int transposed, int stride,
                                      int planes, const float* w_amax, void* stream) {
  if (!w || !packed || Cout <= 0 || Cin <= 0 || Cin % 16 || ((uintptr_t)packed & 15)) return VG_ERR_BAD_ARG;
  if ((stride != 1 && stride != 2) || !planes_ok(planes)) return VG_ERR_BAD_ARG;
  if ((planes & VG_PLANES_F16_FLAG) && !w_amax) return VG_ERR_BAD_ARG;
  const int CoutP = (Cout + 127) & ~127, nsteps = Cin / 16 * 25;
  hipLaunchKernelGGL(pack_bf16split_kernel, dim3(CoutP / PK_CO, Cin / 16), dim3(256), 0,
                     (hipStream_t)stream, w, (bf16x8*)packed, Cout, Cin, CoutP, nsteps, transposed ? 1 : 0, stride, planes,
                     w_amax);
  VG_CHECK_LAUNCH();
  return 0;
}

extern "C" int vg_conv5x5_pack_bf16split_multi(const VgPackEntry* entries, int count, int planes, void* stream) {
  if (count < 0 || (count > 0 && !entries) || !planes_ok(planes)) return VG_ERR_BAD_ARG;
  int i = 0;
  while (i < count) {
    PackBatch P;
    P.count = 0;
    unsigned blocks = 0;
    while (i < count && P.count < PK_MAXE) {
      const VgPackEntry& E = entries[i++];
      if (!E.w || !E.packed || E.Cout <= 0 || E.Cin <= 0 || E.Cin % 16 || ((uintptr_t)E.packed & 15) ||
          (E.stride != 1 && E.stride != 2) || ((planes & VG_PLANES_F16_FLAG) && !E.w_amax))
        return VG_ERR_BAD_ARG;
      const int k = P.count++;
      P.w_amax[k] = E.w_amax;
      P.w[k] = E.w; P.p[k] = (bf16x8*)E.packed; P.Cout[k] = E.Cout; P.Cin[k] = E.Cin;
      P.transposed[k] = E.transposed ? 1 : 0; P.S[k] = E.stride;
      P.first_block[k] = blocks;
      blocks += (unsigned)(((E.Cout + 127) & ~127) / PK_CO) * (unsigned)(E.Cin / 16);
    }
    P.first_block[P.count] = blocks;
    hipLaunchKernelGGL(pack_bf16split_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P, planes);
    VG_CHECK_LAUNCH();
  }
  return 0;
}

// Stride 2 runs on the 8-wave ring kernel (conv_ring.hip) -- except transposed convolutions with <= 64 output
// channels, whose thin tiles (32 cout x 128 / 256 pixels) live here; stride 1 runs here.
static bool tr_on_ring(int Cout, int stride) { return stride == 2 && Cout > 64; }

extern "C" size_t vg_conv5x5_fwd_bf16split_workspace_bytes(int B, int Cin, int H, int W, int Cout, int stride, int planes) {
  if (B <= 0 || Cin <= 0 || Cin % 16 || Cout <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2) || !planes_ok(planes)) return 0;
  if (stride == 2) return vg_internal_ring_workspace_bytes(0, B, Cin, H, W, Cout, planes);
  const int k = fwd_ksplit(B, Cin, H, W, Cout, stride);
  if (k <= 1) return 0;
  return (size_t)k * B * Cout * ((H - 1) / stride + 1) * ((W - 1) / stride + 1) * sizeof(float);
}

static bool fuse_empty(const vg_conv_fusion* f) { return !f || (!f->in_scale && !f->in_shift && !f->stats); }   // in_amax is not a fusion

extern "C" int vg_conv5x5_bf16split_fusable(int transposed, int Cin, int Cout, int stride) {
  if (Cin <= 0 || Cin % 16 || Cout <= 0) return 0;
  return transposed ? (tr_on_ring(Cout, stride) ? 1 : 0) : (stride == 2 ? 1 : 0);
}

extern "C" size_t vg_conv5x5_fwd_bf16split_stats_floats(int B, int Cin, int H, int W, int Cout, int stride, int planes) {
  if (B <= 0 || Cin <= 0 || Cin % 16 || Cout <= 0 || H <= 0 || W <= 0 || stride != 2 || !planes_ok(planes)) return 0;
  return vg_internal_ring_stats_floats(0, B, Cin, H, W, Cout, planes);
}

extern "C" size_t vg_convT5x5_fwd_bf16split_stats_floats(int B, int Cin, int H, int W, int Cout, int stride, int planes) {
  if (B <= 0 || Cin <= 0 || Cin % 16 || Cout <= 0 || H <= 0 || W <= 0 || !tr_on_ring(Cout, stride) || !planes_ok(planes)) return 0;
  return vg_internal_ring_stats_floats(1, B, Cin, H, W, Cout, planes);
}

extern "C" int vg_conv5x5_fwd_bf16split(const float* x, const void* packed, const float* bias, float* y, int B, int Cin,
                                     int H, int W, int Cout, int stride, int planes, void* workspace,
                                     size_t workspace_bytes, const vg_conv_fusion* fuse, void* stream) {
  if (!x_args_ok(x, packed, y, B, Cin, H, W, Cout, stride, planes)) return VG_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (stride == 2)
    return vg_internal_ring_conv(0, x, packed, bias, y, B, Cin, H, W, Cout, planes, workspace, workspace_bytes,
                                 fuse ? fuse->in_scale : nullptr, fuse ? fuse->in_shift : nullptr, fuse ? fuse->in_act : 0,
                                 fuse ? fuse->stats : nullptr, fuse ? fuse->stats_floats : 0, fuse ? fuse->in_amax : nullptr, st);
  if (!fuse_empty(fuse)) return VG_ERR_BAD_ARG;        // vg_conv5x5_bf16split_fusable says which layers take it
  const bf16x8* w = (const bf16x8*)packed;
  const int k = fwd_ksplit(B, Cin, H, W, Cout, stride);
  if (k > 1 && (!workspace || workspace_bytes < vg_conv5x5_fwd_bf16split_workspace_bytes(B, Cin, H, W, Cout, stride, planes)))
    return VG_ERR_WORKSPACE;
  const XSplit xs = {k, (float*)workspace, fuse ? fuse->in_amax : nullptr};
  if (planes & VG_PLANES_F16_FLAG) {
    if (!xs.in_amax) return VG_ERR_BAD_ARG;
    return dispatch_x<X_FWD, 1, 2, true>(x, w, bias, y, B, Cin, H, W, Cout, xs, st);
  }
  if (planes == 2) return dispatch_x<X_FWD, 1, 2>(x, w, bias, y, B, Cin, H, W, Cout, xs, st);
  return dispatch_x<X_FWD, 1, 3>(x, w, bias, y, B, Cin, H, W, Cout, xs, st);
}

extern "C" size_t vg_convT5x5_fwd_bf16split_workspace_bytes(int B, int Cin, int H, int W, int Cout, int stride, int planes) {
  if (B <= 0 || Cin <= 0 || Cin % 16 || Cout <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2) || !planes_ok(planes)) return 0;
  return tr_on_ring(Cout, stride) ? vg_internal_ring_workspace_bytes(1, B, Cin, H, W, Cout, planes) : 0;
}

extern "C" int vg_convT5x5_fwd_bf16split(const float* x, const void* packed, const float* bias, float* y, int B, int Cin,
                                      int H, int W, int Cout, int stride, int planes, void* workspace,
                                      size_t workspace_bytes, const vg_conv_fusion* fuse, void* stream) {
  if (!x_args_ok(x, packed, y, B, Cin, H, W, Cout, stride, planes)) return VG_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (tr_on_ring(Cout, stride))
    return vg_internal_ring_conv(1, x, packed, bias, y, B, Cin, H, W, Cout, planes, workspace, workspace_bytes,
                                 fuse ? fuse->in_scale : nullptr, fuse ? fuse->in_shift : nullptr, fuse ? fuse->in_act : 0,
                                 fuse ? fuse->stats : nullptr, fuse ? fuse->stats_floats : 0, fuse ? fuse->in_amax : nullptr, st);
  if (!fuse_empty(fuse)) return VG_ERR_BAD_ARG;
  const bf16x8* w = (const bf16x8*)packed;
  const XSplit xs = {1, nullptr, fuse ? fuse->in_amax : nullptr};
  if (planes & VG_PLANES_F16_FLAG) {
    if (!xs.in_amax) return VG_ERR_BAD_ARG;
    if (stride == 2) return dispatch_x<X_TR, 2, 2, true>(x, w, bias, y, B, Cin, H, W, Cout, xs, st);
    return dispatch_x<X_TR, 1, 2, true>(x, w, bias, y, B, Cin, H, W, Cout, xs, st);
  }
  if (planes == 2) {
    if (stride == 2) return dispatch_x<X_TR, 2, 2>(x, w, bias, y, B, Cin, H, W, Cout, xs, st);
    return dispatch_x<X_TR, 1, 2>(x, w, bias, y, B, Cin, H, W, Cout, xs, st);
  }
  if (stride == 2) return dispatch_x<X_TR, 2, 3>(x, w, bias, y, B, Cin, H, W, Cout, xs, st);
  return dispatch_x<X_TR, 1, 3>(x, w, bias, y, B, Cin, H, W, Cout, xs, st);
}
