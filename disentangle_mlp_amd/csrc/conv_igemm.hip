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
// Work decomposition (one workgroup = 4 or 8 wavefronts of 64 lanes; tile variants below):
//   * output tile = TN output channels ("rows" of D) x TM = NB*TH*TW pixels
//     ("columns"): D[cout][pixel], so the 32 lanes of an MFMA column group store
//     32 consecutive pixels of an NCHW row.
//   * the K loop runs over input-channel chunks of CK channels.  Per chunk the
//     workgroup stages into LDS (double buffered, next chunk prefetched into
//     registers under the MFMAs): the NCHW input patch [NB][CK][PH][PW] with
//     coalesced row loads and zero-filled halo, and the filter slab [TN][CK*NTAP].
//     Staging loads are never predicated (addresses clamped into the tensor, halo
//     zeroed at the LDS store): predicated loads made hipcc merge the dummy loads
//     and drain vmcnt in the middle of the prefetch.
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


template <int MODE_, int S_, int NB_, int TH_, int TW_, int TN_, int WC_, int WP_, int CK_, int MINW_ = 2, int KS_ = 1,
          bool PK_ = false>
struct Cfg {
  static constexpr int MODE = MODE_, S = S_, NB = NB_, TH = TH_, TW = TW_, TN = TN_, WC = WC_, WP = WP_,
                       CK = CK_, MINW = MINW_, KS = KS_;  // KS: in-workgroup split of the K chunk
  static constexpr bool PK = PK_;  // filter comes pre-packed [class][ci][tap][cout] (vg_conv5x5_pack)
  // packed slab copied global -> LDS by DMA (no VGPR round trip): measured faster on the 8-wave
  // tiles without K split, slower on the 4-wave ones (scripts/tune_conv.py)
  static constexpr bool DMA = PK_ && WC_ * WP_ == 8 && KS_ == 1;
  static constexpr int NT = 64 * WC * WP * KS;
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
  static constexpr int WOFF = (NP + 3) & ~3;         // filter slab starts 16-byte aligned
  static constexpr int STAGE = (WOFF + TN * WSMAX + 1 + 3) & ~3;  // +1: dummy slot for masked-off stores
  static constexpr int NCLS = (MODE == MODE_FWD) ? 1 : S * S;
  static_assert(TN % (32 * WC) == 0 && TM % (32 * WP) == 0 && CK % 2 == 0, "tile shape");
  static_assert(RLMAX <= NT, "one filter row per pass at least");
  static_assert(KS == 1 || (CK / 2) % KS == 0, "split-K groups own whole channel pairs");
  static_assert(KS == 1 || 2 * STAGE >= (KS - 1) * TM * TN, "LDS must hold the partial tiles of the K groups");
};

struct Args {
  const float* x;
  const float* w;    // PK: the packed filter
  const float* bias;
  float* y;
  int B, Cin, XH, XW, Cout, YH, YW;
  int CinP, CoutP;   // PK: padded extents of the packed filter
  int ntiles_n, tiles_w, tiles_hw, blocks_per_cls;
  float* stats;      // optional [slot][Cout][2]: per-channel sum y, sum y^2 of each (pixel tile, wavefront row) --
                     // the statistics of the BatchNorm that follows (forward convolution only; vg_conv_fusion.stats)
};

// Padded extents of the packed filter: whole K chunks (<= 8 channels) and whole cout tiles.
__host__ __device__ constexpr int packed_cin(int Cin) { return (Cin + 7) & ~7; }
__host__ __device__ constexpr int packed_cout(int Cout) { return (Cout + 127) & ~127; }

// Taps of the parity classes that precede class (R, SS) in the packed filter.
__host__ __device__ constexpr int taps_before(int S, int R, int SS) {
  int n = 0;
  for (int r = 0; r < S; ++r)
    for (int s = 0; s < S; ++s) {
      if (r == R && s == SS) return n;
      n += ((5 - r + S - 1) / S) * ((5 - s + S - 1) / S);
    }
  return n;
}

// R, SS: output parity class of the transposed convolution (0,0 for MODE_FWD).
template <class C, int R, int SS>
__device__ __forceinline__ void igemm_body(const Args& A, float* smem, int bid) {
  constexpr int MODE = C::MODE, S = C::S, NB = C::NB, TH = C::TH, TW = C::TW, TN = C::TN, CK = C::CK;
  constexpr int NT = C::NT, FC = C::FC, FP = C::FP, PH = C::PH, PW = C::PW, PWP = C::PWP;
  constexpr int NP = C::NP, NQP = C::NQP, NTMAX = C::NTMAX, WOFF = C::WOFF;
  constexpr bool PK = C::PK;
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
  const int kg = wid / (C::WC * C::WP), w2 = wid % (C::WC * C::WP);   // K group, wave within the tile
  const int wc = w2 % C::WC, wp = w2 / C::WC;
  // XCD-aware placement: workgroups are dealt round-robin over the 8 XCDs (bid % 8 labels the
  // XCD, each with its own L2).  The ntiles_n cout tiles that read the SAME input patch are
  // given to ONE XCD, back to back, so the patch is fetched from HBM once and re-read from that
  // L2; every XCD streams the whole (L2-sized) filter.  Placement affects speed only.
  int nt, pt;
  {
    const int ntn = A.ntiles_n;
    const int npatch = A.blocks_per_cls / ntn;
    const int full = (npatch / 8) * 8 * ntn;
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
  const int th0 = (sp / A.tiles_w) * TH, tw0 = (sp % A.tiles_w) * TW;  // tile origin (tile space)
  const int b0 = bg * NB, n0 = nt * TN;
  const int Cin = A.Cin, Cout = A.Cout, XH = A.XH, XW = A.XW;
  const int HW = XH * XW;
  // patch origin in input coordinates
  const int ih0 = (MODE == MODE_FWD) ? th0 * S - 2 : th0 - (NTMAX - 1 - 2 / S);
  const int iw0 = (MODE == MODE_FWD) ? tw0 * S - 2 : tw0 - (NTMAX - 1 - 2 / S);
  const float* xb = A.x + (size_t)b0 * Cin * HW;

  // ---- per-thread patch staging map (invariant over the K loop).  Every load address is
  // CLAMPED into the tensor (distinct, always valid) instead of being predicated: the prefetch
  // is straight-line code with no dependent or merged loads; halo / tail elements are replaced
  // by 0 when the registers are written to LDS.
  int pofs[NQP];
  unsigned pvalid = 0;   // bit q: spatially inside the image (and a real image)
#pragma unroll
  for (int q = 0; q < NQP; ++q) {
    const int e = tid + q * NT;
    const int col = e % PWP;
    int t = e / PWP;
    const int r = t % PH;
    t /= PH;
    const int nb = t / CK;
    const int ih = ih0 + r, iw = iw0 + col;
    const bool ok = (e < NP) && (col < PW) && ih >= 0 && ih < XH && iw >= 0 && iw < XW && (b0 + nb) < A.B;
    const int nbc = min(nb, A.B - 1 - b0), ihc = min(max(ih, 0), XH - 1), iwc = min(max(iw, 0), XW - 1);
    pofs[q] = nbc * Cin * HW + ihc * XW + iwc;
    pvalid |= ok ? (1u << q) : 0u;
  }
  static_assert(NQP <= 32, "patch validity mask is 32-bit");

  // ---- filter staging map: RL consecutive lanes stage one filter row
  const int w_sub = tid / RL, w_r = tid % RL;
  const bool wrow_ok = w_sub < RP;
  const int w_c = w_r / NTAP, w_t = w_r % NTAP;
  const int w_kh = (MODE == MODE_FWD) ? w_t / 5 : R + S * (w_t / NTW);
  const int w_kw = (MODE == MODE_FWD) ? w_t % 5 : SS + S * (w_t % NTW);
  // conv weight [Cout][Cin][25]; transposed-conv weight [Cin][Cout][25]
  const int w_row_stride = (MODE == MODE_FWD) ? Cin * 25 : 25;
  const int w_ch_stride = (MODE == MODE_FWD) ? 25 : Cout * 25;
  const int w_tap = w_kh * 5 + w_kw;

  // ---- packed filter (PK): the chunk's slab is RL rows x TN floats of the [ci*NTAP+tap][CoutP]
  // image, copied verbatim (16-byte loads / ds_write_b128, zero padding comes from the pack)
  constexpr int TN4 = TN / 4, RPQ = NT / TN4;      // 16-byte columns per row; rows per pass
  constexpr int NQ4 = PK ? cdiv(RL, RPQ) : 1;
  static_assert(NT % TN4 == 0, "packed staging: whole rows per pass");
  const int pk_c4 = tid % TN4, pk_k0 = tid / TN4;
  const float* wpk = A.w + ((size_t)taps_before(S, R, SS) * A.CinP) * A.CoutP + n0 + 4 * pk_c4;

  float preg[NQP], wreg[PK ? 1 : NQW];
  f32x4 wreg4[NQ4];
  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int q = 0; q < NQP; ++q) {
      const int e = tid + q * NT;
      const int c = (e / (PWP * PH)) % CK;
      preg[q] = xb[pofs[q] + min(c0 + c, Cin - 1) * HW];
    }
    if constexpr (C::DMA) {
      // the slab is a verbatim image: DMA it straight into the other LDS stage (wave = 1 KiB)
      const float* src = wpk + (size_t)(c0 * NTAP) * A.CoutP;
      float* dst = smem + (((c0 / CK) & 1) * C::STAGE) + WOFF + wid * 256;
#pragma unroll
      for (int q = 0; q < NQ4; ++q) {
        const int k = pk_k0 + q * RPQ;
        if (k < RL) __builtin_amdgcn_global_load_lds(src + (size_t)k * A.CoutP, dst + q * NT * 4, 16, 0, 0);
      }
    } else if constexpr (PK) {
      const float* src = wpk + (size_t)(c0 * NTAP) * A.CoutP;
#pragma unroll
      for (int q = 0; q < NQ4; ++q) {
        const int k = min(pk_k0 + q * RPQ, RL - 1);   // tail rows re-copy row RL-1
        wreg4[q] = *reinterpret_cast<const f32x4*>(src + (size_t)k * A.CoutP);
      }
    } else {
      const int wbase = min(c0 + w_c, Cin - 1) * w_ch_stride + w_tap;
#pragma unroll
      for (int q = 0; q < NQW; ++q) {
        const int row = min(n0 + w_sub + q * RP, Cout - 1);
        wreg[q] = A.w[row * w_row_stride + wbase];
      }
    }
  };
  auto store_chunk = [&](float* st, int c0) {
    constexpr int DUMMY = C::STAGE - 1;
#pragma unroll
    for (int q = 0; q < NQP; ++q) {
      const int e = tid + q * NT;
      const int c = (e / (PWP * PH)) % CK;
      const bool ok = ((pvalid >> q) & 1u) && (c0 + c) < Cin;
      st[(e < NP) ? e : DUMMY] = ok ? preg[q] : 0.f;
    }
    if constexpr (C::DMA) {
    } else if constexpr (PK) {
#pragma unroll
      for (int q = 0; q < NQ4; ++q) {
        const int k = min(pk_k0 + q * RPQ, RL - 1);
        *reinterpret_cast<f32x4*>(st + WOFF + k * TN + 4 * pk_c4) = wreg4[q];
      }
    } else {
      const bool cok = wrow_ok && (c0 + w_c) < Cin;
#pragma unroll
      for (int q = 0; q < NQW; ++q) {
        const int co_l = w_sub + q * RP;
        const bool ok = cok && co_l < TN && (n0 + co_l) < Cout;
        st[(wrow_ok && co_l < TN) ? WOFF + co_l * WS + w_r : DUMMY] = ok ? wreg[q] : 0.f;
      }
    }
  };

  // ---- per-lane operand bases
  int base_w[FC], base_p[FP];
  constexpr int CPG = CK / 2 / C::KS;   // channel pairs per K group
#pragma unroll
  for (int g = 0; g < FC; ++g)
    base_w[g] = PK ? WOFF + (h * NTAP + kg * CPG * 2 * NTAP) * TN + (wc * FC + g) * 32 + l32
                   : WOFF + ((wc * FC + g) * 32 + l32) * WS + h * NTAP + kg * CPG * 2 * NTAP;
#pragma unroll
  for (int f = 0; f < FP; ++f) {
    const int m = (wp * FP + f) * 32 + l32;
    const int nb = m / (TH * TW), r = m % (TH * TW);
    base_p[f] = ((nb * CK + h + kg * CPG * 2) * PH + PSTEP * (r / TW)) * PWP + PSTEP * (r % TW);
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
  store_chunk(smem, 0);
  __syncthreads();

  for (int ch = 0; ch < nchunks; ++ch) {
    const float* st = smem + (ch & 1) * C::STAGE;
    const bool more = (ch + 1) < nchunks;
    if (more) load_chunk((ch + 1) * CK);
    // (cp, tap) steps of this K group; the compiler interleaves the LDS reads with the MFMAs
    constexpr int NSTEP = CPG * NTAP;
    float a_cur[FC], b_cur[FP];
    auto read_step = [&](int stp, float* a, float* b) {
      const int cp = stp / NTAP, t = stp % NTAP;
      const int ro = (MODE == MODE_FWD) ? t / 5 : NTMAX - 1 - t / NTW;   // patch row / col offset of the tap
      const int cof = (MODE == MODE_FWD) ? t % 5 : NTMAX - 1 - t % NTW;
#pragma unroll
      for (int g = 0; g < FC; ++g) a[g] = st[base_w[g] + (cp * 2 * NTAP + t) * (PK ? TN : 1)];
#pragma unroll
      for (int f = 0; f < FP; ++f) b[f] = st[base_p[f] + (2 * cp * PH + ro) * PWP + cof];
    };
#pragma unroll
    for (int stp = 0; stp < NSTEP; ++stp) {
      read_step(stp, a_cur, b_cur);
#pragma unroll
      for (int g = 0; g < FC; ++g)
#pragma unroll
        for (int f = 0; f < FP; ++f) acc[g][f] = mfma32(a_cur[g], b_cur[f], acc[g][f]);
    }
    if (more) store_chunk(smem + ((ch + 1) & 1) * C::STAGE, (ch + 1) * CK);
    __syncthreads();
  }

  // ---- in-workgroup split-K: groups 1..KS-1 park their partial tiles in LDS, group 0 adds them
  if constexpr (C::KS > 1) {
    constexpr int PER_WAVE = FC * FP * 16 * 64;
    if (kg > 0) {
      float* dst = smem + ((kg - 1) * C::WC * C::WP + w2) * PER_WAVE + lane;
#pragma unroll
      for (int g = 0; g < FC; ++g)
#pragma unroll
        for (int f = 0; f < FP; ++f)
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[((g * FP + f) * 16 + r) * 64] = acc[g][f][r];
    }
    __syncthreads();
    if (kg > 0) return;
#pragma unroll
    for (int k = 1; k < C::KS; ++k) {
      const float* src = smem + ((k - 1) * C::WC * C::WP + w2) * PER_WAVE + lane;
#pragma unroll
      for (int g = 0; g < FC; ++g)
#pragma unroll
        for (int f = 0; f < FP; ++f)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[g][f][r] += src[((g * FP + f) * 16 + r) * 64];
    }
  }
  // ---- epilogue: + bias, NCHW store
  const int YH = A.YH, YW = A.YW;
  if (MODE == MODE_FWD && A.stats) {
    bool pokf[FP];
#pragma unroll
    for (int f = 0; f < FP; ++f) {
      const int m = (wp * FP + f) * 32 + l32;
      const int nb = m / (TH * TW), r = m % (TH * TW);
      pokf[f] = (b0 + nb) < A.B && (th0 + r / TW) < YH && (tw0 + r % TW) < YW;
    }
    float* sb = A.stats + (size_t)(pt * C::WP + wp) * Cout * 2;
#pragma unroll
    for (int g = 0; g < FC; ++g)
#pragma unroll
      for (int r16 = 0; r16 < 16; ++r16) {
        const int co = n0 + (wc * FC + g) * 32 + acc_row(r16, lane);
        const float bvs = A.bias ? A.bias[min(co, Cout - 1)] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int f = 0; f < FP; ++f) {
          const float v = pokf[f] ? acc[g][f][r16] + bvs : 0.f;
          s1 += v;
          s2 = fmaf(v, v, s2);
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {       // the 32 lanes of a half hold the 32 pixels of this cout row
          s1 += __shfl_xor(s1, o, 64);
          s2 += __shfl_xor(s2, o, 64);
        }
        if (l32 == 0 && co < Cout) {
          sb[2 * co] = s1;
          sb[2 * co + 1] = s2;
        }
      }
  }
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
      float bv[16];
#pragma unroll
      for (int r16 = 0; r16 < 16; ++r16) {   // one batch of loads (clamped rows), one wait
        const int co = min(n0 + (wc * FC + g) * 32 + acc_row(r16, lane), Cout - 1);
        bv[r16] = A.bias ? A.bias[co] : 0.f;
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
__global__ __launch_bounds__(C::NT, C::MINW) void conv5x5_igemm_kernel(Args A) {
  __shared__ __attribute__((aligned(16))) float smem[2 * C::STAGE];
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

// statistics request of a launch: `need` != NULL: only report the floats the chosen tile variant would write
struct IgStats {
  float* ptr;
  size_t cap;
  size_t* need;
};

template <class C>
int launch(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int XH, int XW,
           int Cout, hipStream_t st, const IgStats* sx = nullptr) {
  Args A;
  A.x = x; A.w = w; A.bias = bias; A.y = y;
  A.stats = nullptr;
  A.B = B; A.Cin = Cin; A.XH = XH; A.XW = XW; A.Cout = Cout;
  A.CinP = packed_cin(Cin); A.CoutP = packed_cout(Cout);
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
  if (sx) {
    const size_t floats = (C::MODE == MODE_FWD) ? (size_t)(per_cls / A.ntiles_n) * C::WP * Cout * 2 : 0;
    if (sx->need) {
      *sx->need = floats;
      return 0;
    }
    if (sx->ptr) {
      if (!floats || sx->cap < floats) return VG_ERR_BAD_ARG;
      A.stats = sx->ptr;
    }
  }
  hipLaunchKernelGGL(conv5x5_igemm_kernel<C>, dim3((unsigned)grid), dim3(C::NT), 0, st, A);
  VG_CHECK_LAUNCH();
  return 0;
}

// ---- tile variants ------------------------------------------------------------------
// Geometry by tile-space width: W32 -> rows of 32 pixels, W16 -> 16, W8 -> 8x8 images.
//   variant 0: 128 px x 128 cout, 4 waves              variant 4: 256 px x 128 cout, 8 waves
//   variant 1: 256 px x  64 cout, 8 waves              variant 5: 128 px x 64 cout, 8 waves (K split 2)
//   variant 2:  64 px x  64 cout, 4 waves              variant 6: 128 px x 32 cout, 4 waves (thin)
//   variant 3: 128 px x  64 cout, 4 waves              variant 7: 256 px x 32 cout, 8 waves (thin)
// K chunk: 2 channels (forward / stride 1) or 4 (stride-2 transposed); 4 / 8 for the K-split form.
template <int MODE, int S, int WIDTH, int VAR, bool PK>
struct Pick;
#define VG_PICK(WIDTH, VAR, NB, TH, TW, TN, WC, WP, CKF, CKT, KS)                                             \
  template <int MODE, int S, bool PK>                                                                         \
  struct Pick<MODE, S, WIDTH, VAR, PK> {                                                                      \
    using type = Cfg<MODE, S, NB, TH, TW, TN, WC, WP, ((MODE == MODE_FWD || S == 1) ? CKF : CKT), 2, KS, PK>; \
  };
//       W  V NB TH TW   TN WC WP CKF CKT KS
VG_PICK(32, 0, 1, 4, 32, 128, 2, 2, 2, 4, 1)
VG_PICK(16, 0, 1, 8, 16, 128, 2, 2, 2, 4, 1)
VG_PICK(8, 0, 2, 8, 8, 128, 2, 2, 2, 4, 1)
VG_PICK(32, 1, 1, 8, 32, 64, 2, 4, 2, 4, 1)
VG_PICK(16, 1, 1, 16, 16, 64, 2, 4, 2, 4, 1)
VG_PICK(8, 1, 4, 8, 8, 64, 2, 4, 2, 4, 1)
VG_PICK(32, 2, 1, 2, 32, 64, 2, 2, 2, 4, 1)
VG_PICK(16, 2, 1, 4, 16, 64, 2, 2, 2, 4, 1)
VG_PICK(8, 2, 1, 8, 8, 64, 2, 2, 2, 4, 1)
VG_PICK(32, 3, 1, 4, 32, 64, 2, 2, 2, 4, 1)
VG_PICK(16, 3, 1, 8, 16, 64, 2, 2, 2, 4, 1)
VG_PICK(8, 3, 2, 8, 8, 64, 2, 2, 2, 4, 1)
VG_PICK(32, 4, 1, 8, 32, 128, 2, 4, 2, 4, 1)
VG_PICK(16, 4, 1, 16, 16, 128, 2, 4, 2, 4, 1)
VG_PICK(8, 4, 4, 8, 8, 128, 2, 4, 2, 4, 1)
VG_PICK(32, 5, 1, 4, 32, 64, 2, 2, 4, 8, 2)
VG_PICK(16, 5, 1, 8, 16, 64, 2, 2, 4, 8, 2)
VG_PICK(8, 5, 2, 8, 8, 64, 2, 2, 4, 8, 2)
VG_PICK(32, 6, 1, 4, 32, 32, 1, 4, 2, 4, 1)
VG_PICK(16, 6, 1, 8, 16, 32, 1, 4, 2, 4, 1)
VG_PICK(8, 6, 2, 8, 8, 32, 1, 4, 2, 4, 1)
// thin outputs, 256 px x 32 cout, 8 waves along the pixels
VG_PICK(32, 7, 1, 8, 32, 32, 1, 8, 2, 4, 1)
VG_PICK(16, 7, 1, 16, 16, 32, 1, 8, 2, 4, 1)
VG_PICK(8, 7, 4, 8, 8, 32, 1, 8, 2, 4, 1)
#undef VG_PICK
constexpr int NVAR = 8;

#ifdef VG_TUNING
int g_tile_override[2] = {-1, -1};  // tuning build only (vg_debug_set_conv_tile)
#else
constexpr int g_tile_override[2] = {-1, -1};
#endif

template <int MODE, int S, int WIDTH, bool PK>
int launch_var(int var, const float* x, const float* w, const float* bias, float* y, int B, int Cin, int XH,
               int XW, int Cout, hipStream_t st, const IgStats* sx = nullptr) {
  switch (var) {
    case 0: return launch<typename Pick<MODE, S, WIDTH, 0, PK>::type>(x, w, bias, y, B, Cin, XH, XW, Cout, st, sx);
    case 1: return launch<typename Pick<MODE, S, WIDTH, 1, PK>::type>(x, w, bias, y, B, Cin, XH, XW, Cout, st, sx);
    case 2: return launch<typename Pick<MODE, S, WIDTH, 2, PK>::type>(x, w, bias, y, B, Cin, XH, XW, Cout, st, sx);
    case 3: return launch<typename Pick<MODE, S, WIDTH, 3, PK>::type>(x, w, bias, y, B, Cin, XH, XW, Cout, st, sx);
    case 4: return launch<typename Pick<MODE, S, WIDTH, 4, PK>::type>(x, w, bias, y, B, Cin, XH, XW, Cout, st, sx);
    case 5: return launch<typename Pick<MODE, S, WIDTH, 5, PK>::type>(x, w, bias, y, B, Cin, XH, XW, Cout, st, sx);
    case 6: return launch<typename Pick<MODE, S, WIDTH, 6, PK>::type>(x, w, bias, y, B, Cin, XH, XW, Cout, st, sx);
    default: return launch<typename Pick<MODE, S, WIDTH, 7, PK>::type>(x, w, bias, y, B, Cin, XH, XW, Cout, st, sx);
  }
}

// Tile choice: widest pixel row the tile space supports; cout tile to fit Cout; smaller tiles
// when the big ones would leave CUs idle (256 CUs x 2 resident workgroups).
template <int MODE, int S, bool PK>
int dispatch(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int XH, int XW,
             int Cout, hipStream_t st, const IgStats* sx = nullptr) {
  const int tsw = (MODE == MODE_FWD) ? (XW - 1) / S + 1 : XW;
  const int tsh = (MODE == MODE_FWD) ? (XH - 1) / S + 1 : XH;
  const int width = tsw >= 32 ? 32 : (tsw >= 16 ? 16 : 8);
  // Measured on MI355X at B=128 (scripts/tune_conv.py).  Staging (global loads -> LDS) is the
  // cost next to the MFMAs, so the biggest pixel tile that still fills the chip wins: 256 px x
  // 128 / 64 cout (8 waves) on the large grids, 128 px x 64 cout otherwise -- with 8 waves
  // (in-workgroup split of the K chunk) where 4 waves would leave SIMDs with a single wave.
  const long px256 = (long)cdiv(B * cdiv(tsh, 8) * cdiv(tsw, 8) * 64, 256);
  const int ncls = (MODE == MODE_TR) ? S * S : 1;
  int var;
  if (Cout <= 32) {
    var = (px256 * ncls >= 2048) ? 7 : 6;
  } else if (MODE == MODE_FWD) {
    if (px256 * cdiv(Cout, 128) >= 256 && Cout > 64) var = 4;
    else if (px256 * cdiv(Cout, 64) >= 256) var = 1;
    else var = 5;
  } else if (!PK) {
    if (px256 * cdiv(Cout, 64) * ncls >= 1024) var = 1;
    else var = (2 * px256 * cdiv(Cout, 64) * ncls < 768) ? 5 : 3;
  } else {
    // packed filters make the 128-cout tile's slab copy cheap: 128 px x 128 cout where it fills the chip
    if (Cout > 64 && 2 * px256 * cdiv(Cout, 128) * ncls >= 1024) var = 0;
    else if (px256 * cdiv(Cout, 64) * ncls >= 1024) var = 1;
    else var = (2 * px256 * cdiv(Cout, 64) * ncls < 768) ? 2 : 3;
  }
  const int ov = g_tile_override[MODE];
  const int use = (ov >= 0 && ov < NVAR) ? ov : var;
  if (width == 32) return launch_var<MODE, S, 32, PK>(use, x, w, bias, y, B, Cin, XH, XW, Cout, st, sx);
  if (width == 16) return launch_var<MODE, S, 16, PK>(use, x, w, bias, y, B, Cin, XH, XW, Cout, st, sx);
  return launch_var<MODE, S, 8, PK>(use, x, w, bias, y, B, Cin, XH, XW, Cout, st, sx);
}

// ---- filter pre-pack: [class][ci][tap][CoutP], zero padded to CinP x CoutP.  One workgroup
// transposes the 25 taps of 64 output channels of one input channel through LDS.
template <int TRANSPOSED>
__global__ __launch_bounds__(256) void pack_filter_kernel(const float* __restrict__ w, float* __restrict__ p,
                                                          int Cout, int Cin, int CinP, int CoutP, int S) {
  __shared__ float t[64 * 25 + 1];
  const int ci = blockIdx.x, co0 = blockIdx.y * 64, tid = threadIdx.x;
  for (int e = tid; e < 64 * 25; e += 256) {
    const int col = e / 25, tap = e - col * 25, co = co0 + col;
    float v = 0.f;
    if (ci < Cin && co < Cout)
      v = TRANSPOSED ? w[((size_t)ci * Cout + co) * 25 + tap] : w[((size_t)co * Cin + ci) * 25 + tap];
    t[e] = v;
  }
  __syncthreads();
  // packed position of tap (kh, kw): class (kh % S, kw % S), index (kh / S) * ntw + kw / S
  for (int e = tid; e < 64 * 25; e += 256) {
    const int tap = e / 64, col = e % 64;
    const int kh = tap / 5, kw = tap % 5;
    const int R = kh % S, SS = kw % S;
    const int ntw = (5 - SS + S - 1) / S, nth = (5 - R + S - 1) / S;
    const int tb = taps_before(S, R, SS);
    const size_t row = (size_t)tb * CinP + (size_t)ci * (nth * ntw) + (kh / S) * ntw + kw / S;
    p[row * CoutP + co0 + col] = t[col * 25 + tap];
  }
}

}  // namespace

#ifdef VG_TUNING
extern "C" int vg_debug_set_conv_tile(int mode, int variant) {
  if (mode < 0 || mode > 1) return VG_ERR_BAD_ARG;
  g_tile_override[mode] = variant;
  return 0;
}
#endif

extern "C" size_t vg_conv5x5_packed_floats(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0) return 0;
  return (size_t)25 * packed_cin(Cin) * packed_cout(Cout);
}

extern "C" int vg_conv5x5_pack(const float* w, float* packed, int Cout, int Cin, int transposed, int stride,
                               void* stream) {
  if (!w || !packed || Cout <= 0 || Cin <= 0 || (stride != 1 && stride != 2)) return VG_ERR_BAD_ARG;
  if (((uintptr_t)packed & 15) != 0) return VG_ERR_BAD_ARG;
  const int CinP = packed_cin(Cin), CoutP = packed_cout(Cout);
  const dim3 grid((unsigned)CinP, (unsigned)(CoutP / 64));
  if (CoutP / 64 > 65535) return VG_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (transposed)
    hipLaunchKernelGGL(pack_filter_kernel<1>, grid, dim3(256), 0, st, w, packed, Cout, Cin, CinP, CoutP, stride);
  else   // the forward kernel walks all 25 taps as one class
    hipLaunchKernelGGL(pack_filter_kernel<0>, grid, dim3(256), 0, st, w, packed, Cout, Cin, CinP, CoutP, 1);
  VG_CHECK_LAUNCH();
  return 0;
}

static int conv_args_ok(const float* x, const float* w, float* y, int B, int Cin, int H, int W, int Cout,
                        int stride) {
  if (!x || !w || !y || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
  return stride == 1 || stride == 2;
}

extern "C" int vg_conv5x5_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Cin,
                              int H, int W, int Cout, int stride, void* stream) {
  if (!conv_args_ok(x, w, y, B, Cin, H, W, Cout, stride)) return VG_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (stride == 2) return dispatch<MODE_FWD, 2, false>(x, w, bias, y, B, Cin, H, W, Cout, st);
  return dispatch<MODE_FWD, 1, false>(x, w, bias, y, B, Cin, H, W, Cout, st);
}

extern "C" int vg_conv5x5_fwd_packed(const float* x, const float* packed, const float* bias, float* y, int B,
                                     int Cin, int H, int W, int Cout, int stride, void* stream) {
  if (!conv_args_ok(x, packed, y, B, Cin, H, W, Cout, stride) || ((uintptr_t)packed & 15) != 0)
    return VG_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (stride == 2) return dispatch<MODE_FWD, 2, true>(x, packed, bias, y, B, Cin, H, W, Cout, st);
  return dispatch<MODE_FWD, 1, true>(x, packed, bias, y, B, Cin, H, W, Cout, st);
}

extern "C" size_t vg_conv5x5_fwd_packed_stats_floats(int B, int Cin, int H, int W, int Cout, int stride) {
  if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2)) return 0;
  size_t need = 0;
  const IgStats sx = {nullptr, 0, &need};
  const int rc = (stride == 2) ? dispatch<MODE_FWD, 2, true>(nullptr, nullptr, nullptr, nullptr, B, Cin, H, W, Cout, nullptr, &sx)
                               : dispatch<MODE_FWD, 1, true>(nullptr, nullptr, nullptr, nullptr, B, Cin, H, W, Cout, nullptr, &sx);
  return rc == 0 ? need : 0;
}

extern "C" int vg_conv5x5_fwd_packed_stats(const float* x, const float* packed, const float* bias, float* y, int B,
                                           int Cin, int H, int W, int Cout, int stride, float* stats,
                                           size_t stats_floats, void* stream) {
  if (!conv_args_ok(x, packed, y, B, Cin, H, W, Cout, stride) || ((uintptr_t)packed & 15) != 0 || !stats)
    return VG_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const IgStats sx = {stats, stats_floats, nullptr};
  if (stride == 2) return dispatch<MODE_FWD, 2, true>(x, packed, bias, y, B, Cin, H, W, Cout, st, &sx);
  return dispatch<MODE_FWD, 1, true>(x, packed, bias, y, B, Cin, H, W, Cout, st, &sx);
}

extern "C" int vg_convT5x5_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Cin,
                               int H, int W, int Cout, int stride, void* stream) {
  if (!conv_args_ok(x, w, y, B, Cin, H, W, Cout, stride)) return VG_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (stride == 2) return dispatch<MODE_TR, 2, false>(x, w, bias, y, B, Cin, H, W, Cout, st);
  // <= 4 output channels: direct VALU kernel (an MFMA tile would be 3/32 full)
  if (Cout <= 4 && g_tile_override[MODE_TR] < 0) return vg_internal_convT_s1_thin(x, w, bias, y, B, Cin, H, W, Cout, st);
  return dispatch<MODE_TR, 1, false>(x, w, bias, y, B, Cin, H, W, Cout, st);
}

extern "C" int vg_convT5x5_fwd_packed(const float* x, const float* packed, const float* bias, float* y, int B,
                                      int Cin, int H, int W, int Cout, int stride, void* stream) {
  if (!conv_args_ok(x, packed, y, B, Cin, H, W, Cout, stride) || ((uintptr_t)packed & 15) != 0)
    return VG_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (stride == 2) return dispatch<MODE_TR, 2, true>(x, packed, bias, y, B, Cin, H, W, Cout, st);
  return dispatch<MODE_TR, 1, true>(x, packed, bias, y, B, Cin, H, W, Cout, st);
}
