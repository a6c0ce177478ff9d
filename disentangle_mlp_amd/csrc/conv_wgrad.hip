// Weight gradient of the 5x5 / pad 2 / stride {1,2} convolutions (and, with the
// operand roles swapped, of the transposed convolutions) for gfx950: an implicit
// GEMM on the fp32-input MFMA with the reduction over (image, output pixel).
//
//   dw[co][ci][kh][kw] = sum_{b,oh,ow} gy[b][co][oh][ow] * x[b][ci][S*oh+kh-2][S*ow+kw-2]
//
// Autograd of nn.Conv2d / nn.ConvTranspose2d at /root/reference/models/model.py:
// 389-398, 450-456, 495-507, reached from the backward() calls of
// experiments/new_betavaegan.py:103,121,157-163,185-187.
//
// GEMM view: D[co][n], n = ci*25 + tap, K = (b, oh, ow).  A workgroup owns TM
// output channels x 5 input channels (125 of 128 MFMA columns) and a slice of
// K (split-K over images / pixel tiles).  Per K chunk (one image, 64 output
// pixels) it stages gy[TM][64] (contiguous NCHW rows) and the input patch
// [5][PH][PW] in LDS; lane (j,h) reads x-patch[ci(j)][S*oh+kh(j)][S*ow+kw(j)] at a
// per-lane base + compile-time immediate.  Partial slabs go to the workspace
// and are summed in a fixed order by a second kernel (deterministic, no atomics).
#include "common.hpp"
#include "vaegan_hip.h"

namespace {

constexpr int WG_NT = 256;
constexpr int KC = 64;    // output pixels per K chunk

// CIT input channels per column tile: 5 -> 125 of 128 MFMA columns, 10 -> 250 of 256 (the gy tile is
// then staged half as often per FLOP; 8 waves as 2 x 4 instead of 2 x 2 x 2 K groups)
template <int S_, int TW_, int TM_, int KS_ = 1, int CIT_ = 5>
struct WCfg {
  static constexpr int S = S_, TW = TW_, TH = KC / TW_, TM = TM_, KS = KS_, CIT = CIT_;
  static constexpr int TNP = (CIT * 25 + 31) / 32 * 32;   // padded column tile
  static constexpr int WM = (TM >= 64) ? 2 : 1, WN = (TNP / 128) * (4 / WM);
  static constexpr int NT = 64 * WM * WN * KS;   // KS wave groups split the 64-pixel K chunk
  static constexpr int FC = TM / 32 / WM, FN = TNP / 32 / WN;
  static constexpr int PH = S * (TH - 1) + 5, PW = S * (TW - 1) + 5, PWP = PW | 1;
  static constexpr int NPATCH = CIT * PH * PWP;
  static constexpr int AS = KC + 1;
  static constexpr int STAGE = TM * AS + NPATCH;
  static constexpr int NQP = cdiv(NPATCH, NT), NQA = TM / (NT / 64);
  static_assert(NQA % 4 == 0 || NQA == 2, "gy rows per thread: whole 16-byte groups");
  static_assert(KS == 1 || STAGE >= 4 * (128 / 32 / (4 / WM)) * 16 * 64, "LDS holds one fragment row of partials per wave");
};

struct WArgs {
  const float* x;
  const float* gy;
  float* ws;
  int B, Cin, H, W, Cout, OH, OW;
  int mtiles, ntiles, splits, tiles_w, tiles_hw, chunks, chunks_per_split;
  int vec4;   // gy rows can be read as aligned 16-byte pieces (OW % 4 == 0, 16-byte aligned base)
};

template <class C>
__global__ __launch_bounds__(C::NT, 2) void conv5x5_wgrad_kernel(WArgs A) {
  constexpr int S = C::S, TW = C::TW, TH = C::TH, TM = C::TM, FC = C::FC, FN = C::FN;
  constexpr int PH = C::PH, PW = C::PW, PWP = C::PWP, NPATCH = C::NPATCH, AS = C::AS;
  constexpr int NQP = C::NQP, NQA = C::NQA;
  __shared__ float smem[C::STAGE];
  float* gyl = smem;
  float* pl = smem + TM * AS;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int h = lane >> 5, l32 = lane & 31;
  constexpr int CIT = C::CIT, WPT = C::WM * C::WN;
  const int kg = wid / WPT, w4 = wid % WPT;   // K group, wave within the output tile
  const int wm = w4 % C::WM, wn = w4 / C::WM;
  constexpr int NT = C::NT, NWV = C::NT / 64;
  int bid = blockIdx.x;
  const int mt = bid % A.mtiles;
  bid /= A.mtiles;
  const int nt = bid % A.ntiles, split = bid / A.ntiles;
  const int m0 = mt * TM, ci0 = nt * CIT;
  const int Cin = A.Cin, Cout = A.Cout, H = A.H, W = A.W, OH = A.OH, OW = A.OW;
  const int HW = H * W, OHW = OH * OW;

  // ---- staging maps
  int pk[NQP], prc[NQP];
#pragma unroll
  for (int q = 0; q < NQP; ++q) {
    const int e = tid + q * NT;
    const int col = e % PWP;
    int t = e / PWP;
    const int r = t % PH, ci_l = t / PH;
    const bool ok = e < NPATCH && col < PW && (ci0 + ci_l) < Cin;
    pk[q] = ok ? ci_l * HW + r * W + col : -1;
    prc[q] = r | (col << 16);
  }
  const int a_pix = lane;  // this thread's pixel within the chunk tile
  const int a_ph = a_pix / TW, a_pw = a_pix % TW;

  float preg[NQP], areg[NQA];
  auto load_chunk = [&](int chunk) {
    const int b = chunk / A.tiles_hw, sp = chunk % A.tiles_hw;
    const int th0 = (sp / A.tiles_w) * TH, tw0 = (sp % A.tiles_w) * TW;
    const int ih0 = S * th0 - 2, iw0 = S * tw0 - 2;
    const float* xb = A.x + ((size_t)b * Cin + ci0) * HW;
    const int shift = ih0 * W + iw0;
#pragma unroll
    for (int q = 0; q < NQP; ++q) {
      const int ih = ih0 + (prc[q] & 0xffff), iw = iw0 + (prc[q] >> 16);
      const bool ok = pk[q] >= 0 && ih >= 0 && ih < H && iw >= 0 && iw < W;
      preg[q] = ok ? xb[pk[q] + shift] : 0.f;
    }
    if (A.vec4) {
      // OW % 4 == 0: 16-byte loads, 16 lanes per 64-pixel row, 4 rows per wavefront instruction
      const int pix = (lane & 15) * 4, rsub = wid * 4 + (lane >> 4);
      const int oh = th0 + pix / TW, ow = tw0 + pix % TW;
      const bool pok = oh < OH && ow < OW;
      const float* gb = A.gy + ((size_t)b * Cout + m0 + rsub) * OHW + oh * OW + ow;
#pragma unroll
      for (int q = 0; q < NQA / 4; ++q) {
        const bool ok = pok && (m0 + rsub + NWV * 4 * q) < Cout;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) v = *reinterpret_cast<const f32x4*>(gb + (size_t)q * NWV * 4 * OHW);
        areg[4 * q] = v[0]; areg[4 * q + 1] = v[1]; areg[4 * q + 2] = v[2]; areg[4 * q + 3] = v[3];
      }
    } else {
      const int oh = th0 + a_ph, ow = tw0 + a_pw;
      const bool pok = oh < OH && ow < OW;
      const float* gb = A.gy + ((size_t)b * Cout + m0 + wid) * OHW + oh * OW + ow;
#pragma unroll
      for (int q = 0; q < NQA; ++q) {
        const bool ok = pok && (m0 + wid + NWV * q) < Cout;
        areg[q] = ok ? gb[(size_t)q * NWV * OHW] : 0.f;
      }
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int q = 0; q < NQP; ++q) {
      const int e = tid + q * NT;
      if (e < NPATCH) pl[e] = preg[q];
    }
    if (A.vec4) {
      const int pix = (lane & 15) * 4, rsub = wid * 4 + (lane >> 4);
#pragma unroll
      for (int q = 0; q < NQA / 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) gyl[(rsub + NWV * 4 * q) * AS + pix + j] = areg[4 * q + j];
    } else {
#pragma unroll
      for (int q = 0; q < NQA; ++q) gyl[(wid + NWV * q) * AS + a_pix] = areg[q];
    }
  };

  // ---- per-lane operand bases
  int base_a[FC], base_b[FN];
  constexpr int KPG = KC / 2 / C::KS;            // pixel pairs per K group
  constexpr int K0 = 2 * KPG;                    // first pixel of group 1
  constexpr int IMM0 = (S * (K0 / TW)) * PWP + S * (K0 % TW);
#pragma unroll
  for (int g = 0; g < FC; ++g) base_a[g] = ((wm * FC + g) * 32 + l32) * AS + h + kg * K0;
#pragma unroll
  for (int f = 0; f < FN; ++f) {
    const int n = (wn * FN + f) * 32 + l32;
    const int ci_l = n / 25, tap = n % 25;
    base_b[f] = (n < CIT * 25) ? TM * AS + (ci_l * PH + tap / 5) * PWP + tap % 5 + h * S + kg * IMM0 : TM * AS;
  }

  f32x16 acc[FC][FN];
#pragma unroll
  for (int g = 0; g < FC; ++g)
#pragma unroll
    for (int f = 0; f < FN; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[g][f][r] = 0.f;

  const int c_begin = split * A.chunks_per_split;
  const int c_end = min(c_begin + A.chunks_per_split, A.chunks);
  if (c_begin < c_end) {
    load_chunk(c_begin);
    store_chunk();
    __syncthreads();
    for (int ch = c_begin; ch < c_end; ++ch) {
      const bool more = (ch + 1) < c_end;
      if (more) load_chunk(ch + 1);
#pragma unroll
      for (int kp = 0; kp < KPG; ++kp) {
        const int k = 2 * kp;
        const int imm_b = (S * (k / TW)) * PWP + S * (k % TW);
        float a[FC], b[FN];
#pragma unroll
        for (int g = 0; g < FC; ++g) a[g] = smem[base_a[g] + k];
#pragma unroll
        for (int f = 0; f < FN; ++f) b[f] = smem[base_b[f] + imm_b];
#pragma unroll
        for (int g = 0; g < FC; ++g)
#pragma unroll
          for (int f = 0; f < FN; ++f) acc[g][f] = mfma32(a[g], b[f], acc[g][f]);
      }
      __syncthreads();
      if (more) {
        store_chunk();
        __syncthreads();
      }
    }
  }

  // ---- in-workgroup K split: group 1 parks its partial tile in LDS (one 32-row fragment row at
  // a time: FN*16*64 floats per wave), group 0 adds it
  if constexpr (C::KS > 1) {
    static_assert(C::KS == 2, "two K groups");
    __syncthreads();
#pragma unroll
    for (int g = 0; g < FC; ++g) {
      float* slot = smem + (w4 * FN * 16) * 64 + lane;
      if (kg == 1) {
#pragma unroll
        for (int f = 0; f < FN; ++f)
#pragma unroll
          for (int r = 0; r < 16; ++r) slot[(f * 16 + r) * 64] = acc[g][f][r];
      }
      __syncthreads();
      if (kg == 0) {
#pragma unroll
        for (int f = 0; f < FN; ++f)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[g][f][r] += slot[(f * 16 + r) * 64];
      }
      __syncthreads();
    }
    if (kg == 1) return;
  }
  // ---- partial slab store: ws[split][co][ci*25+tap] (lanes = consecutive n: coalesced)
  float* wsb = A.ws + (size_t)split * Cout * Cin * 25;
#pragma unroll
  for (int f = 0; f < FN; ++f) {
    const int n = (wn * FN + f) * 32 + l32;
    const bool nok = n < CIT * 25 && (ci0 + n / 25) < Cin;
#pragma unroll
    for (int g = 0; g < FC; ++g) {
#pragma unroll
      for (int r16 = 0; r16 < 16; ++r16) {
        const int co = m0 + (wm * FC + g) * 32 + acc_row(r16, lane);
        if (nok && co < Cout) wsb[((size_t)co * Cin + ci0) * 25 + n] = acc[g][f][r16];
      }
    }
  }
}

// dw[i] = sum_k ws[k][i] in a fixed order.  A workgroup owns 64 consecutive elements; its NW
// wavefronts each sum every NW-th slab (coalesced 256-B rows), then combine through LDS in a
// fixed tree -- deterministic.  NW = 4 for the big filters (many workgroups, few slabs each), 16
// where a small filter was split hundreds of ways (few workgroups: the slab chain is the latency).
template <int NW>
__global__ __launch_bounds__(64 * NW) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                              int n, int splits, int accumulate) {
  __shared__ float red[NW][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (i < n) {
    int k = wid;
    for (; k + 3 * NW < splits; k += 4 * NW) {        // four slabs with their loads in flight together, added in slab order
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = ws[(size_t)(k + j * NW) * n + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) s += v[j];
    }
    for (; k < splits; k += NW) s += ws[(size_t)k * n + i];
  }
  red[wid][lane] = s;
  __syncthreads();
#pragma unroll
  for (int w = NW / 2; w >= 2; w >>= 1) {     // pairwise tree over the wavefront partials
    if (wid < w) red[wid][lane] += red[wid + w][lane];
    __syncthreads();
  }
  // accumulate: added to what dw holds (a filter used twice before one backward: no separate add launch)
  if (wid == 0 && i < n) dw[i] = (red[0][lane] + red[1][lane]) + (accumulate ? dw[i] : 0.f);
}

// The same sum for the common case -- n % 4 == 0, 16-byte aligned, fewer than 64 slabs: 16 bytes per lane, eight slabs'
// loads in flight per thread, added in slab order.  (The 4-byte form above took 10 us for the 17 MB of a Linear GEMM's
// sixteen slabs and 13 us for a K-split convolution's 32 MB: 5 us and 8 us here.)
__global__ __launch_bounds__(256) void slab_sum4_kernel(const f32x4* __restrict__ slabs, f32x4* __restrict__ out, int n4,
                                                       int splits, int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 8 <= splits; k += 8) {
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = slabs[(size_t)(k + j) * n4 + i];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
  }
  for (; k + 4 <= splits; k += 4) {
    f32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = slabs[(size_t)(k + j) * n4 + i];
#pragma unroll
    for (int j = 0; j < 4; ++j) s += v[j];
  }
  for (; k < splits; ++k) s += slabs[(size_t)k * n4 + i];
  out[i] = accumulate ? s + out[i] : s;
}

struct Plan {
  int tw, tm, cit, mtiles, ntiles, splits, tiles_w, tiles_hw, chunks, cps, OH, OW;
};

VG_KNOB(int, g_wgrad_ks, 2);                 // 8-wave K-split kernel for the 128-row tile (1 = 4-wave form, diagnostics)
VG_KNOB(int, g_wgrad_cit, 5);                // 10 = 250-column tile for the 128-row tile (measured slower: diagnostics)
VG_KNOB(int, g_wgrad_vec4, 1);               // diagnostics: 0 = scalar gy loads everywhere
VG_KNOB(int, g_wgrad_tm_override, -1);       // diagnostics only (vg_debug_set_conv_tile mode 2)
VG_KNOB(int, g_wgrad_blocks_target, -1);     // diagnostics only

Plan make_plan(int B, int Cin, int H, int W, int Cout, int S) {
  Plan p;
  p.OH = (H - 1) / S + 1;
  p.OW = (W - 1) / S + 1;
  p.tw = p.OW <= 8 ? 8 : (p.OW <= 16 ? 16 : (p.OW <= 32 ? 32 : 64));
  p.tm = Cout > 64 ? 128 : (Cout > 32 ? 64 : 32);
  if (p.tm == 128 && Cout <= 128 && Cin <= 32) p.tm = 64;   // few column tiles: more row tiles instead
  if (g_wgrad_tm_override > 0 && g_wgrad_tm_override <= p.tm) p.tm = g_wgrad_tm_override;
  p.mtiles = cdiv(Cout, p.tm);
  p.cit = (g_wgrad_cit == 10 && p.tm == 128 && Cin >= 20) ? 10 : 5;
  p.ntiles = cdiv(Cin, p.cit);
  const int th = KC / p.tw;
  p.tiles_w = cdiv(p.OW, p.tw);
  p.tiles_hw = p.tiles_w * cdiv(p.OH, th);
  p.chunks = B * p.tiles_hw;
  // split-K so that every layer fills the chip (measured, scripts/tune_wgrad.py): ~2048
  // workgroups for the wide layers, fewer slabs for the thin ones (their cost is the reduction)
  const int tiles = p.mtiles * p.ntiles;
  int target = g_wgrad_blocks_target > 0 ? g_wgrad_blocks_target : (tiles >= 32 ? 2048 : (tiles >= 4 ? 1024 : 512));
  int want = cdiv(target, tiles);
  if (want > p.chunks) want = p.chunks;
  if (want < 1) want = 1;
  p.cps = cdiv(p.chunks, want);
  p.splits = cdiv(p.chunks, p.cps);
  return p;
}

template <class C>
int launch_w(const WArgs& A, hipStream_t st) {
  const long grid = (long)A.mtiles * A.ntiles * A.splits;
  hipLaunchKernelGGL(conv5x5_wgrad_kernel<C>, dim3((unsigned)grid), dim3(C::NT), 0, st, A);
  VG_CHECK_LAUNCH();
  return 0;
}

template <int S, int TW>
int dispatch_tm(const WArgs& A, int tm, int cit, hipStream_t st) {
  if (tm == 128 && cit == 10) return launch_w<WCfg<S, TW, 128, 1, 10>>(A, st);
  if (tm == 128) return g_wgrad_ks == 2 ? launch_w<WCfg<S, TW, 128, 2>>(A, st) : launch_w<WCfg<S, TW, 128>>(A, st);
  if (tm == 64) return launch_w<WCfg<S, TW, 64>>(A, st);
  return launch_w<WCfg<S, TW, 32>>(A, st);
}

template <int S>
int dispatch_tw(const WArgs& A, int tw, int tm, int cit, hipStream_t st) {
  switch (tw) {
    case 8: return dispatch_tm<S, 8>(A, tm, cit, st);
    case 16: return dispatch_tm<S, 16>(A, tm, cit, st);
    case 32: return dispatch_tm<S, 32>(A, tm, cit, st);
    default: return dispatch_tm<S, 64>(A, tm, cit, st);
  }
}

}  // namespace

// shared with wgrad_bf16split.hip: dw[i] = sum over `splits` slabs of n floats, fixed order
int vg_internal_wgrad_reduce(const float* slabs, float* dw, int n, int splits, hipStream_t st, int accumulate) {
  if (splits < 64 && n % 4 == 0 && (((uintptr_t)slabs | (uintptr_t)dw) & 15) == 0) {
    hipLaunchKernelGGL(slab_sum4_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, st, (const f32x4*)slabs, (f32x4*)dw, n / 4,
                       splits, accumulate);
    VG_CHECK_LAUNCH();
    return 0;
  }
  if (splits >= 64 && cdiv(n, 64) < 1024)
    hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3(cdiv(n, 64)), dim3(1024), 0, st, slabs, dw, n, splits, accumulate);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3(cdiv(n, 64)), dim3(256), 0, st, slabs, dw, n, splits, accumulate);
  VG_CHECK_LAUNCH();
  return 0;
}

#ifdef VG_TUNING
extern "C" int vg_debug_set_wgrad(int what, int value) {
  if (what == 0) g_wgrad_tm_override = value;
  else if (what == 1) g_wgrad_blocks_target = value;
  else if (what == 2) g_wgrad_ks = value;
  else if (what == 3) g_wgrad_cit = value;
  else if (what == 4) g_wgrad_vec4 = value;
  else if (what == 5) vg_internal_wx_set_th(value);         // split-bf16 kernel: pixel rows of a chunk (1 / 2; 0 = planned)
  else return VG_ERR_BAD_ARG;
  return 0;
}
#endif

extern "C" size_t vg_conv5x5_wgrad_workspace_bytes(int B, int Cin, int H, int W, int Cout, int stride) {
  if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2)) return 0;
  const Plan p = make_plan(B, Cin, H, W, Cout, stride);
  return (size_t)p.splits * Cout * Cin * 25 * sizeof(float);
}

extern "C" int vg_conv5x5_wgrad(const float* x, const float* gy, float* dw, int B, int Cin, int H, int W,
                                int Cout, int stride, void* workspace, size_t workspace_bytes, int accumulate,
                                void* stream) {
  if (!x || !gy || !dw || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return VG_ERR_BAD_ARG;
  if (stride != 1 && stride != 2) return VG_ERR_BAD_ARG;
  const Plan p = make_plan(B, Cin, H, W, Cout, stride);
  const size_t need = (size_t)p.splits * Cout * Cin * 25 * sizeof(float);
  if (!workspace || workspace_bytes < need) return VG_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  WArgs A;
  A.x = x; A.gy = gy; A.ws = (float*)workspace;
  A.B = B; A.Cin = Cin; A.H = H; A.W = W; A.Cout = Cout; A.OH = p.OH; A.OW = p.OW;
  A.mtiles = p.mtiles; A.ntiles = p.ntiles; A.splits = p.splits;
  A.tiles_w = p.tiles_w; A.tiles_hw = p.tiles_hw; A.chunks = p.chunks; A.chunks_per_split = p.cps;
  A.vec4 = (p.OW % 4 == 0 && ((uintptr_t)gy & 15) == 0 && g_wgrad_vec4) ? 1 : 0;
  int rc = (stride == 2) ? dispatch_tw<2>(A, p.tw, p.tm, p.cit, st) : dispatch_tw<1>(A, p.tw, p.tm, p.cit, st);
  if (rc) return rc;
  return vg_internal_wgrad_reduce((const float*)workspace, dw, Cout * Cin * 25, p.splits, st, accumulate ? 1 : 0);
}
