// OPT-IN arithmetic modes of the 5x5 weight gradient (vg_conv5x5_wgrad_bf16split) for gfx950, the
// companion of conv_bf16split.hip: operands split into 2 (hi/lo, 3 products, ~4.5e-6) or 3 (hi/mid/lo,
// 6 products, fp32-equivalent) bf16 planes, v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
//
//   dw[co][ci][kh][kw] = sum_{b,oh,ow} gy[b][co][oh][ow] * x[b][ci][S*oh+kh-2][S*ow+kw-2]
//   (autograd of nn.Conv2d / nn.ConvTranspose2d, /root/reference/models/model.py:389-398, 450-456,
//    495-507; same contract as vg_conv5x5_wgrad)
//
// GEMM view: D[co][n = ci*25 + tap], reduction over (image, output pixel).  The bf16 MFMA wants 8
// consecutive k per lane; "consecutive pixels" would make the patch operand an unaligned 16-byte
// LDS read, so k runs over IMAGES: one MFMA step = one output pixel x 16 images (k-block 0 / 1 =
// images 0-7 / 8-15).  Both operands must therefore be batch-innermost and split into planes:
//   gy: one re-layout pass -> Gp[image group][pixel][plane][k-block][co] x 8 images (16 B)   (LDS transpose);
//   x : no pass -- the kernel stages its patch straight from NCHW (8 loads an image apart per unit) and splits it
//       while it writes it to LDS.
// The kernel (4 wavefronts, 128 co x 5 ci (125 of 128 columns), 2 x 2 fragments per wavefront):
//   * per chunk = (image group, 4 x 8 output pixels) the x patch [plane][k-block][5 ci][rows][cols]
//     is copied to LDS as 16-byte units; MFMA column n reads unit (ci, S*oh+kh, S*ow+kw) at a
//     per-lane base + compile-time pixel offset; row / channel strides are 5 / 9 (mod 16) units so
//     that column n lands on bank group n mod 16 (conflict-free ds_read_b128);
//   * the gy operand goes global -> register (32 consecutive co = 512 contiguous bytes), prefetched
//     one pixel ahead -- no LDS, no barrier for it;
//   * split-K over chunks, partial slabs summed in a fixed order by conv_wgrad.hip's reduction.
#include <algorithm>
#include <vector>

#include "common.hpp"
#include "vaegan_hip.h"

namespace {

constexpr int WTW = 8, WCIT = 5;

struct WXArgs {
  const float* x;      // NCHW fp32 (the patch is split while it is staged)
  int B;
  // fused BatchNorm: x is read as act(x * in_scale[ci] + in_shift[ci]) (vg_conv_fusion semantics); NULL: as it is
  const float* in_scale;
  const float* in_shift;
  float in_slope;      // activation as max(v, 0) + slope * min(v, 0): 1 none, 0 ReLU, 0.2 LeakyReLU
  const bf16x8* gp;
  float* ws;
  int Cin, H, W, Cout, CoP, OH, OW;
  int mtiles, ntiles, tiles_w, tiles_hw, chunks;
  int units, upw;                 // units = output tiles x chunks; every workgroup works off upw consecutive ones
  unsigned short order[256];      // launch slot (blockIdx.x) -> workgroup index
  const float* x_amax;            // fp16 planes: x_amax[0] >= max |activated x| (device); the slab sum undoes the scales
};

// ---- gy[B][Co][P] -> Gp[bg][p][plane][kb][CoP] x 8 images; 32 pixels x 32 channels per workgroup
// optional g_scale / g_shift / g_slope: the operand is read as act(gy * scale[co] + shift[co]) (a fused BatchNorm: the
// weight gradient of a TRANSPOSED convolution takes the layer's input here)
// F16: fp16 planes of gy * f16_scale_of(g_amax[0]) (common.hpp, "split arithmetics")
template <int NP, bool F16>
__global__ __launch_bounds__(256) void relayout_gy_kernel(const float* __restrict__ gy, bf16x8* __restrict__ gp,
                                                         int B, int Co, int CoP, int P, const float* __restrict__ g_scale,
                                                         const float* __restrict__ g_shift, float g_slope,
                                                         const float* __restrict__ g_amax) {
  __shared__ float tile[8][32][33];
  const float gsc16 = F16 ? f16_scale_of(*g_amax) : 1.f;
  const int tid = threadIdx.x;
  const int p0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  const int bg = blockIdx.z >> 1, kb = blockIdx.z & 1;
  if ((P & 3) == 0 && p0 + 32 <= P && ((uintptr_t)gy & 15) == 0) {
    // whole tile inside the image, rows 16-byte aligned: one 16-byte load per (image, channel, 4 pixels) -- 8 loads per
    // thread in flight instead of 32 scalar ones
    const int co = co0 + (tid >> 3), q4 = (tid & 7) * 4;
    const float sc = (g_scale && co < Co) ? g_scale[co] : 1.f, sh = (g_scale && co < Co) ? g_shift[co] : 0.f;
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int b = bg * 16 + kb * 8 + j;
      const float* src = gy + ((size_t)min(b, B - 1) * Co + min(co, Co - 1)) * P + p0 + q4;
      v[j] = *reinterpret_cast<const f32x4*>(src);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = (bg * 16 + kb * 8 + j) < B && co < Co;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = v[j][e];
        if (g_scale) {
          t = fmaf(t, sc, sh);
          t = act_slope(t, g_slope);
        }
        tile[j][tid >> 3][q4 + e] = ok ? t : 0.f;
      }
    }
  } else {
    const int r = tid >> 5, pl = tid & 31;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int b = bg * 16 + kb * 8 + j;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int co = co0 + r + 8 * rr;
        float v = 0.f;
        if (b < B && co < Co && p0 + pl < P) {
          v = gy[((size_t)b * Co + co) * P + p0 + pl];
          if (g_scale) {
            v = fmaf(v, g_scale[co], g_shift[co]);
            v = act_slope(v, g_slope);
          }
        }
        tile[j][r + 8 * rr][pl] = v;
      }
    }
  }
  __syncthreads();
  const int cc = tid & 31;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int pp = (tid >> 5) + 8 * k;
    if (p0 + pp >= P) continue;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = F16 ? tile[j][cc][pp] * gsc16 : tile[j][cc][pp];
    f32x4 pl[NP];
    split_planes16<NP, F16>(v, pl);
#pragma unroll
    for (int p = 0; p < NP; ++p)
      gp[((((size_t)bg * P + p0 + pp) * NP + p) * 2 + kb) * CoP + co0 + cc] = __builtin_bit_cast(bf16x8, pl[p]);
  }
}

// ---- the weight-gradient kernel: one workgroup of 8 wavefronts per CU ------------------------------------------
// Output tile = TCO cout x 5 input channels (125 of 128 columns); WCO wavefronts along cout (32 rows each, so every
// wavefront streams ONLY its own gy rows, global -> register, prefetched PD pixels ahead) x WN = 8 / WCO column
// groups of 128 / WN columns.  A chunk = 16 images x (TH x 8) output pixels, TH = 2 when the output height is even:
// its x patch [plane][k-block][5 ci][S*(TH-1)+5 rows][S*7+5 cols] (72 KB for 3 planes at TH = 2) is double-buffered in
// LDS -- the next chunk's units are loaded into registers at the first pixel of a chunk and split + written one unit
// at a time at later pixels (the two wavefronts of a SIMD at different ones), ONE barrier per chunk -- so the MFMAs
// never wait for a copy.  Work is dealt out in equal shares of (tile, chunk) units (plan_shares below).  (The
// 4-wavefront predecessor staged the patch between two barriers and prefetched gy one pixel ahead: 142 TFLOP/s on
// 128 -> 256 @32 -> 16.)  Plain loads only: hipcc counts them (vmcnt) itself.
constexpr int W8NT = 512;
// timing experiments only (experiments/abl_build.sh): 1 gy always from pixel 0 (cache-resident), 2 x patch loaded once,
// 4 patch split + LDS store once, 8 no MFMAs, 16 patch fragments read once
#ifndef VG_WX_ABL
#define VG_WX_ABL 0
#endif

template <int S_, int NP_, int WCO_, int TH_, bool F16_ = false>
struct W8 {
  static constexpr int S = S_, NP = NP_, WCO = WCO_, WN = 8 / WCO_, FP = 4 / WN, TCO = 32 * WCO_;
  static constexpr bool F16 = F16_;
  static constexpr int TH = TH_, NPIX = TH_ * WTW;                                // output pixels of a chunk: TH rows x 8
  // gy prefetch distance in pixels (register sets - 1): a pixel step is FP * 6 (or 3) MFMAs per wavefront, i.e. 0.35 /
  // 0.7 us, against ~2 us for a first touch of gy from HBM; the 2-fragment tile has the registers for 8 sets
  static constexpr int PD = (FP <= 2) ? 7 : 3;
  static constexpr int PH = S * (TH - 1) + 5, PW = S * (WTW - 1) + 5;
  static constexpr int ROWU = PW + ((5 - PW % 16) + 16) % 16;                    // = 5 (mod 16)
  static constexpr int CIU = PH * ROWU + ((9 - (PH * ROWU) % 16) + 16) % 16;     // = 9 (mod 16)
  static constexpr int KBU = WCIT * CIU;                                          // units per (plane, k-block)
  static constexpr int BUFU = 2 * NP * KBU;                                       // units per patch buffer
  static constexpr int NUNIT = 2 * WCIT * PH * PW;          // staged units (k-block, ci, row, col): 8 images each
  static constexpr int NQ = cdiv(NUNIT, W8NT);
  // the patch of chunk + 1 is loaded at pixel 0 and written to the other buffer one unit per thread at a time, the
  // two wavefronts of a SIMD (w, w + 4) at different pixels, so that one's split arithmetic runs under the other's
  // MFMAs: wavefronts 0-3 at pixels ST0 + q * STEP, wavefronts 4-7 NQ * STEP later
  static constexpr int STEP = TH, ST0 = NPIX / 4;
  static_assert(ST0 + 2 * NQ * STEP <= NPIX, "store schedule");
  static_assert(NPIX % (PD + 1) == 0, "gy ring");
  static_assert(2 * BUFU * 16 <= 160 * 1024, "LDS");
};

template <class C>
__global__ __launch_bounds__(W8NT, 2) void conv5x5_wgrad_split8_kernel(WXArgs A) {
  constexpr int S = C::S, PH = C::PH, PW = C::PW, ROWU = C::ROWU, CIU = C::CIU, KBU = C::KBU, NQ = C::NQ;
  constexpr int NP = C::NP, FP = C::FP, BUFU = C::BUFU, PD = C::PD;
  constexpr bool F16 = C::F16;
  __shared__ f32x4 lds[2 * BUFU];          // [buffer][plane][k-block][ci][row][col] x 8 images
  const float x_scale = F16 ? f16_scale_of(*A.x_amax) : 1.f;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kb = lane >> 5, l32 = lane & 31;
  const int wm = wid % C::WCO, wn = wid / C::WCO;
  const int grp = wid >> 2;                  // which of the two wavefronts of its SIMD
  // This workgroup's share of the work: units [u_begin, u_end) of the (output tile, chunk) space, tile-major -- every
  // workgroup gets the same number of chunks whatever the tile count (a share that crosses a tile boundary is worked
  // off as two segments).  A.order places workgroups with neighbouring chunk ranges on one XCD (speed only).
  const int wg = A.order[blockIdx.x];
  const int u_begin = wg * A.upw, u_end = min(u_begin + A.upw, A.units);
  const int Cin = A.Cin, Cout = A.Cout, H = A.H, W = A.W, OW = A.OW, CoP = A.CoP;
  const int HW = H * W, P = A.OH * A.OW;
  int m0 = 0, ci0 = 0;                       // the current segment's output tile

  // ---- staging map: unit e = (k-block, ci, row, col) = 8 images of one input pixel, read straight from x (NCHW
  // fp32: 8 loads a channel-image apart), split into the NP planes in registers and written as NP 16-byte LDS units.
  // One packed descriptor per unit: row | col << 4 | ci << 9 | kb << 12 | present << 13.
  int desc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int e = tid + q * W8NT;
    const int col = e % PW;
    int t = e / PW;
    const int r = t % PH;
    t /= PH;
    const int ci = t % WCIT, kbs = min(t / WCIT, 1);
    desc[q] = r | (col << 4) | (ci << 9) | (kbs << 12) | ((e < C::NUNIT ? 1 : 0) << 13);
  }
  static_assert(PH <= 16 && PW <= 32 && WCIT <= 8, "descriptor fields");

  // chunk -> (image group, output row, first output column)
  auto chunk_pos = [&](int chunk, int& bg, int& oh, int& ow0) {
    bg = chunk / A.tiles_hw;
    const int sp = chunk % A.tiles_hw;
    oh = (sp / A.tiles_w) * C::TH;
    ow0 = (sp % A.tiles_w) * WTW;
  };
  float preg[NQ][8], psc[NQ], psh[NQ];
  unsigned pvalid = 0;
  const size_t chw = (size_t)Cin * HW;
  auto load_patch = [&](int chunk) {
    int bg, oh, ow0;
    chunk_pos(chunk, bg, oh, ow0);
    const int ih0 = S * oh - 2, iw0 = S * ow0 - 2;
    pvalid = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int d = desc[q];
      const int ih = ih0 + (d & 15), iw = iw0 + ((d >> 4) & 31);
      const int ci = (d >> 9) & 7, kbs = (d >> 12) & 1;
      const bool ok = ih >= 0 && ih < H && iw >= 0 && iw < W && (ci0 + ci) < Cin;   // channels past Cin: zeros
      pvalid |= ok ? (1u << q) : 0u;
      const int cc = min(ci0 + ci, Cin - 1);
      const unsigned off = (unsigned)(cc * HW + min(max(ih, 0), H - 1) * W + min(max(iw, 0), W - 1));
      psc[q] = A.in_scale ? A.in_scale[cc] : 1.f;
      psh[q] = A.in_scale ? A.in_shift[cc] : 0.f;
      const int b0 = bg * 16 + kbs * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float* xb = A.x + (size_t)min(b0 + j, A.B - 1) * chw;      // images past the batch: clamped, zeroed below
        preg[q][j] = xb[off];
      }
    }
  };
  auto store_unit = [&](int q, int buf, int chunk) {
    const int bgrp = chunk / A.tiles_hw;
    const int d = desc[q];
    const int kbs = (d >> 12) & 1;
    const bool live = (pvalid >> q) & 1u;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = fmaf(preg[q][j], psc[q], psh[q]);                    // the producing layer's BatchNorm ...
      t = act_slope(t, A.in_slope);                                  // ... and activation, on load
      t = F16 ? t * x_scale : t;
      v[j] = (live && (bgrp * 16 + kbs * 8 + j) < A.B) ? t : 0.f;    // padding pads the activated tensor
    }
    f32x4 pl[NP];
    split_planes16<NP, F16>(v, pl);
    const int dst = buf * BUFU + kbs * KBU + ((d >> 9) & 7) * CIU + (d & 15) * ROWU + ((d >> 4) & 31);
    if (d >> 13) {
#pragma unroll
      for (int p = 0; p < NP; ++p) lds[dst + p * 2 * KBU] = pl[p];
    }
  };
  auto store_patch = [&](int buf, int chunk) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) store_unit(q, buf, chunk);
  };
  static_assert(NQ <= 32, "validity mask");

  // ---- per-lane base of the patch operand
  int base_b[FP];
#pragma unroll
  for (int f = 0; f < FP; ++f) {
    const int n = min((wn * FP + f) * 32 + l32, WCIT * 25 - 1);     // columns 125..127: any valid unit (result unused)
    const int ci = n / 25, tap = n % 25;
    base_b[f] = kb * KBU + ci * CIU + (tap / 5) * ROWU + tap % 5;
  }
  const size_t gstep = (size_t)2 * NP * CoP;                           // gy units per pixel
  auto pix_off = [&](int k) -> size_t { return (size_t)(k >> 3) * OW + (k & 7); };   // pixel k of a chunk

  // A share that straddles two tiles is worked off second part first: its chunks 0 .. n2-1 of the second tile at steps
  // 0 .. n2-1, like every other workgroup's (see the chunk order below), then the end of the first tile.
  const int u_split = (u_begin / A.chunks + 1) * A.chunks;
#if defined(VG_WX_PLAIN_ORDER)
  const bool two = false;
#else
  const bool two = u_split < u_end && u_end <= u_split + A.chunks;
#endif
  int u = two ? u_split : u_begin, u_lim = u_end, pass = 0, t0 = 0;
  while (true) {
    if (u >= u_lim) {
      if (!two || pass) break;
      pass = 1;
      u = u_begin;
      u_lim = u_split;
      continue;
    }
    // ---- one segment: chunks [c_begin, c_end) of one output tile
    const int tile = u / A.chunks, c_begin = u - tile * A.chunks;
    const int c_end = min(A.chunks, c_begin + (u_lim - u));
    u += c_end - c_begin;
    m0 = (tile % A.mtiles) * C::TCO;
    ci0 = (tile / A.mtiles) * WCIT;
    const bf16x8* ga = A.gp + (size_t)kb * CoP + m0 + wm * 32 + l32;   // this lane's cout row, k-block

    f32x16 acc[FP];
#pragma unroll
    for (int f = 0; f < FP; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    // first pixel (in Gp) of a chunk, clamped to the last chunk of the segment (prefetches past the end are unused)
    auto chunk_pix = [&](int chunk) -> size_t {
      int bg, oh, ow0;
      chunk_pos(min(chunk, c_end - 1), bg, oh, ow0);
      return (size_t)bg * P + (size_t)oh * OW + ow0;
    };
    bf16x8 av[PD + 1][NP];      // gy fragments of pixels t .. t+PD, ring indexed by (pixel & PD) (PD + 1 = 4 or 8 sets)
    static_assert((PD == 3 || PD == 7) && WTW == 8, "ring indices below assume 4 or 8 sets, 8 pixels per row");
    auto load_a = [&](int set, size_t pix) {
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) av[set][pl] = ga[((VG_WX_ABL & 1) ? 0 : pix) * gstep + (size_t)pl * 2 * CoP];
    };
    // Order of the segment's chunks: a share starts at its chunk that is 0 (mod upw) and wraps, so that at step j EVERY
    // workgroup of the launch is at a chunk that is j (mod upw) (exactly for a share inside one tile).  The workgroups of
    // an XCD (A.order: neighbouring chunk ranges, all tiles) then read the same 2-6 chunks of gy at the same time --
    // about 1 MB in a 4 MB L2 -- where shares worked off from their own first chunk kept a window of upw..64 chunks
    // (6-12 MB) in flight and the L2 hit rate of the two big layers at 34-46 % (877 / 1085 MB from beyond L2 per
    // launch against 117 / 167 MB of operands: profiles/r03_hbm_traffic_by_kernel.json).  Speed only: the sum over the
    // chunks is taken in this order whatever the timing.
    const int nch = c_end - c_begin;
#if defined(VG_WX_PLAIN_ORDER)              // timing experiments: the previous order
    const int rot = 0;
#else
    // step t0 + j of the share is at a chunk that is t0 + j (mod upw), as far as the segment allows
    int rot = ((t0 - c_begin) % A.upw + A.upw) % A.upw;
    rot = rot < nch ? rot : 0;
    t0 += nch;
#endif
    auto chunk_at = [&](int j) -> int {
      int i = min(j, nch - 1) + rot;
      i -= (i >= nch) ? nch : 0;
      return c_begin + i;
    };
    load_patch(chunk_at(0));
    store_patch(0, chunk_at(0));
    size_t pcur = chunk_pix(chunk_at(0));
#pragma unroll
    for (int t = 0; t < PD; ++t) load_a(t, pcur + pix_off(t));
    __syncthreads();
    for (int j = 0; j < nch; ++j) {
      const int chn = chunk_at(j + 1);
      const int buf = j & 1;
      const bool more = (j + 1) < nch;
      const size_t pnxt = chunk_pix(chn);
#pragma unroll
      for (int t = 0; t < C::NPIX; ++t) {
        // gy fragments PD pixels ahead (the last PD pixels of a chunk fetch the first ones of the next chunk)
        load_a((t + PD) & PD, (t + PD < C::NPIX) ? pcur + pix_off(t + PD) : pnxt + pix_off(t + PD - C::NPIX));
        if (t == 0 && more && !(VG_WX_ABL & 2)) load_patch(chn);
        bf16x8 bv[FP][NP];
        if (!(VG_WX_ABL & 16) || (j == 0 && t == 0)) {
#pragma unroll
          for (int f = 0; f < FP; ++f)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
              bv[f][pl] = __builtin_bit_cast(
                  bf16x8, lds[buf * BUFU + base_b[f] + S * (t >> 3) * ROWU + S * (t & 7) + pl * 2 * KBU]);
        }
        // products with plane index sum < NP, smallest terms first, product-major
#pragma unroll
        for (int sum = NP - 1; sum >= 0; --sum)
#pragma unroll
          for (int pa = sum; pa >= 0; --pa)
#pragma unroll
            for (int f = 0; f < FP; ++f) {
              if constexpr ((VG_WX_ABL & 32) != 0) {
                // timing experiment: the same FLOPs as two v_mfma_f32_16x16x32 (numerically meaningless here), as
                // conv_ring.hip's VG_RING_ABL = 128
                f32x4 c0 = {acc[f][0], acc[f][1], acc[f][2], acc[f][3]}, c1 = {acc[f][4], acc[f][5], acc[f][6], acc[f][7]};
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[t & PD][pa], bv[f][sum - pa], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[t & PD][pa], bv[f][sum - pa], c1, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) { acc[f][r] = c0[r]; acc[f][4 + r] = c1[r]; }
              } else
              if (!(VG_WX_ABL & 8)) acc[f] = mfma_split16<F16>(av[t & PD][pa], bv[f][sum - pa], acc[f]);
              else acc[f][0] += (float)av[t & PD][pa][0] + (float)bv[f][sum - pa][0];
            }
        // one unit of the next patch per slot of the store schedule (the other buffer was last read a chunk ago).
        // (A branch-free form of this loop -- clamped re-loads instead of `more`, no wavefront groups, dump units for
        // unit-less lanes: three basic blocks per chunk instead of ~40 -- measured 3-20 % SLOWER: hipcc clusters the
        // MFMAs of the big block and the loads lose their run-ahead.)
        {
          const int rel = t - C::ST0;
          if (rel >= 0 && rel % C::STEP == 0 && rel / C::STEP < 2 * NQ) {
            const int q = (rel / C::STEP) % NQ, g = (rel / C::STEP) / NQ;
            if (more && grp == g && !(VG_WX_ABL & 4)) store_unit(q, buf ^ 1, chn);
          }
        }
      }
      __syncthreads();                                   // also frees both patch buffers for the next segment
      pcur = pnxt;
    }

    // ---- this segment's partial slab: ws[piece][co][ci*25 + tap], piece = position among the tile's workgroups
    const int piece = wg - (int)(((long)tile * A.chunks) / A.upw);
    float* wsb = A.ws + (size_t)piece * Cout * Cin * 25;
#pragma unroll
    for (int f = 0; f < FP; ++f) {
      const int n = (wn * FP + f) * 32 + l32;
      const bool nok = n < WCIT * 25 && (ci0 + n / 25) < Cin;
#pragma unroll
      for (int r16 = 0; r16 < 16; ++r16) {
        const int co = m0 + wm * 32 + acc_row(r16, lane);
        if (nok && co < Cout) wsb[((size_t)co * Cin + ci0) * 25 + n] = acc[f][r16];
      }
    }
  }
}

// dw[e] = sum of the pieces of e's output tile, in workgroup order (fixed: the result does not depend on timing)
__global__ __launch_bounds__(256) void wx_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int Cin,
                                                       int tco, int mtiles, int chunks, int upw, unsigned n, int accumulate,
                                                       const float* __restrict__ x_amax, const float* __restrict__ gy_amax) {
  const unsigned e = blockIdx.x * 256u + threadIdx.x;
  if (e >= n) return;
  // fp16 planes: the two operands' power-of-two scales are undone here, once per output (exact)
  const float ux = x_amax ? f16_unscale_of(*x_amax) : 1.f, ug = gy_amax ? f16_unscale_of(*gy_amax) : 1.f;
  const int row = Cin * 25, co = e / row, ci = (e - co * row) / 25;
  const long tile = (long)(ci / WCIT) * mtiles + co / tco;
  const int first = (int)((tile * chunks) / upw), last = (int)(((tile + 1) * chunks - 1) / upw);
  // whole groups of four pieces with their loads in flight together, then the tail; added in piece order either way
  const int cnt = last - first + 1;
  float sum = 0.f;
  int p = 0;
  for (; p + 4 <= cnt; p += 4) {
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = slabs[(size_t)(p + k) * n + e];
#pragma unroll
    for (int k = 0; k < 4; ++k) sum += v[k];
  }
  for (; p < cnt; ++p) sum += slabs[(size_t)p * n + e];
  dw[e] = sum * ux * ug + (accumulate ? dw[e] : 0.f);        // accumulate: see wgrad_reduce_kernel
}

struct XPlan {
  int BG, CoP, OH, OW, wco, th, mtiles, ntiles, tiles_w, tiles_hw, chunks, units, upw, wgs, pieces;
  size_t gp_bytes, slab_bytes;
};

VG_KNOB(int, g_wx_th, 0);     // tuning build: force the pixel rows of a chunk (0: chosen by the plan)

// Work split: units = output tiles x chunks, tile-major, dealt out in equal consecutive shares to at most one
// workgroup per CU -- the tile count (26 for Cin = 128) rarely divides 256, and whole K splits per tile left 9-19 % of
// the CUs idle on the benchmark's layers.  A tile's partial results ("pieces", one per workgroup that touched it) are
// summed in workgroup order by wx_reduce_kernel.
void plan_shares(XPlan& p, int th) {
  p.th = th;
  p.tiles_hw = p.tiles_w * (p.OH / th);
  p.chunks = p.BG * p.tiles_hw;
  p.units = p.mtiles * p.ntiles * p.chunks;
  p.upw = cdiv(p.units, 256);
  p.wgs = cdiv(p.units, p.upw);
}

bool make_xplan(int B, int Cin, int H, int W, int Cout, int S, int planes, XPlan& p) {
  p.OH = (H - 1) / S + 1;
  p.OW = (W - 1) / S + 1;
  if (p.OW % WTW) return false;                         // whole 1 x 8 pixel tiles only (caller falls back to fp32)
  p.BG = cdiv(B, 16);
  p.CoP = (Cout + 127) & ~127;
  p.wco = Cout > 128 ? 8 : 4;                           // wavefronts along cout: 256- or 128-row output tiles
  p.mtiles = cdiv(Cout, 32 * p.wco);
  p.ntiles = cdiv(Cin, WCIT);
  p.tiles_w = p.OW / WTW;
  if ((long)p.mtiles * p.ntiles * p.BG * p.tiles_w * p.OH > 0x3fffffffL) return false;
  // chunk = 16 images x (th x 8) output pixels: two rows whenever the output height allows it (half the barriers,
  // 30 % fewer staged patch units per pixel; measured 2-4 % faster than one row even where one row deals the finer,
  // better balanced shares)
  int th = (p.OH % 2 == 0) ? 2 : 1;
  if (g_wx_th == 1 || (g_wx_th == 2 && p.OH % 2 == 0)) th = g_wx_th;
  plan_shares(p, th);
  p.pieces = 0;
  for (long t = 0; t < (long)p.mtiles * p.ntiles; ++t) {
    const int first = (int)((t * p.chunks) / p.upw), last = (int)(((t + 1) * p.chunks - 1) / p.upw);
    p.pieces = last - first + 1 > p.pieces ? last - first + 1 : p.pieces;
  }
  p.gp_bytes = (size_t)p.BG * p.OH * p.OW * 2 * planes * p.CoP * 16;
  p.slab_bytes = (size_t)p.pieces * Cout * Cin * 25 * sizeof(float);
  return true;
}

// launch slot -> workgroup: workgroups sorted by where their share starts inside a tile (they read the same gy / x
// chunks as their neighbours in that order), dealt to the XCDs in runs (slot b runs on XCD b % 8)
void make_order(const XPlan& p, unsigned short* order) {
  std::vector<int> idx(p.wgs);
  for (int w = 0; w < p.wgs; ++w) idx[w] = w;
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) {
    return ((long)a * p.upw) % p.chunks < ((long)b * p.upw) % p.chunks;
  });
  // XCD k runs the launch slots k, k + 8, ...: it gets the k-th run of the sorted list (runs differ by one workgroup
  // when the count is not a multiple of 8)
  int start[9];
  start[0] = 0;
  for (int k = 0; k < 8; ++k) start[k + 1] = start[k] + (p.wgs - k + 7) / 8;
  for (int b = 0; b < p.wgs; ++b) order[b] = (unsigned short)idx[start[b & 7] + (b >> 3)];
}

template <class C>
int launch_wx(const WXArgs& A, long grid, hipStream_t st) {
  hipLaunchKernelGGL(conv5x5_wgrad_split8_kernel<C>, dim3((unsigned)grid), dim3(W8NT), 0, st, A);
  VG_CHECK_LAUNCH();
  return 0;
}

template <int S, int NP, bool F16 = false>
int launch_wx_by_cout(const WXArgs& A, int wco, int th, long grid, hipStream_t st) {
  if (th == 2) return wco == 8 ? launch_wx<W8<S, NP, 8, 2, F16>>(A, grid, st) : launch_wx<W8<S, NP, 4, 2, F16>>(A, grid, st);
  return wco == 8 ? launch_wx<W8<S, NP, 8, 1, F16>>(A, grid, st) : launch_wx<W8<S, NP, 4, 1, F16>>(A, grid, st);
}

}  // namespace

#ifdef VG_TUNING
void vg_internal_wx_set_th(int th) { g_wx_th = (th == 1 || th == 2) ? th : 0; }
#endif

extern "C" size_t vg_conv5x5_wgrad_bf16split_workspace_bytes(int B, int Cin, int H, int W, int Cout, int stride,
                                                          int planes) {
  if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2)) return 0;
  if (planes != 2 && planes != 3 && planes != (VG_PLANES_F16_FLAG | 2)) return 0;
  XPlan p;
  if (!make_xplan(B, Cin, H, W, Cout, stride, planes & 0xff, p)) return 0;     // 0: shape not supported by this mode
  return p.gp_bytes + p.slab_bytes;
}

extern "C" int vg_conv5x5_wgrad_bf16split(const float* x, const float* gy, float* dw, int B, int Cin, int H, int W,
                                       int Cout, int stride, int planes, void* workspace, size_t workspace_bytes,
                                       const float* in_scale, const float* in_shift, int in_act, int affine_on_gy,
                                       const float* x_amax, const float* gy_amax, int accumulate, void* stream) {
  if (!x || !gy || !dw || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return VG_ERR_BAD_ARG;
  const bool f16 = (planes & VG_PLANES_F16_FLAG) != 0;
  planes &= 0xff;
  if (f16 && (planes != 2 || !x_amax || !gy_amax)) return VG_ERR_BAD_ARG;
  if (!f16) x_amax = gy_amax = nullptr;
  if ((in_scale == nullptr) != (in_shift == nullptr) || in_act < VG_ACT_NONE || in_act > VG_ACT_LRELU) return VG_ERR_BAD_ARG;
  const float slope = (!in_scale || in_act == VG_ACT_NONE) ? 1.f : (in_act == VG_ACT_RELU ? 0.f : 0.2f);
  const float* gsc = affine_on_gy ? in_scale : nullptr;      // coefficients per channel of gy (Cout of them) ...
  const float* gsh = affine_on_gy ? in_shift : nullptr;
  if (affine_on_gy) in_scale = in_shift = nullptr;           // ... or per channel of x (Cin)
  if ((stride != 1 && stride != 2) || (planes != 2 && planes != 3)) return VG_ERR_BAD_ARG;
  XPlan p;
  if (!make_xplan(B, Cin, H, W, Cout, stride, planes, p)) return VG_ERR_BAD_ARG;
  if (!workspace || ((uintptr_t)workspace & 15) || workspace_bytes < p.gp_bytes + p.slab_bytes)
    return VG_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  bf16x8* gp = (bf16x8*)workspace;
  float* slabs = (float*)((char*)workspace + p.gp_bytes);
  const int P = p.OH * p.OW;
  if ((size_t)Cin * H * W * 4 > 0xffffffffUL || p.BG * 2 > 65535 || p.CoP / 32 > 65535) return VG_ERR_BAD_ARG;
  const dim3 gg(cdiv(P, 32), p.CoP / 32, p.BG * 2);
  if (f16) hipLaunchKernelGGL((relayout_gy_kernel<2, true>), gg, dim3(256), 0, st, gy, gp, B, Cout, p.CoP, P, gsc, gsh, slope, gy_amax);
  else if (planes == 2) hipLaunchKernelGGL((relayout_gy_kernel<2, false>), gg, dim3(256), 0, st, gy, gp, B, Cout, p.CoP, P, gsc, gsh, slope, gy_amax);
  else hipLaunchKernelGGL((relayout_gy_kernel<3, false>), gg, dim3(256), 0, st, gy, gp, B, Cout, p.CoP, P, gsc, gsh, slope, gy_amax);
  VG_CHECK_LAUNCH();
  WXArgs A;
  A.x = x; A.B = B; A.gp = gp; A.ws = slabs; A.x_amax = x_amax;
  A.in_scale = in_scale; A.in_shift = in_shift;
  A.in_slope = in_scale ? slope : 1.f;
  A.Cin = Cin; A.H = H; A.W = W; A.Cout = Cout; A.CoP = p.CoP; A.OH = p.OH; A.OW = p.OW;
  A.mtiles = p.mtiles; A.ntiles = p.ntiles; A.tiles_w = p.tiles_w; A.tiles_hw = p.tiles_hw;
  A.chunks = p.chunks; A.units = p.units; A.upw = p.upw;
  make_order(p, A.order);
  const long grid = p.wgs;
  int rc;
  if (f16) rc = (stride == 2) ? launch_wx_by_cout<2, 2, true>(A, p.wco, p.th, grid, st) : launch_wx_by_cout<1, 2, true>(A, p.wco, p.th, grid, st);
  else if (planes == 2) rc = (stride == 2) ? launch_wx_by_cout<2, 2>(A, p.wco, p.th, grid, st) : launch_wx_by_cout<1, 2>(A, p.wco, p.th, grid, st);
  else rc = (stride == 2) ? launch_wx_by_cout<2, 3>(A, p.wco, p.th, grid, st) : launch_wx_by_cout<1, 3>(A, p.wco, p.th, grid, st);
  if (rc) return rc;
  const unsigned n = (unsigned)Cout * Cin * 25;
  hipLaunchKernelGGL(wx_reduce_kernel, dim3(cdiv(n, 256u)), dim3(256), 0, st, slabs, dw, Cin, 32 * p.wco, p.mtiles,
                     p.chunks, p.upw, n, accumulate ? 1 : 0, x_amax, gy_amax);
  VG_CHECK_LAUNCH();
  return 0;
}
