// Weight gradient of a 5x5 convolution with <= 3 input channels in the split-bf16 arithmetic for gfx950: convs.0 of the
// discriminator (Conv2d(3, 32, 5, 1, 2), /root/reference/models/model.py:389), features.0 of the encoder
// (Conv2d(3, 64, 5, 2, 2), model.py:450) and -- roles swapped -- the decoder's deconv4 (ConvTranspose2d(32, 3, 5, 1, 2),
// model.py:507).  One read pass over gy (67 MB at B = 128 for the 32-channel 64 x 64 tensor); x is tiny.
//
//   dw[co][ci][kh][kw] = sum_{b,oh,ow} gy[b][co][oh][ow] * x[b][ci][S*oh+kh-2][S*ow+kw-2]
//
// GEMM view: D[co][n = ci*25 + kh*5 + kw] (32 x 75 per 32-channel group), reduction over pixels; one MFMA step of
// v_mfma_f32_32x32x16_bf16 = 16 consecutive pixels of an output row, three 32-column tiles (75 of 96 used).
//   * A (gy): global -> registers directly in fragment layout (lane = channel, 8 consecutive pixels = 32 aligned
//     bytes), one step ahead; split into planes in registers (each gy element is used by this workgroup only).
//   * B (x patches): 25 taps read every input pixel, so the band of x is split ONCE per workgroup into bf16 planes and
//     stored in LDS as five copies, one per kw, each shifted so that the 8 pixels a lane needs -- x[S*(8g+j)+kw-2],
//     j = 0..7 -- are one aligned 16-byte unit: a B fragment is one ds_read_b128 per plane, no VALU work.
//   * a workgroup (8 wavefronts) takes bands of output rows of one image; its wavefronts share the band's steps;
//     partial D tiles are summed across the wavefronts through LDS in a fixed order into one slab per workgroup, and
//     conv_wgrad.hip's reduction sums the slabs in order (deterministic).
// Arithmetic: operands split exactly into NP bf16 planes, plane products with index sum < NP, fp32 accumulate (NP = 3:
// fp32-equivalent, as conv_ring.hip).
#include "common.hpp"
#include "vaegan_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

constexpr int WNT = 512;
// timing experiments only (experiments/abl_build.sh twg <bits>): 1 every lane reads gy of channel 0 (one cache line per
// load), 2 plane copies built for the first band only, 4 no MFMAs
#ifndef VG_TWG_ABL
#define VG_TWG_ABL 0
#endif

struct TWArgs {
  // the gy operand read as act(gy * gy_scale[c] + gy_shift[c]) -- the weight gradient of deconv4 passes the layer's INPUT
  // in this slot, whose producer's BatchNorm + ReLU is applied here instead of being materialised (NULL: plain)
  const float* gy_scale;
  const float* gy_shift;
  float gy_slope;
  const float* x;
  const float* gy;
  float* slabs;        // [workgroup][Cout][Cin * 25]
  int B, Cin, H, W, Cout, OH, OW;
  int rb, rbi, ng, kg;         // band: output rows, staged input rows, 8-pixel groups per row, 16-pixel steps per row
  int bands, units_per_wg, total_units;       // unit = (image, band)
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

// MT: 32-channel groups of gy handled together (they share the B fragments)
template <int S, int MT, int NP>
__global__ __launch_bounds__(WNT) void conv_thin_wgrad_kernel(TWArgs A) {
  extern __shared__ f32x4 cp[];                        // [kw 5][ci 3][row rbi][group ng][plane NP] x 8 pixels (bf16)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;
  const int H = A.H, W = A.W, OW = A.OW, OH = A.OH, Cin = A.Cin, Cout = A.Cout;
  const int ng = A.ng, rbi = A.rbi;
  const int row_units = ng * NP, ci_units = rbi * row_units, kw_units = 3 * ci_units;

  const bool aff = A.gy_scale != nullptr;
  float a_sc[MT], a_sh[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int co = min(m * 32 + l32, Cout - 1);
    a_sc[m] = aff ? A.gy_scale[co] : 1.f;
    a_sh[m] = aff ? A.gy_shift[co] : 0.f;
  }

  // ---- per-lane B bases: column n = 32 t + l32 = (ci, kh, kw); columns past Cin * 25 read column 0 (weights unused)
  int bbase[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    int n = 32 * t + l32;
    if (n >= Cin * 25) n = 0;
    const int ci = n / 25, kh = (n % 25) / 5, kw = n % 5;
    bbase[t] = kw * kw_units + ci * ci_units + kh * row_units + half * NP;
  }

  f32x16 acc[MT][3];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;

  const float inv_ng = 1.f / (float)ng, inv_kg = 1.f / (float)A.kg;
  const int u_begin = blockIdx.x * A.units_per_wg, u_end = min(u_begin + A.units_per_wg, A.total_units);
  for (int u = u_begin; u < u_end; ++u) {
    const int b = u / A.bands, r0 = (u % A.bands) * A.rb;
    const int rows = min(A.rb, OH - r0);
    __syncthreads();                                   // the previous band's fragments are consumed
    // ---- the band of x: first the raw rows, coalesced, into LDS (one global round trip; read straight from global,
    // item by item, the copies cost five dependent round trips per band) ...
    float* xraw = reinterpret_cast<float*>(cp + 5 * kw_units);        // [ci 3][row rbi][W + 4], zero halo
    const int PWX = W + 4;
    {
      const int ih0 = S * r0 - 2, per_c = rbi * PWX, total = 3 * per_c;
      const float inv_pwx = 1.f / (float)PWX;
      for (int e0 = tid; e0 < total; e0 += 4 * WNT) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int e = e0 + k * WNT;
          const int rr = (int)(((float)e + 0.5f) * inv_pwx), col = e - rr * PWX;      // rr = ci * rbi + row (exact: e < 2^16)
          const int ci = (rr >= 2 * rbi) ? 2 : (rr >= rbi ? 1 : 0), r = rr - ci * rbi;
          const int ih = ih0 + r, iw = col - 2;
          const bool ok = e < total && ci < Cin && ih >= 0 && ih < H && iw >= 0 && iw < W;
          const float t = A.x[(((size_t)b * Cin + min(ci, Cin - 1)) * H + min(max(ih, 0), H - 1)) * W + min(max(iw, 0), W - 1)];
          v[k] = ok ? t : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (e0 + k * WNT < total) xraw[e0 + k * WNT] = v[k];
      }
    }
    __syncthreads();
    // ... then the five shifted plane copies.  Item = (kw, ci, input row, group): 8 pixels S apart.
    if (!(VG_TWG_ABL & 2) || u == u_begin) {
      const int per_kc = rbi * ng, items = 15 * per_kc;
      const float inv_kc = 1.f / (float)per_kc;
      for (int e = tid; e < items; e += WNT) {
        const int kc = (int)(((float)e + 0.5f) * inv_kc);              // kw * 3 + ci  (items < 2^16: exact)
        const int rg = e - kc * per_kc;
        const int r = (int)(((float)rg + 0.5f) * inv_ng), g = rg - r * ng;
        const int kw = kc / 3, ci = kc - 3 * kw;
        const float* xr = xraw + (ci * rbi + r) * PWX + S * 8 * g + kw;      // column of iw = S*8g + kw - 2
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = xr[S * j];
        bf16x8 pl[NP];
        split_frag<NP>(v, pl);
        f32x4* dst = cp + kw * kw_units + ci * ci_units + r * row_units + g * NP;
#pragma unroll
        for (int p = 0; p < NP; ++p) dst[p] = __builtin_bit_cast(f32x4, pl[p]);
      }
    }
    __syncthreads();

    // ---- this wavefront's steps of the band: step = (row, 16-pixel group)
    const int nsteps = rows * A.kg;
    const float* gyb = A.gy + (size_t)b * Cout * OH * OW;
    // gy fragments PF - 1 steps ahead (a step is ~0.35 us of MFMAs, a first touch of gy ~2 us); ring slot = step % PF,
    // compile-time in the PF-unrolled loop
    constexpr int PF = (MT == 1) ? 6 : 3;
    float raw[PF][MT][8];
    auto load_a = [&](int slot, int i) {               // i: index among this wavefront's steps
      const int st = wid + 8 * i;
      const int row = (int)(((float)st + 0.5f) * inv_kg), k16 = st - row * A.kg;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int co = (VG_TWG_ABL & 1) ? 0 : min(m * 32 + l32, Cout - 1);    // channels past Cout: clamped (their rows are not stored)
        const float* p = gyb + ((size_t)co * OH + r0 + row) * OW + k16 * 16 + half * 8;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
        raw[slot][m][0] = lo[0]; raw[slot][m][1] = lo[1]; raw[slot][m][2] = lo[2]; raw[slot][m][3] = lo[3];
        raw[slot][m][4] = hi[0]; raw[slot][m][5] = hi[1]; raw[slot][m][6] = hi[2]; raw[slot][m][7] = hi[3];
      }
    };
    auto step = [&](int slot, int i) {
      const int st = wid + 8 * i;
      const int row = (int)(((float)st + 0.5f) * inv_kg), k16 = st - row * A.kg;
      bf16x8 af[MT][NP];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if (aff) {                                     // wave-uniform: a lane's channel is fixed, its coefficients are loaded once
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float v = fmaf(raw[slot][m][j], a_sc[m], a_sh[m]);
            raw[slot][m][j] = act_slope(v, A.gy_slope);
          }
        }
        split_frag<NP>(raw[slot][m], af[m]);
      }
      const int boff = (S * row) * row_units + (2 * k16) * NP;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        bf16x8 bf[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) bf[p] = __builtin_bit_cast(bf16x8, cp[bbase[t] + boff + p]);
#pragma unroll
        for (int sum = NP - 1; sum >= 0; --sum)
#pragma unroll
          for (int pa = sum; pa >= 0; --pa)
#pragma unroll
            for (int m = 0; m < MT; ++m)
              if (!(VG_TWG_ABL & 4)) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][pa], bf[sum - pa], acc[m][t], 0, 0, 0);
              else acc[m][t][0] += (float)af[m][pa][0] + (float)bf[sum - pa][0];
      }
    };
    const int nmine = (nsteps > wid) ? (nsteps - wid + 7) / 8 : 0;     // steps wid, wid + 8, ...
#pragma unroll
    for (int j = 0; j < PF - 1; ++j)
      if (j < nmine) load_a(j, j);
    for (int i0 = 0; i0 < nmine; i0 += PF) {
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const int i = i0 + j;
        if (i < nmine) {
          if (i + PF - 1 < nmine) load_a((j + PF - 1) % PF, i + PF - 1);
          step(j, i);
        }
      }
    }
  }

  // ---- sum the wavefronts' tiles in wavefront order, one tile at a time through LDS, into this workgroup's slab
  float* red = reinterpret_cast<float*>(cp);           // [8 wavefronts][32 x 32]
  float* slab = A.slabs + (size_t)blockIdx.x * Cout * Cin * 25;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rowi = (r & 3) + 8 * (r >> 2) + 4 * half;             // channel within the group
        red[(wid * 32 + rowi) * 32 + l32] = acc[m][t][r];
      }
      __syncthreads();
      for (int e = tid; e < 1024; e += WNT) {
        const int rowi = e >> 5, col = e & 31;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) s += red[(w * 32 + rowi) * 32 + col];
        const int co = m * 32 + rowi, n = 32 * t + col;
        if (co < Cout && n < Cin * 25) slab[(size_t)co * Cin * 25 + n] = s;
      }
    }
}

struct TWPlan {
  int OH, OW, rb, rbi, ng, kg, bands, mt, wgs, upw, total;
  size_t lds, slab_bytes;
};

bool make_twplan(int B, int Cin, int H, int W, int Cout, int S, int planes, TWPlan& p) {
  if (B < 1 || Cin < 1 || Cin > 3 || Cout < 1 || Cout > 64 || H < 1 || W < 1 || (S != 1 && S != 2)) return false;
  if (planes != 2 && planes != 3) return false;
  p.OH = (H - 1) / S + 1;
  p.OW = (W - 1) / S + 1;
  if (p.OW % 16 || (S == 2 && (W % 2))) return false;
  p.ng = p.OW / 8;
  p.kg = p.OW / 16;
  p.mt = cdiv(Cout, 32);
  // band height: as many output rows as keep the five plane copies under 120 KB
  const size_t per_row = (size_t)15 * p.ng * planes * 16;
  int rb = 32;
  while (rb > 1 && per_row * (S * (rb - 1) + 5) > 120 * 1024) rb >>= 1;
  if (per_row * (S * (rb - 1) + 5) > 120 * 1024) return false;
  if (rb > p.OH) rb = p.OH;
  p.rb = rb;
  p.rbi = S * (rb - 1) + 5;
  p.bands = cdiv(p.OH, rb);
  p.total = B * p.bands;
  p.upw = cdiv(p.total, 256);
  p.wgs = cdiv(p.total, p.upw);
  const size_t red = (size_t)8 * 32 * 32 * sizeof(float);
  const size_t staged = per_row * p.rbi + (size_t)3 * p.rbi * (W + 4) * sizeof(float);       // plane copies + raw rows
  p.lds = staged > red ? staged : red;
  if (p.lds > 156 * 1024) return false;
  p.slab_bytes = (size_t)p.wgs * Cout * Cin * 25 * sizeof(float);
  return (size_t)15 * p.rbi * p.ng < 65536 && (size_t)3 * p.rbi * (W + 4) < 65536 && (size_t)Cout * p.OH * p.OW < ((size_t)1 << 31);
}

template <int S, int MT>
int launch_tw(const TWArgs& A, const TWPlan& p, int planes, hipStream_t st) {
  if (planes == 3) hipLaunchKernelGGL((conv_thin_wgrad_kernel<S, MT, 3>), dim3(p.wgs), dim3(WNT), p.lds, st, A);
  else hipLaunchKernelGGL((conv_thin_wgrad_kernel<S, MT, 2>), dim3(p.wgs), dim3(WNT), p.lds, st, A);
  VG_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" size_t vg_conv5x5_thin_wgrad_bf16split_workspace_bytes(int B, int Cin, int H, int W, int Cout, int stride,
                                                                  int planes) {
  TWPlan p;
  return make_twplan(B, Cin, H, W, Cout, stride, planes, p) ? p.slab_bytes : 0;       // 0: shape not taken
}

extern "C" int vg_conv5x5_thin_wgrad_bf16split(const float* x, const float* gy, float* dw, int B, int Cin, int H, int W,
                                               int Cout, int stride, int planes, void* workspace, size_t workspace_bytes,
                                               const float* gy_scale, const float* gy_shift, int gy_act, int accumulate,
                                               void* stream) {
  if (!x || !gy || !dw) return VG_ERR_BAD_ARG;
  if ((gy_scale == nullptr) != (gy_shift == nullptr) || gy_act < VG_ACT_NONE || gy_act > VG_ACT_LRELU) return VG_ERR_BAD_ARG;
  TWPlan p;
  if (!make_twplan(B, Cin, H, W, Cout, stride, planes, p)) return VG_ERR_BAD_ARG;
  if (!workspace || ((uintptr_t)workspace & 15) || workspace_bytes < p.slab_bytes) return VG_ERR_WORKSPACE;
  if (((uintptr_t)gy & 15) != 0) return VG_ERR_BAD_ARG;
  TWArgs A;
  A.x = x; A.gy = gy; A.slabs = (float*)workspace;
  A.gy_scale = gy_scale; A.gy_shift = gy_shift;
  A.gy_slope = (!gy_scale || gy_act == VG_ACT_NONE) ? 1.f : (gy_act == VG_ACT_RELU ? 0.f : 0.2f);
  A.B = B; A.Cin = Cin; A.H = H; A.W = W; A.Cout = Cout; A.OH = p.OH; A.OW = p.OW;
  A.rb = p.rb; A.rbi = p.rbi; A.ng = p.ng; A.kg = p.kg; A.bands = p.bands; A.units_per_wg = p.upw; A.total_units = p.total;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (stride == 1) rc = p.mt == 1 ? launch_tw<1, 1>(A, p, planes, st) : launch_tw<1, 2>(A, p, planes, st);
  else rc = p.mt == 1 ? launch_tw<2, 1>(A, p, planes, st) : launch_tw<2, 2>(A, p, planes, st);
  if (rc) return rc;
  return vg_internal_wgrad_reduce(A.slabs, dw, Cout * Cin * 25, p.wgs, st, accumulate ? 1 : 0);
}
