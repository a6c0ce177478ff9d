// 5x5 convolution from <= 3 input channels in the split-bf16 arithmetic for gfx950: the first layers of the
// discriminator (convs.0: Conv2d(3, 32, 5, 1, 2), /root/reference/models/model.py:389) and of the encoder (features.0:
// Conv2d(3, 64, 5, 2, 2), model.py:450).  The input is tiny (6 MB at B = 128, 64 x 64); the kernel is one write pass
// over the output (67 MB for convs.0) with the next BatchNorm's statistics taken on the way out.
//
//   y[b][co][oh][ow] = bias[co] + sum_{ci,kh,kw} x[b][ci][S*oh+kh-2][S*ow+kw-2] * w[co][ci][kh][kw]
//
// GEMM view per filter row kh: D[co][px] += A_kh[co][k] * B_kh[k][px] with k = (ci, kw): 15 of the 16 reduction
// indices of one v_mfma_f32_32x32x16_bf16, 32 output channels x 32 consecutive pixels of an output row per wavefront;
// the five kh are five MFMA steps into the same accumulator.
//   * A (the filter) lives in registers for the whole kernel: [kh][plane] fragments, 60 VGPRs for 3 planes.
//   * B: the workgroup stages its band of the input (3 channels x (S*RB + 3) rows, zero halo) in LDS once; a
//     wavefront marches down the output rows of its 32-pixel strip and builds the fragment of each INPUT row once
//     (8 ds_read_b32 + the split into planes), keeping the last five (S = 1) or six (S = 2) rows in a register
//     ring -- every input row serves up to five output rows.
//     The order of the reduction index is chosen so that both lane halves run the same code: k-block h holds a run of
//     five (channel 0 or 2, kw = 0..4) and a run of three (channel 1, kw = 0..2 or 3..5, the sixth with zero weight).
//   * epilogue per output row: bias, 16 stores of 128 contiguous bytes per half wavefront; per-lane running sums of
//     y and y^2, reduced over the pixels once per band into the statistics slot of the wavefront
//     (vg_conv_fusion.stats layout [slot][Cout][2], consumed by vg_bn_finalize_stats).
// No barrier after the staging one.  Arithmetic: operands split exactly into NP bf16 planes, plane products with index
// sum < NP, fp32 accumulate (NP = 3: fp32-equivalent, as conv_ring.hip).
#include "common.hpp"
#include "vaegan_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// timing experiments only (experiments/abl_build.sh tfwd <bits>): 1 no output stores, 2 no MFMAs, 4 input rows built once,
// 8 no statistics
#ifndef VG_TF_ABL
#define VG_TF_ABL 0
#endif

struct FArgs {
  const float* x;
  const float* w;
  const float* bias;
  float* y;
  float* stats;        // [slot][Cout][2] or NULL
  int B, Cin, H, W, Cout, OH, OW, bands, strips;
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

template <int S>
struct FCfg {
  static constexpr int RB = (S == 1) ? 8 : 4;          // output rows per workgroup
  static constexpr int RS = (S == 1) ? 5 : 6;          // register ring of input rows
  static constexpr int UNR = RS / S;                   // steps per unrolled block (ring slots compile-time)
  static constexpr int PR = S * (RB - 1) + 5;          // staged input rows
};

template <int S, int NP>
__global__ __launch_bounds__(512) void conv_thin_fwd_kernel(FArgs A) {
  using C = FCfg<S>;
  extern __shared__ float patch[];                     // [3 channels][PR rows][W + 4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;
  const int strip = wid % A.strips, cg = wid / A.strips;       // 32-pixel strip, 32-channel group
  const int H = A.H, W = A.W, OW = A.OW, Cin = A.Cin, Cout = A.Cout;
  const int b = blockIdx.x / A.bands, band = blockIdx.x % A.bands, r0 = band * C::RB;
  const int PW = W + 4;

  // ---- stage the band of x (zero halo, zero for missing channels): a wavefront takes whole patch rows, 8 loads in
  // flight per lane (one element per round trip made this prologue longer than the rest of the kernel)
  {
    const int ih0 = S * r0 - 2, nw = blockDim.x >> 6;
    const int chunks = (PW + 63) >> 6;                 // 64-column pieces of a patch row
    // a wavefront takes whole patch rows (channel c, row r), four rows' loads in flight per 64-column piece
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* xc = A.x + ((size_t)b * Cin + min(c, Cin - 1)) * H * W;
      float* pc = patch + c * C::PR * PW;
      for (int rq = wid; rq < C::PR; rq += 4 * nw) {
        for (int ch = 0; ch < chunks; ++ch) {
          const int col = ch * 64 + lane, iw = col - 2;
          const bool cok = c < Cin && iw >= 0 && iw < W;
          const int iwc = min(max(iw, 0), W - 1);
          float v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int ih = ih0 + rq + k * nw;
            v[k] = xc[min(max(ih, 0), H - 1) * W + iwc];            // clamped, unconditional; masked below
            v[k] = (cok && ih >= 0 && ih < H) ? v[k] : 0.f;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (rq + k * nw < C::PR && col < PW) pc[(rq + k * nw) * PW + col] = v[k];
        }
      }
    }
  }

  // ---- the filter and the bias go through LDS: read straight from global, a lane's 40 taps sit 75 floats from its
  // neighbour's -- 64 cache lines per load instruction, 2560 per wavefront, and the L1 (one line per cycle) made the
  // prologue longer than the rest of the kernel.  Staged with contiguous loads the same reads cost LDS cycles only
  // (stride 75 = 11 mod 64 banks: conflict-free).
  const int ngroups = (int)(blockDim.x >> 6) / A.strips;
  float* w_l = patch + 3 * C::PR * PW;                 // [channel of the workgroup][Cin * 25]
  float* bias_l = w_l + ngroups * 32 * 75;             // [channel of the workgroup]
  {
    const int nw = min(ngroups * 32, Cout) * Cin * 25;                 // the first min(...) channels: contiguous in w
    for (int e0 = tid; e0 < nw; e0 += 8 * (int)blockDim.x) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = (e0 + k * (int)blockDim.x < nw) ? A.w[e0 + k * (int)blockDim.x] : 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (e0 + k * (int)blockDim.x < nw) w_l[e0 + k * (int)blockDim.x] = v[k];
    }
    if (tid < ngroups * 32) bias_l[tid] = (A.bias && tid < Cout) ? A.bias[tid] : 0.f;
  }
  __syncthreads();

  // ---- filter fragments A[co][k] of every kh.  k-block h, element j: j < 5 -> (channel 2h, kw = j);
  // j >= 5 -> (channel 1, kw = 3h + j - 5), kw = 5 does not exist: zero.  Accumulator register r of this lane is channel
  // (r & 3) + 8 * (r >> 2) + 4 * half of the group.
  const int c5 = 2 * half, o3 = 3 * half;
  const int co_l = cg * 32 + l32;
  bf16x8 wf[5][NP];
  {
    // two per-lane bases, compile-time tap offsets: the reads are ds_read_b32 with immediates
    const float* wl5 = w_l + (min(co_l, Cout - 1) * Cin + min(c5, Cin - 1)) * 25;
    const float* wl3 = w_l + (min(co_l, Cout - 1) * Cin + min(1, Cin - 1)) * 25 + o3;
    const bool ok5 = c5 < Cin && co_l < Cout, ok3 = 1 < Cin && co_l < Cout;
#pragma unroll
    for (int kh = 0; kh < 5; ++kh) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const float t = wl5[kh * 5 + j];               // unconditional read, then select: hipcc does not speculate a
        v[j] = ok5 ? t : 0.f;                          // load under a condition, it branches around it
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float t = wl3[kh * 5 + j];               // kw = 5 (half 1, j = 2): masked, the read stays inside LDS
        v[5 + j] = (ok3 && !(half && j == 2)) ? t : 0.f;
      }
      split_frag<NP>(v, wf[kh]);
    }
  }

  // ---- B fragment of staged input row ir: lane = pixel l32 of the strip; run of 5 from channel c5, run of 3 from channel 1
  const int pcol = S * (strip * 32 + l32);             // patch column of kw = 0
  const float* p5 = patch + (size_t)c5 * C::PR * PW + pcol;
  const float* p3 = patch + (size_t)1 * C::PR * PW + pcol + o3;    // o3 + 2 <= 5 <= PW - pcol - ... (last pixel: col W+3 max)
  bf16x8 win[C::RS][NP];
  auto build_row = [&](int slot, int ir) {
    float v[8];
    if (ir < C::PR) {
#pragma unroll
      for (int j = 0; j < 5; ++j) v[j] = p5[ir * PW + j];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float t = p3[ir * PW + j];               // kw = 5 (half 1, j = 2): past the window, masked
        v[5 + j] = (half && j == 2) ? 0.f : t;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
    split_frag<NP>(v, win[slot]);
  };

  float s1[16], s2[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) s1[r] = s2[r] = 0.f;
  const int ow = strip * 32 + l32;
  const int rows = min(C::RB, A.OH - r0);
  const unsigned plane_byte = (unsigned)(A.OH * OW) * 4u;                       // the plan keeps 32 planes below 2^32 bytes
  const unsigned lane_byte = (unsigned)(4 * half) * plane_byte + (unsigned)ow * 4u;

  // step t finishes output row r0 + t from staged input rows S*t .. S*t + 4 (ring slot = row % RS)
#pragma unroll
  for (int ir = 0; ir < 5 - S; ++ir) build_row(ir, ir);
  for (int t0 = 0; t0 < rows; t0 += C::UNR) {
#pragma unroll
    for (int u = 0; u < C::UNR; ++u) {
      const int t = t0 + u;
      if (t < rows) {                                  // uniform
#pragma unroll
        for (int n = 0; n < S; ++n)
          if (!(VG_TF_ABL & 4) || t == 0) build_row((S * u + 5 - S + n) % C::RS, S * t + 5 - S + n);
        // two accumulators taken in turn: an MFMA that accumulates into the result of the one issued just before it
        // waits out that one's whole latency in compiler-inserted s_nops (one chain: 4158 SALU instructions per
        // wavefront, three quarters of the kernel's time); products in order of size, smallest plane products first
        f32x16 d, d1;
#pragma unroll
        for (int r = 0; r < 16; ++r) d[r] = d1[r] = 0.f;
        {
          int n = 0;
#pragma unroll
          for (int sum = NP - 1; sum >= 0; --sum)
#pragma unroll
            for (int pa = sum; pa >= 0; --pa)
#pragma unroll
              for (int kh = 0; kh < 5; ++kh, ++n) {
                if (VG_TF_ABL & 2) {
                  d[kh] += (float)wf[kh][pa][0] * (float)win[(S * u + kh) % C::RS][sum - pa][0];
                } else if (n & 1) {
                  d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kh][pa], win[(S * u + kh) % C::RS][sum - pa], d1, 0, 0, 0);
                } else {
                  d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kh][pa], win[(S * u + kh) % C::RS][sum - pa], d, 0, 0, 0);
                }
              }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) d[r] += d1[r];
        // row (oh) of the 32 channels: uniform base + per-lane offset (4 * half channels, pixel) + uniform channel step.
        // Channels past Cout have zero weights and bias (v = 0): only their stores need masking, and only in a group
        // that is not full (uniform test -- a per-store exec branch costs more than the store)
        // address = uniform row base (SGPR pair) + 32-bit per-lane byte offset: one v_add per store (with 64-bit
        // per-store addresses the scalar unit spilled its registers into VGPR lanes)
        char* yrow = (char*)(A.y + (((size_t)b * Cout + cg * 32) * A.OH + (r0 + t)) * OW);
        const bool full = (cg + 1) * 32 <= Cout;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          d[r] += bias_l[cg * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
          if (!(VG_TF_ABL & 8)) {
            s1[r] += d[r];
            s2[r] += d[r] * d[r];
          }
        }
        if (!(VG_TF_ABL & 1)) {
          // opaque per row: left visible, hipcc hoists the 16 zero-extended 64-bit offsets out of the row loop (32
          // VGPRs), spills, and waits for every outstanding store (vmcnt(0)) on each reload
          unsigned lb = lane_byte;
          asm volatile("" : "+v"(lb));
          if (full) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              *(float*)(yrow + (lb + (unsigned)((r & 3) + 8 * (r >> 2)) * plane_byte)) = d[r];
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (cg * 32 + (r & 3) + 8 * (r >> 2) + 4 * half < Cout)
                *(float*)(yrow + (lb + (unsigned)((r & 3) + 8 * (r >> 2)) * plane_byte)) = d[r];
          }
        }
      }
    }
  }

  // ---- statistics of the band: sum over the 32 pixels (lanes of a half), one slot per (image, band, strip)
  if (A.stats) {
    float* slot = A.stats + ((size_t)(b * A.bands + band) * A.strips + strip) * Cout * 2;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float a = s1[r], q = s2[r];
#pragma unroll
      for (int msk = 1; msk < 32; msk <<= 1) {
        a += __shfl_xor(a, msk, 64);
        q += __shfl_xor(q, msk, 64);
      }
      const int co = cg * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (l32 == 0 && co < Cout) {
        slot[co * 2] = a;
        slot[co * 2 + 1] = q;
      }
    }
  }
}

struct FPlan {
  int OH, OW, strips, groups, bands, waves;
  size_t lds, nslots;
};

bool make_fplan(int B, int Cin, int H, int W, int Cout, int S, FPlan& p) {
  if (Cin < 1 || Cin > 3 || Cout < 1 || (S != 1 && S != 2) || H < 1 || W < 1 || B < 1) return false;
  p.OH = (H - 1) / S + 1;
  p.OW = (W - 1) / S + 1;
  if (S == 2 && (W % 2)) return false;                 // patch columns S*ow + kw <= W + 3
  if (p.OW % 32 || (size_t)p.OH * p.OW > ((size_t)1 << 24)) return false;     // 32-bit byte offsets inside a 32-channel group
  p.strips = p.OW / 32;
  p.groups = cdiv(Cout, 32);
  p.waves = p.strips * p.groups;
  if (p.waves > 8) return false;
  const int rb = (S == 1) ? 8 : 4, pr = S * (rb - 1) + 5;
  p.bands = cdiv(p.OH, rb);
  p.lds = ((size_t)3 * pr * (W + 4) + 32 * 76 * p.groups) * sizeof(float);     // patch + filter + bias
  if (p.lds > 64 * 1024) return false;
  p.nslots = (size_t)B * p.bands * p.strips;
  return (long)B * p.bands <= 0x7fffffffL && (size_t)B * Cin * H * W <= 0x7fffffffUL;
}

}  // namespace

extern "C" int vg_conv5x5_thin_bf16split_ok(int Cin, int H, int W, int Cout, int stride) {
  FPlan p;
  return make_fplan(1, Cin, H, W, Cout, stride, p) ? 1 : 0;
}

extern "C" size_t vg_conv5x5_thin_bf16split_stats_floats(int B, int Cin, int H, int W, int Cout, int stride) {
  FPlan p;
  if (!make_fplan(B, Cin, H, W, Cout, stride, p)) return 0;
  return p.nslots * Cout * 2;
}

extern "C" int vg_conv5x5_thin_bf16split(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int H,
                                         int W, int Cout, int stride, int planes, float* stats, size_t stats_floats,
                                         void* stream) {
  if (!x || !w || !y || (planes != 2 && planes != 3)) return VG_ERR_BAD_ARG;
  FPlan p;
  if (!make_fplan(B, Cin, H, W, Cout, stride, p)) return VG_ERR_BAD_ARG;
  if (stats && stats_floats < p.nslots * Cout * 2) return VG_ERR_BAD_ARG;
  FArgs A;
  A.x = x; A.w = w; A.bias = bias; A.y = y; A.stats = stats;
  A.B = B; A.Cin = Cin; A.H = H; A.W = W; A.Cout = Cout; A.OH = p.OH; A.OW = p.OW; A.bands = p.bands; A.strips = p.strips;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)(B * p.bands)), block(p.waves * 64);
  if (stride == 1) {
    if (planes == 3) hipLaunchKernelGGL((conv_thin_fwd_kernel<1, 3>), grid, block, p.lds, st, A);
    else hipLaunchKernelGGL((conv_thin_fwd_kernel<1, 2>), grid, block, p.lds, st, A);
  } else {
    if (planes == 3) hipLaunchKernelGGL((conv_thin_fwd_kernel<2, 3>), grid, block, p.lds, st, A);
    else hipLaunchKernelGGL((conv_thin_fwd_kernel<2, 2>), grid, block, p.lds, st, A);
  }
  VG_CHECK_LAUNCH();
  return 0;
}
