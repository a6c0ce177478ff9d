// Train-mode BatchNorm (2-D and 1-D) fused with ReLU / LeakyReLU(0.2), forward and
// backward, plus the per-channel sum used for convolution bias gradients.
// HBM-bound kernels: float4 coalesced streams over NCHW planes, fp64 partial sums
// reduced with 64-lane wavefront shuffles, deterministic two-stage reductions.
//
// Replaces F.batch_norm(training=True) + activation reached from
// /root/reference/models/model.py:451-458 (encoder), :462,468 (heads), :492
// (decoder preprocess), :496-505 (decoder), :390-400 (discriminator).
#include "common.hpp"
#include "vaegan_hip.h"

namespace {

constexpr int NT = 256;
constexpr int NS_MAX = 64;   // slices per channel in the two-stage reductions

__device__ __forceinline__ float act_fwd(float v, int act) {
  if (act == VG_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == VG_ACT_LRELU) return v > 0.f ? v : 0.2f * v;
  return v;
}
__device__ __forceinline__ float act_grad(float pre, float g, int act) {
  if (act == VG_ACT_RELU) return pre > 0.f ? g : 0.f;
  if (act == VG_ACT_LRELU) return pre > 0.f ? g : 0.2f * g;
  return g;
}

struct Slicing {
  int ns;        // slices per channel
  long per;      // virtual elements (b*HW+hw) per slice, multiple of 4 when HW%4==0
};
Slicing make_slicing(int B, int C, int HW) {
  const long total = (long)B * HW;
  long ns = 2048 / C;
  if (ns < 1) ns = 1;
  if (ns > NS_MAX) ns = NS_MAX;
  long per = (total + ns - 1) / ns;
  if (per < 1024) per = 1024;
  per = (per + 3) / 4 * 4;
  ns = (total + per - 1) / per;
  Slicing s;
  s.ns = (int)ns;
  s.per = per;
  return s;
}

// MODE 0: sum x, sum x^2.   MODE 1: sum g_pre, sum g_pre*xhat (backward).   MODE 2: sum x only.
template <int MODE>
__global__ __launch_bounds__(NT) void bn_partial_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, double* __restrict__ part,
                                                        int B, int C, int HW, long per, int ns, int act) {
  __shared__ double red[NT / 64];
  const int c = blockIdx.x, k = blockIdx.y;
  const long total = (long)B * HW;
  const long v0 = (long)k * per, v1 = min(v0 + per, total);
  const int hw_shift = (HW & (HW - 1)) == 0 ? __builtin_ctz(HW) : -1;      // power-of-two planes: no 64-bit division per load
  double s1 = 0.0, s2 = 0.0;
  float mu = 0.f, is = 0.f, sc = 0.f, sh = 0.f;
  if (MODE == 1) {
    mu = mean[c];
    is = invstd[c];
    sc = gamma[c] * is;
    sh = beta[c] - mu * sc;
  }
  auto accum = [&](float xv, float gv) {
    if (MODE == 0) {
      s1 += xv;
      s2 += (double)xv * xv;
    } else if (MODE == 2) {
      s1 += xv;
    } else {
      const float g = act_grad(fmaf(xv, sc, sh), gv, act);
      s1 += g;
      s2 += (double)(g * ((xv - mu) * is));
    }
  };
  if ((HW & 3) == 0) {
    for (long v = v0 + 4L * threadIdx.x; v < v1; v += 4L * NT) {
      const long b = hw_shift >= 0 ? (v >> hw_shift) : v / HW, hw = v - b * HW;
      const size_t off = ((size_t)b * C + c) * HW + hw;
      const float4 xv = *reinterpret_cast<const float4*>(x + off);
      float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (MODE == 1) gv = *reinterpret_cast<const float4*>(gy + off);
      accum(xv.x, gv.x);
      accum(xv.y, gv.y);
      accum(xv.z, gv.z);
      accum(xv.w, gv.w);
    }
  } else {
    for (long v = v0 + threadIdx.x; v < v1; v += NT) {
      const long b = hw_shift >= 0 ? (v >> hw_shift) : v / HW, hw = v - b * HW;
      const size_t off = ((size_t)b * C + c) * HW + hw;
      accum(x[off], MODE == 1 ? gy[off] : 0.f);
    }
  }
  const double t1 = block_sum<NT>(s1, red);
  const double t2 = block_sum<NT>(s2, red);
  if (threadIdx.x == 0) {
    part[((size_t)c * ns + k) * 2 + 0] = t1;
    part[((size_t)c * ns + k) * 2 + 1] = t2;
  }
}

__global__ __launch_bounds__(NT) void channel_sum_finalize_kernel(const double* __restrict__ part, int ns, int C,
                                                                  float* __restrict__ out) {
  const int c = blockIdx.x * NT + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0;
  for (int k = 0; k < ns; ++k) s1 += part[((size_t)c * ns + k) * 2];
  out[c] = (float)s1;
}

// Normalise (FWD) or input-gradient (BWD) pass with the per-channel finalisation folded in: a
// workgroup owns one slice of ONE channel, re-derives that channel's coefficients from the
// (<= 64) fp64 partial sums in a fixed order -- no separate finalize launch, no coefficient
// table, coefficients live in registers -- and streams its slice as float4.  The slice-0
// workgroup of a channel also writes the saved / running statistics (FWD) or dgamma / dbeta (BWD).
//   FWD: y  = act(x*sc + sh)            sc = gamma*invstd, sh = beta - mean*sc
//   BWD: gx = sc * (g_pre - c1 - xhat*c2),  c1 = sum(g_pre)/N, c2 = sum(g_pre*xhat)/N
template <bool BWD>
__global__ __launch_bounds__(NT) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                      const double* __restrict__ part, int ns,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ mean_io,
                                                      float* __restrict__ invstd_io,
                                                      float* __restrict__ running_mean,
                                                      float* __restrict__ running_var, float* __restrict__ dgamma,
                                                      float* __restrict__ dbeta, float* __restrict__ out, int B,
                                                      int C, int HW, long per, float eps, float momentum, int act,
                                                      unsigned* __restrict__ out_amax) {
  // out_amax (backward, may be NULL): max |gx| is added to it (common.hpp block_amax_atomic) -- the bound the fp16-plane
  // convolutions that consume gx scale it by, emitted here instead of by a pass of its own
  const bool accp = (act & 0x100) != 0;        // backward: dgamma / dbeta are accumulated into (bit 8 of `act`)
  act &= 0xff;
  // `per`: elements of this channel per workgroup of THIS pass (independent of the partial pass)
  __shared__ float s_co[4];
  // slices in descending order: the sums pass before this one went through the channel's slices in ascending order (same
  // channel -> same XCD in both launches), so the end of the channel is what its L2 still holds
#if defined(VG_BN_APPLY_ASCENDING)      // timing experiments: the previous order
  const int c = blockIdx.x, k = blockIdx.y;
#else
  const int c = blockIdx.x, k = (int)gridDim.y - 1 - (int)blockIdx.y;
#endif
  const double count = (double)B * HW;
  if (threadIdx.x < 64) {   // wavefront 0: fixed-order reduction of the slice partials
    double s1 = 0.0, s2 = 0.0;
    for (int j = threadIdx.x; j < ns; j += 64) {
      s1 += part[((size_t)c * ns + j) * 2];
      s2 += part[((size_t)c * ns + j) * 2 + 1];
    }
    s1 = wave_allsum(s1);
    s2 = wave_allsum(s2);
    if (threadIdx.x == 0) {
      if (!BWD) {
        const double m = s1 / count;
        double var = s2 / count - m * m;
        if (var < 0.0) var = 0.0;
        const float mu = (float)m, is = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * is;
        s_co[0] = sc;
        s_co[1] = beta[c] - mu * sc;
        if (k == 0) {
          mean_io[c] = mu;
          invstd_io[c] = is;
          if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
          if (running_var) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
          }
        }
      } else {
        const float mu = mean_io[c], is = invstd_io[c], sc = gamma[c] * is;
        s_co[0] = sc;
        s_co[1] = beta[c] - mu * sc;
        s_co[2] = (float)(s1 / count);
        s_co[3] = (float)(s2 / count);
        if (k == 0) {       // accp: added to what is there (a layer used twice before one backward: no add launch)
          if (dbeta) dbeta[c] = (float)s1 + (accp ? dbeta[c] : 0.f);
          if (dgamma) dgamma[c] = (float)s2 + (accp ? dgamma[c] : 0.f);
        }
      }
    }
  }
  __syncthreads();
  const float sc = s_co[0], sh = s_co[1];
  float c1 = 0.f, c2 = 0.f, mu = 0.f, is = 0.f;
  if (BWD) {
    c1 = s_co[2];
    c2 = s_co[3];
    mu = mean_io[c];
    is = invstd_io[c];
  }
  unsigned am = 0;
  auto one = [&](float xv, float gv) -> float {
    const float pre = fmaf(xv, sc, sh);
    if (!BWD) {
      const float r = act_fwd(pre, act);
      am = max(am, abs_bits(r));
      return r;
    }
    const float g = act_grad(pre, gv, act);
    const float r = sc * (g - c1 - ((xv - mu) * is) * c2);
    am = max(am, abs_bits(r));
    return r;
  };
  const long total = (long)B * HW;
  const long v0 = (long)k * per, v1 = min(v0 + per, total);
  const int hw_shift = (HW & (HW - 1)) == 0 ? __builtin_ctz(HW) : -1;
  if ((HW & 3) == 0) {
    for (long v = v0 + 4L * threadIdx.x; v < v1; v += 4L * NT) {
      const long b = hw_shift >= 0 ? (v >> hw_shift) : v / HW, hw = v - b * HW;
      const size_t off = ((size_t)b * C + c) * HW + hw;
      const float4 xv = *reinterpret_cast<const float4*>(x + off);
      float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (BWD) gv = *reinterpret_cast<const float4*>(gy + off);
      float4 o;
      o.x = one(xv.x, gv.x);
      o.y = one(xv.y, gv.y);
      o.z = one(xv.z, gv.z);
      o.w = one(xv.w, gv.w);
      *reinterpret_cast<float4*>(out + off) = o;
    }
  } else {
    for (long v = v0 + threadIdx.x; v < v1; v += NT) {
      const long b = hw_shift >= 0 ? (v >> hw_shift) : v / HW, hw = v - b * HW;
      const size_t off = ((size_t)b * C + c) * HW + hw;
      out[off] = one(x[off], BWD ? gy[off] : 0.f);
    }
  }
  if (out_amax) block_amax_atomic<NT>(am, out_amax);      // wave-uniform condition
}

// Backward in ONE pass for channels that fit a workgroup's registers (B * HW <= ONE_NT * 4 * NV elements: the 16 x 16 and
// 8 x 8 layers at B = 128): a workgroup of 16 wavefronts owns a channel, holds its (x, gy) in registers (2 * NV float4 per
// lane), sums g_pre and g_pre * xhat in fp64 in a fixed order, and writes gx from the registers -- two reads and one write
// per element where the two-pass form above reads (x, gy) twice.  Same formulas as bn_partial_kernel<1> +
// bn_apply_kernel<true>; only the order of the fp64 sums differs.
constexpr int ONE_NT = 1024;

template <int NV>
__global__ __launch_bounds__(ONE_NT) void bn_bwd_onepass_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, float* __restrict__ gx,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                int B, int C, int HW, int act,
                                                                unsigned* __restrict__ gx_amax) {
  __shared__ double red[ONE_NT / 64];
  __shared__ float s_c[2];
  const bool accp = (act & 0x100) != 0;
  act &= 0xff;
  const int c = blockIdx.x;
  const int total = B * HW;
  const int hw_shift = (HW & (HW - 1)) == 0 ? __builtin_ctz(HW) : -1;
  const float mu = mean[c], is = invstd[c], sc = gamma[c] * is, sh = beta[c] - mu * sc;
  float4 xv[NV], gv[NV];
  size_t off[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int v = (j * ONE_NT + (int)threadIdx.x) * 4;
    const int vc = min(v, total - 4);                       // clamped, unconditional; masked below
    const int b = hw_shift >= 0 ? (vc >> hw_shift) : vc / HW, hw = vc - b * HW;
    off[j] = ((size_t)b * C + c) * HW + hw;
    xv[j] = *reinterpret_cast<const float4*>(x + off[j]);
    gv[j] = *reinterpret_cast<const float4*>(gy + off[j]);
  }
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const bool ok = (j * ONE_NT + (int)threadIdx.x) * 4 < total;
    float* xp = reinterpret_cast<float*>(&xv[j]);
    float* gp = reinterpret_cast<float*>(&gv[j]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float g = ok ? act_grad(fmaf(xp[e], sc, sh), gp[e], act) : 0.f;
      gp[e] = g;                                            // g_pre replaces gy
      s1 += g;
      s2 += (double)(g * ((xp[e] - mu) * is));
    }
  }
  const double t1 = block_sum<ONE_NT>(s1, red);
  const double t2 = block_sum<ONE_NT>(s2, red);
  if (threadIdx.x == 0) {
    const double count = (double)total;
    s_c[0] = (float)(t1 / count);
    s_c[1] = (float)(t2 / count);
    if (dbeta) dbeta[c] = (float)t1 + (accp ? dbeta[c] : 0.f);
    if (dgamma) dgamma[c] = (float)t2 + (accp ? dgamma[c] : 0.f);
  }
  __syncthreads();
  const float c1 = s_c[0], c2 = s_c[1];
  unsigned am = 0;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    if ((j * ONE_NT + (int)threadIdx.x) * 4 >= total) continue;
    const float* xp = reinterpret_cast<const float*>(&xv[j]);
    const float* gp = reinterpret_cast<const float*>(&gv[j]);
    float4 o;
    o.x = sc * (gp[0] - c1 - ((xp[0] - mu) * is) * c2);
    o.y = sc * (gp[1] - c1 - ((xp[1] - mu) * is) * c2);
    o.z = sc * (gp[2] - c1 - ((xp[2] - mu) * is) * c2);
    o.w = sc * (gp[3] - c1 - ((xp[3] - mu) * is) * c2);
    am = max(max(am, abs_bits(o.x)), max(max(abs_bits(o.y), abs_bits(o.z)), abs_bits(o.w)));
    *reinterpret_cast<float4*>(gx + off[j]) = o;
  }
  if (gx_amax) block_amax_atomic<ONE_NT>(am, gx_amax);      // max |gx| for the fp16-plane consumers (bn_apply_kernel)
}

// BatchNorm1d: x [B][C].  A workgroup owns 32 consecutive channels (128-byte coalesced rows);
// its 8 row-slices (threads 32*s .. 32*s+31) each sum every 8th batch row in fp64 and combine
// through LDS in a fixed order; the normalise / gradient pass re-reads the rows from L2.
constexpr int B1_CH = 32, B1_SL = 8;

__global__ __launch_bounds__(NT) void bn1d_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ y,
                                                      float* __restrict__ running_mean,
                                                      float* __restrict__ running_var, float* __restrict__ save_mean,
                                                      float* __restrict__ save_invstd, int B, int C, float eps,
                                                      float momentum, int act) {
  __shared__ double r1[B1_SL][B1_CH], r2[B1_SL][B1_CH];
  __shared__ float s_sc[B1_CH], s_sh[B1_CH];
  const int cl = threadIdx.x % B1_CH, sl = threadIdx.x / B1_CH;
  const int c = blockIdx.x * B1_CH + cl;
  const bool cok = c < C;
  double s1 = 0.0, s2 = 0.0;
  if (cok)
    for (int b = sl; b < B; b += B1_SL) {
      const float v = x[(size_t)b * C + c];
      s1 += v;
      s2 += (double)v * v;
    }
  r1[sl][cl] = s1;
  r2[sl][cl] = s2;
  __syncthreads();
  if (sl == 0 && cok) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int k = 0; k < B1_SL; ++k) {
      t1 += r1[k][cl];
      t2 += r2[k][cl];
    }
    const double m = t1 / B;
    double var = t2 / B - m * m;
    if (var < 0.0) var = 0.0;
    const float mu = (float)m, is = (float)(1.0 / sqrt(var + (double)eps));
    save_mean[c] = mu;
    save_invstd[c] = is;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    if (running_var) {
      const double unb = B > 1 ? var * B / (B - 1.0) : var;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
    const float sc = gamma[c] * is;
    s_sc[cl] = sc;
    s_sh[cl] = beta[c] - mu * sc;
  }
  __syncthreads();
  if (cok) {
    const float sc = s_sc[cl], sh = s_sh[cl];
    for (int b = sl; b < B; b += B1_SL) {
      const size_t o = (size_t)b * C + c;
      y[o] = act_fwd(fmaf(x[o], sc, sh), act);
    }
  }
}

__global__ __launch_bounds__(NT) void bn1d_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta,
                                                      const float* __restrict__ mean,
                                                      const float* __restrict__ invstd, float* __restrict__ gx,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta, int B,
                                                      int C, int act, unsigned* __restrict__ gx_amax) {
  const bool accp = (act & 0x100) != 0;        // as bn_apply_kernel
  act &= 0xff;
  __shared__ double r1[B1_SL][B1_CH], r2[B1_SL][B1_CH];
  __shared__ float s_c1[B1_CH], s_c2[B1_CH];
  const int cl = threadIdx.x % B1_CH, sl = threadIdx.x / B1_CH;
  const int c = blockIdx.x * B1_CH + cl;
  const bool cok = c < C;
  float mu = 0.f, is = 0.f, sc = 0.f, sh = 0.f;
  double s1 = 0.0, s2 = 0.0;
  if (cok) {
    mu = mean[c];
    is = invstd[c];
    sc = gamma[c] * is;
    sh = beta[c] - mu * sc;
    for (int b = sl; b < B; b += B1_SL) {
      const size_t o = (size_t)b * C + c;
      const float xv = x[o];
      const float g = act_grad(fmaf(xv, sc, sh), gy[o], act);
      s1 += g;
      s2 += (double)(g * ((xv - mu) * is));
    }
  }
  r1[sl][cl] = s1;
  r2[sl][cl] = s2;
  __syncthreads();
  if (sl == 0 && cok) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int k = 0; k < B1_SL; ++k) {
      t1 += r1[k][cl];
      t2 += r2[k][cl];
    }
    if (dbeta) dbeta[c] = (float)t1 + (accp ? dbeta[c] : 0.f);
    if (dgamma) dgamma[c] = (float)t2 + (accp ? dgamma[c] : 0.f);
    s_c1[cl] = (float)(t1 / B);
    s_c2[cl] = (float)(t2 / B);
  }
  __syncthreads();
  unsigned am = 0;
  if (cok) {
    const float c1 = s_c1[cl], c2 = s_c2[cl];
    for (int b = sl; b < B; b += B1_SL) {
      const size_t o = (size_t)b * C + c;
      const float xv = x[o];
      const float g = act_grad(fmaf(xv, sc, sh), gy[o], act);
      const float r = sc * (g - c1 - ((xv - mu) * is) * c2);
      am = max(am, abs_bits(r));
      gx[o] = r;
    }
  }
  if (gx_amax) block_amax_atomic<NT>(am, gx_amax);
}

// slices of the apply pass: >= 4096 elements per workgroup so that the per-workgroup prologue (partials
// reduction + barrier) is amortised, but enough workgroups to fill the chip
Slicing make_apply_slicing(int B, int C, int HW) {
  const long total = (long)B * HW;
  long ns = 4096 / C;
  if (ns < 1) ns = 1;
  long per = (total + ns - 1) / ns;
  if (per < 4096) per = 4096;
  per = (per + 3) / 4 * 4;
  Slicing s;
  s.per = per;
  s.ns = (int)((total + per - 1) / per);
  return s;
}

size_t part_bytes(int C) { return (size_t)C * NS_MAX * 2 * sizeof(double); }
size_t ws_bytes(int C) { return part_bytes(C) + 64; }

}  // namespace

// Per-channel coefficients of a train-mode BatchNorm from partial sums: `part` holds, for channel c and partial k,
// (sum, sum of squares) at part[(c * cs + k * ks) * 2 + {0, 1}] -- fp32 slots written by a convolution epilogue
// (cs = 1... see vg_bn_finalize_stats) or the fp64 slices of bn_partial_kernel<0>.  Fixed summation order, fp64.
// Writes mean / invstd (saved for backward), scale = gamma * invstd, shift = beta - mean * scale, and updates the
// running statistics exactly as bn_apply_kernel's slice-0 workgroup does.
template <typename T>
__global__ __launch_bounds__(NT) void bn_finalize_kernel(const T* __restrict__ part, int np, long cs, long ks, int C,
                                                         double count, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ running_mean,
                                                         float* __restrict__ running_var, float* __restrict__ mean_out,
                                                         float* __restrict__ invstd_out, float* __restrict__ scale,
                                                         float* __restrict__ shift, float eps, float momentum,
                                                         unsigned* __restrict__ act_amax) {
  // act_amax (may be NULL): an upper bound of max |act(BN(x))| over the whole tensor is added to it (atomic maximum of
  // the channels' bounds) -- what the fp16-plane convolution that applies these coefficients on load scales its input by
  // 32 channels per workgroup x 8 partial-lanes: consecutive threads read consecutive channels (the convolution's
  // slots are [slot][C][2]: 256 contiguous bytes per 32 threads), each partial-lane sums every 8th partial, and the 8
  // sums of a channel are added in a fixed order
  constexpr int CH = 32, KL = NT / CH;
  __shared__ double r1[KL][CH], r2[KL][CH];
  const int cl = threadIdx.x % CH, kl = threadIdx.x / CH;
  const int c = blockIdx.x * CH + cl;
  double s1 = 0.0, s2 = 0.0;
  if (c < C)
    for (int k = kl; k < np; k += KL) {
      s1 += (double)part[((size_t)c * cs + (size_t)k * ks) * 2];
      s2 += (double)part[((size_t)c * cs + (size_t)k * ks) * 2 + 1];
    }
  r1[kl][cl] = s1;
  r2[kl][cl] = s2;
  __syncthreads();
  if (kl == 0 && c < C) {
    s1 = 0.0;
    s2 = 0.0;
#pragma unroll
    for (int k = 0; k < KL; ++k) {
      s1 += r1[k][cl];
      s2 += r2[k][cl];
    }
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    const float mu = (float)m, is = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * is;
    mean_out[c] = mu;
    invstd_out[c] = is;
    scale[c] = sc;
    shift[c] = beta[c] - mu * sc;
    if (act_amax) {      // Chebyshev: |x - mean| <= sigma sqrt(n - 1), so |BN(x)| <= |gamma| sqrt(n) + |beta| (+ fp32 rounding of x * sc + sh)
      const float bound = (fabsf(gamma[c]) * sqrtf((float)count) + fabsf(beta[c])) * 1.001f + fabsf(mu * sc) * 1e-6f;
      atomicMax(act_amax, __float_as_uint(bound));
    }
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    if (running_var) {
      const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
  }
}

// MANY slots in ONE launch (the ring kernels leave 512-2048 of them at B = 128): a workgroup of 1024 threads owns 8
// channels, 128 partial-lanes each -- every lane sums every 128th slot (4-16 independent loads), the 128 fp64 sums of a
// channel are added in a fixed order through LDS (16 lanes x 8, then 16).  Replaces stats_partial + finalize (two launches
// of ~5 us and a kernel boundary) wherever a channel has <= 4096 slots.
__global__ __launch_bounds__(1024) void bn_finalize_wide_kernel(const float* __restrict__ stats, int nslots, int C,
                                                                double count, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta,
                                                                float* __restrict__ running_mean,
                                                                float* __restrict__ running_var, float* __restrict__ mean_out,
                                                                float* __restrict__ invstd_out, float* __restrict__ scale,
                                                                float* __restrict__ shift, float eps, float momentum,
                                                                unsigned* __restrict__ act_amax) {
  constexpr int CH = 8, KL = 128;
  __shared__ double r1[KL][CH], r2[KL][CH];
  const int cl = threadIdx.x % CH, kl = threadIdx.x / CH;
  const int c = blockIdx.x * CH + cl;
  double s1 = 0.0, s2 = 0.0;
  if (c < C)
    for (int k = kl; k < nslots; k += KL) {
      const float2 v = *reinterpret_cast<const float2*>(stats + ((size_t)k * C + c) * 2);
      s1 += (double)v.x;
      s2 += (double)v.y;
    }
  r1[kl][cl] = s1;
  r2[kl][cl] = s2;
  __syncthreads();
  if (kl < 16) {                      // 16 lanes per channel: each adds 8 consecutive partial-lanes, in order
    s1 = 0.0;
    s2 = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      s1 += r1[kl * 8 + k][cl];
      s2 += r2[kl * 8 + k][cl];
    }
  }
  __syncthreads();
  if (kl < 16) {
    r1[kl][cl] = s1;
    r2[kl][cl] = s2;
  }
  __syncthreads();
  if (kl == 0 && c < C) {
    s1 = 0.0;
    s2 = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      s1 += r1[k][cl];
      s2 += r2[k][cl];
    }
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    const float mu = (float)m, is = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * is;
    mean_out[c] = mu;
    invstd_out[c] = is;
    scale[c] = sc;
    shift[c] = beta[c] - mu * sc;
    if (act_amax) {      // Chebyshev: |x - mean| <= sigma sqrt(n - 1), so |BN(x)| <= |gamma| sqrt(n) + |beta| (+ fp32 rounding of x * sc + sh)
      const float bound = (fabsf(gamma[c]) * sqrtf((float)count) + fabsf(beta[c])) * 1.001f + fabsf(mu * sc) * 1e-6f;
      atomicMax(act_amax, __float_as_uint(bound));
    }
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    if (running_var) {
      const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
  }
}

// First stage for MANY slots (a 3 -> 32 first layer at B = 128 leaves 16 384 of them for 32 channels): workgroup
// (channel block, split) sums its share of the slots into fp64 partials part[c][split][2], in a fixed order.
__global__ __launch_bounds__(NT) void stats_partial_kernel(const float* __restrict__ stats, int nslots, int C, int nsplit,
                                                           double* __restrict__ part) {
  constexpr int CH = 32, KL = NT / CH;
  __shared__ double r1[KL][CH], r2[KL][CH];
  const int cl = threadIdx.x % CH, kl = threadIdx.x / CH;
  const int c = blockIdx.x * CH + cl, sp = blockIdx.y;
  const int per = cdiv(nslots, nsplit), k0 = sp * per, k1 = min(k0 + per, nslots);
  double s1 = 0.0, s2 = 0.0;
  if (c < C)
    for (int k = k0 + kl; k < k1; k += KL) {
      s1 += (double)stats[((size_t)k * C + c) * 2];
      s2 += (double)stats[((size_t)k * C + c) * 2 + 1];
    }
  r1[kl][cl] = s1;
  r2[kl][cl] = s2;
  __syncthreads();
  if (kl == 0 && c < C) {
    s1 = 0.0;
    s2 = 0.0;
#pragma unroll
    for (int k = 0; k < KL; ++k) {
      s1 += r1[k][cl];
      s2 += r2[k][cl];
    }
    part[((size_t)c * nsplit + sp) * 2] = s1;
    part[((size_t)c * nsplit + sp) * 2 + 1] = s2;
  }
}

constexpr int AA_U = 4;
// y = act(x * scale[c] + shift[c]): the normalise pass with given coefficients (layers whose consumer cannot apply
// them while it loads)
// hw4_shift >= 0: H*W/4 is a power of two (every layer of the reference) -- the channel is a shift and a 32-bit remainder;
// the 64-bit division per thread that this replaces made the pass run at 0.8 TB/s (20 us for 16 MB).
__global__ __launch_bounds__(NT) void affine_act_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, float* __restrict__ y, int C,
                                                        int HW, size_t n4, int act, int hw4_shift,
                                                        unsigned* __restrict__ y_amax) {
  // A workgroup walks chunks of NT * AA_U units of 16 bytes, grid-stride (<= 2048 workgroups: one maximum and one atomic
  // per workgroup, not per 4 KB); within a chunk a thread's AA_U units are NT apart (contiguous per round, all loads of
  // a chunk in flight together).
  unsigned am = 0;
  for (size_t base = (size_t)blockIdx.x * (NT * AA_U) + threadIdx.x; base < n4 + threadIdx.x;
       base += (size_t)gridDim.x * (NT * AA_U)) {
    float4 v[AA_U];
    float sc[AA_U], sh[AA_U];
#pragma unroll
    for (int u = 0; u < AA_U; ++u) {
      const size_t i0 = base + (size_t)u * NT;
      const size_t i = i0 < n4 ? i0 : n4 - 1;               // past the end: the last unit again (not stored)
      // HW % 4 == 0: the four elements share a channel
      const int c = hw4_shift >= 0 ? (int)((unsigned)(i >> hw4_shift) % (unsigned)C) : (int)(((4 * i) / HW) % C);
      sc[u] = scale[c];
      sh[u] = shift[c];
      v[u] = *reinterpret_cast<const float4*>(x + 4 * i);
    }
#pragma unroll
    for (int u = 0; u < AA_U; ++u) {
      const size_t i0 = base + (size_t)u * NT;
      float4 o;
      o.x = act_fwd(fmaf(v[u].x, sc[u], sh[u]), act);
      o.y = act_fwd(fmaf(v[u].y, sc[u], sh[u]), act);
      o.z = act_fwd(fmaf(v[u].z, sc[u], sh[u]), act);
      o.w = act_fwd(fmaf(v[u].w, sc[u], sh[u]), act);
      if (i0 < n4) *reinterpret_cast<float4*>(y + 4 * i0) = o;
      am = max(max(am, abs_bits(o.x)), max(max(abs_bits(o.y), abs_bits(o.z)), abs_bits(o.w)));
    }
  }
  if (y_amax) block_amax_atomic<NT>(am, y_amax);      // max |y| for an fp16-plane consumer (common.hpp)
}

// the same for H*W not a multiple of 4 (odd image sizes): one element per lane
__global__ __launch_bounds__(NT) void affine_act_scalar_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, float* __restrict__ y,
                                                               int C, int HW, size_t n, int act,
                                                               unsigned* __restrict__ y_amax) {
  const size_t e0 = (size_t)blockIdx.x * NT + threadIdx.x;
  const size_t e = e0 < n ? e0 : n - 1;
  const int c = (int)((e / HW) % C);
  const float r = act_fwd(fmaf(x[e], scale[c], shift[c]), act);
  if (e0 < n) y[e] = r;
  if (y_amax) block_amax_atomic<NT>(abs_bits(r), y_amax);
}

extern "C" size_t vg_bn_workspace_bytes(int C) { return C > 0 ? ws_bytes(C) : 0; }

extern "C" int vg_bn_finalize_stats(const float* stats, int nslots, int C, double count, const float* gamma,
                                    const float* beta, float* running_mean, float* running_var, float* save_mean,
                                    float* save_invstd, float* scale, float* shift, float eps, float momentum,
                                    float* act_amax, void* workspace, size_t workspace_bytes, void* stream) {
  unsigned* am = (unsigned*)act_amax;
  if (!stats || nslots <= 0 || C <= 0 || count <= 0 || !gamma || !beta || !save_mean || !save_invstd || !scale || !shift)
    return VG_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  // few slots: one launch (a workgroup per 32 channels, 8 partial-lanes: up to 8 dependent loads per lane); more:
  // NS_MAX-way first stage into the BatchNorm workspace.  (The one-launch form on 1024 slots x 32 channels ran 56 us
  // -- one workgroup, 128 dependent loads per lane -- against ~6 us for the two stages.)
  if (nslots <= 64) {
    hipLaunchKernelGGL(bn_finalize_kernel<float>, dim3(cdiv(C, 32)), dim3(NT), 0, st, stats, nslots, 1L, (long)C, C,
                       count, gamma, beta, running_mean, running_var, save_mean, save_invstd, scale, shift, eps,
                       momentum, am);      // stats[slot][C][2]: channel stride 1, slot stride C
    VG_CHECK_LAUNCH();
    return 0;
  }
  if (nslots <= 4096) {
    hipLaunchKernelGGL(bn_finalize_wide_kernel, dim3(cdiv(C, 8)), dim3(1024), 0, st, stats, nslots, C, count, gamma, beta,
                       running_mean, running_var, save_mean, save_invstd, scale, shift, eps, momentum, am);
    VG_CHECK_LAUNCH();
    return 0;
  }
  if (!workspace || workspace_bytes < ws_bytes(C)) return VG_ERR_WORKSPACE;
  double* part = (double*)workspace;
  hipLaunchKernelGGL(stats_partial_kernel, dim3(cdiv(C, 32), NS_MAX), dim3(NT), 0, st, stats, nslots, C, NS_MAX, part);
  VG_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_finalize_kernel<double>, dim3(cdiv(C, 32)), dim3(NT), 0, st, (const double*)part, NS_MAX,
                     (long)NS_MAX, 1L, C, count, gamma, beta, running_mean, running_var, save_mean, save_invstd, scale,
                     shift, eps, momentum, am);      // part[c][NS_MAX][2]
  VG_CHECK_LAUNCH();
  return 0;
}

extern "C" int vg_bn_stats(const float* x, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, float* save_mean, float* save_invstd, float* scale, float* shift, int B,
                           int C, int HW, float eps, float momentum, float* act_amax, void* workspace,
                           size_t workspace_bytes, void* stream) {
  if (!x || !gamma || !beta || !save_mean || !save_invstd || !scale || !shift || B <= 0 || C <= 0 || HW <= 0)
    return VG_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < ws_bytes(C)) return VG_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  double* part = (double*)workspace;
  const Slicing s = make_slicing(B, C, HW);
  hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(C, s.ns), dim3(NT), 0, st, x, (const float*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, part, B, C, HW, s.per, s.ns, 0);
  VG_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_finalize_kernel<double>, dim3(cdiv(C, 32)), dim3(NT), 0, st, (const double*)part, s.ns,
                     (long)s.ns, 1L, C, (double)B * HW, gamma, beta, running_mean, running_var, save_mean, save_invstd,
                     scale, shift, eps, momentum, (unsigned*)act_amax);      // part[c][ns][2]
  VG_CHECK_LAUNCH();
  return 0;
}

extern "C" int vg_affine_act(const float* x, const float* scale, const float* shift, float* y, int B, int C, int HW,
                             int act, float* y_amax, void* stream) {
  if (!x || !scale || !shift || !y || B <= 0 || C <= 0 || HW <= 0) return VG_ERR_BAD_ARG;
  if (act < VG_ACT_NONE || act > VG_ACT_LRELU) return VG_ERR_BAD_ARG;
  if (HW & 3) {
    const size_t n = (size_t)B * C * HW;
    if (cdiv((long)n, (long)NT) > 0x7fffffffL) return VG_ERR_BAD_ARG;
    hipLaunchKernelGGL(affine_act_scalar_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream, x,
                       scale, shift, y, C, HW, n, act, (unsigned*)y_amax);
    VG_CHECK_LAUNCH();
    return 0;
  }
  const size_t n4 = (size_t)B * C * HW / 4;
  if (cdiv((long)n4, (long)NT) > 0x7fffffffL) return VG_ERR_BAD_ARG;
  const int hw4 = HW / 4;
  int sh = -1;
  if ((hw4 & (hw4 - 1)) == 0 && n4 < (1ULL << 40)) {      // power of two (and plane indices that fit 32 bits after the shift)
    sh = 0;
    while ((1 << sh) < hw4) ++sh;
    if ((n4 >> sh) > 0xffffffffULL) sh = -1;
  }
  const size_t chunks = (n4 + NT * AA_U - 1) / (NT * AA_U);
  hipLaunchKernelGGL(affine_act_kernel, dim3((unsigned)(chunks < 2048 ? chunks : 2048)), dim3(NT), 0, (hipStream_t)stream, x, scale,
                     shift, y, C, HW, n4, act, sh, (unsigned*)y_amax);
  VG_CHECK_LAUNCH();
  return 0;
}

extern "C" int vg_bn_act_fwd(const float* x, const float* gamma, const float* beta, float* y, float* running_mean,
                             float* running_var, float* save_mean, float* save_invstd, int B, int C, int HW,
                             float eps, float momentum, int act, float* y_amax, void* workspace,
                             size_t workspace_bytes, void* stream) {
  if (!x || !gamma || !beta || !y || !save_mean || !save_invstd || B <= 0 || C <= 0 || HW <= 0)
    return VG_ERR_BAD_ARG;
  if (act < VG_ACT_NONE || act > VG_ACT_LRELU) return VG_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (HW == 1) {
    hipLaunchKernelGGL(bn1d_fwd_kernel, dim3(cdiv(C, B1_CH)), dim3(NT), 0, st, x, gamma, beta, y, running_mean,
                       running_var, save_mean, save_invstd, B, C, eps, momentum, act);
    VG_CHECK_LAUNCH();
    return 0;
  }
  if (!workspace || workspace_bytes < ws_bytes(C)) return VG_ERR_WORKSPACE;
  double* part = (double*)workspace;
  const Slicing s = make_slicing(B, C, HW);
  hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(C, s.ns), dim3(NT), 0, st, x, (const float*)nullptr, gamma, beta,
                     (const float*)nullptr, (const float*)nullptr, part, B, C, HW, s.per, s.ns, act);
  VG_CHECK_LAUNCH();
  const Slicing a = make_apply_slicing(B, C, HW);
  hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(C, a.ns), dim3(NT), 0, st, x, (const float*)nullptr,
                     (const double*)part, s.ns, gamma, beta, save_mean, save_invstd, running_mean, running_var,
                     (float*)nullptr, (float*)nullptr, y, B, C, HW, a.per, eps, momentum, act,
                     reinterpret_cast<unsigned*>(y_amax));
  VG_CHECK_LAUNCH();
  return 0;
}

extern "C" int vg_bn_act_bwd(const float* gy, const float* x, const float* gamma, const float* beta,
                             const float* save_mean, const float* save_invstd, float* gx, float* dgamma,
                             float* dbeta, int B, int C, int HW, int act, int accumulate_param_grads, float* gx_amax,
                             void* workspace, size_t workspace_bytes, void* stream) {
  unsigned* am = (unsigned*)gx_amax;
  if (!gy || !x || !gamma || !beta || !save_mean || !save_invstd || !gx || B <= 0 || C <= 0 || HW <= 0)
    return VG_ERR_BAD_ARG;
  if (act < VG_ACT_NONE || act > VG_ACT_LRELU) return VG_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int act_apply = act | (accumulate_param_grads ? 0x100 : 0);      // bit 8: dgamma / dbeta += (kernels decode it)
  if (HW == 1) {
    hipLaunchKernelGGL(bn1d_bwd_kernel, dim3(cdiv(C, B1_CH)), dim3(NT), 0, st, gy, x, gamma, beta, save_mean,
                       save_invstd, gx, dgamma, dbeta, B, C, act_apply, am);
    VG_CHECK_LAUNCH();
    return 0;
  }
  // one pass when a channel fits a workgroup's registers and there are channels enough to fill the chip's memory system
  const long per_channel = (long)B * HW;
#ifdef VG_BN_TWO_PASS                  // timing experiments only (same-box A/B of the one-pass form)
  constexpr bool one_pass = false;
#else
  constexpr bool one_pass = true;
#endif
  if (one_pass && (HW & 3) == 0 && per_channel >= 4 && per_channel <= (long)ONE_NT * 4 * 8 && C >= 128) {
    if (per_channel <= (long)ONE_NT * 4 * 2)
      hipLaunchKernelGGL(bn_bwd_onepass_kernel<2>, dim3(C), dim3(ONE_NT), 0, st, x, gy, gamma, beta, save_mean, save_invstd,
                         gx, dgamma, dbeta, B, C, HW, act_apply, am);
    else
      hipLaunchKernelGGL(bn_bwd_onepass_kernel<8>, dim3(C), dim3(ONE_NT), 0, st, x, gy, gamma, beta, save_mean, save_invstd,
                         gx, dgamma, dbeta, B, C, HW, act_apply, am);
    VG_CHECK_LAUNCH();
    return 0;
  }
  if (!workspace || workspace_bytes < ws_bytes(C)) return VG_ERR_WORKSPACE;
  double* part = (double*)workspace;
  const Slicing s = make_slicing(B, C, HW);
  hipLaunchKernelGGL(bn_partial_kernel<1>, dim3(C, s.ns), dim3(NT), 0, st, x, gy, gamma, beta, save_mean,
                     save_invstd, part, B, C, HW, s.per, s.ns, act);
  VG_CHECK_LAUNCH();
  const Slicing a = make_apply_slicing(B, C, HW);
  hipLaunchKernelGGL(bn_apply_kernel<true>, dim3(C, a.ns), dim3(NT), 0, st, x, gy, (const double*)part, s.ns, gamma,
                     beta, const_cast<float*>(save_mean), const_cast<float*>(save_invstd), (float*)nullptr,
                     (float*)nullptr, dgamma, dbeta, gx, B, C, HW, a.per, 0.f, 0.f, act_apply, am);
  VG_CHECK_LAUNCH();
  return 0;
}

extern "C" int vg_channel_sum(const float* g, float* out, int B, int C, int HW, void* workspace,
                              size_t workspace_bytes, void* stream) {
  if (!g || !out || B <= 0 || C <= 0 || HW <= 0) return VG_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < ws_bytes(C)) return VG_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  double* part = (double*)workspace;
  const Slicing s = make_slicing(B, C, HW);
  hipLaunchKernelGGL(bn_partial_kernel<2>, dim3(C, s.ns), dim3(NT), 0, st, g, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                     part, B, C, HW, s.per, s.ns, 0);
  VG_CHECK_LAUNCH();
  hipLaunchKernelGGL(channel_sum_finalize_kernel, dim3(cdiv(C, NT)), dim3(NT), 0, st, (const double*)part, s.ns,
                     C, out);
  VG_CHECK_LAUNCH();
  return 0;
}
