// Fused loss / elementwise kernels of the beta-VAE-GAN iteration for gfx950.
// All HBM- or latency-bound: float4 coalesced streams, 64-lane wavefront shuffle
// reductions, fp64 partials, deterministic two-stage sums, loss scalars written to
// device memory (no host synchronisation inside the step).
//
//   reparam + KL   /root/reference/models/model.py:532-535 + experiments/new_betavaegan.py:64-65
//   Dis_l / SIM    new_betavaegan.py:67-69       pixel MSE  new_betavaegan.py:71-75
//   BCE            new_betavaegan.py:53,101,118,153-154 (nn.BCELoss, log clamp -100)
//   LeakyReLU/tanh/sigmoid epilogues  model.py:404, 509/565, 408
#include "common.hpp"
#include "vaegan_hip.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float ew_fwd(float v, int kind) {
  if (kind == VG_EW_LRELU) return v > 0.f ? v : 0.2f * v;
  if (kind == VG_EW_TANH) return tanhf(v);
  return 1.f / (1.f + expf(-v));
}
__device__ __forceinline__ float ew_bwd(float g, float y, int kind) {
  if (kind == VG_EW_LRELU) return y > 0.f ? g : 0.2f * g;
  if (kind == VG_EW_TANH) return g * (1.f - y * y);
  return g * y * (1.f - y);
}

__global__ __launch_bounds__(NT) void bias_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                                          float* __restrict__ y, int C, int HW, size_t n, int kind) {
  const size_t stride = (size_t)gridDim.x * NT;
  if ((HW & 3) == 0) {
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += stride) {
      const float b = bias ? bias[((i << 2) / HW) % C] : 0.f;
      const float4 v = reinterpret_cast<const float4*>(x)[i];
      float4 o;
      o.x = ew_fwd(v.x + b, kind);
      o.y = ew_fwd(v.y + b, kind);
      o.z = ew_fwd(v.z + b, kind);
      o.w = ew_fwd(v.w + b, kind);
      reinterpret_cast<float4*>(y)[i] = o;
    }
  } else {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) {
      const float b = bias ? bias[(i / HW) % C] : 0.f;
      y[i] = ew_fwd(x[i] + b, kind);
    }
  }
}

__global__ __launch_bounds__(NT) void act_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                     float* __restrict__ gx, size_t n, int kind,
                                                     unsigned* __restrict__ gx_amax) {
  const size_t stride = (size_t)gridDim.x * NT;
  const size_t n4 = n >> 2;
  unsigned am = 0;      // max |gx| for an fp16-plane consumer (the Linear data / weight-gradient GEMMs): vg_act_bwd
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += stride) {
    const float4 g = reinterpret_cast<const float4*>(gy)[i];
    const float4 v = reinterpret_cast<const float4*>(y)[i];
    float4 o;
    o.x = ew_bwd(g.x, v.x, kind);
    o.y = ew_bwd(g.y, v.y, kind);
    o.z = ew_bwd(g.z, v.z, kind);
    o.w = ew_bwd(g.w, v.w, kind);
    am = max(max(am, abs_bits(o.x)), max(max(abs_bits(o.y), abs_bits(o.z)), abs_bits(o.w)));
    reinterpret_cast<float4*>(gx)[i] = o;
  }
  for (size_t i = (n4 << 2) + (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) {
    const float o = ew_bwd(gy[i], y[i], kind);
    am = max(am, abs_bits(o));
    gx[i] = o;
  }
  if (gx_amax) block_amax_atomic<NT>(am, gx_amax);
}

// One workgroup of 1024 threads (16 wavefronts); a wavefront owns a row at a time,
// so the per-sample KL is one 64-lane shuffle reduction and the total is summed in a
// fixed order.
__global__ __launch_bounds__(1024) void reparam_kl_fwd_kernel(const float* __restrict__ mu,
                                                              const float* __restrict__ lv,
                                                              const float* __restrict__ eps, float* __restrict__ z,
                                                              float* __restrict__ kl, float* __restrict__ kl_rows,
                                                              int B, int D, float beta) {
  __shared__ double red[16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  double wave_total = 0.0;
  for (int b = wid; b < B; b += 16) {
    float s = 0.f;
    for (int j = lane; j < D; j += 64) {
      const size_t o = (size_t)b * D + j;
      const float m = mu[o], l = lv[o];
      if (z) z[o] = fmaf(eps[o], expf(0.5f * l), m);
      s += 1.f + l - m * m - expf(l);
    }
    s = wave_allsum(s);
    const float row = -0.5f * s;
    if (kl_rows && lane == 0) kl_rows[b] = row;
    wave_total += row;
  }
  if (lane == 0) red[wid] = wave_total;
  __syncthreads();
  if (threadIdx.x == 0 && kl) {
    double t = 0.0;
    for (int i = 0; i < 16; ++i) t += red[i];
    kl[0] = (float)(beta * t);
  }
}

__global__ __launch_bounds__(NT) void reparam_kl_bwd_kernel(const float* __restrict__ gz, const float* __restrict__ mu,
                                                            const float* __restrict__ lv,
                                                            const float* __restrict__ eps, const float* __restrict__ gkl, float beta,
                                                            float* __restrict__ gmu, float* __restrict__ glv,
                                                            size_t n) {
  const size_t stride = (size_t)gridDim.x * NT;
  const float kb = (gkl ? gkl[0] : 0.f) * beta;
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) {
    const float g = gz ? gz[i] : 0.f;
    const float l = lv[i];
    gmu[i] = fmaf(kb, mu[i], g);
    glv[i] = g * eps[i] * 0.5f * expf(0.5f * l) + kb * 0.5f * (expf(l) - 1.f);
  }
}

__global__ __launch_bounds__(NT) void sqdiff_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            float* __restrict__ ga, double* __restrict__ part,
                                                            size_t n, float gmul) {
  __shared__ double red[NT / 64];
  const size_t stride = (size_t)gridDim.x * NT;
  const size_t n4 = n >> 2;
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += stride) {
    const float4 x = reinterpret_cast<const float4*>(a)[i];
    const float4 y = reinterpret_cast<const float4*>(b)[i];
    const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
    s += (double)(d0 * d0) + (double)(d1 * d1) + (double)(d2 * d2) + (double)(d3 * d3);
    if (ga) reinterpret_cast<float4*>(ga)[i] = make_float4(gmul * d0, gmul * d1, gmul * d2, gmul * d3);
  }
  for (size_t i = (n4 << 2) + (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) {
    const float d = a[i] - b[i];
    s += (double)(d * d);
    if (ga) ga[i] = gmul * d;
  }
  const double t = block_sum<NT>(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

__global__ __launch_bounds__(NT) void sum_finalize_kernel(const double* __restrict__ part, int nparts,
                                                          float* __restrict__ out, float scale) {
  __shared__ double red[NT / 64];
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += NT) s += part[i];
  const double t = block_sum<NT>(s, red);
  if (threadIdx.x == 0) out[0] = (float)(scale * t);
}

__global__ __launch_bounds__(NT) void bce_kernel(const float* __restrict__ p, float target,
                                                 const float* __restrict__ target_dev, float* __restrict__ loss,
                                                 float* __restrict__ gp, int B, float inv_div, float gscale) {
  __shared__ double red[NT / 64];
  if (target_dev) target = target_dev[0];       // a label that changes between replays of a captured iteration
  double s = 0.0;
  for (int i = threadIdx.x; i < B; i += NT) {
    const float v = p[i];
    const float lp = fmaxf(logf(v), -100.f), l1p = fmaxf(logf(1.f - v), -100.f);
    s += -(double)(target * lp + (1.f - target) * l1p);
    if (gp) gp[i] = gscale * inv_div * (v - target) / fmaxf(v * (1.f - v), 1e-12f);
  }
  const double t = block_sum<NT>(s, red);
  if (threadIdx.x == 0 && loss) loss[0] = (float)(t * inv_div);
}

// K11 of SURVEY.md section 2.1: the discriminator's head -- Linear(2048 -> 1) + Sigmoid (model.py:406-408) -- and
// nn.BCELoss against the iteration's label (new_betavaegan.py:53,101,118,153-154) in ONE launch each way, instead of a
// GEMV, a sigmoid, a BCE kernel (forward) and a scale, a sigmoid', a bias sum and two GEMMs (backward).
// Forward: a wavefront owns a row (32 elements per lane as 16-byte loads, one 64-lane shuffle sum), 16 rows per
// workgroup; the row terms of the loss go to the workspace in double and are added in a fixed order by the sum kernel that
// follows (one workgroup on one CU for the whole 1 MB of features took as long as the three kernels it replaced).  Same
// expressions, in the same fp32 operations, as bias_act (sigmoid), bce_kernel and act_bwd: p, the loss, and
// dlogit = dBCE/dp * p (1 - p).
__global__ __launch_bounds__(1024) void dot_sigmoid_bce_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ w,
                                                                   const float* __restrict__ bias, float target,
                                                                   const float* __restrict__ target_dev,
                                                                   float* __restrict__ p_out, double* __restrict__ terms,
                                                                   float* __restrict__ dlogit, int B, int K, float inv_div) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (target_dev) target = target_dev[0];
  const float b0 = bias ? bias[0] : 0.f;
  const int b = blockIdx.x * 16 + wid;                 // one row per wavefront, 16 rows per workgroup
  if (b >= B) return;
  const float* row = feat + (size_t)b * K;
  float s = 0.f;
  if ((K & 3) == 0) {
    for (int j = 4 * lane; j < K; j += 256) {
      const float4 x = *reinterpret_cast<const float4*>(row + j);
      const float4 v = *reinterpret_cast<const float4*>(w + j);
      s = fmaf(x.x, v.x, s);
      s = fmaf(x.y, v.y, s);
      s = fmaf(x.z, v.z, s);
      s = fmaf(x.w, v.w, s);
    }
  } else {
    for (int j = lane; j < K; j += 64) s = fmaf(row[j], w[j], s);
  }
  s = wave_allsum(s);
  if (lane == 0) {
    const float v = 1.f / (1.f + expf(-(s + b0)));
    const float lp = fmaxf(logf(v), -100.f), l1p = fmaxf(logf(1.f - v), -100.f);
    p_out[b] = v;
    terms[b] = -(double)(target * lp + (1.f - target) * l1p);
    if (dlogit) dlogit[b] = (inv_div * (v - target) / fmaxf(v * (1.f - v), 1e-12f)) * v * (1.f - v);
  }
}

// Backward: a workgroup (16 wavefronts) owns 64 columns of the 2048; wavefront q takes rows q, q + 16, ... with four rows'
// loads in flight: gfeat[b][k] = g dlogit[b] w[k] (256 contiguous bytes per row), gw[k] = g sum_b dlogit[b] feat[b][k]
// summed over the sixteen wavefronts in a fixed order, gb = g sum_b dlogit[b] by workgroup 0.  g = the upstream gradient
// of the loss scalar (device memory).  (Four wavefronts walking 32 rows each, one row at a time, took 20 us.)
constexpr int DSB_NT = 1024, DSB_NW = DSB_NT / 64;
__global__ __launch_bounds__(DSB_NT) void dot_sigmoid_bce_bwd_kernel(const float* __restrict__ dlogit,
                                                                     const float* __restrict__ gloss,
                                                                     const float* __restrict__ feat, const float* __restrict__ w,
                                                                     float* __restrict__ gfeat, float* __restrict__ gw,
                                                                     float* __restrict__ gb, int B, int K, int accumulate) {
  __shared__ float part[DSB_NW][64];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  const int kc = min(k, K - 1);
  const float g = gloss ? gloss[0] : 1.f;
  const float wk = w[kc];
  float acc = 0.f;
  for (int b0 = q; b0 < B; b0 += 4 * DSB_NW) {
    float d[4], f[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int b = min(b0 + u * DSB_NW, B - 1);
      d[u] = g * dlogit[b];
      f[u] = gw ? feat[(size_t)b * K + kc] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int b = b0 + u * DSB_NW;
      if (b < B && k < K) {
        if (gfeat) gfeat[(size_t)b * K + k] = d[u] * wk;
        acc = fmaf(d[u], f[u], acc);
      }
    }
  }
  part[q][lane] = acc;
  __syncthreads();
  if (q == 0 && gw && k < K) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < DSB_NW; ++i) t += part[i][lane];
    gw[k] = t + (accumulate ? gw[k] : 0.f);      // accumulate: the head is used twice before one backward
  }
  if (blockIdx.x == 0 && gb && q == 1) {           // (a wavefront that is not the one writing gw)
    float t = 0.f;
    for (int b = lane; b < B; b += 64) t += g * dlogit[b];
    t = wave_allsum(t);
    if (lane == 0) gb[0] = t + (accumulate ? gb[0] : 0.f);
  }
}

__global__ __launch_bounds__(NT) void scale_by_scalar_kernel(const float* g, const float* __restrict__ s, float* out,
                                                             size_t n) {
  const float v = s[0];
  const size_t stride = (size_t)gridDim.x * NT;
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += stride) {
    float4 t = reinterpret_cast<const float4*>(g)[i];
    t.x *= v; t.y *= v; t.z *= v; t.w *= v;
    reinterpret_cast<float4*>(out)[i] = t;
  }
  for (size_t i = (n4 << 2) + (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) out[i] = g[i] * v;
}

int flat_grid(size_t n) {
  const size_t blocks = (n / 4 + NT - 1) / NT;
  return (int)(blocks > 2048 ? 2048 : (blocks < 1 ? 1 : blocks));
}
constexpr int SQ_PARTS = 1024;

}  // namespace

extern "C" int vg_version(void) { return VG_ABI_VERSION; }

extern "C" int vg_bias_act_fwd(const float* x, const float* bias, float* y, int B, int C, int HW, int act_kind,
                               void* stream) {
  if (!x || !y || B <= 0 || C <= 0 || HW <= 0 || act_kind < 0 || act_kind > 2) return VG_ERR_BAD_ARG;
  const size_t n = (size_t)B * C * HW;
  hipLaunchKernelGGL(bias_act_fwd_kernel, dim3(flat_grid(n)), dim3(NT), 0, (hipStream_t)stream, x, bias, y, C, HW, n,
                     act_kind);
  VG_CHECK_LAUNCH();
  return 0;
}

extern "C" int vg_act_bwd(const float* gy, const float* y, float* gx, size_t n, int act_kind, float* gx_amax,
                          void* stream) {
  if (!gy || !y || !gx || n == 0 || act_kind < 0 || act_kind > 2) return VG_ERR_BAD_ARG;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(flat_grid(n)), dim3(NT), 0, (hipStream_t)stream, gy, y, gx, n, act_kind,
                     reinterpret_cast<unsigned*>(gx_amax));
  VG_CHECK_LAUNCH();
  return 0;
}

extern "C" int vg_reparam_kl_fwd(const float* mu, const float* logvar, const float* eps, float* z, float* kl,
                                 float* kl_rows, int B, int D, float beta, void* stream) {
  if (!mu || !logvar || B <= 0 || D <= 0 || (z && !eps)) return VG_ERR_BAD_ARG;
  hipLaunchKernelGGL(reparam_kl_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, mu, logvar, eps, z, kl,
                     kl_rows, B, D, beta);
  VG_CHECK_LAUNCH();
  return 0;
}

extern "C" int vg_reparam_kl_bwd(const float* gz, const float* mu, const float* logvar, const float* eps,
                                 const float* gkl, float beta, float* gmu, float* glogvar, int B, int D, void* stream) {
  if (!mu || !logvar || !eps || !gmu || !glogvar || B <= 0 || D <= 0) return VG_ERR_BAD_ARG;
  const size_t n = (size_t)B * D;
  hipLaunchKernelGGL(reparam_kl_bwd_kernel, dim3(flat_grid(n * 4)), dim3(NT), 0, (hipStream_t)stream, gz, mu, logvar,
                     eps, gkl, beta, gmu, glogvar, n);
  VG_CHECK_LAUNCH();
  return 0;
}

extern "C" int vg_scale_by_scalar(const float* g, const float* s, float* out, size_t n, void* stream) {
  if (!g || !s || !out || n == 0) return VG_ERR_BAD_ARG;
  hipLaunchKernelGGL(scale_by_scalar_kernel, dim3(flat_grid(n)), dim3(NT), 0, (hipStream_t)stream, g, s, out, n);
  VG_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t vg_sqdiff_workspace_bytes(size_t n) { return n ? SQ_PARTS * sizeof(double) : 0; }

extern "C" int vg_sqdiff_loss(const float* a, const float* b, float* loss, float* ga, size_t n, float scale,
                              float gscale, void* workspace, size_t workspace_bytes, void* stream) {
  if (!a || !b || !loss || n == 0) return VG_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < SQ_PARTS * sizeof(double)) return VG_ERR_WORKSPACE;
  int grid = flat_grid(n);
  if (grid > SQ_PARTS) grid = SQ_PARTS;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sqdiff_partial_kernel, dim3(grid), dim3(NT), 0, st, a, b, ga, (double*)workspace, n,
                     gscale * 2.f * scale);
  VG_CHECK_LAUNCH();
  hipLaunchKernelGGL(sum_finalize_kernel, dim3(1), dim3(NT), 0, st, (const double*)workspace, grid, loss, scale);
  VG_CHECK_LAUNCH();
  return 0;
}

extern "C" int vg_bce_loss(const float* p, float target, float* loss, float* gp, int B, float divisor, float gscale,
                           void* stream) {
  if (!p || B <= 0 || !(divisor > 0.f)) return VG_ERR_BAD_ARG;
  hipLaunchKernelGGL(bce_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, p, target, (const float*)nullptr, loss, gp, B,
                     1.f / divisor, gscale);
  VG_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t vg_dot_sigmoid_bce_workspace_bytes(int B) { return B > 0 ? (size_t)B * sizeof(double) : 0; }

extern "C" int vg_dot_sigmoid_bce_fwd(const float* feat, const float* w, const float* bias, float target,
                                      const float* target_dev, float* p, float* loss, float* dlogit, int B, int K,
                                      float divisor, void* workspace, size_t workspace_bytes, void* stream) {
  if (!feat || !w || !p || !loss || B <= 0 || K <= 0 || !(divisor > 0.f)) return VG_ERR_BAD_ARG;
  if ((K & 3) == 0 && ((((uintptr_t)feat | (uintptr_t)w) & 15) != 0)) return VG_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < vg_dot_sigmoid_bce_workspace_bytes(B)) return VG_ERR_WORKSPACE;
  double* terms = (double*)workspace;
  hipLaunchKernelGGL(dot_sigmoid_bce_fwd_kernel, dim3(cdiv(B, 16)), dim3(1024), 0, (hipStream_t)stream, feat, w, bias, target,
                     target_dev, p, terms, dlogit, B, K, 1.f / divisor);
  VG_CHECK_LAUNCH();
  hipLaunchKernelGGL(sum_finalize_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, (const double*)terms, B, loss,
                     1.f / divisor);
  VG_CHECK_LAUNCH();
  return 0;
}

extern "C" int vg_dot_sigmoid_bce_bwd(const float* dlogit, const float* gloss, const float* feat, const float* w,
                                      float* gfeat, float* gw, float* gb, int B, int K, int accumulate_param_grads,
                                      void* stream) {
  if (!dlogit || !w || B <= 0 || K <= 0 || (gw && !feat) || (!gfeat && !gw && !gb)) return VG_ERR_BAD_ARG;
  hipLaunchKernelGGL(dot_sigmoid_bce_bwd_kernel, dim3(cdiv(K, 64)), dim3(DSB_NT), 0, (hipStream_t)stream, dlogit, gloss, feat,
                     w, gfeat, gw, gb, B, K, accumulate_param_grads ? 1 : 0);
  VG_CHECK_LAUNCH();
  return 0;
}

extern "C" int vg_bce_loss_dev(const float* p, const float* target_dev, float* loss, float* gp, int B, float divisor,
                               float gscale, void* stream) {
  if (!p || !target_dev || B <= 0 || !(divisor > 0.f)) return VG_ERR_BAD_ARG;
  hipLaunchKernelGGL(bce_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, p, 0.f, target_dev, loss, gp, B, 1.f / divisor,
                     gscale);
  VG_CHECK_LAUNCH();
  return 0;
}
