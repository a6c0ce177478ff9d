// Adam step of the three optimizers (/root/reference/experiments/new_betavaegan.py:49-50:
// optim.Adam defaults -- betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad), SURVEY.md
// section 8 row a14: 182.9 M parameter updates per iteration, 28 bytes each -- purely HBM-bound.
//
//   m <- m + (1 - beta1) (g - m)             (torch: exp_avg.lerp_(grad, 1 - beta1))
//   v <- beta2 v + (1 - beta2) g g           (exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2))
//   p <- p - step_size * m / (sqrt(v) / bias_correction2_sqrt + eps)
//
// One launch updates up to VG_ADAM_MAX_TENSORS tensors: their pointers travel in the kernel
// arguments, a workgroup owns 8192 consecutive elements of one tensor (16-byte loads / stores when
// the four pointers allow it).  Bias corrections are computed on the host in double precision.
#include "common.hpp"
#include "vaegan_hip.h"

namespace {

constexpr int ANT = 256, ACHUNK = 8192, AMAX = 24;

struct AdamPack {
  float* p[AMAX];
  const float* g[AMAX];
  float* m[AMAX];
  float* v[AMAX];
  unsigned long long n[AMAX];
  unsigned* amax[AMAX];               // per tensor, may be NULL: max |p| after the update is added here (atomic max)
  unsigned first_block[AMAX + 1];     // prefix sums of ceil(n / ACHUNK)
  int count;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float omb1, float b2, float omb2,
                                         float step_size, float bc2s, float eps) {
  m = m + omb1 * (g - m);
  v = b2 * v + omb2 * g * g;
  const float denom = sqrtf(v) / bc2s + eps;
  p = p - step_size * (m / denom);
}

// One thread: the scalars of an optimizer step whose step count lives on the device (a step captured in a HIP graph
// cannot take them as kernel arguments: they change from replay to replay).  `advance`: the device counter is advanced
// by one and used (captured steps); otherwise `step_host` is used and, when there is a device counter, stored in it (an
// eager step between replays keeps the counter current).  Both ways the bias corrections are formed here, in double,
// from the same expression -- an eager step and a replayed one give the same bits.
__global__ void adam_prepare_kernel(double step_host, double* __restrict__ step_dev, int advance, double lr, double beta1,
                                    double beta2, float* __restrict__ scalars) {
  double step = step_host;
  if (advance) step = step_dev[0] + 1.0;
  if (step_dev) step_dev[0] = step;
  const double bc1 = 1.0 - pow(beta1, step);
  const double bc2 = 1.0 - pow(beta2, step);
  scalars[0] = (float)(lr / bc1);       // step_size
  scalars[1] = (float)sqrt(bc2);        // bias_correction2_sqrt
}

template <bool DEV>
__global__ __launch_bounds__(ANT) void adam_multi_kernel(AdamPack A, float omb1, float b2, float omb2, float step_size,
                                                        float bc2s, float eps, const float* __restrict__ scalars) {
  if constexpr (DEV) {
    step_size = scalars[0];
    bc2s = scalars[1];
  }
  int t = 0;
  while (t + 1 < A.count && blockIdx.x >= A.first_block[t + 1]) ++t;
  const unsigned long long n = A.n[t];
  const unsigned long long base = (unsigned long long)(blockIdx.x - A.first_block[t]) * ACHUNK;
  const unsigned long long end = min(base + (unsigned long long)ACHUNK, n);
  float* __restrict__ p = A.p[t];
  const float* __restrict__ g = A.g[t];
  float* __restrict__ m = A.m[t];
  float* __restrict__ v = A.v[t];
  const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
  // bound of max |p| for the fp16-plane GEMMs that read this weight next (VgAdamTensor::amax): the step that changes the
  // weight is the one pass that sees every new value anyway -- no separate 134 MB read per weight and iteration
  unsigned am = 0;
  if (vec) {
    const unsigned long long end4 = base + ((end - base) & ~3ULL);
    for (unsigned long long i = base + 4ULL * threadIdx.x; i < end4; i += 4ULL * ANT) {
      f32x4 pv = *reinterpret_cast<f32x4*>(p + i), mv = *reinterpret_cast<f32x4*>(m + i);
      f32x4 vv = *reinterpret_cast<f32x4*>(v + i);
      const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float pj = pv[j], mj = mv[j], vj = vv[j];
        adam_one(pj, gv[j], mj, vj, omb1, b2, omb2, step_size, bc2s, eps);
        pv[j] = pj; mv[j] = mj; vv[j] = vj;
        am = max(am, abs_bits(pj));
      }
      *reinterpret_cast<f32x4*>(p + i) = pv;
      *reinterpret_cast<f32x4*>(m + i) = mv;
      *reinterpret_cast<f32x4*>(v + i) = vv;
    }
    for (unsigned long long i = end4 + threadIdx.x; i < end; i += ANT) {
      adam_one(p[i], g[i], m[i], v[i], omb1, b2, omb2, step_size, bc2s, eps);
      am = max(am, abs_bits(p[i]));
    }
  } else {
    for (unsigned long long i = base + threadIdx.x; i < end; i += ANT) {
      adam_one(p[i], g[i], m[i], v[i], omb1, b2, omb2, step_size, bc2s, eps);
      am = max(am, abs_bits(p[i]));
    }
  }
  if (A.amax[t]) block_amax_atomic<ANT>(am, A.amax[t]);      // (t is uniform over the workgroup)
}

}  // namespace

namespace {
int adam_launch(const VgAdamTensor* tensors, int count, double beta1, double beta2, double eps, float step_size, float bc2s,
                const float* scalars, hipStream_t st) {
  const float omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
  int i = 0;
  while (i < count) {
    AdamPack A;
    A.count = 0;
    unsigned blocks = 0;
    while (i < count && A.count < AMAX) {
      const VgAdamTensor& T = tensors[i++];
      if (T.n == 0) continue;
      if (!T.p || !T.g || !T.m || !T.v) return VG_ERR_BAD_ARG;
      const unsigned long long nb = (T.n + ACHUNK - 1) / ACHUNK;
      if (nb > 0x3fffffffULL - blocks) return VG_ERR_BAD_ARG;
      const int k = A.count++;
      A.p[k] = T.p; A.g[k] = T.g; A.m[k] = T.m; A.v[k] = T.v; A.n[k] = T.n;
      A.amax[k] = reinterpret_cast<unsigned*>(T.amax);
      A.first_block[k] = blocks;
      blocks += (unsigned)nb;
    }
    if (A.count == 0) break;
    A.first_block[A.count] = blocks;
    if (scalars)
      hipLaunchKernelGGL(adam_multi_kernel<true>, dim3(blocks), dim3(ANT), 0, st, A, omb1, (float)beta2, omb2, 0.f, 0.f,
                         (float)eps, scalars);
    else
      hipLaunchKernelGGL(adam_multi_kernel<false>, dim3(blocks), dim3(ANT), 0, st, A, omb1, (float)beta2, omb2, step_size,
                         bc2s, (float)eps, (const float*)nullptr);
    VG_CHECK_LAUNCH();
  }
  return 0;
}
}  // namespace

extern "C" int vg_adam_step(const VgAdamTensor* tensors, int count, double lr, double beta1, double beta2, double eps,
                            double bias_correction1, double bias_correction2_sqrt, void* stream) {
  if (count < 0 || (count > 0 && !tensors) || !(bias_correction1 > 0.0) || !(bias_correction2_sqrt > 0.0))
    return VG_ERR_BAD_ARG;
  // scalars are formed in double and rounded once, as torch does with its Python-side hyper-parameters
  return adam_launch(tensors, count, beta1, beta2, eps, (float)(lr / bias_correction1), (float)bias_correction2_sqrt,
                     nullptr, (hipStream_t)stream);
}

extern "C" int vg_adam_prepare(double step, double* step_dev, int advance_device_counter, double lr, double beta1,
                               double beta2, float* scalars, void* stream) {
  if (!scalars || (advance_device_counter ? !step_dev : !(step >= 1.0)) || !(beta1 >= 0.0 && beta1 < 1.0) ||
      !(beta2 >= 0.0 && beta2 < 1.0))
    return VG_ERR_BAD_ARG;
  hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step, step_dev,
                     advance_device_counter ? 1 : 0, lr, beta1, beta2, scalars);
  VG_CHECK_LAUNCH();
  return 0;
}

extern "C" int vg_adam_step_dev(const VgAdamTensor* tensors, int count, double beta1, double beta2, double eps,
                                const float* scalars, void* stream) {
  if (count < 0 || (count > 0 && !tensors) || !scalars) return VG_ERR_BAD_ARG;
  return adam_launch(tensors, count, beta1, beta2, eps, 0.f, 0.f, scalars, (hipStream_t)stream);
}
