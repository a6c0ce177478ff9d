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

__global__ __launch_bounds__(ANT) void adam_multi_kernel(AdamPack A, float omb1, float b2, float omb2, float step_size,
                                                        float bc2s, float eps) {
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
      }
      *reinterpret_cast<f32x4*>(p + i) = pv;
      *reinterpret_cast<f32x4*>(m + i) = mv;
      *reinterpret_cast<f32x4*>(v + i) = vv;
    }
    for (unsigned long long i = end4 + threadIdx.x; i < end; i += ANT) adam_one(p[i], g[i], m[i], v[i], omb1, b2, omb2, step_size, bc2s, eps);
  } else {
    for (unsigned long long i = base + threadIdx.x; i < end; i += ANT) adam_one(p[i], g[i], m[i], v[i], omb1, b2, omb2, step_size, bc2s, eps);
  }
}

}  // namespace

extern "C" int vg_adam_step(const VgAdamTensor* tensors, int count, double lr, double beta1, double beta2, double eps,
                            double bias_correction1, double bias_correction2_sqrt, void* stream) {
  if (count < 0 || (count > 0 && !tensors) || !(bias_correction1 > 0.0) || !(bias_correction2_sqrt > 0.0))
    return VG_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  // scalars are formed in double and rounded once, as torch does with its Python-side hyper-parameters
  const float step_size = (float)(lr / bias_correction1);
  const float omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2), bc2s = (float)bias_correction2_sqrt;
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
      A.first_block[k] = blocks;
      blocks += (unsigned)nb;
    }
    if (A.count == 0) break;
    A.first_block[A.count] = blocks;
    hipLaunchKernelGGL(adam_multi_kernel, dim3(blocks), dim3(ANT), 0, st, A, omb1, (float)beta2, omb2, step_size, bc2s,
                       (float)eps);
    VG_CHECK_LAUNCH();
  }
  return 0;
}
