// Upper bounds of max |x| for the fp16-plane arithmetic of the convolution kernels (common.hpp, "split arithmetics";
// DESIGN.md section 2): the kernels scale an operand by an exact power of two taken from such a bound, read from device
// memory.  One pass over the tensor, 16-byte loads, a wavefront-shuffle maximum and ONE atomic per workgroup:
// atomicMax on the bit pattern of |x| (for non-negative IEEE floats the integer order is the float order; a NaN's
// pattern is above every number's, so a NaN in the data ends up in the bound and from there in the consumer's output).
// The maximum does not depend on the order of the atomics: deterministic.  The caller zeroes amax[0] beforehand (a bound
// may also be accumulated over several tensors).  Producers that can emit the bound on their way out (vg_bn_act_bwd's
// gx_amax, vg_bn_finalize_stats' act_amax) make these passes unnecessary for most tensors of a training iteration.
//
// Replaces nothing of the reference (/root/reference/models/model.py:389-398, 450-456, 495-507 run fp32 cuDNN / MKL-DNN
// convolutions): it is what lets those convolutions run in 3 instead of 6 16-bit MFMAs per fp32 multiply.
#include "common.hpp"
#include "vaegan_hip.h"

namespace {

constexpr int ANT = 256;

// n4 whole 16-byte units from x (16-byte aligned), then the tail
__device__ __forceinline__ unsigned absmax_span(const float* __restrict__ x, size_t n, size_t first, size_t stride) {
  unsigned m = 0;
  if (((uintptr_t)x & 15) == 0) {
    const size_t n4 = n / 4;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    size_t i = first;
    for (; i + 3 * stride < n4; i += 4 * stride) {          // four loads in flight
      const f32x4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
#pragma unroll
      for (int e = 0; e < 4; ++e) m = max(max(m, abs_bits(a[e])), max(max(abs_bits(b[e]), abs_bits(c[e])), abs_bits(d[e])));
    }
    for (; i < n4; i += stride) {
      const f32x4 a = x4[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) m = max(m, abs_bits(a[e]));
    }
    for (size_t t = n4 * 4 + first; t < n; t += stride) m = max(m, abs_bits(x[t]));
  } else {
    for (size_t t = first; t < n; t += stride) m = max(m, abs_bits(x[t]));
  }
  return m;
}

__global__ __launch_bounds__(ANT) void absmax_kernel(const float* __restrict__ x, size_t n, unsigned* __restrict__ out) {
  const unsigned m = absmax_span(x, n, (size_t)blockIdx.x * ANT + threadIdx.x, (size_t)gridDim.x * ANT);
  block_amax_atomic<ANT>(m, out);
}

// max |act(x * scale[c] + shift[c])| over x[B][C][HW]: workgroup (c, slice)
__global__ __launch_bounds__(ANT) void absmax_affine_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, float slope, int B, int C,
                                                            int HW, unsigned* __restrict__ out) {
  const int c = blockIdx.x;
  const float sc = scale[c], sh = shift[c];
  const long total = (long)B * HW;
  unsigned m = 0;
  auto one = [&](float v) {
    v = fmaf(v, sc, sh);
    v = act_slope(v, slope);
    m = max(m, abs_bits(v));
  };
  const int hw_shift = (HW & (HW - 1)) == 0 ? __builtin_ctz(HW) : -1;
  if ((HW & 3) == 0) {
    for (long v = 4L * ((long)blockIdx.y * ANT + threadIdx.x); v < total; v += 4L * ANT * gridDim.y) {
      const long b = hw_shift >= 0 ? (v >> hw_shift) : v / HW, hw = v - b * HW;
      const f32x4 a = *reinterpret_cast<const f32x4*>(x + ((size_t)b * C + c) * HW + hw);
      one(a[0]), one(a[1]), one(a[2]), one(a[3]);
    }
  } else {
    for (long v = (long)blockIdx.y * ANT + threadIdx.x; v < total; v += (long)ANT * gridDim.y) {
      const long b = hw_shift >= 0 ? (v >> hw_shift) : v / HW, hw = v - b * HW;
      one(x[((size_t)b * C + c) * HW + hw]);
    }
  }
  block_amax_atomic<ANT>(m, out);
}

constexpr int AM_MAXE = 24;
struct AbsmaxBatch {
  const float* x[AM_MAXE];
  unsigned* out[AM_MAXE];
  size_t n[AM_MAXE];
  unsigned first_block[AM_MAXE + 1];
  int count;
};

__global__ __launch_bounds__(ANT) void absmax_multi_kernel(AbsmaxBatch P) {
  int t = 0;
  while (t + 1 < P.count && blockIdx.x >= P.first_block[t + 1]) ++t;
  const unsigned nb = P.first_block[t + 1] - P.first_block[t], b = blockIdx.x - P.first_block[t];
  const unsigned m = absmax_span(P.x[t], P.n[t], (size_t)b * ANT + threadIdx.x, (size_t)nb * ANT);
  block_amax_atomic<ANT>(m, P.out[t]);
}

unsigned blocks_for(size_t n) {
  const size_t per = (size_t)ANT * 16;                     // >= 4 units of 16 bytes per thread
  size_t b = (n + per - 1) / per;
  return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

extern "C" int vg_absmax(const float* x, size_t n, float* amax, void* stream) {
  if (!x || !amax || n == 0) return VG_ERR_BAD_ARG;
  hipLaunchKernelGGL(absmax_kernel, dim3(blocks_for(n)), dim3(ANT), 0, (hipStream_t)stream, x, n, (unsigned*)amax);
  VG_CHECK_LAUNCH();
  return 0;
}

extern "C" int vg_absmax_affine(const float* x, const float* scale, const float* shift, int act, int B, int C, int HW,
                                float* amax, void* stream) {
  if (!x || !scale || !shift || !amax || B <= 0 || C <= 0 || HW <= 0) return VG_ERR_BAD_ARG;
  if (act < VG_ACT_NONE || act > VG_ACT_LRELU || C > 65535 * 32) return VG_ERR_BAD_ARG;
  const float slope = act == VG_ACT_NONE ? 1.f : (act == VG_ACT_RELU ? 0.f : 0.2f);
  const long total = (long)B * HW;
  long ns = 2048 / C;
  ns = ns < 1 ? 1 : ns;
  const long cap = (total + 4L * ANT * 4 - 1) / (4L * ANT * 4);      // >= 4 loads per thread
  ns = ns > cap ? (cap < 1 ? 1 : cap) : ns;
  hipLaunchKernelGGL(absmax_affine_kernel, dim3(C, (unsigned)ns), dim3(ANT), 0, (hipStream_t)stream, x, scale, shift, slope,
                     B, C, HW, (unsigned*)amax);
  VG_CHECK_LAUNCH();
  return 0;
}

extern "C" int vg_absmax_multi(const VgAbsmaxEntry* entries, int count, void* stream) {
  if (count < 0 || (count > 0 && !entries)) return VG_ERR_BAD_ARG;
  int i = 0;
  while (i < count) {
    AbsmaxBatch P;
    P.count = 0;
    unsigned blocks = 0;
    while (i < count && P.count < AM_MAXE) {
      const VgAbsmaxEntry& E = entries[i++];
      if (!E.x || !E.amax || E.n == 0) return VG_ERR_BAD_ARG;
      const int k = P.count++;
      P.x[k] = E.x; P.out[k] = (unsigned*)E.amax; P.n[k] = E.n;
      P.first_block[k] = blocks;
      const unsigned nb = blocks_for(E.n);
      blocks += nb > 64 ? 64 : nb;                         // filters: <= 6.5 MB each, many side by side
    }
    P.first_block[P.count] = blocks;
    hipLaunchKernelGGL(absmax_multi_kernel, dim3(blocks), dim3(ANT), 0, (hipStream_t)stream, P);
    VG_CHECK_LAUNCH();
  }
  return 0;
}
