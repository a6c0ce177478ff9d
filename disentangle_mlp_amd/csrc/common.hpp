// Shared device helpers for the gfx950 (MI355X / CDNA4) beta-VAE-GAN kernels.
// 64-wide wavefronts, fp32-input MFMA (v_mfma_f32_32x32x2_f32: exact fp32,
// bitwise a k-ordered fmaf chain), LDS-staged NCHW patches.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define VG_WAVE 64

// Tuning knobs (forced tile variants, split targets ...): mutable process-globals exist ONLY in the tuning build
// (-DVG_TUNING -> libvaegan_hip_tuning.so, loaded by tests/ and scripts/ through _lib.load_tuning()); in the product
// library they are compile-time constants and the vg_debug_* setters are absent: it holds no global mutable state
// (include/vaegan_hip.h, "Conventions").
#ifdef VG_TUNING
#define VG_KNOB(type, name, value) type name = value
#else
#define VG_KNOB(type, name, value) constexpr type name = value
#endif

#define VG_CHECK_LAUNCH()                         \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return (int)e__;       \
  } while (0)

// D(32x32) += A(32x2) * B(2x32).  Lane l supplies A[i=l&31][k=l>>5] and
// B[k=l>>5][j=l&31]; D register r of lane l is D[i=(r&3)+8*(r>>2)+4*(l>>5)][j=l&31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int acc_row(int r, int lane) {
  return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}

// Activation applied on load: identity (slope 1), ReLU (0), LeakyReLU (0.2).  One compare, one select, one multiply,
// no branch -- and a NaN stays a NaN, as in torch (fmaxf / fminf drop it: max(NaN, 0) = 0).
__device__ __forceinline__ float act_slope(float v, float slope) { return v * (v > 0.f ? 1.f : slope); }

// ---- split arithmetics of the convolution kernels -------------------------------------------
// An fp32 operand becomes NP 16-bit planes; the plane products whose indices sum to < NP go to the 16-bit MFMA with
// fp32 accumulation.  bf16 planes (F16 = false): each plane takes the leading 8 significand bits of what is left --
// 3 planes are the whole fp32 significand, full fp32 range, 6 products.  fp16 planes (F16 = true, NP = 2): hi + lo carry
// 11 + 11 significand bits (residual <= 2^-24 relative), 3 products; fp16 has 5 exponent bits, so the operand is first
// multiplied by an exact power of two that puts the tensor's largest magnitude just under 2^15 (f16_scale_of of an
// UPPER BOUND of max|x| read from device memory), and the product of the two operands' inverse scales is applied to
// the fp32 accumulators at the end.  Elements more than 2^-16 below the bound keep fewer than 22 bits (the lo plane
// runs into fp16's subnormals, which the MFMA honours: profiles/r04_mfma_f16.jsonl): their error is bounded by 2^-40
// of the tensor's bound instead of 2^-24 of themselves.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

#define VG_PLANES_F16_FLAG 0x100      // == VG_PLANES_F16 of vaegan_hip.h

// biased fp32 exponent of an upper bound `amax` >= 0, clamped so that both 2^(14 - E) and 2^(E - 14) are normal fp32
// numbers (amax = 0 or subnormal: the largest scale; inf / NaN: the smallest -- the data then carries the inf / NaN)
__device__ __forceinline__ unsigned f16_bound_exp(float amax) {
  const unsigned e = (__float_as_uint(amax) >> 23) & 0xffu;
  return min(max(e, 15u), 254u);
}
// amax in [2^E, 2^(E+1))  ->  2^(14 - E): the scaled tensor lies in (-2^15, 2^15), fp16's largest finite value is 65504
__device__ __forceinline__ float f16_scale_of(float amax) { return __uint_as_float((268u - f16_bound_exp(amax)) << 23); }
__device__ __forceinline__ float f16_unscale_of(float amax) { return __uint_as_float((f16_bound_exp(amax) - 14u) << 23); }

// v[8] (already scaled when F16) -> NP planes as raw 16-byte units; v is consumed
template <int NP, bool F16>
__device__ __forceinline__ void split_planes16(float* v, f32x4* out) {
  if constexpr (F16) {
    static_assert(NP == 2, "fp16 planes: hi + lo");
    // Two elements at a time, as 2-wide vectors: v_cvt_pk_f16_f32 (round to nearest even, two floats -> one packed
    // register, no separate pack), v_pk_add_f32 for the residual -- 2.5 VALU per element where the scalar form
    // (cvt, cvt back, sub, and a cvt_pk each for packing hi and lo) took 4.  Same roundings: same bits.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    f16x2 hi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x2 v2 = {v[2 * j], v[2 * j + 1]};
      const f16x2 h = __builtin_convertvector(v2, f16x2);                 // |v - h| <= 2^-12 |v|
      hi[j] = h;
      lo[j] = __builtin_convertvector(v2 - __builtin_convertvector(h, f32x2), f16x2);   // exact difference, 11 more bits
    }
    struct P4 { f16x2 q[4]; };
    out[0] = __builtin_bit_cast(f32x4, P4{{hi[0], hi[1], hi[2], hi[3]}});
    out[1] = __builtin_bit_cast(f32x4, P4{{lo[0], lo[1], lo[2], lo[3]}});
  } else {
#pragma unroll
    for (int p = 0; p < NP; ++p) {                // hi, (mid,) lo: each plane takes the leading 8 bits of what is left
      bf16x8 q;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)v[j];
        q[j] = h;
        v[j] -= (float)h;
      }
      out[p] = __builtin_bit_cast(f32x4, q);
    }
  }
}

template <bool F16>
__device__ __forceinline__ f32x16 mfma_split16(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ---- wavefront / block reductions (64 lanes; no 32-lane idioms) -------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;  // valid in lane 0
}

template <typename T>
__device__ __forceinline__ T wave_allsum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Sum over a block of NT threads (NT multiple of 64); result valid in thread 0.
// `red` is LDS scratch of at least NT/64 elements.
template <int NT, typename T>
__device__ __forceinline__ T block_sum(T v, T* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  T t = 0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
  }
  return t;
}

// Bound of max |x| for the fp16-plane arithmetic: bit pattern of |v| (integer order == float order for non-negative
// floats; NaN patterns sort above every number), and the workgroup's maximum added to a device word with ONE atomic
// (order-independent: deterministic).  Every thread of the NT-thread workgroup must call it.
__device__ __forceinline__ unsigned abs_bits(float v) { return __float_as_uint(v) & 0x7fffffffu; }

template <int NT>
__device__ __forceinline__ void block_amax_atomic(unsigned m, unsigned* out) {
  __shared__ unsigned amax_red[NT / 64];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) amax_red[wid] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) m = max(m, amax_red[i]);
    // most workgroups find the word already above their maximum: a relaxed load (possibly stale: then the atomic runs
    // anyway) keeps thousands of them from queueing on one address (16 384 atomics made a 20 us pass take 190)
    if (m > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, m);
  }
}

__host__ __device__ constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }

// cross-file internal entry points (not part of the C ABI)
int vg_internal_wgrad_reduce(const float* slabs, float* dw, int n, int splits, hipStream_t st, int accumulate = 0);
// conv_ring.hip: stride-2 split-bf16 convolution (mode 0) / transposed convolution (mode 1), 8-wave ring kernel
size_t vg_internal_ring_workspace_bytes(int mode, int B, int Cin, int H, int W, int Cout, int planes);
size_t vg_internal_ring_stats_floats(int mode, int B, int Cin, int H, int W, int Cout, int planes);
int vg_internal_ring_conv(int mode, const float* x, const void* packed, const float* bias, float* y, int B, int Cin, int H,
                          int W, int Cout, int planes, void* workspace, size_t workspace_bytes, const float* in_scale,
                          const float* in_shift, int in_act, float* stats, size_t stats_floats, const float* in_amax,
                          hipStream_t st);       // planes may carry VG_PLANES_F16 (then in_amax is required)
#ifdef VG_TUNING
void vg_internal_ring_set_variant(int v);
void vg_internal_wx_set_th(int th);
#endif
int vg_internal_convT_s1_thin(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int H,
                              int W, int Cout, hipStream_t st);
