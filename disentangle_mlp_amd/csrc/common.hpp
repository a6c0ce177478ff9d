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

__host__ __device__ constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }

// cross-file internal entry points (not part of the C ABI)
int vg_internal_wgrad_reduce(const float* slabs, float* dw, int n, int splits, hipStream_t st, int accumulate = 0);
// conv_ring.hip: stride-2 split-bf16 convolution (mode 0) / transposed convolution (mode 1), 8-wave ring kernel
size_t vg_internal_ring_workspace_bytes(int mode, int B, int Cin, int H, int W, int Cout);
size_t vg_internal_ring_stats_floats(int mode, int B, int Cin, int H, int W, int Cout);
int vg_internal_ring_conv(int mode, const float* x, const void* packed, const float* bias, float* y, int B, int Cin, int H,
                          int W, int Cout, int planes, void* workspace, size_t workspace_bytes, const float* in_scale,
                          const float* in_shift, int in_act, float* stats, size_t stats_floats, hipStream_t st);
#ifdef VG_TUNING
void vg_internal_ring_set_variant(int v);
void vg_internal_wx_set_th(int th);
#endif
int vg_internal_convT_s1_thin(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int H,
                              int W, int Cout, hipStream_t st);
