// On-box f16 MFMA check for the 2-plane fp32-equivalent arithmetic (`fp16x3`): (1) are subnormal f16 inputs honoured
// by v_mfma_f32_32x32x16_f16 / v_mfma_f32_16x16x32_f16 (the lo planes live there), (2) what rate and clock the f16
// shapes hold on random operands next to the bf16 ones (scripts/mfma_peak.hip measures those).  Standalone; built and
// run on the GPU box by scripts/mfma_f16.sh -> one JSON line per case.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

// ---- (1) one 32 x 32 x 16 product: A[i][k] (row-major 32 x 16 halfs), B[k][j] (16 x 32), D fp32 32 x 32
__global__ void one_mfma(const _Float16* A, const _Float16* B, float* D, float* D16) {
  const int lane = threadIdx.x, i = lane & 31, kb = lane >> 5;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = A[i * 16 + kb * 8 + e];
    b[e] = B[(kb * 8 + e) * 32 + i];
  }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * kb) * 32 + i] = c[r];
  // 16 x 16 x 32: lane supplies A[i16][k = 8 * (lane >> 4) ..], B[k][j16]; use rows / columns 0..15, K = first 16 of A
  // twice (k-blocks 2, 3 get zeros) so the expected result is the 16 x 16 corner of the same product
  const int i16 = lane & 15, k4 = lane >> 4;
  f16x8 a2, b2;
  for (int e = 0; e < 8; ++e) {
    a2[e] = k4 < 2 ? A[i16 * 16 + k4 * 8 + e] : (_Float16)0.f;
    b2[e] = k4 < 2 ? B[(k4 * 8 + e) * 32 + i16] : (_Float16)0.f;
  }
  f32x4 c2 = {0.f, 0.f, 0.f, 0.f};
  c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b2, c2, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D16[(4 * k4 + r) * 16 + i16] = c2[r];
}

// fp32 -> (hi, lo) f16 split as the kernels would do it, checked against the host
__global__ void split_kernel(const float* x, float scale, _Float16* hi, _Float16* lo, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i] * scale;
  const _Float16 h = (_Float16)v;
  hi[i] = h;
  lo[i] = (_Float16)(v - (float)h);
}

template <int KIND>   // 0: 32x32x16 f16, 1: 16x16x32 f16, 2: 32x32x16 bf16, 3: 16x16x32 bf16
__global__ __launch_bounds__(256) void mfma_loop(const f16x8* __restrict__ ops, float* __restrict__ sink,
                                                 unsigned long long* __restrict__ stamps, int iters) {
  const int lane = threadIdx.x & 63;
  f16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = ops[(i * 64 + lane)];
    b[i] = ops[((4 + i) * 64 + lane)];
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float total = 0.f;
  if (KIND == 0 || KIND == 2) {
    f32x16 c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (KIND == 0) {
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c[0]) : "v"(a[u]), "v"(b[u]));
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c[1]) : "v"(a[u]), "v"(b[(u + 1) & 3]));
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c[2]) : "v"(a[(u + 1) & 3]), "v"(b[u]));
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c[3]) : "v"(a[(u + 1) & 3]), "v"(b[(u + 1) & 3]));
        } else {
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c[0]) : "v"(a[u]), "v"(b[u]));
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c[1]) : "v"(a[u]), "v"(b[(u + 1) & 3]));
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c[2]) : "v"(a[(u + 1) & 3]), "v"(b[u]));
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c[3]) : "v"(a[(u + 1) & 3]), "v"(b[(u + 1) & 3]));
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) total += c[i][r];
  } else {
    f32x4 c[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) c[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (KIND == 1)
              asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c[i * 4 + j]) : "v"(a[(i + u) & 3]), "v"(b[(j + u) & 3]));
            else
              asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c[i * 4 + j]) : "v"(a[(i + u) & 3]), "v"(b[(j + u) & 3]));
          }
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) total += c[i][r];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (lane == 0) {
    stamps[2 * wave] = t1 - t0;
    stamps[2 * wave + 1] = r1 - r0;
  }
  sink[blockIdx.x * blockDim.x + threadIdx.x] = total;
}

static uint16_t bf16_of(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
}
static uint16_t f16_bits(_Float16 h) {
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}

int main(int argc, char** argv) {
  const double warm_s = argc > 1 ? atof(argv[1]) : 2.0;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("{\"device\": \"%s\", \"cus\": %d}\n", prop.name, prop.multiProcessorCount);
  // ---- (1) subnormal inputs.  Cases: A subnormal x B normal; A normal x B subnormal; both subnormal; mixed random.
  {
    _Float16 *dA, *dB;
    float *dD, *dD16;
    CK(hipMalloc(&dA, 32 * 16 * 2));
    CK(hipMalloc(&dB, 16 * 32 * 2));
    CK(hipMalloc(&dD, 32 * 32 * 4));
    CK(hipMalloc(&dD16, 16 * 16 * 4));
    uint32_t s = 777u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.f - 0.5f; };
    const char* names[4] = {"A subnormal, B normal", "A normal, B subnormal", "both subnormal", "lo-plane-like mix"};
    for (int cs = 0; cs < 4; ++cs) {
      std::vector<_Float16> A(32 * 16), B(16 * 32);
      for (auto& v : A) {
        const float r = rnd();
        v = (_Float16)((cs == 0 || cs == 2) ? r * 6e-5f : (cs == 3 ? r * ((s & 256) ? 3e-6f : 4.f) : r * 8.f));
      }
      for (auto& v : B) {
        const float r = rnd();
        v = (_Float16)((cs == 1 || cs == 2) ? r * 6e-5f : (cs == 3 ? r * ((s & 512) ? 2e-7f : 300.f) : r * 8.f));
      }
      int nsub = 0;
      for (auto& v : A) nsub += (f16_bits(v) & 0x7c00) == 0 && (f16_bits(v) & 0x3ff) != 0;
      for (auto& v : B) nsub += (f16_bits(v) & 0x7c00) == 0 && (f16_bits(v) & 0x3ff) != 0;
      CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
      CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dD, dD16);
      std::vector<float> D(32 * 32), D16(16 * 16);
      CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(D16.data(), dD16, D16.size() * 4, hipMemcpyDeviceToHost));
      double worst = 0, worst16 = 0, scale = 0;
      int nz = 0;
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          double ref = 0, mag = 0;
          for (int k = 0; k < 16; ++k) {
            const double p = (double)(float)A[i * 16 + k] * (double)(float)B[k * 32 + j];
            ref += p;
            mag += fabs(p);
          }
          scale = std::max(scale, mag);
          nz += D[i * 32 + j] != 0.f;
          if (mag > 0) worst = std::max(worst, fabs(D[i * 32 + j] - ref) / mag);
          if (i < 16 && j < 16 && mag > 0) worst16 = std::max(worst16, fabs(D16[i * 16 + j] - ref) / mag);
        }
      printf("{\"test\": \"subnormal\", \"case\": \"%s\", \"subnormal_inputs\": %d, \"nonzero_outputs\": %d, "
             "\"worst_err_over_sum_abs_products_32x32x16\": %.3e, \"worst_16x16x32\": %.3e, \"largest_sum_abs\": %.3e}\n",
             names[cs], nsub, nz, worst, worst16, scale);
    }
    // the split itself on the device vs the host (v_cvt_f16_f32 rounding, subnormal lo planes)
    const int n = 4096;
    std::vector<float> x(n);
    for (int i = 0; i < n; ++i) x[i] = rnd() * expf(rnd() * 40.f);
    float* dx;
    _Float16 *dh, *dl;
    CK(hipMalloc(&dx, n * 4));
    CK(hipMalloc(&dh, n * 2));
    CK(hipMalloc(&dl, n * 2));
    CK(hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice));
    float amax = 0;
    for (float v : x) amax = std::max(amax, fabsf(v));
    int e;
    frexpf(amax, &e);                    // amax in [2^(e-1), 2^e)
    const float sc = ldexpf(1.f, 15 - e);  // scaled max in [2^14, 2^15)
    hipLaunchKernelGGL(split_kernel, dim3(n / 256), dim3(256), 0, 0, dx, sc, dh, dl, n);
    std::vector<_Float16> hh(n), hl(n);
    CK(hipMemcpy(hh.data(), dh, n * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hl.data(), dl, n * 2, hipMemcpyDeviceToHost));
    int mism = 0, sub = 0;
    double worst_rel = 0, worst_abs_over_max = 0;
    for (int i = 0; i < n; ++i) {
      const float v = x[i] * sc;
      const _Float16 h = (_Float16)v, l = (_Float16)(v - (float)h);
      mism += f16_bits(h) != f16_bits(hh[i]) || f16_bits(l) != f16_bits(hl[i]);
      sub += (f16_bits(hl[i]) & 0x7c00) == 0 && (f16_bits(hl[i]) & 0x3ff) != 0;
      const double rec = (double)(float)hh[i] + (double)(float)hl[i];
      if (v != 0) worst_rel = std::max(worst_rel, fabs(rec - v) / fabs(v));
      worst_abs_over_max = std::max(worst_abs_over_max, fabs(rec - v) / (amax * sc));
    }
    printf("{\"test\": \"split\", \"n\": %d, \"device_vs_host_mismatches\": %d, \"subnormal_lo\": %d, "
           "\"worst_rel_residual\": %.3e, \"worst_residual_over_max\": %.3e}\n", n, mism, sub, worst_rel, worst_abs_over_max);
  }
  // ---- (2) rates
  const int cus = prop.multiProcessorCount, iters = 20000;
  std::vector<uint16_t> h(8 * 64 * 8);
  f16x8* ops;
  float* sink;
  unsigned long long* stamps;
  const int max_blocks = cus * 2;
  CK(hipMalloc(&ops, h.size() * 2));
  CK(hipMalloc(&sink, (size_t)max_blocks * 256 * 4));
  CK(hipMalloc(&stamps, (size_t)max_blocks * 4 * 2 * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int kind = 0; kind < 4; ++kind) {
    for (int opk = 0; opk < 3; ++opk) {      // 0 random N(0,1) hi-like, 1 zeros, 2 (f16 only) a-hi x b-lo like: b small with 11 random mantissa bits
      if (opk == 2 && kind >= 2) continue;
      uint32_t s = 12345u;
      for (size_t i = 0; i < h.size(); ++i) {
        float v = 0.f;
        for (int k = 0; k < 4; ++k) {
          s = s * 1664525u + 1013904223u;
          v += (float)(s >> 8) / 16777216.f - 0.5f;
        }
        v *= 1.7320508f;
        if (opk == 2 && i >= h.size() / 2) v *= 2.4e-4f;
        _Float16 hv = (_Float16)v;
        h[i] = opk == 1 ? 0 : (kind < 2 ? f16_bits(hv) : bf16_of(v));
      }
      CK(hipMemcpy(ops, h.data(), h.size() * 2, hipMemcpyHostToDevice));
      for (int wps = 1; wps <= 2; ++wps) {
        const int blocks = cus * wps;
        auto launch = [&]() {
          switch (kind) {
            case 0: hipLaunchKernelGGL(mfma_loop<0>, dim3(blocks), dim3(256), 0, 0, ops, sink, stamps, iters); break;
            case 1: hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(256), 0, 0, ops, sink, stamps, iters); break;
            case 2: hipLaunchKernelGGL(mfma_loop<2>, dim3(blocks), dim3(256), 0, 0, ops, sink, stamps, iters); break;
            default: hipLaunchKernelGGL(mfma_loop<3>, dim3(blocks), dim3(256), 0, 0, ops, sink, stamps, iters); break;
          }
        };
        launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float one_ms;
        CK(hipEventElapsedTime(&one_ms, e0, e1));
        const int warm = std::max(1, (int)(warm_s * 1e3 / one_ms));
        for (int i = 0; i < warm; ++i) launch();
        const int reps = 20;
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        const int waves = blocks * 4;
        std::vector<unsigned long long> st(2 * waves);
        CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> ghz(waves);
        for (int w = 0; w < waves; ++w) ghz[w] = (double)st[2 * w] / (double)st[2 * w + 1] * 0.1;
        std::sort(ghz.begin(), ghz.end());
        const double flop = (double)waves * iters * 16.0 * 32768.0;
        const char* kn[4] = {"32x32x16_f16", "16x16x32_f16", "32x32x16_bf16", "16x16x32_bf16"};
        const char* on[3] = {"random", "zeros", "hi x lo"};
        printf("{\"test\": \"rate\", \"mfma\": \"%s\", \"operands\": \"%s\", \"wavefronts_per_simd\": %d, \"ms_per_launch\": %.4f, "
               "\"tflops\": %.1f, \"in_kernel_clock_ghz_median\": %.3f}\n", kn[kind], on[opk], wps, ms,
               flop / (ms * 1e-3) * 1e-12, ghz[waves / 2]);
        fflush(stdout);
      }
    }
  }
  return 0;
}
