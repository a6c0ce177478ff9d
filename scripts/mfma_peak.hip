// On-box bf16 MFMA microbenchmark ("peak stated per dtype and confirmed by an on-box MFMA
// microbenchmark").  A bare loop of independent MFMAs, operands in registers, nothing else in the loop: what the matrix
// pipe of THIS chip delivers on random and on all-zero operands, for both bf16 shapes, with the clock it holds meanwhile
// (in-kernel: s_memtime / s_memrealtime, MI355X_MICROARCH.md "DVFS give-back" item 6).  Standalone; built and run on the
// GPU box by scripts/mfma_peak.sh:  mfma_peak <seconds of warm-up launches per case>  ->  one JSON line per case.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

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

// SHAPE 0: v_mfma_f32_32x32x16_bf16, 4 independent accumulators (64 registers); SHAPE 1: v_mfma_f32_16x16x32_bf16,
// 16 independent accumulators (64 registers): the same output tile per wavefront (64 x 64), the same FLOPs per pass.
template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_loop(const bf16x8* __restrict__ ops, float* __restrict__ sink,
                                                 unsigned long long* __restrict__ stamps, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = ops[(i * 64 + lane)];
    b[i] = ops[((4 + i) * 64 + lane)];
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float total = 0.f;
  if (SHAPE == 0) {
    f32x16 c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {          // 16 MFMAs of 32 x 32 x 16: a 64 x 64 tile over K = 64
        // inline asm with the accumulators pinned ("+a"): through the builtin hipcc rotated the sixteen 16 x 16
        // accumulators of the other shape through v_accvgpr_mov copies on every pass (27 cycles per MFMA instead of 16)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c[0]) : "v"(a[u]), "v"(b[u]));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c[1]) : "v"(a[u]), "v"(b[(u + 1) & 3]));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c[2]) : "v"(a[(u + 1) & 3]), "v"(b[u]));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c[3]) : "v"(a[(u + 1) & 3]), "v"(b[(u + 1) & 3]));
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
      for (int u = 0; u < 2; ++u) {          // 32 MFMAs of 16 x 16 x 32: the same 64 x 64 tile over K = 64
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c[i * 4 + j]) : "v"(a[(i + u) & 3]), "v"(b[(j + u) & 3]));
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
  sink[blockIdx.x * blockDim.x + threadIdx.x] = total;      // keeps the loop alive; own buffer, read by nobody
}

static uint16_t bf16_of(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
}

int main(int argc, char** argv) {
  const double warm_s = argc > 1 ? atof(argv[1]) : 2.0;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int iters = 20000;                                   // 16 x 20000 MFMAs of 32 cycles: ~5 ms per launch at 2 GHz
  std::vector<uint16_t> h(8 * 64 * 8);
  bf16x8* ops;
  float* sink;
  unsigned long long* stamps;
  const int max_blocks = cus * 2;
  CK(hipMalloc(&ops, h.size() * 2));
  CK(hipMalloc(&sink, (size_t)max_blocks * 256 * 4));
  CK(hipMalloc(&stamps, (size_t)max_blocks * 4 * 2 * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("{\"device\": \"%s\", \"cus\": %d, \"max_clock_mhz\": %d}\n", prop.name, cus, prop.clockRate / 1000);
  for (int zero = 0; zero < 2; ++zero) {
    // N(0, 1)-like operands from a fixed linear congruential stream (sum of 4 uniforms), or all zeros
    uint32_t s = 12345u;
    for (size_t i = 0; i < h.size(); ++i) {
      float v = 0.f;
      for (int k = 0; k < 4; ++k) {
        s = s * 1664525u + 1013904223u;
        v += (float)(s >> 8) / 16777216.f - 0.5f;
      }
      h[i] = zero ? 0 : bf16_of(v * 1.7320508f);
    }
    CK(hipMemcpy(ops, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    for (int wps = 1; wps <= 2; ++wps) {                     // wavefronts per SIMD: workgroups of 4 wavefronts, 1 or 2 per CU
      for (int shape = 0; shape < 2; ++shape) {
        const int blocks = cus * wps;
        auto launch = [&]() {
          if (shape == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(blocks), dim3(256), 0, 0, ops, sink, stamps, iters);
          else hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(256), 0, 0, ops, sink, stamps, iters);
        };
        // back-to-back launches for warm_s seconds: the clock the chip settles at, not the one it starts from
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
        std::vector<double> ghz(waves), cyc(waves);
        for (int w = 0; w < waves; ++w) {
          ghz[w] = (double)st[2 * w] / (double)st[2 * w + 1] * 0.1;      // s_memrealtime counts at 100 MHz
          cyc[w] = (double)st[2 * w];
        }
        std::sort(ghz.begin(), ghz.end());
        std::sort(cyc.begin(), cyc.end());
        const double flop = (double)waves * iters * 16.0 * 32768.0;      // both shapes: 64 x 64 x 64 x 2 per iteration
        printf("{\"operands\": \"%s\", \"shape\": \"%s\", \"wavefronts_per_simd\": %d, \"ms_per_launch\": %.4f, "
               "\"tflops\": %.1f, \"in_kernel_clock_ghz_median\": %.3f, \"cycles_per_flop_x1e6\": %.3f}\n",
               zero ? "zeros" : "random", shape == 0 ? "32x32x16" : "16x16x32", wps, ms, flop / (ms * 1e-3) * 1e-12,
               ghz[waves / 2], cyc[waves / 2] / (iters * 16.0 * 32768.0) * 1e6);
        fflush(stdout);
      }
    }
  }
  return 0;
}
