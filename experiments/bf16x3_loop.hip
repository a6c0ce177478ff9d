// Ablation for DESIGN.md section 8 item 1 (split-bf16 convolutions): what can the inner loop of a
// hi/lo-split implicit GEMM deliver on gfx950 when NOTHING is staged?  One workgroup = 4 or 8
// wavefronts, each owning a 2 x 2 tile of 32x32 fragments; per k16 step a wavefront issues
// 8 x ds_read_b128 (A hi/lo x 2 fragments from a [k-step][cout][8] filter image, B hi/lo x 2
// fragments from a channel-innermost [pixel][8] patch image at a per-tap compile-time offset) and
// 12 x v_mfma_f32_32x32x16_bf16 (hi*hi + hi*lo + lo*hi per fragment pair).  LDS holds garbage;
// only the instruction mix, the addressing pattern and the barriers are those of the real kernel.
// Prints fp32-equivalent TFLOP/s (each logical MAC counted once).
//   hipcc -O3 --offload-arch=gfx950 experiments/bf16x3_loop.hip -o experiments/_bf16x3_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NWAVE, int NPROD>
__global__ __launch_bounds__(64 * NWAVE, 2) void loop_kernel(float* out, int chunks) {
  // patch image: 2 planes x 2 k-blocks x (11 rows x 2 parities x 18 cols) pixels x 16 B = 25 KB
  // filter image per tap: 2 planes x 2 k-blocks x 128 cout x 16 B = 8 KB; 5 taps resident = 40 KB
  constexpr int PPIX = 11 * 2 * 18, PIMG = 2 * 2 * PPIX;   // 16-byte units
  constexpr int WTAP = 2 * 2 * 128;
  __shared__ f32x4 lds[PIMG + 5 * WTAP];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int kb = lane >> 5, l32 = lane & 31;
  const int wc = wid & 1, wp = (wid >> 1) % (NWAVE / 2);
  for (int e = tid; e < PIMG + 5 * WTAP; e += 64 * NWAVE) lds[e] = f32x4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  // per-lane bases (16-byte units).  B: fragment f of this wave = 2 output rows x 16 pixels
  int base_b[2], base_a[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int m = (wp * 2 + f) * 32 + l32;
    const int ph = (m / 16) % 4, pw = m % 16;
    base_b[f] = kb * PPIX + (2 * ph * 2) * 18 + pw;       // [kb][row][parity][col]
  }
#pragma unroll
  for (int g = 0; g < 2; ++g) base_a[g] = PIMG + kb * 128 + (wc * 2 + g) * 32 + l32;
  f32x16 acc[2][2];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[g][f][r] = 0.f;
  for (int ch = 0; ch < chunks; ++ch) {
#pragma unroll
    for (int kh = 0; kh < 5; ++kh) {
#pragma unroll
      for (int kw = 0; kw < 5; ++kw) {
        const int boff = (kh * 2 + (kw & 1)) * 18 + (kw >> 1);    // tap offset in the de-interleaved patch
        bf16x8 a_hi[2], a_lo[2], b_hi[2], b_lo[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          a_hi[g] = __builtin_bit_cast(bf16x8, lds[base_a[g] + kw * WTAP]);
          a_lo[g] = __builtin_bit_cast(bf16x8, lds[base_a[g] + kw * WTAP + 2 * 128]);
        }
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          b_hi[f] = __builtin_bit_cast(bf16x8, lds[base_b[f] + boff]);
          b_lo[f] = __builtin_bit_cast(bf16x8, lds[base_b[f] + boff + 2 * PPIX]);
        }
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            acc[g][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[g], b_hi[f], acc[g][f], 0, 0, 0);
            if (NPROD >= 3) {
              acc[g][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[g], b_lo[f], acc[g][f], 0, 0, 0);
              acc[g][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo[g], b_hi[f], acc[g][f], 0, 0, 0);
            }
          }
      }
      __syncthreads();     // the real kernel swaps the 5-tap filter slab here
    }
  }
  float s = 0.f;
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[g][f][r];
  out[(size_t)blockIdx.x * 64 * NWAVE + tid] = s;
}

template <int NWAVE, int NPROD>
void run(const char* name, int blocks, int chunks) {
  float* out;
  hipMalloc(&out, (size_t)blocks * 64 * NWAVE * sizeof(float));
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((loop_kernel<NWAVE, NPROD>), dim3(blocks), dim3(64 * NWAVE), 0, 0, out, chunks);
    hipEventRecord(b);
    hipEventSynchronize(b);
  }
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  // logical MACs: per wave per tap step 4 fragments x 32 x 32 x 16
  const double flop = 2.0 * blocks * NWAVE * (double)chunks * 25 * 4 * 32 * 32 * 16;
  printf("%-34s %4d workgroups x %d waves: %7.3f ms  %7.1f fp32-equivalent TFLOP/s  (%d MFMAs per step)\n", name,
         blocks, NWAVE, ms, flop / ms / 1e9, 4 * NPROD);
  hipFree(out);
}

int main() {
  run<4, 3>("hi/lo split, 3 products, 4 waves", 2048, 64);
  run<8, 3>("hi/lo split, 3 products, 8 waves", 1024, 64);
  run<4, 1>("plain bf16, 1 product, 4 waves", 2048, 64);
  run<8, 1>("plain bf16, 1 product, 8 waves", 1024, 64);
  return 0;
}
