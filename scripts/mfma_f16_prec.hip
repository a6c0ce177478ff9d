// What the f16 / bf16 MFMAs do INSIDE one instruction's K sum (needed for the split arithmetics: a plane-product
// instruction mixes magnitudes as far as the data does).  One 32x32x16 (or 16x16x32) instruction per experiment:
//  (1) one large product 2^e next to 15 unit products: is the sum 2^e + 15 exact?  (alignment width of the adder tree)
//  (2) the same with products carrying 21 significant bits
//  (3) subnormal f16 inputs next to large ones
//  (4) accumulator input C large / small relative to the products
// Standalone; scripts/mfma_f16_prec.sh builds and runs it on the GPU box.
#include <hip/hip_runtime.h>
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
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// A[32][16], B[16][32] given as floats (converted in-kernel to f16 or bf16), C[32][32] in, D out
template <int BF>
__global__ void one32(const float* A, const float* B, const float* C, float* D) {
  const int lane = threadIdx.x, i = lane & 31, kb = lane >> 5;
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = C[((r & 3) + 8 * (r >> 2) + 4 * kb) * 32 + i];
  if (BF) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)A[i * 16 + kb * 8 + e]; b[e] = (__bf16)B[(kb * 8 + e) * 32 + i]; }
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  } else {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)A[i * 16 + kb * 8 + e]; b[e] = (_Float16)B[(kb * 8 + e) * 32 + i]; }
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * kb) * 32 + i] = c[r];
}
// 16x16x32: A[16][32], B[32][16]
template <int BF>
__global__ void one16(const float* A, const float* B, const float* C, float* D) {
  const int lane = threadIdx.x, i = lane & 15, k4 = lane >> 4;
  f32x4 c;
  for (int r = 0; r < 4; ++r) c[r] = C[(4 * k4 + r) * 16 + i];
  if (BF) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)A[i * 32 + k4 * 8 + e]; b[e] = (__bf16)B[(k4 * 8 + e) * 16 + i]; }
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  } else {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)A[i * 32 + k4 * 8 + e]; b[e] = (_Float16)B[(k4 * 8 + e) * 16 + i]; }
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) D[(4 * k4 + r) * 16 + i] = c[r];
}

struct Dev { float *A, *B, *C, *D; };
static void run(int shape, int bf, Dev d, const std::vector<float>& A, const std::vector<float>& B, const std::vector<float>& C, std::vector<float>& D) {
  CK(hipMemcpy(d.A, A.data(), A.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d.B, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d.C, C.data(), C.size() * 4, hipMemcpyHostToDevice));
  if (shape == 0) { if (bf) hipLaunchKernelGGL(one32<1>, dim3(1), dim3(64), 0, 0, d.A, d.B, d.C, d.D); else hipLaunchKernelGGL(one32<0>, dim3(1), dim3(64), 0, 0, d.A, d.B, d.C, d.D); }
  else { if (bf) hipLaunchKernelGGL(one16<1>, dim3(1), dim3(64), 0, 0, d.A, d.B, d.C, d.D); else hipLaunchKernelGGL(one16<0>, dim3(1), dim3(64), 0, 0, d.A, d.B, d.C, d.D); }
  D.resize(shape == 0 ? 1024 : 256);
  CK(hipMemcpy(D.data(), d.D, D.size() * 4, hipMemcpyDeviceToHost));
}

int main() {
  Dev d;
  CK(hipMalloc(&d.A, 4096)); CK(hipMalloc(&d.B, 4096)); CK(hipMalloc(&d.C, 4096)); CK(hipMalloc(&d.D, 4096));
  const char* sn[2] = {"32x32x16", "16x16x32"};
  const char* tn[2] = {"f16", "bf16"};
  for (int shape = 0; shape < 2; ++shape)
    for (int bf = 0; bf < 2; ++bf) {
      const int M = shape == 0 ? 32 : 16, K = shape == 0 ? 16 : 32;
      // row i of A / column j of B: only element (0,0) of D is examined; position `pos` of the large product varies
      for (int exp_kind = 0; exp_kind < 2; ++exp_kind) {      // 0: unit small products; 1: small products (1 + 2^-7)^2 (bf16-exact) or (1+2^-10)^2
        printf("{\"mfma\": \"%s_%s\", \"small_products\": \"%s\", \"first_inexact_e_by_pos\": [", sn[shape], tn[bf], exp_kind ? "1+eps squared" : "1.0");
        for (int pos = 0; pos < K; pos += (K / 8)) {
          int first_bad = -1;
          for (int e = 0; e <= 28 && first_bad < 0; ++e) {
            std::vector<float> A(M * K, 0.f), B(K * M, 0.f), C(M * M, 0.f), D;
            const float eps = bf ? ldexpf(1.f, -7) : ldexpf(1.f, -10);
            const float sm = exp_kind ? 1.f + eps : 1.f;
            for (int k = 0; k < K; ++k) { A[0 * K + k] = sm; B[k * M + 0] = sm; }
            A[pos] = ldexpf(1.f, e / 2);
            B[pos * M] = ldexpf(1.f, e - e / 2);
            run(shape, bf, d, A, B, C, D);
            const double ref = ldexp(1.0, e) + (K - 1) * (double)sm * (double)sm;
            const float reff = (float)ref;              // correctly rounded fp32 of the exact sum
            if (D[0] != reff) first_bad = e;
          }
          printf("%s%d", pos ? ", " : "", first_bad);
        }
        printf("]}\n");
      }
      // detailed: large product at position 0, e = 12..26: what comes out vs exact
      printf("{\"mfma\": \"%s_%s\", \"detail\": [", sn[shape], tn[bf]);
      for (int e = 10; e <= 28; e += 2) {
        std::vector<float> A(M * K, 0.f), B(K * M, 0.f), C(M * M, 0.f), D;
        for (int k = 0; k < K; ++k) { A[k] = 1.f; B[k * M] = 1.f; }
        A[0] = ldexpf(1.f, e / 2); B[0] = ldexpf(1.f, e - e / 2);
        run(shape, bf, d, A, B, C, D);
        printf("%s[%d, %.1f]", e > 10 ? ", " : "", e, (double)D[0] - ldexp(1.0, e));
      }
      printf("]}\n");
      // accumulator: C = 2^e, 16/32 unit products
      printf("{\"mfma\": \"%s_%s\", \"C_large_detail\": [", sn[shape], tn[bf]);
      for (int e = 10; e <= 28; e += 2) {
        std::vector<float> A(M * K, 0.f), B(K * M, 0.f), C(M * M, 0.f), D;
        for (int k = 0; k < K; ++k) { A[k] = 1.f; B[k * M] = 1.f; }
        C[0] = ldexpf(1.f, e);
        run(shape, bf, d, A, B, C, D);
        printf("%s[%d, %.1f]", e > 10 ? ", " : "", e, (double)D[0] - ldexp(1.0, e));
      }
      printf("]}\n");
      if (!bf) {
        // subnormal a (n ulps of 2^-24) times b = 2^10 next to a large product 2^e: is a*b = n * 2^-14 still added exactly?
        printf("{\"mfma\": \"%s_f16\", \"subnormal_next_to_large\": [", sn[shape]);
        for (int e = -14; e <= 10; e += 4) {
          std::vector<float> A(M * K, 0.f), B(K * M, 0.f), C(M * M, 0.f), D;
          A[0] = ldexpf(1.f, e >= 0 ? e : 0); B[0] = ldexpf(1.f, e >= 0 ? 0 : e);
          for (int k = 1; k < K; ++k) { A[k] = ldexpf((float)(k + 1), -24); B[k * M] = 1024.f; }
          double ref = ldexp(1.0, e);
          for (int k = 1; k < K; ++k) ref += ldexp((double)(k + 1), -14);
          run(shape, bf, d, A, B, C, D);
          printf("%s[%d, %.6e, %.6e]", e > -14 ? ", " : "", e, (double)D[0], ref);
        }
        printf("]}\n");
      }
    }
  return 0;
}
