// Linear layers in the split-bf16 arithmetic (vg_gemm_nt_bf16split) for gfx950: the three GEMMs of nn.Linear's
// forward / backward on the 16384 x 2048 layers of the path (encoder heads x_to_mu.0 / x_to_logvar.0,
// /root/reference/models/model.py:460-471; discriminator lth_features.0, :402-404), where the fp32 vendor GEMM is
// MFMA-bound at ~100 TFLOP/s and the weight (134 MB) could stream in a third of that time:
//
//     C[m][n] = sum_k A(m, k) * B(n, k) (+ bias[n]),      A(m, k) = A[m * ars + k * aks],  B(n, k) = B[n * brs + k * bks]
//
// with, per operand, either the reduction index contiguous (k stride 1) or the row index contiguous (row stride 1):
//   forward   y  = x W^T   : A = x  (ars = K,  aks = 1),   B = W (brs = K, bks = 1)
//   data grad gx = gy W    : A = gy (ars = N', aks = 1),   B = W read as [k_in][n] (brs = 1, bks = K_in)
//   weight gr gW = gy^T x  : A = gy read as [n][b] (ars = 1, aks = N'),   B = x read as [k][b] (brs = 1, bks = K_in)
// Same arithmetic as the convolutions (conv_ring.hip): every fp32 operand split into NP bf16 planes (NP = 3: exact,
// fp32-equivalent), the plane products with index sum < NP on v_mfma_f32_32x32x16_bf16, fp32 accumulate.
//
// One workgroup = 8 wavefronts owns a 128 x 128 output tile (2 x 4 wavefronts of 64 x 32) and a slice of the reduction
// (K split over blockIdx.z; partial tiles go to slabs summed in a fixed order).  A stage = 32 reduction indices: every
// thread stages exactly ONE 8-index unit of A and one of B -- two 16-byte loads where the reduction index is
// contiguous, eight 4-byte loads (a row apart; consecutive lanes = consecutive rows, coalesced) where it is not --
// splits it into planes in registers and writes NP 16-byte LDS units; stages are double-buffered in LDS with the
// next stage's loads in flight during the MFMAs, ONE barrier per stage.  Plain loads only: hipcc counts vmcnt itself.
#include "common.hpp"
#include "vaegan_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

constexpr int GNT = 512, GT = 128, GKC = 32;          // threads, tile edge, reduction indices per stage
constexpr int G_OPU = 4 * GT;                          // units of one operand plane per stage: [k-block 4][row 128]

struct GArgs {
  const float* A;
  const float* B;
  const float* bias;
  float* C;            // ksplit == 1: the output [M][N]; else the slabs [ksplit][M][N]
  int M, N, K;
  long ars, aks, brs, bks;
  int kper;            // reduction indices per split (multiple of GKC)
  int ksplit;
};

template <int NP>
__device__ __forceinline__ void split8(float* v, bf16x8* out) {
#pragma unroll
  for (int p = 0; p < NP; ++p) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const __bf16 h = (__bf16)v[j];
      out[p][j] = h;
      v[j] -= (float)h;
    }
  }
}

// AT / BT: the operand's ROW index is the contiguous one (reduction index strided)
template <int NP, bool AT, bool BT>
__global__ __launch_bounds__(GNT, 2) void gemm_nt_split_kernel(GArgs G) {
  __shared__ f32x4 lds[2 * 2 * NP * G_OPU];           // [buffer][operand][plane][k-block][row]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kb = lane >> 5, l32 = lane & 31;
  const int wm = wid & 1, wn = wid >> 1;                // 2 x 4 wavefronts: 64 rows x 32 columns each
  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT, split = blockIdx.z;
  const int k_begin = split * G.kper, k_end = min(k_begin + G.kper, G.K);

  // ---- staging map: one unit (row, k-block) per thread and operand
  //   reduction contiguous: 4 consecutive lanes cover the 32 indices (128 B) of a row;
  //   row contiguous:       consecutive lanes = consecutive rows, the unit's 8 indices are a k-stride apart
  const int a_row = AT ? (tid & (GT - 1)) : (tid >> 2), a_kblk = AT ? (tid >> 7) : (tid & 3);
  const int b_row = BT ? (tid & (GT - 1)) : (tid >> 2), b_kblk = BT ? (tid >> 7) : (tid & 3);
  const bool a_ok = (m0 + a_row) < G.M, b_ok = (n0 + b_row) < G.N;
  const float* ap = G.A + (size_t)min(m0 + a_row, G.M - 1) * G.ars + (size_t)(k_begin + a_kblk * 8) * G.aks;
  const float* bp = G.B + (size_t)min(n0 + b_row, G.N - 1) * G.brs + (size_t)(k_begin + b_kblk * 8) * G.bks;
  const int a_dst = a_kblk * GT + a_row, b_dst = NP * G_OPU + b_kblk * GT + b_row;

  // Register ring: the loads of stage st + PD are issued at the start of stage st (a stage is ~0.8 us of MFMAs, a
  // first touch of the streamed operand ~2 us: one stage of run-ahead measured 85 us on 128 x 2048 x 16384, latency-
  // bound); the ring slot of a stage is (stage % PD), compile-time in the PD-unrolled loop.
  constexpr int PD = 4;
  float ra[PD][8], rb[PD][8];
  auto load_stage = [&](int slot, int st) {            // st: stage index within this split
    const float* a = ap + (size_t)st * GKC * G.aks;
    const float* b = bp + (size_t)st * GKC * G.bks;
    if constexpr (AT) {
#pragma unroll
      for (int j = 0; j < 8; ++j) ra[slot][j] = a[(size_t)j * G.aks];
    } else {
      const f32x4 lo = *reinterpret_cast<const f32x4*>(a), hi = *reinterpret_cast<const f32x4*>(a + 4);
      ra[slot][0] = lo[0]; ra[slot][1] = lo[1]; ra[slot][2] = lo[2]; ra[slot][3] = lo[3];
      ra[slot][4] = hi[0]; ra[slot][5] = hi[1]; ra[slot][6] = hi[2]; ra[slot][7] = hi[3];
    }
    if constexpr (BT) {
#pragma unroll
      for (int j = 0; j < 8; ++j) rb[slot][j] = b[(size_t)j * G.bks];
    } else {
      const f32x4 lo = *reinterpret_cast<const f32x4*>(b), hi = *reinterpret_cast<const f32x4*>(b + 4);
      rb[slot][0] = lo[0]; rb[slot][1] = lo[1]; rb[slot][2] = lo[2]; rb[slot][3] = lo[3];
      rb[slot][4] = hi[0]; rb[slot][5] = hi[1]; rb[slot][6] = hi[2]; rb[slot][7] = hi[3];
    }
  };
  auto store_stage = [&](int slot, int buf) {
    float va[8], vb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      va[j] = a_ok ? ra[slot][j] : 0.f;                 // rows beyond M / N: zeros (their outputs are not stored)
      vb[j] = b_ok ? rb[slot][j] : 0.f;
    }
    bf16x8 pa[NP], pb[NP];
    split8<NP>(va, pa);
    split8<NP>(vb, pb);
    f32x4* base = lds + buf * 2 * NP * G_OPU;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      base[p * G_OPU + a_dst] = __builtin_bit_cast(f32x4, pa[p]);
      base[p * G_OPU + b_dst] = __builtin_bit_cast(f32x4, pb[p]);
    }
  };

  f32x16 acc[2];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

  const int nst = (k_end - k_begin) / GKC;
  if (nst > 0) {
    // stages past the end re-load the last one (never stored): every ring slot always holds valid registers
#pragma unroll
    for (int j = 0; j < PD; ++j) load_stage(j, min(j, nst - 1));
    store_stage(0, 0);
    __syncthreads();
    for (int st0 = 0; st0 < nst; st0 += PD) {
#pragma unroll
      for (int j = 0; j < PD; ++j) {
        const int st = st0 + j;
        if (st < nst) {                                  // wave-uniform
          const int buf = st & 1;
          const bool more = (st + 1) < nst;
          load_stage(j, min(st + PD, nst - 1));          // slot j was stored to LDS at the end of stage st - 1
          const f32x4* base = lds + buf * 2 * NP * G_OPU;
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {               // two 16-index MFMA steps per stage
            bf16x8 av[2][NP], bv[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
#pragma unroll
              for (int g = 0; g < 2; ++g)
                av[g][p] = __builtin_bit_cast(bf16x8, base[p * G_OPU + (2 * s2 + kb) * GT + wm * 64 + g * 32 + l32]);
              bv[p] = __builtin_bit_cast(bf16x8, base[(NP + p) * G_OPU + (2 * s2 + kb) * GT + wn * 32 + l32]);
            }
#pragma unroll
            for (int sum = NP - 1; sum >= 0; --sum)
#pragma unroll
              for (int pa = sum; pa >= 0; --pa)
#pragma unroll
                for (int g = 0; g < 2; ++g)
                  acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[g][pa], bv[sum - pa], acc[g], 0, 0, 0);
          }
          if (more) store_stage((j + 1) % PD, buf ^ 1);  // stage st + 1: loaded PD - 1 stages ago
          __syncthreads();
        }
      }
    }
  }

  // ---- epilogue: C (or this split's slab) row-major [M][N]; the bias goes in with split 0
  float* out = G.C + (G.ksplit > 1 ? (size_t)split * G.M * G.N : 0);
  const int n = n0 + wn * 32 + l32;
  const float bvv = (G.bias && split == 0 && n < G.N) ? G.bias[n] : 0.f;
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int r16 = 0; r16 < 16; ++r16) {
      const int m = m0 + wm * 64 + g * 32 + acc_row(r16, lane);
      if (m < G.M && n < G.N) out[(size_t)m * G.N + n] = acc[g][r16] + bvv;
    }
}

// K split: as many splits as keep >= 8 stages each and bring the grid to about one workgroup per CU
int gemm_ksplit(int M, int N, int K) {
  const long tiles = (long)cdiv(M, GT) * cdiv(N, GT);
  int ks = 1;
  while (tiles * ks * 2 <= 256 && K % (GKC * ks * 2) == 0 && K / (ks * 2) >= 8 * GKC) ks *= 2;
  return ks;
}

bool gemm_ok(int M, int N, int K, long ars, long aks, long brs, long bks) {
  if (M <= 0 || N <= 0 || K <= 0 || K % GKC) return false;
  if (!((aks == 1 && ars % 4 == 0) || ars == 1)) return false;      // 16-byte loads need aligned rows
  if (!((bks == 1 && brs % 4 == 0) || brs == 1)) return false;
  return true;
}

template <int NP>
int launch_gemm(const GArgs& G, bool at, bool bt, dim3 grid, hipStream_t st) {
  if (at && bt) hipLaunchKernelGGL((gemm_nt_split_kernel<NP, true, true>), grid, dim3(GNT), 0, st, G);
  else if (at) hipLaunchKernelGGL((gemm_nt_split_kernel<NP, true, false>), grid, dim3(GNT), 0, st, G);
  else if (bt) hipLaunchKernelGGL((gemm_nt_split_kernel<NP, false, true>), grid, dim3(GNT), 0, st, G);
  else hipLaunchKernelGGL((gemm_nt_split_kernel<NP, false, false>), grid, dim3(GNT), 0, st, G);
  VG_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" size_t vg_gemm_nt_bf16split_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0 || K % GKC) return 0;
  const int ks = gemm_ksplit(M, N, K);
  return ks > 1 ? (size_t)ks * M * N * sizeof(float) : 0;
}

extern "C" int vg_gemm_nt_bf16split(const float* A, const float* B, const float* bias, float* C, int M, int N, int K,
                                    long a_row_stride, long a_k_stride, long b_row_stride, long b_k_stride, int planes,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  if (!A || !B || !C || (planes != 2 && planes != 3)) return VG_ERR_BAD_ARG;
  if (!gemm_ok(M, N, K, a_row_stride, a_k_stride, b_row_stride, b_k_stride)) return VG_ERR_BAD_ARG;
  if ((!(a_k_stride != 1) && ((uintptr_t)A & 15)) || (!(b_k_stride != 1) && ((uintptr_t)B & 15))) return VG_ERR_BAD_ARG;
  const int ks = gemm_ksplit(M, N, K);
  if (ks > 1 && (!workspace || workspace_bytes < (size_t)ks * M * N * sizeof(float))) return VG_ERR_WORKSPACE;
  if ((size_t)M * N > 0x7fffffffUL) return VG_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  GArgs G;
  G.A = A; G.B = B; G.bias = bias; G.C = ks > 1 ? (float*)workspace : C;
  G.M = M; G.N = N; G.K = K;
  G.ars = a_row_stride; G.aks = a_k_stride; G.brs = b_row_stride; G.bks = b_k_stride;
  G.kper = K / ks; G.ksplit = ks;
  const dim3 grid(cdiv(N, GT), cdiv(M, GT), ks);
  const bool at = a_k_stride != 1, bt = b_k_stride != 1;
  const int rc = planes == 3 ? launch_gemm<3>(G, at, bt, grid, st) : launch_gemm<2>(G, at, bt, grid, st);
  if (rc) return rc;
  if (ks > 1) return vg_internal_wgrad_reduce((const float*)workspace, C, M * N, ks, st);   // fixed-order sum of the slabs
  return 0;
}
