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
// One workgroup = 8 wavefronts owns a 128 x 256 output tile (2 x 4 wavefronts of 64 x 64: 12 fragment reads per 24
// MFMAs -- with 64 x 32 wavefront tiles the LDS port, not the matrix pipe, set the pace) and a slice of the reduction
// (K split over blockIdx.z; partial tiles go to slabs summed in a fixed order).  A stage = 32 reduction indices: a
// thread stages ONE 8-index unit of A and two of B -- two 16-byte loads each where the reduction index is contiguous,
// eight 4-byte loads (a row apart; consecutive lanes = consecutive rows, coalesced) where it is not -- PD stages ahead
// in registers, splits them into planes and writes NP 16-byte LDS units each ([plane][k-block][row], k-blocks padded by
// two units so that the 4 lanes that share a row do not share banks); stages are double-buffered in LDS, ONE barrier
// per stage.  Plain loads only: hipcc counts vmcnt itself.
#include "common.hpp"
#include "vaegan_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

constexpr int GNT = 512, GTM = 128, GTN = 256, GKC = 32;   // threads, tile rows / columns, reduction indices per stage
constexpr int GPAD = 2;
constexpr int A_KB = GTM + GPAD, B_KB = GTN + GPAD;          // units per k-block
constexpr int A_PL = 4 * A_KB, B_PL = 4 * B_KB;              // units per plane
constexpr int G_PD = 2;                                      // stages in flight in registers
// timing experiments only (experiments/abl_build.sh gemm <bits>): 1 every workgroup streams the B rows of tile 0 /
// split 0 (cache-resident), 2 the same for A, 4 no plane split (raw bits stored)
#ifndef VG_GEMM_ABL
#define VG_GEMM_ABL 0
#endif

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

__device__ __forceinline__ void load8(float* r, const float* p, long ks, bool strided) {
  if (strided) {
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = p[(size_t)j * ks];
  } else {
    const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  }
}

// AT / BT: the operand's ROW index is the contiguous one (reduction index strided)
template <int NP, bool AT, bool BT>
__global__ __launch_bounds__(GNT, 2) void gemm_nt_split_kernel(GArgs G) {
  constexpr int BUFU = NP * (A_PL + B_PL);
  __shared__ f32x4 lds[2 * BUFU];                      // [buffer][A planes | B planes][k-block][row]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kb = lane >> 5, l32 = lane & 31;
  const int wm = wid & 1, wn = wid >> 1;                // 2 x 4 wavefronts: 64 rows x 64 columns each
  const int m0 = blockIdx.y * GTM, n0 = blockIdx.x * GTN, split = blockIdx.z;
  const int k_begin = split * G.kper, k_end = min(k_begin + G.kper, G.K);
  const int nst = (k_end - k_begin) / GKC;

  // ---- staging map: unit (row, k-block) -> thread.  Reduction contiguous: 4 consecutive lanes cover the 32 indices
  // (128 B) of a row; row contiguous: consecutive lanes = consecutive rows, the unit's 8 indices a k-stride apart.
  const int a_row = AT ? (tid & (GTM - 1)) : (tid >> 2), a_kblk = AT ? (tid >> 7) : (tid & 3);
  const bool a_ok = (m0 + a_row) < G.M;
  const float* ap = G.A + (size_t)min(m0 + a_row, G.M - 1) * G.ars + (size_t)(((VG_GEMM_ABL & 2) ? 0 : k_begin) + a_kblk * 8) * G.aks;
  const int a_dst = a_kblk * A_KB + a_row;
  const float* bp[2];
  int b_dst[2];
  bool b_ok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + GNT * i;
    const int b_row = BT ? (e & (GTN - 1)) : (e >> 2), b_kblk = BT ? (e >> 8) : (e & 3);
    b_ok[i] = (n0 + b_row) < G.N;
    bp[i] = G.B + (size_t)min(((VG_GEMM_ABL & 1) ? 0 : n0) + b_row, G.N - 1) * G.brs +
            (size_t)(((VG_GEMM_ABL & 1) ? 0 : k_begin) + b_kblk * 8) * G.bks;
    b_dst[i] = NP * A_PL + b_kblk * B_KB + b_row;
  }

  float ra[G_PD][8], rb[G_PD][2][8];
  auto load_stage = [&](int slot, int st) {            // st: stage index within this split
    load8(ra[slot], ap + (size_t)st * GKC * G.aks, G.aks, AT);
#pragma unroll
    for (int i = 0; i < 2; ++i) load8(rb[slot][i], bp[i] + (size_t)st * GKC * G.bks, G.bks, BT);
  };
  // one plane of one staged unit (u = 0: the A unit, 1 / 2: the B units): v -> bf16 plane p -> LDS, residual stays in v
  auto piece = [&](int slot, int u, int p, f32x4* base) {
    float* v = (u == 0) ? ra[slot] : rb[slot][u - 1];
    if (p == 0) {
      const bool ok = (u == 0) ? a_ok : b_ok[u == 0 ? 0 : u - 1];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = ok ? v[j] : 0.f;           // rows beyond M / N: zeros (never stored)
    }
    bf16x8 o;
    if (VG_GEMM_ABL & 4) {
      o = __builtin_bit_cast(bf16x8, f32x4{v[0], v[1], v[2], v[3]});
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)v[j];
        o[j] = h;
        v[j] -= (float)h;
      }
    }
    base[(u == 0) ? p * A_PL + a_dst : p * B_PL + b_dst[u == 0 ? 0 : u - 1]] = __builtin_bit_cast(f32x4, o);   // b_dst starts at the B planes
  };
  auto store_stage = [&](int slot, int buf) {
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int p = 0; p < NP; ++p) piece(slot, u, p, lds + buf * BUFU);
  };
  auto read_frags = [&](bf16x8 (&av)[2][NP], bf16x8 (&bv)[2][NP], const f32x4* base, int s2) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        av[g][p] = __builtin_bit_cast(bf16x8, base[p * A_PL + (2 * s2 + kb) * A_KB + wm * 64 + g * 32 + l32]);
        bv[g][p] = __builtin_bit_cast(bf16x8, base[NP * A_PL + p * B_PL + (2 * s2 + kb) * B_KB + wn * 64 + g * 32 + l32]);
      }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[g][h][r] = 0.f;

  constexpr int NPROD = NP * (NP + 1) / 2;               // plane products per MFMA step (index sum < NP)
  if (nst > 0) {
    // stages past the end re-load the last one (never stored): every ring slot always holds valid registers
#pragma unroll
    for (int j = 0; j < G_PD; ++j) load_stage(j, min(j, nst - 1));
    store_stage(0, 0);
    __syncthreads();
    for (int st0 = 0; st0 < nst; st0 += G_PD) {
#pragma unroll
      for (int j = 0; j < G_PD; ++j) {
        const int st = st0 + j;
        if (st < nst) {                                  // uniform
          // One stage, issue order pinned group by group (left alone, hipcc emits the 48 MFMAs back to back and then
          // the ~210 VALU operations of the split with the matrix pipe idle): a group = the 4 MFMAs of one plane
          // product, each followed by its share of one piece of the split of stage st + 1 (which then executes
          // while the MFMA does); the second step's fragments are read during the first step's second group.
          const int buf = st & 1;
          const f32x4* base = lds + buf * BUFU;
          f32x4* nxt = lds + (buf ^ 1) * BUFU;           // stage st + 1 (loaded G_PD - 1 stages ago) goes here --
                                                         // after the last stage too (a re-store nobody reads): no branch
          bf16x8 av0[2][NP], bv0[2][NP], av1[2][NP], bv1[2][NP];
          load_stage(j, min(st + G_PD, nst - 1));        // slot j held stage st: stored during stage st - 1
          read_frags(av0, bv0, base, 0);
          __builtin_amdgcn_sched_barrier(0);
          int grp = 0;
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int sum = NP - 1; sum >= 0; --sum)
#pragma unroll
              for (int pa = sum; pa >= 0; --pa) {
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                  for (int h = 0; h < 2; ++h)
                    acc[g][h] = s2 == 0
                        ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(av0[g][pa], bv0[h][sum - pa], acc[g][h], 0, 0, 0)
                        : __builtin_amdgcn_mfma_f32_32x32x16_bf16(av1[g][pa], bv1[h][sum - pa], acc[g][h], 0, 0, 0);
                if (grp < 3 * NP) piece((j + 1) % G_PD, grp / NP, grp % NP, nxt);
                if (grp == 1) read_frags(av1, bv1, base, 1);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // MFMA
                  if (grp == 1) __builtin_amdgcn_sched_group_barrier(0x100, NP, 0);        // DS reads
                  if (grp < 3 * NP) __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);     // VALU
                }
                if (grp < 3 * NP) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);       // DS write
                __builtin_amdgcn_sched_barrier(0);
                ++grp;
              }
          __syncthreads();
        }
      }
    }
  }

  // ---- epilogue: C (or this split's slab) row-major [M][N]; the bias goes in with split 0
  float* out = G.C + (G.ksplit > 1 ? (size_t)split * G.M * G.N : 0);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int n = n0 + wn * 64 + h * 32 + l32;
    const float bvv = (G.bias && split == 0 && n < G.N) ? G.bias[n] : 0.f;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int r16 = 0; r16 < 16; ++r16) {
        const int m = m0 + wm * 64 + g * 32 + acc_row(r16, lane);
        if (m < G.M && n < G.N) out[(size_t)m * G.N + n] = acc[g][h][r16] + bvv;
      }
  }
}

// K split: as many splits as keep >= 8 stages each and bring the grid to about one workgroup per CU
int gemm_ksplit(int M, int N, int K) {
  const long tiles = (long)cdiv(M, GTM) * cdiv(N, GTN);
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
  const dim3 grid(cdiv(N, GTN), cdiv(M, GTM), ks);
  const bool at = a_k_stride != 1, bt = b_k_stride != 1;
  const int rc = planes == 3 ? launch_gemm<3>(G, at, bt, grid, st) : launch_gemm<2>(G, at, bt, grid, st);
  if (rc) return rc;
  if (ks > 1) return vg_internal_wgrad_reduce((const float*)workspace, C, M * N, ks, st);   // fixed-order sum of the slabs
  return 0;
}
