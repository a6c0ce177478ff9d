// Stride-2 5x5 convolution / transposed convolution forward in split-bf16 arithmetic for gfx950, with BOTH
// MFMA operands read from LDS: vg_conv5x5_fwd_bf16split (stride 2) and vg_convT5x5_fwd_bf16split (stride 2) run
// here.  Same arithmetic as conv_bf16split.hip (every fp32 operand split into NP bf16 planes, the plane products
// whose indices sum to < NP on v_mfma_f32_32x32x16_bf16, fp32 accumulate; NP = 3 is the product default and
// fp32-equivalent); nn.Conv2d / nn.ConvTranspose2d forward of /root/reference/models/model.py:392-398, 453-456,
// 495-505 and each other's data gradients.
//
// Why a second kernel: in conv_bf16split.hip every wavefront fetches its filter fragments global -> VGPR
// (8 waves x 4-6 KB per tap through the CU's vector L1) and the matrix pipe is ~55 % busy.  Here
//   * one workgroup = 8 wavefronts (2 per SIMD) owns a CU: tile = TN cout x TM pixels, WC x WP wavefronts,
//     FC x FP fragments of 32 x 32 each;
//   * the pre-split, pre-packed filter ([class][chunk][step][plane][k-block][cout] x 8 bf16) is a verbatim
//     image of what the MFMA A operand wants, so the slice of one K step (16 k) goes global -> LDS by
//     `global_load_lds_dwordx4` DMA into a 3-slot ring, three steps ahead of its use, with counted
//     `s_waitcnt vmcnt` and ONE raw `s_barrier` per step (never a drain);
//   * the input patch is staged through registers (it needs the fp32 -> bf16-plane split), double-buffered:
//     the next (half-)chunk is loaded, split and written while the current one feeds the MFMAs;
//   * fragments for step k+1 are read (ds_read_b128, conflict-free layouts) while the MFMAs of step k run.
// Forward (stride 2): a K step is 8 channels x 2 taps (k-block 0 / 1 = consecutive taps), so that the two
// resident patch buffers are 8-channel halves: 25 steps per 16-channel chunk, step 12 pairs tap 24 of both
// halves -- no padding step.  Transposed (stride 2): the 4 output-parity classes (3x3, 3x2, 2x3, 2x2 taps) as
// in conv_igemm.hip, K step = 16 channels x 1 tap.
#include "common.hpp"
#include "vaegan_hip.h"

namespace {

enum { R_FWD = 0, R_TR = 1 };
constexpr int RNT = 512;          // 8 wavefronts
// Depth of the filter ring: DMA(k + NSLOT) is issued in body k.  3 slots.  With fp16 planes a step holds half the MFMAs
// of a bf16x6 step and three steps of run-ahead looked marginal against the L2 -> LDS latency, so round 4 timed 4, 5 and
// 6 slots (-DVG_RING_SLOTS=n, fp16 planes only): 169.6-173.0 us against 169.9-172.0 on the dominant layer, no
// difference (profiles/r04_logs/r4_slots.log) -- the wavefronts do not wait for the DMA.
#ifndef VG_RING_SLOTS
#define VG_RING_SLOTS 3
#endif
#ifndef VG_RING_REGF
#define VG_RING_REGF 0
#endif
// Timing experiment for the K = 32 body planned in DESIGN.md section 4.10: NO cross-body fragment prefetch -- body k reads
// the fragments of step k itself, right behind its barrier (the filter DMA of body k then refills the slot of step k - 1,
// one step less of run-ahead: build with -DVG_RING_SLOTS=4 to keep three).
#ifndef VG_RING_NOPF
#define VG_RING_NOPF 0
#endif
constexpr int ring_slots(bool f16, int slotu, int patchu) {
  if (!f16) return 3;
  int n = VG_RING_SLOTS;
  while (n > 3 && (n * slotu + patchu + 1) * 16 > 160 * 1024) --n;
  return n;
}
constexpr int RING_MAX_SLOTS = 6;      // <= VG_PACK_SPARE of conv_bf16split.hip: the run-ahead past the last step reads zeros
static_assert(VG_RING_SLOTS >= 3 && VG_RING_SLOTS <= RING_MAX_SLOTS, "ring depth");

// 16-byte units per patch row; same bank rules as conv_bf16split.hip (fragments of 2 rows x 16 pixels need the
// two rows 0 (mod 16) units apart, 4 rows x 8 pixels 8 (mod 16))
constexpr int ring_cols(int mode, int TW, int PW) {
  if (mode == R_FWD) {                                   // per column parity; a row is 2 * COLS units
    const int need = (PW + 1) / 2;
    if (TW == 8) { int c = need; while ((c & 3) != 2) ++c; return c; }
    return (need + 3) & ~3;
  }
  if (TW == 16) return (PW + 15) & ~15;
  if (TW == 8) { int c = PW; while ((c & 15) != 8) ++c; return c; }
  return (PW + 3) & ~3;
}

template <int MODE_, int NB_, int TH_, int TW_, int WC_, int FC_, int FP_, int NP_, bool F16_ = false>
struct RCfg {
  static constexpr int MODE = MODE_, NB = NB_, TH = TH_, TW = TW_, WC = WC_, WP = 8 / WC_, FC = FC_, FP = FP_, NP = NP_;
  static constexpr bool F16 = F16_;       // fp16 planes (NP = 2, "fp16x3") instead of bf16 ones
  static_assert(!F16_ || NP_ == 2, "fp16 planes: hi + lo");
  static constexpr int S = 2;
  static constexpr int TN = 32 * WC * FC, TM = NB * TH * TW;
  static constexpr int NTMAX = (MODE == R_FWD) ? 5 : 3;
  static constexpr int PH = (MODE == R_FWD) ? S * (TH - 1) + 5 : TH + NTMAX - 1;
  static constexpr int PW = (MODE == R_FWD) ? S * (TW - 1) + 5 : TW + NTMAX - 1;
  static constexpr int COLS = ring_cols(MODE, TW, PW);
  static constexpr int ROWU = (MODE == R_FWD) ? 2 * COLS : COLS;      // units per patch row
  static constexpr int IMGU = NB * PH * ROWU;                         // units per (plane, 8-channel block) image
  // patch buffers: FWD two 8-channel halves [half][plane][image]; TR two 16-channel chunks [buf][plane][kb][image]
  static constexpr int BUFU = (MODE == R_FWD) ? NP * IMGU : NP * 2 * IMGU;
  static constexpr int PATCHU = 2 * BUFU;
  static constexpr int SLOTU = NP * 2 * TN;                            // one K step of the filter: [plane][kb][cout]
  static constexpr int NSLOT = ring_slots(F16_, SLOTU, PATCHU);
  static constexpr int RINGU = NSLOT * SLOTU;
  static constexpr int LDSU = RINGU + PATCHU + 1;                      // + one dummy unit for masked staging writes
  static constexpr int DMA_TOTAL = SLOTU / 64;                         // 1 KiB DMA instructions per step
  static constexpr int NDMA = (DMA_TOTAL + 7) / 8;                     // per wavefront (every wavefront the same count)
  // staged units per staging event: NB * PH * PW pixels of 8 channels (FWD) or of 2 x 8 channels (TR: unit q of a
  // thread belongs to k-block q & 1, so that a unit's channels are wave-uniform)
  static constexpr int NPIX = NB * PH * PW;
  static constexpr int NQ = ((MODE == R_FWD) ? 1 : 2) * ((NPIX + RNT - 1) / RNT);
  static constexpr int NL = NQ * 8;                                    // plain global loads per staging event
  static constexpr int NCLS = (MODE == R_FWD) ? 1 : 4;
  static_assert(TM == 32 * WP * FP, "pixel tile");
  static_assert(WC == 1 || WC == 2 || WC == 4 || WC == 8, "wavefront grid");
  static_assert(SLOTU % 64 == 0, "a step's filter slice is whole DMA instructions");
  static_assert(LDSU * 16 <= 160 * 1024, "LDS");
};

struct RArgs {
  const float* x;
  const bf16x8* w;     // packed filter
  const float* bias;
  float* y;
  int B, Cin, XH, XW, Cout, CoutP, YH, YW;
  int ntiles_n, tiles_w, tiles_hw, blocks_per_cls;
  int cps;             // channel chunks per K split
  size_t ysplit;       // elements per partial output slab (ksplit > 1: y points at the slabs)
  // fused BatchNorm (vg_conv_fusion): the INPUT is read as act(x * in_scale[c] + in_shift[c]) -- the train-mode
  // BatchNorm + activation of the producing layer applied while the patch is staged (zero padding applies to the
  // activated tensor) -- and per-channel sums of the OUTPUT (sum y, sum y^2 over this workgroup's pixels, one slot
  // per wavefront row) are left for the next BatchNorm's statistics.  NULL pointers: plain convolution.
  const float* in_scale;
  const float* in_shift;
  int in_act;
  float* stats;        // [slot][Cout][2]
  // fp16 planes: in_amax[0] >= max |activated input| (device); w_unscale[0] = the inverse of the power of two the pack
  // kernel multiplied the filter by (stored behind the pack).  NULL for bf16 planes.
  const float* in_amax;
  const float* w_unscale;
  int paired;          // transposed: a workgroup runs two parity classes of its tile (see conv5x5_ring_kernel)
};

// steps (of 16 k) of the transposed classes before (R, SS), per chunk of 16 channels
__host__ __device__ constexpr int tr_taps_before(int R, int SS) {
  int n = 0;
  for (int r = 0; r < 2; ++r)
    for (int s = 0; s < 2; ++s) {
      if (r == R && s == SS) return n;
      n += ((5 - r + 1) / 2) * ((5 - s + 1) / 2);
    }
  return n;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// AFF: the input is read through a producer's BatchNorm + activation (RArgs::in_scale); a template parameter, so that
// plain launches (every data gradient) carry none of its arithmetic and the coefficients of the others come by scalar
// loads from the kernel argument itself.  (Round 3 selected between in_scale and a __device__ constant at run time: the
// compiler could not tell the address space, loaded the 16 coefficients of a unit with flat_load and drained vmcnt(0)
// -- the whole filter ring -- in front of every unit's arithmetic: ~2000 cycles per staging event.)
// R2 >= 0 (transposed form only): the workgroup computes TWO output-parity classes of its tile, (R, SS) and (R2, SS2),
// from ONE staged patch: per chunk the K steps of the first class, then those of the second, each into its own
// accumulators; the filter DMA alternates between the two classes' streams (which stream a step belongs to is a
// compile-time property of the unrolled body).  Half the staging work and input traffic of two separate passes.
template <class C, int R, int SS, bool AFF, int R2 = -1, int SS2 = -1>
__device__ __forceinline__ void ring_body(const RArgs& A, f32x4* lds, int bid, int split) {
  constexpr int MODE = C::MODE, S = C::S, NB = C::NB, TH = C::TH, TW = C::TW, PH = C::PH, PW = C::PW;
  constexpr int COLS = C::COLS, ROWU = C::ROWU, IMGU = C::IMGU, NQ = C::NQ, FC = C::FC, FP = C::FP, NP = C::NP;
  constexpr int NTMAX = C::NTMAX, BUFU = C::BUFU, SLOTU = C::SLOTU, NDMA = C::NDMA, NL = C::NL, TN = C::TN;
  constexpr int NSLOT = C::NSLOT;
  constexpr bool F16 = C::F16;
  constexpr int NTH = (MODE == R_FWD) ? 5 : (5 - R + 1) / 2;     // taps along h / w in this class
  constexpr int NTW = (MODE == R_FWD) ? 5 : (5 - SS + 1) / 2;
  constexpr bool TWO = R2 >= 0;
  static_assert(!TWO || MODE == R_TR, "two classes: transposed form");
  constexpr int NTH2 = TWO ? (5 - R2 + 1) / 2 : 0, NTW2 = TWO ? (5 - SS2 + 1) / 2 : 1;
  constexpr int NA = NTH * NTW, NB2 = NTH2 * NTW2;                // K steps of the first / second class per chunk
  constexpr int NSTEP = NA + NB2;                                 // K steps per 16-channel chunk (FWD: 25)
  constexpr int NACC = TWO ? 2 : 1;
  static_assert(!TWO || NA >= NSLOT, "the prologue's DMAs all belong to the first class");
  constexpr int PSTEP = (MODE == R_FWD) ? S : 1;
  constexpr int RING0 = 0, PATCH0 = C::RINGU, DUMMY = C::RINGU + C::PATCHU;

  // Timing experiments only (scripts/abl_ring.py builds one-off libraries with -DVG_RING_ABL=bits; results are
  // then wrong): 1 no filter DMA, 2 no barrier, 4 no patch staging, 8 no counted vmcnt wait, 16 no pixel-fragment
  // reads, 32 no filter-fragment reads, 64 no MFMAs.  Compile-time: a run-time switch changed the schedule.
#ifdef VG_RING_ABL
  constexpr int abl = VG_RING_ABL;
#else
  constexpr int abl = 0;
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform: keeps the DMA's LDS base in SGPRs
#ifndef VG_RING_STAGGER
#define VG_RING_STAGGER 0
#endif
  constexpr bool STG = VG_RING_STAGGER != 0;
  static_assert(!(STG && VG_RING_REGF), "stagger build: DMA path only");
  const bool early = STG && wid >= 4;                             // wave-uniform: meets each barrier half a body late
  const bool late = wid >= 4;      // wave-uniform: the second-dispatched half (timing experiments VG_RING_PLACE = 1, 2)
  const int kb = lane >> 5, l32 = lane & 31;
  const int wc = wid % C::WC, wp = wid / C::WC;
  // XCD-aware placement (conv_igemm.hip): the cout tiles of one pixel tile share an XCD
  int nt, pt;
  {
    const int ntn = A.ntiles_n, npatch = A.blocks_per_cls / ntn, full = (npatch / 8) * 8 * ntn;
    if (bid < full) {
      const int xcd = bid & 7, j = bid >> 3;
#if defined(VG_RING_PT_INTERLEAVED)            // timing experiments: the previous map (neighbouring pixel tiles on 8 XCDs)
      pt = (j / ntn) * 8 + xcd;
#else
      // an XCD works through a contiguous run of pixel tiles: neighbours in the image (shared halo rows, shared cache
      // lines of a row) follow each other through the same L2
      pt = xcd * (npatch / 8) + j / ntn;
#endif
      nt = j % ntn;
    } else {
      const int t = bid - full;
      pt = (npatch / 8) * 8 + t / ntn;
      nt = t % ntn;
    }
  }
  const int sp = pt % A.tiles_hw, bg = pt / A.tiles_hw;
  const int th0 = (sp / A.tiles_w) * TH, tw0 = (sp % A.tiles_w) * TW;
  const int b0 = bg * NB, n0 = nt * TN;
  const int Cin = A.Cin, Cout = A.Cout, XH = A.XH, XW = A.XW, HW = XH * XW;
  const int ih0 = (MODE == R_FWD) ? th0 * S - 2 : th0 - (NTMAX - 1 - 2 / S);
  const int iw0 = (MODE == R_FWD) ? tw0 * S - 2 : tw0 - (NTMAX - 1 - 2 / S);
  const float* xb = A.x + (size_t)b0 * Cin * HW;

  // ---- staging map.  FWD: unit e = (image, row, column) of ONE 8-channel half; TR: (k-block, image, row, column).
  // Recomputed from the thread id at every staging event (a few dozen VALU, four events per 25 steps) instead of
  // living in registers through the MFMA loop: the loop is at the 256-VGPR budget.
  struct Unit {
    unsigned ofs;     // BYTE offset from a wave-uniform base: loads use the SGPR-base + 32-bit VGPR-offset form
    int dst;          // LDS unit within a buffer's plane image, -1: not a unit of the patch
    bool ok;          // inside the image (else zero)
  };
  auto unit_of = [&](int q) -> Unit {
    int t0 = tid;
    asm volatile("" : "+v"(t0));                       // opaque: keeps the map from being hoisted out of the loop
    const int kbs = (MODE == R_FWD) ? 0 : (q & 1);
    const int e = t0 + ((MODE == R_FWD) ? q : (q >> 1)) * RNT;
    const int col = e % PW;
    int t = e / PW;
    const int r = t % PH;
    t /= PH;
    const int nb = t % NB;
    const int ih = ih0 + r, iw = iw0 + col;
    const bool in = e < C::NPIX;
    Unit u;
    u.ok = in && ih >= 0 && ih < XH && iw >= 0 && iw < XW && (b0 + nb) < A.B;
    const int nbc = min(nb, A.B - 1 - b0), ihc = min(max(ih, 0), XH - 1), iwc = min(max(iw, 0), XW - 1);
    u.ofs = 4u * (unsigned)((nbc * Cin + kbs * 8) * HW + ihc * XW + iwc);
    u.dst = in ? kbs * IMGU + (nb * PH + r) * ROWU + ((MODE == R_FWD) ? (col & 1) * COLS + (col >> 1) : col) : -1;
    return u;
  };

  float preg[NQ][8];
  int staged_c0 = 0;      // first channel of the units in preg (wave-uniform)
  const float aff_slope = A.in_act == VG_ACT_NONE ? 1.f : (A.in_act == VG_ACT_RELU ? 0.f : 0.2f);
  // fp16 planes: the activated input times an exact power of two (wave-uniform, from the caller's bound on max |input|)
  float x_scale = 1.f, x_unscale = 1.f;
  if constexpr (F16) {
    const float amax = *A.in_amax;
    x_scale = f16_scale_of(amax);
    x_unscale = f16_unscale_of(amax);
  }
  // c0 = first channel of the 8 (FWD) / 16 (TR) channels to stage; clamped so that the address stays in the tensor
  // (past-the-end events happen at the last chunk and are never consumed)
  auto stage_load = [&](int c0) {
    c0 = min(c0, Cin - ((MODE == R_FWD) ? 8 : 16));
    staged_c0 = c0;
    unsigned ofs[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) ofs[q] = unit_of(q).ofs;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const char* xj = (const char*)(xb + (size_t)(c0 + j) * HW);      // wave-uniform
#pragma unroll
      for (int q = 0; q < NQ; ++q) preg[q][j] = *(const float*)(xj + ofs[q]);
    }
  };
  auto split_unit = [&](int q, int bufu) {          // fp32 -> planes of staged unit q, written to buffer `bufu`
    const Unit u = unit_of(q);
    f32x4 pl[NP];
    float vv[8];
    // BatchNorm + activation of the producing layer, on load.  Scalar loads: a unit's channels are wave-uniform.
    const int cb = __builtin_amdgcn_readfirstlane(staged_c0 + ((MODE == R_FWD) ? 0 : (q & 1) * 8));
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = preg[q][j];
      if constexpr (AFF) {
        v = fmaf(v, A.in_scale[cb + j], A.in_shift[cb + j]);
        v = act_slope(v, aff_slope);                  // slope 1: identity, 0: ReLU, 0.2: LeakyReLU
      }
      v = u.ok ? v : 0.f;                             // zero padding pads the ACTIVATED tensor
      vv[j] = F16 ? v * x_scale : v;
    }
    split_planes16<NP, F16>(vv, pl);
    const int d = PATCH0 + bufu + u.dst;            // masked units go to the dummy unit (no divergent branch)
#pragma unroll
    for (int p = 0; p < NP; ++p)
      lds[(u.dst >= 0) ? d + p * ((MODE == R_FWD) ? IMGU : 2 * IMGU) : DUMMY] = pl[p];
  };
  auto stage_store = [&](int bufu) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) split_unit(q, bufu);
  };

  // ---- filter DMA: step `st` (global step index within this class) -> ring slot `slot`
  const int CoutP = A.CoutP;
  const size_t wstep = (size_t)2 * NP * CoutP;      // units per step in the pack
  const int nchunks = Cin / 16;
  const bf16x8* wcls = A.w + (size_t)((MODE == R_FWD) ? 0 : tr_taps_before(R, SS)) * nchunks * wstep;
  const bf16x8* dma_ptr2 = A.w + (size_t)(TWO ? tr_taps_before(R2 < 0 ? 0 : R2, SS2 < 0 ? 0 : SS2) : 0) * nchunks * wstep;
  // DMA instruction i of a step covers units [64 i, 64 i + 64) of the slot image [plane][kb][TN]
  unsigned dma_src[NDMA];     // per-lane BYTE offset within a pack step
  int dma_dst[NDMA];          // wave-uniform unit offset within the slot
#pragma unroll
  for (int j = 0; j < NDMA; ++j) {
    const int i = (wid * NDMA + j) % C::DMA_TOTAL;  // surplus instructions repeat earlier ones (same bytes, same place)
    const int u = i * 64 + lane;                    // unit within the slot
    const int row = u / TN, col = u % TN;           // row = plane * 2 + kb
    dma_src[j] = 16u * (unsigned)(row * CoutP + n0 + col);
    dma_dst[j] = i * 64;
  }
  const bf16x8* dma_ptr1 = wcls;                    // pack address of the next step to fetch (advanced by dma_next)
  auto dma_next = [&](int slot, bool second = false) {
    const bf16x8*& dma_ptr = second ? dma_ptr2 : dma_ptr1;
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
#if !defined(VG_RING_DMA_VADDR)
      // scalar-base form: SGPR-pair base + one 32-bit lane offset, M0 = the slot's LDS address.  The builtin gives every
      // lane a 64-bit address (a v_lshl_add_u64 per instruction and twice the address registers to read): same-box A/B
      // 151.7-152.1 against 155.9-156.4 us on the dominant layer (profiles/r04_logs/r4_saddr.log).  hipcc does not see
      // this VMEM instruction; its own waits for the staging loads only become more conservative (completion is in order),
      // the waits for the DMA are the counted ones of this file.
      const unsigned m0v = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(lds + RING0 + slot * SLOTU + dma_dst[j]);
      asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(dma_src[j]), "s"(dma_ptr), "s"(m0v) : "memory", "m0");
#else
      __builtin_amdgcn_global_load_lds((const void*)((const char*)dma_ptr + dma_src[j]),
                                       (__attribute__((address_space(3))) void*)(lds + RING0 + slot * SLOTU + dma_dst[j]),
                                       16, 0, 0);
#endif
    }
    dma_ptr += wstep;
  };
  // Filter through registers instead (fp16 planes, -DVG_RING_REGF=1): the step's slice as NDMA plain 16-byte loads per
  // wavefront (scalar base + lane offset, seen by the compiler: it counts their vmcnt itself), written to the ring one
  // body later with ds_write_b128.  Body k writes step k+2 (loaded in body k-1) into slot (k+2) % 3 -- free since the
  // barrier of body k-1: step k-1's fragments were read in body k-2 -- and loads step k+3.
  constexpr bool REGF = (VG_RING_REGF != 0) && F16;
  static_assert(!REGF || NSLOT == 3, "register path: three slots");
  f32x4 freg[REGF ? NDMA : 1];
  auto filt_load = [&](bool second = false) {
    const bf16x8*& dma_ptr = second ? dma_ptr2 : dma_ptr1;
#pragma unroll
    for (int j = 0; j < NDMA; ++j) freg[j] = *(const f32x4*)((const char*)dma_ptr + dma_src[j]);
    dma_ptr += wstep;
  };
  auto filt_store = [&](int slot) {
#pragma unroll
    for (int j = 0; j < NDMA; ++j) lds[RING0 + slot * SLOTU + dma_dst[j] + lane] = freg[j];
  };

  // ---- per-lane operand bases (units)
  int base_a[FC], base_b[FP];
#pragma unroll
  for (int g = 0; g < FC; ++g) base_a[g] = RING0 + kb * TN + (wc * FC + g) * 32 + l32;
#pragma unroll
  for (int f = 0; f < FP; ++f) {
    const int m = (wp * FP + f) * 32 + l32;
    const int nb = m / (TH * TW), r = m % (TH * TW);
    base_b[f] = PATCH0 + (nb * PH + PSTEP * (r / TW)) * ROWU + r % TW + ((MODE == R_TR) ? kb * IMGU : 0);
  }

  f32x16 accs[NACC][FC][FP];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int g = 0; g < FC; ++g)
#pragma unroll
      for (int f = 0; f < FP; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) accs[a][g][f][r] = 0.f;

  const int c_begin = split * A.cps, c_end = min(c_begin + A.cps, nchunks);
  const int nsteps = (c_end - c_begin) * NSTEP;     // this workgroup's K steps

  // patch offset (units) of K step `s` of a chunk for k-block `k`, relative to base_b: FWD buffer = 8-channel half
  auto fwd_tap_off = [](int t) constexpr { return (t / 5) * ROWU + ((t % 5) & 1) * COLS + ((t % 5) >> 1); };
  auto step_off = [&](int s, int k) constexpr -> int {
    if constexpr (MODE == R_FWD) {
      const int half = (s < 12) ? 0 : (s == 12 ? k : 1);
      const int tap = (s < 12) ? 2 * s + k : (s == 12 ? 24 : 2 * (s - 13) + k);
      return half * BUFU + fwd_tap_off(tap);
    } else {
      if (TWO && s >= NA) return (NTMAX - 1 - (s - NA) / NTW2) * ROWU + (NTMAX - 1 - (s - NA) % NTW2);
      return (NTMAX - 1 - s / NTW) * ROWU + (NTMAX - 1 - s % NTW);
    }
  };

  bf16x8 av[2][FC][NP], bv[2][FP][NP];
  // Fragment read j of a step (j < NRD): the FC * NP filter fragments, then the FP * NP pixel fragments.
  constexpr int NRD = (FC + FP) * NP;
  auto read_one = [&](int buf, int j, int so, int bo) {
    if (j < FC * NP) {
      const int g = j / NP, p = j % NP;
      av[buf][g][p] = __builtin_bit_cast(bf16x8, lds[base_a[g] + so + p * 2 * TN]);
    } else {
      const int f = (j - FC * NP) / NP, p = (j - FC * NP) % NP;
      bv[buf][f][p] = __builtin_bit_cast(bf16x8, lds[base_b[f] + bo + p * ((MODE == R_FWD) ? IMGU : 2 * IMGU)]);
    }
  };
  // patch offset of chunk-step s for this lane's k-block (+ TR: the chunk's patch buffer)
  auto patch_off = [&](int s, int pbuf) -> int {
    if constexpr (MODE == R_FWD) return kb ? step_off(s, 1) : step_off(s, 0);
    else return pbuf + step_off(s, 0);
  };
  // MFMA i of a step (i < NMF): plane products with index sum < NP, smallest terms first, all fragments per product
  constexpr int NPROD = NP * (NP + 1) / 2, NMF = NPROD * FC * FP;
  auto mfma_one = [&](int buf, int i, int a = 0) {
    auto& acc = accs[a];
    const int pr = i / (FC * FP), g = (i / FP) % FC, f = i % FP;
    int pa = 0, pb = 0, n = 0;                       // pr-th pair in the order (sum = NP-1 .. 0; pa = sum .. 0)
#pragma unroll
    for (int sum = NP - 1; sum >= 0; --sum)
#pragma unroll
      for (int a = sum; a >= 0; --a) {
        if (n == pr) { pa = a; pb = sum - a; }
        ++n;
      }
    if constexpr ((abl & 128) != 0) {
      // timing experiment: the same FLOPs as two v_mfma_f32_16x16x32_bf16 (numerically meaningless here)
      typedef float f32x4v __attribute__((ext_vector_type(4)));
      f32x4v c0 = {acc[g][f][0], acc[g][f][1], acc[g][f][2], acc[g][f][3]};
      f32x4v c1 = {acc[g][f][4], acc[g][f][5], acc[g][f][6], acc[g][f][7]};
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[buf][g][pa], bv[buf][f][pb], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[buf][g][pa], bv[buf][f][pb], c1, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) { acc[g][f][r] = c0[r]; acc[g][f][4 + r] = c1[r]; }
    } else {
      acc[g][f] = mfma_split16<F16>(av[buf][g][pa], bv[buf][f][pb], acc[g][f]);
    }
  };
  // a step's MFMAs with the next step's fragment reads spread between them (one read per RS MFMAs), pinned
  constexpr int RS = (NMF / NRD) > 0 ? (NMF / NRD) : 1;
  // ---- prologue: first patch buffer, ring slots 0..2, fragments of step 0
  dma_ptr1 += (size_t)c_begin * NA * wstep;
  dma_ptr2 += (size_t)c_begin * NB2 * wstep;
  if constexpr (REGF) {
    filt_load(); filt_store(0);
    filt_load(); filt_store(1);
    filt_load();                                     // step 2 stays in registers until body 0
  } else {
#pragma unroll
    for (int i = 0; i < NSLOT - (VG_RING_NOPF ? 1 : 0); ++i) dma_next(i);
  }
  stage_load(c_begin * 16);
  stage_store(0);
  wait_vmcnt<0>();
  __syncthreads();
  if (!VG_RING_NOPF) {
#pragma unroll
    for (int j = 0; j < NRD; ++j) read_one(0, j, 0, patch_off(0, 0));
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

#ifdef VG_RING_STAMP      // diagnostic build (scripts/ring_stamps.py): where a wavefront's cycles go, summed over its steps --
  // [body top .. barrier passed], [.. last MFMA issued], [.. waits done].  s_memtime answers asynchronously (lgkmcnt): the
  // three stamps of a body are read after the body's own lgkmcnt(0); the sums leave through the statistics buffer.
  unsigned long long tk_a = 0, tk_b = 0, tk_c = 0, tk_d = 0, sum_bar = 0, sum_mf = 0, sum_tail = 0, sum_all = 0;
  unsigned long long tm[NMF < 12 ? NMF : 12];      // the issue times of the LAST step's first 12 MFMAs (overwritten every step)
#pragma unroll
  for (int i = 0; i < NMF && i < 12; ++i) tm[i] = 0;
#endif
  int slot = 0;        // ring slot of the current step
  for (int ch = c_begin; ch < c_end; ++ch) {
    const int pcur = (MODE == R_TR) ? ((ch - c_begin) & 1) * BUFU : 0;
    const int pnxt = (MODE == R_TR) ? BUFU - pcur : 0;
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;
      // staging events of this body.  FWD: half B of this chunk is loaded at body 0 and written at body 4 (needed
      // by the reads of step 12, issued in body 11); half A of the NEXT chunk is loaded at body 13 (its buffer is
      // free once step 12 has been read) and written at body 17.  TR: the next chunk at body 0 / body LW.
      constexpr int LW = (NSTEP >= 6) ? 3 : 2;
      const bool ld = (MODE == R_FWD) ? (s == 0 || s == 13) : (s == 0);
      const bool wr = (MODE == R_FWD) ? (s == 4 || s == 17) : (s == LW);
      const int wbuf = (MODE == R_FWD) ? (s == 4 ? BUFU : 0) : pnxt;
      // B_k: slot (k+1)%3 and the patch writes of body k-1 are visible.  Staggered build: the second-dispatched half of
      // the wavefronts (4-7: the SIMD partners of 0-3) meets the same barrier in the MIDDLE of its body, see below.
#ifdef VG_RING_STAMP
      asm volatile("s_memtime %0" : "=s"(tk_a));
#endif
      if (!(abl & 2) && !early) __builtin_amdgcn_s_barrier();
#ifdef VG_RING_STAMP
      asm volatile("s_memtime %0" : "=s"(tk_b));
#endif
      __builtin_amdgcn_sched_barrier(0);
      const int nslot = (slot == NSLOT - 1) ? 0 : slot + 1;
      const int so = nslot * SLOTU;
      // the next step's patch offset (the last body reads step 0 of the next chunk)
      const int bo = (s + 1 < NSTEP) ? patch_off(s + 1, pcur) : patch_off(0, pnxt);
      if constexpr (STG) {
        // Stagger (MI355X_MICROARCH.md, "Two waves per SIMD", item 9): a body is [H1: the first half of the step's
        // MFMAs, nothing else][H2: the second half + EVERYTHING that touches memory -- filter DMA, staging loads, the
        // next step's fragment reads, patch writes].  Wavefronts 0-3 pass the step's barrier before H1, wavefronts 4-7
        // between H1 and H2: every memory operation of body k still falls between barriers k and k+1 for all eight
        // (the hazard analysis of the file header holds unchanged), but the two wavefronts of a SIMD are half a body
        // apart, so one of them has MFMAs to issue while the other drains its waits and sits at the barrier.
        // Accumulation order per output is unchanged: results are bit-identical to the lockstep build.
        constexpr int H1N = NMF / 2, H2N = NMF - H1N;
        constexpr int RPM = (NRD + H2N - 1) / H2N;           // fragment reads per MFMA of H2
#pragma unroll
        for (int i = 0; i < H1N; ++i)
          if (!(abl & 64)) mfma_one(cur, i, (TWO && s >= NA) ? 1 : 0);
        __builtin_amdgcn_sched_barrier(0);
        if (!(abl & 2) && early) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = H1N; i < NMF; ++i) {
          const int j = i - H1N;
          if (!(abl & 64)) mfma_one(cur, i, (TWO && s >= NA) ? 1 : 0);
          if (j == 0) {
            if (!(abl & 1)) dma_next(slot, TWO && (s + NSLOT) % NSTEP >= NA);      // step k+3 into the slot read during body k-1
            if (ld && !(abl & 4)) {
              if constexpr (MODE == R_FWD) stage_load(ch * 16 + (s == 0 ? 8 : 16));
              else stage_load((ch + 1) * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int r = 0; r < RPM; ++r)
            if (j * RPM + r < NRD && !(abl & ((j * RPM + r < FC * NP) ? 32 : 16))) read_one(nxt, j * RPM + r, so, bo);
          __builtin_amdgcn_sched_barrier(0);
          if (wr && !(abl & 4) && j < NQ) {
            split_unit(j, wbuf);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (wr && !(abl & 4)) {
#pragma unroll
          for (int q = H2N; q < NQ; ++q) split_unit(q, wbuf);
        }
      } else {
      // Placement of the step's vector-memory instructions (filter DMA, a staging event's loads behind it) and of a
      // written unit's split: behind MFMA 0 and between the later MFMAs, the same in all eight wavefronts.  Round 4 timed
      // two other placements with per-MFMA s_memtime stamps (scripts/ring_stamps.py, profiles/r04_logs/r4_stamps*.log):
      // the DMA half a body later in wavefronts 4-7 (VG_RING_PLACE=1), and the whole memory block in front of the MFMAs
      // in wavefronts 4-7 / behind them in 0-3 (VG_RING_PLACE=2).  Both within 2 % of this one (154.7-157.5 us against
      // 159.5-163 on the dominant layer): the two wavefronts of a SIMD do not interleave their MFMAs whatever the
      // placement -- the older one wins every arbitration while it has an instruction ready and runs its twelve first --
      // and an LDS-DMA instruction holds its in-order wavefront 70-190 cycles wherever it stands.
#ifndef VG_RING_PLACE
#define VG_RING_PLACE 0
#endif
      if (VG_RING_NOPF) {          // this step's own fragments, behind the barrier
#pragma unroll
        for (int j = 0; j < NRD; ++j) read_one(cur, j, slot * SLOTU, patch_off(s, pcur));
        __builtin_amdgcn_sched_barrier(0);
      }
      auto mem_block = [&]() {
        if constexpr (VG_RING_NOPF != 0) {
          if (!(abl & 1)) dma_next(slot == 0 ? NSLOT - 1 : slot - 1, TWO && (s + NSLOT - 1) % NSTEP >= NA);
        } else
        if constexpr (REGF) {
          if (!(abl & 1)) {
            filt_store(slot == 0 ? 2 : slot - 1);                                  // step k+2 -> slot (k+2) % 3
            filt_load(TWO && (s + NSLOT) % NSTEP >= NA);                           // step k+3
          }
        } else
        if (!(abl & 1)) dma_next(slot, TWO && (s + NSLOT) % NSTEP >= NA);          // step k+NSLOT into the slot read during body k-1
        if (ld && !(abl & 4)) {
          if constexpr (MODE == R_FWD) stage_load(ch * 16 + (s == 0 ? 8 : 16));
          else stage_load((ch + 1) * 16);
        }
        if (VG_RING_PLACE == 2 && wr && !(abl & 4)) {
#pragma unroll
          for (int q = 0; q < NQ; ++q) split_unit(q, wbuf);
        }
      };
      if (VG_RING_PLACE == 2 && late) {
        mem_block();
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < NMF; ++i) {
        if (!(abl & 64)) mfma_one(cur, i, (TWO && s >= NA) ? 1 : 0);
#ifdef VG_RING_STAMP
        if (i < 12) asm volatile("s_memtime %0" : "=s"(tm[i]));
#endif
        if (VG_RING_PLACE < 2 && (i == 0 || i == NMF / 2)) {
          if ((i != 0) == (VG_RING_PLACE == 1 && late)) mem_block();
          __builtin_amdgcn_sched_barrier(0);
        }
        if (!VG_RING_NOPF && i % RS == RS - 1 && i / RS < NRD) {
          if (!(abl & ((i / RS < FC * NP) ? 32 : 16))) read_one(nxt, i / RS, so, bo);
          __builtin_amdgcn_sched_barrier(0);
        }
        // a written unit's split (VALU) + ds_writes go between the later MFMAs
        if (VG_RING_PLACE < 2 && wr && !(abl & 4) && i >= NMF / 2 && (i - NMF / 2) < NQ) {
          split_unit(i - NMF / 2, wbuf);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int j = NMF / RS; j < NRD && !VG_RING_NOPF; ++j) read_one(nxt, j, so, bo);     // reads the MFMA count left over
      if constexpr ((abl & 256) != 0) {
        // timing experiment: what a v_mfma_f32_16x16x32_bf16 build would read on top -- (FC + FP) * 2 * 2 operand
        // registers per step instead of (FC + FP) * 3: one more read per fragment row / column (results discarded)
#pragma unroll
        for (int j = 0; j < FC + FP; ++j) {
          f32x4 d = (j < FC) ? lds[base_a[j] + so + TN] : lds[base_b[j - FC] + bo + 1];
          asm volatile("" ::"v"(d));
        }
      }
      if (VG_RING_PLACE < 2 && wr && !(abl & 4)) {
#pragma unroll
        for (int q = NMF - NMF / 2; q < NQ; ++q) split_unit(q, wbuf);
      }
      if (VG_RING_PLACE == 2 && !late) {
        __builtin_amdgcn_sched_barrier(0);
        mem_block();
      }
      }
      // DMA(k+2) has landed (DMA(k+3), and a staging event's loads for two bodies, stay in flight);
      // this wavefront's reads of step k+1 and its patch writes are done
      __builtin_amdgcn_sched_barrier(0);
#ifdef VG_RING_STAMP
      asm volatile("s_memtime %0" : "=s"(tk_c));
#endif
      // In flight behind DMA(k+2), in issue order: DMA(k+3) .. DMA(k+NSLOT), and a staging event's loads for the NSLOT - 1
      // bodies from the one that issued them (they follow that body's DMA; hipcc itself waits for them where the
      // registers are used)
      const bool ldw = (MODE == R_FWD) ? ((s >= 0 && s < NSLOT - 1) || (s >= 13 && s < 13 + NSLOT - 1)) : (s >= 0 && s < NSLOT - 1);
      if constexpr (REGF) {
        // every vector-memory instruction is the compiler's: it waits where a register is used
      } else
      if (abl & (8 | 4 | 1)) {                                      // ablations change what is in flight: drain or skip
        if (!(abl & 8)) wait_vmcnt<0>();
      } else if (ldw) wait_vmcnt<(NSLOT - 2) * NDMA + NL>(); else wait_vmcnt<(NSLOT - 2) * NDMA>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef VG_RING_STAMP
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tk_d), "+s"(tk_a), "+s"(tk_b), "+s"(tk_c));
      // the per-MFMA stamps are pinned until here too: s_memtime answers asynchronously, and a register the compiler
      // had meanwhile given to something else (an address) would be overwritten when the answer lands
      static_assert(NMF == 12 || NMF == 6 || NMF == 24, "stamp pinning below");
      if constexpr (NMF >= 12)
        asm volatile("" : "+s"(tm[0]), "+s"(tm[1]), "+s"(tm[2]), "+s"(tm[3]), "+s"(tm[4]), "+s"(tm[5]), "+s"(tm[6]), "+s"(tm[7]),
                     "+s"(tm[8]), "+s"(tm[9]), "+s"(tm[10]), "+s"(tm[11]));
      else
        asm volatile("" : "+s"(tm[0]), "+s"(tm[1]), "+s"(tm[2]), "+s"(tm[3]), "+s"(tm[4]), "+s"(tm[5]));
      sum_bar += tk_b - tk_a;
      sum_mf += tk_c - tk_b;
      sum_tail += tk_d - tk_c;
      sum_all += 1;
#endif
      __builtin_amdgcn_sched_barrier(0);
      slot = nslot;
    }
    if constexpr (NSTEP & 1) {     // odd step count: the fragments just read sit in set 1, the next chunk starts on set 0
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int g = 0; g < FC; ++g) av[0][g][p] = av[1][g][p];
#pragma unroll
        for (int f = 0; f < FP; ++f) bv[0][f][p] = bv[1][f][p];
      }
    }
  }
  wait_vmcnt<0>();     // nothing of the ring is in flight when the workgroup ends
  if constexpr (F16) {   // undo the two operands' power-of-two scales (two exact multiplications: no intermediate overflow)
    const float w_unscale = *A.w_unscale;
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
      for (int g = 0; g < FC; ++g)
#pragma unroll
        for (int f = 0; f < FP; ++f)
#pragma unroll
          for (int r = 0; r < 16; ++r) accs[a][g][f][r] = accs[a][g][f][r] * x_unscale * w_unscale;
  }

  const int YH = A.YH, YW = A.YW;
#ifdef VG_RING_STAMP
  if (A.stats && lane == 0 && bid < 64) {
    unsigned long long* o = (unsigned long long*)A.stats + (size_t)(bid * 8 + wid) * 4;
    o[0] = sum_bar; o[1] = sum_mf; o[2] = sum_tail; o[3] = sum_all;
    unsigned long long* o2 = (unsigned long long*)A.stats + 64 * 8 * 4 + (size_t)(bid * 8 + wid) * 16;
    o2[0] = tk_a; o2[1] = tk_b; o2[2] = tk_c; o2[3] = tk_d;
#pragma unroll
    for (int i = 0; i < NMF && i < 12; ++i) o2[4 + i] = tm[i];
  }
  const bool emit_stats = false;
#else
  const bool emit_stats = A.stats != nullptr;
#endif
  // one class's statistics + output (two classes: called twice)
  auto emit = [&](auto& acc, int Rc, int SSc) {
  // ---- optional: per-channel sums of the output for the next BatchNorm (ksplit == 1 only: the host says so)
  if (emit_stats) {
    bool pokf[FP];
#pragma unroll
    for (int f = 0; f < FP; ++f) {
      const int m = (wp * FP + f) * 32 + l32;
      const int nb = m / (TH * TW), r = m % (TH * TW);
      const int th = th0 + r / TW, tw = tw0 + r % TW;
      const int oh = (MODE == R_FWD) ? th : S * th + Rc, ow = (MODE == R_FWD) ? tw : S * tw + SSc;
      pokf[f] = (b0 + nb) < A.B && oh < YH && ow < YW;
    }
    const int cls = (MODE == R_FWD) ? 0 : 2 * Rc + SSc;
    const int slot = (cls * (A.blocks_per_cls / A.ntiles_n) + pt) * C::WP + wp;
    float* sb = A.stats + (size_t)slot * Cout * 2;
#pragma unroll
    for (int g = 0; g < FC; ++g)
#pragma unroll
      for (int r16 = 0; r16 < 16; ++r16) {
        const int co = n0 + (wc * FC + g) * 32 + acc_row(r16, lane);
        const float bvs = A.bias ? A.bias[min(co, Cout - 1)] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int f = 0; f < FP; ++f) {
          const float v = pokf[f] ? acc[g][f][r16] + bvs : 0.f;
          s1 += v;
          s2 = fmaf(v, v, s2);
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {       // the 32 lanes of a half hold the 32 pixels of this cout row
          s1 += __shfl_xor(s1, o, 64);
          s2 += __shfl_xor(s2, o, 64);
        }
        if (l32 == 0 && co < Cout) {
          sb[2 * co] = s1;
          sb[2 * co + 1] = s2;
        }
      }
  }
  // ---- epilogue: + bias, NCHW store (as conv_igemm.hip)
#pragma unroll
  for (int f = 0; f < FP; ++f) {
    const int m = (wp * FP + f) * 32 + l32;
    const int nb = m / (TH * TW), r = m % (TH * TW);
    const int th = th0 + r / TW, tw = tw0 + r % TW, b = b0 + nb;
    const int oh = (MODE == R_FWD) ? th : S * th + Rc;
    const int ow = (MODE == R_FWD) ? tw : S * tw + SSc;
    const bool pok = b < A.B && oh < YH && ow < YW;
    float* yb = A.y + (size_t)split * A.ysplit + ((size_t)b * Cout * YH + oh) * YW + ow;
#pragma unroll
    for (int g = 0; g < FC; ++g) {
      float bvv[16];
#pragma unroll
      for (int r16 = 0; r16 < 16; ++r16) {
        const int co = min(n0 + (wc * FC + g) * 32 + acc_row(r16, lane), Cout - 1);
        bvv[r16] = (A.bias && split == 0) ? A.bias[co] : 0.f;     // partial slabs: the bias goes in once
      }
#pragma unroll
      for (int r16 = 0; r16 < 16; ++r16) {
        const int co = n0 + (wc * FC + g) * 32 + acc_row(r16, lane);
        if (pok && co < Cout) yb[(size_t)co * YH * YW] = acc[g][f][r16] + bvv[r16];
      }
    }
  }
  };
  emit(accs[0], R, SS);
  if constexpr (TWO) emit(accs[1], R2, SS2);
}

template <class C, bool AFF>
__global__ __launch_bounds__(RNT, 2) void conv5x5_ring_kernel(RArgs A) {
  __shared__ f32x4 lds[C::LDSU];     // ONE array: [filter ring 3 slots][patch buffers][dummy]
  int bid = blockIdx.x;
  if constexpr (C::NCLS == 1) {
    const int split = bid / A.blocks_per_cls;
    ring_body<C, 0, 0, AFF>(A, lds, bid - split * A.blocks_per_cls, split);
  } else {
    const int per = A.blocks_per_cls;
    if (A.paired) {
      // The four parity classes of a tile read the SAME input patch; launched as four workgroups a whole launch apart,
      // every class fetched it from HBM again (232 MB per launch of the 256 -> 128 @16 -> 32 layer against 104 MB of
      // operands: profiles/r04_pmc_dominant_convT.json).  Where the grid allows it, one workgroup runs TWO classes of
      // its tile back to back -- (3x3 taps, then 2x2) or (3x2, then 2x3): 13 / 12 steps per chunk, so the halves are
      // balanced and 2 x 128 tiles are ONE round of the 256 CUs (the four-class grid was two rounds with the shortest
      // class last) -- and the two workgroups of a tile are XCD neighbours in launch order (blocks b, b + 8): the patch
      // comes from HBM once, from that XCD's L2 afterwards.  [pair][split][tile] per XCD-interleaved index.
      const int nsplit = gridDim.x / (2 * per);
      const int xcd = bid & 7, j = bid >> 3;
      const int pair = j & 1;
      bid = (j >> 1) * 8 + xcd;                 // in [0, per * nsplit): keeps its XCD (per % 8 == 0, the host checks)
      const int split = bid / per;
      bid -= split * per;
      // pair 0: classes (0,0) [3x3 taps] + (1,1) [2x2]; pair 1: (0,1) [3x2] + (1,0) [2x3] -- from ONE staged patch
      if constexpr (C::F16) {     // (the host pairs fp16-plane launches only: with three planes two accumulator sets spill)
        if (pair == 0) ring_body<C, 0, 0, AFF, 1, 1>(A, lds, bid, split);
        else ring_body<C, 0, 1, AFF, 1, 0>(A, lds, bid, split);
      }
      return;
    }
    // [class][split][tile]; class 0 (3x3 taps) first: the longest blocks start earliest
    const int nsplit = gridDim.x / (4 * per);
    const int cls = bid / (per * nsplit);
    bid -= cls * per * nsplit;
    const int split = bid / per;
    bid -= split * per;
    switch (cls) {
      case 0: ring_body<C, 0, 0, AFF>(A, lds, bid, split); break;
      case 1: ring_body<C, 0, 1, AFF>(A, lds, bid, split); break;
      case 2: ring_body<C, 1, 0, AFF>(A, lds, bid, split); break;
      default: ring_body<C, 1, 1, AFF>(A, lds, bid, split); break;
    }
  }
}

struct RFuse {
  const float* in_scale;
  const float* in_shift;
  int in_act;
  float* stats;
  const float* in_amax;
};

template <class C>
int launch_ring(const float* x, const bf16x8* w, const float* bias, float* y, int B, int Cin, int XH, int XW, int Cout,
                int ksplit, float* slabs, RFuse fu, hipStream_t st) {
  RArgs A;
  A.x = x; A.w = w; A.bias = bias; A.y = y;
  A.in_scale = fu.in_scale; A.in_shift = fu.in_shift; A.in_act = fu.in_act;
  A.stats = (ksplit == 1) ? fu.stats : nullptr;
  A.in_amax = fu.in_amax;
  // the pack's trailer (conv_bf16split.hip, pack kernel): behind the steps and the spare steps
  A.w_unscale = (const float*)(w + (size_t)(Cin / 16 * 25 + RING_MAX_SLOTS) * 2 * C::NP * ((Cout + 127) & ~127));
  A.B = B; A.Cin = Cin; A.XH = XH; A.XW = XW; A.Cout = Cout; A.CoutP = (Cout + 127) & ~127;
  int tsh, tsw;
  if (C::MODE == R_FWD) {
    A.YH = (XH - 1) / 2 + 1; A.YW = (XW - 1) / 2 + 1;
    tsh = A.YH; tsw = A.YW;
  } else {
    A.YH = XH * 2; A.YW = XW * 2;
    tsh = XH; tsw = XW;
  }
  A.tiles_w = cdiv(tsw, C::TW);
  A.tiles_hw = cdiv(tsh, C::TH) * A.tiles_w;
  A.ntiles_n = cdiv(Cout, C::TN);
  const long per_cls = (long)A.ntiles_n * A.tiles_hw * cdiv(B, C::NB);
  // transposed: two classes per workgroup when that still gives every CU a workgroup (and the XCD arithmetic holds)
  A.paired = (C::NCLS == 4 && C::F16 && per_cls * ksplit * 2 >= 256 && per_cls % 8 == 0) ? 1 : 0;
#ifdef VG_RING_UNPAIRED      // timing experiments
  A.paired = 0;
#endif
  const long grid = per_cls * (A.paired ? 2 : C::NCLS) * ksplit;
  if (grid <= 0 || grid > 0x7fffffffL) return VG_ERR_BAD_ARG;
  A.blocks_per_cls = (int)per_cls;
  A.cps = cdiv(Cin / 16, ksplit);
  A.ysplit = (size_t)B * Cout * A.YH * A.YW;
  if (ksplit > 1) A.y = slabs;
  if (A.in_scale) hipLaunchKernelGGL((conv5x5_ring_kernel<C, true>), dim3((unsigned)grid), dim3(RNT), 0, st, A);
  else hipLaunchKernelGGL((conv5x5_ring_kernel<C, false>), dim3((unsigned)grid), dim3(RNT), 0, st, A);
  VG_CHECK_LAUNCH();
  if (ksplit > 1) {
    if (A.ysplit > 0x7fffffffUL) return VG_ERR_BAD_ARG;
    return vg_internal_wgrad_reduce(slabs, y, (int)A.ysplit, ksplit, st);   // fixed-order sum of the slabs
  }
  return 0;
}

// pixel-tile geometry by the width of the tiled image (forward: output, transposed: input)
template <int MODE, int WC, int FC, int FP, int NP, bool F16>
int ring_geom(const float* x, const bf16x8* w, const float* bias, float* y, int B, int Cin, int XH, int XW, int Cout,
              int ksplit, float* slabs, RFuse fu, hipStream_t st) {
  const int tsw = (MODE == R_FWD) ? (XW - 1) / 2 + 1 : XW;
  constexpr int TM = 32 * (8 / WC) * FP;
  static_assert(TM == 128 || TM == 256, "pixel tile");
  if constexpr (TM == 128) {
    if (tsw > 8) return launch_ring<RCfg<MODE, 1, 8, 16, WC, FC, FP, NP, F16>>(x, w, bias, y, B, Cin, XH, XW, Cout, ksplit, slabs, fu, st);
    // transposed, 8-wide images, 256 cout, three planes: ring + the two 2 x 10 x 10 patches (24 units per row) exceed
    // the LDS; the plan sends those layers to the 128-cout tiles
    if constexpr (MODE == R_TR && WC * FC * 32 == 256 && NP == 3) return VG_ERR_BAD_ARG;
    else return launch_ring<RCfg<MODE, 2, 8, 8, WC, FC, FP, NP, F16>>(x, w, bias, y, B, Cin, XH, XW, Cout, ksplit, slabs, fu, st);
  } else {   // 256 pixels: whole 16 x 16 tiles only (the plan never picks it for narrower images)
    return launch_ring<RCfg<MODE, 1, 16, 16, WC, FC, FP, NP, F16>>(x, w, bias, y, B, Cin, XH, XW, Cout, ksplit, slabs, fu, st);
  }
}

VG_KNOB(int, g_ring_variant, -1);   // tuning build only: forced tile variant

}  // namespace

// Tile variants.  0: 256 cout x 128 px (4 x 2 wavefronts of 64 x 64); 1: 128 cout x 128 px (2 x 4 wavefronts of
// 64 cout x 32 px); 2: 128 cout x 128 px (4 x 2 wavefronts of 32 cout x 64 px); 3 (transposed only, small patch):
// 128 cout x 256 px (2 x 4 wavefronts of 64 x 64).
template <int MODE, int NP, bool F16>
static int ring_dispatch(int variant, const float* x, const bf16x8* w, const float* bias, float* y, int B, int Cin, int XH,
                         int XW, int Cout, int ksplit, float* slabs, RFuse fu, hipStream_t st) {
  switch (variant) {
    case 0: return ring_geom<MODE, 4, 2, 2, NP, F16>(x, w, bias, y, B, Cin, XH, XW, Cout, ksplit, slabs, fu, st);
    case 1: return ring_geom<MODE, 2, 2, 1, NP, F16>(x, w, bias, y, B, Cin, XH, XW, Cout, ksplit, slabs, fu, st);
    case 2: return ring_geom<MODE, 4, 1, 2, NP, F16>(x, w, bias, y, B, Cin, XH, XW, Cout, ksplit, slabs, fu, st);
    default:      // forward: with fp16 planes only (its patch does not fit beside the ring otherwise)
      if constexpr (MODE == R_TR || F16) return ring_geom<MODE, 2, 2, 2, NP, F16>(x, w, bias, y, B, Cin, XH, XW, Cout, ksplit, slabs, fu, st);
      return VG_ERR_BAD_ARG;
  }
}

// ---- decisions shared with conv_bf16split.hip ------------------------------------------------------------------
// K split (channel chunks over workgroups, partial outputs summed in a fixed order) for layers whose tile grid
// leaves CUs idle: the smallest power of two that gives >= 192 workgroups, every split owning >= 2 chunks.
static int ring_ksplit(long wgs, int nchunks) {
  int k = 1;
  while (wgs * k < 192 && k < 8 && nchunks / (2 * k) >= 2 && (2 * k - 1) * cdiv(nchunks, 2 * k) < nchunks) k *= 2;
  return k;
}

struct RingPlan {
  int variant, ksplit;
};

static RingPlan ring_plan(int mode, int B, int Cin, int XH, int XW, int Cout, int planes) {
  const bool f16 = (planes & VG_PLANES_F16_FLAG) != 0;      // two fp16 planes: smaller ring slots and patches, more tiles fit
  const int tsw = (mode == R_FWD) ? (XW - 1) / 2 + 1 : XW, tsh = (mode == R_FWD) ? (XH - 1) / 2 + 1 : XH;
  const int ncls = (mode == R_TR) ? 4 : 1;
  // pixel tiles of 128 (geometry 8 x 16 per image, or 2 images x 8 x 8)
  const long pt128 = (tsw > 8) ? (long)B * cdiv(tsh, 8) * cdiv(tsw, 16) : (long)cdiv(B, 2) * cdiv(tsh, 8) * cdiv(tsw, 8);
  RingPlan p;
  // measured at B = 128 (scripts/time_ring.py): 256-cout tiles for wide layers; for <= 128 output channels the
  // 4 x 2 wavefront arrangement (32 cout x 64 px per wavefront), and for the transposed form -- small patch -- the
  // 256-pixel tile when the image has whole 16 x 16 tiles
  if (Cout > 128) p.variant = 0;
  else p.variant = (mode == R_TR && tsw >= 16 && tsh >= 16) ? 3 : 2;
  // fp16 planes, forward: the 128 cout x 256 px tile fits too (89.6 KB patch + 24 KB ring; with three bf16 planes 170 KB)
  // -- half the workgroups, half the filter DMA per MFMA -- where whole 16 x 16 tiles still give every CU a workgroup
  if (f16 && mode == R_FWD && Cout <= 128 && tsw >= 16 && tsh >= 16 && tsw % 16 == 0 && tsh % 16 == 0 &&
      (long)B * (tsh / 16) * (tsw / 16) * cdiv(Cout, 128) >= 256)
    p.variant = 3;
  if (g_ring_variant >= 0) p.variant = g_ring_variant;
  if (p.variant == 3 && ((mode == R_FWD && !f16) || tsw <= 8 || tsw % 16 || tsh % 16)) p.variant = 1;
  // transposed, 8-wide images: the 256-cout tile's ring + two 2 x 10 x 10 patches exceed the LDS with three planes
  if (p.variant == 0 && mode == R_TR && tsw <= 8 && !f16) p.variant = 2;
  const int tn = (p.variant == 0) ? 256 : 128;
  const long wgs = ((p.variant == 3) ? (pt128 + 1) / 2 : pt128) * cdiv(Cout, tn) * ncls;
  p.ksplit = ring_ksplit(wgs, Cin / 16);
  return p;
}

size_t vg_internal_ring_workspace_bytes(int mode, int B, int Cin, int H, int W, int Cout, int planes) {
  const RingPlan p = ring_plan(mode, B, Cin, H, W, Cout, planes);
  if (p.ksplit <= 1) return 0;
  const size_t yh = (mode == R_FWD) ? (H - 1) / 2 + 1 : 2 * H, yw = (mode == R_FWD) ? (W - 1) / 2 + 1 : 2 * W;
  return (size_t)p.ksplit * B * Cout * yh * yw * sizeof(float);
}

// Floats of the output-statistics buffer ([slot][Cout][2]; one slot per (class, pixel tile, wavefront row)); 0 when
// the layer runs K-split (its partial outputs never meet in one workgroup: statistics come from a pass over y).
size_t vg_internal_ring_stats_floats(int mode, int B, int Cin, int H, int W, int Cout, int planes) {
  const RingPlan p = ring_plan(mode, B, Cin, H, W, Cout, planes);
  if (p.ksplit > 1) return 0;
  const int tsw = (mode == R_FWD) ? (W - 1) / 2 + 1 : W, tsh = (mode == R_FWD) ? (H - 1) / 2 + 1 : H;
  long tiles;
  if (p.variant == 3) tiles = (long)B * cdiv(tsh, 16) * cdiv(tsw, 16);
  else tiles = (tsw > 8) ? (long)B * cdiv(tsh, 8) * cdiv(tsw, 16) : (long)cdiv(B, 2) * cdiv(tsh, 8) * cdiv(tsw, 8);
  const int wp = (p.variant == 0 || p.variant == 2) ? 2 : 4;
  return (size_t)tiles * wp * ((mode == R_TR) ? 4 : 1) * Cout * 2;
}

int vg_internal_ring_conv(int mode, const float* x, const void* packed, const float* bias, float* y, int B, int Cin, int H,
                          int W, int Cout, int planes, void* workspace, size_t workspace_bytes, const float* in_scale,
                          const float* in_shift, int in_act, float* stats, size_t stats_floats, const float* in_amax,
                          hipStream_t st) {
  const RingPlan p = ring_plan(mode, B, Cin, H, W, Cout, planes);
  if (p.ksplit > 1 && (!workspace || workspace_bytes < vg_internal_ring_workspace_bytes(mode, B, Cin, H, W, Cout, planes)))
    return VG_ERR_WORKSPACE;
  if ((in_scale == nullptr) != (in_shift == nullptr) || in_act < VG_ACT_NONE || in_act > VG_ACT_LRELU) return VG_ERR_BAD_ARG;
  if (stats && (p.ksplit > 1 || stats_floats < vg_internal_ring_stats_floats(mode, B, Cin, H, W, Cout, planes))) return VG_ERR_BAD_ARG;
  const bool f16 = (planes & VG_PLANES_F16_FLAG) != 0;
  planes &= 0xff;
  if (f16 && (planes != 2 || !in_amax)) return VG_ERR_BAD_ARG;
  const RFuse fu = {in_scale, in_shift, in_act, stats, in_amax};
  const bf16x8* w = (const bf16x8*)packed;
  float* slabs = (float*)workspace;
  if (mode == R_FWD) {
    if (f16) return ring_dispatch<R_FWD, 2, true>(p.variant, x, w, bias, y, B, Cin, H, W, Cout, p.ksplit, slabs, fu, st);
    if (planes == 3) return ring_dispatch<R_FWD, 3, false>(p.variant, x, w, bias, y, B, Cin, H, W, Cout, p.ksplit, slabs, fu, st);
    return ring_dispatch<R_FWD, 2, false>(p.variant, x, w, bias, y, B, Cin, H, W, Cout, p.ksplit, slabs, fu, st);
  }
  if (f16) return ring_dispatch<R_TR, 2, true>(p.variant, x, w, bias, y, B, Cin, H, W, Cout, p.ksplit, slabs, fu, st);
  if (planes == 3) return ring_dispatch<R_TR, 3, false>(p.variant, x, w, bias, y, B, Cin, H, W, Cout, p.ksplit, slabs, fu, st);
  return ring_dispatch<R_TR, 2, false>(p.variant, x, w, bias, y, B, Cin, H, W, Cout, p.ksplit, slabs, fu, st);
}

#ifdef VG_TUNING
void vg_internal_ring_set_variant(int v) { g_ring_variant = v; }
#endif
