// 256 x 256 block-tile implicit-GEMM kernel for the C3D convolutions with N >= 256 output channels (gfx950, bf16).
//
// Why this tile: the 256x128 staggered kernel (igemm_stagger.hip.h) moves 48 KB of operand tiles L2 -> LDS per
// 64-deep K-tile and runs 14...16 TB/s of LDS-DMA ingest chip-wide on conv3b / conv4b -- 77...90 % of the
// 66-73 GB/s per CU that MI355X_MICROARCH.md measures for rows gathered into LDS from L2.  It is INGEST-bound: the
// matrix pipe idles 41 % of the time waiting for bytes.  Bytes per FLOP are (BM + BN) / (BM * BN): a 256 x 256 tile
// needs two thirds of the 256 x 128 tile's, and 25 % fewer LDS fragment-read bytes per MFMA (wave tile 128 x 64: 12
// fragments feed 32 MFMAs instead of 16).
//
// Structure: the staggered two-group schedule of igemm_stagger.hip.h at a K depth of 32 elements per step
// ("sub-tile": 64-byte operand rows; a 64-deep K-tile of a 256 x 256 block would need 3 x 64 KB of ring):
//
//     LOAD(s):    A-part of the LDS-DMA of sub-tile s+3, all 12 fragment reads of sub-tile s, counted vmcnt, barrier
//     COMPUTE(s): 32 MFMAs from registers, B-part of the DMA of sub-tile s+3 between them, barrier
//
// 8 waves = 2 (M) x 4 (N); waves 0-3 and 4-7 (partners on a SIMD) run one barrier apart.  Four LDS slots of 32 KB keep
// three sub-tiles of DMA in flight; waits are counted, barriers raw.
//
// LDS image of a sub-tile: [256 A rows | 256 B rows] x 64 B.  One LDS-DMA instruction fills 16 rows (1 KiB) = exactly
// one MFMA fragment block; the 16-B chunk c of row r sits at physical chunk c ^ ((-(r >> 2)) & 3), which makes every
// 16-lane group of the ds_read_b128 fragment read touch all 64 banks once (groups {0-3,12-15,20-27}, ...: rows with
// equal r & 3 must land in four different chunks).  The permutation is applied on the DMA's per-lane SOURCE address
// and again on the read (LDS-DMA writes lane-linear).
//
// Hazards (h = half-step; group A loads sub-tile s at h = 2s and computes at 2s+1; group B one later):
//  RAW  sub-tile s is read from h = 2s on.  Its DMA was issued in LOAD(s-3) / COMPUTE(s-3) (A: 2s-6, 2s-5; B: 2s-5,
//       2s-4) and every wave passed `vmcnt(6)` for it in LOAD(s-1) before that phase's barrier (A: end of 2s-2,
//       B: end of 2s-1).
//  WAR  the DMA of sub-tile s+3 overwrites the slot of sub-tile s-1, last read in LOAD(s-1) (A: 2s-2, B: 2s-1, each
//       followed by lgkmcnt(0) + barrier); it is issued at h >= 2s.
#pragma once
#include <type_traits>

#include "igemm.hip.h"

namespace rgp {

// Two block shapes share the code, both 8 waves of 128 x 64:
//   256 x 256 (waves 2 x 4), 4 slots of 32 KB, DMA 3 sub-tiles ahead   -- N % 256 == 0 (conv3a ... conv4b)
//   512 x 128 (waves 4 x 2), 3 slots of 40 KB, DMA 2 sub-tiles ahead   -- N == 128 (conv2a): the A tile is all that can
//       grow there; 5/6 of the 256 x 128 tile's bytes per FLOP and half as many tile prologues / epilogues (conv2a's K
//       loop is only 27 K-tiles long).
template <int BM_, int BN_> struct WideSmem {
  static constexpr int BM = BM_, BN = BN_;
  static constexpr int SLOT_BYTES = (BM + BN) * 64;          // per 32-deep sub-tile
  static constexpr int NSLOT = BM == 256 ? 4 : 3;
  static constexpr int KOFF_OFF = NSLOT * SLOT_BYTES;        // 128 KiB / 120 KiB
  static constexpr int KOFF_MAX = 256;                       // 128-byte K-chunks (int each)
  static constexpr int ROWINFO_OFF = KOFF_OFF + KOFF_MAX * 4;
  static constexpr int ROWSET = BM * 24;                     // rowin + rowout (8 B) + rowimg + rowml (4 B) per row
  static constexpr int BYTES = ROWINFO_OFF + 2 * ROWSET;     // 141 KiB / 145 KiB
  static_assert(BYTES <= 160 * 1024, "LDS budget");
};

template <int BM, int BN, int P, class Epi, int VAR = 0>
__global__ __launch_bounds__(512) void igemm_wide_kernel(const IgemmParams p, const EpiParams e) {
  constexpr bool B_IN_LOAD = (VAR & 1) != 0;                  // dev: the B-part of the DMA rides in LOAD too
  constexpr bool A_SPLIT = (VAR & 2) != 0;                    // dev: half of the A-part rides in COMPUTE (512-row tile: 4 + 1 -> 2 + 3)
  using T = bf16_t;
  using Smem = WideSmem<BM, BN>;
  constexpr int NT = 512;
  constexpr int WTM = 128, WTN = 64, MI = 8, NI = 4;
  constexpr int WN = BN / WTN;                               // waves along N: 4 or 2
  constexpr int SLOT = Smem::SLOT_BYTES;
  constexpr int NSLOT = Smem::NSLOT, AHEAD = NSLOT - 1;
  constexpr int A_BYTES = BM * 64;
  constexpr int A_PER = BM / 16 / 8, B_PER = BN / 16 / 8;    // 1-KiB DMA blocks per wave: 2 + 2 or 4 + 1
  // counted waits: in LOAD(s), after the A-part of sub-tile s+AHEAD, everything up to sub-tile s+1 must have landed
  constexpr int A_LOAD = A_SPLIT ? A_PER / 2 : A_PER;         // A-part DMA instructions issued in LOAD
  constexpr int VM_LOOP = (AHEAD - 2) * (A_PER + B_PER) + (B_IN_LOAD ? A_PER + B_PER : A_LOAD);
  constexpr int VM_PRO = (AHEAD - 1) * (A_PER + B_PER);
  static_assert(BM % 128 == 0 && BN % 64 == 0 && (BM / WTM) * WN == 8, "8 waves of 128 x 64");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_koff = (int*)(smem + Smem::KOFF_OFF);
  auto row_in = [&](int set) { return (long long*)(smem + Smem::ROWINFO_OFF + set * Smem::ROWSET); };
  auto row_out = [&](int set) { return row_in(set) + BM; };
  auto row_img = [&](int set) { return (int*)(row_in(set) + 2 * BM); };
  auto row_ml = [&](int set) { return row_img(set) + BM; };

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const bool group_b = wave >= 4;

  // persistent tile walk, XCD-contiguous (igemm_stagger.hip.h)
  const int n_nt = p.N / BN;
  const int n_mt = (p.M + BM - 1) / BM;
  const int nwg = n_mt * n_nt;
  auto tile_origin = [&](int t, int& m0, int& n0) {
    const int q = nwg >> 3, r = nwg & 7, x = t & 7, y = t >> 3;
    const int bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    m0 = (bid / n_nt) * BM;
    n0 = (bid % n_nt) * BN;
  };
  struct RowRegs { int img, ml, in_off, out_off; bool valid; };
  auto row_lookup = [&](int m0) {
    RowRegs q;
    int m = m0 + tid;
    q.valid = m < p.M;
    if (!q.valid) m = p.M - 1;
    q.img = m / p.Mw;
    q.ml = m - q.img * p.Mw;
    q.in_off = p.in_tab[q.ml];
    q.out_off = (q.valid && tid % P == 0) ? e.out_tab[q.ml / P] : 0;
    return q;
  };
  auto row_commit = [&](int set, const RowRegs& q) {
    row_in(set)[tid] = (long long)q.img * p.in_img_stride + q.in_off;
    row_img(set)[tid] = q.valid ? q.img : -1;
    row_ml(set)[tid] = q.ml;
    row_out(set)[tid] = (q.valid && tid % P == 0) ? (long long)q.img * e.out_img_stride + e.out_extra + q.out_off : 0;
  };

  int tile = blockIdx.x;
  if (tile >= nwg) return;
  int m0, n0;
  tile_origin(tile, m0, n0);
  if (tid < BM) row_commit(0, row_lookup(m0));
  for (int i = tid; i < p.nk; i += NT) s_koff[i] = p.koff[i];
  __syncthreads();
  int set = 0;

  // DMA lane geometry: an instruction fills 16 rows x 64 B; lane l -> row l >> 2, physical chunk l & 3
  const int drow = lane >> 2;
  const int dchunk = (lane & 3) ^ ((-(drow >> 2)) & 3);      // logical chunk this lane fetches
  // fragment-read lane geometry: row lane & 15, logical chunk lane >> 4
  const int frow = lane & 15, fk = lane >> 4;
  const int pc = frow * 64 + ((fk ^ ((-(frow >> 2)) & 3)) << 4);
  const int a_off = (wm * MI) * 1024 + pc;
  const int b_off = A_BYTES + (wn * NI) * 1024 + pc;
  const int nsub = p.nk * 2;

  const char* a_src[A_PER];
  const char* b_src[B_PER];
  // A-part / B-part of one sub-tile's DMA (A_PER + B_PER of this wave's instructions)
  auto dma_a = [&](int slot, int s, long long ko, int j0, int j1) {
    char* abuf = smem + slot * SLOT;
#pragma unroll
    for (int j = j0; j < j1; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[j] + ko),
                                       (__attribute__((address_space(3))) void*)(abuf + (wave * A_PER + j) * 1024), 16, 0, 0);
  };
  auto dma_b = [&](int slot, int s) {
    char* bbuf = smem + slot * SLOT + A_BYTES;
    const long long kb = (long long)s * 64;
#pragma unroll
    for (int j = 0; j < B_PER; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[j] + kb),
                                       (__attribute__((address_space(3))) void*)(bbuf + (wave * B_PER + j) * 1024), 16, 0, 0);
  };

#pragma clang loop unroll(disable)
  while (true) {
    {
      const long long* s_rowin = row_in(set);
#pragma unroll
      for (int j = 0; j < A_PER; ++j) a_src[j] = (const char*)p.A + s_rowin[(wave * A_PER + j) * 16 + drow] * 2 + dchunk * 16;
#pragma unroll
      for (int j = 0; j < B_PER; ++j)
        b_src[j] = (const char*)p.W + ((long long)(n0 + (wave * B_PER + j) * 16 + drow) * p.K) * 2 + dchunk * 16;
    }
    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- prologue: sub-tiles 0 .. AHEAD-1 in flight, sub-tile 0 landed for everybody (nsub >= 4: host-checked) ----
    auto koff_bytes = [&](int s) { return (long long)s_koff[s >> 1] * 2 + (s & 1) * 64; };
#pragma unroll
    for (int s0 = 0; s0 < AHEAD; ++s0) { dma_a(s0, s0, koff_bytes(s0), 0, A_PER); dma_b(s0, s0); }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_PRO) : "memory");
    __builtin_amdgcn_s_barrier();
    if (group_b) __builtin_amdgcn_s_barrier();      // run one half-step behind group A

    int slot = 0;
#pragma clang loop unroll(disable)
    for (int s = 0; s < nsub; ++s) {
      // ---------------- LOAD(s) ----------------
      const bool more = s + AHEAD < nsub;
      int slot3 = slot + AHEAD;
      if (slot3 >= NSLOT) slot3 -= NSLOT;
      const char* sb = smem + slot * SLOT;
      long long ko3 = 0;
      if (more) ko3 = koff_bytes(s + AHEAD);           // read ahead of the fragment reads: its wait covers only itself
      f32x4 af[MI], bf[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *(const f32x4*)(sb + a_off + i * 1024);
#pragma unroll
      for (int i = 0; i < NI; ++i) bf[i] = *(const f32x4*)(sb + b_off + i * 1024);
      if (more) {
        __builtin_amdgcn_sched_barrier(0);
        dma_a(slot3, s + AHEAD, ko3, 0, A_LOAD);
        if constexpr (B_IN_LOAD) dma_b(slot3, s + AHEAD);
        __builtin_amdgcn_sched_barrier(0);
        // all of sub-tile s+1 landed: outstanding may be the parts of s+2 .. s+AHEAD issued so far
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(VM_LOOP) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- COMPUTE(s) ----------------
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int jn = 0; jn < NI; ++jn) Mma<T>::step(acc[i][jn], af[i], bf[jn]);
        if (!B_IN_LOAD && more && i == 1) {
          __builtin_amdgcn_sched_barrier(0);
          dma_b(slot3, s + AHEAD);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (A_SPLIT && more && i == 4) {
          __builtin_amdgcn_sched_barrier(0);
          dma_a(slot3, s + AHEAD, ko3, A_LOAD, A_PER);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      slot = slot + 1 == NSLOT ? 0 : slot + 1;
    }
    if (!group_b) __builtin_amdgcn_s_barrier();
    __syncthreads();

    // ---- epilogue through LDS (the ring is free): 16-B stores of 8 consecutive channels per thread ----
    constexpr int LDS_LD = BN + 4;
    float* stg = (float*)smem;
    constexpr int CG = BN / 8;                       // 32 column groups; NT % CG == 0: a thread keeps its group
    using Bias = EpiBiasSplit<Epi>;
    const int cg = tid % CG;
    const int next_tile = tile + gridDim.x;
    int m0n = 0, n0n = 0;
    RowRegs nxt = {0, 0, 0, 0, false};
    if (next_tile < nwg) {
      tile_origin(next_tile, m0n, n0n);
      if (tid < BM) nxt = row_lookup(m0n);
    }
    const int* s_rowimg = row_img(set);
    const int* s_rowml = row_ml(set);
    const long long* s_rowout = row_out(set);
    float bias8[8];
    if constexpr (Bias::value) {
#pragma unroll
      for (int i = 0; i < 8; ++i) bias8[i] = e.bias[n0 + cg * 8 + i];
    }
    bool pooled_in_regs = false;
    if constexpr (P == 8) pooled_in_regs = e.argmax == nullptr;
    if (pooled_in_regs) {
      // window maximum in registers (a lane holds 4 of a window's 8 rows, lane ^ 16 the other 4): the staged tile is
      // the pooled one, 32 x 256 floats
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jn = 0; jn < NI; ++jn) {
          const f32x4 c = acc[i][jn];
          const float x = fmaxf(fmaxf(c[0], c[1]), fmaxf(c[2], c[3]));
          const float y = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), 0x401F));   // lane ^ 16
          if ((fk & 1) == 0) stg[(wm * (WTM / 8) + i * 2 + (fk >> 1)) * LDS_LD + wn * WTN + jn * 16 + frow] = fmaxf(x, y);
        }
      __syncthreads();
      __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): look-ups + bias complete on every path (igemm_stagger.hip.h)
      if (next_tile < nwg && tid < BM) row_commit(set ^ 1, nxt);
      static_assert(((BM / 8) * CG) % NT == 0, "whole items per thread");
#pragma unroll
      for (int k = 0; k < (BM / 8) * CG / NT; ++k) {
        const int prow = (tid + k * NT) / CG;          // pooled row
        const int rt = prow * 8;
        const int img = s_rowimg[rt];
        if (img >= 0) {
          float v[8];
          const float* src = stg + prow * LDS_LD + cg * 8;
          const f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) { v[i] = v0[i]; v[4 + i] = v1[i]; }
          if constexpr (Bias::value) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += bias8[i];
          }
          Bias::NoBias::apply_at(e, p.N, img, s_rowml[rt] / P, s_rowout[rt], n0 + cg * 8, v);
        }
      }
    } else {
      // the fp32 tile is 256 KB: it goes through the ring in passes of PROWS rows (64 at BN = 256, 128 at BN = 128);
      // pass q holds fragments (q % PPW) * FPP .. +FPP-1 of the waves with wm == q / PPW
      constexpr int PROWS = BN == 256 ? 64 : 128;
      constexpr int PPW = WTM / PROWS, FPP = PROWS / 16;      // passes per wave row, fragments per pass
      constexpr int NPASS = BM / PROWS;
      static_assert(PROWS * LDS_LD * 4 <= NSLOT * SLOT, "a pass fits in the ring");
      constexpr int ITEMS = (PROWS / P) * CG;
      constexpr int NIT = (ITEMS + NT - 1) / NT;
      auto pass = [&](auto qc) {
        constexpr int q = decltype(qc)::value;
        if (wm == q / PPW) {
#pragma unroll
          for (int i4 = 0; i4 < FPP; ++i4)
#pragma unroll
            for (int jn = 0; jn < NI; ++jn)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                stg[(i4 * 16 + fk * 4 + r) * LDS_LD + wn * WTN + jn * 16 + frow] = acc[(q % PPW) * FPP + i4][jn][r];
        }
        __syncthreads();
        if constexpr (q == 0) {
          __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0), see above
          if (next_tile < nwg && tid < BM) row_commit(set ^ 1, nxt);
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
          const int it = tid + k * NT;
          if (ITEMS % NT != 0 && it >= ITEMS) break;
          const int rl = (it / CG) * P;                // row inside the pass
          const int rt = q * PROWS + rl;
          const int img = s_rowimg[rt];
          if (img >= 0) {
            float v[8];
            const float* src = stg + rl * LDS_LD + cg * 8;
            const int mlp = s_rowml[rt] / P;
            if (P > 1 && e.argmax) pool_window_argmax<P>(src, LDS_LD, v, e.argmax + ((long long)img * (p.Mw / P) + mlp) * p.N + n0 + cg * 8);
            else pool_window<P>(src, LDS_LD, v);
            if constexpr (Bias::value) {
#pragma unroll
              for (int i = 0; i < 8; ++i) v[i] += bias8[i];
            }
            Bias::NoBias::apply_at(e, p.N, img, mlp, s_rowout[rt], n0 + cg * 8, v);
          }
        }
        if constexpr (q < NPASS - 1) __syncthreads();   // pass read: the staging area may be overwritten
      };
      static_assert(NPASS == 4, "four passes at both shapes");
      pass(std::integral_constant<int, 0>{});
      pass(std::integral_constant<int, 1>{});
      pass(std::integral_constant<int, 2>{});
      pass(std::integral_constant<int, 3>{});
    }
    if (next_tile >= nwg) break;
    __syncthreads();               // staging read and the next tables in place
    set ^= 1;
    tile = next_tile;
    m0 = m0n;
    n0 = n0n;
  }
}

}  // namespace rgp
