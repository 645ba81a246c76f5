// Staggered two-group implicit-GEMM kernel for the large C3D convolutions (gfx950).
//
// Same contraction, tables, swizzled LDS-DMA staging and LDS epilogue as igemm_kernel
// (igemm.hip.h), but a 256x128 block tile run by 8 waves = two groups of four (waves
// 0-3 and 4-7; waves w and w+4 share a SIMD).  Every wave alternates
//
//     LOAD(j):    issue its share of the LDS-DMA for K-tile j+2, read ALL of K-tile j's
//                 fragments LDS -> registers, counted vmcnt (K-tile j+1 landed), barrier
//     COMPUTE(j): 32 MFMAs from registers only, barrier
//
// and group B runs one barrier behind group A, so on every SIMD one wave is in its MFMA
// phase while its partner is in its load phase: the matrix pipe stays busy across the
// barrier / DMA-wait / LDS-latency that stall the simple one-barrier loop.  Three LDS
// stages (3 x 48 KiB) keep two K-tiles of DMA in flight; waits are counted (never
// vmcnt(0) in the steady state) and barriers are raw s_barrier, so in-flight DMA is not
// drained (cdna_hip_programming.md "Pipelining across barriers").
//
// Hazards (h = half-step; A loads K-tile j at h=2j, computes at 2j+1; B one later):
//  RAW  K-tile j is read from h=2j on.  Its DMA was issued in LOAD(j-2) (A: 2j-4, B: 2j-3)
//       and every wave passed `vmcnt(6)` for it in LOAD(j-1) before that phase's barrier
//       (A: end of 2j-2, B: end of 2j-1), i.e. before h=2j begins.
//  WAR  DMA for K-tile j+2 overwrites the stage of K-tile j-1, last read in LOAD(j-1)
//       (A: 2j-2, B: 2j-1, each followed by lgkmcnt(0) + barrier); issued at h >= 2j.
#pragma once
#include "igemm.hip.h"

namespace rgp {

struct StaggerSmem {
  static constexpr int BM = 256, BN = 128;
  static constexpr int STAGE_BYTES = (BM + BN) * 128;        // 48 KiB
  static constexpr int KOFF_OFF = 3 * STAGE_BYTES;           // 144 KiB
  static constexpr int KOFF_MAX = 256;                       // K-chunks (int each)
  static constexpr int ROWINFO_OFF = KOFF_OFF + KOFF_MAX * 4;
  static constexpr int ROWSET = BM * 24;                     // rowin + rowout (8 B) + rowimg + rowml (4 B) per row
  static constexpr int BYTES = ROWINFO_OFF + 2 * ROWSET;     // 157.0 KiB
};

template <typename T, int P, class Epi, int ABLATE = 0>
__global__ __launch_bounds__(512) void igemm_stagger_kernel(const IgemmParams p, const EpiParams e) {
  constexpr int BM = StaggerSmem::BM, BN = StaggerSmem::BN;
  constexpr int NW = 8, NT = 512;
  constexpr int WTM = 64, WTN = 64, MI = 4, NI = 4;
  constexpr int A_PER_WAVE = (BM / 8) / NW, B_PER_WAVE = (BN / 8) / NW;   // 4, 2
  constexpr int ESZ = sizeof(T);
  constexpr int STAGE = StaggerSmem::STAGE_BYTES;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_koff = (int*)(smem + StaggerSmem::KOFF_OFF);
  // row tables of the current tile and of the next one (two sets, swapped per tile)
  auto row_in = [&](int set) { return (long long*)(smem + StaggerSmem::ROWINFO_OFF + set * StaggerSmem::ROWSET); };
  auto row_out = [&](int set) { return row_in(set) + BM; };
  auto row_img = [&](int set) { return (int*)(row_in(set) + 2 * BM); };
  auto row_ml = [&](int set) { return row_img(set) + BM; };

  unsigned long long t_entry = 0;
  if (ABLATE & 32) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_entry)::"memory");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const bool group_b = wave >= 4;

  // Persistent: the grid is one block per CU and block b takes tiles b, b + gridDim.x, ...  gridDim.x is a
  // multiple of 8 (or the whole problem), so a block stays on "its" XCD's contiguous range of the tile order.
  const int n_nt = (p.N + BN - 1) / BN;
  const int n_mt = (p.M + BM - 1) / BM;
  const int nwg = n_mt * n_nt;
  auto tile_origin = [&](int t, int& m0, int& n0) {
    const int q = nwg >> 3, r = nwg & 7, x = t & 7, y = t >> 3;
    const int bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    // Column tiles: the blocks of an XCD that run side by side (R = gridDim.x / 8 of them) take R consecutive row
    // tiles of ONE column tile, then the next column tile of the same rows: they stream the same 128-column filter
    // panel in near-lockstep (one L2 fill serves all of them) instead of n_nt panels that evict each other
    // (conv4b: 14 MB of filter per round against a 4 MB L2).
    const int R = (p.ntile_group > 0) ? p.ntile_group : 1;
    const int g = bid / (R * n_nt), rr = bid - g * (R * n_nt);
    const int rows = min(R, n_mt - g * R);
    m0 = (g * R + rr % rows) * BM;
    n0 = (rr / rows) * BN;
  };
  // row r = tid of a tile: image, row in image, operand origin, output offset of its pooling window.  The
  // look-ups for the NEXT tile are issued at the start of the epilogue and committed to the other table set
  // at its end, so only the first tile of a block waits on them.
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

  const int lrow = lane >> 3;
  const int lchunk = (lane & 7) ^ lrow;
  const char* a_src[A_PER_WAVE];
  const char* b_src[B_PER_WAVE];

  auto stage = [&](int st, int kt) {
    char* abuf = smem + st * STAGE;
    char* bbuf = abuf + BM * 128;
    const long long ko = (long long)s_koff[kt] * ESZ;
    const long long kb = (long long)kt * 128;
#pragma unroll
    for (int j = 0; j < A_PER_WAVE; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[j] + ko),
                                       (__attribute__((address_space(3))) void*)(abuf + (wave * A_PER_WAVE + j) * 1024),
                                       16, 0, 0);
#pragma unroll
    for (int j = 0; j < B_PER_WAVE; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[j] + kb),
                                       (__attribute__((address_space(3))) void*)(bbuf + (wave * B_PER_WAVE + j) * 1024),
                                       16, 0, 0);
  };

  // one third of a K-tile's DMA (2 of this wave's 6 instructions): issued between MFMA groups
  auto stage_part = [&](int st, int kt, int part, long long ko) {
    char* abuf = smem + st * STAGE;
    char* bbuf = abuf + BM * 128;
    if (part < 2) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[part * 2 + j] + ko),
                                         (__attribute__((address_space(3))) void*)(abuf + (wave * A_PER_WAVE + part * 2 + j) * 1024),
                                         16, 0, 0);
    } else {
      const long long kb = (long long)kt * 128;
#pragma unroll
      for (int j = 0; j < B_PER_WAVE; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[j] + kb),
                                         (__attribute__((address_space(3))) void*)(bbuf + (wave * B_PER_WAVE + j) * 1024),
                                         16, 0, 0);
    }
  };

  const int frow = lane & 15, fk = lane >> 4;
  const int a_off = (wm * WTM + frow) * 128, b_off = BM * 128 + (wn * WTN + frow) * 128;
  const int pc0 = ((0 * 4 + fk) ^ (frow & 7)) * 16, pc1 = ((1 * 4 + fk) ^ (frow & 7)) * 16;

  // diagnostic build only (ABLATE & 32): per-wave s_memtime stamps around the phases; sums go to a
  // debug buffer (EpiParams::c_save, unused by the C3D epilogue) that nothing else reads.
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0, t_start = 0;
  auto stamp = [&](int k) {
    if (ABLATE & 32) {
      unsigned long long t;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      if (k >= 0) seg[k] += t - t_prev;
      else t_start = t;
      t_prev = t;
    }
  };

#pragma clang loop unroll(disable)
  while (true) {
  {
    const long long* s_rowin = row_in(set);
#pragma unroll
    for (int j = 0; j < A_PER_WAVE; ++j) {
      const int r = (wave * A_PER_WAVE + j) * 8 + lrow;
      a_src[j] = (const char*)p.A + s_rowin[r] * ESZ + lchunk * 16;
    }
#pragma unroll
    for (int j = 0; j < B_PER_WAVE; ++j) {
      const int r = (wave * B_PER_WAVE + j) * 8 + lrow;
      b_src[j] = (const char*)p.W + ((long long)(n0 + r) * p.K) * ESZ + lchunk * 16;
    }
  }
  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- prologue: K-tiles 0 and 1 in flight, tile 0 landed for everybody ----
  stage(0, 0);
  if (p.nk > 1) {
    stage(1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (group_b) __builtin_amdgcn_s_barrier();      // run one half-step behind group A

  stamp(-1);
  if (ABLATE & 32) seg[5] += t_start - t_entry;     // dev: prologue (replaces the bar2 column)
  unsigned long long r_start = 0;                   // dev: 100 MHz wall clock around the K loop -> in-kernel shader clock
  if (ABLATE & 32) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r_start)::"memory");
  int st = 0;       // stage of K-tile j
#pragma clang loop unroll(disable)
  for (int j = 0; j < p.nk; ++j) {
    // ---------------- LOAD(j) ----------------
    // Order inside the phases (measured with the s_memtime stamps below): fragment reads are issued
    // first so their LDS latency is covered by the DMA issue, and the B-tile DMA (2 of this wave's 6
    // instructions) rides in COMPUTE(j) so LOAD and COMPUTE are closer in length.  ABLATE & 128 is the
    // earlier order (all 6 DMA, then reads) kept for A/B runs.
    constexpr bool READS_FIRST = !(ABLATE & 128);
    constexpr bool B_IN_COMPUTE = !(ABLATE & 128);
    // INTERLEAVE (ABLATE & 256): the fragment reads and the A-tile DMA go to different units (LDS array / texture
    // addresser); issued as two blocks they serialise at ISSUE -- a wave is in-order, the 16 reads back up behind the
    // LDS queue (~295 cycles for the 4 loading waves' 64 KB), and only then do the 4 DMA instructions start queueing
    // behind the addresser (~300) -- so they are issued 4 reads : 1 DMA and the two queues drain side by side.
    constexpr bool INTERLEAVE = (ABLATE & 256) != 0;
    const bool more = j + 2 < p.nk;
    int st2 = st + 2;
    if (st2 >= 3) st2 -= 3;
    long long ko2 = 0;
    if (more) ko2 = (long long)s_koff[j + 2] * ESZ;
    if (!READS_FIRST && more && !(ABLATE & 2)) stage(st2, j + 2);
    stamp(0);
    const char* sb = smem + ((ABLATE & 4) ? 0 : st) * STAGE;
    f32x4 af[2][MI], bf[2][NI];
    if constexpr (INTERLEAVE) {
      char* abuf2 = smem + st2 * STAGE;
      auto dma_a = [&](int jj) {
        if (more && !(ABLATE & 2))
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[jj] + ko2),
                                           (__attribute__((address_space(3))) void*)(abuf2 + (wave * A_PER_WAVE + jj) * 1024),
                                           16, 0, 0);
      };
#pragma unroll
      for (int i = 0; i < MI; ++i) af[0][i] = *(const f32x4*)(sb + a_off + i * 16 * 128 + pc0);
      __builtin_amdgcn_sched_barrier(0);
      dma_a(0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NI; ++i) bf[0][i] = *(const f32x4*)(sb + b_off + i * 16 * 128 + pc0);
      __builtin_amdgcn_sched_barrier(0);
      dma_a(1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < MI; ++i) af[1][i] = *(const f32x4*)(sb + a_off + i * 16 * 128 + pc1);
      __builtin_amdgcn_sched_barrier(0);
      dma_a(2);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NI; ++i) bf[1][i] = *(const f32x4*)(sb + b_off + i * 16 * 128 + pc1);
      __builtin_amdgcn_sched_barrier(0);
      dma_a(3);
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      af[0][i] = *(const f32x4*)(sb + a_off + i * 16 * 128 + pc0);
      af[1][i] = *(const f32x4*)(sb + a_off + i * 16 * 128 + pc1);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      bf[0][i] = *(const f32x4*)(sb + b_off + i * 16 * 128 + pc0);
      bf[1][i] = *(const f32x4*)(sb + b_off + i * 16 * 128 + pc1);
    }
    stamp(1);
    if (READS_FIRST && more && !(ABLATE & 2)) {
      __builtin_amdgcn_sched_barrier(0);
      stage_part(st2, j + 2, 0, ko2);
      stage_part(st2, j + 2, 1, ko2);
      if (!B_IN_COMPUTE) stage_part(st2, j + 2, 2, ko2);
      __builtin_amdgcn_sched_barrier(0);
    }
    }
    // everything older than this phase's own DMA (i.e. all of K-tile j+1) must have landed
    if (more && !(ABLATE & 2)) {
      if (B_IN_COMPUTE) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    stamp(2);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    stamp(2);                                      // (barrier wait folded into the "wait" column; slot 3 holds the wall clock)
    // ---------------- COMPUTE(j) ----------------
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int jn = 0; jn < NI; ++jn) {
          if (ABLATE & 1) { asm volatile("" ::"v"(af[s][i]), "v"(bf[s][jn])); acc[i][jn][0] += 1.f; }
          else Mma<T>::step(acc[i][jn], af[s][i], bf[s][jn]);
        }
        if (B_IN_COMPUTE && more && !(ABLATE & 2) && s * MI + i == 3) {
          __builtin_amdgcn_sched_barrier(0);
          stage_part(st2, j + 2, 2, ko2);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    __builtin_amdgcn_s_setprio(0);
    stamp(4);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    stamp(4);
    st = st + 1 == 3 ? 0 : st + 1;
  }
  if (ABLATE & 32) {
    seg[6] += t_prev - t_start;
    unsigned long long r_end;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r_end)::"memory");
    seg[3] += r_end - r_start;
  }
  if (!group_b) __builtin_amdgcn_s_barrier();
  __syncthreads();

  // ---- epilogue: the whole 256 x BN fp32 tile goes through LDS in one pass (the three stages are free now),
  // one barrier, then every thread finishes its (BM/P * BN/8) / 512 items back to back.  Output offsets were
  // resolved in the prologue (s_rowout) and the 8 bias values of a thread's column group are fetched while the
  // accumulators are being written, so no item waits on a global load.  (The former slab-by-slab loop paid a
  // table lookup + bias fetch + two barriers per 64 rows: 12k cycles per tile pooled, 40k unpooled -- a quarter
  // of conv2a and 40 % of conv3a on top of their K loops.)
  constexpr int LDS_LD = BN + 4;
  static_assert(BM * LDS_LD * 4 <= 3 * STAGE, "fp32 tile must fit in the stage ring");
  float* stg = (float*)smem;
  constexpr int CG = BN / 8;
  constexpr int ITEMS = (BM / P) * CG;
  constexpr int NIT = (ITEMS + NT - 1) / NT;
  static_assert(NT % CG == 0, "a thread keeps its column group");
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
    for (int i = 0; i < 8; ++i) bias8[i] = (n0 + cg * 8 + i < p.N) ? e.bias[n0 + cg * 8 + i] : 0.f;
  }
  // 2x2x2 pooling without arg-max: a lane holds 4 of a window's 8 rows (registers r = 0..3) and the lane 16 away
  // the other 4, so the window maximum is one v_max3 + v_max in the lane and one ds_swizzle (lane ^ 16) -- the staged
  // tile is the POOLED one, 32 x BN instead of 256 x BN floats: an eighth of the LDS traffic of the general path.
  bool pooled_in_regs = false;
  if constexpr (P == 8) pooled_in_regs = e.argmax == nullptr && (p.pool_regs & 1);
  if (pooled_in_regs) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int jn = 0; jn < NI; ++jn) {
        const f32x4 c = acc[i][jn];
        const float x = fmaxf(fmaxf(c[0], c[1]), fmaxf(c[2], c[3]));
        const float y = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), 0x401F));   // lane ^ 16
        if ((fk & 1) == 0) stg[(wm * (WTM / 8) + i * 2 + (fk >> 1)) * LDS_LD + wn * WTN + jn * 16 + frow] = fmaxf(x, y);
      }
  } else {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int jn = 0; jn < NI; ++jn)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          stg[(wm * WTM + i * 16 + fk * 4 + r) * LDS_LD + wn * WTN + jn * 16 + frow] = acc[i][jn][r];
  }
  __syncthreads();
  // Every load of this epilogue (the look-ups, bias8) is made known-complete HERE, on every path and before any store
  // is issued: (i) a load consumed only under a condition (bias8 under `img >= 0`) otherwise stays "maybe pending" in
  // the compiler's register scoreboard, and the K loop then gets an `s_waitcnt vmcnt(0)` in front of whichever
  // instruction re-uses one of those registers -- it was the fragment reads of the P = 1 kernels (conv3a / conv4a /
  // conv5a / conv5b): the LDS-DMA ring was drained every K-tile (scripts/check_isa_waits.py lints this); (ii) placed
  // after the stores the same wait would sit out their completion latency.  The builtin is an s_waitcnt the compiler
  // accounts for.
  __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0)
  if (next_tile < nwg && tid < BM) row_commit(set ^ 1, nxt);
  if (pooled_in_regs) {
    {
      const int rt = (tid / CG) * 8;                 // 512 threads = 32 pooled rows x 16 column groups
      const int img = s_rowimg[rt];
      if (img >= 0) {
        float v[8];
        const float* src = stg + (tid / CG) * LDS_LD + cg * 8;
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
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int it = tid + k * NT;
    if (ITEMS % NT != 0 && it >= ITEMS) break;
    const int rt = (it / CG) * P;
    const int img = s_rowimg[rt];
    if (img >= 0) {
      float v[8];
      const float* src = stg + rt * LDS_LD + cg * 8;
      const int mlp = s_rowml[rt] / P;
      if (P > 1 && e.argmax && n0 + cg * 8 < p.N) pool_window_argmax<P>(src, LDS_LD, v, e.argmax + ((long long)img * (p.Mw / P) + mlp) * p.N + n0 + cg * 8);
      else pool_window<P>(src, LDS_LD, v);
      if constexpr (Bias::value) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += bias8[i];
      }
      Bias::NoBias::apply_at(e, p.N, img, mlp, s_rowout[rt], n0 + cg * 8, v);
    }
  }
  }
  stamp(7);                                        // seg[7] = everything after the K loop (epilogue)
  if (next_tile >= nwg) break;
  __syncthreads();               // the fp32 tile has been read (the stage ring is free) and the next tables are in place
  if (ABLATE & 32) t_entry = t_prev;
  set ^= 1;
  tile = next_tile;
  m0 = m0n;
  n0 = n0n;
  }
  if ((ABLATE & 32) && e.c_save) {
    if (lane == 0) {
      unsigned long long* dbg = (unsigned long long*)e.c_save + ((size_t)blockIdx.x * 8 + wave) * 8;
      for (int k = 0; k < 8; ++k) dbg[k] = seg[k];
    }
  }
}

}  // namespace rgp
