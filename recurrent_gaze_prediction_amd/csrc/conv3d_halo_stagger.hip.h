// 3x3x3 convolution with an LDS-resident input halo AND the staggered two-group schedule (gfx950):
// conv3d_halo_kernel's data movement (a block owns a BOX of output voxels, its input halo is loaded into LDS
// once per 64-channel chunk and serves all 27 taps; only the filter K-tiles stream) under
// igemm_stagger_kernel's schedule (8 waves = two groups of four, waves w and w+4 share a SIMD; every wave
// alternates LOAD: fragments LDS -> registers, and COMPUTE: 32 MFMAs from registers; group B runs one barrier
// behind group A, so each SIMD always has one wave in its MFMA phase).
//
// Why both: the im2col stagger kernel is LOAD-bound -- its LOAD phase (6 LDS-DMA per wave at the TA's 64 B/clk
// + 16 fragment reads + waits) is ~1000 cycles against ~600 of COMPUTE, so the matrix pipe idles half the time.
// With the halo resident, LOAD has no DMA at all (the per-tap filter tile, 2-4 DMA per wave, is issued from
// inside COMPUTE), which brings the two phases to the same length.
//
//   LDS    : [filter ring, NST stages of BN x 128 B][halo, HP8 x 128 B][halo gather offsets]
//   ring   : COMPUTE(j) issues the DMA of filter tile j+NST-1; a wave's LOAD(j) ends with a counted vmcnt that
//            covers its share of tile j+1, then the barrier publishes it.
//   chunk  : at a channel-chunk switch the pipeline is drained (group A idles one barrier so both groups are
//            aligned), every wave reloads its share of the halo, and group B idles one barrier to re-stagger.
//   hazards (h = half-step; group A runs LOAD(j) at h=2j and COMPUTE(j) at 2j+1, group B one later):
//     RAW  filter tile t is first read by group A in LOAD(t) (h=2t), so every wave's share must be published by the
//          barrier ending h=2t-1: each wave waits for its share of tile j+1 at the end of ITS LOAD(j) (A: h=2j,
//          B: h=2j+1).  The share was issued in COMPUTE(j+1-LOOKAHEAD), 2*LOOKAHEAD-3 half-steps earlier.
//     WAR  tile j+LOOKAHEAD overwrites the stage of tile j-1, last read in group B's LOAD(j-1) (h=2j-1); it is
//          issued in COMPUTE(j) (h=2j+1 / 2j+2).
#pragma once
#include "conv3d_halo.hip.h"

namespace rgp {

template <typename T, int BM, int BN, int WM, int WN, int NST, int P, class Epi>
__global__ __launch_bounds__(512) void conv3d_halo_stagger_kernel(const HaloParams p, EpiParams e) {
  constexpr int NW = WM * WN, NT = NW * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16;
  constexpr int B_PER_WAVE = (BN / 8) / NW;
  constexpr int BKE = Elem<T>::BKE, ESZ = sizeof(T);
  constexpr int B_STAGE = BN * 128;
  constexpr int LOOKAHEAD = NST - 1;                    // COMPUTE(j) issues tile j + LOOKAHEAD
  constexpr int WAIT_KEEP = (NST - 3) * B_PER_WAVE;     // DMA instructions that may stay in flight past LOAD(j)
  static_assert(NW == 8 && MI == 4 && NI == 4 && WTM % P == 0 && (BN / 8) % NW == 0 && NST >= 3, "tile");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int halo_bytes = p.HP8 * 128;
  char* halo0 = smem + NST * B_STAGE;
  (void)halo_bytes;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Waves w and w+4 share a SIMD and are in different groups (A: waves 0-3, B: 4-7); a group owns the upper /
  // lower half of the block's rows.
  const bool group_b = wave >= 4;
  const int wrow = WM == 4 ? wave >> 1 : wave >> 2;
  const int wcol = WM == 4 ? wave & 1 : wave & 3;

  const int n_nt = p.N / BN;
  const int boxes = p.nbx * p.nby * p.nbz;
  const int nwg = p.n_img * boxes * n_nt;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, y = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
  }
  const int nt = bid % n_nt;
  int bb = bid / n_nt;
  const int img = bb / boxes;
  bb -= img * boxes;
  const int bx = bb % p.nbx, by = (bb / p.nbx) % p.nby, bz = bb / (p.nbx * p.nby);
  const int n0 = nt * BN;
  const long long in_origin = (long long)img * p.in_img_stride + (long long)bz * p.box_in_z + (long long)by * p.box_in_y +
                              (long long)bx * p.box_in_x;
  e.out_extra = (long long)bz * p.box_out_z + (long long)by * p.box_out_y + (long long)bx * p.box_out_x;

  const int lrow = lane >> 3;
  const int lchunk = (lane & 7) ^ lrow;
  const char* a_base = (const char*)p.A + in_origin * ESZ + lchunk * 16;
  const char* b_src[B_PER_WAVE];
#pragma unroll
  for (int j = 0; j < B_PER_WAVE; ++j) {
    const int r = (wave * B_PER_WAVE + j) * 8 + lrow;
    b_src[j] = (const char*)p.W + ((long long)(n0 + r) * p.K) * ESZ + lchunk * 16;
  }
  // halo gather offsets of this wave's DMA instructions live in registers: read from an LDS table, every DMA of the
  // halo load would wait for the previous one (the compiler orders an LDS read after all outstanding LDS-DMA)
  constexpr int HQ = 10;                                 // up to 80 halo DMA instructions (640 voxels) per chunk
  const int n_hinst = p.HP8 >> 3;
  int go[HQ];
#pragma unroll
  for (int q = 0; q < HQ; ++q) go[q] = (wave + q * NW < n_hinst) ? p.halo_goff[(wave + q * NW) * 8 + lrow] : 0;
  auto halo_load = [&](int cc) {
#pragma unroll
    for (int q = 0; q < HQ; ++q)
      if (wave + q * NW < n_hinst)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(a_base + ((long long)go[q] + (long long)cc * BKE) * ESZ),
            (__attribute__((address_space(3))) void*)(halo0 + (wave + q * NW) * 1024), 16, 0, 0);
  };
  auto stage_b = [&](int kt) {
    const int tap = kt % 27, cc = kt / 27;
    const long long kb = ((long long)tap * p.Cin + (long long)cc * BKE) * ESZ;
    char* dst = smem + (kt % NST) * B_STAGE;
#pragma unroll
    for (int j = 0; j < B_PER_WAVE; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[j] + kb),
                                       (__attribute__((address_space(3))) void*)(dst + (wave * B_PER_WAVE + j) * 1024), 16, 0, 0);
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fk = lane >> 4;
  int hp[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) hp[i] = p.row_hp[wrow * WTM + i * 16 + frow];
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned halo_lds = lds_base + NST * B_STAGE;
  const unsigned b_off = (wcol * WTN + frow) * 128;
  const unsigned bpc0 = ((0 * 4 + fk) ^ (frow & 7)) * 16, bpc1 = ((1 * 4 + fk) ^ (frow & 7)) * 16;

  const int nk = 27 * p.nchunks;
  // All table values (go[], hp[]) are register-resident before the first DMA is issued: a later use would make
  // the compiler wait on the in-order vmcnt counter and serialise the DMA behind it.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < MI; ++i) asm volatile("" : "+v"(hp[i]));
#pragma unroll
  for (int q = 0; q < HQ; ++q) asm volatile("" : "+v"(go[q]));
  // ---- prologue: halo of chunk 0, then the first LOOKAHEAD filter tiles; tile 0 and the halo must have landed ----
  halo_load(0);
#pragma unroll
  for (int t = 0; t < LOOKAHEAD; ++t)
    if (t < nk) stage_b(t);
  if (nk >= LOOKAHEAD) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LOOKAHEAD - 1) * B_PER_WAVE) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (group_b) __builtin_amdgcn_s_barrier();      // run one half-step behind group A

#pragma clang loop unroll(disable)
  for (int kt = 0; kt < nk; ++kt) {
    const int tap = kt % 27;
    // ---------------- LOAD(kt): fragments of tap `tap` from the halo, filter tile kt from the ring ----------------
    const int sh = (tap / 9) * p.shift_z + ((tap / 3) % 3) * p.shift_y + tap % 3;
    unsigned aa[2][MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int v = hp[i] + sh;
      aa[0][i] = halo_lds + v * 128 + (((0 * 4 + fk) ^ (v & 7)) << 4);
      aa[1][i] = halo_lds + v * 128 + (((1 * 4 + fk) ^ (v & 7)) << 4);
    }
    const unsigned bbase = lds_base + (kt % NST) * B_STAGE + b_off;
    f32x4 a[2][MI], b[2][NI];
    asm volatile(
        "ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\tds_read_b128 %3, %11\n\t"
        "ds_read_b128 %4, %12\n\tds_read_b128 %5, %13\n\tds_read_b128 %6, %14\n\tds_read_b128 %7, %15"
        : "=&v"(a[0][0]), "=&v"(a[0][1]), "=&v"(a[0][2]), "=&v"(a[0][3]), "=&v"(a[1][0]), "=&v"(a[1][1]), "=&v"(a[1][2]), "=&v"(a[1][3])
        : "v"(aa[0][0]), "v"(aa[0][1]), "v"(aa[0][2]), "v"(aa[0][3]), "v"(aa[1][0]), "v"(aa[1][1]), "v"(aa[1][2]), "v"(aa[1][3])
        : "memory");
    asm volatile(
        "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:2048\n\tds_read_b128 %2, %8 offset:4096\n\tds_read_b128 %3, %8 offset:6144\n\t"
        "ds_read_b128 %4, %9\n\tds_read_b128 %5, %9 offset:2048\n\tds_read_b128 %6, %9 offset:4096\n\tds_read_b128 %7, %9 offset:6144"
        : "=&v"(b[0][0]), "=&v"(b[0][1]), "=&v"(b[0][2]), "=&v"(b[0][3]), "=&v"(b[1][0]), "=&v"(b[1][1]), "=&v"(b[1][2]), "=&v"(b[1][3])
        : "v"(bbase + bpc0), "v"(bbase + bpc1)
        : "memory");
    // this wave's share of filter tile kt+1 must have landed before the barrier publishes it
    if (kt + LOOKAHEAD - 1 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WAIT_KEEP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    // (the fragment registers are only valid after the lgkmcnt wait: tie them to it)
    asm volatile("" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3]));
    asm volatile("" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[0][3]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2]), "+v"(b[1][3]));
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- COMPUTE(kt) ----------------
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < NI; ++j) Mma<T>::step(acc[i][j], a[s][i], b[s][j]);
        if (s * MI + i == 1 && kt + LOOKAHEAD < nk) {
          __builtin_amdgcn_sched_barrier(0);
          stage_b(kt + LOOKAHEAD);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- channel-chunk switch: drain, reload the halo, re-stagger ----------------
    if (tap == 26 && kt + 1 < nk) {
      if (!group_b) __builtin_amdgcn_s_barrier();       // group B finishes COMPUTE(kt): nobody reads the halo any more
      asm volatile("" ::: "memory");
      halo_load(kt / 27 + 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (group_b) __builtin_amdgcn_s_barrier();
    }
  }
  if (!group_b) __builtin_amdgcn_s_barrier();
  __syncthreads();

  // ---- epilogue (slab scheme of igemm_kernel; rows = box voxels, pooling-window-major) ----
  constexpr int LDS_LD = BN + 4;
  float* stg = (float*)smem;
  static_assert(WTM * LDS_LD * 4 <= NST * B_STAGE, "epilogue slab fits in the filter ring");
  constexpr int CG = BN / 8;
  constexpr int ITEMS = (WTM / P) * CG;
#pragma unroll 1
  for (int slab = 0; slab < WM; ++slab) {
    if (wrow == slab) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            stg[(i * 16 + fk * 4 + r) * LDS_LD + wcol * WTN + j * 16 + frow] = acc[i][j][r];
    }
    __syncthreads();
    for (int it = tid; it < ITEMS; it += NT) {
      const int g = it / CG, cg = it - g * CG;
      float v[8];
      const float* src = stg + (g * P) * LDS_LD + cg * 8;
      const int prow = (slab * WTM) / P + g;   // pooled row inside the box
      pool_window<P>(src, LDS_LD, v);
      Epi::apply(e, p.N, img, prow, n0 + cg * 8, v);
    }
    __syncthreads();
  }
}

}  // namespace rgp
