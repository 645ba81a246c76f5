// 3x3x3 convolution with an LDS-resident input halo (gfx950) -- the C3D conv2a / conv3a /
// conv3b kernel.
//
// Why: the im2col tile loop (igemm.hip.h) re-gathers every input voxel 27 times from L2
// into LDS; at the MFMA rate of the bf16 16x16x32 tile loop that is ~130 GB/s per CU of
// LDS-DMA ingest against a measured ~70 GB/s per CU ceiling (MI355X_MICROARCH.md, "Indexed
// rows: gather into LDS"), so the matrix pipe idles at ~50 %.  Here a block owns a BOX of
// output voxels, loads the box's input halo (box + 1 voxel each side, one 64-channel chunk =
// 128 B per voxel) into LDS ONCE per channel chunk, and serves all 27 taps from it by
// shifting the fragment read address; only the filter K-tiles stream.  LDS-DMA bytes per
// FLOP drop 2-2.5x and the remaining stream (filters) is shared by every block.
//
//   rows   : the box's output voxels, pooling-window-major (the 8 rows of a 2x2x2 window are
//            consecutive), so the fused max-pool epilogue is the same as igemm_kernel's.
//   A frag : lane (row, kgroup) reads LDS voxel hp[row] + tap_shift[tap], 16-B chunk
//            (kgroup ^ voxel&7)  -- same XOR swizzle as the DMA source permutation.
//   B tile : packed filter rows n0..n0+BN, K-tile (tap, chunk) at k = tap*Cin + chunk*BKE,
//            streamed by LDS-DMA through a 3-stage ring, issued two taps ahead.
//   K loop : for chunk in Cin/BKE: for tap in 27: counted vmcnt + one raw barrier per K-tile; the halo is
//            reloaded at each chunk switch (a second halo buffer does not fit next to the ring).
#pragma once
#include "igemm.hip.h"

namespace rgp {

struct HaloParams {
  const void* A;            // halo-padded NDHWC activations (T)
  const void* W;            // packed filter [Npad][27*Cin] (T)
  const int* halo_goff;     // [HP8] element offset of halo voxel h relative to the box origin (chunk 0)
  const int* row_hp;        // [BM] halo voxel index of row r's receptive-field origin
  const int* tap_shift;     // [27] halo voxel shift of tap (kz,ky,kx)
  long long in_img_stride;  // elements
  int Cin, N, K;            // K = 27*Cin
  int nchunks;              // Cin / BKE
  int HP8;                  // halo voxels, padded to a multiple of 8
  int shift_y, shift_z;     // halo-voxel index step of one tap in y (= HX) and z (= HY*HX)
  int n_img;
  int nbx, nby, nbz;        // boxes per image
  int box_in_x, box_in_y, box_in_z;   // element offset of one box step in the input  (elements)
  int box_out_x, box_out_y, box_out_z; // element offset of one box step in the output (elements)
};

template <int BM, int BN>
struct HaloSmem {
  static constexpr int B_STAGE = BN * 128;
  static constexpr int B_OFF = 0;                         // 3 filter stages (ring)
  static constexpr int HALO_OFF = 3 * B_STAGE;
  static constexpr int halo_bytes(int hp8) { return hp8 * 128; }
};

template <typename T, int BM, int BN, int WM, int WN, int P, class Epi>
__global__ __launch_bounds__(WM* WN * 64) void conv3d_halo_kernel(const HaloParams p, EpiParams e, int halo_bufs) {
  constexpr int NW = WM * WN, NT = NW * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16;
  constexpr int B_PER_WAVE = (BN / 8) / NW;
  constexpr int BKE = Elem<T>::BKE, ESZ = sizeof(T);
  constexpr int B_STAGE = HaloSmem<BM, BN>::B_STAGE;
  static_assert((BN / 8) % NW == 0 && WTM % 16 == 0 && WTN % 16 == 0 && WTM % P == 0, "tile");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int halo_bytes = p.HP8 * 128;
  char* halo0 = smem + HaloSmem<BM, BN>::HALO_OFF;
  int* s_goff = (int*)(halo0 + halo_bufs * halo_bytes);
  int* s_tapshift = s_goff + p.HP8;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // block -> (image, box, n-tile); n-tile fastest so the N/BN blocks of one box share L2
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

  for (int i = tid; i < p.HP8; i += NT) s_goff[i] = p.halo_goff[i];
  if (tid < 27) s_tapshift[tid] = p.tap_shift[tid];
  __syncthreads();

  const int lrow = lane >> 3;
  const int lchunk = (lane & 7) ^ lrow;
  const char* a_base = (const char*)p.A + in_origin * ESZ + lchunk * 16;
  const char* b_src[B_PER_WAVE];
#pragma unroll
  for (int j = 0; j < B_PER_WAVE; ++j) {
    const int r = (wave * B_PER_WAVE + j) * 8 + lrow;
    b_src[j] = (const char*)p.W + ((long long)(n0 + r) * p.K) * ESZ + lchunk * 16;
  }
  const int n_hinst = p.HP8 >> 3;      // halo DMA instructions per chunk (8 voxels each)

  // halo DMA instruction i of channel chunk cc into halo buffer hb
  auto halo_dma = [&](int hb, int cc, int i) {
    const int go = s_goff[i * 8 + lrow];
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)(a_base + ((long long)go + (long long)cc * BKE) * ESZ),
        (__attribute__((address_space(3))) void*)(halo0 + hb * halo_bytes + i * 1024), 16, 0, 0);
  };
  auto stage_b = [&](int buf, int tap, int cc) {
    const long long kb = ((long long)tap * p.Cin + (long long)cc * BKE) * ESZ;
#pragma unroll
    for (int j = 0; j < B_PER_WAVE; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[j] + kb),
                                       (__attribute__((address_space(3))) void*)(smem + buf * B_STAGE + (wave * B_PER_WAVE + j) * 1024),
                                       16, 0, 0);
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fk = lane >> 4;
  int hp[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) hp[i] = p.row_hp[wm * WTM + i * 16 + frow];
  const int b_off = (wn * WTN + frow) * 128;
  const int bpc0 = ((0 * 4 + fk) ^ (frow & 7)) * 16, bpc1 = ((1 * 4 + fk) ^ (frow & 7)) * 16;

  // Fragment reads are inline asm with their own lgkmcnt wait: through plain loads the compiler orders every LDS
  // read after ALL outstanding LDS-DMA (s_waitcnt vmcnt(0)), which would drain the filter ring every tap.
  static_assert(MI == 4 && NI == 4, "64x64 wave tiles");
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned halo_lds = lds_base + (unsigned)(halo0 - smem);
  auto compute = [&](int hb, int bbuf, int tap) {
    const unsigned hbase = halo_lds + hb * halo_bytes;
    const unsigned bbase = lds_base + bbuf * B_STAGE + b_off;
    // halo-voxel shift of tap (kz,ky,kx), computed (an LDS table read here would make the compiler drain the DMA ring)
    const int sh = (tap / 9) * p.shift_z + ((tap / 3) % 3) * p.shift_y + tap % 3;
    unsigned aa[2][MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int v = hp[i] + sh;
      aa[0][i] = hbase + v * 128 + (((0 * 4 + fk) ^ (v & 7)) << 4);
      aa[1][i] = hbase + v * 128 + (((1 * 4 + fk) ^ (v & 7)) << 4);
    }
    f32x4 a[2][MI], b[2][NI];
    asm volatile(
        "ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\tds_read_b128 %3, %11\n\t"
        "ds_read_b128 %4, %12\n\tds_read_b128 %5, %13\n\tds_read_b128 %6, %14\n\tds_read_b128 %7, %15"
        : "=&v"(a[0][0]), "=&v"(a[0][1]), "=&v"(a[0][2]), "=&v"(a[0][3]), "=&v"(a[1][0]), "=&v"(a[1][1]), "=&v"(a[1][2]), "=&v"(a[1][3])
        : "v"(aa[0][0]), "v"(aa[0][1]), "v"(aa[0][2]), "v"(aa[0][3]), "v"(aa[1][0]), "v"(aa[1][1]), "v"(aa[1][2]), "v"(aa[1][3])
        : "memory");
    asm volatile(
        "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:2048\n\tds_read_b128 %2, %8 offset:4096\n\tds_read_b128 %3, %8 offset:6144\n\t"
        "ds_read_b128 %4, %9\n\tds_read_b128 %5, %9 offset:2048\n\tds_read_b128 %6, %9 offset:4096\n\tds_read_b128 %7, %9 offset:6144\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(b[0][0]), "=&v"(b[0][1]), "=&v"(b[0][2]), "=&v"(b[0][3]), "=&v"(b[1][0]), "=&v"(b[1][1]), "=&v"(b[1][2]), "=&v"(b[1][3])
        : "v"(bbase + bpc0), "v"(bbase + bpc1)
        : "memory");
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) Mma<T>::step(acc[i][j], a[s][i], b[s][j]);
  };

  // ---- K loop: the input halo of one channel chunk is LDS-resident for its 27 taps; the filter K-tiles
  // stream through a 3-stage ring issued two taps ahead (counted vmcnt, one raw barrier per tap: a wait
  // for everything in flight would expose the whole L2 latency of the next tile every tap). ----
  const int nk = 27 * p.nchunks;
  auto kt_tap = [&](int kt) { return kt % 27; };
  auto kt_cc = [&](int kt) { return kt / 27; };
  for (int i = wave; i < n_hinst; i += NW) halo_dma(0, 0, i);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // keeps the ring's vmcnt arithmetic free of halo loads
  // hp[] came from a global load: re-define the registers here, behind the wait, or the compiler guards their first
  // use inside the loop with its own vmcnt(0) (the in-order counter then also covers the ring's DMA)
#pragma unroll
  for (int i = 0; i < MI; ++i) asm volatile("" : "+v"(hp[i]));
  stage_b(0, 0, 0);
  if (nk > 1) stage_b(1, kt_tap(1), kt_cc(1));
#pragma clang loop unroll(disable)
  for (int kt = 0; kt < nk; ++kt) {
    const int tap = kt_tap(kt), cc = kt_cc(kt);
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(B_PER_WAVE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // K-tile kt landed for every wave; stage (kt+2)%3 is free
    asm volatile("" ::: "memory");         // the barrier builtin does not order plain LDS loads: keep compute() below it
    if (kt + 2 < nk) stage_b((kt + 2) % 3, kt_tap(kt + 2), kt_cc(kt + 2));
    compute(0, kt % 3, tap);
    if (tap == 26 && cc + 1 < p.nchunks) {
      // chunk switch: every wave is done with the resident halo, reload it (exposed once per chunk)
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      for (int i = wave; i < n_hinst; i += NW) halo_dma(0, cc + 1, i);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();

  // ---- epilogue (same slab scheme as igemm_kernel; rows = box voxels) ----
  constexpr int LDS_LD = BN + 4;
  float* stg = (float*)smem;      // the filter ring is free now (WTM x (BN+4) floats fit in its 3 stages)
  static_assert(WTM * (BN + 4) * 4 <= 3 * B_STAGE, "epilogue slab fits in the filter ring");
  constexpr int CG = BN / 8;
  constexpr int ITEMS = (WTM / P) * CG;
#pragma unroll 1
  for (int slab = 0; slab < WM; ++slab) {
    if (wm == slab) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            stg[(i * 16 + fk * 4 + r) * LDS_LD + wn * WTN + j * 16 + frow] = acc[i][j][r];
    }
    __syncthreads();
    for (int it = tid; it < ITEMS; it += NT) {
      const int g = it / CG, cg = it - g * CG;
      float v[8];
      const float* src = stg + (g * P) * LDS_LD + cg * 8;
      pool_window<P>(src, LDS_LD, v);
      Epi::apply(e, p.N, img, (slab * WTM) / P + g, n0 + cg * 8, v);
    }
    __syncthreads();
  }
}

}  // namespace rgp
