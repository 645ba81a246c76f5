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
//            double-buffered LDS-DMA exactly as in igemm_kernel.
//   K loop : for chunk in Cin/BKE: for tap in 27: one barrier per K-tile; the NEXT chunk's
//            halo is prefetched one DMA per wave per K-tile into the second halo buffer.
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
  int n_img;
  int nbx, nby, nbz;        // boxes per image
  int box_in_x, box_in_y, box_in_z;   // element offset of one box step in the input  (elements)
  int box_out_x, box_out_y, box_out_z; // element offset of one box step in the output (elements)
};

template <int BM, int BN>
struct HaloSmem {
  static constexpr int B_STAGE = BN * 128;
  static constexpr int B_OFF = 0;                         // 2 filter stages
  static constexpr int HALO_OFF = 2 * B_STAGE;
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

  auto compute = [&](int hb, int bbuf, int tap) {
    const char* hbase = halo0 + hb * halo_bytes;
    const char* bbase = smem + bbuf * B_STAGE + b_off;
    const int sh = s_tapshift[tap];
    f32x4 a[2][MI], b[2][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int v = hp[i] + sh;
      const char* row = hbase + v * 128;
      a[0][i] = *(const f32x4*)(row + (((0 * 4 + fk) ^ (v & 7)) << 4));
      a[1][i] = *(const f32x4*)(row + (((1 * 4 + fk) ^ (v & 7)) << 4));
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      b[0][j] = *(const f32x4*)(bbase + j * 16 * 128 + bpc0);
      b[1][j] = *(const f32x4*)(bbase + j * 16 * 128 + bpc1);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) Mma<T>::step(acc[i][j], a[s][i], b[s][j]);
  };

  // ---- prologue: halo of chunk 0 and filter K-tile (tap 0, chunk 0) ----
  for (int i = wave; i < n_hinst; i += NW) halo_dma(0, 0, i);
  stage_b(0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int nk = 27 * p.nchunks;
  int tap = 0, cc = 0, bbuf = 0;
  int hnext = wave;          // next halo DMA instruction (of chunk cc+1) this wave issues
#pragma clang loop unroll(disable)
  for (int kt = 0; kt < nk; ++kt) {
    int tap_n = tap + 1, cc_n = cc;
    if (tap_n == 27) { tap_n = 0; cc_n = cc + 1; }
    if (kt + 1 < nk) stage_b(bbuf ^ 1, tap_n, cc_n);
    if (cc + 1 < p.nchunks && hnext < n_hinst) {     // trickle-prefetch next chunk's halo
      halo_dma((cc + 1) & 1, cc + 1, hnext);
      hnext += NW;
    }
    compute(cc & (halo_bufs - 1), bbuf, tap);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bbuf ^= 1;
    if (tap_n == 0) hnext = wave;
    tap = tap_n;
    cc = cc_n;
  }

  // ---- epilogue (same slab scheme as igemm_kernel; rows = box voxels) ----
  constexpr int LDS_LD = BN + 4;
  float* stg = (float*)halo0;
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
      f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + 4);
#pragma unroll
      for (int q = 1; q < P; ++q) {
        const f32x4 w0 = *(const f32x4*)(src + q * LDS_LD), w1 = *(const f32x4*)(src + q * LDS_LD + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v0[i] = fmaxf(v0[i], w0[i]); v1[i] = fmaxf(v1[i], w1[i]); }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] = v0[i]; v[4 + i] = v1[i]; }
      Epi::apply(e, p.N, img, (slab * WTM) / P + g, n0 + cg * 8, v);
    }
    __syncthreads();
  }
}

}  // namespace rgp
