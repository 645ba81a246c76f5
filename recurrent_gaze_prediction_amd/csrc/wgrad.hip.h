// Filter-gradient implicit GEMM for gfx950 (the wgrad half of tf.gradients through a conv layer,
// base.py:278-281 applied to the C3D stack of feature_extration.prototxt:22-342):
//
//   dW[k][n] += sum_m  X[rowbase_x(m) + koff(k)] * dY[rowbase_y(m) + n]
//
// i.e. (im2col)^T x dY with the REDUCTION over the output rows m.  Both operands are stored with the
// reduction index strided in memory (activations are channels-last), the opposite of what an MFMA
// fragment wants, so:
//  * row tiles [32 rows][128 B] of X (per K-chunk) and [32 rows][BN cols] of dY are staged by LDS-DMA
//    exactly as they lie in HBM (full 128/256/512-B lines, no transposing gather),
//  * bf16 fragments are read with ds_read_b64_tr_b16, the gfx950 transposing LDS read: a 16-lane group
//    fetches a 4-row x 16-column block and each lane receives one column = 4 consecutive reduction
//    indices of its channel; two reads make the 8-deep k-group of v_mfma_f32_16x16x32_bf16,
//  * 32-byte segments are XOR-swizzled per row (on the DMA source side, the LDS image is lane-linear)
//    so the 8 rows a 32-lane half touches fall in 8 different bank groups,
//  * fp32 uses v_mfma_f32_16x16x4_f32 whose fragments are single dwords: plain ds_read_b32.
// Block tile: 4 K-chunks (4 x 64 bf16 / 4 x 32 fp32 filter rows) x BN = 32*WNT output channels, 8 waves as
// 4 (chunk) x 2 (16*WNT columns); the reduction is split over blockIdx.y and combined with fp32 atomics.
// WNT = 4 (BN 128, two blocks per CU) or 8 (BN 256, one block per CU: a third fewer LDS-DMA bytes and a quarter
// fewer fragment reads per FLOP -- the kernel is LDS-DMA-ingest bound).
// 3-stage DMA ring, one raw barrier per 32-row step, counted vmcnt.
#pragma once
#include "igemm.hip.h"

namespace rgp {

struct WgradParams {
  const void* X;           // layer input, halo-padded channels-last (operand dtype)
  const void* dY;          // gradient w.r.t. the conv output before pooling, halo-padded (operand dtype);
                           // its first 128*sizeof(T) bytes must be zero (they are: halo)
  float* dW;               // [nk*BKE][ldw] fp32, accumulated with atomics
  // Row m of an image is the output position (z, y, x), m = (z*H + y)*W + x.  Its window origin in X is
  // z*x_sz + y*x_sy + x*x_sx elements, its position in dY is y_org + z*y_sz + y*y_sy + x*y_sx.  The offsets
  // are computed, not looked up: a per-lane table load inside the loop would sit in the same in-order
  // vmcnt queue as the LDS-DMA and force the ring to drain.
  int D, H, W;
  int x_sz, x_sy, x_sx, y_sz, y_sy, y_sx, y_org;
  float inv_W, inv_H, inv_D;
  const int* koff;         // [nk*G] element offset of each K (sub-)chunk
  long long x_img_stride, y_img_stride;
  long long M;             // rows = images * Mw
  int Mw, N, nk, ldw;
  int k_valid;             // rows of dW that exist (<= nk*BKE); the rest of the last chunk is channel padding
  int steps_per_split;     // 32-row steps per blockIdx.y
  int ablate;              // dev diagnostics (RGP_WG_ABLATE): 1 = no reads/MFMA, 2 = no in-loop DMA, 4 = no barrier; results are garbage
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <typename T, int WNT> struct WgradSmem {
  static constexpr int BN = 32 * WNT;
  static constexpr int XB = 4 * 32 * 128;                    // 4 chunks x 32 rows x 128 B
  static constexpr int YB = 32 * BN * (int)sizeof(T);        // 32 rows x BN columns
  static constexpr int STAGE = XB + YB;
  static constexpr int BYTES = 3 * STAGE;
};

// 16-byte-chunk XOR applied to row r of an X tile / a dY tile (bf16 only; see header comment)
template <typename T> __device__ __forceinline__ int wg_swz_x(int r) {
  return sizeof(T) == 2 ? 2 * (((r >> 1) & 1) | (((r >> 3) & 1) << 1)) : 0;
}
template <typename T> __device__ __forceinline__ int wg_swz_y(int r) {
  return sizeof(T) == 2 ? 2 * ((r & 3) | (((r >> 3) & 1) << 2)) : 0;
}

// STAG (bf16, WNT = 8, one block per CU): the two wave groups 0-3 / 4-7 (partners on a SIMD) run half a step apart,
//   LOAD(s): fragment reads of step s, DMA of step s+2, counted vmcnt, barrier     COMPUTE(s): 32 MFMAs, barrier
// so one wave of a SIMD is in its MFMAs while the other reads (igemm_stagger.hip.h).  Without it the 256-wide tile
// -- a third fewer LDS-DMA bytes per FLOP than two co-resident 128-wide blocks -- loses what it saves to exposed
// LDS latency (its two waves per SIMD read, wait and compute in lockstep).
template <typename T, int G, int WNT, bool STAG = false>
__global__ __launch_bounds__(512, WNT == 4 && sizeof(T) == 2 ? 4 : 2) void wgrad_kernel(const WgradParams p) {
  static_assert(!STAG || (WNT == 8 && sizeof(T) == 2), "staggered schedule: bf16, 256-wide tile");
  constexpr int ESZ = sizeof(T);
  constexpr int BKE = Elem<T>::BKE;               // filter rows per K-chunk (64 bf16, 32 fp32)
  constexpr int CI = BKE / 16;                    // 16-row tiles per chunk
  constexpr int BN = 32 * WNT;                    // output channels per block
  constexpr int YROW = BN * ESZ;                  // bytes per dY tile row
  constexpr int CPR = YROW / 16;                  // 16-byte chunks per dY tile row
  constexpr int NYL = (32 * CPR) / 512;           // dY DMA instructions per thread and step
  constexpr int PER_STEP = 2 + NYL;               // DMA instructions per thread and step
  using S = WgradSmem<T, WNT>;
  static_assert(WNT == 4 || WNT == 8, "wave tile 64x64 or 64x128");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave >> 1, wn = wave & 1;
  const int n_kt = (p.nk + 3) / 4;
  // (Dealing whole row ranges to one XCD so that its L2 serves all their K-tiles was measured: conv4b -10 %, but
  // conv2a / conv3a / conv3b +15...25 % -- the plain order, K-tiles of a row range spread over the XCDs, stays.)
  const int kt = blockIdx.x % n_kt, nt = blockIdx.x / n_kt;
  const int n0 = nt * BN;
  const long long m_begin = (long long)blockIdx.y * p.steps_per_split * 32;
  long long m_end = m_begin + (long long)p.steps_per_split * 32;
  if (m_end > p.M) m_end = p.M;
  if (m_begin >= m_end) return;
  const int nsteps = (int)((m_end - m_begin + 31) / 32);

  // ---- DMA source bookkeeping: each thread owns one X row (2 chunks of it) and NYL dY rows ----
  const int xr = 8 * (wave & 3) + (lane >> 3);              // tile row of this thread's X loads
  const int xc = (lane & 7) ^ wg_swz_x<T>(xr);              // logical 16-B chunk it fetches
  const char* xsrc_k[2];                                    // chunk-dependent part (koff), fixed per block
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    int kc = kt * 4 + (wave >> 2) + 2 * u;
    if (kc >= p.nk) kc = p.nk - 1;                          // duplicate work, never stored
    const int elems_per_sub = BKE / G, ce = xc * (16 / ESZ);
    const int sub = ce / elems_per_sub, off = ce - sub * elems_per_sub;
    xsrc_k[u] = (const char*)p.X + ((long long)p.koff[kc * G + sub] + off) * ESZ;
  }
  // dY: lane-load q = tid + 512 u covers tile row q / CPR, physical chunk q % CPR (the LDS image is lane-linear)
  int yr[NYL], yc[NYL];
#pragma unroll
  for (int u = 0; u < NYL; ++u) {
    const int q = tid + 512 * u;
    yr[u] = q / CPR;
    yc[u] = (q % CPR) ^ wg_swz_y<T>(yr[u]);
  }
  // running (image, z, y, x) of each owned row; a step advances every row by 32
  struct RowPos { int img, z, y, x; };
  auto locate = [&](long long m) {
    RowPos r;
    r.img = (int)(m / p.Mw);
    int ml = (int)(m - (long long)r.img * p.Mw);
    r.x = ml % p.W; ml /= p.W;
    r.y = ml % p.H;
    r.z = ml / p.H;
    return r;
  };
  auto advance = [&](RowPos& r) {
    r.x += 32;
    int t = (int)(((float)r.x + 0.5f) * p.inv_W); r.x -= t * p.W; r.y += t;
    t = (int)(((float)r.y + 0.5f) * p.inv_H); r.y -= t * p.H; r.z += t;
    t = (int)(((float)r.z + 0.5f) * p.inv_D); r.z -= t * p.D; r.img += t;
  };
  RowPos xpos = locate(m_begin + xr), ypos[NYL];
#pragma unroll
  for (int u = 0; u < NYL; ++u) ypos[u] = locate(m_begin + yr[u]);
  int n_issued = 0;

  auto issue = [&](int buf) {
    char* xb = smem + buf * S::STAGE;
    char* yb = xb + S::XB;
    const long long m_step = m_begin + (long long)n_issued * 32;
    {
      const bool ok = m_step + xr < m_end;
      const long long base = ((long long)xpos.img * p.x_img_stride + xpos.z * p.x_sz + xpos.y * p.x_sy + xpos.x * p.x_sx) * ESZ;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const char* src = ok ? xsrc_k[u] + base : (const char*)p.X;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(xb + ((wave >> 2) + 2 * u) * 4096 + (wave & 3) * 1024),
                                         16, 0, 0);
      }
      advance(xpos);
    }
#pragma unroll
    for (int u = 0; u < NYL; ++u) {
      const bool ok = m_step + yr[u] < m_end;
      const RowPos& r = ypos[u];
      const char* src = (const char*)p.dY;
      if (ok) src += ((long long)r.img * p.y_img_stride + p.y_org + r.z * p.y_sz + r.y * p.y_sy + r.x * p.y_sx + n0) * ESZ + yc[u] * 16;
      else src += (yc[u] & 7) * 16;                                        // zeros (halo)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(yb + wave * 1024 + u * 8192), 16, 0, 0);
      advance(ypos[u]);
    }
    ++n_issued;
  };

  f32x4 acc[CI][WNT];
#pragma unroll
  for (int i = 0; i < CI; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fcol = lane & 15, g = lane >> 4;
  // filter rows of this wave's K-chunk that exist: a wave whose chunk lies beyond nk, and 16-row tiles beyond
  // k_valid (channel padding, or a 16-row problem such as the head's Toeplitz filter gradient), skip their MFMAs
  const int my_rows = (kt * 4 + wk < p.nk) ? p.k_valid - (kt * 4 + wk) * BKE : 0;
  const int ci_n = my_rows <= 0 ? 0 : (my_rows >= BKE ? CI : (my_rows + 15) >> 4);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  auto compute = [&](int buf) {
    const char* xb = smem + buf * S::STAGE + wk * 4096;
    const char* yb = smem + buf * S::STAGE + S::XB;
    if constexpr (ESZ == 2) {
      const int q = fcol >> 2, pp = fcol & 3;
      const int row = 8 * g + q;                                 // rows row and row+4 share the swizzle
      const int sx = (wg_swz_x<T>(row) >> 1), sy = (wg_swz_y<T>(row) >> 1);
      // The transposing reads are written as inline asm: issued through the builtin, the compiler orders
      // every LDS read after ALL outstanding LDS-DMA (s_waitcnt vmcnt(0)), which drains the two tiles in
      // flight and serialises the ring.  The asm carries its own lgkmcnt wait; A's registers are consumed
      // only by MFMAs that also need B's, so one wait at the end of the last block covers them all.
      static_assert(CI == 4, "bf16 chunk = 4 x 16 filter rows");
      const unsigned xa = lds_base + (unsigned)(xb - smem) + row * 128 + pp * 8;
      const unsigned ya = lds_base + (unsigned)(yb - smem) + row * YROW + pp * 8;
      i32x2 al[4], ah[4], bl[WNT], bh[WNT];
      asm volatile(
          "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:512\n\t"
          "ds_read_b64_tr_b16 %2, %9\n\tds_read_b64_tr_b16 %3, %9 offset:512\n\t"
          "ds_read_b64_tr_b16 %4, %10\n\tds_read_b64_tr_b16 %5, %10 offset:512\n\t"
          "ds_read_b64_tr_b16 %6, %11\n\tds_read_b64_tr_b16 %7, %11 offset:512"
          : "=&v"(al[0]), "=&v"(ah[0]), "=&v"(al[1]), "=&v"(ah[1]), "=&v"(al[2]), "=&v"(ah[2]), "=&v"(al[3]), "=&v"(ah[3])
          : "v"(xa + ((0 ^ sx) * 32)), "v"(xa + ((1 ^ sx) * 32)), "v"(xa + ((2 ^ sx) * 32)), "v"(xa + ((3 ^ sx) * 32))
          : "memory");
      // rows row+4 of the dY tile are 4*YROW bytes further: 1024 (BN 128) or 2048 (BN 256)
#pragma unroll
      for (int jb = 0; jb < WNT; jb += 4) {
        if constexpr (WNT == 4) {
          asm volatile(
              "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:1024\n\t"
              "ds_read_b64_tr_b16 %2, %9\n\tds_read_b64_tr_b16 %3, %9 offset:1024\n\t"
              "ds_read_b64_tr_b16 %4, %10\n\tds_read_b64_tr_b16 %5, %10 offset:1024\n\t"
              "ds_read_b64_tr_b16 %6, %11\n\tds_read_b64_tr_b16 %7, %11 offset:1024"
              : "=&v"(bl[jb]), "=&v"(bh[jb]), "=&v"(bl[jb + 1]), "=&v"(bh[jb + 1]), "=&v"(bl[jb + 2]), "=&v"(bh[jb + 2]), "=&v"(bl[jb + 3]),
                "=&v"(bh[jb + 3])
              : "v"(ya + (((wn * WNT + jb + 0) ^ sy) * 32)), "v"(ya + (((wn * WNT + jb + 1) ^ sy) * 32)),
                "v"(ya + (((wn * WNT + jb + 2) ^ sy) * 32)), "v"(ya + (((wn * WNT + jb + 3) ^ sy) * 32))
              : "memory");
        } else {
          asm volatile(
              "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:2048\n\t"
              "ds_read_b64_tr_b16 %2, %9\n\tds_read_b64_tr_b16 %3, %9 offset:2048\n\t"
              "ds_read_b64_tr_b16 %4, %10\n\tds_read_b64_tr_b16 %5, %10 offset:2048\n\t"
              "ds_read_b64_tr_b16 %6, %11\n\tds_read_b64_tr_b16 %7, %11 offset:2048"
              : "=&v"(bl[jb]), "=&v"(bh[jb]), "=&v"(bl[jb + 1]), "=&v"(bh[jb + 1]), "=&v"(bl[jb + 2]), "=&v"(bh[jb + 2]), "=&v"(bl[jb + 3]),
                "=&v"(bh[jb + 3])
              : "v"(ya + (((wn * WNT + jb + 0) ^ sy) * 32)), "v"(ya + (((wn * WNT + jb + 1) ^ sy) * 32)),
                "v"(ya + (((wn * WNT + jb + 2) ^ sy) * 32)), "v"(ya + (((wn * WNT + jb + 3) ^ sy) * 32))
              : "memory");
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // the fragment registers are valid only behind the wait: re-define them there
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(al[i]), "+v"(ah[i]));
#pragma unroll
      for (int j = 0; j < WNT; ++j) asm volatile("" : "+v"(bl[j]), "+v"(bh[j]));
      f32x4 a[CI], b[WNT];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const i32x4 va = {al[i][0], al[i][1], ah[i][0], ah[i][1]};
        a[i] = __builtin_bit_cast(f32x4, va);
      }
#pragma unroll
      for (int j = 0; j < WNT; ++j) {
        const i32x4 vb = {bl[j][0], bl[j][1], bh[j][0], bh[j][1]};
        b[j] = __builtin_bit_cast(f32x4, vb);
      }
#pragma unroll
      for (int i = 0; i < CI; ++i)
        if (i < ci_n) {
#pragma unroll
          for (int j = 0; j < WNT; ++j) Mma<T>::step(acc[i][j], a[i], b[j]);
        }
    } else {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int row = 4 * s + g;
        float a[CI], b[WNT];
#pragma unroll
        for (int i = 0; i < CI; ++i) a[i] = *(const float*)(xb + row * 128 + (i * 16 + fcol) * 4);
#pragma unroll
        for (int j = 0; j < WNT; ++j) b[j] = *(const float*)(yb + row * YROW + ((wn * WNT + j) * 16 + fcol) * 4);
#pragma unroll
        for (int i = 0; i < CI; ++i)
          if (i < ci_n) {
#pragma unroll
            for (int j = 0; j < WNT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
          }
      }
    }
  };

  // STAG: the same reads / MFMAs as compute(), split around the group barrier (fragments live in registers across it)
  i32x2 s_al[4], s_ah[4], s_bl[WNT], s_bh[WNT];
  auto stag_reads = [&](int buf) {
    if constexpr (STAG) {
      const char* xb = smem + buf * S::STAGE + wk * 4096;
      const char* yb = smem + buf * S::STAGE + S::XB;
      const int q = fcol >> 2, pp = fcol & 3;
      const int row = 8 * g + q;
      const int sx = (wg_swz_x<T>(row) >> 1), sy = (wg_swz_y<T>(row) >> 1);
      const unsigned xa = lds_base + (unsigned)(xb - smem) + row * 128 + pp * 8;
      const unsigned ya = lds_base + (unsigned)(yb - smem) + row * YROW + pp * 8;
      asm volatile(
          "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:512\n\t"
          "ds_read_b64_tr_b16 %2, %9\n\tds_read_b64_tr_b16 %3, %9 offset:512\n\t"
          "ds_read_b64_tr_b16 %4, %10\n\tds_read_b64_tr_b16 %5, %10 offset:512\n\t"
          "ds_read_b64_tr_b16 %6, %11\n\tds_read_b64_tr_b16 %7, %11 offset:512"
          : "=&v"(s_al[0]), "=&v"(s_ah[0]), "=&v"(s_al[1]), "=&v"(s_ah[1]), "=&v"(s_al[2]), "=&v"(s_ah[2]), "=&v"(s_al[3]), "=&v"(s_ah[3])
          : "v"(xa + ((0 ^ sx) * 32)), "v"(xa + ((1 ^ sx) * 32)), "v"(xa + ((2 ^ sx) * 32)), "v"(xa + ((3 ^ sx) * 32))
          : "memory");
#pragma unroll
      for (int jb = 0; jb < WNT; jb += 4)
        asm volatile(
            "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:2048\n\t"
            "ds_read_b64_tr_b16 %2, %9\n\tds_read_b64_tr_b16 %3, %9 offset:2048\n\t"
            "ds_read_b64_tr_b16 %4, %10\n\tds_read_b64_tr_b16 %5, %10 offset:2048\n\t"
            "ds_read_b64_tr_b16 %6, %11\n\tds_read_b64_tr_b16 %7, %11 offset:2048"
            : "=&v"(s_bl[jb]), "=&v"(s_bh[jb]), "=&v"(s_bl[jb + 1]), "=&v"(s_bh[jb + 1]), "=&v"(s_bl[jb + 2]), "=&v"(s_bh[jb + 2]),
              "=&v"(s_bl[jb + 3]), "=&v"(s_bh[jb + 3])
            : "v"(ya + (((wn * WNT + jb + 0) ^ sy) * 32)), "v"(ya + (((wn * WNT + jb + 1) ^ sy) * 32)),
              "v"(ya + (((wn * WNT + jb + 2) ^ sy) * 32)), "v"(ya + (((wn * WNT + jb + 3) ^ sy) * 32))
            : "memory");
    }
  };
  auto stag_mma = [&]() {
    if constexpr (STAG) {
      // the fragment registers are valid only behind the lgkmcnt(0) of the LOAD phase: re-define them here
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(s_al[i]), "+v"(s_ah[i]));
#pragma unroll
      for (int j = 0; j < WNT; ++j) asm volatile("" : "+v"(s_bl[j]), "+v"(s_bh[j]));
      f32x4 a[CI], b[WNT];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const i32x4 va = {s_al[i][0], s_al[i][1], s_ah[i][0], s_ah[i][1]};
        a[i] = __builtin_bit_cast(f32x4, va);
      }
#pragma unroll
      for (int j = 0; j < WNT; ++j) {
        const i32x4 vb = {s_bl[j][0], s_bl[j][1], s_bh[j][0], s_bh[j][1]};
        b[j] = __builtin_bit_cast(f32x4, vb);
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < CI; ++i)
        if (i < ci_n) {
#pragma unroll
          for (int j = 0; j < WNT; ++j) Mma<T>::step(acc[i][j], a[i], b[j]);
        }
      __builtin_amdgcn_s_setprio(0);
    }
  };

  // ---- 3-stage ring: steps s+1 and s+2 are in flight while step s is consumed.  Steps beyond the
  // block's range are issued too (they fetch zeros), so the vmcnt arithmetic is uniform. ----
  issue(0);
  issue(1);
  if constexpr (STAG) {
    // Hazards as in igemm_stagger.hip.h (h = half-step; group A loads step s at h = 2s, computes at 2s+1; B one later):
    //  RAW  step s is read from h = 2s on; its DMA was issued in LOAD(s-2) and every wave passed vmcnt(PER_STEP) for it
    //       in LOAD(s-1) before that phase's barrier (A: end of 2s-2, B: end of 2s-1).
    //  WAR  the DMA of step s+2 overwrites the stage of step s-1, last read in LOAD(s-1) (A: 2s-2, B: 2s-1, each followed
    //       by lgkmcnt(0) + barrier); it is issued at h >= 2s.
    const bool group_b = wave >= 4;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STEP) : "memory");   // step 0 landed
    __builtin_amdgcn_s_barrier();
    if (group_b) __builtin_amdgcn_s_barrier();
#pragma clang loop unroll(disable)
    for (int s = 0; s < nsteps; ++s) {
      stag_reads(s % 3);
      __builtin_amdgcn_sched_barrier(0);
      issue((s + 2) % 3);                                            // (beyond the range: zeros, uniform vmcnt arithmetic)
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PER_STEP) : "memory");   // step s+1 landed, fragments in registers
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      stag_mma();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!group_b) __builtin_amdgcn_s_barrier();
  } else {
#pragma clang loop unroll(disable)
  for (int s = 0; s < nsteps; ++s) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STEP) : "memory");
    if (!(p.ablate & 4)) __builtin_amdgcn_s_barrier();           // step s landed for every wave; stage (s+2)%3 is free
    if (!(p.ablate & 2)) issue((s + 2) % 3);
    if (!(p.ablate & 1) && ci_n > 0) compute(s % 3);
  }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: D[row = 4*(lane>>4)+r (filter row), col = lane&15 (output channel)] ----
  const int kc = kt * 4 + wk;
  if (kc < p.nk) {
#pragma unroll
    for (int i = 0; i < CI; ++i)
#pragma unroll
      for (int j = 0; j < WNT; ++j) {
        const int n = n0 + (wn * WNT + j) * 16 + fcol;
        if (n < p.N) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const long long k = (long long)kc * BKE + i * 16 + g * 4 + r;
            if (k < p.k_valid) atomicAdd(p.dW + k * p.ldw + n, acc[i][j][r]);
          }
        }
      }
  }
}

}  // namespace rgp
