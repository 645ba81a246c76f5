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
// fewer fragment reads per FLOP -- the kernel is LDS-DMA-ingest bound, docs/HISTORY.md section 4).
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
  // z*x_sz + y*x_sy + x*x_sx elements, its position in dY is y_org + z*y_sz + y*y_sy + x*y_sx: wgrad_row_tables()
  // (rgp_host.h) tabulates both as BYTE offsets from the image, x_tab / y_tab [Mw + 32], entry e >= Mw continuing into
  // the following image(s).  A 32-row step starts at a (scalar) image and row ml0 and row r of it is entry ml0 + r, so
  // the loop's per-lane work is one table load per owned row, issued one step ahead and IN FRONT of that step's LDS-DMA
  // in the in-order vmcnt queue (the counted wait that lets one step of DMA stay in flight then also covers it), and
  // one add per DMA.  (The first version recomputed z*sz + y*sy + x*sx per row and step with float reciprocals: 8 VALU
  // instructions per MFMA, 16 of every 110 the quarter-rate v_mul_lo_u32 -- the matrix pipe was 36 % busy.)
  const int *x_tab, *y_tab;
  int D, H, W;
  int x_sz, x_sy, x_sx, y_sz, y_sy, y_sx, y_org;
  const int* koff;         // [nk*G] element offset of each K (sub-)chunk
  long long x_img_stride, y_img_stride;
  long long M;             // rows = images * Mw
  int Mw, N, nk, ldw;
  int k_valid;             // rows of dW that exist (<= nk*BKE); the rest of the last chunk is channel padding
  int steps_per_split;     // 32-row steps per blockIdx.y
  int ablate;              // dev diagnostics (RGP_WG_ABLATE): 1 = no reads/MFMA, 2 = no in-loop DMA, 4 = no barrier; results are garbage
  // Several problems of ONE geometry in one launch (gridDim.z = nz): problem z reads its images at X + zx[z] / dY + zy[z]
  // (bytes) and accumulates into dW + zw[z] (elements).  The out-of-range fall-back rows stay at X / dY themselves.  The
  // head's filter gradients are a dozen launches of a few hundred 32-row steps each: grouped, one launch fills the chip.
  int nz = 1;
  long long zx[8] = {0}, zy[8] = {0}, zw[8] = {0};
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

template <int V> struct WgInt { static constexpr int value = V; };

template <typename T, int G, int WNT>
__global__ __launch_bounds__(512, WNT == 4 && sizeof(T) == 2 ? 4 : 2) void wgrad_kernel(const WgradParams p) {
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
  unsigned xk[2];                                           // chunk-dependent byte offset (koff), fixed per block
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    int kc = kt * 4 + (wave >> 2) + 2 * u;
    if (kc >= p.nk) kc = p.nk - 1;                          // duplicate work, never stored
    const int elems_per_sub = BKE / G, ce = xc * (16 / ESZ);
    const int sub = ce / elems_per_sub, off = ce - sub * elems_per_sub;
    xk[u] = (unsigned)((p.koff[kc * G + sub] + off) * ESZ);
  }
  // dY: lane-load q = tid + 512 u covers tile row q / CPR, physical chunk q % CPR (the LDS image is lane-linear)
  int yr[NYL];
  unsigned yk[NYL], yz[NYL];
#pragma unroll
  for (int u = 0; u < NYL; ++u) {
    const int q = tid + 512 * u;
    yr[u] = q / CPR;
    const int yc = (q % CPR) ^ wg_swz_y<T>(yr[u]);
    yk[u] = (unsigned)(n0 * ESZ + yc * 16);
    yz[u] = (unsigned)((yc & 7) * 16);                      // zeros (the image's leading halo) for rows beyond the range
  }
  // scalar position of the next step to issue: image img0, row ml0 of it; a step advances it by 32 rows
  const int adv_img = 32 / p.Mw, adv_ml = 32 - adv_img * p.Mw;
  long long img0 = m_begin / p.Mw;
  int ml0 = (int)(m_begin - img0 * p.Mw);
  int rows_left = (int)(m_end - m_begin);                   // rows of this block's range not yet issued
  // table entries of the step to issue next (landed: see the vmcnt waits)
  unsigned tx, ty[NYL];
  auto load_tabs = [&]() {
    const int* xt = p.x_tab + ml0;
    const int* yt = p.y_tab + ml0;
    asm volatile("global_load_dword %0, %1, %2" : "=v"(tx) : "v"(xr * 4), "s"(xt) : "memory");
#pragma unroll
    for (int u = 0; u < NYL; ++u) asm volatile("global_load_dword %0, %1, %2" : "=v"(ty[u]) : "v"(yr[u] * 4), "s"(yt) : "memory");
  };
  auto tabs_landed = [&]() {                                 // behind a wait that covers them: re-define the registers
    asm volatile("" : "+v"(tx));
#pragma unroll
    for (int u = 0; u < NYL; ++u) asm volatile("" : "+v"(ty[u]));
  };

  // issue the LDS-DMA of the next step (its table entries are in tx / ty) and, in front of it, the table loads of the
  // step after
  auto issue = [&](int buf) {
    char* xb = smem + buf * S::STAGE;
    char* yb = xb + S::XB;
    const char* ximg = (const char*)p.X + p.zx[blockIdx.z] + img0 * p.x_img_stride * ESZ;
    const char* yimg = (const char*)p.dY + p.zy[blockIdx.z] + img0 * p.y_img_stride * ESZ;
    unsigned xo[2], yo[NYL];
#pragma unroll
    for (int u = 0; u < 2; ++u) xo[u] = tx + xk[u];
#pragma unroll
    for (int u = 0; u < NYL; ++u) yo[u] = ty[u] + yk[u];
    const int rl = rows_left;
    rows_left -= 32;
    img0 += adv_img;
    ml0 += adv_ml;
    if (ml0 >= p.Mw) { ml0 -= p.Mw; ++img0; }
    load_tabs();
    auto dma = [&](const char* const (&xs)[2], const char* const (&ys)[NYL]) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)xs[u],
                                         (__attribute__((address_space(3))) void*)(xb + ((wave >> 2) + 2 * u) * 4096 + (wave & 3) * 1024),
                                         16, 0, 0);
#pragma unroll
      for (int u = 0; u < NYL; ++u)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ys[u],
                                         (__attribute__((address_space(3))) void*)(yb + wave * 1024 + u * 8192), 16, 0, 0);
    };
    const char *xs[2], *ys[NYL];
    if (rl >= 32) {                                          // scalar image base + 32-bit lane offset
#pragma unroll
      for (int u = 0; u < 2; ++u) xs[u] = ximg + xo[u];
#pragma unroll
      for (int u = 0; u < NYL; ++u) ys[u] = yimg + yo[u];
      dma(xs, ys);
    } else {
      // the range's last (partial) step and the ones beyond it (issued for a uniform vmcnt arithmetic): rows past the
      // end read the zeros at the start of dY against finite values of X
#pragma unroll
      for (int u = 0; u < 2; ++u) xs[u] = xr < rl ? ximg + xo[u] : (const char*)p.X;
#pragma unroll
      for (int u = 0; u < NYL; ++u) ys[u] = yr[u] < rl ? yimg + yo[u] : (const char*)p.dY + yz[u];
      dma(xs, ys);
      asm volatile("; partial step" ::: "memory");           // keeps the two arms apart: merged, the full steps lose the scalar base
    }
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
  // per-thread LDS read addresses of ring stage 0 (stage 1 = + STAGE as an immediate, stage 2 has its own registers:
  // 2 * STAGE does not fit the 16-bit offset field of the 256-wide tile)
  const int rq = fcol >> 2, rp = fcol & 3;
  const int rrow = 8 * g + rq;                                 // rows rrow and rrow+4 share the swizzle
  unsigned xa0[4], ya0[WNT], xa2[4], ya2[WNT];
  {
    const int sx = (wg_swz_x<T>(rrow) >> 1), sy = (wg_swz_y<T>(rrow) >> 1);
    const unsigned xa = lds_base + wk * 4096 + rrow * 128 + rp * 8;
    const unsigned ya = lds_base + S::XB + rrow * YROW + rp * 8;
#pragma unroll
    for (int k = 0; k < 4; ++k) { xa0[k] = xa + ((k ^ sx) * 32); xa2[k] = xa0[k] + 2 * S::STAGE; }
#pragma unroll
    for (int j = 0; j < WNT; ++j) { ya0[j] = ya + (((wn * WNT + j) ^ sy) * 32); ya2[j] = ya0[j] + 2 * S::STAGE; }
  }
  auto compute = [&](auto BUF) {
    constexpr int buf = decltype(BUF)::value;
    if constexpr (ESZ == 2) {
      // The transposing reads are written as inline asm: issued through the builtin, the compiler orders
      // every LDS read after ALL outstanding LDS-DMA (s_waitcnt vmcnt(0)), which drains the two tiles in
      // flight and serialises the ring.  The asm carries its own lgkmcnt wait; A's registers are consumed
      // only by MFMAs that also need B's, so one wait at the end of the last block covers them all.
      static_assert(CI == 4, "bf16 chunk = 4 x 16 filter rows");
      constexpr int OB = buf == 1 ? S::STAGE : 0;
      const unsigned (&xa)[4] = buf == 2 ? xa2 : xa0;
      const unsigned (&ya)[WNT] = buf == 2 ? ya2 : ya0;
      i32x2 al[4], ah[4], bl[WNT], bh[WNT];
      asm volatile(
          "ds_read_b64_tr_b16 %0, %8 offset:%12\n\tds_read_b64_tr_b16 %1, %8 offset:%13\n\t"
          "ds_read_b64_tr_b16 %2, %9 offset:%12\n\tds_read_b64_tr_b16 %3, %9 offset:%13\n\t"
          "ds_read_b64_tr_b16 %4, %10 offset:%12\n\tds_read_b64_tr_b16 %5, %10 offset:%13\n\t"
          "ds_read_b64_tr_b16 %6, %11 offset:%12\n\tds_read_b64_tr_b16 %7, %11 offset:%13"
          : "=&v"(al[0]), "=&v"(ah[0]), "=&v"(al[1]), "=&v"(ah[1]), "=&v"(al[2]), "=&v"(ah[2]), "=&v"(al[3]), "=&v"(ah[3])
          : "v"(xa[0]), "v"(xa[1]), "v"(xa[2]), "v"(xa[3]), "n"(OB), "n"(OB + 512)
          : "memory");
      // rows rrow+4 of the dY tile are 4*YROW bytes further: 1024 (BN 128) or 2048 (BN 256)
#pragma unroll
      for (int jb = 0; jb < WNT; jb += 4)
        asm volatile(
            "ds_read_b64_tr_b16 %0, %8 offset:%12\n\tds_read_b64_tr_b16 %1, %8 offset:%13\n\t"
            "ds_read_b64_tr_b16 %2, %9 offset:%12\n\tds_read_b64_tr_b16 %3, %9 offset:%13\n\t"
            "ds_read_b64_tr_b16 %4, %10 offset:%12\n\tds_read_b64_tr_b16 %5, %10 offset:%13\n\t"
            "ds_read_b64_tr_b16 %6, %11 offset:%12\n\tds_read_b64_tr_b16 %7, %11 offset:%13"
            : "=&v"(bl[jb]), "=&v"(bh[jb]), "=&v"(bl[jb + 1]), "=&v"(bh[jb + 1]), "=&v"(bl[jb + 2]), "=&v"(bh[jb + 2]), "=&v"(bl[jb + 3]),
              "=&v"(bh[jb + 3])
            : "v"(ya[jb]), "v"(ya[jb + 1]), "v"(ya[jb + 2]), "v"(ya[jb + 3]), "n"(OB), "n"(OB + 4 * YROW)
            : "memory");
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
      const char* xb = smem + buf * S::STAGE + wk * 4096;
      const char* yb = smem + buf * S::STAGE + S::XB;
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

  // ---- 3-stage ring: steps s+1 and s+2 are in flight while step s is consumed.  Steps beyond the
  // block's range are issued too (they fetch zeros), so the vmcnt arithmetic is uniform. ----
  load_tabs();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  tabs_landed();
  issue(0);                                                         // queue: T(1) DMA(0)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STEP) : "memory");   // T(1) landed
  tabs_landed();
  issue(1);                                                         // queue: DMA(0) T(2) DMA(1)
  auto step = [&](auto BUF) {
    constexpr int buf = decltype(BUF)::value;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STEP) : "memory");   // all but DMA(s+1): DMA(s) and T(s+2) landed
    tabs_landed();
    if (!(p.ablate & 4)) __builtin_amdgcn_s_barrier();           // step s landed for every wave; stage (s+2)%3 is free
    if (!(p.ablate & 2)) issue((buf + 2) % 3);
    if (!(p.ablate & 1) && ci_n > 0) compute(BUF);
  };
  // written out over the ring's three stages: every LDS address of the loop is a register + an immediate
  for (int s = 0; s < nsteps;) {
    step(WgInt<0>{});
    if (++s >= nsteps) break;
    step(WgInt<1>{});
    if (++s >= nsteps) break;
    step(WgInt<2>{});
    ++s;
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
            if (k < p.k_valid) atomicAdd(p.dW + p.zw[blockIdx.z] + k * p.ldw + n, acc[i][j][r]);
          }
        }
      }
  }
}

}  // namespace rgp
