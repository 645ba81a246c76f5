// Filter gradients of conv2a, conv3a and conv3b (3x3x3, pad 1, 56 x 56 / 28 x 28 planes) for gfx950, bf16: the patch
// scheme of conv_patch.hip.h applied to  dW[tap][c][n] += sum_m X[pos(m) + tap][c] dY[pos(m)][n]
// (tf.gradients w.r.t. the filters of feature_extration.prototxt:67-173, base.py:278-281).
//
// wgrad.hip.h gives every block 4 K-chunks (for these layers: 1-4 taps x 64 channels) and streams X and dY rows for
// them from L2: the 27 taps re-read the same input pixels 27 times and the n_kt K-tiles of a layer re-read dY n_kt
// times -- 11.4 (128-wide tile) or 7.6 (256-wide) KB of LDS-DMA per MFLOP, and the kernel is ingest-bound (matrix pipe
// 48-50 % busy at 2.2-2.3 GHz).  Here a block owns ONE slice of 32 input channels x 64 output channels and ALL 27 taps:
//
//  * it walks columns (window, 4 image rows) through the planes z like conv1a.hip.h: a ring of input plane slabs
//    (6 rows x (W+2) pixels x 64 B) and a double-buffered dY slab (4 rows x (W+2) pixels x 128 B) in LDS; per group of
//    224 positions (one plane at 56 x 56, two at 28 x 28) one new slab of each is fetched while the group computes:
//    2.2 KB of LDS-DMA per MFLOP.
//  * the 54 (tap, 16-channel tile) units are dealt to the 8 waves (7,7,7,7,7,7,6,6); a unit's accumulators are 4 tiles
//    of 16 x 16 (64 output channels): 112 registers per wave, kept for the whole column range of the block and added
//    to dW with fp32 atomics at the end (27 x 32 x 64 floats per block).
//  * fragments: both operands have the reduction index (the position) on the strided axis, so they are read with
//    ds_read_b64_tr_b16 (wgrad.hip.h): a lane supplies the address of ITS row, so the 8 positions of a k-group can be
//    any pixels -- the per-lane addresses of the 14 (step, half) position groups are tabulated once; a tap is a
//    wave-uniform offset.  The dY slab's 32-byte segments are XOR-swizzled per pixel on the DMA source side (8 rows of
//    a read fall in 8 different bank groups); the X slab is not (a tap shifts the pixel, so the swizzle could not be
//    an immediate): its reads are at most 2-way conflicted.
#pragma once
#include "igemm.hip.h"

namespace rgp {

struct WgradPatchParams {
  const bf16_t* x;     // layer input [n][D+2][W+2][W+2][CIN], halo-padded
  const bf16_t* dy;    // gradient w.r.t. the conv output before pooling [n][D+2][W+2][W+2][COUT], halos zero
  float* dw;           // [27 * CIN][COUT] fp32 (DHWIO), accumulated with atomics
  int n_windows;
  int splits;          // column ranges per (channel slice, output slice)
};

template <int CIN, int COUT, int HW, int DEPTH> struct WgpCfg {
  static constexpr int WP = HW + 2;
  static constexpr int ZS = HW == 56 ? 1 : 2;             // planes per group: 224 positions = 7 steps of 32 either way
  static constexpr int NG = DEPTH / ZS;                   // groups per column
  static constexpr int CS = CIN / 32, NS = COUT / 64;     // channel slices, output slices
  static constexpr int COLS = HW / 4;                     // columns per window
  static constexpr int XI = (6 * WP + 15) / 16;           // LDS-DMA instructions per input plane slab (16 pixels x 64 B)
  static constexpr int XBUF = XI * 1024;
  static constexpr int NXB = 2 * ZS + 2;                  // ring: ZS + 2 planes in use, ZS in flight
  static constexpr int DI = 4 * WP / 8;                   // instructions per dY plane slab (8 pixels x 128 B)
  static constexpr int DYPLANE = DI * 1024, DYBUF = ZS * DYPLANE;
  static constexpr int DY_OFF = NXB * XBUF;
  static constexpr int SMEM = DY_OFF + 2 * DYBUF;
  static_assert(ZS * 4 * HW == 224 && (4 * WP) % 8 == 0 && DEPTH % ZS == 0 && CIN % 32 == 0 && COUT % 64 == 0, "group shape");
  static_assert(SMEM <= 160 * 1024, "LDS budget");
};

typedef int i32x2_wg __attribute__((ext_vector_type(2)));
typedef int i32x4_wg __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ i32x2_wg wgp_tr_read(unsigned addr) {
  i32x2_wg v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}

template <int CIN, int COUT, int HW, int DEPTH>
static __global__ __launch_bounds__(512) void wgrad_patch_bf16_kernel(const WgradPatchParams p) {
  using C = WgpCfg<CIN, COUT, HW, DEPTH>;
  constexpr int WP = C::WP, ZS = C::ZS;
  extern __shared__ __attribute__((aligned(16))) char wp_smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)wp_smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fcol = lane & 15, g = lane >> 4, q = fcol >> 2, pp = fcol & 3;
  if (lds0 & 127u) __builtin_trap();                          // the dY segment swizzle is an XOR on address bits 5-6

  // block -> (channel slice cs, output slice ns, column range).  Consecutive workgroup ids sit on different XCDs: the
  // CS x NS blocks of one column range are put on ONE XCD (id & 7), so that every input pixel and dY pixel -- of which
  // each of them reads a different 64 / 128 bytes -- comes through that XCD's L2 once instead of CS x NS times from HBM
  // (splits is a multiple of 8)
  const int xcd = blockIdx.x & 7, wx = blockIdx.x >> 3;
  const int combo = wx % (C::CS * C::NS), split = (wx / (C::CS * C::NS)) * 8 + xcd;
  const int cs = combo % C::CS, ns = combo / C::CS;
  const int ncols = p.n_windows * C::COLS;
  const int cper = (ncols + p.splits - 1) / p.splits;
  int col = split * cper;
  const int col_end = min(col + cper, ncols);
  if (col >= col_end) return;

  // ---- per-lane position tables: (step s, half h) -> position m = 32 s + 8 g + 4 h + q of the group ----
  unsigned xa[7][2], ya[7][2];
  unsigned zlm = 0;                                           // bit 2 s + h: the position lies in the group's second plane
#pragma unroll
  for (int s = 0; s < 7; ++s)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int m = 32 * s + 8 * g + 4 * h + q;
      const int zl = m / (4 * HW), rr = (m / HW) % 4, xx = m % HW;
      xa[s][h] = (unsigned)((rr * WP + xx) * 64 + pp * 8);
      const int pi = zl * (4 * WP) + rr * WP + xx + 1;        // pixel of the dY slab (its rows start at x = -1)
      ya[s][h] = (unsigned)(pi * 128 + pp * 8 + ((((pi >> 1) & 1) | (((pi >> 3) & 1) << 1)) << 5));
      if (zl) zlm |= 1u << (2 * s + h);
    }
  // this wave's units: channel tile ct, taps (wave >> 1) + 4 i
  const int ct = wave & 1, tap0 = wave >> 1;

  // ---- DMA: instruction t of a group's fetch list = X plane slabs (ZS x XI), then dY plane slabs (ZS x DI) ----
  const int xpix = lane >> 2, xchk = lane & 3;                // X: 16 pixels x 4 chunks of 16 B
  // dY: 8 pixels x 8 chunks per instruction; the 32-byte segment seg of slab pixel pi is stored at seg ^ s5(pi),
  // s5 = ((pi >> 1) & 1) | (((pi >> 3) & 1) << 1): the 8 rows a 32-lane half of a transposing read touches (pi .. pi+3
  // and pi+8 .. pi+11) then fall in 8 different 32-byte bank groups.  pi >> 3 = the instruction's index in the slab.
  const int ypix = lane >> 3;
  const int ychk0 = (lane & 7) ^ (2 * ((ypix >> 1) & 1));
  auto x_plane_src = [&](int c, int pz) {                     // column c = (window, row quarter), input plane pz (halo coords)
    const int n = c / C::COLS, yq = c - n * C::COLS;
    return (const char*)(p.x + (((long long)n * (DEPTH + 2) + pz) * WP + 4 * yq) * (long long)(WP * CIN) + cs * 32);
  };
  auto y_plane_src = [&](int c, int z) {                      // dY of conv plane z: rows 4 yq + 1 .. + 4, from x = -1
    const int n = c / C::COLS, yq = c - n * C::COLS;
    return (const char*)(p.dy + (((long long)n * (DEPTH + 2) + z + 1) * WP + 4 * yq + 1) * (long long)(WP * COUT) + ns * 64);
  };
  auto dma_x = [&](const char* src, int slot, int j) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long long)(j * 16 + xpix) * (CIN * 2) + xchk * 16),
                                     (__attribute__((address_space(3))) void*)(wp_smem + slot * C::XBUF + j * 1024), 16, 0, 0);
  };
  auto dma_y = [&](const char* src, int buf, int zl, int j) {
    const int ychk = ychk0 ^ (4 * ((zl * C::DI + j) & 1));
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long long)(j * 8 + ypix) * (COUT * 2) + ychk * 16),
                                     (__attribute__((address_space(3))) void*)(wp_smem + C::DY_OFF + buf * C::DYBUF + zl * C::DYPLANE + j * 1024),
                                     16, 0, 0);
  };
  // fetch input planes [pz0, pz0 + np) and the dY slab of group gq of column c; the instructions are dealt round-robin
  auto fetch = [&](int c, int pz0, int np, int gq) {
    const int nx = np * C::XI, total = nx + ZS * C::DI;
    for (int t = wave; t < total; t += 8) {
      if (t < nx) {
        const int k = t / C::XI, j = t - k * C::XI;
        dma_x(x_plane_src(c, pz0 + k), (pz0 + k) % C::NXB, j);
      } else {
        const int u = t - nx, zl = u / C::DI, j = u - zl * C::DI;
        dma_y(y_plane_src(c, gq * ZS + zl), gq & 1, zl, j);
      }
    }
  };

  f32x4 acc[7][4];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (; col < col_end; ++col) {
    // column prologue: planes 0 .. ZS + 1 and the first dY slab (not overlapped: the ring slots of the previous
    // column's last planes are still in use until its last group is done)
    fetch(col, 0, ZS + 2, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma clang loop unroll(disable)
    for (int gq = 0; gq < C::NG; ++gq) {
      const int z0 = gq * ZS;
      if (gq + 1 < C::NG) fetch(col, z0 + ZS + 2, ZS, gq + 1);   // in flight while this group computes
      // wave-uniform address parts of this wave's units: plane slab of (plane z0 + zl + kz), tap offset, channel tile
      unsigned sa[7][2];
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        int tap = tap0 + 4 * i;
        if (tap > 26) tap = 26;                               // the unit does not exist: computed, never stored
        const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
        const unsigned off = lds0 + (unsigned)((ky * WP + kx) * 64 + ct * 32);
#pragma unroll
        for (int zl = 0; zl < 2; ++zl) sa[i][zl] = off + (unsigned)(((z0 + zl + kz) % C::NXB) * C::XBUF);
      }
      const unsigned yb = lds0 + C::DY_OFF + (gq & 1) * C::DYBUF;
      // fragments of step s: 8 dY reads (4 output tiles x 2 halves) and 14 X reads (7 units x 2 halves); the reads of step
      // s + 1 are issued in front of the MFMAs of step s (two register sets)
      auto reads = [&](int s, i32x2_wg (&bl)[4], i32x2_wg (&bh)[4], i32x2_wg (&al)[7], i32x2_wg (&ah)[7]) {
        const unsigned y0 = yb + ya[s][0], y1 = yb + ya[s][1];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bl[j] = wgp_tr_read(y0 ^ (unsigned)(j << 5));
          bh[j] = wgp_tr_read(y1 ^ (unsigned)(j << 5));
        }
        const bool z0b = ZS == 2 && ((zlm >> (2 * s)) & 1u), z1b = ZS == 2 && ((zlm >> (2 * s + 1)) & 1u);
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          al[i] = wgp_tr_read(xa[s][0] + (z0b ? sa[i][1] : sa[i][0]));
          ah[i] = wgp_tr_read(xa[s][1] + (z1b ? sa[i][1] : sa[i][0]));
        }
      };
      auto landed = [&](i32x2_wg (&bl)[4], i32x2_wg (&bh)[4], i32x2_wg (&al)[7], i32x2_wg (&ah)[7]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(bl[j]), "+v"(bh[j]));
#pragma unroll
        for (int i = 0; i < 7; ++i) asm volatile("" : "+v"(al[i]), "+v"(ah[i]));
      };
      auto mmas = [&](const i32x2_wg (&bl)[4], const i32x2_wg (&bh)[4], const i32x2_wg (&al)[7], const i32x2_wg (&ah)[7]) {
        f32x4 b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = __builtin_bit_cast(f32x4, (i32x4_wg){bl[j][0], bl[j][1], bh[j][0], bh[j][1]});
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          const f32x4 a = __builtin_bit_cast(f32x4, (i32x4_wg){al[i][0], al[i][1], ah[i][0], ah[i][1]});
#pragma unroll
          for (int j = 0; j < 4; ++j) Mma<bf16_t>::step(acc[i][j], a, b[j]);
        }
      };
      i32x2_wg bl0[4], bh0[4], al0[7], ah0[7], bl1[4], bh1[4], al1[7], ah1[7];
      reads(0, bl0, bh0, al0, ah0);
#pragma unroll
      for (int s = 0; s < 7; s += 2) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        landed(bl0, bh0, al0, ah0);
        if (s + 1 < 7) reads(s + 1, bl1, bh1, al1, ah1);
        __builtin_amdgcn_sched_barrier(0);
        mmas(bl0, bh0, al0, ah0);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < 7) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          landed(bl1, bh1, al1, ah1);
          if (s + 2 < 7) reads(s + 2, bl0, bh0, al0, ah0);
          __builtin_amdgcn_sched_barrier(0);
          mmas(bl1, bh1, al1, ah1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the next group's slabs have landed ...
      __builtin_amdgcn_s_barrier();                           // ... for everybody, and everybody is done with this group's
    }
  }

  // ---- dW[(tap * CIN + 32 cs + 16 ct + 4 g + r) * COUT + 64 ns + 16 j + fcol] += D[row 4 g + r][col fcol] ----
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int tap = tap0 + 4 * i;
    if (tap < 27) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          atomicAdd(p.dw + (long long)(tap * CIN + cs * 32 + ct * 16 + 4 * g + r) * COUT + ns * 64 + j * 16 + fcol, acc[i][j][r]);
    }
  }
}

}  // namespace rgp
