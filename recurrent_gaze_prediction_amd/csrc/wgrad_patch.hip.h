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
//    2.2 KB of LDS-DMA per MFLOP.  At 28 x 28 the ring has 8 slots, so the first four planes of the NEXT column are
//    fetched under the last group of this one (no exposed prologue); at 56 x 56 LDS holds 4 slots and a column
//    (16 groups) starts with an exposed fetch.
//  * the 54 (tap, 16-channel tile) units are dealt to the 8 waves (7,7,7,7,7,7,6,6); a unit's accumulators are 4 tiles
//    of 16 x 16 (64 output channels): 112 registers per wave, kept for the whole column range of the block and added
//    to dW with fp32 atomics at the end (27 x 32 x 64 floats per block).
//  * fragments: both operands have the reduction index (the position) on the strided axis, so they are read with
//    ds_read_b64_tr_b16 (wgrad.hip.h): a lane supplies the address of ITS position.  The MFMA's k slot (step s, lane
//    group g, half h, q) is mapped to image row g, column 8 s + 4 h + q (28 x 28: plane and column from 2 s + h): a
//    lane's row and q never change, so every read address is  base register + IMMEDIATE  -- one base per unit for X
//    (lane constant + the tap's wave-uniform offset + ring slot, recomputed per group: 2 VALU per unit), four for dY.
//    No per-read address arithmetic (the first version spent 154 VALU per 196 MFMAs on it).
//  * banks: a 32-lane half of a read touches 2 rows x 4 consecutive pixels x 32 B.  X (64-byte pixel slices, row pitch
//    = 128 mod 256): the 32-byte half of a pixel is stored at ct ^ (slab row & 1); dY (128-byte pixels): its 32-byte
//    segment seg at seg ^ (((x >> 1) & 1) | ((row & 1) << 1)).  Both swizzles are applied on the DMA's SOURCE side (a
//    lane picks which 16 bytes it fetches) and are lane constants (XOR with the tap's row parity for X) on the read
//    side: every read covers the 64 banks exactly once.
//  * the step-0 fragments of the next group are requested right after the barrier that publishes its slabs, in front
//    of the last step's MFMAs of the current group (the two fragment register sets alternate), so the LDS latency at
//    a group boundary is covered.
//
// 14 x 14 planes (conv4a, conv4b; round 5).  Rows of 14 pixels do not fill the 32 k slots of a step with a (row, column)
// product: on the 16-pixel padded row 1/8 of the MFMAs would multiply halo (a costed and rejected variant of round 2).
// The slots are a product of FACTORS of the extents instead: 784 = 2^4 7^2 positions per window hold 16 lane positions
// (x parity, y parity, all 4 planes) and a PAIR of windows supplies the other factor of 2 -- slot (lane group g, half h,
// q) = x parity q & 1, y parity q >> 1, window g & 1, plane (g >> 1) + 2 h -- while the 7 x 7 (x pair, y pair) steps
// carry the odd factors: a column = a window pair, a group = one output row pair of it (7 steps over the x pairs), no
// slot multiplies padding.  A lane's position moves by (2 pixels, 0) per step and by 2 planes per half: immediates.
//  * input ring: 8 ROW slots [window 2][plane 6][16 pixels][64 B] (window pitch = 128 mod 256, slot = 0 mod 256; the
//    two halo planes are zeroed once and never fetched); a group reads padded rows 2 ys .. 2 ys + 3 and the two rows the
//    next-but-one group adds are in flight meanwhile; a new column's four rows go to the four slots the old column has
//    left (16 rows per column = 2 turns of the ring): no exposed prologue.  dY slab of a group: [y parity][plane][x]
//    [window][128 B], double-buffered.  1.7 KB of LDS-DMA per MFLOP.
//  * banks: the 8 positions of a 32-lane half differ in (x parity, y parity, window): 64 B, the 32-byte half swizzle
//    (stored at ct ^ row parity) and the window pitch put them in the 8 different 32-byte groups of a 256-byte bank row;
//    dY: the 2-bit segment swizzle (x parity | y parity << 1) and the 128-byte window interleave.
#pragma once
#include <type_traits>

#include "igemm.hip.h"

namespace rgp {

struct WgradPatchParams {
  const bf16_t* x;     // layer input [n][D+2][W+2][W+2][CIN], halo-padded
  const bf16_t* dy;    // gradient w.r.t. the conv output before pooling [n][D+2][W+2][W+2][COUT], halos zero
  float* dw;           // [27 * CIN][COUT] fp32 (DHWIO), accumulated with atomics
  float* db;           // optional [COUT] fp32 bias gradient (sum of dy over all positions), accumulated with atomics:
                       // waves 6 and 7 own 6 units of the 54, and wave 6 of the blocks of channel slice 0 runs its seventh
                       // slot as the unit "all-ones x dY" -- the column sums, on MFMAs the wave issued anyway (a separate
                       // colsum pass re-read the whole gradient image: 0.3 ms for conv3a at 256 windows)
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
  static constexpr int DI = 4 * WP / 8;                   // instructions per dY plane slab (8 pixels x 128 B)
  static constexpr int DYPLANE = DI * 1024, DYBUF = ZS * DYPLANE;
  // ring of input plane slabs: ZS + 2 planes in use, ZS (next group of the column) or ZS + 2 (first group of the
  // next column) in flight
  static constexpr bool SEAMLESS = (2 * ZS + 4) * XBUF + 2 * DYBUF <= 160 * 1024;
  static constexpr int NXB = SEAMLESS ? 2 * ZS + 4 : 2 * ZS + 2;
  static constexpr int DY_OFF = NXB * XBUF;
  static constexpr int SMEM = DY_OFF + 2 * DYBUF;
  static_assert(ZS * 4 * HW == 224 && (4 * WP) % 8 == 0 && DEPTH % (2 * ZS) == 0 && CIN % 32 == 0 && COUT % 64 == 0, "group shape");
  static_assert((WP * 64) % 256 == 128 && (WP * 128) % 256 == 0, "row pitches assumed by the bank swizzles");
  static_assert(SMEM <= 160 * 1024, "LDS budget");
};

// 14 x 14 x 4 layers: a column = a pair of windows, a group = one pair of output rows of both (header)
template <int CIN, int COUT> struct Wgp14Cfg {
  static constexpr int WP = 16;                           // padded row
  static constexpr int ZS = 1;                            // (one read base per unit)
  static constexpr int NG = 7;                            // groups per column: output row pairs
  static constexpr int CS = CIN / 32, NS = COUT / 64;
  static constexpr int PLANE = 16 * 64;                   // one (window, plane) row segment: 16 pixels x 64 B = one LDS-DMA instruction
  static constexpr int WPITCH = 6 * PLANE + 128;          // window pitch inside a row slot: = 128 (mod 256)
  static constexpr int XBUF = 2 * WPITCH;                 // a row slot: 12 544 B = 0 (mod 256)
  static constexpr int NXB = 8;                           // ring of row slots (a column's 16 padded rows = 2 turns)
  static constexpr int XI = 8;                            // LDS-DMA instructions per row: 2 windows x 4 data planes
  static constexpr int DY_X = 256, DY_Z = 14 * DY_X, DY_Y = 4 * DY_Z;   // dY slab pitches: [y parity][plane][x][window][128 B]
  static constexpr int DYBUF = 2 * DY_Y;                  // 28 672 B
  static constexpr int DI = DYBUF / 1024;                 // 28 instructions of 8 pixels x 128 B
  static constexpr int DY_OFF = NXB * XBUF;
  static constexpr int SMEM = DY_OFF + 2 * DYBUF;         // 157 696 B
  static constexpr bool SEAMLESS = true;
  static_assert(CIN % 32 == 0 && COUT % 64 == 0, "slices");
  static_assert(WPITCH % 256 == 128 && XBUF % 256 == 0 && DY_OFF % 256 == 0, "pitches assumed by the bank argument");
  static_assert(SMEM <= 160 * 1024, "LDS budget");
};

typedef int i32x2_wg __attribute__((ext_vector_type(2)));
typedef int i32x4_wg __attribute__((ext_vector_type(4)));

template <int OFF>
static __device__ __forceinline__ i32x2_wg wgp_tr_read(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
  i32x2_wg v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

// accumulate in place: the tied operand keeps each accumulator in ONE register quad for the whole kernel (left to the
// register allocator, the results rotate through the fragment registers and the loop-carried state needs spills).
// An accumulator is touched once per 28 MFMAs and read by nothing else inside the loop: no software wait states needed.
static __device__ __forceinline__ void wgp_mfma(f32x4& acc, const i32x4_wg& a, const i32x4_wg& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

struct WgpFrags {                                             // fragments of one step: dY (4 output tiles) and X (7 units), two halves
  i32x2_wg bl[4], bh[4], al[7], ah[7];
};

template <int CIN, int COUT, int HW, int DEPTH>
static __global__ __launch_bounds__(512) void wgrad_patch_bf16_kernel(const WgradPatchParams p) {
  constexpr bool P14 = HW == 14;                              // the window-pair scheme of the 14 x 14 x 4 layers (header)
  static_assert(!P14 || DEPTH == 4, "14 x 14 planes come 4 deep");
  using C = std::conditional_t<P14, Wgp14Cfg<CIN, COUT>, WgpCfg<CIN, COUT, HW, DEPTH>>;
  constexpr int WP = C::WP, ZS = C::ZS, NXB = C::NXB;
  extern __shared__ __attribute__((aligned(16))) char wp_smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)wp_smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fcol = lane & 15, g = lane >> 4, q = fcol >> 2, pp = fcol & 3;
  if (lds0 & 255u) __builtin_trap();                          // the swizzles are XORs on address bits 5-6 of 256-byte aligned rows

  // block -> (channel slice cs, output slice ns, column range).  Consecutive workgroup ids sit on different XCDs: the
  // CS x NS blocks of one column range are put on ONE XCD (id & 7), so that every input pixel and dY pixel -- of which
  // each of them reads a different 64 / 128 bytes -- comes through that XCD's L2 once instead of CS x NS times from HBM
  // (splits is a multiple of 8)
  const int xcd = blockIdx.x & 7, wx = blockIdx.x >> 3;
  const int combo = wx % (C::CS * C::NS), split = (wx / (C::CS * C::NS)) * 8 + xcd;
  const int cs = combo % C::CS, ns = combo / C::CS;
  int ncols;
  if constexpr (P14) ncols = (p.n_windows + 1) >> 1;          // window pairs (the last one may hold a single window)
  else ncols = p.n_windows * C::COLS;
  const int cper = (ncols + p.splits - 1) / p.splits;
  const int col0 = split * cper;
  const int col_end = min(col0 + cper, ncols);
  if (col0 >= col_end) return;
  const int n_groups = (col_end - col0) * C::NG;              // even where NG is (56 x 56, 28 x 28)

  // this wave's units: channel tile ct, taps (wave >> 1) + 4 i
  const int ct = wave & 1, tap0 = wave >> 1;
  const bool bias_unit = p.db != nullptr && wave == 6 && cs == 0;      // its seventh unit (tap 27) does not exist: see WgradPatchParams::db

  // ---- read-side lane constants.  k slot (s, g, h, q) of a step = image row g of the column, x = 8 s + 4 h + q
  // (28 x 28: u = 2 s + h, plane u / 7, x = 4 (u % 7) + q) ----
  // 14 x 14: slot (g, h, q) = x parity q & 1, y parity q >> 1, window g & 1, plane (g >> 1) + 2 h; step s = x pair
  const int l_x = q & 1, l_y = q >> 1, l_w = g & 1, l_z = g >> 1;
  unsigned xl0, swl, yl[4];                                                                // X: tap rows of even ky; odd ky: ^ 32
  if constexpr (P14) {
    xl0 = (unsigned)(l_w * C::WPITCH + l_z * C::PLANE + l_x * 64 + pp * 8 + 32 * ((ct ^ l_y) & 1));
    swl = (unsigned)(l_x | (l_y << 1));
#pragma unroll
    for (int j = 0; j < 4; ++j)
      yl[j] = (unsigned)(l_y * C::DY_Y + l_z * C::DY_Z + l_x * C::DY_X + l_w * 128 + pp * 8) + (((unsigned)j ^ swl) << 5);
  } else {
    xl0 = (unsigned)((g * WP + q) * 64 + pp * 8 + 32 * ((ct ^ g) & 1));
    swl = (unsigned)(((q >> 1) & 1) | ((g & 1) << 1));
#pragma unroll
    for (int j = 0; j < 4; ++j) yl[j] = (unsigned)((g * WP + q + 1) * 128 + pp * 8) + (((unsigned)j ^ swl) << 5);   // dY: output tile j -> segment j ^ swl
  }

  // ---- DMA: instruction t of a group's fetch list = X plane slabs (np x XI), then dY plane slabs (ZS x DI) ----
  const int xpix = lane >> 2, xchk = lane & 3;                // X: 16 pixels x 4 chunks of 16 B
  const int ypix = lane >> 3, ychk = lane & 7;                // dY: 8 pixels x 8 chunks
  // fetch what group (column c = col0 + kc, gq) needs and no earlier group of its column has fetched: input planes
  // 0 .. ZS + 1 (gq = 0) or the ZS planes behind them, and its dY slab; the instructions are dealt round-robin.
  // Input plane pz of the block's kc-th column lives in ring slot (kc (DEPTH + 2) + pz) % NXB.
  // 14 x 14: one instruction = one (window, data plane) row segment of 16 pixels x 64 B, or 8 dY pixels x 128 B; a group
  // adds padded rows 2 gq + 2, 2 gq + 3 (the column's first group: rows 0 .. 3); row r of the block's kc-th column lives
  // in ring slot (16 kc + r) % 8
  auto fetch = [&](int kc, int gq) {
    if constexpr (P14) {
      const int c = col0 + kc;
      const int win0 = 2 * c;
      const bool two = win0 + 1 < p.n_windows;                // the pair's second window exists
      const int r0 = gq == 0 ? 0 : 2 * gq + 2, nr = gq == 0 ? 4 : 2;      // padded rows this group adds
      const int nx = nr * C::XI, total = nx + C::DI;
      for (int t = wave; t < total; t += 8) {
        if (t < nx) {
          const int k = t >> 3, w = (t >> 2) & 1, zp = (t & 3) + 1, r = r0 + k;
          // a missing second window is fetched from the first: finite values against its all-zero dY
          const bf16_t* src = p.x + ((((long long)(win0 + (two ? w : 0)) * 6 + zp) * 16 + r) * 16) * (long long)CIN + cs * 32;
          int lx = xpix, lc = xchk;
          asm volatile("" : "+v"(lx), "+v"(lc));
          // the 32-byte half h of a pixel is stored at h ^ (row parity): the lane picks its 16 bytes on the source side
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)((const char*)src + (unsigned)(lx * (CIN * 2) + (lc ^ (2 * (r & 1))) * 16)),
              (__attribute__((address_space(3))) void*)(wp_smem + ((kc * 16 + r) & 7) * C::XBUF + w * C::WPITCH + zp * C::PLANE), 16, 0, 0);
        } else {
          const int jj = t - nx;
          int ly = ypix;
          asm volatile("" : "+v"(ly));
          const int pi = jj * 8 + ly;                          // slab pixel [y parity][plane][x][window]
          const int w = pi & 1, t2 = pi >> 1, t3 = t2 / 14, x = t2 - t3 * 14, z = t3 & 3, yp = t3 >> 2;
          const int sw = (x & 1) | (yp << 1);
          const unsigned chunk = (unsigned)((ychk ^ (2 * sw)) * 16);
          const char* src;
          if (two || w == 0)
            src = (const char*)(p.dy + ((((long long)(win0 + w) * 6 + z + 1) * 16 + 2 * gq + yp + 1) * 16 + x + 1) * (long long)COUT + ns * 64) + chunk;
          else
            src = (const char*)(p.dy + (x + 1) * COUT + ns * 64) + chunk;        // zeros: row 0 of the first halo plane
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(wp_smem + C::DY_OFF + ((kc * C::NG + gq) & 1) * C::DYBUF + jj * 1024),
                                           16, 0, 0);
        }
      }
    } else {
      auto x_plane_src = [&](int c, int pz) {                     // column c = (window, row quarter), input plane pz (halo coords)
        const int n = c / C::COLS, yq = c - n * C::COLS;
        return (const char*)(p.x + (((long long)n * (DEPTH + 2) + pz) * WP + 4 * yq) * (long long)(WP * CIN) + cs * 32);
      };
      auto y_plane_src = [&](int c, int z) {                      // dY of conv plane z: rows 4 yq + 1 .. + 4, from x = -1
        const int n = c / C::COLS, yq = c - n * C::COLS;
        return (const char*)(p.dy + (((long long)n * (DEPTH + 2) + z + 1) * WP + 4 * yq + 1) * (long long)(WP * COUT) + ns * 64);
      };
      auto dma_x = [&](const char* src, int slot, int j) {
        int lx = xpix;
        asm volatile("" : "+v"(lx));                              // recompute the lane's offset per instruction (no hoisted tables)
        const int P = j * 16 + lx, row = P / WP;                  // slab pixel, slab row: half ct of the pixel goes to ct ^ (row & 1)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (unsigned)(P * (CIN * 2) + (xchk ^ (2 * (row & 1))) * 16)),
                                         (__attribute__((address_space(3))) void*)(wp_smem + slot * C::XBUF + j * 1024), 16, 0, 0);
      };
      auto dma_y = [&](const char* src, int buf, int zl, int j) {
        int ly = ypix;
        asm volatile("" : "+v"(ly));
        const int pi = j * 8 + ly, row = pi / WP, xs = pi - row * WP;         // slab pixel (x = xs - 1), slab row
        const int sw = (((xs - 1) >> 1) & 1) | ((row & 1) << 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (unsigned)(pi * (COUT * 2) + (ychk ^ (2 * sw)) * 16)),
                                         (__attribute__((address_space(3))) void*)(wp_smem + C::DY_OFF + buf * C::DYBUF + zl * C::DYPLANE + j * 1024),
                                         16, 0, 0);
      };
      const int c = col0 + kc;
      const int pz0 = gq == 0 ? 0 : gq * ZS + 2, np = gq == 0 ? ZS + 2 : ZS;
      const int nx = np * C::XI, total = nx + ZS * C::DI;
      for (int t = wave; t < total; t += 8) {
        if (t < nx) {
          const int k = t / C::XI, j = t - k * C::XI;
          dma_x(x_plane_src(c, pz0 + k), (kc * (DEPTH + 2) + pz0 + k) % NXB, j);
        } else {
          const int u = t - nx, zl = u / C::DI, j = u - zl * C::DI;
          dma_y(y_plane_src(c, gq * ZS + zl), (kc * C::NG + gq) & 1, zl, j);
        }
      }
    }
  };

  f32x4 acc[7][4];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // read bases of a group: X unit i (tap tap0 + 4 i), plane zl of the group; dY output tile j
  unsigned bx[7][ZS], by[4];
  auto bases = [&](int kc, int gq) {
    unsigned sa[4] = {0, 0, 0, 0};                            // 14 x 14: slot addresses of the group's padded rows 2 gq + k (scalars)
    if constexpr (P14) {
#pragma unroll
      for (int k = 0; k < 4; ++k) sa[k] = lds0 + (unsigned)(((kc * 16 + 2 * gq + k) & 7) * C::XBUF);
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      int tap = tap0 + 4 * i;
      if (tap > 26) tap = 26;                                 // the unit does not exist: computed, never stored
      const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
      unsigned lanepart = xl0;
      asm volatile("" : "+v"(lanepart));                      // one v_xor per unit and group instead of a hoisted register each
      lanepart ^= (unsigned)((ky & 1) << 5);
      if constexpr (P14) {
        // the lane's padded row is 2 gq + (y parity) + ky: one select between two scalar slot addresses
        const unsigned rowaddr = l_y ? sa[ky == 0 ? 1 : (ky == 1 ? 2 : 3)] : sa[ky == 0 ? 0 : (ky == 1 ? 1 : 2)];
        bx[i][0] = lanepart + (rowaddr + (unsigned)(kz * C::PLANE + kx * 64));
      } else {
        const unsigned off = lds0 + (unsigned)((ky * WP + kx) * 64);
#pragma unroll
        for (int zl = 0; zl < ZS; ++zl)
          bx[i][zl] = lanepart + (off + (unsigned)(((kc * (DEPTH + 2) + gq * ZS + zl + kz) % NXB) * C::XBUF));
      }
    }
    const unsigned yb = lds0 + C::DY_OFF + ((kc * C::NG + gq) & 1) * C::DYBUF;
#pragma unroll
    for (int j = 0; j < 4; ++j) by[j] = yl[j] + yb;
  };
  // fragment read r of step S (0-7: dY output tile r / 2, half r & 1; 8-21: X unit (r - 8) / 2, half r & 1)
  auto read1 = [&](auto S, int r, WgpFrags& f) {
    constexpr int s = decltype(S)::value;
    if (r >= 20 && bias_unit) return;                         // the bias wave's seventh fragment stays all ones (below)
    if constexpr (P14) {                                      // step s = x pair s; half h = planes + 2 h
      const int h = r & 1;
      if (r < 8) {
        const int j = r >> 1;
        if (h == 0) f.bl[j] = wgp_tr_read<s * 2 * C::DY_X>(by[j]);
        else f.bh[j] = wgp_tr_read<s * 2 * C::DY_X + 2 * C::DY_Z>(by[j]);
      } else {
        const int i = (r - 8) >> 1;
        if (h == 0) f.al[i] = wgp_tr_read<s * 128>(bx[i][0]);
        else f.ah[i] = wgp_tr_read<s * 128 + 2 * C::PLANE>(bx[i][0]);
      }
    } else {
      constexpr int u0 = 2 * s, u1 = 2 * s + 1;
      constexpr int z0 = ZS == 2 ? u0 / 7 : 0, z1 = ZS == 2 ? u1 / 7 : 0;
      constexpr int c0 = ZS == 2 ? 4 * (u0 % 7) : 4 * u0, c1 = ZS == 2 ? 4 * (u1 % 7) : 4 * u1;     // first column of the half
      const int h = r & 1;
      if (r < 8) {
        const int j = r >> 1;
        if (h == 0) f.bl[j] = wgp_tr_read<z0 * C::DYPLANE + c0 * 128>(by[j]);
        else f.bh[j] = wgp_tr_read<z1 * C::DYPLANE + c1 * 128>(by[j]);
      } else {
        const int i = (r - 8) >> 1;
        if (h == 0) f.al[i] = wgp_tr_read<c0 * 64>(bx[i][z0]);
        else f.ah[i] = wgp_tr_read<c1 * 64>(bx[i][z1]);
      }
    }
  };
  auto reads = [&](auto S, WgpFrags& f) {                     // all 22 of a step (the very first step of a block)
#pragma unroll
    for (int r = 0; r < 22; ++r) read1(S, r, f);
  };
  auto landed = [&](WgpFrags& f) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(f.bl[j]), "+v"(f.bh[j]));
#pragma unroll
    for (int i = 0; i < 7; ++i) asm volatile("" : "+v"(f.al[i]), "+v"(f.ah[i]));
  };
  // the 28 MFMAs of a step on `cur`, with the 22 fragment reads of step S (of this group, or step 0 of the next one)
  // into `nxt` issued one behind each of the first 22: the wave never stops feeding the matrix pipe to issue reads
  // (a read burst in front of the MFMAs idles the pipe whenever the SIMD's two waves are in the same phase, and a
  // barrier per group puts them there), and the last 6 MFMAs cover the last read's latency
  auto step = [&](auto S, const WgpFrags& cur, WgpFrags& nxt) {
    i32x4_wg b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = (i32x4_wg){cur.bl[j][0], cur.bl[j][1], cur.bh[j][0], cur.bh[j][1]};
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const i32x4_wg a = (i32x4_wg){cur.al[i][0], cur.al[i][1], cur.ah[i][0], cur.ah[i][1]};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        wgp_mfma(acc[i][j], a, b[j]);
        if (4 * i + j < 22) read1(S, 4 * i + j, nxt);
      }
    }
  };
  // steps 0 .. 5 of a group whose step-0 fragments are in flight in fa; leaves step 6's in flight in fa
  auto steps_0_5 = [&](WgpFrags& fa, WgpFrags& fb) {
    // (measured and not kept, round 5: entering a step on counted lgkmcnt waits per unit -- 12, 14, 15 -- instead of one
    // full wait: fine-tune step 41.4 - 41.7 vs 41.4 - 41.5 ms, no gain: the SIMD's other wave covers the wait)
    landed(fa); step(std::integral_constant<int, 1>{}, fa, fb);
    landed(fb); step(std::integral_constant<int, 2>{}, fb, fa);
    landed(fa); step(std::integral_constant<int, 3>{}, fa, fb);
    landed(fb); step(std::integral_constant<int, 4>{}, fb, fa);
    landed(fa); step(std::integral_constant<int, 5>{}, fa, fb);
    landed(fb); step(std::integral_constant<int, 6>{}, fb, fa);
  };
  // group G = (kc, gq) from its step-0 fragments in fa; its last step's MFMAs are issued behind the barrier that
  // publishes group G + 1's slabs, interleaved with the reads of that group's step-0 fragments (into fb)
  int kc = 0, gq = 0;
  auto group = [&](int G, WgpFrags& fa, WgpFrags& fb) {
    steps_0_5(fa, fb);
    landed(fa);                                               // step 6's fragments: this group's slabs are no longer read
    int kc1 = kc, gq1 = gq + 1;
    if (gq1 == C::NG) { gq1 = 0; ++kc1; }
    const bool more = G + 1 < n_groups;
    if (more) bases(kc1, gq1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // group G + 1's slabs have landed (SEAMLESS, or same column) ...
    __builtin_amdgcn_s_barrier();                             // ... for everybody, and everybody is done with group G's
    if (!C::SEAMLESS && more && gq1 == 0) {
      // no room to prefetch a column's first planes: they are fetched now, exposed (once per NG groups)
      fetch(kc1, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    step(std::integral_constant<int, 0>{}, fa, fb);           // (after the last group the reads fetch nothing that is used)
    // the DMA issue (address arithmetic) sits behind the MFMAs, where only one fragment set is live
    int kc2 = kc1, gq2 = gq1 + 1;                              // group G + 2
    if (gq2 == C::NG) { gq2 = 0; ++kc2; }
    if (G + 2 < n_groups && (C::SEAMLESS || gq2 != 0)) fetch(kc2, gq2);
    kc = kc1; gq = gq1;
  };

  WgpFrags f0, f1;
  // the bias unit's A fragment is all ones (bf16): acc[6][j] = column sums of dY.  Written ONCE, here -- that wave issues no
  // reads into these registers (read1) -- and never next to its use: the MFMAs are inline asm, so the compiler pads no
  // VALU-write -> MFMA-read hazard for them (a select placed directly in front of the unit's first MFMA made that MFMA read
  // the old registers)
  f0.al[6] = f0.ah[6] = f1.al[6] = f1.ah[6] = (i32x2_wg){0x3F803F80, 0x3F803F80};
  asm volatile("" : "+v"(f0.al[6]), "+v"(f0.ah[6]), "+v"(f1.al[6]), "+v"(f1.ah[6]));
  if constexpr (P14) {
    // the halo planes of the row slots are never fetched: zero the ring once (LDS writes, drained before the first DMA)
    for (int o = tid * 16; o < C::DY_OFF; o += 512 * 16) *(u32x4*)(wp_smem + o) = (u32x4){0u, 0u, 0u, 0u};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  fetch(0, 0);
  bases(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  fetch(0, 1);                                                // NG >= 2: same column
  reads(std::integral_constant<int, 0>{}, f0);
#pragma clang loop unroll(disable)
  for (int G = 0; G < n_groups; G += 2) {
    group(G, f0, f1);
    if (C::NG % 2 == 0 || G + 1 < n_groups) group(G + 1, f1, f0);     // (7 groups per column at 14 x 14: the count may be odd)
  }

  // the last step's (unused) fragment reads have returned; the last MFMAs' results are in the registers
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
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
  if (bias_unit && g == 0) {                                   // every row of the ones x dY tile holds the column sums: row 0
#pragma unroll
    for (int j = 0; j < 4; ++j) atomicAdd(p.db + ns * 64 + j * 16 + fcol, acc[6][j][0]);
  }
}

}  // namespace rgp
