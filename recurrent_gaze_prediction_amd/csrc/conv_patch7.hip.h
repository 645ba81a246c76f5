// C3D conv5a and conv5b (512 -> 512, 3x3x3 pad 1, no pooling) on 2 x 7 x 7 positions for gfx950, bf16, and their input
// gradients: the patch scheme of conv_patch.hip.h for the layers whose planes are 7 x 7.
// Spec: /root/reference/C3D/.../c3d_prototxt/feature_extration.prototxt:241-342 (conv5a, conv5b; the reference stops at
// conv5b, extract_C3D_features.py:689-724); the gradients are tf.gradients through them (base.py:278-281).
//
// What differs from conv_patch.hip.h / conv_patch14.hip.h (read those headers first):
//  * depth 2 with padding 1: output plane z reads input planes z-1 .. z+1, of which z = -1 and z = 2 are the zero halo.
//    A wave owns positions of ONE output plane, so it runs only the 18 taps whose input plane is real (kz = 1, 2 for
//    z = 0; kz = 0, 1 for z = 1): a third of the dense MFMA work is a product with structural zeros and is not issued,
//    and the halo planes are never fetched.  (Rates are still quoted against the dense 27-tap FLOP count of the layer.)
//  * a block tile = 4 clip windows x all 98 positions x 128 output channels; waves 4 (M) x 2 (N), wave tile 112 x 64 as
//    everywhere.  M waves 0, 1 own plane z = 0 (fragments 0..6 / 7..12 + one idle), M waves 2, 3 plane z = 1.  A 16-row
//    fragment is 4 WINDOWS x 4 consecutive positions of the plane: 13 fragments hold the 49 positions (5.8 % padding;
//    with the idle 14th fragment slot 12.5 %).
//  * banks: the window images lie in LDS with a row pitch of 11 pixels (= 3 mod 4: consecutive positions advance by one
//    64-byte quarter of a 256-byte bank row, also across a row end), and the images of windows 2, 3 start 32 bytes
//    (mod 64) after those of windows 0, 1: the 16 lanes of every ds_read_b128 group -- rows of two windows with K chunk
//    c, rows of the other two with chunk c +- 1 -- cover the 64 banks exactly once, for every tap (an immediate offset
//    of (ky 11 + kx) 64 bytes).  No swizzle; checked by enumeration: scripts/check_conv7_banks.py.
//  * K order: channel sweep cc (16 of 32 channels) x input plane (2) x (ky, kx) (9).  The waves of plane z = 0 and those
//    of z = 1 read the same input plane at the same time but multiply it with different filter taps (kz = plane + 1 - z),
//    so a K step brings TWO filter slabs (128 rows x 64 B each): 16 KB + 1/9 of a 32 KB plane buffer per step,
//    5.4 KB of LDS-DMA per issued MFLOP (igemm_stagger_kernel: 11.4).
//  * two plane buffers: plane 1 of a sweep is fetched at the start of its plane-0 phase, plane 0 of the next sweep (or
//    of the next tile) at the start of the plane-1 phase; the filter ring (4 slots x 16 KB, 3 steps ahead) runs across
//    tile boundaries as in conv_patch.hip.h.
#pragma once
#include <type_traits>

#include "conv_patch.hip.h"

namespace rgp {

struct Patch7Cfg {
  static constexpr int CIN = 512, NOUT = 512, TN = 128;
  static constexpr int NCT = NOUT / TN;                   // 4 column tiles
  static constexpr int NCC = CIN / 32;                    // 16 channel sweeps
  static constexpr int WPX = 11;                          // LDS row pitch in pixels (= 3 mod 4)
  static constexpr int WIN_BYTES = 128 * 64;              // one window image: 128 pixels (9 rows x 11 used) x 64 B
  static constexpr int PLANE_BYTES = 4 * WIN_BYTES + 64;  // 4 windows; windows 2, 3 shifted by 32 B
  static constexpr int PLANE_STRIDE = 33024;              // multiple of 256
  static constexpr int BRING_OFF = 66560;                 // >= 2 * PLANE_STRIDE, multiple of 1024
  static constexpr int SLAB = TN * 64;                    // 8 KB: 128 filter rows x 32 K elements
  static constexpr int BSLOT = 2 * SLAB;                  // the z = 0 and the z = 1 tap of a step
  static constexpr int NSLOT = 4, AHEAD = 3;
  static constexpr int BIAS_OFF = BRING_OFF + NSLOT * BSLOT;   // the layer's 512 biases (fp32), read by the epilogues
  static constexpr int SMEM = BIAS_OFF + NOUT * 4;        // 134 144
  static constexpr int NSTEP = NCC * 18;
  static constexpr int K = 27 * CIN;
  static constexpr int IN_PLANE = 81 * CIN, IN_IMG = 4 * IN_PLANE;      // elements, halo-padded [4][9][9][512]
  static_assert(2 * PLANE_STRIDE <= BRING_OFF && PLANE_BYTES <= PLANE_STRIDE && PLANE_STRIDE % 256 == 0, "plane buffers");
  static_assert(SMEM <= 160 * 1024, "LDS budget");
};

// OUT: 0 = halo-padded image [n][4][9][9][512] (conv5a's output; the masked input gradient of conv5b), 1 = conv5b's rows
// [n 49 + y 7 + x][z 512 + c], 2 = dense [n][98][512] (conv5a's input gradient, for the un-pool kernel).
template <int OUT, bool DGRAD = false>
static __global__ __launch_bounds__(512) void conv_patch7_bf16_kernel(const ConvPatchParams p) {
  using C = Patch7Cfg;
  static_assert(OUT >= 0 && OUT <= 2 && (OUT != 2 || DGRAD) && (OUT != 1 || !DGRAD), "output forms");
  extern __shared__ __attribute__((aligned(16))) char c7_smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)c7_smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int zout = wm >> 1;                                 // output plane of this wave
  const bool group_b = wave >= 4;                           // (= the z = 1 waves)
  const int frow = lane & 15, fk = lane >> 4;
  auto win_base = [](int w) { return (unsigned)(w * C::WIN_BYTES + 32 * (w >> 1)); };

  // tiles: (group g of 4 windows, column tile ct), ct innermost; dealt to the XCDs in contiguous ranges
  const int ng = (p.n_windows + 3) >> 2;
  const int nt = ng * C::NCT;
  auto tile_of = [&](int t) {
    const int q = nt >> 3, r = nt & 7, x = t & 7, y = t >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
  };
  int t_seq = blockIdx.x;
  if (t_seq >= nt) return;

  // ---- plane fetch: this wave's 4 LDS-DMA instructions = pixels [64 (wave & 1), +64) of window image wave >> 1 ----
  const int dwin = wave >> 1;
  int dsrc[4];                                              // byte offset of this lane's 16 B inside a source plane
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int l = 64 * (wave & 1) + 16 * u + (lane >> 2);
    const int y = l / C::WPX, x = l - y * C::WPX;
    dsrc[u] = ((y < 9 && x < 9) ? (y * 9 + x) * (C::CIN * 2) : 0) + (lane & 3) * 16;
  }
  auto plane_src = [&](int tile, int cc, int pl) -> const char* {
    int n = 4 * (tile / C::NCT) + dwin;
    if (n >= p.n_windows) n = p.n_windows - 1;              // ragged last group: a valid window, never stored
    return (const char*)(p.in + (long long)n * C::IN_IMG + (long long)(pl + 1) * C::IN_PLANE + cc * 32);
  };
  auto dma_plane = [&](const char* src, int buf) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + dsrc[u]),
                                       (__attribute__((address_space(3))) void*)(c7_smem + buf * C::PLANE_STRIDE + win_base(dwin) +
                                                                                  (64 * (wave & 1) + 16 * u) * 64),
                                       16, 0, 0);
  };
  // ---- filter slabs of a K step: this wave's 1-KB block (16 filter rows x 64 B) of the z = 0 and of the z = 1 tap, chunk-
  // swizzled as in conv_patch.hip.h; MFMA column 16 j + c of a wave carries channel 64 wn + 4 c + j ----
  const int brow = lane >> 2;
  const int bchk = (lane & 3) ^ ((-(brow >> 2)) & 3);
  const char* b_src = (const char*)(p.wp + (long long)((wave >> 2) * 64 + brow * 4 + (wave & 3)) * C::K) + bchk * 16;
  auto dma_b = [&](int slot, int ct, int cc, int r) {      // K step 18 cc + r of a tile: sweep cc, r = 9 h + (3 ky + kx)
    const int h = r >= 9 ? 1 : 0, t9 = r - 9 * h;
    const long long base = (long long)ct * C::TN * C::K * 2 + (((cc >> 1) * 27 + t9) * 64 + (cc & 1) * 32) * 2;
    // input plane h: tap kz = h + 1 for the z = 0 waves, kz = h for the z = 1 waves
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src + base + (h + 1) * 9 * 128),
                                     (__attribute__((address_space(3))) void*)(c7_smem + C::BRING_OFF + slot * C::BSLOT + wave * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src + base + h * 9 * 128),
                                     (__attribute__((address_space(3))) void*)(c7_smem + C::BRING_OFF + slot * C::BSLOT + C::SLAB + wave * 1024), 16, 0, 0);
  };

  // ---- fragment addressing: fragment f = 7 (wm & 1) + i of this wave's plane; row frow = window frow >> 2, position
  // 4 f + (frow & 3) (positions 49 .. 55 are padding: they read on linearly, inside the 128-pixel image) ----
  unsigned rowaddr[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int pos = 4 * (7 * (wm & 1) + i) + (frow & 3);
    const int y = pos / 7, x = pos - y * 7;
    rowaddr[i] = lds0 + win_base(frow >> 2) + (y * C::WPX + x) * 64 + fk * 16;
  }
  const unsigned b_addr = lds0 + C::BRING_OFF + zout * C::SLAB + (wn * 4) * 1024 + frow * 64 + ((fk ^ ((-(frow >> 2)) & 3)) << 4);
  if constexpr (!DGRAD) ((float*)(c7_smem + C::BIAS_OFF))[tid] = p.bias[tid];       // 512 threads, 512 channels
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- prologue (once): plane 0 of the first sweep, filter slabs of steps 0 .. 2 ----
  {
    const int tile0 = tile_of(t_seq);
    dma_plane(plane_src(tile0, 0, 0), 0);
    dma_b(0, tile0 % C::NCT, 0, 0);
    dma_b(1, tile0 % C::NCT, 0, 1);
    dma_b(2, tile0 % C::NCT, 0, 2);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");         // plane 0 and slab 0 landed
    __builtin_amdgcn_s_barrier();
  }
  int slot = 0;
  while (true) {
    const int tile = tile_of(t_seq);
    const int t_next = t_seq + gridDim.x;
    const bool has_next = t_next < nt;
    const int tile_next = has_next ? tile_of(t_next) : tile;
    const int ct = tile % C::NCT, ct_next = tile_next % C::NCT;
    f32x4 acc[7][4];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (group_b) __builtin_amdgcn_s_barrier();               // run one half-step behind group A

    // the 9 (ky, kx) taps on input plane h (buffer h) of sweep cc; the plane fetch `pl_src` -> buffer `pl_buf` is issued in
    // the first LOAD phase
    auto phase = [&](int cc, int h, const char* pl_src, int pl_buf) {
      unsigned ra[7];
#pragma unroll
      for (int i = 0; i < 7; ++i) ra[i] = rowaddr[i] + h * C::PLANE_STRIDE;
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {
        // ---------------- LOAD ----------------
        f32x4 af[7], bf[4];
        const unsigned bb = b_addr + slot * C::BSLOT;
        auto reads = [&](auto T9) {
          constexpr int t = decltype(T9)::value;
          constexpr int imm = ((t / 3) * C::WPX + (t % 3)) * 64;
#pragma unroll
          for (int i = 0; i < 7; ++i) af[i] = cp_lds_read128<imm>(ra[i]);
        };
        switch (t9) {
          case 0: reads(std::integral_constant<int, 0>{}); break;
          case 1: reads(std::integral_constant<int, 1>{}); break;
          case 2: reads(std::integral_constant<int, 2>{}); break;
          case 3: reads(std::integral_constant<int, 3>{}); break;
          case 4: reads(std::integral_constant<int, 4>{}); break;
          case 5: reads(std::integral_constant<int, 5>{}); break;
          case 6: reads(std::integral_constant<int, 6>{}); break;
          case 7: reads(std::integral_constant<int, 7>{}); break;
          default: reads(std::integral_constant<int, 8>{}); break;
        }
        bf[0] = cp_lds_read128<0>(bb);
        bf[1] = cp_lds_read128<1024>(bb);
        bf[2] = cp_lds_read128<2048>(bb);
        bf[3] = cp_lds_read128<3072>(bb);
        __builtin_amdgcn_sched_barrier(0);
        if (t9 == 0) dma_plane(pl_src, pl_buf);
        {
          // filter slabs of step s + 3 (at the end of a tile: steps 0 .. 2 of the next one, in its column tile)
          int r3 = h * 9 + t9 + C::AHEAD, cc3 = cc, ct3 = ct;
          if (r3 >= 18) { r3 -= 18; ++cc3; }
          if (cc3 == C::NCC) { cc3 = 0; ct3 = ct_next; }
          int slot3 = slot + C::AHEAD;
          if (slot3 >= C::NSLOT) slot3 -= C::NSLOT;
          dma_b(slot3, ct3, cc3, r3);
        }
        __builtin_amdgcn_sched_barrier(0);
        // slabs of step s + 1 landed: younger are those of s + 2, s + 3 (2 instructions each) and, in the two steps after
        // a plane fetch, its 4 instructions
        if (t9 < 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 7; ++i) asm volatile("" : "+v"(af[i]));
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(bf[j]));
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- COMPUTE ----------------
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 7; ++i) Mma<bf16_t>::step(acc[i][j], af[i], bf[j]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        slot = slot + 1 == C::NSLOT ? 0 : slot + 1;
      }
    };
#pragma clang loop unroll(disable)
    for (int cc = 0; cc < C::NCC; ++cc) {
      const bool last = cc == C::NCC - 1;
      phase(cc, 0, plane_src(tile, cc, 1), 1);                                   // reads buffer 0; plane 1 of this sweep -> buffer 1
      phase(cc, 1, plane_src(last ? tile_next : tile, last ? 0 : cc + 1, 0), 0); // reads buffer 1; plane 0 of the next sweep -> buffer 0
    }
    if (!group_b) __builtin_amdgcn_s_barrier();               // the groups are level again

    // ---- epilogue: bias + ReLU (or the ReLU mask of the forward activation), 8-byte stores from registers.  Register e of
    // accumulator (i, j): window fk of the group, position 4 f + e of plane zout, channel ct 128 + 64 wn + 4 frow + j ----
    {
      const int n = 4 * (tile / C::NCT) + fk;
      const int ch = ct * C::TN + wn * 64 + frow * 4;
      f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
      if constexpr (!DGRAD) b4 = *(const f32x4*)(c7_smem + C::BIAS_OFF + ch * 4);
      if (n < p.n_windows) {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          const int f = 7 * (wm & 1) + i;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int pos = 4 * f + e;
            if (pos < 49) {
              const int y = pos / 7, x = pos - y * 7;
              long long oe;
              if constexpr (OUT == 0) oe = (long long)n * C::IN_IMG + (zout + 1) * C::IN_PLANE + ((y + 1) * 9 + x + 1) * C::NOUT + ch;
              else if constexpr (OUT == 1) oe = ((long long)n * 49 + pos) * (2 * C::NOUT) + zout * C::NOUT + ch;
              else oe = ((long long)n * 98 + zout * 49 + pos) * C::NOUT + ch;
              float v[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = DGRAD ? acc[i][j][e] : fmaxf(acc[i][j][e] + b4[j], 0.f);
              if constexpr (DGRAD && OUT == 0) {
                const uint2 m = *(const uint2*)(p.mask + oe);
                if (!(bf2f((bf16_t)(m.x & 0xffffu)) > 0.f)) v[0] = 0.f;
                if (!(bf2f((bf16_t)(m.x >> 16)) > 0.f)) v[1] = 0.f;
                if (!(bf2f((bf16_t)(m.y & 0xffffu)) > 0.f)) v[2] = 0.f;
                if (!(bf2f((bf16_t)(m.y >> 16)) > 0.f)) v[3] = 0.f;
              }
              uint2 o;
              o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
              o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
              *(uint2*)(p.out + oe) = o;
            }
          }
        }
      }
    }
    if (!has_next) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the look-ahead DMA lands before the LDS is released
      break;
    }
    t_seq = t_next;
  }
}

}  // namespace rgp
