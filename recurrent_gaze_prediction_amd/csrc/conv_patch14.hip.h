// C3D conv4a (256->512) and conv4b (512->512 + pool4), 3x3x3 pad 1 on 4 x 14 x 14 positions, for gfx950, bf16: the patch
// scheme of conv_patch.hip.h for the layers whose pooled rows hold 7 windows.
// Spec: /root/reference/C3D/.../c3d_prototxt/feature_extration.prototxt:174-240 (conv4a, conv4b, pool4).
//
// What differs from conv_patch.hip.h (read that header first):
//  * 7 pooling windows per pooled row: an m-tile pairs the windows (row r, xp) and (row r + 1, xp) VERTICALLY, a wave
//    owns two consecutive pooled rows x 7 columns (14 windows = 7 m-tiles), and "consecutive" runs through the 14 pooled
//    rows of a clip window (2 pooled planes x 7): the pair (6, 7) straddles the two pooled planes.  A block tile is two
//    such wave tiles (waves 2 (M) x 4 (N): 224 positions x 256 channels) and may straddle two clip windows; the 512
//    output channels are two column tiles that follow each other in the tile order (the second finds the patch in L2).
//  * so every POOLED ROW is a slot of its own in the LDS patch: 4 input planes x 4 input rows x 16 pixels, fetched row by
//    row (one LDS-DMA instruction = one 16-pixel row of one channel slice = 1 KB) -- rows shared by neighbouring pooled
//    rows are fetched twice, 64 KB per 27 K steps against the 432 KB of filter slabs.  LDS rows have a pitch of 1152 B
//    (= 128 mod 256) and the plane buffers start at 32 (k & 1): the bank argument of conv_patch.hip.h holds per window
//    (its 2 x 2 pixels of a plane fall in the four 64-byte quarters of a bank row), whatever the two windows of an
//    m-tile are.
#pragma once
#include <type_traits>

#include "conv_patch.hip.h"

#ifndef RGP_MMA_ORDER
#define RGP_MMA_ORDER 1     // 1: filter fragment outermost in a step (7 consecutive MFMAs share it; 0: the activation fragment, 4): -0.5 % wall
#endif
#ifndef RGP_PLANE_AUX
#define RGP_PLANE_AUX 0      // cache policy of the plane-slab LDS-DMA (2 = nt; measured)
#endif

namespace rgp {

template <int CIN, bool POOL> struct Patch14Cfg {
  static constexpr int NOUT = 512, TN = 256;              // output channels, channels per tile (waves 2 x 4)
  static constexpr int NCT = NOUT / TN;                   // column tiles
  static constexpr int NCC = CIN / 32;                    // channel sweeps: 8 / 16
  static constexpr int LROW = 1152;                       // LDS row pitch (16 pixels x 64 B + 128)
  static constexpr int PLANE_BYTES = 16 * LROW;           // 4 slots x 4 rows
  static constexpr int PLANE_STRIDE = PLANE_BYTES + 256;
  static constexpr int PPW = 2;                           // row fetches per wave and plane
  static constexpr int BRING_OFF = (3 * PLANE_STRIDE + 32 + PLANE_BYTES + 1023) / 1024 * 1024;   // 74 752
  static constexpr int BPW = 2, BSLOT = TN * 64, NSLOT = 4, AHEAD = 3;
  static constexpr int STG_OFF = BRING_OFF + NSLOT * BSLOT;
  static constexpr int WIN = 28, STG_LD = TN + 8;
  static constexpr int STGA_OFF = STG_OFF + WIN * STG_LD * 2;
  static constexpr int SMEM = POOL ? STGA_OFF + WIN * STG_LD : STG_OFF;     // 162 464 / 140 288
  static constexpr int NSTEP = NCC * 27;
  static constexpr int K = 27 * CIN;
  static constexpr int IN_ROW = 16 * CIN, IN_PLANE = 16 * IN_ROW, IN_IMG = 6 * IN_PLANE;                 // elements
  static constexpr int OW = POOL ? 7 : 14, OD = POOL ? 2 : 4;
  static constexpr int OUT_ROW = (OW + 2) * NOUT, OUT_PLANE = (OW + 2) * OUT_ROW, OUT_IMG = (OD + 2) * OUT_PLANE;
  static constexpr int CGN = TN / 8;
  static_assert(SMEM <= 160 * 1024, "LDS budget");
  static_assert(LROW % 256 == 128, "row pitch = 128 (mod 256)");
};

template <int CIN, bool POOL, bool ARGMAX = false, bool DGRAD = false>
static __global__ __launch_bounds__(512) void conv_patch14_bf16_kernel(const ConvPatchParams p) {
  static_assert(POOL || !ARGMAX, "arg-max codes belong to the pooled layer");
  static_assert(!POOL || !DGRAD, "the input gradient is an un-pooled convolution");
  using C = Patch14Cfg<CIN, POOL>;
  extern __shared__ __attribute__((aligned(16))) char cq_smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)cq_smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const bool group_b = wave >= 4;
  const int frow = lane & 15, fk = lane >> 4;
  auto plane_base = [](int k) { return (unsigned)(k * C::PLANE_STRIDE + 32 * (k & 1)); };

  // tiles: (block tile b = wave tiles 2b, 2b+1 of the 7 n_windows, column tile ct), ct innermost; dealt to the XCDs in
  // contiguous ranges
  const int n_wt = 7 * p.n_windows;                          // wave tiles (pairs of pooled rows)
  const int nt = ((n_wt + 1) >> 1) * C::NCT;
  auto tile_of = [&](int t) {
    const int q = nt >> 3, r = nt & 7, x = t & 7, y = t >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
  };
  int t_seq = blockIdx.x;
  if (t_seq >= nt) return;

  // pooled row of slot u (0 .. 3) of block tile b: wave tile 2b + (u >> 1), row 2 j + (u & 1) of the clip's 14
  struct Slot { int n, zp, yp; bool valid; };
  auto slot_of = [&](int b, int u) {
    Slot s;
    int g = 2 * b + (u >> 1);
    s.valid = g < n_wt;
    if (!s.valid) g = n_wt - 1;
    s.n = g / 7;
    const int r = 2 * (g - s.n * 7) + (u & 1);
    s.zp = r >= 7 ? 1 : 0;
    s.yp = r - 7 * s.zp;
    return s;
  };
  // the two input rows (of plane k, channel sweep cc) this wave fetches for a tile: slot wave >> 1, rows 2 (wave & 1), +1
  const int dpix = lane >> 2, dchk = lane & 3;
  auto plane_src = [&](int tile, int cc, int k) -> const char* {
    const Slot s = slot_of(tile / C::NCT, wave >> 1);
    return (const char*)(p.in + (long long)s.n * C::IN_IMG + (long long)(2 * s.zp + k) * C::IN_PLANE + (2 * s.yp + 2 * (wave & 1)) * C::IN_ROW +
                         cc * 32);
  };
  auto dma_plane = [&](const char* src, int k, bool last_touch = false) {
    if (last_touch) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (u * 16 + dpix) * (CIN * 2) + dchk * 16),
                                         (__attribute__((address_space(3))) void*)(cq_smem + plane_base(k) + ((wave >> 1) * 4 + 2 * (wave & 1) + u) * C::LROW),
                                         16, 0, 2 /* nt */);
      return;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (u * 16 + dpix) * (CIN * 2) + dchk * 16),
                                       (__attribute__((address_space(3))) void*)(cq_smem + plane_base(k) + ((wave >> 1) * 4 + 2 * (wave & 1) + u) * C::LROW),
                                       16, 0, RGP_PLANE_AUX);
  };
  // filter slab (conv_patch.hip.h), rows of column tile ct
  const int brow = lane >> 2;
  const int bchk = (lane & 3) ^ ((-(brow >> 2)) & 3);
  auto b_row = [&](int blk) { return POOL ? blk * 16 + brow : (blk >> 2) * 64 + brow * 4 + (blk & 3); };
  const char* b_src[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) b_src[u] = (const char*)(p.wp + (long long)b_row(wave * 2 + u) * C::K) + bchk * 16;
  auto dma_b = [&](int slot, int ct, int cc, int tap) {
    const long long koff = (long long)ct * C::TN * C::K * 2 + (((cc >> 1) * 27 + tap) * 64 + (cc & 1) * 32) * 2;
#pragma unroll
    for (int u = 0; u < 2; ++u)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[u] + koff),
                                       (__attribute__((address_space(3))) void*)(cq_smem + C::BRING_OFF + slot * C::BSLOT + (wave * 2 + u) * 1024), 16, 0, 0);
  };

  // fragment addressing: m-tile i = column xp = i; row frow of it: window (slot 2 wm + (frow >> 3)), dz, dy, dx; K chunk fk
  const int r_ws = frow >> 3, r_dz = (frow >> 2) & 1, r_dy = (frow >> 1) & 1, r_dx = frow & 1;
  unsigned rowaddr[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) rowaddr[i] = lds0 + ((2 * wm + r_ws) * 4 + r_dy) * C::LROW + (2 * i + r_dx) * 64 + fk * 16;
  const unsigned b_addr = lds0 + C::BRING_OFF + (wn * 4) * 1024 + frow * 64 + ((fk ^ ((-(frow >> 2)) & 3)) << 4);
  const int cg = tid % C::CGN;
  float b4ct[C::NCT][4];
#pragma unroll
  for (int t = 0; t < C::NCT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) b4ct[t][q] = DGRAD ? 0.f : p.bias[t * C::TN + (POOL ? wn * 64 + q * 16 + frow : wn * 64 + frow * 4 + q)];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- prologue (once) ----
  {
    const int tile0 = tile_of(t_seq);
    dma_plane(plane_src(tile0, 0, 0), 0);
    dma_plane(plane_src(tile0, 0, 1), 1);
    dma_b(0, tile0 % C::NCT, 0, 0);
    dma_b(1, tile0 % C::NCT, 0, 1);
    dma_b(2, tile0 % C::NCT, 0, 2);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * C::BPW) : "memory");   // planes 0, 1 and slab 0 landed
    __builtin_amdgcn_s_barrier();
  }
  int slot = 0;
  while (true) {
    const int tile = tile_of(t_seq);
    const int t_next = t_seq + gridDim.x;
    const bool has_next = t_next < nt;
    const int tile_next = has_next ? tile_of(t_next) : tile;
    const int ct = tile % C::NCT, ct_next = tile_next % C::NCT;
    float b4[4];                                              // bias of this lane's 4 MFMA columns in this column tile
#pragma unroll
    for (int q = 0; q < 4; ++q) b4[q] = ct ? b4ct[1][q] : b4ct[0][q];

    f32x4 acc[7][4];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (group_b) __builtin_amdgcn_s_barrier();               // run one half-step behind group A

    auto tap_group = [&](auto NPL_, int cc, int kz, const char* pl_a, int ka, const char* pl_b, int kb, bool last_touch = false) {
      constexpr int NPL = decltype(NPL_)::value;
      unsigned ra[7];
      const unsigned pb = r_dz ? plane_base(kz + 1) : plane_base(kz);
#pragma unroll
      for (int i = 0; i < 7; ++i) ra[i] = rowaddr[i] + pb;
      const int s0 = cc * 27 + kz * 9;
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {
        // ---------------- LOAD ----------------
        const int s = s0 + t9;
        f32x4 af[7], bf[4];
        const unsigned bb = b_addr + slot * C::BSLOT;
        auto reads = [&](auto T9) {
          constexpr int t = decltype(T9)::value;
          constexpr int imm = (t / 3) * C::LROW + (t % 3) * 64;
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
        if (t9 == 0) {
          if (NPL >= 1) dma_plane(pl_a, ka, last_touch);
          if (NPL >= 2) dma_plane(pl_b, kb, last_touch);
        }
        {
          // filter slab of step s + 3 (at the end of a tile: steps 0 .. 2 of the next one, in its column tile)
          int s3 = s + C::AHEAD;
          int ct3 = ct;
          if (s3 >= C::NSTEP) { s3 -= C::NSTEP; ct3 = ct_next; }
          const int cc3 = s3 / 27;
          int slot3 = slot + C::AHEAD;
          if (slot3 >= C::NSLOT) slot3 -= C::NSLOT;
          dma_b(slot3, ct3, cc3, s3 - cc3 * 27);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t9 < 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * C::BPW + NPL * C::PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * C::BPW) : "memory");
#pragma unroll
        for (int i = 0; i < 7; ++i) asm volatile("" : "+v"(af[i]));
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(bf[j]));
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- COMPUTE ----------------
        __builtin_amdgcn_s_setprio(1);
#if RGP_MMA_ORDER == 1
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 7; ++i) Mma<bf16_t>::step(acc[i][j], af[i], bf[j]);
#else
#pragma unroll
        for (int i = 0; i < 7; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) Mma<bf16_t>::step(acc[i][j], af[i], bf[j]);
#endif
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        slot = slot + 1 == C::NSLOT ? 0 : slot + 1;
      }
    };
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
#pragma clang loop unroll(disable)
    for (int cc = 0; cc < C::NCC; ++cc) {
      const bool last = cc == C::NCC - 1;
      const int ntile = last ? tile_next : tile;
      const int ncc = last ? 0 : cc + 1;
      // last touch of an input line (conv_patch.hip.h: `nt` hint): an odd sweep (the second 64-byte half) of the LAST
      // column tile of a block tile; conv4a -0.4 %, conv4b -0.7 %; the input gradient was not measured and stays without
      const bool lt = !DGRAD && (cc & 1) != 0 && tile % C::NCT == C::NCT - 1;
      const bool nlt = !DGRAD && (ncc & 1) != 0 && ntile % C::NCT == C::NCT - 1;
      tap_group(I2{}, cc, 0, plane_src(tile, cc, 2), 2, plane_src(tile, cc, 3), 3, lt);
      tap_group(I1{}, cc, 1, plane_src(ntile, ncc, 0), 0, nullptr, 0, nlt);
      tap_group(I1{}, cc, 2, plane_src(ntile, ncc, 1), 1, nullptr, 0, nlt);
    }
    if (!group_b) __builtin_amdgcn_s_barrier();               // the groups are level again

    const int bt = tile / C::NCT;
    if constexpr (POOL) {
      // ---- epilogue: pool in registers, bias + ReLU, pooled bf16 tile (and arg-max codes) through LDS ----
      bf16_t* stg = (bf16_t*)(cq_smem + C::STG_OFF);
      unsigned char* stga = (unsigned char*)(cq_smem + C::STGA_OFF);
#pragma unroll
      for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 c = acc[i][j];
          const int so = ((2 * wm + (fk >> 1)) * 7 + i) * C::STG_LD + wn * 64 + j * 16 + frow;      // window = slot * 7 + xp
          if constexpr (ARGMAX) {
            float best = c[0];
            int idx = 0;
            if (c[1] > best) { best = c[1]; idx = 1; }
            if (c[2] > best) { best = c[2]; idx = 2; }
            if (c[3] > best) { best = c[3]; idx = 3; }
            const float ob = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, best), 0x401F));   // lane ^ 16
            const int oi = __builtin_amdgcn_ds_swizzle(idx, 0x401F);
            if ((fk & 1) == 0) {
              stg[so] = f2bf(fmaxf((ob > best ? ob : best) + b4[j], 0.f));
              stga[so] = (unsigned char)(ob > best ? oi + 4 : idx);
            }
          } else {
            const float x = fmaxf(fmaxf(c[0], c[1]), fmaxf(c[2], c[3]));
            const float y = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), 0x401F));
            if ((fk & 1) == 0) stg[so] = f2bf(fmaxf(fmaxf(x, y) + b4[j], 0.f));
          }
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // raw barrier: __syncthreads() would also drain the look-ahead DMA
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int w = tid / C::CGN + (512 / C::CGN) * k;      // window of the tile: slot u = w / 7, column xp = w % 7
        if (w < C::WIN) {
          const int u = w / 7, xp = w - u * 7;
          const Slot sl = slot_of(bt, u);
          if (sl.valid) {
            bf16_t* o = p.out + (long long)sl.n * C::OUT_IMG + (sl.zp + 1) * C::OUT_PLANE + (sl.yp + 1) * C::OUT_ROW + (xp + 1) * C::NOUT + ct * C::TN +
                        cg * 8;
            *(u32x4*)o = *(const u32x4*)(stg + w * C::STG_LD + cg * 8);
            if constexpr (ARGMAX)
              *(uint2*)(p.argmax + ((((long long)sl.n * 2 + sl.zp) * 7 + sl.yp) * 7 + xp) * C::NOUT + ct * C::TN + cg * 8) =
                  *(const uint2*)(stga + w * C::STG_LD + cg * 8);
          }
        }
      }
    } else {
      // ---- epilogue: bias + ReLU, 8-byte stores from registers (conv_patch.hip.h); this lane's window: slot
      // 2 wm + (fk >> 1), column i; dz = fk & 1; register e: dy = e >> 1, dx = e & 1 ----
      const Slot sl = slot_of(bt, 2 * wm + (fk >> 1));
      if (sl.valid) {
        const long long ow = (long long)sl.n * C::OUT_IMG + (2 * sl.zp + (fk & 1) + 1) * C::OUT_PLANE + (2 * sl.yp + 1) * C::OUT_ROW + C::NOUT + ct * C::TN +
                             wn * 64 + frow * 4;
#pragma unroll
        for (int i = 0; i < 7; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const long long oe = ow + (e >> 1) * C::OUT_ROW + (2 * i + (e & 1)) * C::NOUT;
            uint2 o;
            if constexpr (DGRAD) {
              const uint2 m = *(const uint2*)(p.mask + oe);
              const float v0 = bf2f((bf16_t)(m.x & 0xffffu)) > 0.f ? acc[i][0][e] : 0.f, v1 = bf2f((bf16_t)(m.x >> 16)) > 0.f ? acc[i][1][e] : 0.f;
              const float v2 = bf2f((bf16_t)(m.y & 0xffffu)) > 0.f ? acc[i][2][e] : 0.f, v3 = bf2f((bf16_t)(m.y >> 16)) > 0.f ? acc[i][3][e] : 0.f;
              o.x = (unsigned)f2bf(v0) | ((unsigned)f2bf(v1) << 16);
              o.y = (unsigned)f2bf(v2) | ((unsigned)f2bf(v3) << 16);
            } else {
              o.x = (unsigned)f2bf(fmaxf(acc[i][0][e] + b4[0], 0.f)) | ((unsigned)f2bf(fmaxf(acc[i][1][e] + b4[1], 0.f)) << 16);
              o.y = (unsigned)f2bf(fmaxf(acc[i][2][e] + b4[2], 0.f)) | ((unsigned)f2bf(fmaxf(acc[i][3][e] + b4[3], 0.f)) << 16);
            }
            *(uint2*)(p.out + oe) = o;
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
