// Plane-slab variant of the patch kernel (conv_patch.hip.h), kept for conv2a + pool2 (64 -> 128 channels on 56 x 56 x 16).
// Spec: /root/reference/C3D/.../c3d_prototxt/feature_extration.prototxt:67-107.
//
// Round 3 moved the 56 x 56 / 28 x 28 kernels to dz-pure fragments on a row pitch = 32 (mod 256), which makes the
// halo-plane tap groups of a fragment skippable (conv3a / conv3b: 1/12 of the MFMAs not issued) at the price of a
// ROW-WISE plane fetch: 6 rows x 64 pixels per plane and channel sweep instead of one contiguous slab of 6 x 58 = 348
// pixels (384 fetched) -- the same 24 LDS-DMA instructions, 10 % more bytes that are actually new to L2.  conv2a skips
// nothing (2 of its 8 pooled planes could; measured: no gain, twice the traffic), so it only pays: same-box A/B
// 14.61 -> 14.92 ms per 1024 windows on the slow boxes of the pool, -0.5 % on the fast ones (profiles/r03_ab_vs_r02.txt).
// This file is round 2's kernel for that layer: a plane slab = 348 consecutive pixels of the halo-padded input (full
// rows incl. the x halo) brought in by 3 LDS-DMA instructions per wave; fragments of 2 adjacent pooling windows x
// (dz, dy, dx) on a row pitch of 58 * 64 = 128 (mod 256) bytes, plane buffers at 32 (k & 1) (mod 256): every 16-lane
// group of a ds_read_b128 covers the 64 banks once, no swizzle.  Everything else (K order, filter ring, plane ring,
// staggered wave groups, pooled epilogue, the last-touch nt hint) as described in conv_patch.hip.h; the packed filter and
// the activation layouts are the same, so the two kernels are interchangeable per launch: inference plans take this one
// unless created with RGP_C3D_CONV2A_ROWWISE, and the results are bit-identical (the K order and the accumulation order
// inside a lane are the same; tests/test_c3d_gpu.py compares the two plans with torch.equal).  Same-box A/B, round 5
// (profiles/r05_ab_conv2a_slab.txt): conv2a 14.23 vs 14.35 - 14.45 ms per 1024 windows (-0.9 ... -1.5 %).
#pragma once
#include "conv_patch.hip.h"

namespace rgp {

template <int CIN, int NOUT, int HW, int DEPTH> struct PatchSlabCfg {
  static constexpr int WP = HW + 2;                       // padded row: 58 / 30 pixels
  static constexpr int XPN = HW / 2;                      // pooling windows per pooled row: 28 / 14
  static constexpr int NCC = CIN / 32;                    // channel sweeps: 2 / 8
  static constexpr int WIN = 2 * XPN;                     // pooling windows per tile: 56 / 28
  static constexpr int WMW = WIN / 14, WNW = 8 / WMW;     // waves along M (14 windows = 7 m-tiles each) and N
  static constexpr int PPW = (6 * WP + 127) / 128;        // plane-slab DMA instructions per wave: 3 / 2
  static constexpr int PLANE_PIX = PPW * 128;             // pixels fetched per slab: 384 / 256 (348 / 180 used)
  static constexpr int PLANE_BYTES = PLANE_PIX * 64, PLANE_STRIDE = PLANE_BYTES + 256;
  static constexpr int BRING_OFF = (3 * PLANE_STRIDE + 32 + PLANE_BYTES + 1023) / 1024 * 1024;
  static constexpr int NI = NOUT / (16 * WNW);            // 16-column MFMA tiles per wave: 4
  static constexpr int BPW = (NOUT + 127) / 128;          // filter-slab DMA instructions per wave and step: 1 / 2
  static constexpr int BSLOT = BPW * 128 * 64;            // 8 / 16 KB: filter rows (padded to 128: the packing pads too) x 32 K elements
  static constexpr int NSLOT = 4, AHEAD = 3;
  static constexpr int STG_OFF = BRING_OFF + NSLOT * BSLOT;
  // staged pooled tile (bias + ReLU applied, bf16) [WIN][NOUT + 8] and its arg-max codes [WIN][NOUT + 8] bytes: an area of
  // its own (the filter ring keeps running across tiles)
  static constexpr int STG_LD = NOUT + 8;
  static constexpr int STGA_OFF = STG_OFF + WIN * STG_LD * 2;
  static constexpr int SMEM = STGA_OFF;                   // (no arg-max codes: inference plans only)
  static constexpr int NSTEP = NCC * 27;
  static constexpr int YT = HW / 4;                       // tiles per pooled plane
  static constexpr int TILES_PER_WINDOW = (DEPTH / 2) * YT;
  static constexpr int K = 27 * CIN;
  static constexpr int IN_ROW = WP * CIN, IN_PLANE = WP * IN_ROW, IN_IMG = (DEPTH + 2) * IN_PLANE;      // elements
  static constexpr int OW = HW / 2, OD = DEPTH / 2;       // output extent (pooled)
  static constexpr int OUT_ROW = (OW + 2) * NOUT, OUT_PLANE = (OW + 2) * OUT_ROW, OUT_IMG = (OD + 2) * OUT_PLANE;
  static constexpr int CGN = NOUT / 8;                    // epilogue: 8-channel groups
  static_assert(WNW * 16 * NI == NOUT && NI == 4 && WMW * 14 == WIN && XPN % 2 == 0 && HW % 4 == 0 && CIN % 32 == 0, "tile shape");
  static_assert((WP * 64) % 256 == 128, "row pitch = 128 (mod 256): the bank argument of the header");
  static_assert(SMEM <= 160 * 1024, "LDS budget");
  static_assert(WIN * CGN == 2 * 448, "pooled epilogue: two items per thread (448 of the 512 threads)");
  static_assert((STG_LD * 2) % 16 == 0 && STG_LD % 8 == 0, "staging rows keep 16- / 8-byte alignment");
};

// MFMA column 16 j + c of a wave carries channel 64 wn + 4 c + j (the filter slab is fetched in that row order), so a lane
// holds 4 adjacent channels of a position; 2x2x2 max-pool epilogue: the lane's four pooled channels of a window are staged
// with one 8-byte LDS write.  Inference plans only: no arg-max codes, no input-gradient or un-pooled variants (those are
// conv_patch.hip.h's; round 4 carried them here too, never instantiated).
template <int CIN, int NOUT, int HW, int DEPTH>
static __global__ __launch_bounds__(512) void conv_patch_slab_bf16_kernel(const ConvPatchParams p) {
  using C = PatchSlabCfg<CIN, NOUT, HW, DEPTH>;
  constexpr int NI = C::NI;
  extern __shared__ __attribute__((aligned(16))) char cp_smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)cp_smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / C::WNW, wn = wave % C::WNW;
  const bool group_b = wave >= 4;
  const int frow = lane & 15, fk = lane >> 4;
  auto plane_base = [](int k) { return (unsigned)(k * C::PLANE_STRIDE + 32 * (k & 1)); };

  // persistent tile walk: XCD x (workgroup id & 7) owns a contiguous range of tiles (neighbouring tiles share halo rows
  // and planes: one L2 serves them)
  const int nt = p.n_windows * C::TILES_PER_WINDOW;
  auto tile_of = [&](int t) {
    const int q = nt >> 3, r = nt & 7, x = t & 7, y = t >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
  };
  int t_seq = blockIdx.x;
  if (t_seq >= nt) return;
#ifdef RGP_DEV_KNOBS
  // dev experiment (RGP_CP_ABLATE bits 8..15 = n): block i of an XCD starts i * n * 0.43 us late, so that the 32 CUs of an
  // XCD read the filter out of phase (every slab is then re-touched 32 times per tile time instead of once: it stays in
  // L2).  Measured (docs/HISTORY.md): traffic beyond L2 falls, time RISES -- the re-streams are not what these kernels wait for.
  if ((p.ablate >> 8) & 0xff) {
    const int n = ((p.ablate >> 8) & 0xff) * (blockIdx.x >> 3);
    for (int k = 0; k < n; ++k) __builtin_amdgcn_s_sleep(16);
  }
#endif

  // source of plane k of (tile, channel sweep cc)
  auto plane_src = [&](int tile, int cc, int k) -> const char* {
    const int n = tile / C::TILES_PER_WINDOW, r = tile - n * C::TILES_PER_WINDOW;
    const int zp = r / C::YT, yp = r - zp * C::YT;
    if (RGP_CP_ABL(p, 1)) return (const char*)(p.in + (long long)(blockIdx.x & 7) * C::IN_PLANE);
    return (const char*)(p.in + (long long)n * C::IN_IMG + (long long)(2 * zp + k) * C::IN_PLANE + (4 * yp) * C::IN_ROW + cc * 32);
  };
  // this wave's PPW of a plane slab's DMA instructions: 16 pixels x 64 B each
  const int dpix = lane >> 2, dchk = lane & 3;
  auto dma_plane = [&](const char* src, int k, bool last_touch = false) {
    if (last_touch) {
#pragma unroll
      for (int u = 0; u < C::PPW; ++u) {
        const int j = wave * C::PPW + u;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (j * 16 + dpix) * (CIN * 2) + dchk * 16),
                                         (__attribute__((address_space(3))) void*)(cp_smem + plane_base(k) + j * 1024), 16, 0, 2 /* nt */);
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < C::PPW; ++u) {
      const int j = wave * C::PPW + u;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (j * 16 + dpix) * (CIN * 2) + dchk * 16),
                                       (__attribute__((address_space(3))) void*)(cp_smem + plane_base(k) + j * 1024), 16, 0, RGP_PLANE_AUX);
    }
  };
  // filter slab of K step (cc, tap): this wave's BPW of its 1-KB blocks (16 filter rows x 64 B), chunk-swizzled like
  // igemm_wide.hip.h (physical chunk c of row r holds logical chunk c ^ ((-(r >> 2)) & 3))
  const int brow = lane >> 2;
  const int bchk = (lane & 3) ^ ((-(brow >> 2)) & 3);
  // filter row (output channel) behind row brow of 1-KB block blk: wave blk / NI, column tile blk % NI: channel
  // 16 NI (blk / NI) + NI brow + blk % NI
  auto b_row = [&](int blk) { return (blk / NI) * (16 * NI) + brow * NI + (blk % NI); };
  const char* b_src[C::BPW];
#pragma unroll
  for (int u = 0; u < C::BPW; ++u) b_src[u] = (const char*)(p.wp + (long long)b_row(wave * C::BPW + u) * C::K) + bchk * 16;
  auto dma_b = [&](int slot, int cc, int tap) {
    int koff = (((cc >> 1) * 27 + tap) * 64 + (cc & 1) * 32) * 2;
    if (RGP_CP_ABL(p, 8)) koff = 0;
#pragma unroll
    for (int u = 0; u < C::BPW; ++u)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[u] + koff),
                                       (__attribute__((address_space(3))) void*)(cp_smem + C::BRING_OFF + slot * C::BSLOT + (wave * C::BPW + u) * 1024),
                                       16, 0, 0);
  };

  // fragment addressing.  m-tile i of this wave = pooling windows 2 (7 wm + i), +1; row frow of it: window frow >> 3,
  // dz = (frow >> 2) & 1, dy = (frow >> 1) & 1, dx = frow & 1; K chunk fk.
  const int r_ws = frow >> 3, r_dz = (frow >> 2) & 1, r_dy = (frow >> 1) & 1, r_dx = frow & 1;
  unsigned rowaddr[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int w0 = 2 * (7 * wm + i);
    const int ypl = w0 / C::XPN, xp = w0 - ypl * C::XPN + r_ws;
    rowaddr[i] = lds0 + ((2 * ypl + r_dy) * C::WP + 2 * xp + r_dx) * 64 + fk * 16;
  }
  const unsigned b_addr = lds0 + C::BRING_OFF + (wn * NI) * 1024 + frow * 64 + ((fk ^ ((-(frow >> 2)) & 3)) << 4);

  const int cg = tid % C::CGN;                                // POOL epilogue, store pass: this thread's 8 output channels
  float b4[NI];                                               // bias of this lane's NI MFMA columns
#pragma unroll
  for (int q = 0; q < NI; ++q) b4[q] = p.bias[wn * (16 * NI) + frow * NI + q];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // ---- prologue (once): planes 0, 1 of the first sweep, filter slabs of steps 0 .. 2.  Afterwards the filter ring and
  // the plane prefetch run across tile boundaries: step s of a tile issues slab s + 3 (mod NSTEP) and the last sweep of
  // a tile fetches planes 0, 1 of the next one. ----
  {
    const int tile0 = tile_of(t_seq);
    dma_plane(plane_src(tile0, 0, 0), 0);
    dma_plane(plane_src(tile0, 0, 1), 1);
    dma_b(0, 0, 0);
    dma_b(1, 0, 1);
    dma_b(2, 0, 2);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * C::BPW) : "memory");   // planes 0, 1 and slab 0 landed
    __builtin_amdgcn_s_barrier();
  }
  int slot = 0;
  while (true) {
    const int tile = tile_of(t_seq);
    const int t_next = t_seq + gridDim.x;
    const bool has_next = t_next < nt;
    const int tile_next = has_next ? tile_of(t_next) : tile;

    f32x4 acc[7][NI];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (group_b) __builtin_amdgcn_s_barrier();               // run one half-step behind group A

    // one group of 9 taps (ky, kx) of plane offset kz, K steps s0 .. s0 + 8 of the tile; NPL plane fetches (PPW
    // instructions per wave each) are issued in its first LOAD phase
    auto tap_group = [&](auto NPL_, int cc, int kz, const char* pl_a, int ka, const char* pl_b, int kb, bool odd_sweep = false) {
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
        f32x4 af[7], bf[NI];
        const unsigned bb = b_addr + slot * C::BSLOT;
        auto reads = [&](auto T9) {
          constexpr int t = decltype(T9)::value;
          constexpr int imm = ((t / 3) * C::WP + (t % 3)) * 64;
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
        if constexpr (NI == 4) {
          bf[2] = cp_lds_read128<2048>(bb);
          bf[3] = cp_lds_read128<3072>(bb);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t9 == 0) {
          if (NPL >= 1) dma_plane(pl_a, ka, odd_sweep);
          if (NPL >= 2) dma_plane(pl_b, kb, odd_sweep);
        }
        {
          // filter slab of step s + 3 (at the end of a tile: steps 0 .. 2 of the next one)
          int s3 = s + C::AHEAD;
          if (s3 >= C::NSTEP) s3 -= C::NSTEP;
          const int cc3 = s3 / 27;
          int slot3 = slot + C::AHEAD;
          if (slot3 >= C::NSLOT) slot3 -= C::NSLOT;
          dma_b(slot3, cc3, s3 - cc3 * 27);
        }
        __builtin_amdgcn_sched_barrier(0);
        // slab s+1 landed: younger are slabs s+2, s+3 and, in the two steps after a plane fetch, its instructions
        if (t9 < 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * C::BPW + NPL * C::PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * C::BPW) : "memory");
#pragma unroll
        for (int i = 0; i < 7; ++i) asm volatile("" : "+v"(af[i]));
#pragma unroll
        for (int j = 0; j < NI; ++j) asm volatile("" : "+v"(bf[j]));
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- COMPUTE ----------------
        __builtin_amdgcn_s_setprio(1);
#if RGP_MMA_ORDER == 1
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int i = 0; i < 7; ++i) Mma<bf16_t>::step(acc[i][j], af[i], bf[j]);
#else
#pragma unroll
        for (int i = 0; i < 7; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) Mma<bf16_t>::step(acc[i][j], af[i], bf[j]);
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
      // the sweep after this one: the next channel slice of this tile, or the first one of the next tile
      const bool last = cc == C::NCC - 1;
      const int ntile = last ? tile_next : tile;
      const int ncc = last ? 0 : cc + 1;
      // An odd sweep reads the second 64-byte half of the 128-byte lines its predecessor brought into L2: the tile's
      // last touch of them.  Those fetches carry the `nt` hint (the line becomes the first candidate for eviction), which
      // leaves more of the L2 to the filter and to the neighbouring tiles' rows: conv2a -3.5 %, conv3b -0.5 %; conv3a
      // (CIN = 128) measured +0.5 % and the input gradients were not measured: both stay without it.
      constexpr bool LT = CIN != 128;
      tap_group(I2{}, cc, 0, plane_src(tile, cc, 2), 2, plane_src(tile, cc, 3), 3, LT && (cc & 1) != 0);
      tap_group(I1{}, cc, 1, plane_src(ntile, ncc, 0), 0, nullptr, 0, LT && (ncc & 1) != 0);
      tap_group(I1{}, cc, 2, plane_src(ntile, ncc, 1), 1, nullptr, 0, LT && (ncc & 1) != 0);
    }
    if (!group_b) __builtin_amdgcn_s_barrier();               // the groups are level again

    const int tn = tile / C::TILES_PER_WINDOW, tr = tile - tn * C::TILES_PER_WINDOW;
    const int zp = tr / C::YT, yp = tr - zp * C::YT;
    if (RGP_CP_ABL(p, 2)) {
#pragma unroll
      for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(acc[i][j]));
    } else {
      // ---- epilogue: pool in registers (a lane holds the 4 members of a window with its dz, lane ^ 16 the other 4),
      // bias + ReLU, pooled bf16 tile through LDS, 16-byte stores ----
      bf16_t* stg = (bf16_t*)(cp_smem + C::STG_OFF);
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        unsigned short pv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 c = acc[i][j];
          const float x = cp_max(cp_max3(c[0], c[1], c[2]), c[3]);
          const float y = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), 0x401F));   // lane ^ 16
          pv[j] = f2bf(cp_relu(cp_max(x, y) + b4[j]));
        }
        if ((fk & 1) == 0) {
          const int so = (2 * (7 * wm + i) + (fk >> 1)) * C::STG_LD + wn * 64 + 4 * frow;
          uint2 o;
          o.x = (unsigned)pv[0] | ((unsigned)pv[1] << 16);
          o.y = (unsigned)pv[2] | ((unsigned)pv[3] << 16);
          *(uint2*)(stg + so) = o;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // raw barrier: __syncthreads() would also drain the look-ahead DMA
      __builtin_amdgcn_s_barrier();
      bf16_t* obase = p.out + (long long)tn * C::OUT_IMG + (zp + 1) * C::OUT_PLANE + (2 * yp + 1) * C::OUT_ROW + NOUT;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int w = tid / C::CGN + (512 / C::CGN) * k;      // pooling window of the tile
        if (w < C::WIN && !RGP_CP_ABL(p, 4)) {
          const int ypl = w / C::XPN, xp = w - ypl * C::XPN;
          *(u32x4*)(obase + ypl * C::OUT_ROW + xp * NOUT + cg * 8) = *(const u32x4*)(stg + w * C::STG_LD + cg * 8);
        }
      }
    }
    if (!has_next) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the look-ahead DMA lands before the LDS is released
      break;
    }
    t_seq = t_next;                                            // (the K loop's barriers separate this tile's staging reads from the next one's writes)
  }
}

}  // namespace rgp
