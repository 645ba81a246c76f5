// C3D conv4a (256->512) and conv4b (512->512 + pool4), 3x3x3 pad 1 on 4 x 14 x 14 positions, for gfx950, bf16: the patch
// scheme of conv_patch.hip.h for the layers whose pooled rows hold 7 windows.
// Spec: /root/reference/C3D/.../c3d_prototxt/feature_extration.prototxt:174-240 (conv4a, conv4b, pool4).
//
// What differs from conv_patch.hip.h (read that header first):
//  * 7 pooling windows per pooled row.  Every POOLED ROW is a slot of its own in the LDS patch: 4 input planes x 4 input
//    rows x 16 pixels, fetched row by row (one LDS-DMA instruction = one 16-pixel row of one channel slice = 1 KB) -- rows
//    shared by neighbouring pooled rows are fetched twice, 64 KB per 27 K steps against the 432 KB of filter slabs.  A
//    block tile is 4 consecutive slots = 28 pooling windows = 224 positions x 256 channels (waves 2 (M) x 4 (N)); the 512
//    output channels are two column tiles that follow each other in the tile order (the second finds the patch in L2).
//    Slots are numbered (chunk of 4 clip windows, pooled plane zp, window, pooled row yp), so that a block tile lies in
//    ONE pooled plane (only tiles of a short last chunk can straddle the two; they skip nothing).
//  * fragments are dz-PURE: a 16-row m-tile is 4 pooling windows x (dy, dx) of one output plane z = 2 zp + dz.  The depth
//    is 4 with padding 1, so output plane z = 0 multiplies its kz = 0 taps with the zero halo plane z = -1, and z = 3 its
//    kz = 2 taps with the halo plane z = 4: in a tile of pooled plane 0 the dz = 0 fragments skip the kz = 0 tap group, in
//    a tile of pooled plane 1 the dz = 1 fragments skip kz = 2 -- a sixth of the dense MFMA work, products with structural
//    zeros, is not issued (rates are still quoted against the dense 27-tap FLOP count).  Each M wave holds 4 fragments of
//    one dz and 3 of the other (wave 0: dz 0 f 0..3 + dz 1 f 0..2; wave 1: dz 1 f 3..6 + dz 0 f 4..6), so both waves of a
//    SIMD shrink in the same K steps (12 / 16 MFMAs instead of 28) and the pair stays balanced.
//  * banks: LDS rows have a pitch of 1152 B (= 128 mod 256: the 2 x 2 pixels of a window fall in the four 64-byte
//    quarters of a 256-byte bank row) and the rows of the ODD slots start 32 bytes late; a fragment takes two windows from
//    even slots (rows 0-7) and two from odd slots (rows 8-15), which puts the 16 lanes of every ds_read_b128 group on 16
//    different 16-byte slots for every tap (scripts/check_conv14_banks.py enumerates them).
//  * pooling: max over (dy, dx) in the lane's four accumulator registers, max over dz between two fragments of the same
//    wave -- except fragment 3, whose dz = 0 half lives in M wave 0 and dz = 1 half in M wave 1: wave 1 passes its fp32
//    maxima (and arg-max indices) through the filter-ring slot that is idle during the epilogue.
#pragma once
#include <type_traits>

#include "conv_patch.hip.h"

#ifndef RGP_MMA_ORDER
#define RGP_MMA_ORDER 1     // 1: filter fragment outermost in a step (7 consecutive MFMAs share it; 0: the activation fragment, 4): -0.5 % wall
#endif
#ifndef RGP_PLANE_AUX
#define RGP_PLANE_AUX 0      // cache policy of the plane-slab LDS-DMA (2 = nt; measured, see docs/HISTORY.md)
#endif

namespace rgp {

template <int CIN, bool POOL, int NOUT_ = 512> struct Patch14Cfg {
  static constexpr int NOUT = NOUT_, TN = 256;            // output channels (512; 256: conv4a's input gradient), channels per tile (waves 2 x 4)
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
  static constexpr int BIAS_OFF = POOL ? STGA_OFF : STG_OFF;   // inference kernels: the 512 biases (pooled: over the unused arg-max code area)
  static constexpr int SMEM = POOL ? STGA_OFF + WIN * STG_LD : STG_OFF + NOUT * 4;     // 162 464 / 142 336
  static constexpr int NSTEP = NCC * 27;
  static constexpr int K = 27 * CIN;
  static constexpr int IN_ROW = 16 * CIN, IN_PLANE = 16 * IN_ROW, IN_IMG = 6 * IN_PLANE;                 // elements
  static constexpr int OW = POOL ? 7 : 14, OD = POOL ? 2 : 4;
  static constexpr int OUT_ROW = (OW + 2) * NOUT, OUT_PLANE = (OW + 2) * OUT_ROW, OUT_IMG = (OD + 2) * OUT_PLANE;
  static constexpr int CGN = TN / 8;
  static_assert(SMEM <= 160 * 1024, "LDS budget");
  static_assert(LROW % 256 == 128, "row pitch = 128 (mod 256)");
};

// DENSE (DGRAD only): the output is the dense, un-masked [n][4*14*14][NOUT] image the un-pool kernel consumes (conv4a's
// input gradient, NOUT = 256: the gradient w.r.t. pool3's output).
template <int CIN, bool POOL, bool ARGMAX = false, bool DGRAD = false, int NOUT = 512, bool DENSE = false>
static __global__ __launch_bounds__(512) void conv_patch14_bf16_kernel(const ConvPatchParams p) {
  static_assert(POOL || !ARGMAX, "arg-max codes belong to the pooled layer");
  static_assert(!POOL || !DGRAD, "the input gradient is an un-pooled convolution");
  static_assert(!DENSE || DGRAD, "dense output: input gradients only");
  using C = Patch14Cfg<CIN, POOL, NOUT>;
  extern __shared__ __attribute__((aligned(16))) char cq_smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)cq_smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const bool group_b = wave >= 4;
  const int frow = lane & 15, fk = lane >> 4;
  auto plane_base = [](int k) { return (unsigned)(k * C::PLANE_STRIDE); };

  // tiles: (block tile b = slots 4b .. 4b+3 of the 14 n_windows, column tile ct), ct innermost; dealt to the XCDs in
  // contiguous ranges
  const int n_slots = 14 * p.n_windows;                      // pooled rows
  const int nt = ((n_slots + 3) >> 2) * C::NCT;
  auto tile_of = [&](int t) {
    const int q = nt >> 3, r = nt & 7, x = t & 7, y = t >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
  };
  int t_seq = blockIdx.x;
  if (t_seq >= nt) return;

  // pooled row of slot u (0 .. 3) of block tile b.  Slots are numbered (chunk of 4 clip windows, pooled plane zp, window,
  // pooled row yp): 28 slots = 7 block tiles per pooled plane of a chunk, so no tile of a full chunk straddles the two
  // planes, and the two planes of a window (which share input planes 1, 2) are 7 block tiles apart: same XCD, same round
  struct Slot { int n, zp, yp; bool valid; };
  auto slot_of = [&](int b, int u) {
    Slot s;
    int g = 4 * b + u;
    s.valid = g < n_slots;
    if (!s.valid) g = n_slots - 1;
    const int q = g / 56, r = g - 56 * q;
    const int cw = min(4, p.n_windows - 4 * q);               // windows of this chunk (the last one may be short)
    s.zp = r >= 7 * cw ? 1 : 0;
    const int rr = r - s.zp * 7 * cw;
    const int nn = rr / 7;
    s.n = 4 * q + nn;
    s.yp = rr - nn * 7;
    return s;
  };
  // window w (0 .. 3) of fragment f (0 .. 6) = column xp = f of slot u: w = 0, 1 -> the even slots 0, 2; w = 2, 3 -> the odd
  // slots 1, 3 (whose LDS rows start 32 bytes late: the bank argument of the header)
  auto frag_window = [](int f, int w, int& u, int& xp) {
    u = 2 * (w & 1) + (w >> 1);
    xp = f;
  };
  // the two input rows (of plane k, channel sweep cc) this wave fetches for a tile: slot wave >> 1, rows 2 (wave & 1), +1
  const int dpix = lane >> 2, dchk = lane & 3;
  auto plane_src = [&](int tile, int cc, int k) -> const char* {
    const Slot s = slot_of(tile / C::NCT, wave >> 1);
    return (const char*)(p.in + (long long)s.n * C::IN_IMG + (long long)(2 * s.zp + k) * C::IN_PLANE + (2 * s.yp + 2 * (wave & 1)) * C::IN_ROW +
                         cc * 32);
  };
  // (src is a wave-uniform base; the lane's part is one 32-bit offset, so no 64-bit pointer lives in vector registers)
  const unsigned dlane = (unsigned)(dpix * (CIN * 2) + dchk * 16);
  auto dma_plane = [&](const char* src, int k, bool last_touch = false) {
    if (last_touch) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + u * 16 * (CIN * 2) + dlane),
                                         (__attribute__((address_space(3))) void*)(cq_smem + plane_base(k) + ((wave >> 1) * 4 + 2 * (wave & 1) + u) * C::LROW + 32 * ((wave >> 1) & 1)),
                                         16, 0, 2 /* nt */);
      return;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + u * 16 * (CIN * 2) + dlane),
                                       (__attribute__((address_space(3))) void*)(cq_smem + plane_base(k) + ((wave >> 1) * 4 + 2 * (wave & 1) + u) * C::LROW + 32 * ((wave >> 1) & 1)),
                                       16, 0, RGP_PLANE_AUX);
  };
  // filter slab (conv_patch.hip.h), rows of column tile ct
  const int brow = lane >> 2;
  const int bchk = (lane & 3) ^ ((-(brow >> 2)) & 3);
  auto b_row = [&](int blk) { return POOL ? blk * 16 + brow : (blk >> 2) * 64 + brow * 4 + (blk & 3); };
  unsigned b_off[2];                                          // this lane's byte offset inside the packed filter (< 2^31)
#pragma unroll
  for (int u = 0; u < 2; ++u) b_off[u] = (unsigned)(b_row(wave * 2 + u) * (C::K * 2) + bchk * 16);
  auto dma_b = [&](int slot, int ct, int cc, int tap) {
    const char* base = (const char*)p.wp + ((long long)ct * C::TN * C::K * 2 + (((cc >> 1) * 27 + tap) * 64 + (cc & 1) * 32) * 2);
#pragma unroll
    for (int u = 0; u < 2; ++u)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + b_off[u]),
                                       (__attribute__((address_space(3))) void*)(cq_smem + C::BRING_OFF + slot * C::BSLOT + (wave * 2 + u) * 1024), 16, 0, 0);
  };

  // fragment addressing.  Accumulator slot i of M wave wm: i < 4 -> (dz = wm, fragment f = 3 wm + i), i >= 4 -> (dz = 1 - wm,
  // f = i - 4 + 4 wm); row frow of a fragment: window frow >> 2, (dy, dx) = ((frow >> 1) & 1, frow & 1); K chunk fk
  auto frag_of = [&](int i) { return i < 4 ? 3 * wm + i : i - 4 + 4 * wm; };
  const int r_dy = (frow >> 1) & 1, r_dx = frow & 1;
  // LDS address of this lane's row (window frow >> 2, (dy, dx)) of fragment column 0 in the plane its dz reads for the
  // CURRENT tap group (kz + dz): ra_lo for accumulator slots 0 .. 3 (dz = wm, columns 3 wm + i), ra_hi for slots 4 .. 6
  // (dz = 1 - wm, columns 4 wm + i - 4); the column is an immediate of 128 bytes per fragment.  Advanced in place by one
  // plane per tap group and taken back by two at the end of a sweep.
  unsigned ra_lo, ra_hi;
  {
    int u, xp;
    frag_window(0, frow >> 2, u, xp);
    const unsigned base = lds0 + (u * 4 + r_dy) * C::LROW + 32 * (u & 1) + r_dx * 64 + fk * 16;
    ra_lo = base + (3 * wm) * 128 + plane_base(wm);
    ra_hi = base + (4 * wm) * 128 + plane_base(1 - wm);
  }
  const unsigned b_addr = lds0 + C::BRING_OFF + (wn * 4) * 1024 + frow * 64 + ((fk ^ ((-(frow >> 2)) & 3)) << 4);
  // inference kernels keep the 512 biases in LDS (the arg-max code area is unused there): the epilogue reads them with a
  // ds_read; a global load there waits on vmcnt(0), i.e. for the whole look-ahead DMA.  Training kernels load them per tile.
  constexpr bool BIAS_LDS = !DGRAD && !ARGMAX;
  if constexpr (BIAS_LDS) { if (tid < C::NOUT) ((float*)(cq_smem + C::BIAS_OFF))[tid] = p.bias[tid]; }
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
    f32x4 acc[7][4];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (group_b) __builtin_amdgcn_s_barrier();               // run one half-step behind group A

    // MODE 0: all 7 fragments; 1: only slots 4 .. 6 (the first four multiply a halo plane in this tap group); 2: only 0 .. 3
    auto tap_group = [&](auto NPL_, auto MODE_, int cc, int kz, const char* pl_a, int ka, const char* pl_b, int kb, bool last_touch) {
      constexpr int NPL = decltype(NPL_)::value, MODE = decltype(MODE_)::value;
      constexpr int I0 = MODE == 1 ? 4 : 0, I1 = MODE == 2 ? 4 : 7;
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
          if constexpr (I0 < 4) {
            af[0] = cp_lds_read128<imm>(ra_lo);
            af[1] = cp_lds_read128<imm + 128>(ra_lo);
            af[2] = cp_lds_read128<imm + 256>(ra_lo);
            af[3] = cp_lds_read128<imm + 384>(ra_lo);
          }
          if constexpr (I1 > 4) {
            af[4] = cp_lds_read128<imm>(ra_hi);
            af[5] = cp_lds_read128<imm + 128>(ra_hi);
            af[6] = cp_lds_read128<imm + 256>(ra_hi);
          }
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
        for (int i = I0; i < I1; ++i) asm volatile("" : "+v"(af[i]));
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
          for (int i = I0; i < I1; ++i) Mma<bf16_t>::step(acc[i][j], af[i], bf[j]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        slot = slot + 1 == C::NSLOT ? 0 : slot + 1;
      }
      // next tap group: one plane further (after kz = 2: back to the planes of kz = 0)
      ra_lo += kz == 2 ? (unsigned)(-2 * C::PLANE_STRIDE) : (unsigned)C::PLANE_STRIDE;
      ra_hi += kz == 2 ? (unsigned)(-2 * C::PLANE_STRIDE) : (unsigned)C::PLANE_STRIDE;
    };
    using I0_ = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    // pooled plane of the tile (-1: its slots straddle the two planes or the tile is ragged -- nothing is skipped) and
    // this wave's skip mode in the kz = 0 / kz = 2 tap groups: its dz = 0 fragments are slots 0 .. 3 of M wave 0 and
    // 4 .. 6 of M wave 1
    const int bt = tile / C::NCT;
    int zpu;
    {
      const Slot a0 = slot_of(bt, 0), a3 = slot_of(bt, 3);
      zpu = (a3.valid && a0.zp == a3.zp) ? a0.zp : -1;
      if (RGP_CP_ABL(p, 16)) zpu = -1;                        // dev: no tap group skipped (timing of the layout alone)
    }
    const int mode_k0 = zpu == 0 ? (wm == 0 ? 1 : 2) : 0;     // kz = 0 on pooled plane 0: the dz = 0 slots are idle
    const int mode_k2 = zpu == 1 ? (wm == 0 ? 2 : 1) : 0;     // kz = 2 on pooled plane 1: the dz = 1 slots are idle
#pragma clang loop unroll(disable)
    for (int cc = 0; cc < C::NCC; ++cc) {
      const bool last = cc == C::NCC - 1;
      const int ntile = last ? tile_next : tile;
      const int ncc = last ? 0 : cc + 1;
      // last touch of an input line (conv_patch.hip.h: `nt` hint): an odd sweep (the second 64-byte half) of the LAST
      // column tile of a block tile; conv4a -0.4 %, conv4b -0.7 %; the input gradient was not measured and stays without
      const bool lt = !DGRAD && (cc & 1) != 0 && tile % C::NCT == C::NCT - 1;
      const bool nlt = !DGRAD && (ncc & 1) != 0 && ntile % C::NCT == C::NCT - 1;
      const char* p2 = plane_src(tile, cc, 2);
      const char* p3 = plane_src(tile, cc, 3);
      if (mode_k0 == 0) tap_group(I2{}, I0_{}, cc, 0, p2, 2, p3, 3, lt);
      else if (mode_k0 == 1) tap_group(I2{}, I1{}, cc, 0, p2, 2, p3, 3, lt);
      else tap_group(I2{}, I2{}, cc, 0, p2, 2, p3, 3, lt);
      tap_group(I1{}, I0_{}, cc, 1, plane_src(ntile, ncc, 0), 0, nullptr, 0, nlt);
      const char* n1 = plane_src(ntile, ncc, 1);
      if (mode_k2 == 0) tap_group(I1{}, I0_{}, cc, 2, n1, 1, nullptr, 0, nlt);
      else if (mode_k2 == 1) tap_group(I1{}, I1{}, cc, 2, n1, 1, nullptr, 0, nlt);
      else tap_group(I1{}, I2{}, cc, 2, n1, 1, nullptr, 0, nlt);
    }
    if (!group_b) __builtin_amdgcn_s_barrier();               // the groups are level again

    // Epilogue-only lane values are re-derived here, per tile, from opaque copies of the lane ids (left visible, the compiler
    // hoists staging offsets and window coordinates out of the tile loop and spills them across the K loop; a scratch reload
    // in the epilogue waits on vmcnt(0)).
    int e_fk = fk, e_frow = frow, e_tid = tid;
    asm volatile("" : "+v"(e_fk), "+v"(e_frow), "+v"(e_tid));
    // bias of this lane's 4 MFMA columns in this column tile
    float b4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = ct * C::TN + (POOL ? wn * 64 + q * 16 + e_frow : wn * 64 + e_frow * 4 + q);
      b4[q] = DGRAD ? 0.f : BIAS_LDS ? ((const float*)(cq_smem + C::BIAS_OFF))[ch] : p.bias[ch];
    }
    if constexpr (POOL) {
      // ---- epilogue: pool in registers -- max over (dy, dx) = the four registers, max over dz = two accumulator slots of
      // this wave (fragment 3: one slot here, one in the other M wave, exchanged in fp32 through the idle ring slot) --
      // then bias + ReLU, pooled bf16 tile (and arg-max codes dz 4 + dy 2 + dx, first maximum) through LDS ----
      bf16_t* stg = (bf16_t*)(cq_smem + C::STG_OFF);
      unsigned char* stga = (unsigned char*)(cq_smem + C::STGA_OFF);
      float* xm = (float*)(cq_smem + C::BRING_OFF + ((slot + C::AHEAD) & (C::NSLOT - 1)) * C::BSLOT);   // idle until the next LOAD phase
      unsigned char* xi = (unsigned char*)(xm + 4 * 256);
      float m[7][4];
      int mi[7][4];
#pragma unroll
      for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 c = acc[i][j];
          float best = c[0];
          int idx = 0;
          if (c[1] > best) { best = c[1]; idx = 1; }
          if (c[2] > best) { best = c[2]; idx = 2; }
          if (c[3] > best) { best = c[3]; idx = 3; }
          m[i][j] = best;
          mi[i][j] = idx;
        }
      // staging position of (fragment f, this lane's window fk, column j 16 + frow)
      auto stage_pos = [&](int f, int j) {
        int u, xp;
        frag_window(f, e_fk, u, xp);
        return (u * 7 + xp) * C::STG_LD + wn * 64 + j * 16 + e_frow;
      };
      auto put = [&](int f, int j, float v0, int i0, float v1, int i1) {      // v0: dz = 0 maximum, v1: dz = 1
        const int so = stage_pos(f, j);
        const bool hi = v1 > v0;
        stg[so] = f2bf(fmaxf((hi ? v1 : v0) + b4[j], 0.f));
        if constexpr (ARGMAX) stga[so] = (unsigned char)(hi ? i1 + 4 : i0);
      };
      if (wm == 0) {
#pragma unroll
        for (int f = 0; f < 3; ++f)
#pragma unroll
          for (int j = 0; j < 4; ++j) put(f, j, m[f][j], mi[f][j], m[f + 4][j], mi[f + 4][j]);
      } else {
#pragma unroll
        for (int f = 4; f < 7; ++f)
#pragma unroll
          for (int j = 0; j < 4; ++j) put(f, j, m[f][j], mi[f][j], m[f - 3][j], mi[f - 3][j]);
        // dz = 1 half of fragment 3 (slot 0 here) for M wave 0: [window fk][column] fp32 + index byte
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xm[e_fk * 256 + wn * 64 + j * 16 + e_frow] = m[0][j];
          if constexpr (ARGMAX) xi[e_fk * 256 + wn * 64 + j * 16 + e_frow] = (unsigned char)mi[0][j];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // raw barrier: __syncthreads() would also drain the look-ahead DMA
      __builtin_amdgcn_s_barrier();
      if (wm == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v1 = xm[e_fk * 256 + wn * 64 + j * 16 + e_frow];
          int i1 = 0;
          if constexpr (ARGMAX) i1 = xi[e_fk * 256 + wn * 64 + j * 16 + e_frow];
          put(3, j, m[3][j], mi[3][j], v1, i1);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const int cg = e_tid % C::CGN;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int w = e_tid / C::CGN + (512 / C::CGN) * k;      // window of the tile: slot u = w / 7, column xp = w % 7
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
      // ---- epilogue: bias + ReLU, 8-byte stores from registers (conv_patch.hip.h).  Accumulator slot i, register e: window
      // fk of fragment frag_of(i), dz of the slot, dy = e >> 1, dx = e & 1; channels ct 256 + 64 wn + 4 frow + 0 .. 3 ----
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        int u, xp;
        frag_window(frag_of(i), e_fk, u, xp);
        const Slot sl = slot_of(bt, u);
        const int dz = i < 4 ? wm : 1 - wm;
        if (sl.valid) {
          // position (z, y, x) = (2 zp + dz, 2 yp + dy, 2 xp + dx): halo-padded image, or (DENSE) natural order without halo
          constexpr int ROWS = DENSE ? 14 * C::NOUT : C::OUT_ROW, PLANES = DENSE ? 14 * ROWS : C::OUT_PLANE;
          constexpr long long IMG = DENSE ? 4LL * PLANES : (long long)C::OUT_IMG;
          constexpr int H1 = DENSE ? 0 : 1;
          const long long ow = (long long)sl.n * IMG + (2 * sl.zp + dz + H1) * PLANES + (2 * sl.yp + H1) * ROWS + (2 * xp + H1) * C::NOUT +
                               ct * C::TN + wn * 64 + e_frow * 4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const long long oe = ow + (e >> 1) * ROWS + (e & 1) * C::NOUT;
            uint2 o;
            if constexpr (DGRAD && DENSE) {
              o.x = (unsigned)f2bf(acc[i][0][e]) | ((unsigned)f2bf(acc[i][1][e]) << 16);
              o.y = (unsigned)f2bf(acc[i][2][e]) | ((unsigned)f2bf(acc[i][3][e]) << 16);
            } else if constexpr (DGRAD) {
              const uint2 mk = *(const uint2*)(p.mask + oe);
              const float v0 = bf2f((bf16_t)(mk.x & 0xffffu)) > 0.f ? acc[i][0][e] : 0.f, v1 = bf2f((bf16_t)(mk.x >> 16)) > 0.f ? acc[i][1][e] : 0.f;
              const float v2 = bf2f((bf16_t)(mk.y & 0xffffu)) > 0.f ? acc[i][2][e] : 0.f, v3 = bf2f((bf16_t)(mk.y >> 16)) > 0.f ? acc[i][3][e] : 0.f;
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
    }
    if (!has_next) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the look-ahead DMA lands before the LDS is released
      break;
    }
    t_seq = t_next;
  }
}

}  // namespace rgp
