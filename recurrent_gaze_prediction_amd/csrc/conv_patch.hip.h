// The large 3x3x3 convolutions of the C3D stack on 56 x 56 and 28 x 28 planes for gfx950, bf16 -- conv2a (64->128) + pool2,
// conv3a (128->256), conv3b (256->256) + pool3, forward (inference and training: arg-max codes of the pooling), and
// their input gradients (conv3b's masked by the forward activation; conv2a's / conv3a's dense, for the un-pool kernel).
// Spec: /root/reference/C3D/.../c3d_prototxt/feature_extration.prototxt:67-107 (conv2a, pool2), 108-173 (conv3a, conv3b,
// pool3); the gradients are tf.gradients through them (base.py:278-281).  conv4a / conv4b: conv_patch14.hip.h.
// The text below describes conv2a's forward; the other instances are the same kernel at 28 x 28 x 8 positions with
// 4 / 8 channel sweeps of 32, tiles of 2 x 14 pooling windows x 256 channels (waves 2 (M) x 4 (N)), plane slabs of
// 6 x 30 pixels and 16 KB filter slabs; the two dense input gradients have 64 / 128 output channels and use wave tiles
// of 112 x 32 (NI = 2) on the same block tiles.
//
// Why a kernel of its own: conv2a is the largest layer of the stack (22.7 of 78.8 TFLOP per 1024 windows) and the one
// the general implicit-GEMM tile serves worst.  With N = 128 only the A tile can grow, and the A operand -- one
// 64-byte row per output position and 32-deep K step, gathered from L2 by LDS-DMA -- is re-fetched for each of the 27
// taps: the 512 x 128 tile of igemm_wide.hip.h moves 40 KB L2 -> LDS per 4.2 MFLOP, 39 B per cycle and CU at full
// MFMA rate against the 28...35 B per cycle an MI355X CU ingests (MI355X_MICROARCH.md: 66-73 GB/s), and its matrix
// pipe is 62 % busy.  Here the input PATCH of a tile is brought into LDS once per channel half and the fragments of
// all 27 taps are read from it at shifted addresses (what conv1a.hip.h does for the first layer):
//
//  * tile = 2 pooled rows x 28 pooled columns of one pooled plane of one window = 56 pooling windows = 448 conv
//    outputs x 128 channels; its input is 4 planes x 6 rows x 58 pixels of act1 [n][18][58][58][64].  24 LDS-DMA
//    instructions (4 per slab row, 16 pixels each) bring 64 B (one channel half) of each pixel into a 24 KB plane buffer.
//  * K order: channel half cc (2) x tap (27, kz-major) x 32 channels.  A K step reads, per wave, 7 A fragments from
//    the patch at (row base + plane base) + immediate (ky pitch + 64 kx), and 4 B fragments from a ring of 8 KB filter
//    slabs (the packed filter of the general kernel, [128][tap*64 + c]).  L2 -> LDS per step: 8 KB of filter + 96 KB
//    of patch per 27 steps = 11.6 KB (was 40).
//  * plane buffers: the rows with dz = 0 read plane kz, those with dz = 1 plane kz + 1, so plane 0 is free after the
//    kz = 0 taps and plane 1 after kz = 1: the next sweep's planes 0 / 1 are fetched at the start of this sweep's
//    kz = 1 / kz = 2 tap groups and its planes 2, 3 at the start of its own kz = 0 group -- four buffers, no stall.
//  * banks, row order and pooling: see "Round 3" below (rounds 1-2 ordered the rows (pooling window, dz, dy, dx) on a row
//    pitch of 128 (mod 256) with the plane buffers 32 bytes apart (mod 256); the pooled 56 x 128 tile still goes through
//    LDS -- 8-byte writes of 4 adjacent channels per lane -- for 16-byte stores, which sit in the same vmcnt queue as the
//    DMA: the first two counted waits of the next tile also wait for them -- conservative, never early).
//  * schedule: the two-group staggered loop of igemm_wide.hip.h (waves 0-3 / 4-7, partners on a SIMD, half a step
//    apart: LOAD = fragment reads + DMA issue + counted wait, COMPUTE = 28 MFMAs), waves 4 (M) x 2 (N), 112 x 64 each.
//
// Round 3 -- dz-pure fragments and skipped halo-plane tap groups (the scheme of conv_patch14.hip.h, read its header):
//  * a 16-row fragment is 4 pooling windows x (dy, dx) of ONE output plane z = 2 zp + dz: the column pair (2 f, 2 f + 1) of
//    the tile's two pooled rows, rows 0-3 = (row 0, col 2f), 4-7 = (row 1, col 2f), 8-11 = (row 1, col 2f+1), 12-15 =
//    (row 0, col 2f+1).  A pair of M waves (m = wm & 1) shares 7 column pairs: accumulator slots 0 .. 3 of wave m hold plane
//    dz = m of pairs 3 m + i, slots 4 .. 6 plane 1 - m of pairs 4 m + i - 4.  (conv2a has two such wave pairs, one per
//    half of the 28 columns.)
//  * the plane slabs are fetched row by row (4 / 2 LDS-DMA instructions of 16 pixels per row) into rows of pitch 4128 /
//    2080 bytes = 32 (mod 256): with the fragment order above the 16 lanes of every ds_read_b128 group hit 16 different
//    16-byte slots for every tap (scripts/check_conv_patch_banks.py); no parity shift of the plane buffers any more.  The
//    fragment address is one register per dz set + immediates (256 bytes per column pair, ky pitch + 64 kx per tap).
//  * output plane z = 0 multiplies its kz = 0 taps with the zero halo plane z = -1, and z = DEPTH - 1 its kz = 2 taps with
//    the halo plane z = DEPTH: in the tiles of the first / last pooled plane the dz = 0 / dz = 1 fragments skip that tap
//    group (12 or 16 MFMAs per wave and step instead of 28): 2 of the 4 pooled planes of conv3a / conv3b.  conv2a (2 of 8)
//    measured no gain and runs every tap group; so does conv3b's training forward (register pressure, see the kernel).
//  * pooling: max over (dy, dx) in the lane's registers, over dz between two accumulator slots of the wave; column pair 3
//    has its two planes in different waves, which exchange fp32 maxima through the filter-ring slot that is idle during
//    the epilogue (conv_patch14.hip.h).
#pragma once
#include <type_traits>

#include "igemm.hip.h"

#ifndef RGP_MMA_ORDER
#define RGP_MMA_ORDER 1     // 1: filter fragment outermost in a step (7 consecutive MFMAs share it; 0: the activation fragment, 4): -0.5 % wall
#endif
#ifndef RGP_CP_CHUNK56
#define RGP_CP_CHUNK56 1     // clip windows per tile-order chunk, 56 x 56 planes (see decode() in the kernel)
#endif
#ifndef RGP_CP_CHUNK28
#define RGP_CP_CHUNK28 16    // ... 28 x 28 planes (8: conv3b 2 x FETCH 12.8 GB, 16 and 32: 11.1 GB; times equal)
#endif
#ifndef RGP_PLANE_AUX
#define RGP_PLANE_AUX 0      // cache policy of the plane-slab LDS-DMA (2 = nt; measured, see docs/HISTORY.md)
#endif

namespace rgp {

struct ConvPatchParams {
  const bf16_t* in;     // [n][D+2][HW+2][HW+2][CIN] halo-padded
  const bf16_t* wp;     // packed filter [NOUT][27*CIN], K index = ((c / 64) * 27 + tap) * 64 + c % 64
  const float* bias;    // [NOUT]
  bf16_t* out;          // [n][D/2+2][HW/2+2][HW/2+2][NOUT] halo-padded
  unsigned char* argmax;  // pooled layers, training plans: [n][D/2][HW/2][HW/2][NOUT] index (dz*4 + dy*2 + dx) of the first maximum
  const bf16_t* mask;     // DGRAD kernels: forward activation in the layout of `out`; the result is kept where it is > 0
  int n_windows;
  int ablate;             // dev builds (RGP_CP_ABLATE, timing only, results garbage): 1 plane fetches from one L2-resident
                          // slab, 2 no epilogue, 4 epilogue without its global stores, 8 filter slabs from one L2-resident slab
};

#ifdef RGP_DEV_KNOBS
#define RGP_CP_ABL(p, bit) (((p).ablate & (bit)) != 0)
#else
#define RGP_CP_ABL(p, bit) false
#endif

template <int CIN, int NOUT, int HW, int DEPTH, bool POOL = true> struct PatchCfg {
  static constexpr int WP = HW + 2;                       // padded row: 58 / 30 pixels
  static constexpr int XPN = HW / 2;                      // pooling windows per pooled row: 28 / 14
  static constexpr int NCC = CIN / 32;                    // channel sweeps: 2 / 8
  static constexpr int WIN = 2 * XPN;                     // pooling windows per tile: 56 / 28
  static constexpr int WMW = WIN / 14, WNW = 8 / WMW;     // waves along M (14 windows = 7 m-tiles each) and N
  static constexpr int RPI = (WP + 15) / 16;              // LDS-DMA instructions per slab row (16 pixels x 64 B each): 4 / 2
  static constexpr int LP = RPI * 1024 + 32;              // LDS row pitch: 4128 / 2080 bytes = 32 (mod 256)
  static constexpr int PPW = (6 * RPI + 7) / 8;           // plane-slab DMA instructions per wave: 3 / 2
  static constexpr int NDUMP = 8 * PPW - 6 * RPI;         // instructions beyond the 6 rows (0 / 4): they land in a dump area
  static constexpr int PLANE_BYTES = 6 * LP + NDUMP * 1024;
  static constexpr int PLANE_STRIDE = (PLANE_BYTES + 255) / 256 * 256;     // 24 832 / 16 640
  static constexpr int BRING_OFF = (4 * PLANE_STRIDE + 1023) / 1024 * 1024;
  static constexpr int NI = NOUT / (16 * WNW);            // 16-column MFMA tiles per wave: 4 (2 for the 64 / 128-channel
                                                          // input gradients of conv2a / conv3a: wave tile 112 x 32)
  static constexpr int BPW = (NOUT + 127) / 128;          // filter-slab DMA instructions per wave and step: 1 / 2
  static constexpr int BSLOT = BPW * 128 * 64;            // 8 / 16 KB: filter rows (padded to 128: the packing pads too) x 32 K elements
  static constexpr int NSLOT = 4, AHEAD = 3;
  static constexpr int STG_OFF = BRING_OFF + NSLOT * BSLOT;
  // staged pooled tile (bias + ReLU applied, bf16) [WIN][NOUT + 8] and its arg-max codes [WIN][NOUT + 8] bytes: an area of
  // its own (the filter ring keeps running across tiles)
  static constexpr int STG_LD = NOUT + 8;
  static constexpr int STGA_OFF = STG_OFF + WIN * STG_LD * 2;
  static constexpr int BIAS_OFF = POOL ? STGA_OFF + WIN * STG_LD : STG_OFF;      // the layer's biases (fp32), read by the epilogues
  static constexpr int SMEM = BIAS_OFF + NOUT * 4;
  static constexpr int NSTEP = NCC * 27;
  static constexpr int YT = HW / 4;                       // tiles per pooled plane
  static constexpr int TILES_PER_WINDOW = (DEPTH / 2) * YT;
  static constexpr int K = 27 * CIN;
  static constexpr int IN_ROW = WP * CIN, IN_PLANE = WP * IN_ROW, IN_IMG = (DEPTH + 2) * IN_PLANE;      // elements
  static constexpr int OW = POOL ? HW / 2 : HW, OD = POOL ? DEPTH / 2 : DEPTH;   // output extent
  static constexpr int OUT_ROW = (OW + 2) * NOUT, OUT_PLANE = (OW + 2) * OUT_ROW, OUT_IMG = (OD + 2) * OUT_PLANE;
  static constexpr int CGN = NOUT / 8;                    // epilogue: 8-channel groups
  static_assert(WNW * 16 * NI == NOUT && (NI == 4 || (NI == 2 && !POOL)) && WMW * 14 == WIN && XPN % 2 == 0 && HW % 4 == 0 && CIN % 32 == 0,
                "tile shape");
  static_assert(LP % 256 == 32, "row pitch = 32 (mod 256): the bank argument of the header");
  static_assert(SMEM <= 160 * 1024, "LDS budget");
  static_assert(!POOL || WIN * CGN == 2 * 448, "pooled epilogue: two items per thread (448 of the 512 threads)");
  static_assert((STG_LD * 2) % 16 == 0 && STG_LD % 8 == 0, "staging rows keep 16- / 8-byte alignment");
};

// max of MFMA results without the compiler's canonicalising `v_max_f32 x, x, x` in front of every operand (IEEE mode: the
// hardware instruction quiets a signalling NaN itself and, like fmaxf, returns the other operand for a NaN): 77 of the
// 224 v_max of conv2a's pooled epilogue were such moves once the kernels left the -fno-honor-nans translation unit
static __device__ __forceinline__ float cp_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
static __device__ __forceinline__ float cp_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
static __device__ __forceinline__ float cp_relu(float a) { float r; asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(a)); return r; }

template <int OFF>
static __device__ __forceinline__ f32x4 cp_lds_read128(unsigned addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

// In every variant MFMA column 16 j + c of a wave carries channel 64 wn + 4 c + j (the filter slab is fetched in that row
// order), so a lane holds 4 adjacent channels of a position.
// POOL: 2x2x2 max-pool epilogue (conv2a, conv3b): the lane's four pooled channels of a window are staged with one 8-byte
// LDS write (their arg-max codes with one 4-byte write; four 2-byte writes each before: conv2a -1.5 %).
// !POOL (conv3a): the same tiles -- the row order (2x2x2 blocks of positions) is immaterial to a convolution -- stored
// un-pooled: 8 bytes straight from registers, 16 lanes = 128 contiguous bytes.
// DGRAD (!POOL): the same convolution as the input gradient of a layer (in = dY before pooling, halo-padded; filter =
// the rotated, in/out-swapped one of the backward plan): no bias, no ReLU, the result masked by the forward activation.
// DENSE (DGRAD only): the output is the dense, un-masked [n][D*HW*HW][NOUT] image the un-pool kernel consumes (gradient
// w.r.t. a pooled layer's output).
template <int CIN, int NOUT, int HW, int DEPTH, bool POOL, bool ARGMAX = false, bool DGRAD = false, bool DENSE = false>
static __global__ __launch_bounds__(512) void conv_patch_bf16_kernel(const ConvPatchParams p) {
  static_assert(!DENSE || DGRAD, "dense output: input gradients only");
  static_assert(POOL || !ARGMAX, "arg-max codes belong to the pooled layers");
  static_assert(!POOL || !DGRAD, "the input gradient is an un-pooled convolution");
  using C = PatchCfg<CIN, NOUT, HW, DEPTH, POOL>;
  constexpr int NI = C::NI;
  extern __shared__ __attribute__((aligned(16))) char cp_smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)cp_smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / C::WNW, wn = wave % C::WNW;
  const bool group_b = wave >= 4;
  const int frow = lane & 15, fk = lane >> 4;
  const int wmm = wm & 1, wmh = wm >> 1;                     // member of its M-wave pair, column half (conv2a: 2 pairs)
  auto plane_base = [](int k) { return (unsigned)(k * C::PLANE_STRIDE); };

  // persistent tile walk: XCD x (workgroup id & 7) owns a contiguous range of tiles (neighbouring tiles share halo rows
  // and planes: one L2 serves them)
  const int nt = p.n_windows * C::TILES_PER_WINDOW;
  auto tile_of = [&](int t) {
    const int q = nt >> 3, r = nt & 7, x = t & 7, y = t >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
  };
  int t_seq = blockIdx.x;
  if (t_seq >= nt) return;
  // tile -> (clip window, pooled plane zp, tile row yp).  Tiles are numbered (chunk of RGP_CP_CHUNK56 / 28 windows, zp, window, yp):
  // runs of CHUNK x YT tiles of ONE pooled plane follow each other, so the CUs of an XCD work on tiles of equal length at
  // any time -- the tiles of the first / last pooled plane skip tap groups and are ~7 % shorter; mixed with the others
  // they let the CUs drift apart, which costs the L2 sharing of halo rows and planes between neighbouring tiles
  // (conv3b's traffic beyond L2 went 9.4 -> 18 GB with the (window, zp, yp) order)
  auto decode = [&](int tile, int& tn, int& zp, int& yp) {
    constexpr int NW = HW == 56 ? RGP_CP_CHUNK56 : RGP_CP_CHUNK28, TPW = C::TILES_PER_WINDOW;
    const int c = tile / (NW * TPW), r = tile - c * (NW * TPW);
    const int cw = min(NW, p.n_windows - c * NW);              // windows of this chunk (the last one may be short)
    zp = r / (cw * C::YT);
    const int r2 = r - zp * (cw * C::YT);
    const int w = r2 / C::YT;
    tn = c * NW + w;
    yp = r2 - w * C::YT;
  };
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
    int n, zp, yp;
    decode(tile, n, zp, yp);
    if (RGP_CP_ABL(p, 1)) return (const char*)(p.in + (long long)(blockIdx.x & 7) * C::IN_PLANE);
    return (const char*)(p.in + (long long)n * C::IN_IMG + (long long)(2 * zp + k) * C::IN_PLANE + (4 * yp) * C::IN_ROW + cc * 32);
  };
  // this wave's PPW of a plane slab's DMA instructions: job j = (slab row j / RPI, 16-pixel part j % RPI), 16 pixels x 64 B
  // each, to LDS row pitch LP; jobs past the 6 rows (28 x 28 planes: 4 of 16) re-fetch job 0 into the dump area.  The
  // source is a wave-uniform base + one 32-bit lane offset.
  const unsigned dlane = (unsigned)((lane >> 2) * (CIN * 2) + (lane & 3) * 16);
  auto dma_plane = [&](const char* src, int k, bool last_touch = false) {
#pragma unroll
    for (int u = 0; u < C::PPW; ++u) {
      const int j = wave * C::PPW + u;
      const bool real = j < 6 * C::RPI;
      const int r = real ? j / C::RPI : 0, q = real ? j - r * C::RPI : 0;
      const char* g = src + (long long)r * (C::IN_ROW * 2) + q * 16 * (CIN * 2) + dlane;
      char* l = cp_smem + plane_base(k) + (real ? r * C::LP + q * 1024 : 6 * C::LP + (j - 6 * C::RPI) * 1024);
      if (last_touch)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 2 /* nt */);
      else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, RGP_PLANE_AUX);
    }
  };
  // filter slab of K step (cc, tap): this wave's BPW of its 1-KB blocks (16 filter rows x 64 B), chunk-swizzled like
  // igemm_wide.hip.h (physical chunk c of row r holds logical chunk c ^ ((-(r >> 2)) & 3))
  const int brow = lane >> 2;
  const int bchk = (lane & 3) ^ ((-(brow >> 2)) & 3);
  // filter row (output channel) behind row brow of 1-KB block blk: blk * 16 + brow, or (!POOL) wave blk / NI, column
  // tile blk % NI: channel 16 NI (blk / NI) + NI brow + blk % NI (rows >= NOUT of a 64-channel filter are the packing's zeros)
  auto b_row = [&](int blk) { return (blk / NI) * (16 * NI) + brow * NI + (blk % NI); };
  unsigned b_off[C::BPW];
#pragma unroll
  for (int u = 0; u < C::BPW; ++u) b_off[u] = (unsigned)(b_row(wave * C::BPW + u) * (C::K * 2) + bchk * 16);
  auto dma_b = [&](int slot, int cc, int tap) {
    int koff = (((cc >> 1) * 27 + tap) * 64 + (cc & 1) * 32) * 2;
    if (RGP_CP_ABL(p, 8)) koff = 0;
    const char* base = (const char*)p.wp + koff;
#pragma unroll
    for (int u = 0; u < C::BPW; ++u)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + b_off[u]),
                                       (__attribute__((address_space(3))) void*)(cp_smem + C::BRING_OFF + slot * C::BSLOT + (wave * C::BPW + u) * 1024),
                                       16, 0, 0);
  };

  // fragment addressing (header, round 3).  Row frow of a fragment: window frow >> 2 of the column pair -- (pooled row,
  // column) = (0, 0), (1, 0), (1, 1), (0, 1) -- and (dy, dx) = ((frow >> 1) & 1, frow & 1); K chunk fk.  Column pair cp of the
  // tile (0 .. XPN / 2 - 1): accumulator slot i < 4 holds plane dz = wmm of pair 7 wmh + 3 wmm + i, slot i >= 4 plane 1 - wmm
  // of pair 7 wmh + 4 wmm + i - 4.  ra_lo / ra_hi: this lane's row of the first pair of either set, in the plane its dz reads
  // for the CURRENT tap group; advanced by one plane per tap group, taken back by two at the end of a sweep.
  const int r_w = frow >> 2, r_ypl = (r_w == 1 || r_w == 2) ? 1 : 0, r_xo = r_w >> 1;
  const int r_dy = (frow >> 1) & 1, r_dx = frow & 1;
  auto pair_of = [&](int i) { return 7 * wmh + (i < 4 ? 3 * wmm + i : 4 * wmm + i - 4); };
  unsigned ra_lo, ra_hi;
  {
    const unsigned base = lds0 + (2 * r_ypl + r_dy) * C::LP + (2 * r_xo + r_dx) * 64 + fk * 16;
    ra_lo = base + (7 * wmh + 3 * wmm) * 256 + plane_base(wmm);
    ra_hi = base + (7 * wmh + 4 * wmm) * 256 + plane_base(1 - wmm);
  }
  const unsigned b_addr = lds0 + C::BRING_OFF + (wn * NI) * 1024 + frow * 64 + ((fk ^ ((-(frow >> 2)) & 3)) << 4);

  // the biases go to LDS once (the epilogues read them from there: nothing epilogue-only stays in registers through the K
  // loop, and no global load sits in the epilogue, whose wait would also drain the look-ahead DMA)
  if constexpr (!DGRAD) { if (tid < NOUT) ((float*)(cp_smem + C::BIAS_OFF))[tid] = p.bias[tid]; }
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
    // instructions per wave each) are issued in its first LOAD phase.  MODE 0: all 7 accumulator slots; 1: only slots
    // 4 .. 6 (the first four multiply a halo plane in this tap group); 2: only slots 0 .. 3
    auto tap_group = [&](auto NPL_, auto MODE_, int cc, int kz, const char* pl_a, int ka, const char* pl_b, int kb, bool odd_sweep) {
      constexpr int NPL = decltype(NPL_)::value, MODE = decltype(MODE_)::value;
      constexpr int I0 = MODE == 1 ? 4 : 0, I1 = MODE == 2 ? 4 : 7;
      const int s0 = cc * 27 + kz * 9;
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {
        // ---------------- LOAD ----------------
        const int s = s0 + t9;
        f32x4 af[7], bf[NI];
        const unsigned bb = b_addr + slot * C::BSLOT;
        auto reads = [&](auto T9) {
          constexpr int t = decltype(T9)::value;
          constexpr int imm = (t / 3) * C::LP + (t % 3) * 64;
          if constexpr (I0 < 4) {
            af[0] = cp_lds_read128<imm>(ra_lo);
            af[1] = cp_lds_read128<imm + 256>(ra_lo);
            af[2] = cp_lds_read128<imm + 512>(ra_lo);
            af[3] = cp_lds_read128<imm + 768>(ra_lo);
          }
          if constexpr (I1 > 4) {
            af[4] = cp_lds_read128<imm>(ra_hi);
            af[5] = cp_lds_read128<imm + 256>(ra_hi);
            af[6] = cp_lds_read128<imm + 512>(ra_hi);
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
        for (int i = I0; i < I1; ++i) asm volatile("" : "+v"(af[i]));
#pragma unroll
        for (int j = 0; j < NI; ++j) asm volatile("" : "+v"(bf[j]));
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- COMPUTE ----------------
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < NI; ++j)
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
    int tn, zp, yp;
    decode(tile, tn, zp, yp);
    // this wave's skip mode: in the first pooled plane its dz = 0 slots (0 .. 3 of wave wmm = 0, 4 .. 6 of wmm = 1) idle
    // through the kz = 0 tap groups, in the last one its dz = 1 slots through kz = 2
    int mode_k0 = zp == 0 ? (wmm == 0 ? 1 : 2) : 0;
    int mode_k2 = zp == DEPTH / 2 - 1 ? (wmm == 0 ? 2 : 1) : 0;
    if (RGP_CP_ABL(p, 16)) mode_k0 = mode_k2 = 0;              // dev: the layout without skipping
    // conv3b's training forward (arg-max codes): with the three loop variants the register allocator spills a fragment
    // inside the COMPUTE phases (scripts/check_isa_waits.py); it runs all seven slots in every tap group
    if constexpr (ARGMAX && CIN == 256) mode_k0 = mode_k2 = 0;
    // conv2a (56 x 56 planes, 2 of its 8 pooled planes could skip): measured, skipping gains it nothing (14.10 vs 14.20 ms,
    // both -0.4 % against the round-2 layout) and the unequal tiles let the CUs drift apart: 2 x FETCH 8.2 -> 17.5 GB
    if constexpr (HW == 56) mode_k0 = mode_k2 = 0;
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
      constexpr bool LT = !DGRAD && CIN != 128;
      const char* p2 = plane_src(tile, cc, 2);
      const char* p3 = plane_src(tile, cc, 3);
      const bool lt0 = LT && (cc & 1) != 0, lt1 = LT && (ncc & 1) != 0;
      if (mode_k0 == 0) tap_group(I2{}, I0_{}, cc, 0, p2, 2, p3, 3, lt0);
      else if (mode_k0 == 1) tap_group(I2{}, I1{}, cc, 0, p2, 2, p3, 3, lt0);
      else tap_group(I2{}, I2{}, cc, 0, p2, 2, p3, 3, lt0);
      tap_group(I1{}, I0_{}, cc, 1, plane_src(ntile, ncc, 0), 0, nullptr, 0, lt1);
      const char* n1 = plane_src(ntile, ncc, 1);
      if (mode_k2 == 0) tap_group(I1{}, I0_{}, cc, 2, n1, 1, nullptr, 0, lt1);
      else if (mode_k2 == 1) tap_group(I1{}, I1{}, cc, 2, n1, 1, nullptr, 0, lt1);
      else tap_group(I1{}, I2{}, cc, 2, n1, 1, nullptr, 0, lt1);
    }
    if (!group_b) __builtin_amdgcn_s_barrier();               // the groups are level again

    // Epilogue-only lane values are re-derived HERE, per tile, from opaque copies of the lane ids: left visible, the
    // compiler hoists them (staging offsets, window coordinates, ...) out of the tile loop and then spills them across the
    // K loop -- and a scratch reload in the epilogue waits on vmcnt(0), i.e. for the whole look-ahead DMA.
    int e_fk = fk, e_frow = frow, e_tid = tid;
    asm volatile("" : "+v"(e_fk), "+v"(e_frow), "+v"(e_tid));
    // this lane's window of accumulator slot i: column pair pair_of(i), member fk -> pooled row ypl, column xp
    auto lane_window = [&](int i, int& ypl, int& xp) {
      ypl = (e_fk == 1 || e_fk == 2) ? 1 : 0;
      xp = 2 * pair_of(i) + (e_fk >> 1);
    };
    float b4[NI];                                             // bias of this lane's NI MFMA columns
#pragma unroll
    for (int q = 0; q < NI; ++q) b4[q] = DGRAD ? 0.f : ((const float*)(cp_smem + C::BIAS_OFF))[wn * (16 * NI) + e_frow * NI + q];
    const int cg = e_tid % C::CGN;                            // POOL epilogue, store pass: this thread's 8 output channels
    if (RGP_CP_ABL(p, 2)) {
#pragma unroll
      for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(acc[i][j]));
    } else if constexpr (POOL) {
      // ---- epilogue: pool in registers -- max over (dy, dx) = the four registers of an accumulator, max over dz = two
      // accumulator slots of this wave (column pair 3 of a wave pair: one slot here, one in the other wave, exchanged in
      // fp32 through the idle ring slot) -- then bias + ReLU, pooled bf16 tile (and arg-max codes dz 4 + dy 2 + dx, first
      // maximum) through LDS, 16-byte (8-byte) stores ----
      bf16_t* stg = (bf16_t*)(cp_smem + C::STG_OFF);
      unsigned char* stga = (unsigned char*)(cp_smem + C::STGA_OFF);
      float* xm = (float*)(cp_smem + C::BRING_OFF + ((slot + C::AHEAD) & (C::NSLOT - 1)) * C::BSLOT);   // idle until the next LOAD phase
      unsigned* xi = (unsigned*)(xm + C::WMW / 2 * C::WNW * 4 * 64);
      // maxima over (dy, dx) of accumulator slot i (and, ARGMAX, the member indices dy 2 + dx, 8 bits per n-tile) -- computed
      // slot by slot, right before use: all seven at once would pull the 112 accumulators through the vector registers
      auto pool_slot = [&](int i, float (&m)[4], unsigned& mi) {
        mi = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 c = acc[i][j];
          if constexpr (ARGMAX) {
            float best = c[0];
            unsigned idx = 0;
            if (c[1] > best) { best = c[1]; idx = 1; }
            if (c[2] > best) { best = c[2]; idx = 2; }
            if (c[3] > best) { best = c[3]; idx = 3; }
            m[j] = best;
            mi |= idx << (8 * j);
          } else {
            m[j] = cp_max(cp_max3(c[0], c[1], c[2]), c[3]);
          }
        }
      };
      // v0 / i0: the dz = 0 maxima (and member indices) of a slot's window, v1 / i1: dz = 1; stage the lane's 4 channels
      auto put = [&](int slot_i, const float (&v0)[4], unsigned i0, const float (&v1)[4], unsigned i1) {
        int ypl, xp;
        lane_window(slot_i, ypl, xp);
        const int so = (ypl * C::XPN + xp) * C::STG_LD + wn * 64 + 4 * e_frow;
        unsigned short pv[4];
        unsigned pc = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool hi = v1[j] > v0[j];
          pv[j] = f2bf(cp_relu((hi ? v1[j] : v0[j]) + b4[j]));
          if constexpr (ARGMAX) pc |= (hi ? ((i1 >> (8 * j)) & 0xffu) + 4u : ((i0 >> (8 * j)) & 0xffu)) << (8 * j);
        }
        uint2 o;
        o.x = (unsigned)pv[0] | ((unsigned)pv[1] << 16);
        o.y = (unsigned)pv[2] | ((unsigned)pv[3] << 16);
        *(uint2*)(stg + so) = o;
        if constexpr (ARGMAX) *(unsigned*)(stga + so) = pc;
      };
      const int xo = ((wmh * C::WNW + wn) * 4 + e_fk) * 64 + 4 * e_frow;   // exchange slot of (wave pair, wn, window fk)
      float m3[4];                                            // wave wmm = 0: its dz = 0 half of column pair 3, kept for the second pass
      unsigned mi3 = 0;
      if (wmm == 0) {
#pragma unroll
        for (int f = 0; f < 3; ++f) {
          float a0[4], a1[4];
          unsigned i0, i1;
          pool_slot(f, a0, i0);
          pool_slot(f + 4, a1, i1);
          put(f, a0, i0, a1, i1);
        }
        pool_slot(3, m3, mi3);
      } else {
#pragma unroll
        for (int f = 4; f < 7; ++f) {
          float a0[4], a1[4];
          unsigned i0, i1;
          pool_slot(f, a0, i0);
          pool_slot(f - 3, a1, i1);
          put(f, a0, i0, a1, i1);
        }
        // dz = 1 half of column pair 3 (slot 0 here) for wave wmm = 0 of the pair
        pool_slot(0, m3, mi3);
        *(f32x4*)(xm + xo) = (f32x4){m3[0], m3[1], m3[2], m3[3]};
        if constexpr (ARGMAX) xi[xo >> 2] = mi3;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // raw barrier: __syncthreads() would also drain the look-ahead DMA
      __builtin_amdgcn_s_barrier();
      if (wmm == 0) {
        const f32x4 q = *(const f32x4*)(xm + xo);
        const float v1[4] = {q[0], q[1], q[2], q[3]};
        unsigned i1 = 0;
        if constexpr (ARGMAX) i1 = xi[xo >> 2];
        put(3, m3, mi3, v1, i1);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      bf16_t* obase = p.out + (long long)tn * C::OUT_IMG + (zp + 1) * C::OUT_PLANE + (2 * yp + 1) * C::OUT_ROW + NOUT;
      unsigned char* abase = ARGMAX ? p.argmax + (((long long)tn * (DEPTH / 2) + zp) * (HW / 2) + 2 * yp) * (long long)((HW / 2) * NOUT) : nullptr;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int w = e_tid / C::CGN + (512 / C::CGN) * k;    // pooling window of the tile
        if (w < C::WIN && !RGP_CP_ABL(p, 4)) {
          const int ypl = w / C::XPN, xp = w - ypl * C::XPN;
          *(u32x4*)(obase + ypl * C::OUT_ROW + xp * NOUT + cg * 8) = *(const u32x4*)(stg + w * C::STG_LD + cg * 8);
          if constexpr (ARGMAX) *(uint2*)(abase + (ypl * (HW / 2) + xp) * NOUT + cg * 8) = *(const uint2*)(stga + w * C::STG_LD + cg * 8);
        }
      }
    } else {
      // ---- epilogue: bias + ReLU, 8-byte stores from registers.  Accumulator slot i, register e: this lane's window of
      // the slot's column pair, plane dz of the slot, dy = e >> 1, dx = e & 1; channels 16 NI wn + NI frow + 0 .. NI-1 ----
      // position of (window, dz, dy, dx): halo-padded image, or (DENSE) natural (z, y, x) order without halo
      constexpr int ROWS = DENSE ? HW * NOUT : C::OUT_ROW, PLANES = DENSE ? HW * ROWS : C::OUT_PLANE;
      constexpr long long IMG = DENSE ? (long long)DEPTH * PLANES : (long long)C::OUT_IMG;
      constexpr int H1 = DENSE ? 0 : 1;
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        int ypl, xp;
        lane_window(i, ypl, xp);
        const int dz = i < 4 ? wmm : 1 - wmm;
        const long long ow = (long long)tn * IMG + (2 * zp + H1 + dz) * PLANES + (4 * yp + 2 * ypl + H1) * ROWS + (2 * xp + H1) * NOUT +
                             wn * (16 * NI) + e_frow * NI;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const long long oe = ow + (e >> 1) * ROWS + (e & 1) * NOUT;
          float v[NI];
#pragma unroll
          for (int j = 0; j < NI; ++j) v[j] = DGRAD ? acc[i][j][e] : fmaxf(acc[i][j][e] + b4[j], 0.f);
          if constexpr (DGRAD && !DENSE) {
            unsigned mk[NI / 2];
            if constexpr (NI == 4) { const uint2 mm = *(const uint2*)(p.mask + oe); mk[0] = mm.x; mk[1] = mm.y; }
            else mk[0] = *(const unsigned*)(p.mask + oe);
#pragma unroll
            for (int j = 0; j < NI; ++j)
              if (!(bf2f((bf16_t)((mk[j >> 1] >> (16 * (j & 1))) & 0xffffu)) > 0.f)) v[j] = 0.f;
          }
          if constexpr (NI == 4) {
            uint2 o;
            o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
            o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
            *(uint2*)(p.out + oe) = o;
          } else {
            *(unsigned*)(p.out + oe) = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
          }
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
