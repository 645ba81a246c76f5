// C3D conv1a (3x3x3, 3->64, pad 1) + bias + ReLU + pool1 (1x2x2 max) for gfx950, bf16.
// Spec: /root/reference/C3D/.../c3d_prototxt/feature_extration.prototxt:22-66.
//
// conv1a has K = 81: far too short for the LDS-staged implicit-GEMM tile loop (three K-chunks per tile, so
// tile set-up, barriers and the epilogue dominate), and its output (6.4 MB per window) is 5x its input, so the
// kernel is organised around streaming, and -- the MFMA time of a tile being only 448 cycles -- around the number of
// instructions a wave issues per tile (MI355X guide, 'vector-instruction ISSUE cost': every VALU / LDS / s_nop costs
// 4 issue cycles of the SIMD, a 16x16x32 MFMA 8 of its 16, a 32x32x16 MFMA 8 of its 32):
//
//  * input  act0 [n][18][114][116][4] bf16 (halo-padded, channels 3->4, 8 bytes per pixel), or (FUSED) the caller's
//    fp32 windows converted on the fly.
//  * a block (4 waves) walks COLUMNS = (window, 4 pooled rows) through the 16 planes z: the input of a job (window, z,
//    4 pooled rows) is 3 planes x 10 rows x 116 pixels, and consecutive z share two of the three planes, so LDS holds
//    a ring of 4 plane slabs: every job fetches ONE new slab (10 rows, with coalesced 16-byte loads into registers
//    while the current job computes; one barrier per job).  Round 1's kernel fetched all three planes for every job
//    and left the overlap to L2, which the 6.4 MB / window of output streaming through it evicts: FETCH_SIZE was
//    2.2x the input.
//  * slab rows have a pitch of 144 pixel slots = 1152 B = 128 (mod 256): the two image rows a fragment read touches
//    (16 pixels = 128 B each) fall on disjoint halves of the 64 banks.
//  * K is ordered (kz,ky,kx | c4): 27 taps x 4 channels = 108, padded to 112 = 7 k-steps of v_mfma_f32_32x32x16_bf16
//    (round 1: 128 = 4 steps of 16x16x32, twice the MFMA instructions for the same rows).  A lane's 8-element
//    A-fragment is two taps = two 8-byte pixels = two ds_read_b64.
//  * bias and ReLU ride in the GEMM: the 4th channel of every parked pixel is 1.0, the 4th-channel weight of the
//    centre tap is bf16(bias) and that of the padding tap 27 -- which reads the centre pixel again -- is the bf16
//    remainder (bias to 2^-17 relative); ReLU is the 0 in the second max3 of the pooling.
//  * filter [64][128] bf16 -> 14 B-fragments (56 VGPRs) loaded once per wave.
//  * wave tile = 32 conv rows = 8 pooled pixels x 64 channels; wave w of a block owns pooled row w of the job and walks
//    its 7 tiles in x, so every fragment read is (one per-lane base VGPR) + (compile-time offset).  Rows are ordered
//    (pooling window, dy, dx), so the 4 rows of a window are 4 consecutive accumulator registers of one lane
//    (32x32 C layout: row = (reg&3) + 8 (reg>>2) + 4 (lane>>5)): pool1 + ReLU = two v_max3_f32 per value.
//  * MFMA column c of n-tile nt carries output channel 2c + nt (the filter fragments are loaded in that order), so a
//    lane ends a tile with 2 ADJACENT channels of 4 pooled pixels; lane pairs trade a dword so that each lane stores
//    8 bytes (4 channels) of one pixel straight from registers, 16 lanes = one pixel's 128 B -- no LDS transpose.
//  * the fragment reads are inline asm: left to itself the compiler fuses pairs of them into ds_read2_b64, which
//    banks mod 32 and takes 8 LDS cycles where two ds_read_b64 take 4 (round 1's kernel: 50 % of its LDS cycles were
//    bank conflicts).
//  * this translation unit is compiled with -fno-honor-nans: fmaxf() on MFMA results otherwise costs a quieting
//    v_max_f32 x, x per operand under the IEEE mode (5 instructions per pooled value instead of 2).
//  * columns are dealt to the 8 XCDs in contiguous ranges (consecutive workgroup ids sit on different XCDs), so the
//    y halo between neighbouring columns is served by one XCD's L2.
#pragma once
#include "igemm.hip.h"

namespace rgp {

struct Conv1aParams {
  const float* video;   // FUSED variant: the caller's mean-subtracted clip windows [n][16][112][112][3] fp32, read directly
                        // (the separate video_prep pass -- 2.4 MB read + 1.9 MB written per window, 0.93 ms per 1024
                        // windows at HBM rate -- is folded into the slab fetch)
  const bf16_t* in;     // [n][18][114][116][4]
  const bf16_t* wp;     // [64][128]
  const float* bias;    // [64]
  bf16_t* out;          // [n][18][58][58][64] (halo-padded input of conv2a)
  unsigned char* argmax;  // optional [n][16][56][56][64]: dy*2+dx of the first maximum of each pool1 window, + 4 where the
                          // pooled output stored beside it is zero (ReLU off: no gradient passes; conv1a_wgrad.hip.h then
                          // need not read the activation back for its gate)
  int n_windows;
};

constexpr int C1_D = 16, C1_H = 112, C1_HP = 114, C1_WP = 116, C1_K = 128;
constexpr int C1_KSTEPS = 7;                    // 7 x 16 = 112 >= 27 taps x 4
constexpr int C1_PO = 56;                       // pooled extent
constexpr int C1_XG = C1_PO / 8;                // 8 pooled pixels per wave tile
constexpr int C1_OUT_P = 58;
constexpr int C1_JROWS = 4;                     // pooled rows per job
constexpr int C1_YQ = C1_PO / C1_JROWS;         // 14 columns per window
constexpr int C1_PROWS = 2 * C1_JROWS + 2;      // 10 input rows per slab
constexpr int C1_ROWB = C1_WP * 8;              // 928 bytes per act0 row
constexpr int C1_WPL = 144;                     // LDS slab row pitch in pixel slots (1152 B)
constexpr int C1_SLAB = C1_PROWS * C1_WPL * 8;  // 11 520 bytes
constexpr int C1_SMEM = 4 * C1_SLAB;            // ring of 4 slabs
constexpr int C1_STREAM = C1_D + 2;             // slabs per column: planes -1 .. 16
// act0 variant: a slab row = the 116 pixels of an act0 row (slot 0 = x -1) = 58 16-byte chunks
constexpr int C1_CPR = C1_WP * 8 / 16;                    // 58 chunks per row
constexpr int C1_CHUNKS = C1_PROWS * C1_CPR;              // 580
constexpr int C1_NLD = (C1_CHUNKS + 255) / 256;           // 3 loads per thread
// FUSED variant: slot 1 = x -1 (zero), slots 2..113 = x 0..111, so that the 4-pixel groups converted from fp32 land
// on 16-byte boundaries; every slot the fetch does not write stays zero for the whole kernel
constexpr int C1F_GROUPS = C1_PROWS * 28;                 // 280 groups of 4 pixels (12 floats = three 16-byte loads)
constexpr int C1F_NLD = (C1F_GROUPS + 255) / 256;         // 2 per thread

template <int V> struct C1Int { static constexpr int value = V; };
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x8_c1 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int OFF>
static __device__ __forceinline__ u32x2 c1_lds_read64(unsigned addr) {
  u32x2 v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
// all LDS reads issued so far have landed; the fragments pass through the statement so that no use is scheduled above it
static __device__ __forceinline__ void c1_lds_wait(u32x2 (&a)[C1_KSTEPS][2]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[3][0]),
                 "+v"(a[3][1]), "+v"(a[4][0]), "+v"(a[4][1]), "+v"(a[5][0]), "+v"(a[5][1]), "+v"(a[6][0]), "+v"(a[6][1]));
}
static constexpr __host__ __device__ int c1_tap_kz(int tap) { return (tap > 26 ? 13 : tap) / 9; }

// VAR (dev builds only): 2 = no output stores, 4 = no slab fetch after the first (timing ablations)
template <bool FUSED, bool ARGMAX, int VAR = 0>
static __global__ __launch_bounds__(256, 2) void conv1a_pool_bf16_kernel(const Conv1aParams p) {
  constexpr int WPL = C1_WPL;
  constexpr int XS = FUSED ? 1 : 0;                        // slot of x = -1
  extern __shared__ __attribute__((aligned(16))) char c1_smem[];
  char* ring = c1_smem;                                                    // [4][C1_SLAB]
  for (int i = threadIdx.x; i < C1_SMEM / 16; i += 256) ((u32x4*)ring)[i] = (u32x4){0u, 0u, 0u, 0u};
  __syncthreads();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 31, kh = lane >> 5;

  // columns of this block: XCD x owns the contiguous range [x*per_xcd, (x+1)*per_xcd)
  const long long total = (long long)p.n_windows * C1_YQ;
  const long long per_xcd = (total + 7) / 8;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  long long col = xcd * per_xcd + slot;
  long long col_end = (xcd + 1) * per_xcd;
  if (col_end > total) col_end = total;
  if (col >= col_end) return;

  // filter fragments: B[k = 16s + 8kh + j][MFMA column frow of n-tile nt = output channel 2 frow + nt]; the bias
  // goes into the 4th-channel slots of the centre tap (k = 55: s 3, kh 0, j 7) and of tap 27 (k = 111: s 6, kh 1, j 7)
  u32x4 bfrag[C1_KSTEPS][2];
#pragma unroll
  for (int s = 0; s < C1_KSTEPS; ++s)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
      bfrag[s][nt] = *(const u32x4*)(p.wp + (frow * 2 + nt) * C1_K + s * 16 + kh * 8);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const float b = p.bias[frow * 2 + nt];
    const bf16_t hi = f2bf(b);
    const bf16_t lo = f2bf(b - bf2f(hi));
    if (kh == 0) bfrag[3][nt][3] = (bfrag[3][nt][3] & 0xffffu) | ((unsigned)hi << 16);
    else bfrag[6][nt][3] = (bfrag[6][nt][3] & 0xffffu) | ((unsigned)lo << 16);
  }

  // ---- slab fetch: stream position (column c = (window n, row quarter yq), plane pz = z + 1 in 0..17) ----
  // act0 variant: slab chunk q = tid + 256 u: where it comes from (elements, relative to the slab's first row) and
  // where it goes in the slab (bytes)
  int g_off[C1_NLD], l_off[C1_NLD];
#pragma unroll
  for (int u = 0; u < C1_NLD; ++u) {
    int q = tid + 256 * u;
    if (q >= C1_CHUNKS) q = C1_CHUNKS - 1;                 // duplicate, never written
    const int ry = q / C1_CPR, c = q - ry * C1_CPR;
    g_off[u] = (ry * C1_WP) * 4 + c * 8;
    l_off[u] = ry * (WPL * 8) + c * 16;
  }
  auto fetch = [&](long long c, int pz, u32x4 (&pf)[C1_NLD]) {
    const int yq = (int)(c % C1_YQ);
    const long long n = c / C1_YQ;
    const bf16_t* src = p.in + (((n * (C1_D + 2) + pz) * C1_HP + yq * 2 * C1_JROWS) * (long long)C1_WP) * 4;
#pragma unroll
    for (int u = 0; u < C1_NLD; ++u) pf[u] = *(const u32x4*)(src + g_off[u]);
  };
  auto park = [&](char* slab, const u32x4 (&pf)[C1_NLD]) {
#pragma unroll
    for (int u = 0; u < C1_NLD; ++u)
      if (tid + 256 * u < C1_CHUNKS) {
        u32x4 v = pf[u];
        v[1] |= 0x3f800000u;                               // 4th channel = 1.0 (act0 carries 0 there)
        v[3] |= 0x3f800000u;
        *(u32x4*)(slab + l_off[u]) = v;
      }
  };
  // FUSED: group q = tid + 256 u of the slab = row ry = q / 28, pixels 4 (q % 28)..+3 -> three 16-byte loads of 12
  // floats from the fp32 window.  The loads are buffer loads on the window (num_records = one window); planes -1 and
  // 16 and the image rows -1 and 112 (first / last column of a window) get an out-of-range offset -> zeros.
  constexpr unsigned WIN_BYTES = C1_D * C1_H * C1_H * 12u;
  unsigned f_src[C1F_NLD], f_dst[C1F_NLD], f_edge = 0;     // edge bits: u -> group is slab row 0, 4+u -> slab row 9
#pragma unroll
  for (int u = 0; u < C1F_NLD; ++u) {
    int q = tid + 256 * u;
    if (q >= C1F_GROUPS) q = C1F_GROUPS - 1;                 // duplicate, never written
    const int ry = q / 28, xg = q - ry * 28;
    f_src[u] = (unsigned)((ry * C1_H + 4 * xg) * 12);
    f_dst[u] = (unsigned)((ry * WPL + 2 + 4 * xg) * 8);
    if (ry == 0) f_edge |= 1u << u;
    if (ry == C1_PROWS - 1) f_edge |= 16u << u;
  }
  auto fetch_f = [&](long long c, int pz, f32x4 (&pf)[C1F_NLD][3]) {
    const int yq = (int)(c % C1_YQ);
    const long long n = c / C1_YQ;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.video + n * (long long)(WIN_BYTES / 4)), 0,
                                                        WIN_BYTES, 0x00020000);
    const bool plane_ok = pz >= 1 && pz <= C1_D;
    const unsigned job_off = (unsigned)((((pz - 1) * C1_H + yq * 2 * C1_JROWS - 1) * C1_H) * 12);
    const unsigned edge = plane_ok ? f_edge & ((yq == 0 ? 0xfu : 0u) | (yq == C1_YQ - 1 ? 0xf0u : 0u)) : 0xffu;
#pragma unroll
    for (int u = 0; u < C1F_NLD; ++u) {
      const unsigned off = (edge & (0x11u << u)) ? 0x80000000u : f_src[u] + job_off;
#pragma unroll
      for (int k = 0; k < 3; ++k)
        pf[u][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16 * k, 0, 0));
    }
  };
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  auto pk = [](float x, float y) -> unsigned {               // v_cvt_pk_bf16_f32
    const f32x2_t v = {x, y};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
  };
  auto park_f = [&](char* slab, const f32x4 (&pf)[C1F_NLD][3]) {
#pragma unroll
    for (int u = 0; u < C1F_NLD; ++u) {
      if (tid + 256 * u < C1F_GROUPS) {
        const float f[12] = {pf[u][0][0], pf[u][0][1], pf[u][0][2], pf[u][0][3], pf[u][1][0], pf[u][1][1],
                             pf[u][1][2], pf[u][1][3], pf[u][2][0], pf[u][2][1], pf[u][2][2], pf[u][2][3]};
        // two packed conversions per pixel: {c0, c1}, {c2, 1.0}
        u32x4 w0, w1;
#pragma unroll
        for (int px = 0; px < 4; ++px) {
          const unsigned lo = pk(f[3 * px], f[3 * px + 1]), hi = pk(f[3 * px + 2], 1.f);
          if (px < 2) { w0[2 * px] = lo; w0[2 * px + 1] = hi; } else { w1[2 * (px - 2)] = lo; w1[2 * (px - 2) + 1] = hi; }
        }
        char* dst = slab + f_dst[u];
        *(u32x4*)dst = w0;
        *(u32x4*)(dst + 16) = w1;
      }
    }
  };

  // ---- fragment addressing (bytes).  Row m = frow of the tile: window w = m>>2, dy = (m>>1)&1, dx = m&1; k-step s,
  // half kh = taps 4s + 2kh, 4s + 2kh + 1 (tap 27 = the centre pixel again, for the bias remainder).  ra0: offset
  // inside a slab; the slab of tap plane kz rotates with the job (ring), added per job.
  const int r_w = frow >> 2, r_dy = (frow >> 1) & 1, r_dx = frow & 1;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)c1_smem;
  const unsigned lane_base = lds0 + ((r_dy + 2 * wave) * WPL + 2 * r_w + r_dx + XS) * 8;
  unsigned ra0[C1_KSTEPS][2];
#pragma unroll
  for (int s = 0; s < C1_KSTEPS; ++s)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int tap = 4 * s + 2 * kh + j;
      if (tap > 26) tap = 13;
      const int ky = (tap / 3) % 3, kx = tap % 3;
      ra0[s][j] = (ky * WPL + kx) * 8 + lane_base;
    }

  // fragments of tile OFF / 128 of this wave's row; c1_lds_wait() before the first use
  auto load_frags = [&](auto OFF, const unsigned (&ra)[C1_KSTEPS][2], u32x2 (&a)[C1_KSTEPS][2]) {
    constexpr int off = decltype(OFF)::value;
#pragma unroll
    for (int s = 0; s < C1_KSTEPS; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j) a[s][j] = c1_lds_read64<off>(ra[s][j]);
  };
  // this lane's part of the output address: pixel kh of a pixel pair, channels 2 frow, 2 frow + 1
  const int o_lane = kh * 64 + frow * 2;
  auto tile = [&](bf16_t* orow, unsigned char* arow, int xg, const u32x2 (&a)[C1_KSTEPS][2]) {
    f32x16 acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
#pragma unroll
    for (int s = 0; s < C1_KSTEPS; ++s) {
      const u32x4 av = {a[s][0][0], a[s][0][1], a[s][1][0], a[s][1][1]};
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8_c1, av),
                                                          __builtin_bit_cast(s16x8_c1, bfrag[s][nt]), acc[nt], 0, 0, 0);
    }
    unsigned dpk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {                            // pooled pixel 2j + kh of the tile
      float v[2];
      unsigned codes = 0;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const float c0 = acc[nt][4 * j], c1 = acc[nt][4 * j + 1], c2 = acc[nt][4 * j + 2], c3 = acc[nt][4 * j + 3];
        v[nt] = fmaxf(fmaxf(fmaxf(fmaxf(c0, c1), c2), c3), 0.f);      // pool1 + ReLU (the bias is in the sums): 2 v_max3_f32
        if constexpr (ARGMAX) {
          float best = c0;
          unsigned idx = 0;
          if (c1 > best) { best = c1; idx = 1; }
          if (c2 > best) { best = c2; idx = 2; }
          if (c3 > best) { best = c3; idx = 3; }
          codes |= idx << (8 * nt);
        }
      }
      dpk[j] = pk(v[0], v[1]);
      if constexpr (ARGMAX) {
        // the gate is taken from the bf16 value as stored (what every other consumer of the activation sees)
        if ((dpk[j] & 0x7fffu) == 0) codes |= 4u;
        if ((dpk[j] & 0x7fff0000u) == 0) codes |= 4u << 8;
        *(unsigned short*)(arow + (xg * 8 + 2 * j) * 64 + o_lane) = (unsigned short)codes;
      }
    }
    // the lane pair (2c, 2c+1) holds channels 4c .. 4c+3 of the 4 pixels: each trades one dword per pixel pair with its
    // partner (DPP quad_perm [1,0,3,2]) and stores 8 bytes of ONE pixel -- half the store instructions of the 4-byte
    // form (measured -3 %)
    if (!(VAR & 2) || p.n_windows < 0) {
      const bool odd = (frow & 1) != 0;
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        const unsigned send = odd ? dpk[2 * pp] : dpk[2 * pp + 1];
        const unsigned recv = (unsigned)__builtin_amdgcn_mov_dpp((int)send, 0xB1, 0xF, 0xF, true);
        uint2 o;
        o.x = odd ? recv : dpk[2 * pp];
        o.y = odd ? dpk[2 * pp + 1] : recv;
        *(uint2*)(orow + (xg * 8 + 2 * (2 * pp + (odd ? 1 : 0))) * 64 + kh * 64 + (frow & ~1) * 2) = o;
      }
    }
  };
  // job (column c, plane z) with tap plane kz in ring slab (rb + kz) & 3: this wave's pooled row, 7 tiles, fragments
  // of tile i+1 in flight under tile i.  (Measured and not kept: tile i's pooling issued behind tile i+1's MFMAs with two
  // accumulator sets, +10 %; three blocks per CU at 168 registers, +30 %; non-temporal stores, +-0.)
  auto compute = [&](long long c, int z, int rb) {
    const unsigned so[3] = {(unsigned)(rb & 3) * C1_SLAB, (unsigned)((rb + 1) & 3) * C1_SLAB, (unsigned)((rb + 2) & 3) * C1_SLAB};
    unsigned ra[C1_KSTEPS][2];
#pragma unroll
    for (int s = 0; s < C1_KSTEPS; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j) ra[s][j] = ra0[s][j] + (kh ? so[c1_tap_kz(4 * s + 2 + j)] : so[c1_tap_kz(4 * s + j)]);
    const int yo = (int)(c % C1_YQ) * C1_JROWS + wave;
    const long long n = c / C1_YQ;
    bf16_t* orow = p.out + (((n * (C1_D + 2) + z + 1) * C1_OUT_P + yo + 1) * (long long)C1_OUT_P + 1) * 64;
    unsigned char* arow = ARGMAX ? p.argmax + (((n * C1_D + z) * C1_PO + yo) * (long long)C1_PO) * 64 : nullptr;
    u32x2 a0[C1_KSTEPS][2], a1[C1_KSTEPS][2];
    load_frags(C1Int<0>{}, ra, a0);
    c1_lds_wait(a0);
    load_frags(C1Int<128>{}, ra, a1);
    tile(orow, arow, 0, a0);
    c1_lds_wait(a1);
    load_frags(C1Int<256>{}, ra, a0);
    tile(orow, arow, 1, a1);
    c1_lds_wait(a0);
    load_frags(C1Int<384>{}, ra, a1);
    tile(orow, arow, 2, a0);
    c1_lds_wait(a1);
    load_frags(C1Int<512>{}, ra, a0);
    tile(orow, arow, 3, a1);
    c1_lds_wait(a0);
    load_frags(C1Int<640>{}, ra, a1);
    tile(orow, arow, 4, a0);
    c1_lds_wait(a1);
    load_frags(C1Int<768>{}, ra, a0);
    tile(orow, arow, 5, a1);
    c1_lds_wait(a0);
    tile(orow, arow, 6, a0);
  };

  // ---- the slab stream: position i of this block = (column, plane pz) lives in ring slab i & 3; the job of plane
  // z = pz - 2 runs once slab i is parked, on slabs i-2, i-1, i, while slab i+1 is in flight in registers; it is parked
  // into slab (i+1) & 3 = (i-3) & 3, last read by the previous job, which every wave has left (the barrier).
  u32x4 pf[C1_NLD];
  f32x4 pff[C1F_NLD][3];
  int pz = 0, i = 0;
  if constexpr (FUSED) { fetch_f(col, 0, pff); park_f(ring, pff); }
  else { fetch(col, 0, pf); park(ring, pf); }
  __syncthreads();
  while (true) {
    const bool wrap = pz == C1_STREAM - 1;
    const long long ncol = wrap ? col + nslot : col;
    const int npz = wrap ? 0 : pz + 1;
    const bool more = ncol < col_end;
    if (more && (!(VAR & 4) || i < 4)) { if constexpr (FUSED) fetch_f(ncol, npz, pff); else fetch(ncol, npz, pf); }
    if (pz >= 2) compute(col, pz - 2, i - 2);
    if (!more) break;
    char* slab = ring + ((i + 1) & 3) * C1_SLAB;
    if constexpr (FUSED) park_f(slab, pff); else park(slab, pf);
    __syncthreads();                                // the new slab is visible; everybody is done with the oldest one
    ++i;
    col = ncol;
    pz = npz;
  }
}

}  // namespace rgp
