// C3D conv1a (3x3x3, 3->64, pad 1) + bias + ReLU + pool1 (1x2x2 max) for gfx950, bf16.
// Spec: /root/reference/C3D/.../c3d_prototxt/feature_extration.prototxt:22-66.
//
// conv1a has K = 81: far too short for the LDS-staged implicit-GEMM tile loop (three
// K-chunks per tile, so tile set-up, barriers and the LDS epilogue dominate).  This
// kernel keeps the WHOLE filter in registers and streams activations straight from
// global memory into MFMA A-fragments, one wave per tile, no block-level barrier:
//
//  * input  act0 [n][18][114][116][4] bf16 (halo-padded, channels 3->4): the 3 kx taps x
//    4 channels (+1 zero-weight pixel) of one (kz,ky) are 16 contiguous elements, so with
//    K ordered (kz,ky | kx,c) a lane's 8-element A-fragment is ONE aligned-enough 16-byte
//    load.  K = 10 taps x 16 = 160 (tap 9 and kx = 3 carry zero weights) = 5 MFMA k-steps.
//  * filter packed [64][160] bf16 -> 20 B-fragments (80 VGPRs) loaded once per wave.
//  * wave tile = 32 conv rows = 8 pooled pixels x 64 channels.  Rows are ordered
//    (pooling window, dy, dx), so the 4 rows of a window are the 4 accumulator registers
//    of one lane (v_mfma_f32_16x16x32 C layout: row = 4*(lane>>4)+reg): pool1 is an
//    in-lane max, then bias + ReLU.
//  * the pooled 8x64 bf16 tile is transposed through 1 KiB of wave-private LDS so the
//    store is one fully coalesced 1 KiB run (8 adjacent pixels x 128 B).
//  * next tile's 10 fragment loads are issued before the current tile's 40 MFMAs.
#pragma once
#include "igemm.hip.h"

namespace rgp {

typedef f32x4 __attribute__((aligned(8))) f32x4_a8;   // fragment loads are only 8-byte aligned

struct Conv1aParams {
  const bf16_t* in;     // [n][18][114][116][4]
  const bf16_t* wp;     // [64][160]
  const float* bias;    // [64]
  bf16_t* out;          // [n][18][58][58][64] (halo-padded input of conv2a)
  unsigned char* argmax;  // optional [n][16][56][56][64]: dy*2+dx of the first maximum of each pool1 window
  int n_windows;
};

constexpr int C1_D = 16, C1_H = 112, C1_HP = 114, C1_WP = 116, C1_K = 160;
constexpr int C1_PO = 56;                       // pooled extent
constexpr int C1_XG = C1_PO / 8;                // 8 pooled pixels per wave tile
constexpr int C1_TILES_PER_WINDOW = C1_D * C1_PO * C1_XG;
constexpr int C1_OUT_P = 58;

__global__ __launch_bounds__(256, 2) void conv1a_pool_bf16_kernel(const Conv1aParams p) {
  __shared__ __attribute__((aligned(16))) bf16_t s_out[4][8 * 72];   // per wave: 8 px x (64 ch + 8 pad)
  __shared__ __attribute__((aligned(16))) unsigned char s_arg[4][8 * 72];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int frow = lane & 15, kg = lane >> 4;
  const long long total = (long long)p.n_windows * C1_TILES_PER_WINDOW;
  const long long stride = (long long)gridDim.x * 4;

  // filter fragments: B[k = 32s + 8kg + j][n = 16jn + frow]
  f32x4 bfrag[5][4];
#pragma unroll
  for (int s = 0; s < 5; ++s)
#pragma unroll
    for (int jn = 0; jn < 4; ++jn)
      bfrag[s][jn] = *(const f32x4*)(p.wp + (jn * 16 + frow) * C1_K + s * 32 + kg * 8);
  float bias_v[4];
#pragma unroll
  for (int jn = 0; jn < 4; ++jn) bias_v[jn] = p.bias[jn * 16 + frow];

  // per-lane tap offsets (elements): tap = 2s + (kg>>1); tap 9 has zero weights -> any valid address
  int tapoff[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    int tap = 2 * s + (kg >> 1);
    if (tap > 8) tap = 0;
    tapoff[s] = ((tap / 3) * C1_HP + (tap % 3)) * C1_WP * 4 + (kg & 1) * 8;
  }
  // row inside an m-tile: window w = frow>>2, dy = (frow>>1)&1, dx = frow&1
  const int r_w = frow >> 2, r_dy = (frow >> 1) & 1, r_dx = frow & 1;

  auto tile_base = [&](long long t, int mi) -> long long {
    const int xg = (int)(t % C1_XG);
    const int yo = (int)((t / C1_XG) % C1_PO);
    const int z = (int)((t / (C1_XG * C1_PO)) % C1_D);
    const long long n = t / C1_TILES_PER_WINDOW;
    const int x = 2 * (xg * 8 + mi * 4 + r_w) + r_dx, y = 2 * yo + r_dy;
    return (((n * (C1_D + 2) + z) * C1_HP + y) * (long long)C1_WP + x) * 4;
  };
  auto load_tile = [&](long long t, f32x4 (&a)[2][5]) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const bf16_t* base = p.in + tile_base(t, mi);
#pragma unroll
      for (int s = 0; s < 5; ++s) a[mi][s] = *(const f32x4_a8*)(base + tapoff[s]);
    }
  };
  auto process = [&](long long t, const f32x4 (&a)[2][5]) {
    f32x4 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) acc[mi][jn] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) Mma<bf16_t>::step(acc[mi][jn], a[mi][s], bfrag[s][jn]);
    bf16_t* so = s_out[wave];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) {
        const f32x4 c = acc[mi][jn];
        const float v = fmaxf(fmaxf(fmaxf(c[0], c[1]), fmaxf(c[2], c[3])) + bias_v[jn], 0.f);   // pool1, bias, ReLU
        so[(mi * 4 + kg) * 72 + jn * 16 + frow] = f2bf(v);
        if (p.argmax) {
          float best = c[0];
          unsigned char idx = 0;
#pragma unroll
          for (int r = 1; r < 4; ++r)
            if (c[r] > best) { best = c[r]; idx = (unsigned char)r; }
          s_arg[wave][(mi * 4 + kg) * 72 + jn * 16 + frow] = idx;
        }
      }
    __builtin_amdgcn_wave_barrier();   // wave-private LDS: DS ops of one wave execute in order
    const int px = lane >> 3, chunk = lane & 7;
    const u32x4 row = *(const u32x4*)(so + px * 72 + chunk * 8);
    const int xg = (int)(t % C1_XG);
    const int yo = (int)((t / C1_XG) % C1_PO);
    const int z = (int)((t / (C1_XG * C1_PO)) % C1_D);
    const long long n = t / C1_TILES_PER_WINDOW;
    const long long o = (((n * (C1_D + 2) + z + 1) * C1_OUT_P + yo + 1) * (long long)C1_OUT_P + xg * 8 + px + 1) * 64 + chunk * 8;
    *(u32x4*)(p.out + o) = row;
    if (p.argmax) {
      const uint2 codes = *(const uint2*)(s_arg[wave] + px * 72 + chunk * 8);
      const long long oa = (((n * C1_D + z) * C1_PO + yo) * (long long)C1_PO + xg * 8 + px) * 64 + chunk * 8;
      *(uint2*)(p.argmax + oa) = codes;
    }
    __builtin_amdgcn_wave_barrier();
  };

  long long t = (long long)blockIdx.x * 4 + wave;
  if (t >= total) return;
  f32x4 a0[2][5], a1[2][5];
  load_tile(t, a0);
  while (true) {
    long long tn = t + stride;
    if (tn < total) load_tile(tn, a1);
    process(t, a0);
    if (tn >= total) break;
    t = tn;
    tn = t + stride;
    if (tn < total) load_tile(tn, a0);
    process(t, a1);
    if (tn >= total) break;
    t = tn;
  }
}

}  // namespace rgp
