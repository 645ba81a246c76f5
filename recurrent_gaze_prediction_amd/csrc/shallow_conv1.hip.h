// ShallowNet conv1 (5x5 VALID, 3 -> 32) + bias + ReLU + 2x2 / 2 max-pool for gfx950, bf16
// (/root/reference/models/saliency_shallownet.py:90-117; BASELINE config 1, and the frame-saliency branch of the cascade).
//
// Rounds 1-2 ran this layer on the general implicit-GEMM tile (igemm_kernel<128x32, G = 2, P = 4>): K = 5 x 32 is five
// k-steps, so tile set-up, the gather of every row's five 64-byte runs and the pooled epilogue dominate -- 0.35 ms for
// 512 frames of 112 x 112, 46 % of config 1's forward pass.  Here a workgroup owns a FRAME:
//
//  * the prepared frame [IH][IH][4] bf16 (8 bytes per pixel, frame_prep_kernel) is copied to LDS once by LDS-DMA (100 KB at
//    112 x 112, 77 KB at 98 x 98: one workgroup of 8 waves per CU);
//  * K order = the packed filter of the general path: k-step ky = 8 pixels x 4 channels of input row y + ky (weights of
//    kx >= 5 and of the 4th channel are zero), so the A fragment of 16 consecutive output pixels is, per lane, the 16
//    bytes at pixel x + 2 (lane >> 4) of row y + ky: two ds_read_b64 (8-byte aligned; neighbouring lanes overlap, which
//    the LDS serves as broadcasts).  The ten filter fragments (5 k-steps x 2 column tiles) stay in registers;
//  * a wave's unit = one POOLED row x 16 conv pixels: conv rows 2 py and 2 py + 1 share five of their six input rows, 12
//    fragment reads feed 20 MFMAs; pool, bias, ReLU and the arg-max code (first maximum in (dy, dx) order on the raw sums,
//    as pool_window_argmax of the general path) happen in registers, the 8 pooled pixels x 32 channels go through a
//    per-wave LDS patch for 16-byte stores.
#pragma once
#include "igemm.hip.h"

namespace rgp {

struct ShallowConv1Params {
  const bf16_t* frames4;   // [n][IH][IH][4] (4th channel 0)
  const bf16_t* wp;        // packed filter [>= 32 rows][ldw]: row = output channel, K index ky * 32 + kx * 4 + c (kx < 8)
  const float* bias;       // [32]
  bf16_t* pool1;           // [n][PH][PH][32]
  unsigned char* amax;     // training plans: [n][PH * PH][32] member dy * 2 + dx of the first maximum; else null
  int n;
  int ldw;                 // elements per packed filter row (the K schedule pads the five k-steps to six)
};

template <int IH> struct ShallowConv1Cfg {
  static constexpr int CH = IH - 4, PH = CH / 2, XB = (CH + 15) / 16;
  static constexpr int FRAME_BYTES = IH * IH * 8;
  static constexpr int PAD_OFF = FRAME_BYTES;                 // 64 zero bytes: the last row's reads run 7 pixels past the frame
  static constexpr int STG_OFF = (PAD_OFF + 64 + 255) / 256 * 256;
  static constexpr int STG_WAVE = 512 + 256;                  // per wave: 8 pooled pixels x 32 channels bf16, and their codes
  static constexpr int SMEM = STG_OFF + 8 * STG_WAVE;
  static_assert(CH % 2 == 0 && FRAME_BYTES % 16 == 0, "frame geometry");
  static_assert(SMEM <= 160 * 1024, "LDS budget");
};

template <int IH, bool ARGMAX>
static __global__ __launch_bounds__(512) void shallow_conv1_bf16_kernel(const ShallowConv1Params p) {
  using C = ShallowConv1Cfg<IH>;
  extern __shared__ __attribute__((aligned(16))) char sc_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fk = lane >> 4;
  const long long img = blockIdx.x;

  // the frame, 16 bytes per lane and instruction; the tail instruction is masked by lane
  {
    const char* src = (const char*)(p.frames4 + img * (long long)(IH * IH * 4));
    constexpr int NCH = C::FRAME_BYTES / 16;                  // 16-byte chunks
    for (int c0 = wave * 64; c0 < NCH; c0 += 8 * 64) {
      if (c0 + lane < NCH)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long long)(c0 + lane) * 16),
                                         (__attribute__((address_space(3))) void*)(sc_smem + c0 * 16), 16, 0, 0);
    }
    if (tid < 4) ((u32x4*)(sc_smem + C::PAD_OFF))[tid] = (u32x4){0u, 0u, 0u, 0u};
  }
  // filter fragments (column n = 16 t + frow, k chunk fk of k-step ky) and this lane's two biases
  f32x4 bfr[5][2];
#pragma unroll
  for (int ky = 0; ky < 5; ++ky)
#pragma unroll
    for (int t = 0; t < 2; ++t) bfr[ky][t] = *(const f32x4*)(p.wp + (long long)(16 * t + frow) * p.ldw + ky * 32 + fk * 8);
  const float b0 = p.bias[frow], b1 = p.bias[16 + frow];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  char* stg = sc_smem + C::STG_OFF + wave * C::STG_WAVE;
  for (int unit = wave; unit < C::PH * C::XB; unit += 8) {
    const int py = unit / C::XB, xb = unit - py * C::XB;
    // input rows 2 py .. 2 py + 5, pixels 16 xb + frow + 2 fk, + 1
    const char* a0 = sc_smem + ((2 * py) * IH + 16 * xb + frow + 2 * fk) * 8;
    f32x4 a[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const uint2 lo = *(const uint2*)(a0 + j * IH * 8), hi = *(const uint2*)(a0 + j * IH * 8 + 8);
      a[j] = __builtin_bit_cast(f32x4, (u32x4){lo.x, lo.y, hi.x, hi.y});
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[dy][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 5; ++ky)
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int t = 0; t < 2; ++t) Mma<bf16_t>::step(acc[dy][t], a[dy + ky], bfr[ky][t]);
    // accumulator register r of a lane: conv pixel 16 xb + 4 fk + r, channel 16 t + frow.  Pool over (dy, dx): pooled
    // pixel 8 xb + 2 fk + h takes registers 2 h, 2 h + 1 of both rows
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float best = acc[0][t][2 * h];
        unsigned code = 0;
        const float c1 = acc[0][t][2 * h + 1], c2 = acc[1][t][2 * h], c3 = acc[1][t][2 * h + 1];
        if (c1 > best) { best = c1; code = 1; }
        if (c2 > best) { best = c2; code = 2; }
        if (c3 > best) { best = c3; code = 3; }
        const float v = fmaxf(best + (t ? b1 : b0), 0.f);
        const int pp = 2 * fk + h, ch = 16 * t + frow;
        *(bf16_t*)(stg + pp * 64 + ch * 2) = f2bf(v);
        if (ARGMAX) *(unsigned char*)(stg + 512 + pp * 32 + ch) = (unsigned char)code;
      }
    // DS operations of one wave execute in order; the compiler must not move the 16-byte reads above the 2-byte stores
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    if (lane < 32) {
      const int pp = lane >> 2, part = lane & 3, px = 8 * xb + pp;
      if (px < C::PH)
        *(u32x4*)(p.pool1 + ((img * C::PH + py) * C::PH + px) * 32 + part * 8) = *(const u32x4*)(stg + pp * 64 + part * 16);
    } else if (ARGMAX && lane < 48) {
      const int l = lane - 32, pp = l >> 1, half = l & 1, px = 8 * xb + pp;
      if (px < C::PH)
        *(u32x4*)(p.amax + ((img * C::PH + py) * C::PH + px) * 32 + half * 16) = *(const u32x4*)(stg + 512 + pp * 32 + half * 16);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  }
}

template <int IH>
inline int run_shallow_conv1_t(const ShallowConv1Params& p, hipStream_t s) {
  using C = ShallowConv1Cfg<IH>;
  if (p.amax) {
    auto kern = shallow_conv1_bf16_kernel<IH, true>;
    RGP_TRY(ensure_dyn_smem((const void*)kern, C::SMEM));
    kern<<<p.n, 512, C::SMEM, s>>>(p);
  } else {
    auto kern = shallow_conv1_bf16_kernel<IH, false>;
    RGP_TRY(ensure_dyn_smem((const void*)kern, C::SMEM));
    kern<<<p.n, 512, C::SMEM, s>>>(p);
  }
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

// frame sizes the reference uses: 112 (FramewiseShallowNet) and 98 (the cascade's branch); false = not covered
inline bool shallow_conv1_covers(int IH) { return IH == 112 || IH == 98; }
inline int run_shallow_conv1(int IH, const ShallowConv1Params& p, hipStream_t s) {
  if (p.n <= 0) return RGP_OK;
  return IH == 112 ? run_shallow_conv1_t<112>(p, s) : run_shallow_conv1_t<98>(p, s);
}

}  // namespace rgp
