// The last stage of the saliency head for gfx950, bf16: logits[f][y][x] = sum_{u,v,c} D2p[f][y+u][x+v][c] G[u][v][c] + out_b,
// G = deconv3 (7x7, SAME, 32 -> 12) folded with the 12 -> 1 output filter (/root/reference/models/gaze_grcn.py:346-374;
// the fold is exact algebra, rgp_grcn.hip).  One output channel: as an implicit GEMM over (tap, channel) it would fill 1
// of 16 MFMA columns, so a GEMM row is (y, run of 16 x) and the filter the row-Toeplitz matrix
// Gt[u][(x', c)][n] = G[u][x'-n][c] (kernels_misc.hip.h), K = 7 tap rows x 22 pixels x 32 channels = 4928, N = 16.
//
// Rounds 1-2 ran that GEMM on the general implicit-GEMM tile: every GEMM row gathered its own 7 runs of 1408 bytes by
// LDS-DMA -- 1.5 GB of ingest per 1024 frames for a 198 MB input (rows y and y+1 share 6 of their 7 runs) -- and took
// 0.20 ms (+ 0.02 ms for the 49th pixel column on the one-column path), 40 % of the head's transposed-conv stage.  Here the
// input band of a block is staged in LDS ONCE:
//
//  * block (8 waves) = (frame, band of 16 output rows y0 .. y0+15, y0 = 0, 16, 32, 48): 22 rows x 55 pixels x 64 B of D2p
//    (the last band: output row 48 only, 7 input rows; its other fragment rows read unstaged LDS and are not stored),
//    LDS row pitch 3536 B = 13 x 16 (mod 256): the 16 rows of an A fragment (16 different y, same pixel) fall on 16 different
//    16-byte bank groups -- ds_read_b128 without conflicts.
//  * the block's 4 A fragments are the pixel runs x0 = 0, 16, 32, 33 (the last one overlaps too -- a run starting at 48 would
//    read past the row -- and stores only pixel column 48).  The Toeplitz matrix does not depend on x0.
//  * K is split over the 8 waves (k-step ks -> wave ks & 7): a wave keeps its 20 (19) filter fragments in registers (80
//    VGPRs, loaded once per block from the L2-resident packed filter while the band is staged), so the filter is read
//    once per block, not once per wave; the partial sums of the 8 waves are added through LDS.
#pragma once
#include "igemm.hip.h"

namespace rgp {

struct HeadLogitsParams {
  const bf16_t* d2;      // [F][55][55][32] halo-padded (halo 3) output of deconv2
  const bf16_t* wt;      // packed Toeplitz filter [>= 16 rows][4928]: row n, K index u * 704 + x' * 32 + c
  const float* bias16;   // out_b, 16 times
  float* logits;         // [F][49][49]
  int n_frames;
};

constexpr int HL_PITCH = 3536, HL_ROWS = 22, HL_K = 4928, HL_KSTEPS = HL_K / 32, HL_ROWB = 55 * 64;
constexpr int HL_SMEM = HL_ROWS * HL_PITCH;                  // 77 792 B: two blocks per CU
constexpr int HL_WAVES = 8, HL_THREADS = 64 * HL_WAVES;     // K split 8 ways: 16 waves per CU hide the LDS / L2 latencies of the short chains
static_assert(HL_PITCH % 256 == 208 && HL_PITCH >= HL_ROWB, "row pitch: an odd multiple of 16 (mod 256)");
static_assert(HL_WAVES * 4 * 64 * 16 <= HL_SMEM && (HL_ROWS * 4) % HL_WAVES == 0, "the reduction buffer reuses the band");

static __global__ __launch_bounds__(HL_THREADS, 2) void head_logits_bf16_kernel(const HeadLogitsParams p) {
  extern __shared__ __attribute__((aligned(16))) char hl_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int f = blockIdx.x >> 2, band = blockIdx.x & 3;
  const int y0 = band * 16;
  const int frow = lane & 15, fk = lane >> 4;

  // this wave's filter fragments: k-steps wave, wave + 4, ...
  constexpr int NKW = (HL_KSTEPS + HL_WAVES - 1) / HL_WAVES;   // 20
  f32x4 bfrag[NKW];
  const bf16_t* wrow = p.wt + (long long)frow * HL_K + fk * 8;
#pragma unroll
  for (int i = 0; i < NKW; ++i) {
    const int ks = wave + HL_WAVES * i;
    bfrag[i] = ks < HL_KSTEPS ? *(const f32x4*)(wrow + ks * 32) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // the band: 22 rows of 3520 bytes (contiguous in memory) to LDS rows of pitch 3536
  const char* src = (const char*)(p.d2 + ((long long)f * 55 + y0) * 55 * 32);
  // by LDS-DMA, 16 bytes per lane: 4 instructions per row (3 x 64 lanes + 28), 11 per wave, all in flight together (a copy
  // loop through registers ran one global-load latency per iteration: 19 of them, 10 us per block)
  constexpr int CPR = HL_ROWB / 16;                           // 220 chunks of 16 bytes per row
#pragma unroll
  for (int t = 0; t < HL_ROWS * 4 / HL_WAVES; ++t) {
    const int jj = wave * (HL_ROWS * 4 / HL_WAVES) + t, r = jj >> 2, q = jj & 3;
    const int chunk = q * 64 + lane;
    if (chunk < CPR && y0 + r < 55)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long long)r * HL_ROWB + chunk * 16),
                                       (__attribute__((address_space(3))) void*)(hl_smem + r * HL_PITCH + q * 1024), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x4 acc[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // A fragment of (pixel run b, k-step ks = (u, j)): row frow = output row y0 + frow, elements (x0 + j, channels 8 fk ..)
  const char* abase = hl_smem + frow * HL_PITCH + fk * 16;
#pragma unroll
  for (int i = 0; i < NKW; ++i) {
    const int ks = wave + HL_WAVES * i;
    if (ks < HL_KSTEPS) {
      const int u = ks / 22, j = ks - u * 22;
      const char* a = abase + u * HL_PITCH + j * 64;
      const f32x4 a0 = *(const f32x4*)(a);
      const f32x4 a1 = *(const f32x4*)(a + 16 * 64);
      const f32x4 a2 = *(const f32x4*)(a + 32 * 64);
      const f32x4 a3 = *(const f32x4*)(a + 33 * 64);
      Mma<bf16_t>::step(acc[0], a0, bfrag[i]);
      Mma<bf16_t>::step(acc[1], a1, bfrag[i]);
      Mma<bf16_t>::step(acc[2], a2, bfrag[i]);
      Mma<bf16_t>::step(acc[3], a3, bfrag[i]);
    }
  }
  __syncthreads();                                            // everybody is done with the band: it becomes the reduction buffer
  f32x4* red = (f32x4*)hl_smem;                               // [wave][run][lane]
#pragma unroll
  for (int b = 0; b < 4; ++b) red[(wave * 4 + b) * 64 + lane] = acc[b];
  __syncthreads();
  if (tid < 256) {
    const int b = tid >> 6;                                   // this thread finishes run b, accumulator lane `lane`
    f32x4 v = red[b * 64 + lane];
#pragma unroll
    for (int w = 1; w < HL_WAVES; ++w) {
      const f32x4 t = red[(w * 4 + b) * 64 + lane];
      v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
    }
    const float bias = p.bias16[frow];
    const int x = (b == 3 ? 33 : 16 * b) + frow;              // MFMA column = pixel of the run
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int yl = 4 * fk + r;                              // MFMA row 4 (lane >> 4) + r
      // every pixel is written by ONE run: the overlapping run 3 keeps only column 48 (the same pixel has its taps on
      // other k-steps, i.e. in other waves' partial sums, in run 2: equal up to rounding, not bit-equal)
      if ((b != 3 || frow == 15) && y0 + yl < 49) p.logits[((long long)f * 49 + y0 + yl) * 49 + x] = v[r] + bias;
    }
  }
}

inline int run_head_logits(const bf16_t* d2, const bf16_t* wt, const float* bias16, float* logits, int n_frames, hipStream_t s) {
  HeadLogitsParams p{d2, wt, bias16, logits, n_frames};
  RGP_TRY(ensure_dyn_smem((const void*)head_logits_bf16_kernel, HL_SMEM));
  head_logits_bf16_kernel<<<n_frames * 4, HL_THREADS, HL_SMEM, s>>>(p);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

}  // namespace rgp
