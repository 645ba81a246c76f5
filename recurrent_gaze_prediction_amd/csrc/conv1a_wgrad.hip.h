// Filter and bias gradient of C3D conv1a (+ReLU +pool1) for gfx950, bf16: the backward of conv1a.hip.h
// (tf.gradients through feature_extration.prototxt:22-66; base.py:278-281).
//
//   dW[tap, c, n] = sum over conv rows  X[row + tap, c] * dY[row, n],     db[n] = sum over rows dY[row, n]
//
// where dY, the gradient at conv1a's pre-activation, is the pooled gradient dP routed to the arg-max member of
// each 1x2x2 window and gated by the pooled output's sign.  Going through the generic path this layer cost an
// un-pool pass that WRITES the 6.4 MB-per-window dY image (3.5 ms per 256 windows) and a wgrad that reads it back
// (3.7 ms): both only move zeros.  This kernel never materialises dY:
//
//  * jobs = (window, plane z, 2 pooled rows), dealt to the XCDs in contiguous ranges like the forward kernel;
//    per job the 3 x 6 x 116-pixel input patch (16.3 KiB) and the 112 windows x 64 channels of dP and arg-max codes
//    are fetched once with 16-byte loads into registers while the previous job computes, and parked in LDS
//    (double-buffered, one barrier per job).  dP is gated on the way in by bit 2 of the code, which the forward sets
//    where the pooled output it stored is zero (conv1a.hip.h): round 3 read the 6.4 MB-per-window activation back for
//    that (y > 0), 36 % of this kernel's bytes -- it is bound by its HBM reads (matrix pipe 27 % busy);
//  * MFMA v_mfma_f32_16x16x32_bf16 with M = packed filter index (tap*4 + c: 27 taps x 4 = 108 -> 128),
//    N = output channel, reduction = 32 conv rows = 8 pooling windows x (dy, dx):
//      - A (X^T) fragments come straight out of the patch with the transposing LDS read ds_read_b64_tr_b16: a
//        16-lane group addresses 4 conv rows x 4 taps (one 8-byte pixel each) and every lane receives one
//        (tap, c) column = 4 consecutive reduction rows; two reads = the 8-deep k-group;
//      - B (dY) fragments are built in registers: a lane reads the gated gradient and the arg-max code of its two
//        windows and shifts the bf16 into the member's slot (one 64-bit shift per window) -- the expanded dY tile
//        never exists anywhere;
//  * 8 waves = 2 row-group halves x (2 x 2) quarters of the 128 x 64 output tile; accumulators live across all
//    jobs of a block and are added to the DHWIO fp32 gradient with one atomic per element and wave at the end;
//    the bias gradient is summed from the staged dP on the way into LDS.
#pragma once
#include "conv1a.hip.h"

namespace rgp {

struct Conv1aWgradParams {
  const bf16_t* in;            // act0 [n][18][114][116][4]
  const bf16_t* dyp;           // pooled gradient [n][16][56][56][64]
  const unsigned char* argmax; // [n][16][56][56][64]: dy*2+dx of the window's first maximum
  const bf16_t* y;             // act1 [n][18][58][58][64]: pooled forward output (not read: the gate y > 0 is bit 2 of the code)
  float* dw;                   // DHWIO [27][3][64] fp32, accumulated
  float* db;                   // [64] fp32, accumulated
  int n_windows;
};

constexpr int W1_JROWS = 2;                                // pooled rows per job
constexpr int W1_YH = C1_PO / W1_JROWS;                    // 28 jobs per plane
constexpr int W1_JOBS_PER_WINDOW = C1_D * W1_YH;           // 448
constexpr int W1_PROWS = 2 * W1_JROWS + 2;                 // 6 input rows per plane
constexpr int W1_PATCH = 3 * W1_PROWS * C1_ROWB;           // 16704 bytes
constexpr int W1_PCHUNKS = W1_PATCH / 16;                  // 1044
constexpr int W1_NPL = (W1_PCHUNKS + 511) / 512;           // 3 patch loads per thread
constexpr int W1_WIN = W1_JROWS * C1_PO;                   // 112 windows per job
constexpr int W1_GCHUNKS = W1_WIN * 8;                     // 896 (window, 8-channel group) items
constexpr int W1_NGL = (W1_GCHUNKS + 511) / 512;           // 2 per thread
constexpr int W1_LD = 72;                                  // padded channel stride of the staged gradient / codes
constexpr int W1_G_OFF = W1_PATCH;
constexpr int W1_A_OFF = W1_G_OFF + W1_WIN * W1_LD * 2;
constexpr int W1_BUF = W1_A_OFF + W1_WIN * W1_LD;          // 40896 bytes per buffer
constexpr int W1_SMEM = 2 * W1_BUF;
constexpr int W1_RG = W1_WIN / 8;                          // 14 row groups (8 windows = 32 conv rows) per job

typedef short s16x4_t __attribute__((ext_vector_type(4)));

static __global__ __launch_bounds__(512) void conv1a_wgrad_bf16_kernel(const Conv1aWgradParams p) {
  extern __shared__ __attribute__((aligned(16))) char w1_smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
  const int fi = lane & 15, kg = lane >> 4;

  const long long total = (long long)p.n_windows * W1_JOBS_PER_WINDOW;
  const long long per_xcd = (total + 7) / 8;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  long long job = xcd * per_xcd + slot;
  long long job_end = (xcd + 1) * per_xcd;
  if (job_end > total) job_end = total;
  if (job >= job_end) return;

  // ---- what this thread fetches per job ----
  int p_off[W1_NPL];                                       // patch chunk q = tid + 512 u (elements from the job's first row)
#pragma unroll
  for (int u = 0; u < W1_NPL; ++u) {
    int q = tid + 512 * u;
    if (q >= W1_PCHUNKS) q = W1_PCHUNKS - 1;
    const int row = q / C1_CPR, c = q - row * C1_CPR;
    const int kz = row / W1_PROWS, ry = row - kz * W1_PROWS;
    p_off[u] = ((kz * C1_HP + ry) * C1_WP) * 4 + c * 8;
  }
  struct Fetched { u32x4 px[W1_NPL]; u32x4 g[W1_NGL]; uint2 am[W1_NGL]; };
  auto fetch = [&](long long j, Fetched& f) {
    const int yh = (int)(j % W1_YH);
    const int z = (int)((j / W1_YH) % C1_D);
    const long long n = j / W1_JOBS_PER_WINDOW;
    const bf16_t* src = p.in + (((n * (C1_D + 2) + z) * C1_HP + yh * 2 * W1_JROWS) * (long long)C1_WP) * 4;
#pragma unroll
    for (int u = 0; u < W1_NPL; ++u) f.px[u] = *(const u32x4*)(src + p_off[u]);
    const long long w0 = ((n * C1_D + z) * C1_PO + yh * W1_JROWS) * (long long)C1_PO;      // first window of the job
#pragma unroll
    for (int u = 0; u < W1_NGL; ++u) {
      int c = tid + 512 * u;
      if (c >= W1_GCHUNKS) c = W1_GCHUNKS - 1;
      f.g[u] = *(const u32x4*)(p.dyp + w0 * 64 + (long long)c * 8);
      f.am[u] = *(const uint2*)(p.argmax + w0 * 64 + (long long)c * 8);
    }
  };
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto park = [&](char* buf, const Fetched& f) {
#pragma unroll
    for (int u = 0; u < W1_NPL; ++u)
      if (tid + 512 * u < W1_PCHUNKS) *(u32x4*)(buf + (tid + 512 * u) * 16) = f.px[u];
#pragma unroll
    for (int u = 0; u < W1_NGL; ++u) {
      const int c = tid + 512 * u;
      if (c < W1_GCHUNKS) {
        u32x4 g = f.g[u];
#pragma unroll
        for (int k = 0; k < 4; ++k) {                       // gate by y > 0: bit 2 of the channel's code (conv1a.hip.h)
          const unsigned cw = k < 2 ? f.am[u].x : f.am[u].y;
          const unsigned keep_lo = (cw >> (16 * (k & 1))) & 4u ? 0u : 0xffffu;
          const unsigned keep_hi = (cw >> (16 * (k & 1) + 8)) & 4u ? 0u : 0xffff0000u;
          g[k] &= keep_lo | keep_hi;
          bsum[2 * k] += __uint_as_float(g[k] << 16);
          bsum[2 * k + 1] += __uint_as_float(g[k] & 0xffff0000u);
        }
        const int wj = c >> 3, cg = c & 7;
        *(u32x4*)(buf + W1_G_OFF + (wj * W1_LD + cg * 8) * 2) = g;
        *(uint2*)(buf + W1_A_OFF + wj * W1_LD + cg * 8) = f.am[u];
      }
    }
  };

  // ---- per-lane fragment addressing ----
  // A: lane 4q+p of a 16-lane group addresses conv row q = (dy, dx) of the group's window and tap p of the m-tile
  const int q = fi >> 2, pt = fi & 3, a_dy = q >> 1, a_dx = q & 1;
  int a_const[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    int tap = 4 * (4 * wm + m) + pt;
    if (tap > 26) tap = 0;                                   // rows of dW that do not exist: any finite operand
    const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
    a_const[m] = ((kz * W1_PROWS + a_dy + ky) * C1_WP + 4 * kg + a_dx + kx) * 8;
  }
  // B: lane (n = fi, kg) owns windows 2kg, 2kg+1 of the row group
  const int b_const = (2 * kg) * W1_LD + wn * 32 + fi;

  f32x4 acc[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  Fetched f;
  fetch(job, f);
  park(w1_smem, f);
  __syncthreads();
  int cur = 0;
  while (true) {
    const long long nxt = job + nslot;
    const bool more = nxt < job_end;
    if (more) fetch(nxt, f);
    const char* buf = w1_smem + cur * W1_BUF;
#pragma unroll
    for (int i = 0; i < W1_RG / 2; ++i) {
      const int rg = 2 * i + half;
      const int yl = rg / C1_XG, xg = rg - yl * C1_XG;
      const char* pa = buf + (2 * yl * C1_WP + 16 * xg) * 8;
      f32x4 a[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(pa + a_const[m]));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(pa + a_const[m] + 16));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        a[m] = __builtin_bit_cast(f32x4, v);
      }
      const bf16_t* sg = (const bf16_t*)(buf + W1_G_OFF) + rg * 8 * W1_LD + b_const;
      const unsigned char* sa = (const unsigned char*)(buf + W1_A_OFF) + rg * 8 * W1_LD + b_const;
      f32x4 b[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned long long v0 = (unsigned long long)sg[j * 16] << (16 * (sa[j * 16] & 3));
        const unsigned long long v1 = (unsigned long long)sg[j * 16 + W1_LD] << (16 * (sa[j * 16 + W1_LD] & 3));
        const u32x4 v = {(unsigned)v0, (unsigned)(v0 >> 32), (unsigned)v1, (unsigned)(v1 >> 32)};
        b[j] = __builtin_bit_cast(f32x4, v);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j) Mma<bf16_t>::step(acc[m][j], a[m], b[j]);
    }
    if (!more) break;
    park(w1_smem + (cur ^ 1) * W1_BUF, f);
    __syncthreads();
    cur ^= 1;
    job = nxt;
  }

  // ---- filter gradient: D[row = 4*(lane>>4)+r -> (tap-in-tile kg, c = r)][col = fi -> n] ----
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int tap = 4 * (4 * wm + m) + kg;
    if (tap < 27) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = (2 * wn + j) * 16 + fi;
#pragma unroll
        for (int r = 0; r < 3; ++r) atomicAdd(p.dw + (tap * 3 + r) * 64 + n, acc[m][j][r]);
      }
    }
  }
  // ---- bias gradient: thread = (row lane tid>>3, channel group tid&7) ----
  __syncthreads();
  float* red = (float*)w1_smem;
#pragma unroll
  for (int k = 0; k < 8; ++k) red[(tid >> 3) * 64 + (tid & 7) * 8 + k] = bsum[k];
  __syncthreads();
  if (tid < 64) {
    float s = 0.f;
    for (int r = 0; r < 64; ++r) s += red[r * 64 + tid];
    if (s != 0.f) atomicAdd(p.db + tid, s);
  }
}

}  // namespace rgp
