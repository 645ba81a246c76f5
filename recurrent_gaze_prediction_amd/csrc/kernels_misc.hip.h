// Layout / packing / softmax kernels around the implicit-GEMM core (gfx950).
// All are HBM-bound streaming kernels: 16-byte vector accesses, 256-thread
// blocks, no cross-block communication.
#pragma once
#include "igemm.hip.h"

namespace rgp {

// c3d_input [F][C][49] f32 (the reference's placeholder layout, gaze_rnn.py:118-121)
// -> rows [F*49][C] of T: the transpose of gaze_grcn.py:225-227 fused with the
// operand conversion.  One block = one frame x 64 channels.
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_rows_kernel(const float* __restrict__ x, T* __restrict__ y, int C) {
  __shared__ float tile[64][50];
  const int f = blockIdx.y, c0 = blockIdx.x * 64;
  const float* src = x + ((long long)f * C + c0) * 49;
  // 64 x 49 floats = 784 contiguous 16-byte pieces (the block's slice starts at a multiple of 12 544 bytes)
  for (int q = threadIdx.x; q < 64 * 49 / 4; q += 256) {
    const f32x4 v = *(const f32x4*)(src + 4 * q);
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int i = 4 * q + k; tile[i / 49][i % 49] = v[k]; }
  }
  __syncthreads();
  for (int it = threadIdx.x; it < 49 * 8; it += 256) {
    const int p = it >> 3, cg = it & 7;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = tile[cg * 8 + i][p];
    store8<T>(y + ((long long)f * 49 + p) * C + c0 + cg * 8, v, 8);
  }
}

// Generic filter packer: dst[(row0+n)*K + tap*cin_k + k0 + c] =
//   src[tap_src[tap]*s_tap + n*s_n + c*s_c]   (0 if tap_src<0).
// grouped == 0: every c in [0, cin_k) is written (0 for c >= cin_src: channel padding).
// grouped != 0: several source filters share the packed K axis of one tap (gate-concatenated
//   dgrad filters); only c in [0, cin_src) is written, at column offset k0.
// Covers HWIO/DHWIO conv filters, the [kh,kw,out,in] transposed-conv filters per
// sub-pixel phase, 180-degree rotated filters for dgrad, and channel padding.
template <typename T>
__global__ __launch_bounds__(256) void pack_filter_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                                          const int* __restrict__ tap_src, int ntaps, int cin_k,
                                                          int cin_src, int n_rows, int row0, int K, long long s_tap,
                                                          long long s_n, long long s_c, int k0, int grouped, int row_step,
                                                          int chunk_major = 0) {
  const long long total = (long long)n_rows * ntaps * cin_k;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % cin_k);
    const int tap = (int)((i / cin_k) % ntaps);
    const int n = (int)(i / ((long long)cin_k * ntaps));
    if (grouped && c >= cin_src) continue;
    const int ts = tap_src ? tap_src[tap] : tap;
    float v = 0.f;
    if (ts >= 0 && c < cin_src) v = src[ts * s_tap + n * s_n + c * s_c];
    // chunk_major = B: K index (channel chunk, tap, channel in chunk) instead of (tap, channel)
    const int col = chunk_major ? (c / chunk_major) * (ntaps * chunk_major) + tap * chunk_major + (c % chunk_major) : tap * cin_k + k0 + c;
    dst[(long long)(row0 + n * row_step) * K + col] = Elem<T>::to(v);
  }
}

// The same packing for sources whose OUTPUT-channel index is the contiguous one (s_n == 1: HWIO / DHWIO forward filters,
// re-packed after every optimizer step): a 32 x 32 (n, c) tile of one tap goes through LDS so that both the fp32 reads
// (along n) and the packed writes (along c) are coalesced.  The kernel above reads such a source with a stride of
// cout floats per lane.  grid = (ceil(cin_k / 32), ceil(n_rows / 32), ntaps), 256 threads as 32 x 8.
template <typename T>
__global__ __launch_bounds__(256) void pack_filter_tiled_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                                                const int* __restrict__ tap_src, int ntaps, int cin_k, int cin_src,
                                                                int n_rows, int row0, int K, long long s_tap, long long s_c, int k0,
                                                                int row_step, int chunk_major) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 32, n0 = blockIdx.y * 32, tap = blockIdx.z;
  const int ts = tap_src ? tap_src[tap] : tap;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, n = n0 + tx;
    float v = 0.f;
    if (ts >= 0 && c < cin_src && n < n_rows) v = src[ts * s_tap + n + c * s_c];
    tile[ty + 8 * k][tx] = v;
  }
  __syncthreads();
  const int c = c0 + tx;
  if (c >= cin_k) return;
  const int col = chunk_major ? (c / chunk_major) * (ntaps * chunk_major) + tap * chunk_major + (c % chunk_major) : tap * cin_k + k0 + c;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int n = n0 + ty + 8 * k;
    if (n < n_rows) dst[(long long)(row0 + n * row_step) * K + col] = Elem<T>::to(tile[tx][ty + 8 * k]);
  }
}

// Several packs in ONE launch (the head re-packs 34 filters after every optimizer step: at config 4's shape the step is
// launch-bound and each of those launches cost ~4 us for a few KB of work).  A job is the argument list of one of the
// two kernels above; the table travels as a kernel argument.  Block b belongs to the job j with first[j] <= b < first[j+1].
struct PackJob {
  const float* src;
  void* dst;
  const int* tap_src;
  long long s_tap, s_n, s_c;
  int ntaps, cin_k, cin_src, n_rows, row0, K, k0, grouped, row_step, chunk_major;
  int tiled, gx, gy;                 // tiled: grid of the job = gx x gy x ntaps blocks; else `gx` grid-stride blocks
  // tiled == 2 / 3 (bf16 packs of whole 64-channel chunks, 16-byte aligned rows: the conv stack's 110 MB of filters, re-packed
  // after every optimizer step): 2 = 64 x 64 transposing tiles with float4 loads and 16-byte stores (gx x gy x ntaps
  // blocks); 3 = source contiguous along c: 8 channels per thread, two float4 loads and one 16-byte store, `gx` grid-stride
  // blocks.  (The scalar forms moved 1.6 TB/s: 0.1 ms per pack of the conv stack.)
};
constexpr int PACK_MAX_JOBS = 32;
struct PackJobTable {
  PackJob job[PACK_MAX_JOBS];
  int first[PACK_MAX_JOBS + 1];
  int n;
};

// ---- several small clears in one launch.  A training step clears six buffers (h_0, two sets of phase counters, the atomically
// accumulated gradients, a zero row, the folded head's dK scratch): as hipMemsetAsync calls each is a 5 us launch of its own
// on a stream whose every launch is on the critical path (config 4: 45 launches per 1.3 ms step).
constexpr int ZERO_MAX_REGIONS = 12;
struct ZeroTable {
  void* ptr[ZERO_MAX_REGIONS];
  unsigned long long bytes[ZERO_MAX_REGIONS];      // multiples of 4
  int first[ZERO_MAX_REGIONS + 1];                 // first block of each region
  int n;
};
constexpr int ZERO_BLOCK_BYTES = 256 * 16 * 4;     // 256 threads x 16 B x 4 stores

static __global__ __launch_bounds__(256) void zero_regions_kernel(const ZeroTable t) {
  int j = 0;
  while (j + 1 < t.n && (int)blockIdx.x >= t.first[j + 1]) ++j;
  const unsigned long long off = (unsigned long long)(blockIdx.x - t.first[j]) * ZERO_BLOCK_BYTES;
  char* p = (char*)t.ptr[j] + off;
  const unsigned long long n = t.bytes[j] - off < ZERO_BLOCK_BYTES ? t.bytes[j] - off : ZERO_BLOCK_BYTES;
  if ((((unsigned long long)p) & 15) == 0) {
    for (unsigned long long i = threadIdx.x * 16ull; i + 16 <= n; i += 256 * 16) *(uint4*)(p + i) = make_uint4(0u, 0u, 0u, 0u);
    for (unsigned long long i = (n & ~15ull) + threadIdx.x * 4ull; i < n; i += 256 * 4) *(unsigned*)(p + i) = 0u;
  } else {
    for (unsigned long long i = threadIdx.x * 4ull; i < n; i += 256 * 4) *(unsigned*)(p + i) = 0u;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_filter_batch_kernel(const PackJobTable t) {
  __shared__ float tile[32][33];
  int j = 0;
  while (j + 1 < t.n && (int)blockIdx.x >= t.first[j + 1]) ++j;
  const PackJob& q = t.job[j];
  const int b = blockIdx.x - t.first[j];
  T* dst = (T*)q.dst;
  if constexpr (sizeof(T) == 2) {
    if (q.tiled == 2) {
      // 64 (c) x 64 (n) tile: rows of 64 floats along n in, rows of 64 bf16 along c out
      __shared__ float big[64][65];
      const int bx = b % q.gx, by = (b / q.gx) % q.gy, tap = b / (q.gx * q.gy);
      const int c0 = bx * 64, n0 = by * 64;
      const int ts = q.tap_src ? q.tap_src[tap] : tap;
      const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;                // 16 x float4 per row, 16 rows per pass
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = c0 + ly + 16 * k;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ts >= 0) v = *(const float4*)(q.src + ts * q.s_tap + (long long)c * q.s_c + n0 + 4 * lx);
        big[ly + 16 * k][4 * lx + 0] = v.x; big[ly + 16 * k][4 * lx + 1] = v.y;
        big[ly + 16 * k][4 * lx + 2] = v.z; big[ly + 16 * k][4 * lx + 3] = v.w;
      }
      __syncthreads();
      const int sx = threadIdx.x & 7, sy = threadIdx.x >> 3;                 // 8 x 16 bytes per packed row, 32 rows per pass
      const int c = c0 + 8 * sx;
      const int col = q.chunk_major ? (c / q.chunk_major) * (q.ntaps * q.chunk_major) + tap * q.chunk_major + (c % q.chunk_major)
                                    : tap * q.cin_k + q.k0 + c;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int n = sy + 32 * k;
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          o[e] = (unsigned)f2bf(big[8 * sx + 2 * e][n]) | ((unsigned)f2bf(big[8 * sx + 2 * e + 1][n]) << 16);
        *(u32x4*)((bf16_t*)dst + (long long)(q.row0 + n0 + n) * q.K + col) = o;
      }
      return;
    }
    if (q.tiled == 3) {
      const long long groups = (long long)q.n_rows * q.ntaps * (q.cin_k / 8);
      for (long long i = (long long)b * 256 + threadIdx.x; i < groups; i += (long long)q.gx * 256) {
        const int cgn = q.cin_k / 8;
        const int c = (int)(i % cgn) * 8;
        const int tap = (int)((i / cgn) % q.ntaps);
        const int n = (int)(i / ((long long)cgn * q.ntaps));
        const int ts = q.tap_src ? q.tap_src[tap] : tap;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), bb = a;
        if (ts >= 0) {
          const float* sp = q.src + ts * q.s_tap + n * q.s_n + c;
          a = *(const float4*)sp;
          bb = *(const float4*)(sp + 4);
        }
        const int col = q.chunk_major ? (c / q.chunk_major) * (q.ntaps * q.chunk_major) + tap * q.chunk_major + (c % q.chunk_major)
                                      : tap * q.cin_k + q.k0 + c;
        u32x4 o;
        o[0] = (unsigned)f2bf(a.x) | ((unsigned)f2bf(a.y) << 16);
        o[1] = (unsigned)f2bf(a.z) | ((unsigned)f2bf(a.w) << 16);
        o[2] = (unsigned)f2bf(bb.x) | ((unsigned)f2bf(bb.y) << 16);
        o[3] = (unsigned)f2bf(bb.z) | ((unsigned)f2bf(bb.w) << 16);
        *(u32x4*)((bf16_t*)dst + (long long)(q.row0 + n * q.row_step) * q.K + col) = o;
      }
      return;
    }
  }
  if (q.tiled) {
    const int bx = b % q.gx, by = (b / q.gx) % q.gy, tap = b / (q.gx * q.gy);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = bx * 32, n0 = by * 32;
    const int ts = q.tap_src ? q.tap_src[tap] : tap;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + ty + 8 * k, n = n0 + tx;
      float v = 0.f;
      if (ts >= 0 && c < q.cin_src && n < q.n_rows) v = q.src[ts * q.s_tap + n + c * q.s_c];
      tile[ty + 8 * k][tx] = v;
    }
    __syncthreads();
    const int c = c0 + tx;
    if (c >= q.cin_k) return;
    const int col = q.chunk_major ? (c / q.chunk_major) * (q.ntaps * q.chunk_major) + tap * q.chunk_major + (c % q.chunk_major)
                                  : tap * q.cin_k + q.k0 + c;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int n = n0 + ty + 8 * k;
      if (n < q.n_rows) dst[(long long)(q.row0 + n * q.row_step) * q.K + col] = Elem<T>::to(tile[tx][ty + 8 * k]);
    }
    return;
  }
  const long long total = (long long)q.n_rows * q.ntaps * q.cin_k;
  for (long long i = (long long)b * 256 + threadIdx.x; i < total; i += (long long)q.gx * 256) {
    const int c = (int)(i % q.cin_k);
    const int tap = (int)((i / q.cin_k) % q.ntaps);
    const int n = (int)(i / ((long long)q.cin_k * q.ntaps));
    if (q.grouped && c >= q.cin_src) continue;
    const int ts = q.tap_src ? q.tap_src[tap] : tap;
    float v = 0.f;
    if (ts >= 0 && c < q.cin_src) v = q.src[ts * q.s_tap + n * q.s_n + c * q.s_c];
    const int col = q.chunk_major ? (c / q.chunk_major) * (q.ntaps * q.chunk_major) + tap * q.chunk_major + (c % q.chunk_major)
                                  : tap * q.cin_k + q.k0 + c;
    dst[(long long)(q.row0 + n * q.row_step) * q.K + col] = Elem<T>::to(v);
  }
}

// Exact algebraic fold of the 7x7 transposed conv (12 channels) with the 12->1
// projection (gaze_grcn.py:353-361): G[tap][c] = sum_o F[tap][o][c] * out_W[o].
static __global__ void fold_head_filter_kernel(const float* __restrict__ f, const float* __restrict__ out_w,
                                        float* __restrict__ g, int ntaps, int co, int ci) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ntaps * ci) return;
  const int tap = i / ci, c = i % ci;
  float s = 0.f;
  for (int o = 0; o < co; ++o) s += f[((long long)tap * co + o) * ci + c] * out_w[o];
  g[i] = s;
}

// Row-Toeplitz form of the folded head filter g[tap = a*7+b][c] (rgp_grcn.hip run_d3):
//   gt[u][n][x'*32 + c] = g[(6-u)*7 + 6-(x'-n)][c] for 0 <= x'-n <= 6, else 0;   bias16[n] = out_b
static __global__ void toeplitz_head_filter_kernel(const float* __restrict__ g, const float* __restrict__ out_b,
                                                   float* __restrict__ gt, float* __restrict__ bias16) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 16) bias16[i] = out_b[0];
  if (i >= 7 * 16 * 704) return;
  const int c = i % 32, xp = (i / 32) % 22, n = (i / 704) % 16, u = i / (16 * 704);
  const int v = xp - n;
  gt[i] = (v >= 0 && v <= 6) ? g[((6 - u) * 7 + (6 - v)) * 32 + c] : 0.f;
}

__device__ __forceinline__ float block_reduce(float v, float* sh, bool is_max) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float w = __shfl_xor(v, o);
    v = is_max ? fmaxf(v, w) : v + w;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = is_max ? fmaxf(r, sh[i]) : r + sh[i];
  return r;
}

// Per-frame softmax over the n (=2401) pixels (model_util.py:61-64) and, when
// labels are given, the per-frame cross entropy -sum g*log_softmax(z)
// (model_util.py:66-72).  One block per frame; the row lives in registers.
static __global__ __launch_bounds__(256) void softmax_xent_kernel(const float* __restrict__ z, const float* __restrict__ labels,
                                                           float* __restrict__ probs, float* __restrict__ frame_loss,
                                                           int n) {
  __shared__ float sh[4];
  const long long row = blockIdx.x;
  const float* zr = z + row * n;
  float v[12];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const int j = threadIdx.x + i * 256;
    v[i] = j < n ? zr[j] : -INFINITY;
    mx = fmaxf(mx, v[i]);
  }
  mx = block_reduce(mx, sh, true);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const int j = threadIdx.x + i * 256;
    v[i] = j < n ? expf(v[i] - mx) : 0.f;
    sum += v[i];
  }
  sum = block_reduce(sum, sh, false);
  const float inv = 1.f / sum;
  if (probs) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int j = threadIdx.x + i * 256;
      if (j < n) probs[row * n + j] = v[i] * inv;
    }
  }
  if (labels && frame_loss) {
    const float lse = mx + logf(sum);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int j = threadIdx.x + i * 256;
      if (j < n) acc += labels[row * n + j] * (lse - zr[j]);
    }
    acc = block_reduce(acc, sh, false);
    if (threadIdx.x == 0) frame_loss[row] = acc;
  }
}

// Deterministic sum of the per-frame losses divided by B*T (gaze_rnn.py:406-407).
static __global__ __launch_bounds__(256) void loss_reduce_kernel(const float* __restrict__ frame_loss, float* __restrict__ loss,
                                                          int n, float scale) {
  __shared__ float sh[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) a += frame_loss[i];
  a = block_reduce(a, sh, false);
  if (threadIdx.x == 0) *loss = a * scale;
}

// video [N][D][H][W][3] f32 (mean-subtracted) -> halo-padded channels-last
// [N][D+2][H+2][W+4][4] of T (x halo: 1 left, 3 right; 4th channel 0) so conv1a's
// (kx, c) taps are one contiguous 32-byte run per (kz, ky) and need no bounds test.
template <typename T>
__global__ __launch_bounds__(256) void video_prep_kernel(const float* __restrict__ v, T* __restrict__ out, long long npix,
                                                         int D, int H, int W) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % W);
    const int y = (int)((i / W) % H);
    const int z = (int)((i / ((long long)W * H)) % D);
    const long long n = i / ((long long)W * H * D);
    const float* s = v + i * 3;
    const long long o = (((n * (D + 2) + z + 1) * (H + 2) + y + 1) * (long long)(W + 4) + x + 1) * 4;
    out[o + 0] = Elem<T>::to(s[0]);
    out[o + 1] = Elem<T>::to(s[1]);
    out[o + 2] = Elem<T>::to(s[2]);
    out[o + 3] = Elem<T>::to(0.f);
  }
}

// conv5b rows [M][d*512+c] -> xt [M][c*2+d] (the layout nchw_to_rows_kernel produces from c3d_input)
template <typename T>
__global__ __launch_bounds__(256) void rows_to_xt_kernel(const T* __restrict__ rows, T* __restrict__ xt, long long total) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ch = (int)(i & 1023);
    xt[i] = rows[(i - ch) + (ch & 1) * 512 + (ch >> 1)];
  }
}

// The VIDEO_DATA layer of the C3D prototxt (feature_extration.prototxt:3-21) on device: window w is the
// 16 consecutive uint8 frames from starts[w]; each is resized to 128x171 (bilinear, half-pixel centres,
// edge-clamped, rounded back to an 8-bit level as cv::resize stores it), centre-cropped to 112x112
// (offsets 8, 29), and the mean cube [3][16][128][171] is subtracted at the cropped position.
// Writes the halo-padded operand image of conv1a (layout of video_prep_kernel) and/or a dense fp32
// [N][16][112][112][3] copy.  One thread per output pixel; 4 source taps x 3 channels each.
template <typename T>
__global__ __launch_bounds__(256) void frames_prep_kernel(const unsigned char* __restrict__ frames, int fh, int fw,
                                                          const int* __restrict__ starts, const float* __restrict__ mean,
                                                          T* __restrict__ out, float* __restrict__ video, long long npix) {
  constexpr int D = 16, H = 112, W = 112, RH = 128, RW = 171, OY = (RH - H) / 2, OX = (RW - W) / 2;
  const float sy = (float)fh / RH, sx = (float)fw / RW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % W);
    const int y = (int)((i / W) % H);
    const int z = (int)((i / ((long long)W * H)) % D);
    const long long n = i / ((long long)W * H * D);
    const int ry = y + OY, rx = x + OX;
    float fy = (ry + 0.5f) * sy - 0.5f, fx = (rx + 0.5f) * sx - 0.5f;
    fy = fminf(fmaxf(fy, 0.f), (float)(fh - 1));
    fx = fminf(fmaxf(fx, 0.f), (float)(fw - 1));
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = min(y0 + 1, fh - 1), x1 = min(x0 + 1, fw - 1);
    const float wy = fy - (float)y0, wx = fx - (float)x0;
    const unsigned char* f = frames + (long long)(starts[n] + z) * fh * fw * 3;
    const unsigned char *p00 = f + ((long long)y0 * fw + x0) * 3, *p01 = f + ((long long)y0 * fw + x1) * 3;
    const unsigned char *p10 = f + ((long long)y1 * fw + x0) * 3, *p11 = f + ((long long)y1 * fw + x1) * 3;
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float top = (float)p00[c] + wx * ((float)p01[c] - (float)p00[c]);
      const float bot = (float)p10[c] + wx * ((float)p11[c] - (float)p10[c]);
      float r = floorf(top + wy * (bot - top) + 0.5f);
      if (mean) r -= mean[((long long)(c * D + z) * RH + ry) * RW + rx];
      v[c] = r;
    }
    if (out) {
      const long long o = (((n * (D + 2) + z + 1) * (H + 2) + y + 1) * (long long)(W + 4) + x + 1) * 4;
      out[o + 0] = Elem<T>::to(v[0]);
      out[o + 1] = Elem<T>::to(v[1]);
      out[o + 2] = Elem<T>::to(v[2]);
      out[o + 3] = Elem<T>::to(0.f);
    }
    if (video) {
      video[i * 3 + 0] = v[0];
      video[i * 3 + 1] = v[1];
      video[i * 3 + 2] = v[2];
    }
  }
}

// frames [N][H][W][3] f32 -> [N][H][W][4] of T (4th channel 0): 5 kx taps x 4 channels of the
// ShallowNet's 5x5 conv1 are then one contiguous run per ky (saliency_shallownet.py:90-97).
template <typename T>
__global__ __launch_bounds__(256) void frame_prep_kernel(const float* __restrict__ v, T* __restrict__ out, long long npix) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
    const float* s = v + i * 3;
    T* o = out + i * 4;
    o[0] = Elem<T>::to(s[0]);
    o[1] = Elem<T>::to(s[1]);
    o[2] = Elem<T>::to(s[2]);
    o[3] = Elem<T>::to(0.f);
  }
}

// tf.nn.max_pool(ksize k, stride s, padding SAME) on NHWC (saliency_shallownet.py:117,134):
// out = ceil(in/s), pad_before = max((out-1)*s + k - in, 0) / 2, padded cells ignored.
// dst row stride ld_out elements per image (lets the flattened result be K-padded for the FC GEMM).
// A thread owns 8 channels of one output (16-byte loads; C % 8 == 0).  amax (training plans, may be null): the place of
// the FIRST maximum in the window's (row, column) scan order, a * k + b, one byte per output element -- what the gradient
// routes through (rgp_shallownet.hip maxpool_same_bwd_kernel).
__device__ __forceinline__ void mp_load8(const float* p, float* v) {
  const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
__device__ __forceinline__ void mp_load8(const bf16_t* p, float* v) {
  const u32x4 a = *(const u32x4*)p;
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = bf2f((bf16_t)(a[i] & 0xffffu)); v[2 * i + 1] = bf2f((bf16_t)(a[i] >> 16)); }
}
template <typename T>
__global__ __launch_bounds__(256) void maxpool_same_kernel(const T* __restrict__ src, T* __restrict__ dst, int N, int H,
                                                           int W, int C, int k, int s, int OH, int OW, int pt, int pl,
                                                           long long ld_out, unsigned char* __restrict__ amax) {
  const int CG = C / 8;
  const long long total = (long long)N * OH * OW * CG;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % CG) * 8;
    const int ox = (int)((i / CG) % OW);
    const int oy = (int)((i / ((long long)CG * OW)) % OH);
    const long long n = i / ((long long)CG * OW * OH);
    float m[8];
    unsigned code[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { m[q] = -INFINITY; code[q] = 0; }
    for (int a = 0; a < k; ++a) {
      const int y = oy * s - pt + a;
      if (y < 0 || y >= H) continue;
      for (int b = 0; b < k; ++b) {
        const int x = ox * s - pl + b;
        if (x < 0 || x >= W) continue;
        float v[8];
        mp_load8(src + ((n * H + y) * W + x) * C + c, v);
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (v[q] > m[q]) { m[q] = v[q]; code[q] = (unsigned)(a * k + b); }
      }
    }
    store8<T>(dst + n * ld_out + ((long long)oy * OW + ox) * C + c, m, 8);
    if (amax) {
      uint2 pk;
      pk.x = code[0] | (code[1] << 8) | (code[2] << 16) | (code[3] << 24);
      pk.y = code[4] | (code[5] << 8) | (code[6] << 16) | (code[7] << 24);
      *(uint2*)(amax + ((n * OH + oy) * (long long)OW + ox) * C + c) = pk;
    }
  }
}

// 49x49 -> 7x7 average pool (tf.nn.avg_pool 7x7 stride 7 VALID, gaze_rnn.py:262-269)
static __global__ __launch_bounds__(64) void avgpool7_kernel(const float* __restrict__ src, float* __restrict__ dst) {
  const long long f = blockIdx.x;
  const int t = threadIdx.x;
  if (t >= 49) return;
  const int oy = t / 7, ox = t % 7;
  float a = 0.f;
  for (int y = 0; y < 7; ++y)
    for (int x = 0; x < 7; ++x) a += src[f * 2401 + (oy * 7 + y) * 49 + ox * 7 + x];
  dst[f * 49 + t] = a * (1.0f / 49.0f);
}

// conv5b rows [F*49][d*512+c] (T) -> the reference's feature layout
// [F][1024 = c*2+d][7][7] f32 (gaze_rnn.py:494-497).
template <typename T>
__global__ __launch_bounds__(256) void rows_to_c3d_features_kernel(const T* __restrict__ rows, float* __restrict__ feat,
                                                                   long long total) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int p = (int)(i % 49);
    const int ch = (int)((i / 49) % 1024);
    const long long f = i / (49 * 1024);
    const int c = ch >> 1, d = ch & 1;
    feat[i] = Elem<T>::from(rows[(f * 49 + p) * 1024 + d * 512 + c]);
  }
}

}  // namespace rgp
