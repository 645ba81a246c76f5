// Backward-pass support kernels of the gaze_grcn head (gfx950): loss gradient, the
// element-wise parts of BPTT / batch-norm, gather-transposes that turn weight gradients
// into plain K-contiguous GEMMs for igemm_kernel, the folded 7x7 head filter's dgrad /
// wgrad, and the fused TF-Adam + global-norm-clip optimizer.
// What is differentiated: /root/reference/models/gaze_grcn.py:173-376 with the loss of
// gaze_rnn.py:363-408 (tf.gradients, base.py:278-281).
#pragma once
#include "kernels_misc.hip.h"

namespace rgp {

// d loss / d logits for loss = sum_f xent_f / n_frames (gaze_rnn.py:390-407):
//   dz = (softmax(z) * sum(g) - g) / n_frames.  One block per frame; also the frame's sum
// of dz (for d out_b).  'l2' variant: dz = (z - g) / n_frames.
static __global__ __launch_bounds__(256) void dlogits_kernel(const float* __restrict__ probs_or_logits,
                                                      const float* __restrict__ labels, float* __restrict__ dz,
                                                      float* __restrict__ frame_sum, int n, float scale, int l2) {
  __shared__ float sh[4];
  const long long row = blockIdx.x;
  float gs = 0.f;
  for (int j = threadIdx.x; j < n; j += 256) gs += labels[row * n + j];
  gs = block_reduce(gs, sh, false);
  float acc = 0.f;
  for (int j = threadIdx.x; j < n; j += 256) {
    const float p = probs_or_logits[row * n + j], g = labels[row * n + j];
    const float d = (l2 ? (p - g) : (p * gs - g)) * scale;
    dz[row * n + j] = d;
    acc += d;
  }
  acc = block_reduce(acc, sh, false);
  if (threadIdx.x == 0) frame_sum[row] = acc;
}

// out[0] (+)= scale * sum(x[0..n))   (single block, deterministic order)
static __global__ __launch_bounds__(256) void sum_kernel(const float* __restrict__ x, float* __restrict__ out, long long n,
                                                  float scale) {
  __shared__ float sh[4];
  float a = 0.f;
  for (long long i = threadIdx.x; i < n; i += 256) a += x[i];
  a = block_reduce(a, sh, false);
  if (threadIdx.x == 0) *out = a * scale;
}

// Row sums of a [R][ld] matrix of T over the first n columns: out[r] = sum_m x[r][m]
// (bias gradient from a transposed activation-gradient matrix).  One block per row.
template <typename T>
__global__ __launch_bounds__(256) void rowsum_kernel(const T* __restrict__ x, float* __restrict__ out, long long ld,
                                                     long long n) {
  __shared__ float sh[4];
  const T* r = x + (long long)blockIdx.x * ld;
  float a = 0.f;
  for (long long i = threadIdx.x; i < n; i += 256) a += Elem<T>::from(r[i]);
  a = block_reduce(a, sh, false);
  if (threadIdx.x == 0) out[blockIdx.x] = a;
}

// Gather + transpose: dst[(row0 + tap*C + c) * ld + m] = src[base(img) + tab[tap*Mw + ml] + c]
// (0 where tab < 0), m = img*Mw + ml, base(img) = (img % inner)*stride_inner + (img / inner)*stride_outer.
// Turns "sum over output positions m" (weight gradients) into a K-contiguous operand.
// Block = 64 positions x 64 channels of one tap, transposed through LDS.
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void gather_transpose_kernel(const TS* __restrict__ src, TD* __restrict__ dst,
                                                               const int* __restrict__ tab, int Mw, long long M, int C,
                                                               long long ld, int row0, int inner,
                                                               long long stride_inner, long long stride_outer) {
  __shared__ float tile[64][65];
  const int tap = blockIdx.z;
  const long long m0 = (long long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  for (int it = threadIdx.x; it < 64 * 8; it += 256) {
    const int r = it >> 3, cg = it & 7;
    const long long m = m0 + r;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (m < M) {
      const long long img = m / Mw;
      const int ml = (int)(m - img * Mw);
      const int off = tab[tap * Mw + ml];
      if (off >= 0) {
        const TS* s = src + (img % inner) * stride_inner + (img / inner) * stride_outer + off + c0 + cg * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (c0 + cg * 8 + i < C) v[i] = Elem<TS>::from(s[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) tile[r][cg * 8 + i] = v[i];
  }
  __syncthreads();
  for (int it = threadIdx.x; it < 64 * 8; it += 256) {
    const int c = it >> 3, mg = it & 7;
    if (c0 + c >= C) continue;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = tile[mg * 8 + i][c];
    TD* d = dst + ((long long)row0 + (long long)tap * C + c0 + c) * ld + m0 + mg * 8;
    const long long left = M - (m0 + mg * 8);
    if (left > 0) store8<TD>(d, v, left >= 8 ? 8 : (int)left);
  }
}

// Folded 7x7 head filter, gradient w.r.t. its input (deconv2's output):
//   dd2[f, Y, X, c] = sum_{u,v} dz[f, Y-u, X-v] * Gp[u, v, c]   (Y,X in 0..48 un-padded coords
//   of the 49x49 map shifted by the SAME padding 3: source pixel (Y+3-u, X+3-v) - 3 ...)
// Precisely: logit[y,x] = sum_{u,v,c} d2pad[y+u, x+v, c] Gp[u,v,c], d2pad = d2 with halo 3, so
//   dd2[Y,X,c] = sum_{u,v} dz[Y+3-u, X+3-v] Gp[u,v,c]  with dz = 0 outside [0,49)^2.
// One block per (frame, 7 rows Y0 .. Y0+6); a thread owns 7 consecutive pixels X of one row and 8 channels, so a filter
// tap's 8 channels (two 16-byte LDS reads) serve 7 pixels and a row's 13 dz values serve its 7 taps: 18 LDS reads per 392
// FMAs.  (Round 2's version -- a thread per (pixel, 8 channels), three LDS reads per 8 FMAs -- ran at the LDS bandwidth:
// 61 us for 280 frames, 0.22 ms for 1024.)
template <typename T>
__global__ __launch_bounds__(256) void head_fold_dgrad_kernel(const float* __restrict__ dz, const float* __restrict__ gp,
                                                              T* __restrict__ dd2) {
  __shared__ __attribute__((aligned(16))) float s_g[49 * 32];
  __shared__ float s_dz[13][56];                              // rows Y0-3 .. Y0+9, columns x = -3 .. 52 (zero outside the map)
  const int f = blockIdx.y, Y0 = blockIdx.x * 7;
  for (int i = threadIdx.x; i < 49 * 32; i += 256) s_g[i] = gp[i];
  for (int i = threadIdx.x; i < 13 * 56; i += 256) {
    const int ry = i / 56, xx = i - ry * 56;
    const int y = Y0 - 3 + ry, x = xx - 3;
    s_dz[ry][xx] = (y >= 0 && y < 49 && x >= 0 && x < 49) ? dz[((long long)f * 49 + y) * 49 + x] : 0.f;
  }
  __syncthreads();
  if (threadIdx.x >= 196) return;
  const int cg = threadIdx.x & 3, xg = (threadIdx.x >> 2) % 7, yl = threadIdx.x / 28;
  float acc[7][8];
#pragma unroll
  for (int xi = 0; xi < 7; ++xi)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[xi][i] = 0.f;
  for (int u = 0; u < 7; ++u) {
    // source row Y + 3 - u = s_dz row yl + 6 - u; pixel X = 7 xg + xi reads columns X + 3 - v = s_dz column X + 6 - v
    const float* drow = s_dz[yl + 6 - u] + 7 * xg;
    float d[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) d[k] = drow[k];
#pragma unroll
    for (int v = 0; v < 7; ++v) {
      const f32x4 g0 = *(const f32x4*)(s_g + (u * 7 + v) * 32 + cg * 8);
      const f32x4 g1 = *(const f32x4*)(s_g + (u * 7 + v) * 32 + cg * 8 + 4);
#pragma unroll
      for (int xi = 0; xi < 7; ++xi) {
        const float dv = d[xi + 6 - v];
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc[xi][i] += dv * g0[i]; acc[xi][4 + i] += dv * g1[i]; }
      }
    }
  }
#pragma unroll
  for (int xi = 0; xi < 7; ++xi)
    store8<T>(dd2 + (((long long)f * 49 + Y0 + yl) * 49 + 7 * xg + xi) * 32 + cg * 8, acc[xi], 8);
}

// ... and w.r.t. the folded filter: dGp[u,v,c] += sum_{f,y,x} dz[f,y,x] * d2pad[f, y+u, x+v, c].
// One block per frame; float atomics into the 1568-entry gradient (one add per element per frame).
template <typename T>
__global__ __launch_bounds__(256) void head_fold_wgrad_kernel(const float* __restrict__ dz, const T* __restrict__ d2pad,
                                                              float* __restrict__ dgp) {
  // one block per frame: each thread owns (tap, 8 channels) and sweeps the frame's 2401 pixels,
  // so a frame contributes ONE atomic per gradient element (per-row blocks contended 49x more)
  __shared__ float s_dz[2401];
  const int f = blockIdx.x;
  for (int i = threadIdx.x; i < 2401; i += 256) s_dz[i] = dz[(long long)f * 2401 + i];
  __syncthreads();
  const T* img = d2pad + (long long)f * 55 * 55 * 32;
  for (int it = threadIdx.x; it < 49 * 4; it += 256) {        // (tap, 8-channel group)
    const int tap = it >> 2, cg = it & 3;
    const int u = tap / 7, v = tap % 7;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int y = 0; y < 49; ++y) {
      const T* row = img + ((long long)(y + u) * 55 + v) * 32 + cg * 8;
      for (int x = 0; x < 49; ++x) {
        const float d = s_dz[y * 49 + x];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += d * Elem<T>::from(row[x * 32 + i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd(dgp + tap * 32 + cg * 8 + i, acc[i]);
  }
}

// Un-fold: dF3[a,b,o,c] = dG[a,b,c] * out_W[o],  d out_W[o] = sum_{a,b,c} dG[a,b,c] * F3[a,b,o,c],
// with dG[a,b,c] = dGp[6-a, 6-b, c]  (G = sum_o F3 * out_W, gaze_grcn.py:353-361).  One block.
static __global__ __launch_bounds__(256) void head_unfold_grads_kernel(const float* __restrict__ dgp, const float* __restrict__ f3,
                                                                const float* __restrict__ out_w, float* __restrict__ df3,
                                                                float* __restrict__ dout_w) {
  __shared__ float sh[4];
  float part[12];
#pragma unroll
  for (int o = 0; o < 12; ++o) part[o] = 0.f;
  for (int i = threadIdx.x; i < 49 * 32; i += 256) {
    const int tap = i / 32, c = i % 32;
    const int a = tap / 7, b = tap % 7;
    const float dg = dgp[((6 - a) * 7 + (6 - b)) * 32 + c];
#pragma unroll
    for (int o = 0; o < 12; ++o) {
      const long long idx = ((long long)tap * 12 + o) * 32 + c;
      df3[idx] = dg * out_w[o];
      part[o] += dg * f3[idx];
    }
  }
#pragma unroll
  for (int o = 0; o < 12; ++o) {
    const float r = block_reduce(part[o], sh, false);
    if (threadIdx.x == 0) dout_w[o] = r;
  }
}

// Batch-norm (inference form, one layer per timestep, gaze_grcn.py:325) backward:
//   y = gamma_t * h * inv + beta_t  =>  dgamma_t[c] = inv * sum_{b,p} dy*h, dbeta_t[c] = sum dy,
//   dh = dy * gamma_t[c] * inv.   One block per (t, 8 channels); dy rows are frame-major (b*T+t).
static __global__ __launch_bounds__(256) void bn_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ hall,
                                                     const float* __restrict__ gamma, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, float* __restrict__ dh_head, int B,
                                                     int T_, int S, float inv, int t0) {
  __shared__ float sh[4];
  const int t = blockIdx.y + t0, c0 = blockIdx.x * 8;      // (grid.y = T, t0 = 0: every step; grid.y = 1: step t0)
  float sg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int rows = B * 49;
  for (int r = threadIdx.x; r < rows; r += 256) {
    const int b = r / 49, p = r % 49;
    const float* d = dy + (((long long)b * T_ + t) * 49 + p) * S + c0;
    const float* h = hall + (((long long)(t + 1) * B + b) * 49 + p) * S + c0;     // h_t
    float* o = dh_head + (((long long)t * B + b) * 49 + p) * S + c0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      sg[i] += d[i] * h[i];
      sb[i] += d[i];
      o[i] = d[i] * gamma[(long long)t * S + c0 + i] * inv;
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float a = block_reduce(sg[i], sh, false);
    const float b2 = block_reduce(sb[i], sh, false);
    if (threadIdx.x == 0) {
      dgamma[(long long)t * S + c0 + i] = a * inv;
      dbeta[(long long)t * S + c0 + i] = b2;
    }
  }
}

// BPTT step, part 1 (gaze_grcn.py:122-127 differentiated):  dh = dh_head_t + dh_carry
//   du = dh*(h_prev - c)   dc = dh*(1-u)   dh_carry = dh*u
//   dc_pre = dc*(1-c^2)    dz_pre = du*u*(1-u)
// writes dXpre[:, z] and dXpre[:, c] (frame-major rows), dc_pre as a halo-padded T image
// (operand of the U dgrad conv), and the new carry.
template <typename T>
__global__ __launch_bounds__(256) void gru_bwd1_kernel(const float* __restrict__ dh_head, float* __restrict__ dh_carry,
                                                       const float* __restrict__ h_prev, const float* __restrict__ u,
                                                       const float* __restrict__ c, float* __restrict__ dxpre,
                                                       T* __restrict__ dcp_pad, const int* __restrict__ pad_tab, int B,
                                                       int T_, int t, int S, int first, float* __restrict__ drh_zero,
                                                       const float* __restrict__ dy_frames, const float* __restrict__ gamma_t,
                                                       float bn_inv) {
  const long long total = (long long)B * 49 * S;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ch = (int)(i % S);
    const int p = (int)((i / S) % 49);
    const int b = (int)(i / ((long long)S * 49));
    if (drh_zero) drh_zero[i] = 0.f;                  // the split-K U dgrad that follows adds its partial sums with atomics
    // dy_frames (stepwise backward from external state gradients, frame-major [b*T + t][49][S]): the batch-norm's input
    // gradient of this step is taken here instead of from a bn_bwd pass over all steps
    const float head = dy_frames ? dy_frames[(((long long)b * T_ + t) * 49 + p) * S + ch] * gamma_t[ch] * bn_inv : dh_head[i];
    const float dh = head + (first ? 0.f : dh_carry[i]);
    const float uu = u[i], cc = c[i];
    const float du = dh * (h_prev[i] - cc), dc = dh * (1.f - uu);
    dh_carry[i] = dh * uu;
    const float dcp = dc * (1.f - cc * cc), dzp = du * uu * (1.f - uu);
    float* row = dxpre + (((long long)b * T_ + t) * 49 + p) * 3 * S;
    row[ch] = dzp;
    row[2 * S + ch] = dcp;
    dcp_pad[(long long)b * 81 * S + pad_tab[p] + ch] = Elem<T>::to(dcp);
  }
}

// BPTT step, part 2: d(r*h) from the U dgrad conv ->
//   dr = drh*h_prev, dh_carry += drh*r, dr_pre = dr*r*(1-r); writes dXpre[:, r] and the
//   halo-padded T image [dz_pre | dr_pre] (operand of the Uz|Ur dgrad conv).
template <typename T>
__global__ __launch_bounds__(256) void gru_bwd2_kernel(const float* __restrict__ drh, float* __restrict__ dh_carry,
                                                       const float* __restrict__ h_prev, const float* __restrict__ r,
                                                       float* __restrict__ dxpre, T* __restrict__ dzr_pad,
                                                       const int* __restrict__ pad_tab2, int B, int T_, int t, int S) {
  const long long total = (long long)B * 49 * S;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ch = (int)(i % S);
    const int p = (int)((i / S) % 49);
    const int b = (int)(i / ((long long)S * 49));
    const float d = drh[i], rr = r[i];
    const float drp = d * h_prev[i] * rr * (1.f - rr);
    dh_carry[i] += d * rr;
    float* row = dxpre + (((long long)b * T_ + t) * 49 + p) * 3 * S;
    row[S + ch] = drp;
    T* img = dzr_pad + (long long)b * 81 * 2 * S + pad_tab2[p];
    img[ch] = Elem<T>::to(row[ch]);          // dz_pre written by part 1
    img[S + ch] = Elem<T>::to(drp);
  }
}

// rh[t] = r[t] * h[t]  (h[t] = h_{t-1} of step t; both [T][B][49][S] fp32)
static __global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = a[i] * b[i];
}

// fp32 [rows][C] -> halo-padded T image [rows/49][81][C]
template <typename T>
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                                       const int* __restrict__ pad_tab, long long total, int C) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int p = (int)((i / C) % 49);
    const long long f = i / ((long long)C * 49);
    dst[f * 81 * C + pad_tab[p] + c] = Elem<T>::to(src[i]);
  }
}

// The recurrent filter gradients' two operand images in one pass: h_{t-1} and r . h_{t-1} of every step, fp32 [rows][C]
// -> halo-padded T images (round 4: a multiply into a scratch array and two pad_rows launches)
template <typename T>
__global__ __launch_bounds__(256) void pad_h_rh_kernel(const float* __restrict__ r, const float* __restrict__ h, T* __restrict__ hp,
                                                       T* __restrict__ rhp, const int* __restrict__ pad_tab, long long total, int C) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int p = (int)((i / C) % 49);
    const long long f = i / ((long long)C * 49);
    const long long o = f * 81 * C + pad_tab[p] + c;
    const float hv = h[i];
    hp[o] = Elem<T>::to(hv);
    rhp[o] = Elem<T>::to(r[i] * hv);
  }
}

// ---- optimizer (base.py:262-308; TF semantics) ------------------------------
// partial[b] = sum of squares of block b's grid-stride share (deterministic two-stage norm)
// (16-byte loads, four in flight per thread: the scalar loop it replaces read 123 MB -- the 30.7 M gradients of the
// end-to-end model -- at 0.5 TB/s)
static __global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, long long n,
                                                             float* __restrict__ partial) {
  __shared__ float sh[4];
  const long long tid = (long long)blockIdx.x * 256 + threadIdx.x, stride = (long long)gridDim.x * 256;
  float a = 0.f;
  // leading elements up to a 16-byte boundary, whole 16-byte groups, tail
  const long long head = min(n, (long long)((16 - ((size_t)g & 15)) & 15) / 4);
  const long long n4 = (n - head) >> 2;
  const f32x4* g4 = (const f32x4*)(g + head);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 4
  for (long long i = tid; i < n4; i += stride) {
    const f32x4 v = g4[i];
    a0 += v[0] * v[0]; a1 += v[1] * v[1]; a2 += v[2] * v[2]; a3 += v[3] * v[3];
  }
  a = (a0 + a1) + (a2 + a3);
  for (long long i = tid; i < head; i += stride) a += g[i] * g[i];
  for (long long i = head + 4 * n4 + tid; i < n; i += stride) a += g[i] * g[i];
  a = block_reduce(a, sh, false);
  if (threadIdx.x == 0) partial[blockIdx.x] = a;
}

// clip_by_global_norm + AdamOptimizer.apply_gradients:
//   scale = clip / max(norm, clip);  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
//   theta -= lr_t * m / (sqrt(v) + eps),  lr_t = lr * sqrt(1-b2^t) / (1-b1^t)  (passed in).
static __global__ __launch_bounds__(256) void adam_clip_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, long long n,
                                                        const float* __restrict__ sq_partial, int n_partial, float clip,
                                                        float lr_t, float b1, float b2, float eps,
                                                        float* __restrict__ norm_out, const float* __restrict__ lr_t_dev) {
  if (lr_t_dev) lr_t = *lr_t_dev;       // step size computed on the device (graph-replayable training step)
  // global norm from the partial sums: every lane adds its share, then a butterfly over the wave -- each stage adds the two
  // partners' values (a + b == b + a exactly), so every lane of every wave ends with the SAME bits, hence the same scale.
  // (Round 4 had every thread add all partials one after the other: 256 - 512 dependent adds in front of a 43 us kernel.)
  float sq = 0.f;
  for (int i = threadIdx.x & 63; i < n_partial; i += 64) sq += sq_partial[i];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off, 64);
  const float norm = sqrtf(sq);
  const float scale = clip > 0.f ? clip / fmaxf(norm, clip) : 1.f;
  if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) *norm_out = norm;
  const float c1 = 1.f - b1, c2 = 1.f - b2;
  const long long tid = (long long)blockIdx.x * 256 + threadIdx.x, stride = (long long)gridDim.x * 256;
  auto upd = [&](float gi, float& mi, float& vi, float& pi) {
    gi *= scale;
    mi = b1 * mi + c1 * gi;
    vi = b2 * vi + c2 * gi * gi;
    pi -= lr_t * mi / (sqrtf(vi) + eps);
  };
  // whole 16-byte groups where the four arrays are 16-byte aligned (flat buffers are), then the tail
  const bool al = ((((size_t)p | (size_t)g | (size_t)m | (size_t)v) & 15) == 0);
  const long long n4 = al ? n >> 2 : 0;
  for (long long i = tid; i < n4; i += stride) {
    f32x4 G = ((const f32x4*)g)[i], M = ((f32x4*)m)[i], V = ((f32x4*)v)[i], P = ((f32x4*)p)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) { float mi = M[e], vi = V[e], pi = P[e]; upd(G[e], mi, vi, pi); M[e] = mi; V[e] = vi; P[e] = pi; }
    ((f32x4*)m)[i] = M; ((f32x4*)v)[i] = V; ((f32x4*)p)[i] = P;
  }
  for (long long i = 4 * n4 + tid; i < n; i += stride) {
    float mi = m[i], vi = v[i], pi = p[i];
    upd(g[i], mi, vi, pi);
    m[i] = mi; v[i] = vi; p[i] = pi;
  }
}

// Learning-rate schedule + Adam bias correction evaluated on the device, so that a captured training step
// (HIP graph) can be replayed: lr_t = lr0 * decay^floor(step/decay_steps) * sqrt(1-b2^t)/(1-b1^t), t = step+1
// (gaze_rnn.py:436-444, TF AdamOptimizer); then step += 1.
static __global__ void lr_schedule_kernel(int* __restrict__ step, float lr0, float decay, int decay_steps, float b1, float b2,
                                          float* __restrict__ lr_t) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int s = *step;
    const double t = (double)s + 1.0;
    const double lr = (double)lr0 * pow((double)decay, (double)(s / decay_steps));
    *lr_t = (float)(lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
    *step = s + 1;
  }
}

}  // namespace rgp
