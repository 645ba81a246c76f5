// The whole saliency head of gaze_grcn as ONE GEMM, for inference plans.
// Spec: /root/reference/models/gaze_grcn.py:292-314 (filters), 326-361 (the three transposed convolutions and out_W).
//
// Between BN(h_t) [7,7,128] and the logit map [49,49] the reference applies three transposed convolutions and a 12 -> 1
// projection with NO bias and NO non-linearity in between: the head is one linear map.  Folding
// deconv3 with out_W (rounds 1-3) is the first step of an exact algebra that goes all the way:
//
//   logit[y,x] = out_b + sum_{a,b,c} d2[y-a+3, x-b+3, c] G[a,b,c]            G = fold of weight3 with out_W  (7 x 7 x 32)
//   d2[2i+a', 2j+b', c] += d1[i,j,k] F2[a',b',c,k]                            (5 x 5, stride 2, VALID: 23 -> 49)
//   d1[3m+a", 3n+b", k] += y[m,n,s] F1[a",b",k,s]                             (5 x 5, stride 3, VALID:  7 -> 23)
//
//   => logit[y,x] = out_b + sum_{i,j,k} d1[i,j,k] H[y-2i, x-2j, k]            H[p,q,k] = sum_{a'+a-3 = p, b'+b-3 = q, c} G[a,b,c] F2[a',b',c,k]
//                                                                             p, q in [-3, 7]:  11 x 11 x 64
//   => logit[y,x] = out_b + sum_{m,n,s} y[m,n,s] K[y-6m, x-6n, s]             K[r,t,s] = sum_{2a"+p = r, 2b"+q = t, k} F1[a",b",k,s] H[p,q,k]
//                                                                             r, t in [-3, 15]: 19 x 19 x 128
//
// Exact, borders included: a VALID transposed convolution produces exactly the rows 0 .. 48 (2 * 22 + 4), so the zero
// padding of the SAME 7 x 7 stage never meets a value the fold would have to drop, and the output is simply restricted to
// 0 <= y, x < 49.  The head is then ONE transposed convolution (19 x 19, stride 6, S -> 1 channels), run as GEMM + col2im:
//   Z[(f,m,n), (r,t)] = sum_s y[f,m,n,s] K[(r,t), s]          igemm: M = frames x 49, K = S, N = 361 -> 384: 4.8 MFLOP per frame
//   logit[f,y,x] = out_b + sum_{m,n} Z[(f,m,n), (y-6m, x-6n)]  head_col2im_kernel: <= 4 x 4 terms per pixel, fp32
// instead of the 164.6 MFLOP of the three stages, with no 27 x 27 x 64 / 55 x 55 x 32 intermediates (293 MB per 1024 frames
// written and read back): two launches instead of six.  (First form of this round: the dense 6272 x 2401 matrix, 30.8 MFLOP
// per frame and a 30 MB filter -- 0.08 ms per 1024 frames, but 66 us at config 4's 280 frames: 190 tiles of 64 x 64 each
// streaming K = 6272.)  All reported rates keep dividing by the UNFOLDED 432.79 MFLOP per frame.  Fewer roundings
// than the staged bf16 pipeline (the intermediates are never rounded to bf16; the folded filter is, once).
//
// Training plans fold too (second half of round 4).  The backward never needs the dense matrix, only K and the patches
// Pm[(f,m,n)][(r,t)] = dz[f, 6m+r, 6n+t] of the logit gradient (19 x 19 = 361 taps, padded to 384, zero outside the map):
//
//   dK[(r,t), s]   = sum_{(f,m,n)} Pm[(f,m,n), (r,t)] y[f,m,n,s]              one wgrad_kernel launch (rows = the 7 x 7 positions)
//   dy[(f,m,n), s] = sum_{(r,t)}   Pm[(f,m,n), (r,t)] K[(r,t), s]             one GEMM, K = 384, N = S
//   dF1[a",b",k,s] = sum_{p,q} H[p,q,k] dK[2a"+p, 2b"+q, s]        dH[p,q,k] = sum_{a",b",s} F1[a",b",k,s] dK[2a"+p, 2b"+q, s]
//   dF2[a',b',c,k] = sum_{a,b} G[a,b,c] dH[a'+a-3, b'+b-3, k]      dG[a,b,c] = sum_{a',b',k} F2[a',b',c,k] dH[a'+a-3, b'+b-3, k]
//   dF3 = dG (x) out_W,  d out_W = <dG, F3>   (head_unfold_grads_kernel, as before)
//
// (the chain rule through the fold, checked against autograd to 1e-12 before it was written down here): 7 small launches
// and two GEMMs of 1.3 / 4.3 GFLOP at 280 frames replace three filter-gradient launches, two input-gradient GEMMs, the
// Toeplitz filter gradient and the intermediate maps d1 / d2 / dd1 / dd2.  RGP_GRCN_UNFOLDED_HEAD keeps the three stages, forward
// and backward: the library's second implementation of the head, which the tests compare this one with.
#pragma once
#include "igemm.hip.h"

namespace rgp {

constexpr int HF_HP = 11, HF_KP = 19;      // taps of H and K per axis
constexpr int HF_PK = 384;                 // the 361 taps of K padded to a multiple of the K-chunk (64 bf16 / 32 fp32 elements)

// H[(p+3)*11 + q+3][k] from G [7*7][32] (fold_head_filter_kernel) and weight2 [5,5,32,64] = (kh, kw, out, in)
static __global__ void head_fold_h_kernel(const float* __restrict__ g, const float* __restrict__ f2, float* __restrict__ h) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= HF_HP * HF_HP * 64) return;
  const int k = i % 64, q = (i / 64) % HF_HP - 3, p = i / (64 * HF_HP) - 3;
  float s = 0.f;
  for (int a = 0; a < 7; ++a) {
    const int a1 = p - a + 3;
    if (a1 < 0 || a1 > 4) continue;
    for (int b = 0; b < 7; ++b) {
      const int b1 = q - b + 3;
      if (b1 < 0 || b1 > 4) continue;
      const float* gp = g + (a * 7 + b) * 32;
      const float* fp = f2 + ((long long)(a1 * 5 + b1) * 32) * 64 + k;
      for (int c = 0; c < 32; ++c) s += gp[c] * fp[(long long)c * 64];
    }
  }
  h[i] = s;
}

// K[(r+3)*19 + t+3][s] = sum_a part[a][(r,t)][s],  part[a][(r,t)][s] = sum_{b,k} F1[a,b,k,s] H[r-2a, t-2b, k]: block = ((r,t), a),
// thread = s, then head_fold_sum_kernel adds the five parts in a fixed order (deterministic: two set_weights calls with the
// same weights give the same bits; a thread per (r,t,s) looping over all 25 x 64 terms took 57 us -- 180 blocks of serial loads)
static __global__ void head_fold_k_kernel(const float* __restrict__ h, const float* __restrict__ f1, float* __restrict__ part, int S) {
  const int rt = blockIdx.x, a = blockIdx.y;
  const int r = rt / HF_KP - 3, t = rt % HF_KP - 3;
  const int p = r - 2 * a;
  float* dst = part + ((long long)a * HF_KP * HF_KP + rt) * S;
  for (int s = threadIdx.x; s < S; s += blockDim.x) {
    float acc = 0.f;
    if (p >= -3 && p <= 7) {
      for (int b = 0; b < 5; ++b) {
        const int q = t - 2 * b;
        if (q < -3 || q > 7) continue;
        const float* hp = h + ((p + 3) * HF_HP + q + 3) * 64;
        const float* fp = f1 + ((long long)(a * 5 + b) * 64) * S + s;
#pragma unroll 8
        for (int k = 0; k < 64; ++k) acc += hp[k] * fp[(long long)k * S];
      }
    }
    dst[s] = acc;
  }
}

// out[i] = sum_{j < n_parts} part[j][i]   (fixed order)
static __global__ void head_fold_sum_kernel(const float* __restrict__ part, float* __restrict__ out, int n, int n_parts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float acc = 0.f;
  for (int j = 0; j < n_parts; ++j) acc += part[(long long)j * n + i];
  out[i] = acc;
}

// Forward, second half (col2im of the transposed convolution): the GEMM  Z[(f,m,n)][(r,t)] = sum_s y[f,m,n,s] K[(r,t),s]
// (M = frames x 49, K = S, N = 384: 1/6 of the dense matrix's FLOPs, no 30 MB filter) leaves every product of a 7x7 position
// with the 19x19 filter; a logit pixel gathers its <= 4 x 4 contributions:  logit[f,y,x] = out_b + sum_{m,n} Z[(f,m,n)][(y-6m, x-6n)]
static __global__ __launch_bounds__(256) void head_col2im_kernel(const float* __restrict__ z, const float* __restrict__ out_b,
                                                                float* __restrict__ logits, long long total) {
  const float bias = out_b[0];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int pix = (int)(i % 2401);
    const long long f = i / 2401;
    const int y = pix / 49, x = pix - y * 49;
    // m with -3 <= y - 6m <= 15  <=>  (y - 15) / 6 <= m <= (y + 3) / 6
    const int m0 = y > 15 ? (y - 10) / 6 : 0, m1 = min(6, (y + 3) / 6);      // ceil((y - 15) / 6) = (y - 10) / 6 for y >= 16
    const int n0 = x > 15 ? (x - 10) / 6 : 0, n1 = min(6, (x + 3) / 6);
    float acc = bias;
    for (int m = m0; m <= m1; ++m)
      for (int n = n0; n <= n1; ++n)
        acc += z[((f * 49 + m * 7 + n) * HF_PK) + (y - 6 * m + 3) * HF_KP + (x - 6 * n + 3)];
    logits[i] = acc;
  }
}

// Pm[(f, m, n)][(r+3)*19 + t+3] = dz[f, 6m+r, 6n+t] for r, t in [-3, 15] inside the map, else 0; columns 361 .. 383 zero
template <typename T>
static __global__ void head_fold_patches_kernel(const float* __restrict__ dz, T* __restrict__ pm, long long rows) {
  const long long total = rows * HF_PK;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int k = (int)(i % HF_PK);
    const long long row = i / HF_PK;
    const int pos = (int)(row % 49);
    const long long f = row / 49;
    float v = 0.f;
    if (k < HF_KP * HF_KP) {
      const int y = 6 * (pos / 7) + k / HF_KP - 3, x = 6 * (pos % 7) + k % HF_KP - 3;
      if (y >= 0 && y < 49 && x >= 0 && x < 49) v = dz[f * 2401 + y * 49 + x];
    }
    pm[i] = Elem<T>::to(v);
  }
}

// dF1[a,b,k,s] = sum_{p,q} H[p,q,k] dK[2a+p, 2b+q, s]          (dk: [HF_PK][S], row (r+3)*19 + t+3)
static __global__ void head_unfold_f1_kernel(const float* __restrict__ dk, const float* __restrict__ h, float* __restrict__ df1, int S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 25 * 64 * S) return;
  const int s = i % S, k = (i / S) % 64, ab = i / (S * 64), a = ab / 5, b = ab % 5;
  float acc = 0.f;
  for (int p = 0; p < HF_HP; ++p)
    for (int q = 0; q < HF_HP; ++q)
      acc += h[(p * HF_HP + q) * 64 + k] * dk[((long long)(2 * a + p) * HF_KP + 2 * b + q) * S + s];
  df1[i] = acc;
}

// dH[p,q,k] = sum_{(a,b)} part[(a,b)][(p,q)][k],  part = sum_s F1[a,b,k,s] dK[2a+p, 2b+q, s]: block = ((p,q), (a,b)), thread = k x 4
// contiguous quarters of s (16-byte loads); head_fold_sum_kernel adds the 25 parts in a fixed order
static __global__ __launch_bounds__(256) void head_unfold_h_kernel(const float* __restrict__ dk, const float* __restrict__ f1,
                                                                  float* __restrict__ part, int S) {
  __shared__ float red[256];
  const int pq = blockIdx.x, p = pq / HF_HP, q = pq % HF_HP, ab = blockIdx.y, a = ab / 5, b = ab % 5;
  const int k = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int S4 = S / 4;                                       // floats per quarter (S is a multiple of 64)
  const f32x4* dkr = (const f32x4*)(dk + ((long long)(2 * a + p) * HF_KP + 2 * b + q) * S + sl * S4);
  const f32x4* fr = (const f32x4*)(f1 + ((long long)ab * 64 + k) * S + sl * S4);
  float acc = 0.f;
  for (int i = 0; i < S4 / 4; ++i) {
    const f32x4 x = fr[i], y = dkr[i];
    acc += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < 64) part[((long long)ab * HF_HP * HF_HP + pq) * 64 + k] = (red[k] + red[64 + k]) + (red[128 + k] + red[192 + k]);
}

// dF2[a',b',c,k] = sum_{a,b} G[a,b,c] dH[a'+a-3, b'+b-3, k]    (dh index p+3 = a'+a)
static __global__ void head_unfold_f2_kernel(const float* __restrict__ dh, const float* __restrict__ g, float* __restrict__ df2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 25 * 32 * 64) return;
  const int k = i % 64, c = (i / 64) % 32, ab = i / (64 * 32), a1 = ab / 5, b1 = ab % 5;
  float acc = 0.f;
  for (int a = 0; a < 7; ++a)
    for (int b = 0; b < 7; ++b) acc += g[(a * 7 + b) * 32 + c] * dh[((a1 + a) * HF_HP + b1 + b) * 64 + k];
  df2[i] = acc;
}

// dGp[6-a, 6-b, c] = dG[a,b,c] = sum_{a',b',k} F2[a',b',c,k] dH[a'+a-3, b'+b-3, k]     (flipped: what head_unfold_grads_kernel reads)
// block = tap (a,b), thread = (c, slice of 8 k)
static __global__ __launch_bounds__(256) void head_unfold_g_kernel(const float* __restrict__ dh, const float* __restrict__ f2,
                                                                  float* __restrict__ dgp) {
  __shared__ float red[256];
  const int tap = blockIdx.x, a = tap / 7, b = tap % 7;
  const int c = threadIdx.x & 31, ks = threadIdx.x >> 5;
  float acc = 0.f;
  for (int ab = 0; ab < 25; ++ab) {
    const int a1 = ab / 5, b1 = ab % 5;
    const float* fr = f2 + ((long long)ab * 32 + c) * 64 + ks * 8;
    const float* dr = dh + ((a1 + a) * HF_HP + b1 + b) * 64 + ks * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += fr[k] * dr[k];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < 32) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) t += red[j * 32 + c];
    dgp[((6 - a) * 7 + (6 - b)) * 32 + c] = t;
  }
}

}  // namespace rgp
