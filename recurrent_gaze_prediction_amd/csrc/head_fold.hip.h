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
// 0 <= y, x < 49.  As a matrix: logits[f, (y,x)] = sum_{(m,n,s)} Y[f, (m,n,s)] Wd[(m,n,s), (y,x)], a 6272 x 2401 matrix with
// 15 % non-zeros, run DENSE: 30.8 MFLOP per frame instead of the 164.6 of the three stages, no 27 x 27 x 64 / 55 x 55 x 32
// intermediates (293 MB per 1024 frames written and read back), one launch instead of six.  All reported rates keep
// dividing by the UNFOLDED 432.79 MFLOP per frame.  Fewer roundings than the staged bf16 pipeline (the
// intermediates are never rounded to bf16; the folded filter is, once).
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

// K[(r+3)*19 + t+3][s] from H and weight1 [5,5,64,S] = (kh, kw, out, in)
static __global__ void head_fold_k_kernel(const float* __restrict__ h, const float* __restrict__ f1, float* __restrict__ kf, int S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= HF_KP * HF_KP * S) return;
  const int s = i % S, t = (i / S) % HF_KP - 3, r = i / (S * HF_KP) - 3;
  float acc = 0.f;
  for (int a = 0; a < 5; ++a) {
    const int p = r - 2 * a;
    if (p < -3 || p > 7) continue;
    for (int b = 0; b < 5; ++b) {
      const int q = t - 2 * b;
      if (q < -3 || q > 7) continue;
      const float* hp = h + ((p + 3) * HF_HP + q + 3) * 64;
      const float* fp = f1 + ((long long)(a * 5 + b) * 64) * S + s;
      for (int k = 0; k < 64; ++k) acc += hp[k] * fp[(long long)k * S];
    }
  }
  kf[i] = acc;
}

// Wd[tap = m*7+n][s][col = y*49+x] = K[y-6m, x-6n, s] (0 outside its 19 x 19 taps);  bias[col] = out_b
static __global__ void head_fold_expand_kernel(const float* __restrict__ kf, const float* __restrict__ out_b, float* __restrict__ wd,
                                               float* __restrict__ bias, int S, int n_bias) {
  const long long total = 49LL * S * 2401;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int col = (int)(i % 2401);
    const int s = (int)((i / 2401) % S);
    const int tap = (int)(i / (2401LL * S));
    const int y = col / 49, x = col - y * 49, m = tap / 7, n = tap - m * 7;
    const int r = y - 6 * m + 3, t = x - 6 * n + 3;
    wd[i] = (r >= 0 && r < HF_KP && t >= 0 && t < HF_KP) ? kf[((long long)r * HF_KP + t) * S + s] : 0.f;
    if (i < n_bias) bias[i] = out_b[0];
  }
}

// The packed GEMM filter straight from K, in the operand type: dst[col][(m*7+n)*S + s] = K[y-6m, x-6n, s], rows col >= 2401
// zero; bias[col] = out_b.  (Training plans re-fold after every optimizer step: 30 MB written, no fp32 intermediate.)
template <typename T>
static __global__ void head_fold_pack_kernel(const float* __restrict__ kf, const float* __restrict__ out_b, T* __restrict__ dst,
                                             float* __restrict__ bias, int S, int n_pad) {
  const int SG = S / 8;
  const long long total = (long long)n_pad * 49 * SG;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int sg = (int)(i % SG);
    const int tap = (int)((i / SG) % 49);
    const int col = (int)(i / (49LL * SG));
    const int y = col / 49, x = col - y * 49, m = tap / 7, n = tap - m * 7;
    const int r = y - 6 * m + 3, t = x - 6 * n + 3;
    float v[8];
    if (col < 2401 && r >= 0 && r < HF_KP && t >= 0 && t < HF_KP) {
      const float* src = kf + ((long long)r * HF_KP + t) * S + sg * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = src[k];
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = 0.f;
    }
    store8<T>(dst + (long long)col * (49 * S) + tap * S + sg * 8, v, 8);
    if (i < n_pad) bias[i] = out_b[0];
  }
}

constexpr int HF_PK = 384;                  // 361 taps of K padded to a multiple of the K-chunk (64 bf16 / 32 fp32 elements)

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

// dH[p,q,k] = sum_{a,b,s} F1[a,b,k,s] dK[2a+p, 2b+q, s]        (one block per (p,q), thread = k x 4 slices of s)
static __global__ __launch_bounds__(256) void head_unfold_h_kernel(const float* __restrict__ dk, const float* __restrict__ f1,
                                                                  float* __restrict__ dh, int S) {
  __shared__ float red[256];
  const int pq = blockIdx.x, p = pq / HF_HP, q = pq % HF_HP;
  const int k = threadIdx.x & 63, sl = threadIdx.x >> 6;
  float acc = 0.f;
  for (int ab = 0; ab < 25; ++ab) {
    const int a = ab / 5, b = ab % 5;
    const float* dkr = dk + ((long long)(2 * a + p) * HF_KP + 2 * b + q) * S;
    const float* fr = f1 + ((long long)ab * 64 + k) * S;
    for (int s = sl; s < S; s += 4) acc += fr[s] * dkr[s];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < 64) dh[pq * 64 + k] = red[k] + red[64 + k] + red[128 + k] + red[192 + k];
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
static __global__ void head_unfold_g_kernel(const float* __restrict__ dh, const float* __restrict__ f2, float* __restrict__ dgp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 49 * 32) return;
  const int c = i % 32, tap = i / 32, a = tap / 7, b = tap % 7;
  float acc = 0.f;
  for (int ab = 0; ab < 25; ++ab) {
    const int a1 = ab / 5, b1 = ab % 5;
    const float* fr = f2 + ((long long)ab * 32 + c) * 64;
    const float* dr = dh + ((a1 + a) * HF_HP + b1 + b) * 64;
    for (int k = 0; k < 64; ++k) acc += fr[k] * dr[k];
  }
  dgp[((6 - a) * 7 + (6 - b)) * 32 + c] = acc;
}

}  // namespace rgp
