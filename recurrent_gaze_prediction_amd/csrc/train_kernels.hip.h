// Small element-wise / reduction kernels shared by the backward passes of the maxout read-outs
// (cascade gaze_grcn_cascade.py:383-423, ShallowNet saliency_shallownet.py:139-185).
#pragma once
#include "igemm.hip.h"

namespace rgp {

constexpr int kFcN2 = 4864;   // 4802 FC outputs padded to a multiple of 128 (row width of the gradient rows)

// dz[f+1][j] / dz[f+1][2401+j] = gradient of maxout unit j routed to the half that won (mask 1 / 2), 0 if
// ReLU-gated.  The unit's gradient is (a - b) * scale (loss layer: maps - gt) or a * scale (b == null).
// Row 0 of dz stays zero (wgrad_kernel's out-of-range rows read it).
template <typename T>
__global__ __launch_bounds__(256) void maxout_bwd_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b,
                                                         float scale, const unsigned char* __restrict__ mask, T* __restrict__ dz,
                                                         long long total) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int j = (int)(i % 2401);
    const long long f = i / 2401;
    const float v = (a[f * lda + j] - (b ? b[i] : 0.f)) * scale;
    const unsigned char m = mask[i];
    T* row = dz + (f + 1) * kFcN2;
    row[j] = Elem<T>::to(m == 1 ? v : 0.f);
    row[2401 + j] = Elem<T>::to(m == 2 ? v : 0.f);
  }
}

// bias gradient of an FC: db[col] = sum_f dz[f+1][col].  Block = 32 columns x 8 row lanes (a thread per column looping
// over all F rows -- 19 blocks, F dependent loads each -- took 0.19 ms at 512 frames); launch with (4802 + 31) / 32 blocks.
template <typename T>
__global__ __launch_bounds__(256) void fc_bias_grad_kernel(const T* __restrict__ dz, int F, float* __restrict__ db) {
  __shared__ float red[8][33];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + cl;
  float a = 0.f;
  if (col < 4802)
    for (int f = rl; f < F; f += 8) a += Elem<T>::from(dz[(long long)(f + 1) * kFcN2 + col]);
  red[rl][cl] = a;
  __syncthreads();
  if (rl == 0 && col < 4802) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) t += red[r][cl];
    db[col] = t;
  }
}

}  // namespace rgp
