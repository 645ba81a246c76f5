// librgp_hip.so: element-wise training ops that are not tied to one plan.
//
//  * inverted dropout as an op (tf.nn.dropout, /root/reference/models/gaze_rnn.py:302-303,529 and
//    gaze_grcn_cascade.py:401-402): a counter-based Philox-4x32-10 mask generator on the device, the
//    mask is kept (one byte per element) so the backward pass gates with the SAME draw;
//  * the l2 loss of gaze_rnn.py:387-389 / gaze_grcn_cascade.py:428-441;
//  * the two optimizers of base.py:268-273 next to Adam: RMSPropOptimizer(lr, momentum=0.9) and
//    MomentumOptimizer(lr, momentum=0.9), each fused with clip_by_global_norm like rgp_adam_clip_step_ext.
//
// All kernels are HBM-streaming (one pass, 16 B per lane where the layout allows).
#include <algorithm>

#include "rgp_host.h"
#include "bwd_kernels.hip.h"

using namespace rgp;

namespace {

// Philox-4x32-10 (Salmon et al., SC'11): counter (c0..c3), key (k0,k1) -> 4 x 32 random bits.
__device__ __forceinline__ void philox4x32_10(unsigned c[4], unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

// mask[i] = floor(keep + u_i), u_i uniform in [0,1) with 24 random bits (tf.nn.dropout's own rule:
// random_tensor = keep_prob + random_uniform; binary_tensor = floor(random_tensor)).
// Element i uses word (i & 3) of the Philox block with counter (offset + i / 4): the mask does not depend on the
// launch geometry, so a caller can regenerate any slice of it.
__global__ __launch_bounds__(256) void dropout_mask_kernel(unsigned char* __restrict__ mask, long long n, float keep,
                                                           unsigned long long seed, unsigned long long offset) {
  const long long nblk = (n + 3) / 4;
  for (long long b = (long long)blockIdx.x * 256 + threadIdx.x; b < nblk; b += (long long)gridDim.x * 256) {
    const unsigned long long ctr = offset + (unsigned long long)b;
    unsigned c[4] = {(unsigned)ctr, (unsigned)(ctr >> 32), 0u, 0u};
    philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
    unsigned char m[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float u = (float)(c[j] >> 8) * (1.0f / 16777216.0f);
      m[j] = floorf(keep + u) >= 1.0f ? 1 : 0;
    }
    const long long i = b * 4;
    if (i + 3 < n) *(uchar4*)(mask + i) = make_uchar4(m[0], m[1], m[2], m[3]);
    else for (int j = 0; j < 4 && i + j < n; ++j) mask[i + j] = m[j];
  }
}

// x[r][c] = x[r][c] * mask[r*cols + c] / keep   for c < cols (rows are ld apart)
template <typename T>
__global__ __launch_bounds__(256) void dropout_apply_kernel(T* __restrict__ x, const unsigned char* __restrict__ mask, long long rows,
                                                            int cols, long long ld, float inv_keep) {
  const long long total = rows * cols;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / cols;
    const int c = (int)(i - r * cols);
    T* p = x + r * ld + c;
    *p = Elem<T>::to(mask[i] ? Elem<T>::from(*p) * inv_keep : 0.f);
  }
}

// loss = scale * sum 0.5 (a - b)^2, deterministic two-stage reduction (gaze_rnn.py:387-389)
__global__ __launch_bounds__(256) void l2_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n,
                                                         float* __restrict__ partial) {
  __shared__ float sh[4];
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float d = a[i] - b[i];
    acc += 0.5f * d * d;
  }
  acc = block_reduce(acc, sh, false);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
__global__ void l2_final_kernel(const float* __restrict__ partial, int n, float scale, float* __restrict__ loss) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += partial[i];
    *loss = s * scale;
  }
}

__device__ __forceinline__ float clip_scale(const float* __restrict__ sq_partial, int n_partial, float clip, float* norm_out) {
  float sq = 0.f;
  for (int i = 0; i < n_partial; ++i) sq += sq_partial[i];
  const float norm = sqrtf(sq);
  if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) *norm_out = norm;
  return clip > 0.f ? clip / fmaxf(norm, clip) : 1.f;
}

// tf.train.MomentumOptimizer(lr, momentum): accum = momentum * accum + g;  var -= lr * accum
__global__ __launch_bounds__(256) void momentum_clip_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ accum,
                                                            long long n, const float* __restrict__ sq_partial, int n_partial, float clip,
                                                            float lr, float momentum, float* __restrict__ norm_out) {
  const float scale = clip_scale(sq_partial, n_partial, clip, norm_out);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float a = momentum * accum[i] + g[i] * scale;
    accum[i] = a;
    p[i] -= lr * a;
  }
}

// tf.train.RMSPropOptimizer(lr, decay, momentum, epsilon):
//   ms = decay * ms + (1 - decay) * g^2;  mom = momentum * mom + lr * g / sqrt(ms + epsilon);  var -= mom
// (TF initialises the ms slot to ONES and mom to zeros -- the caller provides the slots.)
__global__ __launch_bounds__(256) void rmsprop_clip_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ ms,
                                                           float* __restrict__ mom, long long n, const float* __restrict__ sq_partial,
                                                           int n_partial, float clip, float lr, float decay, float momentum, float eps,
                                                           float* __restrict__ norm_out) {
  const float scale = clip_scale(sq_partial, n_partial, clip, norm_out);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float gi = g[i] * scale;
    const float m2 = decay * ms[i] + (1.f - decay) * gi * gi;
    const float mo = momentum * mom[i] + lr * gi / sqrtf(m2 + eps);
    ms[i] = m2;
    mom[i] = mo;
    p[i] -= mo;
  }
}

inline int nblocks(long long n, int cap = 4096) { return (int)std::min<long long>((n + 255) / 256, cap); }

}  // namespace

namespace rgp {
// used by the plans that own a dropout site (rgp_fcgru.hip, rgp_cascade.hip)
int dropout_apply(void* x, int dtype, const unsigned char* mask, long long rows, int cols, long long ld, float keep, hipStream_t s) {
  if (rows <= 0 || cols <= 0) return RGP_OK;
  const float inv = 1.0f / keep;
  if (dtype == RGP_BF16) dropout_apply_kernel<bf16_t><<<nblocks(rows * cols), 256, 0, s>>>((bf16_t*)x, mask, rows, cols, ld, inv);
  else dropout_apply_kernel<float><<<nblocks(rows * cols), 256, 0, s>>>((float*)x, mask, rows, cols, ld, inv);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}
}  // namespace rgp

extern "C" {

int rgp_dropout_mask(unsigned char* mask, long long n, float keep_prob, unsigned long long seed, unsigned long long offset,
                     rgp_stream_t stream) {
  RGP_REQUIRE(mask && n > 0, "rgp_dropout_mask: bad arguments");
  RGP_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f, "rgp_dropout_mask: keep_prob %g not in (0, 1]", (double)keep_prob);
  dropout_mask_kernel<<<nblocks((n + 3) / 4), 256, 0, (hipStream_t)stream>>>(mask, n, keep_prob, seed, offset);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

int rgp_dropout_apply(float* x, const unsigned char* mask, long long n, float keep_prob, rgp_stream_t stream) {
  RGP_REQUIRE(x && mask && n > 0, "rgp_dropout_apply: bad arguments");
  RGP_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f, "rgp_dropout_apply: keep_prob %g not in (0, 1]", (double)keep_prob);
  return dropout_apply(x, RGP_F32, mask, 1, (int)std::min<long long>(n, 1 << 30), n, keep_prob, (hipStream_t)stream);
}

int rgp_l2_loss_fwd(const float* maps, const float* labels, long long n, int frames, float* workspace, float* loss,
                    rgp_stream_t stream) {
  RGP_REQUIRE(maps && labels && workspace && loss && n > 0 && frames > 0, "rgp_l2_loss_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  l2_partial_kernel<<<RGP_SQNORM_PARTIALS, 256, 0, s>>>(maps, labels, n, workspace);
  l2_final_kernel<<<1, 64, 0, s>>>(workspace, RGP_SQNORM_PARTIALS, 1.0f / (float)frames, loss);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

int rgp_momentum_clip_step(float* params, const float* grads, float* accum, long long n, const float* partials, int n_partials,
                           float lr, float momentum, float max_grad_norm, float* grad_norm_out, rgp_stream_t stream) {
  RGP_REQUIRE(params && grads && accum && partials && n > 0 && n_partials > 0, "rgp_momentum_clip_step: bad arguments");
  momentum_clip_kernel<<<nblocks(n), 256, 0, (hipStream_t)stream>>>(params, grads, accum, n, partials, n_partials, max_grad_norm, lr,
                                                                    momentum, grad_norm_out);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

int rgp_rmsprop_clip_step(float* params, const float* grads, float* ms, float* mom, long long n, const float* partials,
                          int n_partials, float lr, float decay, float momentum, float eps, float max_grad_norm,
                          float* grad_norm_out, rgp_stream_t stream) {
  RGP_REQUIRE(params && grads && ms && mom && partials && n > 0 && n_partials > 0, "rgp_rmsprop_clip_step: bad arguments");
  rmsprop_clip_kernel<<<nblocks(n), 256, 0, (hipStream_t)stream>>>(params, grads, ms, mom, n, partials, n_partials, max_grad_norm, lr,
                                                                   decay, momentum, eps, grad_norm_out);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

}  // extern "C"
