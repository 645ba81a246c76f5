// librgp_hip.so: the frame-wise ShallowNet saliency model (BASELINE config 1).
// Reference graph: /root/reference/models/saliency_shallownet.py:74-216
// (SaliencyModel.create_shallownet, dropout off as in the gaze models), used by
// models/gaze_framewise_shallownet.py:62-90.
//   conv 5x5 VALID 3->32 +b ReLU, maxpool 2x2/2 SAME | conv 3x3 VALID 32->64 +b ReLU,
//   maxpool 3x3/2 SAME | conv 3x3 VALID 64->32 +b ReLU, maxpool 3x3/2 SAME | flatten (NHWC)
//   fc 4802 +b ReLU maxout(halves) | fc 4802 +b ReLU maxout -> [N,49,49]
// The convolutions and both FC layers are igemm_kernel launches (conv1 with its 2x2 pool
// fused, the FCs with a fused ReLU+maxout epilogue over an interleaved filter packing);
// the two overlapping 3x3/2 pools are a streaming kernel.
#include <algorithm>

#include "train_kernels.hip.h"
#include "wgrad_launch.h"
#include "shallow_conv1.hip.h"

using namespace rgp;

struct rgp_shallownet {
  int N = 0, IH = 0, dtype = RGP_F32;
  int c1 = 0, p1 = 0, c2 = 0, p2 = 0, c3 = 0, p3 = 0, nflat = 0, Kf = 0, K2 = 0;
  ConvDesc conv1, conv2, conv3, fc1, fc2;
  size_t frames4 = 0, pool1 = 0, act2 = 0, pool2 = 0, act3 = 0, pool3 = 0, mo1 = 0, b1i = 0, b2i = 0;
  size_t ws_bytes = 0;
  char* ws = nullptr;
  SideStream side;           // backward: filter and bias gradients beside the data-gradient chain
  bool weights_set = false;
  const float *b_conv1 = nullptr, *b_conv2 = nullptr, *b_conv3 = nullptr;
  // ---- training ----
  bool save = false;
  int last_n = 0;
  size_t amax1 = 0, mask1 = 0, mask2 = 0;         // conv1 pool arg-max [n][p1*p1][32]; maxout masks [n][2401]
  size_t amax2 = 0, amax3 = 0;                    // pool2 / pool3 first-maximum places [n][p2*p2][64], [n][p3*p3][32]
  ConvDesc b_fc2, b_fc1, b_c3, b_c2;               // input-gradient GEMMs / "full" correlations with the rotated filters
  size_t dz2 = 0, dz1 = 0;                         // T [n+1][kFcN2]
  size_t dmo1 = 0, dpool3 = 0, dpool2 = 0, dpool1 = 0;   // fp32 [n][K2], [n][Kf], [n][p2*p2*64], [n][p1*p1*32]
  size_t dy3 = 0, dy2 = 0;                         // T [n][(c+4)^2][C]: gradient w.r.t. the conv outputs, halo 2
  size_t dy1 = 0;                                  // T [128 zeros][n][c1*c1][32]
  size_t dw1 = 0;                                  // fp32 conv1 filter gradient in its packed K order
};

namespace {

// interleave the two halves of a bias: out[2j] = b[j], out[2j+1] = b[j + half]
__global__ void interleave_kernel(const float* __restrict__ b, float* __restrict__ out, int half) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < half) {
    out[2 * j] = b[j];
    out[2 * j + 1] = b[j + half];
  }
}

template <typename T>
int set_weights_impl(rgp_shallownet* g, const rgp_shallownet_weights* w, hipStream_t s) {
  char* ws = g->ws;
  // every pack of this call in one or two launches; no memset of the packed areas (zero from bind time outside the
  // positions a pack writes, rgp_grcn.hip set_weights_impl)
  PackBatch<T> pk(ws, s);
  RGP_TRY(pk.add(g->conv1, w->conv1_w, 32, 0));
  RGP_TRY(pk.add(g->conv2, w->conv2_w, 64, 0));
  RGP_TRY(pk.add(g->conv3, w->conv3_w, 32, 0));
  // FC filters [K][4802], halves interleaved: packed row 2j = unit j, 2j+1 = unit j+2401
  RGP_TRY(pk.add(g->fc1, w->fc1_w, 2401, 0, 0, 0, 2));
  RGP_TRY(pk.add(g->fc1, w->fc1_w + 2401, 2401, 1, 0, 0, 2));
  RGP_TRY(pk.add(g->fc2, w->fc2_w, 2401, 0, 0, 0, 2));
  RGP_TRY(pk.add(g->fc2, w->fc2_w + 2401, 2401, 1, 0, 0, 2));
  interleave_kernel<<<(2401 + 255) / 256, 256, 0, s>>>(w->fc1_b, (float*)(ws + g->b1i), 2401);
  interleave_kernel<<<(2401 + 255) / 256, 256, 0, s>>>(w->fc2_b, (float*)(ws + g->b2i), 2401);
  RGP_HIP(hipGetLastError());
  g->b_conv1 = w->conv1_b; g->b_conv2 = w->conv2_b; g->b_conv3 = w->conv3_b;
  if (g->save) {
    RGP_TRY(pk.add(g->b_fc2, w->fc2_w, 2401, 0));
    RGP_TRY(pk.add(g->b_fc1, w->fc1_w, g->nflat, 0));
    RGP_TRY(pk.add(g->b_c3, w->conv3_w, 64, 0));
    RGP_TRY(pk.add(g->b_c2, w->conv2_w, 32, 0));
  }
  RGP_TRY(pk.flush());
  g->weights_set = true;
  return RGP_OK;
}

template <typename T>
int pool(const T* src, T* dst, int N, int H, int C, int k, int st, long long ld_out, hipStream_t s, unsigned char* amax = nullptr) {
  const int OH = (H + st - 1) / st;
  const int pad = std::max((OH - 1) * st + k - H, 0) / 2;
  const long long total = (long long)N * OH * OH * (C / 8);
  maxpool_same_kernel<T><<<(int)std::min<long long>((total + 255) / 256, 8192), 256, 0, s>>>(src, dst, N, H, H, C, k, st, OH, OH,
                                                                                      pad, pad, ld_out, amax);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

template <typename T>
int forward_impl(rgp_shallownet* g, const float* frames, int n, float* sal, float* sal7, hipStream_t s) {
  char* ws = g->ws;
  constexpr int G32 = sizeof(T) == 2 ? 2 : 1;     // 32-element taps: two per 128-byte chunk in bf16
  const long long npix = (long long)n * g->IH * g->IH;
  g->last_n = n;
  frame_prep_kernel<T><<<(int)std::min<long long>((npix + 255) / 256, 8192), 256, 0, s>>>(frames, (T*)(ws + g->frames4), npix);
  RGP_HIP(hipGetLastError());
  // bf16, 112 / 98 pixel frames, enough frames to give the CUs a workgroup each: the frame kernel (shallow_conv1.hip.h);
  // a handful of frames (config 1's own batch of 2) spreads better as tiles of the general kernel (0.11 vs 0.14 ms)
  if (sizeof(T) == 2 && shallow_conv1_covers(g->IH) && n >= 48) {
    ShallowConv1Params q;
    q.frames4 = (const bf16_t*)(ws + g->frames4); q.wp = (const bf16_t*)(ws + g->conv1.w_off); q.bias = g->b_conv1;
    q.pool1 = (bf16_t*)(ws + g->pool1); q.amax = g->save ? (unsigned char*)(ws + g->amax1) : nullptr;
    q.n = n; q.ldw = g->conv1.K;
    RGP_TRY(run_shallow_conv1(g->IH, q, s));
  } else {
    IgemmParams p = make_params(g->conv1, ws + g->frames4, ws, n);
    EpiParams e = make_epi(g->conv1, ws + g->pool1, ws);
    e.bias = g->b_conv1;
    if (g->save) e.argmax = (unsigned char*)(ws + g->amax1);
    RGP_TRY((launch_igemm<T, G32, 4, EpiStore<T, true, true>>(p, e, s)));
  }
  {
    IgemmParams p = make_params(g->conv2, ws + g->pool1, ws, n);
    EpiParams e = make_epi(g->conv2, ws + g->act2, ws);
    e.bias = g->b_conv2;
    RGP_TRY((launch_igemm<T, G32, 1, EpiStore<T, true, true>>(p, e, s)));
  }
  RGP_TRY(pool<T>((const T*)(ws + g->act2), (T*)(ws + g->pool2), n, g->c2, 64, 3, 2, (long long)g->p2 * g->p2 * 64, s,
                  g->save ? (unsigned char*)(ws + g->amax2) : nullptr));
  {
    IgemmParams p = make_params(g->conv3, ws + g->pool2, ws, n);
    EpiParams e = make_epi(g->conv3, ws + g->act3, ws);
    e.bias = g->b_conv3;
    RGP_TRY((launch_igemm<T, 1, 1, EpiStore<T, true, true>>(p, e, s)));
  }
  RGP_TRY(pool<T>((const T*)(ws + g->act3), (T*)(ws + g->pool3), n, g->c3, 32, 3, 2, g->Kf, s,
                  g->save ? (unsigned char*)(ws + g->amax3) : nullptr));
  {
    IgemmParams p = make_params(g->fc1, ws + g->pool3, ws, n);
    EpiParams e = make_epi(g->fc1, ws + g->mo1, ws);
    e.bias = (const float*)(ws + g->b1i);
    if (g->save) e.argmax = (unsigned char*)(ws + g->mask1);
    RGP_TRY((launch_igemm<T, 1, 1, EpiReluMaxout<T>>(p, e, s)));
  }
  {
    IgemmParams p = make_params(g->fc2, ws + g->mo1, ws, n);
    EpiParams e = make_epi(g->fc2, sal, ws);
    e.bias = (const float*)(ws + g->b2i);
    if (g->save) e.argmax = (unsigned char*)(ws + g->mask2);
    RGP_TRY((launch_igemm<T, 1, 1, EpiReluMaxout<float>>(p, e, s)));
  }
  if (sal7) {
    avgpool7_kernel<<<n, 64, 0, s>>>(sal, sal7);
    RGP_HIP(hipGetLastError());
  }
  return RGP_OK;
}

void conv_valid_desc(ConvDesc& d, int H_in, int k, int Cin, int Cout, int out_h, int dtype, bool& ok) {
  d.Mw = out_h * out_h; d.N = Cout;
  d.in_img_stride = (long long)H_in * H_in * Cin; d.out_img_stride = (long long)out_h * out_h * Cout;
  std::vector<int> tapoff, fidx;
  for (int y = 0; y < out_h; ++y) for (int x = 0; x < out_h; ++x) { d.in_tab.push_back((y * H_in + x) * Cin); d.out_tab.push_back((y * out_h + x) * Cout); }
  for (int ky = 0; ky < k; ++ky) for (int kx = 0; kx < k; ++kx) { tapoff.push_back((ky * H_in + kx) * Cin); fidx.push_back(ky * k + kx); }
  ok &= build_k_schedule(d, tapoff, fidx, Cin, dtype);
  d.s_tap = (long long)Cin * Cout; d.s_c = Cout; d.s_n = 1;     // HWIO
}

// ---------------------------------------------------------------- backward (FramewiseShallowNet trains all
// variables, gaze_framewise_shallownet.py:43-57; everywhere else the ShallowNet has learning rate 0)

// tf.nn.max_pool(3x3 / 2, SAME) + ReLU differentiated: the gradient of a pooled output goes to the first
// maximum of its window; windows overlap, so each input position gathers from the (up to 4) windows that cover it.
// x: the conv output after ReLU, dense [n][H][H][C]; dyp: fp32, image stride ld_img, element (oy*OH+ox)*C + c;
// out: gradient w.r.t. the conv pre-activation in a halo-padded image [n][H+2h][H+2h][C] (h = halo).
// A thread owns 8 channels of one input position; the forward pass recorded each window's first maximum (its place in
// the scan order, kernels_misc.hip.h maxpool_same_kernel), so a window costs one 8-byte and one 32-byte load.  (Round 2's
// version -- a thread per element that re-scanned every covering window, up to 36 two-byte loads each -- was 43 % of
// config 1's training step: 1.16 ms per launch at 512 frames.)
template <typename T>
__global__ __launch_bounds__(256) void maxpool_same_bwd_kernel(const T* __restrict__ x, const float* __restrict__ dyp,
                                                               long long ld_img, T* __restrict__ out, int n, int H, int C, int k,
                                                               int st, int OH, int pad, int halo, const unsigned char* __restrict__ amax) {
  const int CG = C / 8;
  const long long total = (long long)n * H * H * CG;
  const int Hp = H + 2 * halo;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % CG) * 8;
    const int xx = (int)((i / CG) % H);
    const int yy = (int)((i / ((long long)CG * H)) % H);
    const long long img = i / ((long long)CG * H * H);
    float v[8], acc[8];
    mp_load8(x + ((img * H + yy) * H + xx) * C + c, v);
    bool any = false;
#pragma unroll
    for (int q = 0; q < 8; ++q) { acc[q] = 0.f; any |= v[q] > 0.f; }
    if (any) {
      const int oy0 = max(0, (yy + pad - k + 1 + st - 1) / st), oy1 = min(OH - 1, (yy + pad) / st);
      const int ox0 = max(0, (xx + pad - k + 1 + st - 1) / st), ox1 = min(OH - 1, (xx + pad) / st);
      for (int oy = oy0; oy <= oy1; ++oy)
        for (int ox = ox0; ox <= ox1; ++ox) {
          const unsigned own = (unsigned)((yy - (oy * st - pad)) * k + (xx - (ox * st - pad)));   // this position's place in the window
          const long long w = ((img * OH + oy) * (long long)OH + ox) * C + c;
          const uint2 code8 = *(const uint2*)(amax + w);
          float d[8];
          mp_load8(dyp + img * ld_img + ((long long)oy * OH + ox) * C + c, d);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const unsigned code = ((q < 4 ? code8.x : code8.y) >> (8 * (q & 3))) & 0xffu;
            acc[q] += (code == own && v[q] > 0.f) ? d[q] : 0.f;
          }
        }
    }
    store8<T>(out + ((img * Hp + yy + halo) * Hp + xx + halo) * C + c, acc, 8);
  }
}

// conv1's fused 2x2 / 2 pool: route through the arg-max the forward epilogue recorded, gate by pool1 > 0;
// writes all four window members of dy1 (dense [n][c1][c1][32], behind 128 leading zeros).
template <typename T>
__global__ __launch_bounds__(256) void unpool2x2_kernel(const float* __restrict__ dyp, const unsigned char* __restrict__ amax,
                                                        const T* __restrict__ pooled, T* __restrict__ dy, int n, int PH, int C) {
  const int CG = C / 8;                                       // a thread owns 8 channels of a window: 16-byte loads / stores
  const long long total = (long long)n * PH * PH * CG;
  const int H = 2 * PH;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % CG) * 8;
    const int xo = (int)((i / CG) % PH);
    const int yo = (int)((i / ((long long)CG * PH)) % PH);
    const long long img = i / ((long long)CG * PH * PH);
    const long long e = ((img * PH + yo) * PH + xo) * C + c;
    float pv[8], g[8];
    mp_load8(pooled + e, pv);
    mp_load8(dyp + e, g);
    const uint2 code8 = *(const uint2*)(amax + e);
#pragma unroll
    for (int k = 0; k < 8; ++k) g[k] = pv[k] > 0.f ? g[k] : 0.f;
    for (int q = 0; q < 4; ++q) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = (((k < 4 ? code8.x : code8.y) >> (8 * (k & 3))) & 0xffu) == (unsigned)q ? g[k] : 0.f;
      store8<T>(dy + ((img * H + 2 * yo + (q >> 1)) * H + 2 * xo + (q & 1)) * C + c, v, 8);
    }
  }
}

// bias gradient of a conv: out[c] += sum over images and positions of an H x H window of a (possibly padded)
// channels-last image.  Thread = (row lane, 8-channel group): 16-byte reads, per-block LDS reduction, one atomic
// per channel and block (out is zeroed by the caller).
template <typename T>
__global__ __launch_bounds__(256) void conv_bias_grad_kernel(const T* __restrict__ base, long long img_stride, int H, int row_stride,
                                                             int C, long long rows_total, float* __restrict__ out) {
  __shared__ float red[256 * 8];
  const int CG = C / 8;
  const int cg = threadIdx.x % CG, rl = threadIdx.x / CG, RL = 256 / CG;
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (long long r = (long long)blockIdx.x * RL + rl; r < rows_total; r += (long long)gridDim.x * RL) {
    const int xx = (int)(r % H);
    const int yy = (int)((r / H) % H);
    const long long img = r / ((long long)H * H);
    const T* q = base + img * img_stride + (long long)yy * row_stride + (long long)xx * C + cg * 8;
    float v[8];
    mp_load8(q, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) bsum[k] += v[k];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[rl * C + cg * 8 + k] = bsum[k];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f;
    for (int r = 0; r < RL; ++r) a += red[r * C + c];
    if (a != 0.f) atomicAdd(out + c, a);
  }
}

// conv1 filter gradient: packed K order (ky tap: 8 px x 4 ch) -> HWIO [5,5,3,32]
__global__ void conv1_unpack_grad_kernel(const float* __restrict__ dw1, float* __restrict__ grad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 5 * 5 * 3 * 32) return;
  const int n = i % 32, c = (i / 32) % 3, kx = (i / 96) % 5, ky = i / 480;
  grad[i] = dw1[(long long)(ky * 32 + kx * 4 + c) * 32 + n];
}

bool bwd_plan(rgp_shallownet* g, Arena& a) {
  const int dtype = g->dtype, es = esize(dtype);
  const size_t n = g->N;
  bool ok = true;
  // "full" correlation of the halo-2 padded output gradient with the 180-degree rotated, in/out-swapped 3x3 filter
  auto dgrad3 = [&](ConvDesc& d, int out_h, int in_h, int Cout, int Cin) {
    const int Wp = out_h + 4;
    d.Mw = in_h * in_h; d.N = Cin;
    d.in_img_stride = (long long)Wp * Wp * Cout; d.out_img_stride = (long long)in_h * in_h * Cin;
    std::vector<int> tapoff, fidx;
    for (int y = 0; y < in_h; ++y) for (int x = 0; x < in_h; ++x) { d.in_tab.push_back((y * Wp + x) * Cout); d.out_tab.push_back((y * in_h + x) * Cin); }
    for (int t = 0; t < 9; ++t) { tapoff.push_back(((t / 3) * Wp + t % 3) * Cout); fidx.push_back(8 - t); }
    ok &= build_k_schedule(d, tapoff, fidx, Cout, dtype);
    d.s_tap = (long long)Cin * Cout; d.s_n = Cout; d.s_c = 1;          // HWIO [3,3,Cin,Cout]: row = ci, K column = co
  };
  dgrad3(g->b_c3, g->c3, g->p2, 32, 64);
  dgrad3(g->b_c2, g->c2, g->p1, 64, 32);
  auto fcT = [&](ConvDesc& d, int N, long long out_ld) {
    d.Mw = 1; d.N = N; d.in_img_stride = kFcN2; d.out_img_stride = out_ld; d.in_tab = {0}; d.out_tab = {0};
    ok &= build_k_schedule(d, {0}, {0}, kFcN2, dtype);
    d.cin_src = 4802; d.s_tap = 0; d.s_n = 4802; d.s_c = 1;
  };
  fcT(g->b_fc2, 2401, g->K2);
  fcT(g->b_fc1, g->nflat, g->Kf);
  if (!ok) return false;
  for (ConvDesc* d : {&g->b_fc2, &g->b_fc1, &g->b_c3, &g->b_c2}) d->reserve(a, dtype);
  g->amax1 = a.take(n * g->p1 * g->p1 * 32);
  g->amax2 = a.take(n * g->p2 * g->p2 * 64);
  g->amax3 = a.take(n * g->p3 * g->p3 * 32);
  g->mask1 = a.take(n * 2401);
  g->mask2 = a.take(n * 2401);
  g->dz2 = a.take((n + 1) * kFcN2 * es + 1024);
  g->dz1 = a.take((n + 1) * kFcN2 * es + 1024);
  g->dmo1 = a.take(n * g->K2 * 4);
  g->dpool3 = a.take(n * g->Kf * 4);
  g->dpool2 = a.take(n * g->p2 * g->p2 * 64 * 4);
  g->dpool1 = a.take(n * g->p1 * g->p1 * 32 * 4);
  g->dy3 = a.take(n * (g->c3 + 4) * (g->c3 + 4) * 32 * es + 4096);
  g->dy2 = a.take(n * (g->c2 + 4) * (g->c2 + 4) * 64 * es + 4096);
  g->dy1 = a.take((128 + n * g->c1 * g->c1 * 32) * es + 4096);
  g->dw1 = a.take((size_t)g->conv1.nk * bke(dtype) * 32 * 4);
  return true;
}

template <typename T>
int backward_impl(rgp_shallownet* g, int n, const float* d_sal, const rgp_shallownet_weights* gr, hipStream_t s) {
  char* ws = g->ws;
  constexpr int BKE = Elem<T>::BKE;
  constexpr int G32 = sizeof(T) == 2 ? 2 : 1;
  auto Tp = [&](size_t off) { return (T*)(ws + off); };
  auto Fp = [&](size_t off) { return (float*)(ws + off); };
  auto nblk = [](long long x) { return (int)std::min<long long>((x + 255) / 256, 8192); };
  WgradParams wp;
  // filter and bias gradients run on the plan's side stream (SideStream, rgp_host.h) beside the data-gradient chain; the last
  // layer's have nothing left to run beside and stay on s
  hipStream_t sw = s;
  auto fc_wgrad = [&](const void* X, int ldx, const ConvDesc& fwd, size_t dz, float* dW, int k_valid, hipStream_t s) -> int {
    RGP_HIP(hipMemsetAsync(dW, 0, (size_t)k_valid * 4802 * 4, s));
    memset(&wp, 0, sizeof(wp));
    wp.X = X; wp.dY = ws + dz; wp.dW = dW;
    wgrad_grid(wp, 1, 1, n);
    wp.x_sx = ldx; wp.y_sx = kFcN2; wp.y_org = kFcN2;
    wp.koff = (const int*)(ws + fwd.koff_off);
    wp.M = n; wp.N = 4802; wp.nk = fwd.nk; wp.ldw = 4802; wp.k_valid = k_valid;
    return launch_wgrad<T, 1>(wp, s);
  };
  // ---- fully connected read-out (saliency_shallownet.py:139-185)
  maxout_bwd_kernel<T><<<nblk((long long)n * 2401), 256, 0, s>>>(d_sal, 2401, nullptr, 1.0f, (const unsigned char*)(ws + g->mask2),
                                                                 Tp(g->dz2), (long long)n * 2401);
  RGP_TRY(g->side.fork(s, 0, &sw));
  fc_bias_grad_kernel<T><<<(4802 + 31) / 32, 256, 0, sw>>>(Tp(g->dz2), n, (float*)gr->fc2_b);
  RGP_HIP(hipGetLastError());
  RGP_TRY(fc_wgrad(ws + g->mo1, g->K2, g->fc2, g->dz2, (float*)gr->fc2_w, 2401, sw));
  {
    IgemmParams p = make_params(g->b_fc2, Tp(g->dz2) + kFcN2, ws, n);
    EpiParams e = make_epi(g->b_fc2, Fp(g->dmo1), ws);
    RGP_TRY((launch_igemm<T, 1, 1, EpiStore<float, false, false>>(p, e, s)));
  }
  maxout_bwd_kernel<T><<<nblk((long long)n * 2401), 256, 0, s>>>(Fp(g->dmo1), g->K2, nullptr, 1.0f, (const unsigned char*)(ws + g->mask1),
                                                                 Tp(g->dz1), (long long)n * 2401);
  RGP_TRY(g->side.fork(s, 1, &sw));
  fc_bias_grad_kernel<T><<<(4802 + 31) / 32, 256, 0, sw>>>(Tp(g->dz1), n, (float*)gr->fc1_b);
  RGP_HIP(hipGetLastError());
  RGP_TRY(fc_wgrad(ws + g->pool3, g->Kf, g->fc1, g->dz1, (float*)gr->fc1_w, g->nflat, sw));
  {
    IgemmParams p = make_params(g->b_fc1, Tp(g->dz1) + kFcN2, ws, n);
    EpiParams e = make_epi(g->b_fc1, Fp(g->dpool3), ws);
    RGP_TRY((launch_igemm<T, 1, 1, EpiStore<float, false, false>>(p, e, s)));
  }
  // ---- conv3 (3x3 VALID 64 -> 32) behind pool3
  auto conv_wgrad = [&](const void* X, int in_h, int Cin, const ConvDesc& fwd, size_t dy, int out_h, int Cout, float* dW, int G,
                        hipStream_t s) -> int {
    const int Wp = out_h + 4;
    RGP_HIP(hipMemsetAsync(dW, 0, (size_t)9 * Cin * Cout * 4, s));
    memset(&wp, 0, sizeof(wp));
    wp.X = X; wp.dY = ws + dy; wp.dW = dW;
    wgrad_grid(wp, 1, out_h, out_h);
    wp.x_sx = Cin; wp.x_sy = in_h * Cin; wp.x_img_stride = (long long)in_h * in_h * Cin;
    wp.y_sx = Cout; wp.y_sy = Wp * Cout; wp.y_org = (2 * Wp + 2) * Cout; wp.y_img_stride = (long long)Wp * Wp * Cout;
    wp.koff = (const int*)(ws + fwd.koff_off);
    wp.M = (long long)n * out_h * out_h; wp.N = Cout; wp.nk = fwd.nk; wp.ldw = Cout; wp.k_valid = 9 * Cin;
    return G == 1 ? launch_wgrad<T, 1>(wp, s) : launch_wgrad<T, G32>(wp, s);
  };
  {
    const int H = g->c3, OH = g->p3, pad = std::max((OH - 1) * 2 + 3 - H, 0) / 2, Wp = H + 4;
    maxpool_same_bwd_kernel<T><<<nblk((long long)n * H * H * 4), 256, 0, s>>>(Tp(g->act3), Fp(g->dpool3), g->Kf, Tp(g->dy3), n, H, 32, 3, 2,
                                                                              OH, pad, 2, (const unsigned char*)(ws + g->amax3));
    RGP_TRY(g->side.fork(s, 2, &sw));
    RGP_HIP(hipMemsetAsync((void*)gr->conv3_b, 0, 32 * 4, sw));
    conv_bias_grad_kernel<T><<<std::min(nblk((long long)n * H * H * 2), 1024), 256, 0, sw>>>(Tp(g->dy3) + (2 * Wp + 2) * 32, (long long)Wp * Wp * 32, H, Wp * 32,
                                                                           32, (long long)n * H * H, (float*)gr->conv3_b);
    RGP_HIP(hipGetLastError());
    RGP_TRY(conv_wgrad(ws + g->pool2, g->p2, 64, g->conv3, g->dy3, H, 32, (float*)gr->conv3_w, 1, sw));
    IgemmParams p = make_params(g->b_c3, Tp(g->dy3), ws, n);
    EpiParams e = make_epi(g->b_c3, Fp(g->dpool2), ws);
    RGP_TRY((launch_igemm<T, G32, 1, EpiStore<float, false, false>>(p, e, s)));
  }
  // ---- conv2 (3x3 VALID 32 -> 64) behind pool2
  {
    const int H = g->c2, OH = g->p2, pad = std::max((OH - 1) * 2 + 3 - H, 0) / 2, Wp = H + 4;
    maxpool_same_bwd_kernel<T><<<nblk((long long)n * H * H * 8), 256, 0, s>>>(Tp(g->act2), Fp(g->dpool2), (long long)OH * OH * 64, Tp(g->dy2),
                                                                              n, H, 64, 3, 2, OH, pad, 2, (const unsigned char*)(ws + g->amax2));
    RGP_TRY(g->side.fork(s, 3, &sw));
    RGP_HIP(hipMemsetAsync((void*)gr->conv2_b, 0, 64 * 4, sw));
    conv_bias_grad_kernel<T><<<std::min(nblk((long long)n * H * H * 4), 1024), 256, 0, sw>>>(Tp(g->dy2) + (2 * Wp + 2) * 64, (long long)Wp * Wp * 64, H, Wp * 64,
                                                                           64, (long long)n * H * H, (float*)gr->conv2_b);
    RGP_HIP(hipGetLastError());
    RGP_TRY(conv_wgrad(ws + g->pool1, g->p1, 32, g->conv2, g->dy2, H, 64, (float*)gr->conv2_w, G32, sw));
    IgemmParams p = make_params(g->b_c2, Tp(g->dy2), ws, n);
    EpiParams e = make_epi(g->b_c2, Fp(g->dpool1), ws);
    RGP_TRY((launch_igemm<T, 1, 1, EpiStore<float, false, false>>(p, e, s)));
  }
  // ---- conv1 (5x5 VALID 3 -> 32) with its fused 2x2 pool; no input gradient (the input is the image)
  {
    const int H = g->c1, IH = g->IH;
    T* dy1 = Tp(g->dy1) + 128;
    unpool2x2_kernel<T><<<nblk((long long)n * g->p1 * g->p1 * 4), 256, 0, s>>>(Fp(g->dpool1), (const unsigned char*)(ws + g->amax1),
                                                                               Tp(g->pool1), dy1, n, g->p1, 32);
    RGP_HIP(hipMemsetAsync((void*)gr->conv1_b, 0, 32 * 4, s));
    conv_bias_grad_kernel<T><<<std::min(nblk((long long)n * H * H * 2), 1024), 256, 0, s>>>(dy1, (long long)H * H * 32, H, H * 32, 32, (long long)n * H * H,
                                                                           (float*)gr->conv1_b);
    RGP_HIP(hipGetLastError());
    RGP_HIP(hipMemsetAsync(Fp(g->dw1), 0, (size_t)g->conv1.nk * BKE * 32 * 4, s));
    memset(&wp, 0, sizeof(wp));
    wp.X = ws + g->frames4; wp.dY = Tp(g->dy1); wp.dW = Fp(g->dw1);
    wgrad_grid(wp, 1, H, H);
    wp.x_sx = 4; wp.x_sy = IH * 4; wp.x_img_stride = (long long)IH * IH * 4;
    wp.y_sx = 32; wp.y_sy = H * 32; wp.y_org = 128; wp.y_img_stride = (long long)H * H * 32;
    wp.koff = (const int*)(ws + g->conv1.koff_off);
    wp.M = (long long)n * H * H; wp.N = 32; wp.nk = g->conv1.nk; wp.ldw = 32; wp.k_valid = g->conv1.nk * BKE;
    RGP_TRY((launch_wgrad<T, G32>(wp, s)));
    conv1_unpack_grad_kernel<<<(5 * 5 * 3 * 32 + 255) / 256, 256, 0, s>>>(Fp(g->dw1), (float*)gr->conv1_w);
    RGP_HIP(hipGetLastError());
  }
  if (sw != s) RGP_TRY(g->side.join(s));
  return RGP_OK;
}

}  // namespace

extern "C" {

int rgp_shallownet_backward(rgp_shallownet_t* g, int n_frames, const float* d_saliency, const rgp_shallownet_weights* grads,
                            rgp_stream_t stream) {
  RGP_REQUIRE(g && d_saliency && grads, "rgp_shallownet_backward: null argument");
  if (!g->save) return set_err(RGP_ESTATE, "rgp_shallownet_backward: plan was created without save_for_backward");
  if (!g->ws || !g->weights_set) return set_err(RGP_ESTATE, "rgp_shallownet_backward: workspace/weights not set");
  if (n_frames != g->last_n || n_frames <= 0)
    return set_err(RGP_ESTATE, "rgp_shallownet_backward: n_frames %d != frames of the last forward (%d)", n_frames, g->last_n);
  const float* const* ptrs = (const float* const*)grads;
  for (size_t i = 0; i < sizeof(rgp_shallownet_weights) / sizeof(float*); ++i)
    RGP_REQUIRE(ptrs[i], "rgp_shallownet_backward: gradient pointer %zu is null", i);
  hipStream_t s = (hipStream_t)stream;
  return g->dtype == RGP_BF16 ? backward_impl<bf16_t>(g, n_frames, d_saliency, grads, s)
                              : backward_impl<float>(g, n_frames, d_saliency, grads, s);
}

int rgp_shallownet_create(rgp_shallownet_t** plan, int max_frames, int image_hw, int dtype) {
  return rgp_shallownet_create_ex(plan, max_frames, image_hw, dtype, 0);
}

int rgp_shallownet_create_ex(rgp_shallownet_t** plan, int max_frames, int image_hw, int dtype, int save_for_backward) {
  RGP_REQUIRE(plan && max_frames > 0, "rgp_shallownet_create: bad arguments");
  RGP_REQUIRE(image_hw == 98 || image_hw == 112, "rgp_shallownet_create: image %d (98 or 112)", image_hw);
  RGP_REQUIRE(dtype == RGP_F32 || dtype == RGP_BF16, "rgp_shallownet_create: dtype %d", dtype);
  rgp_shallownet* g = new rgp_shallownet();
  g->N = max_frames; g->IH = image_hw; g->dtype = dtype;
  g->save = save_for_backward != 0;
  g->c1 = image_hw - 4; g->p1 = g->c1 / 2; g->c2 = g->p1 - 2; g->p2 = (g->c2 + 1) / 2; g->c3 = g->p2 - 2; g->p3 = (g->c3 + 1) / 2;
  g->nflat = g->p3 * g->p3 * 32;
  g->Kf = (int)align_up(g->nflat, 64);
  g->K2 = (int)align_up(2401, 64);
  const int es = esize(dtype), IH = g->IH;
  bool ok = true;
  {  // conv1 5x5 VALID on [IH,IH,4]; one 32-element "tap" per ky (8 px x 4 ch, kx >= 5 zero); pool 2x2 fused
    ConvDesc& d = g->conv1;
    d.Mw = g->c1 * g->c1; d.N = 32; d.P = 4;
    d.in_img_stride = (long long)IH * IH * 4; d.out_img_stride = (long long)g->p1 * g->p1 * 32;
    for (int yo = 0; yo < g->p1; ++yo) for (int xo = 0; xo < g->p1; ++xo) {
      for (int dy = 0; dy < 2; ++dy) for (int dx = 0; dx < 2; ++dx) d.in_tab.push_back(((2 * yo + dy) * IH + 2 * xo + dx) * 4);
      d.out_tab.push_back((yo * g->p1 + xo) * 32);
    }
    std::vector<int> tapoff, fidx;
    for (int ky = 0; ky < 5; ++ky) { tapoff.push_back(ky * IH * 4); fidx.push_back(ky); }
    ok &= build_k_schedule(d, tapoff, fidx, 32, dtype);
    const int nt = d.pack_taps;
    std::vector<int> ts;
    for (int t = 0; t < nt; ++t) for (int kx = 0; kx < 8; ++kx) ts.push_back((t < 5 && kx < 5) ? t * 5 + kx : -1);
    d.tap_src = ts; d.pack_taps = nt * 8; d.cin_k = 4; d.cin_src = 3;
    d.s_tap = 3LL * 32; d.s_c = 32; d.s_n = 1;
  }
  conv_valid_desc(g->conv2, g->p1, 3, 32, 64, g->c2, dtype, ok);
  conv_valid_desc(g->conv3, g->p2, 3, 64, 32, g->c3, dtype, ok);
  auto fc = [&](ConvDesc& d, int K, long long ldc) {
    d.Mw = 1; d.N = 4802; d.in_img_stride = K; d.out_img_stride = ldc;
    d.in_tab = {0}; d.out_tab = {0};
    ok &= build_k_schedule(d, {0}, {0}, K, dtype);
    d.s_tap = 0; d.s_n = 1; d.s_c = 4802;
  };
  fc(g->fc1, g->Kf, g->K2);
  g->fc1.cin_src = g->nflat;
  fc(g->fc2, g->K2, 2401);
  g->fc2.cin_src = 2401;
  if (!ok) { delete g; return set_err(RGP_EINVAL, "rgp_shallownet_create: K schedule failed"); }
  Arena a;
  for (ConvDesc* d : {&g->conv1, &g->conv2, &g->conv3, &g->fc1, &g->fc2}) d->reserve(a, dtype);
  const size_t n = max_frames;
  g->frames4 = a.take(n * IH * IH * 4 * es + 4096);          // slack: conv1's 8-pixel runs overrun a row end
  g->pool1 = a.take(n * g->p1 * g->p1 * 32 * es + 4096);
  g->act2 = a.take(n * g->c2 * g->c2 * 64 * es);
  g->pool2 = a.take(n * g->p2 * g->p2 * 64 * es + 4096);
  g->act3 = a.take(n * g->c3 * g->c3 * 32 * es);
  g->pool3 = a.take(n * g->Kf * es);
  g->mo1 = a.take(n * g->K2 * es);
  g->b1i = a.take(4802 * 4 + 64);
  g->b2i = a.take(4802 * 4 + 64);
  if (g->save && !bwd_plan(g, a)) { delete g; return set_err(RGP_EINVAL, "rgp_shallownet_create: backward K schedule failed"); }
  g->ws_bytes = a.off;
  *plan = g;
  return RGP_OK;
}

int rgp_shallownet_destroy(rgp_shallownet_t* plan) {
  delete plan;
  return RGP_OK;
}

size_t rgp_shallownet_workspace_bytes(const rgp_shallownet_t* plan) { return plan ? plan->ws_bytes : 0; }

int rgp_shallownet_bind_workspace(rgp_shallownet_t* g, void* workspace, size_t bytes, rgp_stream_t stream) {
  RGP_REQUIRE(g && workspace, "rgp_shallownet_bind_workspace: null argument");
  if (bytes < g->ws_bytes) return set_err(RGP_EWORKSPACE, "workspace %zu < required %zu bytes", bytes, g->ws_bytes);
  RGP_REQUIRE(((size_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  g->ws = (char*)workspace;
  g->weights_set = false;
  RGP_HIP(hipMemsetAsync(g->ws, 0, g->ws_bytes, s));
  for (ConvDesc* d : {&g->conv1, &g->conv2, &g->conv3, &g->fc1, &g->fc2}) RGP_TRY(upload_desc(*d, g->ws, s));
  if (g->save) for (ConvDesc* d : {&g->b_fc2, &g->b_fc1, &g->b_c3, &g->b_c2}) RGP_TRY(upload_desc(*d, g->ws, s));
  return RGP_OK;
}

int rgp_shallownet_set_weights(rgp_shallownet_t* g, const rgp_shallownet_weights* w, rgp_stream_t stream) {
  RGP_REQUIRE(g && w, "rgp_shallownet_set_weights: null argument");
  if (!g->ws) return set_err(RGP_EWORKSPACE, "rgp_shallownet: workspace not bound");
  const float* const* ptrs = (const float* const*)w;
  for (size_t i = 0; i < sizeof(rgp_shallownet_weights) / sizeof(float*); ++i)
    RGP_REQUIRE(ptrs[i], "rgp_shallownet_set_weights: weight pointer %zu is null", i);
  hipStream_t s = (hipStream_t)stream;
  return g->dtype == RGP_BF16 ? set_weights_impl<bf16_t>(g, w, s) : set_weights_impl<float>(g, w, s);
}

int rgp_shallownet_forward(rgp_shallownet_t* g, const float* frames, int n_frames, float* saliency, float* saliency7,
                           rgp_stream_t stream) {
  RGP_REQUIRE(g && frames && saliency && n_frames > 0 && n_frames <= g->N, "rgp_shallownet_forward: bad arguments");
  if (!g->ws) return set_err(RGP_EWORKSPACE, "rgp_shallownet: workspace not bound");
  if (!g->weights_set) return set_err(RGP_ESTATE, "rgp_shallownet: weights not set");
  hipStream_t s = (hipStream_t)stream;
  return g->dtype == RGP_BF16 ? forward_impl<bf16_t>(g, frames, n_frames, saliency, saliency7, s)
                              : forward_impl<float>(g, frames, n_frames, saliency, saliency7, s);
}

}  // extern "C"
