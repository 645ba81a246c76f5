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

#include "rgp_host.h"

using namespace rgp;

struct rgp_shallownet {
  int N = 0, IH = 0, dtype = RGP_F32;
  int c1 = 0, p1 = 0, c2 = 0, p2 = 0, c3 = 0, p3 = 0, nflat = 0, Kf = 0, K2 = 0;
  ConvDesc conv1, conv2, conv3, fc1, fc2;
  size_t frames4 = 0, pool1 = 0, act2 = 0, pool2 = 0, act3 = 0, pool3 = 0, mo1 = 0, b1i = 0, b2i = 0;
  size_t ws_bytes = 0;
  char* ws = nullptr;
  bool weights_set = false;
  const float *b_conv1 = nullptr, *b_conv2 = nullptr, *b_conv3 = nullptr;
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
  for (ConvDesc* d : {&g->conv1, &g->conv2, &g->conv3, &g->fc1, &g->fc2}) RGP_HIP(hipMemsetAsync(ws + d->w_off, 0, d->w_bytes(g->dtype), s));
  RGP_TRY(pack_filter<T>(g->conv1, w->conv1_w, ws, 32, 0, s));
  RGP_TRY(pack_filter<T>(g->conv2, w->conv2_w, ws, 64, 0, s));
  RGP_TRY(pack_filter<T>(g->conv3, w->conv3_w, ws, 32, 0, s));
  // FC filters [K][4802], halves interleaved: packed row 2j = unit j, 2j+1 = unit j+2401
  RGP_TRY(pack_filter<T>(g->fc1, w->fc1_w, ws, 2401, 0, s, 0, 0, 2));
  RGP_TRY(pack_filter<T>(g->fc1, w->fc1_w + 2401, ws, 2401, 1, s, 0, 0, 2));
  RGP_TRY(pack_filter<T>(g->fc2, w->fc2_w, ws, 2401, 0, s, 0, 0, 2));
  RGP_TRY(pack_filter<T>(g->fc2, w->fc2_w + 2401, ws, 2401, 1, s, 0, 0, 2));
  interleave_kernel<<<(2401 + 255) / 256, 256, 0, s>>>(w->fc1_b, (float*)(ws + g->b1i), 2401);
  interleave_kernel<<<(2401 + 255) / 256, 256, 0, s>>>(w->fc2_b, (float*)(ws + g->b2i), 2401);
  RGP_HIP(hipGetLastError());
  g->b_conv1 = w->conv1_b; g->b_conv2 = w->conv2_b; g->b_conv3 = w->conv3_b;
  g->weights_set = true;
  return RGP_OK;
}

template <typename T>
int pool(const T* src, T* dst, int N, int H, int C, int k, int st, long long ld_out, hipStream_t s) {
  const int OH = (H + st - 1) / st;
  const int pad = std::max((OH - 1) * st + k - H, 0) / 2;
  const long long total = (long long)N * OH * OH * C;
  maxpool_same_kernel<T><<<(int)std::min<long long>((total + 255) / 256, 8192), 256, 0, s>>>(src, dst, N, H, H, C, k, st, OH, OH,
                                                                                      pad, pad, ld_out);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

template <typename T>
int forward_impl(rgp_shallownet* g, const float* frames, int n, float* sal, float* sal7, hipStream_t s) {
  char* ws = g->ws;
  constexpr int G32 = sizeof(T) == 2 ? 2 : 1;     // 32-element taps: two per 128-byte chunk in bf16
  const long long npix = (long long)n * g->IH * g->IH;
  frame_prep_kernel<T><<<(int)std::min<long long>((npix + 255) / 256, 8192), 256, 0, s>>>(frames, (T*)(ws + g->frames4), npix);
  RGP_HIP(hipGetLastError());
  {
    IgemmParams p = make_params(g->conv1, ws + g->frames4, ws, n);
    EpiParams e = make_epi(g->conv1, ws + g->pool1, ws);
    e.bias = g->b_conv1;
    RGP_TRY((launch_igemm<T, G32, 4, EpiStore<T, true, true>>(p, e, s)));
  }
  {
    IgemmParams p = make_params(g->conv2, ws + g->pool1, ws, n);
    EpiParams e = make_epi(g->conv2, ws + g->act2, ws);
    e.bias = g->b_conv2;
    RGP_TRY((launch_igemm<T, G32, 1, EpiStore<T, true, true>>(p, e, s)));
  }
  RGP_TRY(pool<T>((const T*)(ws + g->act2), (T*)(ws + g->pool2), n, g->c2, 64, 3, 2, (long long)g->p2 * g->p2 * 64, s));
  {
    IgemmParams p = make_params(g->conv3, ws + g->pool2, ws, n);
    EpiParams e = make_epi(g->conv3, ws + g->act3, ws);
    e.bias = g->b_conv3;
    RGP_TRY((launch_igemm<T, 1, 1, EpiStore<T, true, true>>(p, e, s)));
  }
  RGP_TRY(pool<T>((const T*)(ws + g->act3), (T*)(ws + g->pool3), n, g->c3, 32, 3, 2, g->Kf, s));
  {
    IgemmParams p = make_params(g->fc1, ws + g->pool3, ws, n);
    EpiParams e = make_epi(g->fc1, ws + g->mo1, ws);
    e.bias = (const float*)(ws + g->b1i);
    RGP_TRY((launch_igemm<T, 1, 1, EpiReluMaxout<T>>(p, e, s)));
  }
  {
    IgemmParams p = make_params(g->fc2, ws + g->mo1, ws, n);
    EpiParams e = make_epi(g->fc2, sal, ws);
    e.bias = (const float*)(ws + g->b2i);
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

}  // namespace

extern "C" {

int rgp_shallownet_create(rgp_shallownet_t** plan, int max_frames, int image_hw, int dtype) {
  RGP_REQUIRE(plan && max_frames > 0, "rgp_shallownet_create: bad arguments");
  RGP_REQUIRE(image_hw == 98 || image_hw == 112, "rgp_shallownet_create: image %d (98 or 112)", image_hw);
  RGP_REQUIRE(dtype == RGP_F32 || dtype == RGP_BF16, "rgp_shallownet_create: dtype %d", dtype);
  rgp_shallownet* g = new rgp_shallownet();
  g->N = max_frames; g->IH = image_hw; g->dtype = dtype;
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
