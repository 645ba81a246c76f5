// librgp_hip.so: C3D conv1a..conv5b feature stack as fused implicit-GEMM launches
// (3x3x3 conv pad 1 + bias + ReLU + max-pool in one kernel per layer).
// Spec: /root/reference/C3D/.../c3d_prototxt/feature_extration.prototxt:22-342
// (the reference runs it as an offline Caffe binary, extract_C3D_features.py:689-724).
#include <algorithm>

#include "rgp_host.h"
#include "conv1a.hip.h"

using namespace rgp;

namespace {
struct LayerSpec {
  int cin, cout, D, H, pd, ph;  // input extent (W == H), pooling window (depth, spatial); 1 = none
};
const LayerSpec kLayers[8] = {
    {3, 64, 16, 112, 1, 2},  {64, 128, 16, 56, 2, 2}, {128, 256, 8, 28, 1, 1}, {256, 256, 8, 28, 2, 2},
    {256, 512, 4, 14, 1, 1}, {512, 512, 4, 14, 2, 2}, {512, 512, 2, 7, 1, 1},  {512, 512, 2, 7, 1, 1},
};
}  // namespace

struct rgp_c3d {
  int max_windows = 0, dtype = RGP_BF16;
  ConvDesc L[8];
  size_t act_off[9] = {0};       // act[i] = halo-padded input of layer i; act[8] = conv5b rows
  long long act_stride[9] = {0}; // elements per window
  std::vector<int> unpad_tab[8];
  size_t unpad_off[8] = {0};
  size_t ws_bytes = 0;
  char* ws = nullptr;
  bool weights_set = false;
  const float* bias[8] = {nullptr};
  StageProfiler prof;
};

namespace {

template <typename T, int G, int P>
int run_layer(rgp_c3d* c, int i, int n, hipStream_t s) {
  IgemmParams p = make_params(c->L[i], c->ws + c->act_off[i], c->ws, n);
  EpiParams e = make_epi(c->L[i], c->ws + c->act_off[i + 1], c->ws);
  e.bias = c->bias[i];
  return launch_igemm<T, G, P, EpiStore<T, true, true>>(p, e, s);
}

// bf16 conv1a: dedicated register-resident-filter kernel (conv1a.hip.h)
int run_conv1a_bf16(rgp_c3d* c, int n, hipStream_t s) {
  Conv1aParams p;
  p.in = (const bf16_t*)(c->ws + c->act_off[0]);
  p.wp = (const bf16_t*)(c->ws + c->L[0].w_off);
  p.bias = c->bias[0];
  p.out = (bf16_t*)(c->ws + c->act_off[1]);
  p.n_windows = n;
  const long long tiles = (long long)n * C1_TILES_PER_WINDOW;
  const int grid = (int)std::min<long long>((tiles + 3) / 4, 1024);
  conv1a_pool_bf16_kernel<<<grid, 256, 0, s>>>(p);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

template <typename T>
int layer_dispatch(rgp_c3d* c, int i, int n, hipStream_t s) {
  constexpr int G0 = sizeof(T) == 2 ? 4 : 2;
  if (i == 0 && sizeof(T) == 2) return run_conv1a_bf16(c, n, s);
  switch (i) {
    case 0: return run_layer<T, G0, 4>(c, i, n, s);
    case 1: case 3: case 5: return run_layer<T, 1, 8>(c, i, n, s);
    default: return run_layer<T, 1, 1>(c, i, n, s);
  }
}

template <typename T>
int forward_chunk(rgp_c3d* c, const float* video, int n, float* features, void* rows, hipStream_t s) {
  const long long npix = (long long)n * 16 * 112 * 112;
  int pid = c->prof.begin(8, s);
  video_prep_kernel<T><<<(int)std::min<long long>((npix + 255) / 256, 65536), 256, 0, s>>>(
      video, (T*)(c->ws + c->act_off[0]), npix, 16, 112, 112);
  RGP_HIP(hipGetLastError());
  c->prof.end(pid, s);
  for (int i = 0; i < 8; ++i) {
    pid = c->prof.begin(i, s);
    RGP_TRY(layer_dispatch<T>(c, i, n, s));
    c->prof.end(pid, s);
  }
  const T* r = (const T*)(c->ws + c->act_off[8]);
  if (rows) RGP_HIP(hipMemcpyAsync(rows, r, (size_t)n * 49 * 1024 * sizeof(T), hipMemcpyDeviceToDevice, s));
  if (features) {
    const long long total = (long long)n * 1024 * 49;
    rows_to_c3d_features_kernel<T><<<(int)std::min<long long>((total + 255) / 256, 8192), 256, 0, s>>>(r, features, total);
    RGP_HIP(hipGetLastError());
  }
  return RGP_OK;
}

template <typename T>
int set_weights_impl(rgp_c3d* c, const rgp_c3d_weights* w, hipStream_t s) {
  for (int i = 0; i < 8; ++i) {
    RGP_HIP(hipMemsetAsync(c->ws + c->L[i].w_off, 0, c->L[i].w_bytes(c->dtype), s));
    RGP_TRY(pack_filter<T>(c->L[i], w->w[i], c->ws, kLayers[i].cout, 0, s));
    c->bias[i] = w->b[i];
  }
  c->weights_set = true;
  return RGP_OK;
}

}  // namespace

extern "C" {

int rgp_c3d_create(rgp_c3d_t** plan, int max_windows, int dtype) {
  RGP_REQUIRE(plan && max_windows > 0, "rgp_c3d_create: bad arguments");
  RGP_REQUIRE(dtype == RGP_F32 || dtype == RGP_BF16, "rgp_c3d_create: dtype %d", dtype);
  RGP_REQUIRE((long long)max_windows * 16 * 112 * 112 < (1LL << 31), "rgp_c3d_create: max_windows too large");
  rgp_c3d* c = new rgp_c3d();
  c->max_windows = max_windows;
  c->dtype = dtype;
  bool ok = true;
  Arena a;
  for (int i = 0; i < 8; ++i) {
    const LayerSpec& l = kLayers[i];
    ConvDesc& d = c->L[i];
    const int D = l.D, H = l.H, W = l.H;
    const int C = i == 0 ? 4 : l.cin;                 // conv1a: channels padded 3 -> 4
    const int Hp = H + 2, Wp = i == 0 ? W + 4 : W + 2;  // conv1a: x halo 1 left, 3 right
    c->act_stride[i] = (long long)(D + 2) * Hp * Wp * C;
    const int Do = D / l.pd, Ho = H / l.ph, Wo = W / l.ph;
    d.Mw = D * H * W;
    d.N = l.cout;
    d.P = l.pd * l.ph * l.ph;
    d.in_img_stride = c->act_stride[i];
    // rows ordered pooling-window-major so the P rows of a window are consecutive
    for (int zo = 0; zo < Do; ++zo) for (int yo = 0; yo < Ho; ++yo) for (int xo = 0; xo < Wo; ++xo)
      for (int dz = 0; dz < l.pd; ++dz) for (int dy = 0; dy < l.ph; ++dy) for (int dx = 0; dx < l.ph; ++dx) {
        const int z = zo * l.pd + dz, y = yo * l.ph + dy, x = xo * l.ph + dx;
        d.in_tab.push_back(((z * Hp + y) * Wp + x) * C);
      }
    if (i < 7) {
      d.out_img_stride = (long long)(Do + 2) * (Ho + 2) * (Wo + 2) * l.cout;
      for (int zo = 0; zo < Do; ++zo) for (int yo = 0; yo < Ho; ++yo) for (int xo = 0; xo < Wo; ++xo)
        d.out_tab.push_back((((zo + 1) * (Ho + 2) + yo + 1) * (Wo + 2) + xo + 1) * l.cout);
    } else {  // conv5b -> rows [49][d*512 + c]
      d.out_img_stride = 49LL * 1024;
      for (int zo = 0; zo < Do; ++zo) for (int yo = 0; yo < Ho; ++yo) for (int xo = 0; xo < Wo; ++xo)
        d.out_tab.push_back((yo * 7 + xo) * 1024 + zo * 512);
    }
    c->unpad_tab[i] = d.out_tab;
    std::vector<int> tapoff, fidx;
    if (i == 0) {
      // one "tap" per (kz,ky): the 3 kx taps x 4 channels (+1 zero pixel) are 16 contiguous elements
      for (int kz = 0; kz < 3; ++kz) for (int ky = 0; ky < 3; ++ky) { tapoff.push_back(((kz * Hp + ky) * Wp) * 4); fidx.push_back(kz * 3 + ky); }
      ok &= build_k_schedule(d, tapoff, fidx, 16, dtype);
      const int nt = d.pack_taps;               // (kz,ky) taps incl. zero padding
      std::vector<int> ts;
      for (int t = 0; t < nt; ++t) for (int kx = 0; kx < 4; ++kx) ts.push_back((t < 9 && kx < 3) ? t * 3 + kx : -1);
      d.tap_src = ts;
      d.pack_taps = nt * 4; d.cin_k = 4; d.cin_src = 3;
      d.s_tap = 3LL * l.cout; d.s_c = l.cout; d.s_n = 1;
      if (dtype == RGP_BF16) {   // dedicated kernel: K = 10 (kz,ky) taps x 16, filter [64][160]
        d.tap_src.resize(40);
        d.pack_taps = 40;
        d.K = C1_K;
      }
    } else {
      for (int kz = 0; kz < 3; ++kz) for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
        tapoff.push_back(((kz * Hp + ky) * Wp + kx) * C);
        fidx.push_back((kz * 3 + ky) * 3 + kx);
      }
      ok &= build_k_schedule(d, tapoff, fidx, C, dtype);
      d.s_tap = (long long)l.cin * l.cout; d.s_c = l.cout; d.s_n = 1;   // DHWIO
    }
    d.reserve(a, dtype);
    c->unpad_off[i] = a.take(c->unpad_tab[i].size() * 4);
  }
  c->act_stride[8] = 49LL * 1024;
  if (!ok) { delete c; return set_err(RGP_EINVAL, "rgp_c3d_create: K schedule failed"); }
  for (int i = 0; i < 9; ++i) c->act_off[i] = a.take((size_t)max_windows * c->act_stride[i] * esize(dtype));
  c->ws_bytes = a.off;
  *plan = c;
  return RGP_OK;
}

int rgp_c3d_destroy(rgp_c3d_t* plan) {
  delete plan;
  return RGP_OK;
}

size_t rgp_c3d_workspace_bytes(const rgp_c3d_t* plan) { return plan ? plan->ws_bytes : 0; }

int rgp_c3d_bind_workspace(rgp_c3d_t* c, void* workspace, size_t bytes, rgp_stream_t stream) {
  RGP_REQUIRE(c && workspace, "rgp_c3d_bind_workspace: null argument");
  if (bytes < c->ws_bytes) return set_err(RGP_EWORKSPACE, "workspace %zu < required %zu bytes", bytes, c->ws_bytes);
  RGP_REQUIRE(((size_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  c->ws = (char*)workspace;
  c->weights_set = false;
  RGP_HIP(hipMemsetAsync(c->ws, 0, c->ws_bytes, s));   // halos stay zero afterwards
  for (int i = 0; i < 8; ++i) {
    RGP_TRY(upload_desc(c->L[i], c->ws, s));
    RGP_HIP(hipMemcpyAsync(c->ws + c->unpad_off[i], c->unpad_tab[i].data(), c->unpad_tab[i].size() * 4, hipMemcpyHostToDevice, s));
  }
  return RGP_OK;
}

int rgp_c3d_set_weights(rgp_c3d_t* c, const rgp_c3d_weights* w, rgp_stream_t stream) {
  RGP_REQUIRE(c && w, "rgp_c3d_set_weights: null argument");
  if (!c->ws) return set_err(RGP_EWORKSPACE, "rgp_c3d: workspace not bound");
  for (int i = 0; i < 8; ++i) RGP_REQUIRE(w->w[i] && w->b[i], "rgp_c3d_set_weights: layer %d pointer is null", i);
  hipStream_t s = (hipStream_t)stream;
  return c->dtype == RGP_BF16 ? set_weights_impl<bf16_t>(c, w, s) : set_weights_impl<float>(c, w, s);
}

int rgp_c3d_forward(rgp_c3d_t* c, const float* video, int n_windows, float* features, void* rows, rgp_stream_t stream) {
  RGP_REQUIRE(c && video && n_windows > 0, "rgp_c3d_forward: bad arguments");
  if (!c->ws) return set_err(RGP_EWORKSPACE, "rgp_c3d: workspace not bound");
  if (!c->weights_set) return set_err(RGP_ESTATE, "rgp_c3d: weights not set");
  hipStream_t s = (hipStream_t)stream;
  const size_t es = esize(c->dtype);
  for (int w0 = 0; w0 < n_windows; w0 += c->max_windows) {
    const int n = std::min(c->max_windows, n_windows - w0);
    const float* v = video + (size_t)w0 * 16 * 112 * 112 * 3;
    float* f = features ? features + (size_t)w0 * 1024 * 49 : nullptr;
    void* r = rows ? (char*)rows + (size_t)w0 * 49 * 1024 * es : nullptr;
    RGP_TRY(c->dtype == RGP_BF16 ? forward_chunk<bf16_t>(c, v, n, f, r, s) : forward_chunk<float>(c, v, n, f, r, s));
  }
  return RGP_OK;
}

int rgp_c3d_profile_enable(rgp_c3d_t* c, int enable) {
  RGP_REQUIRE(c, "rgp_c3d_profile_enable: null plan");
  c->prof.enabled = enable != 0;
  return RGP_OK;
}

int rgp_c3d_profile_read(rgp_c3d_t* c, double ms[RGP_C3D_STAGES], long long calls[RGP_C3D_STAGES]) {
  RGP_REQUIRE(c && ms && calls, "rgp_c3d_profile_read: null argument");
  return c->prof.read(ms, calls, RGP_C3D_STAGES);
}

size_t rgp_c3d_layer_elems(const rgp_c3d_t* c, int layer, int n_windows) {
  if (!c || layer < 0 || layer > 7) return 0;
  return (size_t)n_windows * c->unpad_tab[layer].size() * kLayers[layer].cout;
}

int rgp_c3d_read_layer(rgp_c3d_t* c, int layer, int n_windows, float* dst, rgp_stream_t stream) {
  RGP_REQUIRE(c && c->ws && dst && layer >= 0 && layer <= 7 && n_windows > 0 && n_windows <= c->max_windows,
              "rgp_c3d_read_layer: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int rows = (int)c->unpad_tab[layer].size();
  int C = kLayers[layer].cout;
  const long long stride = c->L[layer].out_img_stride;
  if (layer == 7) {
    // rows buffer [49][d*512+c]: expose as NDHWC [2,7,7,512] through the same table
  }
  const long long total = (long long)n_windows * rows * C;
  const int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
  const int* tab = (const int*)(c->ws + c->unpad_off[layer]);
  const char* src = c->ws + c->act_off[layer + 1];
  if (c->dtype == RGP_BF16) unpad_kernel<bf16_t><<<blocks, 256, 0, s>>>((const bf16_t*)src, dst, tab, rows, C, stride, total);
  else unpad_kernel<float><<<blocks, 256, 0, s>>>((const float*)src, dst, tab, rows, C, stride, total);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

}  // extern "C"
