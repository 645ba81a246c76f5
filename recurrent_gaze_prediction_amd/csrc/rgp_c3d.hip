// librgp_hip.so: C3D conv1a..conv5b feature stack as fused implicit-GEMM launches
// (3x3x3 conv pad 1 + bias + ReLU + max-pool in one kernel per layer).
// Spec: /root/reference/C3D/.../c3d_prototxt/feature_extration.prototxt:22-342
// (the reference runs it as an offline Caffe binary, extract_C3D_features.py:689-724).
#include <algorithm>

#include "rgp_c3d_plan.h"
#include "conv1a.hip.h"   // layout constants (C1_K)

using namespace rgp;

namespace {

template <typename T, int G, int P>
int run_layer(rgp_c3d* c, int i, int n, hipStream_t s) {
  IgemmParams p = make_params(c->L[i], c->ws + c->act_off[i], c->ws, n);
  p.tile128 = c->tile128();
  EpiParams e = make_epi(c->L[i], c->ws + c->act_off[i + 1], c->ws);
  e.bias = c->bias[i];
  if (c->save && P > 1) e.argmax = (unsigned char*)(c->ws + c->B[i].argmax_off);
  // dev diagnostics (RGP_ABLATE=32 RGP_STAMP=<layer>): phase stamps of the staggered kernel
#ifdef RGP_DEV_KNOBS
  const int stamp_layer = dev_knob("RGP_STAMP", -1);
  if (stamp_layer == i) {
    static unsigned long long* dbg = nullptr;
    const int nblk = (p.M + 255) / 256 * ((p.N + 127) / 128);
    if (!dbg) RGP_HIP(hipMalloc(&dbg, (size_t)nblk * 64 * 8));
    RGP_HIP(hipMemsetAsync(dbg, 0, (size_t)nblk * 64 * 8, s));
    e.c_save = (float*)dbg;
    const int rc = launch_igemm<T, G, P, EpiStore<T, true, true>>(p, e, s);
    RGP_HIP(hipStreamSynchronize(s));
    std::vector<unsigned long long> h((size_t)nblk * 64);
    RGP_HIP(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
    double sumA[8] = {0}, sumB[8] = {0};
    for (int b = 0; b < nblk; ++b) for (int w = 0; w < 8; ++w) for (int k = 0; k < 8; ++k)
      (w < 4 ? sumA : sumB)[k] += (double)h[((size_t)b * 8 + w) * 8 + k];
    const double n = (double)nblk * 4 * p.nk;
    fprintf(stderr, "[stamp layer %d nk=%d] cycles per K-tile: group A dma %.0f reads %.0f wait %.0f bar1 %.0f mfma %.0f bar2 %.0f | group B dma %.0f reads %.0f wait %.0f bar1 %.0f mfma %.0f bar2 %.0f | loop total/ktile A %.0f | epilogue/loop A %.3f B %.3f | prologue cycles A %.0f B %.0f | in-kernel clock %.2f GHz (K loops: s_memtime / s_memrealtime x 100 MHz; the bar1 columns hold the wall ticks)\n",
            i, p.nk, sumA[0] / n, sumA[1] / n, sumA[2] / n, sumA[3] / n, sumA[4] / n, sumA[5] / n, sumB[0] / n, sumB[1] / n,
            sumB[2] / n, sumB[3] / n, sumB[4] / n, sumB[5] / n, sumA[6] / n, sumA[7] / sumA[6], sumB[7] / sumB[6], sumA[5] / (nblk * 4.0), sumB[5] / (nblk * 4.0), sumA[6] / sumA[3] * 0.1);
    return rc;
  }
#endif
  return launch_igemm<T, G, P, EpiStore<T, true, true>>(p, e, s);
}

template <typename T>
int layer_dispatch(rgp_c3d* c, int i, int n, hipStream_t s) {
  constexpr int G0 = sizeof(T) == 2 ? 4 : 2;
  if (i == 0 && sizeof(T) == 2) return run_conv1a_bf16(c, n, s);
  // conv2a ... conv5b: the patch kernels (conv_patch.hip.h, conv_patch14.hip.h, conv_patch7.hip.h); they read the filter in
  // its chunk-major packing
  if (sizeof(T) == 2 && c->use_patch() && ((i == 1 && dev_knob("RGP_C2PATCH", 1)) || (i == 3 && c->L[3].chunk_major == 64 && dev_knob("RGP_C3PATCH", 1)) ||
                                     (i == 2 && c->L[2].chunk_major == 64 && dev_knob("RGP_C3APATCH", 1)) ||
                                     ((i == 4 || i == 5) && c->L[i].chunk_major == 64 && dev_knob("RGP_C4PATCH", 1)) ||
                                     ((i == 6 || i == 7) && c->L[i].chunk_major == 64 && dev_knob("RGP_C5PATCH", 1))))
    return run_conv_patch_bf16(c, i, n, s);
  switch (i) {
    case 0: return run_layer<T, G0, 4>(c, i, n, s);
    case 1: case 3: case 5: return run_layer<T, 1, 8>(c, i, n, s);
    default: return run_layer<T, 1, 1>(c, i, n, s);
  }
}

// raw uint8 frames of one chunk of windows (the VIDEO_DATA layer's job)
struct FrameSrc {
  const unsigned char* frames = nullptr;
  int fh = 0, fw = 0;
  const float* mean = nullptr;
};

template <typename T>
int forward_chunk(rgp_c3d* c, const float* video, const FrameSrc* fs, int n, float* features, void* rows, hipStream_t s) {
  const long long npix = (long long)n * 16 * 112 * 112;
  const int blocks = (int)std::min<long long>((npix + 255) / 256, 65536);
  // bf16 inference from fp32 windows: conv1a converts and pads the input inside its patch fetch (no act0 image).  A
  // training plan keeps act0: the conv1a filter gradient reads it.
  const bool fused_in = sizeof(T) == 2 && !fs && !c->save && dev_knob("RGP_C1FUSE", 1);
  int pid = c->prof.begin(8, s);
  if (fs)
    frames_prep_kernel<T><<<blocks, 256, 0, s>>>(fs->frames, fs->fh, fs->fw, (const int*)(c->ws + c->starts_off), fs->mean,
                                                 (T*)(c->ws + c->act_off[0]), nullptr, npix);
  else if (!fused_in)
    video_prep_kernel<T><<<blocks, 256, 0, s>>>(video, (T*)(c->ws + c->act_off[0]), npix, 16, 112, 112);
  RGP_HIP(hipGetLastError());
  c->prof.end(pid, s);
  c->last_n = n;
  for (int i = 0; i < 8; ++i) {
    pid = c->prof.begin(i, s);
    if (i == 0 && fused_in) RGP_TRY(run_conv1a_bf16(c, n, s, video));
    else RGP_TRY(layer_dispatch<T>(c, i, n, s));
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
  // one launch for the eight packs (a training step re-packs after every optimizer update: as eight launches of 5 - 30 us
  // each they ran one after the other, none of them filling the chip)
  PackBatch<T> pk(c->ws, s);
  for (int i = 0; i < 8; ++i) {
    // (no memset: the packed area is zero from bind time outside the positions the pack writes)
    RGP_TRY(pk.add(c->L[i], w->w[i], kLayers[i].cout, 0));
    c->bias[i] = w->b[i];
  }
  RGP_TRY(pk.flush());
  c->weights_set = true;
  return RGP_OK;
}

}  // namespace

extern "C" {

int rgp_c3d_create(rgp_c3d_t** plan, int max_windows, int dtype) { return rgp_c3d_create_ex(plan, max_windows, dtype, 0); }

int rgp_c3d_create_ex(rgp_c3d_t** plan, int max_windows, int dtype, int flags) {
  RGP_REQUIRE(plan && max_windows > 0, "rgp_c3d_create: bad arguments");
  RGP_REQUIRE((flags & ~(RGP_C3D_SAVE_FOR_BACKWARD | RGP_C3D_KERNELS_IGEMM | RGP_C3D_KERNELS_TILE128 | RGP_C3D_CONV2A_ROWWISE)) == 0,
              "rgp_c3d_create_ex: unknown flags 0x%x", flags);
  RGP_REQUIRE(!(flags & RGP_C3D_KERNELS_TILE128) || (flags & RGP_C3D_KERNELS_IGEMM), "rgp_c3d_create_ex: RGP_C3D_KERNELS_TILE128 needs RGP_C3D_KERNELS_IGEMM");
  const int save_for_backward = flags & RGP_C3D_SAVE_FOR_BACKWARD;
  RGP_REQUIRE(dtype == RGP_F32 || dtype == RGP_BF16, "rgp_c3d_create: dtype %d", dtype);
  RGP_REQUIRE((long long)max_windows * 16 * 112 * 112 < (1LL << 31), "rgp_c3d_create: max_windows too large");
  rgp_c3d* c = new rgp_c3d();
  c->max_windows = max_windows;
  c->dtype = dtype;
  c->save = save_for_backward != 0;
  c->kernels = flags & (RGP_C3D_KERNELS_IGEMM | RGP_C3D_KERNELS_TILE128 | RGP_C3D_CONV2A_ROWWISE);
  bool ok = true;
  Arena a;
  for (int i = 0; i < 8; ++i) {
    const C3dLayerSpec& l = kLayers[i];
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
      if (dtype == RGP_BF16) {   // dedicated kernel: K = 27 (kz,ky,kx) taps x 4 padded to 128, filter [64][128]
        d.tap_src.clear();
        for (int t = 0; t < C1_K / 4; ++t) d.tap_src.push_back(t < 27 ? t : -1);
        d.pack_taps = C1_K / 4;
        d.K = C1_K;
      }
    } else {
      for (int kz = 0; kz < 3; ++kz) for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
        tapoff.push_back(((kz * Hp + ky) * Wp + kx) * C);
        fidx.push_back((kz * 3 + ky) * 3 + kx);
      }
      ok &= build_k_schedule(d, tapoff, fidx, C, dtype);
      d.s_tap = (long long)l.cin * l.cout; d.s_c = l.cout; d.s_n = 1;   // DHWIO
      if (dev_knob("RGP_KORDER", 1)) make_chunk_major(d, dtype);      // dev RGP_KORDER=0: tap-major K order
    }
    d.reserve(a, dtype);
    c->unpad_off[i] = a.take(c->unpad_tab[i].size() * 4);
  }
  c->act_stride[8] = 49LL * 1024;
  c->starts_off = a.take((size_t)max_windows * 4);
  if (!ok) { delete c; return set_err(RGP_EINVAL, "rgp_c3d_create: K schedule failed"); }
  for (int i = 0; i < 9; ++i) c->act_off[i] = a.take((size_t)max_windows * c->act_stride[i] * esize(dtype));
  if (c->save) {
    const int rc = c3d_bwd_plan(c, a);
    if (rc != RGP_OK) { delete c; return rc; }
  }
  // the patch kernels (conv_patch.hip.h) fetch plane slabs in whole LDS-DMA instructions: up to 76 pixels (39 KB) past the
  // last row a tile uses, i.e. past the end of an activation / gradient image for a window's last tile -- always inside
  // the workspace: this tail covers the case of an image that is the arena's last buffer
  a.take(64 * 1024);
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
  if (c->save) RGP_TRY(c3d_bwd_upload(c, s));
  return RGP_OK;
}

int rgp_c3d_set_weights(rgp_c3d_t* c, const rgp_c3d_weights* w, rgp_stream_t stream) {
  RGP_REQUIRE(c && w, "rgp_c3d_set_weights: null argument");
  if (!c->ws) return set_err(RGP_EWORKSPACE, "rgp_c3d: workspace not bound");
  for (int i = 0; i < 8; ++i) RGP_REQUIRE(w->w[i] && w->b[i], "rgp_c3d_set_weights: layer %d pointer is null", i);
  hipStream_t s = (hipStream_t)stream;
  RGP_TRY(c->dtype == RGP_BF16 ? set_weights_impl<bf16_t>(c, w, s) : set_weights_impl<float>(c, w, s));
  if (c->save) RGP_TRY(c3d_bwd_pack(c, w, s));
  return RGP_OK;
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
    RGP_TRY(c->dtype == RGP_BF16 ? forward_chunk<bf16_t>(c, v, nullptr, n, f, r, s)
                                 : forward_chunk<float>(c, v, nullptr, n, f, r, s));
  }
  return RGP_OK;
}

static int check_windows(const int* starts, int n_windows, int n_frames, const char* who) {
  for (int w = 0; w < n_windows; ++w)
    if (starts[w] < 0 || (long long)starts[w] + 16 > n_frames)
      return set_err(RGP_EINVAL, "%s: window %d starts at frame %d but only %d frames were given", who, w, starts[w], n_frames);
  return RGP_OK;
}

int rgp_c3d_forward_frames(rgp_c3d_t* c, const unsigned char* frames, int n_frames, int frame_h, int frame_w,
                           const int* window_starts, int n_windows, const float* mean_cube, float* features, void* rows,
                           rgp_stream_t stream) {
  RGP_REQUIRE(c && frames && window_starts && n_windows > 0 && n_frames >= 16 && frame_h > 0 && frame_w > 0,
              "rgp_c3d_forward_frames: bad arguments");
  RGP_REQUIRE((long long)n_frames * frame_h * frame_w * 3 < (1LL << 40), "rgp_c3d_forward_frames: frame stack too large");
  if (!c->ws) return set_err(RGP_EWORKSPACE, "rgp_c3d: workspace not bound");
  if (!c->weights_set) return set_err(RGP_ESTATE, "rgp_c3d: weights not set");
  RGP_TRY(check_windows(window_starts, n_windows, n_frames, "rgp_c3d_forward_frames"));
  hipStream_t s = (hipStream_t)stream;
  const size_t es = esize(c->dtype);
  FrameSrc fs;
  fs.frames = frames; fs.fh = frame_h; fs.fw = frame_w; fs.mean = mean_cube;
  for (int w0 = 0; w0 < n_windows; w0 += c->max_windows) {
    const int n = std::min(c->max_windows, n_windows - w0);
    RGP_HIP(hipMemcpyAsync(c->ws + c->starts_off, window_starts + w0, (size_t)n * 4, hipMemcpyHostToDevice, s));
    float* f = features ? features + (size_t)w0 * 1024 * 49 : nullptr;
    void* r = rows ? (char*)rows + (size_t)w0 * 49 * 1024 * es : nullptr;
    RGP_TRY(c->dtype == RGP_BF16 ? forward_chunk<bf16_t>(c, nullptr, &fs, n, f, r, s)
                                 : forward_chunk<float>(c, nullptr, &fs, n, f, r, s));
  }
  return RGP_OK;
}

int rgp_c3d_frames_to_video(rgp_c3d_t* c, const unsigned char* frames, int n_frames, int frame_h, int frame_w,
                            const int* window_starts, int n_windows, const float* mean_cube, float* video,
                            rgp_stream_t stream) {
  RGP_REQUIRE(c && frames && window_starts && video && n_windows > 0 && n_frames >= 16 && frame_h > 0 && frame_w > 0,
              "rgp_c3d_frames_to_video: bad arguments");
  if (!c->ws) return set_err(RGP_EWORKSPACE, "rgp_c3d: workspace not bound");
  RGP_TRY(check_windows(window_starts, n_windows, n_frames, "rgp_c3d_frames_to_video"));
  hipStream_t s = (hipStream_t)stream;
  for (int w0 = 0; w0 < n_windows; w0 += c->max_windows) {
    const int n = std::min(c->max_windows, n_windows - w0);
    RGP_HIP(hipMemcpyAsync(c->ws + c->starts_off, window_starts + w0, (size_t)n * 4, hipMemcpyHostToDevice, s));
    const long long npix = (long long)n * 16 * 112 * 112;
    frames_prep_kernel<float><<<(int)std::min<long long>((npix + 255) / 256, 65536), 256, 0, s>>>(
        frames, frame_h, frame_w, (const int*)(c->ws + c->starts_off), mean_cube, nullptr,
        video + (size_t)w0 * 16 * 112 * 112 * 3, npix);
    RGP_HIP(hipGetLastError());
  }
  return RGP_OK;
}

const char* rgp_c3d_layer_kernel_name(const rgp_c3d_t* c, int i, int n_windows) {
  static thread_local char buf[96];
  buf[0] = 0;
  if (!c || i < 0 || i > 7 || n_windows <= 0) return buf;
  const int n = std::min(n_windows, c->max_windows);
  const C3dLayerSpec& l = kLayers[i];
  const int P = l.pd * l.ph * l.ph;
  const bool bf = c->dtype == RGP_BF16;
  if (bf && i == 0) {
    snprintf(buf, sizeof(buf), "conv1a_pool_bf16_kernel<%s>", c->save ? "act0,argmax" : "fused");
  } else if (bf && c->use_patch() && i >= 1 && i <= 3 && (i == 1 || c->L[i].chunk_major == 64)) {
    snprintf(buf, sizeof(buf), "conv_patch%s_bf16_kernel<%d,%d,%d,%d,pool%d>", i == 1 && c->conv2a_slab() ? "_slab" : "", l.cin, l.cout,
             l.H, l.D, P);
  } else if (bf && c->use_patch() && (i == 4 || i == 5) && c->L[i].chunk_major == 64) {
    snprintf(buf, sizeof(buf), "conv_patch14_bf16_kernel<%d,pool%d>", l.cin, P);
  } else if (bf && c->use_patch() && (i == 6 || i == 7) && c->L[i].chunk_major == 64) {
    snprintf(buf, sizeof(buf), "conv_patch7_bf16_kernel<%s>", i == 6 ? "image" : "rows");
  } else {
    IgemmParams p = make_params(c->L[i], c->ws, c->ws, n);
    p.tile128 = c->tile128();
    IgemmTile t;
    using EB = EpiStore<bf16_t, true, true>;
    using EF = EpiStore<float, true, true>;
    if (bf) t = P == 8 ? igemm_tile_choice<bf16_t, 1, 8, EB>(p, 1) : P == 4 ? igemm_tile_choice<bf16_t, 4, 4, EB>(p, 1) : igemm_tile_choice<bf16_t, 1, 1, EB>(p, 1);
    else t = P == 8 ? igemm_tile_choice<float, 1, 8, EF>(p, 1) : P == 4 ? igemm_tile_choice<float, 2, 4, EF>(p, 1) : igemm_tile_choice<float, 1, 1, EF>(p, 1);
    snprintf(buf, sizeof(buf), "%s,%s,pool%d>", igemm_tile_name(t), bf ? "bf16" : "f32", P);
  }
  return buf;
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
