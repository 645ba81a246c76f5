// Plan object of the C3D conv stack, shared by the forward (rgp_c3d.hip) and backward
// (rgp_c3d_bwd.hip) translation units.
#pragma once
#include <vector>

#include "rgp_host.h"

struct C3dLayerSpec {
  int cin, cout, D, H, pd, ph;  // input extent (W == H), pooling window (depth, spatial); 1 = none
};
// conv1a..conv5b, feature_extration.prototxt:22-342
static const C3dLayerSpec kLayers[8] = {
    {3, 64, 16, 112, 1, 2},  {64, 128, 16, 56, 2, 2}, {128, 256, 8, 28, 1, 1}, {256, 256, 8, 28, 2, 2},
    {256, 512, 4, 14, 1, 1}, {512, 512, 4, 14, 2, 2}, {512, 512, 2, 7, 1, 1},  {512, 512, 2, 7, 1, 1},
};

// Backward state of one layer (present when the plan was created with save_for_backward).
struct C3dBwdLayer {
  rgp::ConvDesc dg;                 // dgrad implicit GEMM over dYpre with the rotated, in/out-swapped filter (layers 1..7)
  std::vector<int> x_tab, y_tab;    // wgrad row tables: window origin in act[i] / position in dYpre[i], natural (z,y,x) order
  std::vector<int> win_tab;         // pooled layers: origin of each pooling window in dYpre[i]
  std::vector<int> q_off;           // pooled layers: offset of window member q (dz,dy,dx order)
  size_t x_tab_off = 0, y_tab_off = 0, win_tab_off = 0, q_off_off = 0;
  size_t dypre_off = 0;             // [max_windows][D+2][H+2][W+2][Cout] operand dtype, halos zero
  long long dypre_stride = 0;       // elements per window
  size_t argmax_off = 0;            // pooled layers: [max_windows][Do*Ho*Wo][Cout] bytes
  size_t grad_w = 0, grad_b = 0;    // element offsets into the flat fp32 gradient vector
};

struct rgp_c3d {
  int max_windows = 0, dtype = RGP_BF16;
  int kernels = 0;               // 0: patch kernels for conv2a..conv4b (bf16); RGP_C3D_KERNELS_IGEMM / _TILE128 bits otherwise
  bool use_patch() const { return dtype == RGP_BF16 && !(kernels & RGP_C3D_KERNELS_IGEMM); }
  bool tile128() const { return (kernels & RGP_C3D_KERNELS_TILE128) != 0; }
  // conv2a's INFERENCE forward on the plane-slab variant of the patch kernel (conv_patch_slab.hip.h) instead of the row-wise
  // one (conv_patch.hip.h): same operands, bit-identical results; chosen by a same-box A/B (profiles/r05_ab_conv2a_slab.txt) and
  // switched off per plan by RGP_C3D_CONV2A_ROWWISE.  Training plans (arg-max codes recorded) stay on the row-wise kernel:
  // the slab variant's training forward measured 4 % slower.
  bool conv2a_slab() const { return use_patch() && !save && !(kernels & RGP_C3D_CONV2A_ROWWISE) && rgp::dev_knob("RGP_C2A_SLAB", 1) != 0; }
  rgp::ConvDesc L[8];
  size_t act_off[9] = {0};       // act[i] = halo-padded input of layer i; act[8] = conv5b rows
  long long act_stride[9] = {0}; // elements per window
  std::vector<int> unpad_tab[8];
  size_t unpad_off[8] = {0};
  size_t starts_off = 0;         // int32 [max_windows] first-frame index of each window (frames entry)
  size_t ws_bytes = 0;
  char* ws = nullptr;
  bool weights_set = false;
  const float* bias[8] = {nullptr};
  rgp::StageProfiler prof;
  // ---- backward ----
  bool save = false;
  C3dBwdLayer B[8];
  size_t dyp_off = 0;            // dense pooled-gradient scratch (largest pooled layer)
  size_t dw1_off = 0;            // conv1a filter gradient in its packed K order, fp32
  size_t n_params = 0;           // 27 655 936 = sum of w[i] + b[i]
  int last_n = 0;                // windows of the last forward (what backward differentiates)
  // recorded on the backward's stream once layer i's filter + bias gradient kernels are enqueued: lets the host
  // start the all-reduce of that layer's slice on another stream while the earlier layers are still differentiating
  hipEvent_t grad_ev[8] = {nullptr};
  bool grad_ev_made = false;
  // backward: conv5a's / conv5b's filter gradients (wgrad_kernel, one round of ingest-bound blocks) beside their input gradients
  rgp::SideStream side;
  ~rgp_c3d() {
    if (grad_ev_made) for (int i = 0; i < 8; ++i) (void)hipEventDestroy(grad_ev[i]);
  }
};

// rgp_conv1a.hip: the dedicated bf16 conv1a + pool1 kernel (video != nullptr: reads the fp32 windows directly)
int run_conv1a_bf16(rgp_c3d* c, int n, hipStream_t s, const float* video = nullptr);
int run_conv_patch_bf16(rgp_c3d* c, int layer, int n, hipStream_t s);   // layers 1 ... 7
int run_conv_patch_dgrad_bf16(rgp_c3d* c, int layer, int n, hipStream_t s);   // layers 1, 2, 4, 6 (dense), 3, 5, 7 (masked); training plans
// rgp_c3d_bwd.hip
int c3d_bwd_plan(rgp_c3d* c, rgp::Arena& a);                                      // tables + workspace layout
int c3d_bwd_upload(rgp_c3d* c, hipStream_t s);
int c3d_bwd_pack(rgp_c3d* c, const rgp_c3d_weights* w, hipStream_t s);            // rotated dgrad filters
