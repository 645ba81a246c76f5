// librgp_hip.so: launches of the patch kernels (conv_patch.hip.h, conv_patch14.hip.h) -- conv2a .. conv4b forward and
// their input gradients, bf16.  Spec: /root/reference/C3D/.../c3d_prototxt/feature_extration.prototxt:67-240.
#include "rgp_c3d_plan.h"
#include "conv_patch.hip.h"
#include "conv_patch_slab.hip.h"
#include "conv_patch14.hip.h"
#include "conv_patch7.hip.h"

using namespace rgp;

// bf16 conv2a + pool2 / conv3a / conv3b + pool3 (conv_patch.hip.h); a training plan records the arg-max codes of the
// pooled layers
template <int CIN, int NOUT, int HW, int DEPTH, bool POOL, bool ARGMAX>
static int run_conv_patch(rgp_c3d* c, int layer, int n, hipStream_t s) {
  using Cfg = PatchCfg<CIN, NOUT, HW, DEPTH, POOL>;
  ConvPatchParams p;
  p.in = (const bf16_t*)(c->ws + c->act_off[layer]);
  p.wp = (const bf16_t*)(c->ws + c->L[layer].w_off);
  p.bias = c->bias[layer];
  p.out = (bf16_t*)(c->ws + c->act_off[layer + 1]);
  p.argmax = ARGMAX ? (unsigned char*)(c->ws + c->B[layer].argmax_off) : nullptr;
  p.mask = nullptr;
  p.n_windows = n;
  p.ablate = dev_knob("RGP_CP_ABLATE", 0);
  int n_cu = 0;
  RGP_TRY(device_cu_count(&n_cu));
  auto kern = conv_patch_bf16_kernel<CIN, NOUT, HW, DEPTH, POOL, ARGMAX>;
  RGP_TRY(ensure_dyn_smem((const void*)kern, Cfg::SMEM));
  kern<<<n_cu, 512, Cfg::SMEM, s>>>(p);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

// conv2a + pool2 on the plane-slab variant (conv_patch_slab.hip.h): inference plans
static int run_conv2a_slab(rgp_c3d* c, int n, hipStream_t s) {
  using Cfg = PatchSlabCfg<64, 128, 56, 16>;
  ConvPatchParams p;
  p.in = (const bf16_t*)(c->ws + c->act_off[1]);
  p.wp = (const bf16_t*)(c->ws + c->L[1].w_off);
  p.bias = c->bias[1];
  p.out = (bf16_t*)(c->ws + c->act_off[2]);
  p.argmax = nullptr;
  p.mask = nullptr;
  p.n_windows = n;
  p.ablate = dev_knob("RGP_CP_ABLATE", 0);
  int n_cu = 0;
  RGP_TRY(device_cu_count(&n_cu));
  auto kern = conv_patch_slab_bf16_kernel<64, 128, 56, 16>;
  RGP_TRY(ensure_dyn_smem((const void*)kern, Cfg::SMEM));
  kern<<<n_cu, 512, Cfg::SMEM, s>>>(p);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

template <int CIN, bool POOL, bool ARGMAX>
static int run_conv_patch14(rgp_c3d* c, int layer, int n, hipStream_t s) {
  using Cfg = Patch14Cfg<CIN, POOL>;
  ConvPatchParams p;
  p.in = (const bf16_t*)(c->ws + c->act_off[layer]);
  p.wp = (const bf16_t*)(c->ws + c->L[layer].w_off);
  p.bias = c->bias[layer];
  p.out = (bf16_t*)(c->ws + c->act_off[layer + 1]);
  p.argmax = ARGMAX ? (unsigned char*)(c->ws + c->B[layer].argmax_off) : nullptr;
  p.mask = nullptr;
  p.n_windows = n;
  p.ablate = dev_knob("RGP_CP_ABLATE", 0);
  int n_cu = 0;
  RGP_TRY(device_cu_count(&n_cu));
  auto kern = conv_patch14_bf16_kernel<CIN, POOL, ARGMAX>;
  RGP_TRY(ensure_dyn_smem((const void*)kern, Cfg::SMEM));
  kern<<<n_cu, 512, Cfg::SMEM, s>>>(p);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

// conv5a (-> halo-padded act7) and conv5b (-> the conv5b rows the projection consumes), conv_patch7.hip.h
template <int OUT>
static int run_conv_patch7(rgp_c3d* c, int layer, int n, hipStream_t s) {
  ConvPatchParams p;
  p.in = (const bf16_t*)(c->ws + c->act_off[layer]);
  p.wp = (const bf16_t*)(c->ws + c->L[layer].w_off);
  p.bias = c->bias[layer];
  p.out = (bf16_t*)(c->ws + c->act_off[layer + 1]);
  p.argmax = nullptr;
  p.mask = nullptr;
  p.n_windows = n;
  p.ablate = 0;
  int n_cu = 0;
  RGP_TRY(device_cu_count(&n_cu));
  auto kern = conv_patch7_bf16_kernel<OUT>;
  RGP_TRY(ensure_dyn_smem((const void*)kern, Patch7Cfg::SMEM));
  kern<<<n_cu, 512, Patch7Cfg::SMEM, s>>>(p);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

int run_conv_patch_bf16(rgp_c3d* c, int layer, int n, hipStream_t s) {
  if (layer == 6) return run_conv_patch7<0>(c, layer, n, s);
  if (layer == 7) return run_conv_patch7<1>(c, layer, n, s);
  if (layer == 4) return run_conv_patch14<256, false, false>(c, layer, n, s);
  if (layer == 5) return c->save ? run_conv_patch14<512, true, true>(c, layer, n, s) : run_conv_patch14<512, true, false>(c, layer, n, s);
  if (layer == 1 && c->conv2a_slab()) return run_conv2a_slab(c, n, s);
  if (layer == 1) return c->save ? run_conv_patch<64, 128, 56, 16, true, true>(c, layer, n, s) : run_conv_patch<64, 128, 56, 16, true, false>(c, layer, n, s);
  if (layer == 2) return run_conv_patch<128, 256, 28, 8, false, false>(c, layer, n, s);
  if (layer == 3) return c->save ? run_conv_patch<256, 256, 28, 8, true, true>(c, layer, n, s) : run_conv_patch<256, 256, 28, 8, true, false>(c, layer, n, s);
  return set_err(RGP_EINVAL, "conv_patch: no kernel for layer %d", layer);
}

// input gradients (bf16): the un-pooled patch kernels on dY with the backward plan's rotated filter.  conv3b / conv4b /
// conv5b: masked by the forward activation, written as dY of conv3a / conv4a / conv5a; conv2a / conv3a / conv5a: dense,
// for the un-pool kernel
int run_conv_patch_dgrad_bf16(rgp_c3d* c, int layer, int n, hipStream_t s) {
  ConvPatchParams p;
  p.in = (const bf16_t*)(c->ws + c->B[layer].dypre_off);
  p.wp = (const bf16_t*)(c->ws + c->B[layer].dg.w_off);
  p.bias = nullptr;
  p.out = (bf16_t*)(c->ws + c->B[layer - 1].dypre_off);
  p.argmax = nullptr;
  p.mask = (const bf16_t*)(c->ws + c->act_off[layer]);
  p.n_windows = n;
  p.ablate = 0;
  int n_cu = 0;
  RGP_TRY(device_cu_count(&n_cu));
  if (layer == 1 || layer == 2) {
    // into a pooled layer: dense, un-masked [n][D*H*W][cin] image for the un-pool kernel (wave tiles of 112 x 32)
    p.out = (bf16_t*)(c->ws + c->dyp_off);
    p.mask = nullptr;
    if (layer == 1) {
      auto kern = conv_patch_bf16_kernel<128, 64, 56, 16, false, false, true, true>;
      constexpr int smem = PatchCfg<128, 64, 56, 16, false>::SMEM;
      RGP_TRY(ensure_dyn_smem((const void*)kern, smem));
      kern<<<n_cu, 512, smem, s>>>(p);
    } else {
      auto kern = conv_patch_bf16_kernel<256, 128, 28, 8, false, false, true, true>;
      constexpr int smem = PatchCfg<256, 128, 28, 8, false>::SMEM;
      RGP_TRY(ensure_dyn_smem((const void*)kern, smem));
      kern<<<n_cu, 512, smem, s>>>(p);
    }
  } else if (layer == 3) {
    auto kern = conv_patch_bf16_kernel<256, 256, 28, 8, false, false, true>;
    constexpr int smem = PatchCfg<256, 256, 28, 8, false>::SMEM;
    RGP_TRY(ensure_dyn_smem((const void*)kern, smem));
    kern<<<n_cu, 512, smem, s>>>(p);
  } else if (layer == 5) {
    auto kern = conv_patch14_bf16_kernel<512, false, false, true>;
    constexpr int smem = Patch14Cfg<512, false>::SMEM;
    RGP_TRY(ensure_dyn_smem((const void*)kern, smem));
    kern<<<n_cu, 512, smem, s>>>(p);
  } else if (layer == 4) {                                   // conv4a: dense [n][784][256] for the un-pool kernel (pool3)
    p.out = (bf16_t*)(c->ws + c->dyp_off);
    p.mask = nullptr;
    auto kern = conv_patch14_bf16_kernel<512, false, false, true, 256, true>;
    constexpr int smem = Patch14Cfg<512, false, 256>::SMEM;
    RGP_TRY(ensure_dyn_smem((const void*)kern, smem));
    kern<<<n_cu, 512, smem, s>>>(p);
  } else if (layer == 7) {                                   // conv5b: masked by conv5a's activation, written as dY of conv5a
    auto kern = conv_patch7_bf16_kernel<0, true>;
    RGP_TRY(ensure_dyn_smem((const void*)kern, Patch7Cfg::SMEM));
    kern<<<n_cu, 512, Patch7Cfg::SMEM, s>>>(p);
  } else if (layer == 6) {                                   // conv5a: dense [n][98][512] for the un-pool kernel (pool4)
    p.out = (bf16_t*)(c->ws + c->dyp_off);
    p.mask = nullptr;
    auto kern = conv_patch7_bf16_kernel<2, true>;
    RGP_TRY(ensure_dyn_smem((const void*)kern, Patch7Cfg::SMEM));
    kern<<<n_cu, 512, Patch7Cfg::SMEM, s>>>(p);
  } else {
    return set_err(RGP_EINVAL, "conv_patch dgrad: no kernel for layer %d", layer);
  }
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}
