// librgp_hip.so: launch of the dedicated bf16 conv1a + bias + ReLU + pool1 kernel (conv1a.hip.h).
// A translation unit of its own because it is compiled with -fno-honor-nans (see the header comment of conv1a.hip.h):
// ONLY this kernel is built that way; the patch kernels of conv2a .. conv4b live in rgp_conv_patch.hip, built with the
// default floating-point semantics like the rest of the library.
#include "rgp_c3d_plan.h"
#include "conv1a.hip.h"

using namespace rgp;

int run_conv1a_bf16(rgp_c3d* c, int n, hipStream_t s, const float* video) {
  Conv1aParams p;
  p.video = video;
  p.in = (const bf16_t*)(c->ws + c->act_off[0]);
  p.wp = (const bf16_t*)(c->ws + c->L[0].w_off);
  p.bias = c->bias[0];
  p.out = (bf16_t*)(c->ws + c->act_off[1]);
  p.argmax = c->save ? (unsigned char*)(c->ws + c->B[0].argmax_off) : nullptr;
  p.n_windows = n;
  // two 4-wave blocks per CU, a multiple of 8 so that every XCD gets the same number of job slots
  int n_cu = 0;
  RGP_TRY(device_cu_count(&n_cu));
  const int grid = 2 * n_cu;
  auto launch = [&](auto kern) -> int {
    RGP_TRY(ensure_dyn_smem((const void*)kern, C1_SMEM));
    kern<<<grid, 256, C1_SMEM, s>>>(p);
    RGP_HIP(hipGetLastError());
    return RGP_OK;
  };
#ifdef RGP_DEV_KNOBS
  if (video && !p.argmax) switch (dev_knob("RGP_C1VAR", 0)) {
    case 2: return launch(conv1a_pool_bf16_kernel<true, false, 2>);
    case 4: return launch(conv1a_pool_bf16_kernel<true, false, 4>);
    case 6: return launch(conv1a_pool_bf16_kernel<true, false, 6>);
    default: break;
  }
#endif
  if (video && !p.argmax) return launch(conv1a_pool_bf16_kernel<true, false>);
  if (video) return launch(conv1a_pool_bf16_kernel<true, true>);
  if (p.argmax) return launch(conv1a_pool_bf16_kernel<false, true>);
  return launch(conv1a_pool_bf16_kernel<false, false>);
}

