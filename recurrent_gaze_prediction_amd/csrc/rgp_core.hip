// librgp_hip.so: error reporting, device info, softmax / cross-entropy entry point.
#include <mutex>
#include <set>
#include <utility>

#include "rgp_host.h"

namespace rgp {

int ensure_dyn_smem(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  int dev = 0;
  RGP_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  if (done.count({dev, kernel})) return RGP_OK;
  RGP_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done.insert({dev, kernel});
  return RGP_OK;
}

int device_cu_count(int* n_cu) {
  static std::mutex mu;
  static int cached[64] = {0};
  int dev = 0;
  RGP_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  if (dev < 0 || dev >= 64 || !cached[dev]) {
    int n = 0;
    RGP_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    n = n / 8 * 8;
    if (n <= 0) n = 256;
    if (dev < 0 || dev >= 64) { *n_cu = n; return RGP_OK; }
    cached[dev] = n;
  }
  *n_cu = cached[dev];
  return RGP_OK;
}

thread_local char g_err[512] = "";

int set_err(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int upload_desc(const ConvDesc& d, char* ws, hipStream_t s) {
  // Tables are tiny; the host vectors outlive the copy (they are plan members), and
  // pageable-memory hipMemcpyAsync stages synchronously, so this is safe.
  if (!d.in_tab.empty()) RGP_HIP(hipMemcpyAsync(ws + d.in_tab_off, d.in_tab.data(), d.in_tab.size() * 4, hipMemcpyHostToDevice, s));
  if (!d.out_tab.empty()) RGP_HIP(hipMemcpyAsync(ws + d.out_tab_off, d.out_tab.data(), d.out_tab.size() * 4, hipMemcpyHostToDevice, s));
  if (!d.koff.empty()) RGP_HIP(hipMemcpyAsync(ws + d.koff_off, d.koff.data(), d.koff.size() * 4, hipMemcpyHostToDevice, s));
  if (!d.koff_tm.empty()) RGP_HIP(hipMemcpyAsync(ws + d.koff_tm_off, d.koff_tm.data(), d.koff_tm.size() * 4, hipMemcpyHostToDevice, s));
  if (!d.tap_src.empty()) RGP_HIP(hipMemcpyAsync(ws + d.tap_src_off, d.tap_src.data(), d.tap_src.size() * 4, hipMemcpyHostToDevice, s));
  return RGP_OK;
}

}  // namespace rgp

using namespace rgp;

extern "C" {

const char* rgp_last_error(void) { return g_err; }

int rgp_version(void) { return 100; }

int rgp_device_arch(char* buf, int buflen) {
  if (!buf || buflen <= 0) return set_err(RGP_EINVAL, "rgp_device_arch: no buffer");
  int dev = 0;
  RGP_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  RGP_HIP(hipGetDeviceProperties(&prop, dev));
  snprintf(buf, buflen, "%s", prop.gcnArchName);
  return RGP_OK;
}

int rgp_softmax_xent_fwd(const float* logits, const float* labels, float* probs, float* frame_loss, float* loss,
                         int frames, int npix, rgp_stream_t stream) {
  RGP_REQUIRE(logits && frames > 0 && npix > 0 && npix <= 12 * 256, "softmax_xent: bad shape frames=%d npix=%d", frames, npix);
  RGP_REQUIRE(!loss || (labels && frame_loss), "softmax_xent: loss needs labels and frame_loss");
  hipStream_t s = (hipStream_t)stream;
  softmax_xent_kernel<<<frames, 256, 0, s>>>(logits, labels, probs, frame_loss, npix);
  RGP_HIP(hipGetLastError());
  if (loss) {
    loss_reduce_kernel<<<1, 256, 0, s>>>(frame_loss, loss, frames, 1.0f / (float)frames);
    RGP_HIP(hipGetLastError());
  }
  return RGP_OK;
}

}  // extern "C"
