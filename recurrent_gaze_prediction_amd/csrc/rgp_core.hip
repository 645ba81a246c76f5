// librgp_hip.so: error reporting, device info, softmax / cross-entropy entry point.
#include <array>
#include <map>
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

// Side streams: ONE pool of three per device for every plan of the process.  The runtime maps streams onto four hardware
// queues in creation order; a plan that made streams of its own late in a process's life (the tenth plan of a benchmark
// script, say) got queues that other streams already shared, and chains meant to run side by side ran one after the other
// (profiles/r05_configs.json before / after).  Calls fork behind the caller's stream and join before they return, so plans
// sharing the pool only ever delay each other.  Never destroyed.
int pool_stream(int i, bool create, hipStream_t* out) {
  static std::mutex mu;
  static hipStream_t pool[64][3] = {};
  int dev = 0;
  RGP_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || i < 0 || i >= 3) return set_err(RGP_EINVAL, "pool_stream: device %d, stream %d", dev, i);
  std::lock_guard<std::mutex> lock(mu);
  if (!pool[dev][i] && create) RGP_HIP(hipStreamCreateWithFlags(&pool[dev][i], hipStreamNonBlocking));
  *out = pool[dev][i];
  return RGP_OK;
}

namespace {
struct PersistentGuard {
  std::mutex mu;
  hipEvent_t ev[64] = {nullptr};
  bool recorded[64] = {false};
};
PersistentGuard g_guard;
bool stream_capturing(hipStream_t s) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}
}  // namespace

PersistentLaunch::PersistentLaunch(hipStream_t s) : s_(s), dev_(-1), locked_(false), rc_(RGP_OK) {
  if (stream_capturing(s)) return;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { rc_ = set_err(RGP_EHIP, "persistent launch: hipGetDevice failed"); return; }
  if (dev < 0 || dev >= 64) return;
  g_guard.mu.lock();                      // held until the destructor: wait-event, launch and record are one critical section
  locked_ = true;
  dev_ = dev;
  if (g_guard.recorded[dev]) {
    const hipError_t e = hipStreamWaitEvent(s, g_guard.ev[dev], 0);
    if (e != hipSuccess) rc_ = set_err(RGP_EHIP, "persistent launch: hipStreamWaitEvent: %s", hipGetErrorString(e));
  }
}

int PersistentLaunch::commit() {
  if (!locked_) return RGP_OK;
  if (!g_guard.ev[dev_]) RGP_HIP(hipEventCreateWithFlags(&g_guard.ev[dev_], hipEventDisableTiming));
  RGP_HIP(hipEventRecord(g_guard.ev[dev_], s_));
  g_guard.recorded[dev_] = true;
  return RGP_OK;
}

PersistentLaunch::~PersistentLaunch() {
  if (locked_) g_guard.mu.unlock();
}

thread_local char g_err[512] = "";

namespace {
__global__ void wgrad_row_tables_kernel(WgradGeom g, int n, int* __restrict__ xt, int* __restrict__ yt) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int Mw = g.D * g.H * g.W;
  const long long img = e / Mw;
  int ml = e - (int)img * Mw;
  const int x = ml % g.W;
  ml /= g.W;
  const int y = ml % g.H, z = ml / g.H;
  xt[e] = (int)((img * g.x_img_stride + (long long)z * g.x_sz + (long long)y * g.x_sy + (long long)x * g.x_sx) * g.esz);
  yt[e] = (int)((img * g.y_img_stride + g.y_org + (long long)z * g.y_sz + (long long)y * g.y_sy + (long long)x * g.y_sx) * g.esz);
}
}  // namespace

int wgrad_row_tables(const WgradGeom& g, hipStream_t s, const int** x_tab, const int** y_tab) {
  typedef std::array<long long, 15> Key;
  static std::mutex mu;
  static std::map<Key, std::pair<int*, int*>> cache;
  int dev = 0;
  RGP_HIP(hipGetDevice(&dev));
  const Key key = {dev, g.D, g.H, g.W, g.x_sz, g.x_sy, g.x_sx, g.y_sz, g.y_sy, g.y_sx, g.y_org, g.esz, g.x_img_stride, g.y_img_stride, 0};
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(key);
  if (it == cache.end()) {
    const long long Mw = (long long)g.D * g.H * g.W;
    if (Mw <= 0 || Mw > (1 << 28)) return set_err(RGP_EINVAL, "wgrad: bad grid %d x %d x %d", g.D, g.H, g.W);
    const long long n = Mw + 32, imgs = (n + Mw - 1) / Mw;
    const long long span_x = (imgs * g.x_img_stride + (long long)g.D * g.x_sz) * g.esz;
    const long long span_y = (imgs * g.y_img_stride + g.y_org + (long long)g.D * g.y_sz) * g.esz;
    if (span_x >= (1LL << 31) || span_y >= (1LL << 31) || span_x < 0 || span_y < 0)
      return set_err(RGP_EINVAL, "wgrad: image too large for 32-bit row offsets");
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    RGP_HIP(hipStreamIsCapturing(s, &cs));
    if (cs != hipStreamCaptureStatusNone)
      return set_err(RGP_EINVAL, "wgrad: first use of a geometry inside a stream capture (run one step eagerly first)");
    int* buf = nullptr;
    RGP_HIP(hipMalloc(&buf, (size_t)n * 2 * sizeof(int)));
    wgrad_row_tables_kernel<<<(int)((n + 255) / 256), 256, 0, s>>>(g, (int)n, buf, buf + n);
    RGP_HIP(hipGetLastError());
    RGP_HIP(hipStreamSynchronize(s));                          // visible to every stream that uses the cached pointers
    it = cache.emplace(key, std::make_pair(buf, buf + n)).first;
  }
  *x_tab = it->second.first;
  *y_tab = it->second.second;
  return RGP_OK;
}

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
