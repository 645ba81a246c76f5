// Host-side plumbing shared by the plan objects: error reporting, workspace arena,
// offset-table construction and the igemm launch dispatcher.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rgp.h"
#include "igemm.hip.h"
#include "kernels_misc.hip.h"
#include "igemm_stagger.hip.h"
#include "igemm_wide.hip.h"

namespace rgp {

extern thread_local char g_err[512];
int set_err(int code, const char* fmt, ...);

#define RGP_HIP(call)                                                                                  \
  do {                                                                                                 \
    hipError_t e_ = (call);                                                                            \
    if (e_ != hipSuccess) return set_err(RGP_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                                         __FILE__, __LINE__);                                          \
  } while (0)
#define RGP_TRY(call)          \
  do {                         \
    int r_ = (call);           \
    if (r_ != RGP_OK) return r_; \
  } while (0)
#define RGP_REQUIRE(cond, ...) \
  do {                         \
    if (!(cond)) return set_err(RGP_EINVAL, __VA_ARGS__); \
  } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Development switches (ablations, tile-order experiments: docs/HISTORY.md, "Development switches") exist only in a build made with
// `make DEV=1`; the product library reads no environment variable and keeps no per-process tuning state.
#ifdef RGP_DEV_KNOBS
inline int dev_knob(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#else
constexpr int dev_knob(const char*, int dflt) { return dflt; }
#endif

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE property of a kernel: cached per (device, kernel),
// so plans created on several devices of one process each get it.  device_cu_count: CUs of the current device,
// rounded down to a multiple of 8 (one slice per XCD).
int ensure_dyn_smem(const void* kernel, int bytes);
int device_cu_count(int* n_cu);

// The persistent ConvGRU kernels (convgru_seq.hip.h, convgru_bptt.hip.h) need every workgroup of a launch resident at
// once: two such launches must never share the device.  Within one process that is enforced here: a launch on stream s
// is made inside the life of a PersistentLaunch object, which takes a process-wide lock, makes s wait for the previous
// persistent launch of the process on the current device (whatever stream, plan or host THREAD issued it), and --
// commit(), after the launch -- records this one; the lock is held from the wait to the record, so two host threads
// cannot both pass the wait before either has recorded.  Skipped while s is being captured (a graph replays on one
// stream).  Other processes on the same device are outside its reach: the kernels' own time-out (RGP_ETIMEOUT) is what
// reports those.
class PersistentLaunch {
 public:
  explicit PersistentLaunch(hipStream_t s);
  ~PersistentLaunch();
  int status() const { return rc_; }   // RGP_OK, or why the wait could not be queued
  int commit();                        // after the kernel launch: record it
  PersistentLaunch(const PersistentLaunch&) = delete;
  PersistentLaunch& operator=(const PersistentLaunch&) = delete;
 private:
  hipStream_t s_;
  int dev_;
  bool locked_;
  int rc_;
};

// Row tables of the filter-gradient kernel (wgrad.hip.h): byte offset of row m = (z*H + y)*W + x of a D x H x W grid
// from its image in X (z*x_sz + y*x_sy + x*x_sx elements) and in dY (y_org + z*y_sz + y*y_sy + x*y_sx), entries
// [0, D*H*W + 32), entry e >= D*H*W continuing into the following image(s).  Built on the device the first time a
// geometry is seen on the current device (one hipMalloc + one small kernel on `s`, then cached for the life of the
// process); the first use of a geometry must therefore not happen inside a stream capture.
struct WgradGeom {
  int D, H, W, x_sz, x_sy, x_sx, y_sz, y_sy, y_sx, y_org, esz;
  long long x_img_stride, y_img_stride;
};
int wgrad_row_tables(const WgradGeom& g, hipStream_t s, const int** x_tab, const int** y_tab);

struct Arena {
  size_t off = 0;
  size_t take(size_t bytes) {
    const size_t o = off;
    off = align_up(off + bytes, 256);
    return o;
  }
};

inline int esize(int dtype) { return dtype == RGP_BF16 ? 2 : 4; }
inline int bke(int dtype) { return dtype == RGP_BF16 ? 64 : 32; }

// One implicit-GEMM problem: offset tables, K schedule and filter packing recipe.
struct ConvDesc {
  // geometry
  int Mw = 0, N = 0, K = 0, nk = 0, G = 1, P = 1;
  long long in_img_stride = 0, out_img_stride = 0;
  std::vector<int> in_tab, out_tab, koff;
  // chunk-major K order (make_chunk_major): koff / the packed filter run (channel chunk, tap) instead of (tap, channel
  // chunk); koff_tm keeps the tap-major table (wgrad_kernel: its row order is the filter's own DHWIO order)
  int chunk_major = 0;               // elements per chunk (BKE) when on
  std::vector<int> koff_tm;
  size_t koff_tm_off = 0;
  // filter packing: dst[n][tap*cin_k + c] = src[tap_src[tap]*s_tap + n*s_n + c*s_c]
  std::vector<int> tap_src;
  int pack_taps = 0, cin_k = 0, cin_src = 0;
  long long s_tap = 0, s_n = 0, s_c = 0;
  // workspace offsets (bytes)
  size_t in_tab_off = 0, out_tab_off = 0, koff_off = 0, tap_src_off = 0, w_off = 0;
  int n_pad() const { return (int)align_up((size_t)N, 128); }
  size_t w_bytes(int dtype) const { return (size_t)n_pad() * K * esize(dtype); }
  void reserve(Arena& a, int dtype) {
    in_tab_off = a.take(in_tab.size() * 4);
    out_tab_off = a.take(out_tab.size() * 4);
    koff_off = a.take(koff.size() * 4);
    koff_tm_off = koff_tm.empty() ? koff_off : a.take(koff_tm.size() * 4);
    tap_src_off = a.take(tap_src.size() * 4);
    w_off = a.take(w_bytes(dtype));
  }
};

// K schedule for taps of `cin` contiguous elements each: chunks of BKE elements.
// If cin < BKE several taps share one 128-byte chunk (G of them); the tap list is
// padded with zero-weight taps to a multiple of G.  Fills koff, K, nk, G and the
// packing tap list (tap_src = index into the filter's own tap order, -1 = zero).
inline bool build_k_schedule(ConvDesc& d, const std::vector<int>& tapoff, const std::vector<int>& tap_filter_idx,
                             int cin, int dtype) {
  const int B = bke(dtype);
  d.koff.clear();
  d.tap_src = tap_filter_idx;
  int ntaps = (int)tapoff.size();
  std::vector<int> offs = tapoff;
  if (cin >= B) {
    if (cin % B) return false;
    d.G = 1;
    for (int t = 0; t < ntaps; ++t)
      for (int c = 0; c < cin / B; ++c) d.koff.push_back(offs[t] + c * B);
  } else {
    if (B % cin) return false;
    d.G = B / cin;
    if (d.G != 2 && d.G != 4) return false;
    while (ntaps % d.G) {
      offs.push_back(offs[0]);
      d.tap_src.push_back(-1);
      ++ntaps;
    }
    d.koff = offs;
  }
  d.pack_taps = ntaps;
  d.cin_k = cin;
  d.cin_src = cin;
  d.K = ntaps * cin;
  d.nk = d.K / B;
  return true;
}

// Re-order the K schedule of a G == 1 problem from (tap, channel chunk) to (channel chunk, tap).  The im2col gather then
// sweeps all taps of ONE 128-byte channel slice before moving to the next: the slice of a tile's input region
// (a few hundred KB per XCD) stays in L2 across its 27 taps, where the tap-major order re-fetched the whole
// multi-MB region per tap (conv3b / conv4a / conv4b read 9...17x their input from beyond L2).
inline bool make_chunk_major(ConvDesc& d, int dtype) {
  const int B = bke(dtype);
  if (d.G != 1 || d.cin_k % B || d.cin_k / B < 2) return false;
  const int nc = d.cin_k / B, nt = d.pack_taps;
  d.koff_tm = d.koff;
  for (int t = 0; t < nt; ++t) for (int c = 0; c < nc; ++c) d.koff[c * nt + t] = d.koff_tm[t * nc + c];
  d.chunk_major = B;
  return true;
}

int upload_desc(const ConvDesc& d, char* ws, hipStream_t s);

template <typename T>
int pack_filter(const ConvDesc& d, const float* src, char* ws, int n_rows, int row0, hipStream_t s, int k0 = 0,
                int grouped = 0, int row_step = 1) {
  // grouped != 0: the packed K axis holds several source filters side by side per tap (column
  // offset k0, cin_src channels each); the buffer is pre-zeroed, padding is never written.
  if (!grouped && d.s_n == 1 && d.cin_k >= 32 && n_rows >= 32) {      // output channel contiguous in the source: tiled transpose
    const dim3 grid((d.cin_k + 31) / 32, (n_rows + 31) / 32, d.pack_taps);
    pack_filter_tiled_kernel<T><<<grid, 256, 0, s>>>(src, (T*)(ws + d.w_off), (const int*)(ws + d.tap_src_off), d.pack_taps, d.cin_k,
                                                     d.cin_src, n_rows, row0, d.K, d.s_tap, d.s_c, k0, row_step, d.chunk_major);
    RGP_HIP(hipGetLastError());
    return RGP_OK;
  }
  const long long total = (long long)n_rows * d.pack_taps * d.cin_k;
  const int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
  pack_filter_kernel<T><<<blocks, 256, 0, s>>>(src, (T*)(ws + d.w_off), (const int*)(ws + d.tap_src_off), d.pack_taps,
                                               d.cin_k, d.cin_src, n_rows, row0, d.K, d.s_tap, d.s_n, d.s_c, k0, grouped,
                                               row_step, d.chunk_major);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

// pack_filter() calls collected into launches of up to PACK_MAX_JOBS jobs (same arguments, same kernel choice per job)
template <typename T>
struct PackBatch {
  PackJobTable t;
  char* ws;
  hipStream_t s;
  int blocks = 0;
  PackBatch(char* ws_, hipStream_t s_) : ws(ws_), s(s_) { t.n = 0; t.first[0] = 0; }
  int flush() {
    if (t.n == 0) return RGP_OK;
    pack_filter_batch_kernel<T><<<blocks, 256, 0, s>>>(t);
    RGP_HIP(hipGetLastError());
    t.n = 0; blocks = 0; t.first[0] = 0;
    return RGP_OK;
  }
  int add(const ConvDesc& d, const float* src, int n_rows, int row0, int k0 = 0, int grouped = 0, int row_step = 1) {
    if (t.n == PACK_MAX_JOBS) RGP_TRY(flush());
    PackJob& q = t.job[t.n];
    q.src = src; q.dst = ws + d.w_off; q.tap_src = (const int*)(ws + d.tap_src_off);
    q.s_tap = d.s_tap; q.s_n = d.s_n; q.s_c = d.s_c;
    q.ntaps = d.pack_taps; q.cin_k = d.cin_k; q.cin_src = d.cin_src; q.n_rows = n_rows; q.row0 = row0; q.K = d.K; q.k0 = k0;
    q.grouped = grouped; q.row_step = row_step; q.chunk_major = d.chunk_major;
    int nb;
    // whole 64-channel chunks of a bf16 pack with 16-byte aligned rows on both sides: the vector forms (kernels_misc.hip.h)
    const bool wide = sizeof(T) == 2 && !grouped && row_step == 1 && d.cin_k % 64 == 0 && d.cin_src == d.cin_k && d.K % 8 == 0 &&
                      k0 % 8 == 0 && (d.chunk_major == 0 || d.chunk_major % 64 == 0) && ((size_t)src & 15) == 0 && d.s_tap % 4 == 0 &&
                      (d.w_off & 15) == 0;
    if (wide && d.s_n == 1 && d.s_c % 4 == 0 && n_rows % 64 == 0) {
      q.tiled = 2; q.gx = d.cin_k / 64; q.gy = n_rows / 64;
      nb = q.gx * q.gy * d.pack_taps;
    } else if (wide && d.s_c == 1 && d.s_n % 4 == 0) {
      const long long groups = (long long)n_rows * d.pack_taps * (d.cin_k / 8);
      q.tiled = 3; q.gy = 1; q.gx = nb = (int)std::min<long long>((groups + 255) / 256, 4096);
    } else if (!grouped && d.s_n == 1 && d.cin_k >= 32 && n_rows >= 32) {      // as pack_filter(): tiled transpose
      q.tiled = 1; q.gx = (d.cin_k + 31) / 32; q.gy = (n_rows + 31) / 32;
      nb = q.gx * q.gy * d.pack_taps;
    } else {
      const long long total = (long long)n_rows * d.pack_taps * d.cin_k;
      q.tiled = 0; q.gy = 1; q.gx = nb = (int)std::min<long long>((total + 255) / 256, 4096);
    }
    blocks += nb;
    t.first[++t.n] = blocks;
    return RGP_OK;
  }
};

// A stream of a plan's own for work its main chain does not wait for (weight gradients beside the data-gradient chain ...):
// fork(s, i, &sc) puts the side stream behind everything queued on s (event i) and hands it out -- or s itself while s is
// being captured and the side stream does not exist yet; join(s) makes s wait for the side stream to drain.  Made on first use.
int pool_stream(int i, bool create, hipStream_t* out);      // rgp_core.hip: side stream i (0 .. 2) of the current device, or null
struct SideStream {
  hipStream_t side = nullptr;                                // pool stream 0 (shared, not owned)
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr}, ev_join = nullptr;
  int fork(hipStream_t s, int i, hipStream_t* sc) {
    *sc = s;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = !(hipStreamIsCapturing(s, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone);
    if (!dev_knob("RGP_SIDE_STREAM", 1)) return RGP_OK;
    RGP_TRY(pool_stream(0, !capturing, &side));
    if (!side) return RGP_OK;
    if (!ev_join) {
      for (hipEvent_t* e : {&ev[0], &ev[1], &ev[2], &ev[3], &ev_join}) RGP_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    RGP_HIP(hipEventRecord(ev[i], s));
    RGP_HIP(hipStreamWaitEvent(side, ev[i], 0));
    *sc = side;
    return RGP_OK;
  }
  int join(hipStream_t s) {
    if (!side || !ev_join) return RGP_OK;
    RGP_HIP(hipEventRecord(ev_join, side));
    RGP_HIP(hipStreamWaitEvent(s, ev_join, 0));
    return RGP_OK;
  }
  ~SideStream() {
    if (ev_join) for (hipEvent_t e : {ev[0], ev[1], ev[2], ev[3], ev_join}) (void)hipEventDestroy(e);
  }
};

// ---- launch dispatch -------------------------------------------------------
template <typename T, int BM, int BN, int WM, int WN, int G, int P, class Epi>
int launch_cfg(const IgemmParams& p, const EpiParams& e, hipStream_t s, int ksplit = 1) {
  auto kern = igemm_kernel<T, BM, BN, WM, WN, G, P, Epi>;
  constexpr int smem = IgemmSmem<BM, BN>::BYTES;
  RGP_TRY(ensure_dyn_smem((const void*)kern, smem));
  const int n_mt = (p.M + BM - 1) / BM, n_nt = (p.N + BN - 1) / BN;
  kern<<<dim3(n_mt * n_nt, ksplit), dim3(WM * WN * 64), smem, s>>>(p, e);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

template <typename T, int P, class Epi, int ABLATE = 0>
int launch_stagger(const IgemmParams& p, const EpiParams& e, hipStream_t s) {
  auto kern = igemm_stagger_kernel<T, P, Epi, ABLATE>;
  constexpr int smem = StaggerSmem::BYTES;
  RGP_TRY(ensure_dyn_smem((const void*)kern, smem));
  const int n_mt = (p.M + 255) / 256, n_nt = (p.N + 127) / 128;
  // persistent: one block per CU walks the tile list (dev RGP_PERSIST=0: one block per tile)
  const int persist = dev_knob("RGP_PERSIST", 1);
  int n_cu = 0;
  RGP_TRY(device_cu_count(&n_cu));
  const int tiles = n_mt * n_nt;
  const int grid = persist ? std::min(tiles, n_cu) : tiles;
  IgemmParams q = p;
  const int ngrp = dev_knob("RGP_NGROUP", 1);      // 1 = column tiles innermost (measured best); -1 = one XCD's worth of row tiles per column tile
  q.ntile_group = ngrp >= 0 ? ngrp : std::max(1, std::min(grid, n_cu) / 8);
  q.pool_regs = dev_knob("RGP_POOLREGS", 1);
  kern<<<dim3(grid), dim3(512), smem, s>>>(q, e);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

// Epilogues the 256x256 kernel is instantiated for (the C3D forward / dgrad convolutions)
template <class E> struct EpiWideOk : std::false_type {};
template <typename TO, bool B, bool R> struct EpiWideOk<EpiStore<TO, B, R>> : std::true_type {};
template <typename TO> struct EpiWideOk<EpiStoreMask<TO>> : std::true_type {};

template <int BM, int BN, int P, class Epi, int VAR = 0>
int launch_wide(const IgemmParams& p, const EpiParams& e, hipStream_t s) {
  auto kern = igemm_wide_kernel<BM, BN, P, Epi, VAR>;
  constexpr int smem = WideSmem<BM, BN>::BYTES;
  RGP_TRY(ensure_dyn_smem((const void*)kern, smem));
  int n_cu = 0;
  RGP_TRY(device_cu_count(&n_cu));
  const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
  kern<<<dim3(std::min(tiles, n_cu)), dim3(512), smem, s>>>(p, e);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

// Tile choice by problem size and output width.  The persistent 256-row kernels (igemm_wide.hip.h, igemm_stagger.hip.h)
// need about a thousand tiles; below that 128x128 (2x2 waves of 64x64) for N >= 128, 128x64 (2x2 waves of 64x32) for N
// in (32, 64], 128x32 (4x1 waves of 32x32) below, and 64x64 for the latency-bound recurrent convs.
enum IgemmTile { TILE_WIDE_256x256, TILE_WIDE_512x128, TILE_STAGGER_256x128, TILE_LOOP_256x128, TILE_64x16, TILE_64x64, TILE_128x128, TILE_128x64, TILE_128x32 };

template <typename T, int G, int P, class Epi>
IgemmTile igemm_tile_choice(const IgemmParams& p, int ksplit) {
  // dev builds -- 2 = staggered 256x128 kernel where eligible (default, fastest measured), 0 = 128x128 only,
  // 1 = 256x128 simple loop (dev comparison)
  const int tile_cfg = p.tile128 ? 0 : dev_knob("RGP_TILE", 2);
  // N >= 256 bf16 convolutions at C3D scale: the 256x256 tile (two thirds of the L2 -> LDS bytes per FLOP)
  if constexpr (sizeof(T) == 2 && G == 1 && (P == 1 || P == 8) && EpiWideOk<Epi>::value) {
    // (at least four tiles per CU: with fewer -- conv5a/b at 1024 windows: 784 -- the last, partly filled round of the
    // persistent walk costs more than the bytes saved, and the 256x128 kernel's 1568 tiles run faster)
    const int wide = p.tile128 ? 0 : dev_knob("RGP_WIDE", 3);
    if (ksplit == 1 && p.nk >= 2 && p.nk <= 256) {
      const long long wtiles = (long long)((p.M + 255) / 256) * (p.N / 256);
      if ((wide & 1) && p.N % 256 == 0 && wtiles >= 1024) return TILE_WIDE_256x256;
      // short reductions (the head's projection: K = 1024, 16 K-tiles): a 256 x 128 tile ingests 48 KB per K-tile of 4.2 MFLOP and
      // is bound by the CU's LDS-DMA ingest (66-73 GB/s), not by its rounds: 256 x 256 tiles move two thirds of the bytes per
      // FLOP and win even at 1.5 rounds (392 tiles at 1024 frames: 82 -> see DESIGN.md us)
      if ((wide & 1) && p.N % 256 == 0 && p.nk <= 32 && wtiles >= 256 && dev_knob("RGP_WIDE_SHORTK", 1)) return TILE_WIDE_256x256;
      if ((wide & 2) && p.N == 128 && (p.M + 511) / 512 >= 1024) return TILE_WIDE_512x128;
    }
  }
  // (256 row tiles, or at least two 256x128 tiles per CU for its persistent walk; round 3: was M >= 65 536 alone, which
  // kept the head's hoisted W.x convolution and projection at 1024 frames -- M = 50 176 -- on the 128x128 tile)
  if (ksplit == 1 && G == 1 && tile_cfg == 2 && p.N % 128 == 0 && p.nk <= StaggerSmem::KOFF_MAX &&
      (p.M >= 256 * 256 || (long long)((p.M + 255) / 256) * (p.N / 128) >= 512)) return TILE_STAGGER_256x128;
  if (ksplit == 1 && p.N > 64 && tile_cfg == 1 && p.M >= 256 * 512) return TILE_LOOP_256x128;
  // one row tile and a wide output (the fc-GRU's recurrent GEMMs: M = B <= 64 rows, N = 1624 / 3248, K = 1624): even
  // 64x64 tiles give only 26 / 51 blocks, each bound by ONE CU's MFMA rate (40 us per launch, 32 launches per forward);
  // igemm_skinny_kernel spreads the same FLOPs over 102 / 203 CUs (64 x 16 columns per block, K split over its 16 waves)
  if constexpr (G == 1 && P == 1) {
    if (ksplit == 1 && p.M <= 64 && p.N >= 512 && (p.N + 63) / 64 < 128) return TILE_64x16;
  }
  // latency-bound problems (the per-timestep recurrent convs: M = B*49): 64x64 tiles give 4x the
  // blocks of 128x128, so more of the 256 CUs have a tile
  {
    const long long tiles128 = (long long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    if (p.N >= 64 && tiles128 * ksplit < 160) return TILE_64x64;
  }
  if (p.N > 64) return TILE_128x128;
  if (p.N > 32) return TILE_128x64;
  return TILE_128x32;
}

inline const char* igemm_tile_name(IgemmTile t) {
  switch (t) {
    case TILE_WIDE_256x256: return "igemm_wide_kernel<256x256";
    case TILE_WIDE_512x128: return "igemm_wide_kernel<512x128";
    case TILE_STAGGER_256x128: return "igemm_stagger_kernel<256x128";
    case TILE_LOOP_256x128: return "igemm_kernel<256x128";
    case TILE_64x16: return "igemm_skinny_kernel<64x16";
    case TILE_64x64: return "igemm_kernel<64x64";
    case TILE_128x128: return "igemm_kernel<128x128";
    case TILE_128x64: return "igemm_kernel<128x64";
    default: return "igemm_kernel<128x32";
  }
}

template <typename T, int G, int P, class Epi>
int launch_igemm(const IgemmParams& p, const EpiParams& e, hipStream_t s, int ksplit = 1) {
  if (p.M <= 0 || p.nk <= 0) return set_err(RGP_EINVAL, "igemm: empty problem M=%d nk=%d", p.M, p.nk);
  const IgemmTile tile = igemm_tile_choice<T, G, P, Epi>(p, ksplit);
  if constexpr (sizeof(T) == 2 && G == 1 && (P == 1 || P == 8) && EpiWideOk<Epi>::value) {
    if (tile == TILE_WIDE_256x256) {
#ifdef RGP_DEV_KNOBS
      if (dev_knob("RGP_WVAR", 0) == 1) return launch_wide<256, 256, P, Epi, 1>(p, e, s);
#endif
      return launch_wide<256, 256, P, Epi>(p, e, s);
    }
    if (tile == TILE_WIDE_512x128) {
#ifdef RGP_DEV_KNOBS
      if (dev_knob("RGP_WVAR", 0) == 2) return launch_wide<512, 128, P, Epi, 2>(p, e, s);
      if (dev_knob("RGP_WVAR", 0) == 3) return launch_wide<512, 128, P, Epi, 1>(p, e, s);
#endif
      return launch_wide<512, 128, P, Epi>(p, e, s);
    }
  }
  if (tile == TILE_STAGGER_256x128) {
#ifdef RGP_DEV_KNOBS
    const int abl = dev_knob("RGP_ABLATE", 0);
    if constexpr (sizeof(T) == 2) { if (abl == 256) return launch_stagger<T, P, Epi, 256>(p, e, s); }
    if constexpr (sizeof(T) == 2 && (P == 1 || P == 4)) { if (abl == 32) return launch_stagger<T, P, Epi, 32>(p, e, s); }
    if (sizeof(T) == 2 && P == 8 && abl) {
      if (abl == 1) return launch_stagger<T, P, Epi, 1>(p, e, s);
      if (abl == 2) return launch_stagger<T, P, Epi, 2>(p, e, s);
      if (abl == 4) return launch_stagger<T, P, Epi, 4>(p, e, s);
      if (abl == 6) return launch_stagger<T, P, Epi, 6>(p, e, s);
      if (abl == 3) return launch_stagger<T, P, Epi, 3>(p, e, s);
      if (abl == 5) return launch_stagger<T, P, Epi, 5>(p, e, s);
      if (abl == 32) return launch_stagger<T, P, Epi, 32>(p, e, s);
      if (abl == 128) return launch_stagger<T, P, Epi, 128>(p, e, s);
    }
#endif
    return launch_stagger<T, P, Epi>(p, e, s);
  }
  if constexpr (G == 1 && P == 1) {
    if (tile == TILE_64x16) {
      auto kern = igemm_skinny_kernel<T, Epi>;
      RGP_TRY(ensure_dyn_smem((const void*)kern, SkinnySmem::BYTES));
      kern<<<dim3((p.N + 15) / 16), dim3(1024), SkinnySmem::BYTES, s>>>(p, e);
      RGP_HIP(hipGetLastError());
      return RGP_OK;
    }
  }
  switch (tile) {
#ifdef RGP_DEV_KNOBS
    case TILE_LOOP_256x128: return launch_cfg<T, 256, 128, 4, 2, G, P, Epi>(p, e, s);
#endif
    case TILE_64x64: return launch_cfg<T, 64, 64, 2, 2, G, P, Epi>(p, e, s, ksplit);
    case TILE_128x128: return launch_cfg<T, 128, 128, 2, 2, G, P, Epi>(p, e, s, ksplit);
    case TILE_128x64: return launch_cfg<T, 128, 64, 2, 2, G, P, Epi>(p, e, s, ksplit);
    // (a 5-stage LDS-DMA ring on this tile -- 100 KB of LDS, one block per CU instead of four -- measured 0.54 -> 0.71 ms on
    // the head's deconvolution phases: co-resident blocks hide the K-tile latency better than a deeper ring; not kept)
    default: return launch_cfg<T, 128, 32, 4, 1, G, P, Epi>(p, e, s, ksplit);
  }
}

// n problems in one launch (igemm_grouped_kernel): dev_p / dev_e are device copies of ps / es.  Tile by the widest N as
// launch_igemm's last three cases (the persistent kernels walk one problem's tile list).
template <typename T, int G, int P, class Epi>
int launch_igemm_grouped(const std::vector<IgemmParams>& ps, const IgemmParams* dev_p, const EpiParams* dev_e, hipStream_t s,
                         long long a_bytes = 0, long long out_bytes = 0) {
  if (ps.empty()) return RGP_OK;
  int n_max = 0;
  for (const IgemmParams& p : ps) {
    if (p.M <= 0 || p.nk <= 0) return set_err(RGP_EINVAL, "igemm: empty problem M=%d nk=%d", p.M, p.nk);
    n_max = std::max(n_max, p.N);
  }
  auto go = [&](auto kern, int BM, int BN, int threads, int smem) -> int {
    RGP_TRY(ensure_dyn_smem((const void*)kern, smem));
    int tiles = 0;
    for (const IgemmParams& p : ps) tiles = std::max(tiles, ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN));
    kern<<<dim3(tiles, 1, (unsigned)ps.size()), dim3(threads), smem, s>>>(dev_p, dev_e, a_bytes, out_bytes);
    RGP_HIP(hipGetLastError());
    return RGP_OK;
  };
  if (n_max > 64) return go(igemm_grouped_kernel<T, 128, 128, 2, 2, G, P, Epi>, 128, 128, 256, IgemmSmem<128, 128>::BYTES);
  if (n_max > 32) return go(igemm_grouped_kernel<T, 128, 64, 2, 2, G, P, Epi>, 128, 64, 256, IgemmSmem<128, 64>::BYTES);
  return go(igemm_grouped_kernel<T, 128, 32, 4, 1, G, P, Epi>, 128, 32, 256, IgemmSmem<128, 32>::BYTES);
}

inline IgemmParams make_params(const ConvDesc& d, const void* A, char* ws, int n_img) {
  IgemmParams p;
  p.A = A;
  p.W = ws + d.w_off;
  p.in_tab = (const int*)(ws + d.in_tab_off);
  p.koff = (const int*)(ws + d.koff_off);
  p.in_img_stride = d.in_img_stride;
  p.Mw = d.Mw;
  p.M = d.Mw * n_img;
  p.N = d.N;
  p.K = d.K;
  p.nk = d.nk;
  return p;
}

inline EpiParams make_epi(const ConvDesc& d, void* out, char* ws) {
  EpiParams e;
  memset(&e, 0, sizeof(e));
  e.out = out;
  e.out_tab = (const int*)(ws + d.out_tab_off);
  e.out_img_stride = d.out_img_stride;
  return e;
}

// Stage timer: pairs of HIP events recorded on the launch stream, summed per stage at read().
struct StageProfiler {
  bool enabled = false;
  struct Rec { hipEvent_t a, b; int stage; };
  std::vector<Rec> pool;
  size_t used = 0;
  int begin(int stage, hipStream_t s) {
    if (!enabled) return -1;
    if (used == pool.size()) {
      Rec r; r.stage = stage;
      if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -1;
      pool.push_back(r);
    }
    pool[used].stage = stage;
    (void)hipEventRecord(pool[used].a, s);
    return (int)used++;
  }
  void end(int id, hipStream_t s) { if (id >= 0) (void)hipEventRecord(pool[id].b, s); }
  int read(double* ms, long long* calls, int nstages) {
    for (int i = 0; i < nstages; ++i) { ms[i] = 0; calls[i] = 0; }
    for (size_t i = 0; i < used; ++i) {
      float t = 0.f;
      RGP_HIP(hipEventSynchronize(pool[i].b));
      RGP_HIP(hipEventElapsedTime(&t, pool[i].a, pool[i].b));
      if (pool[i].stage < nstages) { ms[pool[i].stage] += t; calls[pool[i].stage] += 1; }
    }
    used = 0;
    return RGP_OK;
  }
  ~StageProfiler() { for (auto& r : pool) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); } }
};

// dst[img][i][c] (fp32, dense) = src[img*img_stride + tab[i] + c] for c < C.
template <typename T>
__global__ __launch_bounds__(256) void unpad_kernel(const T* __restrict__ src, float* __restrict__ dst,
                                                    const int* __restrict__ tab, int rows, int C, long long img_stride,
                                                    long long total) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int r = (int)((i / C) % rows);
    const long long img = i / ((long long)C * rows);
    dst[i] = Elem<T>::from(src[img * img_stride + tab[r] + c]);
  }
}

}  // namespace rgp
