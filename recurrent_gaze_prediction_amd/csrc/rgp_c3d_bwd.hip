// librgp_hip.so: backward of the C3D conv stack (end-to-end fine-tune, BASELINE config 5:
// tf.gradients through conv1a..conv5b of feature_extration.prototxt:22-342, base.py:278-281).
//
// Per layer i, from conv5b down:
//   dYpre[i]  gradient w.r.t. the conv output before pooling, in a halo-padded image of the conv's
//             own resolution (so dgrad is the forward implicit GEMM again and wgrad gathers rows of it)
//   wgrad     dW[i] += im2col(act[i])^T x dYpre[i]                       (wgrad.hip.h, transposing LDS reads)
//   dgrad     d act[i] = dYpre[i] (*) rot180(W[i]) with in/out swapped   (igemm.hip.h, packed once per set_weights)
//   unpool    pooled layer below: route each pooled gradient to the window member the forward epilogue
//             recorded as arg-max, gated by ReLU (output > 0); un-pooled layer below: the ReLU gate is
//             fused into the dgrad epilogue (EpiStoreMask).
// Bias gradients are column sums of dYpre[i] (fused into unpool where there is one).
#include <algorithm>

#include "rgp_c3d_plan.h"
#include "wgrad_launch.h"
#include "conv1a_wgrad.hip.h"
#include "wgrad_patch.hip.h"

using namespace rgp;

namespace {

inline bool pooled(int i) { return kLayers[i].pd * kLayers[i].ph > 1; }

// d(conv5b rows) -> dYpre[7] [n][4][9][9][512]: ReLU gate from the forward rows, layout change.
// src_features != 0: src is [n][1024][7][7] with channel c*2+d (gaze_rnn.py:494-497); else [n*49][d*512+c].
template <typename T>
__global__ __launch_bounds__(256) void rows_grad_kernel(const float* __restrict__ src, int src_features, const T* __restrict__ fwd_rows,
                                                        T* __restrict__ dypre, long long img_stride, long long total) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % 512);
    const int d = (int)((i / 512) % 2);
    const int pos = (int)((i / 1024) % 49);
    const long long n = i / (1024 * 49);
    const float g = src_features ? src[(n * 1024 + c * 2 + d) * 49 + pos] : src[(n * 49 + pos) * 1024 + d * 512 + c];
    const bool on = Elem<T>::from(fwd_rows[(n * 49 + pos) * 1024 + d * 512 + c]) > 0.f;
    const int y = pos / 7, x = pos % 7;
    dypre[n * img_stride + (((long long)(d + 1) * 9 + y + 1) * 9 + x + 1) * 512 + c] = Elem<T>::to(on ? g : 0.f);
  }
}

// Per-block reduction of the threads' 8-channel partial sums (thread = (row lane rl, channel group cg)),
// then ONE atomic per channel and block: thousands of threads adding to the same C addresses serialise.
__device__ __forceinline__ void block_colsum(const float* bsum, int cg, int rl, int RL, int C, float* __restrict__ db) {
  __shared__ float red[256 * 8];
#pragma unroll
  for (int k = 0; k < 8; ++k) red[rl * C + cg * 8 + k] = bsum[k];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f;
    for (int r = 0; r < RL; ++r) a += red[r * C + c];
    if (a != 0.f) atomicAdd(db + c, a);
  }
}

// Pooled layer: scatter dYp [n][PR][C] into dYpre through the recorded arg-max, gate by the pooled
// forward output y > 0, and accumulate the bias gradient.  Thread = (row lane, 8-channel group).
template <typename T>
__global__ __launch_bounds__(256) void unpool_kernel(const T* __restrict__ dyp, const unsigned char* __restrict__ amax,
                                                     const T* __restrict__ y, const int* __restrict__ y_tab, long long y_img_stride,
                                                     T* __restrict__ dypre, const int* __restrict__ win_tab,
                                                     const int* __restrict__ q_off, long long dypre_stride, int PR, int C, int P,
                                                     long long rows_total, float* __restrict__ db) {
  const int CG = C / 8;
  const int cg = threadIdx.x % CG, rl = threadIdx.x / CG, RL = 256 / CG;
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (long long r = (long long)blockIdx.x * RL + rl; r < rows_total; r += (long long)gridDim.x * RL) {
    const long long img = r / PR;
    const int pr = (int)(r - img * PR);
    float g[8];
    unsigned code[8];
    const T* gp = dyp + r * C + cg * 8;
    const T* yp = y + img * y_img_stride + y_tab[pr] + cg * 8;
    const unsigned char* ap = amax + r * C + cg * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool on = Elem<T>::from(yp[k]) > 0.f;
      g[k] = on ? Elem<T>::from(gp[k]) : 0.f;
      code[k] = ap[k];
      bsum[k] += g[k];
    }
    T* base = dypre + img * dypre_stride + win_tab[pr] + cg * 8;
    for (int q = 0; q < P; ++q) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = code[k] == (unsigned)q ? g[k] : 0.f;
      store8<T>(base + q_off[q], v, 8);
    }
  }
  block_colsum(bsum, cg, rl, RL, C, db);
}

// The same for the bf16 2x2x2-pooled layers (conv2a, conv3b, conv4b), laid out for the WRITE stream, which is 8x the reads
// (3.3 GB per 256 windows for conv2a): a store instruction of a wave covers ONE contiguous kilobyte of dYpre -- lanes =
// (window of a run of WPI windows adjacent in x, dx, 8-channel group), so the 64/LPP positions 2 xo + dx ... of an image
// row lie side by side -- instead of four 256-byte pieces of four windows.  The two dx lanes of a window load the same
// pooled gradient / code / activation (one request).  grid = (blocks, windows of the batch): no 64-bit division.
// (round 4: the three launches 1.7 -> see DESIGN.md ms per 256 windows)
template <int C>
__global__ __launch_bounds__(256) void unpool8_rows_kernel(const bf16_t* __restrict__ dyp, const unsigned char* __restrict__ amax,
                                                           const bf16_t* __restrict__ y, const int* __restrict__ y_tab,
                                                           long long y_img_stride, bf16_t* __restrict__ dypre,
                                                           const int* __restrict__ win_tab, long long dypre_stride, int PR,
                                                           int q_dz, int q_dy, float* __restrict__ db) {
  constexpr int LPP = C / 8;                       // lanes per position: 16 / 32 / 64
  constexpr int DXL = LPP == 64 ? 1 : 2;           // dx values across lanes (C = 512: dx is looped)
  constexpr int WPI = 64 / (LPP * DXL);            // windows per wave iteration: 2 / 1 / 1
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cg = lane % LPP, dxl = (lane / LPP) % DXL, wsub = lane / (LPP * DXL);
  const long long img = blockIdx.y;
  const bf16_t* gimg = dyp + img * (long long)PR * C;
  const unsigned char* aimg = amax + img * (long long)PR * C;
  const bf16_t* yimg = y + img * y_img_stride;
  bf16_t* oimg = dypre + img * dypre_stride;
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int pr0 = (blockIdx.x * 4 + wave) * WPI; pr0 < PR; pr0 += gridDim.x * 4 * WPI) {
    const int pr = pr0 + wsub;
    float g[8];
    mp_load8(gimg + (long long)pr * C + cg * 8, g);
    float yv[8];
    mp_load8(yimg + y_tab[pr] + cg * 8, yv);
    const uint2 cw = *reinterpret_cast<const uint2*>(aimg + (long long)pr * C + cg * 8);
    unsigned code[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      code[k] = ((k < 4 ? cw.x : cw.y) >> (8 * (k & 3))) & 0xffu;
      g[k] = yv[k] > 0.f ? g[k] : 0.f;
      if (DXL == 1 || dxl == 0) bsum[k] += g[k];
    }
    bf16_t* base = oimg + win_tab[pr] + cg * 8;
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dxi = 0; dxi < 2 / DXL; ++dxi) {
          const int dx = DXL == 2 ? dxl : dxi;
          const unsigned q = (unsigned)(dz * 4 + dy * 2 + dx);
          float v[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = code[k] == q ? g[k] : 0.f;
          store8<bf16_t>(base + dz * q_dz + dy * q_dy + dx * C, v, 8);
        }
  }
  // bias gradient: per-block reduction, one atomic per channel and block
  __shared__ float red[256 * 8];
#pragma unroll
  for (int k = 0; k < 8; ++k) red[threadIdx.x * 8 + k] = bsum[k];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f;
    for (int t = c / 8; t < 256; t += LPP) a += red[t * 8 + (c & 7)];
    if (a != 0.f) atomicAdd(db + c, a);
  }
}

template <int C>
static int launch_unpool8_rows(const bf16_t* dyp, const unsigned char* amax, const bf16_t* y, const int* y_tab, long long y_img_stride,
                               bf16_t* dypre, const int* win_tab, long long dypre_stride, int PR, int q_dz, int q_dy, int n,
                               float* db, hipStream_t s) {
  constexpr int WPI = C == 128 ? 2 : 1;
  const int per_block = 4 * WPI;
  int bx = (PR + per_block - 1) / per_block;
  // enough blocks to fill the chip about eight times over, each wave a few iterations
  const int want = std::max(1, 8192 / std::max(n, 1));
  bx = std::max(1, std::min(bx, want));
  unpool8_rows_kernel<C><<<dim3(bx, n), 256, 0, s>>>(dyp, amax, y, y_tab, y_img_stride, dypre, win_tab, dypre_stride, PR, q_dz, q_dy, db);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

// Bias gradient of an un-pooled layer: column sums over the interior rows of dYpre.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ buf, const int* __restrict__ tab, long long img_stride,
                                                     int Mw, int C, long long rows_total, float* __restrict__ db) {
  const int CG = C / 8;
  const int cg = threadIdx.x % CG, rl = threadIdx.x / CG, RL = 256 / CG;
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // four rows per thread and iteration: four independent 16-byte loads in flight (one per iteration read 0.8 GB of conv3a's dYpre
  // at 3.2 TB/s: latency-bound)
  const long long step = (long long)gridDim.x * RL;
  for (long long r0 = (long long)blockIdx.x * RL + rl; r0 < rows_total; r0 += 4 * step) {
    float v[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long r = r0 + u * step;
      if (r < rows_total) {
        const long long img = r / Mw;
        const int ml = (int)(r - img * Mw);
        mp_load8(buf + img * img_stride + tab[ml] + cg * 8, v[u]);        // (16-byte load: kernels_misc.hip.h)
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[u][k] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < 8; ++k) bsum[k] += v[u][k];
  }
  block_colsum(bsum, cg, rl, RL, C, db);
}

// conv1a: packed K order ((kz,ky) tap, kx 0..3, c 0..3) -> DHWIO [3,3,3,3,64], accumulated
__global__ void conv1a_unpack_grad_kernel(const float* __restrict__ dw1, float* __restrict__ grad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // over 27*3*64
  if (i >= 27 * 3 * 64) return;
  const int n = i % 64, c = (i / 64) % 3, tap = i / 192;
  const int kx = tap % 3, t2 = tap / 3;                  // t2 = kz*3+ky
  grad[i] += dw1[(long long)(t2 * 16 + kx * 4 + c) * 64 + n];
}

// conv2a ... conv4b filter gradients (bf16): one block per (32 input channels, 64 output channels, column range)
template <int CIN, int COUT, int HW, int DEPTH>
int launch_wgrad_patch(const bf16_t* x, const bf16_t* dy, float* dw, float* db, int n, hipStream_t s) {
  using Cfg = std::conditional_t<HW == 14, Wgp14Cfg<CIN, COUT>, WgpCfg<CIN, COUT, HW, DEPTH>>;
  int n_cu = 0;
  RGP_TRY(device_cu_count(&n_cu));
  WgradPatchParams p;
  p.x = x; p.dy = dy; p.dw = dw; p.db = db; p.n_windows = n;
  p.splits = std::max(8, n_cu / (Cfg::CS * Cfg::NS) / 8 * 8);          // a multiple of 8: one XCD per column range
  auto kern = wgrad_patch_bf16_kernel<CIN, COUT, HW, DEPTH>;
  RGP_TRY(ensure_dyn_smem((const void*)kern, Cfg::SMEM));
  kern<<<Cfg::CS * Cfg::NS * p.splits, 512, Cfg::SMEM, s>>>(p);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

// db != nullptr: the layer's bias gradient too (un-pooled layers; the pooled layers' comes from the un-pool kernel)
int run_wgrad_patch(rgp_c3d* c, int layer, int n, float* dw, float* db, hipStream_t s) {
  const bf16_t* x = (const bf16_t*)(c->ws + c->act_off[layer]);
  const bf16_t* dy = (const bf16_t*)(c->ws + c->B[layer].dypre_off);
  if (layer == 1) return launch_wgrad_patch<64, 128, 56, 16>(x, dy, dw, db, n, s);
  if (layer == 2) return launch_wgrad_patch<128, 256, 28, 8>(x, dy, dw, db, n, s);
  if (layer == 3) return launch_wgrad_patch<256, 256, 28, 8>(x, dy, dw, db, n, s);
  if (layer == 4) return launch_wgrad_patch<256, 512, 14, 4>(x, dy, dw, db, n, s);
  if (layer == 5) return launch_wgrad_patch<512, 512, 14, 4>(x, dy, dw, db, n, s);
  return set_err(RGP_EINVAL, "wgrad_patch: no kernel for layer %d", layer);
}

template <typename T>
int backward_impl(rgp_c3d* c, const float* d_features, const float* d_rows, float* grads, hipStream_t s_in) {
  char* ws = c->ws;
  hipStream_t s = s_in, sw = s_in;
  // Layers whose filter gradient runs on the plan's side stream beside the input gradient (both read dYpre[i]).  The large
  // layers' kernels are matrix-pipe bound and hold one block per CU: sharing the chip gains nothing there
  // (profiles/r05_c3d_bwd_fork_experiment.txt); conv5a / conv5b's are one round of ingest-bound blocks.
  const int fork_mask = sizeof(T) == 2 ? dev_knob("RGP_C3D_BWD_FORK", 0xC0) : 0;
  const int n = c->last_n;
  constexpr int G0 = sizeof(T) == 2 ? 4 : 2;
  auto blocks_for = [](long long items, int per_block) { return (int)std::min<long long>((items + per_block - 1) / per_block, 4096); };
  {  // conv5b: external gradient -> dYpre[7], bias gradient
    const long long total = (long long)n * 49 * 1024;
    rows_grad_kernel<T><<<blocks_for(total, 256), 256, 0, s>>>(d_features ? d_features : d_rows, d_features != nullptr,
                                                              (const T*)(ws + c->act_off[8]), (T*)(ws + c->B[7].dypre_off),
                                                              c->B[7].dypre_stride, total);
    RGP_HIP(hipGetLastError());
  }
  for (int i = 7; i >= 0; --i) {
    const C3dLayerSpec& l = kLayers[i];
    const C3dBwdLayer& b = c->B[i];
    const int Mw = l.D * l.H * l.H;
    if ((fork_mask >> i) & 1) {
      RGP_TRY(c->side.fork(s_in, i & 3, &sw));
      s = sw;
    }
    // filter gradient on the patch kernels (wgrad_patch.hip.h; dev builds: one mask bit per layer)?
    const bool wg_patch = sizeof(T) == 2 && c->use_patch() && i >= 1 && i <= 5 && ((dev_knob("RGP_WGPATCH", 31) >> (i - 1)) & 1);
    if (!pooled(i) && !wg_patch) {  // bias gradient (pooled layers: done by unpool below; patch kernels: with the filter gradient)
      const long long rows = (long long)n * Mw;
      colsum_kernel<T><<<std::min(blocks_for(rows, 256 / (l.cout / 8) * 16), 1024), 256, 0, s>>>(      // (every block ends with C atomics)
          
          (const T*)(ws + b.dypre_off), (const int*)(ws + b.y_tab_off), b.dypre_stride, Mw, l.cout, rows, grads + b.grad_b);
      RGP_HIP(hipGetLastError());
    }
    {  // filter gradient
      WgradParams p;
      p.X = ws + c->act_off[i];
      p.dY = ws + b.dypre_off;
      {
        const int Hp = l.H + 2, Wp = l.H + 2, Cx = i == 0 ? 4 : l.cin, Wpx = i == 0 ? l.H + 4 : l.H + 2;
        p.D = l.D; p.H = l.H; p.W = l.H;
        p.x_sx = Cx; p.x_sy = Wpx * Cx; p.x_sz = Hp * Wpx * Cx;
        p.y_sx = l.cout; p.y_sy = Wp * l.cout; p.y_sz = Hp * Wp * l.cout;
        p.y_org = p.y_sz + p.y_sy + p.y_sx;
      }
      p.koff = (const int*)(ws + c->L[i].koff_tm_off);      // tap-major: dW rows in DHWIO order
      p.x_img_stride = c->act_stride[i];
      p.y_img_stride = b.dypre_stride;
      p.M = (long long)n * Mw;
      p.Mw = Mw;
      p.N = l.cout;
      p.nk = c->L[i].nk;
      p.ldw = l.cout;
      p.k_valid = c->L[i].nk * Elem<T>::BKE;
      p.steps_per_split = 0;
      if (i == 0 && sizeof(T) == 2) {
        // dedicated kernel: filter + bias gradient straight from the pooled gradient and the arg-max codes
        Conv1aWgradParams q;
        q.in = (const bf16_t*)(ws + c->act_off[0]);
        q.dyp = (const bf16_t*)(ws + c->dyp_off);
        q.argmax = (const unsigned char*)(ws + b.argmax_off);
        q.y = (const bf16_t*)(ws + c->act_off[1]);
        q.dw = grads + b.grad_w;
        q.db = grads + b.grad_b;
        q.n_windows = n;
        RGP_TRY(ensure_dyn_smem((const void*)conv1a_wgrad_bf16_kernel, W1_SMEM));
        int n_cu_w1 = 0;
        RGP_TRY(device_cu_count(&n_cu_w1));
        conv1a_wgrad_bf16_kernel<<<2 * n_cu_w1, 512, W1_SMEM, s>>>(q);   // two blocks per CU (119 VGPRs, 2 x 81 792 B of LDS): more fetches in flight
        RGP_HIP(hipGetLastError());
      } else if (i == 0) {
        p.dW = (float*)(ws + c->dw1_off);
        RGP_HIP(hipMemsetAsync(p.dW, 0, (size_t)p.nk * Elem<T>::BKE * 64 * 4, s));
        RGP_TRY((launch_wgrad<T, G0>(p, s)));
        conv1a_unpack_grad_kernel<<<(27 * 3 * 64 + 255) / 256, 256, 0, s>>>(p.dW, grads + b.grad_w);
        RGP_HIP(hipGetLastError());
      } else if (wg_patch) {
        // wgrad_patch.hip.h: conv2a 5.2 -> 4.0 ms, conv3a 2.3 -> 2.3, conv3b 4.8 -> 4.3 ms per 256 windows against
        // wgrad_kernel; round 5: conv4a / conv4b on its window-pair form, and the un-pooled layers' bias gradient
        // (conv3a, conv4a) on the kernel's spare MFMA slot instead of a colsum pass over the gradient image
        RGP_TRY(run_wgrad_patch(c, i, n, grads + b.grad_w, pooled(i) ? nullptr : grads + b.grad_b, s));
      } else {
        p.dW = grads + b.grad_w;
        RGP_TRY((launch_wgrad<T, 1>(p, s)));
      }
    }
    // layer i's gradient slice (filter here; bias here or, pooled layers, in the un-pool of the step before) is queued
    if (c->grad_ev_made) RGP_HIP(hipEventRecord(c->grad_ev[i], s));
    s = s_in;
    if (i == 0) break;
    // gradient w.r.t. the layer input = pooled (or plain) output of layer i-1
    const C3dBwdLayer& lo = c->B[i - 1];
    IgemmParams p = make_params(b.dg, ws + b.dypre_off, ws, n);
    p.tile128 = c->tile128();
    if (pooled(i - 1)) {
      if (sizeof(T) == 2 && c->use_patch() && (i == 1 || i == 2 || i == 4 || i == 6) && b.dg.chunk_major == 64 && dev_knob("RGP_DGPATCH", 1)) {
        RGP_TRY(run_conv_patch_dgrad_bf16(c, i, n, s));      // conv_patch.hip.h, dense output
      } else {
        EpiParams e = make_epi(b.dg, ws + c->dyp_off, ws);
        RGP_TRY((launch_igemm<T, 1, 1, EpiStore<T, false, false>>(p, e, s)));
      }
      const C3dLayerSpec& ll = kLayers[i - 1];
      const int PR = (ll.D / ll.pd) * (ll.H / ll.ph) * (ll.H / ll.ph);
      const long long rows = (long long)n * PR;
      // bf16 conv1a consumes the pooled gradient directly (conv1a_wgrad.hip.h): its 6.4 MB-per-window dY image is
      // only built on demand by rgp_c3d_read_grad_image
      const bool rows8 = sizeof(T) == 2 && ll.pd == 2 && ll.ph == 2 && (ll.cout == 128 || ll.cout == 256 || ll.cout == 512) &&
                         (ll.cout != 128 || (ll.H / 2) % 2 == 0) && dev_knob("RGP_UNPOOL_ROWS", 1);
      if (rows8 && dev_knob("RGP_UNPOOL_SKIP", 0)) {
        // dev pricing (timing only): no un-pool launch -- the consumers read the gradient image an earlier step left
      } else if (rows8) {
        const int Hp = ll.H + 2;
        const int q_dz = Hp * Hp * ll.cout, q_dy = Hp * ll.cout;
        const bf16_t* dyp = (const bf16_t*)(ws + c->dyp_off);
        const unsigned char* am = (const unsigned char*)(ws + lo.argmax_off);
        const bf16_t* yf = (const bf16_t*)(ws + c->act_off[i]);
        const int* ytab = (const int*)(ws + c->unpad_off[i - 1]);
        bf16_t* out = (bf16_t*)(ws + lo.dypre_off);
        const int* wtab = (const int*)(ws + lo.win_tab_off);
        if (ll.cout == 128) RGP_TRY(launch_unpool8_rows<128>(dyp, am, yf, ytab, c->act_stride[i], out, wtab, lo.dypre_stride, PR, q_dz, q_dy, n, grads + lo.grad_b, s));
        else if (ll.cout == 256) RGP_TRY(launch_unpool8_rows<256>(dyp, am, yf, ytab, c->act_stride[i], out, wtab, lo.dypre_stride, PR, q_dz, q_dy, n, grads + lo.grad_b, s));
        else RGP_TRY(launch_unpool8_rows<512>(dyp, am, yf, ytab, c->act_stride[i], out, wtab, lo.dypre_stride, PR, q_dz, q_dy, n, grads + lo.grad_b, s));
      } else if (!(i == 1 && sizeof(T) == 2)) unpool_kernel<T><<<blocks_for(rows, 256 / (ll.cout / 8) * 8), 256, 0, s>>>(
          (const T*)(ws + c->dyp_off), (const unsigned char*)(ws + lo.argmax_off), (const T*)(ws + c->act_off[i]),
          (const int*)(ws + c->unpad_off[i - 1]), c->act_stride[i], (T*)(ws + lo.dypre_off), (const int*)(ws + lo.win_tab_off),
          (const int*)(ws + lo.q_off_off), lo.dypre_stride, PR, ll.cout, ll.pd * ll.ph * ll.ph, rows, grads + lo.grad_b);
      RGP_HIP(hipGetLastError());
    } else if (sizeof(T) == 2 && c->use_patch() && (i == 3 || i == 5 || i == 7) && b.dg.chunk_major == 64 && dev_knob("RGP_DGPATCH", 1)) {
      RGP_TRY(run_conv_patch_dgrad_bf16(c, i, n, s));        // conv_patch.hip.h / conv_patch14.hip.h
    } else {
      EpiParams e = make_epi(b.dg, ws + lo.dypre_off, ws);
      e.mask = ws + c->act_off[i];
      RGP_TRY((launch_igemm<T, 1, 1, EpiStoreMask<T>>(p, e, s)));
    }
  }
  if (sw != s_in) RGP_TRY(c->side.join(s_in));
  return RGP_OK;
}

}  // namespace

int c3d_bwd_plan(rgp_c3d* c, Arena& a) {
  const int dtype = c->dtype, es = esize(dtype);
  bool ok = true;
  size_t dyp_elems = 0, goff = 0;
  for (int i = 0; i < 8; ++i) {
    const C3dLayerSpec& l = kLayers[i];
    C3dBwdLayer& b = c->B[i];
    const int D = l.D, H = l.H, W = l.H, Hp = H + 2, Wp = W + 2, Co = l.cout;
    const int Cx = i == 0 ? 4 : l.cin, Wpx = i == 0 ? W + 4 : W + 2;
    b.dypre_stride = (long long)(D + 2) * Hp * Wp * Co;
    for (int z = 0; z < D; ++z) for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
      b.x_tab.push_back(((z * Hp + y) * Wpx + x) * Cx);
      b.y_tab.push_back((((z + 1) * Hp + y + 1) * Wp + x + 1) * Co);
    }
    if (pooled(i)) {
      const int Do = D / l.pd, Ho = H / l.ph, Wo = W / l.ph;
      for (int zo = 0; zo < Do; ++zo) for (int yo = 0; yo < Ho; ++yo) for (int xo = 0; xo < Wo; ++xo)
        b.win_tab.push_back((((zo * l.pd + 1) * Hp + yo * l.ph + 1) * Wp + xo * l.ph + 1) * Co);
      for (int dz = 0; dz < l.pd; ++dz) for (int dy = 0; dy < l.ph; ++dy) for (int dx = 0; dx < l.ph; ++dx)
        b.q_off.push_back(((dz * Hp + dy) * Wp + dx) * Co);
      b.argmax_off = a.take((size_t)c->max_windows * Do * Ho * Wo * Co);
      dyp_elems = std::max(dyp_elems, (size_t)Do * Ho * Wo * Co);
    }
    if (i >= 1) {
      ConvDesc& d = b.dg;
      d.Mw = D * H * W;
      d.N = l.cin;
      d.in_img_stride = b.dypre_stride;
      std::vector<int> tapoff, fidx;
      for (int z = 0; z < D; ++z) for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) d.in_tab.push_back(((z * Hp + y) * Wp + x) * Co);
      for (int kz = 0; kz < 3; ++kz) for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
        tapoff.push_back(((kz * Hp + ky) * Wp + kx) * Co);
        fidx.push_back(26 - ((kz * 3 + ky) * 3 + kx));        // rot180
      }
      ok &= build_k_schedule(d, tapoff, fidx, Co, dtype);
      d.s_tap = (long long)l.cin * l.cout; d.s_n = l.cout; d.s_c = 1;    // W[tap][n = cin][c = cout]
      const bool cm = dev_knob("RGP_KORDER", 1) != 0;
      if (cm) make_chunk_major(d, dtype);                               // same L2 argument as the forward convs
      if (pooled(i - 1)) {
        d.out_img_stride = (long long)d.Mw * l.cin;
        for (int m = 0; m < d.Mw; ++m) d.out_tab.push_back(m * l.cin);
      } else {
        d.out_img_stride = c->B[i - 1].dypre_stride;
        d.out_tab = c->B[i - 1].y_tab;
      }
      d.reserve(a, dtype);
    }
    b.x_tab_off = a.take(b.x_tab.size() * 4);
    b.y_tab_off = a.take(b.y_tab.size() * 4);
    b.win_tab_off = a.take(b.win_tab.size() * 4 + 4);
    b.q_off_off = a.take(b.q_off.size() * 4 + 4);
    b.dypre_off = a.take((size_t)c->max_windows * b.dypre_stride * es);
    b.grad_w = goff; goff += (size_t)27 * l.cin * l.cout;
    b.grad_b = goff; goff += l.cout;
  }
  c->n_params = goff;
  c->dyp_off = a.take((size_t)c->max_windows * dyp_elems * es);
  c->dw1_off = a.take((size_t)c->L[0].nk * bke(dtype) * 64 * 4);
  if (!ok) return set_err(RGP_EINVAL, "rgp_c3d_create: backward K schedule failed");
  return RGP_OK;
}

int c3d_bwd_upload(rgp_c3d* c, hipStream_t s) {
  auto up = [&](const std::vector<int>& t, size_t off) -> int {
    if (!t.empty()) RGP_HIP(hipMemcpyAsync(c->ws + off, t.data(), t.size() * 4, hipMemcpyHostToDevice, s));
    return RGP_OK;
  };
  for (int i = 0; i < 8; ++i) {
    C3dBwdLayer& b = c->B[i];
    if (i >= 1) RGP_TRY(upload_desc(b.dg, c->ws, s));
    RGP_TRY(up(b.x_tab, b.x_tab_off)); RGP_TRY(up(b.y_tab, b.y_tab_off));
    RGP_TRY(up(b.win_tab, b.win_tab_off)); RGP_TRY(up(b.q_off, b.q_off_off));
  }
  return RGP_OK;
}

template <typename T>
static int c3d_bwd_pack_t(rgp_c3d* c, const rgp_c3d_weights* w, hipStream_t s) {
  PackBatch<T> pk(c->ws, s);                       // the seven rotated input-gradient filters in one launch
  for (int i = 1; i < 8; ++i) RGP_TRY(pk.add(c->B[i].dg, w->w[i], kLayers[i].cin, 0));
  return pk.flush();
}

int c3d_bwd_pack(rgp_c3d* c, const rgp_c3d_weights* w, hipStream_t s) {
  return c->dtype == RGP_BF16 ? c3d_bwd_pack_t<bf16_t>(c, w, s) : c3d_bwd_pack_t<float>(c, w, s);
}

extern "C" {

size_t rgp_c3d_param_elems(const rgp_c3d_t* c) {
  if (!c) return 0;
  size_t n = 0;
  for (int i = 0; i < 8; ++i) n += (size_t)27 * kLayers[i].cin * kLayers[i].cout + kLayers[i].cout;
  return n;
}

size_t rgp_c3d_param_offset(const rgp_c3d_t* c, int layer, int is_bias) {
  size_t off = 0;
  for (int i = 0; i < 8 && i <= layer; ++i) {
    if (i == layer) return off + (is_bias ? (size_t)27 * kLayers[i].cin * kLayers[i].cout : 0);
    off += (size_t)27 * kLayers[i].cin * kLayers[i].cout + kLayers[i].cout;
  }
  return off;
}

int rgp_c3d_wait_layer_grads(rgp_c3d_t* c, int layer, rgp_stream_t waiting_stream) {
  RGP_REQUIRE(c && layer >= 0 && layer <= 7, "rgp_c3d_wait_layer_grads: bad arguments");
  if (!c->save || !c->grad_ev_made) return set_err(RGP_ESTATE, "rgp_c3d_wait_layer_grads: no backward has run on this plan");
  RGP_HIP(hipStreamWaitEvent((hipStream_t)waiting_stream, c->grad_ev[layer], 0));
  return RGP_OK;
}

int rgp_c3d_read_grad_image(rgp_c3d_t* c, int layer, int n_windows, float* dst, rgp_stream_t stream) {
  RGP_REQUIRE(c && c->ws && c->save && dst && layer >= 0 && layer <= 7 && n_windows > 0 && n_windows <= c->max_windows,
              "rgp_c3d_read_grad_image: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const C3dBwdLayer& b = c->B[layer];
  const int rows = (int)b.y_tab.size(), C = kLayers[layer].cout;
  const long long total = (long long)n_windows * rows * C;
  const int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
  const int* tab = (const int*)(c->ws + b.y_tab_off);
  if (layer == 0 && c->dtype == RGP_BF16) {
    // the backward pass kept only the pooled gradient of conv1a (still in the dyp buffer: pool1 is the last pooled
    // layer it visits); expand it now.  The bias sums of this pass go to a scratch area.
    const C3dLayerSpec& ll = kLayers[0];
    const int PR = (ll.D / ll.pd) * (ll.H / ll.ph) * (ll.H / ll.ph);
    const long long prow = (long long)n_windows * PR;
    RGP_HIP(hipMemsetAsync(c->ws + b.dypre_off, 0, (size_t)n_windows * b.dypre_stride * 2, s));
    unpool_kernel<bf16_t><<<(int)std::min<long long>((prow + 255) / 256, 4096), 256, 0, s>>>(
        (const bf16_t*)(c->ws + c->dyp_off), (const unsigned char*)(c->ws + b.argmax_off), (const bf16_t*)(c->ws + c->act_off[1]),
        (const int*)(c->ws + c->unpad_off[0]), c->act_stride[1], (bf16_t*)(c->ws + b.dypre_off), (const int*)(c->ws + b.win_tab_off),
        (const int*)(c->ws + b.q_off_off), b.dypre_stride, PR, ll.cout, ll.pd * ll.ph * ll.ph, prow, (float*)(c->ws + c->dw1_off));
    RGP_HIP(hipGetLastError());
  }
  if (c->dtype == RGP_BF16)
    unpad_kernel<bf16_t><<<blocks, 256, 0, s>>>((const bf16_t*)(c->ws + b.dypre_off), dst, tab, rows, C, b.dypre_stride, total);
  else
    unpad_kernel<float><<<blocks, 256, 0, s>>>((const float*)(c->ws + b.dypre_off), dst, tab, rows, C, b.dypre_stride, total);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

int rgp_c3d_backward(rgp_c3d_t* c, const float* d_features, const float* d_rows, int n_windows, float* grads,
                     rgp_stream_t stream) {
  RGP_REQUIRE(c && grads && (d_features != nullptr) != (d_rows != nullptr),
              "rgp_c3d_backward: need grads and exactly one of d_features / d_rows");
  if (!c->save) return set_err(RGP_ESTATE, "rgp_c3d_backward: plan was not created with save_for_backward");
  if (!c->ws || !c->weights_set) return set_err(RGP_ESTATE, "rgp_c3d_backward: workspace/weights not set");
  if (n_windows != c->last_n || n_windows <= 0)
    return set_err(RGP_ESTATE, "rgp_c3d_backward: n_windows %d != windows of the last forward chunk (%d); forward at most "
                   "max_windows windows, then call backward", n_windows, c->last_n);
  hipStream_t s = (hipStream_t)stream;
  if (!c->grad_ev_made) {
    for (int i = 0; i < 8; ++i) RGP_HIP(hipEventCreateWithFlags(&c->grad_ev[i], hipEventDisableTiming));
    c->grad_ev_made = true;
  }
  return c->dtype == RGP_BF16 ? backward_impl<bf16_t>(c, d_features, d_rows, grads, s)
                              : backward_impl<float>(c, d_features, d_rows, grads, s);
}

}  // extern "C"
