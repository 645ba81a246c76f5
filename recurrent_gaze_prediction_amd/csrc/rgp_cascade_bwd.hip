// librgp_hip.so: backward of the two-level cascade (BASELINE config 5) under its l2 loss
// (gaze_grcn_cascade.py:428-441; tf.gradients of every non-ShallowNet variable, base.py:264-281).
//
//   d maps = (maps - gt) / (B*T)
//   fc2, fc1     maxout + ReLU routing from the masks the forward epilogue recorded; filter gradients with
//                wgrad_kernel (rows = frames), input gradients as plain GEMMs over the transposed filters
//   top cell     BPTT over T steps on 49x49 (gaze_grcn_cascade.py:95-129 differentiated): 5x5 dgrad convs of
//                the recurrent filters per step; all filter gradients hoisted out of the loop (wgrad_kernel
//                over the kept operand images of every step)
//   upsampling   stride-7 transposed conv: input gradient = stride-7 11x11 conv of the map gradient,
//                filter gradient = wgrad_kernel with stride-7 row origins
//   bottom cell  rgp_grcn_backward_from_states on the sub-plan (+ rgp_grcn_backward_input for the conv stack)
#include <algorithm>

#include "rgp_cascade_plan.h"
#include "wgrad_launch.h"

using namespace rgp;

namespace {

constexpr long long kImg = (long long)kHp * kHp;   // pixels of a padded 53x53 image

// dz[f+1][j] / dz[f+1][2401+j] = gradient of maxout unit j routed to the half that won (mask 1 / 2), 0 if
// ReLU-gated.  The unit's gradient is (a - b) * scale (loss layer: maps - gt) or a * scale (b == null).
template <typename T>
__global__ __launch_bounds__(256) void maxout_bwd_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b,
                                                         float scale, const unsigned char* __restrict__ mask, T* __restrict__ dz,
                                                         long long total) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int j = (int)(i % 2401);
    const long long f = i / 2401;
    const float v = (a[f * lda + j] - (b ? b[i] : 0.f)) * scale;
    const unsigned char m = mask[i];
    T* row = dz + (f + 1) * kN2;
    row[j] = Elem<T>::to(m == 1 ? v : 0.f);
    row[2401 + j] = Elem<T>::to(m == 2 ? v : 0.f);
  }
}

// bias gradient of an FC: db[col] = sum_f dz[f+1][col].  Block = 32 columns x 8 row lanes (a thread per column looping
// over all F rows -- 19 blocks, F dependent loads each -- took 0.19 ms at 512 frames); launch with (4802 + 31) / 32 blocks.
template <typename T>
__global__ __launch_bounds__(256) void fc_bias_grad_kernel(const T* __restrict__ dz, int F, float* __restrict__ db) {
  __shared__ float red[8][33];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + cl;
  float a = 0.f;
  if (col < 4802)
    for (int f = rl; f < F; f += 8) a += Elem<T>::from(dz[(long long)(f + 1) * kN2 + col]);
  red[rl][cl] = a;
  __syncthreads();
  if (rl == 0 && col < 4802) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) t += red[r][cl];
    db[col] = t;
  }
}

// BPTT step of the top cell, part 1:  dh = d fcin (frame b*T+t) + carry
//   du = dh (h_prev - c), dc = dh (1 - u), carry = dh u;  dz_pre = du u (1-u), dc_pre = dc (1 - c^2)
// -> columns [0,16) and [32,48) of dxpre_pad (frame b*T+t) and the padded dc_pre image of this step.
template <typename T>
__global__ __launch_bounds__(256) void top_bwd1_kernel(const float* __restrict__ dfcin, int ldf, float* __restrict__ carry,
                                                       const float* __restrict__ h_prev, const float* __restrict__ u,
                                                       const float* __restrict__ c, T* __restrict__ dxpre_pad,
                                                       T* __restrict__ dcp_pad, int B, int T_, int t, int first) {
  const long long total = (long long)B * 2401 * kSt;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ch = (int)(i % kSt);
    const int p = (int)((i / kSt) % 2401);
    const int b = (int)(i / ((long long)kSt * 2401));
    const long long f = (long long)b * T_ + t;
    float dh = first ? 0.f : carry[i];
    if (ch < 3) dh += dfcin[f * ldf + p * 3 + ch];
    const float uu = u[i], cc = c[i];
    const float du = dh * (h_prev[i] - cc), dc = dh * (1.f - uu);
    carry[i] = dh * uu;
    const float dcp = dc * (1.f - cc * cc), dzp = du * uu * (1.f - uu);
    const int pos = ((p / 49 + 2) * kHp + p % 49 + 2);
    T* row = dxpre_pad + (f * kImg + pos) * 64;
    row[ch] = Elem<T>::to(dzp);
    row[2 * kSt + ch] = Elem<T>::to(dcp);
    dcp_pad[((long long)b * kImg + pos) * kSt + ch] = Elem<T>::to(dcp);
  }
}

// part 2: d(r.h) from the U dgrad conv -> dr_pre = drh h_prev r (1-r), carry += drh r; writes column block
// [16,32) of dxpre_pad and the padded [dz_pre | dr_pre] image (operand of the Uz|Ur dgrad conv).
template <typename T>
__global__ __launch_bounds__(256) void top_bwd2_kernel(const float* __restrict__ drh, float* __restrict__ carry,
                                                       const float* __restrict__ h_prev, const float* __restrict__ r,
                                                       T* __restrict__ dxpre_pad, T* __restrict__ dzr_pad, int B, int T_, int t) {
  const long long total = (long long)B * 2401 * kSt;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ch = (int)(i % kSt);
    const int p = (int)((i / kSt) % 2401);
    const int b = (int)(i / ((long long)kSt * 2401));
    const long long f = (long long)b * T_ + t;
    const float d = drh[i], rr = r[i];
    const float drp = d * h_prev[i] * rr * (1.f - rr);
    carry[i] += d * rr;
    const int pos = ((p / 49 + 2) * kHp + p % 49 + 2);
    T* row = dxpre_pad + (f * kImg + pos) * 64;
    const T drp_t = Elem<T>::to(drp);
    row[kSt + ch] = drp_t;
    T* img = dzr_pad + ((long long)b * kImg + pos) * 2 * kSt;
    img[ch] = row[ch];          // dz_pre written by part 1
    img[kSt + ch] = drp_t;
  }
}

// packed filter gradients of the top cell -> the TF filters.
//   x part: dwx [tap*128 + c][g*16 + n] -> W_g [tap][c < 65][n < 3]
//   h part: dwh [tap*16 + ci][g*16 + n] (g = z, r) -> U_g [tap][ci < 3][n < 3];  dwu [tap*16 + ci][n] -> U
__global__ void top_unpack_grads_kernel(const float* __restrict__ dwx, const float* __restrict__ dwh, const float* __restrict__ dwu,
                                        float* __restrict__ Wz, float* __restrict__ Wr, float* __restrict__ W, float* __restrict__ Uz,
                                        float* __restrict__ Ur, float* __restrict__ U) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 25 * 65 * 3) {
    const int n = i % 3, c = (i / 3) % 65, tap = i / 195;
    const float* row = dwx + (long long)(tap * kCt + c) * 48;
    Wz[i] = row[n]; Wr[i] = row[kSt + n]; W[i] = row[2 * kSt + n];
  }
  if (i < 25 * 3 * 3) {
    const int n = i % 3, ci = (i / 3) % 3, tap = i / 9;
    Uz[i] = dwh[(long long)(tap * kSt + ci) * 32 + n];
    Ur[i] = dwh[(long long)(tap * kSt + ci) * 32 + kSt + n];
    U[i] = dwu[(long long)(tap * kSt + ci) * 16 + n];
  }
}

template <typename T>
int backward_impl(rgp_cascade* g, const float* maps, const float* gt, const rgp_cascade_weights* gr, float* d_rows, hipStream_t s) {
  char* ws = g->ws;
  const int B = g->B, T_ = g->T, F = g->F;
  constexpr int BKE = Elem<T>::BKE;
  constexpr int G16 = BKE / kSt;                       // 16-channel taps per 128-byte chunk: 4 (bf16) / 2 (fp32)
  constexpr int G32 = BKE / (2 * kSt);                 // 2 / 1
  auto Tp = [&](size_t off) { return (T*)(ws + off); };
  auto Fp = [&](size_t off) { return (float*)(ws + off); };
  auto nblk = [](long long n) { return (int)std::min<long long>((n + 255) / 256, 8192); };
  const size_t st = (size_t)B * 2401 * kSt;

  // ---- fully connected read-out
  maxout_bwd_kernel<T><<<nblk((long long)F * 2401), 256, 0, s>>>(maps, 2401, gt, 1.0f / (float)F, (const unsigned char*)(ws + g->mask2),
                                                                 Tp(g->dz2), (long long)F * 2401);
  fc_bias_grad_kernel<T><<<(4802 + 31) / 32, 256, 0, s>>>(Tp(g->dz2), F, (float*)gr->fc2_b);
  RGP_HIP(hipGetLastError());
  // weight gradients run on the plan's side stream (rgp_cascade_plan.h): fork(i) behind the kernels that complete their operands
  hipStream_t sw = s;
  auto fc_wgrad = [&](const void* X, int ldx, const ConvDesc& fwd, size_t dz, float* dW, int k_valid, hipStream_t s) -> int {
    RGP_HIP(hipMemsetAsync(dW, 0, (size_t)k_valid * 4802 * 4, s));
    WgradParams p;
    memset(&p, 0, sizeof(p));
    p.X = X; p.dY = ws + dz; p.dW = dW;
    wgrad_grid(p, 1, 1, F);
    p.x_sx = ldx; p.y_sx = kN2; p.y_org = kN2;
    p.koff = (const int*)(ws + fwd.koff_off);
    p.M = F; p.N = 4802; p.nk = fwd.nk; p.ldw = 4802; p.k_valid = k_valid;
    return launch_wgrad<T, 1>(p, s);
  };
  RGP_TRY(g->fork(s, 0, &sw));
  RGP_TRY(fc_wgrad(ws + g->mo1, g->K2, g->fc2, g->dz2, (float*)gr->fc2_w, 2401, sw));
  {
    IgemmParams p = make_params(g->b_fc2, Tp(g->dz2) + kN2, ws, F);
    EpiParams e = make_epi(g->b_fc2, Fp(g->dmo1), ws);
    RGP_TRY((launch_igemm<T, 1, 1, EpiStore<float, false, false>>(p, e, s)));
  }
  // fc1's dropout (when on): the winning half of a live unit was kept, so its gradient is d out / keep
  maxout_bwd_kernel<T><<<nblk((long long)F * 2401), 256, 0, s>>>(Fp(g->dmo1), g->K2, nullptr, 1.0f / g->drop_keep, (const unsigned char*)(ws + g->mask1),
                                                                 Tp(g->dz1), (long long)F * 2401);
  fc_bias_grad_kernel<T><<<(4802 + 31) / 32, 256, 0, s>>>(Tp(g->dz1), F, (float*)gr->fc1_b);
  RGP_HIP(hipGetLastError());
  RGP_TRY(g->fork(s, 1, &sw));
  RGP_TRY(fc_wgrad(ws + g->fcin, g->Kfc, g->fc1, g->dz1, (float*)gr->fc1_w, 7203, sw));
  {
    IgemmParams p = make_params(g->b_fc1, Tp(g->dz1) + kN2, ws, F);
    EpiParams e = make_epi(g->b_fc1, Fp(g->dfcin), ws);
    RGP_TRY((launch_igemm<T, 1, 1, EpiStore<float, false, false>>(p, e, s)));
  }

  // ---- top cell BPTT -> input gradient -> stride-7 transposed convolution -> bottom cell BPTT.  Three chains one time step
  // apart (rgp_cascade_plan.h) when the plan has its streams: each chain is per-step launches on a fraction of the CUs.
  const bool pipe = sw != s && dev_knob("RGP_CASCADE_PIPE", 1) && g->bottom->seq_groups <= 0 && g->pipe_ok(s, T_);
  hipStream_t sa = s, sb = s;                                   // top cell BPTT / its input gradient down to the bottom states
  if (pipe) {
    sa = g->side2; sb = g->side3;
    RGP_HIP(hipEventRecord(g->ev_join2, s));                     // d fc input is complete; nothing of an earlier call is in flight
    RGP_HIP(hipStreamWaitEvent(sa, g->ev_join2, 0));
    RGP_HIP(hipStreamWaitEvent(sb, g->ev_join2, 0));
    RGP_HIP(hipMemsetAsync(Fp(g->d_hbn), 0, (size_t)F * 49 * 256 * 4, sb));    // the split-K transposed-conv gradient adds into it
  }
  const long long img64 = kImg * 64;
  // gradient w.r.t. the upsampled maps (input channels 0..63 of the top cell; channel 64 is the frozen saliency), then w.r.t.
  // the bottom states: d y[i,j,c] = sum_{a,b,o} dUp[7i+a-2, 7j+b-2, o] F[a,b,o,c].  t < 0: all frames at once
  auto feed_back = [&](int t, hipStream_t q) -> int {
    const int n_img = t < 0 ? F : B;
    const long long off = t < 0 ? 0 : t * img64;
    IgemmParams p = make_params(g->b_tx, Tp(g->dxpre_pad) + off, ws, n_img);
    EpiParams e = make_epi(g->b_tx, Tp(g->dup_pad) + off, ws);
    if (t >= 0) { p.in_img_stride *= T_; e.out_img_stride *= T_; }
    RGP_TRY((launch_igemm<T, 1, 1, EpiStore<T, false, false>>(p, e, q)));
    if (t < 0) return RGP_OK;                                    // (the hoisted form runs the filter gradient in between: below)
    IgemmParams pu = make_params(g->b_up, Tp(g->dup_pad) + off, ws, n_img);
    EpiParams eu = make_epi(g->b_up, Fp(g->d_hbn) + (long long)t * 49 * 256, ws);
    pu.in_img_stride *= T_; eu.out_img_stride *= T_;
    // B x 49 rows, K = 121 x 64: a dozen tiles walking 121 K-chunks each -- split K four ways, float atomics
    return launch_igemm<T, 1, 1, EpiAtomicAddF32>(pu, eu, q, 4);
  };
  // The top cell's filter gradients over the steps [t0, t0 + k): image = clip b, z = step.  Partial sums are added into the
  // packed scratch with atomics, so chunks of steps accumulate (first: the scratch is cleared on the same stream).
  constexpr int kWgChunk = 7;
  auto top_wgrads = [&](int t0, int k, hipStream_t q, bool first) -> int {
    const long long img16 = kImg * kSt, imgx = kImg * kCt;
    if (first) {
      RGP_HIP(hipMemsetAsync(Fp(g->dwx), 0, (size_t)g->xtop.nk * BKE * 48 * 4, q));
      RGP_HIP(hipMemsetAsync(Fp(g->dwh), 0, (size_t)g->zr.nk * BKE * 32 * 4, q));
      RGP_HIP(hipMemsetAsync(Fp(g->dwu), 0, (size_t)g->c.nk * BKE * 16 * 4, q));
    }
    WgradParams p;
    memset(&p, 0, sizeof(p));
    p.dY = Tp(g->dxpre_pad) + (long long)t0 * img64;
    p.y_sx = 64; p.y_sy = kHp * 64; p.y_sz = (int)img64; p.y_org = (2 * kHp + 2) * 64;
    p.y_img_stride = (long long)T_ * img64;
    wgrad_grid(p, k, 49, 49);
    p.M = (long long)B * k * 2401;
    // x part: X = the top cell's input images (128 channels), all three gates (48 columns)
    p.X = Tp(g->xtopbuf) + (long long)t0 * imgx; p.dW = Fp(g->dwx);
    p.x_sx = kCt; p.x_sy = kHp * kCt; p.x_sz = (int)imgx; p.x_img_stride = (long long)T_ * imgx;
    p.koff = (const int*)(ws + g->xtop.koff_off);
    p.N = 48; p.nk = g->xtop.nk; p.ldw = 48; p.k_valid = g->xtop.nk * BKE;
    RGP_TRY((launch_wgrad<T, 1>(p, q)));
    // h part, gates z and r: X = h_{t-1} images (slot (b, t))
    p.X = Tp(g->hp_all) + (long long)t0 * img16; p.dW = Fp(g->dwh);
    p.x_sx = kSt; p.x_sy = kHp * kSt; p.x_sz = (int)img16; p.x_img_stride = (long long)(T_ + 1) * img16;
    p.koff = (const int*)(ws + g->zr.koff_off);
    p.N = 32; p.nk = g->zr.nk; p.ldw = 32; p.k_valid = g->zr.nk * BKE;
    RGP_TRY((launch_wgrad<T, G16>(p, q)));
    // h part, candidate: X = r (.) h_{t-1} images, gradient columns [32, 48)
    p.X = Tp(g->rhp_all) + (long long)t0 * img16; p.dW = Fp(g->dwu);
    p.x_img_stride = (long long)T_ * img16;
    p.y_org = (2 * kHp + 2) * 64 + 2 * kSt;
    p.koff = (const int*)(ws + g->c.koff_off);
    p.N = 16; p.nk = g->c.nk; p.ldw = 16; p.k_valid = g->c.nk * BKE;
    return launch_wgrad<T, G16>(p, q);
  };
  const float* hall = Fp(g->hall_t);
  for (int t = T_ - 1; t >= 0; --t) {
    const float* h_prev = hall + (size_t)t * st;
    hipStream_t s = sa;
    top_bwd1_kernel<T><<<nblk((long long)st), 256, 0, s>>>(Fp(g->dfcin), g->Kfc, Fp(g->dh_carry), h_prev, Fp(g->uall) + (size_t)t * st,
                                                          Fp(g->call) + (size_t)t * st, Tp(g->dxpre_pad), Tp(g->dcp_pad), B, T_, t,
                                                          t == T_ - 1);
    RGP_HIP(hipGetLastError());
    {
      IgemmParams p = make_params(g->b_tc, Tp(g->dcp_pad), ws, B);
      EpiParams e = make_epi(g->b_tc, Fp(g->drh), ws);
      RGP_TRY((launch_igemm<T, G16, 1, EpiStore<float, false, false>>(p, e, s)));
    }
    top_bwd2_kernel<T><<<nblk((long long)st), 256, 0, s>>>(Fp(g->drh), Fp(g->dh_carry), h_prev, Fp(g->rall) + (size_t)t * st,
                                                          Tp(g->dxpre_pad), Tp(g->dzr_pad), B, T_, t);
    RGP_HIP(hipGetLastError());
    {
      IgemmParams p = make_params(g->b_tzr, Tp(g->dzr_pad), ws, B);
      EpiParams e = make_epi(g->b_tzr, Fp(g->dh_carry), ws);
      RGP_TRY((launch_igemm<T, G32, 1, EpiAccumF32>(p, e, s)));
    }
    if (pipe) {
      RGP_HIP(hipEventRecord(g->ev_b[t], sa));                  // frame (b, t) of dxpre_pad is complete
      RGP_HIP(hipStreamWaitEvent(sb, g->ev_b[t], 0));
      RGP_TRY(feed_back(t, sb));
      RGP_HIP(hipEventRecord(g->ev_x[t], sb));                  // ... and of d_hbn: the bottom cell's step t may run
      // the top cell's filter gradients of the steps [t, t + chunk) just differentiated: on the weight-gradient stream while
      // the BPTT goes on (waiting for its last step with all of them put 1 ms of full-chip launches behind the chains)
      if (t % kWgChunk == 0) {
        RGP_HIP(hipStreamWaitEvent(g->side, g->ev_b[t], 0));
        RGP_TRY(top_wgrads(t, std::min(kWgChunk, T_ - t), g->side, t + kWgChunk >= T_));
      }
    }
  }
  if (!pipe) RGP_TRY(feed_back(-1, s));
  // ---- top cell: filter gradients (one-chain form: all steps at once, on the side stream beside the chain below)
  if (pipe) sw = g->side;
  else {
    RGP_TRY(g->fork(s, 2, &sw));
    RGP_TRY(top_wgrads(0, T_, sw, true));
  }
  top_unpack_grads_kernel<<<(25 * 65 * 3 + 255) / 256, 256, 0, sw>>>(Fp(g->dwx), Fp(g->dwh), Fp(g->dwu), (float*)gr->top_Wz,
                                                                    (float*)gr->top_Wr, (float*)gr->top_W, (float*)gr->top_Uz,
                                                                    (float*)gr->top_Ur, (float*)gr->top_U);
  RGP_HIP(hipGetLastError());
  // ---- stride-7 transposed conv: filter gradient dF[a,b,o,c] = sum dUp[7i+a-2, 7j+b-2, o] y[i,j,c] (side stream); hoisted
  // form: its input gradient
  {
    if (pipe) { RGP_HIP(hipStreamWaitEvent(g->side, g->ev_x[0], 0)); sw = g->side; }      // behind the last frame of dup_pad
    else RGP_TRY(g->fork(s, 3, &sw));
    RGP_HIP(hipMemsetAsync((void*)gr->upsampling_weight, 0, (size_t)121 * 64 * 256 * 4, sw));
    WgradParams p;
    memset(&p, 0, sizeof(p));
    p.X = Tp(g->dup_pad); p.dY = g->bottom->ws + g->bottom->hbn.off; p.dW = (float*)gr->upsampling_weight;
    wgrad_grid(p, 1, 7, 7);
    p.x_sx = 7 * 64; p.x_sy = 7 * kHp * 64; p.x_img_stride = kImg * 64;
    p.y_sx = 256; p.y_sy = 9 * 256; p.y_org = 10 * 256; p.y_img_stride = 81LL * 256;
    p.koff = (const int*)(ws + g->b_up.koff_off);
    p.M = (long long)F * 49; p.N = 256; p.nk = g->b_up.nk; p.ldw = 256; p.k_valid = 121 * 64;
    RGP_TRY((launch_wgrad<T, 1>(p, sw)));
    if (!pipe) {
      IgemmParams q = make_params(g->b_up, Tp(g->dup_pad), ws, F);
      EpiParams e = make_epi(g->b_up, Fp(g->d_hbn), ws);
      RGP_TRY((launch_igemm<T, 1, 1, EpiStore<float, false, false>>(q, e, s)));
    }
  }
  // ---- bottom cell + projection (and the conv stack's input gradient)
  {
    rgp_grcn_weights bg;
    float* sc = Fp(g->scratch_head);
    bg.proj_c3d_W = gr->proj_c3d_W; bg.proj_c3d_b = gr->proj_c3d_b;
    bg.gru_Wz = gr->bottom_Wz; bg.gru_Uz = gr->bottom_Uz; bg.gru_Wr = gr->bottom_Wr; bg.gru_Ur = gr->bottom_Ur;
    bg.gru_W = gr->bottom_W; bg.gru_U = gr->bottom_U;
    bg.bn_gamma = sc; sc += (size_t)T_ * 256;
    bg.bn_beta = sc; sc += (size_t)T_ * 256;
    bg.up_weight1 = sc; sc += 25 * 64 * 256;
    bg.up_weight2 = sc; sc += 25 * 32 * 64;
    bg.up_weight3 = sc; sc += 49 * 12 * 32;
    bg.out_W = sc; sc += 16;
    bg.out_b = sc;
    g->bottom->bwd_step_ev = pipe ? g->ev_x.data() : nullptr;   // step t of its BPTT waits for frame (b, t) of d_hbn
    const int rc = rgp_grcn_backward_from_states(g->bottom, Fp(g->d_hbn), &bg, (rgp_stream_t)s);
    g->bottom->bwd_step_ev = nullptr;
    RGP_TRY(rc);
    if (d_rows) RGP_TRY(rgp_grcn_backward_input(g->bottom, d_rows, (rgp_stream_t)s));
  }
  if (sw != s) RGP_TRY(g->join(s));
  return RGP_OK;
}

}  // namespace

int cascade_bwd_plan(rgp_cascade* g, Arena& a) {
  const int B = g->B, T_ = g->T, F = g->F, dtype = g->dtype, es = esize(dtype);
  const size_t st = (size_t)B * 2401 * kSt;
  bool ok = true;
  for (int y = 0; y < 49; ++y) for (int x = 0; x < 49; ++x) g->tab_pad53_64.push_back(((y + 2) * kHp + x + 2) * 64);
  // 5x5 dgrads of the top cell: correlation of the halo-padded gradient image with the rotated, in/out-swapped filter
  auto dgrad5 = [&](ConvDesc& d, int Cgrad, int N, long long out_ld, bool padded_out) {
    d.Mw = 2401; d.N = N; d.in_img_stride = kImg * Cgrad;
    std::vector<int> tapoff, fidx;
    for (int y = 0; y < 49; ++y) for (int x = 0; x < 49; ++x) {
      d.in_tab.push_back((y * kHp + x) * Cgrad);
      d.out_tab.push_back(padded_out ? ((y + 2) * kHp + x + 2) * (int)out_ld : (y * 49 + x) * (int)out_ld);
    }
    d.out_img_stride = padded_out ? kImg * out_ld : 2401LL * out_ld;
    for (int t = 0; t < 25; ++t) { tapoff.push_back(((t / 5) * kHp + t % 5) * Cgrad); fidx.push_back(24 - t); }
    ok &= build_k_schedule(d, tapoff, fidx, Cgrad, dtype);
  };
  dgrad5(g->b_tc, kSt, kSt, kSt, false);            // d(r.h) = dc_pre (*) rot(U):      U [5,5,ci 3,co 3]
  g->b_tc.cin_src = 3; g->b_tc.s_tap = 9; g->b_tc.s_n = 3; g->b_tc.s_c = 1;
  dgrad5(g->b_tzr, 2 * kSt, kSt, kSt, false);       // carry += [dz_pre|dr_pre] (*) rot([Uz|Ur])
  g->b_tzr.cin_src = 3; g->b_tzr.s_tap = 9; g->b_tzr.s_n = 3; g->b_tzr.s_c = 1;
  dgrad5(g->b_tx, 64, 64, 64, true);                // d up[c < 64] = dxpre (*) rot(W_g[:, :, c, :])
  g->b_tx.cin_src = 3; g->b_tx.s_tap = 65 * 3; g->b_tx.s_n = 3; g->b_tx.s_c = 1;
  {  // d y[i,j,c] = sum_{a,b,o} dUp[7i+a-2, 7j+b-2, o] F[a,b,o,c]   (gaze_grcn_cascade.py:327-333)
    ConvDesc& d = g->b_up;
    d.Mw = 49; d.N = 256; d.in_img_stride = kImg * 64; d.out_img_stride = 49LL * 256;
    std::vector<int> tapoff, fidx;
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 7; ++j) { d.in_tab.push_back((7 * i * kHp + 7 * j) * 64); d.out_tab.push_back((i * 7 + j) * 256); }
    for (int t = 0; t < 121; ++t) { tapoff.push_back(((t / 11) * kHp + t % 11) * 64); fidx.push_back(t); }
    ok &= build_k_schedule(d, tapoff, fidx, 64, dtype);
    d.s_tap = 64LL * 256; d.s_n = 1; d.s_c = 256;
  }
  auto fcT = [&](ConvDesc& d, int N, long long out_ld) {   // d in[f][k] = sum_col dz[f][col] W[k][col]
    d.Mw = 1; d.N = N; d.in_img_stride = kN2; d.out_img_stride = out_ld; d.in_tab = {0}; d.out_tab = {0};
    ok &= build_k_schedule(d, {0}, {0}, kN2, dtype);
    d.cin_src = 4802; d.s_tap = 0; d.s_n = 4802; d.s_c = 1;
  };
  fcT(g->b_fc2, 2401, g->K2);
  fcT(g->b_fc1, 7203, g->Kfc);
  if (!ok) return set_err(RGP_EINVAL, "rgp_cascade_create: backward K schedule failed");
  for (ConvDesc* d : {&g->b_fc2, &g->b_fc1, &g->b_tc, &g->b_tzr, &g->b_tx, &g->b_up}) d->reserve(a, dtype);
  g->o_pad53_64 = a.take(g->tab_pad53_64.size() * 4);
  g->hall_t = a.take((size_t)(T_ + 1) * st * 4);
  g->uall = a.take((size_t)T_ * st * 4);
  g->rall = a.take((size_t)T_ * st * 4);
  g->call = a.take((size_t)T_ * st * 4);
  g->hp_all = a.take((size_t)B * (T_ + 1) * kImg * kSt * es + 4096);
  g->rhp_all = a.take((size_t)F * kImg * kSt * es + 4096);
  g->mask1 = a.take((size_t)F * 2401);
  g->mask2 = a.take((size_t)F * 2401);
  g->dz2 = a.take((size_t)(F + 1) * kN2 * es + 1024);
  g->dz1 = a.take((size_t)(F + 1) * kN2 * es + 1024);
  g->dmo1 = a.take((size_t)F * g->K2 * 4);
  g->dfcin = a.take((size_t)F * g->Kfc * 4);
  g->dh_carry = a.take(st * 4);
  g->drh = a.take(st * 4);
  g->dcp_pad = a.take((size_t)B * kImg * kSt * es + 4096);
  g->dzr_pad = a.take((size_t)B * kImg * 2 * kSt * es + 4096);
  g->dxpre_pad = a.take((size_t)F * kImg * 64 * es + 4096);
  g->dup_pad = a.take((size_t)F * kImg * 64 * es + 4096);
  g->d_hbn = a.take((size_t)F * 49 * 256 * 4);
  g->dwx = a.take((size_t)g->xtop.nk * bke(dtype) * 48 * 4);
  g->dwh = a.take((size_t)g->zr.nk * bke(dtype) * 32 * 4);
  g->dwu = a.take((size_t)g->c.nk * bke(dtype) * 16 * 4);
  g->scratch_head = a.take(((size_t)2 * T_ * 256 + 25 * 64 * 256 + 25 * 32 * 64 + 49 * 12 * 32 + 64) * 4);
  return RGP_OK;
}

int cascade_bwd_upload(rgp_cascade* g, hipStream_t s) {
  for (ConvDesc* d : {&g->b_fc2, &g->b_fc1, &g->b_tc, &g->b_tzr, &g->b_tx, &g->b_up}) RGP_TRY(upload_desc(*d, g->ws, s));
  RGP_HIP(hipMemcpyAsync(g->ws + g->o_pad53_64, g->tab_pad53_64.data(), g->tab_pad53_64.size() * 4, hipMemcpyHostToDevice, s));
  return RGP_OK;
}

template <typename T>
static int pack_impl(rgp_cascade* g, const rgp_cascade_weights* w, hipStream_t s) {
  char* ws = g->ws;
  PackBatch<T> pk(ws, s);                                   // one launch, no memsets (rgp_grcn.hip set_weights_impl)
  RGP_TRY(pk.add(g->b_fc2, w->fc2_w, 2401, 0));
  RGP_TRY(pk.add(g->b_fc1, w->fc1_w, 7203, 0));
  RGP_TRY(pk.add(g->b_tc, w->top_U, 3, 0));
  RGP_TRY(pk.add(g->b_tzr, w->top_Uz, 3, 0, 0, 1));
  RGP_TRY(pk.add(g->b_tzr, w->top_Ur, 3, 0, kSt, 1));
  RGP_TRY(pk.add(g->b_tx, w->top_Wz, 64, 0, 0, 1));
  RGP_TRY(pk.add(g->b_tx, w->top_Wr, 64, 0, kSt, 1));
  RGP_TRY(pk.add(g->b_tx, w->top_W, 64, 0, 2 * kSt, 1));
  RGP_TRY(pk.add(g->b_up, w->upsampling_weight, 256, 0));
  RGP_TRY(pk.flush());
  return RGP_OK;
}

int cascade_bwd_pack(rgp_cascade* g, const rgp_cascade_weights* w, hipStream_t s) {
  return g->dtype == RGP_BF16 ? pack_impl<bf16_t>(g, w, s) : pack_impl<float>(g, w, s);
}

extern "C" {

int rgp_cascade_backward(rgp_cascade_t* g, const float* gazemaps, const float* gt_gazemap, const rgp_cascade_weights* grads,
                         float* d_rows, rgp_stream_t stream) {
  RGP_REQUIRE(g && gazemaps && gt_gazemap && grads, "rgp_cascade_backward: null argument");
  if (!g->save) return set_err(RGP_ESTATE, "rgp_cascade_backward: plan was created without save_for_backward");
  if (!g->ws || !g->weights_set) return set_err(RGP_ESTATE, "rgp_cascade_backward: workspace/weights not set");
  const float* const* ptrs = (const float* const*)grads;
  for (size_t i = 0; i < 19; ++i) RGP_REQUIRE(ptrs[i], "rgp_cascade_backward: gradient pointer %zu is null", i);
  hipStream_t s = (hipStream_t)stream;
  return g->dtype == RGP_BF16 ? backward_impl<bf16_t>(g, gazemaps, gt_gazemap, grads, d_rows, s)
                              : backward_impl<float>(g, gazemaps, gt_gazemap, grads, d_rows, s);
}

}  // extern "C"
