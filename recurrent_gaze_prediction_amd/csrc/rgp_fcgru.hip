// librgp_hip.so: the fully-connected GRU gaze model (BASELINE config 2), forward and backward.
// Reference graph: /root/reference/models/gaze_rnn.py:211-360 (GazePredictionGRU.
// create_gazeprediction_network): 1024->32 projection per pixel, flatten to 1568,
// tf rnn_cell.GRUCell(1617) over T steps, 1617 -> GH*GW output projection.
// TF-1.x GRUCell:  [r,u] = sigmoid([x,h] Wg + bg);  c = tanh([x, r*h] Wc + bc);
//                  h' = u*h + (1-u)*c.
//
// Nothing new on the device for the forward: every contraction is igemm_kernel in plain-GEMM form (one
// row per "image"), the x-parts of both kernels are hoisted over all T steps, and the gate math
// is the ConvGRU epilogue pair (EpiGruZR / EpiGruC) with the gate columns packed as [u | r]
// and an identity "batch-norm".  All K / N extents are zero-padded to multiples of 64.
// Backward (tf.gradients of the loss of gaze_rnn.py:363-408, base.py:278-281): BPTT with two input-gradient
// GEMMs per step; every weight gradient is hoisted over all T steps and computed by wgrad_kernel
// (rows = frames, transposing LDS reads) on the operand rows the forward kept.
#include <algorithm>

#include "wgrad_launch.h"

using namespace rgp;

struct rgp_fcgru {
  int B = 0, T = 0, F = 0, G = 0, Gp = 0, dtype = RGP_F32;
  int Cp = 32, nx = 1568, n = 1617, Kx = 0, np = 0;
  ConvDesc proj, xg, zr, c, out;
  size_t o_zero1 = 0, o_lin = 0;
  size_t xt = 0, E = 0, xpre = 0, hall = 0, u = 0, hp = 0, rh = 0, hrows = 0, xbias = 0, ones = 0, zeros = 0;
  size_t ws_bytes = 0;
  char* ws = nullptr;
  bool weights_set = false;
  const float *proj_b = nullptr, *out_b = nullptr;
  // training-time dropout on c3d_embedded (gaze_rnn.py:302-303): caller-owned keep bytes [F*49*32], null = off
  const unsigned char* drop_mask = nullptr;
  float drop_keep = 1.0f;
  SideStream side;           // backward: the filter and bias gradients, beside the BPTT loop and the projection's input gradient
  // ---- training ----
  bool save = false;
  size_t hall_t = 0, uall = 0, rall = 0, call = 0;   // fp32 [T(+1)][B][np]
  size_t hp_all = 0, rh_all = 0;                      // T [B][T+1][np] (h_{t-1} of step t in slot (b,t)) / [B][T][np]
  ConvDesc b_out, b_c, b_zr, b_x;                     // input-gradient GEMMs over the transposed kernels
  size_t dzo = 0;                                     // T [F+1][Gp]   d loss / d logits, row 0 zero
  size_t dh_head = 0, carry = 0, drh = 0;             // fp32 [F][np], [B][np], [B][np]
  size_t dcp = 0, dzr = 0;                            // T [B][np], [B][2np]
  size_t dxpre = 0;                                   // T [F+1][3np]  [du_pre | dr_pre | dc_pre], row 0 zero
  size_t dE = 0;                                      // T [F+1][Kx]   row 0 zero
  size_t dwx = 0, dwh = 0, dwc = 0;                   // fp32 [Kx][3np], [np][2np], [np][np]
};

namespace rgp {
int dropout_apply(void* x, int dtype, const unsigned char* mask, long long rows, int cols, long long ld, float keep, hipStream_t s);
}

namespace {

void gemm_desc(ConvDesc& d, int N, int K, long long lda, long long ldc, int dtype) {
  d.Mw = 1; d.N = N; d.in_img_stride = lda; d.out_img_stride = ldc;
  d.in_tab = {0}; d.out_tab = {0};
  build_k_schedule(d, {0}, {0}, K, dtype);
}

template <typename T>
int set_weights_impl(rgp_fcgru* g, const rgp_fcgru_weights* w, hipStream_t s) {
  char* ws = g->ws;
  const int n = g->n, nx = g->nx, np = g->np;
  // every pack of this call in ONE launch (the optimizer step re-packs all 15 filter views: 15 launches of ~10 us and 9
  // memsets were 7 % of config 2's training step).  The packed-filter areas were zeroed with the workspace at bind time
  // and a pack writes the same positions every time: their padding stays zero without a memset per call.
  PackBatch<T> pb(ws, s);
  // projection [1024, 32]
  g->proj.s_tap = 0; g->proj.s_n = 1; g->proj.s_c = g->Cp;
  RGP_TRY(pb.add(g->proj, w->proj_c3d_W, g->Cp, 0));
  // gate kernel [nx+n, 2n] columns [r | u]; candidate kernel [nx+n, n].  Packed rows: [u | r | c].
  auto pk = [&](ConvDesc& d, const float* src, long long ld, int k_rows, int row0) -> int {
    d.s_tap = 0; d.s_n = 1; d.s_c = ld; d.cin_src = k_rows;
    return pb.add(d, src, n, row0);                            // (the job copies the strides set above)
  };
  RGP_TRY(pk(g->xg, w->gates_kernel + n, 2LL * n, nx, 0));                 // u, x-part
  RGP_TRY(pk(g->xg, w->gates_kernel, 2LL * n, nx, np));                    // r, x-part
  RGP_TRY(pk(g->xg, w->candidate_kernel, n, nx, 2 * np));                  // c, x-part
  RGP_TRY(pk(g->zr, w->gates_kernel + (long long)nx * 2 * n + n, 2LL * n, n, 0));   // u, h-part
  RGP_TRY(pk(g->zr, w->gates_kernel + (long long)nx * 2 * n, 2LL * n, n, np));      // r, h-part
  RGP_TRY(pk(g->c, w->candidate_kernel + (long long)nx * n, n, n, 0));              // c, (r*h)-part
  g->out.s_tap = 0; g->out.s_n = 1; g->out.s_c = g->G; g->out.cin_src = n;
  RGP_TRY(pb.add(g->out, w->proj_out_W, g->G, 0));
  // bias of the hoisted x-GEMM: [bu | br | bc] padded
  float* xb = (float*)(ws + g->xbias);
  RGP_HIP(hipMemsetAsync(xb, 0, (size_t)3 * np * 4, s));
  RGP_HIP(hipMemcpyAsync(xb, w->gates_bias + n, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
  RGP_HIP(hipMemcpyAsync(xb + np, w->gates_bias, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
  RGP_HIP(hipMemcpyAsync(xb + 2 * np, w->candidate_bias, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
  g->proj_b = w->proj_c3d_b;
  g->out_b = w->proj_out_b;
  if (g->save) {
    // transposed kernels for the input gradients: packed row = the GEMM's output unit, K = the gradient's columns
    auto pkT = [&](ConvDesc& d, const float* src, long long row_stride, int cols, int rows, int k0) -> int {
      d.s_tap = 0; d.s_n = row_stride; d.s_c = 1; d.cin_src = cols;
      return pb.add(d, src, rows, 0, k0, 1);
    };
    RGP_TRY(pkT(g->b_out, w->proj_out_W, g->G, g->G, n, 0));                                        // d h = d logits Wout^T
    RGP_TRY(pkT(g->b_c, w->candidate_kernel + (long long)nx * n, n, n, n, 0));                      // d(r.h) = dc_pre Wc_h^T
    RGP_TRY(pkT(g->b_zr, w->gates_kernel + (long long)nx * 2 * n + n, 2LL * n, n, n, 0));           // carry += du_pre Wu_h^T
    RGP_TRY(pkT(g->b_zr, w->gates_kernel + (long long)nx * 2 * n, 2LL * n, n, n, np));              //        + dr_pre Wr_h^T
    RGP_TRY(pkT(g->b_x, w->gates_kernel + n, 2LL * n, n, nx, 0));                                   // d E = [du|dr|dc]_pre Wx^T
    RGP_TRY(pkT(g->b_x, w->gates_kernel, 2LL * n, n, nx, np));
    RGP_TRY(pkT(g->b_x, w->candidate_kernel, n, n, nx, 2 * np));
  }
  RGP_TRY(pb.flush());
  g->weights_set = true;
  return RGP_OK;
}

__global__ void fill_kernel(float* p, float v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

template <typename T>
int forward_impl(rgp_fcgru* g, const float* c3d_input, float* logits, float* probs, hipStream_t s) {
  char* ws = g->ws;
  const int B = g->B, T_ = g->T, F = g->F, np = g->np;
  const bool save = g->save;
  nchw_to_rows_kernel<T><<<dim3(1024 / 64, F), 256, 0, s>>>(c3d_input, (T*)(ws + g->xt), 1024);
  RGP_HIP(hipGetLastError());
  {  // per-pixel projection, rows of the x-GEMM (gaze_rnn.py:294-308, flatten :340-341)
    IgemmParams p = make_params(g->proj, ws + g->xt, ws, F);
    EpiParams e = make_epi(g->proj, ws + g->E, ws);
    e.bias = g->proj_b;
    RGP_TRY((launch_igemm<T, 1, 1, EpiStore<T, true, false>>(p, e, s)));
  }
  // tf.nn.dropout on the projected features (training only); a frame's 49*32 values are one row of E
  if (g->drop_mask) RGP_TRY(dropout_apply(ws + g->E, g->dtype, g->drop_mask, F, g->nx, g->Kx, g->drop_keep, s));
  {  // hoisted x-parts of both GRU kernels, biases folded in
    IgemmParams p = make_params(g->xg, ws + g->E, ws, F);
    EpiParams e = make_epi(g->xg, ws + g->xpre, ws);
    e.bias = (const float*)(ws + g->xbias);
    RGP_TRY((launch_igemm<T, 1, 1, EpiStore<float, true, false>>(p, e, s)));
  }
  const size_t st = (size_t)B * np;
  if (!save) RGP_HIP(hipMemsetAsync(ws + g->hp, 0, st * sizeof(T), s));
  float* hall = (float*)(ws + (save ? g->hall_t : g->hall));
  RGP_HIP(hipMemsetAsync(hall, 0, st * 4, s));
  for (int t = 0; t < T_; ++t) {
    // training keeps every step's states, gates and operand rows (slot (b, 0) of hp_all is never written = h_0)
    char* hp_in = save ? ws + g->hp_all + (size_t)t * np * sizeof(T) : ws + g->hp;
    char* hp_out = save ? ws + g->hp_all + (size_t)(t + 1) * np * sizeof(T) : ws + g->hp;
    char* rh_buf = save ? ws + g->rh_all + (size_t)t * np * sizeof(T) : ws + g->rh;
    const long long hp_stride = save ? (long long)(T_ + 1) * np : np, rh_stride = save ? (long long)T_ * np : np;
    EpiParams e = make_epi(g->zr, rh_buf, ws);
    e.out_img_stride = rh_stride;
    e.xpre = (const float*)(ws + g->xpre) + (size_t)t * 3 * np;
    e.xpre_img_stride = (long long)T_ * 3 * np;
    e.xpre_ld = 3 * np;
    e.xpre_col = 0;
    e.S = np;
    e.state_rows = 1;
    e.h_prev = hall + (size_t)(save ? t : (t & 1)) * st;
    e.h_next = hall + (size_t)(save ? t + 1 : ((t + 1) & 1)) * st;
    e.u_gate = save ? (float*)(ws + g->uall) + (size_t)t * st : (float*)(ws + g->u);
    e.r_save = save ? (float*)(ws + g->rall) + (size_t)t * st : nullptr;
    e.c_save = save ? (float*)(ws + g->call) + (size_t)t * st : nullptr;
    IgemmParams p = make_params(g->zr, hp_in, ws, B);
    p.in_img_stride = hp_stride;
    RGP_TRY((launch_igemm<T, 1, 1, EpiGruZR<T>>(p, e, s)));
    e.out = hp_out;
    e.out_tab = (const int*)(ws + g->c.out_tab_off);
    e.out_img_stride = hp_stride;
    e.xpre_col = 2 * np;
    e.out2 = ws + g->hrows;
    e.out2_tab = (const int*)(ws + g->c.out_tab_off);
    e.out2_img_stride = np;
    e.out2_img_mul = T_;
    e.out2_img_add = t;
    e.bn_gamma = (const float*)(ws + g->ones);
    e.bn_beta = (const float*)(ws + g->zeros);
    e.bn_inv_std = 1.0f;
    IgemmParams pc = make_params(g->c, rh_buf, ws, B);
    pc.in_img_stride = rh_stride;
    RGP_TRY((launch_igemm<T, 1, 1, EpiGruC<T>>(pc, e, s)));
  }
  {  // output projection (gaze_rnn.py:346-349)
    IgemmParams p = make_params(g->out, ws + g->hrows, ws, F);
    EpiParams e = make_epi(g->out, logits, ws);
    e.bias = g->out_b;
    RGP_TRY((launch_igemm<T, 1, 1, EpiStore<float, true, false>>(p, e, s)));
  }
  if (probs) RGP_TRY(rgp_softmax_xent_fwd(logits, nullptr, probs, nullptr, nullptr, F, g->G, (rgp_stream_t)s));
  return RGP_OK;
}

// ---------------------------------------------------------------- backward kernels
// d loss / d logits (gaze_rnn.py:363-408): xentropy (probs - labels) / F, l2 (logits - labels) / F
template <typename T>
__global__ __launch_bounds__(256) void fc_dlogits_kernel(const float* __restrict__ a, const float* __restrict__ labels, float scale,
                                                         T* __restrict__ dzo, int G, int Gp, long long total) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int gcol = (int)(i % G);
    const long long f = i / G;
    dzo[(f + 1) * Gp + gcol] = Elem<T>::to((a[i] - labels[i]) * scale);
  }
}

// out[c] = sum over rows of base[row * row_stride + c]: bias gradients.  Block = 16 columns x 16 row lanes (a thread per
// column looping over all F rows -- a handful of blocks, F dependent loads each -- took 0.23 ms per call at 1024 frames,
// 28 % of config 2's training step); launch with (ncols + 15) / 16 blocks of 256 threads.
template <typename T>
__global__ __launch_bounds__(256) void fc_colsum_kernel(const T* __restrict__ base, long long row_stride, long long rows, int ncols,
                                                        float* __restrict__ out) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float a = 0.f;
  if (c < ncols)
    for (long long r = rl; r < rows; r += 16) a += Elem<T>::from(base[r * row_stride + c]);
  red[rl][cl] = a;
  __syncthreads();
  if (rl == 0 && c < ncols) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r][cl];
    out[c] = t;
  }
}

// BPTT step (GRUCell differentiated), part 1 / part 2: see top_bwd{1,2}_kernel of the cascade for the algebra
template <typename T>
__global__ __launch_bounds__(256) void fc_bwd1_kernel(const float* __restrict__ dh_head, float* __restrict__ carry,
                                                      const float* __restrict__ h_prev, const float* __restrict__ u,
                                                      const float* __restrict__ c, T* __restrict__ dxpre, T* __restrict__ dcp,
                                                      int B, int T_, int t, int np, int first) {
  const long long total = (long long)B * np;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int j = (int)(i % np);
    const long long f = (i / np) * T_ + t;
    const float dh = dh_head[f * np + j] + (first ? 0.f : carry[i]);
    const float uu = u[i], cc = c[i];
    const float du = dh * (h_prev[i] - cc), dc = dh * (1.f - uu);
    carry[i] = dh * uu;
    const T dcp_t = Elem<T>::to(dc * (1.f - cc * cc));
    T* row = dxpre + (f + 1) * 3 * np;
    row[j] = Elem<T>::to(du * uu * (1.f - uu));
    row[2 * np + j] = dcp_t;
    dcp[i] = dcp_t;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void fc_bwd2_kernel(const float* __restrict__ drh, float* __restrict__ carry,
                                                      const float* __restrict__ h_prev, const float* __restrict__ r,
                                                      T* __restrict__ dxpre, T* __restrict__ dzr, int B, int T_, int t, int np) {
  const long long total = (long long)B * np;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int j = (int)(i % np);
    const long long b = i / np, f = b * T_ + t;
    const float d = drh[i], rr = r[i];
    carry[i] += d * rr;
    const T drp = Elem<T>::to(d * h_prev[i] * rr * (1.f - rr));
    T* row = dxpre + (f + 1) * 3 * np;
    row[np + j] = drp;
    dzr[b * 2 * np + j] = row[j];
    dzr[b * 2 * np + np + j] = drp;
  }
}

// packed gradients -> the TF kernels: gates_kernel [nx+n, 2n] columns [r | u], candidate_kernel [nx+n, n]
__global__ __launch_bounds__(256) void fc_unpack_kernel(const float* __restrict__ dwx, const float* __restrict__ dwh,
                                                        const float* __restrict__ dwc, float* __restrict__ gates,
                                                        float* __restrict__ cand, int nx, int n, int np) {
  const long long total = (long long)(nx + n) * n;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int j = (int)(i % n);
    const int k = (int)(i / n);
    if (k < nx) {
      const float* row = dwx + (long long)k * 3 * np;
      gates[(long long)k * 2 * n + j] = row[np + j];
      gates[(long long)k * 2 * n + n + j] = row[j];
      cand[i] = row[2 * np + j];
    } else {
      const float* row = dwh + (long long)(k - nx) * 2 * np;
      gates[(long long)k * 2 * n + j] = row[np + j];
      gates[(long long)k * 2 * n + n + j] = row[j];
      cand[i] = dwc[(long long)(k - nx) * np + j];
    }
  }
}

template <typename T>
int backward_impl(rgp_fcgru* g, const float* logits, const float* probs, const float* labels, const rgp_fcgru_weights* gr,
                  int loss_l2, hipStream_t s) {
  char* ws = g->ws;
  const int B = g->B, T_ = g->T, F = g->F, np = g->np, n = g->n, nx = g->nx, Kx = g->Kx, G = g->G, Gp = g->Gp;
  auto Tp = [&](size_t off) { return (T*)(ws + off); };
  auto Fp = [&](size_t off) { return (float*)(ws + off); };
  auto nblk = [](long long x) { return (int)std::min<long long>((x + 255) / 256, 8192); };
  const size_t st = (size_t)B * np;
  // 1. loss layer and output projection
  fc_dlogits_kernel<T><<<nblk((long long)F * G), 256, 0, s>>>(loss_l2 ? logits : probs, labels, 1.0f / (float)F, Tp(g->dzo), G, Gp,
                                                             (long long)F * G);
  fc_colsum_kernel<T><<<(G + 15) / 16, 256, 0, s>>>(Tp(g->dzo) + Gp, Gp, F, G, (float*)gr->proj_out_b);
  RGP_HIP(hipGetLastError());
  WgradParams wp;
  auto rows_wgrad = [&](const void* X, int ldx, const ConvDesc& fwd, const void* dY, int ldy, int y_col, int N, float* dW, int ldw,
                        int k_valid, hipStream_t s) -> int {
    RGP_HIP(hipMemsetAsync(dW, 0, (size_t)k_valid * ldw * 4, s));
    memset(&wp, 0, sizeof(wp));
    wp.X = X; wp.dY = dY; wp.dW = dW;
    wgrad_grid(wp, 1, 1, F);
    wp.x_sx = ldx; wp.y_sx = ldy; wp.y_org = ldy + y_col;
    wp.koff = (const int*)(ws + fwd.koff_off);
    wp.M = F; wp.N = N; wp.nk = fwd.nk; wp.ldw = ldw; wp.k_valid = k_valid;
    return launch_wgrad<T, 1>(wp, s);
  };
  hipStream_t sw = s;                                           // the gradients' stream (SideStream, rgp_host.h)
  RGP_TRY(g->side.fork(s, 0, &sw));
  RGP_TRY(rows_wgrad(ws + g->hrows, np, g->out, ws + g->dzo, Gp, 0, G, (float*)gr->proj_out_W, G, n, sw));
  {
    IgemmParams p = make_params(g->b_out, Tp(g->dzo) + Gp, ws, F);
    EpiParams e = make_epi(g->b_out, Fp(g->dh_head), ws);
    RGP_TRY((launch_igemm<T, 1, 1, EpiStore<float, false, false>>(p, e, s)));
  }
  // 2. BPTT
  const float* hall = Fp(g->hall_t);
  for (int t = T_ - 1; t >= 0; --t) {
    const float* h_prev = hall + (size_t)t * st;
    fc_bwd1_kernel<T><<<nblk((long long)st), 256, 0, s>>>(Fp(g->dh_head), Fp(g->carry), h_prev, Fp(g->uall) + (size_t)t * st,
                                                         Fp(g->call) + (size_t)t * st, Tp(g->dxpre), Tp(g->dcp), B, T_, t, np, t == T_ - 1);
    RGP_HIP(hipGetLastError());
    {
      IgemmParams p = make_params(g->b_c, Tp(g->dcp), ws, B);
      EpiParams e = make_epi(g->b_c, Fp(g->drh), ws);
      RGP_TRY((launch_igemm<T, 1, 1, EpiStore<float, false, false>>(p, e, s)));
    }
    fc_bwd2_kernel<T><<<nblk((long long)st), 256, 0, s>>>(Fp(g->drh), Fp(g->carry), h_prev, Fp(g->rall) + (size_t)t * st, Tp(g->dxpre),
                                                         Tp(g->dzr), B, T_, t, np);
    RGP_HIP(hipGetLastError());
    {
      IgemmParams p = make_params(g->b_zr, Tp(g->dzr), ws, B);
      EpiParams e = make_epi(g->b_zr, Fp(g->carry), ws);
      RGP_TRY((launch_igemm<T, 1, 1, EpiAccumF32>(p, e, s)));
    }
  }
  // 3. kernel gradients, hoisted over all steps (side stream: beside step 4)
  RGP_TRY(g->side.fork(s, 1, &sw));
  const T* dx1 = Tp(g->dxpre) + 3 * np;
  {
  hipStream_t s = sw;
  RGP_TRY(rows_wgrad(ws + g->E, Kx, g->xg, ws + g->dxpre, 3 * np, 0, 3 * np, Fp(g->dwx), 3 * np, Kx, s));
  {
    RGP_HIP(hipMemsetAsync(Fp(g->dwh), 0, (size_t)np * 2 * np * 4, s));
    memset(&wp, 0, sizeof(wp));
    wp.X = ws + g->hp_all; wp.dY = ws + g->dxpre; wp.dW = Fp(g->dwh);
    wgrad_grid(wp, T_, 1, 1);                                   // image = clip b, z = step t
    wp.x_sz = np; wp.x_img_stride = (long long)(T_ + 1) * np;
    wp.y_sz = 3 * np; wp.y_img_stride = (long long)T_ * 3 * np; wp.y_org = 3 * np;
    wp.koff = (const int*)(ws + g->zr.koff_off);
    wp.M = F; wp.N = 2 * np; wp.nk = g->zr.nk; wp.ldw = 2 * np; wp.k_valid = np;
    RGP_TRY((launch_wgrad<T, 1>(wp, s)));
    RGP_HIP(hipMemsetAsync(Fp(g->dwc), 0, (size_t)np * np * 4, s));
    wp.X = ws + g->rh_all; wp.dW = Fp(g->dwc);
    wp.x_img_stride = (long long)T_ * np;
    wp.y_org = 3 * np + 2 * np;
    wp.koff = (const int*)(ws + g->c.koff_off);
    wp.N = np; wp.nk = g->c.nk; wp.ldw = np;
    RGP_TRY((launch_wgrad<T, 1>(wp, s)));
  }
  fc_unpack_kernel<<<nblk((long long)(nx + n) * n), 256, 0, s>>>(Fp(g->dwx), Fp(g->dwh), Fp(g->dwc), (float*)gr->gates_kernel,
                                                               (float*)gr->candidate_kernel, nx, n, np);
  // biases: gates_bias [r | u], candidate_bias
  fc_colsum_kernel<T><<<(n + 15) / 16, 256, 0, s>>>(dx1 + np, 3LL * np, F, n, (float*)gr->gates_bias);
  fc_colsum_kernel<T><<<(n + 15) / 16, 256, 0, s>>>(dx1, 3LL * np, F, n, (float*)gr->gates_bias + n);
  fc_colsum_kernel<T><<<(n + 15) / 16, 256, 0, s>>>(dx1 + 2 * np, 3LL * np, F, n, (float*)gr->candidate_bias);
  RGP_HIP(hipGetLastError());
  }
  // 4. projection: d E = dxpre Wx^T, then per-pixel rows [F*49][32]
  {
    IgemmParams p = make_params(g->b_x, dx1, ws, F);
    EpiParams e = make_epi(g->b_x, Tp(g->dE) + Kx, ws);
    RGP_TRY((launch_igemm<T, 1, 1, EpiStore<T, false, false>>(p, e, s)));
  }
  if (g->drop_mask) RGP_TRY(dropout_apply(Tp(g->dE) + Kx, g->dtype, g->drop_mask, F, g->nx, Kx, g->drop_keep, s));
  {
    RGP_HIP(hipMemsetAsync((void*)gr->proj_c3d_W, 0, (size_t)1024 * g->Cp * 4, s));
    memset(&wp, 0, sizeof(wp));
    wp.X = ws + g->xt; wp.dY = ws + g->dE; wp.dW = (float*)gr->proj_c3d_W;
    wgrad_grid(wp, 1, 1, 49);                                   // image = frame, 49 pixel rows
    wp.x_sx = 1024; wp.x_img_stride = 49LL * 1024;
    wp.y_sx = g->Cp; wp.y_img_stride = Kx; wp.y_org = Kx;
    wp.koff = (const int*)(ws + g->proj.koff_off);
    wp.M = (long long)F * 49; wp.N = g->Cp; wp.nk = g->proj.nk; wp.ldw = g->Cp; wp.k_valid = 1024;
    RGP_TRY((launch_wgrad<T, 1>(wp, s)));
    // bias: sum over frames and pixels = column sums of the [F*49][32] view; rows are 32 apart inside a frame row
    // of Kx, so sum per frame-row column first: c3d_b[c] = sum_f sum_pix dE[f][pix*32 + c]
    if (sw != s) RGP_TRY(g->side.join(s));                      // (dwc is the candidate filter's gradient until fc_unpack has read it)
    fc_colsum_kernel<T><<<(Kx + 15) / 16, 256, 0, s>>>(Tp(g->dE) + Kx, Kx, F, Kx, Fp(g->dwc));     // dwc reused as [Kx] scratch
    RGP_HIP(hipGetLastError());
  }
  return RGP_OK;
}

// proj_c3d_b[c] = sum_pix colsum[pix*32 + c]
__global__ void fc_fold_bias_kernel(const float* __restrict__ colsum, float* __restrict__ out, int Cp) {
  const int c = threadIdx.x;
  if (c >= Cp) return;
  float a = 0.f;
  for (int p = 0; p < 49; ++p) a += colsum[p * Cp + c];
  out[c] = a;
}

}  // namespace

extern "C" {

int rgp_fcgru_set_dropout(rgp_fcgru_t* g, float keep_prob, const unsigned char* mask) {
  RGP_REQUIRE(g, "rgp_fcgru_set_dropout: null plan");
  RGP_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f, "rgp_fcgru_set_dropout: keep_prob %g not in (0, 1]", (double)keep_prob);
  g->drop_mask = keep_prob < 1.f ? mask : nullptr;
  g->drop_keep = g->drop_mask ? keep_prob : 1.0f;
  return RGP_OK;
}

int rgp_fcgru_create(rgp_fcgru_t** plan, int batch, int n_steps, int gazemap_h, int gazemap_w, int dtype) {
  return rgp_fcgru_create_ex(plan, batch, n_steps, gazemap_h, gazemap_w, dtype, 0);
}

int rgp_fcgru_create_ex(rgp_fcgru_t** plan, int batch, int n_steps, int gazemap_h, int gazemap_w, int dtype, int save_for_backward) {
  RGP_REQUIRE(plan && batch > 0 && n_steps > 0, "rgp_fcgru_create: bad arguments");
  RGP_REQUIRE((gazemap_h == 49 && gazemap_w == 49) || (gazemap_h == 7 && gazemap_w == 7),
              "rgp_fcgru_create: gaze map %dx%d (reference uses 49x49 or 7x7)", gazemap_h, gazemap_w);
  RGP_REQUIRE(dtype == RGP_F32 || dtype == RGP_BF16, "rgp_fcgru_create: dtype %d", dtype);
  rgp_fcgru* g = new rgp_fcgru();
  g->B = batch; g->T = n_steps; g->F = batch * n_steps; g->G = gazemap_h * gazemap_w; g->dtype = dtype;
  g->save = save_for_backward != 0;
  g->Gp = (int)align_up(g->G, 128);
  g->Kx = (int)align_up(g->nx, 64);
  g->np = (int)align_up(g->n, 64);
  const int es = esize(dtype), F = g->F, np = g->np, Kx = g->Kx;
  // projection: rows = pixels, output scattered into the flattened [49*32] row of its frame
  g->proj.Mw = 49; g->proj.N = g->Cp; g->proj.in_img_stride = 49LL * 1024; g->proj.out_img_stride = Kx;
  for (int p = 0; p < 49; ++p) { g->proj.in_tab.push_back(p * 1024); g->proj.out_tab.push_back(p * g->Cp); }
  build_k_schedule(g->proj, {0}, {0}, 1024, dtype);
  gemm_desc(g->xg, 3 * np, Kx, Kx, 3LL * np, dtype);
  gemm_desc(g->zr, 2 * np, np, np, np, dtype);
  gemm_desc(g->c, np, np, np, np, dtype);
  gemm_desc(g->out, g->G, np, np, g->G, dtype);
  Arena a;
  for (ConvDesc* d : {&g->proj, &g->xg, &g->zr, &g->c, &g->out}) d->reserve(a, dtype);
  g->xt = a.take((size_t)F * 49 * 1024 * es);
  g->E = a.take((size_t)F * Kx * es);
  g->xpre = a.take((size_t)F * 3 * np * 4);
  g->hall = a.take((size_t)2 * batch * np * 4);
  g->u = a.take((size_t)batch * np * 4);
  g->hp = a.take((size_t)batch * np * es);
  g->rh = a.take((size_t)batch * np * es);
  g->hrows = a.take((size_t)F * np * es);
  g->xbias = a.take((size_t)3 * np * 4);
  g->ones = a.take((size_t)np * 4);
  g->zeros = a.take((size_t)np * 4);
  if (g->save) {
    const size_t st = (size_t)batch * np;
    gemm_desc(g->b_out, np, g->Gp, g->Gp, np, dtype);
    gemm_desc(g->b_c, np, np, np, np, dtype);
    gemm_desc(g->b_zr, np, 2 * np, 2LL * np, np, dtype);
    gemm_desc(g->b_x, Kx, 3 * np, 3LL * np, Kx, dtype);
    for (ConvDesc* d : {&g->b_out, &g->b_c, &g->b_zr, &g->b_x}) d->reserve(a, dtype);
    g->hall_t = a.take((size_t)(n_steps + 1) * st * 4);
    g->uall = a.take((size_t)n_steps * st * 4);
    g->rall = a.take((size_t)n_steps * st * 4);
    g->call = a.take((size_t)n_steps * st * 4);
    g->hp_all = a.take((size_t)batch * (n_steps + 1) * np * es + 1024);
    g->rh_all = a.take((size_t)F * np * es + 1024);
    g->dzo = a.take((size_t)(F + 1) * g->Gp * es + 1024);
    g->dh_head = a.take((size_t)F * np * 4);
    g->carry = a.take(st * 4);
    g->drh = a.take(st * 4);
    g->dcp = a.take(st * es);
    g->dzr = a.take(2 * st * es);
    g->dxpre = a.take((size_t)(F + 1) * 3 * np * es + 1024);
    g->dE = a.take((size_t)(F + 1) * Kx * es + 1024);
    g->dwx = a.take((size_t)Kx * 3 * np * 4);
    g->dwh = a.take((size_t)np * 2 * np * 4);
    g->dwc = a.take((size_t)np * np * 4);
  }
  g->ws_bytes = a.off;
  *plan = g;
  return RGP_OK;
}

int rgp_fcgru_destroy(rgp_fcgru_t* plan) {
  delete plan;
  return RGP_OK;
}

size_t rgp_fcgru_workspace_bytes(const rgp_fcgru_t* plan) { return plan ? plan->ws_bytes : 0; }

int rgp_fcgru_bind_workspace(rgp_fcgru_t* g, void* workspace, size_t bytes, rgp_stream_t stream) {
  RGP_REQUIRE(g && workspace, "rgp_fcgru_bind_workspace: null argument");
  if (bytes < g->ws_bytes) return set_err(RGP_EWORKSPACE, "workspace %zu < required %zu bytes", bytes, g->ws_bytes);
  RGP_REQUIRE(((size_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  g->ws = (char*)workspace;
  g->weights_set = false;
  RGP_HIP(hipMemsetAsync(g->ws, 0, g->ws_bytes, s));
  for (ConvDesc* d : {&g->proj, &g->xg, &g->zr, &g->c, &g->out}) RGP_TRY(upload_desc(*d, g->ws, s));
  if (g->save) for (ConvDesc* d : {&g->b_out, &g->b_c, &g->b_zr, &g->b_x}) RGP_TRY(upload_desc(*d, g->ws, s));
  fill_kernel<<<(g->np + 255) / 256, 256, 0, s>>>((float*)(g->ws + g->ones), 1.0f, g->np);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

int rgp_fcgru_set_weights(rgp_fcgru_t* g, const rgp_fcgru_weights* w, rgp_stream_t stream) {
  RGP_REQUIRE(g && w, "rgp_fcgru_set_weights: null argument");
  if (!g->ws) return set_err(RGP_EWORKSPACE, "rgp_fcgru: workspace not bound");
  const float* const* ptrs = (const float* const*)w;
  for (size_t i = 0; i < sizeof(rgp_fcgru_weights) / sizeof(float*); ++i)
    RGP_REQUIRE(ptrs[i], "rgp_fcgru_set_weights: weight pointer %zu is null", i);
  hipStream_t s = (hipStream_t)stream;
  return g->dtype == RGP_BF16 ? set_weights_impl<bf16_t>(g, w, s) : set_weights_impl<float>(g, w, s);
}

int rgp_fcgru_forward(rgp_fcgru_t* g, const float* c3d_input, float* logits, float* probs, rgp_stream_t stream) {
  RGP_REQUIRE(g && c3d_input && logits, "rgp_fcgru_forward: null argument");
  if (!g->ws) return set_err(RGP_EWORKSPACE, "rgp_fcgru: workspace not bound");
  if (!g->weights_set) return set_err(RGP_ESTATE, "rgp_fcgru: weights not set");
  hipStream_t s = (hipStream_t)stream;
  return g->dtype == RGP_BF16 ? forward_impl<bf16_t>(g, c3d_input, logits, probs, s)
                              : forward_impl<float>(g, c3d_input, logits, probs, s);
}

int rgp_fcgru_backward(rgp_fcgru_t* g, const float* logits, const float* probs, const float* labels,
                       const rgp_fcgru_weights* grads, int loss_type, rgp_stream_t stream) {
  RGP_REQUIRE(g && logits && labels && grads, "rgp_fcgru_backward: null argument");
  if (!g->save) return set_err(RGP_ESTATE, "rgp_fcgru_backward: plan was created without save_for_backward");
  if (!g->ws || !g->weights_set) return set_err(RGP_ESTATE, "rgp_fcgru_backward: workspace/weights not set");
  RGP_REQUIRE(loss_type == 0 || loss_type == 1, "rgp_fcgru_backward: loss_type %d (0 xentropy, 1 l2)", loss_type);
  RGP_REQUIRE(loss_type == 1 || probs, "rgp_fcgru_backward: xentropy needs the softmax maps");
  const float* const* ptrs = (const float* const*)grads;
  for (size_t i = 0; i < sizeof(rgp_fcgru_weights) / sizeof(float*); ++i)
    RGP_REQUIRE(ptrs[i], "rgp_fcgru_backward: gradient pointer %zu is null", i);
  hipStream_t s = (hipStream_t)stream;
  RGP_TRY(g->dtype == RGP_BF16 ? backward_impl<bf16_t>(g, logits, probs, labels, grads, loss_type, s)
                               : backward_impl<float>(g, logits, probs, labels, grads, loss_type, s));
  fc_fold_bias_kernel<<<1, 64, 0, s>>>((const float*)(g->ws + g->dwc), (float*)grads->proj_c3d_b, g->Cp);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

}  // extern "C"
