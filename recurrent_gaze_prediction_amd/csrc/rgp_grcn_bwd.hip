// librgp_hip.so: backward pass of the gaze_grcn head and the TF-style optimizer step.
// Differentiates /root/reference/models/gaze_grcn.py:173-376 under the loss of
// gaze_rnn.py:363-408 (what tf.gradients builds, base.py:278-281), then applies
// clip_by_global_norm + AdamOptimizer (base.py:286-297).
//
// Structure: every data gradient (dgrad) is the SAME implicit-GEMM kernel as the forward
// with a re-packed (rotated / transposed) filter; every weight gradient (wgrad) is a plain
// split-K GEMM  dW[k, n] = sum_m A^T[k, m] * dY^T[n, m]  on operands that a gather-transpose
// kernel has laid out K(=m)-contiguous, accumulated with float atomics.  All weight
// gradients of the recurrence are hoisted out of the time loop (one GEMM over all T steps),
// so BPTT itself is 2 dgrad convs + 2 element-wise kernels per step.
#include <algorithm>
#include <cstring>

#include "bwd_kernels.hip.h"
#include "rgp_grcn_plan.h"
#include "wgrad_launch.h"
#include "convgru_bptt.hip.h"
#include "head_fold.hip.h"

using namespace rgp;

struct GrcnBwd {
  ConvDesc b_d2, b_d1, b_c, b_zr, b_x;     // dgrad convolutions
  ConvDesc b_px;                           // projection input gradient: d rows = dE x W^T
  ConvDesc b_hf;                           // folded head (head_fold.hip.h): dy = Pm x K, rows = (frame, 7x7 position), K = 384
  Buf pm, dkf, dhf, dhp;                   // its patches [F*49][384] T, dK [384][S], dH [11,11,64] and dH's 25 partial sums, fp32
  // gather tables [ntaps][Mw] (element offsets, -1 = zero) + offsets in the workspace
  std::vector<int> t_y, t_pad3S, t_pad2S, koff_c;
  size_t o_y = 0, o_pad3S = 0, o_pad2S = 0, o_koff_c = 0;
  long long M = 0, M2 = 0, Mp = 0, M2p = 0;
  Buf dz, frame_sum, dgp, gp, dd2, dd1, dy, dh_head, dh_carry, drh, dcp_pad, dzr_pad, dxpre, dxpre_pad, dE;
  Buf xch_c, xch_z, xch_r, bptt_cnt;       // persistent BPTT kernel (convgru_bptt.hip.h): exchange images + phase counters
  Buf hp_all, rhp_all, dzb, ptoep, sq_partial;   // dzb / ptoep: blocked dz and the Toeplitz partial sums of the head filter gradient     // halo-padded h_{t-1} and r.h_{t-1} of every step, [t][b][9][9][S] (wgrad operands)
  rgp_grcn_weights w;   // forward weights (device fp32) as last set
  // recorded by backward_impl once a group of gradients is final (rgp_grcn_wait_grads): 0 = batch-norm + upsampling +
  // output layer, 1 = the six ConvGRU filters, 2 = the projection (= end of the backward)
  hipEvent_t grad_ev[3] = {nullptr, nullptr, nullptr};
  bool grad_ev_made = false, grad_ev_recorded = false;
  // The folded head's chain rule (dK -> dF1, dH, dF2, dG, dF3, d out_W: six small dependent kernels, 63 us at config 4's shape)
  // feeds nothing the rest of the backward reads: it runs on a stream of the plan's own, beside the dy GEMM, the batch-norm
  // backward and the BPTT launch (which leaves CUs free at <= 24 clips; at more it is simply queued behind them), and is
  // joined at the end of the call (also inside a stream capture: the graph then has two branches).
  // Behind the BPTT launch the same stream takes the six ConvGRU filter gradients (two grouped wgrad launches, each one
  // round of blocks) while the calling stream runs the input convolutions' dgrad and the projection's gradients: neither
  // branch reads what the other writes (ev_wfork / ev_wjoin).
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_bn = nullptr, ev_join = nullptr, ev_wfork = nullptr, ev_wjoin = nullptr;
  ~GrcnBwd() {
    if (grad_ev_made) for (int i = 0; i < 3; ++i) (void)hipEventDestroy(grad_ev[i]);
    if (ev_fork) {                                           // (the stream belongs to the device's pool)
      (void)hipEventDestroy(ev_fork); (void)hipEventDestroy(ev_bn); (void)hipEventDestroy(ev_join);
      (void)hipEventDestroy(ev_wfork); (void)hipEventDestroy(ev_wjoin);
    }
  }
};

namespace {

constexpr int SQ_BLOCKS = 256;

// b->side = side stream 0 of the device's pool (rgp_core.hip pool_stream; `create`: not inside a stream capture) + the plan's events
int make_side_stream(GrcnBwd* b, bool create) {
  if (b->side) return RGP_OK;
  RGP_TRY(pool_stream(0, create, &b->side));
  if (b->side && !b->ev_fork)
    for (hipEvent_t* e : {&b->ev_fork, &b->ev_bn, &b->ev_join, &b->ev_wfork, &b->ev_wjoin})
      RGP_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
  return RGP_OK;
}

size_t put(Arena& a, const std::vector<int>& t) { return a.take(t.size() * 4); }

// Folded 7x7 head filter gradient dGp[u,v,c] = sum_{f,y,x} dz[f,y,x] d2pad[f,y+u,x+v,c] as seven wgrad_kernel launches
// (the Toeplitz view of run_d3): rows = (frame, y, block of 16 x), X = the 16 dz values of the block, dY = the 704-element
// run d2pad[y+u, 16xb .. +21, :]  ->  P[u][n][x'*32+c];  dGp[u,v,c] = sum_n P[u][n][(n+v)*32+c].
// dz in blocks of 16 pixels: [f][y][xb 0..3][n]; block 3 starts at x = 48 and holds one pixel (the rest zero)
template <typename T>
__global__ __launch_bounds__(256) void dz_block_kernel(const float* __restrict__ dz, T* __restrict__ dzb, long long total) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int n = (int)(i & 15), xb = (int)((i >> 4) & 3);
    const long long fy = i >> 6;                               // f*49 + y
    const int x = 16 * xb + n;
    dzb[i] = Elem<T>::to(x < 49 ? dz[fy * 49 + x] : 0.f);
  }
}
__global__ void head_fold_toeplitz_kernel(const float* __restrict__ P, float* __restrict__ dgp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;        // over 49 taps x 32 channels
  if (i >= 49 * 32) return;
  const int c = i % 32, v = (i / 32) % 7, u = i / (32 * 7);
  float a = 0.f;
  for (int n = 0; n < 16; ++n) a += P[((long long)u * 16 + n) * 704 + (n + v) * 32 + c];
  dgp[i] = a;
}

// out[c] += sum over rows of x[r][c]; thread = (row lane, 8-column group), one atomic per column and block
template <typename T>
__global__ __launch_bounds__(256) void dense_colsum_kernel(const T* __restrict__ x, long long rows, int C, float* __restrict__ out) {
  __shared__ float red[256 * 8];
  const int CG = C / 8, cg = threadIdx.x % CG, rl = threadIdx.x / CG, RL = 256 / CG;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (long long r = (long long)blockIdx.x * RL + rl; r < rows; r += (long long)gridDim.x * RL) {
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] += Elem<T>::from(x[r * C + cg * 8 + k]);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[rl * C + cg * 8 + k] = a[k];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float t = 0.f;
    for (int r = 0; r < RL; ++r) t += red[r * C + c];
    atomicAdd(out + c, t);
  }
}

template <typename T>
int backward_impl(rgp_grcn* g, const float* probs, const float* logits, const float* labels,
                  const rgp_grcn_weights* gr, int loss_l2, hipStream_t s, const float* ext_dy = nullptr) {
  GrcnBwd* b = g->bwd;
  char* ws = g->ws;
  const int B = g->B, T_ = g->T, S = g->S, P = g->P, F = g->F;
  const long long M = b->M, M2 = b->M2;
  auto wg_params = []() { WgradParams p; memset(&p, 0, sizeof(p)); return p; };
  const size_t st = (size_t)B * 49 * S;
  auto I = [&](size_t off) { return (const int*)(ws + off); };
  auto Fp = [&](const Buf& x) { return (float*)(ws + x.off); };
  auto Tp = [&](const Buf& x) { return (T*)(ws + x.off); };
  if (!b->grad_ev_made) {
    for (int i = 0; i < 3; ++i) RGP_HIP(hipEventCreateWithFlags(&b->grad_ev[i], hipEventDisableTiming));
    b->grad_ev_made = true;
  }
  // (not while `s` is being captured into a graph: a replay runs on one stream and has nobody to signal)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  const bool mark = !(hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone);
  // the folded head's chain rule on the plan's side stream.  While `s` is being captured the same fork / join is recorded into
  // the graph (two parallel branches) -- provided the side stream exists already: streams are not created during a capture
  const bool persistent = sizeof(T) == 2 && seq_persistent_ok(g) && dev_knob("RGP_SEQ", 1);
  const bool stepwise = ext_dy && g->bwd_step_ev && !persistent && mark;
  RGP_TRY(make_side_stream(b, mark));
  const bool side_ok = b->side != nullptr && dev_knob("RGP_BWD_FORK", 1);
  const bool fork = g->fold_head && !ext_dy && side_ok;
  const bool wfork = side_ok && dev_knob("RGP_BWD_FORK", 1) != 2;


  // zero the gradients that are accumulated with atomics: one region when the caller's gradient tensors are the
  // slices of one flat buffer (engine.py: flat_grads), else one per tensor -- all clears of the call in ONE launch (ZeroBatch)
  ZeroBatch zb(s);
  {
    struct Z { const float* q; size_t n; };
    const Z z[] = {{gr->proj_c3d_W, (size_t)1024 * P}, {gr->proj_c3d_b, (size_t)P},
                   {gr->gru_Wz, (size_t)9 * P * S}, {gr->gru_Wr, (size_t)9 * P * S}, {gr->gru_W, (size_t)9 * P * S},
                   {gr->gru_Uz, (size_t)9 * S * S}, {gr->gru_Ur, (size_t)9 * S * S}, {gr->gru_U, (size_t)9 * S * S},
                   {gr->up_weight1, (size_t)25 * 64 * S}, {gr->up_weight2, (size_t)25 * 32 * 64}};
    const float* lo = z[0].q;
    const float* hi = z[0].q + z[0].n;
    for (const Z& e : z) { lo = std::min(lo, e.q); hi = std::max(hi, e.q + e.n); }
    // every other gradient of the struct is fully overwritten later in this call, so a range that also covers them
    // (bn_gamma/beta, up_weight3, out_W, out_b, biases) may be cleared as a whole -- but only memory the struct owns:
    const float* all_lo = lo;
    const float* all_hi = hi;
    const Z rest[] = {{gr->bn_gamma, (size_t)g->T * S}, {gr->bn_beta, (size_t)g->T * S}, {gr->up_weight3, (size_t)49 * 12 * 32},
                      {gr->out_W, 12}, {gr->out_b, 1}};
    size_t owned = 0;
    for (const Z& e : z) owned += e.n;
    for (const Z& e : rest) { all_lo = std::min(all_lo, e.q); all_hi = std::max(all_hi, e.q + e.n); owned += e.n; }
    if ((size_t)(all_hi - all_lo) == owned) {
      RGP_TRY(zb.add((void*)all_lo, owned * 4));
    } else {
      for (const Z& e : z) RGP_TRY(zb.add((void*)e.q, e.n * 4));
    }
  }
  RGP_TRY(zb.add(ws + b->dE.off, (size_t)P * sizeof(T)));                                  // dE's zero row
  if (!g->fold_head) RGP_TRY(zb.add(ws + b->dgp.off, b->dgp.bytes));                     // (the folded path overwrites dgp)
  // (buffers that only one later kernel of this call adds into: cleared here with the rest, in the one launch)
  if (g->fold_head && !ext_dy) RGP_TRY(zb.add(ws + b->dkf.off, b->dkf.bytes));           // dK: the wgrad's atomics
  if (persistent) RGP_TRY(zb.add(ws + b->bptt_cnt.off, b->bptt_cnt.bytes));             // phase counters: zeroed EVERY call
  if (ext_dy) {                                                                          // (the unused head's gradients)
    RGP_TRY(zb.add((void*)gr->up_weight3, (size_t)49 * 12 * 32 * 4));
    RGP_TRY(zb.add((void*)gr->out_W, 12 * 4));
    RGP_TRY(zb.add((void*)gr->out_b, 4));
  }
  RGP_TRY(zb.flush());

  if (ext_dy) {
    // the gradient w.r.t. the (batch-normalised) states comes from outside (cascade: the stride-7
    // transposed conv above the bottom cell); the head of this plan is unused, its gradients are zero
    // (stepwise: frame (b, t) of ext_dy is complete only behind bwd_step_ev[t] -- read in place, step by step, below)
    if (!stepwise) RGP_HIP(hipMemcpyAsync(Fp(b->dy), ext_dy, (size_t)M * S * 4, hipMemcpyDeviceToDevice, s));
  } else {
  // 1. d loss / d logits, d out_b
  dlogits_kernel<<<F, 256, 0, s>>>(loss_l2 ? logits : probs, labels, Fp(b->dz), Fp(b->frame_sum), 2401, 1.0f / (float)F, loss_l2);
  sum_kernel<<<1, 256, 0, s>>>(Fp(b->frame_sum), (float*)gr->out_b, F, 1.0f);
  if (g->fold_head) {
    // 2'-4'. the folded head (head_fold.hip.h): patches of dz -> dK (one wgrad launch: rows = the 7x7 positions, X = the
    // patches, dY = the padded BN(h) image, as deconv1's filter gradient) -> chain rule through the fold -> dy = Pm x K
    const long long tot = M * HF_PK;
    head_fold_patches_kernel<T><<<(int)std::min<long long>((tot + 255) / 256, 8192), 256, 0, s>>>(Fp(b->dz), Tp(b->pm), M);
    RGP_HIP(hipGetLastError());
    {
      WgradParams p = wg_params();
      p.X = Tp(b->pm); p.dY = Tp(g->hbn); p.dW = Fp(b->dkf);
      wgrad_grid(p, 1, 7, 7);
      p.x_sx = HF_PK; p.x_sy = 7 * HF_PK; p.x_img_stride = 49LL * HF_PK;
      p.y_sx = S; p.y_sy = 9 * S; p.y_org = 10 * S; p.y_img_stride = 81LL * S;
      p.koff = I(b->o_koff_c); p.M = M; p.N = S; p.nk = HF_PK / Elem<T>::BKE; p.ldw = S; p.k_valid = HF_PK;
      RGP_TRY((launch_wgrad<T, 1>(p, s)));
    }
    const float* hf = (const float*)(ws + g->hf_h.off);
    const float* gf = (const float*)(ws + g->gfold.off);
    hipStream_t sc = s;                                          // the chain's stream
    if (fork) {
      RGP_HIP(hipEventRecord(b->ev_fork, s));                    // dK is complete, the gradient buffers are zeroed
      RGP_HIP(hipStreamWaitEvent(b->side, b->ev_fork, 0));
      sc = b->side;
    }
    head_unfold_f1_kernel<<<(25 * 64 * S + 255) / 256, 256, 0, sc>>>(Fp(b->dkf), hf, (float*)gr->up_weight1, S);
    head_unfold_h_kernel<<<dim3(HF_HP * HF_HP, 25), 256, 0, sc>>>(Fp(b->dkf), b->w.up_weight1, Fp(b->dhp), S);
    head_fold_sum_kernel<<<(HF_HP * HF_HP * 64 + 255) / 256, 256, 0, sc>>>(Fp(b->dhp), Fp(b->dhf), HF_HP * HF_HP * 64, 25);
    head_unfold_f2_kernel<<<(25 * 32 * 64 + 255) / 256, 256, 0, sc>>>(Fp(b->dhf), gf, (float*)gr->up_weight2);
    head_unfold_g_kernel<<<49, 256, 0, sc>>>(Fp(b->dhf), b->w.up_weight2, Fp(b->dgp));
    head_unfold_grads_kernel<<<1, 256, 0, sc>>>(Fp(b->dgp), b->w.up_weight3, b->w.out_W, (float*)gr->up_weight3, (float*)gr->out_W);
    RGP_HIP(hipGetLastError());
    {
      IgemmParams p = make_params(b->b_hf, Tp(b->pm), ws, F);
      EpiParams e = make_epi(b->b_hf, Fp(b->dy), ws);
      RGP_TRY((launch_igemm<T, 1, 1, EpiStore<float, false, false>>(p, e, s)));
    }
  } else {
  // 2. folded 7x7 filter: wgrad -> dF3, d out_W ; dgrad -> dd2
  {
    const long long tot = (long long)F * 49 * 64;
    dz_block_kernel<T><<<(int)std::min<long long>((tot + 255) / 256, 4096), 256, 0, s>>>(Fp(b->dz), Tp(b->dzb), tot);
    RGP_HIP(hipGetLastError());
    RGP_HIP(hipMemsetAsync(ws + b->ptoep.off, 0, (size_t)7 * 16 * 704 * 4, s));
    WgradParams p = wg_params();
    p.X = Tp(b->dzb); p.dY = Tp(g->D2);
    wgrad_grid(p, 1, 49, 4);
    p.x_sx = 16; p.x_sy = 64; p.x_img_stride = 49LL * 64;
    p.y_sx = 16 * 32; p.y_sy = 55 * 32; p.y_img_stride = 55LL * 55 * 32;
    p.koff = I(b->o_koff_c); p.M = (long long)F * 196; p.N = 704; p.nk = 1; p.ldw = 704; p.k_valid = 16;
    // the 7 tap rows u of the Toeplitz filter are 7 problems of one geometry (dY shifted by u image rows): one launch
    p.y_org = 0;
    p.dW = Fp(b->ptoep);
    p.nz = 7;
    for (int u = 0; u < 7; ++u) { p.zy[u] = (long long)u * 55 * 32 * sizeof(T); p.zw[u] = (long long)u * 16 * 704; }
    RGP_TRY((launch_wgrad<T, 1>(p, s)));
    head_fold_toeplitz_kernel<<<(49 * 32 + 255) / 256, 256, 0, s>>>(Fp(b->ptoep), Fp(b->dgp));
    RGP_HIP(hipGetLastError());
  }
  head_unfold_grads_kernel<<<1, 256, 0, s>>>(Fp(b->dgp), b->w.up_weight3, b->w.out_W, (float*)gr->up_weight3, (float*)gr->out_W);
  head_fold_dgrad_kernel<T><<<dim3(7, F), 256, 0, s>>>(Fp(b->dz), Fp(b->gp), Tp(b->dd2));
  RGP_HIP(hipGetLastError());
  // 3. deconv2: wgrad (dF2[a,b,o,c] = sum dd2[2i+a,2j+b,o] d1[i,j,c]) and dgrad
  {  // rows = the 23x23 positions of d1; X = dd2 at the stride-2 row origins (taps of b_d2), dY = the padded d1 image
    WgradParams p = wg_params();
    p.X = Tp(b->dd2); p.dY = Tp(g->D1); p.dW = (float*)gr->up_weight2;
    wgrad_grid(p, 1, 23, 23);
    p.x_sx = 2 * 32; p.x_sy = 2 * 49 * 32; p.x_img_stride = 2401LL * 32;
    p.y_sx = 64; p.y_sy = 27 * 64; p.y_org = (2 * 27 + 2) * 64; p.y_img_stride = 27LL * 27 * 64;
    p.koff = I(b->b_d2.koff_off); p.M = M2; p.N = 64; p.nk = b->b_d2.nk; p.ldw = 64; p.k_valid = 25 * 32;
    if (sizeof(T) == 2) RGP_TRY((launch_wgrad<T, 2>(p, s)));
    else RGP_TRY((launch_wgrad<T, 1>(p, s)));
  }
  {
    IgemmParams p = make_params(b->b_d2, Tp(b->dd2), ws, F);
    EpiParams e = make_epi(b->b_d2, Tp(b->dd1), ws);
    if (sizeof(T) == 2) RGP_TRY((launch_igemm<T, 2, 1, EpiStore<T, false, false>>(p, e, s)));
    else RGP_TRY((launch_igemm<T, 1, 1, EpiStore<T, false, false>>(p, e, s)));
  }
  // 4. deconv1: wgrad (dF1[a,b,o,c] = sum dd1[3i+a,3j+b,o] y[i,j,c]) and dgrad -> dy (fp32)
  {  // rows = the 7x7 positions of y; X = dd1 at the stride-3 row origins (taps of b_d1), dY = the padded BN(h) image
    WgradParams p = wg_params();
    p.X = Tp(b->dd1); p.dY = Tp(g->hbn); p.dW = (float*)gr->up_weight1;
    wgrad_grid(p, 1, 7, 7);
    p.x_sx = 3 * 64; p.x_sy = 3 * 23 * 64; p.x_img_stride = 529LL * 64;
    p.y_sx = S; p.y_sy = 9 * S; p.y_org = 10 * S; p.y_img_stride = 81LL * S;
    p.koff = I(b->b_d1.koff_off); p.M = M; p.N = S; p.nk = b->b_d1.nk; p.ldw = S; p.k_valid = 25 * 64;
    RGP_TRY((launch_wgrad<T, 1>(p, s)));
  }
  {
    IgemmParams p = make_params(b->b_d1, Tp(b->dd1), ws, F);
    EpiParams e = make_epi(b->b_d1, Fp(b->dy), ws);
    RGP_TRY((launch_igemm<T, 1, 1, EpiStore<float, false, false>>(p, e, s)));
  }
  }  // !fold_head
  }  // !ext_dy
  // 5. per-timestep batch-norm
  const float inv = 1.0f / sqrtf(1.0f + 1e-3f);
  if (!stepwise)
    bn_bwd_kernel<<<dim3(S / 8, T_), 256, 0, s>>>(Fp(b->dy), Fp(g->hall), b->w.bn_gamma, (float*)gr->bn_gamma,
                                                    (float*)gr->bn_beta, Fp(b->dh_head), B, T_, S, inv, 0);
  // 6. BPTT: t = T-1 .. 0 -- one persistent launch where the plan allows it (bf16, the reference cell, <= 64 clips)
  const int ew_blocks = (int)std::min<size_t>((st + 255) / 256, 4096);
  // bn_gamma/beta, up_weight1..3, out_W, out_b are final here.  Their event lets the host start the group's all-reduce on
  // another stream -- i.e. an RCCL kernel that runs NEXT TO the BPTT launch.  That launch needs all its workgroups resident
  // together, one per CU (157 KB of LDS each): a collective's workgroup that reaches a CU first keeps a member off it until
  // the collective ends, which takes as long as the slowest peer rank.  So the early release is for launches that leave CUs
  // free (config 4: 8 clips per GPU = 64 workgroups); a launch that needs more than n_cu - RGP_RCCL_CU_RESERVE CUs
  // releases the group only behind itself (include/rgp.h, rgp_grcn_wait_grads).
  const bool top_early = (!persistent || grads_top_early(g)) && !stepwise;     // (stepwise: the batch-norm gradients end with the loop)
  // h_{t-1} and r . h_{t-1} of every step, halo-padded [t][b][9][9][S]: the recurrent filter gradients' X operand (step 8)
  bool h_rh_padded = false;
  auto pad_h_rh = [&](hipStream_t q) -> int {
    const long long tot = (long long)T_ * B * 49 * S;
    const int nb = (int)std::min<long long>((tot + 255) / 256, 8192);
    pad_h_rh_kernel<T><<<nb, 256, 0, q>>>(Fp(g->rall), Fp(g->hall), Tp(b->hp_all), Tp(b->rhp_all), I(b->o_y), tot, S);
    RGP_HIP(hipGetLastError());
    return RGP_OK;
  };
  if (fork) {
    // the TOP group is final once BOTH the chain (side stream) and the batch-norm backward (this stream) are: its event is
    // recorded on the side stream behind an event of this one -- ahead of the BPTT launch in queue order, as before
    RGP_HIP(hipEventRecord(b->ev_bn, s));
    RGP_HIP(hipStreamWaitEvent(b->side, b->ev_bn, 0));
    if (mark && top_early) RGP_HIP(hipEventRecord(b->grad_ev[0], b->side));
    RGP_HIP(hipEventRecord(b->ev_join, b->side));
    if (wfork) {                                                           // forward-only operands: in the BPTT launch's shadow
      RGP_TRY(pad_h_rh(b->side));
      h_rh_padded = true;
    }
  } else if (mark && top_early) {
    RGP_HIP(hipEventRecord(b->grad_ev[0], s));
  }
  if (persistent) {
    BpttParams q;
    q.w_c = (const bf16_t*)(ws + b->b_c.w_off);
    q.w_zr = (const bf16_t*)(ws + b->b_zr.w_off);
    q.dh_head = Fp(b->dh_head); q.hall = Fp(g->hall); q.uall = Fp(g->uall); q.rall = Fp(g->rall); q.call = Fp(g->call);
    q.dxpre = Fp(b->dxpre);
    q.xch_c = (bf16_t*)(ws + b->xch_c.off); q.xch_z = (bf16_t*)(ws + b->xch_z.off); q.xch_r = (bf16_t*)(ws + b->xch_r.off);
    q.cnt = (unsigned*)(ws + b->bptt_cnt.off);
    q.B = B; q.T = T_; q.NC = g->seq_nc; q.ngroups = g->seq_groups;
    q.err = g->err_host;
    q.skip_member = (g->fault & 2) ? 7 : -1;
    g->fault &= ~2;
    RGP_REQUIRE(b->b_c.K == 9 * S && b->b_zr.K == 18 * S, "convgru_bptt: unexpected filter packing");
    PersistentLaunch guard(s);
    RGP_TRY(guard.status());
    if (g->seq_nc == 1) {
      RGP_TRY(ensure_dyn_smem((const void*)convgru_bptt_kernel<4>, SEQ_SMEM));
      convgru_bptt_kernel<4><<<g->seq_groups * 8, SEQ_NT, SEQ_SMEM, s>>>(q);
    } else {
      RGP_TRY(ensure_dyn_smem((const void*)convgru_bptt_kernel<7>, SEQ_SMEM));
      convgru_bptt_kernel<7><<<g->seq_groups * 8, SEQ_NT, SEQ_SMEM, s>>>(q);
    }
    RGP_HIP(hipGetLastError());
    RGP_TRY(guard.commit());
  }
  if (fork) RGP_HIP(hipStreamWaitEvent(s, b->ev_join, 0));              // the chain has ended (it had the whole BPTT launch to do so)
  if (mark && !top_early && !stepwise) RGP_HIP(hipEventRecord(b->grad_ev[0], s));     // full-chip launch: the TOP group leaves behind it
  // Per-step BPTT (plans without the persistent launch: fp32, the cascade's 256-channel bottom cell, the fall-back).  A step is
  // four dependent launches; its two dgrad convolutions have B x 49 rows -- a few dozen 64 x 64 tiles, each walking the whole
  // K = 9 S / 18 S alone: split K over 2 / 3 blocks that add their partial sums with float atomics (drh zeroed by part 1,
  // the carry holds part 1's term already).  16 x 35, S = 256: 23.9 + 42.5 us -> see profiles/r05_ab_cascade.txt
  // (more splits cost more in atomics than they save: S = 256 at 16 x 35, (c, zr) = (1,1) 9.55, (2,2) 9.35, (2,3) 9.35, (2,4) 9.70,
  // (3,3) 9.68, (3,6) 10.2 ms per cascade forward + backward, profiles/r05_ab_cascade.txt)
  const int ks_c = std::max(1, std::min(8, dev_knob("RGP_BPTT_KSPLIT_C", b->b_c.nk >= 16 ? 2 : 1)));
  const int ks_zr = std::max(1, std::min(8, dev_knob("RGP_BPTT_KSPLIT_ZR", b->b_zr.nk >= 48 ? 3 : b->b_zr.nk >= 16 ? 2 : 1)));
  for (int t = T_ - 1; t >= 0 && !persistent; --t) {
    const float* h_prev = Fp(g->hall) + (size_t)t * st;
    if (stepwise) RGP_HIP(hipStreamWaitEvent(s, g->bwd_step_ev[t], 0));      // frame (b, t) of ext_dy is complete
    gru_bwd1_kernel<T><<<ew_blocks, 256, 0, s>>>(Fp(b->dh_head) + (size_t)t * st, Fp(b->dh_carry), h_prev,
                                                  Fp(g->uall) + (size_t)t * st, Fp(g->call) + (size_t)t * st, Fp(b->dxpre),
                                                  Tp(b->dcp_pad), I(g->o_pad9_S), B, T_, t, S, t == T_ - 1,
                                                  ks_c > 1 ? Fp(b->drh) : nullptr, stepwise ? ext_dy : nullptr,
                                                  b->w.bn_gamma + (size_t)t * S, inv);
    RGP_HIP(hipGetLastError());
    {
      IgemmParams p = make_params(b->b_c, Tp(b->dcp_pad), ws, B);
      EpiParams e = make_epi(b->b_c, Fp(b->drh), ws);
      if (ks_c > 1) RGP_TRY((launch_igemm<T, 1, 1, EpiAtomicAddF32>(p, e, s, ks_c)));
      else RGP_TRY((launch_igemm<T, 1, 1, EpiStore<float, false, false>>(p, e, s)));
    }
    gru_bwd2_kernel<T><<<ew_blocks, 256, 0, s>>>(Fp(b->drh), Fp(b->dh_carry), h_prev, Fp(g->rall) + (size_t)t * st,
                                                  Fp(b->dxpre), Tp(b->dzr_pad), I(b->o_pad2S), B, T_, t, S);
    RGP_HIP(hipGetLastError());
    {
      IgemmParams p = make_params(b->b_zr, Tp(b->dzr_pad), ws, B);
      EpiParams e = make_epi(b->b_zr, Fp(b->dh_carry), ws);
      if (ks_zr > 1) RGP_TRY((launch_igemm<T, 1, 1, EpiAtomicAddF32>(p, e, s, ks_zr)));
      else RGP_TRY((launch_igemm<T, 1, 1, EpiAccumF32>(p, e, s)));
    }
  }
  if (stepwise) {                                                         // the batch-norm's own gradients, off the step chain
    bn_bwd_kernel<<<dim3(S / 8, T_), 256, 0, s>>>(ext_dy, Fp(g->hall), b->w.bn_gamma, (float*)gr->bn_gamma, (float*)gr->bn_beta,
                                                    Fp(b->dh_head), B, T_, S, inv, 0);
    if (mark) RGP_HIP(hipEventRecord(b->grad_ev[0], s));
  }
  // 7. hoisted input convs: the padded gradient image both branches below read
  {
    const long long tot = (long long)F * 49 * 3 * S;
    pad_rows_kernel<T><<<(int)std::min<long long>((tot + 255) / 256, 8192), 256, 0, s>>>(Fp(b->dxpre), Tp(b->dxpre_pad),
                                                                                      I(b->o_pad3S), tot, 3 * S);
    RGP_HIP(hipGetLastError());
  }
  hipStream_t sw = s;                                                      // the stream of the ConvGRU filter gradients
  if (wfork) {
    RGP_HIP(hipEventRecord(b->ev_wfork, s));
    RGP_HIP(hipStreamWaitEvent(b->side, b->ev_wfork, 0));
    sw = b->side;
  }
  // 8. weight gradients of the recurrence, hoisted over all T steps: wgrad_kernel straight on the halo-padded operand
  //    images (no im2col / transposed copies).  Gate g's gradient columns are [gS, (g+1)S) of the padded dxpre image.
  {
    if (!h_rh_padded) RGP_TRY(pad_h_rh(sw));
    WgradParams p = wg_params();
    p.y_sx = 3 * S; p.y_sy = 27 * S; p.y_org = 30 * S;
    p.N = S; p.ldw = S; p.M = M;
    // input filters: rows = frames x 7x7, X = the padded projected features E
    wgrad_grid(p, 1, 7, 7);
    p.X = Tp(g->E); p.x_sx = P; p.x_sy = 9 * P; p.x_img_stride = 81LL * P; p.y_img_stride = 243LL * S;
    p.koff = I(g->xconv.koff_off); p.nk = g->xconv.nk; p.k_valid = 9 * P;
    float* dWx[3] = {(float*)gr->gru_Wz, (float*)gr->gru_Wr, (float*)gr->gru_W};
    // the three gates are three problems of one geometry (dY columns q S, their own dW): one grouped launch
    p.dY = Tp(b->dxpre_pad); p.dW = dWx[0]; p.nz = 3;
    for (int q = 0; q < 3; ++q) { p.zy[q] = (long long)q * S * sizeof(T); p.zw[q] = dWx[q] - dWx[0]; }
    RGP_TRY((launch_wgrad<T, 1>(p, sw)));
    // recurrent filters: image = clip b, z = step t (h images are step-major, gradient frames clip-major)
    wgrad_grid(p, T_, 7, 7);
    p.x_sx = S; p.x_sy = 9 * S; p.x_sz = B * 81 * S; p.x_img_stride = 81LL * S;
    p.y_sz = 243 * S; p.y_img_stride = (long long)T_ * 243 * S;
    p.koff = I(g->gzr.koff_off); p.nk = g->gzr.nk; p.k_valid = 9 * S;
    float* dWh[3] = {(float*)gr->gru_Uz, (float*)gr->gru_Ur, (float*)gr->gru_U};
    p.X = Tp(b->hp_all); p.dY = Tp(b->dxpre_pad); p.dW = dWh[0]; p.nz = 3;
    for (int q = 0; q < 3; ++q) {
      p.zx[q] = q < 2 ? 0 : (const char*)Tp(b->rhp_all) - (const char*)Tp(b->hp_all);
      p.zy[q] = (long long)q * S * sizeof(T); p.zw[q] = dWh[q] - dWh[0];
    }
    RGP_TRY((launch_wgrad<T, 1>(p, sw)));
    if (mark) RGP_HIP(hipEventRecord(b->grad_ev[1], sw));              // the six ConvGRU filters are final
    if (wfork) RGP_HIP(hipEventRecord(b->ev_wjoin, sw));
  }
  // 9. the input convolutions' dgrad -> dE, then the projection's gradients: one row per (frame, position), X = the
  //    1024-channel C3D rows, dY = dE behind its zero row
  {
    IgemmParams p = make_params(b->b_x, Tp(b->dxpre_pad), ws, F);
    EpiParams e = make_epi(b->b_x, Tp(b->dE) + P, ws);        // row 0 of the buffer stays zero (wgrad_kernel's dY contract)
    RGP_TRY((launch_igemm<T, 1, 1, EpiStore<T, false, false>>(p, e, s)));
  }
  {
    WgradParams p = wg_params();
    p.X = Tp(g->xt); p.dY = Tp(b->dE); p.dW = (float*)gr->proj_c3d_W;
    wgrad_grid(p, 1, 1, (int)M);
    p.x_sx = 1024; p.y_sx = P; p.y_org = P;
    p.koff = I(b->o_koff_c); p.M = M; p.N = P; p.nk = 1024 / Elem<T>::BKE; p.ldw = P; p.k_valid = 1024;
    RGP_TRY((launch_wgrad<T, 1>(p, s)));
    dense_colsum_kernel<T><<<(int)std::min<long long>((M + 63) / 64, 1024), 256, 0, s>>>(Tp(b->dE) + P, M, P, (float*)gr->proj_c3d_b);
    RGP_HIP(hipGetLastError());
  }
  if (wfork) RGP_HIP(hipStreamWaitEvent(s, b->ev_wjoin, 0));
  if (mark) {
    RGP_HIP(hipEventRecord(b->grad_ev[2], s));
    b->grad_ev_recorded = true;
  } else {
    // captured into a graph: nothing was recorded, and events of an earlier eager backward say nothing about a replay --
    // rgp_grcn_wait_grads must not hand them out (RGP_ESTATE until the next eager backward)
    b->grad_ev_recorded = false;
  }
  return RGP_OK;
}

template <typename T>
int pack_impl(rgp_grcn* g, const rgp_grcn_weights* w, hipStream_t s, hipStream_t sc) {
  GrcnBwd* b = g->bwd;
  char* ws = g->ws;
  const int S = g->S, P = g->P;
  PackBatch<T> pk(ws, s);                                                     // one launch for the ten packs
  // (no memset: the areas are zero from bind time outside the positions a pack writes, rgp_grcn.hip set_weights_impl)
  RGP_TRY(pk.add(b->b_px, w->proj_c3d_W, 512, 0));            // d = 0: feature channels 0, 2, 4, ...
  RGP_TRY(pk.add(b->b_px, w->proj_c3d_W + P, 512, 512));      // d = 1: feature channels 1, 3, 5, ...
  if (!g->fold_head) {
    RGP_TRY(pk.add(b->b_d2, w->up_weight2, 64, 0));
    RGP_TRY(pk.add(b->b_d1, w->up_weight1, S, 0));
  }
  RGP_TRY(pk.add(b->b_c, w->gru_U, S, 0));
  RGP_TRY(pk.add(b->b_zr, w->gru_Uz, S, 0, 0, 1));
  RGP_TRY(pk.add(b->b_zr, w->gru_Ur, S, 0, S, 1));
  RGP_TRY(pk.add(b->b_x, w->gru_Wz, P, 0, 0, 1));
  RGP_TRY(pk.add(b->b_x, w->gru_Wr, P, 0, S, 1));
  RGP_TRY(pk.add(b->b_x, w->gru_W, P, 0, 2 * S, 1));
  RGP_TRY(pk.flush());
  if (g->fold_head) {                                          // K of the folded head: set_weights_impl built it on stream sc
    PackBatch<T> pk2(ws, sc);
    RGP_TRY(pk2.add(b->b_hf, (const float*)(ws + g->hf_k.off), S, 0));
    RGP_TRY(pk2.flush());
  }
  // Gp[u,v,c] = G[6-u,6-v,c] in fp32 for the folded-filter dgrad (G itself is in g->gfold)
  // (49*32 elements; reuse the pack kernel with T=float semantics is overkill: tiny copy kernel)
  return RGP_OK;
}

__global__ void flip_fold_kernel(const float* __restrict__ gfold, float* __restrict__ gp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 49 * 32) return;
  const int tap = i / 32, c = i % 32;
  gp[i] = gfold[(48 - tap) * 32 + c];      // (6-u)*7 + (6-v) = 48 - (u*7+v)
}

}  // namespace

int grcn_bwd_plan(rgp_grcn* g, Arena& a) {
  GrcnBwd* b = new GrcnBwd();
  g->bwd = b;
  const int B = g->B, T_ = g->T, S = g->S, P = g->P, F = g->F, dtype = g->dtype, es = esize(dtype);
  b->M = (long long)F * 49;
  b->M2 = (long long)F * 529;
  b->Mp = (b->M + 63) / 64 * 64;
  b->M2p = (b->M2 + 63) / 64 * 64;
  if (b->M2p * 800 >= (1LL << 31)) return set_err(RGP_EINVAL, "rgp_grcn_create: B*T too large for the backward plan");
  bool ok = true;
  // ---- dgrad convs
  {  // deconv2: dd1[i,j,c] = sum_{a,b,o} dd2[2i+a, 2j+b, o] F2[a,b,o,c]   (gaze_grcn.py:336-343)
    ConvDesc& d = b->b_d2;
    d.Mw = 529; d.N = 64; d.in_img_stride = 2401LL * 32; d.out_img_stride = 529LL * 64;
    std::vector<int> tapoff, fidx;
    for (int i = 0; i < 23; ++i) for (int j = 0; j < 23; ++j) { d.in_tab.push_back(((2 * i) * 49 + 2 * j) * 32); d.out_tab.push_back((i * 23 + j) * 64); }
    for (int aa = 0; aa < 5; ++aa) for (int bb = 0; bb < 5; ++bb) { tapoff.push_back((aa * 49 + bb) * 32); fidx.push_back(aa * 5 + bb); }
    ok &= build_k_schedule(d, tapoff, fidx, 32, dtype);
    d.s_tap = 32LL * 64; d.s_n = 1; d.s_c = 64;
  }
  {  // deconv1: dy[i,j,c] = sum_{a,b,o} dd1[3i+a, 3j+b, o] F1[a,b,o,c]     (gaze_grcn.py:326-333)
    ConvDesc& d = b->b_d1;
    d.Mw = 49; d.N = S; d.in_img_stride = 529LL * 64; d.out_img_stride = 49LL * S;
    std::vector<int> tapoff, fidx;
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 7; ++j) { d.in_tab.push_back(((3 * i) * 23 + 3 * j) * 64); d.out_tab.push_back((i * 7 + j) * S); }
    for (int aa = 0; aa < 5; ++aa) for (int bb = 0; bb < 5; ++bb) { tapoff.push_back((aa * 23 + bb) * 64); fidx.push_back(aa * 5 + bb); }
    ok &= build_k_schedule(d, tapoff, fidx, 64, dtype);
    d.s_tap = 64LL * S; d.s_n = 1; d.s_c = S;
  }
  // 3x3 SAME dgrads: correlation of the halo-padded gradient image with the 180-degree rotated,
  // in/out-swapped filter:  dx[p, ci] = sum_{t', o} dy_pad[p + t', o] * W[8 - t'][ci][o]
  auto dgrad3x3 = [&](ConvDesc& d, int Cgrad, int Nout, int filt_cin, int out_cols) {
    d.Mw = 49; d.N = Nout; d.in_img_stride = 81LL * Cgrad; d.out_img_stride = 49LL * out_cols;
    std::vector<int> tapoff, fidx;
    for (int y = 0; y < 7; ++y) for (int x = 0; x < 7; ++x) { d.in_tab.push_back((y * 9 + x) * Cgrad); d.out_tab.push_back((y * 7 + x) * out_cols); }
    for (int t = 0; t < 9; ++t) { tapoff.push_back(((t / 3) * 9 + (t % 3)) * Cgrad); fidx.push_back(8 - t); }
    bool r = build_k_schedule(d, tapoff, fidx, Cgrad, dtype);
    d.cin_src = S;                           // each source filter contributes S gate columns
    d.s_tap = (long long)filt_cin * S; d.s_n = S; d.s_c = 1;   // HWIO filter [3,3,filt_cin,S]
    return r;
  };
  ok &= dgrad3x3(b->b_c, S, S, S, S);
  ok &= dgrad3x3(b->b_zr, 2 * S, S, S, S);
  ok &= dgrad3x3(b->b_x, 3 * S, P, P, P);
  {  // d rows[m][d*512+c] = sum_p dE[m][p] W[c*2+d][p]   (gaze_grcn.py:225-254; rows order of rgp_c3d_forward)
    ConvDesc& d = b->b_px;
    d.Mw = 1; d.N = 1024; d.in_img_stride = P; d.out_img_stride = 1024; d.in_tab = {0}; d.out_tab = {0};
    ok &= build_k_schedule(d, {0}, {0}, P, dtype);
    d.s_tap = 0; d.s_n = 2LL * P; d.s_c = 1;
  }
  {  // folded head: dy[(f,m,n), s] = sum_k Pm[(f,m,n), k] K[k, s]   (head_fold.hip.h; K [361][S] fp32, rows 361..383 zero)
    ConvDesc& d = b->b_hf;
    d.Mw = 49; d.N = S; d.in_img_stride = 49LL * HF_PK; d.out_img_stride = 49LL * S;
    for (int pos = 0; pos < 49; ++pos) { d.in_tab.push_back(pos * HF_PK); d.out_tab.push_back(pos * S); }
    ok &= build_k_schedule(d, {0}, {0}, HF_PK, dtype);
    d.cin_src = HF_KP * HF_KP;
    d.s_tap = 0; d.s_n = 1; d.s_c = S;
  }
  if (!ok) return set_err(RGP_EINVAL, "rgp_grcn_create: backward K schedule failed");
  for (ConvDesc* d : {&b->b_d2, &b->b_d1, &b->b_c, &b->b_zr, &b->b_x, &b->b_px, &b->b_hf}) d->reserve(a, dtype);

  // ---- small tables
  for (int y = 0; y < 7; ++y) for (int x = 0; x < 7; ++x) {
    b->t_y.push_back(((y + 1) * 9 + x + 1) * S);            // interior of a 9x9xS image
    b->t_pad3S.push_back(((y + 1) * 9 + x + 1) * 3 * S);    // interior of a 9x9x3S image
    b->t_pad2S.push_back(((y + 1) * 9 + x + 1) * 2 * S);    // interior of a 9x9x2S image
  }
  for (int k = 0; k < 1024 / bke(dtype); ++k) b->koff_c.push_back(k * bke(dtype));
  b->o_y = put(a, b->t_y); b->o_pad3S = put(a, b->t_pad3S); b->o_pad2S = put(a, b->t_pad2S); b->o_koff_c = put(a, b->koff_c);

  // ---- buffers
  const size_t st = (size_t)B * 49 * S * 4;
  b->dz = take(a, (size_t)F * 2401 * 4);
  b->frame_sum = take(a, (size_t)F * 4);
  b->dgp = take(a, 50 * 32 * 4);
  b->gp = take(a, 50 * 32 * 4);
  if (g->fold_head) {
    b->pm = take(a, (size_t)F * 49 * HF_PK * es);
    b->dkf = take(a, (size_t)HF_PK * S * 4);
    b->dhf = take(a, (size_t)HF_HP * HF_HP * 64 * 4);
    b->dhp = take(a, (size_t)25 * HF_HP * HF_HP * 64 * 4);
  } else {
    b->dd2 = take(a, (size_t)F * 2401 * 32 * es + 4096);
    b->dd1 = take(a, (size_t)F * 529 * 64 * es + 4096);
  }
  b->dy = take(a, (size_t)F * 49 * S * 4);
  b->dh_head = take(a, st * T_);
  b->dh_carry = take(a, st);
  b->drh = take(a, st);
  b->dcp_pad = take(a, (size_t)B * 81 * S * es);
  b->dzr_pad = take(a, (size_t)B * 81 * 2 * S * es);
  b->dxpre = take(a, (size_t)F * 49 * 3 * S * 4);
  b->dxpre_pad = take(a, (size_t)F * 81 * 3 * S * es);
  b->dE = take(a, (size_t)(b->M + 1) * P * es);        // + a leading zero row
  if (g->seq_groups > 0) {
    b->xch_c = take(a, (size_t)g->seq_groups * 98 * 128 * 2);
    b->xch_z = take(a, (size_t)g->seq_groups * 98 * 128 * 2);
    b->xch_r = take(a, (size_t)g->seq_groups * 98 * 128 * 2);
    b->bptt_cnt = take(a, (size_t)g->seq_groups * 2 * T_ * 4);
  }
  if (!g->fold_head) {
    b->dzb = take(a, ((size_t)F * 49 * 64 + 256) * es);
    b->ptoep = take(a, (size_t)7 * 16 * 704 * 4);
  }
  b->hp_all = take(a, (size_t)T_ * B * 81 * S * es);
  b->rhp_all = take(a, (size_t)T_ * B * 81 * S * es);
  b->sq_partial = take(a, SQ_BLOCKS * 4);
  return RGP_OK;
}

int grcn_bwd_upload(rgp_grcn* g, hipStream_t s) {
  GrcnBwd* b = g->bwd;
  for (ConvDesc* d : {&b->b_d2, &b->b_d1, &b->b_c, &b->b_zr, &b->b_x, &b->b_px, &b->b_hf}) RGP_TRY(upload_desc(*d, g->ws, s));
  auto up = [&](const std::vector<int>& t, size_t off) -> int {
    RGP_HIP(hipMemcpyAsync(g->ws + off, t.data(), t.size() * 4, hipMemcpyHostToDevice, s));
    return RGP_OK;
  };
  RGP_TRY(up(b->t_y, b->o_y)); RGP_TRY(up(b->t_pad3S, b->o_pad3S)); RGP_TRY(up(b->t_pad2S, b->o_pad2S));
  RGP_TRY(up(b->koff_c, b->o_koff_c));
  return RGP_OK;
}

int grcn_bwd_pack(rgp_grcn* g, const rgp_grcn_weights* w, hipStream_t s, hipStream_t sc) {
  g->bwd->w = *w;
  RGP_TRY(g->dtype == RGP_BF16 ? pack_impl<bf16_t>(g, w, s, sc) : pack_impl<float>(g, w, s, sc));
  flip_fold_kernel<<<(49 * 32 + 255) / 256, 256, 0, sc>>>((const float*)(g->ws + g->gfold.off), (float*)(g->ws + g->bwd->gp.off));
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

// rgp_grcn_set_weights of a training plan: the head's fold (G -> H -> K: four dependent kernels) and the packs that read it run
// on the plan's side stream beside the packs of the other filters.  *sc = the side stream, forked behind everything queued on s
// (the optimizer step that wrote the weights); s itself when s is being captured and the side stream does not exist yet.
int grcn_bwd_fork_fold(rgp_grcn* g, hipStream_t s, hipStream_t* sc) {
  GrcnBwd* b = g->bwd;
  *sc = s;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  const bool capturing = !(hipStreamIsCapturing(s, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone);
  if (!g->fold_head || !dev_knob("RGP_BWD_FORK", 1)) return RGP_OK;
  RGP_TRY(make_side_stream(b, !capturing));
  if (!b->side) return RGP_OK;
  RGP_HIP(hipEventRecord(b->ev_fork, s));
  RGP_HIP(hipStreamWaitEvent(b->side, b->ev_fork, 0));
  *sc = b->side;
  return RGP_OK;
}

int grcn_bwd_join_fold(rgp_grcn* g, hipStream_t s) {
  GrcnBwd* b = g->bwd;
  RGP_HIP(hipEventRecord(b->ev_join, b->side));
  RGP_HIP(hipStreamWaitEvent(s, b->ev_join, 0));
  return RGP_OK;
}

void grcn_bwd_destroy(rgp_grcn* g) {
  delete g->bwd;
  g->bwd = nullptr;
}

extern "C" {

int rgp_grcn_backward(rgp_grcn_t* g, const float* logits, const float* probs, const float* labels,
                      const rgp_grcn_weights* grads, int loss_type, rgp_stream_t stream) {
  RGP_REQUIRE(g && logits && labels && grads, "rgp_grcn_backward: null argument");
  if (!g->ws) return set_err(RGP_EWORKSPACE, "rgp_grcn: workspace not bound");
  if (!g->save || !g->bwd) return set_err(RGP_ESTATE, "rgp_grcn_backward: plan was created without save_for_backward");
  if (!g->weights_set) return set_err(RGP_ESTATE, "rgp_grcn: weights not set");
  RGP_TRY(grcn_check_error(g));
  RGP_REQUIRE(loss_type == 0 || loss_type == 1, "rgp_grcn_backward: loss_type %d (0 xentropy, 1 l2)", loss_type);
  RGP_REQUIRE(loss_type == 1 || probs, "rgp_grcn_backward: xentropy needs the softmax maps");
  const float* const* ptrs = (const float* const*)grads;
  for (size_t i = 0; i < sizeof(rgp_grcn_weights) / sizeof(float*); ++i)
    RGP_REQUIRE(ptrs[i], "rgp_grcn_backward: gradient pointer %zu is null", i);
  hipStream_t s = (hipStream_t)stream;
  return g->dtype == RGP_BF16 ? backward_impl<bf16_t>(g, probs, logits, labels, grads, loss_type, s)
                              : backward_impl<float>(g, probs, logits, labels, grads, loss_type, s);
}

int rgp_grcn_backward_from_states(rgp_grcn_t* g, const float* d_states, const rgp_grcn_weights* grads, rgp_stream_t stream) {
  RGP_REQUIRE(g && d_states && grads, "rgp_grcn_backward_from_states: null argument");
  if (!g->ws || !g->save || !g->bwd || !g->weights_set)
    return set_err(RGP_ESTATE, "rgp_grcn_backward_from_states: needs a save_for_backward plan with weights and a forward");
  const float* const* ptrs = (const float* const*)grads;
  for (size_t i = 0; i < sizeof(rgp_grcn_weights) / sizeof(float*); ++i)
    RGP_REQUIRE(ptrs[i], "rgp_grcn_backward_from_states: gradient pointer %zu is null", i);
  hipStream_t s = (hipStream_t)stream;
  return g->dtype == RGP_BF16 ? backward_impl<bf16_t>(g, nullptr, nullptr, nullptr, grads, 0, s, d_states)
                              : backward_impl<float>(g, nullptr, nullptr, nullptr, grads, 0, s, d_states);
}

int rgp_grcn_persistent_workgroups(const rgp_grcn_t* g) {
  return (g && g->dtype == RGP_BF16 && seq_persistent_ok(g) && dev_knob("RGP_SEQ", 1)) ? g->seq_groups * 8 : 0;
}

int rgp_grcn_grads_top_early(const rgp_grcn_t* g) {
  if (!g) return 0;
  return rgp_grcn_persistent_workgroups(g) > 0 ? (grads_top_early(g) ? 1 : 0) : 1;
}

int rgp_grcn_wait_grads(rgp_grcn_t* g, int group, rgp_stream_t waiting_stream) {
  RGP_REQUIRE(g && group >= 0 && group <= 2, "rgp_grcn_wait_grads: bad arguments");
  if (!g->save || !g->bwd || !g->bwd->grad_ev_recorded)
    return set_err(RGP_ESTATE, "rgp_grcn_wait_grads: no backward has run on this plan");
  RGP_HIP(hipStreamWaitEvent((hipStream_t)waiting_stream, g->bwd->grad_ev[group], 0));
  return RGP_OK;
}

int rgp_grcn_backward_input(rgp_grcn_t* g, float* d_rows, rgp_stream_t stream) {
  RGP_REQUIRE(g && d_rows, "rgp_grcn_backward_input: null argument");
  if (!g->ws || !g->save || !g->bwd || !g->weights_set) return set_err(RGP_ESTATE, "rgp_grcn_backward_input: call after rgp_grcn_backward");
  hipStream_t s = (hipStream_t)stream;
  GrcnBwd* b = g->bwd;
  IgemmParams p = make_params(b->b_px, g->ws + b->dE.off + (size_t)g->P * esize(g->dtype), g->ws, (int)b->M);
  EpiParams e = make_epi(b->b_px, d_rows, g->ws);
  return g->dtype == RGP_BF16 ? launch_igemm<bf16_t, 1, 1, EpiStore<float, false, false>>(p, e, s)
                              : launch_igemm<float, 1, 1, EpiStore<float, false, false>>(p, e, s);
}

int rgp_adam_clip_step(float* params, const float* grads, float* m, float* v, long long n, float* workspace, int step,
                       float lr, float beta1, float beta2, float eps, float max_grad_norm, float* grad_norm_out,
                       rgp_stream_t stream) {
  RGP_REQUIRE(params && grads && m && v && workspace && n > 0 && step >= 0, "rgp_adam_clip_step: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  sqnorm_partial_kernel<<<SQ_BLOCKS, 256, 0, s>>>(grads, n, workspace);
  const double t = (double)step + 1.0;
  const float lr_t = (float)((double)lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t)));
  const int blocks = (int)std::min<long long>((n + 255) / 256, 4096);
  adam_clip_kernel<<<blocks, 256, 0, s>>>(params, grads, m, v, n, workspace, SQ_BLOCKS, max_grad_norm, lr_t, beta1, beta2,
                                          eps, grad_norm_out, nullptr);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

int rgp_global_sqnorm(const float* grads, long long n, float* partials, rgp_stream_t stream) {
  RGP_REQUIRE(grads && partials && n > 0, "rgp_global_sqnorm: bad arguments");
  sqnorm_partial_kernel<<<SQ_BLOCKS, 256, 0, (hipStream_t)stream>>>(grads, n, partials);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

int rgp_adam_clip_step_ext(float* params, const float* grads, float* m, float* v, long long n, const float* partials,
                           int n_partials, int step, float lr, float beta1, float beta2, float eps, float max_grad_norm,
                           float* grad_norm_out, rgp_stream_t stream) {
  RGP_REQUIRE(params && grads && m && v && partials && n > 0 && n_partials > 0 && step >= 0, "rgp_adam_clip_step_ext: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const double t = (double)step + 1.0;
  const float lr_t = (float)((double)lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t)));
  const int blocks = (int)std::min<long long>((n + 255) / 256, 4096);
  adam_clip_kernel<<<blocks, 256, 0, s>>>(params, grads, m, v, n, partials, n_partials, max_grad_norm, lr_t, beta1, beta2, eps,
                                          grad_norm_out, nullptr);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

int rgp_lr_schedule_step(int* step_dev, float lr0, float decay, int decay_steps, float beta1, float beta2, float* lr_t_dev,
                         rgp_stream_t stream) {
  RGP_REQUIRE(step_dev && lr_t_dev && decay_steps > 0, "rgp_lr_schedule_step: bad arguments");
  lr_schedule_kernel<<<1, 64, 0, (hipStream_t)stream>>>(step_dev, lr0, decay, decay_steps, beta1, beta2, lr_t_dev);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

int rgp_adam_clip_step_dev(float* params, const float* grads, float* m, float* v, long long n, const float* partials,
                           int n_partials, const float* lr_t_dev, float beta1, float beta2, float eps, float max_grad_norm,
                           float* grad_norm_out, rgp_stream_t stream) {
  RGP_REQUIRE(params && grads && m && v && partials && lr_t_dev && n > 0 && n_partials > 0, "rgp_adam_clip_step_dev: bad arguments");
  const int blocks = (int)std::min<long long>((n + 255) / 256, 4096);
  adam_clip_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(params, grads, m, v, n, partials, n_partials, max_grad_norm, 0.f, beta1,
                                                            beta2, eps, grad_norm_out, lr_t_dev);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

}  // extern "C"
