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

#include "bwd_kernels.hip.h"
#include "rgp_grcn_plan.h"

using namespace rgp;

struct GrcnBwd {
  ConvDesc b_d2, b_d1, b_c, b_zr, b_x;     // dgrad convolutions
  ConvDesc b_px;                           // projection input gradient: d rows = dE x W^T
  // gather tables [ntaps][Mw] (element offsets, -1 = zero) + offsets in the workspace
  std::vector<int> t_E9, t_h9, t_dd1, t_dd2, t_y, t_d1, t_pad3S, t_pad2S, t_zero1, koff_m, koff_m2;
  size_t o_E9 = 0, o_h9 = 0, o_dd1 = 0, o_dd2 = 0, o_y = 0, o_d1 = 0, o_pad3S = 0, o_pad2S = 0,
         o_zero1 = 0, o_koff_m = 0, o_koff_m2 = 0;
  long long M = 0, M2 = 0, Mp = 0, M2p = 0;
  Buf dz, frame_sum, dgp, gp, dd2, dd1, dy, dh_head, dh_carry, drh, dcp_pad, dzr_pad, dxpre, dxpre_pad, dE, rh_all;
  Buf xT, dET, EcolT, dXpreT, HcolT, RHcolT, dd1colT, yT, dd2colT, d1T, sq_partial;
  rgp_grcn_weights w;   // forward weights (device fp32) as last set
};

namespace {

constexpr int SQ_BLOCKS = 256;

size_t put(Arena& a, const std::vector<int>& t) { return a.take(t.size() * 4); }

template <typename TS, typename TD>
int gather_T(const TS* src, TD* dst, const int* tab, int ntaps, int Mw, long long M, int C, long long ld, int row0,
             int inner, long long s_in, long long s_out, hipStream_t s) {
  dim3 grid((unsigned)((M + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)ntaps);
  gather_transpose_kernel<TS, TD><<<grid, 256, 0, s>>>(src, dst, tab, Mw, M, C, ld, row0, inner, s_in, s_out);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

// dW[rows x N] (+)= AT[rows x Kp] * BT[N x Kp]^T, K-contiguous operands, split-K atomics.
template <typename T>
int wgrad_gemm(const rgp_grcn* g, const T* AT, int rows, const T* BT, int N, long long Kp, size_t koff_off, float* out,
               hipStream_t s) {
  const GrcnBwd* b = g->bwd;
  IgemmParams p;
  p.A = AT;
  p.W = BT;
  p.in_tab = (const int*)(g->ws + b->o_zero1);
  p.koff = (const int*)(g->ws + koff_off);
  p.in_img_stride = Kp;
  p.Mw = 1;
  p.M = rows;
  p.N = N;
  p.K = (int)Kp;
  p.nk = (int)(Kp / Elem<T>::BKE);
  EpiParams e;
  memset(&e, 0, sizeof(e));
  e.out = out;
  e.out_tab = (const int*)(g->ws + b->o_zero1);
  e.out_img_stride = N;
  const int tiles = ((rows + 127) / 128) * ((N + 127) / 128);
  // split-K: every split adds rows x N fp32 atomics (the 1024-block rule used before spent 40 % of config 4's
  // training step in these GEMMs).  Measured optimum on MI355X: ~128 blocks in total for short reductions
  // (B*T*49 = 13.7 k rows: 7.3 -> 5.1 ms per step), ~256 for long ones (50 k rows: 9.9 -> 7.7 ms); RGP_WG_BLOCKS overrides.
  static const int target_env = getenv("RGP_WG_BLOCKS") ? atoi(getenv("RGP_WG_BLOCKS")) : 0;
  const int target = target_env > 0 ? target_env : (p.nk > 400 ? 256 : 128);
  int ksplit = std::max(1, std::min(target / std::max(tiles, 1), p.nk / 8));
  ksplit = std::max(1, std::min(ksplit, 64));
  return launch_igemm<T, 1, 1, EpiAtomicAddF32>(p, e, s, ksplit);
}

template <typename T>
int backward_impl(rgp_grcn* g, const float* probs, const float* logits, const float* labels,
                  const rgp_grcn_weights* gr, int loss_l2, hipStream_t s, const float* ext_dy = nullptr) {
  GrcnBwd* b = g->bwd;
  char* ws = g->ws;
  const int B = g->B, T_ = g->T, S = g->S, P = g->P, F = g->F;
  const long long M = b->M, Mp = b->Mp, M2 = b->M2, M2p = b->M2p;
  const size_t st = (size_t)B * 49 * S;
  auto I = [&](size_t off) { return (const int*)(ws + off); };
  auto Fp = [&](const Buf& x) { return (float*)(ws + x.off); };
  auto Tp = [&](const Buf& x) { return (T*)(ws + x.off); };

  // zero the gradients that are accumulated with atomics
  RGP_HIP(hipMemsetAsync((void*)gr->proj_c3d_W, 0, (size_t)1024 * P * 4, s));
  for (const float* q : {gr->gru_Wz, gr->gru_Wr, gr->gru_W}) RGP_HIP(hipMemsetAsync((void*)q, 0, (size_t)9 * P * S * 4, s));
  for (const float* q : {gr->gru_Uz, gr->gru_Ur, gr->gru_U}) RGP_HIP(hipMemsetAsync((void*)q, 0, (size_t)9 * S * S * 4, s));
  RGP_HIP(hipMemsetAsync((void*)gr->up_weight1, 0, (size_t)25 * 64 * S * 4, s));
  RGP_HIP(hipMemsetAsync((void*)gr->up_weight2, 0, (size_t)25 * 32 * 64 * 4, s));
  RGP_HIP(hipMemsetAsync(ws + b->dgp.off, 0, b->dgp.bytes, s));

  if (ext_dy) {
    // the gradient w.r.t. the (batch-normalised) states comes from outside (cascade: the stride-7
    // transposed conv above the bottom cell); the head of this plan is unused, its gradients are zero
    RGP_HIP(hipMemcpyAsync(Fp(b->dy), ext_dy, (size_t)M * S * 4, hipMemcpyDeviceToDevice, s));
    RGP_HIP(hipMemsetAsync((void*)gr->up_weight3, 0, (size_t)49 * 12 * 32 * 4, s));
    RGP_HIP(hipMemsetAsync((void*)gr->out_W, 0, 12 * 4, s));
    RGP_HIP(hipMemsetAsync((void*)gr->out_b, 0, 4, s));
  } else {
  // 1. d loss / d logits, d out_b
  dlogits_kernel<<<F, 256, 0, s>>>(loss_l2 ? logits : probs, labels, Fp(b->dz), Fp(b->frame_sum), 2401, 1.0f / (float)F, loss_l2);
  sum_kernel<<<1, 256, 0, s>>>(Fp(b->frame_sum), (float*)gr->out_b, F, 1.0f);
  // 2. folded 7x7 filter: wgrad -> dF3, d out_W ; dgrad -> dd2
  head_fold_wgrad_kernel<T><<<F, 256, 0, s>>>(Fp(b->dz), Tp(g->D2), Fp(b->dgp));
  head_unfold_grads_kernel<<<1, 256, 0, s>>>(Fp(b->dgp), b->w.up_weight3, b->w.out_W, (float*)gr->up_weight3, (float*)gr->out_W);
  head_fold_dgrad_kernel<T><<<dim3(49, F), 256, 0, s>>>(Fp(b->dz), Fp(b->gp), Tp(b->dd2));
  RGP_HIP(hipGetLastError());
  // 3. deconv2: wgrad (dF2[a,b,o,c] = sum dd2[2i+a,2j+b,o] d1[i,j,c]) and dgrad
  RGP_TRY((gather_T<T, T>(Tp(b->dd2), Tp(b->dd2colT), I(b->o_dd2), 25, 529, M2, 32, M2p, 0, 1, 0, 2401LL * 32, s)));
  RGP_TRY((gather_T<T, T>(Tp(g->D1), Tp(b->d1T), I(b->o_d1), 1, 529, M2, 64, M2p, 0, 1, 0, 27LL * 27 * 64, s)));
  RGP_TRY(wgrad_gemm<T>(g, Tp(b->dd2colT), 25 * 32, Tp(b->d1T), 64, M2p, b->o_koff_m2, (float*)gr->up_weight2, s));
  {
    IgemmParams p = make_params(b->b_d2, Tp(b->dd2), ws, F);
    EpiParams e = make_epi(b->b_d2, Tp(b->dd1), ws);
    if (sizeof(T) == 2) RGP_TRY((launch_igemm<T, 2, 1, EpiStore<T, false, false>>(p, e, s)));
    else RGP_TRY((launch_igemm<T, 1, 1, EpiStore<T, false, false>>(p, e, s)));
  }
  // 4. deconv1: wgrad (dF1[a,b,o,c] = sum dd1[3i+a,3j+b,o] y[i,j,c]) and dgrad -> dy (fp32)
  RGP_TRY((gather_T<T, T>(Tp(b->dd1), Tp(b->dd1colT), I(b->o_dd1), 25, 49, M, 64, Mp, 0, 1, 0, 529LL * 64, s)));
  RGP_TRY((gather_T<T, T>(Tp(g->hbn), Tp(b->yT), I(b->o_y), 1, 49, M, S, Mp, 0, 1, 0, 81LL * S, s)));
  RGP_TRY(wgrad_gemm<T>(g, Tp(b->dd1colT), 25 * 64, Tp(b->yT), S, Mp, b->o_koff_m, (float*)gr->up_weight1, s));
  {
    IgemmParams p = make_params(b->b_d1, Tp(b->dd1), ws, F);
    EpiParams e = make_epi(b->b_d1, Fp(b->dy), ws);
    RGP_TRY((launch_igemm<T, 1, 1, EpiStore<float, false, false>>(p, e, s)));
  }
  }  // !ext_dy
  // 5. per-timestep batch-norm
  const float inv = 1.0f / sqrtf(1.0f + 1e-3f);
  bn_bwd_kernel<<<dim3(S / 8, T_), 256, 0, s>>>(Fp(b->dy), Fp(g->hall), b->w.bn_gamma, (float*)gr->bn_gamma,
                                                  (float*)gr->bn_beta, Fp(b->dh_head), B, T_, S, inv);
  // 6. BPTT: t = T-1 .. 0
  const int ew_blocks = (int)std::min<size_t>((st + 255) / 256, 4096);
  for (int t = T_ - 1; t >= 0; --t) {
    const float* h_prev = Fp(g->hall) + (size_t)t * st;
    gru_bwd1_kernel<T><<<ew_blocks, 256, 0, s>>>(Fp(b->dh_head) + (size_t)t * st, Fp(b->dh_carry), h_prev,
                                                  Fp(g->uall) + (size_t)t * st, Fp(g->call) + (size_t)t * st, Fp(b->dxpre),
                                                  Tp(b->dcp_pad), I(g->o_pad9_S), B, T_, t, S, t == T_ - 1);
    RGP_HIP(hipGetLastError());
    {
      IgemmParams p = make_params(b->b_c, Tp(b->dcp_pad), ws, B);
      EpiParams e = make_epi(b->b_c, Fp(b->drh), ws);
      RGP_TRY((launch_igemm<T, 1, 1, EpiStore<float, false, false>>(p, e, s)));
    }
    gru_bwd2_kernel<T><<<ew_blocks, 256, 0, s>>>(Fp(b->drh), Fp(b->dh_carry), h_prev, Fp(g->rall) + (size_t)t * st,
                                                  Fp(b->dxpre), Tp(b->dzr_pad), I(b->o_pad2S), B, T_, t, S);
    RGP_HIP(hipGetLastError());
    {
      IgemmParams p = make_params(b->b_zr, Tp(b->dzr_pad), ws, B);
      EpiParams e = make_epi(b->b_zr, Fp(b->dh_carry), ws);
      RGP_TRY((launch_igemm<T, 1, 1, EpiAccumF32>(p, e, s)));
    }
  }
  // 7. hoisted input convs: dgrad -> dE, then the projection's gradients
  {
    const long long tot = (long long)F * 49 * 3 * S;
    pad_rows_kernel<T><<<(int)std::min<long long>((tot + 255) / 256, 8192), 256, 0, s>>>(Fp(b->dxpre), Tp(b->dxpre_pad),
                                                                                      I(b->o_pad3S), tot, 3 * S);
    RGP_HIP(hipGetLastError());
    IgemmParams p = make_params(b->b_x, Tp(b->dxpre_pad), ws, F);
    EpiParams e = make_epi(b->b_x, Tp(b->dE), ws);
    RGP_TRY((launch_igemm<T, 1, 1, EpiStore<T, false, false>>(p, e, s)));
  }
  // 8. weight gradients of the recurrence and the projection, hoisted over all T steps
  RGP_TRY((gather_T<float, T>(Fp(b->dxpre), Tp(b->dXpreT), I(b->o_zero1), 1, 1, M, 3 * S, Mp, 0, 1, 0, 3LL * S, s)));
  RGP_TRY((gather_T<T, T>(Tp(g->E), Tp(b->EcolT), I(b->o_E9), 9, 49, M, P, Mp, 0, 1, 0, 81LL * P, s)));
  // h_{t-1} of frame (b,t) is hall[t][b]; frames are b-major: img = b*T+t -> (img % T)*B*49*S + (img / T)*49*S
  RGP_TRY((gather_T<float, T>(Fp(g->hall), Tp(b->HcolT), I(b->o_h9), 9, 49, M, S, Mp, 0, T_, (long long)st, 49LL * S, s)));
  mul_kernel<<<(int)std::min<size_t>((st * T_ + 255) / 256, 8192), 256, 0, s>>>(Fp(g->rall), Fp(g->hall), Fp(b->rh_all), (long long)st * T_);
  RGP_HIP(hipGetLastError());
  RGP_TRY((gather_T<float, T>(Fp(b->rh_all), Tp(b->RHcolT), I(b->o_h9), 9, 49, M, S, Mp, 0, T_, (long long)st, 49LL * S, s)));
  RGP_TRY((gather_T<T, T>(Tp(b->dE), Tp(b->dET), I(b->o_zero1), 1, 1, M, P, Mp, 0, 1, 0, (long long)P, s)));
  RGP_TRY((gather_T<T, T>(Tp(g->xt), Tp(b->xT), I(b->o_zero1), 1, 1, M, 1024, Mp, 0, 1, 0, 1024LL, s)));
  const T* dXT = Tp(b->dXpreT);
  RGP_TRY(wgrad_gemm<T>(g, Tp(b->EcolT), 9 * P, dXT, S, Mp, b->o_koff_m, (float*)gr->gru_Wz, s));
  RGP_TRY(wgrad_gemm<T>(g, Tp(b->EcolT), 9 * P, dXT + (size_t)S * Mp, S, Mp, b->o_koff_m, (float*)gr->gru_Wr, s));
  RGP_TRY(wgrad_gemm<T>(g, Tp(b->EcolT), 9 * P, dXT + (size_t)2 * S * Mp, S, Mp, b->o_koff_m, (float*)gr->gru_W, s));
  RGP_TRY(wgrad_gemm<T>(g, Tp(b->HcolT), 9 * S, dXT, S, Mp, b->o_koff_m, (float*)gr->gru_Uz, s));
  RGP_TRY(wgrad_gemm<T>(g, Tp(b->HcolT), 9 * S, dXT + (size_t)S * Mp, S, Mp, b->o_koff_m, (float*)gr->gru_Ur, s));
  RGP_TRY(wgrad_gemm<T>(g, Tp(b->RHcolT), 9 * S, dXT + (size_t)2 * S * Mp, S, Mp, b->o_koff_m, (float*)gr->gru_U, s));
  RGP_TRY(wgrad_gemm<T>(g, Tp(b->xT), 1024, Tp(b->dET), P, Mp, b->o_koff_m, (float*)gr->proj_c3d_W, s));
  rowsum_kernel<T><<<P, 256, 0, s>>>(Tp(b->dET), (float*)gr->proj_c3d_b, Mp, M);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

template <typename T>
int pack_impl(rgp_grcn* g, const rgp_grcn_weights* w, hipStream_t s) {
  GrcnBwd* b = g->bwd;
  char* ws = g->ws;
  const int S = g->S, P = g->P;
  for (ConvDesc* d : {&b->b_d2, &b->b_d1, &b->b_c, &b->b_zr, &b->b_x, &b->b_px}) RGP_HIP(hipMemsetAsync(ws + d->w_off, 0, d->w_bytes(g->dtype), s));
  RGP_TRY(pack_filter<T>(b->b_px, w->proj_c3d_W, ws, 512, 0, s));            // d = 0: feature channels 0, 2, 4, ...
  RGP_TRY(pack_filter<T>(b->b_px, w->proj_c3d_W + P, ws, 512, 512, s));      // d = 1: feature channels 1, 3, 5, ...
  RGP_TRY(pack_filter<T>(b->b_d2, w->up_weight2, ws, 64, 0, s));
  RGP_TRY(pack_filter<T>(b->b_d1, w->up_weight1, ws, S, 0, s));
  RGP_TRY(pack_filter<T>(b->b_c, w->gru_U, ws, S, 0, s));
  RGP_TRY(pack_filter<T>(b->b_zr, w->gru_Uz, ws, S, 0, s, 0, 1));
  RGP_TRY(pack_filter<T>(b->b_zr, w->gru_Ur, ws, S, 0, s, S, 1));
  RGP_TRY(pack_filter<T>(b->b_x, w->gru_Wz, ws, P, 0, s, 0, 1));
  RGP_TRY(pack_filter<T>(b->b_x, w->gru_Wr, ws, P, 0, s, S, 1));
  RGP_TRY(pack_filter<T>(b->b_x, w->gru_W, ws, P, 0, s, 2 * S, 1));
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
  if (!ok) return set_err(RGP_EINVAL, "rgp_grcn_create: backward K schedule failed");
  for (ConvDesc* d : {&b->b_d2, &b->b_d1, &b->b_c, &b->b_zr, &b->b_x, &b->b_px}) d->reserve(a, dtype);

  // ---- gather tables
  for (int t = 0; t < 9; ++t) for (int y = 0; y < 7; ++y) for (int x = 0; x < 7; ++x) {
    const int ky = t / 3, kx = t % 3;
    b->t_E9.push_back(((y + ky) * 9 + x + kx) * P);
    const int yy = y + ky - 1, xx = x + kx - 1;
    b->t_h9.push_back((yy >= 0 && yy < 7 && xx >= 0 && xx < 7) ? (yy * 7 + xx) * S : -1);
  }
  for (int aa = 0; aa < 5; ++aa) for (int bb = 0; bb < 5; ++bb) {
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 7; ++j) b->t_dd1.push_back(((3 * i + aa) * 23 + 3 * j + bb) * 64);
  }
  for (int aa = 0; aa < 5; ++aa) for (int bb = 0; bb < 5; ++bb) {
    for (int i = 0; i < 23; ++i) for (int j = 0; j < 23; ++j) b->t_dd2.push_back(((2 * i + aa) * 49 + 2 * j + bb) * 32);
  }
  for (int y = 0; y < 7; ++y) for (int x = 0; x < 7; ++x) {
    b->t_y.push_back(((y + 1) * 9 + x + 1) * S);            // interior of a 9x9xS image
    b->t_pad3S.push_back(((y + 1) * 9 + x + 1) * 3 * S);    // interior of a 9x9x3S image
    b->t_pad2S.push_back(((y + 1) * 9 + x + 1) * 2 * S);    // interior of a 9x9x2S image
  }
  for (int i = 0; i < 23; ++i) for (int j = 0; j < 23; ++j) b->t_d1.push_back(((i + 2) * 27 + j + 2) * 64);
  b->t_zero1.push_back(0);
  for (long long k = 0; k < b->Mp / 32; ++k) b->koff_m.push_back((int)(k * bke(dtype)));
  for (long long k = 0; k < b->M2p / 32; ++k) b->koff_m2.push_back((int)(k * bke(dtype)));
  b->o_E9 = put(a, b->t_E9); b->o_h9 = put(a, b->t_h9); b->o_dd1 = put(a, b->t_dd1); b->o_dd2 = put(a, b->t_dd2);
  b->o_y = put(a, b->t_y); b->o_d1 = put(a, b->t_d1); b->o_pad3S = put(a, b->t_pad3S); b->o_pad2S = put(a, b->t_pad2S);
  b->o_zero1 = put(a, b->t_zero1); b->o_koff_m = put(a, b->koff_m); b->o_koff_m2 = put(a, b->koff_m2);

  // ---- buffers
  const size_t st = (size_t)B * 49 * S * 4;
  b->dz = take(a, (size_t)F * 2401 * 4);
  b->frame_sum = take(a, (size_t)F * 4);
  b->dgp = take(a, 50 * 32 * 4);
  b->gp = take(a, 50 * 32 * 4);
  b->dd2 = take(a, (size_t)F * 2401 * 32 * es + 4096);
  b->dd1 = take(a, (size_t)F * 529 * 64 * es + 4096);
  b->dy = take(a, (size_t)F * 49 * S * 4);
  b->dh_head = take(a, st * T_);
  b->dh_carry = take(a, st);
  b->drh = take(a, st);
  b->dcp_pad = take(a, (size_t)B * 81 * S * es);
  b->dzr_pad = take(a, (size_t)B * 81 * 2 * S * es);
  b->dxpre = take(a, (size_t)F * 49 * 3 * S * 4);
  b->dxpre_pad = take(a, (size_t)F * 81 * 3 * S * es);
  b->dE = take(a, (size_t)b->M * P * es);
  b->rh_all = take(a, st * T_);
  auto rows128 = [](int n) { return (size_t)((n + 127) / 128 * 128 + 128); };
  b->xT = take(a, rows128(1024) * b->Mp * es);
  b->dET = take(a, rows128(P) * b->Mp * es);
  b->EcolT = take(a, rows128(9 * P) * b->Mp * es);
  b->dXpreT = take(a, rows128(3 * S) * b->Mp * es);
  b->HcolT = take(a, rows128(9 * S) * b->Mp * es);
  b->RHcolT = take(a, rows128(9 * S) * b->Mp * es);
  b->dd1colT = take(a, rows128(25 * 64) * b->Mp * es);
  b->yT = take(a, rows128(S) * b->Mp * es);
  b->dd2colT = take(a, rows128(25 * 32) * b->M2p * es);
  b->d1T = take(a, rows128(64) * b->M2p * es);
  b->sq_partial = take(a, SQ_BLOCKS * 4);
  return RGP_OK;
}

int grcn_bwd_upload(rgp_grcn* g, hipStream_t s) {
  GrcnBwd* b = g->bwd;
  for (ConvDesc* d : {&b->b_d2, &b->b_d1, &b->b_c, &b->b_zr, &b->b_x, &b->b_px}) RGP_TRY(upload_desc(*d, g->ws, s));
  auto up = [&](const std::vector<int>& t, size_t off) -> int {
    RGP_HIP(hipMemcpyAsync(g->ws + off, t.data(), t.size() * 4, hipMemcpyHostToDevice, s));
    return RGP_OK;
  };
  RGP_TRY(up(b->t_E9, b->o_E9)); RGP_TRY(up(b->t_h9, b->o_h9)); RGP_TRY(up(b->t_dd1, b->o_dd1)); RGP_TRY(up(b->t_dd2, b->o_dd2));
  RGP_TRY(up(b->t_y, b->o_y)); RGP_TRY(up(b->t_d1, b->o_d1)); RGP_TRY(up(b->t_pad3S, b->o_pad3S)); RGP_TRY(up(b->t_pad2S, b->o_pad2S));
  RGP_TRY(up(b->t_zero1, b->o_zero1)); RGP_TRY(up(b->koff_m, b->o_koff_m)); RGP_TRY(up(b->koff_m2, b->o_koff_m2));
  return RGP_OK;
}

int grcn_bwd_pack(rgp_grcn* g, const rgp_grcn_weights* w, hipStream_t s) {
  g->bwd->w = *w;
  RGP_TRY(g->dtype == RGP_BF16 ? pack_impl<bf16_t>(g, w, s) : pack_impl<float>(g, w, s));
  flip_fold_kernel<<<(49 * 32 + 255) / 256, 256, 0, s>>>((const float*)(g->ws + g->gfold.off), (float*)(g->ws + g->bwd->gp.off));
  RGP_HIP(hipGetLastError());
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

int rgp_grcn_backward_input(rgp_grcn_t* g, float* d_rows, rgp_stream_t stream) {
  RGP_REQUIRE(g && d_rows, "rgp_grcn_backward_input: null argument");
  if (!g->ws || !g->save || !g->bwd || !g->weights_set) return set_err(RGP_ESTATE, "rgp_grcn_backward_input: call after rgp_grcn_backward");
  hipStream_t s = (hipStream_t)stream;
  GrcnBwd* b = g->bwd;
  IgemmParams p = make_params(b->b_px, g->ws + b->dE.off, g->ws, (int)b->M);
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
