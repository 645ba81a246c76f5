// librgp_hip.so: the gaze_grcn head (projection, ConvGRU, transposed-conv saliency
// head) -- plan object, workspace layout and stage launches.
// Reference graph: /root/reference/models/gaze_grcn.py:173-376.
#include <algorithm>
#include <map>

#include "rgp_grcn_plan.h"
#include "convgru_seq.hip.h"
#include "head_logits.hip.h"
#include "head_fold.hip.h"

using namespace rgp;

namespace {

// Gather-form sub-pixel phases of tf.nn.conv2d_transpose(VALID, stride s, k x k,
// filter [k,k,Cout,Cin]) (gaze_grcn.py:326-343):
//   out[s*i+py, s*j+px, o] = sum_{a',b',c} in[i-a', j-b', c] * F[py+s*a', px+s*b', o, c]
// Input image is halo-padded by hin (>= taps-1), output image by hout.
// One GEMM per ROW phase py: its s column phases px share their input rows, and the s x Cout values of output pixels
// s*j .. s*j+s-1 are contiguous in memory, so they are the N = s*Cout columns (px, o) of one problem (round 3; before:
// s*s launches of N = Cout, each re-gathering the input).  The tap window is the widest phase's (px = 0: tbm taps in
// x); a phase that lacks a tap gets zero weights for it (pack[]: one tap table per px, -1 = zero).  Rows j beyond a
// phase's own extent compute positions x >= OH: every input they touch is halo, so exact zeros land in the output halo.
bool build_deconv_phases(std::vector<ConvDesc>& out, std::vector<ConvDesc>& pack, int k, int s, int H, int hin, int Cin, int OH,
                         int hout, int Cout, int dtype) {
  const int Wp = H + 2 * hin, OWp = OH + 2 * hout;
  const int tbm = (k + s - 1) / s, Wph = (OH + s - 1) / s;
  if (hin < tbm - 1 || Wph > H + hin || s * (Wph - 1) + s - 1 + hout >= OWp) return false;
  for (int py = 0; py < s; ++py) {
    const int ta = (k - py + s - 1) / s;
    const int Hph = (OH - py + s - 1) / s;
    if (hin < ta - 1 || Hph > H + hin) return false;
    ConvDesc d;
    d.Mw = Hph * Wph;
    d.N = s * Cout;
    d.in_img_stride = (long long)Wp * Wp * Cin;
    d.out_img_stride = (long long)OWp * OWp * Cout;
    for (int i = 0; i < Hph; ++i)
      for (int j = 0; j < Wph; ++j) {
        d.in_tab.push_back(((i - (ta - 1) + hin) * Wp + (j - (tbm - 1) + hin)) * Cin);
        d.out_tab.push_back(((s * i + py + hout) * OWp + (s * j + hout)) * Cout);
      }
    std::vector<int> tapoff, fidx;
    for (int u = 0; u < ta; ++u)
      for (int v = 0; v < tbm; ++v) {
        tapoff.push_back((u * Wp + v) * Cin);
        fidx.push_back((py + s * (ta - 1 - u)) * k + s * (tbm - 1 - v));
      }
    if (!build_k_schedule(d, tapoff, fidx, Cin, dtype)) return false;
    d.s_tap = (long long)Cout * Cin;
    d.s_n = Cin;
    d.s_c = 1;
    // packing aliases: rows px*Cout .. of the same packed filter, taps shifted by px (beyond the filter: zero)
    for (int px = 0; px < s; ++px) {
      ConvDesc a = d;
      a.in_tab.clear(); a.out_tab.clear(); a.koff.clear(); a.koff_tm.clear();
      for (size_t t = 0; t < a.tap_src.size(); ++t) {
        if (t >= fidx.size() || a.tap_src[t] < 0) { a.tap_src[t] = -1; continue; }
        const int kx = fidx[t] % k + px;
        a.tap_src[t] = kx < k ? fidx[t] + px : -1;
      }
      pack.push_back(a);
    }
    out.push_back(d);
  }
  return true;
}

std::vector<int> pad_tab(int H, int halo, int C) {
  std::vector<int> t;
  const int Wp = H + 2 * halo;
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < H; ++x) t.push_back(((y + halo) * Wp + x + halo) * C);
  return t;
}

size_t put_tab(Arena& a, const std::vector<int>& t) { return a.take(t.size() * 4); }

// logit[y,x] = sum_{u,v,c} D2p[y+u, x+v, c] G[6-u, 6-v, c] + out_b has ONE output channel: as an implicit GEMM over
// (tap, channel) it fills 1 of the 32 columns of the narrowest tile.  Instead a GEMM row is (y, block of 16 x): for
// tap row u its operand is the contiguous run D2p[y+u, x0 .. x0+21, 0..31] (704 elements) and the filter is the
// Toeplitz matrix Gt[u][(x', c)][n] = G[6-u, 6-(x'-n), c] (0 outside the 7 taps) -- 3.1x the MACs, all 16 columns
// useful, 5x less MFMA work than before.  49 = 3 x 16 + 1: the last pixel column goes through the 1-column path.
template <typename T, int G>
int run_d3(rgp_grcn* g, float* logits, hipStream_t s) {
  if constexpr (sizeof(T) == 2)                                // bf16: the band kernel (head_logits.hip.h), all 49 columns
    return run_head_logits((const bf16_t*)(g->ws + g->D2.off), (const bf16_t*)(g->ws + g->d3t.w_off),
                           (const float*)(g->ws + g->bias16.off), logits, g->F, s);
  {
    IgemmParams p = make_params(g->d3t, g->ws + g->D2.off, g->ws, g->F);
    EpiParams e = make_epi(g->d3t, logits, g->ws);
    e.bias = (const float*)(g->ws + g->bias16.off);
    RGP_TRY((launch_igemm<T, 1, 1, EpiStore<float, true, false>>(p, e, s)));
  }
  IgemmParams p = make_params(g->d3, g->ws + g->D2.off, g->ws, g->F);
  EpiParams e = make_epi(g->d3, logits, g->ws);
  e.bias = g->out_b;
  return launch_igemm<T, G, 1, EpiStore<float, true, false>>(p, e, s);
}

template <typename T>
int proj_impl(rgp_grcn* g, const float* c3d_input, const void* rows, hipStream_t s) {
  const ConvDesc& d = rows ? g->proj_rows : g->proj;
  const void* A = rows;
  if (!rows) {
    nchw_to_rows_kernel<T><<<dim3(1024 / 64, g->F), 256, 0, s>>>(c3d_input, (T*)(g->ws + g->xt.off), 1024);
    RGP_HIP(hipGetLastError());
    A = g->ws + g->xt.off;
  } else if (g->save) {
    // the backward's projection wgrad reads xt in the reference's channel order c*2+d (gaze_rnn.py:494-497)
    const long long total = (long long)g->F * 49 * 1024;
    rows_to_xt_kernel<T><<<(int)std::min<long long>((total + 255) / 256, 8192), 256, 0, s>>>((const T*)rows, (T*)(g->ws + g->xt.off), total);
    RGP_HIP(hipGetLastError());
  }
  IgemmParams p = make_params(d, A, g->ws, g->F);
  EpiParams e = make_epi(d, g->ws + g->E.off, g->ws);
  e.bias = g->proj_b;
  return launch_igemm<T, 1, 1, EpiStore<T, true, false>>(p, e, s);
}

template <typename T>
int xconv_impl(rgp_grcn* g, hipStream_t s) {
  IgemmParams p = make_params(g->xconv, g->ws + g->E.off, g->ws, g->F);
  EpiParams e = make_epi(g->xconv, g->ws + g->xpre.off, g->ws);
  return launch_igemm<T, 1, 1, EpiStore<float, false, false>>(p, e, s);
}

}  // namespace

// The persistent ConvGRU kernels need all their workgroups (8 per group, one per CU) resident together.
bool seq_persistent_ok(const rgp_grcn* g) {
  int n_cu = 0;
  return g->seq_groups > 0 && device_cu_count(&n_cu) == RGP_OK && g->seq_groups * 8 <= n_cu;
}

// Whether the TOP gradient group may be released (its all-reduce started on another stream) BEFORE the persistent BPTT
// launch: only when that launch leaves at least RGP_RCCL_CU_RESERVE CUs to the collective's workgroups.
bool grads_top_early(const rgp_grcn* g) {
  int n_cu = 0;
  if (g->seq_groups <= 0 || device_cu_count(&n_cu) != RGP_OK) return true;      // per-step plans: nothing needs co-residency
  return g->seq_groups * 8 <= n_cu - RGP_RCCL_CU_RESERVE;
}

namespace {

// All T steps in one persistent launch (convgru_seq.hip.h): recurrent filters resident in registers, state on chip.
int seq_persistent(rgp_grcn* g, hipStream_t s) {
  const int B = g->B, T_ = g->T, S = g->S;
  const size_t st = (size_t)B * 49 * S;
  {
    ZeroBatch z(s);
    RGP_TRY(z.add(g->ws + g->hall.off, st * 4));                             // h_0 = 0 (gaze_grcn.py:262)
    RGP_TRY(z.add(g->ws + g->seq_cnt.off, g->seq_cnt.bytes));                // phase counters: zeroed EVERY call
    RGP_TRY(z.flush());
  }
  SeqParams p;
  p.w_zr = (const bf16_t*)(g->ws + g->gzr.w_off);
  p.w_c = (const bf16_t*)(g->ws + g->gc.w_off);
  p.xpre = (const float*)(g->ws + g->xpre.off);
  p.hall = (float*)(g->ws + g->hall.off);
  p.uall = (float*)(g->ws + g->uall.off);
  p.rall = g->save ? (float*)(g->ws + g->rall.off) : nullptr;
  p.call = g->save ? (float*)(g->ws + g->call.off) : nullptr;
  p.hbn = (bf16_t*)(g->ws + g->hbn.off);
  p.bn_gamma = g->bn_gamma;
  p.bn_beta = g->bn_beta;
  p.bn_inv_std = 1.0f / sqrtf(1.0f + 1e-3f);   // moving mean 0 / var 1, eps 1e-3
  p.xch_h = (bf16_t*)(g->ws + g->xch_h.off);
  p.xch_rh = (bf16_t*)(g->ws + g->xch_rh.off);
  p.cnt = (unsigned*)(g->ws + g->seq_cnt.off);
  p.B = B; p.T = T_; p.NC = g->seq_nc; p.ngroups = g->seq_groups; p.K = g->gzr.K;
  p.err = g->err_host;
  p.skip_member = (g->fault & 1) ? 7 : -1;
  g->fault &= ~1;
  RGP_REQUIRE(g->gzr.K == 9 * S && g->gc.K == 9 * S && g->gzr.chunk_major == 0, "convgru_seq: unexpected filter packing");
  PersistentLaunch guard(s);
  RGP_TRY(guard.status());
  if (g->seq_nc == 1) {
    RGP_TRY(ensure_dyn_smem((const void*)convgru_seq_kernel<4>, SEQ_SMEM));
    convgru_seq_kernel<4><<<g->seq_groups * 8, SEQ_NT, SEQ_SMEM, s>>>(p);
  } else {
    RGP_TRY(ensure_dyn_smem((const void*)convgru_seq_kernel<7>, SEQ_SMEM));
    convgru_seq_kernel<7><<<g->seq_groups * 8, SEQ_NT, SEQ_SMEM, s>>>(p);
  }
  RGP_HIP(hipGetLastError());
  return guard.commit();
}

template <typename T>
int seq_impl(rgp_grcn* g, hipStream_t s) {
  const int B = g->B, T_ = g->T, S = g->S;
  if (sizeof(T) == 2 && seq_persistent_ok(g) && dev_knob("RGP_SEQ", 1)) return seq_persistent(g, s);
  const size_t st = (size_t)B * 49 * S;  // fp32 elements per state snapshot
  RGP_HIP(hipMemsetAsync(g->ws + g->hp.off, 0, g->hp.bytes, s));      // h_0 = 0 (gaze_grcn.py:262)
  RGP_HIP(hipMemsetAsync(g->ws + g->hall.off, 0, st * 4, s));
  float* hall = (float*)(g->ws + g->hall.off);
  float* uall = (float*)(g->ws + g->uall.off);
  float* rall = g->save ? (float*)(g->ws + g->rall.off) : nullptr;
  float* call = g->save ? (float*)(g->ws + g->call.off) : nullptr;
  for (int t = 0; t < T_; ++t) {
    EpiParams e = make_epi(g->gzr, g->ws + g->rhp.off, g->ws);
    e.xpre = (const float*)(g->ws + g->xpre.off) + (size_t)t * 49 * 3 * S;
    e.xpre_img_stride = (long long)T_ * 49 * 3 * S;
    e.xpre_ld = 3 * S;
    e.xpre_col = 0;
    e.S = S;
    e.state_rows = 49;
    e.h_prev = hall + (size_t)t * st;
    e.h_next = hall + (size_t)(t + 1) * st;
    e.u_gate = uall + (size_t)t * st;
    e.r_save = rall ? rall + (size_t)t * st : nullptr;
    e.c_save = call ? call + (size_t)t * st : nullptr;
    IgemmParams p = make_params(g->gzr, g->ws + g->hp.off, g->ws, B);
    RGP_TRY((launch_igemm<T, 1, 1, EpiGruZR<T>>(p, e, s)));
    // candidate conv on r*h, blend, BN -> next operand + head image b*T+t
    e.out = g->ws + g->hp.off;
    e.out_tab = (const int*)(g->ws + g->gc.out_tab_off);
    e.out_img_stride = g->gc.out_img_stride;
    e.xpre_col = 2 * S;
    e.out2 = g->ws + g->hbn.off;
    e.out2_tab = (const int*)(g->ws + g->gc.out_tab_off);
    e.out2_img_stride = 81LL * S;
    e.out2_img_mul = T_;
    e.out2_img_add = t;
    e.bn_gamma = g->bn_gamma + (size_t)t * S;
    e.bn_beta = g->bn_beta + (size_t)t * S;
    e.bn_inv_std = 1.0f / sqrtf(1.0f + 1e-3f);  // moving mean 0 / var 1, eps 1e-3
    IgemmParams pc = make_params(g->gc, g->ws + g->rhp.off, g->ws, B);
    RGP_TRY((launch_igemm<T, 1, 1, EpiGruC<T>>(pc, e, s)));
    if (g->step_ev) RGP_HIP(hipEventRecord(g->step_ev[t], s));
  }
  return RGP_OK;
}

template <typename T>
int head_impl(rgp_grcn* g, float* logits, hipStream_t s) {
  if (g->fold_head) {                                          // head_fold.hip.h: Z = BN(h) x K^T, then col2im (+ out_b)
    IgemmParams p = make_params(g->hfold, g->ws + g->hbn.off, g->ws, g->F);
    EpiParams e = make_epi(g->hfold, g->ws + g->hf_z.off, g->ws);
    RGP_TRY((launch_igemm<T, 1, 1, EpiStore<float, false, false>>(p, e, s)));
    const long long total = (long long)g->F * 2401;
    head_col2im_kernel<<<(int)std::min<long long>((total + 255) / 256, 8192), 256, 0, s>>>((const float*)(g->ws + g->hf_z.off), g->out_b,
                                                                                        logits, total);
    RGP_HIP(hipGetLastError());
    return RGP_OK;
  }
  for (const ConvDesc& d : g->d1) {
    IgemmParams p = make_params(d, g->ws + g->hbn.off, g->ws, g->F);
    EpiParams e = make_epi(d, g->ws + g->D1.off, g->ws);
    RGP_TRY((launch_igemm<T, 1, 1, EpiStore<T, false, false>>(p, e, s)));
  }
  for (const ConvDesc& d : g->d2) {
    IgemmParams p = make_params(d, g->ws + g->D1.off, g->ws, g->F);
    EpiParams e = make_epi(d, g->ws + g->D2.off, g->ws);
    RGP_TRY((launch_igemm<T, 1, 1, EpiStore<T, false, false>>(p, e, s)));
  }
  if (sizeof(T) == 2) return run_d3<T, 2>(g, logits, s);
  return run_d3<T, 1>(g, logits, s);
}

template <typename T>
int set_weights_impl(rgp_grcn* g, const rgp_grcn_weights* w, hipStream_t s, hipStream_t sc) {
  char* ws = g->ws;
  const int S = g->S, P = g->P;
  PackBatch<T> pk(ws, s);       // every pack of this call in one launch, behind the fold / Toeplitz kernels it reads from
  // (the packed-filter areas were zeroed with the workspace at bind time and a pack writes the same positions every
  // time: their channel / row padding stays zero without a memset per call -- 17 launches per optimizer step)
  RGP_TRY(pk.add(g->proj, w->proj_c3d_W, P, 0));
  RGP_TRY(pk.add(g->proj_rows, w->proj_c3d_W, P, 0));
  RGP_TRY(pk.add(g->xconv, w->gru_Wz, S, 0));
  RGP_TRY(pk.add(g->xconv, w->gru_Wr, S, S));
  RGP_TRY(pk.add(g->xconv, w->gru_W, S, 2 * S));
  RGP_TRY(pk.add(g->gzr, w->gru_Uz, S, 0));
  RGP_TRY(pk.add(g->gzr, w->gru_Ur, S, S));
  RGP_TRY(pk.add(g->gc, w->gru_U, S, 0));
  // sc: the stream of the head's fold and the pack that reads it (= s, or a training plan's side stream: rgp_grcn_set_weights)
  float* gf = (float*)(ws + g->gfold.off);
  fold_head_filter_kernel<<<(49 * 32 + 255) / 256, 256, 0, g->fold_head ? sc : s>>>(w->up_weight3, w->out_W, gf, 49, 12, 32);
  RGP_HIP(hipGetLastError());
  if (!g->fold_head) {                                         // the three-stage head's operand filters
    for (size_t i = 0; i < g->d1_pack.size(); ++i) RGP_TRY(pk.add(g->d1_pack[i], w->up_weight1, 64, (int)(i % 3) * 64));   // column phase px = i % 3
    for (size_t i = 0; i < g->d2_pack.size(); ++i) RGP_TRY(pk.add(g->d2_pack[i], w->up_weight2, 32, (int)(i % 2) * 32));
    RGP_TRY(pk.add(g->d3, gf, 1, 0));
    toeplitz_head_filter_kernel<<<(7 * 16 * 704 + 255) / 256, 256, 0, s>>>(gf, w->out_b, (float*)(ws + g->gtoep.off),
                                                                            (float*)(ws + g->bias16.off));
    RGP_HIP(hipGetLastError());
    RGP_TRY(pk.add(g->d3t, (const float*)(ws + g->gtoep.off), 16, 0));
  }
  if (g->fold_head) {
    // the head as one 19x19 stride-6 transposed convolution (head_fold.hip.h): G (above) -> H = G o weight2 -> K = H o weight1
    float* hf = (float*)(ws + g->hf_h.off);
    float* kf = (float*)(ws + g->hf_k.off);
    head_fold_h_kernel<<<(HF_HP * HF_HP * 64 + 255) / 256, 256, 0, sc>>>(gf, w->up_weight2, hf);
    float* part = (float*)(ws + g->hf_part.off);
    head_fold_k_kernel<<<dim3(HF_KP * HF_KP, 5), 128, 0, sc>>>(hf, w->up_weight1, part, S);
    head_fold_sum_kernel<<<(HF_KP * HF_KP * S + 255) / 256, 256, 0, sc>>>(part, kf, HF_KP * HF_KP * S, 5);
    RGP_HIP(hipGetLastError());
    PackBatch<T> pk2(ws, sc);
    RGP_TRY(pk2.add(g->hfold, kf, HF_KP * HF_KP, 0));         // GEMM filter [(r,t)][s]; rows 361 .. 383 stay zero
    RGP_TRY(pk2.flush());
  }
  RGP_TRY(pk.flush());
  g->bn_gamma = w->bn_gamma;
  g->bn_beta = w->bn_beta;
  g->proj_b = w->proj_c3d_b;
  g->out_b = w->out_b;
  g->weights_set = true;
  return RGP_OK;
}

int check_ready(rgp_grcn* g) {
  if (!g) return set_err(RGP_EINVAL, "null plan");
  if (!g->ws) return set_err(RGP_EWORKSPACE, "rgp_grcn: workspace not bound");
  if (!g->weights_set) return set_err(RGP_ESTATE, "rgp_grcn: weights not set");
  return grcn_check_error(g);
}

}  // namespace

int grcn_check_error(rgp_grcn* g) {
  if (g->err_host && *(volatile unsigned*)g->err_host) {
    *(volatile unsigned*)g->err_host = 0u;
    return set_err(RGP_ETIMEOUT, "rgp_grcn: a persistent ConvGRU launch of this plan lost a group member (another launch was "
                   "resident on the device?): its outputs were NaN-poisoned");
  }
  return RGP_OK;
}

extern "C" {

int rgp_grcn_create(rgp_grcn_t** plan, int batch, int n_steps, int dim_proj, int dim_state, int dtype,
                    int flags) {
  RGP_REQUIRE(plan, "rgp_grcn_create: null out pointer");
  RGP_REQUIRE((flags & ~(RGP_GRCN_SAVE_FOR_BACKWARD | RGP_GRCN_PER_STEP | RGP_GRCN_UNFOLDED_HEAD)) == 0,
              "rgp_grcn_create: unknown flags 0x%x", flags);
  const int save_for_backward = flags & RGP_GRCN_SAVE_FOR_BACKWARD;
  RGP_REQUIRE(batch > 0 && n_steps > 0, "rgp_grcn_create: batch=%d n_steps=%d", batch, n_steps);
  RGP_REQUIRE(dtype == RGP_F32 || dtype == RGP_BF16, "rgp_grcn_create: dtype %d", dtype);
  const int Bk = bke(dtype);
  RGP_REQUIRE(dim_proj > 0 && dim_proj % 64 == 0 && dim_state > 0 && dim_state % 64 == 0,
              "rgp_grcn_create: dim_proj=%d dim_state=%d must be multiples of 64", dim_proj, dim_state);
  RGP_REQUIRE((long long)batch * n_steps * 2401 < (1LL << 31), "rgp_grcn_create: B*T too large");
  (void)Bk;
  rgp_grcn* g = new rgp_grcn();
  g->B = batch; g->T = n_steps; g->P = dim_proj; g->S = dim_state; g->dtype = dtype; g->save = save_for_backward;
  g->F = batch * n_steps;
  g->fold_head = !(flags & RGP_GRCN_UNFOLDED_HEAD);
  const int P = g->P, S = g->S, F = g->F, es = esize(dtype);
  bool ok = true;

  // projection  E = X W + b  (gaze_grcn.py:239-242), output halo-padded 9x9xP
  for (ConvDesc* d : {&g->proj, &g->proj_rows}) {
    d->Mw = 49; d->N = P;
    d->in_img_stride = 49LL * 1024; d->out_img_stride = 81LL * P;
    for (int p = 0; p < 49; ++p) d->in_tab.push_back(p * 1024);
    d->out_tab = pad_tab(7, 1, P);
  }
  ok &= build_k_schedule(g->proj, {0}, {0}, 1024, dtype);
  g->proj.s_tap = 0; g->proj.s_n = 1; g->proj.s_c = P;
  // rows from C3D carry K order d*512+c; reference channel = c*2+d
  ok &= build_k_schedule(g->proj_rows, {0, 512}, {0, 1}, 512, dtype);
  g->proj_rows.s_tap = P; g->proj_rows.s_n = 1; g->proj_rows.s_c = 2LL * P;

  // 3x3 SAME convs on the padded 7x7 maps
  auto conv3x3 = [&](ConvDesc& d, int Cin, int N) {
    d.Mw = 49; d.N = N; d.in_img_stride = 81LL * Cin;
    for (int y = 0; y < 7; ++y) for (int x = 0; x < 7; ++x) d.in_tab.push_back((y * 9 + x) * Cin);
    std::vector<int> tapoff, fidx;
    for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) { tapoff.push_back((ky * 9 + kx) * Cin); fidx.push_back(ky * 3 + kx); }
    bool r = build_k_schedule(d, tapoff, fidx, Cin, dtype);
    d.s_tap = (long long)Cin * S;  // HWIO filters [3,3,Cin,S]: src[(tap*Cin + c)*S + n]
    d.s_n = 1; d.s_c = S;
    return r;
  };
  ok &= conv3x3(g->xconv, P, 3 * S);
  for (int p = 0; p < 49; ++p) g->xconv.out_tab.push_back(p * 3 * S);
  g->xconv.out_img_stride = 49LL * 3 * S;
  ok &= conv3x3(g->gzr, S, 2 * S);
  g->gzr.out_tab = pad_tab(7, 1, S); g->gzr.out_img_stride = 81LL * S;
  ok &= conv3x3(g->gc, S, S);
  g->gc.out_tab = pad_tab(7, 1, S); g->gc.out_img_stride = 81LL * S;

  ok &= build_deconv_phases(g->d1, g->d1_pack, 5, 3, 7, 1, S, 23, 2, 64, dtype);    // gaze_grcn.py:326-333
  ok &= build_deconv_phases(g->d2, g->d2_pack, 5, 2, 23, 2, 64, 49, 3, 32, dtype);  // gaze_grcn.py:336-343

  // 7x7 SAME stride-1 transposed conv folded with the 12->1 projection
  // (gaze_grcn.py:353-361): logit[y,x] = sum in[y-a+3, x-b+3, c] G[a,b,c] + out_b.
  {
    ConvDesc& d = g->d3;
    d.Mw = 49; d.N = 1; d.in_img_stride = 55LL * 55 * 32; d.out_img_stride = 2401;     // pixel column x = 48 only
    for (int y = 0; y < 49; ++y) { d.in_tab.push_back((y * 55 + 48) * 32); d.out_tab.push_back(y * 49 + 48); }
    std::vector<int> tapoff, fidx;
    for (int u = 0; u < 7; ++u) for (int v = 0; v < 7; ++v) { tapoff.push_back((u * 55 + v) * 32); fidx.push_back((6 - u) * 7 + (6 - v)); }
    ok &= build_k_schedule(d, tapoff, fidx, 32, dtype);
    d.s_tap = 32; d.s_n = 0; d.s_c = 1;
  }
  {  // pixel columns 0..47 in blocks of 16 (run_d3)
    ConvDesc& d = g->d3t;
    d.Mw = 49 * 3; d.N = 16; d.in_img_stride = 55LL * 55 * 32; d.out_img_stride = 2401;
    for (int y = 0; y < 49; ++y) for (int xb = 0; xb < 3; ++xb) { d.in_tab.push_back((y * 55 + 16 * xb) * 32); d.out_tab.push_back(y * 49 + 16 * xb); }
    std::vector<int> tapoff, fidx;
    for (int u = 0; u < 7; ++u) { tapoff.push_back(u * 55 * 32); fidx.push_back(u); }
    ok &= build_k_schedule(d, tapoff, fidx, 22 * 32, dtype);
    d.s_tap = 16LL * 704; d.s_n = 704; d.s_c = 1;           // gtoep [u][n][x'*32 + c]
  }
  if (g->fold_head) {
    // the folded head (head_fold.hip.h): GEMM rows = the 7x7 positions of the padded BN(h) image, K = S, N = the 19x19 taps
    ConvDesc& d = g->hfold;
    d.Mw = 49; d.N = HF_PK; d.in_img_stride = 81LL * S; d.out_img_stride = 49LL * HF_PK;
    for (int m = 0; m < 7; ++m) for (int n = 0; n < 7; ++n) { d.in_tab.push_back(((m + 1) * 9 + n + 1) * S); d.out_tab.push_back((m * 7 + n) * HF_PK); }
    ok &= build_k_schedule(d, {0}, {0}, S, dtype);
    d.s_tap = 0; d.s_n = S; d.s_c = 1;                          // source K [(r,t)][s]
  }
  if (!ok) { delete g; return set_err(RGP_EINVAL, "rgp_grcn_create: unsupported channel geometry P=%d S=%d", P, S); }

  Arena a;
  if (g->fold_head) g->hfold.reserve(a, dtype);
  for (ConvDesc* d : {&g->proj, &g->proj_rows, &g->xconv, &g->gzr, &g->gc, &g->d3, &g->d3t}) d->reserve(a, dtype);
  for (ConvDesc& d : g->d1) d.reserve(a, dtype);
  for (ConvDesc& d : g->d2) d.reserve(a, dtype);
  // packing aliases: a tap table of their own, the packed filter of the problem they belong to
  auto reserve_aliases = [&](std::vector<ConvDesc>& al, const std::vector<ConvDesc>& ds) {
    const size_t per = al.size() / ds.size();
    for (size_t i = 0; i < al.size(); ++i) {
      const int n = al[i].N;
      al[i].N = 0;                                             // (no filter area of its own)
      al[i].reserve(a, dtype);
      al[i].N = n;
      al[i].w_off = ds[i / per].w_off;
    }
  };
  reserve_aliases(g->d1_pack, g->d1);
  reserve_aliases(g->d2_pack, g->d2);
  g->tab_pad9_P = pad_tab(7, 1, P); g->o_pad9_P = put_tab(a, g->tab_pad9_P);
  g->tab_pad9_S = pad_tab(7, 1, S); g->o_pad9_S = put_tab(a, g->tab_pad9_S);
  g->tab_pad27 = pad_tab(23, 2, 64); g->o_pad27 = put_tab(a, g->tab_pad27);
  g->tab_pad55 = pad_tab(49, 3, 32); g->o_pad55 = put_tab(a, g->tab_pad55);
  for (int p = 0; p < 49; ++p) { g->tab_lin49_3S.push_back(p * 3 * S); g->tab_lin49_S.push_back(p * S); }
  g->o_lin49_3S = put_tab(a, g->tab_lin49_3S); g->o_lin49_S = put_tab(a, g->tab_lin49_S);

  const size_t st = (size_t)batch * 49 * S * 4;
  g->xt = take(a, (size_t)F * 49 * 1024 * es);
  g->E = take(a, (size_t)F * 81 * P * es);
  g->xpre = take(a, (size_t)F * 49 * 3 * S * 4);
  g->hall = take(a, st * (n_steps + 1));
  g->uall = take(a, st * n_steps);
  if (g->save) { g->rall = take(a, st * n_steps); g->call = take(a, st * n_steps); }
  g->hp = take(a, (size_t)batch * 81 * S * es);
  g->rhp = take(a, (size_t)batch * 81 * S * es);
  g->hbn = take(a, (size_t)F * 81 * S * es);
  if (g->fold_head) {                                      // no intermediate maps: the fold's small fp32 work areas instead
    g->hf_h = take(a, (size_t)HF_HP * HF_HP * 64 * 4);
    g->hf_k = take(a, (size_t)HF_KP * HF_KP * S * 4);
    g->hf_z = take(a, (size_t)F * 49 * HF_PK * 4);
    g->hf_part = take(a, (size_t)5 * HF_KP * HF_KP * S * 4);
  } else {
    g->D1 = take(a, (size_t)F * 27 * 27 * 64 * es);
    g->D2 = take(a, (size_t)F * 55 * 55 * 32 * es + 4096);   // slack: the Toeplitz filter-gradient rows of pixel block 3 read past the last row
  }
  g->gfold = take(a, 50 * 32 * 4);
  g->gtoep = take(a, (size_t)7 * 16 * 704 * 4);
  g->bias16 = take(a, 16 * 4);
  g->frame_loss = take(a, (size_t)F * 4);
  // persistent sequence kernel (convgru_seq.hip.h): the reference cell (128 state channels on 7x7), bf16 operands,
  // up to 2 clips per group of 8 workgroups and at most 32 groups (= the 256 CUs)
  if (dtype == RGP_BF16 && S == 128 && batch <= 64 && !(flags & RGP_GRCN_PER_STEP)) {
    g->seq_nc = (batch + 31) / 32;
    g->seq_groups = (batch + g->seq_nc - 1) / g->seq_nc;
    g->xch_h = take(a, (size_t)g->seq_groups * 98 * 128 * 2);
    g->xch_rh = take(a, (size_t)g->seq_groups * 98 * 128 * 2);
    g->seq_cnt = take(a, (size_t)g->seq_groups * 2 * n_steps * 4);
  }
  if (g->save) {
    const int rc = grcn_bwd_plan(g, a);
    if (rc != RGP_OK) { delete g; return rc; }
  }
  g->ws_bytes = a.off;
  *plan = g;
  return RGP_OK;
}

int rgp_grcn_destroy(rgp_grcn_t* plan) {
  if (plan) grcn_bwd_destroy(plan);
  if (plan && plan->err_host) (void)hipHostFree(plan->err_host);
  delete plan;
  return RGP_OK;
}

int rgp_grcn_status(rgp_grcn_t* g, rgp_stream_t stream) {
  RGP_REQUIRE(g, "rgp_grcn_status: null plan");
  RGP_HIP(hipStreamSynchronize((hipStream_t)stream));
  return grcn_check_error(g);
}

int rgp_grcn_inject_fault(rgp_grcn_t* g, int kind) {
  RGP_REQUIRE(g && (kind == RGP_FAULT_SEQ_LOST_MEMBER || kind == RGP_FAULT_BPTT_LOST_MEMBER), "rgp_grcn_inject_fault: bad arguments");
  if (!seq_persistent_ok(g)) return set_err(RGP_ESTATE, "rgp_grcn_inject_fault: the plan does not use the persistent ConvGRU kernels");
  g->fault |= kind;
  return RGP_OK;
}

size_t rgp_grcn_workspace_bytes(const rgp_grcn_t* plan) { return plan ? plan->ws_bytes : 0; }

int rgp_grcn_bind_workspace(rgp_grcn_t* g, void* workspace, size_t bytes, rgp_stream_t stream) {
  RGP_REQUIRE(g && workspace, "rgp_grcn_bind_workspace: null argument");
  if (bytes < g->ws_bytes) return set_err(RGP_EWORKSPACE, "workspace %zu < required %zu bytes", bytes, g->ws_bytes);
  RGP_REQUIRE(((size_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  if (g->seq_groups > 0 && !g->err_host) {
    // error word of the persistent kernels: pinned host memory the device writes directly, so the host can test it at
    // the start of any later call without a synchronisation (host memory, not device memory: header conventions)
    void* e = nullptr;
    RGP_HIP(hipHostMalloc(&e, 64, hipHostMallocMapped));
    g->err_host = (unsigned*)e;
    *(volatile unsigned*)g->err_host = 0u;
  }
  g->ws = (char*)workspace;
  g->weights_set = false;
  // zero everything once: halos of E / Hp / RHp / Hbn / D1 / D2 stay zero because
  // epilogues only ever write interiors.
  RGP_HIP(hipMemsetAsync(g->ws, 0, g->ws_bytes, s));
  for (ConvDesc* d : {&g->proj, &g->proj_rows, &g->xconv, &g->gzr, &g->gc, &g->d3, &g->d3t}) RGP_TRY(upload_desc(*d, g->ws, s));
  if (g->fold_head) RGP_TRY(upload_desc(g->hfold, g->ws, s));
  for (ConvDesc& d : g->d1) RGP_TRY(upload_desc(d, g->ws, s));
  for (ConvDesc& d : g->d2) RGP_TRY(upload_desc(d, g->ws, s));
  for (ConvDesc& d : g->d1_pack) RGP_TRY(upload_desc(d, g->ws, s));
  for (ConvDesc& d : g->d2_pack) RGP_TRY(upload_desc(d, g->ws, s));
  auto up = [&](const std::vector<int>& t, size_t off) -> int {
    RGP_HIP(hipMemcpyAsync(g->ws + off, t.data(), t.size() * 4, hipMemcpyHostToDevice, s));
    return RGP_OK;
  };
  RGP_TRY(up(g->tab_pad9_P, g->o_pad9_P)); RGP_TRY(up(g->tab_pad9_S, g->o_pad9_S));
  RGP_TRY(up(g->tab_pad27, g->o_pad27)); RGP_TRY(up(g->tab_pad55, g->o_pad55));
  RGP_TRY(up(g->tab_lin49_3S, g->o_lin49_3S)); RGP_TRY(up(g->tab_lin49_S, g->o_lin49_S));
  if (g->save) RGP_TRY(grcn_bwd_upload(g, s));
  return RGP_OK;
}

int rgp_grcn_set_weights(rgp_grcn_t* g, const rgp_grcn_weights* w, rgp_stream_t stream) {
  RGP_REQUIRE(g && w, "rgp_grcn_set_weights: null argument");
  if (!g->ws) return set_err(RGP_EWORKSPACE, "rgp_grcn: workspace not bound");
  const float* const* ptrs = (const float* const*)w;
  for (size_t i = 0; i < sizeof(rgp_grcn_weights) / sizeof(float*); ++i)
    RGP_REQUIRE(ptrs[i], "rgp_grcn_set_weights: weight pointer %zu is null", i);
  hipStream_t s = (hipStream_t)stream;
  hipStream_t sc = s;
  if (g->save) RGP_TRY(grcn_bwd_fork_fold(g, s, &sc));
  RGP_TRY(g->dtype == RGP_BF16 ? set_weights_impl<bf16_t>(g, w, s, sc) : set_weights_impl<float>(g, w, s, sc));
  if (g->save) RGP_TRY(grcn_bwd_pack(g, w, s, sc));
  if (sc != s) RGP_TRY(grcn_bwd_join_fold(g, s));
  return RGP_OK;
}

int rgp_proj_fwd(rgp_grcn_t* g, const float* c3d_input, rgp_stream_t stream) {
  RGP_TRY(check_ready(g));
  RGP_REQUIRE(c3d_input, "rgp_proj_fwd: null input");
  hipStream_t s = (hipStream_t)stream;
  const int pid = g->prof.begin(0, s);
  const int rc = g->dtype == RGP_BF16 ? proj_impl<bf16_t>(g, c3d_input, nullptr, s) : proj_impl<float>(g, c3d_input, nullptr, s);
  g->prof.end(pid, s);
  return rc;
}

int rgp_convgru_xconv_fwd(rgp_grcn_t* g, rgp_stream_t stream) {
  RGP_TRY(check_ready(g));
  hipStream_t s = (hipStream_t)stream;
  const int pid = g->prof.begin(1, s);
  const int rc = g->dtype == RGP_BF16 ? xconv_impl<bf16_t>(g, s) : xconv_impl<float>(g, s);
  g->prof.end(pid, s);
  return rc;
}

int rgp_convgru_seq_fwd(rgp_grcn_t* g, rgp_stream_t stream) {
  RGP_TRY(check_ready(g));
  hipStream_t s = (hipStream_t)stream;
  const int pid = g->prof.begin(2, s);
  const int rc = g->dtype == RGP_BF16 ? seq_impl<bf16_t>(g, s) : seq_impl<float>(g, s);
  g->prof.end(pid, s);
  return rc;
}

int rgp_head_fwd(rgp_grcn_t* g, float* logits, rgp_stream_t stream) {
  RGP_TRY(check_ready(g));
  RGP_REQUIRE(logits, "rgp_head_fwd: null logits");
  hipStream_t s = (hipStream_t)stream;
  const int pid = g->prof.begin(3, s);
  const int rc = g->dtype == RGP_BF16 ? head_impl<bf16_t>(g, logits, s) : head_impl<float>(g, logits, s);
  g->prof.end(pid, s);
  return rc;
}

static int grcn_tail(rgp_grcn_t* g, float* logits, float* probs, rgp_stream_t stream) {
  RGP_TRY(rgp_convgru_xconv_fwd(g, stream));
  RGP_TRY(rgp_convgru_seq_fwd(g, stream));
  RGP_TRY(rgp_head_fwd(g, logits, stream));
  if (probs) {
    const int pid = g->prof.begin(4, (hipStream_t)stream);
    RGP_TRY(rgp_softmax_xent_fwd(logits, nullptr, probs, nullptr, nullptr, g->F, 2401, stream));
    g->prof.end(pid, (hipStream_t)stream);
  }
  return RGP_OK;
}

int rgp_grcn_forward(rgp_grcn_t* g, const float* c3d_input, float* logits, float* probs, rgp_stream_t stream) {
  RGP_TRY(rgp_proj_fwd(g, c3d_input, stream));
  return grcn_tail(g, logits, probs, stream);
}

int rgp_grcn_forward_rows(rgp_grcn_t* g, const void* c3d_rows, float* logits, float* probs, rgp_stream_t stream) {
  RGP_TRY(check_ready(g));
  RGP_REQUIRE(c3d_rows && logits, "rgp_grcn_forward_rows: null argument");
  hipStream_t s = (hipStream_t)stream;
  const int pid = g->prof.begin(0, s);
  RGP_TRY(g->dtype == RGP_BF16 ? proj_impl<bf16_t>(g, nullptr, c3d_rows, s) : proj_impl<float>(g, nullptr, c3d_rows, s));
  g->prof.end(pid, s);
  return grcn_tail(g, logits, probs, stream);
}

int rgp_grcn_profile_enable(rgp_grcn_t* g, int enable) {
  RGP_REQUIRE(g, "rgp_grcn_profile_enable: null plan");
  g->prof.enabled = enable != 0;
  return RGP_OK;
}

int rgp_grcn_profile_read(rgp_grcn_t* g, double ms[RGP_GRCN_STAGES], long long calls[RGP_GRCN_STAGES]) {
  RGP_REQUIRE(g && ms && calls, "rgp_grcn_profile_read: null argument");
  return g->prof.read(ms, calls, RGP_GRCN_STAGES);
}

struct BufView {
  size_t off; size_t tab; int rows, C; long long img_stride; long long imgs; bool f32;
};

static bool find_view(const rgp_grcn* g, const char* name, BufView& v) {
  const int S = g->S, P = g->P;
  const std::string n(name ? name : "");
  if (n == "c3d_embedded") v = {g->E.off, g->o_pad9_P, 49, P, 81LL * P, g->F, false};
  else if (n == "xpre") v = {g->xpre.off, g->o_lin49_3S, 49, 3 * S, 49LL * 3 * S, g->F, true};
  else if (n == "bn") v = {g->hbn.off, g->o_pad9_S, 49, S, 81LL * S, g->F, false};
  else if ((n == "d1" || n == "d2") && g->fold_head) return false;      // the folded head has no intermediate maps
  else if (n == "d1") v = {g->D1.off, g->o_pad27, 529, 64, 27LL * 27 * 64, g->F, false};
  else if (n == "d2") v = {g->D2.off, g->o_pad55, 2401, 32, 55LL * 55 * 32, g->F, false};
  else if (n == "u") v = {g->uall.off, g->o_lin49_S, 49, S, 49LL * S, (long long)g->T * g->B, true};
  else if (n == "r" && g->save) v = {g->rall.off, g->o_lin49_S, 49, S, 49LL * S, (long long)g->T * g->B, true};
  else if (n == "c" && g->save) v = {g->call.off, g->o_lin49_S, 49, S, 49LL * S, (long long)g->T * g->B, true};
  else if (n == "h_all") v = {g->hall.off + (size_t)g->B * 49 * S * 4, g->o_lin49_S, 49, S, 49LL * S, (long long)g->T * g->B, true};
  else return false;
  return true;
}

size_t rgp_grcn_buffer_elems(const rgp_grcn_t* g, const char* name) {
  BufView v;
  if (!g) return 0;
  if (name && std::string(name) == "rcn_outputs") return (size_t)g->F * 49 * g->S;
  if (!find_view(g, name, v)) return 0;
  return (size_t)v.imgs * v.rows * v.C;
}

// [T,B,49,S] -> [B,T,49,S]
static __global__ void tb_to_bt_kernel(const float* __restrict__ src, float* __restrict__ dst, int T_, int B_, long long inner) {
  const long long total = (long long)T_ * B_ * inner;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long in = i % inner;
    const int t = (int)((i / inner) % T_);
    const int b = (int)(i / (inner * T_));
    dst[i] = src[((long long)t * B_ + b) * inner + in];
  }
}

int rgp_grcn_read_buffer(rgp_grcn_t* g, const char* name, float* dst, rgp_stream_t stream) {
  RGP_REQUIRE(g && g->ws && name && dst, "rgp_grcn_read_buffer: null argument");
  hipStream_t s = (hipStream_t)stream;
  if (std::string(name) == "rcn_outputs") {   // h_1..h_T as [B,T,7,7,S] (gaze_grcn.py:288)
    const float* src = (const float*)(g->ws + g->hall.off) + (size_t)g->B * 49 * g->S;
    tb_to_bt_kernel<<<1024, 256, 0, s>>>(src, dst, g->T, g->B, 49LL * g->S);
    RGP_HIP(hipGetLastError());
    return RGP_OK;
  }
  BufView v;
  if (!find_view(g, name, v)) return set_err(RGP_EINVAL, "rgp_grcn_read_buffer: unknown buffer '%s'", name);
  const long long total = v.imgs * v.rows * v.C;
  const int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
  const int* tab = (const int*)(g->ws + v.tab);
  if (v.f32) unpad_kernel<float><<<blocks, 256, 0, s>>>((const float*)(g->ws + v.off), dst, tab, v.rows, v.C, v.img_stride, total);
  else if (g->dtype == RGP_BF16) unpad_kernel<bf16_t><<<blocks, 256, 0, s>>>((const bf16_t*)(g->ws + v.off), dst, tab, v.rows, v.C, v.img_stride, total);
  else unpad_kernel<float><<<blocks, 256, 0, s>>>((const float*)(g->ws + v.off), dst, tab, v.rows, v.C, v.img_stride, total);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

}  // extern "C"
