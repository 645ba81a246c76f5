// librgp_hip.so: the two-level cascade model (BASELINE config 5), forward.
// Reference graph: /root/reference/models/gaze_grcn_cascade.py:188-441.  That file does not
// build as committed (the top cell is declared with 65 input channels but fed 64,
// removed TF APIs); this implements its evident intent (commented block :370-377): the top
// cell's input is concat(upsampled bottom state [64 ch], ShallowNet saliency [1 ch]).
//
//   ShallowNet(frames) -> sal [B,T,49,49]                        (saliency_shallownet.py:74-216)
//   proj 1024->512, bottom ConvGRU 512->256 on 7x7               (:267-313)
//   conv2d_transpose 11x11 stride 7 SAME, 256->64, 7x7 -> 49x49  (:317-336)
//   top GRU_RCN_Cell(3 units, 65 features, 49x49, 5x5)            (:346-381)
//   flatten 49*49*3 -> fc 4802 ReLU maxout -> fc 4802 ReLU maxout -> [B,T,49,49]   (:383-423)
//
// Device side: the bottom level is an rgp_grcn sub-plan (its batch-norm set to identity), the
// frame saliency an rgp_shallownet sub-plan; the stride-7 transposed conv is 49 sub-pixel
// phases; the top cell reuses the ConvGRU epilogues with its 3 units / 65 features zero-padded
// to 16 / 128 channels; the FCs reuse the ShallowNet's ReLU+maxout epilogue.
#include <algorithm>
#include <string>

#include "rgp_cascade_plan.h"

using namespace rgp;

namespace {

// tf.nn.conv2d_transpose(SAME, stride s, k x k, filter [k,k,Cout,Cin]) as gather-form phases:
//   out[Y, X, o] = sum in[i - a', j - b', c] F[py + s a', px + s b', o, c],  s*i + py = Y + crop
// (crop = SAME pad_before of the forward conv).  Output rows are written with channel stride
// out_cs into a halo-padded image.
bool build_deconv_same_phases(std::vector<ConvDesc>& out, int k, int s, int H, int hin, int Cin, int OH, int crop,
                              int hout, int out_cs, int Cout, int dtype) {
  const int Wp = H + 2 * hin, OWp = OH + 2 * hout;
  auto range = [&](int py, int& lo, int& hi) {
    lo = std::max(0, (crop - py + s - 1) / s);
    hi = (OH - 1 + crop - py) / s;            // inclusive
  };
  for (int py = 0; py < s; ++py)
    for (int px = 0; px < s; ++px) {
      const int ta = (k - py + s - 1) / s, tb = (k - px + s - 1) / s;
      int ilo, ihi, jlo, jhi;
      range(py, ilo, ihi);
      range(px, jlo, jhi);
      if (ihi < ilo || jhi < jlo) continue;
      if (ilo - (ta - 1) < -hin || jlo - (tb - 1) < -hin || ihi > H - 1 + hin || jhi > H - 1 + hin) return false;
      ConvDesc d;
      d.Mw = (ihi - ilo + 1) * (jhi - jlo + 1);
      d.N = Cout;
      d.in_img_stride = (long long)Wp * Wp * Cin;
      d.out_img_stride = (long long)OWp * OWp * out_cs;
      for (int i = ilo; i <= ihi; ++i)
        for (int j = jlo; j <= jhi; ++j) {
          d.in_tab.push_back(((i - (ta - 1) + hin) * Wp + (j - (tb - 1) + hin)) * Cin);
          d.out_tab.push_back(((s * i + py - crop + hout) * OWp + (s * j + px - crop + hout)) * out_cs);
        }
      std::vector<int> tapoff, fidx;
      for (int u = 0; u < ta; ++u)
        for (int v = 0; v < tb; ++v) {
          tapoff.push_back((u * Wp + v) * Cin);
          fidx.push_back((py + s * (ta - 1 - u)) * k + (px + s * (tb - 1 - v)));
        }
      if (!build_k_schedule(d, tapoff, fidx, Cin, dtype)) return false;
      d.s_tap = (long long)Cout * Cin;
      d.s_n = Cin;
      d.s_c = 1;
      out.push_back(d);
    }
  return true;
}

bool conv5x5_desc(ConvDesc& d, int Cin, int N, long long ldc, int dtype) {
  d.Mw = 2401; d.N = N; d.in_img_stride = (long long)kHp * kHp * Cin; d.out_img_stride = 2401LL * ldc;
  std::vector<int> tapoff, fidx;
  for (int y = 0; y < 49; ++y) for (int x = 0; x < 49; ++x) { d.in_tab.push_back((y * kHp + x) * Cin); d.out_tab.push_back((y * 49 + x) * (int)ldc); }
  for (int ky = 0; ky < 5; ++ky) for (int kx = 0; kx < 5; ++kx) { tapoff.push_back((ky * kHp + kx) * Cin); fidx.push_back(ky * 5 + kx); }
  return build_k_schedule(d, tapoff, fidx, Cin, dtype);
}

// saliency map -> channel 64 of the top cell's halo-padded input image
template <typename T>
// (frames f = j * f_mul + f_add, j < total / 2401: all of them, or the B frames of one time step)
__global__ __launch_bounds__(256) void put_saliency_kernel(const float* __restrict__ sal, T* __restrict__ xtop, long long total, int f_mul,
                                                           int f_add) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int p = (int)(i % 2401);
    const long long f = (i / 2401) * f_mul + f_add;
    const int y = p / 49, x = p % 49;
    xtop[(f * kHp * kHp + (y + 2) * kHp + x + 2) * kCt + 64] = Elem<T>::to(sal[f * 2401 + p]);
  }
}

// top-cell states [F][2401][kSt] -> FC rows [F][Kfc] with the reference's flatten order (y, x, unit)
template <typename T>
__global__ __launch_bounds__(256) void compact_units_kernel(const T* __restrict__ h, T* __restrict__ rows, long long F,
                                                            int Kfc) {
  const long long total = F * 2401 * 3;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % 3);
    const int p = (int)((i / 3) % 2401);
    const long long f = i / (3 * 2401);
    rows[f * Kfc + p * 3 + c] = h[(f * 2401 + p) * kSt + c];
  }
}

__global__ void interleave2_kernel(const float* __restrict__ b, float* __restrict__ out, int half) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < half) { out[2 * j] = b[j]; out[2 * j + 1] = b[j + half]; }
}

__global__ void fill2_kernel(float* p, float v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

template <typename T>
int set_weights_impl(rgp_cascade* g, const rgp_cascade_weights* w, hipStream_t s) {
  char* ws = g->ws;
  // bottom level: ConvGRU 512 -> 256 through the grcn sub-plan; its head filters are unused (zeros),
  // its per-timestep BN is the identity: gamma = sqrt(1 + eps), beta = 0
  rgp_grcn_weights bw;
  memset(&bw, 0, sizeof(bw));
  bw.proj_c3d_W = w->proj_c3d_W; bw.proj_c3d_b = w->proj_c3d_b;
  bw.gru_Wz = w->bottom_Wz; bw.gru_Uz = w->bottom_Uz; bw.gru_Wr = w->bottom_Wr; bw.gru_Ur = w->bottom_Ur;
  bw.gru_W = w->bottom_W; bw.gru_U = w->bottom_U;
  float* bn_id = (float*)(ws + g->bn_id);            // [T*256] gamma | [T*256] beta | dummy head weights (zeros)
  const int nbn = g->T * 256;
  fill2_kernel<<<(nbn + 255) / 256, 256, 0, s>>>(bn_id, sqrtf(1.0f + 1e-3f), nbn);
  bw.bn_gamma = bn_id; bw.bn_beta = bn_id + nbn;
  const float* zeros = bn_id + 2 * nbn;              // >= 25*64*256 zero floats
  bw.up_weight1 = zeros; bw.up_weight2 = zeros; bw.up_weight3 = zeros; bw.out_W = zeros; bw.out_b = zeros;
  RGP_TRY(rgp_grcn_set_weights(g->bottom, &bw, (rgp_stream_t)s));
  rgp_shallownet_weights sw = w->shallownet;
  RGP_TRY(rgp_shallownet_set_weights(g->shallow, &sw, (rgp_stream_t)s));
  // (packs batched, no memsets of the packed areas: rgp_grcn.hip set_weights_impl)
  PackBatch<T> pk(ws, s);
  // stride-7 transposed conv phases, filter [11,11,64,256]
  for (ConvDesc& d : g->up) {
    RGP_TRY(pk.add(d, w->upsampling_weight, 64, 0));
  }
  // top cell: filters [5,5,65,3] (x) and [5,5,3,3] (h); gate g, unit n -> packed row g*kSt + n
  g->xtop.cin_src = 65; g->xtop.s_tap = 65LL * 3; g->xtop.s_c = 3; g->xtop.s_n = 1;
  RGP_TRY(pk.add(g->xtop, w->top_Wz, 3, 0));
  RGP_TRY(pk.add(g->xtop, w->top_Wr, 3, kSt));
  RGP_TRY(pk.add(g->xtop, w->top_W, 3, 2 * kSt));
  for (ConvDesc* d : {&g->zr, &g->c}) { d->cin_src = 3; d->s_tap = 3LL * 3; d->s_c = 3; d->s_n = 1; }
  RGP_TRY(pk.add(g->zr, w->top_Uz, 3, 0));
  RGP_TRY(pk.add(g->zr, w->top_Ur, 3, kSt));
  RGP_TRY(pk.add(g->c, w->top_U, 3, 0));
  // FCs with interleaved halves (see EpiReluMaxout)
  g->fc1.cin_src = 7203; g->fc2.cin_src = 2401;
  for (ConvDesc* d : {&g->fc1, &g->fc2}) { d->s_tap = 0; d->s_n = 1; d->s_c = 4802; }
  RGP_TRY(pk.add(g->fc1, w->fc1_w, 2401, 0, 0, 0, 2));
  RGP_TRY(pk.add(g->fc1, w->fc1_w + 2401, 2401, 1, 0, 0, 2));
  RGP_TRY(pk.add(g->fc2, w->fc2_w, 2401, 0, 0, 0, 2));
  RGP_TRY(pk.add(g->fc2, w->fc2_w + 2401, 2401, 1, 0, 0, 2));
  RGP_TRY(pk.flush());
  interleave2_kernel<<<(2401 + 255) / 256, 256, 0, s>>>(w->fc1_b, (float*)(ws + g->b1i), 2401);
  interleave2_kernel<<<(2401 + 255) / 256, 256, 0, s>>>(w->fc2_b, (float*)(ws + g->b2i), 2401);
  RGP_HIP(hipGetLastError());
  g->weights_set = true;
  return RGP_OK;
}

template <typename T>
int forward_impl(rgp_cascade* g, const float* frames, const float* c3d_input, float* maps, hipStream_t s) {
  char* ws = g->ws;
  const int B = g->B, T_ = g->T, F = g->F;
  constexpr int G16 = sizeof(T) == 2 ? 4 : 2;      // 16-channel taps per 128-byte chunk
  rgp_stream_t rs = (rgp_stream_t)s;
  // Three chains, one time step apart (rgp_cascade_plan.h): `s` runs the bottom level, `sc` feeds the top cell (frame saliency
  // first, then per step: the 49 upsampling phases, the saliency channel, the input convolution), `st` runs the top cell.
  // Every link of each chain is a launch on a fraction of the CUs: together 35 x 49 us instead of 35 x (49 + 30) us plus the
  // hoisted convolutions.  Inside a stream capture, or with RGP_CASCADE_PIPE=0 (dev), one chain with the hoisted forms.
  hipStream_t sc = s, st2 = s;
  RGP_TRY(g->fork(s, 0, &sc));
  // (the per-step events exist only where the bottom recurrence is per-step launches: S = 256 always is)
  const bool pipe = sc != s && dev_knob("RGP_CASCADE_PIPE", 1) && g->bottom->seq_groups <= 0 && g->pipe_ok(s, T_);
  if (pipe) {
    st2 = g->side2;
    RGP_HIP(hipStreamWaitEvent(st2, g->ev[0], 0));            // behind everything queued on s, as sc
  }
  RGP_TRY(rgp_shallownet_forward(g->shallow, frames, F, (float*)(ws + g->sal), nullptr, (rgp_stream_t)sc));
  // The feed chain is the longest of the three (three launches per step) and its steps do not depend on each other: odd
  // steps go to the plan's third side stream, behind the frame saliency
  hipStream_t sc_odd = sc;
  if (pipe && dev_knob("RGP_CASCADE_FEED2", 1)) {
    sc_odd = g->side3;
    RGP_HIP(hipEventRecord(g->ev_join2, sc));
    RGP_HIP(hipStreamWaitEvent(sc_odd, g->ev_join2, 0));
  }
  g->bottom->step_ev = pipe ? g->ev_b.data() : nullptr;
  int rc = rgp_proj_fwd(g->bottom, c3d_input, rs);
  if (rc == RGP_OK) rc = rgp_convgru_xconv_fwd(g->bottom, rs);
  if (rc == RGP_OK) rc = rgp_convgru_seq_fwd(g->bottom, rs);
  g->bottom->step_ev = nullptr;
  RGP_TRY(rc);
  const int es_ = (int)sizeof(T);
  const long long up_in_img = 81LL * 256, xtop_img = (long long)kHp * kHp * kCt, xpre_img = 2401LL * 3 * kSt;
  // (3) 7x7x256 -> 49x49x64 into channels 0..63 of the top cell's input; saliency into channel 64
  //     (49 sub-pixel phases, 215 tiles each: one grouped launch -- one by one they cost 49 x 24 us at 16 x 35)
  // (4) top cell: the x-part of its gates
  auto feed = [&](int t, hipStream_t q) -> int {                // t < 0: every frame at once
    const int n_img = t < 0 ? F : B;
    const long long a_off = t < 0 ? 0 : t * up_in_img * es_, o_off = t < 0 ? 0 : t * xtop_img * es_;
    if (dev_knob("RGP_CASCADE_GROUPED", 1)) {
      RGP_TRY((launch_igemm_grouped<T, 1, 1, EpiStore<T, false, false>>(t < 0 ? g->up_p : g->up_p_step,
                                                                         (const IgemmParams*)(ws + (t < 0 ? g->up_p_off : g->up_ps_off)),
                                                                         (const EpiParams*)(ws + (t < 0 ? g->up_e_off : g->up_es_off)), q, a_off, o_off)));
    } else {
      for (size_t i = 0; i < g->up.size(); ++i) {
        IgemmParams p = t < 0 ? g->up_p[i] : g->up_p_step[i];
        EpiParams e = t < 0 ? g->up_e[i] : g->up_e_step[i];
        p.A = (const char*)p.A + a_off;
        e.out = (char*)e.out + o_off;
        RGP_TRY((launch_igemm<T, 1, 1, EpiStore<T, false, false>>(p, e, q)));
      }
    }
    const long long tot = (long long)n_img * 2401;
    put_saliency_kernel<T><<<(int)std::min<long long>((tot + 255) / 256, 8192), 256, 0, q>>>((const float*)(ws + g->sal), (T*)(ws + g->xtopbuf),
                                                                                       tot, t < 0 ? 1 : T_, t < 0 ? 0 : t);
    RGP_HIP(hipGetLastError());
    IgemmParams p = make_params(g->xtop, ws + g->xtopbuf + o_off, ws, n_img);
    EpiParams e = make_epi(g->xtop, (float*)(ws + g->xpre) + (t < 0 ? 0 : t * xpre_img), ws);
    if (t >= 0) { p.in_img_stride *= T_; e.out_img_stride *= T_; }
    return launch_igemm<T, 1, 1, EpiStore<float, false, false>>(p, e, q);
  };
  if (!pipe) {
    if (sc != s) RGP_TRY(g->join(s));
    RGP_TRY(feed(-1, s));
  }
  const size_t st = (size_t)B * 2401 * kSt;
  const long long img16 = (long long)kHp * kHp * kSt;
  // inference: two state snapshots and one operand image per role; training: every step's states, gates and
  // operand images are kept (hp_all slot (b, t) = h_{t-1} of step t, slot (b, 0) is never written = h_0 = 0)
  const bool save = g->save;
  if (!save) RGP_HIP(hipMemsetAsync(ws + g->hp, 0, (size_t)B * img16 * sizeof(T), st2));
  float* hall = (float*)(ws + (save ? g->hall_t : g->hall));
  RGP_HIP(hipMemsetAsync(hall, 0, st * 4, st2));
  for (int t = 0; t < T_; ++t) {
    if (pipe) {
      hipStream_t sf = (t & 1) ? sc_odd : sc;
      RGP_HIP(hipStreamWaitEvent(sf, g->ev_b[t], 0));           // the bottom state of step t
      RGP_TRY(feed(t, sf));
      RGP_HIP(hipEventRecord(g->ev_x[t], sf));
      RGP_HIP(hipStreamWaitEvent(st2, g->ev_x[t], 0));          // the x-part of step t's gates
    }
    hipStream_t s = st2;
    char* hp_in = save ? ws + g->hp_all + (size_t)t * img16 * sizeof(T) : ws + g->hp;
    char* hp_out = save ? ws + g->hp_all + (size_t)(t + 1) * img16 * sizeof(T) : ws + g->hp;
    char* rh_buf = save ? ws + g->rhp_all + (size_t)t * img16 * sizeof(T) : ws + g->rh;
    const long long hp_stride = save ? (long long)(T_ + 1) * img16 : img16, rh_stride = save ? (long long)T_ * img16 : img16;
    EpiParams e = make_epi(g->zr, rh_buf, ws);
    e.out_tab = (const int*)(ws + g->o_pad53_t);
    e.out_img_stride = rh_stride;
    e.xpre = (const float*)(ws + g->xpre) + (size_t)t * 2401 * 3 * kSt;
    e.xpre_img_stride = (long long)T_ * 2401 * 3 * kSt;
    e.xpre_ld = 3 * kSt;
    e.xpre_col = 0;
    e.S = kSt;
    e.state_rows = 2401;
    e.h_prev = hall + (size_t)(save ? t : (t & 1)) * st;
    e.h_next = hall + (size_t)(save ? t + 1 : ((t + 1) & 1)) * st;
    e.u_gate = save ? (float*)(ws + g->uall) + (size_t)t * st : (float*)(ws + g->u);
    e.r_save = save ? (float*)(ws + g->rall) + (size_t)t * st : nullptr;
    e.c_save = save ? (float*)(ws + g->call) + (size_t)t * st : nullptr;
    IgemmParams p = make_params(g->zr, hp_in, ws, B);
    p.in_img_stride = hp_stride;
    RGP_TRY((launch_igemm<T, G16, 1, EpiGruZR<T>>(p, e, s)));
    e.out = hp_out;
    e.out_img_stride = hp_stride;
    e.xpre_col = 2 * kSt;
    e.out2 = ws + g->hrows;
    e.out2_tab = (const int*)(ws + g->c.out_tab_off);      // dense [2401][kSt] rows
    e.out2_img_stride = 2401LL * kSt;
    e.out2_img_mul = T_;
    e.out2_img_add = t;
    e.bn_gamma = (const float*)(ws + g->ones);
    e.bn_beta = (const float*)(ws + g->zeros);
    e.bn_inv_std = 1.0f;
    IgemmParams pc = make_params(g->c, rh_buf, ws, B);
    pc.in_img_stride = rh_stride;
    RGP_TRY((launch_igemm<T, G16, 1, EpiGruC<T>>(pc, e, s)));
  }
  if (pipe) {
    RGP_HIP(hipEventRecord(g->ev_join2, st2));
    RGP_HIP(hipStreamWaitEvent(s, g->ev_join2, 0));
    RGP_TRY(g->join(s));
  }
  // (5) flatten + two maxout FCs
  {
    const long long tot = (long long)F * 2401 * 3;
    compact_units_kernel<T><<<(int)std::min<long long>((tot + 255) / 256, 8192), 256, 0, s>>>((const T*)(ws + g->hrows),
                                                                                        (T*)(ws + g->fcin), F, g->Kfc);
    RGP_HIP(hipGetLastError());
  }
  {
    IgemmParams p = make_params(g->fc1, ws + g->fcin, ws, F);
    EpiParams e = make_epi(g->fc1, ws + g->mo1, ws);
    e.bias = (const float*)(ws + g->b1i);
    if (save) e.argmax = (unsigned char*)(ws + g->mask1);
    if (g->drop_mask && g->drop_keep < 1.0f) { e.drop_mask = g->drop_mask; e.drop_inv_keep = 1.0f / g->drop_keep; }
    RGP_TRY((launch_igemm<T, 1, 1, EpiReluMaxout<T>>(p, e, s)));
  }
  {
    IgemmParams p = make_params(g->fc2, ws + g->mo1, ws, F);
    EpiParams e = make_epi(g->fc2, maps, ws);
    e.bias = (const float*)(ws + g->b2i);
    if (save) e.argmax = (unsigned char*)(ws + g->mask2);
    RGP_TRY((launch_igemm<T, 1, 1, EpiReluMaxout<float>>(p, e, s)));
  }
  return RGP_OK;
}

}  // namespace

extern "C" {

int rgp_cascade_set_dropout(rgp_cascade_t* g, float keep_prob, const unsigned char* mask) {
  RGP_REQUIRE(g, "rgp_cascade_set_dropout: null plan");
  RGP_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f, "rgp_cascade_set_dropout: keep_prob %g not in (0, 1]", (double)keep_prob);
  g->drop_mask = keep_prob < 1.f ? mask : nullptr;
  g->drop_keep = g->drop_mask ? keep_prob : 1.0f;
  return RGP_OK;
}

int rgp_cascade_create(rgp_cascade_t** plan, int batch, int n_steps, int image_hw, int dtype) {
  return rgp_cascade_create_ex(plan, batch, n_steps, image_hw, dtype, 0);
}

int rgp_cascade_create_ex(rgp_cascade_t** plan, int batch, int n_steps, int image_hw, int dtype, int save_for_backward) {
  RGP_REQUIRE(plan && batch > 0 && n_steps > 0, "rgp_cascade_create: bad arguments");
  RGP_REQUIRE(dtype == RGP_F32 || dtype == RGP_BF16, "rgp_cascade_create: dtype %d", dtype);
  rgp_cascade* g = new rgp_cascade();
  g->B = batch; g->T = n_steps; g->F = batch * n_steps; g->dtype = dtype; g->image_hw = image_hw;
  g->save = save_for_backward != 0;
  int rc = rgp_grcn_create(&g->bottom, batch, n_steps, 512, 256, dtype,
                           (g->save ? RGP_GRCN_SAVE_FOR_BACKWARD : 0) | RGP_GRCN_UNFOLDED_HEAD);   // (the bottom plan's own head is never run: no fold)
  if (rc == RGP_OK) rc = rgp_shallownet_create(&g->shallow, g->F, image_hw, dtype);
  if (rc != RGP_OK) { rgp_cascade_destroy(g); return rc; }
  const int es = esize(dtype), F = g->F;
  g->Kfc = (int)align_up(7203, 64);
  g->K2 = (int)align_up(2401, 64);
  bool ok = build_deconv_same_phases(g->up, 11, 7, 7, 1, 256, 49, 2, 2, kCt, 64, dtype);   // cascade.py:327-333
  ok &= conv5x5_desc(g->xtop, kCt, 3 * kSt, 3 * kSt, dtype);
  ok &= conv5x5_desc(g->zr, kSt, 2 * kSt, kSt, dtype);
  ok &= conv5x5_desc(g->c, kSt, kSt, kSt, dtype);
  auto fc = [&](ConvDesc& d, int K, long long ldc) {
    d.Mw = 1; d.N = 4802; d.in_img_stride = K; d.out_img_stride = ldc; d.in_tab = {0}; d.out_tab = {0};
    ok &= build_k_schedule(d, {0}, {0}, K, dtype);
  };
  fc(g->fc1, g->Kfc, g->K2);
  fc(g->fc2, g->K2, 2401);
  if (!ok) { rgp_cascade_destroy(g); return set_err(RGP_EINVAL, "rgp_cascade_create: K schedule failed"); }
  for (int y = 0; y < 49; ++y)
    for (int x = 0; x < 49; ++x) {
      g->tab_pad53_t.push_back(((y + 2) * kHp + x + 2) * kSt);
      g->tab_pad53_x.push_back(((y + 2) * kHp + x + 2) * kCt);
    }
  Arena a;
  g->off_bottom = a.take(rgp_grcn_workspace_bytes(g->bottom));
  g->off_shallow = a.take(rgp_shallownet_workspace_bytes(g->shallow));
  for (ConvDesc& d : g->up) d.reserve(a, dtype);
  g->up_p_off = a.take(g->up.size() * sizeof(IgemmParams));
  g->up_e_off = a.take(g->up.size() * sizeof(EpiParams));
  g->up_ps_off = a.take(g->up.size() * sizeof(IgemmParams));
  g->up_es_off = a.take(g->up.size() * sizeof(EpiParams));
  for (ConvDesc* d : {&g->xtop, &g->zr, &g->c, &g->fc1, &g->fc2}) d->reserve(a, dtype);
  g->o_pad53_t = a.take(g->tab_pad53_t.size() * 4);
  g->o_pad53_x = a.take(g->tab_pad53_x.size() * 4);
  g->sal = a.take((size_t)F * 2401 * 4);
  g->xtopbuf = a.take((size_t)F * kHp * kHp * kCt * es + 4096);
  g->xpre = a.take((size_t)F * 2401 * 3 * kSt * 4);
  g->hall = a.take((size_t)2 * batch * 2401 * kSt * 4);
  g->u = a.take((size_t)batch * 2401 * kSt * 4);
  g->hp = a.take((size_t)batch * kHp * kHp * kSt * es + 4096);
  g->rh = a.take((size_t)batch * kHp * kHp * kSt * es + 4096);
  g->hrows = a.take((size_t)F * 2401 * kSt * es);
  g->fcin = a.take((size_t)F * g->Kfc * es);
  g->mo1 = a.take((size_t)F * g->K2 * es);
  g->b1i = a.take(4802 * 4 + 64);
  g->b2i = a.take(4802 * 4 + 64);
  g->ones = a.take(kSt * 4);
  g->zeros = a.take(kSt * 4);
  g->bn_id = a.take(((size_t)2 * n_steps * 256 + 25 * 64 * 256) * 4);
  if (g->save) {
    rc = cascade_bwd_plan(g, a);
    if (rc != RGP_OK) { rgp_cascade_destroy(g); return rc; }
  }
  g->ws_bytes = a.off;
  *plan = g;
  return RGP_OK;
}

int rgp_cascade_destroy(rgp_cascade_t* g) {
  if (g) {
    if (g->bottom) rgp_grcn_destroy(g->bottom);
    if (g->shallow) rgp_shallownet_destroy(g->shallow);
    delete g;
  }
  return RGP_OK;
}

size_t rgp_cascade_workspace_bytes(const rgp_cascade_t* plan) { return plan ? plan->ws_bytes : 0; }

int rgp_cascade_bind_workspace(rgp_cascade_t* g, void* workspace, size_t bytes, rgp_stream_t stream) {
  RGP_REQUIRE(g && workspace, "rgp_cascade_bind_workspace: null argument");
  if (bytes < g->ws_bytes) return set_err(RGP_EWORKSPACE, "workspace %zu < required %zu bytes", bytes, g->ws_bytes);
  RGP_REQUIRE(((size_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  g->ws = (char*)workspace;
  g->weights_set = false;
  RGP_HIP(hipMemsetAsync(g->ws, 0, g->ws_bytes, s));
  RGP_TRY(rgp_grcn_bind_workspace(g->bottom, g->ws + g->off_bottom, rgp_grcn_workspace_bytes(g->bottom), stream));
  RGP_TRY(rgp_shallownet_bind_workspace(g->shallow, g->ws + g->off_shallow, rgp_shallownet_workspace_bytes(g->shallow), stream));
  for (ConvDesc& d : g->up) RGP_TRY(upload_desc(d, g->ws, s));
  g->up_p.clear(); g->up_e.clear();
  for (const ConvDesc& d : g->up) {
    g->up_p.push_back(make_params(d, g->bottom->ws + g->bottom->hbn.off, g->ws, g->F));
    g->up_e.push_back(make_epi(d, g->ws + g->xtopbuf, g->ws));
  }
  g->up_p_step = g->up_p; g->up_e_step = g->up_e;                       // the B frames of one step: image stride x T
  for (IgemmParams& p : g->up_p_step) { p.M = p.Mw * g->B; p.in_img_stride *= g->T; }
  for (EpiParams& e : g->up_e_step) e.out_img_stride *= g->T;
  const size_t np = g->up_p.size() * sizeof(IgemmParams), ne = g->up_e.size() * sizeof(EpiParams);
  RGP_HIP(hipMemcpyAsync(g->ws + g->up_p_off, g->up_p.data(), np, hipMemcpyHostToDevice, s));
  RGP_HIP(hipMemcpyAsync(g->ws + g->up_e_off, g->up_e.data(), ne, hipMemcpyHostToDevice, s));
  RGP_HIP(hipMemcpyAsync(g->ws + g->up_ps_off, g->up_p_step.data(), np, hipMemcpyHostToDevice, s));
  RGP_HIP(hipMemcpyAsync(g->ws + g->up_es_off, g->up_e_step.data(), ne, hipMemcpyHostToDevice, s));
  for (ConvDesc* d : {&g->xtop, &g->zr, &g->c, &g->fc1, &g->fc2}) RGP_TRY(upload_desc(*d, g->ws, s));
  RGP_HIP(hipMemcpyAsync(g->ws + g->o_pad53_t, g->tab_pad53_t.data(), g->tab_pad53_t.size() * 4, hipMemcpyHostToDevice, s));
  RGP_HIP(hipMemcpyAsync(g->ws + g->o_pad53_x, g->tab_pad53_x.data(), g->tab_pad53_x.size() * 4, hipMemcpyHostToDevice, s));
  fill2_kernel<<<1, 64, 0, s>>>((float*)(g->ws + g->ones), 1.0f, kSt);
  RGP_HIP(hipGetLastError());
  if (g->save) RGP_TRY(cascade_bwd_upload(g, s));
  return RGP_OK;
}

int rgp_cascade_set_weights(rgp_cascade_t* g, const rgp_cascade_weights* w, rgp_stream_t stream) {
  RGP_REQUIRE(g && w, "rgp_cascade_set_weights: null argument");
  if (!g->ws) return set_err(RGP_EWORKSPACE, "rgp_cascade: workspace not bound");
  const float* const* ptrs = (const float* const*)w;
  for (size_t i = 0; i < sizeof(rgp_cascade_weights) / sizeof(float*); ++i)
    RGP_REQUIRE(ptrs[i], "rgp_cascade_set_weights: weight pointer %zu is null", i);
  hipStream_t s = (hipStream_t)stream;
  RGP_TRY(g->dtype == RGP_BF16 ? set_weights_impl<bf16_t>(g, w, s) : set_weights_impl<float>(g, w, s));
  if (g->save) RGP_TRY(cascade_bwd_pack(g, w, s));
  return RGP_OK;
}

int rgp_cascade_forward(rgp_cascade_t* g, const float* frame_images, const float* c3d_input, float* gazemaps,
                        rgp_stream_t stream) {
  RGP_REQUIRE(g && frame_images && c3d_input && gazemaps, "rgp_cascade_forward: null argument");
  if (!g->ws) return set_err(RGP_EWORKSPACE, "rgp_cascade: workspace not bound");
  if (!g->weights_set) return set_err(RGP_ESTATE, "rgp_cascade: weights not set");
  hipStream_t s = (hipStream_t)stream;
  return g->dtype == RGP_BF16 ? forward_impl<bf16_t>(g, frame_images, c3d_input, gazemaps, s)
                              : forward_impl<float>(g, frame_images, c3d_input, gazemaps, s);
}

int rgp_cascade_read_buffer(rgp_cascade_t* g, const char* name, float* dst, rgp_stream_t stream) {
  RGP_REQUIRE(g && g->ws && name && dst, "rgp_cascade_read_buffer: null argument");
  hipStream_t s = (hipStream_t)stream;
  const std::string n(name);
  const int F = g->F;
  if (n == "rcn_outputs") return rgp_grcn_read_buffer(g->bottom, "rcn_outputs", dst, stream);
  if (n == "frm_sal") {
    RGP_HIP(hipMemcpyAsync(dst, g->ws + g->sal, (size_t)F * 2401 * 4, hipMemcpyDeviceToDevice, s));
    return RGP_OK;
  }
  size_t off, tab;
  int C;
  long long stride;
  if (n == "rcn_upsampled_outputs") { off = g->xtopbuf; tab = g->o_pad53_x; C = 64; stride = (long long)kHp * kHp * kCt; }
  else if (n == "gaze_rcn_outputs") { off = g->hrows; tab = g->c.out_tab_off; C = 3; stride = 2401LL * kSt; }
  else return set_err(RGP_EINVAL, "rgp_cascade_read_buffer: unknown buffer '%s'", name);
  const long long total = (long long)F * 2401 * C;
  const int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
  if (g->dtype == RGP_BF16)
    unpad_kernel<bf16_t><<<blocks, 256, 0, s>>>((const bf16_t*)(g->ws + off), dst, (const int*)(g->ws + tab), 2401, C, stride, total);
  else
    unpad_kernel<float><<<blocks, 256, 0, s>>>((const float*)(g->ws + off), dst, (const int*)(g->ws + tab), 2401, C, stride, total);
  RGP_HIP(hipGetLastError());
  return RGP_OK;
}

}  // extern "C"
