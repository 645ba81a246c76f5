// librgp_hip.so: the fully-connected GRU gaze model (BASELINE config 2).
// Reference graph: /root/reference/models/gaze_rnn.py:211-360 (GazePredictionGRU.
// create_gazeprediction_network): 1024->32 projection per pixel, flatten to 1568,
// tf rnn_cell.GRUCell(1617) over T steps, 1617 -> GH*GW output projection.
// TF-1.x GRUCell:  [r,u] = sigmoid([x,h] Wg + bg);  c = tanh([x, r*h] Wc + bc);
//                  h' = u*h + (1-u)*c.
//
// Nothing new on the device: every contraction is igemm_kernel in plain-GEMM form (one row
// per "image"), the x-parts of both kernels are hoisted over all T steps, and the gate math
// is the ConvGRU epilogue pair (EpiGruZR / EpiGruC) with the gate columns packed as [u | r]
// and an identity "batch-norm".  All K / N extents are zero-padded to multiples of 64.
#include <algorithm>

#include "rgp_host.h"

using namespace rgp;

struct rgp_fcgru {
  int B = 0, T = 0, F = 0, G = 0, dtype = RGP_F32;
  int Cp = 32, nx = 1568, n = 1617, Kx = 0, np = 0;
  ConvDesc proj, xg, zr, c, out;
  size_t o_zero1 = 0, o_lin = 0;
  size_t xt = 0, E = 0, xpre = 0, hall = 0, u = 0, hp = 0, rh = 0, hrows = 0, xbias = 0, ones = 0, zeros = 0;
  size_t ws_bytes = 0;
  char* ws = nullptr;
  bool weights_set = false;
  const float *proj_b = nullptr, *out_b = nullptr;
};

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
  for (ConvDesc* d : {&g->proj, &g->xg, &g->zr, &g->c, &g->out}) RGP_HIP(hipMemsetAsync(ws + d->w_off, 0, d->w_bytes(g->dtype), s));
  // projection [1024, 32]
  g->proj.s_tap = 0; g->proj.s_n = 1; g->proj.s_c = g->Cp;
  RGP_TRY(pack_filter<T>(g->proj, w->proj_c3d_W, ws, g->Cp, 0, s));
  // gate kernel [nx+n, 2n] columns [r | u]; candidate kernel [nx+n, n].  Packed rows: [u | r | c].
  auto pk = [&](ConvDesc& d, const float* src, long long ld, int k_rows, int row0) -> int {
    d.s_tap = 0; d.s_n = 1; d.s_c = ld; d.cin_src = k_rows;
    return pack_filter<T>(d, src, ws, n, row0, s);
  };
  RGP_TRY(pk(g->xg, w->gates_kernel + n, 2LL * n, nx, 0));                 // u, x-part
  RGP_TRY(pk(g->xg, w->gates_kernel, 2LL * n, nx, np));                    // r, x-part
  RGP_TRY(pk(g->xg, w->candidate_kernel, n, nx, 2 * np));                  // c, x-part
  RGP_TRY(pk(g->zr, w->gates_kernel + (long long)nx * 2 * n + n, 2LL * n, n, 0));   // u, h-part
  RGP_TRY(pk(g->zr, w->gates_kernel + (long long)nx * 2 * n, 2LL * n, n, np));      // r, h-part
  RGP_TRY(pk(g->c, w->candidate_kernel + (long long)nx * n, n, n, 0));              // c, (r*h)-part
  g->out.s_tap = 0; g->out.s_n = 1; g->out.s_c = g->G; g->out.cin_src = n;
  RGP_TRY(pack_filter<T>(g->out, w->proj_out_W, ws, g->G, 0, s));
  // bias of the hoisted x-GEMM: [bu | br | bc] padded
  float* xb = (float*)(ws + g->xbias);
  RGP_HIP(hipMemsetAsync(xb, 0, (size_t)3 * np * 4, s));
  RGP_HIP(hipMemcpyAsync(xb, w->gates_bias + n, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
  RGP_HIP(hipMemcpyAsync(xb + np, w->gates_bias, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
  RGP_HIP(hipMemcpyAsync(xb + 2 * np, w->candidate_bias, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
  g->proj_b = w->proj_c3d_b;
  g->out_b = w->proj_out_b;
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
  nchw_to_rows_kernel<T><<<dim3(1024 / 64, F), 256, 0, s>>>(c3d_input, (T*)(ws + g->xt), 1024);
  RGP_HIP(hipGetLastError());
  {  // per-pixel projection, rows of the x-GEMM (gaze_rnn.py:294-308, flatten :340-341)
    IgemmParams p = make_params(g->proj, ws + g->xt, ws, F);
    EpiParams e = make_epi(g->proj, ws + g->E, ws);
    e.bias = g->proj_b;
    RGP_TRY((launch_igemm<T, 1, 1, EpiStore<T, true, false>>(p, e, s)));
  }
  {  // hoisted x-parts of both GRU kernels, biases folded in
    IgemmParams p = make_params(g->xg, ws + g->E, ws, F);
    EpiParams e = make_epi(g->xg, ws + g->xpre, ws);
    e.bias = (const float*)(ws + g->xbias);
    RGP_TRY((launch_igemm<T, 1, 1, EpiStore<float, true, false>>(p, e, s)));
  }
  const size_t st = (size_t)B * np;
  RGP_HIP(hipMemsetAsync(ws + g->hp, 0, st * sizeof(T), s));
  RGP_HIP(hipMemsetAsync(ws + g->hall, 0, st * 4, s));
  float* hall = (float*)(ws + g->hall);
  for (int t = 0; t < T_; ++t) {
    EpiParams e = make_epi(g->zr, ws + g->rh, ws);
    e.xpre = (const float*)(ws + g->xpre) + (size_t)t * 3 * np;
    e.xpre_img_stride = (long long)T_ * 3 * np;
    e.xpre_ld = 3 * np;
    e.xpre_col = 0;
    e.S = np;
    e.state_rows = 1;
    e.h_prev = hall + (size_t)(t & 1) * st;
    e.h_next = hall + (size_t)((t + 1) & 1) * st;
    e.u_gate = (float*)(ws + g->u);
    IgemmParams p = make_params(g->zr, ws + g->hp, ws, B);
    RGP_TRY((launch_igemm<T, 1, 1, EpiGruZR<T>>(p, e, s)));
    e.out = ws + g->hp;
    e.out_tab = (const int*)(ws + g->c.out_tab_off);
    e.out_img_stride = g->c.out_img_stride;
    e.xpre_col = 2 * np;
    e.out2 = ws + g->hrows;
    e.out2_tab = (const int*)(ws + g->c.out_tab_off);
    e.out2_img_stride = np;
    e.out2_img_mul = T_;
    e.out2_img_add = t;
    e.bn_gamma = (const float*)(ws + g->ones);
    e.bn_beta = (const float*)(ws + g->zeros);
    e.bn_inv_std = 1.0f;
    IgemmParams pc = make_params(g->c, ws + g->rh, ws, B);
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

}  // namespace

extern "C" {

int rgp_fcgru_create(rgp_fcgru_t** plan, int batch, int n_steps, int gazemap_h, int gazemap_w, int dtype) {
  RGP_REQUIRE(plan && batch > 0 && n_steps > 0, "rgp_fcgru_create: bad arguments");
  RGP_REQUIRE((gazemap_h == 49 && gazemap_w == 49) || (gazemap_h == 7 && gazemap_w == 7),
              "rgp_fcgru_create: gaze map %dx%d (reference uses 49x49 or 7x7)", gazemap_h, gazemap_w);
  RGP_REQUIRE(dtype == RGP_F32 || dtype == RGP_BF16, "rgp_fcgru_create: dtype %d", dtype);
  rgp_fcgru* g = new rgp_fcgru();
  g->B = batch; g->T = n_steps; g->F = batch * n_steps; g->G = gazemap_h * gazemap_w; g->dtype = dtype;
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

}  // extern "C"
