// Plan object of the two-level cascade, shared by rgp_cascade.hip (forward) and rgp_cascade_bwd.hip.
#pragma once
#include <vector>

#include "rgp_grcn_plan.h"

constexpr int kCt = 128;   // top-cell input channels: 64 upsampled + 1 saliency, zero-padded
constexpr int kSt = 16;    // top-cell state channels: 3 units, zero-padded
constexpr int kHp = 53;    // 49 + 2*2 halo for the 5x5 SAME convs
constexpr int kN2 = 4864;  // 4802 FC outputs padded to a multiple of 128 (gradient rows)

struct rgp_cascade {
  int B = 0, T = 0, F = 0, dtype = RGP_BF16, image_hw = 98;
  rgp_grcn* bottom = nullptr;
  rgp_shallownet_t* shallow = nullptr;
  std::vector<rgp::ConvDesc> up;             // 49 phases of the stride-7 transposed conv
  // ... run as ONE grouped launch: their kernel parameters, built at bind time (host copies outlive the upload), and the
  // device arrays igemm_grouped_kernel reads
  std::vector<rgp::IgemmParams> up_p, up_p_step;       // all frames at once / the B frames of one time step (image stride x T)
  std::vector<rgp::EpiParams> up_e, up_e_step;
  size_t up_p_off = 0, up_e_off = 0, up_ps_off = 0, up_es_off = 0;
  rgp::ConvDesc xtop, zr, c, fc1, fc2;
  std::vector<int> tab_pad53_t, tab_pad53_x;   // interior of a 53x53xkSt / 53x53xkCt image
  size_t o_pad53_t = 0, o_pad53_x = 0;
  size_t off_bottom = 0, off_shallow = 0, sal = 0, xtopbuf = 0, xpre = 0, hall = 0, u = 0, hp = 0, rh = 0, hrows = 0,
         fcin = 0, mo1 = 0, b1i = 0, b2i = 0, ones = 0, zeros = 0, bn_id = 0;
  int Kfc = 0, K2 = 0;
  size_t ws_bytes = 0;
  char* ws = nullptr;
  bool weights_set = false;
  // training-time dropout on fc1 (gaze_grcn_cascade.py:401-402): caller-owned keep bytes [F][4802], null = off
  const unsigned char* drop_mask = nullptr;
  float drop_keep = 1.0f;

  // ---- training (save_for_backward) ----
  bool save = false;
  // forward state kept for BPTT of the top cell: fp32 [T(+1)][B][2401][kSt]; operand images of every step
  size_t hall_t = 0, uall = 0, rall = 0, call = 0;
  size_t hp_all = 0;         // T  [B][T+1][53*53][kSt]   h_{t-1} of step t in slot (b, t); slot (b, 0) stays zero
  size_t rhp_all = 0;        // T  [B][T][53*53][kSt]     r (.) h_{t-1}
  size_t mask1 = 0, mask2 = 0;   // bytes [F][2401]: winning half of each maxout unit (0 = ReLU-gated)
  // backward operands
  rgp::ConvDesc b_fc2, b_fc1, b_tc, b_tzr, b_tx, b_up;   // input-gradient GEMMs / convs
  std::vector<int> tab_pad53_64;                         // interior of a 53x53x64 image
  size_t o_pad53_64 = 0;
  size_t dz2 = 0, dz1 = 0;   // T  [F+1][kN2], row 0 zero: gradient w.r.t. the FC pre-activations (natural column order)
  size_t dmo1 = 0, dfcin = 0;   // fp32 [F][K2] / [F][Kfc]
  size_t dh_carry = 0, drh = 0; // fp32 [B][2401][kSt]
  size_t dcp_pad = 0;        // T  [B][53*53][kSt]
  size_t dzr_pad = 0;        // T  [B][53*53][2*kSt]
  size_t dxpre_pad = 0;      // T  [F][53*53][64]: columns [dz_pre | dr_pre | dc_pre | 0]
  size_t dup_pad = 0;        // T  [F][53*53][64]: gradient w.r.t. the upsampled maps
  size_t d_hbn = 0;          // fp32 [F][49][256]: gradient w.r.t. the bottom states
  size_t dwx = 0, dwh = 0, dwu = 0;   // fp32 scratch of the top cell's filter gradients in packed K order
  size_t scratch_head = 0;   // fp32: dummy head / batch-norm gradients of the bottom sub-plan

  // A stream of the plan's own for work the main chain does not wait for.  Forward: the frame saliency (ShallowNet, its own
  // inputs) beside the projection and the bottom level's 35 per-step launches.  Backward: every weight gradient (the two FC
  // layers, the top cell's three, the stride-7 filter) beside the data-gradient chain, whose BPTT loops are per-step launches
  // on a fraction of the CUs.  ev[i]: "the operands of side task i are complete" on the caller's stream; ev_join: the side
  // stream has drained.  Also recorded into a stream capture when the side stream exists already.
  hipStream_t side = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr}, ev_join = nullptr;
  // Forward as three chains that advance together, one time step apart (rgp_cascade.hip forward_impl): the bottom cell's T steps
  // on the caller's stream (ev_b[t] behind step t), upsampling + saliency + the top cell's input convolution of step t on `side`
  // (ev_x[t]), the top cell's step t on `side2`.  All three are per-step launches on a fraction of the CUs.
  // Backward, the same way in reverse: the top cell's BPTT on `side2` (ev_b[t] behind step t), step t's input gradient through
  // the input convolution and the stride-7 transposed convolution on `side3` (ev_x[t]), the bottom cell's BPTT on the caller's
  // stream (waits for ev_x[t] before its step t: rgp_grcn::bwd_step_ev); `side` keeps the weight gradients.
  hipStream_t side2 = nullptr, side3 = nullptr;
  std::vector<hipEvent_t> ev_b, ev_x;
  hipEvent_t ev_join2 = nullptr;
  // the side stream (made on first use) behind everything queued on s, or s itself when a capture of s finds none yet
  int fork(hipStream_t s, int i, hipStream_t* sc) {
    using namespace rgp;
    *sc = s;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = !(hipStreamIsCapturing(s, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone);
    if (!rgp::dev_knob("RGP_CASCADE_FORK", 1)) return RGP_OK;
    RGP_TRY(pool_stream(0, !capturing, &side));                 // the device's pool (rgp_core.hip): shared, not owned
    if (!side) return RGP_OK;
    if (!ev_join) {
      for (hipEvent_t* e : {&ev[0], &ev[1], &ev[2], &ev[3], &ev_join}) RGP_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    RGP_HIP(hipEventRecord(ev[i], s));
    RGP_HIP(hipStreamWaitEvent(side, ev[i], 0));
    *sc = side;
    return RGP_OK;
  }
  // second side stream + per-step events (made on first use; not inside a stream capture)
  bool pipe_ok(hipStream_t s, int n_steps) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (!(hipStreamIsCapturing(s, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone)) return false;
    if (!side2 || !side3 || !ev_join2) {
      if (rgp::pool_stream(1, true, &side2) != RGP_OK || rgp::pool_stream(2, true, &side3) != RGP_OK || !side2 || !side3) return false;
      bool ok = hipEventCreateWithFlags(&ev_join2, hipEventDisableTiming) == hipSuccess;
      for (int i = 0; i < 2 * n_steps && ok; ++i) {
        hipEvent_t e = nullptr;
        ok = hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        if (ok) (i < n_steps ? ev_b : ev_x).push_back(e);
      }
      if (!ok) return false;
    }
    return side3 && (int)ev_b.size() == n_steps && (int)ev_x.size() == n_steps;
  }
  int join(hipStream_t s) {
    using namespace rgp;
    if (!side || !ev_join) return RGP_OK;
    RGP_HIP(hipEventRecord(ev_join, side));
    RGP_HIP(hipStreamWaitEvent(s, ev_join, 0));
    return RGP_OK;
  }
  ~rgp_cascade() {
    if (ev_join) for (hipEvent_t e : {ev[0], ev[1], ev[2], ev[3], ev_join}) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev_b) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev_x) (void)hipEventDestroy(e);
    if (ev_join2) (void)hipEventDestroy(ev_join2);
  }
};

// rgp_cascade_bwd.hip
int cascade_bwd_plan(rgp_cascade* g, rgp::Arena& a);
int cascade_bwd_upload(rgp_cascade* g, hipStream_t s);
int cascade_bwd_pack(rgp_cascade* g, const rgp_cascade_weights* w, hipStream_t s);
