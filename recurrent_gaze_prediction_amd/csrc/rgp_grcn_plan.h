// Plan object of the gaze_grcn path, shared by the forward (rgp_grcn.hip) and backward
// (rgp_grcn_bwd.hip) translation units.
#pragma once
#include "rgp_host.h"

struct Buf {
  size_t off = 0, bytes = 0;
};

namespace rgp {
// host side: add() regions, flush() = one launch (or a plain memset for a single region)
struct ZeroBatch {
  ZeroTable t;
  hipStream_t s;
  explicit ZeroBatch(hipStream_t stream) : s(stream) { t.n = 0; t.first[0] = 0; }
  int add(void* p, size_t bytes) {
    if (bytes == 0) return RGP_OK;
    if ((bytes & 3) || (((size_t)p) & 3)) return set_err(RGP_EINVAL, "ZeroBatch: region not 4-byte aligned");
    if (t.n == ZERO_MAX_REGIONS) RGP_TRY(flush());
    t.ptr[t.n] = p; t.bytes[t.n] = bytes;
    t.first[t.n + 1] = t.first[t.n] + (int)((bytes + ZERO_BLOCK_BYTES - 1) / ZERO_BLOCK_BYTES);
    ++t.n;
    return RGP_OK;
  }
  int flush() {
    if (t.n == 1) RGP_HIP(hipMemsetAsync(t.ptr[0], 0, t.bytes[0], s));
    else if (t.n > 1) {
      zero_regions_kernel<<<t.first[t.n], 256, 0, s>>>(t);
      RGP_HIP(hipGetLastError());
    }
    t.n = 0; t.first[0] = 0;
    return RGP_OK;
  }
};
}  // namespace rgp

struct rgp_grcn {
  // (owner: a cascade plan) step_ev[t] is recorded on the launch stream behind step t of the per-timestep recurrence: lets
  // another stream consume state t while the later steps run (rgp_cascade.hip).  Null = nothing recorded
  hipEvent_t* step_ev = nullptr;
  // (same owner) backward from external state gradients with per-timestep BPTT: the launch stream waits for bwd_step_ev[t]
  // before step t reads frame (b, t) of the gradient -- its producer runs one step ahead on another stream.  Null = the whole
  // gradient is complete when the call is made
  hipEvent_t* bwd_step_ev = nullptr;
  int B = 0, T = 0, P = 0, S = 0, dtype = RGP_BF16, save = 0, F = 0;
  rgp::ConvDesc proj, proj_rows, xconv, gzr, gc, d3;
  rgp::ConvDesc d3t;   // the folded 7x7 conv as a row-Toeplitz GEMM: 16 output pixels of a row per GEMM row (d3: x = 48 only)
  std::vector<rgp::ConvDesc> d1, d2;            // transposed convolutions: one problem per row phase py, N = (px, channel)
  std::vector<rgp::ConvDesc> d1_pack, d2_pack;  // their filter-packing aliases (one per (py, px): a tap table of its own)
  // read_buffer tables (host copies + offsets)
  std::vector<int> tab_pad9_P, tab_pad9_S, tab_pad27, tab_pad55, tab_lin49_3S, tab_lin49_S;
  size_t o_pad9_P = 0, o_pad9_S = 0, o_pad27 = 0, o_pad55 = 0, o_lin49_3S = 0, o_lin49_S = 0;
  Buf xt, E, xpre, hall, uall, rall, call, hp, rhp, hbn, D1, D2, gfold, frame_loss, gtoep, bias16;
  // unless RGP_GRCN_UNFOLDED_HEAD: the three transposed convolutions + out_W folded into ONE 19x19 stride-6 transposed
  // convolution on BN(h), run as GEMM + col2im (head_fold.hip.h), forward and backward
  bool fold_head = false;
  rgp::ConvDesc hfold;
  Buf hf_part;                     // the five partial sums of K (summed in a fixed order)
  Buf hf_h, hf_k, hf_z;            // H [11,11,64], K [361][S] fp32; Z [F*49][384] fp32 (the GEMM's output, gathered by col2im)
  Buf xch_h, xch_rh, seq_cnt;   // persistent ConvGRU sequence kernel: exchange images [groups][98][128] + phase counters
  int seq_nc = 0, seq_groups = 0;   // clips per group / groups (0 = the per-step path)
  unsigned* err_host = nullptr;     // pinned, device-visible error word: a persistent launch that timed out sets it
  int fault = 0;                    // rgp_grcn_inject_fault: bit 0 next sequence launch, bit 1 next BPTT launch lose a member
  size_t ws_bytes = 0;
  char* ws = nullptr;
  bool weights_set = false;
  const float *bn_gamma = nullptr, *bn_beta = nullptr, *proj_b = nullptr, *out_b = nullptr;
  rgp::StageProfiler prof;
  struct GrcnBwd* bwd = nullptr;   // backward plan (save_for_backward only), rgp_grcn_bwd.hip
};


inline Buf take(rgp::Arena& a, size_t bytes) {
  Buf b;
  b.bytes = bytes;
  b.off = a.take(bytes);
  return b;
}

// rgp_grcn.hip: the persistent ConvGRU kernels apply to this plan on the current device
bool seq_persistent_ok(const rgp_grcn* g);
// ... and its persistent BPTT launch leaves RGP_RCCL_CU_RESERVE CUs free (the TOP gradient group may leave before it)
bool grads_top_early(const rgp_grcn* g);
// rgp_grcn_bwd.hip
// returns RGP_ETIMEOUT (and clears the word) if a persistent launch of this plan reported a lost group member
int grcn_check_error(rgp_grcn* g);
int grcn_bwd_plan(rgp_grcn* g, rgp::Arena& a);
int grcn_bwd_upload(rgp_grcn* g, hipStream_t s);
int grcn_bwd_pack(rgp_grcn* g, const rgp_grcn_weights* w, hipStream_t s, hipStream_t sc);   // sc: the stream the head's fold ran on
int grcn_bwd_fork_fold(rgp_grcn* g, hipStream_t s, hipStream_t* sc);
int grcn_bwd_join_fold(rgp_grcn* g, hipStream_t s);
void grcn_bwd_destroy(rgp_grcn* g);
