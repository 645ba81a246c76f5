/* librgp_hip.so -- C ABI of the MI355X-native recurrent gaze-prediction path.
 *
 * The reference (yj-yu/Recurrent_Gaze_Prediction) has no FFI/plugin boundary: its
 * seam is the Python graph builder `create_gazeprediction_network(frame_images,
 * c3d_input, dropout_keep_prob, net)` (models/gaze_grcn.py:173-188) executed by a
 * single `session.run(feed_dict)` (models/gaze_rnn.py:523-531, 603-611).  This
 * header is the boundary a maintainer would bind in its place: each entry point
 * cites the reference lines whose computation it replaces.
 *
 * Conventions
 *  - every function returns 0 (RGP_OK) or a negative RGP_E* code; the message is in
 *    rgp_last_error() (thread-local);
 *  - every tensor argument is a caller-owned DEVICE pointer (fp32 unless stated);
 *    the library never allocates device memory: the caller provides one workspace
 *    of rgp_*_workspace_bytes() bytes per plan;
 *  - every call is asynchronous on the given HIP stream (a hipStream_t passed as
 *    void*); a plan may be used from one stream at a time.  Training plans (and the
 *    cascade's forward) run independent parts of a call -- weight gradients beside the
 *    data-gradient chain, the cascade's two cells one time step apart -- on streams of
 *    the library's own (three per device, shared by every plan of the process), forked
 *    behind `stream` and joined into it before the call returns: to the caller the call
 *    is ordered on `stream` as if it ran there alone.  The same holds inside a stream
 *    capture (the forks become branches of the graph once the library's streams exist,
 *    i.e. after one eager call; before that the call is serial).  Because those streams
 *    are shared, they join a capture for its duration: while one host thread captures
 *    calls of this library, no other thread may issue training calls on that device;
 *  - layouts are the reference's: c3d_input [B,T,1024,7,7], maps [B,T,49,49],
 *    conv filters HWIO / DHWIO, transposed-conv filters [kh,kw,out,in];
 *  - dtype selects the MFMA operand type of the contractions (RGP_BF16:
 *    v_mfma_f32_16x16x32_bf16, RGP_F32: v_mfma_f32_16x16x4_f32); accumulation,
 *    gate math, recurrent state, logits and losses are always fp32.
 */
#ifndef RGP_H_
#define RGP_H_
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RGP_OK 0
#define RGP_EINVAL (-1)   /* bad argument / unsupported shape */
#define RGP_EHIP (-2)     /* a HIP runtime call failed */
#define RGP_EWORKSPACE (-3) /* workspace missing or too small */
#define RGP_ESTATE (-4)   /* call order violated (e.g. forward before set_weights) */
#define RGP_ETIMEOUT (-5) /* a persistent ConvGRU launch lost a group member: its outputs are NaN-poisoned (rgp_grcn_status) */

#define RGP_F32 0
#define RGP_BF16 1

typedef void* rgp_stream_t; /* hipStream_t */

const char* rgp_last_error(void);
int rgp_version(void);
/* Writes the gcnArchName of the current device ("gfx950...") into buf. */
int rgp_device_arch(char* buf, int buflen);

/* ------------------------------------------------------------------ gaze_grcn */
typedef struct rgp_grcn rgp_grcn_t;

/* fp32 device pointers, reference variable names:
 * proj_c3d_W [1024,P] proj_c3d_b [P]                    gaze_grcn.py:234-237
 * gru_W{z,r,}  [3,3,P,S]   gru_U{z,r,} [3,3,S,S]        gaze_grcn.py:64-81
 * bn_gamma, bn_beta [T,S] (one BN layer per timestep)   gaze_grcn.py:325
 * up_weight1 [5,5,64,S] up_weight2 [5,5,32,64] up_weight3 [7,7,12,32]   gaze_grcn.py:292-310
 * out_W [12,1] out_b [1]                                gaze_grcn.py:311-314 */
typedef struct rgp_grcn_weights {
  const float *proj_c3d_W, *proj_c3d_b;
  const float *gru_Wz, *gru_Uz, *gru_Wr, *gru_Ur, *gru_W, *gru_U;
  const float *bn_gamma, *bn_beta;
  const float *up_weight1, *up_weight2, *up_weight3;
  const float *out_W, *out_b;
} rgp_grcn_weights;

/* Plan for GazePredictionGRCN.create_gazeprediction_network (gaze_grcn.py:173-376)
 * at fixed batch B, timesteps T, dim_cnn_proj P (512), rnn_state_size S (128). */
int rgp_grcn_create(rgp_grcn_t** plan, int batch, int n_steps, int dim_proj, int dim_state, int dtype,
                    int flags);
/* flags (0 / 1 keep the meaning of the former `save_for_backward` argument):
 *  RGP_GRCN_SAVE_FOR_BACKWARD  training plan (gates, states and operand images kept for rgp_grcn_backward);
 *  RGP_GRCN_PER_STEP           run the ConvGRU recurrence and its BPTT as per-timestep launches even where the
 *                              persistent kernels apply (bf16, 128 state channels, <= 64 clips): the library's second
 *                              implementation of the recurrence, always used by f32 plans;
 *  RGP_GRCN_UNFOLDED_HEAD      by default a plan runs the saliency head (gaze_grcn.py:326-361) folded: the three transposed
 *                              convolutions and out_W have no bias or non-linearity between them and are combined, exactly,
 *                              into one 19x19 stride-6 transposed convolution on BN(h_t) when the weights are set (GEMM +
 *                              col2im forward; the backward is the chain rule through the fold and returns the gradients of
 *                              weight1 / weight2 / weight3 / out_W themselves).  With this flag the plan runs the three
 *                              stages one by one, forward and backward: the library's second implementation of the head;
 *                              the buffers "d1" / "d2" of rgp_grcn_read_buffer exist only then. */
#define RGP_GRCN_SAVE_FOR_BACKWARD 1
#define RGP_GRCN_PER_STEP 2
#define RGP_GRCN_UNFOLDED_HEAD 4
/* The persistent ConvGRU kernels (one launch for all T steps, forward and BPTT) need all their workgroups resident
 * together: keep ONE of them in flight per device.  Launches issued through this library from one process are
 * serialised against each other automatically (any stream, any plan, any host thread: the wait for the previous
 * launch, the launch and its record are one critical section); a launch that nevertheless loses a group member
 * -- another process running the same kernels on the device -- gives up after about a second, NaN-poisons everything
 * computed from it (logits, maps, states, gradients) and raises the plan's error state: the next call on the plan
 * returns RGP_ETIMEOUT, and so does rgp_grcn_status, which first waits for `stream`.  The state is cleared by being
 * reported. */
int rgp_grcn_status(rgp_grcn_t* plan, rgp_stream_t stream);
/* Test hook: the next persistent sequence (kind 1) / BPTT (kind 2) launch of the plan runs without one member of its
 * first group, i.e. exercises the time-out path above (about one second).  RGP_ESTATE if the plan uses per-step launches. */
#define RGP_FAULT_SEQ_LOST_MEMBER 1
#define RGP_FAULT_BPTT_LOST_MEMBER 2
int rgp_grcn_inject_fault(rgp_grcn_t* plan, int kind);
int rgp_grcn_destroy(rgp_grcn_t* plan);
size_t rgp_grcn_workspace_bytes(const rgp_grcn_t* plan);
/* Uploads the offset tables and zeroes the halos.  Once per workspace. */
int rgp_grcn_bind_workspace(rgp_grcn_t* plan, void* workspace, size_t bytes, rgp_stream_t stream);
/* Packs the fp32 weights into MFMA operand form ([N][K], operand dtype). Call after
 * every weight update. */
int rgp_grcn_set_weights(rgp_grcn_t* plan, const rgp_grcn_weights* w, rgp_stream_t stream);

/* Whole graph: c3d_input [B,T,1024,7,7] -> logits [B,T,49,49]
 * (gaze_grcn.py:173-376) and, if probs != NULL, the per-frame softmax that
 * build_model applies for loss_type xentropy (gaze_rnn.py:149-159). */
int rgp_grcn_forward(rgp_grcn_t* plan, const float* c3d_input, float* logits, float* probs, rgp_stream_t stream);
/* Same graph fed by C3D conv5b rows produced by rgp_c3d_forward (operand dtype,
 * [B*T*49][1024] with K order d*512+c), skipping the transpose of gaze_grcn.py:225-227. */
int rgp_grcn_forward_rows(rgp_grcn_t* plan, const void* c3d_rows, float* logits, float* probs, rgp_stream_t stream);

/* Stages of the same graph on the plan's workspace (for tests / profiling):
 * rgp_proj_fwd          transpose + xw_plus_b                  gaze_grcn.py:225-254
 * rgp_convgru_xconv_fwd W_z,W_r,W convs of all T steps at once gaze_grcn.py:108-109,112-113,122-123
 * rgp_convgru_seq_fwd   T x { U convs, gates, blend } + BN     gaze_grcn.py:110-127,259-288,325
 * rgp_head_fwd          3 transposed convs + 12->1             gaze_grcn.py:326-366 */
int rgp_proj_fwd(rgp_grcn_t* plan, const float* c3d_input, rgp_stream_t stream);
int rgp_convgru_xconv_fwd(rgp_grcn_t* plan, rgp_stream_t stream);
int rgp_convgru_seq_fwd(rgp_grcn_t* plan, rgp_stream_t stream);
int rgp_head_fwd(rgp_grcn_t* plan, float* logits, rgp_stream_t stream);

/* Copies an intermediate, un-padded and widened to fp32, into dst:
 * "c3d_embedded" [B,T,7,7,P]   "xpre" [B,T,7,7,3S] (z|r|c pre-activations of W*x)
 * "rcn_outputs" [B,T,7,7,S] (h_t)   "bn" [B,T,7,7,S]   "d1" [B*T,23,23,64]   "d2" [B*T,49,49,32]
 * "u" / "r" / "c" [T,B,7,7,S] (r,c only with save_for_backward). */
int rgp_grcn_read_buffer(rgp_grcn_t* plan, const char* name, float* dst, rgp_stream_t stream);
/* Number of fp32 elements rgp_grcn_read_buffer writes for name (0 if unknown). */
size_t rgp_grcn_buffer_elems(const rgp_grcn_t* plan, const char* name);

/* Per-frame softmax (model_util.py:61-64) and cross entropy with summed/averaged
 * loss (model_util.py:66-72, gaze_rnn.py:390-407): logits, labels [frames, npix];
 * probs, frame_loss [frames], loss [1] may each be NULL.  loss = sum(frame_loss)/frames. */
int rgp_softmax_xent_fwd(const float* logits, const float* labels, float* probs, float* frame_loss, float* loss,
                         int frames, int npix, rgp_stream_t stream);

/* Backward of the same graph under the loss of gaze_rnn.py:363-408 (what tf.gradients builds,
 * base.py:278-281).  Needs a plan created with save_for_backward=1 and a preceding
 * rgp_grcn_forward on the same inputs (its saved states live in the workspace).
 * logits / probs: outputs of that forward; labels: per-frame-normalised ground-truth maps
 * [B,T,49,49] (normalize_probability_map, model_util.py:40-58).  loss_type 0 = xentropy,
 * 1 = l2.  grads: fp32 device arrays shaped like the weights; fully overwritten. */
int rgp_grcn_backward(rgp_grcn_t* plan, const float* logits, const float* probs, const float* labels,
                      const rgp_grcn_weights* grads, int loss_type, rgp_stream_t stream);

/* Backward of the projection + ConvGRU only, for a network that stacks its own layers on the states
 * (the cascade, gaze_grcn_cascade.py:289-336): d_states [B*T*49, S] fp32 is the gradient w.r.t. the
 * batch-normalised states in frame order b*T+t; the head fields of grads receive zeros. */
int rgp_grcn_backward_from_states(rgp_grcn_t* plan, const float* d_states, const rgp_grcn_weights* grads,
                                  rgp_stream_t stream);

/* Data-parallel training: rgp_grcn_backward / _from_states record an event on their stream once a group
 * of gradients is final, in this order -- RGP_GRCN_GRADS_TOP: bn_gamma, bn_beta, up_weight1..3, out_W, out_b (before
 * the BPTT starts); RGP_GRCN_GRADS_GRU: the six ConvGRU filters; RGP_GRCN_GRADS_PROJ: proj_c3d_W / _b (the end of the
 * backward).  rgp_grcn_wait_grads makes `waiting_stream` wait for that event, so the host can issue the all-reduce
 * of the group's slice there while the rest of the backward is still running.  RGP_ESTATE before the first backward,
 * and after a backward that was captured into a HIP graph (nothing is recorded during capture; a replay has no events). */
#define RGP_GRCN_GRADS_TOP 0
#define RGP_GRCN_GRADS_GRU 1
#define RGP_GRCN_GRADS_PROJ 2
int rgp_grcn_wait_grads(rgp_grcn_t* plan, int group, rgp_stream_t waiting_stream);
/* Co-residency rule for that overlap.  The persistent BPTT launch occupies one CU per workgroup (8 per group of 1 - 2
 * clips; 157 KB of LDS each, so nothing else fits beside one) and cannot make progress until ALL of them are resident.
 * A collective started before it may hold CUs for as long as the slowest peer rank takes.  Therefore the TOP group's
 * event is recorded ahead of the BPTT launch only when that launch needs at most (CUs of the device -
 * RGP_RCCL_CU_RESERVE) workgroups (on a 256-CU MI355X: up to 24 clips per GPU, e.g. BASELINE config 4's 8); larger
 * per-GPU batches (config 3's 64 clips = 256 workgroups) record it BEHIND the launch, so no collective of this step can
 * run next to it.  Per-step plans (RGP_GRCN_PER_STEP, f32) always release it early.  rgp_grcn_grads_top_early reports
 * which of the two the plan does on the current device: 1 = before the BPTT, 0 = behind it. */
#define RGP_RCCL_CU_RESERVE 64
int rgp_grcn_grads_top_early(const rgp_grcn_t* plan);
/* Workgroups (= CUs) a persistent ConvGRU / BPTT launch of this plan occupies on the current device; 0 if the plan runs
 * the recurrence as per-timestep launches (RGP_GRCN_PER_STEP, f32, other widths, more clips than the device has CUs for). */
int rgp_grcn_persistent_workgroups(const rgp_grcn_t* plan);

/* After rgp_grcn_backward: the gradient w.r.t. the network input, d_rows [B*T*49, 1024] fp32 in the column
 * order of the conv5b rows (d*512 + c) -- what rgp_c3d_backward takes when the conv stack is fine-tuned
 * end to end (BASELINE config 5). */
int rgp_grcn_backward_input(rgp_grcn_t* plan, float* d_rows, rgp_stream_t stream);

/* tf.clip_by_global_norm + tf.train.AdamOptimizer.apply_gradients on one flat fp32 parameter
 * buffer (base.py:286-297; TF form: lr_t = lr*sqrt(1-b2^t)/(1-b1^t), theta -= lr_t*m/(sqrt(v)+eps),
 * t = step+1).  workspace: 256 floats.  grad_norm_out (optional, device): the global norm.
 * max_grad_norm <= 0 disables clipping. */
int rgp_adam_clip_step(float* params, const float* grads, float* m, float* v, long long n, float* workspace,
                       int step, float lr, float beta1, float beta2, float eps, float max_grad_norm,
                       float* grad_norm_out, rgp_stream_t stream);

/* The same step when the variables live in several flat buffers (conv stack + head, config 5): the clip
 * norm is global over ALL of them (base.py:286-292).  rgp_global_sqnorm writes RGP_SQNORM_PARTIALS partial
 * sums of squares of one buffer; rgp_adam_clip_step_ext applies the step to one buffer given the
 * concatenated partials of all buffers. */
#define RGP_SQNORM_PARTIALS 256
int rgp_global_sqnorm(const float* grads, long long n, float* partials, rgp_stream_t stream);
int rgp_adam_clip_step_ext(float* params, const float* grads, float* m, float* v, long long n, const float* partials,
                           int n_partials, int step, float lr, float beta1, float beta2, float eps, float max_grad_norm,
                           float* grad_norm_out, rgp_stream_t stream);

/* Graph-replayable optimizer step: the learning-rate schedule (gaze_rnn.py:436-444: lr0 * decay^floor(step /
 * decay_steps)) and Adam's bias correction are evaluated ON THE DEVICE from a device-resident step counter, so a
 * whole training step captured in a HIP graph replays correctly.  rgp_lr_schedule_step writes lr_t and advances
 * the counter (once per training step); rgp_adam_clip_step_dev is rgp_adam_clip_step_ext reading lr_t. */
int rgp_lr_schedule_step(int* step_dev, float lr0, float decay, int decay_steps, float beta1, float beta2, float* lr_t_dev,
                         rgp_stream_t stream);
int rgp_adam_clip_step_dev(float* params, const float* grads, float* m, float* v, long long n, const float* partials,
                           int n_partials, const float* lr_t_dev, float beta1, float beta2, float eps, float max_grad_norm,
                           float* grad_norm_out, rgp_stream_t stream);

/* The other two optimizers of create_train_op (base.py:268-273), same clip-then-apply contract and the same
 * `partials` (rgp_global_sqnorm of every flat buffer, concatenated) as rgp_adam_clip_step_ext:
 *  rgp_momentum_clip_step : tf.train.MomentumOptimizer(lr, momentum=0.9)   accum = momentum*accum + g; var -= lr*accum
 *  rgp_rmsprop_clip_step  : tf.train.RMSPropOptimizer(lr, decay=0.9, momentum=0.9, epsilon=1e-10)
 *                           ms = decay*ms + (1-decay)*g^2; mom = momentum*mom + lr*g/sqrt(ms+eps); var -= mom
 *                           (TF initialises the ms slot to ones, mom to zeros: the caller owns the slots). */
int rgp_momentum_clip_step(float* params, const float* grads, float* accum, long long n, const float* partials, int n_partials,
                           float lr, float momentum, float max_grad_norm, float* grad_norm_out, rgp_stream_t stream);
int rgp_rmsprop_clip_step(float* params, const float* grads, float* ms, float* mom, long long n, const float* partials,
                          int n_partials, float lr, float decay, float momentum, float eps, float max_grad_norm,
                          float* grad_norm_out, rgp_stream_t stream);

/* l2 loss (gaze_rnn.py:387-389, gaze_grcn_cascade.py:428-441): loss[0] = sum 0.5 (maps - labels)^2 / frames over
 * n elements; workspace: RGP_SQNORM_PARTIALS floats (deterministic two-stage sum). */
int rgp_l2_loss_fwd(const float* maps, const float* labels, long long n, int frames, float* workspace, float* loss,
                    rgp_stream_t stream);

/* Inverted dropout as an op (tf.nn.dropout: keep element i iff floor(keep_prob + u_i) = 1, kept values / keep_prob).
 * rgp_dropout_mask draws the keep mask (1 byte per element, 0 / 1) on the device with Philox-4x32-10 keyed by `seed`;
 * element i uses word i&3 of counter block offset + i/4, so the draw is independent of the launch geometry and a
 * caller advances `offset` by ceil(n/4) per training step.  The mask is the caller's: the training forward of a plan
 * with a dropout site reads it (rgp_fcgru_set_dropout, rgp_cascade_set_dropout) and the backward gates with the same
 * bytes.  rgp_dropout_apply is the op on a dense fp32 vector, in place (also its own backward). */
int rgp_dropout_mask(unsigned char* mask, long long n, float keep_prob, unsigned long long seed, unsigned long long offset,
                     rgp_stream_t stream);
int rgp_dropout_apply(float* x, const unsigned char* mask, long long n, float keep_prob, rgp_stream_t stream);

/* Stage timing (HIP events recorded on the caller's stream around each stage launch
 * group; costs two hipEventRecord per stage).  Stages: 0 proj (incl. transpose),
 * 1 xconv, 2 convgru sequence, 3 head (transposed convs), 4 softmax.
 * rgp_grcn_profile_read synchronises on the recorded events, writes the accumulated
 * milliseconds and launch-group counts since the last read, and resets them. */
#define RGP_GRCN_STAGES 5
int rgp_grcn_profile_enable(rgp_grcn_t* plan, int enable);
int rgp_grcn_profile_read(rgp_grcn_t* plan, double ms[RGP_GRCN_STAGES], long long calls[RGP_GRCN_STAGES]);

/* ------------------------------------------------------------------ fc-GRU (gaze_rnn) */
typedef struct rgp_fcgru rgp_fcgru_t;

/* GazePredictionGRU.create_gazeprediction_network (models/gaze_rnn.py:211-360), fp32 device
 * pointers: proj_c3d_W [1024,32] proj_c3d_b [32] (gaze_rnn.py:294-295); TF-1.x GRUCell(1617)
 * variables gates_kernel [1568+1617, 2*1617] (columns [r | u]), gates_bias [2*1617],
 * candidate_kernel [1568+1617, 1617], candidate_bias [1617] (gaze_rnn.py:315);
 * proj_out_W [1617, GH*GW], proj_out_b [GH*GW] (gaze_rnn.py:319-320). */
typedef struct rgp_fcgru_weights {
  const float *proj_c3d_W, *proj_c3d_b;
  const float *gates_kernel, *gates_bias, *candidate_kernel, *candidate_bias;
  const float *proj_out_W, *proj_out_b;
} rgp_fcgru_weights;

int rgp_fcgru_create(rgp_fcgru_t** plan, int batch, int n_steps, int gazemap_h, int gazemap_w, int dtype);
int rgp_fcgru_destroy(rgp_fcgru_t* plan);
size_t rgp_fcgru_workspace_bytes(const rgp_fcgru_t* plan);
int rgp_fcgru_bind_workspace(rgp_fcgru_t* plan, void* workspace, size_t bytes, rgp_stream_t stream);
int rgp_fcgru_set_weights(rgp_fcgru_t* plan, const rgp_fcgru_weights* w, rgp_stream_t stream);
/* c3d_input [B,T,1024,7,7] -> logits [B,T,GH,GW]; probs (optional) = per-frame softmax. */
int rgp_fcgru_forward(rgp_fcgru_t* plan, const float* c3d_input, float* logits, float* probs, rgp_stream_t stream);

/* Training (config 2): a plan created with save_for_backward = 1 keeps the gates and operand rows of every step;
 * rgp_fcgru_backward then differentiates the loss of gaze_rnn.py:363-408 (loss_type 0 xentropy, 1 l2) w.r.t.
 * all eight variables (grads: arrays shaped like the weights, fully overwritten), as rgp_grcn_backward does. */
int rgp_fcgru_create_ex(rgp_fcgru_t** plan, int batch, int n_steps, int gazemap_h, int gazemap_w, int dtype,
                        int save_for_backward);
int rgp_fcgru_backward(rgp_fcgru_t* plan, const float* logits, const float* probs, const float* labels,
                       const rgp_fcgru_weights* grads, int loss_type, rgp_stream_t stream);

/* Training-time dropout on the projected features c3d_embedded [B*T*49, 32] (gaze_rnn.py:302-303; single_step feeds
 * keep 0.5 when training, :529).  mask: device bytes [B*T*49*32] from rgp_dropout_mask (or the caller's own draw), read
 * by every following forward AND backward until changed; keep_prob = 1 or mask = NULL switches the site off
 * (inference, the default). */
int rgp_fcgru_set_dropout(rgp_fcgru_t* plan, float keep_prob, const unsigned char* mask);
/* ------------------------------------------------------------------ frame-wise ShallowNet */
typedef struct rgp_shallownet rgp_shallownet_t;

/* SaliencyModel.create_shallownet variables (models/saliency_shallownet.py:90-185), fp32 device
 * pointers: conv1_w [5,5,3,32] conv2_w [3,3,32,64] conv3_w [3,3,64,32] (HWIO) + biases;
 * fc1_w [n_flat, 4802] (n_flat = 3872 at 98x98, 4608 at 112x112), fc2_w [2401, 4802] + biases. */
typedef struct rgp_shallownet_weights {
  const float *conv1_w, *conv1_b, *conv2_w, *conv2_b, *conv3_w, *conv3_b;
  const float *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} rgp_shallownet_weights;

int rgp_shallownet_create(rgp_shallownet_t** plan, int max_frames, int image_hw, int dtype);
int rgp_shallownet_destroy(rgp_shallownet_t* plan);
size_t rgp_shallownet_workspace_bytes(const rgp_shallownet_t* plan);
int rgp_shallownet_bind_workspace(rgp_shallownet_t* plan, void* workspace, size_t bytes, rgp_stream_t stream);
int rgp_shallownet_set_weights(rgp_shallownet_t* plan, const rgp_shallownet_weights* w, rgp_stream_t stream);
/* frames [n,H,W,3] fp32 in [0,1] -> saliency [n,49,49] (saliency_shallownet.py:213); saliency7
 * (optional) [n,7,7] = its 7x7 average pool (gaze_rnn.py:262-269). */
int rgp_shallownet_forward(rgp_shallownet_t* plan, const float* frames, int n_frames, float* saliency,
                           float* saliency7, rgp_stream_t stream);

/* Training (FramewiseShallowNet trains every variable, gaze_framewise_shallownet.py:43-57): with a plan created
 * by rgp_shallownet_create_ex(save_for_backward = 1), rgp_shallownet_backward takes d loss / d saliency
 * [n_frames,49,49] fp32 for the frames of the last forward and overwrites grads (arrays shaped like the weights). */
int rgp_shallownet_create_ex(rgp_shallownet_t** plan, int max_frames, int image_hw, int dtype, int save_for_backward);
int rgp_shallownet_backward(rgp_shallownet_t* plan, int n_frames, const float* d_saliency, const rgp_shallownet_weights* grads,
                            rgp_stream_t stream);

/* ------------------------------------------------------------------ two-level cascade (config 5) */
typedef struct rgp_cascade rgp_cascade_t;

/* Variables of GazePredictionGRCN in models/gaze_grcn_cascade.py (fp32 device pointers):
 * proj_c3d_W [1,1,1024,512] + b (:269-275); bottom cell filters GRU_Conv_* (3x3, 512 -> 256, :290-303);
 * Upsampling/weight [11,11,64,256] (:317-321); top cell filters (5x5, x: [5,5,65,3], h: [5,5,3,3],
 * :346-357, input = concat(upsampled [64], ShallowNet saliency [1]));
 * fc1_w [7203,4802] fc2_w [2401,4802] + biases (:383-423); the ShallowNet's own variables. */
typedef struct rgp_cascade_weights {
  const float *proj_c3d_W, *proj_c3d_b;
  const float *bottom_Wz, *bottom_Uz, *bottom_Wr, *bottom_Ur, *bottom_W, *bottom_U;
  const float* upsampling_weight;
  const float *top_Wz, *top_Uz, *top_Wr, *top_Ur, *top_W, *top_U;
  const float *fc1_w, *fc1_b, *fc2_w, *fc2_b;
  rgp_shallownet_weights shallownet;
} rgp_cascade_weights;

int rgp_cascade_create(rgp_cascade_t** plan, int batch, int n_steps, int image_hw, int dtype);
int rgp_cascade_destroy(rgp_cascade_t* plan);
size_t rgp_cascade_workspace_bytes(const rgp_cascade_t* plan);
int rgp_cascade_bind_workspace(rgp_cascade_t* plan, void* workspace, size_t bytes, rgp_stream_t stream);
int rgp_cascade_set_weights(rgp_cascade_t* plan, const rgp_cascade_weights* w, rgp_stream_t stream);
/* frame_images [B*T,H,W,3] fp32 in [0,1], c3d_input [B,T,1024,7,7] -> gazemaps [B,T,49,49]
 * (predicted_gazemaps of gaze_grcn_cascade.py:423, trained with loss_type l2). */
int rgp_cascade_forward(rgp_cascade_t* plan, const float* frame_images, const float* c3d_input, float* gazemaps,
                        rgp_stream_t stream);
/* Training (BASELINE config 5).  rgp_cascade_create_ex(save_for_backward = 1) keeps the gates, states and
 * maxout masks of the forward.  rgp_cascade_backward differentiates the l2 loss of gaze_grcn_cascade.py:428-441
 * (sum_t 0.5 ||maps - gt||^2 / (B*T)) w.r.t. every non-ShallowNet variable (the ShallowNet has learning rate
 * 0, base.py:264-265): grads is shaped like the weights, every listed array is fully overwritten, the
 * shallownet sub-struct is ignored.  d_rows (optional, [B*T*49, 1024] fp32) receives the gradient w.r.t. the
 * C3D conv5b rows for rgp_c3d_backward (end-to-end fine-tune). */
int rgp_cascade_create_ex(rgp_cascade_t** plan, int batch, int n_steps, int image_hw, int dtype, int save_for_backward);
int rgp_cascade_backward(rgp_cascade_t* plan, const float* gazemaps, const float* gt_gazemap, const rgp_cascade_weights* grads,
                         float* d_rows, rgp_stream_t stream);
/* Training-time dropout on fc1's ReLU output, before the maxout (gaze_grcn_cascade.py:401-402).  mask: device bytes
 * [B*T, 4802] in the layer's own unit order (unit j and j + 2401 are maxout partners); keep_prob = 1 or NULL = off. */
int rgp_cascade_set_dropout(rgp_cascade_t* plan, float keep_prob, const unsigned char* mask);
/* Intermediates of the last forward as dense fp32 (net[...] keys of gaze_grcn_cascade.py):
 * "frm_sal" [B*T,49,49], "rcn_outputs" [B,T,7,7,256], "rcn_upsampled_outputs" [B*T,49,49,64],
 * "gaze_rcn_outputs" [B*T,49,49,3] (top-cell states). */
int rgp_cascade_read_buffer(rgp_cascade_t* plan, const char* name, float* dst, rgp_stream_t stream);

/* ------------------------------------------------------------------ C3D conv stack */
typedef struct rgp_c3d rgp_c3d_t;

/* conv1a..conv5b (prototxt:22-342): w[i] DHWIO [3,3,3,Cin,Cout] fp32, b[i] [Cout]. */
typedef struct rgp_c3d_weights {
  const float* w[8];
  const float* b[8];
} rgp_c3d_weights;

/* Plan for up to max_windows 16x112x112x3 windows per call. */
int rgp_c3d_create(rgp_c3d_t** plan, int max_windows, int dtype);
int rgp_c3d_destroy(rgp_c3d_t* plan);
size_t rgp_c3d_workspace_bytes(const rgp_c3d_t* plan);
int rgp_c3d_bind_workspace(rgp_c3d_t* plan, void* workspace, size_t bytes, rgp_stream_t stream);
int rgp_c3d_set_weights(rgp_c3d_t* plan, const rgp_c3d_weights* w, rgp_stream_t stream);
/* video [n,16,112,112,3] fp32 (mean-subtracted, channels last) -> conv5b after ReLU.
 * features (optional): [n,1024,7,7] fp32, channel = c*2+d (gaze_rnn.py:494-497).
 * rows (optional): [n*49][1024] in the plan's operand dtype, K order d*512+c, the
 * form rgp_grcn_forward_rows consumes. */
int rgp_c3d_forward(rgp_c3d_t* plan, const float* video, int n_windows, float* features, void* rows,
                    rgp_stream_t stream);
/* The VIDEO_DATA layer (feature_extration.prototxt:3-21) + rgp_c3d_forward in one call: frames is a
 * device stack [n_frames, frame_h, frame_w, 3] of 8-bit pixels in the channel order the weights were
 * trained with (OpenCV BGR for the Sports-1M model); window w is the 16 consecutive frames from
 * window_starts[w] (HOST int32 array; extract_C3D_features.py:866 uses 0, 16, 32, ...).  Each frame is
 * resized to 128x171 (bilinear), centre-cropped to 112x112 and mean_cube [3,16,128,171] (device fp32,
 * the parsed sport1m_train16_128_mean.binaryproto; NULL = no subtraction) is subtracted. */
int rgp_c3d_forward_frames(rgp_c3d_t* plan, const unsigned char* frames, int n_frames, int frame_h, int frame_w,
                           const int* window_starts, int n_windows, const float* mean_cube, float* features, void* rows,
                           rgp_stream_t stream);
/* Only the VIDEO_DATA step: writes video [n_windows,16,112,112,3] fp32 (what rgp_c3d_forward takes). */
int rgp_c3d_frames_to_video(rgp_c3d_t* plan, const unsigned char* frames, int n_frames, int frame_h, int frame_w,
                            const int* window_starts, int n_windows, const float* mean_cube, float* video,
                            rgp_stream_t stream);
/* Copies layer i's (0..7) pooled, post-ReLU output, un-padded fp32 NDHWC, into dst. */
int rgp_c3d_read_layer(rgp_c3d_t* plan, int layer, int n_windows, float* dst, rgp_stream_t stream);
size_t rgp_c3d_layer_elems(const rgp_c3d_t* plan, int layer, int n_windows);

/* ---- end-to-end fine-tune of the conv stack (BASELINE config 5; tf.gradients, base.py:278-281) ----
 * rgp_c3d_create_ex(save_for_backward = 1) makes the forward record the pooling arg-max and reserves the
 * gradient images.  After ONE forward of n_windows <= max_windows windows, rgp_c3d_backward takes the
 * gradient w.r.t. the conv5b feature -- d_features [n,1024,7,7] fp32 (layout of `features`) or d_rows
 * [n*49,1024] fp32 (layout of `rows`), exactly one non-NULL -- and ACCUMULATES (+=) the parameter
 * gradients into grads, a flat fp32 vector laid out w[0] (DHWIO), b[0], w[1], b[1], ... (the caller zeroes
 * it; rgp_c3d_param_offset gives each piece's element offset, rgp_c3d_param_elems the total). */
int rgp_c3d_create_ex(rgp_c3d_t** plan, int max_windows, int dtype, int flags);
/* flags of rgp_c3d_create_ex (0 / 1 keep the meaning of the former `save_for_backward` argument):
 *  RGP_C3D_SAVE_FOR_BACKWARD  training plan (arg-max codes, gradient images).
 *  RGP_C3D_KERNELS_IGEMM      bf16 plans: conv2a..conv4b (forward and input gradients) run through the general
 *                             implicit-GEMM kernels and their filter gradients through the general filter-gradient
 *                             kernel instead of the layer-specific patch kernels -- the library's second, independent
 *                             implementation of those layers (tile chosen by problem size: 256x256 / 512x128 /
 *                             staggered 256x128 / 128x128), kept for cross-checking the default path.
 *  RGP_C3D_KERNELS_TILE128    with RGP_C3D_KERNELS_IGEMM: every implicit GEMM of the plan on the 128x128 tile loop.
 *  RGP_C3D_CONV2A_ROWWISE     inference plans on the patch kernels: conv2a + pool2 through the row-wise fetch of
 *                             conv_patch.hip.h (what training plans run) instead of the plane-slab fetch of
 *                             conv_patch_slab.hip.h; bit-identical results (a cross-check and A/B switch). */
#define RGP_C3D_SAVE_FOR_BACKWARD 1
#define RGP_C3D_KERNELS_IGEMM 2
#define RGP_C3D_KERNELS_TILE128 4
#define RGP_C3D_CONV2A_ROWWISE 8
/* Name of the kernel instantiation the plan launches for layer i's forward at n_windows windows ("" if unknown):
 * what a profiler shows for the stage rgp_c3d_profile_read times as index i. */
const char* rgp_c3d_layer_kernel_name(const rgp_c3d_t* plan, int layer, int n_windows);
size_t rgp_c3d_param_elems(const rgp_c3d_t* plan);
size_t rgp_c3d_param_offset(const rgp_c3d_t* plan, int layer, int is_bias);
int rgp_c3d_backward(rgp_c3d_t* plan, const float* d_features, const float* d_rows, int n_windows, float* grads,
                     rgp_stream_t stream);
/* Data-parallel fine-tune (gradient buckets launched as their wgrads complete, late layers first).
 * rgp_c3d_backward records an event on its stream once layer i's slice of `grads` (w[i] then b[i], see
 * rgp_c3d_param_offset) is final; rgp_c3d_wait_layer_grads makes `waiting_stream` wait for that event, so the
 * host can issue the RCCL all-reduce of the slice there while the backward of layers i-1..0 is still running. */
int rgp_c3d_wait_layer_grads(rgp_c3d_t* plan, int layer, rgp_stream_t waiting_stream);
/* After rgp_c3d_backward: the gradient w.r.t. layer i's conv output before ReLU/pooling (what dgrad and
 * wgrad of that layer consumed) as dense fp32 [n, D, H, W, Cout].  Valid until the next forward/backward of
 * the plan.  (bf16 layer 0: the backward pass works from the pooled gradient and never builds this image; the
 * call expands it on demand from the pooled gradient and arg-max codes the pass left behind.) */
int rgp_c3d_read_grad_image(rgp_c3d_t* plan, int layer, int n_windows, float* dst, rgp_stream_t stream);

/* Per-layer timing, as rgp_grcn_profile_*: index 0..7 = conv1a..conv5b (one fused
 * conv+bias+ReLU+pool kernel launch each), 8 = video_prep. */
#define RGP_C3D_STAGES 9
int rgp_c3d_profile_enable(rgp_c3d_t* plan, int enable);
int rgp_c3d_profile_read(rgp_c3d_t* plan, double ms[RGP_C3D_STAGES], long long calls[RGP_C3D_STAGES]);

#ifdef __cplusplus
}
#endif
#endif /* RGP_H_ */
