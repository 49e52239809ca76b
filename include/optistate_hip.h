/*
 * optistate_hip.h -- C ABI of liboptistate_hip.so, the MI355X (gfx950) implementation of the
 * OptiState Kalman+GRU hot path.
 *
 * The reference (AlexS28/OptiState) exposes a Python call surface only; there is no FFI in it.
 * Each entry point below names the reference interface it replaces (file:line into the
 * reference tree).  The Python host side (optistate_amd/) binds these with ctypes and mirrors
 * the reference's `Kalman_Filter` and `RNN` classes; INTEGRATION.md shows the stub a
 * reference maintainer would add.
 *
 * Conventions
 *   - Every data pointer is a DEVICE pointer (hipMalloc / torch CUDA tensor storage) unless it
 *     is marked "host".  The caller owns all buffers; the library allocates only context scratch.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are
 *     stream-ordered and asynchronous; no call synchronises the device.
 *   - Return value: 0 ok, <0 error (os_last_error() gives text); -20 = a layer-pipelined launch lost a producer (os_gru_set_stack).  Per-trajectory numerical
 *     status is reported in `status[B]`: bit0 = innovation covariance S not positive definite /
 *     non-finite, bit1 = non-finite state (the reference raises LinAlgError /
 *     propagates NaN: kalman_filter/kalman_filter.py:168), bit2 = QP iteration cap (os_kf_mpc_run),
 *     bit3 = OS_KF_SYMMETRIC_P was requested but the caller's P0 is not symmetric (the kernel used its
 *     upper triangle; the reference never symmetrises P, kalman_filter/kalman_filter.py:172, so re-run
 *     that trajectory without the flag), bit4 (16, informational) = at some step an entry of the float64
 *     rotation matrix was within 2^-40 of +-1 at an attitude other than the exact start theta = 0: the
 *     reference stores R^T into an int64 array (misc/force_controller.py:248-251,271), so whether that
 *     entry integrates dt*omega into theta is decided by the last bits of the reference's own float64 state
 *     (gimbal lock: pitch = pi/2 with roll = yaw); on such a trajectory the 1e-4 state bar is not promised,
 *     the deviation is one dt*omega step per flagged decision (tests/test_gpu_fullsize.py).
 *     FAILURE = (status & OS_STATUS_FAIL_MASK) != 0.  Bit 4 is NOT a failure: a consumer that treats any non-zero
 *     word as "failed" must mask it (the Python engine does: Engine.failed(status) / Engine.trunc_edge(status)).
 *   - Stream layout (structure of arrays, trajectory index fastest, float32):
 *       p, f, dp, body_ref : [T][12][B]      imu, accel : [T][6][B]
 *       contact            : [T][B] of 4 packed bytes (byte k = leg k, 0 swing / 1 stance)
 *       x                  : [12][B]         P : [144][B] (row-major 12x12 per trajectory)
 *       x_out, p_rot_out   : [T][12][B]      ptrace_out, kgain_out : [T][B]
 *   - State order: thx thy thz  x y z  wx wy wz  vx vy vz  (kalman_filter/kalman_filter.py:9);
 *     measurement = state rows {0,1,2,5,6,7,8,9,10,11} (kalman_filter/kalman_filter.py:15-24).
 */
#ifndef OPTISTATE_HIP_H
#define OPTISTATE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct os_ctx os_ctx;

/* Model constants: settings.py:5-23, kalman_filter/kalman_filter.py:33-59, misc/force_controller.py:244-262. */
typedef struct os_kf_config {
    int32_t device;        /* HIP device ordinal */
    float dt;              /* settings.py:5   DT_mpc = 0.01 */
    float mass;            /* settings.py:11  8.8 kg */
    float inertia[3];      /* settings.py:20-23  body-frame diag(Ixx,Iyy,Izz) */
    float gz;              /* kalman_filter/kalman_filter.py:56  -9.81 (added to vz) */
} os_kf_config;

/* Bits of the per-trajectory status word (see "Conventions"). */
enum {
    OS_STATUS_S_NOT_PD   = 1,   /* bit0: innovation covariance not positive definite / non-finite */
    OS_STATUS_NONFINITE  = 2,   /* bit1: non-finite state */
    OS_STATUS_QP_ITER    = 4,   /* bit2: QP iteration cap (os_kf_mpc_run) */
    OS_STATUS_P0_ASYM    = 8,   /* bit3: OS_KF_SYMMETRIC_P with a non-symmetric P0 */
    OS_STATUS_TRUNC_EDGE = 16,  /* bit4: INFORMATIONAL, int64-truncation knife edge (not a failure) */
    OS_STATUS_FAIL_MASK  = 15   /* failure = (status & OS_STATUS_FAIL_MASK) != 0 */
};

/* Flags for os_kf_run / os_fused_run. */
enum {
    OS_KF_SEQUENTIAL_UPDATE = 1,  /* process the 10 measurements one at a time (requires diagonal R; same
                                     posterior as the batch form in exact arithmetic).  Without it the
                                     batch form K = P H^T S^-1 (LU of S as it is, float64) is used, as written in
                                     kalman_filter/kalman_filter.py:166-172. */
    OS_KF_DENSE_FD          = 2,  /* predict_mpc covariance: F_d = element-wise exp(dt F), R from body_ref
                                     (kalman_filter/kalman_filter.py:153-158); needs body_ref. */
    OS_KF_SYMMETRIC_P       = 4,  /* with OS_KF_SEQUENTIAL_UPDATE (and not DENSE_FD): keep only the upper triangle of
                                     P in registers (P is symmetric in exact arithmetic); P0's upper triangle is used
                                     and the final P is written back mirrored.  ~2x faster; same 1e-4 parity bar. */
    OS_FUSED_TWO_KERNEL     = 8,  /* os_fused_run only: force the general two-kernel path (Kalman kernel writes the
                                     normalised feature rows to context scratch, GRU kernels read them) even where
                                     the single-kernel path applies (60 features, hidden 64, sequential+symmetric). */
    OS_FUSED_ONE_KERNEL     = 128, /* os_fused_run: take the single fused kernel whenever the shapes allow it, also for batches
                                     that fill less than a third of the chip (default: the two-kernel path there, it is
                                     faster below ~20 k trajectories; OS_FUSED_TWO_KERNEL forces that path) */
    OS_MPC_COLD_START       = 64, /* os_kf_mpc_run: do not reuse the previous step's active set (development / tests) */
    OS_KF_LANE_PER_TRAJECTORY = 32, /* os_kf_run: never use the small-batch kernel (16 lanes per trajectory, row-parallel P),
                                     which is otherwise chosen for sequential updates when B <= 8,192 (the measured crossover).
                                     Without OS_KF_SYMMETRIC_P the full, never symmetrised P of the reference is carried -- in float64,
                                     16 lanes per trajectory (kf_dense_rows_kernel). */
    OS_KF_P_FLOAT64         = 256, /* os_kf_predict / os_kf_update: P (and K_out) are DOUBLE arrays and the covariance
                                     arithmetic runs in float64.  predict_mpc's element-wise exp(dt F) (kalman_filter.py:157)
                                     leaves a P that float32 cannot carry to the following update within the 1e-4 bar; the
                                     drop-in class uses this for the split predict_mpc() -> update() sequence. */
    OS_FUSED_SPLIT_BF16     = 512, /* os_fused_run: opt-in gate GEMM on the bf16 MFMA (fp32 operands split into THREE bf16
                                     terms = all 24 mantissa bits, six MFMAs per product block, fp32 accumulate); NOT the
                                     exact-fp32 default.  Measured GRU head l-inf vs the float64 oracle: 8e-8. */
    OS_FUSED_SPLIT_BF16_2   = 1024, /* the same with TWO bf16 terms (16 mantissa bits, three MFMAs per block): 5e-7. */
    OS_KF_WAVE_PER_TRAJECTORY = 4096, /* os_kf_run: ONE TRAJECTORY PER WAVEFRONT with x and P in LDS for all T steps (the layout
                                     BASELINE.json's north_star names), float64 batch update across the lanes.  Never chosen by default:
                                     it exists so that the layout is measured, not argued (20-50x slower than the
                                     lane-per-trajectory / 16-lanes-per-trajectory kernels). */
    OS_FUSED_LATENT_IN_PLACE = 2048 /* os_fused_run with n_latent > 0: `latent` is the caller's WHOLE GRU input buffer [T][60 + NL][B]
                                     with rows 60.. already holding the latent stream (os_pack_stream_rows writes them there);
                                     the Kalman kernel fills rows 0..59 in place and the GRU reads the buffer directly -- no
                                     feature scratch, no copy of the latent (1 KB per step at NL = 128).  The buffer is WRITTEN. */
};

/* Replaces Kalman_Filter.__init__ (kalman_filter/kalman_filter.py:8-62): creates a context on cfg->device. */
int os_create(const os_kf_config *cfg, os_ctx **out);
void os_destroy(os_ctx *ctx);
const char *os_last_error(const os_ctx *ctx);
/* Library/ABI version and build target string ("gfx950"). */
int os_version(void);
const char *os_build_arch(void);
/* "<hash of the sources, headers and compile flags>-<hash of `hipcc --version`>", stamped by optistate_amd/build.py;
 * the Python loader refuses a library whose first half does not match the sources it sits beside. */
const char *os_build_id(void);

/* ONE filter step of ONE trajectory with every argument in HOST memory (plain float64 arrays): replaces, for a single
 * `Kalman_Filter` instance, any combination of
 *   OS_STEP_ODOM     odom = get_odom(p, dp, contact, imu); set_measurements(imu, odom)   (kalman_filter/kalman_filter.py:79-117) -> z
 *   OS_STEP_PREDICT  predict(p, f) (:119-138) or, with OS_STEP_DENSE_FD, the covariance / next_state half of
 *                    predict_mpc(p, body_ref, .) with the forces supplied (:153-161) -> x, P, p_rot (p rotated in place by
 *                    next_state, misc/force_controller.py:274-277), x_model, P_trace
 *   OS_STEP_UPDATE   update() (:164-174) -> x, P, K (12 x 10), P_trace, K_gain
 * in ONE kernel launch and ONE stream synchronise (the caller loop data_conversion_Kalman_to_Training.py:193-199 is these
 * four calls per time step).  The call is SYNCHRONOUS: results are in the caller's arrays when it returns.  All arithmetic is
 * float64 (one wavefront, covariance in LDS); Q (144) and R (100) come with the call, like the reference's per-instance
 * attributes.  model: {dt, mass, Ixx, Iyy, Izz, g_z} in float64 or NULL for the context's (float32) configuration.
 * contact: 4 bytes (0/1).  status: bit 0 S not positive definite (the reference's np.linalg.inv raises), bit 1 non-finite x.
 * Unused pointers may be NULL (e.g. K, p_rot). */
enum { OS_STEP_ODOM = 1, OS_STEP_PREDICT = 2, OS_STEP_UPDATE = 4, OS_STEP_DENSE_FD = 8,
       OS_STEP_MPC = 16 /* the predict's forces come from the convex-MPC QP solved on the device in front of the step (os_kf_step_mpc) */ };
int os_kf_step(os_ctx *ctx, uint32_t what, const double *model, const double *p, const double *f, const double *dp,
               const double *imu, const uint8_t *contact, const double *body_ref, const double *Q, const double *R,
               double *x, double *P, double *z, double *p_rot, double *x_model, double *K, double *ptrace, double *kgain,
               int32_t *status, void *stream);

/* Kalman_Filter.estimate_state_mpc (kalman_filter/kalman_filter.py:176-182, the body of the caller loop
 * data_collection/data_conversion_Kalman_to_Training.py:193-199) in ONE call: the stance controller's QP (:141-152;
 * misc/force_controller.py:70-225) is solved on the device from the PRIOR state x, body_ref, p and the contact pattern, its horizon-step-0
 * forces feed next_state, then get_odom + set_measurements + predict_mpc + update as `what` says (OS_STEP_MPC | OS_STEP_PREDICT are
 * implied; pass OS_STEP_ODOM | OS_STEP_DENSE_FD | OS_STEP_UPDATE for the reference's sequence).  Two launches (the QP instance the
 * contact word needs, warm-started from the context's previous solve when the contact pattern is unchanged; the float64 step kernel)
 * and ONE stream synchronise.  f_all (optional): the (12, 5) control matrix the reference keeps in self.f, row-major; qp_iters
 * (optional): active-set iterations; status bit 2: QP iteration cap.  Everything else as os_kf_step. */
int os_kf_step_mpc(os_ctx *ctx, uint32_t what, const double *model, const double *p, const double *dp, const double *imu,
                   const uint8_t *contact, const double *body_ref, const double *Q, const double *R, double *x, double *P,
                   double *z, double *p_rot, double *x_model, double *K, double *ptrace, double *kgain, double *f_all,
                   int32_t *qp_iters, int32_t *status, void *stream);

/* Replaces the attribute writes KF.Q = Q; KF.R = R (data_collection/data_conversion_Kalman_to_Training.py:139-143).
 * Q host float[144], R host float[100], row-major. */
int os_kf_set_noise(os_ctx *ctx, const float *Q_host, const float *R_host);

/* Replaces the per-trajectory loop body
 *   odom = get_odom(p, dp, contact, imu); set_measurements(imu, odom); predict(p, f); update()
 * (kalman_filter/kalman_filter.py:79-138,164-174; caller loop
 * data_collection/data_conversion_Kalman_to_Training.py:193-199) for B trajectories x T steps.
 * x, P are in/out.  p_rot_out (world-rotated p, the in-place side effect of next_state,
 * misc/force_controller.py:274-277), ptrace_out (P_trace), kgain_out (K_gain = np.trace(K), kalman_filter.py:174: the batch
 * form sums the diagonal of the K it built; the sequential / symmetric / 16-lane forms, which never form K, evaluate
 * trace(P+ H^T R^-1) = sum_a P+[a][sel a] / R[a][a] on their posterior covariance -- the same number for the optimal gain; against the
 * reference's float64 value: <= 5e-7 under the default noise, <= 3.3e-5 where R = 1e-4 divides a float32 posterior; test bar 1e-4 absolute),
 * body_ref may be NULL. */
int os_kf_run(os_ctx *ctx, int32_t B, int32_t T,
              const float *p, const float *f, const float *dp, const float *imu, const uint32_t *contact,
              const float *body_ref,
              float *x, float *P,
              float *x_out, float *p_rot_out, float *ptrace_out, float *kgain_out,
              int32_t *status, uint32_t flags, void *stream);

/* os_kf_run with PER-TRAJECTORY noise: q_diag [12][B] and r_diag [10][B] (device) replace the context-wide Q / R of
 * os_kf_set_noise, so filters tuned differently share one launch.  The reference sets Q and R per filter instance
 * (data_collection/data_conversion_Kalman_to_Training.py:138-144: KF2.Q = Q, KF2.R = R with R[0:3] forced to 1e-4; both
 * np.diag, :87,:103-105).  Diagonal noise only (sequential update); lane-per-trajectory kernels at every batch size. */
int os_kf_run_noise(os_ctx *ctx, int32_t B, int32_t T,
                    const float *p, const float *f, const float *dp, const float *imu, const uint32_t *contact,
                    float *x, float *P, const float *q_diag, const float *r_diag,
                    float *x_out, float *p_rot_out, float *ptrace_out, int32_t *status, uint32_t flags, void *stream);

/* Single-instance pieces for the drop-in Kalman_Filter class (B = 1 views over the same kernels).
 * os_kf_odom   : get_odom + set_measurements (kalman_filter/kalman_filter.py:79-117) -> z [10][B]
 * os_kf_predict: predict(p, f) (kalman_filter/kalman_filter.py:119-138); p [12][B] is rotated in place.
 * os_kf_update : update() (kalman_filter/kalman_filter.py:164-174); K_out [120][B] (12x10 row-major) optional; with
 *                OS_KF_SEQUENTIAL_UPDATE (diagonal R) K_out / kgain_out are formed from the posterior: K = P+ H^T R^-1. */
int os_kf_odom(os_ctx *ctx, int32_t B, const float *p, const float *dp, const uint32_t *contact, const float *imu,
               float *z, void *stream);
int os_kf_predict(os_ctx *ctx, int32_t B, float *p, const float *f, const float *body_ref, float *x, float *P,
                  float *ptrace_out, uint32_t flags, void *stream);
int os_kf_update(os_ctx *ctx, int32_t B, const float *z, float *x, float *P, float *K_out, float *ptrace_out,
                 float *kgain_out, int32_t *status, uint32_t flags, void *stream);

/* GRU head dims: RNN(input_size, hidden_size, num_layers, num_classes) (gru/gru_model.py:8-24). */
typedef struct os_gru_dims {
    int32_t input_size, hidden_size, num_layers, num_classes, use_sigmoid;
} os_gru_dims;

/* Number of floats in the flat weight vector: per layer W_ih [3H][I_l], W_hh [3H][H], b_ih [3H], b_hh [3H]
 * (gate order r|z|n, torch.nn.GRU), then fc.weight [C][H], fc.bias [C]. */
size_t os_gru_param_count(const os_gru_dims *d);

/* Replaces model.load_state_dict(...) (gru/gru_test.py:160): w_flat is a DEVICE float vector in the flat
 * layout above; the library re-packs it into MFMA fragment order in context scratch. */
int os_gru_load(os_ctx *ctx, const os_gru_dims *d, const float *w_flat, void *stream);
/* Every following os_gru_backward / os_gru_backward_ws on this context records `event` (a hipEvent_t; NULL switches it off) on
 * its stream right behind the weight-gradient kernel of GRU layer `layer`: from then on the gradients of layers layer..L-1 and
 * of the head (the tail of the flat gradient vector from that layer's offset) are complete, while the layers below are still
 * being swept.  The data-parallel step (gru/gru_train.py:232-251 over RCCL) starts the all-reduce of that half on a side
 * stream there. */
int os_gru_backward_mark(os_ctx *ctx, int32_t layer, void *event);
/* The same with a caller-chosen key != 0 naming (these weights, in this state): the context keeps the packed images of the
 * four most recently used keys, so models that alternate on one context (the ensemble of gru/gru_train.py:205-217,
 * `num_models`) are re-selected without re-packing.  The caller must use a NEW key whenever the weights behind w_flat
 * change (and keep w_flat alive while the key is in use); os_gru_generation counts the packs actually done. */
int os_gru_load_keyed(os_ctx *ctx, const os_gru_dims *d, const float *w_flat, uint64_t key, void *stream);
/* Counts os_gru_load calls on this context.  A context holds ONE loaded model; a host-side weight container that shares
 * a context with others (several RNN modules on one GPU, gru/gru_train.py:205-217 trains num_models of them) compares
 * this with the value it saw after its own load to know whether its weights are still the resident ones. */
uint64_t os_gru_generation(const os_ctx *ctx);

/* Replaces RNN.forward (gru/gru_model.py:25-49): x [B][T][I] (batch_first, as the reference passes it)
 * -> out [B][C]; h0 = 0; fc + sigmoid on the last step.  h_last (optional) [L][B][H]. */
int os_gru_forward(os_ctx *ctx, int32_t B, int32_t T, const float *x, float *out, float *h_last, void *stream);

/* Same with the input already in the library's stream layout xs [T][I][B]; h_last_soa (optional) [L][H][B]. */
int os_gru_forward_soa(os_ctx *ctx, int32_t B, int32_t T, const float *xs, float *out, float *h_last_soa, void *stream);

/* Replaces the reference's inference loop over sliding windows (gru/gru_test.py:138-140 builds window i = rows i .. i + W - 1 of ONE
 * time-ordered row stream, :155,174-191 evaluates them one by one from h0 = 0): rows [N][I] row-major -> out [N - W + 1][C], output
 * i = RNN.forward(rows[i : i + W]).  The windows are never materialised (that tensor is W times the stream) and the input half of
 * the first layer's gate GEMM, x W_ih^T, is computed once per ROW and shared by the W windows that contain it (17 % of the model's
 * flops at RNN(188,128,4)).  hidden_size 128 or 64, input_size <= 192; returns -4 for other shapes (materialise + os_gru_forward). */
int os_gru_forward_windows(os_ctx *ctx, int32_t n_rows, int32_t window, const float *rows, float *out, void *stream);

/* Small batches run their layer stack as ONE launch whose layers hand each other time steps through progress counters
 * (gru_stack_kernel, bwd_sweep_stack_kernel).  Every wait in there is bounded; a consumer whose wait expires sets the context's error
 * word, stops waiting and poisons its input with NaN.  mode 1 (default; OS_GRU_STACK): a call that made such a launch waits for it and
 * returns -20 when the word is set (os_last_error says which kernel): re-run that call after os_gru_set_stack(ctx, 0).  mode 2: the
 * launch stays asynchronous; the word is reported (-20) by the NEXT os_gru_forward* / os_gru_backward* / os_adam_step call that finds
 * it, and os_adam_step's kernel skips its update while it is set (the model is never stepped on poisoned gradients).
 * mode 0: a launch per layer, no progress counters. */
int os_gru_set_stack(os_ctx *ctx, int32_t mode);
/* The mode in force (the OS_GRU_STACK value os_create read, or the last os_gru_set_stack): 0 | 1 | 2; < 0: invalid context.  A host layer
 * that switches modes temporarily (the trainer: 2 for its own step) restores THIS value, not a guess. */
int os_gru_get_stack(const os_ctx *ctx);
/* Mode 2's explicit verification point: if an asynchronous stacked launch has gone out on this context since the last check, waits for
 * `stream` and returns -20 when the error word is set (cleared, the device twin too: the stream is drained), 0 otherwise; returns 0 at
 * once, without waiting, when there was no such launch.  The trainer calls it once per step between the backward and the gradient
 * all-reduce: a lost step is redone with a launch per layer BEFORE its gradients reach a collective or the optimiser
 * (gru/gru_train.py:232-249), so no rank ever contributes a poisoned gradient. */
int os_stack_check(os_ctx *ctx, void *stream);

/* OPT-IN reduced precision for the GRU layer GEMMs at large batches (never the default; env OS_GRU_SPLIT_BF16 = 2 | 3 at os_create).
 * mode 0: exact fp32 (v_mfma_f32_32x32x2_f32).  mode 3 / 2: every fp32 operand of the gate GEMM of an H = 128 inference layer is
 * split into 3 / 2 bf16 terms and the products run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (gru_layer_bf16_kernel:
 * 6 / 3 products per operand pair; 3 terms keep every product term down to 2^-16, measured GRU l-inf vs float64 in
 * tests/test_gpu_gru_bf16.py).  Applies to layers with input width <= 188 at batches of at least 128 x CUs trajectories (mode |
 * OS_GRU_SPLIT_ANY_BATCH: any batch that is a multiple of 4 -- tests); every other shape and the training path stay on the fp32
 * kernels.  The reference computes this GEMM in fp32 (torch.nn.GRU, gru/gru_model.py:12); returns -4 for another mode. */
#define OS_GRU_SPLIT_ANY_BATCH 0x100
/* round 6: | OS_GRU_SPLIT_TRAIN extends the opt-in to the TRAINING step's weight-gradient products (gru/gru_train.py:247 loss.backward():
 * dW_ih += dG^T X, dW_hh += dG^T H_prev on dw3_bf16_kernel, fp32 accumulation and fp32 atomics unchanged); the training forward and the
 * backward sweep stay on the fp32 matrix instruction.  With 3 terms the gradients equal the fp32 path's to fp32 rounding. */
#define OS_GRU_SPLIT_TRAIN 0x200
int os_gru_set_split_bf16(os_ctx *ctx, int32_t mode);

/* The post-processing of the reference's evaluation loop (gru/gru_test.py:184-189,208-213) in one launch: out [B][2 n] = [prediction |
 * error] (normalised) -> pred = p (max - min) + min, above = (p + e) (max - min) + min, below = (p - e) (max - min) + min, each
 * [B][n]; min_v, max_v: device float[n] (the label scaling of gru/gru_train.py:59-62). */
int os_gru_bands(os_ctx *ctx, int32_t B, int32_t n, const float *out, const float *min_v, const float *max_v, float *pred, float *above,
                 float *below, void *stream);

/* Fused path (single kernel where it applies, see OS_FUSED_TWO_KERNEL): KF loop + 60-feature row [x_post | accel | f | p_world | dp | imu]
 * (data_collection/data_conversion_Kalman_to_Training.py:245-254) + min-max normalisation
 * (gru/gru_test.py:99-101; minmax = device float[2][60]: mins then maxs) + optional latent [T][NL][B] appended
 * (gru/gru_test.py:135-136) + GRU over the T-step sequence + head, without writing feature rows to HBM.
 * Requires os_gru_load with input_size == 60 + n_latent.  out [B][C]. */
/* os_pack_stream into rows [row0, row0 + F) of a wider stream: src [B][T][F] (batch-major, e.g. the ViT encoder's latent
 * (N, 128) = (B, T, 128)) -> dst [T][F_total][B].  With row0 = 60, F_total = 60 + F this builds the OS_FUSED_LATENT_IN_PLACE input. */
int os_pack_stream_rows(os_ctx *ctx, int32_t B, int32_t T, int32_t F, const float *src, float *dst, int32_t F_total, int32_t row0,
                        void *stream);

/* The single fused kernel exists for five tile shapes -- trajectories per workgroup (= per CU): 256 | 128 (64 | 32 trajectories per
 * wavefront, 32x32x2 MFMA), 64 (16 per wavefront, 16x16x4 MFMA), 32 | 16 (a 16-trajectory tile's hidden units split over 2 | 4
 * wavefronts).  os_fused_run picks the shape that minimises rounds x measured cost for the batch (a shard of 8,192 trajectories of a
 * 65,536 batch takes 32 per CU and still fills the chip); tile = 0 restores that choice, another value pins the shape (tests, sweeps;
 * env OS_FUSED_TILE at os_create).  Shapes below 128 exist for one-layer models only; -2 for any other value.  Every shape computes
 * the same exact-fp32 arithmetic on data_collection/data_conversion_Kalman_to_Training.py:193-199,245-254 + gru/gru_model.py:27-48. */
int os_fused_set_tile(os_ctx *ctx, int32_t tile);

int os_fused_run(os_ctx *ctx, int32_t B, int32_t T,
                 const float *p, const float *f, const float *dp, const float *imu, const uint32_t *contact,
                 const float *accel, const float *body_ref, const float *latent, int32_t n_latent,
                 const float *minmax,
                 float *x, float *P, float *x_out, float *out, int32_t *status, uint32_t flags, void *stream);

/* ---- Training step of the GRU head (gru/gru_train.py:232-249) ----
 * os_gru_forward_train: RNN.forward (gru/gru_train.py:236) keeping the per-step activations in context scratch.
 * os_gru_loss          : the target construction + nn.MSELoss of gru/gru_train.py:237-245:
 *                        target = [y | |out[:, :C/2] - y|] with `out` detached; loss (device scalar) = mean squared error
 *                        over B*C entries; dout [B][C] = d(loss)/d(out); target [B][C] optional.  y [B][C/2].
 * os_gru_backward      : loss.backward() (gru/gru_train.py:248): all parameter gradients into grad_flat (flat layout of
 *                        os_gru_param_count; this is the single bucket a data-parallel step all-reduces), optional
 *                        d(loss)/d(x) in dx [B][T][I].
 * os_adam_step         : torch.optim.Adam's update (gru/gru_train.py:219,249) fused over flat vectors; step counts from 1. */
int os_gru_forward_train(os_ctx *ctx, int32_t B, int32_t T, const float *x, float *out, void *stream);
int os_gru_loss(os_ctx *ctx, int32_t B, const float *out, const float *y, float *target, float *dout, float *loss,
                void *stream);
int os_gru_backward(os_ctx *ctx, int32_t B, int32_t T, const float *x, const float *out, const float *dout,
                    float *grad_flat, float *dx, void *stream);
int os_adam_step(os_ctx *ctx, size_t n, float *w, const float *g, float *m, float *v, float lr, float beta1, float beta2,
                 float eps, int32_t step, void *stream);
/* The same forward / backward with the saved activations in a CALLER-OWNED workspace (os_gru_train_ws_floats floats), so
 * that several forwards may be outstanding before their backwards run (loss(model(a)) + loss(model(b)), gradient
 * accumulation, an evaluation forward in between: torch autograd allows all of these around gru/gru_train.py:236-248).
 * os_gru_backward_ws takes the dims and the flat weights the forward ran with explicitly: it does not depend on what is
 * loaded in the context at that time. */
size_t os_gru_train_ws_floats(const os_gru_dims *d, int32_t B, int32_t T);
int os_gru_forward_train_ws(os_ctx *ctx, int32_t B, int32_t T, const float *x, float *out, float *ws, void *stream);
int os_gru_backward_ws(os_ctx *ctx, const os_gru_dims *d, const float *w_flat, int32_t B, int32_t T, const float *x,
                       const float *out, const float *dout, const float *ws, float *grad_flat, float *dx, void *stream);

/* ---- Optional ViT-encoder latent (transformer/transformer_model.py:113-135; BASELINE config 5).  PARITY UNPINNED: timm 0.3.2
 * (PatchEmbed, Block) and the trained weights are absent from the build image; this follows that release's published
 * definition and is checked against this repo's own float64 restatement only.
 * Flat weight layout (device floats): patch_embed.proj.weight [D][P*P], .bias [D], cls_token [D], pos_embed [L][D]; per block:
 * norm1.weight, norm1.bias, attn.qkv.weight [3D][D], attn.qkv.bias [3D], attn.proj.weight [D][D], attn.proj.bias, norm2.weight,
 * norm2.bias, mlp.fc1.weight [M][D], mlp.fc1.bias [M], mlp.fc2.weight [D][M], mlp.fc2.bias [D]; then norm.weight, norm.bias.
 * os_vit_encode: images [N][img][img] float in [0,1] (gru/gru_test.py:49-53) -> latent [N][D] = sigmoid(LN(cls token)). */
typedef struct os_vit_dims {
    int32_t img_size, patch_size, in_chans, embed_dim, depth, num_heads, mlp_hidden;
} os_vit_dims;
size_t os_vit_param_count(const os_vit_dims *d);
int os_vit_load(os_ctx *ctx, const os_vit_dims *d, const float *w_flat);
int os_vit_encode(os_ctx *ctx, int32_t N, const float *images, float *latent, void *stream);

/* ---- Convex-MPC ground-reaction forces ("next" row f.2): the QP the reference solves with casadi + qpOASES inside
 * predict_mpc (misc/force_controller.py:70-162 builds it, kalman_filter/kalman_filter.py:141-152 fills the parameters
 * and takes column 0 of the solution).  N = 5 horizon steps x 12 forces; per trajectory:
 *     min  sum_i (x_{i+1} - body_ref)^T Q (x_{i+1} - body_ref) + u_i^T R u_i,
 *          x_{i+1} = (I + A_i dt) x_i + B_i dt u_i + dt g   (A_0, B_0 from the current x, A_i, B_i (i >= 1) from body_ref)
 *     s.t. contact byte 0: u_leg = 0;  contact byte 1: 0 <= fz <= fz_max, |fx| <= mu fz, |fy| <= mu fz.
 * Solved exactly (float64 primal active-set iteration, one wavefront per trajectory); PARITY UNPINNED against qpOASES
 * (absent), checked against the KKT-certified oracle/mpc_oracle.py.
 * os_mpc_set_weights: diag(Q) (12), R scalar, mu, fz_max; defaults are the reference's (kalman_filter.py:64-70,
 *   force_controller.py:147-149).
 * os_mpc_solve: x, body_ref, p [12][B] float32 device arrays, contact [B] packed bytes -> f_out [12][B] (the forces of
 *   horizon step 0 = self.f[:, 0], kalman_filter.py:161), optional u_out [60][B] (all steps) and iters [B];
 *   status [B] is OR-ed with 4 where the iteration cap max_iter (<= 0: 200) was reached. */
int os_mpc_set_weights(os_ctx *ctx, const double *q_weights, double r_weight, double mu, double fz_max);
int os_mpc_solve(os_ctx *ctx, int32_t B, const float *x, const float *body_ref, const float *p, const uint32_t *contact,
                 float *f_out, float *u_out, int32_t *iters, int32_t *status, int32_t max_iter, void *stream);

/* B trajectories x T steps of Kalman_Filter.estimate_state_mpc (kalman_filter/kalman_filter.py:176-182) exactly as
 * data_collection/data_conversion_Kalman_to_Training.py:194-199 drives it: per step the forces come from the QP above
 * (solved from the state before the predict), then get_odom + set_measurements + predict_mpc (dense F_d covariance,
 * next_state with f[:, 0]) + update.  Streams as os_kf_run; f_out [T][12][B] receives the forces (KF2.f[:, 0], the
 * feature columns 18..29 of :248-250), mpc_iters [T][B] (optional) the active-set iteration counts; status [B] is
 * written (bits 0/1 as os_kf_run, bit 2 = QP iteration cap).  Up to 32 trajectories per compute unit (B <= 8192 on an
 * MI355X) this is ONE persistent kernel, a wavefront per trajectory for all T steps (QP with its warm start in registers,
 * then the filter step with the float64 covariance in LDS spread over the 64 lanes): no per-step launches, nothing read
 * back, no stream synchronisation; P is carried in float64 between steps, as in the reference, and rounded to float32
 * only when it is written back at the end of the call.  Larger batches take the launch sequence (per step a QP launch per
 * leg count + the T = 1 filter step; the leg-count histogram is read back once at entry), which has the higher throughput
 * there.  OS_MPC_PERSISTENT=0 / 2 in the environment forces the sequence / the persistent kernel.
 * Round 6, the sequence at batches of 64 and more whose trajectories carry force on at most two legs in a step: the QPs run
 * sixteen lanes each (mpc_quad.hip), and -- for the plain call: no flag but OS_MPC_COLD_START, no P_trace / K_gain output --
 * the filter step of a trajectory runs INSIDE the QP launch that solved its forces (same arithmetic: x, P, x_out, status are
 * bit-identical to the separate launch; OS_MPC_FUSE_KF=0 switches back).  When every step of the call qualifies and B > 8,192
 * the batch runs as two contiguous halves on two internal streams, forked from and joined into `stream` by events (the halves
 * share nothing; OS_MPC_SHARDS=1 keeps one part): the call stays asynchronous and ordered with respect to `stream`.
 * Batches of 8 .. 200 trajectories per compute unit (2,048 .. 51,200) whose every step carries force on the same number (one or two) of
 * legs or on none take a third form in the plain call: kf_mpc_rows_kernel, a 16-lane row per trajectory for ALL T steps (QP -> filter
 * step -> next QP with nothing synchronised between steps; x and P travel through memory as in the launch sequence, so its numbers
 * are the sequence's to ~1e-7); OS_MPC_ROWS=0 switches it off, OS_MPC_ROWS=<lo>:<hi> moves the range.
 * (status bit 6 = 64: a filter step inside a QP launch gave up waiting for its trajectory's forces after 2^24 polls -- a lost
 * device; never seen, tested beside a process that holds 240 of the 256 compute units.)
 * OS_KF_SEQUENTIAL_UPDATE in `flags` selects the scalar-update form in the launch sequence only; the persistent kernel always
 * uses the batch form of kalman_filter.py:166-172 (LU of S) -- the same posterior for the diagonal R the flag requires. */
int os_kf_mpc_run(os_ctx *ctx, int32_t B, int32_t T, const float *p, const float *dp, const float *imu,
                  const uint32_t *contact, const float *body_ref, float *x, float *P, float *x_out, float *f_out,
                  float *p_rot_out, float *ptrace_out, float *kgain_out, int32_t *mpc_iters, int32_t *status,
                  uint32_t flags, void *stream);

/* Per-kernel device timing (HIP events recorded on the launch stream around each internal kernel), used by
 * bench.py for the roofline of the dominant kernel.  os_profile_read synchronises on the recorded events, adds up the
 * elapsed milliseconds and launch counts per phase since os_profile_enable(ctx, 1), and resets them.
 * os_profile_kernel_name: the kernel VARIANT most recently launched in a phase (e.g. "kf_run_rows_kernel" for a small
 * batch, "kf_run_sym_kernel" for the paired-triangle fast path), so that a report names the kernel that actually ran. */
#define OS_PROF_PHASES 12
enum {
    OS_PHASE_KF = 0,          /* Kalman kernels (os_kf_run and the Kalman half of the two-kernel fused path) */
    OS_PHASE_GRU_LAYER = 1,   /* GRU layer kernels (inference and training forward) */
    OS_PHASE_GRU_HEAD = 2,    /* fc + sigmoid head */
    OS_PHASE_FUSED = 3,       /* fused Kalman+GRU kernel */
    OS_PHASE_MPC = 4,         /* convex-MPC force QP kernels */
    OS_PHASE_TRAIN_SWEEP = 5, /* backward-in-time sweep (bwd_sweep_kernel) */
    OS_PHASE_TRAIN_DW = 6,    /* weight-gradient reductions (dw_kernel) */
    OS_PHASE_TRAIN_MISC = 7,  /* loss, head backward, weight re-packs, bias column sums, Adam */
    OS_PHASE_VIT_GEMM = 8,    /* ViT dense projections */
    OS_PHASE_VIT_ATTN = 9,    /* ViT attention */
    OS_PHASE_VIT_MISC = 10,   /* ViT patch extraction, LayerNorm, token assembly */
    OS_PHASE_PACK = 11        /* layout conversions ([B][T][F] <-> [T][F][B]) */
};
int os_profile_enable(os_ctx *ctx, int enable);
int os_profile_read(os_ctx *ctx, double *ms_sum /* host [OS_PROF_PHASES] */, int32_t *launches /* host [OS_PROF_PHASES] */);
const char *os_profile_kernel_name(const os_ctx *ctx, int phase);

/* Layout helper: [B][T][F] (the reference's per-trajectory row lists) -> [T][F][B]. */
int os_pack_stream(os_ctx *ctx, int32_t B, int32_t T, int32_t F, const float *src_btf, float *dst_tfb, void *stream);
int os_unpack_stream(os_ctx *ctx, int32_t B, int32_t T, int32_t F, const float *src_tfb, float *dst_btf, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* OPTISTATE_HIP_H */
