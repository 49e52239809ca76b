// launch.hpp -- context object and error plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/optistate_hip.h"
#include "kf_device.hpp"

struct GruPacked;   // gru_kernels.hip

int os_stack_verify(struct os_ctx *ctx, hipStream_t s, const char *what);     // gru_kernels.hip
int os_stack_pending(struct os_ctx *ctx, const char *what);

struct os_ctx {
    uint32_t magic;
    int device;
    osk::KfConst k;
    bool r_is_diagonal, q_is_diagonal;
    float *kf_qr;                        // device copy of Q (144) and R (100) for per-lane indexing (small-batch kernel)
    int rows_kernel_below;               // use the 16-lanes-per-trajectory kernel when B is below this
    int tune_sym_pre;                    // OS_KF_SYM_PRE=0: kf_run_sym_kernel without the half-step-ahead LDS pick-up of the inputs (A/B runs)
    // development knobs, read from the environment ONCE in os_create (OS_KF_ROWS_BELOW, OS_GRU_SPLIT, OS_DW_RPS, OS_SWEEP_NW)
    int tune_gru_split;                  // -1 automatic, 0 never use the eight-wave split layer kernel
    int tune_gru_wide;                   // OS_GRU_WIDE=0: small H = 128 batches stay on gru_stack_kernel (one CU per (layer, tile)) instead of gru_wide_kernel (four)
    bool wide_attr_set;
    int tune_gru_stack;                  // small batches run their layer stack as one pipelined launch (gru_stack_kernel / bwd_sweep_stack_kernel):
                                         // 1 = yes, and the call waits for the launch and reads the error word (default); 2 = yes, asynchronous (a lost
                                         // producer surfaces at the next os_gru_* call); 0 = a launch per layer.  OS_GRU_STACK / os_gru_set_stack
    uint32_t *stack_flags;               // gru_stack_kernel's progress counters [layers][tiles]
    size_t stack_flags_n;
    // error word of the progress-counter kernels: pinned host memory mapped into the device's address space.  A consumer whose
    // bounded wait expires ORs bit 0 into it (system scope) before it poisons its input with NaN; the host reads it without a
    // copy (os_stack_verify after a stacked launch, os_stack_pending at the entry of later calls); adam_kernel reads it on the
    // device and leaves the weights alone while it is set.
    int32_t *stack_err_host, *stack_err_dev;
    int32_t *stack_err_local;            // the same bit in DEVICE memory: what adam_kernel polls (422 k threads reading a word of host
                                         // memory over PCIe doubled the batch-64 training step); cleared on the stream when the error is reported
    bool stack_dirty;                    // mode 2: a stacked launch has gone out since the last os_stack_check / report
    uint32_t stack_max_polls;            // OS_STACK_DBG_POLLS (development / tests): polls before a wait gives up (default 2^22: seconds)
    int stack_dbg_drop_layer, stack_dbg_drop_step;   // OS_STACK_DBG_DROP="layer,step" (tests): that workgroup row stops publishing from that step on
    int tune_dw_dbg;                     // development: Dw3Args.dbg (OS_DW_DBG)
    int tune_gru_stage;                  // 1: large-batch H = 128 inference layers use gru_layer_stage_kernel (x tile by LDS-DMA), 0: gru_layer_kernel<2,2>
    int tune_gru_ahead;                  // 1: H = 128 small-batch layers use gru_layer_ahead_kernel (input half one step ahead), 0: split kernel
    int tune_mpc_persistent;             // os_kf_mpc_run: 1 = one persistent kernel up to 24 trajectories per CU (default), 2 = always, 0 = the per-step launch sequence
    float *mpc_hand; size_t mpc_hand_floats;   // device: hand-over record, todo lists and counters of the two-pass QP (mpc_kernels.hip launch_instances)
    int tune_mpc_cap;                    // OS_MPC_CAP: iterations after which a 16-lane row hands its problem to a wavefront of its own (default 0: never -- measured at B = 65,536: cap 8 moves 0.35 of 1.38 ms into the second launch, the sum does not change)
    int tune_mpc_quad;                   // OS_MPC_QUAD: batches of at least this many QPs with one / two stance legs run four to a wavefront
                                         // (mpc_quad.hip: 16 lanes per QP); 0 = never; default 64
    int tune_vit_mlp_fused;              // ViT block tail: 3 = projection + LayerNorm + MLP + the next block's LayerNorm / qkv in one kernel (default),
                                         // 2 = projection + LayerNorm + MLP, 1 = LayerNorm + MLP, 0 = separate launches
    int tune_dw_rps;                     // rows per dW slice
    int tune_vit_att_dma;                // 1: persistent attention workgroups with LDS-DMA K / V double buffering (OS_VIT_ATT_DMA=0: one workgroup per head)
    int tune_vit_mlp_bm;                 // rows per vit_mlp tile: 128 (one eight-wave workgroup per CU) or 64 (two four-wave workgroups) (OS_VIT_MLP_BM)
    int tune_vit_tail_split;             // 1: the last partial round of vit_mlp_kernel tiles runs as 32- / 64-row tiles (OS_VIT_TAIL_SPLIT=0: full tiles)
    int tune_dw_fused;                   // 1: W_ih and W_hh gradients of a layer in one launch (dw3_kernel) when eligible, 0: two launches,
                                         // 2: one launch only for inputs of at most 128 columns (the round-2d state)
    int tune_sweep_wr;                   // backward sweep: 32 leading k-pairs of a wave's weight chunk kept in registers (OS_SWEEP_WR=0: none)
    int tune_sweep_nw;                   // 0 automatic, 4 / 8 waves per backward-sweep workgroup
    int tune_train_overlap;              // OS_TRAIN_OVERLAP=0: weight-gradient kernels on the caller's stream (no side stream)
    char err[512];
    // GRU state (owned scratch)
    os_gru_dims gru;
    bool gru_loaded;
    float *gru_packed;        // device: weights re-packed into MFMA fragment order (the active slot of gru_slots)
    // LRU of packed images (os_gru_load_keyed): an ensemble of models alternating on one context (gru_train.py:205-217
    // `num_models`) re-selects its image instead of re-packing it on every forward
    struct GruSlot {
        uint64_t key; os_gru_dims d; const float *flat; float *packed; size_t cap; uint64_t stamp;
        float *vec; size_t vec_cap; bool vec_valid;       // transposed image for gru_vec_kernel, packed on first use
        float *bf; size_t bf_cap; int bf_spl;             // bf16 term image of gru_layer_bf16_kernel (dwords), built on first use; bf_spl = 0: stale
    } gru_slots[4];
    GruSlot *gru_slot;                   // the slot os_gru_load selected last
    int gru_split_bf16;                  // os_gru_set_split_bf16 / OS_GRU_SPLIT_BF16: 0 exact fp32 (default), 2 | 3 = bf16 terms per operand in the H = 128
                                         // large-batch layer kernel (gru_layer_bf16_kernel), | OS_GRU_SPLIT_ANY_BATCH
    bool bf16_layer_attr_set;
    int tune_gru_vec;                    // 1: B <= 4 inference runs the whole model in one single-workgroup launch (gru_vec_kernel)
    bool vec_attr_set;
    uint64_t gru_clock;
    const float *gru_flat;    // caller-owned flat weights (kept for the head / biases)
    float *gru_wide_seq; size_t gru_wide_seq_floats;   // gru_wide_kernel's exchange buffers: L x [T][B][H] (inference)
    float *gru_seq;  size_t gru_seq_floats;   // inter-layer sequences [T][H][B] x2 + h_last
    float *gru_xs;   size_t gru_xs_floats;    // SoA copy of a (B,T,I) input
    float *gru_hl;   size_t gru_hl_floats;    // SoA h_last of all layers
    float *gru_gi;   size_t gru_gi_floats;    // os_gru_forward_windows: rows . W_ih^T of the row stream [N][3H]
    bool gi_attr_set;
    float *nrm;                               // fused path: [min | 1/(max-min)] (120 floats)
    float *fused_img;                         // fused v2: per-call LDS image (weights with folded scales, k order of the register-resident h)
    bool fused2_attr_set, fusedbf_attr_set, fused3_attr_set;
    float *fused_img3;                        // fused v3 (16-trajectory tiles): its LDS image
    int tune_fused_tile;                      // OS_FUSED_TILE (development / sweeps): trajectories per workgroup of the single fused kernel
                                              // (256 | 128 | 64 | 32 | 16), 0 = chosen from the batch
    float *fused_img_bf;                      // LDS image of the opt-in split-bf16 kernel
    float *feat;     size_t feat_floats;      // fused path v0: normalised feature rows [T][I][B]
    int cu_count;
    // convex-MPC force QP (mpc_kernels.hip): weights of kalman_filter.py:64-72, constraints of force_controller.py:143-155
    double mpc_w[12], mpc_rw, mpc_mu, mpc_fzmax;
    double mass64, inertia64[3], gz64;
    float *mpc_scratch; size_t mpc_scratch_floats;
    bool fused_attr_set, sweep_attr_set, layer_attr_set, split_attr_set, ahead_attr_set, stage_attr_set, stack_attr_set;  // hipFuncSetAttribute(MaxDynamicSharedMemorySize) done on this device
    void *vit;                           // os_vit_state (vit_kernels.hip), created by os_vit_load
    void *train;                         // os_train_state (gru_train_kernels.hip), created on first use
    int bwd_mark_layer; void *bwd_mark_event;   // os_gru_backward_mark: event recorded behind this layer's weight-gradient kernel
    void *step;                          // os_step_state (kf_step.hip): pinned, device-mapped staging block of os_kf_step
    // per-kernel timing (os_profile_*): ring of event pairs
    bool prof;
    int prof_n;                          // recorded pairs
    hipEvent_t prof_ev[2 * 512];
    int prof_phase[512];
    double prof_ms[OS_PROF_PHASES];
    int prof_cnt[OS_PROF_PHASES];
    const char *prof_name[OS_PROF_PHASES];   // kernel variant most recently launched in each phase (static strings)
    uint64_t gru_generation;                 // bumped by every os_gru_load (os_gru_generation)
};

// adds the elapsed times of the recorded event pairs to the per-phase sums (synchronises on them) and empties the ring
static inline void os_prof_drain(os_ctx *ctx)
{
    for (int i = 0; i < ctx->prof_n; i++) {
        float ms = 0.f;
        if (hipEventSynchronize(ctx->prof_ev[2 * i + 1]) == hipSuccess &&
            hipEventElapsedTime(&ms, ctx->prof_ev[2 * i], ctx->prof_ev[2 * i + 1]) == hipSuccess) {
            ctx->prof_ms[ctx->prof_phase[i]] += ms;
            ctx->prof_cnt[ctx->prof_phase[i]] += 1;
        }
    }
    ctx->prof_n = 0;
}

// RAII-less helpers: bracket a kernel launch with events when profiling is on
static inline int os_prof_begin(os_ctx *ctx, int phase, hipStream_t s, const char *kernel_name = nullptr)
{
    if (kernel_name) ctx->prof_name[phase] = kernel_name;
    if (!ctx->prof) return -1;
    if (ctx->prof_n >= 512) os_prof_drain(ctx);      // ring full (per-step launch loops): fold it into the sums
    const int i = ctx->prof_n;
    if (!ctx->prof_ev[2 * i]) {
        if (hipEventCreate(&ctx->prof_ev[2 * i]) != hipSuccess || hipEventCreate(&ctx->prof_ev[2 * i + 1]) != hipSuccess)
            return -1;
    }
    ctx->prof_phase[i] = phase;
    (void)hipEventRecord(ctx->prof_ev[2 * i], s);
    return i;
}
static inline void os_prof_end(os_ctx *ctx, int slot, hipStream_t s)
{
    if (slot < 0) return;
    (void)hipEventRecord(ctx->prof_ev[2 * slot + 1], s);
    ctx->prof_n = slot + 1;
}

#define OS_MAGIC 0x4f53414du

static inline int os_fail(os_ctx *ctx, int code, const char *msg)
{
    if (ctx) snprintf(ctx->err, sizeof(ctx->err), "%s", msg);
    return code;
}

#define OS_CHECK_CTX(ctx)                                         \
    do {                                                          \
        if (!(ctx) || (ctx)->magic != OS_MAGIC) return -1;        \
    } while (0)

#define OS_HIP(ctx, expr)                                                                          \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            snprintf((ctx)->err, sizeof((ctx)->err), "%s:%d: %s", __FILE__, __LINE__,              \
                     hipGetErrorString(_e));                                                       \
            return -10;                                                                            \
        }                                                                                          \
    } while (0)

// ---- internal cross-translation-unit helpers (not part of the C ABI) ----
namespace osg { struct LayerArgs; struct WideArgs; }
size_t os_layer_packed_floats(int K, int H);
int os_ensure_scratch(os_ctx *ctx, float **buf, size_t *cap, size_t need_floats);
int os_gru_launch_layer(os_ctx *ctx, const osg::LayerArgs &a, hipStream_t s);
int os_gru_try_layer_bf16(os_ctx *ctx, const osg::LayerArgs &a, hipStream_t s, bool *done);   // gru_bf16_kernels.hip: the opt-in split-bf16 layer kernel
bool os_gru_stack_eligible(os_ctx *ctx, int B, int T, int Kfirst, int H, int nlayers);   // gru_kernels.hip: small batch, all (layer, tile) workgroups resident
int os_gru_launch_stack(os_ctx *ctx, const osg::LayerArgs *layers, int n, hipStream_t s);   // n <= 8 consecutive layers as one pipelined launch
bool os_gru_layer_takes_btf(os_ctx *ctx, int B, int T, int K, int H);   // gru_kernels.hip: the layer kernel for this shape reads (B, T, K) inputs itself
int os_gru_head_launch(os_ctx *ctx, int B, const float *top, const float *fcw, float *out, hipStream_t s);
bool os_gru_wide_eligible(os_ctx *ctx, int B, int T, int K0, int H, int n);                 // gru_wide_kernel.hip
int os_gru_launch_wide(os_ctx *ctx, osg::WideArgs &a, bool save, hipStream_t s);
void os_train_destroy(os_ctx *ctx);
void os_step_destroy(os_ctx *ctx);
void os_vit_destroy(os_ctx *ctx);
