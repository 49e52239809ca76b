// capi.hip -- context lifetime and configuration entry points of liboptistate_hip.so.
#include "launch.hpp"

#include <math.h>
#include <stdlib.h>

extern "C" {

int os_version(void) { return 1; }
const char *os_build_arch(void) { return "gfx950"; }
#ifndef OS_BUILD_ID
#define OS_BUILD_ID "unstamped"
#endif
const char *os_build_id(void) { return OS_BUILD_ID; }

int os_create(const os_kf_config *cfg, os_ctx **out)
{
    if (!cfg || !out) return -2;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return -11;   // no HIP device: fail loudly
    if (cfg->device < 0 || cfg->device >= ndev) return -12;
    os_ctx *c = (os_ctx *)calloc(1, sizeof(os_ctx));
    if (!c) return -13;
    c->magic = OS_MAGIC;
    c->device = cfg->device;
    c->k.dt = cfg->dt;
    c->k.inv_mass = 1.0f / cfg->mass;
    c->k.gz = cfg->gz;
    for (int i = 0; i < 3; i++) c->k.inv_inertia[i] = 1.0f / cfg->inertia[i];
    c->mass64 = (double)cfg->mass; c->gz64 = (double)cfg->gz;
    for (int i = 0; i < 3; i++) c->inertia64[i] = (double)cfg->inertia[i];
    {   // StanceController weights: Q = diag(10,10,10,100,100,100,1,1,5,1,1,1), R = 1e-6 I (kalman_filter.py:64-70)
        static const double wq[12] = {10, 10, 10, 100, 100, 100, 1, 1, 5, 1, 1, 1};
        for (int i = 0; i < 12; i++) c->mpc_w[i] = wq[i];
        c->mpc_rw = 1e-6; c->mpc_mu = 0.6; c->mpc_fzmax = 150.0;      // force_controller.py:147-149
    }
    // defaults: settings.py:28-31
    static const float qd[12] = {0.01f, 0.01f, 0.01f, 0.01f, 0.0001f, 0.01f, 0.01f, 0.01f, 0.01f, 0.01f, 0.01f, 0.0001f};
    for (int i = 0; i < 12; i++) c->k.Q[i * 12 + i] = qd[i];
    for (int i = 0; i < 10; i++) c->k.R[i * 10 + i] = 0.01f;
    c->r_is_diagonal = true;
    c->q_is_diagonal = true;
    // measured (tools/rows_crossover.py, round 3): the rows kernel holds 2,048 waves = 8,192 trajectories at two waves per SIMD
    // (4.35e9 steps/s there; lane kernel 3.67e9) and needs a second, half-empty round beyond that (10,240: 3.3e9 against 4.6e9)
    c->rows_kernel_below = 8193;
    {
        const char *e = getenv("OS_KF_ROWS_BELOW");      // tuning knobs (development): read once, here
        if (e) c->rows_kernel_below = atoi(e);
        e = getenv("OS_KF_SYM_PRE"); c->tune_sym_pre = e ? atoi(e) : 1;
        e = getenv("OS_GRU_SPLIT"); c->tune_gru_split = e ? (atoi(e) != 0 ? -1 : 0) : -1;
        e = getenv("OS_GRU_AHEAD"); c->tune_gru_ahead = e ? atoi(e) : 1;
        e = getenv("OS_GRU_STAGE"); c->tune_gru_stage = e ? atoi(e) : 1;
        e = getenv("OS_GRU_SPLIT_BF16"); c->gru_split_bf16 = (e && (atoi(e) == 2 || atoi(e) == 3)) ? atoi(e) : 0;
        e = getenv("OS_GRU_STACK"); c->tune_gru_stack = e ? atoi(e) : 1;
        e = getenv("OS_GRU_VEC"); c->tune_gru_vec = e ? atoi(e) : 1;
        e = getenv("OS_STACK_DBG_POLLS"); c->stack_max_polls = e && atoi(e) > 0 ? (uint32_t)atoi(e) : (1u << 22);
        c->stack_dbg_drop_layer = c->stack_dbg_drop_step = -1;
        e = getenv("OS_STACK_DBG_DROP");
        if (e) (void)sscanf(e, "%d,%d", &c->stack_dbg_drop_layer, &c->stack_dbg_drop_step);
        e = getenv("OS_FUSED_TILE"); c->tune_fused_tile = e ? atoi(e) : 0;
        e = getenv("OS_MPC_PERSISTENT"); c->tune_mpc_persistent = e ? atoi(e) : 1;
        e = getenv("OS_MPC_QUAD"); c->tune_mpc_quad = e ? atoi(e) : 64;
        e = getenv("OS_MPC_CAP"); c->tune_mpc_cap = e ? atoi(e) : 0;
        e = getenv("OS_VIT_MLP_FUSED"); c->tune_vit_mlp_fused = e ? atoi(e) : 3;
        // rows per weight-gradient slice; unset (0) = about 32 slices, between one 32-row tile and 512 rows (gru_train_kernels.hip)
        e = getenv("OS_DW_RPS"); c->tune_dw_rps = e && atoi(e) > 0 ? atoi(e) : 0;
        e = getenv("OS_DW_DBG"); c->tune_dw_dbg = e ? atoi(e) : 0;
        e = getenv("OS_DW_FUSED"); c->tune_dw_fused = e ? atoi(e) : 1;
        e = getenv("OS_VIT_TAIL_SPLIT"); c->tune_vit_tail_split = e ? atoi(e) : 1;
        e = getenv("OS_VIT_ATT_DMA"); c->tune_vit_att_dma = e ? atoi(e) : 1;
        e = getenv("OS_VIT_MLP_BM"); c->tune_vit_mlp_bm = e ? atoi(e) : 64;
        e = getenv("OS_GRU_WIDE"); c->tune_gru_wide = e ? atoi(e) : 1; c->wide_attr_set = false;
        e = getenv("OS_SWEEP_WR"); c->tune_sweep_wr = e ? atoi(e) : 32;
        e = getenv("OS_SWEEP_NW"); c->tune_sweep_nw = e ? atoi(e) : 0;
        // weight-gradient kernels on a side stream underneath the next layer's sweep: unset = where CUs are idle and the launches are long
        // (gru_train_kernels.hip); 8,192 windows: measured, no gain (both kernels are bound by the shared fp32 pipe); 0 / 1 force it off / on
        e = getenv("OS_TRAIN_OVERLAP"); c->tune_train_overlap = e ? atoi(e) : -1;
    }
    if (hipSetDevice(cfg->device) != hipSuccess || hipMalloc((void **)&c->kf_qr, 244 * sizeof(float)) != hipSuccess ||
        hipMemcpy(c->kf_qr, c->k.Q, 144 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->kf_qr + 144, c->k.R, 100 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        free(c);
        return -10;
    }
    if (hipHostMalloc((void **)&c->stack_err_host, sizeof(int32_t), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&c->stack_err_dev, c->stack_err_host, 0) != hipSuccess) {
        (void)hipFree(c->kf_qr);
        free(c);
        return -10;
    }
    *c->stack_err_host = 0;
    if (hipMalloc((void **)&c->stack_err_local, sizeof(int32_t)) != hipSuccess || hipMemset(c->stack_err_local, 0, sizeof(int32_t)) != hipSuccess) {
        (void)hipFree(c->kf_qr); (void)hipHostFree(c->stack_err_host);
        free(c);
        return -10;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) { (void)hipFree(c->kf_qr); (void)hipHostFree(c->stack_err_host); free(c); return -10; }
    c->cu_count = prop.multiProcessorCount;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        // built for gfx950 only; refuse to pretend on anything else
        (void)hipFree(c->kf_qr);
        free(c);
        return -14;
    }
    *out = c;
    return 0;
}

void os_destroy(os_ctx *ctx)
{
    if (!ctx || ctx->magic != OS_MAGIC) return;
    (void)hipSetDevice(ctx->device);
    os_train_destroy(ctx);
    os_step_destroy(ctx);
    os_vit_destroy(ctx);
    for (auto &sl : ctx->gru_slots)
    {
        if (sl.packed) (void)hipFree(sl.packed);
        if (sl.vec) (void)hipFree(sl.vec);
        if (sl.bf) (void)hipFree(sl.bf);
    }
    if (ctx->stack_flags) (void)hipFree(ctx->stack_flags);
    if (ctx->stack_err_host) (void)hipHostFree(ctx->stack_err_host);
    if (ctx->stack_err_local) (void)hipFree(ctx->stack_err_local);
    float *bufs[] = {ctx->gru_seq, ctx->gru_wide_seq, ctx->gru_xs, ctx->gru_hl, ctx->gru_gi, ctx->feat, ctx->nrm, ctx->kf_qr, ctx->mpc_scratch, ctx->fused_img, ctx->fused_img_bf, ctx->fused_img3, ctx->mpc_hand};
    for (float *b : bufs)
        if (b) (void)hipFree(b);
    for (int i = 0; i < 2 * 512; i++)
        if (ctx->prof_ev[i]) (void)hipEventDestroy(ctx->prof_ev[i]);
    ctx->magic = 0;
    free(ctx);
}

const char *os_last_error(const os_ctx *ctx) { return (ctx && ctx->magic == OS_MAGIC) ? ctx->err : "invalid context"; }

int os_kf_set_noise(os_ctx *ctx, const float *Q_host, const float *R_host)
{
    OS_CHECK_CTX(ctx);
    if (!Q_host || !R_host) return os_fail(ctx, -2, "os_kf_set_noise: null pointer");
    memcpy(ctx->k.Q, Q_host, sizeof(float) * 144);
    memcpy(ctx->k.R, R_host, sizeof(float) * 100);
    bool diag = true;
    for (int a = 0; a < 10; a++)
        for (int b = 0; b < 10; b++)
            if (a != b && R_host[a * 10 + b] != 0.0f) diag = false;
    ctx->r_is_diagonal = diag;
    bool qd = true;
    for (int a = 0; a < 12; a++)
        for (int b = 0; b < 12; b++)
            if (a != b && Q_host[a * 12 + b] != 0.0f) qd = false;
    ctx->q_is_diagonal = qd;
    OS_HIP(ctx, hipSetDevice(ctx->device));
    OS_HIP(ctx, hipMemcpy(ctx->kf_qr, ctx->k.Q, 144 * sizeof(float), hipMemcpyHostToDevice));
    OS_HIP(ctx, hipMemcpy(ctx->kf_qr + 144, ctx->k.R, 100 * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

int os_profile_enable(os_ctx *ctx, int enable)
{
    OS_CHECK_CTX(ctx);
    OS_HIP(ctx, hipSetDevice(ctx->device));
    os_prof_drain(ctx);
    for (int i = 0; i < OS_PROF_PHASES; i++) { ctx->prof_ms[i] = 0.0; ctx->prof_cnt[i] = 0; }
    ctx->prof = enable != 0;
    return 0;
}

const char *os_profile_kernel_name(const os_ctx *ctx, int phase)
{
    if (!ctx || ctx->magic != OS_MAGIC || phase < 0 || phase >= OS_PROF_PHASES || !ctx->prof_name[phase]) return "";
    return ctx->prof_name[phase];
}

uint64_t os_gru_generation(const os_ctx *ctx) { return (ctx && ctx->magic == OS_MAGIC) ? ctx->gru_generation : 0; }

int os_profile_read(os_ctx *ctx, double *ms_sum, int32_t *launches)
{
    OS_CHECK_CTX(ctx);
    if (!ms_sum || !launches) return os_fail(ctx, -2, "os_profile_read: null pointer");
    OS_HIP(ctx, hipSetDevice(ctx->device));
    os_prof_drain(ctx);
    for (int i = 0; i < OS_PROF_PHASES; i++) {
        ms_sum[i] = ctx->prof_ms[i]; launches[i] = ctx->prof_cnt[i];
        ctx->prof_ms[i] = 0.0; ctx->prof_cnt[i] = 0;
    }
    return 0;
}

}  // extern "C"
