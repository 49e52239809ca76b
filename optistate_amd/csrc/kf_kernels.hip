// kf_kernels.hip -- batched Kalman filter kernels for gfx950: one trajectory per lane, the whole T-step
// recurrence inside one launch, x/P resident in VGPRs, inputs streamed coalesced from [T][field][B].
#include <type_traits>

#include "kf_device.hpp"
#include "kf_args.hpp"
#include "launch.hpp"

namespace osk {

// 64-lane workgroups: lanes are independent, so small groups give the dispatcher the most freedom to
// spread waves over the 1024 SIMDs.  State: 156 VGPRs (x, P) + 43 input dwords + update temporaries.
// Feature row layout (data_collection/data_conversion_Kalman_to_Training.py:245-254):
//   [0:12) x_post | [12:18) accel | [18:30) f | [30:42) p_world | [42:54) dp | [54:60) imu ; then (v - min)/(max - min).
__device__ __forceinline__ void store_feat(rsrc_t rf, const float *mm, uint32_t voff, uint32_t rowB, int j, float v)
{
    const float mn = mm[j], mx = mm[60 + j];
    buf_store_nt(rf, voff, j * rowB, (v - mn) / (mx - mn));
}

// (kf_run_kernel -- one trajectory per lane with the full 12 x 12 P in 144 registers, the fallback for a non-symmetric Q, a
// non-diagonal R or the predict_mpc covariance -- lived here until round 5: 0.1-1.5 k spilled VGPRs per instance, and its float32
// batch update lost the filter on ill-conditioned runs.  Every one of its cases runs on the float64 row layout now:
// kf_dense_rows.hip.)

// Development builds only (-DOS_ROWS_TS: tools/rows_ts.sh, kf_run_rows2_kernel; -DOS_SYM_TS: kf_run_sym_kernel): shader-clock stamps at the
// phase boundaries of a step, summed over the steps by lane 0 of workgroup 0 (written to kgain_out as 8 x uint64 / printed).
#if defined(OS_ROWS_TS) || defined(OS_SYM_TS)
#define OS_TS_DECL unsigned long long ts_prev = 0, ts_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define OS_TS(i)                                                                   \
    {                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                         \
        const unsigned long long now = __builtin_readcyclecounter();               \
        if ((i) > 0) ts_sum[i] += now - ts_prev;                                   \
        ts_prev = now;                                                             \
        __builtin_amdgcn_sched_barrier(0);                                         \
    }
#else
#define OS_TS_DECL
#define OS_TS(i)
#endif

// Fast path: symmetric P storage (78 VGPRs), sequential update, predict(p,f) covariance.  OUT: 0 plain, 1 P_trace, 2 features.
// status bit 3: the caller's P0 is not symmetric (checked once per launch against the lower triangle, 66 extra loads per
// trajectory): the symmetric-storage kernels would silently run a different filter than the reference, which never
// symmetrises P (kalman_filter/kalman_filter.py:172)
template <int OUT, bool QDIAG, bool PRE = false>
__device__ __forceinline__ void kf_run_sym_body(const KfRunArgs &a, const KfConst &kc, const int b_raw)
{
    // lanes past the end of the batch stay alive (the 16-byte input DMA needs every lane of the wave) as shadows of the last
    // trajectory with their stores masked
    const bool live = b_raw < a.B;
    const int b = live ? b_raw : a.B - 1;
    const size_t B = (size_t)a.B;
    const uint32_t voff = (uint32_t)b * 4u, rowB = (uint32_t)a.B * 4u;
    const uint32_t vo4 = step_dma_offset(threadIdx.x & 63, b_raw - (int)(threadIdx.x & 63), rowB);
    // stores of the shadow lanes: an offset no descriptor covers (dropped by the range check) instead of a branch around every
    // group of stores -- branches cut the straight-line step into blocks and cost ~300 instructions of register shuffling
    const uint32_t vst = live ? voff : 0x7ffffff0u;
    f2 X[6];                       // the state as pairs (x[2i], x[2i+1])
    f2 U[NU];
    int status = 0;
    {
        rsrc_t rx = make_rsrc(a.x, 12 * rowB), rP = make_rsrc(a.P, 144 * rowB);
#pragma unroll
        for (int i = 0; i < 6; i++) X[i] = (f2){buf_load(rx, voff, 2 * i * rowB), buf_load(rx, voff, (2 * i + 1) * rowB)};
        status = p0_asymmetry_status([&](int e) { return buf_load(rP, voff, (uint32_t)e * rowB); });
        sym_load(U, [&](int e) { return buf_load(rP, voff, (uint32_t)e * rowB); });
    }
    // Step inputs by LDS-DMA.  General form: double buffer, step t + 1 requested at the top of step t, picked up at the top of
    // step t + 1.  PRE (plain runs: x_out the only output, so the number of stores per step is known): THREE stages, step
    // t + 2 requested at the top of step t, and step t + 1's LDS reads issued in the middle of step t (asynchronously, into
    // registers of their own: lds_issue_step) so that they complete underneath the measurement update -- the top of a step
    // used to wait out the 22 dependent LDS reads with nothing else to issue (~0.9 k of a step's 6.3 k cycles, PMC).
    constexpr int NSTAGE = PRE ? 3 : 2;
    __shared__ float stage[NSTAGE][STEP_DWORDS * 64];
    const int lane = threadIdx.x & 63;
    StepInP in;
    StepRaw raw;
    float smin = 3.0e38f;                  // the smallest innovation variance of the run (status bit 0)
    load_step_dma(a, 0, vo4, rowB, stage[0]);
    if (PRE) {
        load_step_dma(a, a.T > 1 ? 1 : 0, vo4, rowB, stage[1]);
        __builtin_amdgcn_s_waitcnt(0x0f7c);                            // vmcnt(12): step 0 has landed, step 1's twelve loads may be in flight
        __builtin_amdgcn_wave_barrier();
        lds_issue_step(stage[0], lane, raw);
    } else {
        // step 0's twelve DMA loads are the only VMEM operations in flight here, so the loop's vmcnt(12) would return before any
        // of them has landed: wait for them explicitly once (until round 4 only hipcc's own vmcnt(0) in front of the visible
        // ds_reads of read_step_lds_p ordered the first pick-up)
        __builtin_amdgcn_s_waitcnt(0x0f70);                            // vmcnt(0)
        __builtin_amdgcn_wave_barrier();
    }
#ifdef OS_SYM_TS
    OS_TS_DECL
#define OS_STS(i) OS_TS(i)
#else
#define OS_STS(i)
#endif
    for (int t = 0; t < a.T; t++) {
        OS_STS(0)
        if (PRE) {
            lds_fence_step(raw, in);
            const int tn = t + 2 < a.T ? t + 2 : a.T - 1;
            load_step_dma(a, tn, vo4, rowB, stage[(t + 2) % 3]);
        } else {
            // vmcnt(12), t >= 1: step t has landed (requested a whole step ago).  Not vmcnt(0): stores count too and complete in
            // issue order with the loads, and the twelve x_out stores of step t - 1 were issued a few instructions ago --
            // everything older than the last twelve operations includes every DMA load.  (t = 0: see the wait before the loop.)
            __builtin_amdgcn_s_waitcnt(0x0f7c);
            __builtin_amdgcn_wave_barrier();
            read_step_lds_p(stage[t & 1], lane, in);
            load_step_dma(a, (t + 1 < a.T) ? t + 1 : t, vo4, rowB, stage[(t + 1) & 1]);
        }
        OS_STS(1)                                       // input pick-up + next request
        float z[NM];
        f2 PW[2][3];
        float g9[9];
        status |= kf_step_inputs_sym(X, in, kc, z, PW, g9);
        OS_STS(2)                                       // rotations, odometry, next_state
        cov_predict_sym_blk<QDIAG>(U, g9, kc);
        OS_STS(3)                                       // covariance predict
        auto pw = [&](int i) { return PW[(i / 3) >> 1][i % 3][(i / 3) & 1]; };          // world-frame foot position 3 leg + component
        auto lg = [](const f2 (*v)[3], int i) { return v[(i / 3) >> 1][i % 3][(i / 3) & 1]; };
        rsrc_t rfeat;
        if (OUT == 2) {
            rfeat = make_rsrc(a.feat_out + (size_t)t * a.feat_I * B, (uint32_t)a.feat_I * rowB);
            rsrc_t ra = make_rsrc(a.accel + (size_t)t * 6 * B, 6 * rowB);
#pragma unroll
            for (int i = 0; i < 6; i++) store_feat(rfeat, a.minmax, vst, rowB, 12 + i, buf_load_nt(ra, voff, i * rowB));
#pragma unroll
            for (int i = 0; i < 12; i++) {
                store_feat(rfeat, a.minmax, vst, rowB, 18 + i, lg(in.f, i));
                store_feat(rfeat, a.minmax, vst, rowB, 30 + i, pw(i));
                store_feat(rfeat, a.minmax, vst, rowB, 42 + i, lg(in.dp, i));
            }
#pragma unroll
            for (int i = 0; i < 6; i++) store_feat(rfeat, a.minmax, vst, rowB, 54 + i, in.imu[i]);
        }
        if (a.p_rot_out) {
            rsrc_t ro = make_rsrc(a.p_rot_out + (size_t)t * 12 * B, 12 * rowB);
#pragma unroll
            for (int i = 0; i < 12; i++) buf_store_nt(ro, vst, i * rowB, pw(i));
        }
        if (PRE) {
            // step t + 1 (requested at the top of step t - 1): younger than it are step t - 1's twelve stores and step t + 2's
            // twelve loads; vmcnt(12) waits for the stores too (issued most of a step ago)
            __builtin_amdgcn_s_waitcnt(0x0f7c);
            __builtin_amdgcn_wave_barrier();
            lds_issue_step(stage[(t + 1) % 3], lane, raw);
        }
        OS_STS(4)                                       // optional outputs, LDS reads of the next step issued
        smin = fminf(smin, update_sequential_sym(X, U, z, kc));            // non-finite states stay non-finite: checked once after the loop
        OS_STS(5)                                       // ten measurement updates
        {
            rsrc_t ro = make_rsrc(a.x_out + (size_t)t * 12 * B, 12 * rowB);
#pragma unroll
            for (int i = 0; i < NS; i++) buf_store_nt(ro, vst, i * rowB, X[i / 2][i & 1]);
        }
        OS_STS(6)                                       // x_out stores
        if (OUT == 2) {
#pragma unroll
            for (int i = 0; i < NS; i++) store_feat(rfeat, a.minmax, vst, rowB, i, X[i / 2][i & 1]);
        }
        if (OUT == 1 && a.ptrace_out && live) a.ptrace_out[(size_t)t * B + b] = trace_sym(U);
        if (OUT == 1 && a.kgain_out && live) a.kgain_out[(size_t)t * B + b] = kgain_from_posterior_sym(U, kc);
    }
    {
        rsrc_t rx = make_rsrc(a.x, 12 * rowB), rP = make_rsrc(a.P, 144 * rowB);
#pragma unroll
        for (int i = 0; i < NS; i++) buf_store(rx, vst, i * rowB, X[i / 2][i & 1]);
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int j = 0; j < NS; j++) buf_store(rP, vst, (i * NS + j) * rowB, OSK_SYM(U, i, j));
    }
#ifdef OS_SYM_TS
    if (blockIdx.x == 0 && threadIdx.x == 0)
        printf("kf_run_sym cycles per step: pick-up %llu | front %llu | predict %llu | outputs+issue %llu | update %llu | stores %llu\n",
               ts_sum[1] / a.T, ts_sum[2] / a.T, ts_sum[3] / a.T, ts_sum[4] / a.T, ts_sum[5] / a.T, ts_sum[6] / a.T);
#endif
#undef OS_STS
    if (live) a.status[b] = status | singular_status(smin) | finite_status_p(X);
}

template <int OUT, bool QDIAG, bool PRE = false>
__global__ __launch_bounds__(64, 1) void kf_run_sym_kernel(const KfRunArgs a)
{
    kf_run_sym_body<OUT, QDIAG, PRE>(a, a.k, blockIdx.x * 64 + threadIdx.x);
}

// Per-trajectory diagonal noise (os_kf_run_noise): each lane overwrites the diagonals of its own copy of the constants
// with its trajectory's q_diag / r_diag (22 more VGPRs; every index is a compile-time constant, so the copy is registers).
template <int OUT>
__global__ __launch_bounds__(64, 1) void kf_run_sym_noise_kernel(const KfRunArgs a)
{
    const int b_raw = blockIdx.x * 64 + threadIdx.x;
    const int b = b_raw < a.B ? b_raw : a.B - 1;          // lanes past the end shadow the last trajectory (stores masked in the body)
    const uint32_t voff = (uint32_t)b * 4u, rowB = (uint32_t)a.B * 4u;
    KfConst kc = a.k;
    rsrc_t rq = make_rsrc(a.q_diag, 12 * rowB), rr = make_rsrc(a.r_diag, 10 * rowB);
#pragma unroll
    for (int i = 0; i < NS; i++) kc.Q[i * NS + i] = buf_load(rq, voff, i * rowB);
#pragma unroll
    for (int i = 0; i < NM; i++) kc.R[i * NM + i] = buf_load(rr, voff, i * rowB);
    kf_run_sym_body<OUT, true>(a, kc, b_raw);
}

// ---- single pieces for the drop-in Kalman_Filter class (B is tiny there; latency-bound by design) ----

__global__ void kf_odom_kernel(int B, const float *p, const float *dp, const uint32_t *contact, const float *imu,
                               float *z)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    StepIn in;
#pragma unroll
    for (int i = 0; i < 12; i++) { in.p[i] = p[(size_t)i * B + b]; in.dp[i] = dp[(size_t)i * B + b]; in.f[i] = 0.f; }
#pragma unroll
    for (int i = 0; i < 6; i++) in.imu[i] = imu[(size_t)i * B + b];
    in.contact = contact[b];
    float zz[NM];
    measurement(in, zz);
#pragma unroll
    for (int i = 0; i < NM; i++) z[(size_t)i * B + b] = zz[i];
}

// (kf_predict / kf_update: kf_dense_rows.hip, 16 lanes per trajectory, float64 inside)

// ---- layout helpers: [B][T][F] <-> [T][F][B] through a 32x32 LDS tile (coalesced on both sides) ----
__global__ void pack_btf_to_tfb(int B, int TF, const float *__restrict__ src, float *__restrict__ dst)
{
    __shared__ float tile[32][33];
    // src viewed as [B][TF], dst as [TF][B]
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;   // bx over TF, by over B
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        int bb = by + r, c = bx + threadIdx.x;
        if (bb < B && c < TF) tile[r][threadIdx.x] = src[(size_t)bb * TF + c];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        int c = bx + r, bb = by + threadIdx.x;
        if (bb < B && c < TF) dst[(size_t)c * B + bb] = tile[threadIdx.x][r];
    }
}

// src [B][T][F] -> rows [row0, row0 + F) of dst [T][F_total][B]; blockIdx.z = time step
__global__ void pack_btf_rows(int B, int T, int F, int F_total, int row0, const float *__restrict__ src, float *__restrict__ dst)
{
    __shared__ float tile[32][33];
    const int fx = blockIdx.x * 32, by = blockIdx.y * 32, t = blockIdx.z;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int bb = by + r, c = fx + threadIdx.x;
        if (bb < B && c < F) tile[r][threadIdx.x] = src[((size_t)bb * T + t) * F + c];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int c = fx + r, bb = by + threadIdx.x;
        if (bb < B && c < F) dst[((size_t)t * F_total + row0 + c) * B + bb] = tile[threadIdx.x][r];
    }
}

}  // namespace osk

using namespace osk;


int os_kf_run_wave(os_ctx *ctx, const KfRunArgs &a, hipStream_t s);      // kf_step.hip: one trajectory per wavefront, P in LDS

// Shared by os_kf_run and os_fused_run (v0: Kalman kernel emits normalised feature rows for the GRU kernels).
int os_kf_run_impl(os_ctx *ctx, KfRunArgs &a, uint32_t flags, hipStream_t s)
{
    const bool seq = flags & OS_KF_SEQUENTIAL_UPDATE, dense = flags & OS_KF_DENSE_FD;
    if (flags & OS_KF_WAVE_PER_TRAJECTORY) {
        // the north_star's literal layout, kept for measurement: float64 batch update (LU of S) on one wavefront per trajectory
        if (dense || a.q_diag || a.feat_out) return os_fail(ctx, -3, "os_kf_run: OS_KF_WAVE_PER_TRAJECTORY runs predict(p, f) + update with the context-wide noise only");
        a.k = ctx->k;
        return os_kf_run_wave(ctx, a, s);
    }
    if (dense && !a.body_ref) return os_fail(ctx, -2, "os_kf_run: OS_KF_DENSE_FD needs body_ref");
    if (seq && !ctx->r_is_diagonal) return os_fail(ctx, -3, "os_kf_run: sequential update needs a diagonal R");
    if ((size_t)a.B * 144 * 4 >= 0xffffffffull) return os_fail(ctx, -2, "os_kf_run: B too large for 32-bit buffer offsets");
    a.k = ctx->k;
#ifdef OS_ROWS_TS
    const bool aux = a.ptrace_out != nullptr, feat = a.feat_out != nullptr;
#else
    const bool aux = a.ptrace_out || a.kgain_out, feat = a.feat_out != nullptr;
#endif
    if (feat && aux) return os_fail(ctx, -3, "os_kf_run: feature emission and P_trace/K_gain outputs are exclusive");
    hipError_t e;
    const bool noise = a.q_diag != nullptr;
    if (noise && (!seq || dense || !a.r_diag))
        return os_fail(ctx, -3, "os_kf_run_noise: per-trajectory noise needs the sequential update and the predict(p,f) covariance");
    // (the rows kernel carries a step's position in a 32-bit SGPR offset: T * 48 B bytes must fit)
    const bool use_rows = !noise && seq && !dense && !(flags & OS_KF_LANE_PER_TRAJECTORY) &&
                          a.B < ctx->rows_kernel_below && ctx->kf_qr && (uint64_t)a.T * 48ull * (uint64_t)a.B < 0xffffffffull;
    const bool use_sym = !use_rows && seq && !dense && ((flags & OS_KF_SYMMETRIC_P) || noise);
    const char *kname = noise ? "kf_run_sym_noise_kernel" : use_rows ? "kf_run_rows2_kernel" : use_sym ? "kf_run_sym_kernel"
                        : dense ? (seq ? "kf_dense_rows_kernel<SEQ>" : "kf_dense_rows_kernel<BATCH>")
                                : (seq ? "kf_dense_rows_kernel<SEQ,predict(p,f)>" : "kf_dense_rows_kernel<BATCH,predict(p,f)>");
    const int slot = os_prof_begin(ctx, OS_PHASE_KF, s, kname);
    if (dense) {
        // predict_mpc covariance (element-wise exp(dt F)): float64, 16 lanes per trajectory, every batch size (kf_dense_rows.hip)
        e = launch_kf_dense_rows(a, (const float *)ctx->kf_qr, seq, feat, aux, s);
    } else if (noise) {
        dim3 grid((a.B + 63) / 64), block(64);
        if (aux) hipLaunchKernelGGL((kf_run_sym_noise_kernel<1>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((kf_run_sym_noise_kernel<0>), grid, block, 0, s, a);
        e = hipGetLastError();
    } else if (use_rows) {
        // small batch: 16 lanes per trajectory so that every SIMD gets a wave
        dim3 grid((a.B + 15) / 16), block(256);                       // 4 waves x 4 trajectories per workgroup
        e = launch_kf_rows2(a, (const float *)ctx->kf_qr, feat, aux, s);
    } else if (use_sym) {
        dim3 grid((a.B + 63) / 64), block(64);
        const bool qd = ctx->q_is_diagonal;
#define OS_SYM(OUT)                                                                                        \
    do {                                                                                                   \
        if (qd) hipLaunchKernelGGL((kf_run_sym_kernel<OUT, true>), grid, block, 0, s, a);                  \
        else hipLaunchKernelGGL((kf_run_sym_kernel<OUT, false>), grid, block, 0, s, a);                    \
    } while (0)
        if (feat) OS_SYM(2);
        else if (aux) OS_SYM(1);
        else if (a.p_rot_out || ctx->tune_sym_pre == 0) OS_SYM(0);
        else if (qd) hipLaunchKernelGGL((kf_run_sym_kernel<0, true, true>), grid, block, 0, s, a);      // plain run: inputs picked up half a step ahead
        else hipLaunchKernelGGL((kf_run_sym_kernel<0, false, true>), grid, block, 0, s, a);
#undef OS_SYM
        e = hipGetLastError();
    } else   // full P wanted (a Q that is not symmetric, a non-diagonal R, OS_KF_LANE_PER_TRAJECTORY without the symmetric flag): float64 rows
        e = launch_kf_dense_rows(a, (const float *)ctx->kf_qr, seq, feat, aux, s, false);
    os_prof_end(ctx, slot, s);
    OS_HIP(ctx, e);
    return 0;
}

extern "C" {

int os_kf_run(os_ctx *ctx, int32_t B, int32_t T, const float *p, const float *f, const float *dp, const float *imu,
              const uint32_t *contact, const float *body_ref, float *x, float *P, float *x_out, float *p_rot_out,
              float *ptrace_out, float *kgain_out, int32_t *status, uint32_t flags, void *stream)
{
    OS_CHECK_CTX(ctx);
    if (B <= 0 || T <= 0) return os_fail(ctx, -2, "os_kf_run: B and T must be positive");
    if (!p || !f || !dp || !imu || !contact || !x || !P || !x_out || !status)
        return os_fail(ctx, -2, "os_kf_run: null required pointer");
    OS_HIP(ctx, hipSetDevice(ctx->device));
    KfRunArgs a;
    a.B = B; a.T = T; a.p = p; a.f = f; a.dp = dp; a.imu = imu; a.contact = contact; a.body_ref = body_ref;
    a.x = x; a.P = P; a.x_out = x_out; a.p_rot_out = p_rot_out; a.ptrace_out = ptrace_out; a.kgain_out = kgain_out;
    a.status = status; a.accel = nullptr; a.minmax = nullptr; a.feat_out = nullptr; a.feat_I = 0;
    return os_kf_run_impl(ctx, a, flags, (hipStream_t)stream);
}

int os_kf_run_noise(os_ctx *ctx, int32_t B, int32_t T, const float *p, const float *f, const float *dp, const float *imu,
                    const uint32_t *contact, float *x, float *P, const float *q_diag, const float *r_diag, float *x_out,
                    float *p_rot_out, float *ptrace_out, int32_t *status, uint32_t flags, void *stream)
{
    OS_CHECK_CTX(ctx);
    if (B <= 0 || T <= 0) return os_fail(ctx, -2, "os_kf_run_noise: B and T must be positive");
    if (!p || !f || !dp || !imu || !contact || !x || !P || !x_out || !status || !q_diag || !r_diag)
        return os_fail(ctx, -2, "os_kf_run_noise: null required pointer");
    if (flags & OS_KF_DENSE_FD) return os_fail(ctx, -3, "os_kf_run_noise: the predict_mpc covariance is not supported here");
    OS_HIP(ctx, hipSetDevice(ctx->device));
    KfRunArgs a;
    a.B = B; a.T = T; a.p = p; a.f = f; a.dp = dp; a.imu = imu; a.contact = contact; a.body_ref = nullptr;
    a.x = x; a.P = P; a.x_out = x_out; a.p_rot_out = p_rot_out; a.ptrace_out = ptrace_out; a.kgain_out = nullptr;
    a.status = status; a.accel = nullptr; a.minmax = nullptr; a.feat_out = nullptr; a.feat_I = 0;
    a.q_diag = q_diag; a.r_diag = r_diag;
    // R is diagonal by construction here whatever the context-wide R is
    const bool saved = ctx->r_is_diagonal;
    ctx->r_is_diagonal = true;
    const int rc = os_kf_run_impl(ctx, a, (flags | OS_KF_SEQUENTIAL_UPDATE | OS_KF_SYMMETRIC_P | OS_KF_LANE_PER_TRAJECTORY), (hipStream_t)stream);
    ctx->r_is_diagonal = saved;
    return rc;
}

int os_kf_odom(os_ctx *ctx, int32_t B, const float *p, const float *dp, const uint32_t *contact, const float *imu,
               float *z, void *stream)
{
    OS_CHECK_CTX(ctx);
    if (B <= 0 || !p || !dp || !contact || !imu || !z) return os_fail(ctx, -2, "os_kf_odom: bad argument");
    OS_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(kf_odom_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, B, p, dp, contact, imu, z);
    OS_HIP(ctx, hipGetLastError());
    return 0;
}

int os_kf_predict(os_ctx *ctx, int32_t B, float *p, const float *f, const float *body_ref, float *x, float *P,
                  float *ptrace_out, uint32_t flags, void *stream)
{
    OS_CHECK_CTX(ctx);
    if (B <= 0 || !p || !f || !x || !P) return os_fail(ctx, -2, "os_kf_predict: bad argument");
    const bool dense = flags & OS_KF_DENSE_FD;
    if (dense && !body_ref) return os_fail(ctx, -2, "os_kf_predict: OS_KF_DENSE_FD needs body_ref");
    OS_HIP(ctx, hipSetDevice(ctx->device));
    OS_HIP(ctx, launch_kf_predict_rows(B, p, f, body_ref, x, P, (flags & OS_KF_P_FLOAT64) != 0, ptrace_out, ctx->k, (const float *)ctx->kf_qr, dense,
                                        (hipStream_t)stream));
    return 0;
}

int os_kf_update(os_ctx *ctx, int32_t B, const float *z, float *x, float *P, float *K_out, float *ptrace_out,
                 float *kgain_out, int32_t *status, uint32_t flags, void *stream)
{
    OS_CHECK_CTX(ctx);
    if (B <= 0 || !z || !x || !P) return os_fail(ctx, -2, "os_kf_update: bad argument");
    const bool seq = flags & OS_KF_SEQUENTIAL_UPDATE;
    if (seq && !ctx->r_is_diagonal) return os_fail(ctx, -3, "os_kf_update: sequential update needs a diagonal R");
    OS_HIP(ctx, hipSetDevice(ctx->device));
    OS_HIP(ctx, launch_kf_update_rows(B, z, x, P, K_out, (flags & OS_KF_P_FLOAT64) != 0, ptrace_out, kgain_out, status, ctx->k, (const float *)ctx->kf_qr, seq,
                                       (hipStream_t)stream));
    return 0;
}

int os_pack_stream(os_ctx *ctx, int32_t B, int32_t T, int32_t F, const float *src, float *dst, void *stream)
{
    OS_CHECK_CTX(ctx);
    if (B <= 0 || T <= 0 || F <= 0 || !src || !dst) return os_fail(ctx, -2, "os_pack_stream: bad argument");
    OS_HIP(ctx, hipSetDevice(ctx->device));
    const int TF = T * F;
    dim3 grid((TF + 31) / 32, (B + 31) / 32), block(32, 8);
    hipLaunchKernelGGL(pack_btf_to_tfb, grid, block, 0, (hipStream_t)stream, B, TF, src, dst);
    OS_HIP(ctx, hipGetLastError());
    return 0;
}

int os_pack_stream_rows(os_ctx *ctx, int32_t B, int32_t T, int32_t F, const float *src, float *dst, int32_t F_total, int32_t row0,
                        void *stream)
{
    OS_CHECK_CTX(ctx);
    if (B <= 0 || T <= 0 || F <= 0 || !src || !dst || row0 < 0 || row0 + F > F_total)
        return os_fail(ctx, -2, "os_pack_stream_rows: bad argument");
    OS_HIP(ctx, hipSetDevice(ctx->device));
    dim3 grid((F + 31) / 32, (B + 31) / 32, T), block(32, 8);
    hipLaunchKernelGGL(pack_btf_rows, grid, block, 0, (hipStream_t)stream, B, T, F, F_total, row0, src, dst);
    OS_HIP(ctx, hipGetLastError());
    return 0;
}

int os_unpack_stream(os_ctx *ctx, int32_t B, int32_t T, int32_t F, const float *src, float *dst, void *stream)
{
    OS_CHECK_CTX(ctx);
    if (B <= 0 || T <= 0 || F <= 0 || !src || !dst) return os_fail(ctx, -2, "os_unpack_stream: bad argument");
    OS_HIP(ctx, hipSetDevice(ctx->device));
    // [TF][B] -> [B][TF] is the same transpose with the roles of the two extents swapped
    const int TF = T * F;
    dim3 grid((B + 31) / 32, (TF + 31) / 32), block(32, 8);
    hipLaunchKernelGGL(pack_btf_to_tfb, grid, block, 0, (hipStream_t)stream, TF, B, src, dst);
    OS_HIP(ctx, hipGetLastError());
    return 0;
}

}  // extern "C"
