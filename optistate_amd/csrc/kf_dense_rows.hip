// kf_dense_rows.hip -- the predict_mpc filter (kalman_filter/kalman_filter.py:140-174) in float64, 16 lanes per trajectory.
//
// The reference's real loop (`estimate_state_mpc`, data_collection/data_conversion_Kalman_to_Training.py:193-199) discretises
// with the ELEMENT-WISE exp(dt F) (kalman_filter.py:157): F_d = 1 1^T + E is dense, adds a common-mode term ~sum(P) to every
// entry of P that the update cancels again against R ~ 1e-4, and so needs float64 covariance arithmetic to hold the 1e-4 state
// bar.  One trajectory per lane (kf_run_kernel<*, DENSE>, rounds 1-4) needs P and M = F_d P, 2 x 144 doubles, in 512
// registers: 1.3-1.5 k spilled VGPRs, 3.7-5.1 KB of scratch per lane.  Here lane r of a 16-lane DPP row holds ROW r of P
// (12 doubles = 24 VGPRs) and x[r]; four trajectories per wavefront; no scratch, no LDS.
//   * every cross-lane term is a broadcast fused into a multiply-add: `v_fmac_f64_dpp acc, src row_newbcast:S, m` reads src
//     from lane S of the row inside the instruction (gfx90a+: 64-bit DPP exists for row_newbcast only, and v_fmac_f64 is the
//     one VOP2 float64 arithmetic instruction), sums over lanes are the same instruction with m = 1;
//   * M is never formed: with c = 1^T P (column sums), rho = P 1 (row sums, in-lane), sigma = 1^T P 1,
//         F_d P F_d^T = sigma 1 1^T + 1 (E c)^T + (E rho) 1^T + E P E^T,
//     and E has rows 0..5 and columns 6..11 only (E[i][6+k] = expm1(dt Rb[k][i]), E[3+i][9+i] = expm1(dt)): six column
//     sums, six row-sum broadcasts and a 6 x 6 block of E P instead of 2 x 12^3 multiply-adds;
//   * both update forms of the lane kernels: sequential (diagonal R: ten rank-1 updates, 12 fused multiply-adds per lane
//     each) and batch as the reference writes it (S = P[sel,sel] + R -> LU of S as it is, row a of the factors in the lane of measurement a ->
//     K = P[:,sel] S^-1 by forward / back substitution, one ROW of K per lane -> P -= K P[sel,:] column by column);
//   * the float32 front (odometry, z, next_state with the int64-truncation quirk) is kf_device.hpp's code, computed
//     redundantly by a trajectory's 16 lanes; the nine sincos and the nine expm1 of a step are shared (one per lane).
// DPP hazard (inline asm is invisible to hipcc's hazard recogniser): a VALU result must be two instructions old before a DPP
// operand reads it.  Every assembly statement here starts with `s_nop 1`, except the inner members of a chain whose DPP
// sources are pinned (an empty volatile asm right behind their definition) in front of the chain's first member;
// tools/isa_dpp_hazard_scan.py checks the compiled code (tests/test_isa_dpp_hazards.py).
#include "kf_device.hpp"
#include "kf_dense_rows.hpp"
#include "kf_args.hpp"
#include "launch.hpp"

namespace osk {

using namespace rows64;

// AUX: P_trace / K_gain / p_rot outputs where the pointers are set; FEAT: the normalised 60-feature row of the two-kernel
// fused path [x_post | accel | f | p_world | dp | imu] (lane r writes the r-th element of each block).
template <bool SEQ, bool AUX, bool FEAT, bool DENSE = true>
__global__ __launch_bounds__(256, 2) void kf_dense_rows_kernel(const KfRunArgs a, const float *__restrict__ qr /* Q 144 | R 100 */)
{
    const int lane = threadIdx.x & 63, r = lane & 15, grp = lane >> 4;
    const int b_raw = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4 + grp;
    const bool live = b_raw < a.B;
    const int b = live ? b_raw : a.B - 1;
    const int rr = r < 12 ? r : 11;                    // idle lanes 12-15 shadow row 11 (never broadcast from, never stored)
    const size_t B = (size_t)a.B;
    const uint32_t voff = (uint32_t)b * 4u, rowB = (uint32_t)a.B * 4u;
    const KfConst &k = a.k;
    // the measurement this lane's state row is (SEL^-1; -1: rows 3, 4 and the idle lanes own none)
    const int am = row_measurement(r);

    // Q and R as float64 in LDS: a lane re-reads ITS row of Q at the end of every predict and its row of R at the top of
    // every batch update (six / five 16-byte reads) instead of holding 24 + 20 registers across the whole step
    __shared__ __attribute__((aligned(16))) double qs[NS * NS], rs[(NM + 2) * NM];
    for (int i = threadIdx.x; i < NS * NS; i += blockDim.x) qs[i] = (double)qr[i];
    for (int i = threadIdx.x; i < (NM + 2) * NM; i += blockDim.x) rs[i] = i < NM * NM ? (double)qr[144 + i] : 0.0;     // R as it is (the reference adds it to H P H^T as given); rows 10, 11: zeros for the lanes that own no measurement
    __syncthreads();
    const int qoff = rr * NS, roff = (am >= 0 ? am : NM) * NM;

    double P[NS];
    float xr;
    {
        rsrc_t rx = make_rsrc(a.x, 12 * rowB), rP = make_rsrc(a.P, 144 * rowB);
        xr = buf_load(rx, voff + (uint32_t)rr * rowB, 0);
#pragma unroll
        for (int j = 0; j < NS; j++) P[j] = (double)buf_load(rP, voff + (uint32_t)(rr * NS + j) * rowB, 0);
    }
    double one = 1.0;
    asm volatile("" : "+v"(one));                      // a register operand for the DPP sums
    const double ed = expm1((double)k.dt);
    int status = 0;
    StepIn in;
    float bref[3];
    load_step(a, 0, voff, rowB, in);
    if (DENSE) {
        rsrc_t rb = make_rsrc(a.body_ref, 12 * rowB);
#pragma unroll
        for (int i = 0; i < 3; i++) bref[i] = buf_load(rb, voff, i * rowB);
    } else bref[0] = bref[1] = bref[2] = 0.f;
    for (int t = 0; t < a.T; t++) {
        // (an offset the optimiser cannot see through: the LDS reads of Q / R stay inside the step instead of being hoisted
        // into 44 loop-invariant registers)
        int oz = 0;
        asm volatile("" : "+v"(oz));
        const double *qrow = qs + qoff + oz, *rrow_b = rs + roff + oz, *rdiag = rs + oz;
        // ---- the prior state, replicated per lane; z, covariance predict, next_state (kf_dense_rows.hpp) ----
        float x[NS], z[NM], pw[12];
        gather_state(xr, x);
        status |= front_row<DENSE>(x, xr, P, in, bref, k, ed, qrow, one, r, z, pw);
        float xn = x[0];
#pragma unroll
        for (int i = 1; i < NS; i++) xn = (rr == i) ? x[i] : xn;
        if (a.p_rot_out && live && r < 12) {
            float pv = pw[0];
#pragma unroll
            for (int i = 1; i < 12; i++) pv = (r == i) ? pw[i] : pv;
            a.p_rot_out[((size_t)t * 12 + r) * B + b] = pv;
        }
        if (FEAT && live && r < 12) {
            float fv = in.f[0], pv = pw[0], dv = in.dp[0];
#pragma unroll
            for (int i = 1; i < 12; i++) { fv = (r == i) ? in.f[i] : fv; pv = (r == i) ? pw[i] : pv; dv = (r == i) ? in.dp[i] : dv; }
            float *fo = a.feat_out + (size_t)t * a.feat_I * B + b;
            const float *mm = a.minmax;
            auto put = [&](int j, float v) { __builtin_nontemporal_store((v - mm[j]) / (mm[60 + j] - mm[j]), fo + (size_t)j * B); };
            put(18 + r, fv); put(30 + r, pv); put(42 + r, dv);
            if (r < 6) {
                float iv = in.imu[0];
#pragma unroll
                for (int i = 1; i < 6; i++) iv = (r == i) ? in.imu[i] : iv;
                put(54 + r, iv);
                put(12 + r, a.accel[((size_t)t * 6 + r) * B + b]);
            }
        }
        // the next step's inputs, underneath the update (the variants with P_trace / K_gain outputs request them behind those:
        // 46 more live registers across the reductions did not fit the 256 of two waves per SIMD)
        const int tn = (t + 1 < a.T) ? t + 1 : t;
        auto prefetch = [&]() {
            load_step(a, tn, voff, rowB, in);
            if (DENSE) {
                rsrc_t rb = make_rsrc(a.body_ref + (size_t)tn * 12 * B, 12 * rowB);
#pragma unroll
                for (int i = 0; i < 3; i++) bref[i] = buf_load(rb, voff, i * rowB);
            }
        };
        if (!AUX) prefetch();
        // ---- update (kalman_filter.py:164-174) ----
        double xd = (double)xn;
        float kgain = 0.f;
        if (SEQ) {
            double rrow[NM];                           // diag R
#pragma unroll
            for (int q = 0; q < NM; q++) rrow[q] = rdiag[q * NM + q];
            status |= update_seq_row(xd, P, z, rrow);
            if (AUX && a.kgain_out) kgain = kgain_posterior_rows(P, rrow);
        } else {
            double K[NM], rrow[NM];
#pragma unroll
            for (int q = 0; q < NM; q++) rrow[q] = rrow_b[q];
            if (AUX) status |= update_batch_row<true>(xd, P, z, rrow, K, am, one, &kgain);
            else status |= update_batch_row(xd, P, z, rrow, K, am);
        }
        xr = (float)xd;
        if (!(xr * 0.f == 0.f)) status |= 2;
        if (live && r < 12) a.x_out[((size_t)t * 12 + r) * B + b] = xr;
        if (FEAT && live && r < 12)
            __builtin_nontemporal_store((xr - a.minmax[r]) / (a.minmax[60 + r] - a.minmax[r]), a.feat_out + ((size_t)t * a.feat_I + r) * B + b);
        if (AUX) {
            if (a.ptrace_out) {
                const float tr = ptrace_rows(P, one);
                if (live && r == 0) a.ptrace_out[(size_t)t * B + b] = tr;
            }
            if (a.kgain_out && live && r == 0) a.kgain_out[(size_t)t * B + b] = kgain;
            prefetch();
        }
    }
    // ---- final state: lane r writes x[r] and row r of P; the status word is OR-reduced over the 16 lanes ----
    status |= __shfl_xor(status, 1, 64); status |= __shfl_xor(status, 2, 64);
    status |= __shfl_xor(status, 4, 64); status |= __shfl_xor(status, 8, 64);
    if (live && r < 12) {
        // (opaque copies of the trajectory and row indices: the addresses of these stores are formed HERE, not hoisted above the T loop
        // and kept -- or spilled: 12 B of scratch in the <batch, AUX, dense> instance -- across it)
        int bb = b, rq = r;
        asm volatile("" : "+v"(bb), "+v"(rq));
        a.x[(size_t)rq * B + bb] = xr;
#pragma unroll
        for (int j = 0; j < NS; j++) a.P[(size_t)(rq * NS + j) * B + bb] = (float)P[j];
        if (r == 0) a.status[bb] = status;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The batched single-step pieces of the C-ABI (os_kf_predict / os_kf_update: Engine.kf_predict / kf_update, the Q / R fitter of
// pipeline.fit_noise_covariances) on the same row layout, float64 inside whatever the storage type of P (PT = float or, with
// OS_KF_P_FLOAT64, double).  They replace the one-trajectory-per-lane kf_predict_kernel / kf_update_kernel (190-616 spilled VGPRs).
// ---------------------------------------------------------------------------------------------------------------------------
template <bool DENSE, typename PT>
__global__ __launch_bounds__(256, 2) void kf_predict_rows_kernel(int B_, float *p, const float *f, const float *body_ref, float *x, PT *Pm,
                                                                float *ptrace_out, const KfConst k, const float *__restrict__ qr)
{
    const int lane = threadIdx.x & 63, r = lane & 15, grp = lane >> 4;
    const int b_raw = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4 + grp;
    const bool live = b_raw < B_;
    const int b = live ? b_raw : B_ - 1;
    const int rr = r < 12 ? r : 11;
    const size_t B = (size_t)B_;
    float xr = x[(size_t)rr * B + b];
    double P[NS], qrow[NS];
#pragma unroll
    for (int j = 0; j < NS; j++) { P[j] = (double)Pm[(size_t)(rr * NS + j) * B + b]; qrow[j] = (double)qr[rr * NS + j]; }
    StepIn in;
#pragma unroll
    for (int i = 0; i < 12; i++) { in.p[i] = p[(size_t)i * B + b]; in.f[i] = f[(size_t)i * B + b]; in.dp[i] = 0.f; }
#pragma unroll
    for (int i = 0; i < 6; i++) in.imu[i] = 0.f;
    in.contact = 0u;
    float bref[3] = {0.f, 0.f, 0.f};
    if (DENSE) {
#pragma unroll
        for (int i = 0; i < 3; i++) bref[i] = body_ref[(size_t)i * B + b];
    }
    double one = 1.0;
    asm volatile("" : "+v"(one));
    float xs[NS], pw[12];
    gather_state(xr, xs);
    // sincos shared over the lanes: prior attitude on lanes 0..2, body_ref attitude on lanes 6..8
    float sv, cv;
    sincos_f32(r < 3 ? xr : r == 6 ? bref[0] : r == 7 ? bref[1] : bref[2], &sv, &cv);
    const Rot rot = rotation_sc(bc32<0>(sv), bc32<0>(cv), bc32<1>(sv), bc32<1>(cv), bc32<2>(sv), bc32<2>(cv));
    double cf[6];
    if (DENSE) {
        const Rot rbr = rotation_sc(bc32<6>(sv), bc32<6>(cv), bc32<7>(sv), bc32<7>(cv), bc32<8>(sv), bc32<8>(cv));
        float rbn = rbr.m[0];
#pragma unroll
        for (int n = 1; n < 9; n++) rbn = (r == n) ? rbr.m[3 * (n % 3) + n / 3] : rbn;
        const double en = expm1((double)k.dt * (double)rbn), ed = expm1((double)k.dt);
        double e[9];
        e[0] = bc64<0>(en); e[1] = bc64<1>(en); e[2] = bc64<2>(en); e[3] = bc64<3>(en); e[4] = bc64<4>(en);
        e[5] = bc64<5>(en); e[6] = bc64<6>(en); e[7] = bc64<7>(en); e[8] = bc64<8>(en);
#pragma unroll
        for (int kk = 0; kk < 3; kk++) {
            cf[kk] = r == 0 ? e[kk] : r == 1 ? e[3 + kk] : r == 2 ? e[6 + kk] : 0.0;
            cf[3 + kk] = (r == 3 + kk) ? ed : 0.0;
        }
        predict_dense_row(P, qrow, e, ed, cf, one);
    } else {
        double g[9];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int kk = 0; kk < 3; kk++) g[3 * i + kk] = (double)k.dt * (double)rot.m[3 * kk + i];
#pragma unroll
        for (int kk = 0; kk < 3; kk++) {
            cf[kk] = r == 0 ? g[kk] : r == 1 ? g[3 + kk] : r == 2 ? g[6 + kk] : 0.0;
            cf[3 + kk] = (r == 3 + kk) ? (double)k.dt : 0.0;
        }
        predict_struct_row(P, qrow, g, (double)k.dt, cf);
    }
    dynamics(xs, rot, in.p, in.f, pw, k);
    float xn = xs[0], pv = pw[0];
#pragma unroll
    for (int i = 1; i < NS; i++) { xn = (rr == i) ? xs[i] : xn; pv = (rr == i) ? pw[i] : pv; }
    const float tr = ptrace_out ? ptrace_rows(P, one) : 0.f;            // (wave-uniform branch: every lane takes part)
    if (live && r < 12) {
        x[(size_t)r * B + b] = xn;
        p[(size_t)r * B + b] = pv;                                      // the foot positions rotated in place (force_controller.py:274-277)
#pragma unroll
        for (int j = 0; j < NS; j++) Pm[(size_t)(r * NS + j) * B + b] = (PT)P[j];
        if (r == 0 && ptrace_out) ptrace_out[b] = tr;
    }
}

template <bool SEQ, typename PT>
__global__ __launch_bounds__(256, 2) void kf_update_rows_kernel(int B_, const float *z, float *x, PT *Pm, PT *K_out, float *ptrace_out,
                                                               float *kgain_out, int32_t *status, const KfConst k, const float *__restrict__ qr)
{
    const int lane = threadIdx.x & 63, r = lane & 15, grp = lane >> 4;
    const int b_raw = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4 + grp;
    const bool live = b_raw < B_;
    const int b = live ? b_raw : B_ - 1;
    const int rr = r < 12 ? r : 11, am = row_measurement(r);
    const size_t B = (size_t)B_;
    double P[NS];
#pragma unroll
    for (int j = 0; j < NS; j++) P[j] = (double)Pm[(size_t)(rr * NS + j) * B + b];
    float zz[NM];
#pragma unroll
    for (int i = 0; i < NM; i++) zz[i] = z[(size_t)i * B + b];
    double xd = (double)x[(size_t)rr * B + b], one = 1.0;
    asm volatile("" : "+v"(one));
    double K[NM], rrow[NM];
    int st;
    float kg;
    if (SEQ) {
#pragma unroll
        for (int q = 0; q < NM; q++) rrow[q] = (double)k.R[q * NM + q];
        st = update_seq_row(xd, P, zz, rrow);
        kg = kgain_posterior_rows(P, rrow);
        // K = P+ H^T R^-1 (diagonal R): the batch gain from the posterior, this lane's row
        for_sel([&](auto A, auto SA) { K[decltype(A)::v] = P[decltype(SA)::v] * rcp64(rrow[decltype(A)::v]); });
    } else {
#pragma unroll
        for (int q = 0; q < NM; q++)                    // the lane's row of R, as given
            rrow[q] = am >= 0 ? (double)qr[144 + am * NM + q] : 0.0;
        st = update_batch_row<true>(xd, P, zz, rrow, K, am, one, &kg);
    }
    const float xo = (float)xd;
    if (!(xo * 0.f == 0.f)) st |= 2;
    st |= __shfl_xor(st, 1, 64); st |= __shfl_xor(st, 2, 64); st |= __shfl_xor(st, 4, 64); st |= __shfl_xor(st, 8, 64);
    const float tr = ptrace_out ? ptrace_rows(P, one) : 0.f;
    if (live && r < 12) {
        x[(size_t)r * B + b] = xo;
#pragma unroll
        for (int j = 0; j < NS; j++) Pm[(size_t)(r * NS + j) * B + b] = (PT)P[j];
        if (K_out) {
#pragma unroll
            for (int a = 0; a < NM; a++) K_out[(size_t)(r * NM + a) * B + b] = (PT)K[a];
        }
        if (r == 0) {
            if (ptrace_out) ptrace_out[b] = tr;
            if (kgain_out) kgain_out[b] = kg;
            if (status) status[b] = st;
        }
    }
}

hipError_t launch_kf_predict_rows(int B, float *p, const float *f, const float *body_ref, float *x, void *P, bool p64, float *ptrace_out,
                                  const KfConst &k, const float *qr, bool dense, hipStream_t s)
{
    dim3 grid((B + 15) / 16), block(256);
    if (p64) {
        if (dense) hipLaunchKernelGGL((kf_predict_rows_kernel<true, double>), grid, block, 0, s, B, p, f, body_ref, x, (double *)P, ptrace_out, k, qr);
        else hipLaunchKernelGGL((kf_predict_rows_kernel<false, double>), grid, block, 0, s, B, p, f, body_ref, x, (double *)P, ptrace_out, k, qr);
    } else {
        if (dense) hipLaunchKernelGGL((kf_predict_rows_kernel<true, float>), grid, block, 0, s, B, p, f, body_ref, x, (float *)P, ptrace_out, k, qr);
        else hipLaunchKernelGGL((kf_predict_rows_kernel<false, float>), grid, block, 0, s, B, p, f, body_ref, x, (float *)P, ptrace_out, k, qr);
    }
    return hipGetLastError();
}

hipError_t launch_kf_update_rows(int B, const float *z, float *x, void *P, void *K_out, bool p64, float *ptrace_out, float *kgain_out,
                                 int32_t *status, const KfConst &k, const float *qr, bool seq, hipStream_t s)
{
    dim3 grid((B + 15) / 16), block(256);
    if (p64) {
        if (seq) hipLaunchKernelGGL((kf_update_rows_kernel<true, double>), grid, block, 0, s, B, z, x, (double *)P, (double *)K_out, ptrace_out, kgain_out, status, k, qr);
        else hipLaunchKernelGGL((kf_update_rows_kernel<false, double>), grid, block, 0, s, B, z, x, (double *)P, (double *)K_out, ptrace_out, kgain_out, status, k, qr);
    } else {
        if (seq) hipLaunchKernelGGL((kf_update_rows_kernel<true, float>), grid, block, 0, s, B, z, x, (float *)P, (float *)K_out, ptrace_out, kgain_out, status, k, qr);
        else hipLaunchKernelGGL((kf_update_rows_kernel<false, float>), grid, block, 0, s, B, z, x, (float *)P, (float *)K_out, ptrace_out, kgain_out, status, k, qr);
    }
    return hipGetLastError();
}

// host side: picks the instantiation (kf_kernels.hip: os_kf_run_impl)
hipError_t launch_kf_dense_rows(const KfRunArgs &a, const float *qr, bool seq, bool feat, bool aux, hipStream_t s, bool dense)
{
    dim3 grid((a.B + 15) / 16), block(256);
#define OSD_GO(SEQ_, AUX_, FEAT_, DENSE_) hipLaunchKernelGGL((kf_dense_rows_kernel<SEQ_, AUX_, FEAT_, DENSE_>), grid, block, 0, s, a, qr)
    if (!dense && !seq) {
        // the predict(p, f) covariance with the BATCH update (a non-diagonal R, or the caller asked for the form that builds K):
        // float64 here -- the float32 one-trajectory-per-lane kernel this replaces lost the filter on ill-conditioned S (fitted
        // noise, cond ~1e6, plus flight phases: state errors of 1e-3 .. 1e+1 after 40-100 steps; found by tools/fuzz_kf.py, round 5)
        if (feat) OSD_GO(false, false, true, false);
        else if (aux) OSD_GO(false, true, false, false);
        else OSD_GO(false, false, false, false);
    } else if (!dense) {
        // ... and with the sequential update where the full (not symmetrised) P is wanted: a Q that is not symmetric, or the caller's flag
        if (feat) OSD_GO(true, false, true, false);
        else if (aux) OSD_GO(true, true, false, false);
        else OSD_GO(true, false, false, false);
    } else if (seq) {
        if (feat) OSD_GO(true, false, true, true);
        else if (aux) OSD_GO(true, true, false, true);
        else OSD_GO(true, false, false, true);
    } else {
        if (feat) OSD_GO(false, false, true, true);
        else if (aux) OSD_GO(false, true, false, true);
        else OSD_GO(false, false, false, true);
    }
#undef OSD_GO
    return hipGetLastError();
}

}  // namespace osk
