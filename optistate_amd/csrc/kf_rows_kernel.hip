// kf_rows_kernel.hip -- the small-batch Kalman kernel: 16 lanes per trajectory, one lane per row of P (BASELINE configs[1]:
// 4096 trajectories = 1024 waves = one wave per SIMD, so a step's time is its instruction count).  Its own translation unit
// because it is compiled with -fno-slp-vectorize (build.py EXTRA_FLAGS): hipcc's SLP vectoriser pairs the scalar 3x3 products
// of this kernel and pays two v_mov_b32 per pair (11 instructions per step more than the scalar code); the lane-per-trajectory
// kernels of kf_kernels.hip are hand-packed and still gain from it on their cold paths.
#include "kf_device.hpp"
#include "kf_args.hpp"
#include "launch.hpp"

namespace osk {

// Development builds only (-DOS_ROWS_TS: tools/rows_ts.sh, kf_run_rows2_kernel; -DOS_SYM_TS: kf_run_sym_kernel): shader-clock stamps at the
// phase boundaries of a step, summed over the steps by lane 0 of workgroup 0 (written to kgain_out as 8 x uint64 / printed).
#if defined(OS_ROWS_TS)
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

template <int SRC>
__device__ __forceinline__ float row_bcast(float v)
{
    // value of lane SRC of this lane's 16-lane row (DPP row_share)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + SRC, 0xf, 0xf, true));
}

// ---------------------------------------------------------------------------------------------------------------
// kf_run_rows2_kernel -- 16 lanes per trajectory (four trajectories per wave): lane r holds row r of P and x[r] (round 3; the
// round-2 kernel with the same layout was deleted in round 5: never launched outside A/B runs).
// At B = 4096 there is exactly one wave per SIMD and each wave is a chain of T dependent steps, so a step costs what the wave
// issues: 409 VALU instructions (PMC) and ~2.1 k cycles (round 2: ~930 and 4.7 k; profiles/r03_rows2_timestamps.md has the
// build-by-build table).  What is where:
//   * every row broadcast is FUSED into the multiply-add that consumes it: `v_fmac_f32_dpp acc, src row_newbcast:S, m` reads
//     src from lane S of the 16-lane row inside the instruction (a rank-1 update of a row is 12 instructions); the predict
//     needs no ds_bpermute (row r + 6 arrives by `row_shl:6`) and runs IN PLACE (its source rows carry zero coefficients);
//   * the lanes share the input-dependent work instead of repeating it: one LEG per lane (leg = lane & 3; contact decode,
//     stance / swing selects, force rotation, torque), the sums over the legs by two quad-permute DPP adds; both rotations
//     (prior attitude | IMU attitude) side by side in register pairs;
//   * each lane integrates ITS component of next_state and holds ITS measurement through weights that are one-hot in the
//     lane's row (no select chains: a chain costs an SGPR mask pair per comparison, and hipcc spilled SGPRs into VGPR lanes);
//   * the filter runs on e = x - z: the state update is the thirteenth DPP multiply-add of a measurement's chain; the ten
//     chains are ONE generated inline-assembly statement (kf_rows_chain.inc) -- hipcc pads every hand-over between its own
//     instructions and an assembly statement with an s_nop, and its order is not ours;
//   * inputs by five LDS-DMA instructions per wave two steps ahead (constant descriptors, the step offset in the VGPR, stage
//     addresses rotating in SGPRs), picked up by inline-assembly LDS reads half a step early: an LDS read hipcc can see is
//     ordered behind EVERY outstanding LDS-DMA (s_waitcnt vmcnt(0)).
// DPP hazards (inline asm is invisible to hipcc's hazard recogniser): a VALU result must be two instructions old before a
// DPP operand reads it, an EXEC write five.  Every DPP source of the assembly here was written by the previous chain (twelve or
// more instructions earlier), or is guarded by an s_nop whose operands pin the producers in front of it.  All 64 lanes stay
// active through the loop (quad_perm and row reads must not meet a disabled lane); lanes past the batch shadow the last trajectory.
// ---------------------------------------------------------------------------------------------------------------
template <int SRC>
__device__ __forceinline__ void fmac_bcast(float &acc, float src, float m)     // acc += (src of lane SRC of this 16-lane row) * m
{
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(src), "v"(m), "n"(SRC));
}
template <int N>
__device__ __forceinline__ void fmac_shl(float &acc, float src, float m)       // acc += (src of lane + N, 0 past the row's end) * m
{
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shl:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(src), "v"(m), "n"(N));
}

template <bool AUX, bool FEAT, bool PROT>
__global__ __launch_bounds__(256, 2) void kf_run_rows2_kernel(const KfRunArgs a, const float *__restrict__ qmat)
{
    const int lane = threadIdx.x & 63, r = lane & 15, grp = lane >> 4, wv = threadIdx.x >> 6;
    // XCD-aware order: consecutive workgroup ids go round-robin to the eight XCDs, each with its own L2, and two workgroups
    // (32 trajectories) share every 128-byte line of an input row -- so XCD x takes a CONTIGUOUS range of trajectory blocks
    // (measured: 2.46x the algorithmic read traffic with the plain order, every line fetched by two L2s)
    const int nblk = gridDim.x, q8 = nblk >> 3, r8 = nblk & 7, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int lblk = xcd * q8 + (xcd < r8 ? xcd : r8) + slot;
    const int first = (lblk * (blockDim.x >> 6) + wv) * 4;              // this wave's first trajectory
    const int b_raw = first + grp;
    const bool live = b_raw < a.B;
    const int b = live ? b_raw : a.B - 1;
    const int rr = r < 12 ? r : 11;                    // idle lanes 12-15 shadow row 11 (never broadcast from, never stored)
    const size_t B = (size_t)a.B;
    const uint32_t voff = (uint32_t)b * 4u, rowB = (uint32_t)a.B * 4u;
    const KfConst &k = a.k;

    float Prow[NS], qrow[NS], xr;
    {
        // (the row is per lane: in the VGPR offset -- as an SGPR offset hipcc wraps every load in a waterfall loop)
        rsrc_t rx = make_rsrc(a.x, 12 * rowB), rP = make_rsrc(a.P, 144 * rowB);
        xr = buf_load(rx, voff + (uint32_t)rr * rowB, 0);
#pragma unroll
        for (int j = 0; j < NS; j++) {
            Prow[j] = buf_load(rP, voff + (uint32_t)(rr * NS + j) * rowB, 0);
            qrow[j] = qmat[rr * NS + j];
        }
    }
    float svmin = 3.0e38f;                             // the smallest innovation variance of the run
    // Per-lane constants (one-hot in the lane's row r, scaled): each lane integrates ITS component of next_state and holds ITS
    // measurement, by multiply-adds with these weights instead of selects (a select chain costs an SGPR mask pair per
    // comparison, and hipcc kept spilling SGPRs in this loop).  Opaque to the optimiser, or it rebuilds them from masks per step.
    const KfConst &kk = a.k;
    float e0 = r == 0 ? kk.dt : 0.f, e1 = r == 1 ? kk.dt : 0.f, e2 = r == 2 ? kk.dt : 0.f;          // rows 0..2: theta
    float cd = (r >= 3 && r < 6) ? kk.dt : 0.f;                                                  // rows 3..5: position += dt * velocity
    float wa0 = r == 6 ? kk.dt : 0.f, wa1 = r == 7 ? kk.dt : 0.f, wa2 = r == 8 ? kk.dt : 0.f;    // rows 6..8: omega += dt * aw
    float wf0 = r == 9 ? kk.dt * kk.inv_mass : 0.f, wf1 = r == 10 ? kk.dt * kk.inv_mass : 0.f, wf2 = rr == 11 ? kk.dt * kk.inv_mass : 0.f;
    float wg = rr == 11 ? kk.dt * kk.gz : 0.f;                                                   // (rr: the idle lanes shadow row 11)
    float zi = (r < 3 || (r >= 6 && r < 9)) ? 1.f : 0.f, z5 = r == 5 ? 1.f : 0.f;                // measurement held by the lane
    float z9 = r == 9 ? 1.f : 0.f, z10 = r == 10 ? 1.f : 0.f, z11 = rr == 11 ? 1.f : 0.f;
    asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(cd), "+v"(wa0), "+v"(wa1), "+v"(wa2), "+v"(wf0), "+v"(wf1), "+v"(wf2), "+v"(wg));
    asm volatile("" : "+v"(zi), "+v"(z5), "+v"(z9), "+v"(z10), "+v"(z11));
    float Rd[NM];                                      // diag R in registers: the DPP add that forms S = P[s][s] + R takes no SGPR
#pragma unroll
    for (int i = 0; i < NM; i++) { Rd[i] = kk.R[i * NM + i]; asm volatile("" : "+v"(Rd[i])); }
    // step inputs: three LDS stages per wave, step t + 2 requested at the top of step t (rows_*: kf_args.hpp).  The stage
    // addresses live in SGPRs and rotate; the descriptors are loop constants.
    __shared__ __attribute__((aligned(16))) float stage_all[4][3][ROWS_STAGE];
    const uint32_t stage0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float *)stage_all[wv][0]);
    uint32_t st_cur = stage0, st_nxt = stage0 + 4u * ROWS_STAGE, st_nn = stage0 + 8u * ROWS_STAGE;
    const RowsLane rd = rows_lane(grp, r, rr);
    uint32_t sh8 = 8u * (uint32_t)(r & 3);                              // the lane's leg in the contact word
    float rw0 = rr % 3 == 0 ? 1.f : 0.f, rw1 = rr % 3 == 1 ? 1.f : 0.f, rw2 = rr % 3 == 2 ? 1.f : 0.f;    // PROT: row rr % 3 of R
    asm volatile("" : "+v"(sh8), "+v"(rw0), "+v"(rw1), "+v"(rw2));
    const RowsDma dma = rows_dma_setup(lane, first, a.B);
    const RowsSrc src = rows_src(a, rowB);
    // the plain variant issues exactly one store per step (x_out), so its waits can be counted (loads and stores retire in
    // issue order).  With optional outputs the store count is not a compile-time constant: wait for everything at the top of a
    // step (one step of latency hiding instead of two).
    constexpr bool plain = !AUX && !FEAT && !PROT;
    OS_TS_DECL
    int edge = 0;                                      // status bit 4: int64-truncation knife edge (trunc_block_f64), rare path only
    // (the prologue's loads retire here: otherwise hipcc re-checks them with a dozen s_waitcnt in every iteration)
    __builtin_amdgcn_s_waitcnt(0x0070);
#pragma unroll
    for (int j = 0; j < NS; j++) asm volatile("" : "+v"(qrow[j]), "+v"(Prow[j]));
    rows_dma_request(src, 0, dma, rowB, st_cur);
    rows_dma_request(src, a.T > 1 ? 1 : 0, dma, rowB, st_nxt);
    // Plain runs pick step t + 1's inputs up in the MIDDLE of step t (the reads complete underneath the measurement updates,
    // the top of a step only takes the registers over): `nx` travels across the loop edge.
    RowsRaw nx;
    if (plain) {
        __builtin_amdgcn_s_waitcnt(0x0f75);                                 // vmcnt(5): step 0 has landed, step 1's five loads may be in flight
        __builtin_amdgcn_wave_barrier();
        rows_issue<FEAT, PROT>(st_cur, rd, nx);
    }
    for (int t = 0; t < a.T; t++) {
        OS_TS(0)
        RowsRaw in;
        {
            const int tn = t + 2 < a.T ? t + 2 : a.T - 1;
            if (plain) {
                rows_fence<FEAT, PROT>(nx);
                in = nx;
                rows_dma_request(src, (uint32_t)tn, dma, rowB, st_nn);
            } else {
                // optional outputs: the number of stores per step is not a compile-time constant -- wait for everything at the
                // top (one step of latency hiding) and read here
                __builtin_amdgcn_s_waitcnt(0x0f70);                         // vmcnt(0)
                __builtin_amdgcn_wave_barrier();
                rows_issue<FEAT, PROT>(st_cur, rd, in);
                rows_dma_request(src, (uint32_t)tn, dma, rowB, st_nn);
                rows_fence<FEAT, PROT>(in);
            }
        }
        OS_TS(1)                                        // wait + LDS reads + next request
        // ---- both rotations (every lane), then ONE LEG per lane (leg = lane & 3) and quad sums over the legs ----
        const float th[3] = {row_bcast<0>(xr), row_bcast<1>(xr), row_bcast<2>(xr)};      // the prior attitude
        const float im[3] = {in.i4[0], in.i4[1], in.i4[2]};
        f2 Rp[9];
        rotation2(th, im, Rp);
        float R0[9];
#pragma unroll
        for (int i = 0; i < 9; i++) R0[i] = Rp[i][0];
        // get_odom (kalman_filter/kalman_filter.py:79-100): stance legs vote on v_xy and height, swing legs on v_z; selects,
        // not 0/1 weights (a NaN in an entry the reference never reads must stay out)
        const uint32_t cb = __builtin_amdgcn_ubfe(__builtin_bit_cast(uint32_t, in.c), sh8, 8u);
        const bool st = cb == 1u, sw = cb == 0u;
        float sum_c = (float)cb, vx = st ? in.dxy[0] : 0.f, vy = st ? in.dxy[1] : 0.f, vz = sw ? in.dz : 0.f, hz = st ? in.pz : 0.f;
        // next_state's input half (misc/force_controller.py:269-291).  The reference forms sum(R p x f) in the world frame and
        // applies I_hat^-1 = R diag(1/I) R^T; R^T (R p x f) = p x (R^T f) exactly (a rotation preserves the cross product), so the
        // lane rotates its leg's FORCE into the body frame and takes the torque there: 9 + 6 multiply-adds instead of 9 + 6 + 9.
        const float fb0 = fmaf(R0[6], in.fz, fmaf(R0[3], in.fxy[1], R0[0] * in.fxy[0]));
        const float fb1 = fmaf(R0[7], in.fz, fmaf(R0[4], in.fxy[1], R0[1] * in.fxy[0]));
        const float fb2 = fmaf(R0[8], in.fz, fmaf(R0[5], in.fxy[1], R0[2] * in.fxy[0]));
        float tau0 = fmaf(-in.pz, fb1, in.pxy[1] * fb2), tau1 = fmaf(-in.pxy[0], fb2, in.pz * fb0), tau2 = fmaf(-in.pxy[1], fb0, in.pxy[0] * fb1);
        float fs0 = in.fxy[0], fs1 = in.fxy[1], fs2 = in.fz;
        quad_sum5(sum_c, vx, vy, vz, hz);
        quad_sum6(tau0, tau1, tau2, fs0, fs1, fs2);
        // no stance leg -> odom = 0 (:97-98).  sum_c is 1, 2, 3 or 4 here: v_rcp_f32 (1 ulp) instead of an IEEE division
        const float inv = (sum_c != 0.f) ? __builtin_amdgcn_rcpf(sum_c) : 0.f;
        const float bx = -vx * inv, by = -vy * inv, bz = -vz * inv;
        const float zh = -hz * inv;
        const float zv0 = Rp[0][1] * bx + Rp[1][1] * by + Rp[2][1] * bz;                   // body velocity to world: the IMU rotation
        const float zv1 = Rp[3][1] * bx + Rp[4][1] * by + Rp[5][1] * bz;
        const float zv2 = Rp[6][1] * bx + Rp[7][1] * by + Rp[8][1] * bz;
        const float zr = fmaf(z11, zv2, fmaf(z10, zv1, fmaf(z9, zv0, fmaf(z5, zh, zi * in.il))));
        // body-frame torque (summed over the legs above), scaled by 1/I, back to world
        const float tb0 = tau0 * k.inv_inertia[0], tb1 = tau1 * k.inv_inertia[1], tb2 = tau2 * k.inv_inertia[2];
        const float aw0 = R0[0] * tb0 + R0[1] * tb1 + R0[2] * tb2;
        const float aw1 = R0[3] * tb0 + R0[4] * tb1 + R0[5] * tb2;
        const float aw2 = R0[6] * tb0 + R0[7] * tb1 + R0[8] * tb2;
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < 9; i++) amax = fmaxf(amax, fabsf(R0[i]));
        OS_TS(2)                                        // rotations + odometry + torque / force sums
        // ---- covariance predict, row-parallel: M = F_d P (rows: lane r needs rows 6..8 or row r + 6), then P' = M F_d^T + Q (local) ----
        // F_d[0:3, 6:9] = dt R^T: lane r < 3 needs row r of it = dt * column r of R (e0..e2 carry the dt)
        const float cg0 = fmaf(e2, R0[2], fmaf(e1, R0[1], e0 * R0[0]));
        const float cg1 = fmaf(e2, R0[5], fmaf(e1, R0[4], e0 * R0[3]));
        const float cg2 = fmaf(e2, R0[8], fmaf(e1, R0[7], e0 * R0[6]));
        // this lane's component of next_state (misc/force_controller.py:269-291): the PRIOR state everywhere on the right
        float xn = fmaf(wa0, aw0, fmaf(wa1, aw1, fmaf(wa2, aw2, fmaf(wf0, fs0, fmaf(wf1, fs1, fmaf(wf2, fs2, xr + wg))))));
        asm volatile("s_nop 1" : "+v"(xr), "+v"(Prow[0]), "+v"(Prow[1]), "+v"(Prow[2]), "+v"(Prow[3]), "+v"(Prow[4]), "+v"(Prow[5]), "+v"(Prow[6]),
                                 "+v"(Prow[7]), "+v"(Prow[8]), "+v"(Prow[9]), "+v"(Prow[10]), "+v"(Prow[11]));
        // IN PLACE: the rows the DPP operands come from (6..11) carry zero coefficients (cg, cd are zero there), so they stay what
        // they were while rows 0..5 accumulate -- no copy of the row.  Four multiply-adds per column, issued column-interleaved:
        // consecutive instructions never touch the same accumulator.
#pragma unroll
        for (int j = 0; j < NS; j++) fmac_bcast<6>(Prow[j], Prow[j], cg0);
#pragma unroll
        for (int j = 0; j < NS; j++) fmac_bcast<7>(Prow[j], Prow[j], cg1);
#pragma unroll
        for (int j = 0; j < NS; j++) fmac_bcast<8>(Prow[j], Prow[j], cg2);
#pragma unroll
        for (int j = 0; j < NS; j++) fmac_shl<6>(Prow[j], Prow[j], cd);    // row r + 6 (rows 9..11 for lanes 3..5)
        fmac_shl<6>(xn, xr, cd);                                           // position += dt * the prior velocity
        // columns: P' = (F_d P) F_d^T + Q with this lane's row of F_d P now in Prow
        const float md0 = k.dt * Prow[6], md1 = k.dt * Prow[7], md2 = k.dt * Prow[8];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            Prow[j] = fmaf(R0[6 + j], md2, fmaf(R0[3 + j], md1, fmaf(R0[j], md0, Prow[j] + qrow[j])));     // += dt (M[:, 6:9] R)[j]
            Prow[3 + j] = fmaf(k.dt, Prow[9 + j], Prow[3 + j] + qrow[3 + j]);
        }
#pragma unroll
        for (int j = 6; j < NS; j++) Prow[j] += qrow[j];
        // theta += dt trunc(R^T) omega: zero unless an entry of R reaches +-1 in float64 (see trunc_block_f64)
        if (__builtin_amdgcn_ballot_w64(amax >= 0.9999995f) != 0ull) {
            float A[9];
            const int e = trunc_block_f64(th[0], th[1], th[2], A);
            edge |= (amax >= 0.9999995f) ? e : 0;                              // status bit 4
            const float w0 = row_bcast<6>(xr), w1 = row_bcast<7>(xr), w2 = row_bcast<8>(xr);
            const float d0 = A[0] * w0 + A[1] * w1 + A[2] * w2, d1 = A[3] * w0 + A[4] * w1 + A[5] * w2, d2 = A[6] * w0 + A[7] * w1 + A[8] * w2;
            const float dth = k.dt * (r == 0 ? d0 : r == 1 ? d1 : r == 2 ? d2 : 0.f);
            xn += (amax >= 0.9999995f) ? dth : 0.f;
        }
        OS_TS(3)                                        // covariance predict + this lane's component of next_state
        // ---- optional outputs: the lane's own row of f / dp / imu came straight from the stage; its row of the rotated foot
        // positions (leg rr / 3, component rr % 3) = row rr % 3 of R, picked by one-hot weights, times that leg's p ----
        float pv = 0.f;
        if (PROT) {
            const float q0 = fmaf(rw2, R0[6], fmaf(rw1, R0[3], rw0 * R0[0])), q1 = fmaf(rw2, R0[7], fmaf(rw1, R0[4], rw0 * R0[1])),
                        q2 = fmaf(rw2, R0[8], fmaf(rw1, R0[5], rw0 * R0[2]));
            pv = fmaf(q2, in.qz, fmaf(q1, in.qxy[1], q0 * in.qxy[0]));
        }
        if (PROT && a.p_rot_out && live && r < 12) a.p_rot_out[((size_t)t * 12 + r) * B + b] = pv;
        if (FEAT && live && r < 12) {
            float *fo = a.feat_out + (size_t)t * a.feat_I * B + b;
            const float *mm = a.minmax;
            auto put = [&](int j, float v) { __builtin_nontemporal_store((v - mm[j]) / (mm[60 + j] - mm[j]), fo + (size_t)j * B); };
            put(18 + r, in.fr); put(30 + r, pv); put(42 + r, in.dr);
            if (r < 6) {
                put(54 + r, in.ir);
                put(12 + r, a.accel[((size_t)t * 6 + r) * B + b]);
            }
        }
        OS_TS(4)                                        // component selects, optional outputs
        // ---- ten sequential scalar measurement updates (kalman_filter.py:164-172 for diagonal R) ----
        float en = xn - zr;
        // The filter runs on e = x - z_lane (the lane's state component minus the lane's measurement, z constant within the
        // step): the innovation of measurement A is e of lane S, and the state update is one more multiply-add of the chain,
        // e += e[S] * (-K) -- the last one, so that its DPP operand is thirteen instructions old.  x = e + z afterwards.
        // Ten inline-assembly blocks, generated (tools/gen_rows_chain.py -> kf_rows_chain.inc) so that the ORDER is ours: the
        // column the NEXT measurement reads (SN) is updated first; two instructions later one DPP add forms its
        // S = P[sn][sn] + R[a'][a'], v_rcp_f32 follows under the other multiply-adds, and the block ends with the next gain
        // factor -P[.][sn] / S -- no instruction waits for its operand and hipcc has nothing to pad (it put an s_nop in front
        // of each of its own instructions that read an assembly result: three per measurement).  The smallest S of the whole
        // run is tested once at the end (status bit 0).  Hazards: a DPP source is at least two instructions old (P[.][sn]:
        // written first, read by the add three later; everything else was written by the previous block), v_rcp_f32's
        // result is read six or more instructions later, the prologue's s_nop 4 covers whatever hipcc scheduled in front.
        if (plain) {
            // step t + 1 (requested at the top of step t - 1): younger than it are x_out(t - 1) and step t + 2's five loads;
            // vmcnt(5) also waits for that store, issued most of a step ago
            __builtin_amdgcn_s_waitcnt(0x0f75);
            __builtin_amdgcn_wave_barrier();
            rows_issue<FEAT, PROT>(st_nxt, rd, nx);
        }
        {
            const uint32_t st_old = st_cur;
            st_cur = st_nxt; st_nxt = st_nn; st_nn = st_old;
        }
#include "kf_rows_chain.inc"
        xr = en + zr;
        OS_TS(5)                                        // ten measurement updates
        if (live && r < 12) a.x_out[((size_t)t * 12 + r) * B + b] = xr;
        if (FEAT && live && r < 12)
            __builtin_nontemporal_store((xr - a.minmax[r]) / (a.minmax[60 + r] - a.minmax[r]), a.feat_out + ((size_t)t * a.feat_I + r) * B + b);
        if (AUX && a.ptrace_out) {
            float dg = Prow[0];
#pragma unroll
            for (int i = 1; i < NS; i++) dg = (r == i) ? Prow[i] : dg;
            dg = r < 12 ? dg : 0.f;
            dg += __shfl_xor(dg, 1, 64); dg += __shfl_xor(dg, 2, 64); dg += __shfl_xor(dg, 4, 64); dg += __shfl_xor(dg, 8, 64);
            if (live && r == 0) a.ptrace_out[(size_t)t * B + b] = dg;
        }
#ifndef OS_ROWS_TS
        if (AUX && a.kgain_out) {
            // K_gain = trace(P+ H^T R^-1) (kgain_from_posterior, kf_device.hpp): lane r < 10 holds row r, its term is P+[r][SEL[r]] / R[r][r]
            float kg = Prow[0] / Rd[0];
#pragma unroll
            for (int i = 1; i < NM; i++) kg = (r == i) ? Prow[SEL[i]] / Rd[i] : kg;
            kg = r < NM ? kg : 0.f;
            kg += __shfl_xor(kg, 1, 64); kg += __shfl_xor(kg, 2, 64); kg += __shfl_xor(kg, 4, 64); kg += __shfl_xor(kg, 8, 64);
            if (live && r == 0) a.kgain_out[(size_t)t * B + b] = kg;
        }
#endif
    }
#ifdef OS_ROWS_TS
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.kgain_out) {
        unsigned long long *o = reinterpret_cast<unsigned long long *>(a.kgain_out);
        for (int i = 0; i < 8; i++) o[i] = ts_sum[i];
    }
#endif
    // ---- final state: lane r writes x[r] and row r of P; the status word is OR-reduced over the 16 lanes ----
    int status = (__builtin_amdgcn_classf(svmin, 0x180) ? 0 : 1) | ((xr * 0.f == 0.f) ? 0 : 2) | edge;
    status |= __shfl_xor(status, 1, 64); status |= __shfl_xor(status, 2, 64);
    status |= __shfl_xor(status, 4, 64); status |= __shfl_xor(status, 8, 64);
    if (live && r < 12) {
        a.x[(size_t)r * B + b] = xr;
#pragma unroll
        for (int j = 0; j < NS; j++) a.P[(size_t)(r * NS + j) * B + b] = Prow[j];
        if (r == 0) a.status[b] = status;
    }
}

// feat / aux / p_rot_out pick the instantiation (os_kf_run_impl: kf_kernels.hip)
hipError_t launch_kf_rows2(const KfRunArgs &a, const float *qmat, bool feat, bool aux, hipStream_t s)
{
    dim3 grid((a.B + 15) / 16), block(256);
    if (feat) hipLaunchKernelGGL((kf_run_rows2_kernel<false, true, true>), grid, block, 0, s, a, qmat);
    else if (aux) hipLaunchKernelGGL((kf_run_rows2_kernel<true, false, true>), grid, block, 0, s, a, qmat);
    else if (a.p_rot_out) hipLaunchKernelGGL((kf_run_rows2_kernel<false, false, true>), grid, block, 0, s, a, qmat);
    else hipLaunchKernelGGL((kf_run_rows2_kernel<false, false, false>), grid, block, 0, s, a, qmat);
    return hipGetLastError();
}

}  // namespace osk
