// kf_step.hip -- os_kf_step: ONE filter step of ONE trajectory with inputs and outputs in HOST memory, for the drop-in
// `Kalman_Filter` class (the reference's caller loop steps one filter instance at a time:
// data_collection/data_conversion_Kalman_to_Training.py:193-199).
//
// B = 1 is pure latency, so the design removes round trips instead of arithmetic:
//   * the context owns one pinned, device-mapped staging block; the kernel reads its inputs from it and writes its
//     outputs to it directly over PCIe (a few KB) -- no hipMemcpy, no device buffers, ONE launch and ONE stream
//     synchronise per call, whatever combination of get_odom / predict / update the call asks for;
//   * one wavefront works on the step: the 12 x 12 covariance sits in LDS in FLOAT64, three entries per lane, and the
//     LU of S / gain / covariance update run across the lanes (the dependent chain of a single lane would be ~20 k
//     instructions long); the short vector part (odometry, rotation, next_state) is computed redundantly on every lane.
// Everything is float64 here: one wavefront issues an fp64 instruction as fast as an fp32 one (profiles/r01j_valu_rates.md),
// so the drop-in class gets the reference's own precision (float64 NumPy) for free.
//
// Math, derived from the reference's equations (not its code):
//   get_odom / set_measurements   kalman_filter/kalman_filter.py:79-117
//   predict                        kalman_filter/kalman_filter.py:119-138   (F_d = I + dt F)
//   predict_mpc covariance         kalman_filter/kalman_filter.py:153-161   (F_d = element-wise exp(dt F), R from body_ref)
//   next_state                     misc/force_controller.py:269-291        (int64 truncation of A[0:3,6:9], :248-251,:271)
//   update                         kalman_filter/kalman_filter.py:164-174   (K returned; P <- (I - K H) P, unsymmetrised)
#include "launch.hpp"
#include "kf_args.hpp"

#include <stddef.h>

namespace oss {

using osk::SEL;

// staging block, host-visible and device-mapped.  Inputs first, outputs behind them; all doubles (8-byte aligned).
struct StepIO {
    double x[12], P[144], z[10], p[12], f[12], dp[12], imu[6], bref[12], Q[144], R[100];
    double dt, inv_mass, gz, inv_inertia[3];
    uint32_t contact, what;
    // outputs (x, P, z, p are in/out)
    double x_model[12], K[120], ptrace, kgain;
    int32_t status, pad;
};
constexpr int IO_DOUBLES = sizeof(StepIO) / 8;
constexpr int IO_INOUT_DOUBLES = 12 + 144 + 10 + 12;                 // x, P, z, p: read and written
constexpr int IO_IN_DOUBLES = offsetof(StepIO, x_model) / 8;         // everything the kernel reads

struct StepMem {
    StepIO io;
    double Mt[144], L[10][10], Kk[12][10], dinv[10], cs[12], rs[12], e[10], G[9];
};

__device__ __forceinline__ void rot64(double tx, double ty, double tz, double *R)
{
    double sx, cx, sy, cy, sz, cz;
    osk::sincos_f64(tx, &sx, &cx); osk::sincos_f64(ty, &sy, &cy); osk::sincos_f64(tz, &sz, &cz);
    // Rz Ry Rx (kalman_filter.py:187-191), closed form
    R[0] = cz * cy; R[1] = cz * sy * sx - sz * cx; R[2] = cz * sy * cx + sz * sx;
    R[3] = sz * cy; R[4] = sz * sy * sx + cz * cx; R[5] = sz * sy * cx - cz * sx;
    R[6] = -sy;     R[7] = cy * sx;                R[8] = cy * cx;
}

__device__ __forceinline__ double rsqrt_nr(double a)
{
    double r = __builtin_amdgcn_rsq(a);
    r = r * (1.5 - 0.5 * a * r * r);
    r = r * (1.5 - 0.5 * a * r * r);
    return r;
}

// One filter step on the LDS-resident block M (one wavefront; the caller has filled M.io and synchronised).  Returns the status bits.
__device__ __forceinline__ int step_body(StepMem &M, const int lane);

// the QP's side of an OS_STEP_MPC step: float32 copies of its inputs, its outputs (pinned, device-mapped like StepIO)
struct StepMpc {
    float x[12], ref[12], p[12];
    uint32_t contact;
    int32_t iters, status, pad;
    float u[60];
};

// mpc_f: null, or the forces of horizon step 0 the QP launch in front of this kernel left in device memory (OS_STEP_MPC)
__global__ __launch_bounds__(64, 1) void kf_step_kernel(StepIO *__restrict__ g, const float *__restrict__ mpc_f)
{
    __shared__ StepMem M;
    const int lane = threadIdx.x;
    StepIO &io = M.io;
    {
        // the staging block lives in host memory: every load is a PCIe round trip, so all of them go out before the first
        // one is waited for (fully unrolled: eight loads per lane in flight)
        const double *src = reinterpret_cast<const double *>(g);
        double *dst = reinterpret_cast<double *>(&io);
        constexpr int NIT = (IO_IN_DOUBLES + 63) / 64;
        double v[NIT];
#pragma unroll
        for (int i = 0; i < NIT; i++) v[i] = (lane + 64 * i < IO_IN_DOUBLES) ? __builtin_nontemporal_load(src + lane + 64 * i) : 0.0;
#pragma unroll
        for (int i = 0; i < NIT; i++)
            if (lane + 64 * i < IO_IN_DOUBLES) dst[lane + 64 * i] = v[i];
    }
    __syncthreads();
    if (mpc_f) {                     // f = self.f[:, 0] of the QP just solved (kalman_filter.py:150-152,161)
        if (lane < 12) io.f[lane] = (double)mpc_f[lane];
        __syncthreads();
    }
    const int status = step_body(M, lane);
    __syncthreads();
    if (lane == 0) io.status = status;
    __syncthreads();
    {
        // write back the in/out head of the block (x, P, z, p) and the outputs behind the constant inputs
        double *dst = reinterpret_cast<double *>(g);
        const double *src = reinterpret_cast<const double *>(&io);
        for (int i = lane; i < IO_INOUT_DOUBLES; i += 64) dst[i] = src[i];
        for (int i = IO_IN_DOUBLES + lane; i < IO_DOUBLES; i += 64) dst[i] = src[i];
    }
}

__device__ __forceinline__ int step_body(StepMem &M, const int lane)
{
    StepIO &io = M.io;
    const uint32_t what = io.what;
    const double dt = io.dt;
    int status = 0;

    // ---- get_odom + set_measurements (every lane computes the same ten numbers; lane 0 stores them) ----
    if (what & OS_STEP_ODOM) {
        double sum_c = 0.0, vx = 0.0, vy = 0.0, vz = 0.0, pz = 0.0;
#pragma unroll
        for (int l = 0; l < 4; l++) {
            const uint32_t cb = (io.contact >> (8 * l)) & 0xffu;
            sum_c += (double)cb;
            if (cb == 1u) { vx += io.dp[3 * l]; vy += io.dp[3 * l + 1]; pz += io.p[3 * l + 2]; }
            if (cb == 0u) vz += io.dp[3 * l + 2];
        }
        double v[3] = {0.0, 0.0, 0.0};
        if (sum_c != 0.0) { v[0] = -vx / sum_c; v[1] = -vy / sum_c; v[2] = -vz / sum_c; pz = -pz / sum_c; }   // no stance leg: odom = 0 (:97-98)
        double R[9];
        rot64(io.imu[0], io.imu[1], io.imu[2], R);
        __syncthreads();
        if (lane == 0) {
            io.z[0] = io.imu[0]; io.z[1] = io.imu[1]; io.z[2] = io.imu[2];
            io.z[3] = pz;
            io.z[4] = io.imu[3]; io.z[5] = io.imu[4]; io.z[6] = io.imu[5];
#pragma unroll
            for (int i = 0; i < 3; i++) io.z[7 + i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2];
        }
        __syncthreads();
    }

    if (what & OS_STEP_PREDICT) {
        double R[9], x[12];
#pragma unroll
        for (int i = 0; i < 12; i++) x[i] = io.x[i];
        rot64(x[0], x[1], x[2], R);                         // prior attitude: F_d of predict() and next_state
        // ---- covariance ----
        if (what & OS_STEP_DENSE_FD) {
            // F_d = exp(dt F) element-wise = 1 1^T + E, E[0:3,6:9] = expm1(dt Rb^T), E[3:6,9:12] = (e^dt - 1) I, Rb from body_ref
            double Rb[9];
            rot64(io.bref[0], io.bref[1], io.bref[2], Rb);
            if (lane < 9) M.e[lane] = expm1(dt * Rb[3 * (lane % 3) + lane / 3]);      // e[3 i + k] = expm1(dt Rb[k][i])
            if (lane == 9) M.e[9] = expm1(dt);
            if (lane < 12) {
                double c = 0.0;
#pragma unroll
                for (int i = 0; i < 12; i++) c += io.P[i * 12 + lane];
                M.cs[lane] = c;
            }
            __syncthreads();
            const double ed = M.e[9];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const int el = lane + 64 * r;
                if (el < 144) {
                    const int i = el / 12, j = el % 12;
                    double m = M.cs[j];
                    if (i < 3) m += M.e[3 * i] * io.P[6 * 12 + j] + M.e[3 * i + 1] * io.P[7 * 12 + j] + M.e[3 * i + 2] * io.P[8 * 12 + j];
                    else if (i < 6) m += ed * io.P[(i + 6) * 12 + j];
                    M.Mt[el] = m;
                }
            }
            __syncthreads();
            if (lane < 12) {
                double c = 0.0;
#pragma unroll
                for (int j = 0; j < 12; j++) c += M.Mt[lane * 12 + j];
                M.rs[lane] = c;
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const int el = lane + 64 * r;
                if (el < 144) {
                    const int i = el / 12, j = el % 12;
                    double v = M.rs[i];
                    if (j < 3) v += M.e[3 * j] * M.Mt[i * 12 + 6] + M.e[3 * j + 1] * M.Mt[i * 12 + 7] + M.e[3 * j + 2] * M.Mt[i * 12 + 8];
                    else if (j < 6) v += ed * M.Mt[i * 12 + j + 6];
                    io.P[el] = v + io.Q[el];
                }
            }
            __syncthreads();
        } else {
            // F_d = I + G, G[0:3,6:9] = dt R^T, G[3:6,9:12] = dt I:  M = F_d P, then P' = M F_d^T + Q
            if (lane < 9) M.G[lane] = dt * R[3 * (lane % 3) + lane / 3];                 // G[3 i + k] = dt R[k][i]
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const int el = lane + 64 * r;
                if (el < 144) {
                    const int i = el / 12, j = el % 12;
                    double m = io.P[el];
                    if (i < 3) m += M.G[3 * i] * io.P[6 * 12 + j] + M.G[3 * i + 1] * io.P[7 * 12 + j] + M.G[3 * i + 2] * io.P[8 * 12 + j];
                    else if (i < 6) m += dt * io.P[(i + 6) * 12 + j];
                    M.Mt[el] = m;
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const int el = lane + 64 * r;
                if (el < 144) {
                    const int i = el / 12, j = el % 12;
                    double v = M.Mt[el];
                    if (j < 3) v += M.G[3 * j] * M.Mt[i * 12 + 6] + M.G[3 * j + 1] * M.Mt[i * 12 + 7] + M.G[3 * j + 2] * M.Mt[i * 12 + 8];
                    else if (j < 6) v += dt * M.Mt[i * 12 + j + 6];
                    io.P[el] = v + io.Q[el];
                }
            }
            __syncthreads();
        }
        // ---- next_state (replicated): x <- (I + A dt) x + B dt f + dt g, p rotated to the world frame in place ----
        double pw[12], tau[3] = {0.0, 0.0, 0.0}, fs[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int l = 0; l < 4; l++) {
            const double a = io.p[3 * l], b = io.p[3 * l + 1], c = io.p[3 * l + 2];
            const double wx = R[0] * a + R[1] * b + R[2] * c, wy = R[3] * a + R[4] * b + R[5] * c, wz = R[6] * a + R[7] * b + R[8] * c;
            pw[3 * l] = wx; pw[3 * l + 1] = wy; pw[3 * l + 2] = wz;
            const double fx = io.f[3 * l], fy = io.f[3 * l + 1], fz = io.f[3 * l + 2];
            tau[0] += wy * fz - wz * fy; tau[1] += wz * fx - wx * fz; tau[2] += wx * fy - wy * fx;
            fs[0] += fx; fs[1] += fy; fs[2] += fz;
        }
        // I_hat^-1 = R diag(1/I) R^T (R orthogonal)
        const double tb0 = (R[0] * tau[0] + R[3] * tau[1] + R[6] * tau[2]) * io.inv_inertia[0];
        const double tb1 = (R[1] * tau[0] + R[4] * tau[1] + R[7] * tau[2]) * io.inv_inertia[1];
        const double tb2 = (R[2] * tau[0] + R[5] * tau[1] + R[8] * tau[2]) * io.inv_inertia[2];
        const double aw0 = R[0] * tb0 + R[1] * tb1 + R[2] * tb2, aw1 = R[3] * tb0 + R[4] * tb1 + R[5] * tb2,
                     aw2 = R[6] * tb0 + R[7] * tb1 + R[8] * tb2;
        const double w0 = x[6], w1 = x[7], w2 = x[8];
        // A[0:3,6:9] = R^T stored into an int64 array: truncated toward zero
        {   // status bit 4 (see trunc_block_f64, kf_device.hpp): an entry within 2^-40 of +-1 away from the exact start theta = 0
            double rmax = 0.0;
#pragma unroll
            for (int i = 0; i < 9; i++) rmax = fmax(rmax, fabs(R[i]));
            if (rmax >= 1.0 - 0x1p-40 && !(x[0] == 0.0 && x[1] == 0.0 && x[2] == 0.0)) status |= 16;
        }
#pragma unroll
        for (int i = 0; i < 3; i++) x[i] += dt * (trunc(R[i]) * w0 + trunc(R[3 + i]) * w1 + trunc(R[6 + i]) * w2);
        x[3] += dt * x[9]; x[4] += dt * x[10]; x[5] += dt * x[11];
        x[6] = w0 + dt * aw0; x[7] = w1 + dt * aw1; x[8] = w2 + dt * aw2;
        x[9] += dt * (fs[0] * io.inv_mass); x[10] += dt * (fs[1] * io.inv_mass); x[11] += dt * (fs[2] * io.inv_mass) + dt * io.gz;
        __syncthreads();
        if (lane < 12) {
            double xv = x[0], pv = pw[0];
#pragma unroll
            for (int i = 1; i < 12; i++) { xv = lane == i ? x[i] : xv; pv = lane == i ? pw[i] : pv; }
            io.x[lane] = xv; io.x_model[lane] = xv; io.p[lane] = pv;
        }
        __syncthreads();
        if (lane == 0) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < 12; i++) t += io.P[i * 13];
            io.ptrace = t;
        }
        __syncthreads();
    }

    if (what & OS_STEP_UPDATE) {
        // S = P[sel,sel] + R AS IT IS: rounding leaves P (and S) not exactly symmetric and the reference inverts that S
        // (np.linalg.inv, kalman_filter.py:169).  K from a symmetrised S is unstable with P -= K H P: P's antisymmetric part then
        // grows step by step (kf_dense_rows.hpp update_batch_row; found by tools/fuzz_kf.py, round 5).  LU without pivoting.
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int el = lane + 64 * r;
            if (el < 100) {
                const int a = el / 10, b = el % 10;
                M.L[a][b] = io.P[SEL[a] * 12 + SEL[b]] + io.R[a * 10 + b];
            }
        }
        __syncthreads();
        // row a of the factors in M.L[a][:]: multipliers L[a][q] (q < a) | U[a][q] (q >= a); lane i > j eliminates its row
        for (int j = 0; j < 10; j++) {
            double d = M.L[j][j];
            if (!(d > 0.0) || !(d < 1.0e300)) { status |= 1; d = 1.0; }
            const double di = 1.0 / d;
            if (lane > j && lane < 10) {
                const double m = M.L[lane][j] * di;
                for (int q = j + 1; q < 10; q++) M.L[lane][q] -= m * M.L[j][q];
                M.L[lane][j] = m;
            }
            if (lane == 0) M.dinv[j] = di;
            __syncthreads();
        }
        // K[i,:] S = P[i,sel], one row per lane: w U = P[i,sel], then K L = w (unit diagonal)
        {
            const int i = lane < 12 ? lane : 0;
            double Kr[10];
#pragma unroll
            for (int c = 0; c < 10; c++) {
                double s = io.P[i * 12 + SEL[c]];
#pragma unroll
                for (int b = 0; b < c; b++) s -= Kr[b] * M.L[b][c];
                Kr[c] = s * M.dinv[c];
            }
#pragma unroll
            for (int c = 8; c >= 0; c--) {
#pragma unroll
                for (int b = c + 1; b < 10; b++) Kr[c] -= Kr[b] * M.L[b][c];
            }
            double sx = 0.0;
#pragma unroll
            for (int a = 0; a < 10; a++) sx += Kr[a] * (io.z[a] - io.x[SEL[a]]);
            __syncthreads();                                   // every lane has read the prior x
            if (lane < 12) {
#pragma unroll
                for (int a = 0; a < 10; a++) { M.Kk[lane][a] = Kr[a]; io.K[lane * 10 + a] = Kr[a]; }
                io.x[lane] += sx;
            }
        }
        __syncthreads();
        // P <- P - K P[sel,:]: every entry reads only OLD rows; a lane forms its three values before any is written
        double pn[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const int el = lane + 64 * r;
            pn[r] = 0.0;
            if (el < 144) {
                const int i = el / 12, j = el % 12;
                double s = 0.0;
#pragma unroll
                for (int a = 0; a < 10; a++) s += M.Kk[i][a] * io.P[SEL[a] * 12 + j];
                pn[r] = io.P[el] - s;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const int el = lane + 64 * r;
            if (el < 144) io.P[el] = pn[r];
        }
        __syncthreads();
        if (lane == 0) {
            double tk = 0.0, tp = 0.0, fin = 0.0;
#pragma unroll
            for (int a = 0; a < 10; a++) tk += M.Kk[a][a];      // np.trace of the 12 x 10 K: its ten main-diagonal entries (:174)
#pragma unroll
            for (int i = 0; i < 12; i++) { tp += io.P[i * 13]; fin += io.x[i] * 0.0; }
            io.kgain = tk; io.ptrace = tp;
            if (!(fin == 0.0)) status |= 2;
        }
    }
    return status;
}

// ---------------------------------------------------------------------------------------------------------------
// kf_run_wave_kernel -- the layout BASELINE.json's north_star names literally: ONE TRAJECTORY PER WAVEFRONT, state and
// covariance tile in LDS, the whole T loop in one launch (OS_KF_WAVE_PER_TRAJECTORY; never chosen by default).
// It is step_body above (float64 on one wavefront: covariance three entries per lane, LU / gain / update across the
// lanes) run over the SoA streams: x, P, Q, R stay in the workgroup's LDS block for all T steps; per step 43 lanes fetch
// the trajectory's 43 input dwords (one 4-byte element out of each stream row: with the trajectory index fastest in memory,
// a wavefront that owns ONE trajectory cannot coalesce -- every dword costs a 64-byte sector) and twelve lanes store x_out.
// Built to MEASURE the layout against the lane-per-trajectory and 16-lanes-per-trajectory kernels (profiles/
// r04_wave_per_trajectory.md), not to be fast: a step is ~35 barrier-separated cross-lane phases on one wavefront.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void kf_run_wave_kernel(const osk::KfRunArgs a)
{
    __shared__ StepMem M;
    const int lane = threadIdx.x;
    // neighbouring trajectories share the 64-byte sectors of every input row: keep them on one XCD's L2 (consecutive workgroup
    // ids go round-robin to the eight XCDs)
    const int nblk = gridDim.x, q8 = nblk >> 3, r8 = nblk & 7, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int b = xcd * q8 + (xcd < r8 ? xcd : r8) + slot;
    if (b >= a.B) return;
    const size_t B = (size_t)a.B;
    StepIO &io = M.io;
    for (int i = lane; i < 144; i += 64) {
        io.P[i] = (double)a.P[(size_t)i * B + b];
        io.Q[i] = (double)a.k.Q[i];
        if (i < 100) io.R[i] = (double)a.k.R[i];
    }
    if (lane < 12) io.x[lane] = (double)a.x[(size_t)lane * B + b];
    if (lane == 0) {
        io.dt = (double)a.k.dt; io.inv_mass = (double)a.k.inv_mass; io.gz = (double)a.k.gz;
        for (int i = 0; i < 3; i++) io.inv_inertia[i] = (double)a.k.inv_inertia[i];
        io.what = OS_STEP_ODOM | OS_STEP_PREDICT | OS_STEP_UPDATE;
    }
    int status = 0;
    for (int t = 0; t < a.T; t++) {
        // lanes 0-11: p, 12-23: f, 24-35: dp, 36-41: imu, 42: the contact word
        float v = 0.f;
        uint32_t cw = 0u;
        if (lane < 12) v = __builtin_nontemporal_load(a.p + ((size_t)t * 12 + lane) * B + b);
        else if (lane < 24) v = __builtin_nontemporal_load(a.f + ((size_t)t * 12 + lane - 12) * B + b);
        else if (lane < 36) v = __builtin_nontemporal_load(a.dp + ((size_t)t * 12 + lane - 24) * B + b);
        else if (lane < 42) v = __builtin_nontemporal_load(a.imu + ((size_t)t * 6 + lane - 36) * B + b);
        else if (lane == 42) cw = __builtin_nontemporal_load(a.contact + (size_t)t * B + b);
        __syncthreads();                                     // the previous step's readers of io are done
        if (lane < 12) io.p[lane] = (double)v;
        else if (lane < 24) io.f[lane - 12] = (double)v;
        else if (lane < 36) io.dp[lane - 24] = (double)v;
        else if (lane < 42) io.imu[lane - 36] = (double)v;
        else if (lane == 42) io.contact = cw;
        __syncthreads();
        status |= step_body(M, lane);
        __syncthreads();
        if (lane < 12) {
            __builtin_nontemporal_store((float)io.x[lane], a.x_out + ((size_t)t * 12 + lane) * B + b);
            if (a.p_rot_out) __builtin_nontemporal_store((float)io.p[lane], a.p_rot_out + ((size_t)t * 12 + lane) * B + b);
        }
        if (lane == 0) {
            if (a.ptrace_out) a.ptrace_out[(size_t)t * B + b] = (float)io.ptrace;
            if (a.kgain_out) a.kgain_out[(size_t)t * B + b] = (float)io.kgain;
        }
    }
    for (int i = lane; i < 144; i += 64) a.P[(size_t)i * B + b] = (float)io.P[i];
    if (lane < 12) a.x[(size_t)lane * B + b] = (float)io.x[lane];
    if (lane == 0) a.status[b] = status;
}

}  // namespace oss

// os_kf_run with OS_KF_WAVE_PER_TRAJECTORY (called from os_kf_run_impl)
int os_kf_run_wave(os_ctx *ctx, const osk::KfRunArgs &a, hipStream_t s)
{
    const int slot = os_prof_begin(ctx, OS_PHASE_KF, s, "kf_run_wave_kernel");
    hipLaunchKernelGGL(oss::kf_run_wave_kernel, dim3((unsigned)a.B), dim3(64), 0, s, a);
    os_prof_end(ctx, slot, s);
    OS_HIP(ctx, hipGetLastError());
    return 0;
}

struct os_step_state {
    oss::StepIO *host, *dev;
    oss::StepMpc *mhost, *mdev;          // behind the StepIO in the same pinned allocation
    float *mpc_f;                        // device: the QP's horizon-step-0 forces, read by the step kernel
    double *warm_u; uint8_t *warm_state; uint32_t *warm_contact;     // device: the context's QP warm start (os_mpc_solve_one)
};

int os_mpc_solve_one(os_ctx *ctx, const float *x, const float *body_ref, const float *p, const uint32_t *contact, uint32_t contact_word,
                     float *f_out, float *u_out, int32_t *iters, int32_t *status, double *warm_u, uint8_t *warm_state, uint32_t *warm_contact,
                     hipStream_t s);      // mpc_kernels.hip

void os_step_destroy(os_ctx *ctx)
{
    os_step_state *st = (os_step_state *)ctx->step;
    if (!st) return;
    if (st->host) (void)hipHostFree(st->host);
    if (st->mpc_f) (void)hipFree(st->mpc_f);
    if (st->warm_u) (void)hipFree(st->warm_u);
    free(st);
    ctx->step = nullptr;
}

static int kf_step_impl(os_ctx *ctx, uint32_t what, const double *model, const double *p, const double *f, const double *dp, const double *imu,
                        const uint8_t *contact, const double *body_ref, const double *Q, const double *R, double *x, double *P,
                        double *z, double *p_rot, double *x_model, double *K, double *ptrace, double *kgain, int32_t *status,
                        double *f_all, int32_t *qp_iters, void *stream)
{
    OS_CHECK_CTX(ctx);
    const bool odom = what & OS_STEP_ODOM, pred = what & OS_STEP_PREDICT, upd = what & OS_STEP_UPDATE, dense = what & OS_STEP_DENSE_FD;
    const bool mpc = what & OS_STEP_MPC;
    if (!(odom || pred || upd)) return os_fail(ctx, -2, "os_kf_step: nothing to do (what = 0)");
    if (odom && (!p || !dp || !imu || !contact || !z)) return os_fail(ctx, -2, "os_kf_step: OS_STEP_ODOM needs p, dp, imu, contact, z");
    if (mpc && (!pred || !body_ref || !contact || !p || !x)) return os_fail(ctx, -2, "os_kf_step: OS_STEP_MPC needs OS_STEP_PREDICT and x, p, body_ref, contact");
    if (pred && (!p || (!f && !mpc) || !x || !P || !Q)) return os_fail(ctx, -2, "os_kf_step: OS_STEP_PREDICT needs p, f, x, P, Q");
    if (pred && dense && !body_ref) return os_fail(ctx, -2, "os_kf_step: OS_STEP_DENSE_FD needs body_ref");
    if (upd && (!x || !P || !z || !R)) return os_fail(ctx, -2, "os_kf_step: OS_STEP_UPDATE needs x, P, z, R");
    OS_HIP(ctx, hipSetDevice(ctx->device));
    os_step_state *st = (os_step_state *)ctx->step;
    if (!st) {
        st = (os_step_state *)calloc(1, sizeof(os_step_state));
        if (!st) return os_fail(ctx, -13, "os_kf_step: out of memory");
        if (hipHostMalloc((void **)&st->host, sizeof(oss::StepIO) + sizeof(oss::StepMpc), hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void **)&st->dev, st->host, 0) != hipSuccess) {
            if (st->host) (void)hipHostFree(st->host);
            free(st);
            return os_fail(ctx, -10, "os_kf_step: cannot allocate the pinned staging block");
        }
        memset(st->host, 0, sizeof(oss::StepIO) + sizeof(oss::StepMpc));
        st->mhost = reinterpret_cast<oss::StepMpc *>(st->host + 1);
        st->mdev = reinterpret_cast<oss::StepMpc *>(st->dev + 1);
        ctx->step = st;
    }
    if (mpc && !st->mpc_f) {
        // the QP's device-side state: forces of horizon step 0 (12 floats) and the warm start (u [64] doubles, faces [64] bytes, contact word)
        if (hipMalloc((void **)&st->mpc_f, 16 * sizeof(float)) != hipSuccess || hipMalloc((void **)&st->warm_u, 64 * sizeof(double) + 64 + 16) != hipSuccess)
            return os_fail(ctx, -10, "os_kf_step: cannot allocate the QP state");
        st->warm_state = reinterpret_cast<uint8_t *>(st->warm_u + 64);
        st->warm_contact = reinterpret_cast<uint32_t *>(st->warm_state + 64);
        OS_HIP(ctx, hipMemset(st->warm_u, 0, 64 * sizeof(double)));
        OS_HIP(ctx, hipMemset(st->warm_state, 0x05, 64));
        OS_HIP(ctx, hipMemset(st->warm_contact, 0xff, 16));
    }
    oss::StepIO &io = *st->host;
    auto put = [](double *d, const double *s, int n) { if (s) memcpy(d, s, sizeof(double) * n); };
    put(io.x, x, 12); put(io.P, P, 144); put(io.p, p, 12); put(io.f, f, 12); put(io.dp, dp, 12); put(io.imu, imu, 6);
    put(io.bref, body_ref, 12); put(io.Q, Q, 144); put(io.R, R, 100);
    if (!odom) put(io.z, z, 10);
    io.contact = contact ? ((uint32_t)contact[0] | ((uint32_t)contact[1] << 8) | ((uint32_t)contact[2] << 16) | ((uint32_t)contact[3] << 24)) : 0u;
    io.what = what;
    if (model) {          // dt, mass, Ixx, Iyy, Izz, g_z in float64 (settings.py:5-23): the context's configuration is float32
        io.dt = model[0]; io.inv_mass = 1.0 / model[1]; io.gz = model[5];
        for (int i = 0; i < 3; i++) io.inv_inertia[i] = 1.0 / model[2 + i];
    } else {
        io.dt = (double)ctx->k.dt; io.inv_mass = 1.0 / ctx->mass64; io.gz = ctx->gz64;
        for (int i = 0; i < 3; i++) io.inv_inertia[i] = 1.0 / ctx->inertia64[i];
    }
    hipStream_t s = (hipStream_t)stream;
    if (mpc) {
        // kalman_filter.py:141-152: the stance controller's QP from the PRIOR state, the reference pose, the foot positions and the
        // contact pattern (float32 inputs as os_mpc_solve takes them, float64 inside); its launch goes out in front of the step
        // kernel on the same stream -- one synchronise for both
        oss::StepMpc &m = *st->mhost;
        for (int i = 0; i < 12; i++) { m.x[i] = (float)x[i]; m.ref[i] = (float)body_ref[i]; m.p[i] = (float)p[i]; }
        m.contact = io.contact; m.status = 0; m.iters = 0;
        if (int rc = os_mpc_solve_one(ctx, st->mdev->x, st->mdev->ref, st->mdev->p, &st->mdev->contact, io.contact, st->mpc_f, st->mdev->u,
                                      &st->mdev->iters, &st->mdev->status, st->warm_u, st->warm_state, st->warm_contact, s))
            return rc;
    }
    const int slot = os_prof_begin(ctx, OS_PHASE_KF, s, "kf_step_kernel");
    hipLaunchKernelGGL(oss::kf_step_kernel, dim3(1), dim3(64), 0, s, st->dev, mpc ? (const float *)st->mpc_f : (const float *)nullptr);
    os_prof_end(ctx, slot, s);
    OS_HIP(ctx, hipGetLastError());
    OS_HIP(ctx, hipStreamSynchronize(s));                     // the results are in host memory from here on
    if (mpc) {
        // the (12, 5) control matrix the reference keeps in self.f, column h = horizon step h (u is [h][12])
        if (f_all)
            for (int h = 0; h < 5; h++)
                for (int i = 0; i < 12; i++) f_all[i * 5 + h] = (double)st->mhost->u[h * 12 + i];
        if (qp_iters) *qp_iters = st->mhost->iters;
        if (st->mhost->status & 4) io.status |= 4;            // QP iteration cap (as os_mpc_solve reports it)
    }
    auto get = [](double *d, const double *s, int n) { if (d) memcpy(d, s, sizeof(double) * n); };
    if (odom) get(z, io.z, 10);
    if (pred) { get(x, io.x, 12); get(P, io.P, 144); get(p_rot, io.p, 12); get(x_model, io.x_model, 12); }
    if (upd) { get(x, io.x, 12); get(P, io.P, 144); get(K, io.K, 120); if (kgain) *kgain = io.kgain; }
    if ((pred && !dense) || upd) { if (ptrace) *ptrace = io.ptrace; }
    if (status) *status = io.status;
    return 0;
}

extern "C" int os_kf_step(os_ctx *ctx, uint32_t what, const double *model, const double *p, const double *f, const double *dp, const double *imu,
                          const uint8_t *contact, const double *body_ref, const double *Q, const double *R, double *x, double *P,
                          double *z, double *p_rot, double *x_model, double *K, double *ptrace, double *kgain, int32_t *status,
                          void *stream)
{
    return kf_step_impl(ctx, what, model, p, f, dp, imu, contact, body_ref, Q, R, x, P, z, p_rot, x_model, K, ptrace, kgain, status, nullptr, nullptr,
                        stream);
}

extern "C" int os_kf_step_mpc(os_ctx *ctx, uint32_t what, const double *model, const double *p, const double *dp, const double *imu,
                              const uint8_t *contact, const double *body_ref, const double *Q, const double *R, double *x, double *P,
                              double *z, double *p_rot, double *x_model, double *K, double *ptrace, double *kgain, double *f_all,
                              int32_t *qp_iters, int32_t *status, void *stream)
{
    return kf_step_impl(ctx, what | OS_STEP_MPC | OS_STEP_PREDICT, model, p, nullptr, dp, imu, contact, body_ref, Q, R, x, P, z, p_rot, x_model, K, ptrace,
                        kgain, status, f_all, qp_iters, stream);
}
