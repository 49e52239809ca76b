// mpc_common.hpp -- problem description, argument block and small float64 helpers shared by the convex-MPC QP solvers:
// mpc_kernels.hip (one wavefront per QP; the persistent estimate_state_mpc kernel) and mpc_quad.hip (round 6: sixteen lanes per QP,
// four QPs per wavefront).  Reference: misc/force_controller.py:70-225, kalman_filter/kalman_filter.py:64-72,141-152.
#pragma once
#include "launch.hpp"

#include <math.h>

#include <type_traits>

#include "kf_args.hpp"

namespace osm {

constexpr int NVMAX = 60, NLSMAX = 20;

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N) -- where a loop body needs its index as an immediate (a DPP lane
// select) or the unroller must not be allowed to give up
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
enum : int { SZ_FREE = 0, SZ_ZERO = 1, SZ_MAX = 2 };

struct MpcParams {
    double w[12];            // state weights (kalman_filter/kalman_filter.py:64), terminal = stage (:72)
    double rw, mu, fzmax;    // control weight (:66), friction coefficient and force cap (force_controller.py:147-149)
    double dt, inv_mass, inv_inertia[3], gz;
};

struct MpcArgs {
    int B;                         // row stride of every [rows][B] stream
    int n;                         // problems of this launch: indices 0 .. n - 1 behind the pointers (n = B, or a shard of the batch whose
                                   // pointers are offset to its first trajectory: os_kf_mpc_run's two concurrent halves)
    const float *x, *ref, *p;      // [12][B]
    const uint32_t *contact;       // [B] 4 packed bytes
    float *f_out;                  // [12][B] forces of horizon step 0
    float *u_out;                  // [60][B] or null
    int32_t *iters;                // [B] or null
    int32_t *status;               // [B] bit 2 set when the iteration cap was hit (or-ed in)
    int max_iter;
    // warm start across the steps of os_kf_mpc_run (null: always cold): the previous step's solution and faces are reused
    // when the trajectory's contact word is unchanged (the pyramids do not move, so the old u stays feasible)
    double *warm_u;                // [B][64]
    uint8_t *warm_state;           // [B][64]  (sx+1) | (sy+1) << 2 | sz << 4 of the lane's leg-step
    uint32_t *warm_contact;        // [B]
    // round 6, the two-pass form (mpc_quad.hip -> mpc_kernels.hip): a 16-lane row gives a problem up after `cap` active-set iterations,
    // leaves its state (point + faces = the warm-start record) in the warm arrays and appends the index to todo[]; the
    // wavefront-per-QP instance then continues exactly there.  The iteration count is long-tailed (mean 4, maximum 30-40 at every
    // step): a straggler costs 30-50 k cycles per iteration on a row and 15 k on a wavefront of its own.
    int cap;                       // 0: no cap
    int cold_in;                   // 1: ignore the warm arrays' CONTENT on entry (they are the hand-over buffer of a cold solve)
    int32_t *todo;                 // [2][B]: indices handed over, per leg count 1 / 2
    int32_t *todo_count;           // [2]
    // round 6: the per-problem records of mpc_quad.hip (QuadRec: generators, linear term; written by mpc_prep_kernel, read by the rows)
    double *rec;                   // [B] x 1,792 bytes
    MpcParams prm;
};

// Round 6, the filter step inside the QP launch (mpc_quad.hip): a row that finishes a QP marks its trajectory done; wavefronts whose
// rows have run out of QPs take blocks of four consecutive trajectories (a ticket counter), wait for their marks and run the step of
// kf_dense_rows_kernel<BATCH> on them -- the work of the filter kernel (0.10 ms per step at B = 65,536) moves under the stragglers of
// the QP launch (its last ~0.3 ms keep a few percent of the chip busy), and the step has one dependent launch fewer.
struct PostArgs {
    osk::KfRunArgs kf;             // this step's streams and state (T = 1); kf.status = the run's status words (or-ed into)
    const float *qr;               // Q 144 | R 100 (the context's device copy)
    uint32_t *done;                // [B]: the launch sequence number once the trajectory's forces are written
    uint32_t seq;                  // this launch's sequence number (marks of earlier launches never match)
};

// 1/a to full double precision: v_rcp_f64 + two Newton steps (an IEEE division costs ~4x as many instructions)
__device__ __forceinline__ double rcp64(double a)
{
    double r = __builtin_amdgcn_rcp(a);
    r = fma(fma(-a, r, 1.0), r, r);
    r = fma(fma(-a, r, 1.0), r, r);
    return r;
}

// 1/sqrt(a) to full double precision: v_rsq_f64 + two Newton steps (1.0 / sqrt(a) in IEEE form is ~500 cycles of a
// ten-column Cholesky's critical path here, this ~100)
__device__ __forceinline__ double rsqrt64(double a)
{
    double r = __builtin_amdgcn_rsq(a);
    const double h = 0.5 * a;
    r = r * fma(-h * r, r, 1.5);
    r = r * fma(-h * r, r, 1.5);
    return r;
}

__device__ __forceinline__ double readlane_f64(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// sin / cos in double for the attitude angles of a walking robot: below 0.5 rad the Taylor series to theta^15 / theta^16 is
// exact to 1e-16 (sixteen FMAs instead of the library's range reduction); anything larger takes the library (wave-uniform)
__device__ __forceinline__ void sincos_small(double t, double *s, double *c)
{
    if (__builtin_amdgcn_ballot_w64(!(fabs(t) < 0.5)) != 0ull) { sincos(t, s, c); return; }
    const double u = t * t;
    double ps = -1.0 / 1307674368000.0;                                     // -1/15!
    ps = fma(ps, u, 1.0 / 6227020800.0); ps = fma(ps, u, -1.0 / 39916800.0); ps = fma(ps, u, 1.0 / 362880.0);
    ps = fma(ps, u, -1.0 / 5040.0); ps = fma(ps, u, 1.0 / 120.0); ps = fma(ps, u, -1.0 / 6.0); ps = fma(ps, u, 1.0);
    double pc = 1.0 / 20922789888000.0;                                     // 1/16!
    pc = fma(pc, u, -1.0 / 87178291200.0); pc = fma(pc, u, 1.0 / 479001600.0); pc = fma(pc, u, -1.0 / 3628800.0);
    pc = fma(pc, u, 1.0 / 40320.0); pc = fma(pc, u, -1.0 / 720.0); pc = fma(pc, u, 1.0 / 24.0); pc = fma(pc, u, -0.5);
    pc = fma(pc, u, 1.0);
    *s = ps * t;
    *c = pc;
}

__device__ __forceinline__ void rotation64(double tx, double ty, double tz, double *R)
{
    double sx, cx, sy, cy, sz, cz;
    sincos_small(tx, &sx, &cx); sincos_small(ty, &sy, &cy); sincos_small(tz, &sz, &cz);
    R[0] = cz * cy; R[1] = cz * sy * sx - sz * cx; R[2] = cz * sy * cx + sz * sx;
    R[3] = sz * cy; R[4] = sz * sy * sx + cz * cx; R[5] = sz * sy * cx - cz * sx;
    R[6] = -sy;     R[7] = cy * sx;                R[8] = cy * cx;
}

// al, be of the header comment for horizon steps i, l
__device__ __forceinline__ void alpha_beta(int i, int l, double dt, double &al, double &be)
{
    const int m = i > l ? i : l;
    int s = 0;
#pragma unroll
    for (int k = 1; k <= 5; k++) s += (k > m) ? (k - 1 - i) * (k - 1 - l) : 0;
    const double dt2 = dt * dt;
    al = dt2 * (double)(5 - m);
    be = dt2 * dt2 * (double)s;
}

// The non-zero contact bytes in leg order (the swing legs squeezed out).  A warm start is valid whenever THIS word is unchanged:
// lane v of an instance is (horizon step, rank among the force-carrying legs, component), so the previous solution and faces
// map onto the new legs rank by rank, the pyramids are the same for every leg, and the start stays feasible.  (Requiring the
// identical contact word made every gait phase change a cold start: ~38 active-set iterations against 2-4.)
__device__ __forceinline__ uint32_t contact_ranks(uint32_t c)
{
    uint32_t out = 0;
    int n = 0;
#pragma unroll
    for (int l = 0; l < 4; l++) {
        const uint32_t byte = (c >> (8 * l)) & 0xffu;
        if (byte) { out |= byte << (8 * n); n++; }
    }
    return out;
}

}  // namespace osm
