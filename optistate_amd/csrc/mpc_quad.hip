// mpc_quad.hip -- the convex-MPC force QP (misc/force_controller.py:70-225, solved by casadi + qpOASES at
// kalman_filter/kalman_filter.py:150) with SIXTEEN LANES PER QP, four QPs per wavefront (round 6).
//
// mpc_kernels.hip gives a QP a whole wavefront, one row of the face-restricted system per lane: a trot problem (30 variables) keeps
// 30 of 64 lanes busy, and the elimination broadcasts the pivot row with two v_readlane per entry -- three instructions per entry,
// 12 k VALU instructions per QP of which 7 % are useful flops (profiles/r05_pmc_mpc.md).  Here the same exact primal active-set
// algorithm (same faces, same ratio test, same multiplier rule: read mpc_kernels.hip's header for the mathematics) runs on the
// 16-lane DPP row: lane l of a row holds variables l and l + 16 (two rows of the system, NV <= 32: one or two stance legs), and
// every cross-lane term of the elimination is a broadcast FUSED into the FMA (v_fmac_f64_dpp ... row_newbcast: the one fp64 VOP2
// arithmetic instruction with a 64-bit DPP operand, kf_dense_rows.hpp).  One instruction updates an entry of the rows of FOUR
// problems: ~0.83 k elimination instructions per wavefront-solve = ~0.2 k per QP against 1.4 k.
// Three and four stance legs (45 / 60 variables: four rows per lane = 488 registers) stay on mpc_kernels.hip's instances.
//
// How a launch runs (profiles/r06_pmc_mpc.md for the measurements behind each piece):
//   * mpc_prep_kernel writes a per-problem RECORD (generators, linear term: everything constant over the iterations) -- four rows in lock
//     step, many wavefronts per SIMD;
//   * mpc_solve_quad_kernel: a persistent grid, each 16-lane ROW takes a problem index from a work counter, copies its record into LDS,
//     iterates until its KKT conditions hold, writes its outputs and takes the next index.  The solve of an iteration is unconditional
//     for the wavefront (a row without a problem rides along on an all-dead system); set-up, ratio test, multiplier check and outputs
//     run under the rows' own predicates.  One LDS exchange hands a leg-step's solution and point to its three lanes, which compute
//     every leg-step quantity redundantly and without branches;
//   * os_kf_mpc_run's form (POST_STEP): a row that finishes a QP marks its trajectory; wavefronts whose rows have run out of QPs take
//     tickets of sixteen consecutive trajectories and run the filter step of kf_dense_rows_kernel<BATCH> on them (drain phase).
// The hazard rule inline assembly must respect itself (hipcc does not look inside): a VALU write needs two wait states before a DPP
// operand reads the register (tools/isa_dpp_hazard_scan.py scans this file too).
// Development switches (tools/quad_variants.sh): -DOSQ_TS in-kernel stamps; -DOSQ_X_FIXED=n every problem exactly n iterations,
// -DOSQ_X_TRUNC=n at most n (what the tail costs), -DOSQ_OCC=1 one wavefront per SIMD.
#include "mpc_common.hpp"

#include <stdlib.h>
#include <type_traits>

#include "kf_dense_rows.hpp"
#include "kf_args.hpp"

namespace osq {

// Development build only (-DOSQ_TS): shader-clock stamps between the phases of an active-set iteration, summed by lane 0 of workgroup 0
#ifdef OSQ_TS
__shared__ unsigned long long osq_ts_sum[16];            // (LDS: a stamp costs one LDS round trip of lane 0, ~100 cycles)
__shared__ unsigned long long osq_ts_prev;
#define OSQ_STAMP(i)                                                                       \
    {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                         \
            const unsigned long long now = __builtin_readcyclecounter();                   \
            if ((i) > 0) osq_ts_sum[i] += now - osq_ts_prev;                               \
            osq_ts_prev = now;                                                             \
        }                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                 \
    }
#else
#define OSQ_STAMP(i)
#endif

using namespace osm;
using osm::static_for;
using osk::rows64::bc64;

// The per-problem RECORD (round 6): everything of a problem that does not change over its active-set iterations -- generators of the
// variables, the linear term per horizon step, R1 diag(w_theta) R1^T -- is computed ONCE by mpc_prep_kernel (all four rows of a
// wavefront in lock step, many wavefronts per SIMD) and written to global memory in exactly the layout of the head of a row's LDS
// block; a row of mpc_solve_quad_kernel copies it in with REC_CHUNKS 16-byte loads per lane.  Inside the persistent rows the same
// set-up (~2.2 k instructions, five dependent global round trips) ran once per ROW, not once per wavefront: the rows do not start
// their problems together.
constexpr int REC_BYTES = 1792;                          // 7 x (16 lanes x 16 bytes); the global stride of a record
template <int NV>
struct QuadRec {
    double gen0[NV][6];      // generators (a, b) of the variables
    double cwv[5][6];        // (cw | cv) of the linear term per horizon step
    double Cth[9];           // R1 diag(w_theta) R1^T
};
template <int NV>
constexpr int rec_chunks() { return ((int)sizeof(QuadRec<NV>) + 255) / 256; }

// per-QP LDS block
template <int NV>
struct QuadMem {
    union {
        QuadRec<NV> rec;
        double rec_image[rec_chunks<NV>() * 32];
    };
    double gent[NV][6];      // generators of the face coordinates
    double vec[32];          // per-variable exchange (solution / u / u0)
    double rowbuf[32];
    double S[32];            // the 5 x 6 per-step sums of form_dot
    double al[10];
    int code[10];
};
struct QuadShared {
    double ab[25][2];        // (al, be) of alpha_beta() at [5 i + l]: depends on dt only, one copy per wavefront
};

// acc += (src of lane S of this lane's 16-lane row) * m; the same register may be accumulator and source (rank-1 form)
template <int S, bool NOP>
__device__ __forceinline__ void fmac_self(double &acc, double m)
{
    if (NOP) asm volatile("s_nop 1\nv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(m), "n"(S));
    else asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(m), "n"(S));
}
template <int S, bool NOP>
__device__ __forceinline__ void fmac_other(double &acc, double src, double m)
{
    if (NOP) asm volatile("s_nop 1\nv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+&v"(acc) : "v"(src), "v"(m), "n"(S));
    else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+&v"(acc) : "v"(src), "v"(m), "n"(S));
}

// does any lane of this lane's 16-lane row satisfy p?
__device__ __forceinline__ bool any16(bool p, int lane)
{
    const unsigned long long m = __ballot(p);
    return ((m >> (lane & 48)) & 0xffffull) != 0ull;
}

// The forces cross from the row that solved the QP to the row that runs the filter step -- another CU, usually another XCD -- inside
// one launch.  The L2 of an XCD is not coherent with the others': an agent-scope release / acquire pair is a write-back of the whole
// L2 and an invalidate of it (measured: every finished QP paid one, every poll the other -- the launch took 2.6 ms instead of 1.1).
// Here only the twelve force values and the mark travel, with agent-scope (sc1) stores and loads that go through to memory
// themselves; `s_waitcnt vmcnt(0)` orders the mark behind them.
__device__ __forceinline__ void store_agent(float *p, float v)
{
    __hip_atomic_store(reinterpret_cast<uint32_t *>(p), __builtin_bit_cast(uint32_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float load_agent(const float *p)
{
    return __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const uint32_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
template <int NST>
struct Quad {
    static constexpr int NV = 15 * NST, NPS = 3 * NST, NLS = 5 * NST, VPL = NV > 16 ? 2 : 1, RCH = rec_chunks<15 * NST>();
    typedef QuadMem<NV> Mem;

    struct Var {              // one of the lane's variables
        int v, i, c, ls;      // variable, horizon step, component, leg-step
        bool pad;
    };
    struct Lane {
        int lane, l;
        Var var[VPL];
    };
    static __device__ __forceinline__ Lane this_lane() { return lane_of(threadIdx.x & 63); }
    // (the lane's constants again from an opaque copy of the lane index: at two wavefronts per SIMD hipcc keeps ~25 of these loop
    // invariants -- LDS addresses of the leg-step's entries, component selectors -- in scratch and reloads them in every iteration
    // behind an s_waitcnt vmcnt(0); recomputed they are a dozen integer instructions)
    static __device__ __forceinline__ Lane again(const Lane &L0)
    {
        int lane = L0.lane;
        asm volatile("" : "+v"(lane));
        return lane_of(lane);
    }
    static __device__ __forceinline__ Lane lane_of(int lane)
    {
        Lane L;
        L.lane = lane;
        L.l = L.lane & 15;
#pragma unroll
        for (int h = 0; h < VPL; h++) {
            Var &V = L.var[h];
            const int v = L.l + 16 * h;
            V.pad = v >= NV;
            V.v = V.pad ? NV - 1 : v;
            V.i = V.v / NPS; V.c = V.v % 3; V.ls = V.v / 3;
        }
        return L;
    }

    // z-vectors of a face generator g at horizon step i against the columns of step l: val(w) = zA . a_w + zB . b_w.
    // Cg = (R1 diag(w_theta) R1^T) g[0..2] and the lane's five (al, be) pairs do not depend on l: formed once per generator
    struct ZCtx { double Cg[3], ab[5][2]; };
    static __device__ __forceinline__ ZCtx zctx(const Mem &M, const QuadShared &Sh, const double *g, int i)
    {
        ZCtx Z;
        double C[9];
#pragma unroll
        for (int r = 0; r < 9; r++) C[r] = M.rec.Cth[r];
#pragma unroll
        for (int l = 0; l < 5; l++) { Z.ab[l][0] = Sh.ab[5 * i + l][0]; Z.ab[l][1] = Sh.ab[5 * i + l][1]; }
#pragma unroll
        for (int r = 0; r < 3; r++) Z.Cg[r] = C[3 * r] * g[0] + C[3 * r + 1] * g[1] + C[3 * r + 2] * g[2];
        return Z;
    }
    static __device__ __forceinline__ void zvec(const MpcParams &P, const ZCtx &Z, const double *g, int l, double *zA, double *zB)
    {
        const double al = Z.ab[l][0], be = Z.ab[l][1];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            zA[r] = al * P.w[6 + r] * g[r] + be * Z.Cg[r];
            zB[r] = (al * P.w[9 + r] + be * P.w[3 + r]) * g[3 + r];
        }
    }

    // sum_w form(g_h, G[w]) vec[w] for the lane's variables (mpc_kernels.hip form_dot: the sum over a step's columns factors into
    // per-step six-vector sums, formed by thirty lanes -- here fifteen lanes, two sums each).  One LDS round trip for the sums, one for
    // reading all thirty back (a rolled loop over the steps was five dependent round trips).
    static __device__ __forceinline__ void form_dot(const Lane &L, const MpcParams &P, const QuadShared &Sh, const double (*gg)[6], const double (*G)[6],
                                                    const double *vec, Mem &M, double *out)
    {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int e = L.l + 16 * q;
            const int ec = e < 30 ? e : 29, l = ec / 6, r = ec % 6;
            double gv[NPS], vv[NPS];
#pragma unroll
            for (int j = 0; j < NPS; j++) { gv[j] = G[NPS * l + j][r]; vv[j] = vec[NPS * l + j]; }
            double sum = 0.0;
#pragma unroll
            for (int j = 0; j < NPS; j++) sum = fma(gv[j], vv[j], sum);
            if (e < 30) M.S[e] = sum;
        }
        __builtin_amdgcn_wave_barrier();
        double Sv[30];
#pragma unroll
        for (int e = 0; e < 30; e++) Sv[e] = M.S[e];
#pragma unroll
        for (int h = 0; h < VPL; h++) {
            const ZCtx Z = zctx(M, Sh, gg[h], L.var[h].i);
            double acc = 0.0;
#pragma unroll
            for (int l = 0; l < 5; l++) {
                double zA[3], zB[3];
                zvec(P, Z, gg[h], l, zA, zB);
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    acc = fma(zA[r], Sv[6 * l + r], acc);
                    acc = fma(zB[r], Sv[6 * l + 3 + r], acc);
                }
            }
            out[h] = acc;
        }
        __builtin_amdgcn_wave_barrier();
    }

    // Minimiser of the QP restricted to the faces (sx, sy, sz per variable = of its leg-step); returns the lane's components IN FACE
    // COORDINATES (iterate_row maps them to forces in the same LDS exchange that hands a leg-step's point to its three lanes).
    // the face of a variable's leg-step, packed: (sx + 1) | (sy + 1) << 2 | sz << 4 (the warm-start format); by value: arrays handed
    // around by pointer stay in scratch memory
    struct Faces { int f[VPL]; };
    static __device__ __forceinline__ int fsx(int f) { return (f & 3) - 1; }
    static __device__ __forceinline__ int fsy(int f) { return ((f >> 2) & 3) - 1; }
    static __device__ __forceinline__ int fsz(int f) { return (f >> 4) & 3; }
    static __device__ __forceinline__ int fpack(int sx, int sy, int sz) { return (sx + 1) | ((sy + 1) << 2) | (sz << 4); }
    struct Sol { double u[VPL]; };

    static __device__ __forceinline__ Sol solve_face(const Lane &L0, const MpcParams &P, const QuadShared &Sh, Mem &M, const Faces F)
    {
        const Lane L = again(L0);
        OSQ_STAMP(0)
        int sx[VPL], sy[VPL], sz[VPL];
#pragma unroll
        for (int h = 0; h < VPL; h++) { sx[h] = fsx(F.f[h]); sy[h] = fsy(F.f[h]); sz[h] = fsz(F.f[h]); }
        Sol out;
        bool live[VPL];
        double g[VPL][6], tt[VPL], u0[VPL];
        // the three generators of the lane's leg-steps in ONE batch of reads, selects instead of branches (a branch per variable kind was
        // a dependent LDS round trip each)
        double g3[VPL][3][6];
#pragma unroll
        for (int h = 0; h < VPL; h++)
#pragma unroll
            for (int q = 0; q < 3; q++)
#pragma unroll
                for (int r = 0; r < 6; r++) g3[h][q][r] = M.rec.gen0[3 * L.var[h].ls + q][r];
#pragma unroll
        for (int h = 0; h < VPL; h++) {
            const Var &V = L.var[h];
            const double fx = sx[h] * P.mu, fy = sy[h] * P.mu;
            const bool isz = V.c == 2;
            live[h] = (isz ? sz[h] == SZ_FREE : (sz[h] != SZ_ZERO && (V.c == 0 ? sx[h] : sy[h]) == 0)) && !V.pad;
            tt[h] = isz ? 1.0 + P.mu * P.mu * (double)(sx[h] * sx[h] + sy[h] * sy[h]) : 1.0;
#pragma unroll
            for (int r = 0; r < 6; r++) {
                const double gz = g3[h][2][r] + fx * g3[h][0][r] + fy * g3[h][1][r];
                const double gxy = V.c == 0 ? g3[h][0][r] : g3[h][1][r];
                g[h][r] = live[h] ? (isz ? gz : gxy) : 0.0;
            }
            // the fixed part of the face (fz = fz_max faces)
            u0[h] = (sz[h] == SZ_MAX && !V.pad) ? (isz ? P.fzmax : (double)(V.c == 0 ? sx[h] : sy[h]) * P.mu * P.fzmax) : 0.0;
        }
        bool anyu = false;
#pragma unroll
        for (int h = 0; h < VPL; h++) anyu = anyu || u0[h] != 0.0;
        const bool any_u0 = __ballot(anyu) != 0ull;                    // (wave-uniform: the sums below are zero for a QP without such a face)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int h = 0; h < VPL; h++) {
            if (!L.var[h].pad) {
#pragma unroll
                for (int r = 0; r < 6; r++) M.gent[L.var[h].v][r] = g[h][r];
                M.vec[L.var[h].v] = u0[h];
            }
        }
        __builtin_amdgcn_wave_barrier();

        OSQ_STAMP(1)                                 // face generators -> LDS
        // ---- rows of the reduced system: A[h][w], w < NV, and the right-hand side in A[h][NV] ----
        double A[VPL][NV + 1];
        double dot[VPL] = {};
        if (any_u0) form_dot(L, P, Sh, g, M.rec.gen0, M.vec, M, dot);
        ZCtx Z[VPL];
#pragma unroll
        for (int h = 0; h < VPL; h++) {
            Z[h] = zctx(M, Sh, g[h], L.var[h].i);
            const double *cwv = M.rec.cwv[L.var[h].i];
            const double rhs = -(g[h][0] * cwv[0] + g[h][1] * cwv[1] + g[h][2] * cwv[2] + g[h][3] * cwv[3] + g[h][4] * cwv[4] + g[h][5] * cwv[5]) - dot[h];
            A[h][NV] = live[h] ? rhs : 0.0;
        }
        // a dead slot's generator is zero, so its row and column are zero here; the diagonal (r + ..., or 1 for a dead slot) is added
        // where the elimination reads the pivot.  Column w + 1's generator is requested before column w's products are formed (hipcc
        // left to itself waits for each column's three reads right behind them: ~110 exposed LDS round trips per solve)
        {
            double zA[VPL][3], zB[VPL][3], gn[6];
#pragma unroll
            for (int r = 0; r < 6; r++) gn[r] = M.gent[0][r];
            static_for<0, NV>([&](auto wc) {
                constexpr int w = decltype(wc)::value;
                if constexpr (w % NPS == 0) {
#pragma unroll
                    for (int h = 0; h < VPL; h++) zvec(P, Z[h], g[h], w / NPS, zA[h], zB[h]);
                }
                double gw[6];
#pragma unroll
                for (int r = 0; r < 6; r++) gw[r] = gn[r];
                if constexpr (w + 1 < NV) {
#pragma unroll
                    for (int r = 0; r < 6; r++) gn[r] = M.gent[w + 1][r];
                }
#pragma unroll
                for (int h = 0; h < VPL; h++)
                    A[h][w] = zA[h][0] * gw[0] + zA[h][1] * gw[1] + zA[h][2] * gw[2] + zB[h][0] * gw[3] + zB[h][1] * gw[4] + zB[h][2] * gw[5];
                if constexpr (w % NPS == NPS - 1) __builtin_amdgcn_sched_barrier(0);
            });
        }
        OSQ_STAMP(2)                                 // right-hand side + rows
        // ---- forward elimination: pivot k lives in lane k & 15 of the row, half k >> 4; its row is broadcast inside the FMA.  Compile-time
        // recursion (static_for), not `#pragma unroll` + a switch over the lane select: the DPP control is an immediate, and the
        // unroller gives up on 30 x 31 / 2 bodies of sixteen cases each (the arrays then live in scratch) ----
        double dsel[VPL], dinv[VPL];
#pragma unroll
        for (int h = 0; h < VPL; h++) { dsel[h] = live[h] ? P.rw * tt[h] : 1.0; dinv[h] = 1.0; }
        // (an opaque copy of the lane index: the ~90 lane predicates of the pivots below -- `row > k`, `l == k & 15` -- are otherwise
        // hoisted out of the solver loop as lane masks in SGPR pairs, spilled to VGPR lanes and fetched back with two v_readlane each;
        // recomputed, a predicate is one v_cmp against an inline constant)
        int ll = L.l;
        asm volatile("" : "+v"(ll));
        // (the NEXT pivot's inverse -- broadcast, v_rcp_f64, two Newton steps: a ~150-cycle dependent chain -- is started as soon as this
        // pivot's first entry has finalised that diagonal, and runs underneath the rest of this pivot's entries)
        double inv_next = rcp64(bc64<0>(A[0][0] + dsel[0]));
        static_for<0, NV>([&](auto kc) {
            constexpr int k = decltype(kc)::value, hk = k >> 4, kk = k & 15;
            const double inv = inv_next;
            if (ll == kk) dinv[hk] = inv;
            // rows below the pivot (the rows of the other half are all below or all above; a pad row is zero, its factor too)
            double nf[VPL];
#pragma unroll
            for (int h = 0; h < VPL; h++) nf[h] = 0.0;
            nf[hk] = ll > kk ? -A[hk][k] * inv : 0.0;
            if constexpr (VPL == 2 && hk == 0) nf[1] = -A[1][k] * inv;
            // entry j of every row below the pivot: the upper half first (it reads the pivot row's register, which the lower half's
            // own update then rewrites: a DPP operand must not have been written by the two preceding instructions)
            static_for<k + 1, NV + 1>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr bool first = j == k + 1;
                if constexpr (VPL == 2 && hk == 0) {
                    fmac_other<kk, first>(A[VPL - 1][j], A[0][j], nf[VPL - 1]);
                    fmac_self<kk, false>(A[0][j], nf[0]);
                } else {
                    fmac_self<kk, first>(A[hk][j], nf[hk]);
                }
                if constexpr (first && k + 1 < NV) {
                    constexpr int k1 = k + 1, hk1 = k1 >> 4, kk1 = k1 & 15;
                    inv_next = rcp64(bc64<kk1>(A[hk1][k1] + dsel[hk1]));
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        OSQ_STAMP(3)                                 // elimination
        // ---- back substitution ----
        double r[VPL];
#pragma unroll
        for (int h = 0; h < VPL; h++) r[h] = A[h][NV];
        static_for<0, NV>([&](auto ic) {
            constexpr int k = NV - 1 - decltype(ic)::value, hk = k >> 4, kk = k & 15;
            const double wk = bc64<kk>(r[hk] * dinv[hk]);
#pragma unroll
            for (int h = 0; h <= hk; h++)
                if (ll + 16 * h < k) r[h] = fma(-A[h][k], wk, r[h]);
            __builtin_amdgcn_sched_barrier(0);
        });
        OSQ_STAMP(4)                                 // back substitution
        // (a row's right-hand side is final once its own step has passed -- later steps touch the rows above -- so every lane forms its
        // own component at the end: the product the step broadcast, without a select per step)
#pragma unroll
        for (int h = 0; h < VPL; h++) out.u[h] = r[h] * dinv[h];
        return out;
    }

    // Per-row solver state: a 16-lane row owns ONE problem at a time and fetches the next from the batch's work counter when it has
    // converged -- the four rows of a wavefront do not wait for each other's iteration counts (measured on the bench's data: mean
    // 4.1 iterations per QP, mean of the maximum over four consecutive QPs 7.7; tools/mpc_iter_stats.py).
    struct Row {
        bool has, exhausted, first, done, converged;
        int b, iters, nx_raw;       // nx_raw (lane 0 of the row): the work-counter value reserved for the row's NEXT problem, requested a problem ahead
        uint32_t cbits;
        bool stance[VPL];
        Faces F;
        double u[VPL];
    };

    // The record of problem b (mpc_prep_kernel; every row of the wavefront is here): into the row's LDS block, then out to a.rec.
    template <typename MemT>
    static __device__ __forceinline__ void prep_row(const Lane &L, const MpcArgs &a, const MpcParams &P, MemT &M, double *prob /* LDS [36] */, int b, uint32_t cbits)
    {
        const size_t B = (size_t)a.B;
        __builtin_amdgcn_wave_barrier();
        if (L.l < 12) {
            prob[L.l] = (double)a.x[(size_t)L.l * B + b];
            prob[12 + L.l] = (double)a.ref[(size_t)L.l * B + b];
            prob[24 + L.l] = (double)a.p[(size_t)L.l * B + b];
        }
        prep_compute(L, a.rec, P, M, prob, b, cbits);
    }
    // prob (LDS): x | ref | p of the problem in float64, written by the row's lanes before the call.  rec_base null: the record stays in
    // the row's LDS block (kf_mpc_rows_kernel builds it where the solver reads it)
    template <typename MemT>
    static __device__ __forceinline__ void prep_compute(const Lane &L0, double *rec_base, const MpcParams &P, MemT &M, const double *prob, int b, uint32_t cbits)
    {
        const Lane L = again(L0);
        __builtin_amdgcn_wave_barrier();
        int legs[4] = {0, 0, 0, 0};
        {
            int n = 0;
#pragma unroll
            for (int lg = 0; lg < 4; lg++) {
                if (((cbits >> (8 * lg)) & 0xffu) != 0u) {
                    if (n == 0) legs[0] = lg; else if (n == 1) legs[1] = lg; else if (n == 2) legs[2] = lg; else legs[3] = lg;
                    n++;
                }
            }
        }
        int leg[VPL];
#pragma unroll
        for (int h = 0; h < VPL; h++) {
            const int rank = (L.var[h].v % NPS) / 3;
            leg[h] = rank == 0 ? legs[0] : rank == 1 ? legs[1] : rank == 2 ? legs[2] : legs[3];
        }
        const double *x = prob, *ref = prob + 12, *p = prob + 24;
        double R0[9], R1[9];
        rotation64(x[0], x[1], x[2], R0);
        rotation64(ref[0], ref[1], ref[2], R1);
        if (L.l < 9) {
            const int r = L.l / 3, s = L.l % 3;
            double v = 0.0;
#pragma unroll
            for (int rr = 0; rr < 3; rr++)
#pragma unroll
                for (int ss = 0; ss < 3; ss++)
                    if (rr == r && ss == s) v = R1[3 * rr] * P.w[0] * R1[3 * ss] + R1[3 * rr + 1] * P.w[1] * R1[3 * ss + 1] + R1[3 * rr + 2] * P.w[2] * R1[3 * ss + 2];
            M.rec.Cth[L.l] = v;
        }
        // generators of the lane's variables: a = I_hat^-1 (R p_leg x e_c), b = e_c / m, with R of the variable's horizon step
#pragma unroll
        for (int h = 0; h < VPL; h++) {
            const Var &V = L.var[h];
            double Rm[9];
#pragma unroll
            for (int r = 0; r < 9; r++) Rm[r] = V.i == 0 ? R0[r] : R1[r];
            const double px = p[3 * leg[h]], py = p[3 * leg[h] + 1], pz = p[3 * leg[h] + 2];
            const double wx = Rm[0] * px + Rm[1] * py + Rm[2] * pz, wy = Rm[3] * px + Rm[4] * py + Rm[5] * pz, wz = Rm[6] * px + Rm[7] * py + Rm[8] * pz;
            double cr[3];                                  // pw x e_c
            if (V.c == 0) { cr[0] = 0.0; cr[1] = wz; cr[2] = -wy; }
            else if (V.c == 1) { cr[0] = -wz; cr[1] = 0.0; cr[2] = wx; }
            else { cr[0] = wy; cr[1] = -wx; cr[2] = 0.0; }
            double tb[3];                                  // I_hat^-1 = R diag(1/I) R^T
#pragma unroll
            for (int q = 0; q < 3; q++) tb[q] = (Rm[q] * cr[0] + Rm[3 + q] * cr[1] + Rm[6 + q] * cr[2]) * P.inv_inertia[q];
            if (!V.pad) {
#pragma unroll
                for (int r = 0; r < 3; r++) M.rec.gen0[V.v][r] = Rm[3 * r] * tb[0] + Rm[3 * r + 1] * tb[1] + Rm[3 * r + 2] * tb[2];
#pragma unroll
                for (int r = 0; r < 3; r++) M.rec.gen0[V.v][3 + r] = (r == V.c) ? P.inv_mass : 0.0;
            }
        }
        // linear term per horizon step (mpc_kernels.hip: closed forms in the step index): lanes 0..4 of the row compute step l's (cw | cv)
        if (L.l < 5) {
            const double dt = P.dt, dt2 = dt * dt;
            const double di = (double)L.l, Mm = 4.0 - di, n = Mm + 1.0;
            const double s1 = 0.5 * Mm * n, s2m = Mm * n * (2.0 * Mm + 1.0) * (1.0 / 6.0), s3m = s1 * s1;
            const double S1 = s1, S2 = s2m + di * s1, T1 = s1 + n * (di + 1.0), T2 = s2m + (di + 1.0) * s1;
            const double T3 = s3m + (2.0 * di + 1.0) * s2m + di * (di + 1.0) * s1;
            double e0[3], dd[3], wt[3];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const double r0w = R0[r] * x[6] + R0[3 + r] * x[7] + R0[6 + r] * x[8];
                const double r1w = R1[r] * x[6] + R1[3 + r] * x[7] + R1[6 + r] * x[8];
                e0[r] = x[r] + dt * r0w - ref[r];
                dd[r] = dt * r1w;
            }
#pragma unroll
            for (int r = 0; r < 3; r++) wt[r] = P.w[r] * (S1 * e0[r] + S2 * dd[r]);
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const double ew = x[6 + r] - ref[6 + r];
                M.rec.cwv[L.l][r] = n * dt * P.w[6 + r] * ew + dt2 * (R1[3 * r] * wt[0] + R1[3 * r + 1] * wt[1] + R1[3 * r + 2] * wt[2]);
                double sev = n * (x[9 + r] - ref[9 + r]);
                double ser = S1 * (x[3 + r] - ref[3 + r]) + dt * x[9 + r] * T2;
                if (r == 2) { sev += dt * P.gz * T1; ser += 0.5 * dt2 * P.gz * T3; }
                M.rec.cwv[L.l][3 + r] = dt * P.w[9 + r] * sev + dt2 * P.w[3 + r] * ser;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // the record image out: chunk c = 16 lanes x 16 bytes
        if (rec_base) {
            typedef double d2_t __attribute__((ext_vector_type(2)));
            d2_t *dst = reinterpret_cast<d2_t *>(reinterpret_cast<char *>(rec_base) + (size_t)b * REC_BYTES);
            const d2_t *src = reinterpret_cast<const d2_t *>(M.rec_image);
#pragma unroll
            for (int c = 0; c < RCH; c++) dst[c * 16 + L.l] = src[c * 16 + L.l];
        }
        __builtin_amdgcn_wave_barrier();
    }
    // the stance flags of the lane's variables from the contact word (load_row; kf_mpc_rows_kernel)
    static __device__ __forceinline__ void set_stance(const Lane &L0, uint32_t cbits, Row &R)
    {
        const Lane L = again(L0);
        int legs[4] = {0, 0, 0, 0};
        {
            int n = 0;
#pragma unroll
            for (int lg = 0; lg < 4; lg++) {
                if (((cbits >> (8 * lg)) & 0xffu) != 0u) {
                    if (n == 0) legs[0] = lg; else if (n == 1) legs[1] = lg; else if (n == 2) legs[2] = lg; else legs[3] = lg;
                    n++;
                }
            }
        }
#pragma unroll
        for (int h = 0; h < VPL; h++) {
            const int rank = (L.var[h].v % NPS) / 3;
            const int leg = rank == 0 ? legs[0] : rank == 1 ? legs[1] : rank == 2 ? legs[2] : legs[3];
            R.stance[h] = ((cbits >> (8 * leg)) & 0xffu) == 1u;      // any other non-zero value: unconstrained (force_controller.py:114-131)
        }
    }

    // problem b -> the row's LDS block and state (row-uniform control flow: every lane of the row is here): the record in, the stance
    // flags from the contact word, the warm-start record where it applies.  In two halves: every global read of the problem is
    // requested at once (request_row), load_row picks the data up, and only then does the caller reserve the row's NEXT index at the
    // work counter (its round trip travels under the problem's iterations): one exposed round trip per problem instead of three
    // (counter, contact word, record).
    typedef double d2_t __attribute__((ext_vector_type(2)));
    struct Pre {
        d2_t rc[RCH];
        uint32_t cb, wc;
        double wu[VPL];
        int wf[VPL];
    };
    static __device__ __forceinline__ Pre request_row(const Lane &L0, const MpcArgs &a, int b)
    {
        const Lane L = again(L0);
        Pre pre;
        pre.cb = a.contact[b];
        const d2_t *src = reinterpret_cast<const d2_t *>(reinterpret_cast<const char *>(a.rec) + (size_t)b * REC_BYTES);
#pragma unroll
        for (int c = 0; c < RCH; c++) pre.rc[c] = src[c * 16 + L.l];
        pre.wc = 0xffffffffu;
#pragma unroll
        for (int h = 0; h < VPL; h++) { pre.wu[h] = 0.0; pre.wf[h] = 0; }
        if (a.warm_u && !a.cold_in) {
            pre.wc = a.warm_contact[b];
#pragma unroll
            for (int h = 0; h < VPL; h++) {
                const int v = L.l + 16 * h;
                pre.wu[h] = a.warm_u[(size_t)b * 64 + v]; pre.wf[h] = a.warm_state[(size_t)b * 64 + v] & 0x3f;
            }
        }
        return pre;
    }
    static __device__ __forceinline__ void load_row(const Lane &L0, const MpcArgs &a, Mem &M, Row &R, int b, const Pre &pre)
    {
        const Lane L = again(L0);
        const uint32_t cbits = pre.cb;
        int legs[4] = {0, 0, 0, 0};
        {
            int n = 0;
#pragma unroll
            for (int lg = 0; lg < 4; lg++) {
                if (((cbits >> (8 * lg)) & 0xffu) != 0u) {
                    if (n == 0) legs[0] = lg; else if (n == 1) legs[1] = lg; else if (n == 2) legs[2] = lg; else legs[3] = lg;
                    n++;
                }
            }
        }
#pragma unroll
        for (int h = 0; h < VPL; h++) {
            const int rank = (L.var[h].v % NPS) / 3;
            const int leg = rank == 0 ? legs[0] : rank == 1 ? legs[1] : rank == 2 ? legs[2] : legs[3];
            R.stance[h] = ((cbits >> (8 * leg)) & 0xffu) == 1u;      // any other non-zero value: unconstrained (force_controller.py:114-131)
        }
        __builtin_amdgcn_wave_barrier();
        d2_t *dst = reinterpret_cast<d2_t *>(M.rec_image);
#pragma unroll
        for (int c = 0; c < RCH; c++) dst[c * 16 + L.l] = pre.rc[c];
        __builtin_amdgcn_wave_barrier();
        // cold start: stance (and unconstrained) legs free, swing legs zero; warm: the previous solve's point and faces when the
        // force-carrying legs' contact bytes are unchanged (the pyramids do not move: the old u stays feasible)
        const bool warm = a.warm_u && !a.cold_in && pre.wc != 0xffffffffu && contact_ranks(pre.wc) == contact_ranks(cbits);
#pragma unroll
        for (int h = 0; h < VPL; h++) {
            const int v = L.l + 16 * h;
            const bool w = warm && v < NV;
            R.F.f[h] = w ? pre.wf[h] : fpack(0, 0, SZ_FREE); R.u[h] = w ? pre.wu[h] : 0.0;
        }
        R.first = !warm; R.done = false; R.converged = false; R.iters = 0; R.b = b; R.cbits = cbits; R.has = true;
    }

    // the converged (or capped) problem's outputs in the reference's variable order (12 per horizon step, leg-major; swing legs zero)
    // the problem goes on in the second pass: its state into the warm arrays (the hand-over record), its index onto the todo list
    static __device__ __forceinline__ void hand_over_row(const Lane &L, const MpcArgs &a, Row &R)
    {
        const size_t B = (size_t)a.B;
        const int b = R.b;
#pragma unroll
        for (int h = 0; h < VPL; h++) {
            const int v = L.l + 16 * h;
            a.warm_u[(size_t)b * 64 + v] = L.var[h].pad ? 0.0 : R.u[h];
            a.warm_state[(size_t)b * 64 + v] = (uint8_t)R.F.f[h];
        }
        if (VPL == 1) { a.warm_u[(size_t)b * 64 + 16 + L.l] = 0.0; a.warm_state[(size_t)b * 64 + 16 + L.l] = 0x05; }
        // (lanes 32..63 of the wavefront-per-QP layout are pad lanes at one / two legs: their record is never read as a variable)
        if (L.l == 0) {
            a.warm_contact[b] = R.cbits;
            if (a.iters) a.iters[b] = R.iters;
            const int at = atomicAdd(&a.todo_count[NST - 1], 1);
            a.todo[(size_t)(NST - 1) * B + at] = b;
        }
        R.has = false;
    }

    template <bool SCOPED = false>
    static __device__ __forceinline__ void finish_row(const Lane &L0, const MpcArgs &a, Mem &M, Row &R)
    {
        const Lane L = again(L0);
        const size_t B = (size_t)a.B;
        const int b = R.b;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int h = 0; h < VPL; h++)
            if (!L.var[h].pad) M.vec[L.var[h].v] = R.u[h];
        __builtin_amdgcn_wave_barrier();
        // rank of each leg among the force-carrying ones (-1: swing), from the contact word: a few scalar-free selects
        int rank_of[4], nr = 0;
#pragma unroll
        for (int lg = 0; lg < 4; lg++) {
            const bool on = ((R.cbits >> (8 * lg)) & 0xffu) != 0u;
            rank_of[lg] = on ? nr : -1;
            nr += on ? 1 : 0;
        }
        // horizon step 0 (the forces the filter step wants) always, the other four steps when the caller asked for the whole control
        const int mmax = a.u_out ? 4 : 1;
        for (int m = 0; m < mmax; m++) {
            const int o = L.l + 16 * m;
            if (o < 60) {
                const int oi = o / 12, oleg = (o % 12) / 3, oc = o % 3;
                const int rank = oleg == 0 ? rank_of[0] : (oleg == 1 ? rank_of[1] : (oleg == 2 ? rank_of[2] : rank_of[3]));
                const float val = rank < 0 ? 0.f : (float)M.vec[oi * NPS + 3 * (rank < 0 ? 0 : rank) + oc];
                if (o < 12) { if (SCOPED) store_agent(&a.f_out[(size_t)o * B + b], val); else a.f_out[(size_t)o * B + b] = val; }
                if (a.u_out) a.u_out[(size_t)o * B + b] = val;
            }
        }
        if (a.warm_u) {
#pragma unroll
            for (int h = 0; h < VPL; h++) {
                const int v = L.l + 16 * h;
                a.warm_u[(size_t)b * 64 + v] = L.var[h].pad ? 0.0 : R.u[h];
                a.warm_state[(size_t)b * 64 + v] = (uint8_t)R.F.f[h];
            }
            if (L.l == 0) a.warm_contact[b] = R.cbits;
        }
        if (L.l == 0) {
            if (a.iters) a.iters[b] = R.iters;
            if (!R.converged) atomicOr(&a.status[b], 4);
        }
        __builtin_amdgcn_wave_barrier();
        R.has = false;
    }

    // One active-set iteration of the row's problem behind the (unconditional) solve.  S: the face-restricted minimiser in face
    // coordinates.  Every quantity of a leg-step (point, candidate point, step, ratio test, snap, multiplier test) is computed by ALL
    // THREE of its lanes from one exchange through LDS, without branches: the round-5 form (only the fz lane decides, every stage
    // hands its result on through LDS, six candidate ratios behind six branches) was ~100 dependent LDS waits = 12 k of an
    // iteration's 40 k cycles on a wavefront alone (in-kernel stamps, round 6).  Exchanges now: 1 (point + solution) + 1 (the row's
    // minimum ratio) per iteration, + 4 when the subspace minimiser is reached (gradient: 2, multipliers: 2).
    static __device__ __forceinline__ double pick3(const double (&a)[3], int c) { return c == 0 ? a[0] : (c == 1 ? a[1] : a[2]); }

    static __device__ __forceinline__ void iterate_row(const Lane &L0, const MpcParams &P, const QuadShared &Sh, Mem &M, Row &R, const Sol &S)
    {
        const Lane L = again(L0);
        Faces &F = R.F;
        double (&u)[VPL] = R.u;
        const bool (&stance)[VPL] = R.stance;
        constexpr double EPS = 1e-11, TOL = 1e-12;
        R.iters++;
        // ---- the leg-step's solution (face coordinates) and current point to each of its lanes ----
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int h = 0; h < VPL; h++)
            if (!L.var[h].pad) { M.vec[L.var[h].v] = S.u[h]; M.rowbuf[L.var[h].v] = u[h]; }
        __builtin_amdgcn_wave_barrier();
        double s3[VPL][3], u3[VPL][3], c3[VPL][3];
#pragma unroll
        for (int h = 0; h < VPL; h++)
#pragma unroll
            for (int q = 0; q < 3; q++) { s3[h][q] = M.vec[3 * L.var[h].ls + q]; u3[h][q] = M.rowbuf[3 * L.var[h].ls + q]; }
        // face coordinates -> forces: the candidate point of the face
#pragma unroll
        for (int h = 0; h < VPL; h++) {
            const int sx = fsx(F.f[h]), sy = fsy(F.f[h]), sz = fsz(F.f[h]);
            const double fz = sz == SZ_MAX ? P.fzmax : (sz == SZ_ZERO ? 0.0 : s3[h][2]);
            c3[h][0] = sx != 0 ? (double)sx * P.mu * fz : (sz == SZ_ZERO ? 0.0 : s3[h][0]);
            c3[h][1] = sy != 0 ? (double)sy * P.mu * fz : (sz == SZ_ZERO ? 0.0 : s3[h][1]);
            c3[h][2] = fz;
        }
        if (R.first) {
            // (row-uniform) clamp the stance-free minimiser into the pyramids; the faces come from the clamps.  Nothing clamped: optimal.
            R.first = false;
            bool clamped = false;
#pragma unroll
            for (int h = 0; h < VPL; h++) {
                const Var &V = L.var[h];
                double fx = c3[h][0], fy = c3[h][1], fz = c3[h][2];
                int sx = 0, sy = 0, sz = SZ_FREE;
                bool cl = false;
                if (fz <= 0.0) { sz = SZ_ZERO; fx = fy = fz = 0.0; cl = true; }
                else {
                    if (fz >= P.fzmax) { sz = SZ_MAX; fz = P.fzmax; cl = true; }
                    const double lim = P.mu * fz;
                    if (fx >= lim) { sx = 1; fx = lim; cl = true; } else if (fx <= -lim) { sx = -1; fx = -lim; cl = true; }
                    if (fy >= lim) { sy = 1; fy = lim; cl = true; } else if (fy <= -lim) { sy = -1; fy = -lim; cl = true; }
                }
                const bool act = stance[h] && !V.pad;
                F.f[h] = act ? fpack(sx, sy, sz) : F.f[h];
                const double own = V.pad ? 0.0 : pick3(c3[h], V.c);
                u[h] = act ? (V.c == 0 ? fx : (V.c == 1 ? fy : fz)) : own;
                clamped = clamped || (act && cl);
            }
            if (!any16(clamped, L.lane)) { R.done = true; R.converged = true; }
            return;
        }
        // ---- ratio test of the lane's leg-steps (all six candidates, no branches) ----
        double d3[VPL][3];
#pragma unroll
        for (int h = 0; h < VPL; h++) {
            const Var &V = L.var[h];
            const int sx = fsx(F.f[h]), sy = fsy(F.f[h]), sz = fsz(F.f[h]);
#pragma unroll
            for (int q = 0; q < 3; q++) d3[h][q] = c3[h][q] - u3[h][q];
            const double fx = u3[h][0], fy = u3[h][1], fz = u3[h][2], dx = d3[h][0], dy = d3[h][1], dz = d3[h][2];
            const bool on = stance[h] && sz != SZ_ZERO;
            const double num[6] = {fz, P.fzmax - fz, P.mu * fz - fx, P.mu * fz + fx, P.mu * fz - fy, P.mu * fz + fy};
            const double den[6] = {-dz, dz, dx - P.mu * dz, -dx - P.mu * dz, dy - P.mu * dz, -dy - P.mu * dz};
            const bool ok[6] = {on && sz == SZ_FREE, on && sz == SZ_FREE, on && sx == 0, on && sx == 0, on && sy == 0, on && sy == 0};
            double best = 2.0; int code = 0;
#pragma unroll
            for (int q = 0; q < 6; q++) {
                const bool v = ok[q] && den[q] > EPS;
                const double al = fmax(0.0, num[q] * rcp64(v ? den[q] : 1.0));
                const bool take = v && al < best;
                best = take ? al : best; code = take ? q + 1 : code;
            }
            if (V.c == 2 && !V.pad) { M.al[V.ls] = best; M.code[V.ls] = code; }
        }
        __builtin_amdgcn_wave_barrier();
        double amin = 1.0; int lsmin = -1, cmin = 0;
        {
            double alq[NLS]; int cdq[NLS];
#pragma unroll
            for (int q = 0; q < NLS; q++) { alq[q] = M.al[q]; cdq[q] = M.code[q]; }
#pragma unroll
            for (int q = 0; q < NLS; q++)
                if (alq[q] < amin) { amin = alq[q]; lsmin = q; cmin = cdq[q]; }
        }
        // the step, the blocking face and the snap onto the face equalities: all three components on every lane of the leg-step
#pragma unroll
        for (int h = 0; h < VPL; h++) {
            const Var &V = L.var[h];
#pragma unroll
            for (int q = 0; q < 3; q++) u3[h][q] += amin * d3[h][q];
            int sx = fsx(F.f[h]), sy = fsy(F.f[h]), sz = fsz(F.f[h]);
            {
                int nx = sx, ny = sy, nz = sz;
                if (cmin == 1) { nx = 0; ny = 0; nz = SZ_ZERO; }
                else if (cmin == 2) nz = SZ_MAX;
                else if (cmin == 3) nx = 1;
                else if (cmin == 4) nx = -1;
                else if (cmin == 5) ny = 1;
                else if (cmin == 6) ny = -1;
                const bool hit = lsmin >= 0 && V.ls == lsmin && !V.pad;
                sx = hit ? nx : sx; sy = hit ? ny : sy; sz = hit ? nz : sz;
                F.f[h] = fpack(sx, sy, sz);
            }
            const double fz = sz == SZ_ZERO ? 0.0 : (sz == SZ_MAX ? P.fzmax : u3[h][2]);
            const double own = pick3(u3[h], V.c);
            const int sgn = V.c == 0 ? sx : sy;
            const double un = V.c == 2 ? fz : (sz == SZ_ZERO ? 0.0 : (sgn != 0 ? (double)sgn * P.mu * fz : own));
            u[h] = V.pad ? u[h] : (stance[h] ? un : own);
        }
#ifdef OS_MPC_DBG
        if (L.lane == 0) printf("quad it %d ratio: amin %.6e lsmin %d cmin %d\n", R.iters, amin, lsmin, cmin);
#endif
        if (lsmin >= 0) return;

        // ---- subspace minimiser reached (row-uniform): multiplier signs ----
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int h = 0; h < VPL; h++)
            if (!L.var[h].pad) M.vec[L.var[h].v] = u[h];
        __builtin_amdgcn_wave_barrier();
        double g0[VPL][6], fd[VPL];
#pragma unroll
        for (int h = 0; h < VPL; h++)
#pragma unroll
            for (int r = 0; r < 6; r++) g0[h][r] = M.rec.gen0[L.var[h].v][r];
        form_dot(L, P, Sh, g0, M.rec.gen0, M.vec, M, fd);
#pragma unroll
        for (int h = 0; h < VPL; h++) {
            const double *cwv = M.rec.cwv[L.var[h].i];
            const double q0 = g0[h][0] * cwv[0] + g0[h][1] * cwv[1] + g0[h][2] * cwv[2] + g0[h][3] * cwv[3] + g0[h][4] * cwv[4] + g0[h][5] * cwv[5];
            const double grad = 2.0 * (fd[h] + P.rw * u[h] + q0);
            if (!L.var[h].pad) M.rowbuf[L.var[h].v] = grad;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int h = 0; h < VPL; h++) {
            const Var &V = L.var[h];
            const int sx = fsx(F.f[h]), sy = fsy(F.f[h]), sz = fsz(F.f[h]);
            const double gx = M.rowbuf[3 * V.ls], gy = M.rowbuf[3 * V.ls + 1], gz = M.rowbuf[3 * V.ls + 2];
            double best = TOL; int code = 0;
            // apex: optimal iff the gradient lies in the dual cone; otherwise release along the steepest edge ray
            const double val = gz - P.mu * fabs(gx) - P.mu * fabs(gy);
            const bool apex = sz == SZ_ZERO;
            if (apex && -val > best) { best = -val; code = 16 + (gx > 0.0 ? 0 : 1) + (gy > 0.0 ? 0 : 2); }
            if (!apex && sx == 1 && gx > best) { best = gx; code = 1; }
            if (!apex && sx == -1 && -gx > best) { best = -gx; code = 1; }
            if (!apex && sy == 1 && gy > best) { best = gy; code = 2; }
            if (!apex && sy == -1 && -gy > best) { best = -gy; code = 2; }
            const double lamU = -gz - sx * P.mu * gx - sy * P.mu * gy;
            if (!apex && sz == SZ_MAX && -lamU > best) { best = -lamU; code = 3; }
            if (!stance[h]) { best = TOL; code = 0; }
            if (V.c == 2 && !V.pad) { M.al[V.ls] = best; M.code[V.ls] = code; }
        }
        __builtin_amdgcn_wave_barrier();
        double rmax = TOL; int lsr = -1, cr = 0;
        {
            double alq[NLS]; int cdq[NLS];
#pragma unroll
            for (int q = 0; q < NLS; q++) { alq[q] = M.al[q]; cdq[q] = M.code[q]; }
#pragma unroll
            for (int q = 0; q < NLS; q++)
                if (cdq[q] != 0 && alq[q] > rmax) { rmax = alq[q]; lsr = q; cr = cdq[q]; }
        }
#ifdef OS_MPC_DBG
        if (L.lane == 0) printf("quad it %d mult: rmax %.6e lsr %d cr %d\n", R.iters, rmax, lsr, cr);
#endif
        if (lsr < 0) { R.done = true; R.converged = true; return; }
#pragma unroll
        for (int h = 0; h < VPL; h++) {
            int sx = fsx(F.f[h]), sy = fsy(F.f[h]), sz = fsz(F.f[h]);
            if (cr == 1) sx = 0;
            else if (cr == 2) sy = 0;
            else if (cr == 3) sz = SZ_FREE;
            else { sx = (cr & 1) ? 1 : -1; sy = (cr & 2) ? 1 : -1; sz = SZ_FREE; }
            const bool hit = L.var[h].ls == lsr && !L.var[h].pad;
            F.f[h] = hit ? fpack(sx, sy, sz) : F.f[h];
        }
    }
};

// The records of a batch (see QuadRec): one 16-lane row per problem, the four rows of a wavefront in lock step.
template <int NV>
struct PrepMem {
    union {
        QuadRec<NV> rec;
        double rec_image[rec_chunks<NV>() * 32];
    };
    double prob[36];         // x | ref | p (float64)
};
template <int NST>
__global__ __launch_bounds__(64) void mpc_prep_kernel(const MpcArgs a)
{
    typedef Quad<NST> Q;
    __shared__ PrepMem<Q::NV> Mp[4];
    const typename Q::Lane L = Q::this_lane();
    PrepMem<Q::NV> &M = Mp[L.lane >> 4];
    const int b = blockIdx.x * 4 + (L.lane >> 4);
    if (b >= a.n) return;
    const uint32_t cb = a.contact[b];
    const int nst = ((cb & 0xffu) != 0) + (((cb >> 8) & 0xffu) != 0) + (((cb >> 16) & 0xffu) != 0) + (((cb >> 24) & 0xffu) != 0);
    if (nst != NST) return;
    for (int i = L.l; i < Q::RCH * 32; i += 16) M.rec_image[i] = 0.0;
    Q::prep_row(L, a, a.prm, M, M.prob, b, cb);
}

// One workgroup = one wavefront = four independent 16-lane rows.  A row takes a problem index from the launch's work counter, sets the
// problem up, iterates until its KKT conditions hold, writes its outputs and takes the next index; problems with another number of
// force-carrying legs are skipped (their instance handles them).  The grid is sized to fill the chip, not to the batch.
#ifndef OSQ_OCC
#define OSQ_OCC 2            // wavefronts per SIMD the register budget is sized for
#endif
// counters of an instance (zeroed by the host before the launch)
enum : int { CNT_WORK = 0, CNT_TICKET = 1, CNT_INTS = 8 };
// what a launch does beyond the QPs: nothing | marks finished trajectories (an instance that is followed by another one) | marks them
// and runs the filter step of EVERY trajectory of the batch in its drain phase (the last instance of the step)
enum : int { POST_NONE = 0, POST_MARK = 1, POST_STEP = 2 };

// the trajectory of a finished QP is marked done (row-uniform), behind the row's force stores
__device__ __forceinline__ void mark_done(int l, const PostArgs &post, int b)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (l == 0) __hip_atomic_store(&post.done[b], post.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One predict_mpc + update step of trajectory b on this 16-lane row: kf_dense_rows_kernel<BATCH, dense> at T = 1 (kf_dense_rows.hip),
// the same building blocks in the same order -- the results are bit-identical to the separate launch.  qs / rs: Q and R in float64 (LDS).
// TV: the step index t differs between the rows of the wavefront (kf_mpc_rows_kernel).  A buffer descriptor has to be wave-uniform: with the
// step in its base address every stream access would be a waterfall loop over the rows' distinct steps (it was: 150 spilled
// registers at two wavefronts per SIMD); the descriptors then span the whole [T][rows][B] stream and the step travels in the lane's offset.
template <bool TV = false>
__device__ __forceinline__ float kf_step_row(const osk::KfRunArgs &a, const double *qs, const double *rs, int b, bool live, int r, int t, double ed /* expm1(dt) */)
{
    using namespace osk;
    namespace rw = osk::rows64;
    const int rr = r < 12 ? r : 11, am = rw::row_measurement(r);
    const size_t B = (size_t)a.B;
    const uint32_t voff = (uint32_t)b * 4u, rowB = (uint32_t)a.B * 4u;
    const KfConst &k = a.k;
    const double *qrow = qs + rr * NS, *rrow_b = rs + (am >= 0 ? am : NM) * NM;
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
    StepIn in;
    float bref[3];
    if constexpr (TV) {
        const uint32_t span = (uint32_t)a.T * rowB;                    // (the host holds T x 12 x B x 4 below 2^32)
        rsrc_t rp = make_rsrc(a.p, 12 * span), rd = make_rsrc(a.dp, 12 * span), ri = make_rsrc(a.imu, 6 * span), rc = make_rsrc(a.contact, span);
        const uint32_t v12 = voff + (uint32_t)t * 12u * rowB, v6 = voff + (uint32_t)t * 6u * rowB, v1 = voff + (uint32_t)t * rowB;
#pragma unroll
        for (int i = 0; i < 12; i++) { in.p[i] = buf_load_nt(rp, v12, i * rowB); in.dp[i] = buf_load_nt(rd, v12, i * rowB); }
#pragma unroll
        for (int i = 0; i < 6; i++) in.imu[i] = buf_load_nt(ri, v6, i * rowB);
        in.contact = buf_load_u32_nt(rc, v1, 0);
        rsrc_t rb = make_rsrc(a.body_ref, 12 * span);
#pragma unroll
        for (int i = 0; i < 3; i++) bref[i] = buf_load(rb, v12, i * rowB);
    } else {
        load_step(a, t, voff, rowB, in);
        rsrc_t rb = make_rsrc(a.body_ref + (size_t)t * 12 * B, 12 * rowB);
#pragma unroll
        for (int i = 0; i < 3; i++) bref[i] = buf_load(rb, voff, i * rowB);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) in.f[i] = load_agent(a.f + ((size_t)t * 12 + i) * B + b);      // (written in this launch, by another CU)
    float x[NS], z[NM], pw[12];
    rw::gather_state(xr, x);
    int status = rw::front_row<true>(x, xr, P, in, bref, k, ed, qrow, one, r, z, pw);
    float xn = x[0];
#pragma unroll
    for (int i = 1; i < NS; i++) xn = (rr == i) ? x[i] : xn;
    if (a.p_rot_out && live && r < 12) {
        float pv = pw[0];
#pragma unroll
        for (int i = 1; i < 12; i++) pv = (r == i) ? pw[i] : pv;
        a.p_rot_out[((size_t)t * 12 + r) * B + b] = pv;
    }
    double xd = (double)xn;
    {
        double K[NM], rrow[NM];
#pragma unroll
        for (int q = 0; q < NM; q++) rrow[q] = rrow_b[q];
        status |= rw::update_batch_row(xd, P, z, rrow, K, am);
    }
    xr = (float)xd;
    if (!(xr * 0.f == 0.f)) status |= 2;
    status |= __shfl_xor(status, 1, 64); status |= __shfl_xor(status, 2, 64);
    status |= __shfl_xor(status, 4, 64); status |= __shfl_xor(status, 8, 64);
    if (live && r < 12) {
        a.x_out[((size_t)t * 12 + r) * B + b] = xr;
        a.x[(size_t)r * B + b] = xr;
#pragma unroll
        for (int j = 0; j < NS; j++) a.P[(size_t)(r * NS + j) * B + b] = (float)P[j];
        if (r == 0 && status) atomicOr(&a.status[b], status);
    }
    return xr;                                         // lane r < 12: component r of the new state (what a.x now holds)
}

template <int NST, int POST>
__global__ __launch_bounds__(64, OSQ_OCC) void mpc_solve_quad_kernel(const MpcArgs a, int *__restrict__ cnt, const PostArgs post)
{
    typedef Quad<NST> Q;
    int *const counter = cnt + CNT_WORK;
    __shared__ typename Q::Mem Mq[4];
    __shared__ QuadShared Sh;
    const typename Q::Lane L = Q::this_lane();
    if (L.lane < 25) {
        double al, be;
        alpha_beta(L.lane / 5, L.lane % 5, a.prm.dt, al, be);
        Sh.ab[L.lane][0] = al; Sh.ab[L.lane][1] = be;
    }
    __builtin_amdgcn_wave_barrier();
#ifdef OSQ_TS
    if (threadIdx.x < 16) osq_ts_sum[threadIdx.x] = 0;
#endif
    typename Q::Mem &M = Mq[L.lane >> 4];
    const size_t B = (size_t)a.B;
    typename Q::Row R;
    R.has = false; R.exhausted = false; R.first = false; R.done = true; R.converged = true; R.b = 0; R.iters = 0; R.cbits = 0;
    R.nx_raw = 0;
    if (L.l == 0) R.nx_raw = atomicAdd(counter, 1);
#pragma unroll
    for (int h = 0; h < Q::VPL; h++) { R.F.f[h] = Q::fpack(0, 0, SZ_ZERO); R.u[h] = 0.0; R.stance[h] = false; }
    // a row that has never had a problem rides along in the solve with an all-dead system (identity rows): its LDS block holds zeros
    for (int i = L.l; i < (int)(sizeof(typename Q::Mem) / 8); i += 16) reinterpret_cast<double *>(&M)[i] = 0.0;
    __builtin_amdgcn_wave_barrier();
    for (;;) {
        if (!R.has && !R.exhausted) {
            // (row-uniform) next problem of this leg count; its index was requested when the previous problem's data had arrived (the work
            // counter's round trip travels under a whole problem's iterations)
            for (;;) {
                const int nb = __builtin_amdgcn_update_dpp(0, R.nx_raw, 0x150, 0xf, 0xf, true);      // row_newbcast:0
                if (nb >= a.n) { R.exhausted = true; break; }
                const typename Q::Pre pre = Q::request_row(L, a, nb);
                const uint32_t cb = pre.cb;
                const int nst = ((cb & 0xffu) != 0) + (((cb >> 8) & 0xffu) != 0) + (((cb >> 16) & 0xffu) != 0) + (((cb >> 24) & 0xffu) != 0);
                if (NST == 1 && nst == 0) {          // no leg on the ground: all forces zero (force_controller.py:114-123)
                    if (L.l < 12) { if (POST != POST_NONE) store_agent(&a.f_out[(size_t)L.l * B + nb], 0.f); else a.f_out[(size_t)L.l * B + nb] = 0.f; }
                    if (a.u_out)
                        for (int o = L.l; o < 60; o += 16) a.u_out[(size_t)o * B + nb] = 0.f;
                    if (L.l == 0 && a.iters) a.iters[nb] = 0;
                    if (L.l == 0 && a.warm_contact) a.warm_contact[nb] = cb;
                    if constexpr (POST != POST_NONE) mark_done(L.l, post, nb);
                }
                // (the row's next index: requested BEHIND everything that waits for this problem's reads -- the vector memory counter
                // is in order, and across the branches hipcc's wait placement falls back to "all of them")
                const bool mine = nst == NST;
                if (mine) Q::load_row(L, a, M, R, nb, pre);
                R.nx_raw = 0;
                if (L.l == 0) R.nx_raw = atomicAdd(counter, 1);
                if (!mine) continue;
                OSQ_STAMP(10)                        // (the row has its next problem)
                break;
            }
        }
        if (__ballot(R.has) == 0ull) break;
        OSQ_STAMP(7)                                 // next problems: record + warm start
        OSQ_STAMP(8)                                 // (nothing: the cost of a stamp)
        const typename Q::Sol S = Q::solve_face(L, a.prm, Sh, M, R.F);
        OSQ_STAMP(5)                                 // (solve_face returns)
        if (R.has) {
            Q::iterate_row(L, a.prm, Sh, M, R, S);
            OSQ_STAMP(9)                             // ratio test / multipliers
#ifdef OSQ_X_FIXED
            R.done = R.iters >= OSQ_X_FIXED; R.converged = true;
#endif
#ifdef OSQ_X_TRUNC
            if (R.iters >= OSQ_X_TRUNC) { R.done = true; R.converged = true; }
#endif
            if (R.done || R.iters >= a.max_iter) {
                Q::template finish_row<POST != POST_NONE>(L, a, M, R);
                if constexpr (POST != POST_NONE) mark_done(L.l, post, R.b);
            }
            else if (a.cap > 0 && R.iters >= a.cap) Q::hand_over_row(L, a, R);
        }
        OSQ_STAMP(6)                                 // outputs of finished problems
#ifdef OSQ_TS
        if (blockIdx.x == 0 && threadIdx.x == 0) osq_ts_sum[0] += 1;
#endif
    }
    if constexpr (POST == POST_STEP) {
        // ---- drain phase: this wavefront has solved everything it will; it now takes blocks of four CONSECUTIVE trajectories (the
        // [rows][B] streams want neighbours together: in completion order a step touched ~380 cache lines per trajectory and the
        // launch took 2.8 ms) from a ticket counter, each row waits for its trajectory's mark and the four filter steps run in
        // lock step.  The wait is for a row that is running (this launch's, or an earlier instance's that has completed): no
        // co-residency assumption; the poll limit only guards against a lost device (status bit 6 of the trajectory).
        double *qs = reinterpret_cast<double *>(&Mq[0]), *rs = qs + osk::NS * osk::NS;      // (the rows' LDS blocks are idle now)
        const double ed = expm1((double)post.kf.k.dt);
        __builtin_amdgcn_wave_barrier();
        for (int i = L.lane; i < osk::NS * osk::NS; i += 64) qs[i] = (double)post.qr[i];
        for (int i = L.lane; i < (osk::NM + 2) * osk::NM; i += 64) rs[i] = i < osk::NM * osk::NM ? (double)post.qr[144 + i] : 0.0;
        __builtin_amdgcn_wave_barrier();
        for (;;) {
            // a ticket = sixteen consecutive trajectories, four at a time: the four sub-steps touch the same cache lines of every stream
            // (four trajectories per ticket fetched each line on four different XCDs)
            int t = 0;
            if (L.lane == 0) t = atomicAdd(&cnt[CNT_TICKET], 1);
            t = __builtin_amdgcn_readfirstlane(t);
            if (16 * t >= a.n) break;
            for (int j = 0; j < 4; j++) {
                const int tb = 16 * t + 4 * j + (L.lane >> 4);
                if (16 * t + 4 * j >= a.n) break;
                const bool have = tb < a.n;
                if (have) {
                    unsigned polls = 0;
                    while (__hip_atomic_load(&post.done[tb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != post.seq) {
                        if (++polls > (1u << 24)) { if (L.l == 0) atomicOr(&post.kf.status[tb], 64); break; }
                        __builtin_amdgcn_s_sleep(8);
                    }
                }
                kf_step_row(post.kf, qs, rs, have ? tb : a.n - 1, have, L.l, 0, ed);
            }
        }
    }
#ifdef OSQ_TS
    if (blockIdx.x == 0 && threadIdx.x == 0 && NST == 2) {
        const unsigned long long n = osq_ts_sum[0] ? osq_ts_sum[0] : 1;
        printf("quad<2> cycles per wave-iteration (%llu iterations): faces %llu | rows %llu | elimination %llu | back-substitution %llu | ratio/multipliers %llu | outputs %llu | "
               "counter+contact %llu | record + warm start %llu | one stamp %llu\n",
               n, osq_ts_sum[1] / n, osq_ts_sum[2] / n, osq_ts_sum[3] / n, osq_ts_sum[4] / n, osq_ts_sum[5] / n + osq_ts_sum[9] / n, osq_ts_sum[6] / n, osq_ts_sum[10] / n,
               osq_ts_sum[7] / n, osq_ts_sum[8] / n);
    }
#endif
}

// =====================================================================================================================
// estimate_state_mpc with a 16-lane ROW per trajectory for ALL T steps (round 6; batches between the wavefront-per-trajectory persistent
// kernel and the launch sequence).  The launch sequence pays, per step, for the few problems with 30-45 active-set iterations: at
// B = 65,536 two concurrent parts hide that drain, at 8-24 k trajectories a launch IS the drain (a part of 8,192 problems has one
// problem per row).  Here nothing is synchronised between steps: a row takes a trajectory from a work counter and runs QP -> filter
// step -> next QP for all its steps -- the record of the next QP is built in the row's LDS block from the state the filter step has
// just produced (prep_compute, no global round trip), the warm start stays in the row's registers, x and P travel through global
// memory as in the launch sequence (P rounded to float32 at every step: the sequence's numbers up to the last bits of the record --
// the same source inlined into another kernel contracts differently).  The four rows of a wavefront are at different steps of
// different trajectories; the solve of an iteration is shared, set-up / outputs / filter step run under the rows' predicates.
// Every trajectory at every step must carry force on NST legs or on none (the host checks the leg-count histogram).
// =====================================================================================================================
struct RowsArgs {
    osk::KfRunArgs kf;             // all T steps' streams, x / P in place, x_out, p_rot_out, status
    MpcParams prm;
    float *f_out;                  // [T][12][B] (kf.f points at the same array)
    int32_t *iters;                // [T][B] or null
    int max_iter, cold;
    const float *qr;               // Q 144 | R 100
};

// OCC: wavefronts per SIMD the register budget is sized for -- 1 (no spills: the filter step alone wants ~250 registers beside the row's
// QP state) while a wavefront per SIMD covers the batch (B <= 16 x CUs: measured at B = 4,096 3.4e7 against 2.3e7 steps/s), else 2
template <int NST, int OCC>
__global__ __launch_bounds__(64, OCC) void kf_mpc_rows_kernel(const RowsArgs a, int *__restrict__ counter)
{
    typedef Quad<NST> Q;
    __shared__ typename Q::Mem Mq[4];
    __shared__ QuadShared Sh;
    __shared__ double qrs[osk::NS * osk::NS + (osk::NM + 2) * osk::NM];
    const typename Q::Lane L = Q::this_lane();
    if (L.lane < 25) {
        double al, be;
        alpha_beta(L.lane / 5, L.lane % 5, a.prm.dt, al, be);
        Sh.ab[L.lane][0] = al; Sh.ab[L.lane][1] = be;
    }
    double *qs = qrs, *rs = qrs + osk::NS * osk::NS;
    for (int i = L.lane; i < osk::NS * osk::NS; i += 64) qs[i] = (double)a.qr[i];
    for (int i = L.lane; i < (osk::NM + 2) * osk::NM; i += 64) rs[i] = i < osk::NM * osk::NM ? (double)a.qr[144 + i] : 0.0;
    typename Q::Mem &M = Mq[L.lane >> 4];
    for (int i = L.l; i < (int)(sizeof(typename Q::Mem) / 8); i += 16) reinterpret_cast<double *>(&M)[i] = 0.0;
    __builtin_amdgcn_wave_barrier();
    const int T = a.kf.T, nB = a.kf.B;
    const size_t B = (size_t)a.kf.B;
    const double ed = expm1((double)a.kf.k.dt);
    typename Q::Row R;
    R.has = false; R.exhausted = false; R.first = false; R.done = true; R.converged = true; R.b = 0; R.iters = 0; R.cbits = 0; R.nx_raw = 0;
#pragma unroll
    for (int h = 0; h < Q::VPL; h++) { R.F.f[h] = Q::fpack(0, 0, SZ_ZERO); R.u[h] = 0.0; R.stance[h] = false; }
    int b = -1, t = 0;                      // the row's trajectory and its current step
    uint32_t prev_c = 0xffffffffu;          // contact word of the previous step's solve (warm start valid while the leg ranks are unchanged)
    float xr = 0.f;                         // lane l < 12: component l of the trajectory's state
    bool airborne = false;                  // this step has no leg on the ground: no QP, the filter step only
    MpcArgs ma;                             // the per-step view finish_row writes through (only the fields it reads)
    ma.B = a.kf.B; ma.u_out = nullptr; ma.warm_u = nullptr; ma.warm_state = nullptr; ma.warm_contact = nullptr; ma.status = a.kf.status;
    for (;;) {
        if (!R.has && !R.exhausted) {
            for (;;) {                      // (row-uniform)
                if (b < 0 || t >= T) {
                    int nx = 0;
                    if (L.l == 0) nx = atomicAdd(counter, 1);
                    nx = __builtin_amdgcn_update_dpp(0, nx, 0x150, 0xf, 0xf, true);
                    if (nx >= nB) { R.exhausted = true; break; }
                    b = nx; t = 0; prev_c = 0xffffffffu;
                    xr = a.kf.x[(size_t)(L.l < 12 ? L.l : 11) * B + b];
                }
                const uint32_t cb = a.kf.contact[(size_t)t * B + b];
                const int nst = ((cb & 0xffu) != 0) + (((cb >> 8) & 0xffu) != 0) + (((cb >> 16) & 0xffu) != 0) + (((cb >> 24) & 0xffu) != 0);
                if (nst == 0) {             // no leg on the ground: all forces zero (force_controller.py:114-123); the row rides through one
                                            // solve and takes the filter step below (one copy of it in the kernel)
                    if (L.l < 12) a.f_out[((size_t)t * 12 + L.l) * B + b] = 0.f;
                    if (L.l == 0 && a.iters) a.iters[(size_t)t * B + b] = 0;
                    prev_c = cb;
                    airborne = true; R.has = true;
                    break;
                }
                // the record of this step's QP from the state the row holds, straight into the row's LDS block
                double *prob = &M.gent[0][0];
                __builtin_amdgcn_wave_barrier();
                if (L.l < 12) {
                    prob[L.l] = (double)xr;
                    prob[12 + L.l] = (double)a.kf.body_ref[((size_t)t * 12 + L.l) * B + b];
                    prob[24 + L.l] = (double)a.kf.p[((size_t)t * 12 + L.l) * B + b];
                }
                Q::prep_compute(L, nullptr, a.prm, M, prob, b, cb);
                Q::set_stance(L, cb, R);
                const bool warm = !a.cold && prev_c != 0xffffffffu && contact_ranks(prev_c) == contact_ranks(cb);
#pragma unroll
                for (int h = 0; h < Q::VPL; h++) {
                    const bool keep = warm && L.l + 16 * h < Q::NV;
                    R.F.f[h] = keep ? R.F.f[h] : Q::fpack(0, 0, SZ_FREE); R.u[h] = keep ? R.u[h] : 0.0;
                }
                R.first = !warm; R.done = false; R.converged = false; R.iters = 0; R.b = b; R.cbits = cb; R.has = true;
                break;
            }
        }
        if (__ballot(R.has) == 0ull) break;
        const typename Q::Sol S = Q::solve_face(L, a.prm, Sh, M, R.F);
        if (R.has) {
            if (!airborne) Q::iterate_row(L, a.prm, Sh, M, R, S);
            if (airborne || R.done || R.iters >= a.max_iter) {
                if (!airborne) {
                    ma.f_out = a.f_out + (size_t)t * 12 * B;
                    ma.iters = a.iters ? a.iters + (size_t)t * B : nullptr;
                    Q::template finish_row<false>(L, ma, M, R);         // forces of horizon step 0, iteration count, status bit 2
                    prev_c = R.cbits;
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (the filter step reads this row's forces back)
                xr = kf_step_row<true>(a.kf, qs, rs, b, true, L.l, t, ed);
                t++;
                R.has = false; airborne = false;
            }
        }
    }
}

}  // namespace osq

// the instances for one / two force-carrying legs over the whole batch (mpc_kernels.hip launch_instances); counters: CNT_INTS zeroed
// ints per instance; post: the filter step of the finished trajectories inside the same launch (null: QP only)
void os_mpc_launch_quad(const osm::MpcArgs &a, uint32_t nst_mask, int *counters, int cu_count, hipStream_t s, const osm::PostArgs *post)
{
    // two wavefronts per SIMD fill the chip; fewer rows than problems never hurts (a row takes the next index), more would idle
    const int rows = (a.n + 3) / 4, full = cu_count * 4 * OSQ_OCC;
    const dim3 grid(rows < full ? rows : full), block(64);
    if (nst_mask & 3u) hipLaunchKernelGGL(osq::mpc_prep_kernel<1>, dim3(rows), block, 0, s, a);
    if (nst_mask & 4u) hipLaunchKernelGGL(osq::mpc_prep_kernel<2>, dim3(rows), block, 0, s, a);
    if (post) {
        // the last instance of the step runs the filter step of every trajectory; one in front of it only marks its own
        if ((nst_mask & 3u) && (nst_mask & 4u)) {
            hipLaunchKernelGGL((osq::mpc_solve_quad_kernel<1, osq::POST_MARK>), grid, block, 0, s, a, counters, *post);
            hipLaunchKernelGGL((osq::mpc_solve_quad_kernel<2, osq::POST_STEP>), grid, block, 0, s, a, counters + osq::CNT_INTS, *post);
        } else if (nst_mask & 3u) hipLaunchKernelGGL((osq::mpc_solve_quad_kernel<1, osq::POST_STEP>), grid, block, 0, s, a, counters, *post);
        else if (nst_mask & 4u) hipLaunchKernelGGL((osq::mpc_solve_quad_kernel<2, osq::POST_STEP>), grid, block, 0, s, a, counters + osq::CNT_INTS, *post);
    } else {
        static const osm::PostArgs none = {};
        if (nst_mask & 3u) hipLaunchKernelGGL((osq::mpc_solve_quad_kernel<1, osq::POST_NONE>), grid, block, 0, s, a, counters, none);
        if (nst_mask & 4u) hipLaunchKernelGGL((osq::mpc_solve_quad_kernel<2, osq::POST_NONE>), grid, block, 0, s, a, counters + osq::CNT_INTS, none);
    }
}

// kf_mpc_rows_kernel over the whole batch (mpc_kernels.hip os_kf_mpc_run); counter: one zeroed int
void os_mpc_launch_rows(const osq::RowsArgs &a, int nst, int *counter, int cu_count, hipStream_t s)
{
    const int rows = (a.kf.B + 3) / 4, simds = cu_count * 4;
    const dim3 block(64);
    if (rows <= simds) {
        if (nst == 1) hipLaunchKernelGGL((osq::kf_mpc_rows_kernel<1, 1>), dim3(rows), block, 0, s, a, counter);
        else hipLaunchKernelGGL((osq::kf_mpc_rows_kernel<2, 1>), dim3(rows), block, 0, s, a, counter);
    } else {
        const dim3 grid(rows < 2 * simds ? rows : 2 * simds);
        if (nst == 1) hipLaunchKernelGGL((osq::kf_mpc_rows_kernel<1, 2>), grid, block, 0, s, a, counter);
        else hipLaunchKernelGGL((osq::kf_mpc_rows_kernel<2, 2>), grid, block, 0, s, a, counter);
    }
}
