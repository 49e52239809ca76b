// mpc_kernels.hip -- the convex-MPC force QP of the reference (misc/force_controller.py:70-162, solved there by
// casadi + qpOASES at kalman_filter/kalman_filter.py:150) as a batched exact active-set solver on gfx950.
//
// One wavefront per QP (3 forces x 5 horizon steps per leg that carries force: swing legs are eliminated, so 15 / 30 /
// 45 / 60 variables), lane v = variable v, float64 throughout
// (the Hessian's spectrum spans 1e-6 ... 0.17: the 1e-6 R-term alone fixes the force distribution in the directions the
// body wrench does not see, which float32 cannot resolve).
//
// Structure that keeps it cheap:
//   * H and q are never stored.  With a_v = I_hat^-1 (R p_leg x e_c) and b_v = e_c / m the generators of variable v,
//       H[v][w] = a_v^T (al W_w + be R1 W_th R1^T) a_w + b_v^T (al W_v + be W_r) b_w + r delta_vw,
//       al = dt^2 (5 - max(i,l)),  be = dt^4 sum_{k>max(i,l)} (k-1-i)(k-1-l)       (i, l = horizon steps of v, w)
//     because x_k depends on the inputs only through the per-step wrench (derivation in docs/DESIGN_history_r01-r04.md section 4.5); a row is rebuilt
//     from the six-vectors in LDS whenever it is needed (one horizon-step block at a time: 3.5 - 13 KB of LDS per
//     wavefront, two to three wavefronts per SIMD).
//   * the feasible set is a product of truncated friction pyramids, so an active set is a FACE per (leg, step):
//     (sx, sy) in {-1, 0, +1} (fx = +-mu fz active) and sz in {free, fz = 0 (apex, f = 0), fz = fz_max}.  Restricting
//     the QP to a face only replaces each generator by a linear combination of its leg's three generators, so the
//     reduced Hessian has the same closed form: no KKT system, no projections of a stored matrix.
//   * the face-restricted system (dead slots = identity rows) is solved by Gaussian elimination with one row per lane.  The system
//     is symmetric: entry j of pivot row k is element k of row j, ONE register across the lanes, which a 16-lane DPP row reads
//     inside the FMA (round 6; rounds 1-5 broadcast the pivot row through LDS, then by two v_readlane per entry).  Dead pivots are
//     skipped wave-uniformly when most are dead; with most alive the elimination is straight-line code with the next pivot's
//     inverse computed under the current pivot's entries (solve_face).
// Algorithm: cold start (stance legs free, swing legs zero) -> subspace minimiser -> clamp into the pyramids -> primal
// active-set iteration (ratio test adds the blocking constraint, multiplier signs release one constraint; at the apex the
// dual-cone test g_z >= mu(|g_x|+|g_y|) decides and the steepest edge ray is released) until the KKT conditions hold.
// In normal walking the clamp never triggers and one solve is the answer.
#include "launch.hpp"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <mutex>

#include "kf_dense_rows.hpp"
#include "kf_args.hpp"
#include "mpc_common.hpp"

namespace osm {

// Development build only (-DOSM_TS): shader-clock stamps between the phases of an active-set iteration of the wavefront-per-QP solver and
// around the persistent kernel's two halves, summed by lane 0 of workgroup 0 (printed at the end of kf_mpc_persistent_kernel)
#ifdef OSM_TS
__shared__ unsigned long long osm_ts_sum[16];
__shared__ unsigned long long osm_ts_prev;
#define OSM_STAMP(i)                                                                       \
    {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                         \
            const unsigned long long now = __builtin_readcyclecounter();                   \
            if ((i) > 0) osm_ts_sum[i] += now - osm_ts_prev;                               \
            osm_ts_prev = now;                                                             \
        }                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                 \
    }
#else
#define OSM_STAMP(i)
#endif

template <int NV_>
struct WaveMemT {
    static constexpr int LW = NV_ <= 32 ? 32 : 64;   // lanes that own a row
    double gen0[NV_][6];     // generators (a, b) of the variables
    double gent[NV_][6];     // generators of the face coordinates
    double rows[NV_ / 5][LW]; // staging of one horizon-step block of the reduced system's rows: rows[w][lane] (built by a rolled loop, then read into registers)
    double rowbuf[64];
    double vec[64];          // broadcast vector (solution / u / u0)
    double al[NLSMAX];
    int code[NLSMAX];
    double ab[25][2];        // (al, be) of alpha_beta() for the horizon-step pairs (i, l) at [5 i + l]: one LDS read instead of ~25
                             // integer / conversion instructions per block of the row formation
};

struct LaneCtx {
    int lane, v, i, leg, c, ls;
    bool pad;
    double Cth[9];                         // R1 diag(w_theta) R1^T (wave-uniform: kept in SGPRs via readfirstlane)
};

// One horizon-step block (NPS columns, step l) of the reduced system's row for face generator g (6) at horizon step L.i
// against the face generators M.gent[w], staged in M.rows[j][lane] (the caller then reads it into registers).
// UNR columns at a time: fully unrolled, hipcc hoists all generator loads to the top and spills them; rolled, every column is
// one exposed LDS round trip.  Three where the instantiation has registers to spare (the stand-alone instances and the
// one-wave-per-SIMD persistent kernel: B = 8 50.4 -> 49.3 us per step, B = 65,536 2.71e7 -> 2.90e7 steps/s), one in the
// two-waves-per-SIMD persistent kernel, whose spills grow otherwise (B = 4096: 1.17e7 against 1.04e7 steps/s).
template <int NPS, int UNR, typename WaveMem>
__device__ __forceinline__ void form_row_block(const LaneCtx &L, const MpcParams &P, const double *g, int l, WaveMem &M)
{
    // opaque copy of the step index: otherwise the (al, be) x weight products of all five blocks are hoisted out of the
    // active-set loop and spilled
    int li = L.i;
    asm volatile("" : "+v"(li));
    const double al = M.ab[5 * li + l][0], be = M.ab[5 * li + l][1];
    double zA[3], zB[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        zA[r] = al * P.w[6 + r] * g[r] + be * (L.Cth[3 * r] * g[0] + L.Cth[3 * r + 1] * g[1] + L.Cth[3 * r + 2] * g[2]);
        zB[r] = (al * P.w[9 + r] + be * P.w[3 + r]) * g[3 + r];
    }
#pragma clang loop unroll_count(UNR)
    for (int j = 0; j < NPS; j++) {
        const int w = NPS * l + j;
        const double *gw = M.gent[w];
        // a dead slot's generator is zero, so its row is zero here; the diagonal (r + ..., or 1 for a dead slot) is added
        // where the elimination reads the pivot -- not by a compare and two selects per column
        const double val = zA[0] * gw[0] + zA[1] * gw[1] + zA[2] * gw[2] + zB[0] * gw[3] + zB[1] * gw[4] + zB[2] * gw[5];
        if (L.lane < WaveMem::LW) M.rows[j][L.lane] = val;
    }
}

// The whole row at once, straight into registers (UNR = 0; round 6): column w + 1's generator is requested before column w's products are
// formed, the (al, be) pairs of the lane's step and C_theta g once per row.  In-kernel stamps at the reference's shape (B = 8, 3.1
// iterations per step): the staged form above cost 7.6 k cycles per iteration -- more than the elimination (4.4 k).  Same expressions,
// same order: the same bits.  Needs ~70 registers beside the row: the one-wave-per-SIMD persistent kernel's solver call uses it.
template <int NST, typename WaveMem>
__device__ __forceinline__ void form_rows_direct(const LaneCtx &L, const MpcParams &P, const double *g, WaveMem &M, double *A)
{
    constexpr int NV = 15 * NST, NPS = 3 * NST;
    int li = L.i;
    asm volatile("" : "+v"(li));
    double ab[5][2];
#pragma unroll
    for (int l = 0; l < 5; l++) { ab[l][0] = M.ab[5 * li + l][0]; ab[l][1] = M.ab[5 * li + l][1]; }
    double Cg[3];
#pragma unroll
    for (int r = 0; r < 3; r++) Cg[r] = L.Cth[3 * r] * g[0] + L.Cth[3 * r + 1] * g[1] + L.Cth[3 * r + 2] * g[2];
    double zA[3], zB[3], gn[6];
#pragma unroll
    for (int r = 0; r < 6; r++) gn[r] = M.gent[0][r];
    static_for<0, NV>([&](auto wc) {
        constexpr int w = decltype(wc)::value;
        if constexpr (w % NPS == 0) {
            const double al = ab[w / NPS][0], be = ab[w / NPS][1];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                zA[r] = al * P.w[6 + r] * g[r] + be * Cg[r];
                zB[r] = (al * P.w[9 + r] + be * P.w[3 + r]) * g[3 + r];
            }
        }
        double gw[6];
#pragma unroll
        for (int r = 0; r < 6; r++) gw[r] = gn[r];
        if constexpr (w + 1 < NV) {
#pragma unroll
            for (int r = 0; r < 6; r++) gn[r] = M.gent[w + 1][r];
        }
        A[w] = zA[0] * gw[0] + zA[1] * gw[1] + zA[2] * gw[2] + zB[0] * gw[3] + zB[1] * gw[4] + zB[2] * gw[5];
        if constexpr (w % NPS == NPS - 1) __builtin_amdgcn_sched_barrier(0);
    });
}

// sum_w form(g, G[w]) * vec[w].  form() depends on the column w only through its generator and, via (al, be), its horizon
// step, so the sum over the NPS columns of a step factors: sum_w (zA . a_w + zB . b_w) vec[w] = zA . A_l + zB . B_l with the
// per-step sums (A_l | B_l) = sum_{w in step l} G[w] vec[w].  Thirty lanes form the 5 x 6 sums (NPS loads in flight each),
// then every lane needs five steps x six products instead of 15 NST columns x seven behind a rolled, LDS-latency-bound
// loop (~10 k -> ~2 k cycles per multiplier check; the rounding differs from the column-by-column sum at the 1e-16 level).
template <int NPS, bool FLAT = false, typename WaveMem>
__device__ __forceinline__ double form_dot(const LaneCtx &L, const MpcParams &P, const double *g, const double (*G)[6], const double *vec,
                                           WaveMem &M)
{
    double *S = &M.rows[0][0];                 // scratch: the row staging area is idle here (>= 96 doubles)
    __builtin_amdgcn_wave_barrier();
    if (L.lane < 30) {
        const int l = L.lane / 6, r = L.lane % 6;
        double gv[NPS], vv[NPS];
#pragma unroll
        for (int j = 0; j < NPS; j++) { gv[j] = G[NPS * l + j][r]; vv[j] = vec[NPS * l + j]; }
        double sum = 0.0;
#pragma unroll
        for (int j = 0; j < NPS; j++) sum = fma(gv[j], vv[j], sum);
        S[L.lane] = sum;
    }
    __builtin_amdgcn_wave_barrier();
    // opaque copy of the step index (see form_row_block): keeps the five (al, be) sets out of the active-set loop's live range
    int li = L.i;
    asm volatile("" : "+v"(li));
    double acc = 0.0;
    if constexpr (FLAT) {
        // (all thirty sums and the five (al, be) pairs in one batch of reads: the rolled loop below is five dependent round trips)
        double Sv[30], ab[5][2], Cg[3];
#pragma unroll
        for (int e = 0; e < 30; e++) Sv[e] = S[e];
#pragma unroll
        for (int l = 0; l < 5; l++) { ab[l][0] = M.ab[5 * li + l][0]; ab[l][1] = M.ab[5 * li + l][1]; }
#pragma unroll
        for (int r = 0; r < 3; r++) Cg[r] = L.Cth[3 * r] * g[0] + L.Cth[3 * r + 1] * g[1] + L.Cth[3 * r + 2] * g[2];
#pragma unroll
        for (int l = 0; l < 5; l++) {
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const double zA = ab[l][0] * P.w[6 + r] * g[r] + ab[l][1] * Cg[r];
                const double zB = (ab[l][0] * P.w[9 + r] + ab[l][1] * P.w[3 + r]) * g[3 + r];
                acc = fma(zA, Sv[6 * l + r], acc);
                acc = fma(zB, Sv[6 * l + 3 + r], acc);
            }
        }
    } else {
#pragma clang loop unroll(disable)              // unrolled, the thirty sums are all read up front: spills in the 168-register instances
        for (int l = 0; l < 5; l++) {
            const double al = M.ab[5 * li + l][0], be = M.ab[5 * li + l][1];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const double zA = al * P.w[6 + r] * g[r] + be * (L.Cth[3 * r] * g[0] + L.Cth[3 * r + 1] * g[1] + L.Cth[3 * r + 2] * g[2]);
                const double zB = (al * P.w[9 + r] + be * P.w[3 + r]) * g[3 + r];
                acc = fma(zA, S[6 * l + r], acc);
                acc = fma(zB, S[6 * l + 3 + r], acc);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    return acc;
}

#ifndef OS_MPC_DPP_ELIM
#define OS_MPC_DPP_ELIM(NST) true
#endif
#ifndef OS_MPC_PIVOT_AHEAD
#define OS_MPC_PIVOT_AHEAD(NST) ((NST) <= 2 ? 4 : 1000)      // straight-line elimination from this many sixths of the variables live (never for 3-4 legs)
#endif
// acc += (src of lane S of this lane's 16-lane DPP row) * m.  Inline assembly answers for its own hazard: a VALU write needs two wait
// states before a DPP operand reads the register (NOP on the first use after src was written; tools/isa_dpp_hazard_scan.py).
template <int S, bool NOP>
__device__ __forceinline__ void fmac_bcast(double &acc, double src, double m)
{
    if (NOP) asm volatile("s_nop 1\nv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+&v"(acc) : "v"(src), "v"(m), "n"(S));
    else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+&v"(acc) : "v"(src), "v"(m), "n"(S));
}
typedef unsigned osm_u2 __attribute__((ext_vector_type(2)));
// (a, b) -> a's even DPP rows in both rows of each pair, b's odd rows likewise
__device__ __forceinline__ void swap16(double &a, double &b)
{
    const unsigned long long ua = __builtin_bit_cast(unsigned long long, a), ub = __builtin_bit_cast(unsigned long long, b);
    const osm_u2 lo = __builtin_amdgcn_permlane16_swap((unsigned)ua, (unsigned)ub, false, false);
    const osm_u2 hi = __builtin_amdgcn_permlane16_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
    a = __builtin_bit_cast(double, (unsigned long long)lo.x | ((unsigned long long)hi.x << 32));
    b = __builtin_bit_cast(double, (unsigned long long)lo.y | ((unsigned long long)hi.y << 32));
}
__device__ __forceinline__ void swap32(double &a, double &b)
{
    const unsigned long long ua = __builtin_bit_cast(unsigned long long, a), ub = __builtin_bit_cast(unsigned long long, b);
    const osm_u2 lo = __builtin_amdgcn_permlane32_swap((unsigned)ua, (unsigned)ub, false, false);
    const osm_u2 hi = __builtin_amdgcn_permlane32_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
    a = __builtin_bit_cast(double, (unsigned long long)lo.x | ((unsigned long long)hi.x << 32));
    b = __builtin_bit_cast(double, (unsigned long long)lo.y | ((unsigned long long)hi.y << 32));
}
// bc[r] (r >= KR: the rows a pivot of DPP row KR still touches) = the sixteen values of c held by DPP row r, in every DPP row
template <int NROWS, int KR>
__device__ __forceinline__ void dpp_row_copies(double c, double (&bc)[4])
{
    bc[0] = bc[1] = bc[2] = bc[3] = c;
    if constexpr (NROWS == 2) {
        if constexpr (KR == 0) swap16(bc[0], bc[1]);           // (a pivot of the last row: every row it touches is its own)
    } else if constexpr (NROWS > 2) {
        if constexpr (KR < NROWS - 1) {
            double lo = c, hi = c;
            swap32(lo, hi);                                    // lo = rows (0, 1, 0, 1), hi = rows (2, 3, 2, 3)
            bc[0] = bc[1] = lo; bc[2] = bc[3] = hi;
            if constexpr (KR <= 1) swap16(bc[0], bc[1]);
            swap16(bc[2], bc[3]);
        }
    }
}

// Minimiser of the QP restricted to the face (sx, sy, sz of this lane's leg-step); returns this lane's component.
template <int NST, int UNR, bool RAW = false, typename WaveMem>
__device__ __forceinline__ double solve_face(const LaneCtx &L, const MpcParams &P, WaveMem &M, int sx, int sy, int sz,
                                             const double *cw, const double *cv /* this lane's step */)
{
    constexpr int NV = 15 * NST, NPS = 3 * NST;
    const int b0 = 3 * L.ls;
    OSM_STAMP(0)
    // ---- face generators ----
    bool live;
    double g[6], tt = 1.0;
    if (L.c == 2) {
        live = sz == SZ_FREE;
        const double fx = sx * P.mu, fy = sy * P.mu;
#pragma unroll
        for (int r = 0; r < 6; r++) g[r] = M.gen0[b0 + 2][r] + fx * M.gen0[b0][r] + fy * M.gen0[b0 + 1][r];
        tt = 1.0 + P.mu * P.mu * (double)(sx * sx + sy * sy);
    } else {
        live = sz != SZ_ZERO && (L.c == 0 ? sx : sy) == 0;
#pragma unroll
        for (int r = 0; r < 6; r++) g[r] = M.gen0[L.v][r];
    }
    live = live && !L.pad;
    if (!live) {
#pragma unroll
        for (int r = 0; r < 6; r++) g[r] = 0.0;
    }
    // the fixed part of the face (fz = fz_max faces)
    const double u0 = (sz == SZ_MAX && !L.pad) ? (L.c == 2 ? P.fzmax : (double)(L.c == 0 ? sx : sy) * P.mu * P.fzmax) : 0.0;
    const bool any_u0 = __ballot(u0 != 0.0) != 0ull;
    __builtin_amdgcn_wave_barrier();
    if (!L.pad) {
#pragma unroll
        for (int r = 0; r < 6; r++) M.gent[L.v][r] = g[r];
        M.vec[L.v] = u0;
    }
    __builtin_amdgcn_wave_barrier();

    OSM_STAMP(1)                                     // faces
    // ---- row of the reduced system ----
    double rhs = -(g[0] * cw[0] + g[1] * cw[1] + g[2] * cw[2] + g[3] * cv[0] + g[4] * cv[1] + g[5] * cv[2]);
    if (any_u0) rhs -= form_dot<NPS>(L, P, g, M.gen0, M.vec, M);
    double A[NV + 1];
    if constexpr (UNR == 0) form_rows_direct<NST>(L, P, g, M, A);
    else {
#pragma unroll
        for (int l = 0; l < 5; l++) {
            form_row_block<NPS, UNR>(L, P, g, l, M);
#pragma unroll
            for (int j = 0; j < NPS; j++) A[NPS * l + j] = M.rows[j][L.lane & (WaveMem::LW - 1)];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    A[NV] = live ? rhs : 0.0;
    const unsigned long long live_mask = __ballot(live);
    OSM_STAMP(2)                                     // rows

    // ---- forward elimination, one row per lane ----
    const double dsel = live ? P.rw * tt : 1.0;            // this lane's diagonal term (see form_row_block)
    double dinv = 1.0;
    // (an opaque copy of the lane index, as in mpc_quad.hip: the lane predicates of the pivots -- `lane > k`, `lane == k` -- are otherwise
    // hoisted out of the active-set loop as ~2 NV lane masks in SGPR pairs, spilled to VGPR lanes at the solver's entry and fetched
    // back with two v_readlane each; recomputed, a predicate is one v_cmp against an inline constant)
    int ll = L.lane;
    asm volatile("" : "+v"(ll));
    if constexpr (OS_MPC_DPP_ELIM(NST)) {
        // The system is symmetric and stays so under the elimination: entry j of pivot row k is ALSO element k of row j, i.e. register
        // A[k] of lane j -- ONE register across the lanes instead of one register per entry in lane k.  A 16-lane DPP row reads it inside
        // the FMA (v_fmac_f64_dpp row_newbcast: one instruction per entry where the broadcast out of lane k's registers took two
        // v_readlane_b32, a two-cycle s_nop and the FMA); rows of the system that sit in another DPP row get that row's sixteen values
        // of A[k] from one v_permlane16_swap (+ one v_permlane32_swap from 33 variables on) per pivot.  Only the right-hand side
        // (register NV, no transpose partner) still crosses by v_readlane.
        // The pivots' inverses are a chain of their own (~10 dependent instructions each: as long as a pivot's entries from the
        // second stance leg on).  With most variables on the face the elimination runs WITHOUT the dead-pivot skips, as straight-line
        // code: a dead variable's pivot is the 1.0 of its dsel over an all-zero row and column -- a no-op, bit for bit -- and pivot k
        // computes what its update will leave on lane k + 1's diagonal (the same FMA on the same operands) so that the next inverse
        // is on its way while the entries run.  With few variables on the face the skips are worth more than the overlap.
        if (__builtin_popcountll(live_mask) * 6 >= NV * OS_MPC_PIVOT_AHEAD(NST)) {
            double inv = rcp64(readlane_f64(A[0] + dsel, 0));
            static_for<0, NV>([&](auto kc) {
                constexpr int k = decltype(kc)::value, KR = k >> 4, E = NV - k - 1;
                if (ll == k) dinv = inv;
                const double nfa = -(A[k] * inv);
                const double nf = ll > k ? nfa : 0.0;            // (lanes from NV on hold all-zero rows: their nfa is a zero already)
                double dn = 1.0, rr = 0.0, ee = 0.0;             // next pivot (uniform), its inverse in the making
                if constexpr (k + 1 < NV) dn = readlane_f64(fma(A[k], nfa, A[k + 1]) + dsel, k + 1);
                double bc[4];
                dpp_row_copies<(NV + 15) / 16, KR>(A[k], bc);
                // rcp64()'s five instructions, one after every second entry (the entries are volatile assembly: hipcc schedules
                // nothing between them by itself)
                auto chain = [&](auto sc) {
                    constexpr int st = decltype(sc)::value;
                    if constexpr (k + 1 < NV) {
                        if constexpr (st == 0) asm volatile("s_nop 1\nv_rcp_f64 %0, %1" : "=v"(rr) : "s"(dn));
                        if constexpr (st == 1 || st == 3) asm volatile("s_nop 0\nv_fma_f64 %0, -%1, %2, 1.0" : "=&v"(ee) : "s"(dn), "v"(rr));
                        if constexpr (st == 2 || st == 4) asm volatile("v_fma_f64 %0, %1, %0, %0" : "+v"(rr) : "v"(ee));
                    }
                };
                static_for<k + 1, NV>([&](auto jc) {
                    constexpr int j = decltype(jc)::value, i = j - k;
                    fmac_bcast<j & 15, j == k + 1 || ((j & 15) == 0)>(A[j], bc[j >> 4], nf);
                    if constexpr (i % 2 == 0 && i / 2 <= 5) chain(std::integral_constant<int, i / 2 - 1>{});
                });
                static_for<(E / 2 < 5 ? E / 2 : 5), 5>(chain);
                A[NV] = fma(nf, readlane_f64(A[NV], k), A[NV]);
                // (column k below the pivot is spent: with two DPP rows its register keeps bc[0], whose first row -- the finished
                // rows' entries the back substitution reads -- is A[k]'s own: one copy less per pivot)
                if constexpr ((NV + 15) / 16 == 2 && KR == 0) A[k] = bc[0];
                inv = rr;
            });
        } else {
            static_for<0, NV>([&](auto kc) {
                constexpr int k = decltype(kc)::value, KR = k >> 4;
                if (!((live_mask >> k) & 1ull)) return;            // wave-uniform
                const double inv = rcp64(readlane_f64(A[k] + dsel, k));
                if (ll == k) dinv = inv;
                const double nf = ll > k ? -(A[k] * inv) : 0.0;
                double bc[4];
                dpp_row_copies<(NV + 15) / 16, KR>(A[k], bc);      // bc[r] = DPP row r's sixteen values of A[k] in every row that needs them
                static_for<k + 1, NV>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    fmac_bcast<j & 15, j == k + 1 || ((j & 15) == 0)>(A[j], bc[j >> 4], nf);
                });
                A[NV] = fma(nf, readlane_f64(A[NV], k), A[NV]);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
    } else {
        // the pivot row broadcast straight from lane k's registers (k is a compile-time lane index here: two v_readlane_b32 per entry
        // into an SGPR pair that the FMA reads) -- no LDS round trip and no barrier per pivot
#pragma unroll
        for (int k = 0; k < NV; k++) {
            if (!((live_mask >> k) & 1ull)) continue;          // wave-uniform
            const double inv = rcp64(readlane_f64(A[k] + dsel, k));
            if (ll == k) dinv = inv;
            const double f = (ll > k && ll < NV) ? A[k] * inv : 0.0;
#pragma unroll
            for (int j = k + 1; j <= NV; j++) {
                A[j] = fma(-f, readlane_f64(A[j], k), A[j]);
                if (((j - k) & 15) == 0) __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    OSM_STAMP(3)                                     // elimination
    // ---- back substitution ----
    double r = A[NV];
#pragma unroll
    for (int k = NV - 1; k >= 0; k--) {
        if (!((live_mask >> k) & 1ull)) continue;
        const double wk = readlane_f64(r * dinv, k);
        if (ll < k) r = fma(-A[k], wk, r);
        __builtin_amdgcn_sched_barrier(0);
    }
    const double sol = live ? r * dinv : 0.0;              // (lane k's r is final when pivot k is reached and no later pivot touches it)
    OSM_STAMP(4)                                     // back substitution
    if constexpr (RAW) return L.pad ? 0.0 : sol;     // (face coordinates: the caller maps them to forces in its own exchange)
    // ---- face coordinates -> forces ----
    __builtin_amdgcn_wave_barrier();
    if (!L.pad) M.vec[L.v] = sol;
    __builtin_amdgcn_wave_barrier();
    const double fz = sz == SZ_MAX ? P.fzmax : (sz == SZ_ZERO ? 0.0 : M.vec[b0 + 2]);
    double u;
    if (L.c == 2) u = fz;
    else {
        const int s = L.c == 0 ? sx : sy;
        u = s != 0 ? (double)s * P.mu * fz : (sz == SZ_ZERO ? 0.0 : sol);
    }
    OSM_STAMP(5)                                     // forces
    return L.pad ? 0.0 : u;
}

// Per-lane solver state that survives a call: the lane's variable and the face of its leg-step (warm start of the next step)
struct QpLane {
    double u;
    int face;                 // (sx + 1) | (sy + 1) << 2 | sz << 4
};

// One QP on the calling wavefront.  x, ref, p: the problem data (wave-uniform); warm: start from io (same contact word as the
// call that produced it).  Returns in `val` (lanes 0..59) the optimal control in the reference's variable order (12 per horizon
// step, leg-major, swing legs zero), in io the state for the next warm start.
template <int NST, int UNR = 3>
__device__ __forceinline__ void mpc_solve_wave(const MpcParams &P, uint32_t cbits, const int (&legs)[4], const double (&x)[12],
                                               const double (&ref)[12], const double (&p)[12], int max_iter, bool warm,
                                               WaveMemT<15 * NST> &M, QpLane &io, float &val, int &iters_out, bool &converged_out)
{
    constexpr int NV = 15 * NST, NPS = 3 * NST, NLS = 5 * NST;
    LaneCtx L;
    L.lane = threadIdx.x & 63;
    L.pad = L.lane >= NV;
    L.v = L.pad ? NV - 1 : L.lane;
    L.i = L.v / NPS; L.c = L.v % 3; L.ls = L.v / 3;
    {
        const int rank = (L.v % NPS) / 3;
        // (one select at a time, opaque in between: as a chain hipcc turns it into legs[rank] on a copy of legs in scratch)
        int lg = legs[0];
        if (rank == 1) lg = legs[1];
        asm volatile("" : "+v"(lg));
        if (rank == 2) lg = legs[2];
        asm volatile("" : "+v"(lg));
        if (rank >= 3) lg = legs[3];
        L.leg = lg;
    }

    OSM_STAMP(12)                                    // call + argument block
    // ---- problem data (wave-uniform) ----
    const uint32_t cleg = (cbits >> (8 * L.leg)) & 0xffu;
    const bool stance = cleg == 1u;      // any other non-zero value: unconstrained (force_controller.py:114-131)

    double R0[9], R1[9];
    rotation64(x[0], x[1], x[2], R0);
    rotation64(ref[0], ref[1], ref[2], R1);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int s = 0; s < 3; s++)
            L.Cth[3 * r + s] = readlane_f64(R1[3 * r] * P.w[0] * R1[3 * s] + R1[3 * r + 1] * P.w[1] * R1[3 * s + 1] + R1[3 * r + 2] * P.w[2] * R1[3 * s + 2], 0);

    OSM_STAMP(13)                                    // rotations, C_theta
    // generators of this lane's variable: a = I_hat^-1 (R p_leg x e_c), b = e_c / m, with R of the lane's horizon step
    {
        const double *R = L.i == 0 ? R0 : R1;
        double Rm[9];
#pragma unroll
        for (int r = 0; r < 9; r++) Rm[r] = L.i == 0 ? R0[r] : R1[r];
        (void)R;
        double px = p[0], py = p[1], pz = p[2];
#pragma unroll
        for (int l = 1; l < 4; l++) {
            if (L.leg == l) { px = p[3 * l]; py = p[3 * l + 1]; pz = p[3 * l + 2]; }
            // (opaque between the selects: hipcc otherwise turns the chain into p[3 * leg + c] on a 96-byte copy of p in SCRATCH --
            // the 112 B per lane every solver instance carried, code-object metadata round 5)
            asm volatile("" : "+v"(px), "+v"(py), "+v"(pz));
        }
        const double wx = Rm[0] * px + Rm[1] * py + Rm[2] * pz, wy = Rm[3] * px + Rm[4] * py + Rm[5] * pz,
                     wz = Rm[6] * px + Rm[7] * py + Rm[8] * pz;
        double cr[3];                                  // pw x e_c
        if (L.c == 0) { cr[0] = 0.0; cr[1] = wz; cr[2] = -wy; }
        else if (L.c == 1) { cr[0] = -wz; cr[1] = 0.0; cr[2] = wx; }
        else { cr[0] = wy; cr[1] = -wx; cr[2] = 0.0; }
        // I_hat^-1 = R diag(1/I) R^T
        double tb[3];
#pragma unroll
        for (int k = 0; k < 3; k++) tb[k] = (Rm[k] * cr[0] + Rm[3 + k] * cr[1] + Rm[6 + k] * cr[2]) * P.inv_inertia[k];
        double g0[6];
#pragma unroll
        for (int r = 0; r < 3; r++) g0[r] = Rm[3 * r] * tb[0] + Rm[3 * r + 1] * tb[1] + Rm[3 * r + 2] * tb[2];
#pragma unroll
        for (int r = 0; r < 3; r++) g0[3 + r] = (r == L.c) ? P.inv_mass : 0.0;
        if (!L.pad) {
#pragma unroll
            for (int r = 0; r < 6; r++) M.gen0[L.v][r] = g0[r];
        }
    }

    OSM_STAMP(14)                                    // generators
    // linear term: q_v = a_v . cw_i + b_v . cv_i from the zero-input trajectory (see header / docs/DESIGN_history_r01-r04.md section 4.5).  The errors of the
    // zero-input trajectory at horizon step k are polynomials in k (attitude and velocity linear, position quadratic through
    // gravity), so the sums over the steps k > i that variable v still influences are closed forms in the lane's step i:
    // with Mm = 4 - i and m = k - 1 - i = 0..Mm,
    //   n = Mm + 1,  S1 = sum m,  S2 = sum m (m + i),  T1 = sum (m + i + 1),  T2 = sum m (m + i + 1),  T3 = sum m (m + i + 1)(m + i)
    // (~60 float64 instructions instead of the ~375 of the five-step loop; the rounding differs at the 1e-16 level).
    double cw[3], cv[3];
    {
        const double dt = P.dt, dt2 = dt * dt;
        const double di = (double)L.i, Mm = 4.0 - di, n = Mm + 1.0;
        const double s1 = 0.5 * Mm * n;                                   // sum m
        const double s2m = Mm * n * (2.0 * Mm + 1.0) * (1.0 / 6.0);        // sum m^2
        const double s3m = s1 * s1;                                      // sum m^3
        const double S1 = s1, S2 = s2m + di * s1;
        const double T1 = s1 + n * (di + 1.0);
        const double T2 = s2m + (di + 1.0) * s1;
        const double T3 = s3m + (2.0 * di + 1.0) * s2m + di * (di + 1.0) * s1;
        double e0[3], dd[3];               // eth_k = e0 + (k - 1) dd,  k - 1 = m + i
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const double r0w = R0[r] * x[6] + R0[3 + r] * x[7] + R0[6 + r] * x[8];
            const double r1w = R1[r] * x[6] + R1[3 + r] * x[7] + R1[6 + r] * x[8];
            e0[r] = x[r] + dt * r0w - ref[r];
            dd[r] = dt * r1w;
        }
        double wt[3];                      // sum_k lev_k w_theta eth_k = w_theta (S1 e0 + S2 dd)
#pragma unroll
        for (int r = 0; r < 3; r++) wt[r] = P.w[r] * (S1 * e0[r] + S2 * dd[r]);
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const double ew = x[6 + r] - ref[6 + r];
            cw[r] = n * dt * P.w[6 + r] * ew + dt2 * (R1[3 * r] * wt[0] + R1[3 * r + 1] * wt[1] + R1[3 * r + 2] * wt[2]);
            double sev = n * (x[9 + r] - ref[9 + r]);                                  // sum_k ev_k
            double ser = S1 * (x[3 + r] - ref[3 + r]) + dt * x[9 + r] * T2;             // sum_k lev_k er_k
            if (r == 2) { sev += dt * P.gz * T1; ser += 0.5 * dt2 * P.gz * T3; }
            cv[r] = dt * P.w[9 + r] * sev + dt2 * P.w[3 + r] * ser;
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (L.lane < 25) {
        double al, be;
        alpha_beta(L.lane / 5, L.lane % 5, P.dt, al, be);
        M.ab[L.lane][0] = al; M.ab[L.lane][1] = be;
    }
    __builtin_amdgcn_wave_barrier();

    // ---- cold start: stance (and unconstrained) legs free, swing legs zero; then the primal active-set iteration ----
    int sx = 0, sy = 0, sz = SZ_FREE;
    double u = 0.0;
    int iters = 0;
    bool first = true, done = false, converged = false;
    if (warm) {
        u = io.u;
        sx = (io.face & 3) - 1; sy = ((io.face >> 2) & 3) - 1; sz = (io.face >> 4) & 3;
        first = false;
    }
    constexpr double EPS = 1e-11, TOL = 1e-12;
    OSM_STAMP(7)                                     // set-up
    // One active-set iteration = solve + ONE exchange of the leg-step's solution and point to its three lanes + the row's minimum ratio
    // (+ four exchanges when the subspace minimiser is reached); every leg-step quantity is computed by all three of its lanes without
    // branches -- mpc_quad.hip's iterate_row on one lane per variable (round 6: the round-5 form handed every stage's result on
    // through LDS, with six candidate ratios behind six branches: 4.4 k cycles of an iteration at the reference's shape).
    while (!done && iters < max_iter) {
        iters++;
        const double sol = solve_face<NST, UNR, true>(L, P, M, sx, sy, sz, cw, cv);
#ifdef OS_MPC_DBG
        if (L.lane == 0) printf("wave it %d\n", iters);
#endif
        __builtin_amdgcn_wave_barrier();
        if (!L.pad) { M.vec[L.v] = sol; M.rowbuf[L.v] = u; }
        __builtin_amdgcn_wave_barrier();
        double s3[3], u3[3], c3[3];
#pragma unroll
        for (int q = 0; q < 3; q++) { s3[q] = M.vec[3 * L.ls + q]; u3[q] = M.rowbuf[3 * L.ls + q]; }
        {
            // face coordinates -> forces: the candidate point of the face
            const double fz = sz == SZ_MAX ? P.fzmax : (sz == SZ_ZERO ? 0.0 : s3[2]);
            c3[0] = sx != 0 ? (double)sx * P.mu * fz : (sz == SZ_ZERO ? 0.0 : s3[0]);
            c3[1] = sy != 0 ? (double)sy * P.mu * fz : (sz == SZ_ZERO ? 0.0 : s3[1]);
            c3[2] = fz;
        }
        if (first) {
            // clamp the stance-free minimiser into the pyramids; the faces come from the clamps.  Nothing clamped: optimal.
            first = false;
            double fx = c3[0], fy = c3[1], fz = c3[2];
            int nx = 0, ny = 0, nz = SZ_FREE;
            bool cl = false;
            if (fz <= 0.0) { nz = SZ_ZERO; fx = fy = fz = 0.0; cl = true; }
            else {
                if (fz >= P.fzmax) { nz = SZ_MAX; fz = P.fzmax; cl = true; }
                const double lim = P.mu * fz;
                if (fx >= lim) { nx = 1; fx = lim; cl = true; } else if (fx <= -lim) { nx = -1; fx = -lim; cl = true; }
                if (fy >= lim) { ny = 1; fy = lim; cl = true; } else if (fy <= -lim) { ny = -1; fy = -lim; cl = true; }
            }
            const bool act = stance && !L.pad;
            sx = act ? nx : sx; sy = act ? ny : sy; sz = act ? nz : sz;
            const double own = L.pad ? 0.0 : (L.c == 0 ? c3[0] : (L.c == 1 ? c3[1] : c3[2]));
            u = act ? (L.c == 0 ? fx : (L.c == 1 ? fy : fz)) : own;
            if (__ballot(act && cl) == 0ull) { done = true; converged = true; }
            OSM_STAMP(6)
            continue;
        }
        // ---- ratio test of the lane's leg-step (all six candidates, no branches) ----
        double d3[3];
#pragma unroll
        for (int q = 0; q < 3; q++) d3[q] = c3[q] - u3[q];
        {
            const double fx = u3[0], fy = u3[1], fz = u3[2], dx = d3[0], dy = d3[1], dz = d3[2];
            const bool on = stance && sz != SZ_ZERO;
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
            if (L.c == 2 && !L.pad) { M.al[L.ls] = best; M.code[L.ls] = code; }
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
        {
#pragma unroll
            for (int q = 0; q < 3; q++) u3[q] += amin * d3[q];
            int nx = sx, ny = sy, nz = sz;
            if (cmin == 1) { nx = 0; ny = 0; nz = SZ_ZERO; }
            else if (cmin == 2) nz = SZ_MAX;
            else if (cmin == 3) nx = 1;
            else if (cmin == 4) nx = -1;
            else if (cmin == 5) ny = 1;
            else if (cmin == 6) ny = -1;
            const bool hit = lsmin >= 0 && L.ls == lsmin && !L.pad;
            sx = hit ? nx : sx; sy = hit ? ny : sy; sz = hit ? nz : sz;
            const double fz = sz == SZ_ZERO ? 0.0 : (sz == SZ_MAX ? P.fzmax : u3[2]);
            const double own = L.c == 0 ? u3[0] : (L.c == 1 ? u3[1] : u3[2]);
            const int sgn = L.c == 0 ? sx : sy;
            const double un = L.c == 2 ? fz : (sz == SZ_ZERO ? 0.0 : (sgn != 0 ? (double)sgn * P.mu * fz : own));
            u = L.pad ? u : (stance ? un : own);
        }
#ifdef OS_MPC_DBG
        if (L.lane == 0) printf("wave it %d ratio: amin %.6e lsmin %d cmin %d\n", iters, amin, lsmin, cmin);
#endif
        if (lsmin >= 0) { OSM_STAMP(6) continue; }

        // ---- subspace minimiser reached: multiplier signs ----
        __builtin_amdgcn_wave_barrier();
        if (!L.pad) M.vec[L.v] = u;
        __builtin_amdgcn_wave_barrier();
        double g0[6];
#pragma unroll
        for (int r = 0; r < 6; r++) g0[r] = M.gen0[L.v][r];
        const double q0 = g0[0] * cw[0] + g0[1] * cw[1] + g0[2] * cw[2] + g0[3] * cv[0] + g0[4] * cv[1] + g0[5] * cv[2];
        const double grad = 2.0 * (form_dot<NPS, UNR == 0>(L, P, g0, M.gen0, M.vec, M) + P.rw * u + q0);
        __builtin_amdgcn_wave_barrier();
        if (!L.pad) M.rowbuf[L.v] = grad;
        __builtin_amdgcn_wave_barrier();
        {
            const double gx = M.rowbuf[3 * L.ls], gy = M.rowbuf[3 * L.ls + 1], gz = M.rowbuf[3 * L.ls + 2];
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
            if (!stance) { best = TOL; code = 0; }
            if (L.c == 2 && !L.pad) { M.al[L.ls] = best; M.code[L.ls] = code; }
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
        if (L.lane == 0) printf("wave it %d mult: rmax %.6e lsr %d cr %d\n", iters, rmax, lsr, cr);
#endif
        if (lsr < 0) { done = true; converged = true; OSM_STAMP(6) break; }
        if (L.ls == lsr && !L.pad) {
            if (cr == 1) sx = 0;
            else if (cr == 2) sy = 0;
            else if (cr == 3) sz = SZ_FREE;
            else { sx = (cr & 1) ? 1 : -1; sy = (cr & 2) ? 1 : -1; sz = SZ_FREE; }
        }
        OSM_STAMP(6)                                 // ratio test / multipliers
    }

    // ---- outputs in the reference's variable order (12 per horizon step, leg-major); swing legs are zero ----
    __builtin_amdgcn_wave_barrier();
    if (!L.pad) M.vec[L.v] = u;
    __builtin_amdgcn_wave_barrier();
    val = 0.f;
    if (L.lane < 60) {
        const int oi = L.lane / 12, oleg = (L.lane % 12) / 3, oc = L.lane % 3;
        int rank = -1;
#pragma unroll
        for (int r = 0; r < NST; r++)
            if ((r == 0 ? legs[0] : (r == 1 ? legs[1] : (r == 2 ? legs[2] : legs[3]))) == oleg) rank = r;
        val = rank < 0 ? 0.f : (float)M.vec[oi * NPS + 3 * rank + oc];
    }
    __builtin_amdgcn_wave_barrier();
    io.u = L.pad ? 0.0 : u;
    io.face = (sx + 1) | ((sy + 1) << 2) | (sz << 4);
    iters_out = iters;
    converged_out = converged;
}

// The same solver behind a real call: its own register allocation instead of the union of four inlined instances and the
// filter state around them (the persistent kernel at small batch: 1,100 SGPR and 50-370 VGPR spills inlined; B = 8:
// 109 -> 92 us per step as a call.  At large batch the call's stack costs occupancy: 7.9e6 -> 5.8e6 steps/s, so the
// two-waves-per-SIMD instantiation keeps the inlined form).
// Round 5: nothing crosses the call through the stack.  References to the caller's kernel argument (`a.prm`) and to its local
// arrays (x, ref, p, legs, the outputs) made hipcc copy the whole 1.3 KB argument block and the arrays into scratch: 1,748 B
// per lane.  The problem data now sits in an LDS block the caller fills (QpCall; the callee copies it into registers), the
// per-lane solver state goes in and out by value (QpRet comes back in six VGPRs).
struct QpCall {
    MpcParams prm;
    double x[12], ref[12], p[12];
    int legs[4];
};
struct QpRet { double u; int face; float val; int iters; int conv; };
typedef const __attribute__((address_space(3))) QpCall *QpCallLds;
template <int NST>
static __device__ __attribute__((noinline)) QpRet mpc_solve_wave_call(QpCallLds qc, uint32_t cbits, int max_iter, bool warm,
                                                               WaveMemT<15 * NST> &M, double u_in, int face_in)
{
    MpcParams P;
    double x[12], ref[12], p[12];
    int legs[4];
#pragma unroll
    for (int j = 0; j < 12; j++) { P.w[j] = qc->prm.w[j]; x[j] = qc->x[j]; ref[j] = qc->ref[j]; p[j] = qc->p[j]; }
    P.rw = qc->prm.rw; P.mu = qc->prm.mu; P.fzmax = qc->prm.fzmax; P.dt = qc->prm.dt; P.inv_mass = qc->prm.inv_mass; P.gz = qc->prm.gz;
#pragma unroll
    for (int j = 0; j < 3; j++) P.inv_inertia[j] = qc->prm.inv_inertia[j];
#pragma unroll
    for (int j = 0; j < 4; j++) legs[j] = qc->legs[j];
    QpLane io = {u_in, face_in};
    QpRet r;
    bool conv;
    mpc_solve_wave<NST, 0>(P, cbits, legs, x, ref, p, max_iter, warm, M, io, r.val, r.iters, conv);
    r.u = io.u; r.face = io.face; r.conv = conv ? 1 : 0;
    return r;
}

// NST = number of legs that carry force variables (contact byte != 0).  Swing legs are eliminated up front: a trot
// problem has 30 variables, not 60 (elimination work ~ n^3), and each instance gets the register budget its row needs.
// Every instance is launched over the whole batch; a wavefront whose problem has a different leg count exits at once.
template <int NST>
#ifndef OS_MPC_SOLVE_OCC
// waves per SIMD the register budget is sized for: 4 at one stance leg (109 registers), 2 at two (the straight-line elimination beside
// the skipping one: eight registers spilled at a budget of 168; since round 6 this instance serves batches below 64 and OS_MPC_QUAD=0
// only), 3 at three (152), 2 at four (185)
#define OS_MPC_SOLVE_OCC (NST <= 1 ? 4 : NST == 3 ? 3 : 2)
#endif
__global__ __launch_bounds__(64, OS_MPC_SOLVE_OCC) void mpc_solve_kernel(const MpcArgs a, int handover)
{
    typedef WaveMemT<15 * NST> WaveMem;
    __shared__ WaveMem M;
    const size_t B = (size_t)a.B;
    // handover = 1 (second pass behind mpc_quad.hip's rows): this wavefront continues problem todo[blockIdx.x] from the state the row
    // left in the warm arrays; the iteration counts add up
    int b = blockIdx.x;
    if (handover) {
        if (NST > 2 || b >= a.todo_count[NST <= 2 ? NST - 1 : 0]) return;
        b = a.todo[(size_t)(NST <= 2 ? NST - 1 : 0) * B + b];
    }
    const uint32_t cbits = a.contact[b];
    int legs[4] = {0, 0, 0, 0}, nst = 0;
#pragma unroll
    for (int l = 0; l < 4; l++) {
        if (((cbits >> (8 * l)) & 0xffu) != 0u) {
            if (nst == 0) legs[0] = l; else if (nst == 1) legs[1] = l; else if (nst == 2) legs[2] = l; else legs[3] = l;
            nst++;
        }
    }
    if (nst == 0 && NST == 1) {          // no leg on the ground: all forces zero (force_controller.py:114-123)
        const int t = threadIdx.x;
        if (t < 12) a.f_out[(size_t)t * B + b] = 0.f;
        if (a.u_out && t < 60) a.u_out[(size_t)t * B + b] = 0.f;
        if (t == 0 && a.iters) a.iters[b] = 0;
        if (t == 0 && a.warm_contact) a.warm_contact[b] = cbits;
        return;
    }
    if (nst != NST) return;
    const int lane = threadIdx.x;
    double x[12], ref[12], p[12];
#pragma unroll
    for (int j = 0; j < 12; j++) {
        x[j] = (double)a.x[(size_t)j * B + b];
        ref[j] = (double)a.ref[(size_t)j * B + b];
        p[j] = (double)a.p[(size_t)j * B + b];
    }
    QpLane io = {0.0, 0};
    const bool warm = handover || (a.warm_u && !a.cold_in && a.warm_contact[b] != 0xffffffffu && contact_ranks(a.warm_contact[b]) == contact_ranks(cbits));
    if (warm) { io.u = a.warm_u[(size_t)b * 64 + lane]; io.face = a.warm_state[(size_t)b * 64 + lane]; }
    float val;
    int iters;
    bool converged;
    mpc_solve_wave<NST>(a.prm, cbits, legs, x, ref, p, handover ? a.max_iter - a.cap : a.max_iter, warm, M, io, val, iters, converged);
    if (lane < 12) a.f_out[(size_t)lane * B + b] = val;
    if (a.u_out && lane < 60) a.u_out[(size_t)lane * B + b] = val;
    if (a.warm_u) {
        a.warm_u[(size_t)b * 64 + lane] = io.u;
        a.warm_state[(size_t)b * 64 + lane] = (uint8_t)io.face;
        if (lane == 0) a.warm_contact[b] = cbits;
    }
    if (lane == 0) {
        if (a.iters) a.iters[b] = handover ? a.iters[b] + iters : iters;
        if (!converged) a.status[b] |= 4;
    }
}

// =====================================================================================================================
// estimate_state_mpc as ONE persistent kernel: a wavefront owns a trajectory for all T steps -- QP (mpc_solve_wave, warm
// start in registers) -> get_odom / set_measurements / next_state (float32, computed redundantly on every lane: a few
// hundred instructions) -> predict_mpc covariance and the batch update with the 12 x 12 covariance in LDS, FLOAT64, spread
// over the 64 lanes.  The per-step launch sequence (two to five kernels, a T = 1 filter launch whose one-trajectory-per-lane
// float64 update is a ~100 k-cycle dependent chain, a status kernel) cost 215-280 us per step at the reference's own shape
// (B = 8, T ~ 4000); here a step is the QP plus ~10 k cycles of filter.  P stays in float64 between steps, as in the
// reference (the launch sequence rounded it to float32 at every step boundary).
// =====================================================================================================================
struct KfWave {
    double Q[144];               // float64 copies for the row layout: a lane reads ITS row of Q ...
    double R[12][10];            // ... and the row of R of the measurement its state row owns (rows 10, 11: zeros, lanes without one)
    QpCall qc;                   // the QP's problem data for mpc_solve_wave_call
};

struct MpcRunArgs {
    osk::KfRunArgs kf;             // streams, state, outputs, noise (k)
    float *f_out;                  // [T][12][B]
    int32_t *iters;                // [T][B] or null
    int max_iter, cold;
    MpcParams prm;
};

union QpMem {
    WaveMemT<15> m1;
    WaveMemT<30> m2;
    WaveMemT<45> m3;
    WaveMemT<60> m4;
    __device__ QpMem() {}
};

// OCC = wavefronts per SIMD the register budget is sized for: 1 (512 registers, nothing spilled in the elimination) for
// the few-long-trajectories shape, 2 when the batch fills the chip twice over (measured B = 4096: 8.5e6 against 7.9e6 steps/s)
template <int OCC>
__global__ __launch_bounds__(64, OCC) void kf_mpc_persistent_kernel(const MpcRunArgs a)
{
    __shared__ QpMem QM;
    __shared__ KfWave W;
    using namespace osk;
    namespace rw = osk::rows64;
    const int b = blockIdx.x, lane = threadIdx.x;
    // Filter state on the 16-lanes-per-trajectory layout of kf_dense_rows.hpp, the SAME trajectory in all four DPP rows of the
    // wavefront: lane (lane & 15) = r holds row r of P in float64 (24 registers) and x[r]; rows 12-15 of every group shadow row 11.
    const int r = lane & 15, rr = r < 12 ? r : 11, am = rw::row_measurement(r);
    const size_t B = (size_t)a.kf.B;
    const uint32_t voff = (uint32_t)b * 4u, rowB = (uint32_t)a.kf.B * 4u;
    float xr = a.kf.x[(size_t)rr * B + b];
    double P[NS];
#pragma unroll
    for (int j = 0; j < NS; j++) P[j] = (double)a.kf.P[(size_t)(rr * NS + j) * B + b];
    for (int el = lane; el < 144; el += 64) W.Q[el] = (double)a.kf.k.Q[el];
    for (int el = lane; el < 120; el += 64)             // R as given (kf_dense_rows.hpp update_batch_row)
        (&W.R[0][0])[el] = el < 100 ? (double)a.kf.k.R[el] : 0.0;
    if (lane == 0) {
        // MpcParams, field by field: a reference to a.prm would make hipcc copy the whole argument block to scratch
#pragma unroll
        for (int j = 0; j < 12; j++) W.qc.prm.w[j] = a.prm.w[j];
        W.qc.prm.rw = a.prm.rw; W.qc.prm.mu = a.prm.mu; W.qc.prm.fzmax = a.prm.fzmax; W.qc.prm.dt = a.prm.dt;
        W.qc.prm.inv_mass = a.prm.inv_mass; W.qc.prm.gz = a.prm.gz;
#pragma unroll
        for (int j = 0; j < 3; j++) W.qc.prm.inv_inertia[j] = a.prm.inv_inertia[j];
    }
    __builtin_amdgcn_wave_barrier();
    const double *qrow = W.Q + rr * NS, *rrow_l = &W.R[am >= 0 ? am : NM][0];
    double one = 1.0;
    asm volatile("" : "+v"(one));
    const double ed = expm1((double)a.kf.k.dt);
    QpLane qio = {0.0, 0};
    uint32_t prev_c = 0xffffffffu;
    int status = 0;
    // inputs of a step: every lane reads the same 55 dwords (broadcast); the next step's are requested before the QP and land
    // underneath it (un-prefetched, their HBM round trips were 8-15 k cycles of every step)
    StepIn in_n;
    float bref_n[12];
    auto fetch = [&](int t) {
        load_step(a.kf, t, voff, rowB, in_n);
        rsrc_t rb = make_rsrc(a.kf.body_ref + (size_t)t * 12 * B, 12 * rowB);
#pragma unroll
        for (int i = 0; i < 12; i++) bref_n[i] = buf_load_nt(rb, voff, i * rowB);
    };
    fetch(0);
#ifdef OSM_TS
    if (threadIdx.x < 16) osm_ts_sum[threadIdx.x] = 0;
#endif
    for (int t = 0; t < a.kf.T; t++) {
        OSM_STAMP(9)                                 // filter + stores of the previous step
        StepIn in = in_n;
        float bref[12];
#pragma unroll
        for (int i = 0; i < 12; i++) bref[i] = bref_n[i];
        fetch(t + 1 < a.kf.T ? t + 1 : t);
        float x[NS];
        rw::gather_state(xr, x);                       // the replicated prior state
        // ---- forces from the state BEFORE this step's predict (kalman_filter.py:141-152) ----
        const uint32_t cbits = __builtin_amdgcn_readfirstlane(in.contact);
        int legs[4] = {0, 0, 0, 0}, nst = 0;
#pragma unroll
        for (int l = 0; l < 4; l++) {
            if (((cbits >> (8 * l)) & 0xffu) != 0u) {
                if (nst == 0) legs[0] = l; else if (nst == 1) legs[1] = l; else if (nst == 2) legs[2] = l; else legs[3] = l;
                nst++;
            }
        }
        float fval = 0.f;
        int iters = 0;
        if (nst > 0) {
            const bool warm = !a.cold && prev_c != 0xffffffffu && contact_ranks(cbits) == contact_ranks(prev_c);
            bool conv = true;
            if constexpr (OCC == 1) {
                if (lane == 0) {
#pragma unroll
                    for (int j = 0; j < 12; j++) { W.qc.x[j] = (double)x[j]; W.qc.ref[j] = (double)bref[j]; W.qc.p[j] = (double)in.p[j]; }
#pragma unroll
                    for (int j = 0; j < 4; j++) W.qc.legs[j] = legs[j];
                }
                __builtin_amdgcn_wave_barrier();
                const QpCallLds qc = (QpCallLds)&W.qc;
                QpRet q;
                if (nst == 1) q = mpc_solve_wave_call<1>(qc, cbits, a.max_iter, warm, QM.m1, qio.u, qio.face);
                else if (nst == 2) q = mpc_solve_wave_call<2>(qc, cbits, a.max_iter, warm, QM.m2, qio.u, qio.face);
                else if (nst == 3) q = mpc_solve_wave_call<3>(qc, cbits, a.max_iter, warm, QM.m3, qio.u, qio.face);
                else q = mpc_solve_wave_call<4>(qc, cbits, a.max_iter, warm, QM.m4, qio.u, qio.face);
                qio.u = q.u; qio.face = q.face; fval = q.val; iters = q.iters; conv = q.conv != 0;
            } else {
                double xd[12], rd[12], pd[12];
#pragma unroll
                for (int j = 0; j < 12; j++) { xd[j] = (double)x[j]; rd[j] = (double)bref[j]; pd[j] = (double)in.p[j]; }
                if (nst == 1) mpc_solve_wave<1, 1>(a.prm, cbits, legs, xd, rd, pd, a.max_iter, warm, QM.m1, qio, fval, iters, conv);
                else if (nst == 2) mpc_solve_wave<2, 1>(a.prm, cbits, legs, xd, rd, pd, a.max_iter, warm, QM.m2, qio, fval, iters, conv);
                else if (nst == 3) mpc_solve_wave<3, 1>(a.prm, cbits, legs, xd, rd, pd, a.max_iter, warm, QM.m3, qio, fval, iters, conv);
                else mpc_solve_wave<4, 1>(a.prm, cbits, legs, xd, rd, pd, a.max_iter, warm, QM.m4, qio, fval, iters, conv);
            }
            if (!conv) status |= 4;
        }
        OSM_STAMP(8)                                 // outputs of the QP (behind its last iteration)
        prev_c = cbits;
#ifdef OSM_TS
        if (blockIdx.x == 0 && threadIdx.x == 0) { osm_ts_sum[10] += iters; osm_ts_sum[11] += 1; }
#endif
        if (lane < 12) a.f_out[((size_t)t * 12 + lane) * B + b] = fval;
        if (a.iters && lane == 0) a.iters[(size_t)t * B + b] = iters;
#pragma unroll
        for (int j = 0; j < 12; j++) in.f[j] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fval), j));
        // ---- get_odom + set_measurements + predict_mpc + next_state + update (kalman_filter.py:176-182), float64 covariance on
        // the row layout: every cross-lane term is a broadcast fused into a v_fmac_f64_dpp (kf_dense_rows.hpp) ----
        float z[NM], pw[12];
        status |= rw::front_row(x, xr, P, in, bref, a.kf.k, ed, qrow, one, r, z, pw);
        float xn = x[0];
#pragma unroll
        for (int i = 1; i < NS; i++) xn = (rr == i) ? x[i] : xn;
        double xd = (double)xn, K[NM], rrow[NM];
#pragma unroll
        for (int q = 0; q < NM; q++) rrow[q] = rrow_l[q];
        float kg = 0.f;
        status |= rw::update_batch_row<true>(xd, P, z, rrow, K, am, one, &kg);
        xr = (float)xd;
        if (!(xr * 0.f == 0.f)) status |= 2;
        if (lane < 12) a.kf.x_out[((size_t)t * 12 + lane) * B + b] = xr;
        if (a.kf.p_rot_out && lane == 0) {
#pragma unroll
            for (int i = 0; i < 12; i++) a.kf.p_rot_out[((size_t)t * 12 + i) * B + b] = pw[i];
        }
        if (a.kf.ptrace_out) {                         // (wave-uniform branches: every lane takes part in the DPP sums)
            const float tr = rw::ptrace_rows(P, one);
            if (lane == 0) a.kf.ptrace_out[(size_t)t * B + b] = tr;
        }
        if (a.kf.kgain_out && lane == 0) a.kf.kgain_out[(size_t)t * B + b] = kg;
    }
#ifdef OSM_TS
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long n = osm_ts_sum[11] ? osm_ts_sum[11] : 1;
        printf("persistent kernel, cycles per step (%llu steps, %.2f iterations per step): faces %llu | rows %llu | elimination %llu | back-substitution %llu | forces %llu | "
               "ratio/multipliers %llu | QP set-up: call + arguments %llu, rotations %llu, generators %llu, linear term + tables %llu | QP outputs %llu | filter step %llu\n", n, (double)osm_ts_sum[10] / (double)n, osm_ts_sum[1] / n, osm_ts_sum[2] / n,
               osm_ts_sum[3] / n, osm_ts_sum[4] / n, osm_ts_sum[5] / n, osm_ts_sum[6] / n, osm_ts_sum[12] / n, osm_ts_sum[13] / n, osm_ts_sum[14] / n, osm_ts_sum[7] / n, osm_ts_sum[8] / n, osm_ts_sum[9] / n);
    }
#endif
    // the status word is OR-reduced over the wavefront (bit 1 is per state component)
    for (int m = 1; m < 64; m <<= 1) status |= __shfl_xor(status, m, 64);
    if (lane < 12) {
        a.kf.x[(size_t)lane * B + b] = xr;
#pragma unroll
        for (int j = 0; j < NS; j++) a.kf.P[(size_t)(lane * NS + j) * B + b] = (float)P[j];
    }
    if (lane == 0) a.kf.status[b] = status;
}

// which leg counts occur at each step of a [T][B] contact stream: flags[t] bit n set <=> some trajectory has n legs on the ground
__global__ void nst_presence_kernel(int B, int T, const uint32_t *contact, uint32_t *flags)
{
    const int t = blockIdx.y;
    uint32_t m = 0;
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
        const uint32_t c = contact[(size_t)t * B + b];
        const int n = ((c & 0xffu) != 0) + (((c >> 8) & 0xffu) != 0) + (((c >> 16) & 0xffu) != 0) + (((c >> 24) & 0xffu) != 0);
        m |= 1u << n;
    }
    if (m) atomicOr(&flags[t], m);
}

__global__ void or_status_kernel(int B, int32_t *dst, const int32_t *src)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) dst[b] |= src[b];
}

static void fill_args(os_ctx *ctx, MpcArgs &a)
{
    for (int i = 0; i < 12; i++) a.prm.w[i] = ctx->mpc_w[i];
    a.prm.rw = ctx->mpc_rw; a.prm.mu = ctx->mpc_mu; a.prm.fzmax = ctx->mpc_fzmax;
    a.prm.dt = (double)ctx->k.dt; a.prm.inv_mass = 1.0 / ctx->mass64; a.prm.gz = ctx->gz64;
    for (int i = 0; i < 3; i++) a.prm.inv_inertia[i] = 1.0 / ctx->inertia64[i];
}

}  // namespace osm
void os_mpc_launch_quad(const osm::MpcArgs &a, uint32_t nst_mask, int *counters, int cu_count, hipStream_t s, const osm::PostArgs *post);      // mpc_quad.hip
namespace osq { struct RowsArgs { osk::KfRunArgs kf; osm::MpcParams prm; float *f_out; int32_t *iters; int max_iter, cold; const float *qr; }; }      // mpc_quad.hip (same layout)
void os_mpc_launch_rows(const osq::RowsArgs &a, int nst, int *counter, int cu_count, hipStream_t s);
namespace osm {

// layout of the context's QP scratch (floats): u [B][64] doubles | faces [B][64] bytes | contact [B] | todo [2][B] | counters (32 ints per
// shard) | records [B] x 1,792 bytes (16-byte aligned) | done marks [B]
constexpr int MAX_SHARDS = 4;
struct HandLayout {
    size_t faces, contact, todo, counters, rec, done, need;
    explicit HandLayout(size_t Bz)
    {
        faces = Bz * 128; contact = faces + Bz * 16; todo = contact + Bz; counters = todo + 2 * Bz;
        rec = (counters + 32 * MAX_SHARDS + 3) & ~(size_t)3; done = rec + Bz * 448; need = done + Bz;
    }
};
static std::atomic<uint32_t> g_post_seq{0};
// a contiguous part of the batch solved by one launch sequence: the caller's pointers are offset to trajectory b0, n trajectories behind
// them, the row stride stays the batch's B (os_kf_mpc_run: two halves on two streams, one in the drain of its QP launch -- a few
// stragglers, most of the chip idle -- while the other is in the bulk of its own)
struct Shard { int index, b0, n; int32_t *counters; };      // counters: 32 zeroed ints of this launch (null: the scratch's, zeroed here)

// nst_mask: bit n set = launch the instance for n legs on the ground (bit 0 rides on the 1-leg instance).  Batches of at least
// ctx->tune_mpc_quad problems: those with one / two force-carrying legs (15 / 30 variables) run sixteen lanes each, four to a
// wavefront, rows fetching problems from a work counter (mpc_quad.hip); three / four legs stay on the wavefront-per-QP instances.
// post (os_kf_mpc_run, one / two legs only): the filter step of every trajectory inside the same launch; post->done / seq are set here.
static bool quad_path(const os_ctx *ctx, int n, uint32_t nst_mask) { return ctx->tune_mpc_quad != 0 && n >= ctx->tune_mpc_quad && (nst_mask & 7u); }
static int launch_instances(os_ctx *ctx, const MpcArgs &a_in, uint32_t nst_mask, hipStream_t s, PostArgs *post = nullptr, const Shard *shard = nullptr)
{
    MpcArgs a = a_in;
    const Shard sh = shard ? *shard : Shard{0, 0, a.B, nullptr};
    a.n = sh.n;
    a.cap = 0; a.cold_in = 0; a.todo = nullptr; a.todo_count = nullptr; a.rec = nullptr;
    const dim3 grid(a.n), block(64);
    if (quad_path(ctx, a.n, nst_mask)) {
        // pass 1: sixteen lanes per QP, rows fetch problems from a work counter and give one up after `cap` iterations;
        // pass 2: the problems handed over continue on a wavefront of their own.  The hand-over record is the warm-start record: a cold
        // solve borrows the context's scratch for it.
        const size_t Bz = (size_t)a.B, b0 = (size_t)sh.b0;
        const HandLayout hl(Bz);
        if (os_ensure_scratch(ctx, &ctx->mpc_hand, &ctx->mpc_hand_floats, hl.need)) return -10;
        int32_t *counters = sh.counters ? sh.counters : (int32_t *)(ctx->mpc_hand + hl.counters) + 32 * sh.index;
        a.todo = (int32_t *)(ctx->mpc_hand + hl.todo);
        a.todo_count = counters + 16;
        a.rec = (double *)(ctx->mpc_hand + hl.rec) + b0 * 224;
        a.cap = (post || shard) ? 0 : ctx->tune_mpc_cap;
        if (!a.warm_u) {
            a.warm_u = (double *)ctx->mpc_hand + b0 * 64; a.warm_state = (uint8_t *)(ctx->mpc_hand + hl.faces) + b0 * 64;
            a.warm_contact = (uint32_t *)(ctx->mpc_hand + hl.contact) + b0;
            a.cold_in = 1;
        }
        if (post) {
            post->done = (uint32_t *)(ctx->mpc_hand + hl.done) + b0;
            uint32_t q = ++g_post_seq;
            if (q == 0) q = ++g_post_seq;
            post->seq = q;
        }
        if (!sh.counters && hipMemsetAsync(counters, 0, 128, s) != hipSuccess) return os_fail(ctx, -10, "os_mpc_solve: hipMemsetAsync failed");
        os_mpc_launch_quad(a, nst_mask & 7u, counters, ctx->cu_count, s, post);
        if (a.cap > 0) {
            if (nst_mask & 3u) hipLaunchKernelGGL(mpc_solve_kernel<1>, grid, block, 0, s, a, 1);
            if (nst_mask & 4u) hipLaunchKernelGGL(mpc_solve_kernel<2>, grid, block, 0, s, a, 1);
        }
        nst_mask &= ~7u;
        a.cap = 0; a.todo = nullptr;
        if (a.cold_in) { a.warm_u = nullptr; a.warm_state = nullptr; a.warm_contact = nullptr; a.cold_in = 0; }
    }
    if (nst_mask & 3u) hipLaunchKernelGGL(mpc_solve_kernel<1>, grid, block, 0, s, a, 0);
    if (nst_mask & 4u) hipLaunchKernelGGL(mpc_solve_kernel<2>, grid, block, 0, s, a, 0);
    if (nst_mask & 8u) hipLaunchKernelGGL(mpc_solve_kernel<3>, grid, block, 0, s, a, 0);
    if (nst_mask & 16u) hipLaunchKernelGGL(mpc_solve_kernel<4>, grid, block, 0, s, a, 0);
    return 0;
}

// the extra streams of the sharded os_kf_mpc_run, one set per device for the life of the process (contexts of a device share them: their
// work is ordered by the fork / join events of each call)
static hipStream_t g_shard_stream[16][MAX_SHARDS];
static std::mutex g_shard_mutex;
static hipStream_t shard_stream(int device, int i)
{
    std::lock_guard<std::mutex> lk(g_shard_mutex);
    if (device < 0 || device >= 16) return nullptr;
    if (!g_shard_stream[device][i] && hipStreamCreateWithFlags(&g_shard_stream[device][i], hipStreamNonBlocking) != hipSuccess) g_shard_stream[device][i] = nullptr;
    return g_shard_stream[device][i];
}

}  // namespace osm

int os_kf_run_impl(os_ctx *ctx, osk::KfRunArgs &a, uint32_t flags, hipStream_t s);   // kf_kernels.hip
int os_ensure_scratch(os_ctx *ctx, float **buf, size_t *cap, size_t need_floats);   // gru_kernels.hip

extern "C" {

int os_mpc_set_weights(os_ctx *ctx, const double *q_weights, double r_weight, double mu, double fz_max)
{
    OS_CHECK_CTX(ctx);
    if (!q_weights) return os_fail(ctx, -2, "os_mpc_set_weights: null weights");
    for (int i = 0; i < 12; i++) {
        if (!(q_weights[i] >= 0.0)) return os_fail(ctx, -2, "os_mpc_set_weights: weights must be non-negative");
        ctx->mpc_w[i] = q_weights[i];
    }
    if (!(r_weight > 0.0) || !(mu >= 0.0) || !(fz_max > 0.0)) return os_fail(ctx, -2, "os_mpc_set_weights: need r_weight > 0, mu >= 0, fz_max > 0");
    ctx->mpc_rw = r_weight; ctx->mpc_mu = mu; ctx->mpc_fzmax = fz_max;
    return 0;
}

int os_mpc_solve(os_ctx *ctx, int32_t B, const float *x, const float *body_ref, const float *p, const uint32_t *contact,
                 float *f_out, float *u_out, int32_t *iters, int32_t *status, int32_t max_iter, void *stream)
{
    OS_CHECK_CTX(ctx);
    if (B <= 0) return os_fail(ctx, -2, "os_mpc_solve: B must be positive");
    if (!x || !body_ref || !p || !contact || !f_out || !status) return os_fail(ctx, -2, "os_mpc_solve: null required pointer");
    OS_HIP(ctx, hipSetDevice(ctx->device));
    osm::MpcArgs a;
    a.B = B; a.x = x; a.ref = body_ref; a.p = p; a.contact = contact; a.f_out = f_out; a.u_out = u_out; a.iters = iters;
    a.status = status; a.max_iter = max_iter > 0 ? max_iter : 200;
    a.warm_u = nullptr; a.warm_state = nullptr; a.warm_contact = nullptr;
    osm::fill_args(ctx, a);
    hipStream_t s = (hipStream_t)stream;
    const int slot = os_prof_begin(ctx, 4, s, osm::quad_path(ctx, B, 7u) ? "mpc_prep_kernel + mpc_solve_quad_kernel<NST>" : "mpc_solve_kernel<NST>");
    if (int rcl = osm::launch_instances(ctx, a, 31u, s)) return rcl;       // all leg counts: a wavefront whose problem has another count exits at once
    os_prof_end(ctx, slot, s);
    OS_HIP(ctx, hipGetLastError());
    return 0;
}

}  // extern "C"

// One QP for os_kf_step's OS_STEP_MPC form (kf_step.hip): B = 1, only the solver instance the contact word needs is launched, warm
// start from the context's previous solve when the contact pattern is unchanged (the pyramids do not move: the old u stays feasible).
int os_mpc_solve_one(os_ctx *ctx, const float *x, const float *body_ref, const float *p, const uint32_t *contact, uint32_t contact_word,
                     float *f_out, float *u_out, int32_t *iters, int32_t *status, double *warm_u, uint8_t *warm_state, uint32_t *warm_contact,
                     hipStream_t s)
{
    osm::MpcArgs a;
    a.B = 1; a.x = x; a.ref = body_ref; a.p = p; a.contact = contact; a.f_out = f_out; a.u_out = u_out; a.iters = iters;
    a.status = status; a.max_iter = 200;
    a.warm_u = warm_u; a.warm_state = warm_state; a.warm_contact = warm_contact;
    osm::fill_args(ctx, a);
    int nst = 0;
    for (int l = 0; l < 4; l++) nst += ((contact_word >> (8 * l)) & 0xffu) == 1u;
    const int slot = os_prof_begin(ctx, 4, s, "mpc_solve_kernel<NST>");
    if (int rcl = osm::launch_instances(ctx, a, 1u << nst, s)) return rcl;
    os_prof_end(ctx, slot, s);
    if (hipGetLastError() != hipSuccess) return os_fail(ctx, -10, "os_kf_step: QP launch failed");
    return 0;
}

extern "C" {

int os_kf_mpc_run(os_ctx *ctx, int32_t B, int32_t T, const float *p, const float *dp, const float *imu,
                  const uint32_t *contact, const float *body_ref, float *x, float *P, float *x_out, float *f_out,
                  float *p_rot_out, float *ptrace_out, float *kgain_out, int32_t *mpc_iters, int32_t *status,
                  uint32_t flags, void *stream)
{
    OS_CHECK_CTX(ctx);
    if (B <= 0 || T <= 0) return os_fail(ctx, -2, "os_kf_mpc_run: B and T must be positive");
    if (!p || !dp || !imu || !contact || !body_ref || !x || !P || !x_out || !f_out || !status)
        return os_fail(ctx, -2, "os_kf_mpc_run: null required pointer");
    OS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    // Three forms (OS_MPC_PERSISTENT=1, the default, picks; 0 / 2 force the launch sequence / the wavefront-per-trajectory kernel):
    //  * kf_mpc_persistent_kernel, a wavefront per trajectory for all T steps: up to 32 trajectories per CU unless the rows form takes the
    //    batch (it spends a whole wavefront on one trajectory and saturates at ~2.2e7 steps/s);
    //  * kf_mpc_rows_kernel (mpc_quad.hip), a 16-lane row per trajectory for all T steps: batches of 8 .. 200 trajectories per CU whose every
    //    step carries force on NST legs or none, the plain call (no flag but OS_MPC_COLD_START, no P_trace / K_gain output); OS_MPC_ROWS=0
    //    switches it off, OS_MPC_ROWS=<lo>:<hi> moves the range;
    //  * the per-step launch sequence (two concurrent parts, the filter step inside the QP launch) for everything else.
    // Measured on 256 CUs at the end of round 6, steps/s at T = 100, wavefront kernel / rows kernel / sequence: B = 2,048 1.76e7 / 1.82e7 /
    // 0.59e7, 2,560 1.70e7 / 2.39e7 / -, 4,096 1.84e7 / 3.80e7 / 1.10e7, 8,192 2.06e7 / 5.13e7 / 2.14e7, 16,384 2.20e7 / 5.74e7 / 3.20e7,
    // 32,768 2.24e7 / 6.37e7 / 5.13e7, 40,960 - / 6.53e7 / 5.91e7, 57,344 - / 6.74e7 / 6.70e7, 65,536 2.27e7 / 6.72e7 / 6.98e7, 98,304 - /
    // 6.87e7 / 7.73e7, 131,072 - / 6.99e7 / 8.16e7.
    int rows_lo = 8 * ctx->cu_count, rows_hi = 200 * ctx->cu_count;
    if (const char *re = getenv("OS_MPC_ROWS")) {
        if (sscanf(re, "%d:%d", &rows_lo, &rows_hi) != 2) { rows_lo = 0; rows_hi = -1; }       // ("0": off)
    }
    const bool rows_plain = (flags & ~(uint32_t)OS_MPC_COLD_START) == 0 && !ptrace_out && !kgain_out && ctx->kf_qr && ctx->tune_mpc_quad != 0 &&
                            (size_t)B * 144 * 4 < 0xffffffffull && (uint64_t)T * 12ull * (uint64_t)B * 4ull < 0xffffffffull;      // (a stream under one descriptor)
    const bool rows_wanted = rows_plain && B >= rows_lo && B <= rows_hi && ctx->tune_mpc_persistent == 1;
    auto launch_persistent = [&]() -> int {
        // one launch, one wavefront per trajectory for all T steps; nothing is read back, nothing synchronises
        osm::MpcRunArgs m;
        osk::KfRunArgs &a = m.kf;
        a.B = B; a.T = T; a.p = p; a.f = f_out; a.dp = dp; a.imu = imu; a.contact = contact; a.body_ref = body_ref;
        a.x = x; a.P = P; a.x_out = x_out; a.p_rot_out = p_rot_out; a.ptrace_out = ptrace_out; a.kgain_out = kgain_out;
        a.status = status; a.accel = nullptr; a.minmax = nullptr; a.feat_out = nullptr; a.feat_I = 0;
        a.k = ctx->k;
        m.f_out = f_out; m.iters = mpc_iters; m.max_iter = 200; m.cold = (flags & OS_MPC_COLD_START) ? 1 : 0;
        {
            osm::MpcArgs tmp;
            osm::fill_args(ctx, tmp);
            m.prm = tmp.prm;
        }
        const int slot = os_prof_begin(ctx, 4, s, "kf_mpc_persistent_kernel");
        // (a two-waves-per-SIMD instance, 256 registers, served B >= 8 x CUs until round 5: 325 spilled VGPRs; since the filter half
        // runs on the row layout the one-wave-per-SIMD instance is 15-34 % faster there too: B = 3,072 1.15e7 -> 1.54e7 steps/s)
        hipLaunchKernelGGL(osm::kf_mpc_persistent_kernel<1>, dim3(B), dim3(64), 0, s, m);
        os_prof_end(ctx, slot, s);
        OS_HIP(ctx, hipGetLastError());
        return 0;
    };
    const bool small = ctx->tune_mpc_persistent == 2 || (ctx->tune_mpc_persistent == 1 && B <= 32 * ctx->cu_count);
    if (small && !rows_wanted) return launch_persistent();
    // scratch: warm-start store (u [B][64] doubles, faces [B][64] bytes, contact [B]) + per-step Kalman status [B] +
    // leg-count presence flags [T]
    // + the work / ticket counters of every (step, part) launch, zeroed once per call (a fill launch per step otherwise)
    const size_t cnt_at = (size_t)B * 146 + (size_t)T, cnt_floats = (size_t)T * osm::MAX_SHARDS * 32;
    const size_t need = cnt_at + cnt_floats;
    if (os_ensure_scratch(ctx, &ctx->mpc_scratch, &ctx->mpc_scratch_floats, need)) return -10;
    double *warm_u = (double *)ctx->mpc_scratch;                       // hipMalloc alignment covers the doubles
    uint8_t *warm_state = (uint8_t *)(ctx->mpc_scratch + (size_t)B * 128);
    uint32_t *warm_contact = (uint32_t *)(ctx->mpc_scratch + (size_t)B * 144);
    int32_t *st_step = (int32_t *)(ctx->mpc_scratch + (size_t)B * 145);
    uint32_t *flags_d = (uint32_t *)(ctx->mpc_scratch + (size_t)B * 146);
    // no previous solution: u = 0 with all faces free is feasible for every contact word, so even a match is harmless
    OS_HIP(ctx, hipMemsetAsync(warm_u, 0, (size_t)B * 64 * sizeof(double), s));
    OS_HIP(ctx, hipMemsetAsync(warm_state, 0x05, (size_t)B * 64, s));
    OS_HIP(ctx, hipMemsetAsync(warm_contact, 0xff, (size_t)B * sizeof(uint32_t), s));
    OS_HIP(ctx, hipMemsetAsync(status, 0, (size_t)B * sizeof(int32_t), s));
    OS_HIP(ctx, hipMemsetAsync(flags_d, 0, ((size_t)T + cnt_floats) * sizeof(uint32_t), s));      // (the counters lie behind the flags)
    int32_t *step_counters = (int32_t *)(ctx->mpc_scratch + cnt_at);
    {
        const int gx = (B + 255) / 256 < 64 ? (B + 255) / 256 : 64;
        hipLaunchKernelGGL(osm::nst_presence_kernel, dim3(gx, T), dim3(256), 0, s, B, T, contact, flags_d);
    }
    // The filter step inside the QP launch (mpc_quad.hip drain phase) where the step is the plain one: batch update, no P_trace /
    // K_gain outputs, every trajectory with at most two force-carrying legs.  OS_MPC_FUSE_KF=0: the separate launch everywhere.
    const char *fuse_env = getenv("OS_MPC_FUSE_KF");
    const bool can_fuse = !(fuse_env && fuse_env[0] == '0') && (flags & ~(uint32_t)OS_MPC_COLD_START) == 0 && !ptrace_out && !kgain_out &&
                          ctx->kf_qr && (size_t)B * 144 * 4 < 0xffffffffull && ctx->tune_mpc_cap == 0;
    if (can_fuse && osm::quad_path(ctx, B, 7u)) {
        // (the done marks once per call: whatever an earlier call with another batch size left at these addresses never matches)
        const osm::HandLayout hl((size_t)B);
        if (os_ensure_scratch(ctx, &ctx->mpc_hand, &ctx->mpc_hand_floats, hl.need)) return -10;
        OS_HIP(ctx, hipMemsetAsync(ctx->mpc_hand + hl.done, 0, (size_t)B * 4, s));
    }
    // one host read-back per call: only the solver instances a step needs are launched
    uint32_t *flags_h = (uint32_t *)malloc((size_t)T * sizeof(uint32_t));
    if (!flags_h) return os_fail(ctx, -13, "os_kf_mpc_run: out of host memory");
    hipError_t e = hipMemcpyAsync(flags_h, flags_d, (size_t)T * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { free(flags_h); return os_fail(ctx, -10, hipGetErrorString(e)); }

    if (rows_wanted) {
        // every step: force on exactly NST legs or on none, the same NST throughout
        uint32_t all = 0;
        for (int t = 0; t < T; t++) all |= flags_h[t];
        const uint32_t nz = all & ~1u;
        if (nz == 2u || nz == 4u) {
            free(flags_h);
            osq::RowsArgs r;
            osk::KfRunArgs &a = r.kf;
            a.B = B; a.T = T; a.p = p; a.f = f_out; a.dp = dp; a.imu = imu; a.contact = contact; a.body_ref = body_ref;
            a.x = x; a.P = P; a.x_out = x_out; a.p_rot_out = p_rot_out; a.ptrace_out = nullptr; a.kgain_out = nullptr;
            a.status = status; a.accel = nullptr; a.minmax = nullptr; a.feat_out = nullptr; a.feat_I = 0;
            a.k = ctx->k;
            r.f_out = f_out; r.iters = mpc_iters; r.max_iter = 200; r.cold = (flags & OS_MPC_COLD_START) ? 1 : 0;
            r.qr = (const float *)ctx->kf_qr;
            {
                osm::MpcArgs tmp;
                osm::fill_args(ctx, tmp);
                r.prm = tmp.prm;
            }
            // (status was zeroed above; the counter: the first of the per-step counters, zeroed with them)
            const int slot = os_prof_begin(ctx, 4, s, "kf_mpc_rows_kernel<NST> (a 16-lane row per trajectory for all T steps)");
            os_mpc_launch_rows(r, nz == 4u ? 2 : 1, step_counters, ctx->cu_count, s);
            os_prof_end(ctx, slot, s);
            OS_HIP(ctx, hipGetLastError());
            return 0;
        }
        if (small) { free(flags_h); return launch_persistent(); }      // (mixed leg counts: the wavefront-per-trajectory kernel takes them all)
    }
    // Shards: when EVERY step of the call runs the fused form, the batch is cut into contiguous parts on streams of their own (the
    // caller's stream forks into them and joins them).  A part's QP launch ends with a few stragglers on an almost idle chip; the
    // other part's launches fill it (measured at B = 65,536: 1.15 -> 1.05 ms per step with two parts; three and four gain nothing; at
    // 12,288 / 16,384 / 24,576: +22 / +17 / +15 %).  OS_MPC_SHARDS=1 keeps one part.
    int S = 1;
    {
        const char *sh_env = getenv("OS_MPC_SHARDS");
        int want = sh_env ? atoi(sh_env) : 2;
        if (want > osm::MAX_SHARDS) want = osm::MAX_SHARDS;
        bool all_fused = can_fuse && want > 1 && B / want >= 4096;
        for (int t = 0; t < T && all_fused; t++) all_fused = (flags_h[t] & ~7u) == 0 && osm::quad_path(ctx, B / want - 16, flags_h[t]);
        if (all_fused) S = want;
    }
    hipStream_t st[osm::MAX_SHARDS] = {s, nullptr, nullptr, nullptr};
    osm::Shard shard[osm::MAX_SHARDS];
    hipEvent_t ev_fork = nullptr, ev_join[osm::MAX_SHARDS] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0, b0 = 0; i < S; i++) {
        const int b1 = i + 1 == S ? B : (int)(((int64_t)B * (i + 1) / S) / 16 * 16);
        shard[i] = osm::Shard{i, b0, b1 - b0, nullptr};
        b0 = b1;
        if (i > 0 && !(st[i] = osm::shard_stream(ctx->device, i))) S = 1;
    }
    if (S == 1) shard[0] = osm::Shard{0, 0, B, nullptr};
    if (S > 1) {
        bool ok = hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) == hipSuccess && hipEventRecord(ev_fork, s) == hipSuccess;
        for (int i = 1; i < S && ok; i++)
            ok = hipEventCreateWithFlags(&ev_join[i], hipEventDisableTiming) == hipSuccess && hipStreamWaitEvent(st[i], ev_fork, 0) == hipSuccess;
        if (!ok) { free(flags_h); return os_fail(ctx, -10, "os_kf_mpc_run: could not fork the shard streams"); }
    }

    osm::MpcArgs m;
    m.B = B; m.n = B; m.u_out = nullptr; m.max_iter = 200;
    osm::fill_args(ctx, m);
    const bool keep_warm = !(flags & OS_MPC_COLD_START);
    int rc = 0;
    for (int t = 0; t < T && rc == 0; t++) {
        const size_t o12 = (size_t)t * 12 * B, o6 = (size_t)t * 6 * B, o1 = (size_t)t * B;
        const bool fuse = can_fuse && osm::quad_path(ctx, shard[0].n, flags_h[t]) && (flags_h[t] & ~7u) == 0;
        for (int i = 0; i < S && rc == 0; i++) {
            const size_t b0 = (size_t)shard[i].b0;
            // forces from the state BEFORE this step's predict (kalman_filter.py:141-152) ...
            m.x = x + b0; m.status = status + b0;
            m.ref = body_ref + o12 + b0; m.p = p + o12 + b0; m.contact = contact + o1 + b0; m.f_out = f_out + o12 + b0;
            m.iters = mpc_iters ? mpc_iters + o1 + b0 : nullptr;
            m.warm_u = keep_warm ? warm_u + b0 * 64 : nullptr; m.warm_state = warm_state + b0 * 64; m.warm_contact = warm_contact + b0;
            // ... then get_odom + set_measurements + predict_mpc covariance + next_state + update (kalman_filter.py:176-182)
            osk::KfRunArgs a;
            a.B = B; a.T = 1; a.p = p + o12 + b0; a.f = f_out + o12 + b0; a.dp = dp + o12 + b0; a.imu = imu + o6 + b0; a.contact = contact + o1 + b0;
            a.body_ref = body_ref + o12 + b0; a.x = x + b0; a.P = P + b0; a.x_out = x_out + o12 + b0;
            a.p_rot_out = p_rot_out ? p_rot_out + o12 + b0 : nullptr;
            a.ptrace_out = ptrace_out ? ptrace_out + o1 : nullptr; a.kgain_out = kgain_out ? kgain_out + o1 : nullptr;
            a.status = st_step; a.accel = nullptr; a.minmax = nullptr; a.feat_out = nullptr; a.feat_I = 0;
            const int slot = os_prof_begin(ctx, 4, st[i], fuse ? (S > 1 ? "mpc_prep_kernel + mpc_solve_quad_kernel<NST, filter step inside> (a part of the batch, two in flight)"
                                                                          : "mpc_prep_kernel + mpc_solve_quad_kernel<NST, filter step inside>")
                                                               : osm::quad_path(ctx, B, flags_h[t]) ? "mpc_prep_kernel + mpc_solve_quad_kernel<NST>" : "mpc_solve_kernel<NST>");
            if (fuse) {
                osm::PostArgs post;
                post.kf = a; post.kf.k = ctx->k; post.kf.status = status + b0;
                post.qr = (const float *)ctx->kf_qr; post.done = nullptr; post.seq = 0;
                shard[i].counters = step_counters + ((size_t)t * osm::MAX_SHARDS + i) * 32;
                rc = osm::launch_instances(ctx, m, flags_h[t], st[i], &post, &shard[i]);          // (a failure still joins the parts below)
                os_prof_end(ctx, slot, st[i]);
                continue;
            }
            // (the separate launches: one part only -- S > 1 needs every step fused)
            rc = osm::launch_instances(ctx, m, flags_h[t], s);
            os_prof_end(ctx, slot, s);
            if (rc) break;
            rc = os_kf_run_impl(ctx, a, (flags | OS_KF_DENSE_FD) & ~(uint32_t)OS_KF_SYMMETRIC_P, s);
            hipLaunchKernelGGL(osm::or_status_kernel, dim3((B + 255) / 256), dim3(256), 0, s, B, status, st_step);
        }
    }
    for (int i = 1; i < S; i++) {
        if (hipEventRecord(ev_join[i], st[i]) != hipSuccess || hipStreamWaitEvent(s, ev_join[i], 0) != hipSuccess) rc = rc ? rc : -10;
        hipEventDestroy(ev_join[i]);
    }
    if (ev_fork) hipEventDestroy(ev_fork);
    free(flags_h);
    if (rc) return rc;
    OS_HIP(ctx, hipGetLastError());
    return 0;
}

}  // extern "C"
