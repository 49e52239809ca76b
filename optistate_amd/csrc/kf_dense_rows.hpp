// kf_dense_rows.hpp -- float64 filter arithmetic on the 16-lanes-per-trajectory layout: lane r of a DPP row holds row r of P.
// Shared by kf_dense_rows_kernel (kf_dense_rows.hip: the layout, the algebra and the DPP hazard rules are described there) and by
// the filter half of kf_mpc_persistent_kernel (mpc_kernels.hip), where all four 16-lane rows of the wavefront carry the SAME
// trajectory.
#pragma once
#include "kf_device.hpp"

namespace osk {
namespace rows64 {

template <int SRC>
__device__ __forceinline__ float bc32(float v)       // value of lane SRC of this lane's 16-lane row
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + SRC, 0xf, 0xf, true));
}
template <int SRC>
__device__ __forceinline__ double bc64(double v)
{
    double o;
    asm volatile("s_nop 1\nv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v), "n"(SRC));
    return o;
}
// acc += (src of lane SRC) * m.  NOP = false inside a chain whose DPP sources were written before the chain started (the
// chain's first member carries the s_nop).
template <int SRC, bool NOP = true>
__device__ __forceinline__ void fmacb(double &acc, double src, double m)
{
    if (NOP) asm volatile("s_nop 1\nv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+&v"(acc) : "v"(src), "v"(m), "n"(SRC));
    else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+&v"(acc) : "v"(src), "v"(m), "n"(SRC));
}
// (the accumulators of the multi-instruction statements are EARLY-CLOBBER operands: an accumulator initialised with the value of the
// broadcast source -- M[j] = P[j] in predict_struct_row -- must not share its register, or a later member of the statement reads, in
// other lanes, a register the earlier members have just written)
#define OSD_F(acc, src, m, S) "v_fmac_f64_dpp %" #acc ", %" #src ", %" #m " row_newbcast:" #S " row_mask:0xf bank_mask:0xf\n"
// acc += sum over lanes 0..11 of the row of src (m = 1.0 in a register: a DPP instruction takes no inline constant)
__device__ __forceinline__ void rowsum12(double &acc, double src, double one)
{
    asm volatile("s_nop 1\n" OSD_F(0, 1, 2, 0) OSD_F(0, 1, 2, 1) OSD_F(0, 1, 2, 2) OSD_F(0, 1, 2, 3) OSD_F(0, 1, 2, 4) OSD_F(0, 1, 2, 5)
                 OSD_F(0, 1, 2, 6) OSD_F(0, 1, 2, 7) OSD_F(0, 1, 2, 8) OSD_F(0, 1, 2, 9) OSD_F(0, 1, 2, 10) OSD_F(0, 1, 2, 11)
                 : "+&v"(acc) : "v"(src), "v"(one));
}
// acc += sum_k cf[k] * (src of lane 6 + k): row i of E applied to a column held one entry per lane (rows 6..11)
__device__ __forceinline__ void ecomb6(double &acc, double src, const double *cf)
{
    asm volatile("s_nop 1\n" OSD_F(0, 1, 2, 6) OSD_F(0, 1, 3, 7) OSD_F(0, 1, 4, 8) OSD_F(0, 1, 5, 9) OSD_F(0, 1, 6, 10) OSD_F(0, 1, 7, 11)
                 : "+&v"(acc) : "v"(src), "v"(cf[0]), "v"(cf[1]), "v"(cf[2]), "v"(cf[3]), "v"(cf[4]), "v"(cf[5]));
}
// P[j] += (P[j] of lane S) * m for the twelve entries of the lane's row: the rank-1 update of a sequential measurement
template <int S>
__device__ __forceinline__ void rank1_row(double *P, double m)
{
#define OSD_R(j) "v_fmac_f64_dpp %" #j ", %" #j ", %12 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n"
    asm volatile("s_nop 1\n" OSD_R(0) OSD_R(1) OSD_R(2) OSD_R(3) OSD_R(4) OSD_R(5) OSD_R(6) OSD_R(7) OSD_R(8) OSD_R(9) OSD_R(10) OSD_R(11)
                 : "+v"(P[0]), "+v"(P[1]), "+v"(P[2]), "+v"(P[3]), "+v"(P[4]), "+v"(P[5]), "+v"(P[6]), "+v"(P[7]), "+v"(P[8]),
                   "+v"(P[9]), "+v"(P[10]), "+v"(P[11])
                 : "v"(m), "n"(S));
#undef OSD_R
}
// acc += sum_a K[a] * (src of lane SEL[a]): one column of K P[sel,:]
__device__ __forceinline__ void kdot_sel(double &acc, double src, const double *K)
{
    asm volatile("s_nop 1\n" OSD_F(0, 1, 2, 0) OSD_F(0, 1, 3, 1) OSD_F(0, 1, 4, 2) OSD_F(0, 1, 5, 5) OSD_F(0, 1, 6, 6) OSD_F(0, 1, 7, 7)
                 OSD_F(0, 1, 8, 8) OSD_F(0, 1, 9, 9) OSD_F(0, 1, 10, 10) OSD_F(0, 1, 11, 11)
                 : "+&v"(acc) : "v"(src), "v"(K[0]), "v"(K[1]), "v"(K[2]), "v"(K[3]), "v"(K[4]), "v"(K[5]), "v"(K[6]), "v"(K[7]),
                   "v"(K[8]), "v"(K[9]));
}

__device__ __forceinline__ double rcp64(double s)          // v_rcp_f64 (~2^-26) + two Newton steps
{
    double y = __builtin_amdgcn_rcp(s);
    double e = fma(-s, y, 1.0); y = fma(y, e, y);
    e = fma(-s, y, 1.0); y = fma(y, e, y);
    return y;
}
__device__ __forceinline__ double rsqrt64(double d)        // v_rsq_f64 + two coupled (Goldschmidt) steps: returns 1 / sqrt(d)
{
    double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y, r = fma(-g, h, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    r = fma(-g, h, 0.5); h = fma(h, r, h);
    return 2.0 * h;
}

template <int N> struct IC { static constexpr int v = N; };
// for_sel(f): f(IC<a>{}, IC<SEL[a]>{}) for the ten measurements in order
template <typename F>
__device__ __forceinline__ void for_sel(F &&f)
{
    f(IC<0>{}, IC<0>{}); f(IC<1>{}, IC<1>{}); f(IC<2>{}, IC<2>{}); f(IC<3>{}, IC<5>{}); f(IC<4>{}, IC<6>{});
    f(IC<5>{}, IC<7>{}); f(IC<6>{}, IC<8>{}); f(IC<7>{}, IC<9>{}); f(IC<8>{}, IC<10>{}); f(IC<9>{}, IC<11>{});
}

// ---- covariance predict: P <- F_d P F_d^T + Q with F_d = 1 1^T + E (header comment), row r of P in lane r ----
// e[3 i + k] = expm1(dt Rb[k][i]) (same in every lane of the trajectory), ed = expm1(dt); cf[0..5]: this lane's row of E over
// the source rows 6..11 (zero for lanes >= 6); qrow: the lane's row of Q.
__device__ __forceinline__ void predict_dense_row(double *P, const double *qrow, const double *e, double ed, const double *cf, double one)
{
    double rho = P[0];
#pragma unroll
    for (int j = 1; j < NS; j++) rho += P[j];
    double sb = 0.0;                                   // sigma + (E rho)[i]
    rowsum12(sb, rho, one);
    ecomb6(sb, rho, cf);
    double c[6], T[6];
#pragma unroll
    for (int l = 0; l < 6; l++) {
        c[l] = 0.0; T[l] = 0.0;
        rowsum12(c[l], P[6 + l], one);                 // column sums 6..11
        ecomb6(T[l], P[6 + l], cf);                    // (E P)[i][6 + l]
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
        // (E c)[j] + (E P E^T)[i][j] = sum_k e[3 j + k] (c[k] + T[k])      (columns 0..2: E[j][6 + k] = e[3 j + k])
        P[j] = sb + qrow[j] + (e[3 * j] * (c[0] + T[0]) + e[3 * j + 1] * (c[1] + T[1]) + e[3 * j + 2] * (c[2] + T[2]));
        P[3 + j] = sb + qrow[3 + j] + ed * (c[3 + j] + T[3 + j]);           // columns 3..5: E[3 + j][9 + j] = ed
    }
#pragma unroll
    for (int j = 6; j < NS; j++) P[j] = sb + qrow[j];
}

// ---- sequential update (diagonal R): ten scalar measurements, kalman_filter.py:164-172 one row of H at a time ----
__device__ __forceinline__ int update_seq_row(double &xd, double *P, const float *z, const double *rdiag)
{
    int status = 0;
    for_sel([&](auto A, auto S) {
        constexpr int a = decltype(A)::v, s = decltype(S)::v;
        double sv = bc64<s>(P[s]) + rdiag[a];
        if (!(sv > 0.0) || !(sv < 1.0e300)) { status |= 1; sv = 1.0; }
        const double inv = rcp64(sv);
        const double innov = (double)z[a] - bc64<s>(xd);
        const double kc = P[s] * inv;                  // K[r] = P[r][s] / S
        xd = fma(kc, innov, xd);
        rank1_row<s>(P, -kc);                          // P[r][:] -= K[r] P[s][:]
    });
    return status;
}

// np.trace of the 12 x 10 gain (kalman_filter.py:174): K[a][a], a < 10, from the lanes' rows of K
__device__ __forceinline__ float kgain_rows(const double (&K)[NM], double one)
{
    double t = 0.0;
    asm volatile("s_nop 1\n" OSD_F(0, 1, 11, 0) OSD_F(0, 2, 11, 1) OSD_F(0, 3, 11, 2) OSD_F(0, 4, 11, 3) OSD_F(0, 5, 11, 4) OSD_F(0, 6, 11, 5)
                 OSD_F(0, 7, 11, 6) OSD_F(0, 8, 11, 7) OSD_F(0, 9, 11, 8) OSD_F(0, 10, 11, 9)
                 : "+&v"(t) : "v"(K[0]), "v"(K[1]), "v"(K[2]), "v"(K[3]), "v"(K[4]), "v"(K[5]), "v"(K[6]), "v"(K[7]), "v"(K[8]), "v"(K[9]), "v"(one));
    return (float)t;
}
// ---- batch update as the reference writes it: S = H P H^T + R, K = P H^T S^-1, x += K y, P -= K H P ----
// S is taken as it IS -- rounding leaves P, and with it S, not exactly symmetric, and the reference inverts that S
// (np.linalg.inv, kalman_filter.py:169).  K from a SYMMETRISED S (rounds 1-5a: a Cholesky of the lower triangle) is unstable with
// this covariance update: the antisymmetric part of P then grows step by step instead of staying at rounding level -- in float64
// the filter was lost after 70-110 steps of ill-conditioned runs (fitted noise, flight phases; found by tools/fuzz_kf.py; numpy
// reproduces it: the reference's form keeps |P - P^T| at 1e-15 over the same 160 steps).  So: LU of the full S, no pivoting (S is a
// positive definite matrix plus rounding noise), one row per lane.
// The lane of state row SEL[a] also owns measurement a (am = a; am < 0: none) and holds row a of S, then of the factors, in
// M[0..9]: multipliers L[a][q] (q < a) | 1 / U[a][a] | U[a][q] (q > a).  rrow: that lane's row of R.  K[0..9]: this lane's row of the gain.
template <int S>
__device__ __forceinline__ void fmac_self(double &acc, double m)      // acc += (acc of lane S) * m
{
    asm volatile("s_nop 1\nv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(m), "n"(S));
}
template <bool WANT_KGAIN = false>
__device__ __forceinline__ int update_batch_row(double &xd, double *P, const float *z, const double *rrow, double (&K)[NM], int am, double one = 1.0,
                                                float *kgain = nullptr)
{
    int status = 0;
    double M[NM];
    for_sel([&](auto Q, auto SQ) { M[decltype(Q)::v] = P[decltype(SQ)::v] + rrow[decltype(Q)::v]; });      // row a of S (row SEL[a] of P is this lane's)
    for_sel([&](auto J, auto SJ) {
        constexpr int j = decltype(J)::v, sj = decltype(SJ)::v;
        double d = bc64<sj>(M[j]);                     // the pivot U[j][j]
        if (!(d > 0.0) || !(d < 1.0e300)) { status |= 1; d = 1.0; }
        const double di = rcp64(d);
        const double m = am > j ? M[j] * di : 0.0;     // rows below the pivot: L[a][j]; every other lane: untouched
        const double nm = -m;
#pragma unroll
        for (int q = j + 1; q < NM; q++) fmac_self<sj>(M[q], nm);     // S[a][q] -= L[a][j] U[j][q]
        M[j] = am > j ? m : (am == j ? di : M[j]);     // (the pivot lane keeps 1 / U[j][j]: all the substitution needs of it)
        // M[j] is a DPP source of the substitutions below: pin its definition HERE (see tools/isa_dpp_hazard_scan.py)
        asm volatile("" : "+v"(M[j]));
    });
    // K[i,:] S = P[i,sel]:  w U = P[i,sel] (forward over the columns of U), then K L = w (backward, unit diagonal);
    // U[b][c] / L[b][c] = register c of the lane of measurement b
    // (w is built in K's registers and turned into K in place)
    for_sel([&](auto C_, auto SC) {
        constexpr int c = decltype(C_)::v, sc = decltype(SC)::v;
        double t = 0.0;
        for_sel([&](auto B_, auto SB) {
            constexpr int b = decltype(B_)::v, sb = decltype(SB)::v;
            if constexpr (b < c) fmacb<sb, true>(t, M[c], K[b]);
        });
        K[c] = (P[sc] - t) * bc64<sc>(M[c]);
    });
#define OSD_BACK(c, ...)                                            \
    {                                                               \
        double t = 0.0;                                             \
        __VA_ARGS__                                                 \
        K[c] -= t;                                                  \
    }
    OSD_BACK(9, )
    OSD_BACK(8, fmacb<11, true>(t, M[8], K[9]);)
    OSD_BACK(7, fmacb<10, true>(t, M[7], K[8]); fmacb<11, true>(t, M[7], K[9]);)
    OSD_BACK(6, fmacb<9, true>(t, M[6], K[7]); fmacb<10, true>(t, M[6], K[8]); fmacb<11, true>(t, M[6], K[9]);)
    OSD_BACK(5, fmacb<8, true>(t, M[5], K[6]); fmacb<9, true>(t, M[5], K[7]); fmacb<10, true>(t, M[5], K[8]); fmacb<11, true>(t, M[5], K[9]);)
    OSD_BACK(4, fmacb<7, true>(t, M[4], K[5]); fmacb<8, true>(t, M[4], K[6]); fmacb<9, true>(t, M[4], K[7]); fmacb<10, true>(t, M[4], K[8]); fmacb<11, true>(t, M[4], K[9]);)
    OSD_BACK(3, fmacb<6, true>(t, M[3], K[4]); fmacb<7, true>(t, M[3], K[5]); fmacb<8, true>(t, M[3], K[6]); fmacb<9, true>(t, M[3], K[7]); fmacb<10, true>(t, M[3], K[8]);
             fmacb<11, true>(t, M[3], K[9]);)
    OSD_BACK(2, fmacb<5, true>(t, M[2], K[3]); fmacb<6, true>(t, M[2], K[4]); fmacb<7, true>(t, M[2], K[5]); fmacb<8, true>(t, M[2], K[6]); fmacb<9, true>(t, M[2], K[7]);
             fmacb<10, true>(t, M[2], K[8]); fmacb<11, true>(t, M[2], K[9]);)
    OSD_BACK(1, fmacb<2, true>(t, M[1], K[2]); fmacb<5, true>(t, M[1], K[3]); fmacb<6, true>(t, M[1], K[4]); fmacb<7, true>(t, M[1], K[5]); fmacb<8, true>(t, M[1], K[6]);
             fmacb<9, true>(t, M[1], K[7]); fmacb<10, true>(t, M[1], K[8]); fmacb<11, true>(t, M[1], K[9]);)
    OSD_BACK(0, fmacb<1, true>(t, M[0], K[1]); fmacb<2, true>(t, M[0], K[2]); fmacb<5, true>(t, M[0], K[3]); fmacb<6, true>(t, M[0], K[4]); fmacb<7, true>(t, M[0], K[5]);
             fmacb<8, true>(t, M[0], K[6]); fmacb<9, true>(t, M[0], K[7]); fmacb<10, true>(t, M[0], K[8]); fmacb<11, true>(t, M[0], K[9]);)
#undef OSD_BACK
    // x += K (z - H x)
    double dx = 0.0;
    for_sel([&](auto A, auto SA) {
        constexpr int a = decltype(A)::v, sa = decltype(SA)::v;
        dx = fma(K[a], (double)z[a] - bc64<sa>(xd), dx);
    });
    xd += dx;
    if (WANT_KGAIN) *kgain = kgain_rows(K, one);       // here: K is live until the covariance update anyway
    // P[i][j] -= sum_a K[i][a] P[SEL[a]][j], column by column: within column j only the registers P[j] are read (from the
    // lanes of the selected rows) and they are written after all ten reads -- the OLD rows, as (I - K H) P needs
#pragma unroll
    for (int j = 0; j < NS; j++) {
        double t = 0.0;
        kdot_sel(t, P[j], K);
        P[j] -= t;
    }
    return status;
}

// sum over the 12 row lanes of a per-lane float64 value -> the same total in every lane
__device__ __forceinline__ double group_sum12(double v, double one)
{
    double t = 0.0;
    rowsum12(t, v, one);
    return t;
}


// ---- covariance predict of predict(p, f) (kalman_filter.py:124-135) on the row layout: F_d = I + G with G[0:3,6:9] = dt R^T,
// G[3:6,9:12] = dt I -- the sparsity of E above.  M = P + G P (the lane's row of G over the source rows 6..11 is cf), then
// P' = M + M G^T + Q in-lane.  g[3 i + k] = dt R[k][i].
__device__ __forceinline__ void predict_struct_row(double *P, const double *qrow, const double *g, double dt, const double *cf)
{
    double M[NS];
#pragma unroll
    for (int j = 0; j < NS; j++) {
        M[j] = P[j];
        ecomb6(M[j], P[j], cf);
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
        P[j] = M[j] + qrow[j] + (g[3 * j] * M[6] + g[3 * j + 1] * M[7] + g[3 * j + 2] * M[8]);
        P[3 + j] = M[3 + j] + qrow[3 + j] + dt * M[9 + j];
    }
#pragma unroll
    for (int j = 6; j < NS; j++) P[j] = M[j] + qrow[j];
}

// The replicated prior state from the lanes' own components
__device__ __forceinline__ void gather_state(float xr, float *x /*12*/)
{
    x[0] = bc32<0>(xr); x[1] = bc32<1>(xr); x[2] = bc32<2>(xr); x[3] = bc32<3>(xr); x[4] = bc32<4>(xr); x[5] = bc32<5>(xr);
    x[6] = bc32<6>(xr); x[7] = bc32<7>(xr); x[8] = bc32<8>(xr); x[9] = bc32<9>(xr); x[10] = bc32<10>(xr); x[11] = bc32<11>(xr);
}

// The front of a step on the row layout: get_odom + set_measurements (z), the predict_mpc covariance on the lane's row of P,
// next_state (x is the replicated prior on entry and the replicated prediction on return; pw = the rotated foot positions).
// r = lane & 15, xr = the lane's own prior component (angles for the sincos sharing), bref = body_ref's three angles,
// ed = expm1(dt), qrow = the lane's row of Q (float64), one = 1.0 in a register.  Returns status bit 4 or 0.
// DENSE = false: the covariance of predict(p, f) instead (kalman_filter.py:124-135: F_d = I + dt F from the PRIOR attitude;
// bref, ed unused).
template <bool DENSE = true>
__device__ __forceinline__ int front_row(float *x, float xr, double *P, const StepIn &in, const float *bref, const KfConst &k, double ed,
                                         const double *qrow, double one, int r, float *z, float *pw)
{
    // nine sincos per step (prior attitude | IMU attitude | body_ref attitude): lane r < 9 evaluates one angle
    float sv, cv;
    {
        const float ang = r < 3 ? xr : r == 3 ? in.imu[0] : r == 4 ? in.imu[1] : r == 5 ? in.imu[2] : r == 6 ? bref[0] : r == 7 ? bref[1] : bref[2];
        sincos_f32(ang, &sv, &cv);
    }
    const Rot rot = rotation_sc(bc32<0>(sv), bc32<0>(cv), bc32<1>(sv), bc32<1>(cv), bc32<2>(sv), bc32<2>(cv));
    const Rot rimu = rotation_sc(bc32<3>(sv), bc32<3>(cv), bc32<4>(sv), bc32<4>(cv), bc32<5>(sv), bc32<5>(cv));
    const Rot rbr = rotation_sc(bc32<6>(sv), bc32<6>(cv), bc32<7>(sv), bc32<7>(cv), bc32<8>(sv), bc32<8>(cv));
    measurement_r(in, rimu, z);
    // ---- covariance predict (kalman_filter.py:153-158) ----
    if (!DENSE) {
        double g[9], cf[6];
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
    } else {
        // e[3 i + kk] = expm1(dt Rb[kk][i]): lane n < 9 evaluates entry n
        float rbn = rbr.m[0];
#pragma unroll
        for (int n = 1; n < 9; n++) rbn = (r == n) ? rbr.m[3 * (n % 3) + n / 3] : rbn;
        const double en = expm1((double)k.dt * (double)rbn);
        double e[9];
        e[0] = bc64<0>(en); e[1] = bc64<1>(en); e[2] = bc64<2>(en); e[3] = bc64<3>(en); e[4] = bc64<4>(en);
        e[5] = bc64<5>(en); e[6] = bc64<6>(en); e[7] = bc64<7>(en); e[8] = bc64<8>(en);
        double cf[6];
#pragma unroll
        for (int kk = 0; kk < 3; kk++) {
            cf[kk] = r == 0 ? e[kk] : r == 1 ? e[3 + kk] : r == 2 ? e[6 + kk] : 0.0;
            cf[3 + kk] = (r == 3 + kk) ? ed : 0.0;
        }
        predict_dense_row(P, qrow, e, ed, cf, one);
    }
    // ---- next_state (replicated) ----
    return dynamics(x, rot, in.p, in.f, pw, k);
}

// the measurement a state row owns (SEL^-1); -1: rows 3, 4 and the idle lanes own none
__device__ __forceinline__ int row_measurement(int r) { return r < 3 ? r : (r >= 5 && r < 12) ? r - 2 : -1; }

// Diagonal sums without select chains (hipcc turns `r == i ? P[i] : ...` into a dynamically indexed array in scratch): entry
// (i, i) is register i of lane i, so the sum is twelve fused broadcasts, each from its own lane.
// trace(P) (P_trace, kalman_filter.py:138,173)
__device__ __forceinline__ float ptrace_rows(const double (&P)[NS], double one)
{
    double t = 0.0;
    asm volatile("s_nop 1\n" OSD_F(0, 1, 13, 0) OSD_F(0, 2, 13, 1) OSD_F(0, 3, 13, 2) OSD_F(0, 4, 13, 3) OSD_F(0, 5, 13, 4) OSD_F(0, 6, 13, 5)
                 OSD_F(0, 7, 13, 6) OSD_F(0, 8, 13, 7) OSD_F(0, 9, 13, 8) OSD_F(0, 10, 13, 9) OSD_F(0, 11, 13, 10) OSD_F(0, 12, 13, 11)
                 : "+&v"(t) : "v"(P[0]), "v"(P[1]), "v"(P[2]), "v"(P[3]), "v"(P[4]), "v"(P[5]), "v"(P[6]), "v"(P[7]), "v"(P[8]), "v"(P[9]),
                   "v"(P[10]), "v"(P[11]), "v"(one));
    return (float)t;
}
// the same from the posterior of the sequential form (diagonal R): K = P+ H^T R^-1, trace = sum_a P+[a][SEL[a]] / R[a][a]
__device__ __forceinline__ float kgain_posterior_rows(const double (&P)[NS], const double (&rdiag)[NM])
{
    double t = 0.0;
    for_sel([&](auto A, auto SA) {
        constexpr int a = decltype(A)::v, sa = decltype(SA)::v;
        fmacb<a>(t, P[sa], rcp64(rdiag[a]));           // (register SEL[a] of lane a) / R[a][a]
    });
    return (float)t;
}
#undef OSD_F

}  // namespace rows64
}  // namespace osk
