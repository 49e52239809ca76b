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
//     each) and batch as the reference writes it (S = P[sel,sel] + R -> Cholesky, row a of L in the lane of measurement a ->
//     K = P[:,sel] S^-1 by forward / back substitution, one ROW of K per lane -> P -= K P[sel,:] column by column);
//   * the float32 front (odometry, z, next_state with the int64-truncation quirk) is kf_device.hpp's code, computed
//     redundantly by a trajectory's 16 lanes; the nine sincos and the nine expm1 of a step are shared (one per lane).
// DPP hazard (inline asm is invisible to hipcc's hazard recogniser): a VALU result must be two instructions old before a DPP
// operand reads it.  Every assembly statement here starts with `s_nop 1`, except the inner members of a chain whose DPP
// sources are pinned (an empty volatile asm right behind their definition) in front of the chain's first member;
// tools/isa_dpp_hazard_scan.py checks the compiled code (tests/test_isa_dpp_hazards.py).
#include "kf_device.hpp"
#include "kf_args.hpp"
#include "launch.hpp"

namespace osk {

namespace {

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
    if (NOP) asm volatile("s_nop 1\nv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(m), "n"(SRC));
    else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(m), "n"(SRC));
}
#define OSD_F(acc, src, m, S) "v_fmac_f64_dpp %" #acc ", %" #src ", %" #m " row_newbcast:" #S " row_mask:0xf bank_mask:0xf\n"
// acc += sum over lanes 0..11 of the row of src (m = 1.0 in a register: a DPP instruction takes no inline constant)
__device__ __forceinline__ void rowsum12(double &acc, double src, double one)
{
    asm volatile("s_nop 1\n" OSD_F(0, 1, 2, 0) OSD_F(0, 1, 2, 1) OSD_F(0, 1, 2, 2) OSD_F(0, 1, 2, 3) OSD_F(0, 1, 2, 4) OSD_F(0, 1, 2, 5)
                 OSD_F(0, 1, 2, 6) OSD_F(0, 1, 2, 7) OSD_F(0, 1, 2, 8) OSD_F(0, 1, 2, 9) OSD_F(0, 1, 2, 10) OSD_F(0, 1, 2, 11)
                 : "+v"(acc) : "v"(src), "v"(one));
}
// acc += sum_k cf[k] * (src of lane 6 + k): row i of E applied to a column held one entry per lane (rows 6..11)
__device__ __forceinline__ void ecomb6(double &acc, double src, const double *cf)
{
    asm volatile("s_nop 1\n" OSD_F(0, 1, 2, 6) OSD_F(0, 1, 3, 7) OSD_F(0, 1, 4, 8) OSD_F(0, 1, 5, 9) OSD_F(0, 1, 6, 10) OSD_F(0, 1, 7, 11)
                 : "+v"(acc) : "v"(src), "v"(cf[0]), "v"(cf[1]), "v"(cf[2]), "v"(cf[3]), "v"(cf[4]), "v"(cf[5]));
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
                 : "+v"(acc) : "v"(src), "v"(K[0]), "v"(K[1]), "v"(K[2]), "v"(K[3]), "v"(K[4]), "v"(K[5]), "v"(K[6]), "v"(K[7]),
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

// ---- batch update as the reference writes it: S = H P H^T + R, K = P H^T S^-1, x += K y, P -= K H P ----
// The lane of state row SEL[a] also owns measurement a: row a of the Cholesky factor L (entries q <= a) lives in ITS registers
// L[0..9]; rrow: that lane's row of R.  K[0..9]: this lane's row of the 12 x 10 gain.
__device__ __forceinline__ int update_batch_row(double &xd, double *P, const float *z, const double *rrow, double *K)
{
    int status = 0;
    double L[NM], dinv[NM];
    // S[a][q], q <= a, from the lower triangle (row SEL[a] of P is this lane's): what a Cholesky reads
    for_sel([&](auto Q, auto SQ) { L[decltype(Q)::v] = P[decltype(SQ)::v] + rrow[decltype(Q)::v]; });
    for_sel([&](auto J, auto SJ) {
        constexpr int j = decltype(J)::v, sj = decltype(SJ)::v;
        // s = S[a][j] - sum_{q<j} L[a][q] L[j][q]; in lane SEL[j] this is the pivot d
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < j; q++) {
            if (q == 0) fmacb<sj, true>(t, L[q], L[q]);
            else fmacb<sj, false>(t, L[q], L[q]);
        }
        const double s = L[j] - t;
        double d = bc64<sj>(s);
        if (!(d > 0.0) || !(d < 1.0e300)) { status |= 1; d = 1.0; }
        const double di = rsqrt64(d);
        dinv[j] = di;
        L[j] = s * di;                                 // lane SEL[j]: sqrt(d); lanes of later measurements: L[a][j]
        // L[j] is the DPP source of later chain members that carry no s_nop of their own: pin its definition HERE (volatile
        // assembly statements keep their order), or hipcc sinks the multiply to just in front of its first use -- inside the next
        // column's chain, zero wait states ahead of the DPP read (found by tools/isa_dpp_hazard_scan.py after G8 failed by 9e-4)
        asm volatile("" : "+v"(L[j]));
    });
    // K[i,:] = solve(S, P[i,sel]): forward then back substitution, L[a][q] broadcast from the lane of measurement a
    double y[NM];
    for_sel([&](auto A, auto SA) {
        constexpr int a = decltype(A)::v, sa = decltype(SA)::v;
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < a; q++) {
            if (q == 0) fmacb<sa, true>(t, L[q], y[q]);
            else fmacb<sa, false>(t, L[q], y[q]);
        }
        y[a] = (P[sa] - t) * dinv[a];
    });
    // back: K[a] = (y[a] - sum_{q>a} L[q][a] K[q]) dinv[a];  L[q][a] = register a of the lane of measurement q
#define OSD_BACK(a, ...)                                            \
    {                                                               \
        double t = 0.0;                                             \
        __VA_ARGS__                                                 \
        K[a] = (y[a] - t) * dinv[a];                                \
    }
    OSD_BACK(9, )
    OSD_BACK(8, fmacb<11, true>(t, L[8], K[9]);)
    OSD_BACK(7, fmacb<10, true>(t, L[7], K[8]); fmacb<11, false>(t, L[7], K[9]);)
    OSD_BACK(6, fmacb<9, true>(t, L[6], K[7]); fmacb<10, false>(t, L[6], K[8]); fmacb<11, false>(t, L[6], K[9]);)
    OSD_BACK(5, fmacb<8, true>(t, L[5], K[6]); fmacb<9, false>(t, L[5], K[7]); fmacb<10, false>(t, L[5], K[8]); fmacb<11, false>(t, L[5], K[9]);)
    OSD_BACK(4, fmacb<7, true>(t, L[4], K[5]); fmacb<8, false>(t, L[4], K[6]); fmacb<9, false>(t, L[4], K[7]); fmacb<10, false>(t, L[4], K[8]); fmacb<11, false>(t, L[4], K[9]);)
    OSD_BACK(3, fmacb<6, true>(t, L[3], K[4]); fmacb<7, false>(t, L[3], K[5]); fmacb<8, false>(t, L[3], K[6]); fmacb<9, false>(t, L[3], K[7]); fmacb<10, false>(t, L[3], K[8]);
             fmacb<11, false>(t, L[3], K[9]);)
    OSD_BACK(2, fmacb<5, true>(t, L[2], K[3]); fmacb<6, false>(t, L[2], K[4]); fmacb<7, false>(t, L[2], K[5]); fmacb<8, false>(t, L[2], K[6]); fmacb<9, false>(t, L[2], K[7]);
             fmacb<10, false>(t, L[2], K[8]); fmacb<11, false>(t, L[2], K[9]);)
    OSD_BACK(1, fmacb<2, true>(t, L[1], K[2]); fmacb<5, false>(t, L[1], K[3]); fmacb<6, false>(t, L[1], K[4]); fmacb<7, false>(t, L[1], K[5]); fmacb<8, false>(t, L[1], K[6]);
             fmacb<9, false>(t, L[1], K[7]); fmacb<10, false>(t, L[1], K[8]); fmacb<11, false>(t, L[1], K[9]);)
    OSD_BACK(0, fmacb<1, true>(t, L[0], K[1]); fmacb<2, false>(t, L[0], K[2]); fmacb<5, false>(t, L[0], K[3]); fmacb<6, false>(t, L[0], K[4]); fmacb<7, false>(t, L[0], K[5]);
             fmacb<8, false>(t, L[0], K[6]); fmacb<9, false>(t, L[0], K[7]); fmacb<10, false>(t, L[0], K[8]); fmacb<11, false>(t, L[0], K[9]);)
#undef OSD_BACK
    // x += K (z - H x)
    double dx = 0.0;
    for_sel([&](auto A, auto SA) {
        constexpr int a = decltype(A)::v, sa = decltype(SA)::v;
        dx = fma(K[a], (double)z[a] - bc64<sa>(xd), dx);
    });
    xd += dx;
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

}  // namespace

// AUX: P_trace / K_gain / p_rot outputs where the pointers are set; FEAT: the normalised 60-feature row of the two-kernel
// fused path [x_post | accel | f | p_world | dp | imu] (lane r writes the r-th element of each block).
template <bool SEQ, bool AUX, bool FEAT>
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
    const int am = r < 3 ? r : (r >= 5 && r < 12) ? r - 2 : -1;

    // Q and R as float64 in LDS: a lane re-reads ITS row of Q at the end of every predict and its row of R at the top of
    // every batch update (six / five 16-byte reads) instead of holding 24 + 20 registers across the whole step
    __shared__ __attribute__((aligned(16))) double qs[NS * NS], rs[(NM + 2) * NM];
    for (int i = threadIdx.x; i < NS * NS; i += blockDim.x) qs[i] = (double)qr[i];
    for (int i = threadIdx.x; i < (NM + 2) * NM; i += blockDim.x) rs[i] = i < NM * NM ? (double)qr[144 + i] : 0.0;     // rows 10, 11: zeros for the lanes that own no measurement
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
    {
        rsrc_t rb = make_rsrc(a.body_ref, 12 * rowB);
#pragma unroll
        for (int i = 0; i < 3; i++) bref[i] = buf_load(rb, voff, i * rowB);
    }
    for (int t = 0; t < a.T; t++) {
        // (an offset the optimiser cannot see through: the LDS reads of Q / R stay inside the step instead of being hoisted
        // into 44 loop-invariant registers)
        int oz = 0;
        asm volatile("" : "+v"(oz));
        const double *qrow = qs + qoff + oz, *rrow_b = rs + roff + oz, *rdiag = rs + oz;
        // ---- the prior state, replicated per lane ----
        float x[NS];
        x[0] = bc32<0>(xr); x[1] = bc32<1>(xr); x[2] = bc32<2>(xr); x[3] = bc32<3>(xr); x[4] = bc32<4>(xr); x[5] = bc32<5>(xr);
        x[6] = bc32<6>(xr); x[7] = bc32<7>(xr); x[8] = bc32<8>(xr); x[9] = bc32<9>(xr); x[10] = bc32<10>(xr); x[11] = bc32<11>(xr);
        // nine sincos per step (prior attitude | IMU attitude | body_ref attitude): lane r < 9 evaluates one angle
        float sv, cv;
        {
            const float ang = r < 3 ? xr : r == 3 ? in.imu[0] : r == 4 ? in.imu[1] : r == 5 ? in.imu[2] : r == 6 ? bref[0] : r == 7 ? bref[1] : bref[2];
            sincos_f32(ang, &sv, &cv);
        }
        const Rot rot = rotation_sc(bc32<0>(sv), bc32<0>(cv), bc32<1>(sv), bc32<1>(cv), bc32<2>(sv), bc32<2>(cv));
        const Rot rimu = rotation_sc(bc32<3>(sv), bc32<3>(cv), bc32<4>(sv), bc32<4>(cv), bc32<5>(sv), bc32<5>(cv));
        const Rot rbr = rotation_sc(bc32<6>(sv), bc32<6>(cv), bc32<7>(sv), bc32<7>(cv), bc32<8>(sv), bc32<8>(cv));
        float z[NM], pw[12];
        measurement_r(in, rimu, z);
        // ---- covariance predict (kalman_filter.py:153-158) ----
        {
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
        // ---- next_state (replicated), keep this lane's component ----
        status |= dynamics(x, rot, in.p, in.f, pw, k);
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
            rsrc_t rb = make_rsrc(a.body_ref + (size_t)tn * 12 * B, 12 * rowB);
#pragma unroll
            for (int i = 0; i < 3; i++) bref[i] = buf_load(rb, voff, i * rowB);
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
            if (AUX && a.kgain_out) {
                // K = P+ H^T R^-1 for a diagonal R: trace = sum_a P+[a][SEL[a]] / R[a][a], entry a from lane a
                double pv = 0.0, rv = 1.0;
                for_sel([&](auto A, auto SA) { if (r == decltype(A)::v) { pv = P[decltype(SA)::v]; rv = rrow[decltype(A)::v]; } });
                kgain = (float)group_sum12(pv * rcp64(rv), one);
            }
        } else {
            double K[NM], rrow[NM];
#pragma unroll
            for (int q = 0; q < NM; q++) rrow[q] = rrow_b[q];
            status |= update_batch_row(xd, P, z, rrow, K);
            if (AUX && a.kgain_out) {
                double kv = 0.0;                       // np.trace of the 12 x 10 K (kalman_filter.py:174): K[a][a], a < 10
#pragma unroll
                for (int q = 0; q < NM; q++) kv = (r == q) ? K[q] : kv;
                kgain = (float)group_sum12(kv, one);
            }
        }
        xr = (float)xd;
        if (!(xr * 0.f == 0.f)) status |= 2;
        if (live && r < 12) a.x_out[((size_t)t * 12 + r) * B + b] = xr;
        if (FEAT && live && r < 12)
            __builtin_nontemporal_store((xr - a.minmax[r]) / (a.minmax[60 + r] - a.minmax[r]), a.feat_out + ((size_t)t * a.feat_I + r) * B + b);
        if (AUX) {
            if (a.ptrace_out) {
                double dg = P[0];
#pragma unroll
                for (int i = 1; i < NS; i++) dg = (r == i) ? P[i] : dg;
                const float tr = (float)group_sum12(dg, one);
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
        a.x[(size_t)r * B + b] = xr;
#pragma unroll
        for (int j = 0; j < NS; j++) a.P[(size_t)(r * NS + j) * B + b] = (float)P[j];
        if (r == 0) a.status[b] = status;
    }
}

// host side: picks the instantiation (kf_kernels.hip: os_kf_run_impl)
hipError_t launch_kf_dense_rows(const KfRunArgs &a, const float *qr, bool seq, bool feat, bool aux, hipStream_t s)
{
    dim3 grid((a.B + 15) / 16), block(256);
#define OSD_GO(SEQ_, AUX_, FEAT_) hipLaunchKernelGGL((kf_dense_rows_kernel<SEQ_, AUX_, FEAT_>), grid, block, 0, s, a, qr)
    if (seq) {
        if (feat) OSD_GO(true, false, true);
        else if (aux) OSD_GO(true, true, false);
        else OSD_GO(true, false, false);
    } else {
        if (feat) OSD_GO(false, false, true);
        else if (aux) OSD_GO(false, true, false);
        else OSD_GO(false, false, false);
    }
#undef OSD_GO
    return hipGetLastError();
}

}  // namespace osk
