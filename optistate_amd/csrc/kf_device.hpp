// kf_device.hpp -- per-lane Kalman filter arithmetic for gfx950 (one trajectory per lane).
//
// All state (x: 12, P: 144 floats) lives in VGPRs of the lane that owns the trajectory; every loop is
// fully unrolled with compile-time indices so nothing is spilled to scratch.  The math is derived from
// the reference's equations (not its code):
//   rotation          kalman_filter/kalman_filter.py:184-193
//   odometry + z      kalman_filter/kalman_filter.py:79-117
//   dynamics          misc/force_controller.py:269-291 (with the int64 truncation of A[0:3,6:9], :248-251,:271)
//   covariance        kalman_filter/kalman_filter.py:124-135 (F_d = I + dt F has two non-trivial 3x3 blocks)
//   update            kalman_filter/kalman_filter.py:164-174 (H is a row selection)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace osk {

constexpr int NS = 12;
constexpr int NM = 10;
// H selects these state rows (kalman_filter/kalman_filter.py:15-24)
__device__ constexpr int SEL[NM] = {0, 1, 2, 5, 6, 7, 8, 9, 10, 11};

struct KfConst {
    float dt, inv_mass, gz;
    float inv_inertia[3];
    float Q[NS * NS];
    float R[NM * NM];
};

struct Rot { float m[9]; };

// ---- buffer addressing: wave-uniform descriptor + SGPR row offset + one constant per-lane VGPR offset ----
// Every stream is [rows][B] with the trajectory index fastest, so element (row, b) is
//   base + row*B*4 (uniform -> SGPR soffset)  +  b*4 (per lane -> the same voffset VGPR for every access).
// With flat 64-bit addresses hipcc materialises one address pair per access (hundreds of VGPRs in a fully
// unrolled 144-element state load); buffer_load/store needs none (cdna_hip_programming.md T8).
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void *base, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(rsrc_t r, uint32_t voff, uint32_t soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ uint32_t buf_load_u32(rsrc_t r, uint32_t voff, uint32_t soff)
{
    return __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
}
__device__ __forceinline__ void buf_store(rsrc_t r, uint32_t voff, uint32_t soff, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, voff, soff, 0);
}
// Non-temporal forms (aux bit 1 = nt) for the read-once input streams and write-once outputs: 1.3 GB per pass flows
// through the 4 MiB L2s, and with the default policy it evicts everything else there (measured: the fused kernel's
// few spilled registers were being re-fetched from HBM, +40 % FETCH_SIZE).
__device__ __forceinline__ float buf_load_nt(rsrc_t r, uint32_t voff, uint32_t soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 2));
}
__device__ __forceinline__ uint32_t buf_load_u32_nt(rsrc_t r, uint32_t voff, uint32_t soff)
{
    return __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 2);
}
__device__ __forceinline__ void buf_store_nt(rsrc_t r, uint32_t voff, uint32_t soff, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, voff, soff, 2);
}

// Branch-free sincos for the attitude angles (Cody-Waite reduction by pi/2 + Cephes-style minimax polynomials on
// [-pi/4, pi/4]; |error| <~ 1.5e-7 for |x| < 1e3).  ocml's sincosf carries a Payne-Hanek slow path per call that costs
// code size and registers in a kernel that is otherwise straight-line.
__device__ __forceinline__ void sincos_f32(float x, float *s, float *c)
{
    const float k = rintf(x * 0.636619772367581343f);
    float r = fmaf(-k, 1.57079637050628662109375f, x);
    r = fmaf(-k, -4.37113900018624283e-8f, r);
    const float z = r * r;
    const float sp = fmaf(r * z, fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
    const float cp = fmaf(z * z, fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f),
                          fmaf(-0.5f, z, 1.0f));
    const int q = (int)k;
    const float ss = (q & 1) ? cp : sp, cc = (q & 1) ? sp : cp;
    *s = (q & 2) ? -ss : ss;
    *c = ((q + 1) & 2) ? -cc : cc;
}

// the k = 0 branch of sincos_f32 (|x| < pi/4): identical polynomials in identical order
__device__ __forceinline__ void sincos_small_f32(float x, float *s, float *c)
{
    const float z = x * x;
    *s = fmaf(x * z, fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), x);
    *c = fmaf(z * z, fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f), fmaf(-0.5f, z, 1.0f));
}

// R = Rz(thz) * Ry(thy) * Rx(thx), closed form of the product at kalman_filter/kalman_filter.py:187-191
__device__ __forceinline__ Rot rotation_sc(float sx, float cx, float sy, float cy, float sz, float cz)
{
    Rot r;
    r.m[0] = cz * cy; r.m[1] = cz * sy * sx - sz * cx; r.m[2] = cz * sy * cx + sz * sx;
    r.m[3] = sz * cy; r.m[4] = sz * sy * sx + cz * cx; r.m[5] = sz * sy * cx - cz * sx;
    r.m[6] = -sy;     r.m[7] = cy * sx;                r.m[8] = cy * cx;
    return r;
}

__device__ __forceinline__ Rot rotation(float thx, float thy, float thz)
{
    float sx, cx, sy, cy, sz, cz;
    // A walking robot's attitude angles are small: when every lane of the wave has all three below pi/4 (wave-uniform test)
    // the reduction's k is 0, r == x exactly and no quadrant select fires, so the polynomials alone give the SAME bits as
    // sincos_f32 -- 10 instructions per angle instead of 24, and a lane's result never depends on its neighbours' data.
    const bool big = !(fabsf(thx) < 0.785f && fabsf(thy) < 0.785f && fabsf(thz) < 0.785f);       // NaN counts as big
    if (__builtin_amdgcn_ballot_w64(big) == 0ull) {
        sincos_small_f32(thx, &sx, &cx);
        sincos_small_f32(thy, &sy, &cy);
        sincos_small_f32(thz, &sz, &cz);
    } else {
        sincos_f32(thx, &sx, &cx);
        sincos_f32(thy, &sy, &cy);
        sincos_f32(thz, &sz, &cz);
    }
    Rot r;
    r.m[0] = cz * cy; r.m[1] = cz * sy * sx - sz * cx; r.m[2] = cz * sy * cx + sz * sx;
    r.m[3] = sz * cy; r.m[4] = sz * sy * sx + cz * cx; r.m[5] = sz * sy * cx - cz * sx;
    r.m[6] = -sy;     r.m[7] = cy * sx;                r.m[8] = cy * cx;
    return r;
}

// float64 sincos for the truncation predicate: fdlibm-style two-term reduction by pi/2 (33 + 53 bits of pi/2) and
// the classic degree-13/14 kernels (published constants).  < 1 ulp for |x| < ~1e5, and in particular cos(e) == 1.0
// exactly for |e| < 1.05e-8 and sin/cos(k*pi/2 as a double) land where a correctly rounded libm puts them.
__device__ __forceinline__ void sincos_f64(double x, double *s, double *c)
{
    const double k = rint(x * 6.36619772367581382433e-01);
    double r = fma(-k, 1.57079632673412561417e+00, x);
    r = fma(-k, 6.07710050650619224932e-11, r);
    const double z = r * r;
    const double sp = r + r * z * (-1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
                      z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)))));
    const double w = z * z;
    const double rr = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * 2.48015872894767294178e-05)) +
                      (w * w) * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11));
    const double hz = 0.5 * z, w1 = 1.0 - hz;
    const double cp = w1 + (((1.0 - w1) - hz) + z * rr);
    const long long q = (long long)k;
    const double ss = (q & 1) ? cp : sp, cc = (q & 1) ? sp : cp;
    *s = (q & 2) ? -ss : ss;
    *c = ((q + 1) & 2) ? -cc : cc;
}

// The reference's module-global `A` is int64, so A[0:3,6:9] = R^T truncates toward zero
// (misc/force_controller.py:248-251,271): an entry survives only when |R_ji| >= 1 in float64.
// The decision is made in float64 from the float32 angles promoted to double: cosf() rounds to
// 1.0f for |th| up to ~3e-4 where the float64 cosine is still < 1, and that would integrate omega into theta
// where the reference adds 0.  The float64 path only runs for lanes whose float32 entry is within a few ulp of 1.
// Returns status bit 4 (OS_STATUS_TRUNC_EDGE = 16) when the decision sits on a knife edge: some |R entry| >= 1 - 2^-40 in
// float64 at an attitude other than the reference's own exact start theta = 0 (settings.py:25).  There the reference's
// int64 store picks 0 or +-1 from the last bits of ITS float64 state (1 - cos(d) < 2^-40 for |d| < 1.3e-6, the size of a
// float32 filter's state error), so a float32 filter may integrate dt*omega where the reference adds 0 or the reverse
// (gimbal lock, pitch = pi/2 with roll = yaw: R[1][1] = cos(yaw - roll) stays within one rounding of 1 for several steps).
__device__ __forceinline__ int trunc_block_f64(float thx, float thy, float thz, float *A /* 9, A[i][j] = trunc(R[j][i]) */)
{
    double sx, cx, sy, cy, sz, cz;
    sincos_f64((double)thx, &sx, &cx);
    sincos_f64((double)thy, &sy, &cy);
    sincos_f64((double)thz, &sz, &cz);
    // M = Ry*Rx, R = Rz*M with the zero terms dropped (adding exact zeros changes nothing)
    double m00 = cy, m01 = sy * sx, m02 = sy * cx;
    double m10 = 0.0, m11 = cx, m12 = -sx;
    double m20 = -sy, m21 = cy * sx, m22 = cy * cx;
    double r[9];
    r[0] = cz * m00 + (-sz) * m10; r[1] = cz * m01 + (-sz) * m11; r[2] = cz * m02 + (-sz) * m12;
    r[3] = sz * m00 + cz * m10;    r[4] = sz * m01 + cz * m11;    r[5] = sz * m02 + cz * m12;
    r[6] = m20;                    r[7] = m21;                    r[8] = m22;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) A[3 * i + j] = (float)trunc(r[3 * j + i]);
    double rmax = 0.0;
#pragma unroll
    for (int i = 0; i < 9; i++) rmax = fmax(rmax, fabs(r[i]));
    const bool exact_start = thx == 0.f && thy == 0.f && thz == 0.f;
    return (rmax >= 1.0 - 0x1p-40 && !exact_start) ? 16 : 0;
}

struct StepIn {
    float p[12], f[12], dp[12], imu[6];
    uint32_t contact;   // 4 packed bytes
};

// get_odom + set_measurements (kalman_filter/kalman_filter.py:79-117).  p is the body-frame foot position.
// r = R(imu angles) (supplied by the caller: the 16-lanes-per-trajectory kernel shares its sincos across the lanes).
__device__ __forceinline__ void measurement_r(const StepIn &in, const Rot &r, float *z /*10*/)
{
    // per leg: stance (byte == 1) entries feed vx, vy, pz, swing (byte == 0) entries feed vz.  SELECTS, not 0/1 weights: the
    // reference's `if contact_cur[i] == 1 / == 0` (:83-90) never touches the other entries, so a NaN / Inf in a swing leg's
    // dp_x must not reach z (0 * NaN = NaN would poison the whole trajectory)
    float sum_c = 0.f, vx = 0.f, vy = 0.f, vz = 0.f, pz = 0.f;
#pragma unroll
    for (int l = 0; l < 4; l++) {
        const uint32_t cb = (in.contact >> (8 * l)) & 0xffu;
        const bool st = cb == 1u, sw = cb == 0u;
        sum_c += (float)cb;
        vx += st ? in.dp[3 * l] : 0.f;
        vy += st ? in.dp[3 * l + 1] : 0.f;
        pz += st ? in.p[3 * l + 2] : 0.f;
        vz += sw ? in.dp[3 * l + 2] : 0.f;
    }
    float inv = (sum_c != 0.f) ? (1.0f / sum_c) : 0.f;   // no stance leg -> odom = 0 (:97-98)
    float bx = -vx * inv, by = -vy * inv, bz = -vz * inv;
    float bpz = -pz * inv;
    z[0] = in.imu[0]; z[1] = in.imu[1]; z[2] = in.imu[2];
    z[3] = bpz;
    z[4] = in.imu[3]; z[5] = in.imu[4]; z[6] = in.imu[5];
    z[7] = r.m[0] * bx + r.m[1] * by + r.m[2] * bz;
    z[8] = r.m[3] * bx + r.m[4] * by + r.m[5] * bz;
    z[9] = r.m[6] * bx + r.m[7] * by + r.m[8] * bz;
}

__device__ __forceinline__ void measurement(const StepIn &in, float *z /*10*/)
{
    const Rot r = rotation(in.imu[0], in.imu[1], in.imu[2]);
    measurement_r(in, r, z);
}

// next_state (misc/force_controller.py:269-291): x <- (I + A dt) x + B dt f + dt g, foot positions rotated
// to the world frame (returned in pw: the reference mutates the caller's p, :274-277).
// I_hat^-1 = R diag(1/I) R^T (R orthogonal), tau = sum_j pw_j x f_j.
__device__ __forceinline__ int dynamics(float *x, const Rot &r, const float *p, const float *f, float *pw,
                                        const KfConst &k)
{
    int edge = 0;
    float tau[3] = {0.f, 0.f, 0.f}, fs[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int l = 0; l < 4; l++) {
        float a = p[3 * l], b = p[3 * l + 1], c = p[3 * l + 2];
        float wx = r.m[0] * a + r.m[1] * b + r.m[2] * c;
        float wy = r.m[3] * a + r.m[4] * b + r.m[5] * c;
        float wz = r.m[6] * a + r.m[7] * b + r.m[8] * c;
        pw[3 * l] = wx; pw[3 * l + 1] = wy; pw[3 * l + 2] = wz;
        float fx = f[3 * l], fy = f[3 * l + 1], fz = f[3 * l + 2];
        tau[0] += wy * fz - wz * fy;
        tau[1] += wz * fx - wx * fz;
        tau[2] += wx * fy - wy * fx;
        fs[0] += fx; fs[1] += fy; fs[2] += fz;
    }
    // body-frame torque, scaled by 1/I, back to world
    float tb0 = (r.m[0] * tau[0] + r.m[3] * tau[1] + r.m[6] * tau[2]) * k.inv_inertia[0];
    float tb1 = (r.m[1] * tau[0] + r.m[4] * tau[1] + r.m[7] * tau[2]) * k.inv_inertia[1];
    float tb2 = (r.m[2] * tau[0] + r.m[5] * tau[1] + r.m[8] * tau[2]) * k.inv_inertia[2];
    float aw0 = r.m[0] * tb0 + r.m[1] * tb1 + r.m[2] * tb2;
    float aw1 = r.m[3] * tb0 + r.m[4] * tb1 + r.m[5] * tb2;
    float aw2 = r.m[6] * tb0 + r.m[7] * tb1 + r.m[8] * tb2;

    // theta: A[0:3,6:9] = trunc(R^T) -- zero unless an entry of R reaches +-1 in float64
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 9; i++) amax = fmaxf(amax, fabsf(r.m[i]));
    float w0 = x[6], w1 = x[7], w2 = x[8];
    if (amax >= 0.9999995f) {
        float A[9];
        edge = trunc_block_f64(x[0], x[1], x[2], A);
        x[0] += k.dt * (A[0] * w0 + A[1] * w1 + A[2] * w2);
        x[1] += k.dt * (A[3] * w0 + A[4] * w1 + A[5] * w2);
        x[2] += k.dt * (A[6] * w0 + A[7] * w1 + A[8] * w2);
    }
    // position integrates the PRIOR velocity
    x[3] += k.dt * x[9]; x[4] += k.dt * x[10]; x[5] += k.dt * x[11];
    x[6] = w0 + k.dt * aw0; x[7] = w1 + k.dt * aw1; x[8] = w2 + k.dt * aw2;
    x[9] += k.dt * (fs[0] * k.inv_mass);
    x[10] += k.dt * (fs[1] * k.inv_mass);
    x[11] += k.dt * (fs[2] * k.inv_mass) + k.dt * k.gz;
    return edge;
}

// P <- F_d P F_d^T + Q with F_d = I + dt F, F[0:3,6:9] = R^T, F[3:6,9:12] = I
// (kalman_filter/kalman_filter.py:124-128,135).  ~300 FMA instead of two dense 12^3 products.
template <bool QDIAG, typename PT = float>
__device__ __forceinline__ void cov_predict(PT *P, const Rot &r, const KfConst &k)
{
    PT g[9];   // g[i][kk] = dt * R^T[i][kk] = dt * R[kk][i]
    const PT dt = (PT)k.dt;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int kk = 0; kk < 3; kk++) g[3 * i + kk] = dt * (PT)r.m[3 * kk + i];
    // rows: M = F_d P
#pragma unroll
    for (int j = 0; j < NS; j++) {
        PT a6 = P[6 * NS + j], a7 = P[7 * NS + j], a8 = P[8 * NS + j];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            P[i * NS + j] += g[3 * i] * a6 + g[3 * i + 1] * a7 + g[3 * i + 2] * a8;
            P[(3 + i) * NS + j] += dt * P[(9 + i) * NS + j];
        }
    }
    // columns: M F_d^T
#pragma unroll
    for (int i = 0; i < NS; i++) {
        PT a6 = P[i * NS + 6], a7 = P[i * NS + 7], a8 = P[i * NS + 8];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            P[i * NS + j] += g[3 * j] * a6 + g[3 * j + 1] * a7 + g[3 * j + 2] * a8;
            P[i * NS + 3 + j] += dt * P[i * NS + 9 + j];
        }
    }
    if (QDIAG) {
        // every Q the reference uses is diagonal (settings.py:28, np.diag at Kalman_to_Training.py:87): 12 scalars
        // instead of 144 keeps the noise terms in SGPRs without spilling
#pragma unroll
        for (int i = 0; i < NS; i++) P[i * NS + i] += (PT)k.Q[i * NS + i];
    } else {
#pragma unroll
        for (int i = 0; i < NS * NS; i++) P[i] += (PT)k.Q[i];
    }
}

// predict_mpc covariance (kalman_filter/kalman_filter.py:153-158): F_d = element-wise exp(dt F), i.e.
// ones everywhere except exp(dt R^T_ij) in [0:3,6:9] and e^dt on the diagonal of [3:6,9:12].
// F_d = 1 1^T + E  ->  F_d P F_d^T evaluated through column/row sums plus the sparse E.
// PT = double in the batched kernel: the all-ones F_d adds a common-mode term ~sum(P) (1e2..1e4) to every entry, the
// following update has to cancel it again against R ~ 1e-4, and float32 keeps only ~1e-2 of the state through that
// (measured); float64 covariance arithmetic restores the 1e-4 bar for this variant.
template <typename PT>
__device__ __forceinline__ void cov_predict_dense(PT *P, const Rot &rb, const KfConst &k)
{
    PT e[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int kk = 0; kk < 3; kk++) e[3 * i + kk] = (PT)expm1((double)k.dt * (double)rb.m[3 * kk + i]);
    PT ed = (PT)expm1((double)k.dt);
    PT M[NS * NS];
    // M = F_d P = 1 (1^T P) + E P
#pragma unroll
    for (int j = 0; j < NS; j++) {
        PT cs = 0;
#pragma unroll
        for (int i = 0; i < NS; i++) cs += P[i * NS + j];
#pragma unroll
        for (int i = 0; i < NS; i++) M[i * NS + j] = cs;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            M[i * NS + j] += e[3 * i] * P[6 * NS + j] + e[3 * i + 1] * P[7 * NS + j] + e[3 * i + 2] * P[8 * NS + j];
            M[(3 + i) * NS + j] += ed * P[(9 + i) * NS + j];
        }
    }
    // P = M F_d^T = (M 1) 1^T + M E^T
#pragma unroll
    for (int i = 0; i < NS; i++) {
        PT rs = 0;
#pragma unroll
        for (int j = 0; j < NS; j++) rs += M[i * NS + j];
#pragma unroll
        for (int j = 0; j < NS; j++) P[i * NS + j] = rs + k.Q[i * NS + j];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            P[i * NS + j] += e[3 * j] * M[i * NS + 6] + e[3 * j + 1] * M[i * NS + 7] + e[3 * j + 2] * M[i * NS + 8];
            P[i * NS + 3 + j] += ed * M[i * NS + 9 + j];
        }
    }
}

// (The one-trajectory-per-lane batch update lived here until round 5: a float32 Cholesky of the symmetrised S.  It lost the filter
// on ill-conditioned runs -- kf_dense_rows.hpp update_batch_row has the account -- and the batch form now runs on the float64 row
// layout for every covariance model.)

// Sequential (one measurement at a time) form of the same update; exact-arithmetic identical to the batch
// form when R is diagonal, which every Q/R set of the reference is (settings.py:30,
// data_collection/data_conversion_Kalman_to_Training.py:103-105 np.diag).  No factorisation, 24 temporaries.
template <typename PT = float>
__device__ __forceinline__ int update_sequential(float *x, PT *P, const float *z, const KfConst &k)
{
    int status = 0;
#pragma unroll
    for (int a = 0; a < NM; a++) {
        const int sa = SEL[a];
        PT s = P[sa * NS + sa] + (PT)k.R[a * NM + a];
        if (!(s > 0) || !(s < (PT)3.0e38)) { status |= 1; s = 1; }
        PT inv = (PT)1 / s;
        PT innov = (PT)z[a] - (PT)x[sa];
        PT kc[NS], row[NS];
#pragma unroll
        for (int i = 0; i < NS; i++) { kc[i] = P[i * NS + sa] * inv; row[i] = P[sa * NS + i]; }
#pragma unroll
        for (int i = 0; i < NS; i++) {
            x[i] = (float)((PT)x[i] + kc[i] * innov);
#pragma unroll
            for (int j = 0; j < NS; j++) P[i * NS + j] -= kc[i] * row[j];
        }
        // keep hipcc's scheduler from interleaving successive measurement updates: they are a dependent chain,
        // and letting them overlap only stretches live ranges past the 256 architectural VGPRs
        __builtin_amdgcn_sched_barrier(0);
    }
    return status;
}

// ---------------------------------------------------------------------------------------------------------------
// Scalar upper triangle (78 floats, row-major) -- the storage the fused kernel keeps: there the paired layout below costs
// more in register-pair alignment than it saves (36 -> 139 spilled VGPRs, 3.09 -> 4.08 ms), so fused_kf_gru_kernel stays
// on this form and lets hipcc's SLP vectoriser pack what it can.
constexpr int NT = NS * (NS + 1) / 2;   // 78
__host__ __device__ constexpr int uidx(int i, int j) { return i * NS - i * (i - 1) / 2 + (j - i); }   // i <= j
#define OSK_TRI(U, i, j) ((i) <= (j) ? (U)[uidx((i), (j))] : (U)[uidx((j), (i))])

// P <- F_d P F_d^T + Q, F_d = I + G with G[0:3,6:9] = dt R^T, G[3:6,9:12] = dt I (kalman_filter.py:124-135):
//   P' = P + M + M^T + M G^T,  M = G P (rows 0..5 only).
template <bool QDIAG>
__device__ __forceinline__ void cov_predict_tri(float *U, const Rot &r, const KfConst &k)
{
    float g[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int kk = 0; kk < 3; kk++) g[3 * i + kk] = k.dt * r.m[3 * kk + i];
    float M[6 * NS];
#pragma unroll
    for (int j = 0; j < NS; j++) {
        const float a6 = OSK_TRI(U, 6, j), a7 = OSK_TRI(U, 7, j), a8 = OSK_TRI(U, 8, j);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            M[i * NS + j] = g[3 * i] * a6 + g[3 * i + 1] * a7 + g[3 * i + 2] * a8;
            M[(3 + i) * NS + j] = k.dt * OSK_TRI(U, 9 + i, j);
        }
    }
#pragma unroll
    for (int i = 0; i < 6; i++) {
#pragma unroll
        for (int j = i; j < NS; j++) {
            float v = U[uidx(i, j)] + M[i * NS + j];
            if (j < 6) {
                v += M[j * NS + i];
                // (M G^T)[i][j] = sum_k M[i][k] G[j][k]
                if (j < 3) v += M[i * NS + 6] * g[3 * j] + M[i * NS + 7] * g[3 * j + 1] + M[i * NS + 8] * g[3 * j + 2];
                else v += k.dt * M[i * NS + 9 + (j - 3)];
            }
            U[uidx(i, j)] = v;
        }
    }
    if (QDIAG) {
#pragma unroll
        for (int i = 0; i < NS; i++) U[uidx(i, i)] += k.Q[i * NS + i];
    } else {
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int j = i; j < NS; j++) U[uidx(i, j)] += 0.5f * (k.Q[i * NS + j] + k.Q[j * NS + i]);
    }
}

// Sequential scalar updates on the packed upper triangle (diagonal R).
__device__ __forceinline__ int update_sequential_tri(float *x, float *U, const float *z, const KfConst &k)
{
    int status = 0;
#pragma unroll
    for (int a = 0; a < NM; a++) {
        const int sa = SEL[a];
        float s = U[uidx(sa, sa)] + k.R[a * NM + a];
        if (!(s > 0.f) || !(s < 3.0e38f)) { status |= 1; s = 1.0f; }
        float inv = __builtin_amdgcn_rcpf(s);
        inv = inv * (2.0f - s * inv);          // one Newton step: <= 1 ulp, 3 instructions instead of the ~10 of a division
        const float innov = z[a] - x[sa];
        float c[NS], kc[NS];
#pragma unroll
        for (int i = 0; i < NS; i++) { c[i] = OSK_TRI(U, i, sa); kc[i] = c[i] * inv; }
#pragma unroll
        for (int i = 0; i < NS; i++) {
            x[i] += kc[i] * innov;
#pragma unroll
            for (int j = i; j < NS; j++) U[uidx(i, j)] -= kc[i] * c[j];
        }
    }
    return status;
}


// ---------------------------------------------------------------------------------------------------------------
// Symmetric-storage fast path.  P is symmetric in exact arithmetic (P0 = Q diagonal; F P F^T + Q and the scalar
// updates P - c c^T / s preserve symmetry), so only the upper triangle is kept.  The reference never symmetrises its
// float64 P (kalman_filter.py:172), whose asymmetry stays at rounding level (~1e-17 relative); the difference is far
// below the fp32 rounding of either form.
//
// Storage: float2 PAIRS of adjacent columns, U[pidx(i, jp)] = (P[i][2jp], P[i][2jp+1]) for jp >= i/2 -- 42 pairs.  An
// even row starts at its diagonal; an odd row's first pair also carries P[i][i-1], a duplicate of P[i-1][i] that is
// updated with the same formulas (and never read as the source of truth).  With one wave per SIMD a VALU instruction
// costs an issue slot whether it is packed or not (profiles/r01j_valu_rates.md), and on this layout the rank-1 updates
// and the covariance products are v_pk_fma_f32 on aligned pairs with the scalar factor broadcast by op_sel: 42
// instructions per measurement update instead of 78 + the shuffles hipcc's own SLP packing needed.
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int NU = 42;   // pairs
__host__ __device__ constexpr int prow0(int i)
{
    int o = 0;
    for (int r = 0; r < i; r++) o += 6 - r / 2;
    return o;
}
__host__ __device__ constexpr int pidx(int i, int jp) { return prow0(i) + (jp - i / 2); }   // jp >= i/2
#define OSK_SYM(U, i, j) ((i) <= (j) ? (U)[pidx((i), (j) / 2)][(j) & 1] : (U)[pidx((j), (i) / 2)][(i) & 1])
__device__ __forceinline__ f2 splat2(float v) { return (f2){v, v}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }


// (a[IA], b[IB]) in ONE instruction: v_pk_mov_b32 picks a half of each source pair (op_sel bit set = upper half).  hipcc
// builds such a pair with two v_mov_b32 more often than not, and the ~50 pairs a filter step assembles across the diagonal of
// the triangle were ~100 of its instructions.  Inline asm is invisible to hipcc's hazard recogniser; the only VALU hazard
// that could apply here (a transcendental result read by the next instruction) cannot: the sources are covariance entries,
// results of v_pk_fma_f32.
template <int IA, int IB>
__device__ __forceinline__ f2 pkmov(f2 a, f2 b)
{
    f2 d;
    if constexpr (IA == 0 && IB == 0) asm("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0]" : "=v"(d) : "v"(a), "v"(b));
    else if constexpr (IA == 1 && IB == 0) asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(a), "v"(b));
    else if constexpr (IA == 0 && IB == 1) asm("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    else asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f2 pkmov_rt(int ia, int ib, f2 a, f2 b)      // ia / ib: compile-time constants once the loops are unrolled
{
    if (ia == 0 && ib == 0) return pkmov<0, 0>(a, b);
    if (ia == 1 && ib == 0) return pkmov<1, 0>(a, b);
    if (ia == 0 && ib == 1) return pkmov<0, 1>(a, b);
    return pkmov<1, 1>(a, b);
}
// (P[k][2jp], P[k][2jp+1]) for any row k: the stored pair where it exists, otherwise assembled through symmetry from rows
// 2jp and 2jp+1 (k, jp: compile-time constants after unrolling)
__device__ __forceinline__ f2 rowpair(const f2 *U, int k, int jp)
{
    if (jp >= k / 2) return U[pidx(k, jp)];
    return pkmov_rt(k & 1, k & 1, U[pidx(2 * jp, k / 2)], U[pidx(2 * jp + 1, k / 2)]);
}

// P <- F_d P F_d^T + Q, F_d = I + Gs with Gs = [[dt R^T, 0], [0, dt I]] (kalman_filter.py:124-135), in BLOCK form: with
// a = rows 0-5 (theta, r) and b = rows 6-11 (omega, v),
//   P_ab' = P_ab + Gs P_bb,      P_aa' = P_aa + Gs P_ba + P_ab' Gs^T      (P_bb unchanged),
// because Gs P_ba + P_ab Gs^T + Gs P_bb Gs^T = Gs P_ba + (P_ab + Gs P_bb) Gs^T.  156 multiply-adds instead of the ~300 of
// M = G P over all twelve columns followed by P + M + M^T + M G^T, and only 18 + 6 pairs assembled through symmetry.
// g[3 i + kk] = Gs[i][kk] = dt R[kk][i]   (i, kk < 3).
template <bool QDIAG>
__device__ __forceinline__ void cov_predict_sym_blk(f2 *U, const float *g, const KfConst &k)
{
    const float dt = k.dt;
    // 1. P_aa += Gs P_ba with the OLD P_ab: the pair (2jp, 2jp+1) of row i takes sum_k Gs[i][k] (P_ab[2jp][k], P_ab[2jp+1][k])
#pragma unroll
    for (int jp = 0; jp < 3; jp++) {
        f2 c[6];
#pragma unroll
        for (int kk = 0; kk < 6; kk++) c[kk] = pkmov_rt(kk & 1, kk & 1, U[pidx(2 * jp, 3 + kk / 2)], U[pidx(2 * jp + 1, 3 + kk / 2)]);
#pragma unroll
        for (int i = 0; i < 6; i++) {
            if (jp < i / 2) continue;
            if (i < 3) U[pidx(i, jp)] = fma2(splat2(g[3 * i + 2]), c[2], fma2(splat2(g[3 * i + 1]), c[1], fma2(splat2(g[3 * i]), c[0], U[pidx(i, jp)])));
            else U[pidx(i, jp)] = fma2(splat2(dt), c[i], U[pidx(i, jp)]);
        }
    }
    // 2. P_ab += Gs P_bb
#pragma unroll
    for (int jp = 3; jp < 6; jp++) {
        const f2 r6 = rowpair(U, 6, jp), r7 = rowpair(U, 7, jp), r8 = rowpair(U, 8, jp);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            U[pidx(i, jp)] = fma2(splat2(g[3 * i + 2]), r8, fma2(splat2(g[3 * i + 1]), r7, fma2(splat2(g[3 * i]), r6, U[pidx(i, jp)])));
            const f2 r9 = rowpair(U, 9 + i, jp);
            U[pidx(3 + i, jp)] = fma2(splat2(dt), r9, U[pidx(3 + i, jp)]);
        }
    }
    // 3. P_aa += P_ab' Gs^T
#define OSK_AB(j, kk) U[pidx((j), 3 + (kk) / 2)][(kk) & 1]        /* P_ab[j][kk] = P[j][6 + kk] */
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const float b0 = OSK_AB(i, 0), b1 = OSK_AB(i, 1), b2 = OSK_AB(i, 2), b3 = OSK_AB(i, 3);
        if (i / 2 <= 0) {
            const f2 ga = {g[0], g[3]}, gb = {g[1], g[4]}, gc = {g[2], g[5]};          // (Gs[0][k], Gs[1][k])
            U[pidx(i, 0)] = fma2(splat2(b2), gc, fma2(splat2(b1), gb, fma2(splat2(b0), ga, U[pidx(i, 0)])));
        }
        if (i / 2 <= 1) {
            f2 v = U[pidx(i, 1)];
            v[0] = fmaf(b2, g[8], fmaf(b1, g[7], fmaf(b0, g[6], v[0])));                // column 2: sum_k P_ab'[i][k] Gs[2][k]
            v[1] = fmaf(dt, b3, v[1]);                                                  // column 3: dt P_ab'[i][3]
            U[pidx(i, 1)] = v;
        }
        U[pidx(i, 2)] = fma2(splat2(dt), U[pidx(i, 5)], U[pidx(i, 2)]);                 // columns 4, 5: dt (P_ab'[i][4], P_ab'[i][5])
    }
#undef OSK_AB
    if (QDIAG) {
#pragma unroll
        for (int i = 0; i < NS; i++) U[pidx(i, i / 2)][i & 1] += k.Q[i * NS + i];
    } else {
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int jp = i / 2; jp < 6; jp++) {
                const int j0 = 2 * jp, j1 = 2 * jp + 1;
                U[pidx(i, jp)] += (f2){0.5f * (k.Q[i * NS + j0] + k.Q[j0 * NS + i]), 0.5f * (k.Q[i * NS + j1] + k.Q[j1 * NS + i])};
            }
    }
}

// Ten sequential scalar updates on the paired upper triangle (diagonal R).  The state rides along as six pairs
// X[jp] = (x[2jp], x[2jp+1]): per measurement 1 add (s) + 1 class test + 1 reciprocal + 1 subtract (innovation) + 6 + 6 + 42
// packed multiply-adds (w = -c / s; x += w (x_s - z); P += c_i w) + the pairs assembled through symmetry.
// v_rcp_f32 is accurate to 1 ulp: the gain's relative error of ~1e-7 is that of any other float32 operation of the step.
// Returns true when some S entry was not a positive finite number (status bit 0; nothing is patched up: the state of that
// trajectory is then garbage or non-finite, which status bit 1 reports as well).
// Returns the innovation variance S: the callers keep the smallest one of the whole run and test it once at the end (status
// bit 0: not +normal / +denormal; a NaN S drops out of the minimum and turns the state into NaN, which the final check reports)
template <int A>
__device__ __forceinline__ float update_one_sym(f2 *X, f2 *U, const float *z, const KfConst &k)
{
    constexpr int sa = SEL[A];
    const float s = OSK_SYM(U, sa, sa) + k.R[A * NM + A];
    const float ninv = -__builtin_amdgcn_rcpf(s);
    const float ninnov = X[sa / 2][sa & 1] - z[A];
    f2 c[6], w[6];
#pragma unroll
    for (int jp = 0; jp < 6; jp++) { c[jp] = rowpair(U, sa, jp); w[jp] = c[jp] * splat2(ninv); }
#pragma unroll
    for (int jp = 0; jp < 6; jp++) X[jp] = fma2(w[jp], splat2(ninnov), X[jp]);
#pragma unroll
    for (int i = 0; i < NS; i++) {
        const float ci = c[i / 2][i & 1];
#pragma unroll
        for (int jp = i / 2; jp < 6; jp++) U[pidx(i, jp)] = fma2(splat2(ci), w[jp], U[pidx(i, jp)]);
    }
    return s;
}
__device__ __forceinline__ float update_sequential_sym(f2 *X, f2 *U, const float *z, const KfConst &k)      // -> the smallest S
{
    const float s0 = update_one_sym<0>(X, U, z, k), s1 = update_one_sym<1>(X, U, z, k), s2 = update_one_sym<2>(X, U, z, k);
    const float s3 = update_one_sym<3>(X, U, z, k), s4 = update_one_sym<4>(X, U, z, k), s5 = update_one_sym<5>(X, U, z, k);
    const float s6 = update_one_sym<6>(X, U, z, k), s7 = update_one_sym<7>(X, U, z, k), s8 = update_one_sym<8>(X, U, z, k);
    const float s9 = update_one_sym<9>(X, U, z, k);
    return fminf(fminf(fminf(fminf(s0, s1), s2), fminf(fminf(s3, s4), s5)), fminf(fminf(fminf(s6, s7), s8), s9));
}
__device__ __forceinline__ int singular_status(float smin) { return __builtin_amdgcn_classf(smin, 0x180) ? 0 : 1; }

// ---- the part of a step that consumes its inputs, hand-packed: legs in pairs (0,1) and (2,3), the step's two rotations (prior
// attitude | IMU attitude) side by side in one register pair.  hipcc's SLP vectoriser packs this code too, but pays for
// every pair with two v_mov_b32; here the pairs are the natural layout (the loaders deliver them) ----
struct StepInP {
    f2 p[2][3], f[2][3], dp[2][3];     // [leg pair q][component c] = (value of leg 2q, value of leg 2q + 1)
    float imu[6];
    uint32_t contact;                  // 4 packed bytes
};
__device__ __forceinline__ f2 lo2(f2 v) { return (f2){v[0], v[0]}; }

// sincos_small_f32 on a pair: identical polynomials in identical order, so each half has the bits of the scalar form
__device__ __forceinline__ void sincos_small2(f2 v, f2 *s, f2 *c)
{
    const f2 z = v * v;
    const f2 ps = fma2(z, fma2(z, splat2(-1.9515295891e-4f), splat2(8.3321608736e-3f)), splat2(-1.6666654611e-1f));
    *s = fma2(v * z, ps, v);
    const f2 pc = fma2(z, fma2(z, splat2(2.443315711809948e-5f), splat2(-1.388731625493765e-3f)), splat2(4.166664568298827e-2f));
    *c = fma2(z * z, pc, fma2(splat2(-0.5f), z, splat2(1.0f)));
}

// Rp[i] = (R(prior attitude)[i], R(IMU attitude)[i]), R = Rz Ry Rx (kalman_filter.py:184-193)
__device__ __forceinline__ void rotation2(const float *xs, const float *im, f2 *Rp)
{
    f2 s[3], c[3];
    // wave-uniform small-angle path as in rotation(): all six angles of every lane below pi/4
    // (two v_max3_f32 and one compare; a NaN angle drops out of the maximum and gives NaN on either path)
    const float amax3 = fmaxf(fmaxf(fabsf(xs[0]), fabsf(xs[1])), fabsf(xs[2])), imax3 = fmaxf(fmaxf(fabsf(im[0]), fabsf(im[1])), fabsf(im[2]));
    const bool big = !(fmaxf(amax3, imax3) < 0.785f);
    if (__builtin_amdgcn_ballot_w64(big) == 0ull) {
#pragma unroll
        for (int i = 0; i < 3; i++) sincos_small2((f2){xs[i], im[i]}, &s[i], &c[i]);
    } else {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            float s0, c0, s1, c1;
            sincos_f32(xs[i], &s0, &c0); sincos_f32(im[i], &s1, &c1);
            s[i] = (f2){s0, s1}; c[i] = (f2){c0, c1};
        }
    }
    const f2 sx = s[0], cx = c[0], sy = s[1], cy = c[1], sz = s[2], cz = c[2];
    const f2 t1 = sy * sx, t2 = sy * cx;
    Rp[0] = cz * cy; Rp[1] = fma2(cz, t1, -(sz * cx)); Rp[2] = fma2(cz, t2, sz * sx);
    Rp[3] = sz * cy; Rp[4] = fma2(sz, t1, cz * cx);    Rp[5] = fma2(sz, t2, -(cz * sx));
    Rp[6] = -sy;     Rp[7] = cy * sx;                  Rp[8] = cy * cx;
}

// get_odom + set_measurements (kalman_filter/kalman_filter.py:79-117) on the paired inputs; the IMU rotation is the upper
// half of Rp.  Selects, not 0/1 weights (see measurement_r).
__device__ __forceinline__ void measurement_p(const StepInP &in, const f2 *Rp, float *z /*10*/)
{
    float sum_c = 0.f, vx = 0.f, vy = 0.f, vz = 0.f, pz = 0.f;
#pragma unroll
    for (int l = 0; l < 4; l++) {
        const uint32_t cb = (in.contact >> (8 * l)) & 0xffu;
        const bool st = cb == 1u, sw = cb == 0u;
        sum_c += (float)cb;
        vx += st ? in.dp[l >> 1][0][l & 1] : 0.f;
        vy += st ? in.dp[l >> 1][1][l & 1] : 0.f;
        pz += st ? in.p[l >> 1][2][l & 1] : 0.f;
        vz += sw ? in.dp[l >> 1][2][l & 1] : 0.f;
    }
    // no stance leg -> odom = 0 (:97-98).  sum_c is 1, 2, 3 or 4 here: v_rcp_f32 (1 ulp) instead of the ~10 instructions of an
    // IEEE division
    const float inv = (sum_c != 0.f) ? __builtin_amdgcn_rcpf(sum_c) : 0.f;
    const float bx = -vx * inv, by = -vy * inv, bz = -vz * inv;
    z[0] = in.imu[0]; z[1] = in.imu[1]; z[2] = in.imu[2];
    z[3] = -pz * inv;
    z[4] = in.imu[3]; z[5] = in.imu[4]; z[6] = in.imu[5];
    z[7] = Rp[0][1] * bx + Rp[1][1] * by + Rp[2][1] * bz;
    z[8] = Rp[3][1] * bx + Rp[4][1] * by + Rp[5][1] * bz;
    z[9] = Rp[6][1] * bx + Rp[7][1] * by + Rp[8][1] * bz;
}

// The input-dependent half of next_state (misc/force_controller.py:269-291) on paired legs: PW[q][c] = world-frame foot
// positions of legs (2q, 2q+1) (the reference mutates the caller's p, :274-277), aw = I_hat^-1 sum(pw x f) (world frame),
// fs = sum f, amax = the largest |entry| of the prior rotation (the int64-truncation predicate).  Same equations as dynamics().
__device__ __forceinline__ void dynamics_rates_p(const f2 *Rp, const StepInP &in, f2 (*PW)[3], const KfConst &k, float *aw, float *fs,
                                                 float &amax)
{
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int c = 0; c < 3; c++)
            PW[q][c] = fma2(lo2(Rp[3 * c + 2]), in.p[q][2], fma2(lo2(Rp[3 * c + 1]), in.p[q][1], lo2(Rp[3 * c]) * in.p[q][0]));
    float tau[3], r[9];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
        f2 t = PW[0][c1] * in.f[0][c2];                     // sum over the four legs of (pw x f)[c], two legs per half
        t = fma2(-PW[0][c2], in.f[0][c1], t);
        t = fma2(PW[1][c1], in.f[1][c2], t);
        t = fma2(-PW[1][c2], in.f[1][c1], t);
        const f2 fsum = in.f[0][c] + in.f[1][c];
        tau[c] = t[0] + t[1];
        fs[c] = fsum[0] + fsum[1];
    }
#pragma unroll
    for (int i = 0; i < 9; i++) r[i] = Rp[i][0];
    // body-frame torque, scaled by 1/I, back to world: I_hat^-1 = R diag(1/I) R^T (R orthogonal)
    const float tb0 = (r[0] * tau[0] + r[3] * tau[1] + r[6] * tau[2]) * k.inv_inertia[0];
    const float tb1 = (r[1] * tau[0] + r[4] * tau[1] + r[7] * tau[2]) * k.inv_inertia[1];
    const float tb2 = (r[2] * tau[0] + r[5] * tau[1] + r[8] * tau[2]) * k.inv_inertia[2];
    aw[0] = r[0] * tb0 + r[1] * tb1 + r[2] * tb2;
    aw[1] = r[3] * tb0 + r[4] * tb1 + r[5] * tb2;
    aw[2] = r[6] * tb0 + r[7] * tb1 + r[8] * tb2;
    amax = 0.f;
#pragma unroll
    for (int i = 0; i < 9; i++) amax = fmaxf(amax, fabsf(r[i]));
}

// next_state on the replicated state: X in/out.
__device__ __forceinline__ int dynamics_p(f2 *X, const f2 *Rp, const StepInP &in, f2 (*PW)[3], const KfConst &k)
{
    int edge = 0;
    float aw[3], fs[3], amax;
    dynamics_rates_p(Rp, in, PW, k, aw, fs, amax);
    const float aw0 = aw[0], aw1 = aw[1], aw2 = aw[2];
    float x[NS];
#pragma unroll
    for (int i = 0; i < 6; i++) { x[2 * i] = X[i][0]; x[2 * i + 1] = X[i][1]; }
    const float w0 = x[6], w1 = x[7], w2 = x[8];
    // theta: A[0:3,6:9] = trunc(R^T) -- zero unless an entry of R reaches +-1 in float64 (see trunc_block_f64)
    if (amax >= 0.9999995f) {
        float A[9];
        edge = trunc_block_f64(x[0], x[1], x[2], A);
        x[0] += k.dt * (A[0] * w0 + A[1] * w1 + A[2] * w2);
        x[1] += k.dt * (A[3] * w0 + A[4] * w1 + A[5] * w2);
        x[2] += k.dt * (A[6] * w0 + A[7] * w1 + A[8] * w2);
    }
    x[3] += k.dt * x[9]; x[4] += k.dt * x[10]; x[5] += k.dt * x[11];      // position integrates the PRIOR velocity
    x[6] = w0 + k.dt * aw0; x[7] = w1 + k.dt * aw1; x[8] = w2 + k.dt * aw2;
    x[9] += k.dt * (fs[0] * k.inv_mass);
    x[10] += k.dt * (fs[1] * k.inv_mass);
    x[11] += k.dt * (fs[2] * k.inv_mass) + k.dt * k.gz;
#pragma unroll
    for (int i = 0; i < 6; i++) X[i] = (f2){x[2 * i], x[2 * i + 1]};
    return edge;
}

// Everything of a step that reads its inputs: measurement vector, next_state, and g = dt R^T of the PRIOR attitude for the
// covariance predict (kalman_filter.py:124: F_d and next_state both use the prior state).  The callers run the predict
// (cov_predict_sym_blk(U, g, k)) afterwards, when the input registers are free again.
// Returns status bit 4 (the int64-truncation knife edge, see trunc_block_f64) or 0.
__device__ __forceinline__ int kf_step_inputs_sym(f2 *X, const StepInP &in, const KfConst &k, float *z, f2 (*PW)[3], float *g /*9*/)
{
    f2 Rp[9];
    const float xs[3] = {X[0][0], X[0][1], X[1][0]};
    rotation2(xs, in.imu, Rp);
    measurement_p(in, Rp, z);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int kk = 0; kk < 3; kk++) g[3 * i + kk] = k.dt * Rp[3 * kk + i][0];     // g[3 i + kk] = dt R[kk][i]
    return dynamics_p(X, Rp, in, PW, k);
}

template <bool QDIAG>
__device__ __forceinline__ int kf_step_front_sym(f2 *X, f2 *U, const StepInP &in, const KfConst &k, float *z, f2 (*PW)[3])
{
    float g[9];
    const int edge = kf_step_inputs_sym(X, in, k, z, PW, g);
    cov_predict_sym_blk<QDIAG>(U, g, k);
    return edge;
}

__device__ __forceinline__ int finite_status_p(const f2 *X)
{
    f2 s = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 6; i++) s = fma2(X[i], splat2(0.f), s);   // NaN / Inf propagate
    return (s[0] + s[1] == 0.f) ? 0 : 2;
}

__device__ __forceinline__ float trace_sym(const f2 *U)
{
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NS; i++) t += OSK_SYM(U, i, i);
    return t;
}

// status bit 3 for the symmetric-storage kernels: the caller's P0 against its own transpose (66 extra loads, once per launch)
template <typename LoadF>
__device__ __forceinline__ int p0_asymmetry_status(LoadF ld)
{
    float worst = 0.f;
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
        for (int j = i + 1; j < NS; j++) {
            const float up = ld(i * NS + j), lo = ld(j * NS + i);
            worst = fmaxf(worst, fabsf(up - lo) - 1e-5f * fmaxf(fabsf(up), fabsf(lo)));
        }
    return worst > 0.f ? 8 : 0;
}

// state load / store of the paired triangle from / to the row-major P [144][B] stream
template <typename LoadF>
__device__ __forceinline__ void sym_load(f2 *U, LoadF ld)
{
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
        for (int jp = i / 2; jp < 6; jp++) U[pidx(i, jp)] = (f2){ld(i * NS + 2 * jp), ld(i * NS + 2 * jp + 1)};
}

// K_gain = np.trace(K) (kalman_filter.py:174: the ten main-diagonal entries of the 12 x 10 gain) WITHOUT forming K: for the
// optimal gain K = P- H^T S^-1 = P+ H^T R^-1, so with the diagonal R the sequential update requires K[i][a] = P+[i][sel a] / R[a][a]
// and trace(K) = sum_a P+[a][sel a] / R[a][a] -- ten divisions on the posterior covariance the sequential forms end with.
template <typename PT>
__device__ __forceinline__ float kgain_from_posterior(const PT *P, const KfConst &k)
{
    PT t = 0;
#pragma unroll
    for (int a = 0; a < NM; a++) t += P[a * NS + SEL[a]] / (PT)k.R[a * NM + a];
    return (float)t;
}

__device__ __forceinline__ float kgain_from_posterior_sym(const f2 *U, const KfConst &k)
{
    float t = 0.f;
#pragma unroll
    for (int a = 0; a < NM; a++) t += OSK_SYM(U, a, SEL[a]) / k.R[a * NM + a];
    return t;
}

template <typename PT>
__device__ __forceinline__ float trace12(const PT *P)
{
    PT t = 0;
#pragma unroll
    for (int i = 0; i < NS; i++) t += P[i * NS + i];
    return (float)t;
}

__device__ __forceinline__ int finite_status(const float *x)
{
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NS; i++) s += x[i] * 0.f;   // NaN/Inf propagate
    return (s == 0.f) ? 0 : 2;
}

// One full filter step, split in two so the caller can reuse the input registers between the halves:
//   kf_step_front: everything that consumes the step's inputs (measurement, covariance predict, dynamics)
//   kf_step_back : the measurement update (the long part; needs only z)
template <bool DENSE, bool QDIAG, typename PT>
__device__ __forceinline__ int kf_step_front(float *x, PT *P, const StepIn &in, const float *body_ref /*3 angles*/,
                                             const KfConst &k, float *z, float *pw)
{
    measurement(in, z);
    Rot r = rotation(x[0], x[1], x[2]);     // prior attitude drives both F_d and next_state
    if (DENSE) {
        Rot rb = rotation(body_ref[0], body_ref[1], body_ref[2]);
        cov_predict_dense(P, rb, k);
    } else {
        if constexpr (sizeof(PT) == 4) cov_predict<QDIAG>(P, r, k);
    }
    return dynamics(x, r, in.p, in.f, pw, k);      // status bit 4 or 0
}

// the same halves on the scalar triangle (fused kernel)
template <bool QDIAG>
__device__ __forceinline__ int kf_step_front_tri(float *x, float *U, const StepIn &in, const KfConst &k, float *z, float *pw)
{
    measurement(in, z);
    Rot r = rotation(x[0], x[1], x[2]);
    cov_predict_tri<QDIAG>(U, r, k);
    return dynamics(x, r, in.p, in.f, pw, k);
}

__device__ __forceinline__ int kf_step_back_tri(float *x, float *U, const float *z, const KfConst &k)
{
    return update_sequential_tri(x, U, z, k) | finite_status(x);
}

template <bool SEQ, bool AUX, typename PT>
__device__ __forceinline__ int kf_step_back(float *x, PT *P, const float *z, const KfConst &k, float *ptrace,
                                            float *kgain)
{
    int st;
    if (SEQ) {
        st = update_sequential(x, P, z, k);
        if (AUX) *kgain = kgain_from_posterior(P, k);
    } else {
        static_assert(SEQ, "the batch update runs on the float64 row layout (kf_dense_rows.hpp)");
        st = 0;
    }
    if (AUX) *ptrace = trace12(P);
    return st | finite_status(x);
}

}  // namespace osk
