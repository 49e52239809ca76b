// kf_args.hpp -- argument block and input-stream loader shared by the Kalman kernels and the fused kernel.
#pragma once
#include "kf_device.hpp"

namespace osk {

struct KfRunArgs {
    int B, T;
    const float *p, *f, *dp, *imu;
    const uint32_t *contact;
    const float *body_ref;
    float *x, *P;
    float *x_out, *p_rot_out, *ptrace_out, *kgain_out;
    int32_t *status;
    // optional feature-row emission (fused path v0): normalised rows [T][feat_I][B]
    const float *accel;      // [T][6][B]
    const float *minmax;     // [2][60]: mins, maxs
    float *feat_out;
    int feat_I;
    // per-trajectory diagonal noise (os_kf_run_noise): q_diag [12][B], r_diag [10][B]; null = the context-wide Q / R in k
    const float *q_diag = nullptr, *r_diag = nullptr;
    KfConst k;
};

// Loads one step's 43 input dwords for this lane.  rowB = B*4 (bytes per row), voff = b*4.
__device__ __forceinline__ void load_step(const KfRunArgs &a, int t, uint32_t voff, uint32_t rowB, StepIn &in)
{
    const size_t B = (size_t)a.B;
    rsrc_t rp = make_rsrc(a.p + (size_t)t * 12 * B, 12 * rowB);
    rsrc_t rf = make_rsrc(a.f + (size_t)t * 12 * B, 12 * rowB);
    rsrc_t rd = make_rsrc(a.dp + (size_t)t * 12 * B, 12 * rowB);
    rsrc_t ri = make_rsrc(a.imu + (size_t)t * 6 * B, 6 * rowB);
    rsrc_t rc = make_rsrc(a.contact + (size_t)t * B, rowB);
#pragma unroll
    for (int i = 0; i < 12; i++) {
        in.p[i] = buf_load_nt(rp, voff, i * rowB);
        in.f[i] = buf_load_nt(rf, voff, i * rowB);
        in.dp[i] = buf_load_nt(rd, voff, i * rowB);
    }
#pragma unroll
    for (int i = 0; i < 6; i++) in.imu[i] = buf_load_nt(ri, voff, i * rowB);
    in.contact = buf_load_u32_nt(rc, voff, 0);
}


}  // namespace osk
