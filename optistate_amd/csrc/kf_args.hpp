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

// The same 43 dwords by LDS-DMA (buffer_load_dword ... lds: no destination registers): row i of the step lands in
// lds[i * 64 + lane].  A one-wave workgroup can request step t + 1 at the TOP of step t and pick it up a whole step later
// (read_step_lds after s_waitcnt vmcnt(0)); with register destinations hipcc sinks the prefetch to ~100 instructions ahead
// of its first use (43 more live registers through the update otherwise) and a step then waits ~1.9 k of its 8.9 k cycles.
__device__ __forceinline__ void lds_dma4(rsrc_t r, float *l, uint32_t voff, uint32_t soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)l, 4, voff, soff, 0, 2);
}
constexpr int STEP_DWORDS = 43;
__device__ __forceinline__ void load_step_dma(const KfRunArgs &a, int t, uint32_t voff, uint32_t rowB, float *lds /* [43][64] */)
{
    const size_t B = (size_t)a.B;
    rsrc_t rp = make_rsrc(a.p + (size_t)t * 12 * B, 12 * rowB);
    rsrc_t rf = make_rsrc(a.f + (size_t)t * 12 * B, 12 * rowB);
    rsrc_t rd = make_rsrc(a.dp + (size_t)t * 12 * B, 12 * rowB);
    rsrc_t ri = make_rsrc(a.imu + (size_t)t * 6 * B, 6 * rowB);
    rsrc_t rc = make_rsrc(a.contact + (size_t)t * B, rowB);
#pragma unroll
    for (int i = 0; i < 12; i++) {
        lds_dma4(rp, lds + i * 64, voff, i * rowB);
        lds_dma4(rf, lds + (12 + i) * 64, voff, i * rowB);
        lds_dma4(rd, lds + (24 + i) * 64, voff, i * rowB);
    }
#pragma unroll
    for (int i = 0; i < 6; i++) lds_dma4(ri, lds + (36 + i) * 64, voff, i * rowB);
    lds_dma4(rc, lds + 42 * 64, voff, 0);
}
__device__ __forceinline__ void read_step_lds(const float *lds, int lane, StepIn &in)
{
#pragma unroll
    for (int i = 0; i < 12; i++) {
        in.p[i] = lds[i * 64 + lane];
        in.f[i] = lds[(12 + i) * 64 + lane];
        in.dp[i] = lds[(24 + i) * 64 + lane];
    }
#pragma unroll
    for (int i = 0; i < 6; i++) in.imu[i] = lds[(36 + i) * 64 + lane];
    in.contact = __builtin_bit_cast(uint32_t, lds[42 * 64 + lane]);
}

// The paired form the symmetric-storage kernels consume (StepInP: legs (0,1) and (2,3) side by side).  Row r and row r + 3
// of a stream are two components of neighbouring legs; from LDS the two dwords of a pair arrive with one ds_read2st64_b32.
__device__ __forceinline__ void read_step_lds_p(const float *lds, int lane, StepInP &in)
{
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int r = 6 * q + c;
            in.p[q][c] = (f2){lds[r * 64 + lane], lds[(r + 3) * 64 + lane]};
            in.f[q][c] = (f2){lds[(12 + r) * 64 + lane], lds[(12 + r + 3) * 64 + lane]};
            in.dp[q][c] = (f2){lds[(24 + r) * 64 + lane], lds[(24 + r + 3) * 64 + lane]};
        }
#pragma unroll
    for (int i = 0; i < 6; i++) in.imu[i] = lds[(36 + i) * 64 + lane];
    in.contact = __builtin_bit_cast(uint32_t, lds[42 * 64 + lane]);
}

// the same from global memory straight into the pairs' halves (fused kernel: 43 loads in flight underneath the GRU cell)
__device__ __forceinline__ void load_step_p(const KfRunArgs &a, int t, uint32_t voff, uint32_t rowB, StepInP &in)
{
    const size_t B = (size_t)a.B;
    rsrc_t rp = make_rsrc(a.p + (size_t)t * 12 * B, 12 * rowB);
    rsrc_t rf = make_rsrc(a.f + (size_t)t * 12 * B, 12 * rowB);
    rsrc_t rd = make_rsrc(a.dp + (size_t)t * 12 * B, 12 * rowB);
    rsrc_t ri = make_rsrc(a.imu + (size_t)t * 6 * B, 6 * rowB);
    rsrc_t rc = make_rsrc(a.contact + (size_t)t * B, rowB);
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const uint32_t r = 6 * q + c;
            in.p[q][c] = (f2){buf_load_nt(rp, voff, r * rowB), buf_load_nt(rp, voff, (r + 3) * rowB)};
            in.f[q][c] = (f2){buf_load_nt(rf, voff, r * rowB), buf_load_nt(rf, voff, (r + 3) * rowB)};
            in.dp[q][c] = (f2){buf_load_nt(rd, voff, r * rowB), buf_load_nt(rd, voff, (r + 3) * rowB)};
        }
#pragma unroll
    for (int i = 0; i < 6; i++) in.imu[i] = buf_load_nt(ri, voff, i * rowB);
    in.contact = buf_load_u32_nt(rc, voff, 0);
}

}  // namespace osk
