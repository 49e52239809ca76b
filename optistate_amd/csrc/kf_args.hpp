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
// Round 3: SIXTEEN-byte LDS-DMA (buffer_load_dwordx4 ... lds, gfx950).  A lane fetches four consecutive trajectories' values of
// one row, 16 lanes cover a row of the wave's 64 trajectories and an instruction covers four rows: 12 instructions per step
// instead of 43 (in-kernel timestamps: a VMEM instruction costs ~16 issue cycles at one wave per SIMD, and the 43 + 12 of a step
// were ~0.9 k of its 6.3 k cycles).  The LDS layout stays [row][64]: rows 0-11 p, 12-23 f, 24-35 dp, 36-41 imu (42, 43: the
// zeros the second imu instruction reads past its six rows), 44 contact (45-47 likewise).  EVERY lane of the wave must take
// part (a lane fetches other lanes' values): the callers keep lanes past the end of the batch alive and mask their stores.
// vo4: per-lane byte offset = (lane >> 4) rows + the first of the lane's four trajectories (step_dma_offset).
constexpr int STEP_DWORDS = 48;
constexpr int STEP_CONTACT_ROW = 44;
__device__ __forceinline__ uint32_t step_dma_offset(int lane, int wave_first_traj, uint32_t rowB)
{
    return (uint32_t)(lane >> 4) * rowB + (uint32_t)(wave_first_traj + 4 * (lane & 15)) * 4u;
}
__device__ __forceinline__ void lds_dma16(rsrc_t r, float *l, uint32_t voff, uint32_t soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)l, 16, voff, soff, 0, 2);
}
__device__ __forceinline__ void load_step_dma(const KfRunArgs &a, int t, uint32_t vo4, uint32_t rowB, float *lds /* [48][64] */)
{
    // The range check of a raw buffer access covers the VGPR offset only (not the SGPR offset), so every access here carries
    // its whole offset in the VGPR and each descriptor spans exactly the rows it may touch: a lane whose trajectories lie past
    // the end of the batch reads the next row (harmless) or, in the last row, nothing (zeros) -- never past the allocation.
    const size_t B = (size_t)a.B;
    const uint32_t vo[3] = {vo4, vo4 + 4u * rowB, vo4 + 8u * rowB};
    rsrc_t rp = make_rsrc(a.p + (size_t)t * 12 * B, 12 * rowB);
    rsrc_t rf = make_rsrc(a.f + (size_t)t * 12 * B, 12 * rowB);
    rsrc_t rd = make_rsrc(a.dp + (size_t)t * 12 * B, 12 * rowB);
    rsrc_t ri = make_rsrc(a.imu + (size_t)t * 6 * B, 6 * rowB);
    rsrc_t rc = make_rsrc(a.contact + (size_t)t * B, rowB);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        lds_dma16(rp, lds + 4 * i * 64, vo[i], 0);
        lds_dma16(rf, lds + (12 + 4 * i) * 64, vo[i], 0);
        lds_dma16(rd, lds + (24 + 4 * i) * 64, vo[i], 0);
    }
    lds_dma16(ri, lds + 36 * 64, vo[0], 0);
    lds_dma16(ri, lds + 40 * 64, vo[1], 0);                 // rows 4, 5; the lanes of rows 6, 7 are out of range: zeros
    lds_dma16(rc, lds + STEP_CONTACT_ROW * 64, vo[0], 0);   // row 0; rows 1-3 out of range
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
    in.contact = __builtin_bit_cast(uint32_t, lds[STEP_CONTACT_ROW * 64 + lane]);
}

// Asynchronous form for a software-pipelined loop: lds_issue_step starts the 22 LDS reads of a staged step into a StepRaw
// WITHOUT waiting for them (inline asm: hipcc would put an s_waitcnt in front of the first use, wherever its scheduler moved
// that); lds_fence_step is the matching wait and hands the values over as a StepInP.  Between the two calls nothing reads the
// raw registers: the fence takes them as read-write operands, so every use is ordered behind it (a compiler-made copy in
// front of the wait would read registers whose data has not arrived).
struct StepRaw { f2 v[21]; float c; };
__device__ __forceinline__ void lds_issue_step(const float *stage, int lane, StepRaw &r)
{
    const uint32_t addr = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float *)stage + (uint32_t)lane * 4u;
#define OSK_RD2(i, o0, o1) asm volatile("ds_read2st64_b32 %0, %1 offset0:" #o0 " offset1:" #o1 : "=v"(r.v[i]) : "v"(addr))
    OSK_RD2(0, 0, 3);    OSK_RD2(1, 1, 4);    OSK_RD2(2, 2, 5);    OSK_RD2(3, 6, 9);    OSK_RD2(4, 7, 10);   OSK_RD2(5, 8, 11);     // p
    OSK_RD2(6, 12, 15);  OSK_RD2(7, 13, 16);  OSK_RD2(8, 14, 17);  OSK_RD2(9, 18, 21);  OSK_RD2(10, 19, 22); OSK_RD2(11, 20, 23);   // f
    OSK_RD2(12, 24, 27); OSK_RD2(13, 25, 28); OSK_RD2(14, 26, 29); OSK_RD2(15, 30, 33); OSK_RD2(16, 31, 34); OSK_RD2(17, 32, 35);   // dp
    OSK_RD2(18, 36, 37); OSK_RD2(19, 38, 39); OSK_RD2(20, 40, 41);                                                                    // imu
#undef OSK_RD2
    asm volatile("ds_read_b32 %0, %1 offset:11264" : "=v"(r.c) : "v"(addr));       // row 44 (STEP_CONTACT_ROW): the contact word
}
__device__ __forceinline__ void lds_fence_step(StepRaw &r, StepInP &in)
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(r.v[0]), "+v"(r.v[1]), "+v"(r.v[2]), "+v"(r.v[3]), "+v"(r.v[4]), "+v"(r.v[5]), "+v"(r.v[6]), "+v"(r.v[7]),
                   "+v"(r.v[8]), "+v"(r.v[9]), "+v"(r.v[10]), "+v"(r.v[11]), "+v"(r.v[12]), "+v"(r.v[13]), "+v"(r.v[14]), "+v"(r.v[15]),
                   "+v"(r.v[16]), "+v"(r.v[17]), "+v"(r.v[18]), "+v"(r.v[19]), "+v"(r.v[20]), "+v"(r.c));
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int c = 0; c < 3; c++) { in.p[q][c] = r.v[3 * q + c]; in.f[q][c] = r.v[6 + 3 * q + c]; in.dp[q][c] = r.v[12 + 3 * q + c]; }
#pragma unroll
    for (int i = 0; i < 3; i++) { in.imu[2 * i] = r.v[18 + i][0]; in.imu[2 * i + 1] = r.v[18 + i][1]; }
    in.contact = __builtin_bit_cast(uint32_t, r.c);
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

// ---- rows kernels (16 lanes per trajectory, four trajectories per wave): the step's 4 x 43 input dwords by FIVE LDS-DMA
// instructions per wave (one per stream; lane L fetches one dword of one trajectory) instead of 43 loads per lane that every
// one of a trajectory's 16 lanes repeats; the lanes then pick their trajectory's values up with 13 broadcast LDS reads.
// No destination registers, so a step can be requested two steps ahead.  Stage layout (dwords): [p 64][f 64][dp 64][imu 64][contact 64],
// a 12-vector stored in PAIR order (0,3,1,4,2,5,6,9,7,10,8,11) so that the leg pairs of StepInP are adjacent.
constexpr int ROWS_STAGE = 5 * 64;
struct RowsDma { uint32_t vo12, vo6, vo1; };      // per-lane byte offsets inside one step's block of a 12-, 6-, 1-row stream
__device__ __forceinline__ RowsDma rows_dma_setup(int lane, int first_traj, int B)
{
    const int ord[12] = {0, 3, 1, 4, 2, 5, 6, 9, 7, 10, 8, 11};
    RowsDma d;
    const int l12 = lane < 48 ? lane : 0, l6 = lane < 24 ? lane : 0, l1 = lane < 4 ? lane : 0;
    int row = ord[0];
#pragma unroll
    for (int i = 1; i < 12; i++) row = (l12 % 12 == i) ? ord[i] : row;
    auto tr = [&](int g) { const int b = first_traj + g; return b < B ? b : B - 1; };
    d.vo12 = (uint32_t)(row * B + tr(l12 / 12)) * 4u;
    d.vo6 = (uint32_t)((l6 % 6) * B + tr(l6 / 6)) * 4u;
    d.vo1 = (uint32_t)tr(l1) * 4u;
    return d;
}
__device__ __forceinline__ void rows_dma_request(const KfRunArgs &a, int t, const RowsDma &d, uint32_t rowB, float *stage)
{
    const size_t B = (size_t)a.B;
    lds_dma4(make_rsrc(a.p + (size_t)t * 12 * B, 12 * rowB), stage, d.vo12, 0);
    lds_dma4(make_rsrc(a.f + (size_t)t * 12 * B, 12 * rowB), stage + 64, d.vo12, 0);
    lds_dma4(make_rsrc(a.dp + (size_t)t * 12 * B, 12 * rowB), stage + 128, d.vo12, 0);
    lds_dma4(make_rsrc(a.imu + (size_t)t * 6 * B, 6 * rowB), stage + 192, d.vo6, 0);
    lds_dma4(make_rsrc(a.contact + (size_t)t * B, rowB), stage + 256, d.vo1, 0);
}
__device__ __forceinline__ void rows_dma_read(const float *stage, int grp, StepInP &in)
{
    const float4 *p4 = reinterpret_cast<const float4 *>(stage + grp * 12), *f4 = reinterpret_cast<const float4 *>(stage + 64 + grp * 12),
                 *d4 = reinterpret_cast<const float4 *>(stage + 128 + grp * 12);
    float v[3][12];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float4 a = p4[i], b = f4[i], c = d4[i];
        v[0][4 * i] = a.x; v[0][4 * i + 1] = a.y; v[0][4 * i + 2] = a.z; v[0][4 * i + 3] = a.w;
        v[1][4 * i] = b.x; v[1][4 * i + 1] = b.y; v[1][4 * i + 2] = b.z; v[1][4 * i + 3] = b.w;
        v[2][4 * i] = c.x; v[2][4 * i + 1] = c.y; v[2][4 * i + 2] = c.z; v[2][4 * i + 3] = c.w;
    }
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int j = 2 * (3 * q + c);
            in.p[q][c] = (f2){v[0][j], v[0][j + 1]};
            in.f[q][c] = (f2){v[1][j], v[1][j + 1]};
            in.dp[q][c] = (f2){v[2][j], v[2][j + 1]};
        }
    const float2 *i2 = reinterpret_cast<const float2 *>(stage + 192 + grp * 6);
#pragma unroll
    for (int i = 0; i < 3; i++) { const float2 w = i2[i]; in.imu[2 * i] = w.x; in.imu[2 * i + 1] = w.y; }
    in.contact = __builtin_bit_cast(uint32_t, stage[256 + grp]);
}

}  // namespace osk
