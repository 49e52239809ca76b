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
    // Every access here carries its whole offset in the VGPR and each descriptor spans exactly the rows it may touch, so the
    // range check sees the complete offset whatever the hardware does with an SGPR part: a lane whose trajectories lie past
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
// Every stream's block is laid out [trajectory of the wave (4)][12 dwords]: lane L < 48 fetches dword L % 12 of trajectory L / 12
// (p, f, dp: component L % 12, leg-major; imu: component (L % 12) % 6; contact: the word, twelve times), so a lane reaches
// everything of its trajectory from stage + 48 grp bytes plus immediate offsets.  Lanes 48-63 repeat lane 0's fetch into the tail.
struct RowsDma { uint32_t vo12, vo6, vo1; };      // per-lane byte offsets inside one step's block of a 12-, 6-, 1-row stream
__device__ __forceinline__ RowsDma rows_dma_setup(int lane, int first_traj, int B)
{
    RowsDma d;
    const int l = lane < 48 ? lane : 0, i = l % 12, g = l / 12;
    const int tr = first_traj + g < B ? first_traj + g : B - 1;
    d.vo12 = (uint32_t)(i * B + tr) * 4u;
    d.vo6 = (uint32_t)((i < 6 ? i : i - 6) * B + tr) * 4u;
    d.vo1 = (uint32_t)tr * 4u;
    return d;
}
// Constant descriptors over the whole streams, the step's position added to the lane's VGPR offset: t * 48 B bytes, so the host
// only picks this kernel while T * 48 B < 2^32.  (Measured: with one-step descriptors and the step's position in the SGPR
// offset, every step but the first came back as zeros -- the range check took the SGPR offset into account there.)
struct RowsSrc { rsrc_t p, f, dp, imu, contact; };
__device__ __forceinline__ RowsSrc rows_src(const KfRunArgs &a, uint32_t rowB)
{
    const uint32_t T = (uint32_t)a.T;
    return RowsSrc{make_rsrc(a.p, T * 12u * rowB), make_rsrc(a.f, T * 12u * rowB), make_rsrc(a.dp, T * 12u * rowB),
                   make_rsrc(a.imu, T * 6u * rowB), make_rsrc(a.contact, T * rowB)};
}
__device__ __forceinline__ void lds_dma4_at(rsrc_t r, uint32_t lds_byte, uint32_t voff)      // lds_byte: wave-uniform
{
    // default cache policy, not the non-temporal hint of the lane kernels' streams: a 128-byte line of a row serves two
    // workgroups here (32 trajectories), and the second one should find it in the L2
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(uintptr_t)lds_byte, 4, voff, 0, 0, 0);
}
__device__ __forceinline__ void rows_dma_request(const RowsSrc &src, uint32_t t, const RowsDma &d, uint32_t rowB, uint32_t stage)
{
    const uint32_t s1 = t * rowB, s6 = 6u * s1, s12 = 12u * s1;
    const uint32_t v12 = d.vo12 + s12;
    lds_dma4_at(src.p, stage, v12);
    lds_dma4_at(src.f, stage + 256u, v12);
    lds_dma4_at(src.dp, stage + 512u, v12);
    lds_dma4_at(src.imu, stage + 768u, d.vo6 + s6);
    lds_dma4_at(src.contact, stage + 1024u, d.vo1 + s1);
}
// The pick-up as inline assembly: hipcc orders an LDS read it can see behind EVERY outstanding LDS-DMA (s_waitcnt vmcnt(0): the
// request of the step after next and the last x_out store included), which would undo the two-step prefetch.  rows_issue starts
// the reads, rows_fence waits for them.  A lane picks up ONE leg (leg = lane & 3: every quad of the 16-lane row holds the four
// legs, the sums over the legs are two quad-permute adds), the IMU attitude, the contact word and its own measurement; the
// variants with optional outputs add the lane's own row of f / dp / imu and the leg of its row of the rotated foot positions.
typedef float f4 __attribute__((ext_vector_type(4)));
struct RowsLane { uint32_t leg, grp, imu, row, rleg; };   // byte offsets inside a stage (rows_lane)
__device__ __forceinline__ RowsLane rows_lane(int grp, int r, int rr)
{
    RowsLane a;
    a.grp = (uint32_t)grp * 48u;
    a.leg = a.grp + 12u * (uint32_t)(r & 3);
    // the lane's own IMU measurement: imu[r] on rows 0..2, imu[r - 3] on rows 6..8 (anything finite elsewhere: weight 0)
    a.imu = a.grp + 768u + 4u * (uint32_t)(r < 3 ? r : (r >= 6 && r < 9) ? r - 3 : 0);
    a.row = a.grp + 4u * (uint32_t)rr;
    a.rleg = a.grp + 12u * (uint32_t)(rr / 3);
    return a;
}
struct RowsRaw {
    f2 pxy, fxy, dxy; float pz, fz, dz;       // the lane's leg
    f4 i4; float c, il;                       // imu[0..3], contact word, the lane's IMU measurement
    f2 qxy; float qz;                         // PROT: p of leg rr / 3
    float fr, dr, ir;                         // FEAT: f[rr], dp[rr], imu[rr] (rr < 6)
};
template <bool FEAT, bool PROT>
__device__ __forceinline__ void rows_issue(uint32_t stage, const RowsLane &a, RowsRaw &r)
{
    const uint32_t aleg = stage + a.leg, agrp = stage + a.grp, aimu = stage + a.imu;
    asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1" : "=v"(r.pxy) : "v"(aleg));
    asm volatile("ds_read_b32 %0, %1 offset:8" : "=v"(r.pz) : "v"(aleg));
    asm volatile("ds_read2_b32 %0, %1 offset0:64 offset1:65" : "=v"(r.fxy) : "v"(aleg));
    asm volatile("ds_read_b32 %0, %1 offset:264" : "=v"(r.fz) : "v"(aleg));
    asm volatile("ds_read2_b32 %0, %1 offset0:128 offset1:129" : "=v"(r.dxy) : "v"(aleg));
    asm volatile("ds_read_b32 %0, %1 offset:520" : "=v"(r.dz) : "v"(aleg));
    asm volatile("ds_read_b128 %0, %1 offset:768" : "=v"(r.i4) : "v"(agrp));
    asm volatile("ds_read_b32 %0, %1 offset:1024" : "=v"(r.c) : "v"(agrp));
    asm volatile("ds_read_b32 %0, %1" : "=v"(r.il) : "v"(aimu));
    if (PROT) {
        const uint32_t arl = stage + a.rleg;
        asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1" : "=v"(r.qxy) : "v"(arl));
        asm volatile("ds_read_b32 %0, %1 offset:8" : "=v"(r.qz) : "v"(arl));
    }
    if (FEAT) {
        const uint32_t arow = stage + a.row;
        asm volatile("ds_read_b32 %0, %1 offset:256" : "=v"(r.fr) : "v"(arow));
        asm volatile("ds_read_b32 %0, %1 offset:512" : "=v"(r.dr) : "v"(arow));
        asm volatile("ds_read_b32 %0, %1 offset:768" : "=v"(r.ir) : "v"(arow));
    }
}
template <bool FEAT, bool PROT>
__device__ __forceinline__ void rows_fence(RowsRaw &r)
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(r.pxy), "+v"(r.pz), "+v"(r.fxy), "+v"(r.fz), "+v"(r.dxy), "+v"(r.dz), "+v"(r.i4), "+v"(r.c), "+v"(r.il));
    if (PROT) asm volatile("" : "+v"(r.qxy), "+v"(r.qz));
    if (FEAT) asm volatile("" : "+v"(r.fr), "+v"(r.dr), "+v"(r.ir));
}
// v = sum over the four lanes of the quad, in every lane: v += v of lane ^ 1, then v += v of lane ^ 2 (DPP reads need the
// source two instructions old: the values interleave, and one s_nop covers whatever the compiler put in front)
#define OSK_QADD(i, perm) "v_add_f32_dpp %" #i ", %" #i ", %" #i " quad_perm:" perm " row_mask:0xf bank_mask:0xf\n"
__device__ __forceinline__ void quad_sum5(float &a, float &b, float &c, float &d, float &e)
{
    asm volatile("s_nop 1\n" OSK_QADD(0, "[1,0,3,2]") OSK_QADD(1, "[1,0,3,2]") OSK_QADD(2, "[1,0,3,2]") OSK_QADD(3, "[1,0,3,2]")
                 OSK_QADD(4, "[1,0,3,2]") OSK_QADD(0, "[2,3,0,1]") OSK_QADD(1, "[2,3,0,1]") OSK_QADD(2, "[2,3,0,1]")
                 OSK_QADD(3, "[2,3,0,1]") OSK_QADD(4, "[2,3,0,1]")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e));
}
__device__ __forceinline__ void quad_sum6(float &a, float &b, float &c, float &d, float &e, float &f)
{
    asm volatile("s_nop 1\n" OSK_QADD(0, "[1,0,3,2]") OSK_QADD(1, "[1,0,3,2]") OSK_QADD(2, "[1,0,3,2]") OSK_QADD(3, "[1,0,3,2]")
                 OSK_QADD(4, "[1,0,3,2]") OSK_QADD(5, "[1,0,3,2]") OSK_QADD(0, "[2,3,0,1]") OSK_QADD(1, "[2,3,0,1]")
                 OSK_QADD(2, "[2,3,0,1]") OSK_QADD(3, "[2,3,0,1]") OSK_QADD(4, "[2,3,0,1]") OSK_QADD(5, "[2,3,0,1]")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
}
#undef OSK_QADD

// kf_rows_kernel.hip (its own translation unit: compile flags): picks kf_run_rows2_kernel's instantiation and launches it
hipError_t launch_kf_rows2(const KfRunArgs &a, const float *qmat, bool feat, bool aux, hipStream_t s);
// kf_dense_rows.hip: the predict_mpc (dense F_d) filter in float64, 16 lanes per trajectory; qr = Q (144) | R (100) on the device
hipError_t launch_kf_dense_rows(const KfRunArgs &a, const float *qr, bool seq, bool feat, bool aux, hipStream_t s, bool dense = true);
// the batched single-step pieces on the same layout (P as float, or double with OS_KF_P_FLOAT64)
hipError_t launch_kf_predict_rows(int B, float *p, const float *f, const float *body_ref, float *x, void *P, bool p64, float *ptrace_out,
                                  const KfConst &k, const float *qr, bool dense, hipStream_t s);
hipError_t launch_kf_update_rows(int B, const float *z, float *x, void *P, void *K_out, bool p64, float *ptrace_out, float *kgain_out,
                                 int32_t *status, const KfConst &k, const float *qr, bool seq, hipStream_t s);

}  // namespace osk
