// fused_kernels.hip -- Kalman filter + feature pack + min-max normalise + GRU layer 0 (+ head) in ONE kernel for
// the headline configuration (60 features -> GRU hidden 64).  Nothing but the input streams, the 48-byte state
// history and the final 24 outputs touches HBM: feature rows never leave registers.
//
// Wave-autonomous design (one CU = one 256-thread workgroup = 4 waves, no inter-wave synchronisation in the T loop):
//   * each wave owns 64 trajectories; lane = trajectory for the Kalman part (x: 12, upper triangle of P: 78 VGPRs, scalar storage: kf_device.hpp);
//   * the 60 normalised features of step t sit in the owning lane's registers; one v_permlane32_swap per feature
//     pair turns them into the two 32-row MFMA A-fragments (lanes 0-31: feature k, lanes 32-63: feature k+1) -- no LDS;
//   * the wave then runs its own rows through the GRU cell: gates[64 x 192] = [x_t | h] [W_ih | W_hh]^T on
//     v_mfma_f32_32x32x2_f32, B-fragments from the LDS-resident fragment-ordered weights (95 KB, loaded once per
//     workgroup), recurrent A-fragments from the wave-private h tile in LDS ([64][65] floats);
//   * the two 32-column chunks are processed one after the other (128 accumulator registers at a time); chunk 0's
//     new h is parked in 32 VGPRs until chunk 1's MFMAs no longer need the old h.
// LDS: 2 x 12,032 floats weights+bias (96,256 B) + 4 x 64 x 65 floats h (66,560 B) = 162,816 B of the 163,840 B per CU.
#include "launch.hpp"

#include "gru_common.hpp"
#include "kf_args.hpp"

namespace osf {

using namespace osk;

constexpr int H = 64, KX = 60, KPX = KX / 2, KPH = H / 2, HS = H + 1;
constexpr int CHF = (KPX + KPH) * 3 * 64 + 128;   // floats per 32-column chunk in the packed layout (incl. biases)
constexpr int WAVES = 4;
constexpr size_t LDS_BYTES = (size_t)(2 * CHF + WAVES * 64 * HS) * sizeof(float);

struct FusedArgs {
    KfRunArgs kf;              // streams, x/P in-out, x_out, status, accel, minmax
    const float *wpacked;      // layer-0 weights in fragment order (2 chunks x CHF floats)
    const float *nrm;          // [120]: mins, then 1/(max-min) (norm_prep_kernel)
    const float *fcw, *fcb;    // head (used when seq_out == nullptr)
    int C, use_sigmoid;
    float *out;                // [B][C]
    float *seq_out;            // [T][64][B] layer-0 output sequence for deeper stacks, or nullptr
    float *h_last;             // v2 only: h_T [64][B] for the trailing head launch (when seq_out == nullptr)
};

// =====================================================================================================================
// element i = 3 leg + component of a per-leg quantity stored as leg pairs (StepInP, kf_device.hpp); state element i of the pairs
#define OSF_LEG(v, i) (v)[((i) / 3) >> 1][(i) % 3][((i) / 3) & 1]
#define OSF_X(i) X[(i) >> 1][(i) & 1]

// fused_kf_gru_kernel_v2 -- the same path with the GRU cell TRANSPOSED and the hidden state in registers.
//
// gates^T[unit][trajectory] = [W_ih | W_hh] . [x_t | h]^T: the weights are the MFMA A operand (from LDS, one ds_read_b128 per
// k-pair fetches the three gates' fragments), the activations the B operand.  The result tile puts the trajectory on
// the lane (col = lane & 31) and 16 hidden units in the registers (unit = 32c + (e&3) + 8(e>>2) + 4(lane>>5)) -- and that is
// exactly the B-operand layout of a k-pair whose two k values are the units (e, lane half 0) and (e, lane half 1).  The
// recurrent weights are therefore packed with their k axis in THAT order (fused_pack_kernel), and h_t never leaves the 64
// registers it is computed in: no LDS tile, no transpose, no barrier, no address arithmetic.  What else moved out of the loop:
//   * the gate scale factors (-log2 e for r and z, 2 log2 e for n) and the min-max scale 1/(max-min) are folded into the
//     packed weights, the four biases into the accumulators' initial value (read from LDS, 16 ds_read_b128 per chunk):
//     sigmoid is exp2 + add + rcp, the normalisation one subtraction;
//   * the fc + sigmoid head is a trailing launch (gru_head_kernel on h_T [64][B]), not 64 + 24 registers of this kernel;
//   * the Kalman step runs on the paired upper triangle (kf_device.hpp: v_pk_fma_f32 on aligned pairs, no shuffles).
// LDS: 2 chunks x (62 k-pairs x 64 lanes x 4 floats + 128 bias floats) + 64 minima = 128,256 B.
// =====================================================================================================================
constexpr int CHF2 = (KPX + KPH) * 256 + 128;

// Values that live in the ACCUMULATOR half of the register file (AGPRs) and are only touched through these helpers.  gfx950
// has 512 registers per lane, but only the 256 architectural VGPRs can be VALU operands; MFMA operands may come from either
// half.  The B operands of the gate GEMM (the 60 features of a step, the 64 hidden-state registers) are written once and
// then read by six MFMAs each, so they are kept in AGPRs by construction ("a" constraints): hipcc on its own puts the
// accumulators there and shuffles ~700 values per step between the halves (v_accvgpr_read/write) to make room.
__device__ __forceinline__ float agpr_put(float v)
{
    float a;
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v));
    return a;
}
// buffer_store straight from an AGPR (gfx90a+: one register file, VMEM data operands may be accumulator registers): the
// layer-0 sequence of the SEQOUT variants leaves without a v_accvgpr_read per element.  vo: per-lane byte offset, so: wave-uniform.
// Two stores per statement behind ONE `s_nop 4`: an SGPR written by a VALU instruction (v_readfirstlane, and the v_readlane that
// reloads a spilled SGPR -- this kernel has ~160 of those) needs five wait states before a VMEM instruction reads it, and hipcc's
// hazard recogniser does not look inside inline assembly (without the nop: random wrong addresses, GRU l-inf 1.5e-2).
__device__ __forceinline__ void buf_store_agpr2(osk::rsrc_t r, uint32_t vo, uint32_t so0, float a0, uint32_t so1, float a1)
{
    asm volatile("s_nop 4\n\tbuffer_store_dword %0, %2, %3, %4 offen\n\tbuffer_store_dword %1, %2, %3, %5 offen"
                 : : "a"(a0), "a"(a1), "v"(vo), "s"(r), "s"(so0), "s"(so1) : "memory");
}
// a descriptor hipcc knows to be wave-uniform (an "s" operand of inline assembly must be: nothing legalises it afterwards)
__device__ __forceinline__ osk::rsrc_t make_rsrc_uniform(const void *base, uint32_t bytes)
{
    const uint64_t p = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ float agpr_get(float a)
{
    float v;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
    return v;
}
// acc (VGPRs) += W-fragment (VGPR, from LDS) x B-fragment (AGPR).  hipcc's hazard recogniser does not look inside inline asm:
// successive MFMAs here never touch the same accumulator within five instructions (>= 320 cycles), and mfma_drain() below
// covers the MFMA -> VALU read of the results.
__device__ __forceinline__ void mfma_va(f32x16 &acc, float w, float b_agpr)
{
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "a"(b_agpr));
}
__device__ __forceinline__ float agpr_mov(float a)
{
    float d;
    asm volatile("v_accvgpr_mov_b32 %0, %1" : "=a"(d) : "a"(a));
    return d;
}
__device__ __forceinline__ void mfma_drain(f32x16 (&acc)[4])
{
    // 16-pass MFMA result -> VALU read: 18 wait states (cdna4 ISA, MFMA hazard table); the operands tie the nops to the values
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
}
constexpr int IMG2 = 2 * CHF2 + 64;          // + the 60 feature minima (read per step with wave-uniform ds_read_b128)
constexpr size_t LDS2_BYTES = (size_t)IMG2 * sizeof(float);

__global__ void fused_pack_kernel(const float *__restrict__ w0 /* layer 0, torch layout */, const float *__restrict__ minmax,
                                  float *__restrict__ img /* [IMG2] */)
{
    constexpr float LOG2E = 1.44269504088896341f;
    const float *Wih = w0, *Whh = Wih + 3 * H * KX, *bih = Whh + 3 * H * H, *bhh = bih + 3 * H;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * CHF2; i += gridDim.x * blockDim.x) {
        const int c = i / CHF2, r = i % CHF2;
        float v = 0.f;
        if (r < (KPX + KPH) * 256) {
            const int q = r >> 8, lane = (r >> 2) & 63, g = r & 3, li = lane & 31, lh = lane >> 5;
            if (g < 3) {
                const int col = g * H + 32 * c + li;
                const float gs = g < 2 ? -LOG2E : 2.0f * LOG2E;
                if (q < KPX) {
                    const int k = 2 * q + lh;
                    v = Wih[col * KX + k] * (1.0f / (minmax[KX + k] - minmax[k])) * gs;
                } else {
                    const int qq = q - KPX, cp = qq >> 4, e = qq & 15;
                    const int k = 32 * cp + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    v = Whh[col * H + k] * gs;
                }
            }
        } else {
            const int j = r - (KPX + KPH) * 256, lh = j >> 6, g4 = (j >> 4) & 3, e = j & 15;
            const int u = 32 * c + (e & 3) + 8 * (e >> 2) + 4 * lh;
            if (g4 == 0) v = (bih[u] + bhh[u]) * -LOG2E;
            else if (g4 == 1) v = (bih[H + u] + bhh[H + u]) * -LOG2E;
            else if (g4 == 2) v = bih[2 * H + u] * (2.0f * LOG2E);
            else v = bhh[2 * H + u] * (2.0f * LOG2E);
        }
        img[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) img[2 * CHF2 + threadIdx.x] = threadIdx.x < KX ? minmax[threadIdx.x] : 0.f;
}

// Development build only (-DOS_FUSED_TS, tools/fused_ts.sh): shader-clock stamps between the phases of a step, summed over the
// steps by lane 0 of workgroup 0 and printed at the end of the kernel.
#ifdef OS_FUSED_TS
#define OSF_TS_DECL unsigned long long ts_prev = 0, ts_sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define OSF_TS(i)                                                                  \
    {                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                         \
        const unsigned long long now = __builtin_readcyclecounter();               \
        if ((i) > 0) ts_sum[i] += now - ts_prev;                                   \
        ts_prev = now;                                                             \
        __builtin_amdgcn_sched_barrier(0);                                         \
    }
#else
#define OSF_TS_DECL
#define OSF_TS(i)
#endif

// load j (0..48) of a step's 49 input dwords into its place in the paired input block (load_step_p's order: p, f, dp rows as leg
// pairs, imu, contact, accel): the fused kernels issue them one at a time between their MFMA groups (j is a constant after unrolling)
struct StepSrc { rsrc_t p, f, dp, imu, contact, accel; };
__device__ __forceinline__ StepSrc step_src(const KfRunArgs &a, int t, uint32_t rowB)
{
    const size_t B = (size_t)a.B;
    return StepSrc{make_rsrc(a.p + (size_t)t * 12 * B, 12 * rowB), make_rsrc(a.f + (size_t)t * 12 * B, 12 * rowB), make_rsrc(a.dp + (size_t)t * 12 * B, 12 * rowB),
                   make_rsrc(a.imu + (size_t)t * 6 * B, 6 * rowB), make_rsrc(a.contact + (size_t)t * B, rowB), make_rsrc(a.accel + (size_t)t * 6 * B, 6 * rowB)};
}
__device__ __forceinline__ void prefetch_input(const StepSrc &src, uint32_t voff, uint32_t rowB, int j, StepInP &in, float (&acl)[6])
{
    if (j < 36) {
        const int st = j / 12, R = j % 12, q = R / 6, c = (R % 6) % 3, hf = (R % 6) / 3;
        const float v = buf_load_nt(st == 0 ? src.p : st == 1 ? src.f : src.dp, voff, (uint32_t)R * rowB);
        if (st == 0) in.p[q][c][hf] = v; else if (st == 1) in.f[q][c][hf] = v; else in.dp[q][c][hf] = v;
    } else if (j < 42) {
        in.imu[j - 36] = buf_load_nt(src.imu, voff, (uint32_t)(j - 36) * rowB);
    } else if (j == 42) {
        in.contact = buf_load_u32_nt(src.contact, voff, 0);
    } else {
        acl[j - 43] = buf_load_nt(src.accel, voff, (uint32_t)(j - 43) * rowB);
    }
}

// NRB = 32-trajectory column blocks per wave.  NRB = 2: a wave owns 64 trajectories, lane = trajectory (the layout above).  NRB = 1 (round 6,
// batches that leave half the chip idle at 256 trajectories per CU): a wave owns 32 trajectories and BOTH lane halves run the filter
// of trajectory wbase + (lane & 31) redundantly -- the B fragment of a k-pair is then one v_cndmask (lanes 0-31 take feature 2kp, lanes
// 32-63 feature 2kp + 1) instead of a v_permlane32_swap, the GRU half is the rb = 0 passes alone, only lanes 0-31 store.
template <bool QDIAG, bool SEQOUT, int NRB>
__global__ __launch_bounds__(256, 1) void fused_kf_gru_kernel_v2(const FusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const KfRunArgs &k = a.kf;
    const size_t B = (size_t)k.B;
    {
        const float4 *src = reinterpret_cast<const float4 *>(a.wpacked);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < IMG2 / 4; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const float4 *mins4 = reinterpret_cast<const float4 *>(lds + 2 * CHF2);      // minima from LDS: 60 SGPRs would spill

    const int wbase = (blockIdx.x * 4 + (threadIdx.x >> 6)) * (32 * NRB);      // first trajectory of this wave
    const int b = wbase + (NRB == 2 ? lane : li);
    const bool live = b < k.B && (NRB == 2 || lh == 0);      // the lanes that store
    const int bb = b < k.B ? b : k.B - 1;              // dead lanes shadow the last trajectory, stores masked
    const uint32_t voff = (uint32_t)bb * 4u, rowB = (uint32_t)k.B * 4u;

    f2 X[6];                       // the filter state as pairs (x[2i], x[2i+1])
    f2 U[NU];
    int status = 0;
    float smin = 3.0e38f;                  // the smallest innovation variance of the run (status bit 0)
    {
        rsrc_t rx = make_rsrc(k.x, 12 * rowB), rP = make_rsrc(k.P, 144 * rowB);
#pragma unroll
        for (int i = 0; i < 6; i++) X[i] = (f2){buf_load(rx, voff, 2 * i * rowB), buf_load(rx, voff, (2 * i + 1) * rowB)};
        // status bit 3: P0 not symmetric (the paired triangle reads the upper half only; see include/optistate_hip.h)
        status = p0_asymmetry_status([&](int e) { return buf_load(rP, voff, (uint32_t)e * rowB); });
        sym_load(U, [&](int e) { return buf_load(rP, voff, (uint32_t)e * rowB); });
    }
    // h_t of this wave's 64 trajectories: hreg[rb][c][e] = h[unit 32c + (e&3) + 8(e>>2) + 4 lh][trajectory wbase + 32 rb + li]
    // (AGPR-resident: agpr_put / agpr_get)
    float hreg[NRB][2][16];
#pragma unroll
    for (int rb = 0; rb < NRB; rb++)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int e = 0; e < 16; e++) hreg[rb][c][e] = agpr_put(0.f);      // h0 = 0 (gru/gru_model.py:27)

    StepInP in;
    float acl[6];
    load_step_p(k, 0, voff, rowB, in);
    {
        rsrc_t ra = make_rsrc(k.accel, 6 * rowB);
#pragma unroll
        for (int i = 0; i < 6; i++) acl[i] = buf_load_nt(ra, voff, i * rowB);
    }

    // SEQOUT: lanes 0-31 / 32-63 write two 128-byte row segments of seq_out[t][unit][trajectory]; trajectories past the batch get
    // an offset no descriptor covers
    const uint32_t vo_seq0 = (wbase + li) < k.B ? (uint32_t)(wbase + li) * 4u + (uint32_t)(4 * lh) * rowB : 0x7ffffff0u;
    const uint32_t vo_seq1 = (wbase + 32 + li) < k.B ? (uint32_t)(wbase + 32 + li) * 4u + (uint32_t)(4 * lh) * rowB : 0x7ffffff0u;
    OSF_TS_DECL
    for (int t = 0; t < k.T; t++) {
        OSF_TS(0)
        const osk::rsrc_t rs_prev = make_rsrc_uniform(SEQOUT ? a.seq_out + (size_t)(t > 0 ? t - 1 : 0) * H * B : nullptr,
                                                      (SEQOUT && t > 0) ? (uint32_t)H * rowB : 0u);
        // ================= Kalman step (lane = trajectory) =================
        // Order chosen for register pressure: everything that reads the step's 55 input registers runs first (measurement,
        // dynamics, the 48 raw-input features, which go straight to AGPRs); the covariance predict and the update then work
        // with the filter state alone.  (predict's F_d and next_state both use the PRIOR attitude: kalman_filter.py:124,133.)
        float z[NM], FA[KX];
        f2 PW[2][3];
        float g9[9];
        status |= kf_step_inputs_sym(X, in, k.k, z, PW, g9);
        __builtin_amdgcn_sched_barrier(0);
        OSF_TS(1)                                        // wait for the prefetched inputs + rotations, odometry, next_state
        // Features [x_post | accel | f | p_world | dp | imu] minus their minimum (the 1/(max-min) scale sits in the packed
        // weights).  v_permlane32_swap turns a feature pair into the two B fragments of its k-pair: afterwards the first
        // register holds trajectories 0-31 (lanes 0-31: feature 2kp, lanes 32-63: feature 2kp+1), the second trajectories
        // 32-63.  Inline asm because hipcc (ROCm 7.2) drops the second result of __builtin_amdgcn_permlane32_swap here; six
        // swaps per block share one leading / trailing s_nop for the VALU <-> permlane wait states.
        auto feat6 = [&](int j0, float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7, float a8, float a9,
                         float a10, float a11) {
            float v[12] = {a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11};
#pragma unroll
            for (int i4 = 0; i4 < 3; i4++) {
                const float4 mn = mins4[j0 / 4 + i4];
                v[4 * i4] -= mn.x; v[4 * i4 + 1] -= mn.y; v[4 * i4 + 2] -= mn.z; v[4 * i4 + 3] -= mn.w;
            }
            if (NRB == 2) {
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\t"
                             "v_permlane32_swap_b32 %6, %7\n\tv_permlane32_swap_b32 %8, %9\n\tv_permlane32_swap_b32 %10, %11\n\ts_nop 1"
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                               "+v"(v[9]), "+v"(v[10]), "+v"(v[11]));
#pragma unroll
                for (int i = 0; i < 12; i++) FA[j0 + i] = agpr_put(v[i]);
            } else {
                // both lane halves hold the same trajectory: the k-pair's fragment is a select on the lane half
#pragma unroll
                for (int i = 0; i < 12; i += 2) FA[j0 + i] = agpr_put(lh ? v[i + 1] : v[i]);
            }
        };
        feat6(12, acl[0], acl[1], acl[2], acl[3], acl[4], acl[5], OSF_LEG(in.f, 0), OSF_LEG(in.f, 1), OSF_LEG(in.f, 2), OSF_LEG(in.f, 3), OSF_LEG(in.f, 4), OSF_LEG(in.f, 5));
        feat6(24, OSF_LEG(in.f, 6), OSF_LEG(in.f, 7), OSF_LEG(in.f, 8), OSF_LEG(in.f, 9), OSF_LEG(in.f, 10), OSF_LEG(in.f, 11), OSF_LEG(PW, 0), OSF_LEG(PW, 1), OSF_LEG(PW, 2), OSF_LEG(PW, 3), OSF_LEG(PW, 4), OSF_LEG(PW, 5));
        feat6(36, OSF_LEG(PW, 6), OSF_LEG(PW, 7), OSF_LEG(PW, 8), OSF_LEG(PW, 9), OSF_LEG(PW, 10), OSF_LEG(PW, 11), OSF_LEG(in.dp, 0), OSF_LEG(in.dp, 1), OSF_LEG(in.dp, 2), OSF_LEG(in.dp, 3), OSF_LEG(in.dp, 4), OSF_LEG(in.dp, 5));
        feat6(48, OSF_LEG(in.dp, 6), OSF_LEG(in.dp, 7), OSF_LEG(in.dp, 8), OSF_LEG(in.dp, 9), OSF_LEG(in.dp, 10), OSF_LEG(in.dp, 11), in.imu[0], in.imu[1], in.imu[2], in.imu[3], in.imu[4],
              in.imu[5]);
        __builtin_amdgcn_sched_barrier(0);          // the inputs are in AGPRs now: the covariance work below starts with their registers free
        OSF_TS(2)                                        // 48 raw features: subtract, swap, AGPR
        cov_predict_sym_blk<QDIAG>(U, g9, k.k);
        __builtin_amdgcn_sched_barrier(0);
        OSF_TS(3)                                        // covariance predict
        smin = fminf(smin, update_sequential_sym(X, U, z, k.k));      // non-finite states stay non-finite: checked once after the loop
        // (the twelve x_out stores of this step are issued inside the first MFMA pass below: a VMEM instruction costs ~16 issue
        // cycles here, free underneath the matrix pipe; X does not change until the next step's filter phase)
        feat6(0, OSF_X(0), OSF_X(1), OSF_X(2), OSF_X(3), OSF_X(4), OSF_X(5), OSF_X(6), OSF_X(7), OSF_X(8), OSF_X(9), OSF_X(10), OSF_X(11));
        OSF_TS(4)                                        // ten measurement updates, x_out stores, 12 state features

        // the next step's input rows (requested under this step's last MFMA pass; the descriptors' scalar arithmetic hides under the first)
        const StepSrc nsrc = step_src(k, (t + 1 < k.T) ? t + 1 : t, rowB);
        // One 32-trajectory column block and one 32-unit chunk at a time: 64 accumulator registers, in VGPRs, where the cell
        // update reads them directly (the weight fragments are simply read from LDS again for each of the four passes).  Chunk
        // 0's new h waits in 16 spare AGPRs until chunk 1's MFMAs no longer need the old one.
#pragma unroll
        for (int rb = 0; rb < NRB; rb++) {
            float park[16];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const float *Wc = lds + c * CHF2;
                f32x16 acc[4];
                {
                    // accumulators start at the (pre-scaled) biases of this lane's 16 units
                    const float4 *bc = reinterpret_cast<const float4 *>(Wc + (KPX + KPH) * 256 + lh * 64);
#pragma unroll
                    for (int g = 0; g < 4; g++)
#pragma unroll
                        for (int v4 = 0; v4 < 4; v4++) {
                            const float4 bv = bc[g * 4 + v4];
                            acc[g][4 * v4] = bv.x; acc[g][4 * v4 + 1] = bv.y; acc[g][4 * v4 + 2] = bv.z; acc[g][4 * v4 + 3] = bv.w;
                        }
                }
                // two-deep software pipeline over the 62 k-pairs: the weight fragments of k-pair q+1 (one ds_read_b128: the three
                // gates' fragments) are requested before the three MFMAs of k-pair q issue
                const float4 *Wq = reinterpret_cast<const float4 *>(Wc) + lane;
                float4 wb[2];
                wb[0] = Wq[0];
#pragma unroll
                for (int q = 0; q < KPX + KPH; q++) {
                    const int cur = q & 1, nxt = cur ^ 1;
                    if (rb == 0 && c == 0 && q == 8) {
                        // shadow lanes: an offset no descriptor covers (the range check drops the store) instead of a branch
                        const uint32_t vst = live ? voff : 0x7ffffff0u;
                        rsrc_t ro = make_rsrc(k.x_out + (size_t)t * 12 * B, 12 * rowB);
#pragma unroll
                        for (int i = 0; i < NS; i++) buf_store_nt(ro, vst, i * rowB, OSF_X(i));
                    }
                    if (SEQOUT && rb == 0 && c == 0 && q >= 12 && q < 12 + 32) {
                        // layer 0's h_{t-1} -> seq_out[t-1] (deeper stacks), two stores per k-pair of this pass, straight from the
                        // AGPRs: every hreg keeps step t-1's value until the cell update of pass (0, 1).  At t = 0 the descriptor
                        // covers nothing; lanes past the batch use an offset no descriptor covers (no branch in the loop).
                        const int i0 = 2 * (q - 12), srb = i0 >> 5, sc = (i0 >> 4) & 1, se = i0 & 15;      // elements i0, i0 + 1: se even
                        if (srb < NRB)
                            buf_store_agpr2(rs_prev, srb ? vo_seq1 : vo_seq0, (uint32_t)(32 * sc + (se & 3) + 8 * (se >> 2)) * rowB, hreg[srb < NRB ? srb : 0][sc][se],
                                            (uint32_t)(32 * sc + ((se + 1) & 3) + 8 * ((se + 1) >> 2)) * rowB, hreg[srb < NRB ? srb : 0][sc][se + 1]);
                    }
#ifdef OSF_V2_PREFETCH_BLOCK
                    if (rb == NRB - 1 && c == 1 && q == KPX) {
                        // the next step's 49 input loads go out underneath the last ~100 MFMAs and the cell update
                        const int tn = (t + 1 < k.T) ? t + 1 : t;
                        load_step_p(k, tn, voff, rowB, in);
                        rsrc_t ra = make_rsrc(k.accel + (size_t)tn * 6 * B, 6 * rowB);
#pragma unroll
                        for (int i = 0; i < 6; i++) acl[i] = buf_load_nt(ra, voff, i * rowB);
                    }
#else
                    if (rb == NRB - 1 && c == 1 && q >= KPX - 18) {
                        // the next step's 49 input loads, one per k-pair of the last pass's final 49 (round 6: a VMEM instruction costs ~16 issue
                        // cycles at one wave per SIMD; one at a time the matrix pipe covers them, 49 in a row it ran dry for ~0.8 k cycles)
                        if (q - (KPX - 18) < 49) prefetch_input(nsrc, voff, rowB, q - (KPX - 18), in, acl);
                    }
#endif
                    const float bv = q < KPX ? FA[2 * (q < KPX ? q : 0) + rb]
                                             : hreg[rb][(q - KPX) >> 4 & 1][(q >= KPX ? q - KPX : 0) & 15];
                    const int gn = q < KPX ? 2 : 3;          // input part feeds gi_n, recurrent part gh_n
                    mfma_va(acc[0], wb[cur].x, bv);
                    // Next k-pair's fragments: requested behind this k-pair's FIRST MFMA, never in front of it.  The matrix pipe reads
                    // an MFMA's A/B registers when the instruction starts, not when it issues; a ds_read placed right behind the
                    // previous k-pair's last MFMA can land in registers that MFMA has not read yet (seen in the split-bf16 kernel
                    // below; tools/bf16_determinism.py), and hipcc does not guard inline-asm MFMAs.
                    __builtin_amdgcn_sched_barrier(0);
                    if (q + 1 < KPX + KPH) wb[nxt] = Wq[(q + 1) * 64];
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_va(acc[1], wb[cur].y, bv);
                    mfma_va(acc[gn], wb[cur].z, bv);
                    __builtin_amdgcn_sched_barrier(0);
                }
                OSF_TS(5)                                // bias init + 186 MFMAs (x 4 passes)
                mfma_drain(acc);
                // Cell update in place on the accumulator registers (scales folded into the weights: sigmoid = rcp(1 + exp2(a))),
                // stage by stage over the 16 elements so that no instruction depends on its predecessor, the seven
                // non-transcendental operations per element as v_pk_add/fma on register pairs: with one wavefront per SIMD an
                // instruction costs an issue slot whether it is packed or not.
                {
                    f2 *A0 = reinterpret_cast<f2 *>(&acc[0]), *A1 = reinterpret_cast<f2 *>(&acc[1]), *A2 = reinterpret_cast<f2 *>(&acc[2]),
                       *A3 = reinterpret_cast<f2 *>(&acc[3]);
                    const f2 one = {1.0f, 1.0f};
#pragma unroll
                    for (int e = 0; e < 16; e++) { acc[0][e] = __builtin_amdgcn_exp2f(acc[0][e]); acc[1][e] = __builtin_amdgcn_exp2f(acc[1][e]); }
#pragma unroll
                    for (int p2 = 0; p2 < 8; p2++) { A0[p2] += one; A1[p2] += one; }
#pragma unroll
                    for (int e = 0; e < 16; e++) { acc[0][e] = __builtin_amdgcn_rcpf(acc[0][e]); acc[1][e] = __builtin_amdgcn_rcpf(acc[1][e]); }      // r, z
#pragma unroll
                    for (int p2 = 0; p2 < 8; p2++) A2[p2] = fma2(A0[p2], A3[p2], A2[p2]);                      // gi_n + r gh_n (pre-scaled)
#pragma unroll
                    for (int e = 0; e < 16; e++) acc[2][e] = __builtin_amdgcn_exp2f(acc[2][e]);
#pragma unroll
                    for (int p2 = 0; p2 < 8; p2++) A2[p2] += one;
#pragma unroll
                    for (int e = 0; e < 16; e++) { acc[2][e] = __builtin_amdgcn_rcpf(acc[2][e]); acc[3][e] = agpr_get(hreg[rb][c][e]); }
#pragma unroll
                    for (int p2 = 0; p2 < 8; p2++) A2[p2] = fma2((f2){-2.0f, -2.0f}, A2[p2], one);             // n = tanh(.)
#pragma unroll
                    for (int p2 = 0; p2 < 8; p2++) A3[p2] = A3[p2] - A2[p2];                                   // h - n
#pragma unroll
                    for (int p2 = 0; p2 < 8; p2++) A3[p2] = fma2(A1[p2], A3[p2], A2[p2]);                      // (1 - z) n + z h
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const float hn = agpr_put(acc[3][e]);
                        if (c == 0) park[e] = hn;                 // old h[0:32] is still an operand of chunk 1
                        else hreg[rb][1][e] = hn;
                    }
                }
                OSF_TS(6)                                // drain + cell update (x 4 passes)
            }
#pragma unroll
            for (int e = 0; e < 16; e++) hreg[rb][0][e] = agpr_mov(park[e]);
        }
        OSF_TS(7)
    }
    if (SEQOUT) {
        // the last step's h
        const osk::rsrc_t rs = make_rsrc_uniform(a.seq_out + (size_t)(k.T - 1) * H * B, (uint32_t)H * rowB);
#pragma unroll
        for (int i = 0; i < 32 * NRB; i += 2) {
            const int srb = i >> 5, sc = (i >> 4) & 1, se = i & 15;
            buf_store_agpr2(rs, srb ? vo_seq1 : vo_seq0, (uint32_t)(32 * sc + (se & 3) + 8 * (se >> 2)) * rowB, hreg[srb][sc][se],
                            (uint32_t)(32 * sc + ((se + 1) & 3) + 8 * ((se + 1) >> 2)) * rowB, hreg[srb][sc][se + 1]);
        }
    }

#ifdef OS_FUSED_TS
    if (blockIdx.x == 0 && threadIdx.x == 0)
        printf("fused_v2 cycles per step: inputs %llu | features %llu | predict %llu | update %llu | mfma(4 passes) %llu | cell(4 passes) %llu | park %llu\n",
               ts_sum[1] / k.T, ts_sum[2] / k.T, ts_sum[3] / k.T, ts_sum[4] / k.T, ts_sum[5] / k.T, ts_sum[6] / k.T, ts_sum[7] / k.T);
#endif
    // ---- final state, status, h_T for the head kernel ----
    status |= singular_status(smin) | finite_status_p(X);
    if (live) {
        rsrc_t rx = make_rsrc(k.x, 12 * rowB), rP = make_rsrc(k.P, 144 * rowB);
#pragma unroll
        for (int i = 0; i < NS; i++) buf_store(rx, voff, i * rowB, OSF_X(i));
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int j = 0; j < NS; j++) buf_store(rP, voff, (i * NS + j) * rowB, OSK_SYM(U, i, j));
        k.status[b] = status;
    }
    if (!SEQOUT) {
#pragma unroll
        for (int rb = 0; rb < NRB; rb++) {
            const int tr = wbase + 32 * rb + li;
            if (tr < k.B) {
                rsrc_t rs = make_rsrc(a.h_last, (uint32_t)H * rowB);
                const uint32_t vo = (uint32_t)tr * 4u + (uint32_t)(4 * lh) * rowB;
#pragma unroll
                for (int c = 0; c < 2; c++)
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        buf_store(rs, vo, (uint32_t)(32 * c + (e & 3) + 8 * (e >> 2)) * rowB, agpr_get(hreg[rb][c][e]));
            }
        }
    }
}

// =====================================================================================================================
// fused_kf_gru_kernel_v3<QDIAG, NSPLIT> (round 6) -- the same path for batches that cannot fill the chip with 32 trajectories per
// wave: a 16-TRAJECTORY tile on v_mfma_f32_16x16x4_f32 (same flop rate as the 32x32x2 form: 8 passes for a quarter of the work).
//
// Lane l = (quarter q = l >> 4, trajectory lt = l & 15).  All four quarters run the filter of trajectory wbase + lt redundantly (the
// filter's cost per wave does not depend on how many lanes carry distinct trajectories).  gates^T tile = 16 units x 16 trajectories:
// A = weights (lane: unit row lt, k = q), B = activations (lane: k = q, trajectory lt), D: register v of lane l = unit 4 q + v of the
// tile -- so, as in v2, the result registers ARE the B operand of a recurrent k-step whose four k values are the units
// {16 ut + 4 q' + v : q' = 0..3}: h_t stays in 16 AGPRs.  The B fragment of a feature k-step (features 4 ks .. 4 ks + 3) is a select
// over the quarter (three v_cndmask) minus the quarter's minimum.
//   NSPLIT = 1: one wave per tile, all four unit tiles (372 MFMAs of 32 cycles per step), 64 trajectories per CU.
//   NSPLIT = 2 / 4: the tile's hidden units are split over 2 / 4 waves of the workgroup (each on its own SIMD): every wave repeats the
//     filter, computes 4 / NSPLIT unit tiles of the gates and their cell update, writes its slice of h_t to a parity-double-buffered
//     LDS block and picks the other slices up behind ONE workgroup barrier placed in the middle of the NEXT step (after the filter
//     and the feature k-steps: the waves' skew is absorbed there).  32 / 16 trajectories per CU.
//   A wave keeps its own slice in hreg slots 0 .. UTW-1 and the others' behind it (slot s = unit tile (sw UTW + s) % 4): the image
//   holds the recurrent fragments per wave position in that rotated k order, so every register index is a compile-time constant.
// LDS: the image (31 k-steps x 4 unit tiles x 64 lanes x (r, z, n, -) floats + biases + minima = 128,256 B) + the exchange block.
// =====================================================================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int KS3X = KX / 4, KS3H = H / 4, KS3 = KS3X + KS3H;            // 15 feature + 16 recurrent k-steps
constexpr int IMG3_W = KS3 * 4 * 256, IMG3_BIAS = IMG3_W, IMG3_MINS = IMG3_BIAS + 256, IMG3 = IMG3_MINS + 64;
constexpr size_t LDS3_IMG_BYTES = (size_t)IMG3 * sizeof(float);
template <int NSPLIT> constexpr size_t lds3_bytes() { return LDS3_IMG_BYTES + (NSPLIT > 1 ? (size_t)(4 / NSPLIT) * 2 * 4 * 64 * 16 : 0); }

__global__ void fused_pack3_kernel(const float *__restrict__ w0 /* layer 0, torch layout */, const float *__restrict__ minmax,
                                   float *__restrict__ img /* [IMG3] */, int nsplit)
{
    constexpr float LOG2E = 1.44269504088896341f;
    const float *Wih = w0, *Whh = Wih + 3 * H * KX, *bih = Whh + 3 * H * H, *bhh = bih + 3 * H;
    const int utw = 4 / nsplit;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < IMG3_W; i += gridDim.x * blockDim.x) {
        // [sw][ks][u][lane][4]
        const int g = i & 3, lane = (i >> 2) & 63, grp = i >> 8;
        const int u = grp % utw, ks = (grp / utw) % KS3, sw = grp / (utw * KS3);
        const int lt = lane & 15, q = lane >> 4, ut = sw * utw + u;
        float v = 0.f;
        if (g < 3) {
            const int row = g * H + 16 * ut + lt;
            const float gs = g < 2 ? -LOG2E : 2.0f * LOG2E;
            if (ks < KS3X) {
                const int k = 4 * ks + q;
                v = Wih[row * KX + k] * (1.0f / (minmax[KX + k] - minmax[k])) * gs;
            } else {
                const int kh = ks - KS3X, slot = kh >> 2, vv = kh & 3;
                const int uin = 16 * ((sw * utw + slot) & 3) + 4 * q + vv;
                v = Whh[row * H + uin] * gs;
            }
        }
        img[i] = v;
    }
    if (blockIdx.x == 0) {
        // biases [q][ut][g4][v] of unit 16 ut + 4 q + v; minima
        const int j = threadIdx.x, q = j >> 6, ut = (j >> 4) & 3, g4 = (j >> 2) & 3, vv = j & 3, u = 16 * ut + 4 * q + vv;
        float v;
        if (g4 == 0) v = (bih[u] + bhh[u]) * -LOG2E;
        else if (g4 == 1) v = (bih[H + u] + bhh[H + u]) * -LOG2E;
        else if (g4 == 2) v = bih[2 * H + u] * (2.0f * LOG2E);
        else v = bhh[2 * H + u] * (2.0f * LOG2E);
        img[IMG3_BIAS + j] = v;
        if (j < 64) img[IMG3_MINS + j] = j < KX ? minmax[j] : 0.f;
    }
}

// acc (4 VGPRs) += W-fragment (VGPR, from LDS) x B-fragment (AGPR); see mfma_va for the hazards inline assembly has to respect itself
__device__ __forceinline__ void mfma16_va(f32x4 &acc, float w, float b_agpr)
{
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "a"(b_agpr));
}

template <bool QDIAG, int NSPLIT>
__global__ __launch_bounds__(256, 1) void fused_kf_gru_kernel_v3(const FusedArgs a)
{
    constexpr int UTW = 4 / NSPLIT;            // unit tiles per wave
    constexpr int TPW = 4 / NSPLIT;            // trajectory tiles per workgroup
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lt = lane & 15, q = lane >> 4;
    const int sw = wave % NSPLIT, tw = wave / NSPLIT;
    const KfRunArgs &k = a.kf;
    const size_t B = (size_t)k.B;
    {
        const float4 *src = reinterpret_cast<const float4 *>(a.wpacked);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < IMG3 / 4; i += 256) dst[i] = src[i];
    }
    if (NSPLIT > 1)                  // step 0 picks "h_{-1}" = 0 up from parity 1 like any other step (no branch in the loop)
        for (int i = threadIdx.x; i < TPW * 2 * 4 * 64 * 4; i += 256) lds[IMG3 + i] = 0.f;
    __syncthreads();
    float minA[KS3X];                // the quarter's minimum of feature k-step ks (feature 4 ks + q), AGPR-resident
#pragma unroll
    for (int ks = 0; ks < KS3X; ks++) minA[ks] = agpr_put(lds[IMG3_MINS + 4 * ks + q]);
    const float4 *Wq = reinterpret_cast<const float4 *>(lds) + (size_t)sw * (KS3 * UTW * 64) + lane;      // this wave position's fragments
    const float4 *bias4 = reinterpret_cast<const float4 *>(lds + IMG3_BIAS) + q * 16 + sw * UTW * 4;
    float4 *xch = reinterpret_cast<float4 *>(lds + IMG3) + (size_t)tw * (2 * 4 * 64) + lane;   // [tile][parity][unit tile][lane] x 16 B

    const int wbase = (blockIdx.x * TPW + tw) * 16;                                // first trajectory of this wave's tile
    const int b = wbase + lt;
    const bool live = b < k.B && q == 0 && sw == 0;                                // the lanes that store
    const int bb = b < k.B ? b : k.B - 1;
    const uint32_t voff = (uint32_t)bb * 4u, rowB = (uint32_t)k.B * 4u;
    const bool q1 = (q & 1) != 0, q2 = (q & 2) != 0;

    f2 X[6];
    f2 U[NU];
    int status = 0;
    float smin = 3.0e38f;
    {
        rsrc_t rx = make_rsrc(k.x, 12 * rowB), rP = make_rsrc(k.P, 144 * rowB);
#pragma unroll
        for (int i = 0; i < 6; i++) X[i] = (f2){buf_load(rx, voff, 2 * i * rowB), buf_load(rx, voff, (2 * i + 1) * rowB)};
        status = p0_asymmetry_status([&](int e) { return buf_load(rP, voff, (uint32_t)e * rowB); });
        sym_load(U, [&](int e) { return buf_load(rP, voff, (uint32_t)e * rowB); });
    }
    // hreg[s][v] = h[unit 16 ((sw UTW + s) % 4) + 4 q + v][trajectory wbase + lt]  (AGPRs)
    float hreg[4][4];
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int v = 0; v < 4; v++) hreg[s][v] = agpr_put(0.f);                    // h0 = 0 (gru/gru_model.py:27)

    StepInP in;
    float acl[6];
    load_step_p(k, 0, voff, rowB, in);
    {
        rsrc_t ra = make_rsrc(k.accel, 6 * rowB);
#pragma unroll
        for (int i = 0; i < 6; i++) acl[i] = buf_load_nt(ra, voff, i * rowB);
    }

    OSF_TS_DECL
    for (int t = 0; t < k.T; t++) {
        OSF_TS(0)
        if (NSPLIT > 1) {
            // the other waves' slices of h_{t-1}: written at the end of step t - 1 into parity (t - 1) & 1 (zeros at t = 0).  The matrix
            // pipe is idle here anyway (the filter phase follows), and the waves arrive together: they all just finished the same step.
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const float4 *xr = xch + (size_t)((t + 1) & 1) * (4 * 64);
            float4 hv[NSPLIT > 1 ? 4 - UTW : 1];
#pragma unroll
            for (int s = UTW; s < 4; s++) hv[s - UTW] = xr[((sw * UTW + s) & 3) * 64];
#pragma unroll
            for (int s = UTW; s < 4; s++) {
                hreg[s][0] = agpr_put(hv[s - UTW].x); hreg[s][1] = agpr_put(hv[s - UTW].y);
                hreg[s][2] = agpr_put(hv[s - UTW].z); hreg[s][3] = agpr_put(hv[s - UTW].w);
            }
        }
        OSF_TS(7)
        float z[NM], FA[KS3X];
        f2 PW[2][3];
        float g9[9];
        status |= kf_step_inputs_sym(X, in, k.k, z, PW, g9);
        __builtin_amdgcn_sched_barrier(0);
        OSF_TS(1)
        // B fragment of feature k-step ks: the quarter's feature 4 ks + q minus its minimum (the 1/(max-min) scale sits in the weights)
        auto feat4 = [&](int ks, float v0, float v1, float v2, float v3) {
            const float lo = q1 ? v1 : v0, hi = q1 ? v3 : v2;
            FA[ks] = agpr_put((q2 ? hi : lo) - agpr_get(minA[ks]));
        };
        // feature order [x 0-11 | accel 12-17 | f 18-29 | p_world 30-41 | dp 42-53 | imu 54-59]
        feat4(3, acl[0], acl[1], acl[2], acl[3]);
        feat4(4, acl[4], acl[5], OSF_LEG(in.f, 0), OSF_LEG(in.f, 1));
        feat4(5, OSF_LEG(in.f, 2), OSF_LEG(in.f, 3), OSF_LEG(in.f, 4), OSF_LEG(in.f, 5));
        feat4(6, OSF_LEG(in.f, 6), OSF_LEG(in.f, 7), OSF_LEG(in.f, 8), OSF_LEG(in.f, 9));
        feat4(7, OSF_LEG(in.f, 10), OSF_LEG(in.f, 11), OSF_LEG(PW, 0), OSF_LEG(PW, 1));
        feat4(8, OSF_LEG(PW, 2), OSF_LEG(PW, 3), OSF_LEG(PW, 4), OSF_LEG(PW, 5));
        feat4(9, OSF_LEG(PW, 6), OSF_LEG(PW, 7), OSF_LEG(PW, 8), OSF_LEG(PW, 9));
        feat4(10, OSF_LEG(PW, 10), OSF_LEG(PW, 11), OSF_LEG(in.dp, 0), OSF_LEG(in.dp, 1));
        feat4(11, OSF_LEG(in.dp, 2), OSF_LEG(in.dp, 3), OSF_LEG(in.dp, 4), OSF_LEG(in.dp, 5));
        feat4(12, OSF_LEG(in.dp, 6), OSF_LEG(in.dp, 7), OSF_LEG(in.dp, 8), OSF_LEG(in.dp, 9));
        feat4(13, OSF_LEG(in.dp, 10), OSF_LEG(in.dp, 11), in.imu[0], in.imu[1]);
        feat4(14, in.imu[2], in.imu[3], in.imu[4], in.imu[5]);
        __builtin_amdgcn_sched_barrier(0);
        OSF_TS(2)
        cov_predict_sym_blk<QDIAG>(U, g9, k.k);
        __builtin_amdgcn_sched_barrier(0);
        OSF_TS(3)
        smin = fminf(smin, update_sequential_sym(X, U, z, k.k));
        __builtin_amdgcn_sched_barrier(0);
        feat4(0, OSF_X(0), OSF_X(1), OSF_X(2), OSF_X(3));
        feat4(1, OSF_X(4), OSF_X(5), OSF_X(6), OSF_X(7));
        feat4(2, OSF_X(8), OSF_X(9), OSF_X(10), OSF_X(11));
        OSF_TS(4)

        // ================= GRU cell: UTW unit tiles x (r, z, gi_n, gh_n) accumulators of 4 registers =================
        const StepSrc nsrc = step_src(k, (t + 1 < k.T) ? t + 1 : t, rowB);      // the next step's input rows (requested between the MFMA groups)
        f32x4 acc[UTW][4];
#pragma unroll
        for (int u = 0; u < UTW; u++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const float4 bv = bias4[u * 4 + g];
                acc[u][g] = (f32x4){bv.x, bv.y, bv.z, bv.w};
            }
        // groups gi = ks UTW + u: one ds_read_b128 (the three gates' fragments) and three MFMAs; fragments three groups deep, the read
        // of group gi + 2 goes out behind the SECOND MFMA of group gi (its buffer was last read by group gi - 1: see mfma_va's note on
        // when the matrix pipe reads its operands)
        constexpr int NG = KS3 * UTW;
        float4 wb[3];
        wb[0] = Wq[0];
        if (NG > 1) wb[1] = Wq[64];
#pragma unroll
        for (int ks = 0; ks < KS3; ks++)
#pragma unroll
        for (int u = 0; u < UTW; u++) {
            const int gi = ks * UTW + u, cur = gi % 3, nx2 = (gi + 2) % 3;
            // 12 x_out stores of this step (X does not change until the next step's filter phase; shadow lanes: an offset no descriptor
            // covers) and the NEXT step's 49 input loads, spread over the groups -- a VMEM instruction costs ~16 issue cycles at one
            // wave per SIMD, which the matrix pipe covers one at a time but not 49 in a row (round 6: in-kernel timestamps).  Every
            // input register of this step is dead by now: its features sit in AGPRs.
            constexpr int VPG = (NS + 49 + NG - 1) / NG;                // VMEM instructions per group
#pragma unroll
            for (int j = gi * VPG; j < (gi + 1) * VPG && j < NS + 49; j++) {
                if (j < NS) {
                    rsrc_t ro = make_rsrc(k.x_out + (size_t)t * 12 * B, 12 * rowB);
                    buf_store_nt(ro, live ? voff : 0x7ffffff0u, j * rowB, OSF_X(j < NS ? j : 0));
                } else {
                    prefetch_input(nsrc, voff, rowB, j - NS, in, acl);
                }
            }
            const float bv = ks < KS3X ? FA[ks < KS3X ? ks : 0] : hreg[(ks >= KS3X ? ks - KS3X : 0) >> 2][(ks >= KS3X ? ks - KS3X : 0) & 3];
            const int gn = ks < KS3X ? 2 : 3;          // input part feeds gi_n, recurrent part gh_n
            mfma16_va(acc[u][0], wb[cur].x, bv);
            mfma16_va(acc[u][1], wb[cur].y, bv);
            __builtin_amdgcn_sched_barrier(0);
            if (gi + 2 < NG) wb[nx2] = Wq[(gi + 2) * 64];
            __builtin_amdgcn_sched_barrier(0);
            mfma16_va(acc[u][gn], wb[cur].z, bv);
            __builtin_amdgcn_sched_barrier(0);
        }
        OSF_TS(5)
        {
            // 8-pass MFMA result -> VALU read (the 16-pass count of mfma_drain covers it)
#pragma unroll
            for (int u = 0; u < UTW; u++) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[u][0]), "+v"(acc[u][1]), "+v"(acc[u][2]), "+v"(acc[u][3]));
        }
        // cell update on the accumulators (scales folded into the weights: sigmoid = rcp(1 + exp2(a))), stage by stage
        {
            const f2 one = {1.0f, 1.0f};
            f2 R[UTW][2], Z[UTW][2], N[UTW][2], G[UTW][2];
#pragma unroll
            for (int u = 0; u < UTW; u++)
#pragma unroll
                for (int v = 0; v < 4; v++) { acc[u][0][v] = __builtin_amdgcn_exp2f(acc[u][0][v]); acc[u][1][v] = __builtin_amdgcn_exp2f(acc[u][1][v]); }
#pragma unroll
            for (int u = 0; u < UTW; u++)
#pragma unroll
                for (int p2 = 0; p2 < 2; p2++) {
                    R[u][p2] = (f2){acc[u][0][2 * p2], acc[u][0][2 * p2 + 1]} + one;
                    Z[u][p2] = (f2){acc[u][1][2 * p2], acc[u][1][2 * p2 + 1]} + one;
                }
#pragma unroll
            for (int u = 0; u < UTW; u++)
#pragma unroll
                for (int p2 = 0; p2 < 2; p2++) {
                    R[u][p2] = (f2){__builtin_amdgcn_rcpf(R[u][p2][0]), __builtin_amdgcn_rcpf(R[u][p2][1])};
                    Z[u][p2] = (f2){__builtin_amdgcn_rcpf(Z[u][p2][0]), __builtin_amdgcn_rcpf(Z[u][p2][1])};
                }
#pragma unroll
            for (int u = 0; u < UTW; u++)
#pragma unroll
                for (int p2 = 0; p2 < 2; p2++) {
                    const f2 gin = {acc[u][2][2 * p2], acc[u][2][2 * p2 + 1]}, ghn = {acc[u][3][2 * p2], acc[u][3][2 * p2 + 1]};
                    N[u][p2] = fma2(R[u][p2], ghn, gin);                                  // gi_n + r gh_n (pre-scaled)
                    N[u][p2] = (f2){__builtin_amdgcn_exp2f(N[u][p2][0]), __builtin_amdgcn_exp2f(N[u][p2][1])} + one;
                }
#pragma unroll
            for (int u = 0; u < UTW; u++)
#pragma unroll
                for (int p2 = 0; p2 < 2; p2++) {
                    N[u][p2] = (f2){__builtin_amdgcn_rcpf(N[u][p2][0]), __builtin_amdgcn_rcpf(N[u][p2][1])};
                    N[u][p2] = fma2((f2){-2.0f, -2.0f}, N[u][p2], one);                    // n = tanh(.)
                    G[u][p2] = (f2){agpr_get(hreg[u][2 * p2]), agpr_get(hreg[u][2 * p2 + 1])};   // h_{t-1} of the wave's own units
                    G[u][p2] = fma2(Z[u][p2], G[u][p2] - N[u][p2], N[u][p2]);             // (1 - z) n + z h
                }
            if (NSPLIT > 1) {
                float4 *xw = xch + (size_t)(t & 1) * (4 * 64);
#pragma unroll
                for (int u = 0; u < UTW; u++) xw[(sw * UTW + u) * 64] = make_float4(G[u][0][0], G[u][0][1], G[u][1][0], G[u][1][1]);
            }
#pragma unroll
            for (int u = 0; u < UTW; u++)
#pragma unroll
                for (int p2 = 0; p2 < 2; p2++) { hreg[u][2 * p2] = agpr_put(G[u][p2][0]); hreg[u][2 * p2 + 1] = agpr_put(G[u][p2][1]); }
        }
        OSF_TS(6)
    }
#ifdef OS_FUSED_TS
    if (blockIdx.x == 0 && threadIdx.x == 0)
        printf("fused_v3<NSPLIT=%d> cycles per step: barrier + exchange %llu | inputs %llu | features %llu | predict %llu | update %llu | mfma %llu | cell %llu\n", NSPLIT,
               ts_sum[7] / k.T, ts_sum[1] / k.T, ts_sum[2] / k.T, ts_sum[3] / k.T, ts_sum[4] / k.T, ts_sum[5] / k.T, ts_sum[6] / k.T);
#endif
    status |= singular_status(smin) | finite_status_p(X);
    if (live) {
        rsrc_t rx = make_rsrc(k.x, 12 * rowB), rP = make_rsrc(k.P, 144 * rowB);
#pragma unroll
        for (int i = 0; i < NS; i++) buf_store(rx, voff, i * rowB, OSF_X(i));
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int j = 0; j < NS; j++) buf_store(rP, voff, (i * NS + j) * rowB, OSK_SYM(U, i, j));
        k.status[b] = status;
    }
    if (NSPLIT > 1) {
        // h_T of the other waves' units (the loop picks a step's slices up in the NEXT step)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const float4 *xr = xch + (size_t)((k.T - 1) & 1) * (4 * 64);
#pragma unroll
        for (int s = UTW; s < 4; s++) {
            const float4 hv = xr[((sw * UTW + s) & 3) * 64];
            hreg[s][0] = agpr_put(hv.x); hreg[s][1] = agpr_put(hv.y); hreg[s][2] = agpr_put(hv.z); hreg[s][3] = agpr_put(hv.w);
        }
    }
    if (sw == 0 && b < k.B) {
        // h_T [64][B] for the head launch: wave position 0 holds unit tile s in slot s
        rsrc_t rs = make_rsrc(a.h_last, (uint32_t)H * rowB);
        const uint32_t vo = (uint32_t)b * 4u + (uint32_t)(4 * q) * rowB;
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int v = 0; v < 4; v++) buf_store(rs, vo, (uint32_t)(16 * s + v) * rowB, agpr_get(hreg[s][v]));
    }
}

// =====================================================================================================================
// fused_kf_gru_bf16_kernel<QDIAG, SPL> -- OPT-IN reduced-precision gate GEMM (OS_FUSED_SPLIT_BF16), never the default.
//
// fp32 MFMA runs at 1/16 of the bf16 rate on gfx950.  Here every fp32 operand of the gate GEMM is split into SPL bf16 terms
// (v = hi + mid + lo, 8 mantissa bits each: SPL = 3 represents an fp32 value exactly, SPL = 2 keeps 16 bits) and the products
// are formed on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: SPL = 3 issues hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid
// (every term down to 2^-16 of the product; what is dropped is below fp32 rounding), SPL = 2 issues hi.hi, hi.lo, lo.hi
// (relative 2^-16 per product).  576 (288) matrix instructions of 32 cycles per step instead of 744 of 64.  The weights are
// split once per call into the LDS image, the activations on the VALU just before use (5.5 / 3 instructions per value): the
// kernel is VALU-bound; the Kalman step and the cell update are the fp32 code of v2.  Same transposed layout: trajectory on
// the lane, h in AGPRs; a k-block is 16 features, or 16 hidden units in the order the accumulators hold them.
// LDS: [2 chunks][8 k-blocks][3 gates][SPL terms][64 lanes][4 dwords of bf16 pairs] + biases + minima
//      = 148,736 B (SPL = 3) / 99,584 B (SPL = 2).
// =====================================================================================================================
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int KBX = 4, KBH = 4, KBT = KBX + KBH;
template <int SPL> struct Bf16Img {
    static constexpr int CH = KBT * 3 * SPL * 256;                // dwords per chunk of weight fragments
    static constexpr int BIAS = 2 * CH;                           // [2 chunks][2 lane halves][4 gate slots][16] floats
    static constexpr int MINS = BIAS + 256;
    static constexpr int TOTAL = MINS + 64;
    static constexpr size_t BYTES = (size_t)TOTAL * 4;
};

using osg::bf16_hi_f32; using osg::bf16_lo_f32; using osg::pack_bf16; using osg::split_pair;   // gru_common.hpp

template <int SPL>
__global__ void fused_pack_bf16_kernel(const float *__restrict__ w0, const float *__restrict__ minmax, uint32_t *__restrict__ img)
{
    using I = Bf16Img<SPL>;
    constexpr float LOG2E = 1.44269504088896341f;
    const float *Wih = w0, *Whh = Wih + 3 * H * KX, *bih = Whh + 3 * H * H, *bhh = bih + 3 * H;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * I::CH / SPL; i += gridDim.x * blockDim.x) {
        // one thread per (chunk, k-block, gate, lane, dword): the SPL term dwords of two adjacent k values
        int r = i;
        const int jj = r & 3; r >>= 2;
        const int lane = r & 63; r >>= 6;
        const int g = r % 3; r /= 3;
        const int kb = r % KBT, c = r / KBT;
        const int li = lane & 31, lh = lane >> 5, col = g * H + 32 * c + li;
        const float gs = g < 2 ? -LOG2E : 2.0f * LOG2E;
        float w[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int j = 2 * jj + q;
            if (kb < KBX) {
                const int kk = 16 * kb + 8 * lh + j;
                w[q] = kk < KX ? Wih[col * KX + kk] * (1.0f / (minmax[KX + kk] - minmax[kk])) * gs : 0.f;
            } else {
                const int kh = kb - KBX, cp = kh >> 1, e = 8 * (kh & 1) + j;
                w[q] = Whh[col * H + 32 * cp + (e & 3) + 8 * (e >> 2) + 4 * lh] * gs;
            }
        }
        uint32_t t[SPL];
        split_pair<SPL>(w[0], w[1], t);
#pragma unroll
        for (int sp = 0; sp < SPL; sp++) img[c * I::CH + (((kb * 3 + g) * SPL + sp) * 64 + lane) * 4 + jj] = t[sp];
    }
    if (blockIdx.x == 0) {
        float *fb = reinterpret_cast<float *>(img);
        const int j = threadIdx.x;                     // 256 threads: [c][lh][g4][e]
        const int lh = (j >> 6) & 1, g4 = (j >> 4) & 3, e = j & 15, u = 32 * (j >> 7) + (e & 3) + 8 * (e >> 2) + 4 * lh;
        float v;
        if (g4 == 0) v = (bih[u] + bhh[u]) * -LOG2E;
        else if (g4 == 1) v = (bih[H + u] + bhh[H + u]) * -LOG2E;
        else if (g4 == 2) v = bih[2 * H + u] * (2.0f * LOG2E);
        else v = bhh[2 * H + u] * (2.0f * LOG2E);
        fb[I::BIAS + j] = v;
        if (j < 64) fb[I::MINS + j] = j < KX ? minmax[j] : 0.f;
    }
}

// acc (VGPRs) += W-fragment x B-fragment, both 8 bf16 per lane in VGPRs.  Inline asm for the same reason as mfma_va: the
// accumulators must stay in VGPRs for the cell update.  MFMAs on one accumulator are at least two instructions apart.
__device__ __forceinline__ void mfma_bf16_vv(f32x16 &acc, i32x4 w, i32x4 b)
{
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(b));
}

template <bool QDIAG, int SPL>
__global__ __launch_bounds__(256, 1) void fused_kf_gru_bf16_kernel(const FusedArgs a)
{
    using I = Bf16Img<SPL>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const KfRunArgs &k = a.kf;
    const size_t B = (size_t)k.B;
    {
        const float4 *src = reinterpret_cast<const float4 *>(a.wpacked);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < I::TOTAL / 4; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const float4 *mins4 = reinterpret_cast<const float4 *>(lds + I::MINS);
    const i32x4 *Wl = reinterpret_cast<const i32x4 *>(lds) + lane;

    const int wbase = blockIdx.x * 256 + (threadIdx.x & ~63);
    const int b = wbase + lane;
    const bool live = b < k.B;
    const int bb = live ? b : k.B - 1;
    const uint32_t voff = (uint32_t)bb * 4u, rowB = (uint32_t)k.B * 4u;

    f2 X[6];
    f2 U[NU];
    int status = 0;
    float smin = 3.0e38f;                  // the smallest innovation variance of the run (status bit 0)
    {
        rsrc_t rx = make_rsrc(k.x, 12 * rowB), rP = make_rsrc(k.P, 144 * rowB);
#pragma unroll
        for (int i = 0; i < 6; i++) X[i] = (f2){buf_load(rx, voff, 2 * i * rowB), buf_load(rx, voff, (2 * i + 1) * rowB)};
        status = p0_asymmetry_status([&](int e) { return buf_load(rP, voff, (uint32_t)e * rowB); });
        sym_load(U, [&](int e) { return buf_load(rP, voff, (uint32_t)e * rowB); });
    }
    float hreg[2][2][16];               // fp32 h_t (AGPR-resident), layout as in v2
#pragma unroll
    for (int rb = 0; rb < 2; rb++)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int e = 0; e < 16; e++) hreg[rb][c][e] = agpr_put(0.f);

    StepInP in;
    float acl[6];
    load_step_p(k, 0, voff, rowB, in);
    {
        rsrc_t ra = make_rsrc(k.accel, 6 * rowB);
#pragma unroll
        for (int i = 0; i < 6; i++) acl[i] = buf_load_nt(ra, voff, i * rowB);
    }

    OSF_TS_DECL
    for (int t = 0; t < k.T; t++) {
        OSF_TS(0)
        float z[NM], FA[2][KBX][8];        // FA[rb][kb][j]: feature 16 kb + 8 (lane half) + j of trajectory block rb (AGPRs)
        f2 PW[2][3];
        float g9[9];
        status |= kf_step_inputs_sym(X, in, k.k, z, PW, g9);
        __builtin_amdgcn_sched_barrier(0);
        OSF_TS(1)                                        // wait for the prefetched inputs + rotations, odometry, next_state
        // k-block kb = features 16 kb .. 16 kb + 15; v_permlane32_swap pairs feature j with feature j + 8 of the block, so that
        // the first register serves trajectories 0-31 (lanes 0-31: feature j, lanes 32-63: feature j + 8), the second 32-63
        auto feat8 = [&](int kb, float l0, float l1, float l2, float l3, float l4, float l5, float l6, float l7, float h0, float h1,
                         float h2, float h3, float h4, float h5, float h6, float h7) {
            float lo[8] = {l0, l1, l2, l3, l4, l5, l6, l7}, hi[8] = {h0, h1, h2, h3, h4, h5, h6, h7};
#pragma unroll
            for (int i4 = 0; i4 < 2; i4++) {
                const float4 ml = mins4[4 * kb + i4], mh = mins4[4 * kb + 2 + i4];
                lo[4 * i4] -= ml.x; lo[4 * i4 + 1] -= ml.y; lo[4 * i4 + 2] -= ml.z; lo[4 * i4 + 3] -= ml.w;
                hi[4 * i4] -= mh.x; hi[4 * i4 + 1] -= mh.y; hi[4 * i4 + 2] -= mh.z; hi[4 * i4 + 3] -= mh.w;
            }
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %8\n\tv_permlane32_swap_b32 %1, %9\n\tv_permlane32_swap_b32 %2, %10\n\t"
                         "v_permlane32_swap_b32 %3, %11\n\tv_permlane32_swap_b32 %4, %12\n\tv_permlane32_swap_b32 %5, %13\n\t"
                         "v_permlane32_swap_b32 %6, %14\n\tv_permlane32_swap_b32 %7, %15\n\ts_nop 1"
                         : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(lo[4]), "+v"(lo[5]), "+v"(lo[6]), "+v"(lo[7]),
                           "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]), "+v"(hi[4]), "+v"(hi[5]), "+v"(hi[6]), "+v"(hi[7]));
#pragma unroll
            for (int j = 0; j < 8; j++) { FA[0][kb][j] = agpr_put(lo[j]); FA[1][kb][j] = agpr_put(hi[j]); }
        };
        // feature order [x 0-11 | accel 12-17 | f 18-29 | p_world 30-41 | dp 42-53 | imu 54-59 | 60-63 zero padding]
        feat8(1, acl[4], acl[5], OSF_LEG(in.f, 0), OSF_LEG(in.f, 1), OSF_LEG(in.f, 2), OSF_LEG(in.f, 3), OSF_LEG(in.f, 4), OSF_LEG(in.f, 5), OSF_LEG(in.f, 6), OSF_LEG(in.f, 7), OSF_LEG(in.f, 8), OSF_LEG(in.f, 9), OSF_LEG(in.f, 10),
              OSF_LEG(in.f, 11), OSF_LEG(PW, 0), OSF_LEG(PW, 1));
        feat8(2, OSF_LEG(PW, 2), OSF_LEG(PW, 3), OSF_LEG(PW, 4), OSF_LEG(PW, 5), OSF_LEG(PW, 6), OSF_LEG(PW, 7), OSF_LEG(PW, 8), OSF_LEG(PW, 9), OSF_LEG(PW, 10), OSF_LEG(PW, 11), OSF_LEG(in.dp, 0), OSF_LEG(in.dp, 1), OSF_LEG(in.dp, 2), OSF_LEG(in.dp, 3),
              OSF_LEG(in.dp, 4), OSF_LEG(in.dp, 5));
        feat8(3, OSF_LEG(in.dp, 6), OSF_LEG(in.dp, 7), OSF_LEG(in.dp, 8), OSF_LEG(in.dp, 9), OSF_LEG(in.dp, 10), OSF_LEG(in.dp, 11), in.imu[0], in.imu[1], in.imu[2], in.imu[3], in.imu[4],
              in.imu[5], 0.f, 0.f, 0.f, 0.f);
        __builtin_amdgcn_sched_barrier(0);          // the inputs are in AGPRs now: the covariance work below starts with their registers free
        OSF_TS(2)                                        // 48 raw features: subtract, swap, AGPR
        cov_predict_sym_blk<QDIAG>(U, g9, k.k);
        __builtin_amdgcn_sched_barrier(0);
        OSF_TS(3)                                        // covariance predict
        smin = fminf(smin, update_sequential_sym(X, U, z, k.k));
        {
            // shadow lanes: an offset no descriptor covers (the range check drops the store) instead of a branch around the stores
            const uint32_t vst = live ? voff : 0x7ffffff0u;
            rsrc_t ro = make_rsrc(k.x_out + (size_t)t * 12 * B, 12 * rowB);
#pragma unroll
            for (int i = 0; i < NS; i++) buf_store_nt(ro, vst, i * rowB, OSF_X(i));
        }
        feat8(0, OSF_X(0), OSF_X(1), OSF_X(2), OSF_X(3), OSF_X(4), OSF_X(5), OSF_X(6), OSF_X(7), OSF_X(8), OSF_X(9), OSF_X(10), OSF_X(11), acl[0], acl[1], acl[2], acl[3]);

        OSF_TS(4)                                        // ten measurement updates, x_out stores, 12 state features
        // ================= GRU cell: one 32-trajectory block at a time, both unit chunks together =================
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
            f32x16 acc[2][4];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const float4 *bc = reinterpret_cast<const float4 *>(lds + I::BIAS + c * 128 + lh * 64);
#pragma unroll
                for (int g = 0; g < 4; g++)
#pragma unroll
                    for (int v4 = 0; v4 < 4; v4++) {
                        const float4 bv = bc[g * 4 + v4];
                        acc[c][g][4 * v4] = bv.x; acc[c][g][4 * v4 + 1] = bv.y; acc[c][g][4 * v4 + 2] = bv.z; acc[c][g][4 * v4 + 3] = bv.w;
                    }
            }
            // weight fragments of (k-block, gate): 2 chunks x SPL terms, double-buffered one (k-block, gate) ahead
            i32x4 W[2][2][SPL];
            auto loadW = [&](int buf, int kb, int g) {
#pragma unroll
                for (int c = 0; c < 2; c++)
#pragma unroll
                    for (int sp = 0; sp < SPL; sp++) W[buf][c][sp] = Wl[(c * I::CH) / 4 + ((kb * 3 + g) * SPL + sp) * 64];
            };
            loadW(0, 0, 0);
#pragma unroll
            for (int kb = 0; kb < KBT; kb++) {
                // the 8 fp32 B values of this lane for the k-block, split into SPL bf16 fragments just before use
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; j++)
                    v[j] = agpr_get(kb < KBX ? FA[rb][kb < KBX ? kb : 0][j] : hreg[rb][(kb - KBX) >> 1 & 1][8 * ((kb - KBX) & 1) + j]);
                i32x4 Bf[SPL];
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    uint32_t tt[SPL];
                    split_pair<SPL>(v[2 * jj], v[2 * jj + 1], tt);
#pragma unroll
                    for (int sp = 0; sp < SPL; sp++) Bf[sp][jj] = (int)tt[sp];
                }
#pragma unroll
                for (int g = 0; g < 3; g++) {
                    const int u = kb * 3 + g, cur = u & 1;
                    const int gn = g < 2 ? g : (kb < KBX ? 2 : 3);        // input part feeds gi_n, recurrent part gh_n
                    // (weight term, activation term), largest first; consecutive MFMAs alternate between the two chunks
                    constexpr int NP = SPL == 3 ? 6 : 3;
                    constexpr int PW[6] = {0, 0, 1, 0, 2, 1}, PB[6] = {0, 1, 0, 2, 0, 1};
                    constexpr int PW2[3] = {0, 0, 1}, PB2[3] = {0, 1, 0};
#pragma unroll
                    for (int pi = 0; pi < NP; pi++) {
                        const int wt = SPL == 3 ? PW[pi] : PW2[pi], bt = SPL == 3 ? PB[pi] : PB2[pi];
                        mfma_bf16_vv(acc[0][gn], W[cur][0][wt], Bf[bt]);
                        mfma_bf16_vv(acc[1][gn], W[cur][1][wt], Bf[bt]);
                        // The next (k-block, gate)'s fragments are requested only AFTER this group's first MFMA pair: the matrix pipe
                        // reads an MFMA's A/B registers when the instruction starts, not when it issues, and a ds_read that returns
                        // into the registers of the previous group's last (still queued) MFMAs corrupted them -- measured with the
                        // two-term split, whose groups are half as long (tools/bf16_determinism.py: hundreds of trajectories differed
                        // from run to run; none since).  hipcc cannot see this hazard inside inline asm.
                        if (pi == 0 && u + 1 < KBT * 3) { __builtin_amdgcn_sched_barrier(0); loadW(cur ^ 1, (u + 1) / 3, (u + 1) % 3); __builtin_amdgcn_sched_barrier(0); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][0]),
                         "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]));
            OSF_TS(5)                                    // bias init + activation split + MFMAs + drain (x 2 blocks)
            if (rb == 1) {
                const int tn = (t + 1 < k.T) ? t + 1 : t;
                load_step_p(k, tn, voff, rowB, in);
                rsrc_t ra = make_rsrc(k.accel + (size_t)tn * 6 * B, 6 * rowB);
#pragma unroll
                for (int i = 0; i < 6; i++) acl[i] = buf_load_nt(ra, voff, i * rowB);
            }
            // cell update (fp32, as v2): every MFMA of this block is done, so h_t goes straight into its registers
#pragma unroll
            for (int c = 0; c < 2; c++) {
                f2 *A0 = reinterpret_cast<f2 *>(&acc[c][0]), *A1 = reinterpret_cast<f2 *>(&acc[c][1]), *A2 = reinterpret_cast<f2 *>(&acc[c][2]),
                   *A3 = reinterpret_cast<f2 *>(&acc[c][3]);
                const f2 one = {1.0f, 1.0f};
#pragma unroll
                for (int e = 0; e < 16; e++) { acc[c][0][e] = __builtin_amdgcn_exp2f(acc[c][0][e]); acc[c][1][e] = __builtin_amdgcn_exp2f(acc[c][1][e]); }
#pragma unroll
                for (int p2 = 0; p2 < 8; p2++) { A0[p2] += one; A1[p2] += one; }
#pragma unroll
                for (int e = 0; e < 16; e++) { acc[c][0][e] = __builtin_amdgcn_rcpf(acc[c][0][e]); acc[c][1][e] = __builtin_amdgcn_rcpf(acc[c][1][e]); }
#pragma unroll
                for (int p2 = 0; p2 < 8; p2++) A2[p2] = fma2(A0[p2], A3[p2], A2[p2]);
#pragma unroll
                for (int e = 0; e < 16; e++) acc[c][2][e] = __builtin_amdgcn_exp2f(acc[c][2][e]);
#pragma unroll
                for (int p2 = 0; p2 < 8; p2++) A2[p2] += one;
#pragma unroll
                for (int e = 0; e < 16; e++) { acc[c][2][e] = __builtin_amdgcn_rcpf(acc[c][2][e]); acc[c][3][e] = agpr_get(hreg[rb][c][e]); }
#pragma unroll
                for (int p2 = 0; p2 < 8; p2++) A2[p2] = fma2((f2){-2.0f, -2.0f}, A2[p2], one);
#pragma unroll
                for (int p2 = 0; p2 < 8; p2++) A3[p2] = A3[p2] - A2[p2];
#pragma unroll
                for (int p2 = 0; p2 < 8; p2++) A3[p2] = fma2(A1[p2], A3[p2], A2[p2]);
#pragma unroll
                for (int e = 0; e < 16; e++) hreg[rb][c][e] = agpr_put(acc[c][3][e]);
            }
            OSF_TS(6)                                    // next step's input request (block 1) + cell update (x 2 blocks)
        }
    }
#ifdef OS_FUSED_TS
    if (blockIdx.x == 0 && threadIdx.x == 0)
        printf("fused_bf16<SPL=%d> cycles per step: inputs %llu | features %llu | predict %llu | update %llu | split + mfma (2 blocks) %llu | cell (2 blocks) %llu | sum %llu\n",
               SPL, ts_sum[1] / k.T, ts_sum[2] / k.T, ts_sum[3] / k.T, ts_sum[4] / k.T, ts_sum[5] / k.T, ts_sum[6] / k.T,
               (ts_sum[1] + ts_sum[2] + ts_sum[3] + ts_sum[4] + ts_sum[5] + ts_sum[6]) / k.T);
#endif

    status |= singular_status(smin) | finite_status_p(X);
    if (live) {
        rsrc_t rx = make_rsrc(k.x, 12 * rowB), rP = make_rsrc(k.P, 144 * rowB);
#pragma unroll
        for (int i = 0; i < NS; i++) buf_store(rx, voff, i * rowB, OSF_X(i));
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int j = 0; j < NS; j++) buf_store(rP, voff, (i * NS + j) * rowB, OSK_SYM(U, i, j));
        k.status[b] = status;
    }
#pragma unroll
    for (int rb = 0; rb < 2; rb++) {
        const int tr = wbase + 32 * rb + li;
        if (tr < k.B) {
            rsrc_t rs = make_rsrc(a.h_last, (uint32_t)H * rowB);
            const uint32_t vo = (uint32_t)tr * 4u + (uint32_t)(4 * lh) * rowB;
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int e = 0; e < 16; e++)
                    buf_store(rs, vo, (uint32_t)(32 * c + (e & 3) + 8 * (e >> 2)) * rowB, agpr_get(hreg[rb][c][e]));
        }
    }
}

}  // namespace osf

int os_kf_run_impl(os_ctx *ctx, osk::KfRunArgs &a, uint32_t flags, hipStream_t s);   // kf_kernels.hip
int os_gru_scratch(os_ctx *ctx, int B, int T, float **seq0, float **seq1, float **hlast);   // gru_kernels.hip
int os_gru_layers_impl(os_ctx *ctx, int B, int T, const float *in, int first_layer, float *out, float *h_last_all,
                       hipStream_t s);
int os_gru_head_launch(os_ctx *ctx, int B, const float *top, const float *fcw, float *out, hipStream_t s);

extern "C" {

int os_fused_set_tile(os_ctx *ctx, int32_t tile)
{
    OS_CHECK_CTX(ctx);
    if (tile != 0 && tile != 256 && tile != 128 && tile != 64 && tile != 32 && tile != 16)
        return os_fail(ctx, -2, "os_fused_set_tile: 0 (automatic) or 256 | 128 | 64 | 32 | 16 trajectories per workgroup");
    ctx->tune_fused_tile = tile;
    return 0;
}

int os_fused_run(os_ctx *ctx, int32_t B, int32_t T, const float *p, const float *f, const float *dp, const float *imu,
                 const uint32_t *contact, const float *accel, const float *body_ref, const float *latent,
                 int32_t n_latent, const float *minmax, float *x, float *P, float *x_out, float *out, int32_t *status,
                 uint32_t flags, void *stream)
{
    OS_CHECK_CTX(ctx);
    if (B <= 0 || T <= 0) return os_fail(ctx, -2, "os_fused_run: B and T must be positive");
    if (!p || !f || !dp || !imu || !contact || !accel || !minmax || !x || !P || !x_out || !out || !status)
        return os_fail(ctx, -2, "os_fused_run: null required pointer");
    if (!ctx->gru_loaded) return os_fail(ctx, -5, "os_fused_run: call os_gru_load first");
    if (n_latent < 0 || (n_latent > 0 && !latent)) return os_fail(ctx, -2, "os_fused_run: bad latent");
    const int I = 60 + n_latent;
    if (ctx->gru.input_size != I) return os_fail(ctx, -4, "os_fused_run: GRU input_size must be 60 + n_latent");
    if ((size_t)B * 144 * 4 >= 0xffffffffull) return os_fail(ctx, -2, "os_fused_run: B too large for 32-bit buffer offsets");
    OS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    const os_gru_dims &d = ctx->gru;
    osk::KfRunArgs a;
    a.B = B; a.T = T; a.p = p; a.f = f; a.dp = dp; a.imu = imu; a.contact = contact; a.body_ref = body_ref;
    a.x = x; a.P = P; a.x_out = x_out; a.p_rot_out = nullptr; a.ptrace_out = nullptr; a.kgain_out = nullptr;
    a.status = status; a.accel = accel; a.minmax = minmax; a.feat_out = nullptr; a.feat_I = I;

    // Round 6: the single kernel exists for five tile shapes (trajectories per workgroup of four waves = per CU: the LDS image allows
    // one workgroup per CU): 256 / 128 (v2: 64 / 32 per wave, 32x32x2 MFMA), 64 (v3: 16 per wave, 16x16x4 MFMA), 32 / 16 (v3 with the
    // tile's hidden units split over 2 / 4 waves).  A workgroup's time per step does not depend on the batch, so the shape is chosen
    // to minimise rounds x cost with the measured cycles per step below (profiles/r06_shard_sweep.md); the two-kernel path (Kalman
    // kernel + layer kernel) remains for shapes the single kernel does not cover and behind OS_FUSED_TWO_KERNEL.
    const bool shapes_ok = (flags & OS_KF_SEQUENTIAL_UPDATE) && (flags & OS_KF_SYMMETRIC_P) && !(flags & OS_KF_DENSE_FD) &&
                           ctx->r_is_diagonal && n_latent == 0 && d.hidden_size == 64 && d.input_size == 60;
    int tile = 0;                      // trajectories per workgroup, 0 = two-kernel path
    if (shapes_ok && !(flags & OS_FUSED_TWO_KERNEL)) {
        // cost = measured microseconds per time step of one workgroup (T = 100 sweeps on MI355X, profiles/r06_shard_sweep.md)
        static const struct { int tpw; float cost; } shapes[5] = {{256, 25.9f}, {128, 14.4f}, {64, 9.0f}, {32, 5.7f}, {16, 4.4f}};
        const int nshape = d.num_layers > 1 ? 2 : 5;          // the v3 shapes have no layer-0 sequence output
        float best = 0.f;
        for (int i = 0; i < nshape; i++) {
            const int nwg = (B + shapes[i].tpw - 1) / shapes[i].tpw, rounds = (nwg + ctx->cu_count - 1) / ctx->cu_count;
            const float c = rounds * shapes[i].cost;
            if (!tile || c < best) { tile = shapes[i].tpw; best = c; }
        }
        if (ctx->tune_fused_tile > 0) {
            const int want = ctx->tune_fused_tile;
            if (want == 256 || want == 128 || (d.num_layers == 1 && (want == 64 || want == 32 || want == 16))) tile = want;
        }
        // B <= 80 CUs used to go to the two-kernel path (round 5 and before); the small tiles are faster there now, but a deeper stack
        // at a small batch still prefers its layer kernels' own input handling unless the caller insists
        if (d.num_layers > 1 && !(flags & OS_FUSED_ONE_KERNEL) && ctx->tune_fused_tile <= 0 && B <= 80 * ctx->cu_count) tile = 0;
    }
    const bool single_kernel = tile != 0;
    if (flags & (OS_FUSED_SPLIT_BF16 | OS_FUSED_SPLIT_BF16_2)) {
        // opt-in reduced-precision gate GEMM (never chosen by default): bf16 split terms on the bf16 MFMA, fp32 accumulate
        if (!shapes_ok || d.num_layers != 1)
            return os_fail(ctx, -4, "os_fused_run: OS_FUSED_SPLIT_BF16 needs the single-kernel shapes (60 features, hidden 64, one layer, "
                                    "diagonal R, sequential + symmetric flags)");
        const int spl = (flags & OS_FUSED_SPLIT_BF16_2) ? 2 : 3;
        const size_t bytes = spl == 3 ? osf::Bf16Img<3>::BYTES : osf::Bf16Img<2>::BYTES;
        osf::FusedArgs fa;
        fa.kf = a; fa.kf.k = ctx->k;
        fa.fcw = nullptr; fa.fcb = nullptr; fa.C = d.num_classes; fa.use_sigmoid = d.use_sigmoid; fa.out = out; fa.seq_out = nullptr;
        float *seq0 = nullptr, *seq1 = nullptr, *hlast = nullptr;
        if (os_gru_scratch(ctx, B, T, &seq0, &seq1, &hlast)) return -10;
        fa.h_last = hlast; fa.nrm = nullptr;
        if (!ctx->fusedbf_attr_set) {
            const void *fns[4] = {(const void *)osf::fused_kf_gru_bf16_kernel<true, 3>, (const void *)osf::fused_kf_gru_bf16_kernel<false, 3>,
                                  (const void *)osf::fused_kf_gru_bf16_kernel<true, 2>, (const void *)osf::fused_kf_gru_bf16_kernel<false, 2>};
            for (const void *fn : fns)
                OS_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)osf::Bf16Img<3>::BYTES));
            ctx->fusedbf_attr_set = true;
        }
        if (!ctx->fused_img_bf) OS_HIP(ctx, hipMalloc((void **)&ctx->fused_img_bf, osf::Bf16Img<3>::BYTES));
        if (spl == 3) hipLaunchKernelGGL(osf::fused_pack_bf16_kernel<3>, dim3(64), dim3(256), 0, s, ctx->gru_flat, minmax, (uint32_t *)ctx->fused_img_bf);
        else hipLaunchKernelGGL(osf::fused_pack_bf16_kernel<2>, dim3(64), dim3(256), 0, s, ctx->gru_flat, minmax, (uint32_t *)ctx->fused_img_bf);
        fa.wpacked = ctx->fused_img_bf;
        dim3 grid((B + 255) / 256), block(256);
        const int slot = os_prof_begin(ctx, OS_PHASE_FUSED, s, spl == 3 ? "fused_kf_gru_bf16_kernel<3>" : "fused_kf_gru_bf16_kernel<2>");
        const bool qd = ctx->q_is_diagonal;
        if (spl == 3 && qd) hipLaunchKernelGGL((osf::fused_kf_gru_bf16_kernel<true, 3>), grid, block, bytes, s, fa);
        else if (spl == 3) hipLaunchKernelGGL((osf::fused_kf_gru_bf16_kernel<false, 3>), grid, block, bytes, s, fa);
        else if (qd) hipLaunchKernelGGL((osf::fused_kf_gru_bf16_kernel<true, 2>), grid, block, bytes, s, fa);
        else hipLaunchKernelGGL((osf::fused_kf_gru_bf16_kernel<false, 2>), grid, block, bytes, s, fa);
        os_prof_end(ctx, slot, s);
        OS_HIP(ctx, hipGetLastError());
        const float *fcw = ctx->gru_flat + (os_gru_param_count(&d) - ((size_t)d.num_classes * 64 + d.num_classes));
        return os_gru_head_launch(ctx, B, hlast, fcw, out, s);
    }
    if (single_kernel) {
        // transposed GRU cell, h in registers, scales folded into a per-call LDS image, head as a trailing launch
        osf::FusedArgs fa;
        fa.kf = a; fa.kf.k = ctx->k;
        fa.fcw = nullptr; fa.fcb = nullptr; fa.C = d.num_classes; fa.use_sigmoid = d.use_sigmoid; fa.out = out;
        float *seq0 = nullptr, *seq1 = nullptr, *hlast = nullptr;
        if (os_gru_scratch(ctx, B, T, &seq0, &seq1, &hlast)) return -10;
        fa.seq_out = d.num_layers > 1 ? seq0 : nullptr;
        fa.h_last = d.num_layers > 1 ? nullptr : hlast;
        fa.nrm = nullptr;
        const bool qd = ctx->q_is_diagonal, so = d.num_layers > 1;
        dim3 grid((B + tile - 1) / tile), block(256);
        if (tile >= 128) {
            if (!ctx->fused2_attr_set) {
                const void *fns[8] = {(const void *)osf::fused_kf_gru_kernel_v2<true, false, 2>, (const void *)osf::fused_kf_gru_kernel_v2<false, false, 2>,
                                      (const void *)osf::fused_kf_gru_kernel_v2<true, true, 2>, (const void *)osf::fused_kf_gru_kernel_v2<false, true, 2>,
                                      (const void *)osf::fused_kf_gru_kernel_v2<true, false, 1>, (const void *)osf::fused_kf_gru_kernel_v2<false, false, 1>,
                                      (const void *)osf::fused_kf_gru_kernel_v2<true, true, 1>, (const void *)osf::fused_kf_gru_kernel_v2<false, true, 1>};
                for (const void *fn : fns)
                    OS_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)osf::LDS2_BYTES));
                ctx->fused2_attr_set = true;
            }
            if (!ctx->fused_img) OS_HIP(ctx, hipMalloc((void **)&ctx->fused_img, osf::LDS2_BYTES));
            hipLaunchKernelGGL(osf::fused_pack_kernel, dim3(32), dim3(256), 0, s, ctx->gru_flat, minmax, ctx->fused_img);
            fa.wpacked = ctx->fused_img;
            const int slot = os_prof_begin(ctx, OS_PHASE_FUSED, s, tile == 256 ? "fused_kf_gru_kernel_v2" : "fused_kf_gru_kernel_v2<32 per wave>");
#define OSF_LAUNCH2(QD, SO, NRB) hipLaunchKernelGGL((osf::fused_kf_gru_kernel_v2<QD, SO, NRB>), grid, block, osf::LDS2_BYTES, s, fa)
            if (tile == 256) {
                if (qd && !so) OSF_LAUNCH2(true, false, 2); else if (qd) OSF_LAUNCH2(true, true, 2);
                else if (!so) OSF_LAUNCH2(false, false, 2); else OSF_LAUNCH2(false, true, 2);
            } else {
                if (qd && !so) OSF_LAUNCH2(true, false, 1); else if (qd) OSF_LAUNCH2(true, true, 1);
                else if (!so) OSF_LAUNCH2(false, false, 1); else OSF_LAUNCH2(false, true, 1);
            }
#undef OSF_LAUNCH2
            os_prof_end(ctx, slot, s);
        } else {
            const int nsplit = 64 / tile;          // 1, 2, 4 waves per 16-trajectory tile
            if (!ctx->fused3_attr_set) {
                OS_HIP(ctx, hipFuncSetAttribute((const void *)osf::fused_kf_gru_kernel_v3<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)osf::lds3_bytes<1>()));
                OS_HIP(ctx, hipFuncSetAttribute((const void *)osf::fused_kf_gru_kernel_v3<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)osf::lds3_bytes<1>()));
                OS_HIP(ctx, hipFuncSetAttribute((const void *)osf::fused_kf_gru_kernel_v3<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)osf::lds3_bytes<2>()));
                OS_HIP(ctx, hipFuncSetAttribute((const void *)osf::fused_kf_gru_kernel_v3<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)osf::lds3_bytes<2>()));
                OS_HIP(ctx, hipFuncSetAttribute((const void *)osf::fused_kf_gru_kernel_v3<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)osf::lds3_bytes<4>()));
                OS_HIP(ctx, hipFuncSetAttribute((const void *)osf::fused_kf_gru_kernel_v3<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)osf::lds3_bytes<4>()));
                ctx->fused3_attr_set = true;
            }
            if (!ctx->fused_img3) OS_HIP(ctx, hipMalloc((void **)&ctx->fused_img3, osf::LDS3_IMG_BYTES));
            hipLaunchKernelGGL(osf::fused_pack3_kernel, dim3(32), dim3(256), 0, s, ctx->gru_flat, minmax, ctx->fused_img3, nsplit);
            fa.wpacked = ctx->fused_img3;
            const int slot = os_prof_begin(ctx, OS_PHASE_FUSED, s, nsplit == 1 ? "fused_kf_gru_kernel_v3<1>" : nsplit == 2 ? "fused_kf_gru_kernel_v3<2>" : "fused_kf_gru_kernel_v3<4>");
#define OSF_LAUNCH3(QD, NSP) hipLaunchKernelGGL((osf::fused_kf_gru_kernel_v3<QD, NSP>), grid, block, osf::lds3_bytes<NSP>(), s, fa)
            if (nsplit == 1) { if (qd) OSF_LAUNCH3(true, 1); else OSF_LAUNCH3(false, 1); }
            else if (nsplit == 2) { if (qd) OSF_LAUNCH3(true, 2); else OSF_LAUNCH3(false, 2); }
            else { if (qd) OSF_LAUNCH3(true, 4); else OSF_LAUNCH3(false, 4); }
#undef OSF_LAUNCH3
            os_prof_end(ctx, slot, s);
        }
        OS_HIP(ctx, hipGetLastError());
        if (d.num_layers > 1) return os_gru_layers_impl(ctx, B, T, seq0, 1, out, nullptr, s);
        const float *fcw = ctx->gru_flat + (os_gru_param_count(&d) - ((size_t)d.num_classes * 64 + d.num_classes));
        return os_gru_head_launch(ctx, B, hlast, fcw, out, s);
    }
    if (n_latent > 0 && (flags & OS_FUSED_LATENT_IN_PLACE)) {
        // the caller's buffer IS the GRU input [T][I][B] (rows 60.. = latent): features written in place, nothing copied
        float *buf = const_cast<float *>(latent);
        a.feat_out = buf;
        const int rc = os_kf_run_impl(ctx, a, flags & ~(OS_FUSED_TWO_KERNEL | OS_FUSED_LATENT_IN_PLACE), s);
        if (rc) return rc;
        return os_gru_layers_impl(ctx, B, T, buf, 0, out, nullptr, s);
    }
    // general path: Kalman kernel emits normalised feature rows [T][I][B] into context scratch, GRU kernels consume them
    const size_t need = (size_t)T * I * B;
    if (ctx->feat_floats < need) {
        if (ctx->feat) OS_HIP(ctx, hipFree(ctx->feat));
        ctx->feat = nullptr; ctx->feat_floats = 0;
        OS_HIP(ctx, hipMalloc((void **)&ctx->feat, need * sizeof(float)));
        ctx->feat_floats = need;
    }
    a.feat_out = ctx->feat;
    int rc = os_kf_run_impl(ctx, a, flags & ~OS_FUSED_TWO_KERNEL, s);
    if (rc) return rc;
    if (n_latent > 0) {
        // rows [60, 60+NL) of every step: latent [T][NL][B] -> feat [T][I][B] (gru/gru_test.py:135-136)
        OS_HIP(ctx, hipMemcpy2DAsync(ctx->feat + (size_t)60 * B, (size_t)I * B * sizeof(float), latent,
                                     (size_t)n_latent * B * sizeof(float), (size_t)n_latent * B * sizeof(float), T,
                                     hipMemcpyDeviceToDevice, s));
    }
    return os_gru_layers_impl(ctx, B, T, ctx->feat, 0, out, nullptr, s);
}

}  // extern "C"
